"""Many seeds of the pose-optimisation parity check: outlier masks / inlier counts must equal the oracle's exactly, poses
within 1e-4 (the kernel is compiled with fused multiply-adds and sums in a different order than the oracle)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import hip_lib as H, oracle_lib as O
from fishbirdeyevisualslam_amd import cabi, problems as P, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8   # problems per call: 64 or more run k_pose_opt_reg<256>, fewer k_pose_opt_split
bad = 0; worst = 0.0
for mode in (cabi.FB_POSE_FRONT_BIRD, cabi.FB_POSE_FRONT, cabi.FB_POSE_BIRD):
    for s0 in range(0, n, B):
        probs = [synth.make_pose_problem(7000 + s0 + i, n_front=500 + 37 * ((s0 + i) % 40), n_bird=200 + 19 * ((s0 + i) % 40)) for i in range(B)]
        a, oo, k = P.pose_args(probs, mode=mode); O.call("orc_pose_opt", a)
        a2, oh, k2 = P.pose_args(probs, mode=mode); H.call("fb_pose_opt", a2)
        To, Th = oo["Tcw"].reshape(-1, 3, 4), oh["Tcw"].reshape(-1, 3, 4)
        rel = (np.abs(Th - To) / np.maximum(1.0, np.abs(To).max(axis=(1, 2), keepdims=True))).max()
        worst = max(worst, rel)
        same = np.array_equal(oh["ninliers"], oo["ninliers"]) and np.array_equal(oh["front_outlier"], oo["front_outlier"]) and np.array_equal(oh["bird_outlier"], oo["bird_outlier"])
        if not same or rel > 1e-4:
            bad += 1
            print("MISMATCH mode", mode, "seeds", 7000 + s0, "rel", rel, flush=True)
print("pose sweep: %d problems x 3 modes in calls of %d, %d mismatching batches, worst relative pose difference %.3g" % (n, B, bad, worst))
