"""Phase times of one LM evaluation of k_pose_opt (shader clock of thread 0, workgroup 0).  Needs the diagnostic build:
    FB_BUILD_DEFS=-DFB_POSE_STAMPS python -m fishbirdeyevisualslam_amd.build --force && python profiles/probes/pose_stamps.py
"""
import ctypes as C, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import fishbirdeyevisualslam_amd as fb
from fishbirdeyevisualslam_amd import cabi, problems as P, synth
L = fb.lib()
names = ["front edges", "bird edges", "wave reduce", "cross-wave reduce + 2 barriers", "ldlt6 (lane 0)", "exp + mul (lane 0)", "barrier after solve",
         "classification pass"]
for B in (1, 8):
    probs = [synth.make_pose_problem(3000 + i) for i in range(B)]
    for rep in range(3):
        a, out, keep = P.pose_args(probs, mode=cabi.FB_POSE_FRONT_BIRD)
        fb.check(L.fb_pose_opt(C.byref(a)), "pose")
        t = (C.c_uint64 * 16)()
        fb.check(L.fb_pose_debug_stamps(t), "stamps")
    nev = max(t[15], 1)
    print("B=%d: %d LM evaluations, kernel %d cycles (%.1f us at 100 MHz s_memtime? see below)" % (B, t[15], t[14], t[14] / 100.0))
    tot = sum(t[i] for i in range(8))
    for i in range(8):
        per = nev if i < 7 else 4
        print("   %-32s %9d cycles total  %7.0f per %s  %5.1f %%" % (names[i], t[i], t[i] / per, "evaluation" if i < 7 else "round", 100.0 * t[i] / max(t[14], 1)))
    print("   accounted %.1f %% of the kernel" % (100.0 * tot / max(t[14], 1)))
