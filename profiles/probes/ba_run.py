"""Runs LocalBundleAdjustmentWithOdom of BASELINE config 4 a few times (for rocprofv3 --kernel-trace --stats) and prints ms per BA."""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import fishbirdeyevisualslam_amd as fb
from fishbirdeyevisualslam_amd import ba_problem, synth
L = fb.lib()
p = synth.make_ba_problem(4000, n_kf=20, n_mp=8000, n_mpb=2000)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ts = []
for i in range(n):
    a, out, keep = ba_problem.local_ba_args(p, with_odom=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fb.check(L.fb_local_ba(C.byref(a)), "ba")
    ts.append((time.perf_counter() - t0) * 1e3)
print("ms per BA:", [round(t, 3) for t in ts], "median", round(sorted(ts)[len(ts) // 2], 3))
