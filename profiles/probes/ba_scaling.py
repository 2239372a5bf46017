#!/usr/bin/env python3
"""fb_local_ba wall time vs number of key frames (LDS-resident path up to 23 free key frames, HBM path beyond)."""
import ctypes as C, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import fishbirdeyevisualslam_amd as fb
import oracle_lib as O
from fishbirdeyevisualslam_amd import ba_problem, synth
L = fb.lib()
for n_kf, n_mp, n_mpb in ((20, 8000, 2000), (25, 8000, 2000), (40, 8000, 2000), (70, 12000, 3000), (120, 20000, 4000)):
    p = synth.make_ba_problem(4200, n_kf=n_kf, n_fixed=2, n_mp=n_mp, n_mpb=n_mpb)
    ts = []
    for _ in range(3):
        a, o, k = ba_problem.local_ba_args(p, with_odom=1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fb.check(L.fb_local_ba(C.byref(a)), "ba"); ts.append(time.perf_counter() - t0)
    a, o, k = ba_problem.local_ba_args(p, with_odom=1)
    t0 = time.perf_counter(); O.call("orc_local_ba", a); tc = time.perf_counter() - t0
    print("n_kf=%d edges=%d  hip %.2f ms  oracle %.1f ms" % (n_kf, len(p["obs_kf"]) + len(p["bobs_kf"]), sorted(ts)[1] * 1e3, tc * 1e3))
