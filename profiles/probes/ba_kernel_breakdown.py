#!/usr/bin/env python3
"""Per-kernel device time of one fb_local_ba (event profiler) for a given number of key frames."""
import ctypes as C, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import fishbirdeyevisualslam_amd as fb
from fishbirdeyevisualslam_amd import ba_problem, cabi, synth
L = fb.lib()
for n_kf in [int(x) for x in sys.argv[1:]] or [20, 40]:
    p = synth.make_ba_problem(4200, n_kf=n_kf, n_fixed=2, n_mp=8000, n_mpb=2000)
    a, o, k = ba_problem.local_ba_args(p, with_odom=1)
    fb.check(L.fb_local_ba(C.byref(a)), "warm")
    a, o, k = ba_problem.local_ba_args(p, with_odom=1)
    L.fb_prof_reset(); L.fb_prof_enable(1)
    fb.check(L.fb_local_ba(C.byref(a)), "ba")
    L.fb_prof_enable(0)
    ents = (cabi.ProfEntry * 40)()
    n = L.fb_prof_report(ents, 40)
    print("n_kf", n_kf, {ents[i].name.decode(): (ents[i].launches, round(ents[i].total_ms, 3)) for i in range(n)})
