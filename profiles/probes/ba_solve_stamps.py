"""Phase times (shader clock, thread 0) of the look-ahead LDL^T of the local BA.  Needs the diagnostic build:
    FB_BUILD_DEFS=-DFB_BA_STAMPS python -m fishbirdeyevisualslam_amd.build --force && python profiles/probes/ba_solve_stamps.py"""
import ctypes as C, sys
sys.path.insert(0, ".")
import fishbirdeyevisualslam_amd as fb
from fishbirdeyevisualslam_amd import ba_problem, synth
L = fb.lib()
p = synth.make_ba_problem(4000, n_kf=20, n_mp=8000, n_mpb=2000)
for rep in range(3):
    a, out, keep = ba_problem.local_ba_args(p, with_odom=1)
    fb.check(L.fb_local_ba(C.byref(a)), "ba")
    t = (C.c_uint64 * 16)()
    fb.check(L.fb_ba_debug_stamps(t), "stamps")
names = ["load + assemble", "first diagonal factor", "read factor + panel", "barrier after panel", "wave 0: next block update + factor",
         "wait for the trailing update", "backward substitution"]
n = max(t[15], 1)
tot = sum(t[i] for i in range(7))
print("%d solves, %d cycles per solve" % (t[15], tot / n))
for i in range(7):
    print("   %-38s %8.0f cycles per solve  %5.1f %%" % (names[i], t[i] / n, 100.0 * t[i] / tot))
