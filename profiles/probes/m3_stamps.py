"""Phase times (shader clock, workgroup 0) of k_proj_frame inside the tracking chain at batch 1.  Needs the diagnostic build:
    FB_BUILD_DEFS=-DFB_MATCH_STAMPS python -m fishbirdeyevisualslam_amd.build --force && python profiles/probes/m3_stamps.py"""
import ctypes as C, sys
sys.path.insert(0, ".")
import numpy as np, torch
from fishbirdeyevisualslam_amd import sequence as SQ, track as TR
B = 1
seq = SQ.Sequence(B, 7, seed=9001, device="cuda:0")
imgs = [seq.render(k) for k in range(7)]
mask = torch.from_numpy(seq.mask).cuda()
tc = TR.TrackChain(B, (1280, 720), (512, 512), K=seq.Kc, D=seq.D, map_cap=2 * 2064, bird_cap=8 * 2064, bird_nfeatures=1000)
tc.extract(*imgs[0], mask)
v0 = tc.view("cur")
M, MB, mp0, mpb0, Tcw0 = seq.build_map(v0, tc.tables, map_cap=tc.map_cap, bird_cap=tc.bird_cap)
tc.set_map(M, MB)
tc.init_first(mp0, mpb0, Tcw0)
t = (C.c_uint64 * 8)()
for k in range(1, 7):
    tc.set_delta(seq.delta(k))
    tc.track(*imgs[k], mask)
    c, _ = tc.counts()
    if k == 2:
        tc.L.fb_match_debug_m3(t)   # reset after the warm-up frames
tc.L.fb_match_debug_m3(t)
n = max(t[6], 1)
names = ["staging (descriptors, positions, grid -> LDS)", "octave order", "round 0 (grid walk + distances)", "later rounds (from the cache)", "commit + rotation histogram"]
tot = sum(t[i] for i in range(5))
print("%d launches, %.1f rounds per launch, %d cycles per launch" % (t[6], t[5] / n, tot / n))
for i in range(5):
    print("   %-48s %8.0f cycles  %5.1f %%" % (names[i], t[i] / n, 100.0 * t[i] / max(tot, 1)))
