"""Per-kernel table (event brackets) of one frame tracked through TrackReferenceKeyFrame + TrackLocalMap on the device frames.
usage: chain_ref_kernels.py [BATCH] [reference|motion]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from fishbirdeyevisualslam_amd import sequence as SQ, track as TR, cabi
from fishbirdeyevisualslam_amd.bow_problem import make_vocabulary

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
MODE = sys.argv[2] if len(sys.argv) > 2 else "reference"
dev = "cuda:0"
seq = SQ.Sequence(B, 5, seed=9000 + B, device=dev)
imgs = [seq.render(k) for k in range(5)]
mask = torch.from_numpy(seq.mask).to(dev)
tc = TR.TrackChain(B, (1280, 720), (512, 512), K=seq.Kc, D=seq.D, map_cap=2 * 2064, bird_cap=8 * 2064, device=dev, bird_nfeatures=1000)
tc.extract(*imgs[0], mask)
v0 = tc.view("cur")
M, MB, mp0, mpb0, Tcw0 = seq.build_map(v0, tc.tables, map_cap=tc.map_cap, bird_cap=tc.bird_cap)
tc.set_vocabulary(make_vocabulary(9900, k=10, L=6)[1], 6)
tc.set_map(M, MB)
tc.init_first(mp0, mpb0, Tcw0)
tc.make_keyframe("last")
L = tc.L
ents = (cabi.ProfEntry * 48)()
prev = 0
for rep in range(3):
    for k in (1, 2, 1, 0):
        if rep == 2 and k == 1 and prev == 0:
            torch.cuda.synchronize(); L.fb_prof_only(None); L.fb_prof_reset(); L.fb_prof_enable(1)
        tc.set_delta(seq.delta_between(prev, k)); tc.set_delta_kf(seq.delta_between(0, k))
        tc.track_modes(*imgs[k], mask, mode=MODE)
        c, _ = tc.counts()
        prev = k
L.fb_prof_enable(0)
n = L.fb_prof_report(ents, 48)
tot = sum(ents[i].total_ms for i in range(n))
print("B=%d: 4 frames through %s + TrackLocalMap, kernel sum %.3f ms per frame; BoW matches %s" % (B, "TrackReferenceKeyFrame" if MODE == "reference" else "TrackWithMotionModel", tot / 4, c[cabi.FB_CNT["BOW_MATCHES"]][:4].tolist()))
for i in sorted(range(n), key=lambda i: -ents[i].total_ms):
    print("   %-28s %3d launches  %8.1f us per frame" % (ents[i].name.decode(), ents[i].launches, ents[i].total_ms / 4 * 1e3))
