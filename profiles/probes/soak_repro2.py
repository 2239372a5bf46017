import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from fishbirdeyevisualslam_amd import sequence as S, track as T
import test_track_chain_gpu as TT
seed, B, K = 9391575, 2, 4
wh, bwh, fx = (640, 480), (384, 384), 250.0
seq = S.Sequence(B, K, seed=seed, front_wh=wh, bird_wh=bwh, fx=fx, fy=fx, device="cuda:0")
frames = [seq.render(k) for k in range(K)]
def run(mode):
    tc = T.TrackChain(B, wh, bwh, K=seq.Kc, D=seq.D, use_lists=True, bird_nfeatures=600)
    f, b, c = frames[0]
    tc.extract(f, b, None, None)
    v0 = tc.view("cur")
    M, MB, mp0, mpb0, Tcw0 = seq.build_map(v0, tc.tables, map_cap=tc.map_cap, bird_cap=tc.bird_cap)
    lm, lb = TT._lists(M, MB, seed + 5)
    tc.set_map(M, MB, lm, lb)
    tc.init_first(mp0, mpb0, Tcw0)
    out = []
    if mode != "serial":
        tc.prefetch(frames[1][0], frames[1][1], None, None)
    for k in range(1, K):
        d = torch.from_numpy(seq.delta(k)).cuda()
        if mode == "serial":
            tc.set_delta(seq.delta(k)); tc.track(frames[k][0], frames[k][1], None, None)
        else:
            if k + 1 < K:
                tc.prefetch(frames[k + 1][0], frames[k + 1][1], None, None)
            if mode == "pipe_sync":
                torch.cuda.synchronize()
            tc.track_prefetched(d)
        torch.cuda.synchronize()
        out.append(tc.view())
    tc.close()
    return out
ref = run("serial")
for mode in ("pipe", "pipe_sync", "pipe"):
    o = run(mode)
    for k, (a, r) in enumerate(zip(o, ref)):
        msg = []
        for key in ("n", "n_bird"):
            if not np.array_equal(a[key], r[key]): msg.append("%s %s vs %s" % (key, a[key], r[key]))
        for b in range(B):
            nb = min(int(a["n_bird"][b]), int(r["n_bird"][b]))
            if not np.array_equal(a["kps_bird"][b, :nb], r["kps_bird"][b, :nb]):
                ka, kr = a["kps_bird"][b, :int(a["n_bird"][b])], r["kps_bird"][b, :int(r["n_bird"][b])]
                msg.append("seq %d bird kps differ; per-octave counts %s vs %s" % (b, np.bincount(ka["octave"], minlength=8).tolist(), np.bincount(kr["octave"], minlength=8).tolist()))
            n = min(int(a["n"][b]), int(r["n"][b]))
            if not np.array_equal(a["kps"][b, :n], r["kps"][b, :n]): msg.append("seq %d front kps differ" % b)
        print(mode, "frame", k + 1, "OK" if not msg else msg)
