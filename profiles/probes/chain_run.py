"""Probe: B sequences of the tracking chain for N frames (serial driver unless PIPE=1); meant to run under rocprofv3."""
import os, sys, time
sys.path.insert(0, ".")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
from fishbirdeyevisualslam_amd import sequence as SQ, track as TR
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
pipe = os.environ.get("PIPE") == "1"
nframes = 11
seq = SQ.Sequence(B, nframes, seed=9000 + B, device="cuda:0")
imgs = [seq.render(k) for k in range(nframes)]
mask = torch.from_numpy(seq.mask.copy()).cuda()
tc = TR.TrackChain(B, (1280, 720), (512, 512), K=seq.Kc, D=seq.D, map_cap=2 * 2064, bird_cap=8 * 2064, bird_nfeatures=int(os.environ.get('BIRD', '1000')))
tc.extract(*imgs[nframes - 1], mask); v_end = tc.view("cur")
tc.extract(*imgs[0], mask); v0 = tc.view("cur")
M, MB, mp0, mpb0, Tcw0 = seq.build_map(v0, tc.tables, map_cap=tc.map_cap, bird_cap=tc.bird_cap, extra_views=[(nframes - 1, v_end)])
tc.set_map(M, MB); tc.init_first(mp0, mpb0, Tcw0)
path = bench._triangle_path(nframes, N + 30)
dl = {(a, b): torch.from_numpy(seq.delta_between(a, b)).cuda() for a, b in set(zip(path[:-1], path[1:]))}
pos = 0
def run(n):
    global pos
    for _ in range(n):
        a, b = path[pos], path[pos + 1]
        if pipe:
            tc.prefetch(*imgs[path[pos + 2]], mask); tc.track_prefetched(dl[(a, b)], sync=True)
        else:
            tc.delta.copy_(dl[(a, b)], non_blocking=True); tc.track(*imgs[b], mask); tc.counts()
        pos += 1
    torch.cuda.synchronize()
if pipe:
    tc.prefetch(*imgs[path[1]], mask)
run(10)
t0 = time.perf_counter(); run(N); dt = time.perf_counter() - t0
print("B=%d pipe=%s: %.3f ms per step" % (B, pipe, dt / N * 1e3))

if hasattr(tc.L, "fb_match_debug_m2"):
    import ctypes as C
    d = (C.c_int * 4)()
    tc.L.fb_match_debug_m2(d)
    print("M2 (sequence 0): %d launches, %.1f rounds per launch, %.0f points in view per launch" % (d[2], d[0] / max(d[2], 1), d[1] / max(d[2], 1)))
