"""Probe: the tracking chain at small batch, eager launches vs hipGraph replay (one graph per frame-handle rotation: Frame
construction from fixed input buffers + fb_frame_track_dev), per-frame counter read-back in both.  usage: chain_graph.py B N"""
import os, sys, time
sys.path.insert(0, ".")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import ctypes as C
import numpy as np, torch
from fishbirdeyevisualslam_amd import sequence as SQ, track as TR, check
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 120
nframes = 11
seq = SQ.Sequence(B, nframes, seed=9000 + B, device="cuda:0")
imgs = [seq.render(k) for k in range(nframes)]
mask = torch.from_numpy(seq.mask.copy()).cuda()

def make():
    tc = TR.TrackChain(B, (1280, 720), (512, 512), K=seq.Kc, D=seq.D, map_cap=2 * 2064, bird_cap=8 * 2064, bird_nfeatures=1000)
    tc.extract(*imgs[nframes - 1], mask); v_end = tc.view("cur")
    tc.extract(*imgs[0], mask); v0 = tc.view("cur")
    M, MB, mp0, mpb0, Tcw0 = seq.build_map(v0, tc.tables, map_cap=tc.map_cap, bird_cap=tc.bird_cap, extra_views=[(nframes - 1, v_end)])
    tc.set_map(M, MB); tc.init_first(mp0, mpb0, Tcw0)
    return tc
path = bench._triangle_path(nframes, 2 * N + 40)
dl = {(a, b): torch.from_numpy(seq.delta_between(a, b)).cuda() for a, b in set(zip(path[:-1], path[1:]))}

# ---- eager
tc = make()
pos = 0
def eager(n):
    global pos
    for _ in range(n):
        a, b = path[pos], path[pos + 1]
        tc.delta.copy_(dl[(a, b)], non_blocking=True); tc.track(*imgs[b], mask); tc.counts(); pos += 1
    torch.cuda.synchronize()
eager(12)
t0 = time.perf_counter(); eager(N); t_eager = (time.perf_counter() - t0) / N
ref_counts, ref_T = tc.counts()
tc.close()

# ---- graphs: fixed input buffers, one graph per rotation of the three handles
tc = make()
fin = [torch.empty_like(x) for x in imgs[0]]
s = torch.cuda.Stream()
graphs = {}
pos = 0
def step_graph():
    global pos
    a, b = path[pos], path[pos + 1]
    with torch.cuda.stream(s):
        for dst, src in zip(fin, imgs[b]):
            dst.copy_(src, non_blocking=True)
        tc.delta.copy_(dl[(a, b)], non_blocking=True)
        h = tc.k % 3
        if h not in graphs:
            # warm the path once eagerly on this stream (workspaces, attributes), then capture the same calls for this rotation
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                tc.extract(*fin, mask)
                check(tc.L.fb_frame_track_dev(tc.cur, tc.last, C.byref(tc.targs), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "track")
            graphs[h] = g
        graphs[h].replay()
        tc.k += 1
        tc.counts("last")
    pos += 1
with torch.cuda.stream(s):
    for _ in range(3):   # eager warm-up of every rotation before any capture
        a, b = path[pos], path[pos + 1]
        tc.delta.copy_(dl[(a, b)], non_blocking=True); tc.track(*imgs[b], mask); tc.counts(); pos += 1
torch.cuda.synchronize()
for _ in range(9):
    step_graph()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    step_graph()
torch.cuda.synchronize()
t_graph = (time.perf_counter() - t0) / N
c2, T2 = tc.counts()
print("B=%d: eager %.3f ms per frame, graph replay %.3f ms per frame; same final counters: %s, pose difference %.3g" %
      (B, t_eager * 1e3, t_graph * 1e3, bool(np.array_equal(c2[:12], ref_counts[:12])), float(np.abs(T2 - ref_T).max())))
