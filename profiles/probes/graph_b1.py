"""B=1 latency: eager launches vs one hipGraph replay of the whole per-frame chain (torch.cuda.CUDAGraph capture of step_serial)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from fishbirdeyevisualslam_amd import synth
from fishbirdeyevisualslam_amd.pipeline import FramePipeline
for B in (1, 4):
    pipe = FramePipeline(B)
    f = np.stack([synth.synth_image(1000 + k, 1280, 720) for k in range(B)])
    g = np.stack([synth.synth_image(1500 + k, 512, 512) for k in range(B)])
    pipe.set_images(f, g)
    pipe.build_world(seed=3000)
    for _ in range(5): pipe.step_serial()
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K): pipe.step_serial()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / K
    ref = pipe.results_host()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        pipe.step_serial()          # warm on the capture stream
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            pipe.step_serial()
    torch.cuda.synchronize()
    for _ in range(5): gr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): gr.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / K
    res = pipe.results_host()
    same = all(np.array_equal(res[k], ref[k]) for k in ("n_front", "nm_front", "nm_bird", "ninliers")) and np.array_equal(res["Tcw"], ref["Tcw"])
    print("B=%d eager %.3f ms/step  graph %.3f ms/step  identical results: %s" % (B, eager * 1e3, graph * 1e3, same), flush=True)
    pipe.close()
