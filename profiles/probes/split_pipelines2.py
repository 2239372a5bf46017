"""One pipeline of B pairs per step vs P pipelines of B/P pairs with stream sets of their own (GPU_MAX_HW_QUEUES=8)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import numpy as np, torch
from fishbirdeyevisualslam_amd import synth
from fishbirdeyevisualslam_amd.pipeline import FramePipeline
B = 256
for P in (1, 2, 1, 2):
    b = B // P
    pipes = []
    for _ in range(P):
        FramePipeline._streams = {}
        pipes.append(FramePipeline(b))
    for i, p in enumerate(pipes):
        f = np.stack([synth.synth_image(1000 + (i * b + k) % 16, 1280, 720) for k in range(b)])
        g = np.stack([synth.synth_image(1500 + (i * b + k) % 16, 512, 512) for k in range(b)])
        p.set_images(f, g)
        p.build_world(seed=3000 + 100 * i)
    for _ in range(3):
        for p in pipes: p.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        for p in pipes: p.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("P=%d x %d pairs: %.3f ms per %d pairs = %.1f k pairs/s" % (P, b, dt * 1e3, B, B / dt / 1e3), flush=True)
    for p in pipes: p.close()
    del pipes
