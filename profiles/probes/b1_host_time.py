"""Is the B = 1 pipeline bound by the host?  Time of enqueueing `pipe.step()` (no synchronisation) vs the synchronised step."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from fishbirdeyevisualslam_amd import synth
from fishbirdeyevisualslam_amd.pipeline import FramePipeline
for B in (1, 8):
    f = np.stack([synth.synth_image(1000 + i, 1280, 720) for i in range(B)])
    b = np.stack([synth.synth_image(1500 + i, 512, 512) for i in range(B)])
    pipe = FramePipeline(B)
    pipe.set_images(f, b)
    pipe.build_world(seed=5000)
    for _ in range(20):
        pipe.step()
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        pipe.step()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("B=%d: enqueue %.3f ms per step, with the GPU drained %.3f ms per step" % (B, t_enq / n * 1e3, t_all / n * 1e3))
    pipe.close()
