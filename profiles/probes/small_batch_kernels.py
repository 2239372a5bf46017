"""Per-kernel single-stream times at B = 1 and B = 8 (which kernels make up the latency of a small batch)."""
import os, sys, ctypes as C
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import numpy as np, torch
import fishbirdeyevisualslam_amd as fb
from fishbirdeyevisualslam_amd import cabi, synth
from fishbirdeyevisualslam_amd.pipeline import FramePipeline
L = fb.lib()
ents = (cabi.ProfEntry * 48)()
for B in (1, 8):
    pipe = FramePipeline(B)
    f = np.stack([synth.synth_image(1000 + i % 16, 1280, 720) for i in range(B)])
    g = np.stack([synth.synth_image(1500 + i % 16, 512, 512) for i in range(B)])
    pipe.set_images(f, g); pipe.build_world(seed=5000)
    for _ in range(5): pipe.step()
    torch.cuda.synchronize(); L.fb_prof_only(None); L.fb_prof_reset(); L.fb_prof_enable(1)
    n = 20
    with torch.cuda.stream(pipe.sP):
        for _ in range(n): pipe.step_serial()
    torch.cuda.synchronize(); L.fb_prof_enable(0)
    k = L.fb_prof_report(ents, 48)
    d = {ents[i].name.decode(): round(ents[i].total_ms / n, 4) for i in range(k)}
    print("B =", B, dict(sorted(d.items(), key=lambda kv: -kv[1])), "sum", round(sum(d.values()), 3), flush=True)
    pipe.close()
