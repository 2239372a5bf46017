"""What does the pose stage cost the overlapped step?  The same pipeline with the pose launch skipped."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import numpy as np, torch
from fishbirdeyevisualslam_amd import synth
from fishbirdeyevisualslam_amd.pipeline import FramePipeline
B = 256
pipe = FramePipeline(B)
f = np.stack([synth.synth_image(1000 + i, 1280, 720) for i in range(B)])
g = np.stack([synth.synth_image(1500 + i, 512, 512) for i in range(B)])
pipe.set_images(f, g); pipe.build_world(seed=5000)
def run(tag, K=20):
    for _ in range(3): pipe.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): pipe.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(tag, "%.3f ms per step, %.1f k pairs/s" % (dt * 1e3, B / dt / 1e3), flush=True)
run("with pose   ")
orig = pipe.pose
pipe.pose = lambda s, S: None
run("without pose")
pipe.pose = orig
run("with pose   ")
