"""Does k_pose_opt_reg (768 B of scratch per lane) slow down on a queue whose first kernels needed little scratch?"""
import os, sys, ctypes as C
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import numpy as np, torch
import fishbirdeyevisualslam_amd as fb
from fishbirdeyevisualslam_amd import cabi, synth
from fishbirdeyevisualslam_amd.pipeline import FramePipeline
B = 256
pipe = FramePipeline(B)
f = np.stack([synth.synth_image(1000 + i % 16, 1280, 720) for i in range(B)])
g = np.stack([synth.synth_image(1500 + i % 16, 512, 512) for i in range(B)])
pipe.set_images(f, g)
pipe.build_world(seed=5000)   # runs a serial step internally? (setup)
L = fb.lib()
ents = (cabi.ProfEntry * 48)()
def prof(fn, n, tag):
    torch.cuda.synchronize(); L.fb_prof_only(None); L.fb_prof_reset(); L.fb_prof_enable(1)
    for _ in range(n): fn()
    torch.cuda.synchronize(); L.fb_prof_enable(0)
    k = L.fb_prof_report(ents, 48)
    d = {ents[i].name.decode(): round(ents[i].total_ms / max(ents[i].launches, 1), 3) for i in range(k)}
    print(tag, "k_pose_opt per launch", d.get("k_pose_opt"), "k_fast<44>", d.get("k_fast<44>"), flush=True)
fresh = torch.cuda.Stream()
def on(stream):
    def run():
        with torch.cuda.stream(stream): pipe.step_serial()
    return run
prof(on(fresh), 3, "fresh stream, extractor kernels first:")
prof(on(fresh), 3, "same stream again:")
prof(pipe.step_serial, 3, "default stream:")
for _ in range(3): pipe.step()
prof(on(pipe.sP), 3, "pose stream after 3-stream steps:")
prof(pipe.step_serial, 3, "default stream again:")
