import sys, os, ctypes as C, time
sys.path.insert(0, '.')
import numpy as np, torch
import fishbirdeyevisualslam_amd as fb
from fishbirdeyevisualslam_amd import cabi, synth
from fishbirdeyevisualslam_amd.pipeline import FramePipeline
B = 64
f = np.stack([synth.synth_image(1000 + i, 1280, 720) for i in range(8)] * 8)
b = np.stack([synth.synth_image(1500 + i, 512, 512) for i in range(8)] * 8)
pipe = FramePipeline(B)
pipe.set_images(f, b)
L = fb.lib()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3): pipe.extract(s)
torch.cuda.synchronize()
L.fb_prof_reset(); L.fb_prof_enable(1)
for _ in range(10): pipe.extract(s)
torch.cuda.synchronize()
ents = (cabi.ProfEntry * 32)(); n = L.fb_prof_report(ents, 32)
print(os.environ.get('FB_FAST_DBG'), {ents[i].name.decode(): round(ents[i].total_ms / 10, 3) for i in range(n)})
