"""k_fast per-phase wave time (shader clock, summed over waves) -- run with FB_FAST_DBG=20."""
import sys, os, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
import fishbirdeyevisualslam_amd as fb
from fishbirdeyevisualslam_amd import synth
from fishbirdeyevisualslam_amd.pipeline import FramePipeline
assert os.environ.get("FB_FAST_DBG") == "20"
B = 64
f = np.stack([synth.synth_image(1000 + i, 1280, 720) for i in range(8)] * 8)
b = np.stack([synth.synth_image(1500 + i, 512, 512) for i in range(8)] * 8)
pipe = FramePipeline(B)
pipe.set_images(f, b)
L = fb.lib()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3): pipe.extract(s)
torch.cuda.synchronize()
names = ["decode+tile wait+stage", "prefetch issue", "sweep", "score ini", "nms ini", "emit ini", "score min", "nms min", "emit min", "group decode"]
for which, orb in (("front", pipe.orb_f), ("bird", pipe.orb_b)):
    t = (C.c_uint64 * 16)()
    fb.check(L.fb_orb_debug_timers(orb, t), "timers")   # reset
    pipe.extract(s); torch.cuda.synchronize()
    fb.check(L.fb_orb_debug_timers(orb, t), "timers")
    n = max(t[11], 1)
    print(which, "timed waves", t[11], "cycles/wave", round(t[10] / n))
    for i in (9, 0, 1, 2, 3, 4, 5, 6, 7, 8):
        print("   %-24s %8.0f cycles/wave  %5.1f %%" % (names[i], t[i] / n, 100.0 * t[i] / max(t[10], 1)))
from fishbirdeyevisualslam_amd import cabi
L.fb_prof_reset(); L.fb_prof_enable(1)
for _ in range(5): pipe.extract(s)
torch.cuda.synchronize()
ents = (cabi.ProfEntry * 32)(); n = L.fb_prof_report(ents, 32)
print("kernel ms (timed instantiation):", {ents[i].name.decode(): round(ents[i].total_ms / 5, 3) for i in range(n) if "fast" in ents[i].name.decode()})
