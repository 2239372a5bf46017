#!/usr/bin/env python3
"""Time the front match stage (k_proj_frame) of the device pipeline alone, for several batch sizes."""
import ctypes as C, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from fishbirdeyevisualslam_amd.pipeline import FramePipeline
for B in (1, 8, 64, 128):
    front, bird = bench.make_images(B, 0)
    p = FramePipeline(B, bench.FRONT_WH, bench.BIRD_WH, device="cuda:0")
    p.set_images(front, bird)
    p.build_world(seed=5000)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p.step_serial(); torch.cuda.synchronize()
    for name, fn in (("match_front", lambda: p.match_front(s)), ("match_bird", lambda: p.match_bird(s)), ("grids", lambda: p.grids(s)), ("pose", lambda: p.pose(s))):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print("B=%d %s %.1f us" % (B, name, e0.elapsed_time(e1) / 20 * 1e3))
