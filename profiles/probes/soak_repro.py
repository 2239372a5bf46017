import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import test_track_chain_gpu as T
for pipe in (True, False, True):
    try:
        w, st = T._run(2, 7, (640, 480), (384, 384), 250.0, seed=9391575, use_lists=True, granular=False, contour=False, pipelined=pipe, bird_nfeatures=600, check_workload=False)
        print("pipelined", pipe, "ok", w)
    except AssertionError as e:
        print("pipelined", pipe, "FAIL", str(e)[:600])
