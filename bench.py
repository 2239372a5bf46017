#!/usr/bin/env python3
"""bench.py -- frames/s of the per-frame hot path (extract + match + pose-opt) on MI355X.

A "step" is one pass of the hot path over one batch of B synthetic frame pairs (front 1280x720 +
bird 512x512, BASELINE.json configs[2]: HIP ORB extract + Hamming match + PoseOptimizationWithBird
with ~2k front + ~1k bird edges).  All inputs are resident in HBM before the timed region.
    python bench.py --gpus N --steps K --warmup W
For N>1 the driver launches one rank per GPU with torch.distributed.run; every rank processes its
own batch (independent sequences -> weak scaling, no data-path collective).
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRONT_WH, BIRD_WH = (1280, 720), (512, 512)
# algorithmic bytes per image (pyramid-streaming model)
P_FRONT, P_BIRD = 2853088, 811960          # sum of pyramid pixels
PX0 = {"front": 1280 * 720, "bird": 512 * 512}
PX7 = {"front": 357 * 201, "bird": 143 * 143}
HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: 8 TB/s HBM3E (spec)


def algorithmic_bytes_per_pair():
    """Per frame pair and kernel: bytes the algorithm must move.  Used for roofline.achieved."""
    p = {"front": P_FRONT, "bird": P_BIRD}
    resize = sum((p[k] - PX7[k]) + (p[k] - PX0[k]) for k in p)       # read level l-1, write level l
    fast = sum(p.values())                                            # every level read once
    describe = sum(p.values())                                        # every level read once for blur+BRIEF
    match_front = 32 * 2000 + 32 * 2000 + 16 * 2000 + 8 * 2000        # 184,000 B per 2000x2000 problem
    match_bird = 32 * 2000 + 32 * 1000 + 16 * 2000 + 8 * 1000
    pose = 2000 * 24 + 1000 * 28                                      # edge inputs read once (LDS staged)
    return {"k_resize": resize, "k_fast": fast, "k_describe": describe, "k_proj_frame": match_front,
            "k_bird_mappoints": match_bird, "k_pose_opt": pose}


def make_images(batch, rank):
    from fishbirdeyevisualslam_amd import synth
    f = np.stack([synth.synth_image(1000 + rank * 10000 + i, *FRONT_WH) for i in range(batch)])
    b = np.stack([synth.synth_image(1500 + rank * 10000 + i, *BIRD_WH) for i in range(batch)])
    return f, b


def cpu_baseline(front, bird, world, nsample):
    """Oracle (CPU restatement of the reference path, single thread like the reference) on a bounded sample."""
    from oracle import pyoracle as O
    params = O.orb_params()
    O.frame_pipeline(params, front[0], bird[0], world[0])  # warm-up (page in, build .so)
    t0 = time.perf_counter()
    for i in range(nsample):
        O.frame_pipeline(params, front[i], bird[i], world[i])
    dt = time.perf_counter() - t0
    return {"value": nsample / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d frame pairs of the same workload (extract front+bird, grids, M3, M9, PoseOptimizationWithBird), "
                      "oracle/ C++ -O3 -march=native, 1 thread, %.1f s" % (nsample, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="frame pairs per step and per GPU")
    ap.add_argument("--cpu-sample", type=int, default=24, help="frame pairs timed on the host for cpu_baseline (0 = skip)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import fishbirdeyevisualslam_amd as fb
    from fishbirdeyevisualslam_amd import cabi
    from fishbirdeyevisualslam_amd.pipeline import FramePipeline
    L = fb.lib()
    fb.check(L.fb_set_device(local_rank), "fb_set_device")

    B = a.batch
    front, bird = make_images(B, rank)
    pipe = FramePipeline(B, FRONT_WH, BIRD_WH, device="cuda:%d" % local_rank)
    pipe.set_images(front, bird)
    world = pipe.build_world(seed=5000 + rank * 10000)

    def barrier():
        torch.cuda.synchronize()
        if world_size > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        pipe.step()
    barrier()
    L.fb_prof_reset()
    L.fb_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        pipe.step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    L.fb_prof_enable(0)
    if world_size > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel device time of the timed region (HIP events on the launch stream)
    ents = (cabi.ProfEntry * 32)()
    n = L.fb_prof_report(ents, 32)
    kern = {ents[i].name.decode(): (ents[i].launches, ents[i].total_ms) for i in range(n)}
    res = pipe.results_host()

    if rank == 0:
        total_ms = sum(v[1] for v in kern.values())
        dom = max(kern, key=lambda k: kern[k][1])
        alg = algorithmic_bytes_per_pair()
        launches, ms = kern[dom]
        per_launch_ms = ms / launches
        alg_per_launch = alg.get(dom, 0) * B * a.steps / launches   # bytes one launch must move, averaged over its launches
        achieved = alg_per_launch / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
        out = {
            "metric": "frames/s (extract+match+pose-opt), 1280x720+512x512 pair",
            "value": world_size * B * a.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world_size,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8 (extract, Hamming) + f64 (pose optimisation)",
            "data": "synthetic",
            "config": {"workload": "configs[2]: 1280x720 front + 512x512 bird pair, HIP ORB extract + M3/M9 Hamming match + "
                                   "PoseOptimizationWithBird (~2k front + ~1k bird edges)",
                       "frame_pairs_per_step_per_gpu": B, "nfeatures": 2000, "parallelism": "replicas x%d" % world_size},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "avg_launch_ms": per_launch_ms, "algorithmic_bytes_per_launch": alg_per_launch},
            "kernels_ms_per_step": {k: v[1] / a.steps for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])},
            "device_busy_frac": total_ms / (elapsed * 1e3),
            "workload_check": {"kps_front": float(res["n_front"].mean()), "kps_bird": float(res["n_bird"].mean()),
                               "front_matches": float(res["nm_front"].mean()), "bird_matches": float(res["nm_bird"].mean()),
                               "pose_inliers": float(res["ninliers"].mean())},
        }
        if a.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(front, bird, world, min(a.cpu_sample, B))
        print(json.dumps(out), flush=True)
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()
    pipe.close()


if __name__ == "__main__":
    main()
