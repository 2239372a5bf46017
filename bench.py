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
FP64_MFMA_PEAK_TFLOPS = 78.6                # MI355X public spec, FP64 matrix (= FP64 vector); the local guide lists no f64 row


def algorithmic_bytes_per_pair():
    """Per frame pair and kernel: bytes the algorithm must move.  Used for roofline.achieved."""
    p = {"front": P_FRONT, "bird": P_BIRD}
    resize = sum((p[k] - PX7[k]) + (p[k] - PX0[k]) for k in p)       # read level l-1, write level l
    fast = sum(p.values())                                            # every level read once
    blur = 2 * sum(p.values())                                        # every level read once, blurred level written once
    describe = 2 * sum(p.values())                                    # raw level (orientation) + blurred level (BRIEF), once each
    match_front = 32 * 2000 + 32 * 2000 + 16 * 2000 + 8 * 2000        # 184,000 B per 2000x2000 problem
    match_bird = 32 * 2000 + 32 * 1000 + 16 * 2000 + 8 * 1000
    pose = 2000 * 24 + 1000 * 28                                      # edge inputs read once (LDS staged)
    return {"k_resize": resize, "k_fast": fast, "k_blur": blur, "k_describe": describe, "k_proj_frame": match_front,
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


def local_ba_leg(L, rank, world_size, local_rank, reps=3):
    """Secondary measurement (not part of `value`): one LocalBundleAdjustmentWithOdom of BASELINE config 4
    (20 keyframes x 8k map points + 2k bird points).  N=1: fb_local_ba; N>1: the same problem landmark-sharded
    over all ranks (fb_local_ba_sharded, RCCL all-reduce of the reduced normal equations)."""
    import ctypes as C
    import torch
    from fishbirdeyevisualslam_amd import ba_problem, synth, dist as fbd
    p = synth.make_ba_problem(4000, n_kf=20, n_mp=8000, n_mpb=2000)
    cb = fbd.make_allreduce(stage_device=torch.device("cuda", local_rank)) if world_size > 1 else None
    times = []
    for _ in range(reps):
        a, out, keep = ba_problem.local_ba_args(p, with_odom=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = L.fb_local_ba_sharded(C.byref(a), rank, world_size, cb, None) if world_size > 1 else L.fb_local_ba(C.byref(a))
        dt = time.perf_counter() - t0
        if rc != 0:
            raise RuntimeError(L.fb_last_error().decode())
        times.append(dt)
    res = {"ms_per_ba": sorted(times)[len(times) // 2] * 1e3, "workload": "configs[3]: 20 keyframes x 8000 map points + 2000 bird "
           "points, %d front + %d bird + %d odometry edges" % (len(p["obs_kf"]), len(p["bobs_kf"]), len(p["odom_kf_i"])),
           "mode": "sharded over %d ranks (landmark partition, all-reduce of S,b,chi2)" % world_size if world_size > 1 else "1 GPU",
           "includes": "host<->device copies and the host-driven LM loop"}
    # MFMA utilisation of the Schur kernel (north_star: "MFMA utilisation for J^T J"): one more BA under the event
    # profiler.  k_ba_schur multiplies dense LDS panels with v_mfma_f64_16x16x4_f64, upper-triangular 16x16 tiles only.
    from fishbirdeyevisualslam_amd import cabi
    a, out, keep = ba_problem.local_ba_args(p, with_odom=1)
    L.fb_prof_reset()
    L.fb_prof_enable(1)
    rc = L.fb_local_ba_sharded(C.byref(a), rank, world_size, cb, None) if world_size > 1 else L.fb_local_ba(C.byref(a))
    L.fb_prof_enable(0)
    ents = (cabi.ProfEntry * 40)()
    n = L.fb_prof_report(ents, 40)
    kern = {ents[i].name.decode(): (ents[i].launches, ents[i].total_ms) for i in range(n)}
    if rc == 0 and "k_ba_schur" in kern and kern["k_ba_schur"][1] > 0:
        n_free = int((np.asarray(keep["kf_fixed"] if "kf_fixed" in keep else p["kf_fixed"]) == 0).sum())
        nt = (6 * n_free + 1 + 15) // 16                       # 16x16 tiles per side of the reduced system (+ rhs column)
        n_lm = (len(p["mp_xw"]) + len(p["mpb_xw"]) + world_size - 1) // world_size  # landmarks of this rank
        launches, ms = kern["k_ba_schur"]
        flop = 2.0 * (nt * (nt + 1) // 2) * 256 * 3 * n_lm      # MFMA flops issued per launch (upper tiles, K = 3 per landmark)
        tf = flop * launches / (ms * 1e-3) / 1e12
        res["mfma"] = {"kernel": "k_ba_schur", "dtype": "f64", "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": tf / FP64_MFMA_PEAK_TFLOPS, "avg_launch_ms": ms / launches, "launches_per_ba": launches,
                       "flop_per_launch": flop, "free_keyframes": n_free, "tiles": nt,
                       "kernels_ms_per_ba": {k: v[1] for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])}}
    if rank == 0 and world_size == 1:
        from oracle import pyoracle as O
        a, out, keep = ba_problem.local_ba_args(p, with_odom=1)
        t0 = time.perf_counter()
        O.call("orc_local_ba", a)
        res["cpu_oracle_ms"] = (time.perf_counter() - t0) * 1e3
    return res


def match_pair_rate(pipe, world, kern_serial, B, nprob=4, th=15.0):
    fk, _ = pipe.keypoints_host("front")
    sf = np.array(list(pipe.tables.scale_factor)[: pipe.params.nlevels], np.float32)
    pairs = []
    for b in range(min(nprob, B)):
        w, k = world[b], fk[b]
        T = np.asarray(w["Tcw0"], np.float64).reshape(3, 4)
        X = w["last_xw"].astype(np.float64) @ T[:, :3].T + T[:, 3]
        ok = X[:, 2] > 0
        u = pipe.fx * X[:, 0] / X[:, 2] + pipe.cx
        v = pipe.fy * X[:, 1] / X[:, 2] + pipe.cy
        ok &= (u >= 0) & (u < pipe.fw) & (v >= 0) & (v < pipe.fh)
        octv = w["last_octave"]
        r = th * sf[octv]
        kx, ky, ko = k["x"].astype(np.float64), k["y"].astype(np.float64), k["octave"]
        n = 0
        for q in np.nonzero(ok)[0]:
            m = (np.abs(kx - u[q]) < r[q]) & (np.abs(ky - v[q]) < r[q]) & (ko >= octv[q] - 1) & (ko <= octv[q] + 1)
            n += int(m.sum())
        pairs.append(n)
    per_problem = float(np.mean(pairs))
    ms = kern_serial["k_proj_frame"][1] / kern_serial["k_proj_frame"][0] if "k_proj_frame" in kern_serial else None
    return {"kernel": "k_proj_frame", "descriptor_pairs_per_problem": per_problem, "problems_sampled": len(pairs),
            "gpairs_per_s_no_overlap": per_problem * B / (ms * 1e-3) / 1e9 if ms else None,
            "note": "256-bit Hamming distances per second of the front matcher alone (single-stream launch time, per "
                    "fixed-point round once); window th=15"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frame pairs per step and per GPU")
    ap.add_argument("--cpu-sample", type=int, default=96, help="frame pairs timed on the host for cpu_baseline (0 = skip)")
    ap.add_argument("--no-ba", action="store_true", help="skip the secondary local-BA measurement")
    ap.add_argument("--serial", action="store_true", help="single-stream steps (kernels do not overlap; for profiling)")
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

    step = pipe.step_serial if a.serial else pipe.step
    for _ in range(a.warmup):
        step()
    ents = (cabi.ProfEntry * 40)()

    def prof_pass(fn, nsteps):
        """nsteps untimed steps with every kernel bracketed by HIP events -> {kernel: (launches, total ms)}"""
        torch.cuda.synchronize()
        L.fb_prof_only(None)
        L.fb_prof_reset()
        L.fb_prof_enable(1)
        for _ in range(nsteps):
            fn()
        torch.cuda.synchronize()
        L.fb_prof_enable(0)
        n_ = L.fb_prof_report(ents, 40)
        return {ents[i].name.decode(): (ents[i].launches, ents[i].total_ms) for i in range(n_)}

    # probe (untimed): the same steps with every kernel bracketed, to find the dominant kernel.  Two events per launch
    # cost ~8 us of stream bubble each, so in the timed region only the dominant kernel stays bracketed.
    PROBE = 3
    kern_all = prof_pass(step, PROBE)
    dom = max(kern_all, key=lambda k: kern_all[k][1])
    barrier()
    L.fb_prof_only(dom.encode())
    L.fb_prof_reset()
    L.fb_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    L.fb_prof_enable(0)
    L.fb_prof_only(None)
    if world_size > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # device time of the dominant kernel over the timed region (HIP events on its launch stream)
    n = L.fb_prof_report(ents, 40)
    kern = {ents[i].name.decode(): (ents[i].launches, ents[i].total_ms) for i in range(n)}
    res = pipe.results_host()
    # the same kernels without stream overlap (3 single-stream steps, outside the timed region): each kernel's own speed
    kern_serial = prof_pass(pipe.step_serial, 3)
    ba = None
    if not a.no_ba:
        try:
            ba = local_ba_leg(L, rank, world_size, local_rank)
        except Exception as e:  # the secondary leg must never take the headline line down
            ba = {"error": str(e)}

    if rank == 0:
        total_ms = sum(v[1] for v in kern_all.values()) / PROBE * a.steps
        # HBM traffic of the dominant kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE cannot run inside this process; the
        # committed profiles/r01_pmc_traffic.json holds the per-launch means of separate PMC passes over this same
        # workload (profiles/probes/make_summary.py), used only when the batch size matches
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                pt = json.load(f)
            kk = "k_resize_rows" if dom == "k_resize" else dom
            if pt.get("batch") == B and kk in pt["kernels"]:
                traffic = pt["kernels"][kk]["fetch_bytes_per_launch"] + pt["kernels"][kk]["write_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            traffic = None
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
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE, raw, mean per launch)" if traffic else None,
                         "avg_launch_ms": per_launch_ms, "algorithmic_bytes_per_launch": alg_per_launch,
                         "note": "dominant kernel chosen in an untimed probe pass with every kernel bracketed; in the timed region (3 "
                                 "concurrent streams) only this kernel carries events; *_no_overlap = same kernel in single-stream steps",
                         "avg_launch_ms_no_overlap": kern_serial[dom][1] / kern_serial[dom][0],
                         "achieved_no_overlap": alg.get(dom, 0) * B * 3 / kern_serial[dom][0] / (kern_serial[dom][1] / kern_serial[dom][0] * 1e-3) / 1e9},
            "kernels_ms_per_step_no_overlap": {k: v[1] / 3 for k, v in sorted(kern_serial.items(), key=lambda kv: -kv[1][1])},
            "kernels_ms_per_step": {k: v[1] / PROBE for k, v in sorted(kern_all.items(), key=lambda kv: -kv[1][1])},  # probe pass
            # north_star: "HBM GB/s for matching" -- algorithmic bytes / single-stream kernel time, every kernel with a
            # SURVEY 8(d) byte count (fraction of the 8 TB/s roof)
            "hbm_gbs_no_overlap": {k: {"gbs": alg[k] * B * 3 / (kern_serial[k][1] * 1e-3) / 1e9,
                                       "frac": alg[k] * B * 3 / (kern_serial[k][1] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                                   for k in alg if k in kern_serial and kern_serial[k][1] > 0},
            "kernel_ms_sum_over_wall": total_ms / (elapsed * 1e3),  # >1: the three streams overlap
            "workload_check": {"kps_front": float(res["n_front"].mean()), "kps_bird": float(res["n_bird"].mean()),
                               "front_matches": float(res["nm_front"].mean()), "bird_matches": float(res["nm_bird"].mean()),
                               "pose_inliers": float(res["ninliers"].mean())},
        }
        # the matchers' integer work = descriptor pairs actually compared.  Counted on the host for the first
        # problems of the batch with the window rule of SearchByProjection(CurrentFrame, LastFrame) (ORBmatcher.cc:1361-
        # 1411: radius th * scale[octave], levels octave-1 .. octave+1, |dx|,|dy| < r as GetFeaturesInArea, Frame.cc:498-546)
        try:
            out["match"] = match_pair_rate(pipe, world, kern_serial, B)
        except Exception as e:  # a reporting extra must never cost the bench line
            out["match"] = {"error": str(e)[:200]}
        out["local_ba"] = ba
        # the CPU leg is timed on rank 0 of the single-GPU run only (N > 1 would stall the other ranks)
        out["cpu_baseline"] = cpu_baseline(front, bird, world, min(a.cpu_sample, B)) if (a.cpu_sample > 0 and world_size == 1) else None
        print(json.dumps(out), flush=True)
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()
    pipe.close()


if __name__ == "__main__":
    main()
