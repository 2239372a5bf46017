#!/usr/bin/env python3
"""bench.py -- frames/s of the per-frame hot path (extract + match + pose-opt) on MI355X.

A "step" is one pass of the hot path over one batch of B synthetic frame pairs (front 1280x720 +
bird 512x512, BASELINE.json configs[2]: HIP ORB extract + Hamming match + PoseOptimizationWithBird
with ~2k front + ~1k bird edges).  All inputs are resident in HBM before the timed region.
    python bench.py --gpus N --steps K --warmup W
For N>1 the driver launches one rank per GPU with torch.distributed.run; every rank processes its
own batch (independent sequences -> weak scaling, no data-path collective).
Prints ONE JSON line on rank 0.

What the line carries besides the contract fields:
  roofline      the dominant kernel of the SINGLE-STREAM pass (every kernel bracketed by HIP events on its launch stream,
                no overlap: the same ranking `rocprofv3 --kernel-trace --stats` gives), its SURVEY 8(d) algorithmic bytes
                per launch over its average launch duration inside the timed region; + the whole step against the roof
  parity_check  the oracle results of the cpu_baseline sample compared with the same entries of the timed GPU batch
                (key points, descriptors, match indices, outlier masks, inlier counts bit-exact, pose within 1e-4);
                a mismatch makes the process exit non-zero
  cpu_baseline  the oracle on 1 core (per-stage medians over >= 50 frames after 5 warm-ups) and as
                N independent sequences on N cores
  single_sequence  ms per frame pair at B = 1 and B = 8 (config 5's sequence count) with the same pipeline
"""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The pipeline overlaps three HIP streams per tracked batch (front chain, bird chain, pose optimisation).  The HIP runtime
# folds streams onto 4 hardware queues by default, and two streams that land on one queue serialise: with the streams of
# the single-sequence leg created after the BA leg's, B = 1 measured 0.73 ms instead of 0.39 ms per pair.  Must be set
# before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

FRONT_WH, BIRD_WH = (1280, 720), (512, 512)
# algorithmic bytes per image (pyramid-streaming model)
P_FRONT, P_BIRD = 2853088, 811960          # sum of pyramid pixels
PX0 = {"front": 1280 * 720, "bird": 512 * 512}
PX7 = {"front": 357 * 201, "bird": 143 * 143}
PAIR_BYTES = 13384242                       # extract, per frame pair: (P - px7) + (P - px0) + 2 P, front + bird
HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: 8 TB/s HBM3E (spec)
FP64_MFMA_PEAK_TFLOPS = 78.6                # MI355X public spec, FP64 matrix (= FP64 vector); the local guide lists no f64 row
REL_TOL = 1e-4                              # BASELINE.json north_star: pose within 1e-4 relative


def algorithmic_bytes_per_pair():
    """Per frame pair and kernel (profiler name): bytes the algorithm must move, SURVEY 8(d): every level >= 1 is produced by
    reading level l-1 and writing level l; every level is read once for FAST and once for blur + descriptor TOGETHER (the
    blurred image is not re-materialised in the model, although k_blur does write it: that traffic counts against us)."""
    p = {"front": P_FRONT, "bird": P_BIRD}
    return {"k_resize": sum((p[k] - PX7[k]) + (p[k] - PX0[k]) for k in p),
            "k_fast<44>": P_FRONT, "k_fast<56>": P_BIRD,
            "k_blur+k_describe": P_FRONT + P_BIRD,
            "k_proj_frame": 32 * 2000 + 32 * 2000 + 16 * 2000 + 8 * 2000,       # 184,000 B per 2000x2000 problem
            "k_bird_mappoints": 32 * 2000 + 32 * 1000 + 16 * 2000 + 8 * 1000,
            "k_pose_opt": 2000 * 24 + 1000 * 28}                                # edge inputs, read once


def make_images(batch, rank):
    from fishbirdeyevisualslam_amd import synth
    f = np.stack([synth.synth_image(1000 + rank * 10000 + i, *FRONT_WH) for i in range(batch)])
    b = np.stack([synth.synth_image(1500 + rank * 10000 + i, *BIRD_WH) for i in range(batch)])
    return f, b


STAGES = ("extract_front", "extract_bird", "match_front", "match_bird", "pose_opt")


def _cpu_worker(args):
    """One sequence on one core: `nframes` frame pairs through the oracle (5 warm-ups first).  Runs in its own process for
    the N-core leg; the images are synthesised in the worker from their seeds (identical bytes as the GPU's)."""
    seeds, worlds, warm = args[:3]
    keep = len(args) > 3 and args[3]
    from fishbirdeyevisualslam_amd import synth
    from oracle import pyoracle as O
    params = O.orb_params()
    imgs = [(synth.synth_image(1000 + s, *FRONT_WH), synth.synth_image(1500 + s, *BIRD_WH)) for s in seeds]
    for i in range(warm):
        O.frame_pipeline(params, imgs[i % len(imgs)][0], imgs[i % len(imgs)][1], worlds[i % len(imgs)])
    per_frame, stages, results = [], {k: [] for k in STAGES}, []
    t_all = time.perf_counter()
    for (f, b), w in zip(imgs, worlds):
        tm = {}
        t0 = time.perf_counter()
        r = O.frame_pipeline(params, f, b, w, timings=tm)
        per_frame.append(time.perf_counter() - t0)
        for k in STAGES:
            stages[k].append(tm[k])
        if keep:
            results.append(r)
    return time.perf_counter() - t_all, per_frame, stages, results


def parity_check(refs, fk, fd, bk, bd, res):
    """The oracle results the cpu_baseline leg computed for the first frame pairs against the same entries of the GPU batch."""
    bad, worst = [], 0.0
    nsample = len(refs)
    for b, ref in enumerate(refs):
        n, nb = len(ref["fk"]), len(ref["bk"])
        checks = {
            "front keypoints": len(fk[b]) == n and np.array_equal(fk[b], ref["fk"]),
            "front descriptors": len(fd[b]) == n and np.array_equal(fd[b], ref["fd"]),
            "bird keypoints": len(bk[b]) == nb and np.array_equal(bk[b], ref["bk"]),
            "bird descriptors": len(bd[b]) == nb and np.array_equal(bd[b], ref["bd"]),
            "front match indices": np.array_equal(res["m_front"][b, :n], ref["m_front"][:n]),
            "bird match indices": np.array_equal(res["m_bird"][b, :nb], ref["m_bird"][:nb]),
            "front outlier mask": np.array_equal(res["front_outlier"][b, :n][ref["fv"] == 1], ref["front_outlier"][:n][ref["fv"] == 1]),
            "bird outlier mask": np.array_equal(res["bird_outlier"][b, :nb][ref["bv"] == 1], ref["bird_outlier"][:nb][ref["bv"] == 1]),
            "inlier count": int(res["ninliers"][b]) == ref["ninliers"],
        }
        rel = float(np.abs(res["Tcw"][b] - ref["Tcw"]).max() / max(1.0, np.abs(ref["Tcw"]).max()))
        worst = max(worst, rel)
        checks["pose within 1e-4"] = rel <= REL_TOL
        for k, ok in checks.items():
            if not ok:
                bad.append("pair %d: %s" % (b, k))
    return {"pairs": nsample, "mismatches": len(bad), "details": bad[:8], "worst_relative_pose_difference": worst,
            "checked": "key points + descriptors (front, bird), M3 / M9 match indices, front / bird outlier masks, inlier count: "
                       "bit-exact; pose <= 1e-4 relative; GPU values are those of the timed batch"}


def cpu_baseline(world, nsample, rank, refs_out):
    """The oracle (CPU restatement of the reference path, compiled -O3 -march=native, single-threaded per frame like the
    reference) on the host cores of this box: (a) one sequence on one core, per-stage medians; (b) N sequences on N cores."""
    seeds = [rank * 10000 + i for i in range(nsample)]
    t1, per_frame, stages, results = _cpu_worker((seeds, world[:nsample], 5, True))
    refs_out.extend(results)
    one = {"frames": nsample, "warmup_frames": 5, "seconds": t1, "frames_per_s_mean": nsample / t1,
           "ms_per_frame_median": statistics.median(per_frame) * 1e3, "ms_per_frame_mean": statistics.mean(per_frame) * 1e3,
           "stage_ms_median": {k: statistics.median(v) * 1e3 for k, v in stages.items()},
           "stage_ms_mean": {k: statistics.mean(v) * 1e3 for k, v in stages.items()}}
    # N sequences on N cores.  Two lines: N = this box's cores per GPU (usable cores / GPUs on the box: what a
    # host has per MI355X), and N = 16 = the CPU share the one-GPU bench box grants a run (labelled as such)
    import glob
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # (render nodes: one per GPU, or one per partition when the GPUs are partitioned -- the bench box lists 64 for its 8 MI355X)
    nodes = len(glob.glob("/sys/class/drm/renderD*"))
    gpus_on_box = nodes if 1 <= nodes <= 8 else 8
    per = max(10, min(nsample, 16))

    def many_cores(ncore, label):
        jobs = [([rank * 10000 + (w * 7 + i) % nsample for i in range(per)], [world[(w * 7 + i) % nsample] for i in range(per)], 2)
                for w in range(ncore)]
        t0 = time.perf_counter()
        with mp.get_context("spawn").Pool(ncore) as pool:
            outs = pool.map(_cpu_worker, jobs)
        wall = time.perf_counter() - t0
        busy = max(o[0] for o in outs)  # slowest worker, without process start-up and image synthesis
        return {"cores": ncore, "label": label, "frames_per_sequence": per, "frames_per_s": ncore * per / busy, "wall_s_incl_startup": wall}
    share = max(1, avail // gpus_on_box)
    many = {"cores_available": avail, "os_cpu_count": os.cpu_count(), "gpus_on_box": gpus_on_box,
            "per_gpu_share": many_cores(share, "usable cores / GPUs on the box = %d / %d" % (avail, gpus_on_box)),
            "note": "N independent sequences, one process per core, timed from each worker's first to last timed frame (slowest worker)"}
    if share != 16 and avail >= 16:
        many["bench_box_cpu_share"] = many_cores(16, "16 = the CPU share of the one-GPU bench box")
    many["cores"], many["frames_per_s"] = many["per_gpu_share"]["cores"], many["per_gpu_share"]["frames_per_s"]
    return {"value": one["frames_per_s_mean"], "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d frame pairs of the same workload (extract front+bird, grids, M3, M9, PoseOptimizationWithBird) after 5 warm-ups, "
                      "oracle/ C++ -O3 -march=native, 1 thread, %.1f s; stage times = inside the oracle's C++ only" % (nsample, t1),
            "one_core": one, "n_cores": many}


def local_mapping_leg(L, local_rank, reps=5):
    """What the LocalMapping thread asks of the GPU per new key frame (LocalMapping.cc:76-96, 231-476, 478-560), from
    device-resident inputs, enqueued back to back on one stream: SearchForTriangulation of the key frame against 20
    neighbours (M7, one batched call), the search half of Fuse into 20 neighbours and back (two batched calls),
    LocalBundleAdjustmentWithOdom of BASELINE config 4 (fb_local_ba_dev, which ends with the one synchronisation).
    The three problems are synthetic and independent of each other (the map mutation between them stays on the host)."""
    import torch
    from fishbirdeyevisualslam_amd import ba_problem, bow_problem as BP, kf_problems as KP, synth, cabi
    dev = torch.device("cuda", local_rank)
    held = []

    def put(struct, keep):
        made = {}
        for k, v in keep.items():
            if isinstance(v, np.ndarray):
                t = torch.from_numpy(np.ascontiguousarray(v.view(np.uint8) if v.dtype == cabi.KP_DTYPE else v)).to(dev)
                held.append(t)
                made[k] = t
                cabi.fill(struct, **{k: t})
        return made
    NB = 20
    tp = [BP.make_triangulation_problem(6400 + i, 2000, 2000) for i in range(NB)]
    ta, tout, (tk, k1, k2) = BP.triangulation_args(tp)
    put(ta, tk)
    t_out = put(ta, tout)
    fvn = ("n_nodes", "node_ids", "node_start", "items")
    put(ta.fv1, dict(zip(fvn, k1))); put(ta.fv2, dict(zip(fvn, k2)))
    fuse, f_out = [], []
    for d in range(2):
        fp = [KP.make_kf_points_problem(6500 + 100 * d + i, 2000, 2000) for i in range(NB)]
        ks = max(len(q["kf_kps"]) for q in fp)
        cs = np.zeros((NB, 64 * 48 + 1), np.int32); ci = np.zeros((NB, ks), np.int32)
        fa, fo, (kk, mk, fk) = KP.fuse_args(fp, cs, ci, th=3.0)
        put(fa.kf, kk); put(fa.mp, mk); put(fa, fk)
        f_out.append(put(fa, fo)["best_idx"])
        fuse.append(fa)
    s = torch.cuda.current_stream(dev)
    sp = C.c_void_p(s.cuda_stream)
    for fa in fuse:   # the neighbours' grids, built by the product's own kernel (KeyFrame::AssignFeaturesToGrid)
        rc = L.fb_grid_build_batch_dev(C.c_void_p(fa.kf.kf_kps), C.c_void_p(fa.kf.n_kf), NB, fa.kf.kf_stride, C.byref(fa.kf.grid),
                                       C.c_void_p(fa.kf.kf_cell_start), C.c_void_p(fa.kf.kf_cell_items), sp)
        if rc != 0:
            raise RuntimeError(L.fb_last_error().decode())
    p = synth.make_ba_problem(4000, n_kf=20, n_mp=8000, n_mpb=2000)
    parts = {"search_for_triangulation_x20": [], "fuse_search_2x20": [], "total": []}
    for rep in range(reps + 1):
        ba, dv, keep = ba_problem.local_ba_args_dev(p, device=dev, with_odom=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = L.fb_match_triangulation_dev(C.byref(ta), sp)
        for fa in fuse:
            rc = rc or L.fb_fuse_search_dev(C.byref(fa), sp)
        rc = rc or L.fb_local_ba_dev(C.byref(ba), sp)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        if rc != 0:
            raise RuntimeError(L.fb_last_error().decode())
        if rep > 0:
            parts["total"].append(t3 - t0)
    # the parts on their own (each with its own synchronisation)
    for name, fn in (("search_for_triangulation_x20", lambda: L.fb_match_triangulation_dev(C.byref(ta), sp)),
                     ("fuse_search_2x20", lambda: [L.fb_fuse_search_dev(C.byref(fa), sp) for fa in fuse])):
        for rep in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            parts[name].append(time.perf_counter() - t0)
    med = lambda v: sorted(v)[len(v) // 2] * 1e3 if v else None
    return {"ms_per_key_frame": med(parts["total"]),
            "triangulation_matches_per_neighbour_mean": float(t_out["nmatches"].float().mean().item()),
            "fuse_candidates_per_neighbour_mean": float(sum((t >= 0).sum().item() for t in f_out)) / (2 * NB),
            "parts_ms": {"search_for_triangulation_x20": med(parts["search_for_triangulation_x20"]),
                                                                 "fuse_search_2x20": med(parts["fuse_search_2x20"])},
            "workload": "20 neighbours x (2000 x 2000 key points) for M7, 2 x 20 x (2000 map points into 2000 key points) for Fuse, "
                        "configs[3] BA; device pointers, one stream, one synchronisation at the end",
            "reference": "LocalMapping.cc:292 (SearchForTriangulation), :513,538 (Fuse), :87-96 (LocalBundleAdjustmentWithOdom)"}


def local_ba_leg(L, rank, world_size, local_rank, reps=3):
    """Secondary measurement (not part of `value`): one LocalBundleAdjustmentWithOdom of BASELINE config 4
    (20 keyframes x 8k map points + 2k bird points).  N=1: fb_local_ba; N>1: the same problem landmark-sharded
    over all ranks (fb_local_ba_sharded, all-reduce of the reduced normal equations)."""
    import torch
    from fishbirdeyevisualslam_amd import ba_problem, synth, dist as fbd, cabi
    p = synth.make_ba_problem(4000, n_kf=20, n_mp=8000, n_mpb=2000)
    cb, comm, transport = None, None, "1 GPU"
    if world_size > 1:
        import torch.distributed as dist
        gloo = dist.get_backend() == "gloo"   # rehearsal on one GPU: RCCL cannot put two ranks on one device
        ok, err = 1.0, ""
        if not gloo:
            try:  # RCCL inside the library
                comm = fbd.RcclComm(L, rank, world_size, device=torch.device("cuda", local_rank))
            except Exception as e:
                comm, ok, err = None, 0.0, str(e)[:120]
        # every rank takes the same transport: the host-staged callback if ANY rank has no communicator
        flag = torch.tensor([ok if not gloo else 0.0], dtype=torch.float64, device="cpu" if gloo else torch.device("cuda", local_rank))
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag.item()) > 0.0:
            transport = fbd.TRANSPORT_RCCL + "; multi-rank RCCL had never run before this line was produced (round 3: no box with > 1 GPU) -- unverified until a run prints rccl_ranks_seen = the world size"
        else:
            if comm is not None:
                comm.close()
                comm = None
            cb = fbd.make_allreduce(stage_device=None if gloo else torch.device("cuda", local_rank))
            transport = fbd.TRANSPORT_HOST + (" (gloo rehearsal)" if gloo else " (no RCCL communicator: %s)" % err)

    def run(a):
        if world_size == 1:
            return L.fb_local_ba(C.byref(a))
        if comm is not None:
            return L.fb_local_ba_sharded_rccl(C.byref(a), rank, world_size, comm.comm)
        return L.fb_local_ba_sharded(C.byref(a), rank, world_size, cb, None)
    times = []
    for _ in range(reps):
        a, out, keep = ba_problem.local_ba_args(p, with_odom=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = run(a)
        dt = time.perf_counter() - t0
        if rc != 0:
            raise RuntimeError(L.fb_last_error().decode())
        times.append(dt)
    seen = None
    if comm is not None:
        try:  # what RCCL itself says: ncclCommCount / ncclCommUserRank of the library's communicator, gathered over the ranks
            import torch.distributed as dist
            n_, r_ = comm.info()
            t_ = torch.tensor([n_, r_], dtype=torch.int64, device=torch.device("cuda", local_rank))
            lst = [torch.zeros_like(t_) for _ in range(world_size)]
            dist.all_gather(lst, t_)
            seen = {"comm_count_per_rank": [int(x[0]) for x in lst], "comm_user_rank_per_rank": [int(x[1]) for x in lst]}
        except Exception as e:  # noqa: BLE001
            seen = {"error": str(e)[:120]}
    res = {"rccl_ranks_seen": seen, "ms_per_ba": sorted(times)[len(times) // 2] * 1e3, "workload": "configs[3]: 20 keyframes x 8000 map points + 2000 bird "
           "points, %d front + %d bird + %d odometry edges" % (len(p["obs_kf"]), len(p["bobs_kf"]), len(p["odom_kf_i"])),
           "mode": ("sharded over %d ranks (landmark partition, all-reduce of S,b,chi2; " % world_size + transport + ")") if world_size > 1 else "1 GPU",
           "includes": "host<->device copies of the graph and the results"}
    # MFMA utilisation of the Schur kernel (north_star: "MFMA utilisation for J^T J"): one more BA under the event profiler
    a, out, keep = ba_problem.local_ba_args(p, with_odom=1)
    L.fb_prof_reset()
    L.fb_prof_enable(1)
    rc = run(a)
    L.fb_prof_enable(0)
    ents = (cabi.ProfEntry * 48)()
    n = L.fb_prof_report(ents, 48)
    kern = {ents[i].name.decode(): (ents[i].launches, ents[i].total_ms) for i in range(n)}
    if rc == 0 and "k_ba_schur" in kern and kern["k_ba_schur"][1] > 0:
        n_free = int((np.asarray(p["kf_fixed"]) == 0).sum())
        nt = (6 * n_free + 1 + 15) // 16                       # 16x16 tiles per side of the reduced system (+ rhs column)
        n_lm = (len(p["mp_xw"]) + len(p["mpb_xw"]) + world_size - 1) // world_size  # landmarks of this rank
        launches, ms = kern["k_ba_schur"]
        issued = 2.0 * (nt * (nt + 1) // 2) * 256 * 3 * n_lm    # MFMA flops issued per launch (upper tiles, K = 3 per landmark)
        # USEFUL flops: landmark with k observers contributes (6k+1 choose 2 upper) x 3 multiply-adds (SURVEY 8d's Schur term)
        deg = np.bincount(np.asarray(p["obs_mp"]), minlength=len(p["mp_xw"]))
        degb = np.bincount(np.asarray(p["bobs_mpb"]), minlength=len(p["mpb_xw"]))
        rows = np.concatenate([6 * deg + 1, 6 * degb + 1]).astype(np.float64)
        useful = float((2.0 * 3.0 * rows * (rows + 1) / 2).sum()) / world_size
        tf_issued = issued * launches / (ms * 1e-3) / 1e12
        tf_useful = useful * launches / (ms * 1e-3) / 1e12
        res["mfma"] = {"kernel": "k_ba_schur", "dtype": "f64", "achieved": tf_useful, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": tf_useful / FP64_MFMA_PEAK_TFLOPS, "achieved_issued": tf_issued, "frac_issued": tf_issued / FP64_MFMA_PEAK_TFLOPS,
                       "avg_launch_ms": ms / launches, "launches_per_ba": launches, "useful_flop_per_launch": useful,
                       "issued_flop_per_launch": issued, "free_keyframes": n_free, "tiles": nt,
                       "note": "achieved counts only the block products a landmark's observer set requires (useful flops); "
                               "achieved_issued counts the zero padding of the dense panels as well",
                       "kernels_ms_per_ba": {k: v[1] for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])}}
    if rank == 0 and world_size == 1:
        from oracle import pyoracle as O
        a, out, keep = ba_problem.local_ba_args(p, with_odom=1)
        t0 = time.perf_counter()
        O.call("orc_local_ba", a)
        res["cpu_oracle_ms"] = (time.perf_counter() - t0) * 1e3
    if comm is not None:
        comm.close()
    return res


def match_pair_rate(pipe, world, kern_serial, nsteps, B, nprob=4, th=15.0):
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
            "gpairs_per_s_single_stream": per_problem * B / (ms * 1e-3) / 1e9 if ms else None,
            "note": "256-bit Hamming distances per second of the front matcher alone (single-stream launch time); window th=15"}


def host_fed_leg(pipe, front, bird, B, steps=12):
    """The configs[2] step fed from HOST memory: the next batch of images travels from page-locked host buffers into a second set
    of device buffers on a copy stream while the current batch is processed (value itself is HBM resident, as the contract says)."""
    import torch
    dev = pipe.dev
    pin_f, pin_b = torch.from_numpy(np.ascontiguousarray(front)).pin_memory(), torch.from_numpy(np.ascontiguousarray(bird)).pin_memory()
    bufs = [(torch.empty_like(pipe.f_img), torch.empty_like(pipe.b_img)) for _ in range(2)]
    keep = (pipe.f_img, pipe.b_img)
    sC = torch.cuda.Stream(device=dev)
    ev = [torch.cuda.Event() for _ in range(2)]
    cur = torch.cuda.current_stream(dev)

    def upload(k):
        if k >= 2:   # the step that read this buffer pair last must be done with it
            sC.wait_event(pipe.evF)
            sC.wait_event(pipe.evB)
        with torch.cuda.stream(sC):
            bufs[k & 1][0].copy_(pin_f, non_blocking=True)
            bufs[k & 1][1].copy_(pin_b, non_blocking=True)
            ev[k & 1].record(sC)
    # copy alone (one untimed pass first: the first touch of freshly pinned pages is slow)
    upload(0); upload(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(4):
        upload(k & 1)
    torch.cuda.synchronize()
    t_copy = (time.perf_counter() - t0) / 4
    upload(0)
    for k in range(2):   # warm-up with the double buffering
        upload(k + 1)
        cur.wait_event(ev[k & 1])
        pipe.f_img, pipe.b_img = bufs[k & 1]
        pipe.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(2, 2 + steps):
        upload(k + 1)
        cur.wait_event(ev[k & 1])
        pipe.f_img, pipe.b_img = bufs[k & 1]
        pipe.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pipe.f_img, pipe.b_img = keep
    nbytes = pin_f.numel() + pin_b.numel()
    return {"frames_per_s": B * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps, "bytes_per_step": nbytes,
            "h2d_alone_ms": t_copy * 1e3, "h2d_alone_gbs": nbytes / t_copy / 1e9,
            "note": "images from page-locked host memory, next batch uploaded on a copy stream beside the current step; the same images every step"}


def single_sequence_leg(Pipeline, rank, local_rank, batches=(1, 8), steps=200, warmup=20):
    """ms per frame pair when one GPU tracks 1 / 8 sequences (BASELINE config 5 has 8): same pipeline, small batch."""
    import torch
    out = {}
    for b in batches:
        f, bd = make_images(b, rank)
        pipe = Pipeline(b, FRONT_WH, BIRD_WH, device="cuda:%d" % local_rank)
        pipe.set_images(f, bd)
        pipe.build_world(seed=5000 + rank * 10000)
        for _ in range(warmup):
            pipe.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["b%d" % b] = {"ms_per_step": dt / steps * 1e3, "ms_per_pair": dt / steps / b * 1e3, "frames_per_s": b * steps / dt, "steps": steps}
        pipe.close()
    return out


def _triangle_path(nframes, steps):
    """0,1,..,K-1,K-2,..,0,1,.. : the vehicle drives the rendered stretch forth and back, every step a real motion."""
    period = list(range(nframes)) + list(range(nframes - 2, 0, -1))
    return [period[i % len(period)] for i in range(steps + 1)]


def track_chain_leg(rank, local_rank, batches=(1, 8, 256), steps=(200, 100, 20), warm=(20, 20, 4), parity_pairs=2, parity_frames=5,
                    nframes=11):
    """The sequence-faithful per-frame chain (fb_frame_*, csrc/track.hip): Frame construction (extract x2, fisheye
    undistortion with fisheye.yaml's coefficients, bird guidance on a contour image + detect mask, camera XYZ, grids), then
    TrackWithMotionModel + TrackLocalMap (Tracking.cc:1312-1441): pose prediction from the PREVIOUS frame's optimised pose,
    M9, M3, PoseOptimizationWithBird, outlier discard, M8 + FilterBirdOutlierInFront, SearchLocalPoints (isInFrustum + M2),
    second PoseOptimizationWithBird, clean-up.  Frame k+1 is enqueued after frame k on one stream and reads its results.
      dependent: the host reads the counter block after every frame (fb_frame_counts = one synchronisation per frame), as a
                 Tracking state machine that decides on retries / LOST per frame must
      free_running: no per-frame read-back (counters checked afterwards)"""
    import torch
    from fishbirdeyevisualslam_amd import sequence as SQ, track as TR, cabi
    dev = "cuda:%d" % local_rank
    ground = SQ.make_ground(9000 + rank)
    out = {"workload": "synthetic drive over a textured ground plane (fishbirdeyevisualslam_amd/sequence.py): 1280x720 fisheye front "
                       "(fisheye.yaml k1..k4, 2000 features) + 512x512 bird (1000 features: BASELINE configs[2]'s ~1k bird edges) + contour + mask "
                       "per frame, %d rendered frames driven forth and back, map = key points of the two end frames" % nframes,
           "order": "Frame.cc:262-379, Tracking.cc:1312-1385, 1387-1441, 690-725"}
    for B, nsteps, nwarm in zip(batches, steps, warm):
        seq = SQ.Sequence(B, nframes, seed=9000 + rank * 1000 + B, device=dev, ground=ground)
        imgs = [seq.render(k) for k in range(nframes)]
        mask = torch.from_numpy(seq.mask).to(dev)
        cap0 = 2064
        tc = TR.TrackChain(B, FRONT_WH, BIRD_WH, K=seq.Kc, D=seq.D, map_cap=2 * cap0, bird_cap=8 * cap0, device=dev, bird_nfeatures=1000)
        tc.extract(*imgs[nframes - 1], mask)
        v_end = tc.view("cur")
        tc.extract(*imgs[0], mask)
        v0 = tc.view("cur")
        M, MB, mp0, mpb0, Tcw0 = seq.build_map(v0, tc.tables, map_cap=tc.map_cap, bird_cap=tc.bird_cap, extra_views=[(nframes - 1, v_end)])
        tc.set_map(M, MB)
        tc.init_first(mp0, mpb0, Tcw0)
        path = _triangle_path(nframes, 2 * nwarm + 4 * nsteps + 8)
        dl = {}
        for a_, b_ in set(zip(path[:-1], path[1:])):
            dl[(a_, b_)] = torch.from_numpy(seq.delta_between(a_, b_)).to(dev)
        res = {}
        pos = 0

        def run(n, sync_each, pipelined):
            nonlocal pos
            for _ in range(n):
                a_, b_ = path[pos], path[pos + 1]
                if pipelined:   # frame pos+2 is constructed on the extraction stream while frame pos+1 is tracked
                    tc.prefetch(*imgs[path[pos + 2]], mask)
                    tc.track_prefetched(dl[(a_, b_)], sync=sync_each)
                else:
                    tc.delta.copy_(dl[(a_, b_)], non_blocking=True)
                    tc.track(*imgs[b_], mask)
                    if sync_each:
                        tc.counts()
                pos += 1
            torch.cuda.synchronize()
        run(nwarm, True, False)
        for name, sync_each in (("dependent", True), ("free_running", False)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(nsteps, sync_each, False)
            dt = time.perf_counter() - t0
            res[name] = {"ms_per_step": dt / nsteps * 1e3, "ms_per_frame_pair": dt / nsteps / B * 1e3, "frames_per_s": B * nsteps / dt, "steps": nsteps}
        # pipelined: the NEXT frame's construction (it needs only the images) overlaps the tracking of the current frame;
        # the frame k -> frame k+1 dependency of pose / associations is untouched
        tc.prefetch(*imgs[path[pos + 1]], mask)
        run(min(nwarm, 8), True, True)
        for name, sync_each in (("dependent_pipelined", True), ("free_running_pipelined", False)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(nsteps, sync_each, True)
            dt = time.perf_counter() - t0
            res[name] = {"ms_per_step": dt / nsteps * 1e3, "ms_per_frame_pair": dt / nsteps / B * 1e3, "frames_per_s": B * nsteps / dt, "steps": nsteps}
        tc.track_prefetched(dl[(path[pos], path[pos + 1])], sync=True)   # drain the frame already constructed
        pos += 1
        if B == batches[-1]:   # per-kernel table of the chain (every kernel bracketed by HIP events, serial driver, no overlap)
            L_ = tc.L
            ents_ = (cabi.ProfEntry * 48)()
            torch.cuda.synchronize()
            L_.fb_prof_only(None); L_.fb_prof_reset(); L_.fb_prof_enable(1)
            run(3, False, False)
            L_.fb_prof_enable(0)
            n_ = L_.fb_prof_report(ents_, 48)
            res["kernels_ms_per_step_bracketed"] = {ents_[i].name.decode(): ents_[i].total_ms / 3 for i in sorted(range(n_), key=lambda i: -ents_[i].total_ms)}
            res["kernels_ms_sum_bracketed"] = sum(ents_[i].total_ms for i in range(n_)) / 3
        c, T = tc.counts()
        res["counters_mean_last_frame"] = {k: float(c[i].mean()) for k, i in cabi.FB_CNT.items()}
        tp = np.stack([np.asarray(seq.Tcw_true(path[pos], b))[:3, :4].reshape(12) for b in range(B)])
        res["max_abs_pose_error_vs_truth"] = float(np.abs(T - tp).max())
        res["bird_table_points"] = float(tc.mpb["n"].float().mean().item())
        # TrackReferenceKeyFrame (Tracking.cc:1180-1244) instead of TrackWithMotionModel: the key frame = frame 0 of the drive
        # (copied aside, BoW computed), every frame tracked against it with its own odometry increment, TrackLocalMap behind it,
        # one counter read-back per frame.  Vocabulary: a synthetic 10-ary tree of depth 6, the shape of the stock ORB vocabulary (1.1 M nodes; the reference ships none).
        try:
            from fishbirdeyevisualslam_amd.bow_problem import make_vocabulary
            if "voc" not in out:
                out["voc"] = make_vocabulary(9900, k=10, L=6)[1]
            tc.set_vocabulary(out["voc"], 6)
            tc.set_map(M, MB)
            # re-seat frame 0 as the last frame and as the key frame
            tc.extract(*imgs[0], mask)
            tc.init_first(mp0, mpb0, Tcw0)
            tc.make_keyframe("last")
            dkf = {b_: torch.from_numpy(seq.delta_between(0, b_)).to(dev) for b_ in range(nframes)}
            near = [1, 2, 3, 2, 1, 0]   # frames close to the key frame (BoW matches need common features)
            d01 = {(a_, b_): torch.from_numpy(seq.delta_between(a_, b_)).to(dev) for a_ in range(4) for b_ in range(4)}

            def run_ref(n):
                prev = 0
                for i in range(n):
                    b_ = near[i % len(near)]
                    tc.delta.copy_(d01[(prev, b_)], non_blocking=True)
                    tc.delta_kf.copy_(dkf[b_], non_blocking=True)
                    tc.track_modes(*imgs[b_], mask, mode="reference")
                    tc.counts()
                    prev = b_
            run_ref(len(near))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nref = max(len(near), (nsteps // len(near)) * len(near))
            run_ref(nref)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            cR, _ = tc.counts()
            res["reference_keyframe_dependent"] = {"ms_per_step": dt / nref * 1e3, "ms_per_frame_pair": dt / nref / B * 1e3, "frames_per_s": B * nref / dt,
                                                   "steps": nref, "bow_matches_mean_last_frame": float(cR[cabi.FB_CNT["BOW_MATCHES"]].mean()),
                                                   "matches_inliers_mean_last_frame": float(cR[cabi.FB_CNT["MATCHES_INLIERS"]].mean())}
        except Exception as e:  # the motion-model numbers above stand on their own
            res["reference_keyframe_dependent"] = {"error": repr(e)[:300]}
        out["b%d" % B] = res
        tc.close()
        del imgs, seq, tc
        torch.cuda.empty_cache()
    out.pop("voc", None)
    # parity of the chain inside the bench: a few sequences, consecutive dependent frames, GPU vs oracle chain
    try:
        from oracle import pyoracle as O
        seq = SQ.Sequence(parity_pairs, parity_frames + 1, seed=9700 + rank, device=dev, ground=ground)
        tc = TR.TrackChain(parity_pairs, FRONT_WH, BIRD_WH, K=seq.Kc, D=seq.D, device=dev, bird_nfeatures=1000)
        oc = O.OracleChain(tc.params, tc.map_cap, tc.bird_cap)
        mask = torch.from_numpy(seq.mask).to(dev)
        h = lambda t: t.cpu().numpy()
        f, b, c = seq.render(0)
        tc.extract(f, b, c, mask)
        oc.extract(h(f), h(b), h(c), seq.mask)
        v0 = tc.view("cur")
        M, MB, mp0, mpb0, Tcw0 = seq.build_map(v0, tc.tables, map_cap=tc.map_cap, bird_cap=tc.bird_cap)
        for ch in (tc, oc):
            ch.set_map(M, MB)
            ch.init_first(mp0, mpb0, Tcw0)
        bad, worst, cpu_ms, stages = [], 0.0, [], []
        for k in range(1, parity_frames + 1):
            f, b, c = seq.render(k)
            d = seq.delta(k)
            tc.set_delta(d)
            tc.track(f, b, c, mask)
            t0 = time.perf_counter()
            oc.track(h(f), h(b), h(c), seq.mask, d)
            cpu_ms.append((time.perf_counter() - t0) * 1e3 / parity_pairs)
            stages.append(oc.stage_seconds())
            g, o = tc.view(), oc.view()
            for bb in range(parity_pairs):
                n, nb = int(o["n"][bb]), int(o["n_bird"][bb])
                for key, m in (("kps", n), ("desc", n), ("map_point", n), ("outlier", n), ("kps_bird", nb), ("desc_bird", nb), ("map_point_bird", nb), ("bird_outlier", nb)):
                    if not np.array_equal(g[key][bb, :m], o[key][bb, :m]):
                        bad.append("frame %d sequence %d: %s" % (k, bb, key))
                rel = float(np.abs(g["Tcw"][bb] - o["Tcw"][bb]).max() / max(1.0, np.abs(o["Tcw"][bb]).max()))
                worst = max(worst, rel)
                if rel > REL_TOL:
                    bad.append("frame %d sequence %d: pose %.3g" % (k, bb, rel))
            if not np.array_equal(g["counts"][:16], o["counts"][:16]):
                bad.append("frame %d: counters" % k)
        out["parity_check"] = {"sequences": parity_pairs, "consecutive_frames": parity_frames, "mismatches": len(bad), "details": bad[:8],
                               "worst_relative_pose_difference": worst,
                               "checked": "every frame: key points, descriptors, mvpMapPoints / mvpMapPointsBird indices, outlier masks, all counters "
                                          "bit-exact; pose <= 1e-4 relative; frame k's pose and associations are the inputs of frame k+1"}
        st = np.array(stages)
        out["cpu_oracle_chain"] = {"ms_per_frame_pair_1_core": float(statistics.median(cpu_ms)),
                                   "includes": "Frame construction (extract x2 ...) + the whole chain, oracle/track_oracle.cpp, 1 thread",
                                   "stage_ms_median": {"extract_front": float(np.median(st[:, 0])) * 1e3 / parity_pairs, "extract_bird": float(np.median(st[:, 1])) * 1e3 / parity_pairs,
                                                       "m9": float(np.median(st[:, 2])) * 1e3 / parity_pairs, "m3": float(np.median(st[:, 3])) * 1e3 / parity_pairs,
                                                       "pose1": float(np.median(st[:, 4])) * 1e3 / parity_pairs, "m8_filter": float(np.median(st[:, 5])) * 1e3 / parity_pairs,
                                                       "local_points_m2": float(np.median(st[:, 6])) * 1e3 / parity_pairs, "pose2": float(np.median(st[:, 7])) * 1e3 / parity_pairs}}
        tc.close()
        oc.close()
    except Exception as e:  # noqa: BLE001
        out["parity_check"] = {"error": str(e)[:300]}
    return out


def host_abi_leg(rank):
    """What one frame costs through the HOST-POINTER C-ABI the INTEGRATION.md shims call (fb_orb_extract x2, host grids,
    fb_match_projection_frame, fb_match_bird_mappoints, fb_pose_opt, synchronous, one 1280x720 + 512x512 pair per call) and
    through the device-resident frame handle (fb_frame_extract from host images + fb_frame_track_dev + one fb_frame_counts):
    tests/cpp/host_abi_bench.cpp, a plain g++ program, run as a child process."""
    import subprocess
    import tempfile
    import fishbirdeyevisualslam_amd as fb
    from fishbirdeyevisualslam_amd import synth
    d = tempfile.mkdtemp()
    exe = os.path.join(d, "host_abi_bench")
    libdir = os.path.dirname(fb.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "cpp", "host_abi_bench.cpp"), "-o", exe,
                           "-L", libdir, "-lfishbird_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    synth.synth_image(1000 + rank * 10000, *FRONT_WH).tofile(os.path.join(d, "front.raw"))
    synth.synth_image(1500 + rank * 10000, *BIRD_WH).tofile(os.path.join(d, "bird.raw"))
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    txt = subprocess.check_output([exe, os.path.join(d, "front.raw"), os.path.join(d, "bird.raw"), "40"], env=env, timeout=300).decode()
    r = json.loads(txt.strip().splitlines()[-1])
    r["note"] = ("host_pointer = every call uploads its inputs, runs, synchronises and downloads (the drop-in of INTEGRATION.md 1-3); "
                 "frame_handle = images copied through pinned staging, everything else device resident, one synchronisation per frame; "
                 "the frame-handle chain also runs M8 + filter, SearchLocalPoints / M2 and the second pose optimisation")
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frame pairs per step and per GPU")
    ap.add_argument("--cpu-sample", type=int, default=55, help="frame pairs timed on the host for cpu_baseline and checked against the GPU batch (0 = skip)")
    ap.add_argument("--no-ba", action="store_true", help="skip the secondary local-BA measurement")
    ap.add_argument("--no-single", action="store_true", help="skip the B=1 / B=8 single-sequence sub-results")
    ap.add_argument("--no-chain", action="store_true", help="skip the tracking-chain and host-ABI sub-results")
    ap.add_argument("--serial", action="store_true", help="single-stream steps (kernels do not overlap; for profiling)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # FB_BENCH_REHEARSAL=1: several ranks on ONE GPU with gloo (the multi-rank control flow on a one-GPU box; not a measurement)
    rehearsal = os.environ.get("FB_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
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

    def step_serial_on_pose_stream():
        # Single-stream steps run on the pipeline's pose stream, not on torch's default stream: the pose kernel needs far more
        # scratch than the kernels before it, and on a queue whose scratch was sized by those the runtime can end up throttling
        # its waves (seen on some boxes: 2.93 instead of 0.39 ms for k_pose_opt in the single-stream pass only; the pose
        # stream's queue has the large allocation from the warm-up steps).
        with torch.cuda.stream(pipe.sP):
            pipe.step_serial()

    step = step_serial_on_pose_stream if a.serial else pipe.step
    for _ in range(a.warmup):
        step()
    ents = (cabi.ProfEntry * 48)()

    def prof_pass(fn, nsteps):
        """nsteps untimed steps with every kernel bracketed by HIP events on its launch stream -> {kernel: (launches, total ms)}"""
        torch.cuda.synchronize()
        L.fb_prof_only(None)
        L.fb_prof_reset()
        L.fb_prof_enable(1)
        for _ in range(nsteps):
            fn()
        torch.cuda.synchronize()
        L.fb_prof_enable(0)
        n_ = L.fb_prof_report(ents, 48)
        return {ents[i].name.decode(): (ents[i].launches, ents[i].total_ms) for i in range(n_)}

    # untimed: single-stream steps with every kernel bracketed -> each kernel's own speed; the dominant kernel is the one
    # with the largest total there (with three overlapped streams the event times of a kernel include its neighbours)
    PROBE = 3
    kern_serial = prof_pass(step_serial_on_pose_stream, PROBE)
    # ranking by kernel: the instantiations of one kernel (k_fast<44> for the front levels, k_fast<56> / <72> for other cell
    # sizes) count together; of the winning kernel the instantiation with the largest total is the one bracketed and reported
    def family(name):
        return name.split("<")[0]
    fam_total = {}
    for k_, (n_, ms_) in kern_serial.items():
        fam_total[family(k_)] = fam_total.get(family(k_), 0.0) + ms_
    dom_family = max(fam_total, key=fam_total.get)
    dom = max((k_ for k_ in kern_serial if family(k_) == dom_family), key=lambda k_: kern_serial[k_][1])
    for _ in range(2):
        step()
    barrier()
    # timed region: only the dominant kernel stays bracketed (two events per launch cost ~8 us of stream bubble each)
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
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n = L.fb_prof_report(ents, 48)
    kern = {ents[i].name.decode(): (ents[i].launches, ents[i].total_ms) for i in range(n)}
    res = pipe.results_host()
    # BASELINE configs[4], the replica form SURVEY 8e recommends: 8 sequences per GPU, every GPU its own 8 (weak scaling)
    config5 = None
    try:
        if a.no_single:
            raise RuntimeError("skipped (--no-single)")
        f8, b8 = make_images(8, rank)
        p8 = FramePipeline(8, FRONT_WH, BIRD_WH, device="cuda:%d" % local_rank)
        p8.set_images(f8, b8)
        p8.build_world(seed=5000 + rank * 10000)
        for _ in range(10):
            p8.step()
        barrier()
        t8 = time.perf_counter()
        for _ in range(100):
            p8.step()
        torch.cuda.synchronize()
        barrier()
        e8 = time.perf_counter() - t8
        if world_size > 1:
            tt = torch.tensor([e8], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            e8 = float(tt.item())
        config5 = {"workload": "configs[4] as replicas: 8 sequences per GPU (independent frame pairs, no data-path collective)",
                   "frame_pairs_per_step_per_gpu": 8, "steps": 100, "ms_per_step": e8 / 100 * 1e3, "frames_per_s": world_size * 8 * 100 / e8,
                   "n_gpus": world_size, "note": "the landmark-sharded BA of the same config is local_ba (mode names the transport)"}
        p8.close()
    except Exception as e:  # noqa: BLE001
        config5 = {"error": str(e)[:200]}
    ba = None
    ba_hung = False
    if not a.no_ba:
        # the secondary leg must never take the headline line down: exceptions are caught, and with several ranks (collectives
        # inside the BA) it runs under a deadline -- a rank stuck in an exchange is reported instead of waited for
        box = {}

        def _ba():
            try:
                fb.check(L.fb_set_device(local_rank), "fb_set_device")  # the HIP device is per thread
                torch.cuda.set_device(local_rank)
                box["ba"] = local_ba_leg(L, rank, world_size, local_rank)
                if world_size == 1:
                    try:
                        box["ba"]["local_mapping_chain"] = local_mapping_leg(L, local_rank)
                    except Exception as e:
                        box["ba"]["local_mapping_chain"] = {"error": repr(e)[:300]}
            except Exception as e:
                box["ba"] = {"error": str(e)}
        if world_size > 1:
            import threading
            th = threading.Thread(target=_ba, daemon=True)
            th.start()
            th.join(timeout=120.0)
            if th.is_alive():
                ba_hung = True
                box["ba"] = {"error": "the sharded local-BA leg did not finish within 120 s on rank %d" % rank}
        else:
            _ba()
        ba = box.get("ba")

    rc = 0
    if rank == 0:
        alg = algorithmic_bytes_per_pair()

        def alg_key(k):  # the byte model lists blur and describe together
            return "k_blur+k_describe" if k in ("k_blur", "k_describe") else k

        def ser(k):      # (launches, total ms) of a byte-model entry in the single-stream pass
            if k == "k_blur+k_describe":
                kb, kd = kern_serial.get("k_blur", (0, 0.0)), kern_serial.get("k_describe", (0, 0.0))
                return (max(kb[0], kd[0]), kb[1] + kd[1])
            return kern_serial.get(k)
        launches, ms = kern[dom]
        per_launch_ms = ms / launches
        # bytes ONE launch of the dominant kernel must move = per-pair bytes x B pairs x steps / launches in the timed region
        alg_per_launch = alg.get(alg_key(dom), 0) * B * a.steps / launches
        if alg_key(dom) == "k_blur+k_describe":  # the pair shares P: charge the launch with the pair's time
            other = "k_describe" if dom == "k_blur" else "k_blur"
            per_launch_model_ms = per_launch_ms + kern_serial[other][1] / kern_serial[other][0]
        else:
            per_launch_model_ms = per_launch_ms
        achieved = alg_per_launch / (per_launch_model_ms * 1e-3) / 1e9 if per_launch_model_ms > 0 else 0.0
        traffic, traffic_src = None, None
        try:  # PMC counters cannot be read inside this process: per-launch means of separate rocprofv3 --pmc passes
            tj = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
            if not os.path.exists(tj):
                tj = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")
            with open(tj) as f:
                pt = json.load(f)
            if pt.get("batch") == B and dom in pt["kernels"]:
                traffic = pt["kernels"][dom]["hbm_bytes_per_launch"]
                traffic_src = "profiles/" + os.path.basename(tj) + ": " + pt.get("method", "")
        except (OSError, ValueError, KeyError):
            pass
        ms_step = elapsed / a.steps * 1e3
        runner_up = None
        try:  # the second throughput bottleneck: blur + describe on their shared byte model (they move P once between them)
            bd_ms = (kern_serial["k_blur"][1] + kern_serial["k_describe"][1]) / PROBE
            bd_alg = alg["k_blur+k_describe"] * B
            runner_up = {"kernel": "k_blur+k_describe", "bound": "hbm", "achieved": bd_alg / (bd_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": bd_alg / (bd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms_per_step_single_stream": bd_ms,
                         "algorithmic_bytes_per_step": bd_alg, "traffic": None}
            with open(os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")) as f:
                pt = json.load(f)
            if pt.get("batch") == B:
                # two launches of each per step (front images, bird images); the json holds the mean over both kinds
                tb = sum(pt["kernels"][k]["hbm_bytes_per_launch"] * 2 for k in ("k_blur", "k_describe") if k in pt["kernels"])
                runner_up["traffic"] = tb
                runner_up["traffic_over_algorithmic"] = tb / bd_alg
        except (OSError, ValueError, KeyError, ZeroDivisionError):
            pass
        out = {
            "metric": "frames/s (extract+match+pose-opt), 1280x720+512x512 pair",
            "value": world_size * B * a.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world_size,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8 (extract, Hamming) + f64 (pose optimisation)",
            "data": "synthetic",
            "config": {"workload": "configs[2]: 1280x720 front + 512x512 bird pair, HIP ORB extract + M3/M9 Hamming match + "
                                   "PoseOptimizationWithBird (~2k front + ~1k bird edges)",
                       "frame_pairs_per_step_per_gpu": B, "nfeatures": 2000, "parallelism": "replicas x%d" % world_size},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": per_launch_ms, "launches_timed": launches, "algorithmic_bytes_per_launch": alg_per_launch,
                         "avg_launch_ms_single_stream": kern_serial[dom][1] / kern_serial[dom][0],
                         "byte_model": "SURVEY 8(d): FAST reads every level once (P); blur + describe share P; resize (P-px7)+(P-px0)",
                         "note": "dominant kernel = largest total in the untimed single-stream pass (every kernel bracketed by HIP "
                                 "events on its launch stream); achieved = its algorithmic bytes per launch / its average launch "
                                 "duration in the timed region, where only this kernel carries events",
                         "step": {"algorithmic_bytes_per_pair": PAIR_BYTES, "achieved": PAIR_BYTES * B / (ms_step * 1e-3) / 1e9,
                                  "frac": PAIR_BYTES * B / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, "unit": "GB/s",
                                  "note": "whole step: extract bytes of B pairs / ms_per_step"}},
            "roofline_runner_up": runner_up,
            "kernels_ms_per_step_single_stream": {k: v[1] / PROBE for k, v in sorted(kern_serial.items(), key=lambda kv: -kv[1][1])},
            "kernel_ms_sum_single_stream": sum(v[1] for v in kern_serial.values()) / PROBE,
            # north_star: "HBM GB/s for matching" -- algorithmic bytes / single-stream kernel time for every byte-model entry
            "hbm_gbs_single_stream": {k: {"gbs": alg[k] * B * PROBE / (ser(k)[1] * 1e-3) / 1e9,
                                          "frac": alg[k] * B * PROBE / (ser(k)[1] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                                      for k in alg if ser(k) and ser(k)[1] > 0},
            "workload_check": {"kps_front": float(res["n_front"].mean()), "kps_bird": float(res["n_bird"].mean()),
                               "front_matches": float(res["nm_front"].mean()), "bird_matches": float(res["nm_bird"].mean()),
                               "pose_inliers": float(res["ninliers"].mean())},
        }
        try:
            out["match"] = match_pair_rate(pipe, world, kern_serial, PROBE, B)
        except Exception as e:  # a reporting extra must never cost the bench line
            out["match"] = {"error": str(e)[:200]}
        try:
            out["host_fed"] = host_fed_leg(pipe, front, bird, B)
        except Exception as e:  # noqa: BLE001
            out["host_fed"] = {"error": str(e)[:200]}
        out["local_ba"] = ba
        out["config5_replicas"] = config5
        # the CPU legs run on rank 0 of the single-GPU run only (N > 1 would stall the other ranks)
        out["cpu_baseline"], out["parity_check"] = None, None
        if a.cpu_sample > 0 and world_size == 1:
            ns = min(a.cpu_sample, B)
            fk, fd = pipe.keypoints_host("front")
            bk, bd = pipe.keypoints_host("bird")
            refs = []
            out["cpu_baseline"] = cpu_baseline(world, ns, rank, refs)
            out["parity_check"] = parity_check(refs, fk, fd, bk, bd, res)
            if out["parity_check"]["mismatches"]:
                rc = 3
        if not a.no_single and world_size == 1:
            try:
                out["single_sequence"] = {"independent_frames": single_sequence_leg(FramePipeline, rank, local_rank),
                                          "note": "independent_frames = the configs[2] step at B = 1 / 8 with NO data dependency between "
                                                  "consecutive steps (throughput of unrelated frame pairs); dependent = the tracking chain below, "
                                                  "frame k+1 predicted from frame k's optimised pose: THAT is a single sequence"}
            except Exception as e:
                out["single_sequence"] = {"error": str(e)[:200]}
        if not a.no_chain and world_size == 1:
            try:
                tcl = track_chain_leg(rank, local_rank)
                out["track_chain"] = tcl
                if isinstance(out.get("single_sequence"), dict):
                    out["single_sequence"]["dependent"] = {k: {"serial": tcl[k]["dependent"], "pipelined_frame_construction": tcl[k]["dependent_pipelined"]}
                                                           for k in ("b1", "b8") if k in tcl}
                if tcl.get("parity_check", {}).get("mismatches"):
                    rc = 3
            except Exception as e:
                out["track_chain"] = {"error": str(e)[:300]}
            try:
                out["host_abi"] = host_abi_leg(rank)
                if out.get("cpu_baseline"):
                    out["host_abi"]["cpu_oracle_stage_ms_median"] = out["cpu_baseline"]["one_core"]["stage_ms_median"]
            except Exception as e:
                out["host_abi"] = {"error": str(e)[:300]}
        print(json.dumps(out), flush=True)
        if rc:
            print("bench.py: GPU results differ from the oracle: %s" % out["parity_check"]["details"], file=sys.stderr, flush=True)
    sys.stdout.flush()
    if ba_hung:  # a thread of this process is stuck inside a collective: leave without waiting for it
        os._exit(rc)
    if world_size > 1:
        # orderly shutdown when every rank gets here; a peer that left through the branch above must not hold this one
        import threading
        done = threading.Event()

        def _bye():
            try:
                dist.barrier()
                dist.destroy_process_group()
            finally:
                done.set()
        threading.Thread(target=_bye, daemon=True).start()
        if not done.wait(timeout=60.0):
            os._exit(rc)
    pipe.close()
    sys.exit(rc)


if __name__ == "__main__":
    main()
