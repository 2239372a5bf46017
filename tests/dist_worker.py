"""Worker of the world_size-2 tests (spawned by test_dist_*.py with RANK/WORLD_SIZE/MASTER_* set)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    mode = sys.argv[1]
    import torch
    import torch.distributed as dist
    from fishbirdeyevisualslam_amd import dist as fbd
    rank, world = fbd.init_from_env(backend="gloo")
    assert world == 2
    if mode == "helpers":
        # max over ranks, shard assignment, the all-reduce callback called through its C function pointer
        assert fbd.max_over_ranks(1.0 + rank) == 2.0
        assert fbd.shard_sequences(8, rank, world) == [rank, rank + 2, rank + 4, rank + 6]
        cb = fbd.make_allreduce()
        buf = (C.c_double * 5)(*[float(rank + 1) * (i + 1) for i in range(5)])
        assert cb(None, buf, 5, 0) == 0
        assert list(buf) == [3.0 * (i + 1) for i in range(5)]
        buf2 = (C.c_double * 2)(float(rank), -float(rank))
        assert cb(None, buf2, 2, 1) == 0
        assert list(buf2) == [1.0, 0.0]
        print("rank %d helpers ok" % rank, flush=True)
    elif mode in ("ba", "ba_config5", "ba_stop"):
        # landmark-sharded local BA on ONE GPU shared by the two ranks, gloo for the exchange (host-callback transport; the
        # protocol -- slots, exchange blocks, device-side decisions -- is the one the RCCL transport runs)
        import fishbirdeyevisualslam_amd as fb
        import oracle_lib as O
        from fishbirdeyevisualslam_amd import ba_problem, synth
        L = fb.lib()
        if mode == "ba_config5":   # BASELINE config 5's shape: 20 key frames x 8000 + 2000 points
            p = synth.make_ba_problem(4000, n_kf=20, n_mp=8000, n_mpb=2000)
        else:
            p = synth.make_ba_problem(4000, n_kf=8, n_mp=1200, n_mpb=300)
        cb = fbd.make_allreduce()
        if mode == "ba_stop":
            # pbStopFlag raised on rank 1 ONLY, before the call: rank 0 must not hang in an exchange rank 1 never enters, and
            # both ranks must come back with the same (barely optimised) result
            stop = np.array([1 if rank == 1 else 0], np.uint8)
            a, out_s, keep = ba_problem.local_ba_args(p, with_odom=1, stop_flag=stop)
            fbd.local_ba_sharded(L, a, rank, world, cb)
            t = torch.from_numpy(np.concatenate([out_s["kf_Tcw"].ravel(), out_s["mp_xw"].ravel()]).astype(np.float64))
            tmax, tmin = t.clone(), t.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
            identical = bool(torch.equal(tmax, tmin))
            print("rank %d ba_stop returned identical_across_ranks=%s" % (rank, identical), flush=True)
            assert identical
        else:
            a, out_s, keep = ba_problem.local_ba_args(p, with_odom=1)
            fbd.local_ba_sharded(L, a, rank, world, cb)
            a1, out_1, keep1 = ba_problem.local_ba_args(p, with_odom=1)
            O.call("orc_local_ba", a1)   # the CPU oracle on the unsharded problem
            rel = lambda x, y: float(np.abs(x - y).max() / max(1.0, np.abs(y).max()))
            r = [rel(out_s["kf_Tcw"], out_1["kf_Tcw"]), rel(out_s["mp_xw"], out_1["mp_xw"]), rel(out_s["mpb_xw"], out_1["mpb_xw"])]
            same = bool(np.array_equal(out_s["obs_outlier"], out_1["obs_outlier"]) and
                        np.array_equal(out_s["bobs_outlier"][: len(p["bobs_kf"])], out_1["bobs_outlier"][: len(p["bobs_kf"])]))
            # both ranks must hold the identical complete result
            t = torch.from_numpy(out_s["kf_Tcw"].astype(np.float64).copy())
            tmax = t.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            tmin = t.clone()
            dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
            identical = bool(torch.equal(tmax, tmin))
            print("rank %d %s vs oracle rel=%s flags_equal=%s identical_across_ranks=%s" % (rank, mode, ["%.2e" % x for x in r], same, identical), flush=True)
            assert max(r) <= 1e-4 and same and identical
    elif mode in ("ba_rccl", "ba_rccl_stop"):
        # the REAL multi-rank transport: one GPU per rank, ncclAllReduce inside the library over a 2-rank communicator made
        # through fb_rccl_* (gloo only carries the unique id and the final comparison).  Needs a box with >= 2 GPUs.
        import fishbirdeyevisualslam_amd as fb
        import oracle_lib as O
        from fishbirdeyevisualslam_amd import ba_problem, synth
        L = fb.lib()
        fb.check(L.fb_set_device(rank), "fb_set_device")
        torch.cuda.set_device(rank)
        comm = fbd.RcclComm(L, rank, world, device=torch.device("cuda", rank))
        n_, r_ = comm.info()
        assert (n_, r_) == (world, rank), (n_, r_)
        p = synth.make_ba_problem(4000, n_kf=20, n_mp=8000, n_mpb=2000) if mode == "ba_rccl" else synth.make_ba_problem(4000, n_kf=8, n_mp=1200, n_mpb=300)
        stop = np.array([1 if (mode == "ba_rccl_stop" and rank == 1) else 0], np.uint8)
        a, out_s, keep = ba_problem.local_ba_args(p, with_odom=1, stop_flag=stop)
        fbd.local_ba_sharded_rccl(L, a, rank, world, comm)
        t = torch.from_numpy(np.concatenate([out_s["kf_Tcw"].ravel(), out_s["mp_xw"].ravel(), out_s["mpb_xw"].ravel()]).astype(np.float64))
        tmax, tmin = t.clone(), t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        identical = bool(torch.equal(tmax, tmin))
        same, r = True, [0.0]
        if mode == "ba_rccl":
            a1, out_1, keep1 = ba_problem.local_ba_args(p, with_odom=1)
            O.call("orc_local_ba", a1)
            rel = lambda x, y: float(np.abs(x - y).max() / max(1.0, np.abs(y).max()))
            r = [rel(out_s["kf_Tcw"], out_1["kf_Tcw"]), rel(out_s["mp_xw"], out_1["mp_xw"]), rel(out_s["mpb_xw"], out_1["mpb_xw"])]
            same = bool(np.array_equal(out_s["obs_outlier"], out_1["obs_outlier"]) and
                        np.array_equal(out_s["bobs_outlier"][: len(p["bobs_kf"])], out_1["bobs_outlier"][: len(p["bobs_kf"])]))
        print("rank %d %s rccl_ranks_seen=%d rel=%s flags_equal=%s identical_across_ranks=%s" % (rank, mode, n_, ["%.2e" % x for x in r], same, identical), flush=True)
        assert max(r) <= 1e-4 and same and identical
        comm.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
