"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).

* The per-frame path (extract + match + pose-opt) shards by sequence: replicas, no data-path collective
  (SURVEY 8e).  `shard_sequences` is the assignment, `max_over_ranks` the only reduction bench.py needs.
* One local BA over several GPUs is landmark-partitioned inside the library (fb_local_ba_sharded); the library
  asks the host for an all-reduce of a small host buffer twice per LM trial.  `make_allreduce` turns a
  torch.distributed process group into that callback.
"""
import ctypes as C
import os

import numpy as np

from . import cabi

# what bench.py prints next to the sharded-BA numbers
TRANSPORT_NOTE = "host-staged all-reduce callback: D2H, torch.distributed all_reduce, H2D"


def init_from_env(backend=None, device=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun sets them)."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_sequences(n_sequences, rank, world):
    """Sequence s runs on GPU s mod G (independent sequences, round-robin)."""
    return [s for s in range(n_sequences) if s % world == rank]


def max_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def make_allreduce(group=None, stage_device=None):
    """fb_allreduce_fn for fb_local_ba_sharded.  The buffer is host memory; with a NCCL/RCCL group it is staged
    through `stage_device` (a cuda device), with gloo it is reduced in place."""
    import torch
    import torch.distributed as dist

    def _cb(ctx, buf, n, op):
        try:
            a = np.ctypeslib.as_array(buf, shape=(n,))
            t = torch.from_numpy(a)
            rop = dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX
            if stage_device is not None:
                d = t.to(stage_device)
                dist.all_reduce(d, op=rop, group=group)
                t.copy_(d)
            else:
                dist.all_reduce(t, op=rop, group=group)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("allreduce callback failed:", e, flush=True)
            return 1

    return cabi.ALLREDUCE_FN(_cb)


def local_ba_sharded(lib, args, rank, world, allreduce):
    from . import check
    check(lib.fb_local_ba_sharded(C.byref(args), rank, world, allreduce, None), "fb_local_ba_sharded")
