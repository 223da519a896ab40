"""Data-parallel plumbing: one process per GPU, rows / samples sharded by global
index, ONE all-reduce(sum) per step of a small flat buffer -- the float64 Gram accumulator or the
flat weight-gradient vector.  Backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in CPU tests.
Nothing in the reference does this (it has no distributed code)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard(n: int, r: int | None = None, w: int | None = None):
    """Contiguous range of the n global rows owned by rank r: (start, count).  Ranges differ by at
    most one row and depend only on (n, world), so a counter-based generator keyed on the global
    row index produces the same data set for every GPU count."""
    r = rank() if r is None else r
    w = world() if w is None else w
    if not 0 <= r < w:
        raise ValueError(f"rank {r} outside world of {w}")
    base, rem = divmod(int(n), w)
    start = r * base + min(r, rem)
    return start, base + (1 if r < rem else 0)


def allreduce_sum_(t: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks of one contiguous buffer (no-op for a single process)."""
    if world() > 1:
        if not t.is_contiguous():
            raise ValueError("all-reduce needs one contiguous buffer (do not fragment it)")
        if t.is_cuda and dist.get_backend() == "gloo":
            # rehearsals of the N > 1 path on a one-GPU box: stage through the host (gloo's own
            # device path was seen to dead-lock with two ranks sharing one GPU)
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            t.copy_(host)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def broadcast_(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    """In-place broadcast of one contiguous buffer from rank `src` (no-op for a single process):
    replicas must start from the SAME weights, whatever each process's RNG produced."""
    if world() > 1:
        if not t.is_contiguous():
            raise ValueError("broadcast needs one contiguous buffer")
        if t.is_cuda and dist.get_backend() == "gloo":
            host = t.cpu()
            dist.broadcast(host, src=src)
            t.copy_(host)
        else:
            dist.broadcast(t, src=src)
    return t


def reduce_gram_(acc: torch.Tensor) -> torch.Tensor:
    """Partial Gram matrices of the row shards -> the Gram of all rows, on every rank."""
    if acc.dtype != torch.float64:
        raise TypeError("the Gram accumulator is reduced in float64")
    return allreduce_sum_(acc)


def reduce_grads_(flat: torch.Tensor) -> torch.Tensor:
    """Flat gradient of each rank's share of a mean loss (already scaled by 1/B_global, see
    wm_model_train_f32) -> gradient of the global mean loss."""
    return allreduce_sum_(flat)
