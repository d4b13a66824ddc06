"""Multi-GPU plumbing for the hot path.

Queries (set 1) are independent, the set-2 index is read-only and the matrix is
a commutative integer sum, so the path shards by query range with ONE exchange
step: a sum-reduction of the R1 x R2 matrix.  torch.distributed is plumbing
only (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""

from __future__ import annotations

from typing import Callable, Tuple

import numpy as np


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) of `n` queries for `rank` of `world`
    (the reference hands out contiguous 1000-query chunks, overlap.cc:421-433)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allreduce_matrix(t):
    """In-place sum over ranks of an int64 tensor holding uint64 bit patterns
    (two's-complement addition is the same operation)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def sharded_overlap(compute: Callable, set1, set2, rank: int, world: int, device="cpu"):
    """Strong-scaling form: rank computes its query shard with `compute(shard,
    set2) -> uint64 matrix`, then the matrices are summed across ranks.
    Returns the full matrix (same on every rank)."""
    import torch
    lo, hi = shard_bounds(set1.n, rank, world)
    part = compute(set1.subset(slice(lo, hi)), set2)
    t = torch.from_numpy(np.ascontiguousarray(part).view(np.int64).copy()).to(device)
    allreduce_matrix(t)
    return t.cpu().numpy().view(np.uint64).reshape(part.shape)
