"""Multi-GPU plumbing for the hot path, one process per GPU.

The set-2 index is replicated, the matrix is a commutative integer sum, and the WORK of a step
divides by filter slice (library tunables work_shard_count / work_shard_index).  What crosses
the links:

  * once per query set: every rank uploads and keys only ITS share of the queries
    (cmpr_route_queries), the records move to the ranks that work on them with one
    all-to-all over xGMI (`exchange_queries`: counts, then the records), and every rank lays
    out what it received (cmpr_set_queries_routed);
  * once per step: one sum-reduce of the R1 x R2 matrix (`allreduce_matrix`).

The reference's counterpart is the chunk hand-out of overlap.cc:421-433 and the merge of the
threads' private matrices, overlap.cc:510-527.  torch.distributed is plumbing only (backend
"nccl" = RCCL on the GPU box, "gloo" in the CPU tests, which drive this module with a
stand-in for the library context: `h` below is anything with route_queries / route_pack /
set_queries_routed).
"""

from __future__ import annotations

from typing import Callable, Optional, Tuple

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


def exchange_queries(h, share, first_index: int, n_total: int, rank: int, world: int,
                     device="cuda", group=None) -> dict:
    """One query set over `world` contexts: this rank hands `share` (sequences first_index ..
    first_index + share.n - 1 of the whole set of n_total) to its context `h`, the records go
    where their work is, `h` ends up with what it works on.  Returns the sizes that moved.

    With a process group up (also at world size 1, so that a one-GPU torchrun executes it)
    the exchange is torch.distributed.all_to_all_single on `device` buffers; without one it
    is the identity."""
    import torch
    import torch.distributed as dist
    use_dist = dist.is_available() and dist.is_initialized()
    assert (dist.get_world_size(group) if use_dist else 1) == world
    counts, record_bytes, totals = h.route_queries(share, first_index, world)
    counts = np.asarray(counts, dtype=np.int64)
    nsend = int(counts.sum())
    send = torch.empty(max(nsend, 1) * record_bytes, dtype=torch.uint8, device=device)
    h.route_pack(send.data_ptr(), nsend * record_bytes)
    if use_dist:
        # how many records each rank sends me
        sc = torch.from_numpy(counts).to(device)
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=group)
        recv_counts = rc.cpu().numpy()
        nrecv = int(recv_counts.sum())
        recv = torch.empty(max(nrecv, 1) * record_bytes, dtype=torch.uint8, device=device)
        dist.all_to_all_single(recv[:nrecv * record_bytes], send[:nsend * record_bytes],
                               output_split_sizes=[int(x) * record_bytes for x in recv_counts],
                               input_split_sizes=[int(x) * record_bytes for x in counts], group=group)
        # duplicate_count totals per repertoire of the WHOLE set: they bound the summed cells
        tot = torch.from_numpy(np.ascontiguousarray(totals, dtype=np.float64)).to(device)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
        totals = tot.cpu().numpy()
    else:
        assert world == 1
        recv, nrecv = send, nsend
    if str(device) != "cpu":
        torch.cuda.synchronize()
    h.set_queries_routed(recv.data_ptr(), nrecv, share.n_repertoires, n_total, totals)
    return {"records_sent": nsend, "records_received": nrecv, "record_bytes": record_bytes,
            "share": share.n}


def sharded_overlap(compute: Callable, set1, set2, rank: int, world: int, device="cpu"):
    """The literal split of the north star: rank computes a contiguous query shard with
    `compute(shard, set2) -> uint64 matrix`, then the matrices are summed across ranks.
    Returns the full matrix (same on every rank)."""
    import torch
    lo, hi = shard_bounds(set1.n, rank, world)
    part = compute(set1.subset(slice(lo, hi)), set2)
    t = torch.from_numpy(np.ascontiguousarray(part).view(np.int64).copy()).to(device)
    allreduce_matrix(t)
    return t.cpu().numpy().view(np.uint64).reshape(part.shape)


def routed_overlap(h, matrix_of: Callable, set1, rank: int, world: int, device="cuda",
                   first_and_share: Optional[Tuple[int, object]] = None):
    """The whole multi-GPU path once: contiguous share -> exchange -> this rank's part of the
    matrix (`matrix_of(h) -> uint64 matrix`) -> sum over ranks."""
    import torch
    if first_and_share is None:
        lo, hi = shard_bounds(set1.n, rank, world)
        first_and_share = (lo, set1.subset(slice(lo, hi)))
    moved = exchange_queries(h, first_and_share[1], first_and_share[0], set1.n, rank, world, device)
    part = matrix_of(h)
    t = torch.from_numpy(np.ascontiguousarray(part).view(np.int64).copy()).to(device)
    allreduce_matrix(t)
    return t.cpu().numpy().view(np.uint64).reshape(part.shape), moved


# ---- starting the ranks (bench.py --gpus N without a launcher around it) ----

def free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launcher_command(script: str, argv, n: int, port: Optional[int] = None):
    """The command line the driver itself uses for N > 1: one rank per GPU of ONE node,
    rendezvous on 127.0.0.1 (the container's hostname may not resolve)."""
    import sys
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script] + list(argv)


def spawn_ranks(script: str, argv, n: int, env=None, timeout: Optional[float] = None) -> int:
    """Start `script` on `n` ranks as a CHILD process (torch.distributed.run) and relay what it
    prints and its exit code.  To be called before anything in this process has touched the GPU
    -- the parent never initialises HIP and is never replaced by another program (an exec from a
    process that holds the GPU takes the box down on this pool); it only waits.  The reference
    starts its workers at overlap.cc:926-936; this is that loop one level up, a process per GPU."""
    import os
    import subprocess
    import sys
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
    e.setdefault("MASTER_ADDR", "127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "MASTER_PORT"):
        e.pop(k, None)
    # (a session of its own: the launcher and its N ranks are one process group, known exactly)
    p = subprocess.Popen(launcher_command(script, argv, n), env=e, stdout=subprocess.PIPE, start_new_session=True)
    try:
        out, _ = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        # SIGTERM to the group first: torch.distributed.run forwards it to its workers (a SIGKILL of the
        # launcher alone would orphan them with their GPUs and the rendezvous port); then the group is killed
        import signal
        try:
            os.killpg(p.pid, signal.SIGTERM)
        except ProcessLookupError:
            pass
        try:
            out, _ = p.communicate(timeout=10)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            out, _ = p.communicate()
        sys.stdout.write(out.decode(errors="replace"))
        sys.stdout.flush()
        return 124
    sys.stdout.write(out.decode(errors="replace"))
    sys.stdout.flush()
    return p.returncode
