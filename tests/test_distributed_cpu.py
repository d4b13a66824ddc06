"""N > 1 path on CPU: world size 2 over gloo.  The compute leg is the oracle
(there is no GPU here); what is under test is the sharding + the single
sum-reduction of the matrix that bench.py / compairr_amd.dist use on RCCL."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _oracle
from compairr_amd import Options, synth
from compairr_amd.dist import shard_bounds, sharded_overlap


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 64, 1000, 12345):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a = synth.make_set(6000, 1, prefix="A", pool_size=1500)
    b = synth.make_set(5000, 2, prefix="B", pool_size=1500)
    opt = Options(differences=1, indels=True, n_v_genes=synth.N_V, n_j_genes=synth.N_J)

    def compute(q, ref):
        m, _ = _oracle.overlap(q, ref, opt)
        return _oracle.integer_cells(m, opt)

    full = sharded_overlap(compute, a, b, rank, world)
    if rank == 0:
        np.save(out, full)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo_equals_single(tmp_path):
    out = str(tmp_path / "m.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    a = synth.make_set(6000, 1, prefix="A", pool_size=1500)
    b = synth.make_set(5000, 2, prefix="B", pool_size=1500)
    opt = Options(differences=1, indels=True, n_v_genes=synth.N_V, n_j_genes=synth.N_J)
    want, _ = _oracle.overlap(a, b, opt)
    assert np.array_equal(got, _oracle.integer_cells(want, opt))
    assert got.sum() > 0
