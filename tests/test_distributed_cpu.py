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
from compairr_amd import RepertoireSet
from compairr_amd.dist import routed_overlap, shard_bounds, sharded_overlap


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


# ---- the routed exchange (compairr_amd.dist.exchange_queries: what bench.py runs on RCCL) ----

REC = np.dtype([("cnt", "<u8"), ("v", "<u4"), ("j", "<u4"), ("rep", "<u4"), ("len", "<u4"),
                ("res", "u1", (36,)), ("orig", "<u4")])
assert REC.itemsize == 64              # layout.h QueryRec, the unit of the real exchange too


class StandInContext:
    """What dist.exchange_queries needs of a library context, in numpy: records of 64 bytes
    (the library's own record for sequences of up to 36 residues), a query owned by the rank
    its sequence hashes to and, like a class-position item, sometimes wanted by a second rank
    that only keeps its record.  The compute leg is the oracle."""

    def __init__(self, rank, world, set2, opt):
        self.rank, self.world, self.set2, self.opt = rank, world, set2, opt

    @staticmethod
    def owner(seq_bytes, world):
        h = 2166136261
        for c in seq_bytes:
            h = ((h ^ int(c)) * 16777619) & 0xffffffff
        return h % world, (h >> 8) % world

    def route_queries(self, share, first_index, n_dest):
        assert n_dest == self.world
        self.share, self.first = share, first_index
        self.dests = []
        for i in range(share.n):
            a, b = int(share.offsets[i]), int(share.offsets[i + 1])
            o, second = self.owner(share.residues[a:b], n_dest)
            self.dests.append(sorted({o, second}) if (b - a) % 3 == 0 else [o])
        counts = np.zeros(n_dest, dtype=np.int64)
        for d in self.dests:
            for x in d:
                counts[x] += 1
        tot = np.zeros(share.n_repertoires)
        np.add.at(tot, share.repertoire, share.count.astype(np.float64))
        return counts, REC.itemsize, tot

    def route_pack(self, d_send, capacity_bytes):
        import ctypes
        s = self.share
        recs = [[] for _ in range(self.world)]
        for i, ds in enumerate(self.dests):
            a, b = int(s.offsets[i]), int(s.offsets[i + 1])
            r = np.zeros((), dtype=REC)
            r["cnt"], r["v"], r["j"], r["rep"] = s.count[i], s.v_gene[i], s.j_gene[i], s.repertoire[i]
            r["len"], r["orig"] = b - a, self.first + i
            r["res"][:b - a] = s.residues[a:b]
            for d in ds:
                recs[d].append(r)
        flat = np.array([r for d in recs for r in d], dtype=REC)
        assert flat.nbytes <= capacity_bytes
        ctypes.memmove(d_send, flat.ctypes.data, flat.nbytes)

    def set_queries_routed(self, d_records, n, n_rep, n_total, totals):
        import ctypes
        buf = (ctypes.c_char * (n * REC.itemsize)).from_address(d_records) if n else b""
        r = np.frombuffer(buf, dtype=REC, count=n).copy()
        mine = np.array([self.owner(x["res"][:x["len"]], self.world)[0] == self.rank for x in r], dtype=bool)
        self.received, self.n_total, self.totals = n, n_total, totals
        r = r[mine]                                 # (the others are here for their records only)
        lens = r["len"].astype(np.int64)
        off = np.zeros(len(r) + 1, dtype=np.uint64)
        np.cumsum(lens, out=off[1:])
        res = np.concatenate([x["res"][:x["len"]] for x in r]) if len(r) else np.zeros(0, np.uint8)
        self.queries = RepertoireSet(res, off, r["v"], r["j"], r["rep"], r["cnt"],
                                     ["R%d" % k for k in range(n_rep)])
        self.orig = r["orig"]

    def matrix(self):
        m, _ = _oracle.overlap(self.queries, self.set2, self.opt)
        m = _oracle.integer_cells(m, self.opt)
        return m


def _routed_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a = synth.make_set(6001, 1, prefix="A", pool_size=1500)
    b = synth.make_set(5000, 2, prefix="B", pool_size=1500)
    opt = Options(differences=1, n_v_genes=synth.N_V, n_j_genes=synth.N_J)
    h = StandInContext(rank, world, b, opt)
    full, moved = routed_overlap(h, lambda ctx: ctx.matrix(), a, rank, world, device="cpu")
    # every query is computed by exactly one rank, the sequence numbers are those of the whole set
    seen = torch.zeros(a.n, dtype=torch.int64)
    seen[torch.from_numpy(h.orig.astype(np.int64))] = 1
    dist.all_reduce(seen)
    assert bool((seen == 1).all())
    assert h.n_total == a.n and moved["records_received"] == h.received >= len(h.orig)
    want_tot = np.zeros(a.n_repertoires)
    np.add.at(want_tot, a.repertoire, a.count.astype(np.float64))
    assert np.array_equal(h.totals, want_tot)              # summed over the shares
    if rank == 0:
        np.save(out, full)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_routed_exchange_gloo_equals_single(tmp_path, world):
    out = str(tmp_path / "m.npy")
    mp.spawn(_routed_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    a = synth.make_set(6001, 1, prefix="A", pool_size=1500)
    b = synth.make_set(5000, 2, prefix="B", pool_size=1500)
    opt = Options(differences=1, n_v_genes=synth.N_V, n_j_genes=synth.N_J)
    want, _ = _oracle.overlap(a, b, opt)
    assert np.array_equal(got, _oracle.integer_cells(want, opt))
    assert got.sum() > 0


# ---- bench.py --gpus N starts its own ranks (compairr_amd.dist.spawn_ranks) ----

def _json_lines(text):
    import json
    return [json.loads(l) for l in text.splitlines() if l.startswith("{")]


def test_spawn_ranks_relays_the_line_and_the_exit_code(capfd):
    from compairr_amd.dist import spawn_ranks
    stub = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rank_stub.py")
    rc = spawn_ranks(stub, ["--steps", "3"], 2, timeout=300)
    out = capfd.readouterr().out
    assert rc == 0
    lines = _json_lines(out)
    assert len(lines) == 1 and lines[0] == {"ranks_seen": 2, "sum": 3, "argv": ["--steps", "3"]}
    # a rank that dies takes the job's exit code with it
    rc = spawn_ranks(stub, ["--fail", "1"], 2, timeout=300)
    capfd.readouterr()
    assert rc != 0


def test_spawn_ranks_ignores_a_stale_rank_environment(capfd):
    from compairr_amd.dist import spawn_ranks
    stub = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rank_stub.py")
    env = dict(os.environ, RANK="5", WORLD_SIZE="9", MASTER_PORT="1")
    assert spawn_ranks(stub, [], 3, env=env, timeout=300) == 0
    assert _json_lines(capfd.readouterr().out) == [{"ranks_seen": 3, "sum": 6, "argv": []}]


def test_bench_refuses_more_gpus_than_the_box_has():
    """`python3 bench.py --gpus 8` on a box without eight devices: one clear line, non-zero exit, no
    traceback -- and with a launcher around it, a world size that is not --gpus is refused the same way."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    have = torch.cuda.device_count()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(have + 8)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE")})
    err = p.stderr.decode()
    assert p.returncode != 0 and "Traceback" not in err
    assert "--gpus %d but this box shows %d HIP device" % (have + 8, have) in err
    assert not p.stdout
