"""bench.py itself, on the GPU box: the JSON contract, and the RCCL path (process
group on backend nccl + all-reduce of the matrix on the kernels' stream) executed
at world size 1 under torch.distributed.run -- the code the driver runs at N > 1."""

import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SMALL = ["--steps", "3", "--warmup", "1", "--queries", "300000", "--refs", "300000",
         "--cpu-sample", "-1"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(cmd):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract_and_rccl_path_at_world_1():
    plain = _run([sys.executable, "bench.py", "--gpus", "1"] + SMALL)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
                "cpu_baseline", "value_incl_layout", "value_incl_layout_cold", "value_from_device_soa",
                "step_ms_incl_d2h", "parity_vs_reference_full_size", "value_resident_step", "step_kernels_ms",
                "roofline_kernels", "step"):
        assert key in plain, key
    assert plain["n_gpus"] == 1 and plain["ranks_seen"] == 1 and plain["steps"] == 3 and plain["value"] > 0
    assert plain["scaling"] == "strong" and plain["dtype"] == "u64"
    assert set(plain["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert plain["device_resident_inputs"]["same_matrix"] is True
    assert plain["resident_steps_same_matrix"] is True
    # a step is a query set from its device arrays to the matrix: the launches alone are faster, the set
    # from host buffers slower
    assert plain["value_resident_step"] > plain["value"] > plain["value_incl_layout"] > 0
    # (amino acids at d = 1 run on record tiles: fill_tiles_kernel -- "tiles" -- does not run)
    assert set(plain["step_kernels_ms"]) >= {"keys", "scatter", "probe", "resolve"}
    assert set(plain["roofline_kernels"]) >= {"keys", "scatter", "probe"}
    assert plain["config"]["layout"]["record_tiles"] == 1 and "tiles" not in plain["step_kernels_ms"]
    # ... `--step resident` is rounds 1-5's definition
    res = _run([sys.executable, "bench.py", "--gpus", "1", "--step", "resident"] + SMALL)
    assert res["config"]["matrix_checksum"] == plain["config"]["matrix_checksum"]
    assert res["value"] > plain["value"]
    # ... `--skip-host-layout` (the profiling passes): no set from host buffers is laid out, nothing else changes
    skip = _run([sys.executable, "bench.py", "--gpus", "1", "--skip-host-layout"] + SMALL)
    assert skip["value_incl_layout"] is None and skip["config"]["query_layout_ms"] is None
    assert skip["config"]["matrix_checksum"] == plain["config"]["matrix_checksum"] and skip["value"] > 0
    # the same workload through torch.distributed.run: process group on nccl (= RCCL),
    # all-reduce of the matrix inside every step
    dist = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                 "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                 "bench.py", "--gpus", "1"] + SMALL)
    assert dist["n_gpus"] == 1
    assert dist["config"]["matrix_checksum"] == plain["config"]["matrix_checksum"]
    assert dist["config"]["query_layout_ms"]["exchange"] is None       # (query shards: nothing but the matrix moves)
    # ... the work-shard split: the queries go through compairr_amd.dist.exchange_queries (route, pack,
    # all-to-all, receive) in every step
    work = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                 "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                 "bench.py", "--gpus", "1", "--shard-by", "work"] + SMALL)
    assert work["config"]["matrix_checksum"] == plain["config"]["matrix_checksum"]
    ex = work["config"]["query_layout_ms"]["exchange"]
    assert ex["records_sent"] == ex["records_received"] == 300000 and ex["record_bytes"] == 64
    rep = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                "bench.py", "--gpus", "1", "--shard-by", "work", "--layout", "replicated"] + SMALL)
    assert rep["config"]["matrix_checksum"] == plain["config"]["matrix_checksum"]
    assert rep["config"]["query_layout_ms"]["exchange"] is None
    # ... and the way the driver starts N > 1 (`python3 bench.py --gpus N`, no launcher around it): bench.py starts
    # its ranks itself as a child torch.distributed.run -- forced here at N = 1
    own = _run([sys.executable, "bench.py", "--gpus", "1", "--launcher", "always"] + SMALL)
    assert own["n_gpus"] == 1 and own["ranks_seen"] == 1
    assert own["config"]["matrix_checksum"] == plain["config"]["matrix_checksum"]
    # weak scaling keeps the same shard at N = 1
    weak = _run([sys.executable, "bench.py", "--gpus", "1", "--scaling", "weak"] + SMALL)
    assert weak["scaling"] == "weak"
    assert weak["config"]["matrix_checksum"] == plain["config"]["matrix_checksum"]


def test_more_gpus_than_the_box_has_is_one_clear_line():
    import torch
    have = torch.cuda.device_count()
    p = subprocess.run([sys.executable, "bench.py", "--gpus", str(have + 1)] + SMALL, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and b"Traceback" not in p.stderr
    assert b"HIP device" in p.stderr
