"""bench.py without a GPU: it refuses to run (no CPU fallback), and the arithmetic of its roofline objects --
the layout kernels against the HBM peak, the probe kernel against the guide's VALU / LDS / HBM peaks from the
committed counter summary -- is what DESIGN.md section 5 says it is."""

import importlib.util
import json
import os
import subprocess
import sys
import types

import pytest

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    p = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "0", "--queries", "1000",
                        "--refs", "1000", "--cpu-sample", "-1"], cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 1
    assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]      # no bench line
    assert "no CPU fallback" in p.stderr.decode() + p.stdout.decode()


def test_layout_rooflines_are_algorithmic_bytes_over_time_against_8_tb_s():
    b = _bench()
    n, residues, slots, items = 10_000_000, 150_000_000, 10_100_000, 9_000_000
    times = {"keys": 0.5, "scatter": 0.5, "tiles": 0.25}
    out = b.layout_rooflines("no such workload", n, residues, slots, items, False, times)
    assert set(out) == {"keys", "scatter", "tiles"}
    # keys_kernel: residues + offset 8, v 4, j 4, repertoire 4, count 8 read; group 4, rank 4, hash 8, class key 4
    # written; one 4-byte counter read-modify-write
    keys_bytes = residues + n * (28 + 8 + 8 + 4 + 8)
    assert out["keys"]["algorithmic_bytes_per_launch"] == keys_bytes
    assert out["keys"]["achieved"] == pytest.approx(keys_bytes / 0.5e-3 / 1e9)
    assert out["keys"]["peak"] == 8000.0 and out["keys"]["unit"] == "GB/s" and out["keys"]["bound"] == "hbm"
    assert out["keys"]["frac"] == pytest.approx(out["keys"]["achieved"] / 8000.0)
    # scatter_kernel: the same reads + what keys_kernel left + the group base; the 64-byte record; 16 bytes per item
    assert out["scatter"]["algorithmic_bytes_per_launch"] == residues + n * (28 + 8 + 8 + 4 + 4 + 64) + items * 16
    # -i: three hashes travel
    indel = b.layout_rooflines("no such workload", n, residues, slots, items, True, times)
    assert indel["keys"]["algorithmic_bytes_per_launch"] == keys_bytes + n * 16
    # a kernel that did not run (record tiles: no fill_tiles_kernel) has no entry
    assert "tiles" not in b.layout_rooflines("x", n, residues, slots, items, False, {"keys": 0.5, "scatter": 0.5})
    assert out["keys"]["traffic"] is None                                             # (no counters for that name)


def test_probe_roofline_prices_the_vector_unit_at_the_guides_peak():
    b = _bench()
    with open(os.path.join(ROOT, "profiles", "roofline_inputs.json")) as fh:
        inp = json.load(fh)
    name = "synthetic 10M-vs-10M CDR3aa, d=1 substitutions only, V/J matched"
    w = inp["workloads"][name]
    st = types.SimpleNamespace(algorithmic_bytes=23_230_076_653, variants=2_860_009_519,
                               filter_reads=int(w["filter_reads"]) if w.get("filter_reads") else 111_363_495,
                               bloom_positive=1_895_585, matches=2_013_144)
    probe_ms = 0.44
    r = b.roofline(name, st, probe_ms, probe_ms + 0.14, "probe_rows_kernel")
    # the guide: 256 CUs x 4 SIMDs, 2.4 GHz, a wave64 instruction issues over 2 cycles
    assert r["unit"] == "wave-instructions/s" and r["peak"] == pytest.approx(1024 * 2.4e9 / 2)
    assert r["bound"] == "valu"
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    assert r["frac"] == max(r["utilisation"].values())
    assert 0.2 < r["frac"] < 0.6 and r["frac_mix_priced"] > r["frac"]
    assert r["traffic"] == pytest.approx(w["hbm_bytes"], rel=0.05)
    # SURVEY 8d's byte model is reported, never as this kernel's rate
    assert r["algorithmic_equiv"]["of_hbm_peak"] > 1 and "NOT a rate" in r["algorithmic_equiv"]["note"]
    # counters taken on other sources would be flagged
    assert r["counters_stale"] is False
