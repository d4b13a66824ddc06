"""Shared fixtures.  `-m "not gpu"` runs here on CPU (oracle vs golden vectors,
host logic, C-ABI symbols); `-m gpu` runs on the MI355X box and goes through
the C ABI of libcompairr_hip.so."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ARTIFACTS = [
    "compairr_amd/lib/libcompairr_hip.so",
    "bin/compairr",
    "oracle/liboracle.so",
    "tests/bin/compairr_oracle_cli",
    "tests/bin/compairr_oracle_cli_asan",
]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    missing = [a for a in ARTIFACTS if not os.path.exists(os.path.join(ROOT, a))]
    if missing:
        subprocess.run(["make", "-s", "-C", ROOT, "all"], check=True)


def has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


GOLDEN_INPUTS = os.path.join(ROOT, "tests", "golden", "inputs")
GOLDEN_EXPECTED = os.path.join(ROOT, "tests", "golden", "expected")


def load_manifest():
    with open(os.path.join(ROOT, "tests", "golden", "manifest.json")) as fh:
        return json.load(fh)


def run_cli(binary, case, timeout=120, log=None, pairs=None, extra=()):
    argv = [os.path.join(ROOT, binary), case.get("cmd", "-m")] + case["files"] + \
           case["args"].split() + list(extra) + ["-l", log or os.devnull]
    if case.get("pairs"):
        argv += ["-p", pairs or os.devnull]
    return subprocess.run(argv, cwd=GOLDEN_INPUTS, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout)


def warnings_of(logfile):
    with open(logfile, errors="replace") as fh:
        return [l.rstrip("\n") for l in fh if l.startswith("Warning:")]


def sorted_pairs(path) -> bytes:
    """Pairs file with its data lines sorted (their order is unspecified)."""
    with open(path, "rb") as fh:
        lines = fh.read().splitlines(keepends=True)
    return b"".join(lines[:1] + sorted(lines[1:]))


def expected_pairs_of(case) -> bytes:
    with open(os.path.join(GOLDEN_EXPECTED, case["name"] + ".pairs.tsv"), "rb") as fh:
        return fh.read()


def expected_of(case) -> bytes:
    with open(os.path.join(GOLDEN_EXPECTED, case["name"] + ".tsv"), "rb") as fh:
        return fh.read()
