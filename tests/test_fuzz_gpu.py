"""A seeded, time-boxed slice of the two randomised sweeps, inside the suite the driver runs
(`-m gpu`): tests/fuzz_gpu.py (library through the C ABI against the oracle: random inputs,
options, layout tunables -- class anchors included --, work shards, routed shards) and
tests/fuzz_cli_gpu.py (bin/compairr against the reference binary, random command lines).
The long sweeps are still run by hand with other seeds."""

import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _sweep(script, seconds, seed):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", script), "--seconds", str(seconds),
                        "--seed", str(seed)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=seconds + 600)
    return p.returncode, (p.stdout.decode() + p.stderr.decode())[-3000:]


@pytest.mark.parametrize("seed", [20261003, 4])
def test_library_fuzz_slice(seed):
    rc, text = _sweep("fuzz_gpu.py", 35, seed)
    assert rc == 0 and "all bit-exact" in text, text


def test_cli_fuzz_slice():
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "compairr")):
        pytest.skip("oracle/_ref/compairr (the compiled reference) did not travel to this box")
    rc, text = _sweep("fuzz_cli_gpu.py", 40, 20261003)
    assert rc == 0 and "all identical to the reference" in text, text
