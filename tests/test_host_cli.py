"""Host logic of the product binary that needs no GPU: option table and
validation rules of the reference (/root/reference/src/compairr.cc:327-689)."""

import os
import subprocess

import pytest

from conftest import GOLDEN_INPUTS, ROOT

EXE = os.path.join(ROOT, "bin", "compairr")


def run(*args):
    return subprocess.run([EXE] + list(args), cwd=GOLDEN_INPUTS, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE)


def test_help_and_version():
    p = run("--help")
    assert p.returncode == 0 and b"--matrix" in p.stderr
    assert run("-v").returncode == 0


@pytest.mark.parametrize("args,msg", [
    ([], b"Please specify a command"),
    (["-m", "-x", "seta.tsv", "setb.tsv"], b"just one command"),
    (["-m"], b"Incorrect number of arguments"),
    (["-m", "a", "b", "c"], b"Incorrect number of arguments"),
    (["-m", "seta.tsv", "-d", "-1"], b"cannot be negative"),
    (["-m", "seta.tsv", "-d", "2", "-i"], b"Indels are only allowed when d=1"),
    (["-m", "seta.tsv", "-d", "1x"], b"Invalid numeric argument"),
    (["-m", "seta.tsv", "-t", "0"], b"Illegal number of threads"),
    (["-m", "seta.tsv", "-t", "257"], b"Illegal number of threads"),
    (["-m", "seta.tsv", "-s", "bogus"], b"must be MH, Jaccard"),
    (["-m", "seta.tsv", "-d", "1", "-s", "MH"], b"Morisita-Horn index is not defined"),
    (["-m", "seta.tsv", "-d", "1", "-s", "jaccard"], b"Jaccard index is not defined"),
    (["-m", "seta.tsv", "-g", "-g"], b"specified more than once"),
    (["-m", "seta.tsv", "-k", "x"], b"--keep-columns only allowed with --pairs"),
    (["-x", "seta.tsv"], b"Two input files must be specified"),
    (["-x", "setc.tsv", "setb.tsv", "-s", "MH"], b"only allowed when computing repertoire overlap"),
    (["-c", "seta.tsv", "setb.tsv"], b"One input file must be specified"),
    (["-c", "seta.tsv", "-p", "/dev/null"], b"not allowed with -c or --cluster"),
    (["-z", "seta.tsv"], b"not part of the MI355X build"),
    (["-m", "seta.tsv", "-p", "/dev/null", "-k", "a,,b"], b"Illegal list of columns"),
])
def test_rejected_command_lines(args, msg):
    p = run(*args)
    assert p.returncode == 1
    assert msg in p.stderr
    assert p.stdout == b""


@pytest.mark.parametrize("threads", ["1", "3"])
def test_standard_input_as_a_set(threads):
    """"-" = standard input (util.cc:156-170), here for set 1 and for set 2, through the
    host program linked with the CPU oracle backend (no GPU needed)."""
    exe = os.path.join(ROOT, "tests", "bin", "compairr_oracle_cli")
    with open(os.path.join(ROOT, "tests", "golden", "expected", "ref_test_sh.tsv"), "rb") as fh:
        want = fh.read()
    for files, feed in ((["-", "setb.tsv"], "seta.tsv"), (["seta.tsv", "-"], "setb.tsv")):
        with open(os.path.join(GOLDEN_INPUTS, feed), "rb") as fh:
            p = subprocess.run([exe, "-m"] + files + ["-d", "1", "-i", "-t", threads, "-l", os.devnull],
                               cwd=GOLDEN_INPUTS, stdin=fh, stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()
        assert p.stdout == want
