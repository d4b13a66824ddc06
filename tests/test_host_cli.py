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


def _both(args, tmp_path):
    """the same command line through this build's host program (oracle backend) and the reference binary"""
    ours = os.path.join(ROOT, "tests", "bin", "compairr_oracle_cli")
    ref = os.path.join(ROOT, "oracle", "_ref", "compairr")
    out = []
    for exe in (ours, ref):
        if not os.path.exists(exe):
            out.append(None)
            continue
        log = str(tmp_path / (os.path.basename(exe) + ".log"))
        p = subprocess.run([exe] + args + ["-l", log], cwd=GOLDEN_INPUTS, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE)
        out.append((p.returncode, p.stdout, p.stderr, open(log).read() if os.path.exists(log) else ""))
    return out


def _error_lines(text):
    return [l for l in text.splitlines() if ("rror" in l or "Missing" in l) and not l.startswith(("Log file", "Repertoire set"))]


def test_errors_of_two_files_come_in_the_reference_order(tmp_path):
    """The two input files are read side by side (overlap_host.cc); what is reported is what the reference,
    which reads them one after the other, reports: file 1's error when both are broken, file 2's -- behind
    the log lines of file 1 -- when only it is, and nothing of file 2 when file 1 is missing."""
    good = os.path.join(GOLDEN_INPUTS, "seta.tsv")
    bad1 = str(tmp_path / "bad1.tsv")
    bad2 = str(tmp_path / "bad2.tsv")
    with open(good) as fh:
        lines = fh.read().splitlines()
    header = lines[0].split("\t")
    ci = header.index("duplicate_count")
    row = lines[2].split("\t")
    row[ci] = "-5"
    with open(bad1, "w") as fh:
        fh.write("\n".join(lines[:2] + ["\t".join(row)] + lines[3:]) + "\n")
    with open(bad2, "w") as fh:                      # (no v_call column)
        vi = header.index("v_call")
        fh.write("\n".join("\t".join(c for k, c in enumerate(l.split("\t")) if k != vi) for l in lines) + "\n")
    for a, b in ((bad1, bad2), (good, bad2), (bad1, good), (str(tmp_path / "absent.tsv"), bad2), (good, str(tmp_path / "absent.tsv"))):
        ours, ref = _both(["-m", a, b, "-d", "1"], tmp_path)
        assert ours[0] == 1 and ours[1] == b""
        text = (ours[2].decode() + ours[3])
        if a == bad1:
            assert "duplicate_count" in text and "Missing essential" not in text
        elif a == good and b == bad2:
            assert "Missing essential column(s)" in text and "v_call" in text
            # file 1's block of the log stands in front of file 2's error
            assert text.index("Immune receptor repertoire set 1") < text.index("Missing essential")
        else:
            assert "Unable to open input data file" in text
        if ref is not None:
            assert ref[0] == ours[0]
            assert _error_lines(ref[2].decode() + ref[3]) == _error_lines(text)


def test_reader_on_a_file_large_enough_for_huge_pages(tmp_path):
    """The reader keeps the file text, a range's parsed arrays and the merged arrays on 2-MiB pages where the system
    gives them on request (airr_tsv.cc advise_huge / HugeArena): 300 000 sequences are several aligned 2-MiB pieces of
    each.  Same output with the advice, without it (COMPAIRR_NO_HUGEPAGES), on 1 / 5 / 8 threads -- and as the
    reference binary's, where that is built."""
    import sys
    sys.path.insert(0, ROOT)
    from compairr_amd import synth
    a = synth.make_set(300000, 11, prefix="A", pool_size=2000)
    b = synth.make_set(3000, 12, prefix="B", pool_size=2000)
    fa, fb = str(tmp_path / "a.tsv"), str(tmp_path / "b.tsv")
    a.write_tsv_fast(fa)
    b.write_tsv_fast(fb)
    ours = os.path.join(ROOT, "tests", "bin", "compairr_oracle_cli")
    outs = []
    for threads, env in (("1", {}), ("5", {}), ("8", {}), ("8", {"COMPAIRR_NO_HUGEPAGES": "1"})):
        p = subprocess.run([ours, "-m", fa, fb, "-d", "0", "-t", threads, "-l", os.devnull],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        outs.append(p.stdout)
    assert len(set(outs)) == 1 and len(outs[0]) > 100
    # (the same host sources under AddressSanitizer + UBSan: the arena's bump allocation and the fall-back to malloc)
    asan = os.path.join(ROOT, "tests", "bin", "compairr_oracle_cli_asan")
    if os.path.exists(asan):
        p = subprocess.run([asan, "-m", fa, fb, "-d", "0", "-t", "5", "-l", os.devnull],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()[-3000:]
        assert p.stdout == outs[0]
    ref = os.path.join(ROOT, "oracle", "_ref", "compairr")
    if os.path.exists(ref):
        q = subprocess.run([ref, "-m", fa, fb, "-d", "0", "-t", "4", "-l", os.devnull],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert q.returncode == 0 and q.stdout == outs[0]
