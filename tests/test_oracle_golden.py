"""The oracle is pinned here: host code + oracle must reproduce, byte for byte,
every matrix the REAL reference (oracle/_ref/compairr, compiled from
/root/reference) wrote for tests/golden/ -- including the reference's own
test/expected.tsv (case ref_test_sh) -- and must agree with an independent
brute-force evaluation of the pair definition."""

import os

import numpy as np
import pytest

import _oracle
from compairr_amd import Options, synth
from conftest import (expected_of, expected_pairs_of, load_manifest, run_cli, sorted_pairs,
                      warnings_of)

CASES = load_manifest()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_cli_matches_reference(case, tmp_path):
    log = str(tmp_path / "log.txt")
    pairs = str(tmp_path / "pairs.tsv")
    p = run_cli("tests/bin/compairr_oracle_cli", case, log=log, pairs=pairs)
    if case["exit"] != 0:
        assert p.returncode == case["exit"], p.stderr.decode()
        return
    assert p.returncode == 0, p.stderr.decode()
    assert p.stdout == expected_of(case)
    assert warnings_of(log) == case["warnings"]
    if case.get("pairs"):
        assert sorted_pairs(pairs) == expected_pairs_of(case)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_host_code_under_sanitizers(case, tmp_path):
    """The host program (option table, threaded AIRR-TSV reader, matrix / pairs /
    cluster output) + the oracle backend built with -fsanitize=address,undefined
    (make asan): every golden case once more, and no report from either sanitizer."""
    log = str(tmp_path / "log.txt")
    pairs = str(tmp_path / "pairs.tsv")
    p = run_cli("tests/bin/compairr_oracle_cli_asan", case, log=log, pairs=pairs, timeout=300)
    err = p.stderr.decode(errors="replace")
    assert "AddressSanitizer" not in err and "runtime error" not in err and "LeakSanitizer" not in err, err[-3000:]
    assert p.returncode == case["exit"], err[-2000:]
    if case["exit"] == 0:
        assert p.stdout == expected_of(case)


def test_reference_own_golden_file():
    """test/test.sh:9-11 of the reference: -m seta.tsv setb.tsv -d 1 -i."""
    case = next(c for c in CASES if c["name"] == "ref_test_sh")
    assert expected_of(case) == b"#\tB1\tB2\nA1\t0\t7\nA2\t45\t0\n"


def _opts(seed, d, indels, A):
    return Options(differences=d, indels=indels, nucleotides=(A == 4), n_v_genes=2,
                   n_j_genes=2, ignore_genes=(seed % 3 == 0),
                   ignore_counts=(seed % 5 == 0),
                   score=["product", "min", "max", "mean"][seed % 4])


@pytest.mark.parametrize("d,indels", [(0, False), (1, False), (1, True), (2, False)])
def test_oracle_equals_bruteforce(d, indels):
    for seed in range(60):
        A = 4 if seed % 2 else 20
        x = synth.tiny_set(40, seed, alphabet_size=A, letters=2 + seed % 2)
        y = synth.tiny_set(35, seed + 1000, alphabet_size=A, letters=2 + seed % 2)
        o = _opts(seed, d, indels, A)
        m, st = _oracle.overlap(x, y, o)
        assert np.array_equal(m, _oracle.bruteforce(x, y, o))
        m, _ = _oracle.overlap(x, x, o, threads=3)     # one-file mode, threaded
        assert np.array_equal(m, _oracle.bruteforce(x, x, o))


def test_oracle_thread_invariance_and_stats():
    a = synth.make_set(20000, 1, prefix="A", pool_size=3000)
    b = synth.make_set(20000, 2, prefix="B", pool_size=3000)
    o = Options(differences=1, indels=True, n_v_genes=synth.N_V, n_j_genes=synth.N_J)
    m1, s1 = _oracle.overlap(a, b, o, threads=1)
    m4, s4 = _oracle.overlap(a, b, o, threads=4)
    assert np.array_equal(m1, m4)
    assert s1.variants == s4.variants and s1.matches == s4.matches > 0
    assert s1.bloom_positive >= s1.hash_equal >= s1.matches


def test_oracle_empty_and_ragged():
    e = synth.make_set(0, 1)
    a = synth.tiny_set(10, 3)
    o = Options(differences=1, n_v_genes=2, n_j_genes=2)
    m, _ = _oracle.overlap(a, a, o)
    assert m.shape == (a.n_repertoires, a.n_repertoires)
    assert e.n == 0 and e.n_repertoires == 0


# ---- the oracle port against the reference's matrices at BASELINE sizes ----
# (tests/golden/full_size.json, written by tests/golden/make_full_size.py from oracle/_ref)

@pytest.mark.parametrize("name", ["cfg2", "cfg5_sub"])
def test_oracle_port_matches_reference_at_full_size(name):
    """cfg2 (1M x 1M aa, d = 0) and the cfg5 sub-shape (200k x 10M nucleotides, d = 2, -g): the C
    restatement gives the matrix the reference binary printed, digit for digit."""
    import _full_size
    import _oracle
    from compairr_amd import Options
    w = _full_size.load()[name]
    a, b = _full_size.sets_of(w)
    opt = Options(**_full_size.options_of(w))
    m, st = _oracle.overlap(a, b, opt, threads=8)
    got = _oracle.integer_cells(m, opt)
    assert _full_size.mismatch(w, got) is None, _full_size.mismatch(w, got)
    dups = {int(l.split()[-1]): int(l.split()[1]) for l in w["warnings"]}
    assert st.dup_set2 == dups[2] and st.dup_set1 == dups.get(1, 0)


def test_every_pinned_workload_is_asserted_on_the_gpu():
    """Every workload of tests/golden/full_size.json is named by a `-m gpu` test that holds the HIP
    path against it (test_full_size_matches_reference's list or a properties test): a fixture nobody
    reads pins nothing."""
    import _full_size
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_gpu_parity.py")).read()
    for name in _full_size.load():
        assert '"%s"' % name in src, name
