"""Reader of tests/golden/full_size.json -- expected matrices of the BASELINE-size workloads,
written by tests/golden/make_full_size.py from the REAL reference binary (data only: the
reference is not needed, and not present, where this is read).  TEST INFRASTRUCTURE: imported by tests/ and by bench.py's parity check."""

from __future__ import annotations

import json
import os
from typing import Optional

import numpy as np

from compairr_amd import synth

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full_size.json")


def load() -> dict:
    with open(PATH) as fh:
        return {w["name"]: w for w in json.load(fh)["workloads"]}


def sets_of(w: dict):
    """(queries, references) of a recorded workload, regenerated from its seeds."""
    b = synth.make_set(**w["set2"])
    a = b if w["one_file_mode"] else synth.make_set(**w["set1"])
    return a, b


def options_of(w: dict) -> dict:
    """keyword arguments of compairr_amd.Options for the workload's reference argv"""
    argv = w["argv"]
    return dict(differences=int(argv[argv.index("-d") + 1]), indels="-i" in argv,
                nucleotides="-n" in argv, ignore_genes="-g" in argv,
                n_v_genes=synth.N_V, n_j_genes=synth.N_J)


def mismatch(w: dict, matrix: np.ndarray) -> Optional[str]:
    """None when `matrix` (integer cells, repertoire-number order) is what the reference
    printed -- digit for digit through its own %.10lg, and bit for bit (checksum) where every
    printed cell is an exact integer; else a description of the first difference."""
    want = w["cells_printed"]
    if matrix.shape != (len(want), len(want[0]) if want else 0):
        return "shape %s, reference %dx%d" % (matrix.shape, len(want), len(want[0]) if want else 0)
    for i, row in enumerate(want):
        for j, x in enumerate(row):
            got = "%.10g" % float(matrix[i, j])
            if got != x:
                return "cell [%d][%d]: %s, reference %s" % (i, j, got, x)
    if w["exact"] and synth.checksum(matrix) != w["matrix_checksum"]:
        return "checksum %s, reference %s" % (synth.checksum(matrix), w["matrix_checksum"])
    return None


def by_bench_args(refs: int, queries: int, differences: int, indels: bool, nucleotides: bool,
                  ignore_genes: bool, self_cmp: bool, law: str = "uniform", repertoires: int = 16) -> Optional[dict]:
    """the recorded workload bench.py's arguments name, if any (bench.py seeds: set 1 = 1,
    set 2 = 2, pool = refs // 4)"""
    for w in load().values():
        o = options_of(w)
        if (o["differences"], o["indels"], o["nucleotides"], o["ignore_genes"]) != \
                (differences, indels, nucleotides, ignore_genes):
            continue
        extra = dict(**({"nucleotides": True} if nucleotides else {}), **({"law": law} if law != "uniform" else {}),
                     **({"n_repertoires": repertoires} if repertoires != 16 else {}))
        if w["set2"] != dict(n=refs, seed=2, prefix="B", pool_size=refs // 4, **extra):
            continue
        if self_cmp != w["one_file_mode"]:
            continue
        if not self_cmp and w["set1"] != dict(n=queries, seed=1, prefix="A", pool_size=refs // 4, **extra):
            continue
        return w
    return None
