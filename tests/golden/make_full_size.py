#!/usr/bin/env python3
"""Reference-derived expected values at BASELINE.json's FULL sizes -> tests/golden/full_size.json.

For every workload below: generate the two synthetic sets with the seeded generator
(compairr_amd.synth -- the very calls bench.py and tests/test_gpu_parity.py make), write them
as AIRR TSV, run the REAL reference (oracle/_ref/compairr, compiled from /root/reference by
oracle/Makefile) on the files, and record
  * the generator arguments and the md5 of each input file,
  * the md5 of the matrix file the reference wrote (overlap.cc:944-1039),
  * the cells as the reference printed them (%.10lg), rows/columns in repertoire-NUMBER order
    (first appearance, overlap.cc:222) -- what cmpr_overlap_matrix returns,
  * synth.checksum of those cells (valid as an integer checksum when every cell < 10^10,
    `exact` says so), and the reference's own 'Analysing:' time.
The GPU tests (`-m gpu`) and bench.py compare their full-size matrices with these values; the
GPU box never runs this script (it needs /root/reference for oracle/_ref).

Run from the repository root in the build container:
    python tests/golden/make_full_size.py [name ...]
"""

import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from compairr_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "compairr")
OUT = os.path.join(HERE, "full_size.json")

M = 1_000_000
# name -> (set 1 generator args | None for one-file mode, set 2 generator args, reference argv)
WORKLOADS = {
    # BASELINE configs[1]
    "cfg2": (dict(n=1 * M, seed=1, prefix="A", pool_size=M // 4),
             dict(n=1 * M, seed=2, prefix="B", pool_size=M // 4), ["-d", "0"]),
    # BASELINE configs[2]: the bench default
    "cfg3": (dict(n=10 * M, seed=1, prefix="A", pool_size=10 * M // 4),
             dict(n=10 * M, seed=2, prefix="B", pool_size=10 * M // 4), ["-d", "1"]),
    # BASELINE configs[3]
    "cfg4": (dict(n=10 * M, seed=1, prefix="A", pool_size=10 * M // 4),
             dict(n=10 * M, seed=2, prefix="B", pool_size=10 * M // 4), ["-d", "1", "-i"]),
    # the reference's own published benchmark shape is self-vs-self (README.md:726-755)
    "self10m": (None, dict(n=10 * M, seed=2, prefix="B", pool_size=10 * M // 4), ["-d", "1"]),
    # the robustness workload: configs[2] / [3] under the cdr3 law of compairr_amd/synth.py (conserved
    # ends, skewed composition, Zipf clone sizes)
    "cfg3_cdr3": (dict(n=10 * M, seed=1, prefix="A", pool_size=10 * M // 4, law="cdr3"),
                  dict(n=10 * M, seed=2, prefix="B", pool_size=10 * M // 4, law="cdr3"), ["-d", "1"]),
    "cfg4_cdr3": (dict(n=10 * M, seed=1, prefix="A", pool_size=10 * M // 4, law="cdr3"),
                  dict(n=10 * M, seed=2, prefix="B", pool_size=10 * M // 4, law="cdr3"), ["-d", "1", "-i"]),
    # the shape of the reference's published benchmark (README.md:726-755: 24 205 557 sequences in 120
    # repertoires against themselves) on the cdr3 law: d = 0, 1, 1 with indels (d = 2 takes the reference hours)
    "pub_d0": (None, dict(n=24_200_000, seed=2, prefix="B", pool_size=24_200_000 // 4, law="cdr3", n_repertoires=120),
               ["-d", "0"]),
    "pub_d1": (None, dict(n=24_200_000, seed=2, prefix="B", pool_size=24_200_000 // 4, law="cdr3", n_repertoires=120),
               ["-d", "1"]),
    "pub_d1i": (None, dict(n=24_200_000, seed=2, prefix="B", pool_size=24_200_000 // 4, law="cdr3", n_repertoires=120),
                ["-d", "1", "-i"]),
    # d = 2 on the published shape: the whole 24.2M-sequence set as set 2 (its filter, its pages, its class parts),
    # a 500 000-sequence set 1 the reference finishes in a minute on 8 cores
    "pub_d2_sub": (dict(n=500_000, seed=1, prefix="A", pool_size=24_200_000 // 4, law="cdr3", n_repertoires=120),
                   dict(n=24_200_000, seed=2, prefix="B", pool_size=24_200_000 // 4, law="cdr3", n_repertoires=120),
                   ["-d", "2"]),
    # BASELINE configs[4], a sub-shape the reference finishes in minutes on 8 cores
    "cfg5_sub": (dict(n=200_000, seed=3, prefix="A", pool_size=10 * M // 4, nucleotides=True),
                 dict(n=10 * M, seed=4, prefix="B", pool_size=10 * M // 4, nucleotides=True),
                 ["-d", "2", "-n", "-g"]),
}


def md5_of(path):
    h = hashlib.md5()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def run(name):
    g1, g2, argv = WORKLOADS[name]
    nt = bool(g2.get("nucleotides"))
    t0 = time.time()
    b = synth.make_set(**g2)
    a = b if g1 is None else synth.make_set(**g1)
    with tempfile.TemporaryDirectory(prefix="cmpr_full_", dir="/tmp") as tmp:
        fb = os.path.join(tmp, "set2.tsv")
        b.write_tsv_fast(fb, nt)
        fa = fb
        if g1 is not None:
            fa = os.path.join(tmp, "set1.tsv")
            a.write_tsv_fast(fa, nt)
        log, out = os.path.join(tmp, "log"), os.path.join(tmp, "out.tsv")
        threads = min(os.cpu_count() or 1, 256)
        files = [fa] if g1 is None else [fa, fb]
        t1 = time.time()
        p = subprocess.run([REF, "-m"] + files + argv + ["-t", str(threads), "-l", log, "-o", out],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        wall = time.time() - t1
        if p.returncode != 0:
            raise SystemExit("%s: the reference failed: %s" % (name, p.stderr.decode()[-500:]))
        text = open(log).read()
        sec = float(re.search(r"Analysing:\s+100% \(([0-9.]+)s\)", text).group(1))
        rows = [l.rstrip("\n").split("\t") for l in open(out)]
        cols = rows[0][1:]
        cell = {(r[0], cid): x for r in rows[1:] for cid, x in zip(cols, r[1:])}
        cells = [[cell[(ra, cb)] for cb in b.repertoire_ids] for ra in a.repertoire_ids]
        vals = np.array([[float(x) for x in r] for r in cells])
        exact = bool((vals < 1e10).all() and (vals == np.floor(vals)).all())
        rec = {
            "name": name,
            "argv": argv,
            "one_file_mode": g1 is None,
            "set1": g1, "set2": g2,
            "input_md5": {"set1": md5_of(fa), "set2": md5_of(fb)},
            "matrix_tsv_md5": md5_of(out),
            "cells_printed": cells,
            "exact": exact,
            "matrix_checksum": synth.checksum(vals.astype(np.uint64)) if exact else None,
            "warnings": [l for l in text.splitlines() if l.startswith("Warning:")],
            "reference": {"version": "CompAIRR 1.13.0 (oracle/_ref)", "threads": threads,
                          "analysing_seconds": sec, "wall_seconds": round(wall, 1)},
        }
    print("%s: Analysing %.1f s, wall %.1f s (+%.1f s generate/write), checksum %s" %
          (name, sec, wall, t1 - t0, rec["matrix_checksum"]), flush=True)
    return rec


def main():
    names = sys.argv[1:] or list(WORKLOADS)
    have = {}
    if os.path.exists(OUT):
        have = {r["name"]: r for r in json.load(open(OUT))["workloads"]}
    for n in names:
        have[n] = run(n)
        doc = {"made_by": "tests/golden/make_full_size.py",
               "note": "expected values from the reference binary; cells in repertoire-number order as "
                       "printed (%.10lg)",
               "workloads": [have[k] for k in WORKLOADS if k in have]}
        with open(OUT, "w") as fh:
            json.dump(doc, fh, indent=1)
            fh.write("\n")


if __name__ == "__main__":
    main()
