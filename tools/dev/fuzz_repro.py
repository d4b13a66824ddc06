#!/usr/bin/env python3
"""Regenerates case number N of tests/fuzz_gpu.py --seed S and runs it under variations of its tunables.
usage: tools/dev/fuzz_repro.py S N"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle
import fuzz_gpu
from compairr_amd import HipOverlap
seed, N = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for n in range(N + 1):
    a, b, o, tun = fuzz_gpu.random_case(rng)
    shards = int(rng.integers(2, 6)) if (tun.get("variant", -1) != 0 and rng.random() < 0.2) else 1
print("case", N, o, tun, "shards", shards, "n", a.n, b.n, "longest", a.longest, b.longest)
want, ost = _oracle.overlap(a, b, o, threads=8)
want = _oracle.integer_cells(want, o)
def run(tun, shards):
    got, nm, nv = None, 0, 0
    for index in range(shards):
        with HipOverlap(o) as h:
            for k in ["variant"] + [k for k in tun if k != "variant"]:
                if k in tun: h.set_tunable(k, tun[k])
            if shards > 1:
                h.set_tunable("work_shard_count", shards); h.set_tunable("work_shard_index", index)
            h.set_reference(b, a.longest); h.set_queries(a)
            m = h.overlap_matrix(); st = h.stats()
            k = h.get_tunable("class_residues"); c0 = h.get_tunable("class_anchor")
        got = m if got is None else got + m; nm += st.matches; nv += st.variants
    ok = np.array_equal(got, want) and nm == ost.matches and nv == ost.variants
    print("  %-60s shards=%d -> %s matches %d/%d variants %d/%d (K=%d c0=%d)" % (tun, shards, "ok" if ok else "MISMATCH", nm, ost.matches, nv, ost.variants, k, c0))
run(tun, shards)
run(tun, 1)
t = dict(tun); t.pop("deferred_resolve", None); run(t, 1)
t = dict(tun); t.pop("class_anchor", None); run(t, 1)
for var in (2, 1):
    for kk in (1, 2, 3):
        for c0 in range(0, 12):
            t = {"variant": var, "class_residues": kk, "class_anchor": c0}; run(t, 1)

# which pairs are missing (class_anchor 5, K 1)?
t = {"variant": 2, "class_residues": tun.get("class_residues", 1), "class_anchor": 5}
with HipOverlap(o) as h:
    for k in ["variant"] + [k for k in t if k != "variant"]:
        h.set_tunable(k, t[k])
    h.set_reference(b, a.longest); h.set_queries(a)
    gp = h.overlap_pairs()
op = _oracle.pairs(a, b, o)
gs = set(map(tuple, gp.tolist())); os_ = set(map(tuple, op.tolist()))
miss = sorted(os_ - gs); extra = sorted(gs - os_)
print("missing", len(miss), "extra", len(extra))
def seq(s, i):
    return "".join("ACGT"[x] if o.nucleotides else chr(65 + x) for x in s.residues[int(s.offsets[i]):int(s.offsets[i + 1])])
import collections
kinds = collections.Counter()
for q, hh in miss[:4000]:
    sq, sh = seq(a, q), seq(b, hh)
    kinds[(len(sq), len(sh))] += 1
print(sorted(kinds.items()))
for q, hh in miss[:12]:
    print("  q", seq(a, q), "v", a.v_gene[q], a.j_gene[q], " hit", seq(b, hh), "v", b.v_gene[hh], b.j_gene[hh])
