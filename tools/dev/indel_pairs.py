#!/usr/bin/env python3
"""dev check of the -i pair rows: a few random sets under chosen tunables against the oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle
from compairr_amd import HipOverlap, Options, synth
FULL = dict(n_v_genes=synth.N_V, n_j_genes=synth.N_J)
layouts = {
  "k0": {"variant": 2, "class_residues": 0},
  "tiny_k0": {"variant": 2, "slice_words_log2": 4, "class_residues": 0},
  "rows": {"variant": 2},
  "tiny_k3": {"variant": 2, "slice_words_log2": 2, "class_residues": 3, "chunk_tiles": 2, "heavy_threshold": 0},
  "tiny_k1": {"variant": 2, "slice_words_log2": 3, "class_residues": 1, "chunk_tiles": 3, "heavy_threshold": 0},
  "tiny_mixed": {"variant": 2, "slice_words_log2": 2, "class_residues": 2, "heavy_threshold": 2, "waves_per_block": 4},
  "anchor3": {"variant": 2, "slice_words_log2": 2, "class_residues": 2, "heavy_threshold": 0, "class_anchor": 3},
  "anchor5_k3": {"variant": 2, "slice_words_log2": 3, "class_residues": 3, "heavy_threshold": 0, "class_anchor": 5},
}
want = sys.argv[1:] or list(layouts)
bad = 0
for name in want:
    tun = layouts[name]
    for nt in (False, True):
        for seed in (1, 2):
            a = synth.make_set(3000, 10 + seed, prefix="A", pool_size=600, nucleotides=nt)
            b = synth.make_set(3000, 20 + seed, prefix="B", pool_size=600, nucleotides=nt)
            o = Options(differences=1, indels=True, nucleotides=nt, **FULL)
            w, ost = _oracle.overlap(a, b, o, threads=8)
            w = _oracle.integer_cells(w, o)
            try:
                with HipOverlap(o) as h:
                    for k, v in tun.items():
                        h.set_tunable(k, v)
                    h.set_reference(b, a.longest)
                    h.set_queries(a)
                    m = h.overlap_matrix(); st = h.stats()
                ok = np.array_equal(m, w) and st.variants == ost.variants and st.matches == ost.matches
                print("%-12s nt=%d seed=%d %s  variants %d/%d matches %d/%d diffcells %d" % (name, nt, seed, "ok" if ok else "MISMATCH", st.variants, ost.variants, st.matches, ost.matches, int((m != w).sum())))
                bad += 0 if ok else 1
            except Exception as e:
                print("%-12s nt=%d seed=%d ERROR %s" % (name, nt, seed, e)); bad += 1
print("bad:", bad)
