"""The synthetic generator is part of the measurement contract: same seed, same
bytes, on every box."""

import hashlib

import numpy as np

from compairr_amd import synth


def digest(s):
    h = hashlib.md5()
    for a in (s.residues, s.offsets, s.v_gene, s.j_gene, s.repertoire, s.count):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def test_deterministic_and_seed_sensitive():
    a = synth.make_set(5000, 1)
    assert digest(a) == digest(synth.make_set(5000, 1))
    assert digest(a) != digest(synth.make_set(5000, 2))
    assert a.n_repertoires == 16 and a.n == 5000


def test_shape_of_the_law():
    a = synth.make_set(50000, 1)
    L = a.lengths
    assert 8 <= L.min() and L.max() <= 24 and abs(L.mean() - 15) < 0.2
    assert a.residues.max() < 20 and a.count.min() >= 1 and a.count.max() <= 99
    n = synth.make_set(20000, 3, nucleotides=True)
    assert 24 <= n.lengths.min() and n.lengths.max() <= 72 and n.residues.max() < 4
    # repertoire numbers follow first appearance
    first = [int(np.flatnonzero(a.repertoire == r)[0]) for r in range(a.n_repertoires)]
    assert first == sorted(first)


def test_subset_roundtrip():
    a = synth.make_set(2000, 5)
    b = a.subset(slice(100, 900))
    assert b.n == 800 and b.sequence(0) == a.sequence(100) and b.sequence(799) == a.sequence(899)
