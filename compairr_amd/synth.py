"""Seeded synthetic repertoire sets for the BASELINE.json configurations
(SURVEY.md section 8d).

Law (fixed; changing it changes every recorded checksum):
  * lengths  ~ round(Normal(mean, sd)) clipped to [lo, hi]
      aa: mean 15, sd 2.0, [8, 24];   nt: mean 45, sd 6.0, [24, 72]
  * residues uniform over the alphabet
  * V genes: 60 names, P(k) ~ 1/(k+1); J genes: 13 names, P(k) ~ 1/(k+1)
  * repertoires: `n_repertoires` (16), uniform; duplicate_count uniform 1..99
  * a shared "public" pool of max(1000, n_pool) clonotypes (sequence + V + J)
    is generated from `pool_seed` alone; every set draws 30 % of its sequences
    from it, and half of those are mutated once (60 % substitution, 20 %
    deletion, 20 % insertion), so that d = 0 / 1 / 2 and --indels all give
    non-trivial matrices between two sets built from the same pool.
Everything is numpy-vectorised: 10M sequences take a few seconds.

A second law, `law="cdr3"` (amino acids), is closer to what TCR-beta CDR3 repertoires look
like -- the robustness workload, not a BASELINE configuration:
  * a conserved 4-residue start (C A S S, each kept with probability 0.92, else a residue of
    the background composition) and a 3-residue end that is a function of the J gene
    (13 motifs, kept with probability 0.9 per residue);
  * the positions in between drawn from a skewed background composition (G S A T E Q D N L
    R Y P frequent; W C M H I K rare) instead of uniformly;
  * clone sizes (duplicate_count) Zipf-distributed: P(c) ~ c^-2, capped at `max_count`;
  * lengths, genes, repertoires, the public pool and its mutations as above (mutations draw
    replacement residues from the background composition).
Neighbourhoods at d = 1 are far denser than under the uniform law, the residue entropy at the
first and last positions is low (the class positions of the sliced filter have to be found in
the middle), and (length, V, J) classes are heavy.
"""

from __future__ import annotations

import numpy as np

from .sets import AA, NT, RepertoireSet

N_V, N_J = 60, 13


def _zipf_choice(rng, k: int, n: int) -> np.ndarray:
    p = 1.0 / np.arange(1, k + 1)
    p /= p.sum()
    return rng.choice(k, size=n, p=p).astype(np.uint32)


def _lengths(rng, n: int, nucleotides: bool) -> np.ndarray:
    mean, sd, lo, hi = (45.0, 6.0, 24, 72) if nucleotides else (15.0, 2.0, 8, 24)
    return np.clip(np.rint(rng.normal(mean, sd, size=n)), lo, hi).astype(np.int64)


# background composition of the cdr3 law, in the order of sets.AA ("ACDEFGHIKLMNPQRSTVWY")
_CDR3_BACKGROUND = np.array([7.0, 0.6, 4.5, 6.0, 3.0, 11.0, 1.5, 1.5, 1.5, 5.0,
                             0.8, 4.5, 4.0, 6.0, 4.5, 11.0, 7.0, 3.0, 1.0, 5.0])
_CDR3_START = [AA.index(c) for c in "CASS"]


def _j_motifs() -> np.ndarray:
    """the 3-residue end of each of the N_J genes (fixed, seedless)"""
    rng = np.random.default_rng(0x4A4D4F54)
    ends = ["QYF", "AFF", "LFF", "HFG", "QHF", "IYF", "TFG", "LHF", "YTF", "QFF", "AYF", "KFF", "QYV"]
    out = np.array([[AA.index(c) for c in e] for e in ends[:N_J]], dtype=np.uint8)
    assert out.shape == (N_J, 3)
    del rng
    return out


def _cdr3_residues(rng, lens: np.ndarray, j: np.ndarray, W: int) -> np.ndarray:
    """padded residue matrix under the cdr3 law (see the module docstring)"""
    n = len(lens)
    p = _CDR3_BACKGROUND / _CDR3_BACKGROUND.sum()
    pad = rng.choice(20, size=(n, W), p=p).astype(np.uint8)
    keep = rng.random((n, 4)) < 0.92
    for k in range(4):
        pad[keep[:, k], k] = _CDR3_START[k]
    motifs = _j_motifs()[j]                                  # (n, 3)
    keep = rng.random((n, 3)) < 0.9
    rows = np.arange(n)
    for k in range(3):
        pos = lens - 3 + k                                   # lengths are >= 8
        sel = keep[:, k]
        pad[rows[sel], pos[sel]] = motifs[sel, k]
    return pad


def _pool(pool_seed: int, size: int, nucleotides: bool, law: str = "uniform"):
    rng = np.random.default_rng([pool_seed, 0x9E3779B9])
    A = 4 if nucleotides else 20
    hi = 72 if nucleotides else 24
    lens = _lengths(rng, size, nucleotides)
    if law == "cdr3":
        v, j = _zipf_choice(rng, N_V, size), _zipf_choice(rng, N_J, size)
        return _cdr3_residues(rng, lens, j, hi + 1), lens, v, j
    pad = rng.integers(0, A, size=(size, hi + 1), dtype=np.uint8)
    return pad, lens, _zipf_choice(rng, N_V, size), _zipf_choice(rng, N_J, size)


def make_set(n: int, seed: int, *, nucleotides: bool = False, pool_seed: int = 12345,
             pool_size: int = 0, n_repertoires: int = 16, prefix: str = "R",
             public_fraction: float = 0.3, max_count: int = 99, law: str = "uniform") -> RepertoireSet:
    assert law in ("uniform", "cdr3") and not (law == "cdr3" and nucleotides)
    rng = np.random.default_rng([seed, 0x51ED270B] if law == "uniform" else [seed, 0x51ED270B, 0xCD23])
    A = 4 if nucleotides else 20
    hi = 72 if nucleotides else 24
    W = hi + 1                                     # room for one insertion
    pool_size = max(1000, pool_size or n // 4)
    ppad, plen, pv, pj = _pool(pool_seed, pool_size, nucleotides, law)

    lens = _lengths(rng, n, nucleotides)
    if law == "cdr3":
        v = _zipf_choice(rng, N_V, n)
        j = _zipf_choice(rng, N_J, n)
        pad = _cdr3_residues(rng, lens, j, W)
        bg = _CDR3_BACKGROUND / _CDR3_BACKGROUND.sum()
    else:
        pad = rng.integers(0, A, size=(n, W), dtype=np.uint8)
        v = _zipf_choice(rng, N_V, n)
        j = _zipf_choice(rng, N_J, n)

    pub = np.flatnonzero(rng.random(n) < public_fraction)
    pick = rng.integers(0, pool_size, size=len(pub))
    pad[pub] = ppad[pick]
    lens[pub] = plen[pick]
    v[pub] = pv[pick]
    j[pub] = pj[pick]

    mut = pub[rng.random(len(pub)) < 0.5]
    kind = rng.random(len(mut))
    cols = np.arange(W, dtype=np.int64)[None, :]

    sub = mut[kind < 0.6]
    if len(sub):
        pos = (rng.integers(0, 1 << 30, size=len(sub)) % lens[sub])
        if law == "cdr3":                            # another residue of the background composition
            new = rng.choice(20, size=len(sub), p=bg).astype(np.uint8)
            same = new == pad[sub, pos]
            new[same] = (new[same] + 1) % A
            pad[sub, pos] = new
        else:
            delta = rng.integers(1, A, size=len(sub)).astype(np.uint8)
            pad[sub, pos] = (pad[sub, pos] + delta) % A

    dele = mut[(kind >= 0.6) & (kind < 0.8)]
    dele = dele[lens[dele] > (24 if nucleotides else 8)]
    if len(dele):
        pos = (rng.integers(0, 1 << 30, size=len(dele)) % lens[dele])[:, None]
        src = np.minimum(cols + (cols >= pos), W - 1)
        pad[dele] = np.take_along_axis(pad[dele], src, axis=1)
        lens[dele] -= 1

    ins = mut[kind >= 0.8]
    ins = ins[lens[ins] < hi]
    if len(ins):
        pos = (rng.integers(0, 1 << 30, size=len(ins)) % (lens[ins] + 1))
        src = np.maximum(cols - (cols > pos[:, None]), 0)
        block = np.take_along_axis(pad[ins], src, axis=1)
        block[np.arange(len(ins)), pos] = (rng.choice(20, size=len(ins), p=bg).astype(np.uint8)
                                           if law == "cdr3" else rng.integers(0, A, size=len(ins), dtype=np.uint8))
        pad[ins] = block
        lens[ins] += 1

    mask = cols < lens[:, None]
    residues = pad[mask]
    offsets = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])

    rep = rng.integers(0, n_repertoires, size=n, dtype=np.uint32)
    # repertoire numbers must follow first appearance (db.cc:510-520)
    first = np.full(n_repertoires, n, dtype=np.int64)
    np.minimum.at(first, rep, np.arange(n, dtype=np.int64))
    order = np.argsort(first, kind="stable")
    present = int((first < n).sum())
    renum = np.empty(n_repertoires, dtype=np.uint32)
    renum[order] = np.arange(n_repertoires, dtype=np.uint32)
    rep = renum[rep]
    ids = ["%s%02d" % (prefix, int(k) + 1) for k in order[:present]]

    if law == "cdr3":                                # clone sizes: P(c) ~ c^-2, capped
        count = np.minimum(rng.zipf(2.0, size=n), max_count).astype(np.uint64)
    else:
        count = rng.integers(1, max_count + 1, size=n, dtype=np.uint64)
    return RepertoireSet(residues, offsets, v, j, rep, count, ids,
                         ["TRBV%02d" % (k + 1) for k in range(N_V)],
                         ["TRBJ%02d" % (k + 1) for k in range(N_J)],
                         NT if nucleotides else AA)


def tiny_set(n: int, seed: int, *, alphabet_size: int = 20, letters: int = 2,
             min_len: int = 1, max_len: int = 6, n_repertoires: int = 3,
             n_v: int = 2, n_j: int = 2, max_count: int = 9,
             prefix: str = "T") -> RepertoireSet:
    """Adversarial small sets: few letters (homopolymer runs, many neighbours and
    exact duplicates), very short sequences, few genes."""
    rng = np.random.default_rng([seed, 0x7F4A7C15])
    lens = rng.integers(min_len, max_len + 1, size=n)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    letters = min(letters, alphabet_size)
    codes = rng.choice(alphabet_size, size=letters, replace=False)
    residues = codes[rng.integers(0, letters, size=int(offsets[-1]))].astype(np.uint8)
    rep = rng.integers(0, n_repertoires, size=n, dtype=np.uint32)
    _, first_idx = np.unique(rep, return_index=True)
    order = np.argsort(first_idx)
    uniq = np.unique(rep)[order]
    renum = np.zeros(n_repertoires, dtype=np.uint32)
    renum[uniq] = np.arange(len(uniq), dtype=np.uint32)
    rep = renum[rep]
    ids = ["%s%d" % (prefix, int(k) + 1) for k in uniq]
    return RepertoireSet(residues, offsets,
                         rng.integers(0, n_v, size=n, dtype=np.uint32),
                         rng.integers(0, n_j, size=n, dtype=np.uint32),
                         rep, rng.integers(1, max_count + 1, size=n, dtype=np.uint64),
                         ids, ["V%d" % k for k in range(n_v)],
                         ["J%d" % k for k in range(n_j)],
                         NT if alphabet_size == 4 else AA)


def checksum(m: np.ndarray) -> str:
    """Order-sensitive digest of an integer matrix (recorded per configuration)."""
    import hashlib
    return hashlib.md5(np.ascontiguousarray(m, dtype=np.uint64).tobytes()).hexdigest()
