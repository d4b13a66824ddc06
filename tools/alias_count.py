#!/usr/bin/env python3
"""How many of the pair-row filter's positives on skewed data are ALIASES of the sum / difference coding?

kernels_rows.h files the entry of (t, pair p) under the hash of t with both positions blanked and codes the
two residues (a, b) as bit (a_k + a + b) mod 32 and bit (a_k + a - b) mod 32.  (a, b) and (a +- 16, b +- 16)
set the same bits: "q with v at p" is answered yes when set 2 holds q with v at p (a hit) -- or q with
v +- 16 at p AND rb +- 16 at p + 1 (an alias: a distance-2 neighbour of q, a false positive of the filter).
With the residue codes in alphabet order (ACDEFGHIKLMNPQRSTVWY = 0..19) the codes that have a partner are
0..3 and 16..19: A, C, D, E <-> T, V, W, Y -- six of the commonest residues of a CDR3.

This script counts, exactly (set membership, no filter), for a sample of queries of the 10M x 10M cdr3-law
workload at d = 1 (substitutions): the variant tests that are hits, and the ones that are alias-only -- under
the alphabet-order codes and under codes ranked by residue frequency (the twelve commonest residues on the
twelve codes 4..15 that have no partner).  CPU only; uses compairr_amd.synth.  usage: tools/alias_count.py [sample]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from compairr_amd import synth  # noqa: E402

N = 10_000_000
SAMPLE = int(sys.argv[1]) if len(sys.argv) > 1 else 30000


def keys_of(s):
    """bytes(v, j, residues...) of every sequence"""
    res, off = s.residues, s.offsets.astype(np.int64)
    v, j = s.v_gene.astype(np.uint8), s.j_gene.astype(np.uint8)
    raw = res.tobytes()
    return [bytes((v[i], j[i])) + raw[off[i]:off[i + 1]] for i in range(s.n)]


def count(queries, S, code):
    """(variant tests, hits, alias-only positives) over the pairs (p, p + 1), p even, of the queries"""
    inv = np.argsort(code)                       # code -> residue
    partner = {}                                 # residue -> the residue whose code is +- 16 away, if any
    for r in range(20):
        for c in (code[r] + 16, code[r] - 16):
            if 0 <= c < 20:
                partner[r] = int(inv[c])
    tests = hits = alias = 0
    for k in queries:
        head, q = k[:2], bytearray(k[2:])
        L = len(q)
        for p in range(0, L, 2):
            ra = q[p]
            rb = q[p + 1] if p + 1 < L else None
            for pos, other in ((p, p + 1), (p + 1, p)):
                if pos >= L:
                    continue
                own = q[pos]
                for v in range(20):
                    if v == own:
                        continue
                    tests += 1
                    q[pos] = v
                    hit = head + bytes(q) in S
                    al = False
                    if not hit and other < L and v in partner and q[other] in partner:
                        keep = q[other]
                        q[pos] = partner[v]
                        q[other] = partner[keep]
                        al = head + bytes(q) in S
                        q[other] = keep
                    q[pos] = own
                    hits += hit
                    alias += al
    return tests, hits, alias


def count_shared_word(queries, S):
    """The same question with EVERY entry that shares the query's pair-blanked hash: set-2 sequences that equal
    q outside the pair (p, p + 1) set sum bit a + b and difference bit a - b under the very a_k of the query, and a
    test passes when ITS sum bit and ITS difference bit are set -- by one entry (a hit, or the +- 16 alias) or by
    two different ones.  (tests, hits, false positives from entries of the query's own word-and-key)."""
    tests = hits = false = false4 = 0
    for k in queries:
        head, q = k[:2], bytearray(k[2:])
        L = len(q)
        for p in range(0, L - 1, 2):             # (a pair that hangs over the end: one position, no combination)
            ra, rb = q[p], q[p + 1]
            ent = []
            for a in range(20):
                q[p] = a
                for b in range(20):
                    q[p + 1] = b
                    if head + bytes(q) in S:
                        ent.append((a, b))
            q[p], q[p + 1] = ra, rb
            if not ent:
                tests += 38
                continue
            sums = {(a + b) & 31 for a, b in ent}
            diffs = {(a - b) & 31 for a, b in ent}
            firsts = {a for a, b in ent}
            seconds = {b for a, b in ent}
            es = set(ent)
            for v in range(20):
                if v != ra:
                    tests += 1
                    pos = ((v + rb) & 31) in sums and ((v - rb) & 31) in diffs
                    hit = (v, rb) in es
                    hits += hit
                    false += pos and not hit
                    # FOUR forms, two dwords each: sum, difference, the first residue alone, the second alone
                    false4 += pos and not hit and v in firsts and rb in seconds
                if v != rb:
                    tests += 1
                    pos = ((ra + v) & 31) in sums and ((ra - v) & 31) in diffs
                    hit = (ra, v) in es
                    hits += hit
                    false += pos and not hit
                    false4 += pos and not hit and ra in firsts and v in seconds
    return tests, hits, false, false4


def main():
    t0 = time.time()
    a = synth.make_set(N, 1, prefix="A", pool_size=N // 4, law="cdr3")
    b = synth.make_set(N, 2, prefix="B", pool_size=N // 4, law="cdr3")
    print("generated in %.0f s" % (time.time() - t0), flush=True)
    S = set(keys_of(b))
    print("set 2: %d distinct (V, J, sequence) of %d, %.0f s" % (len(S), b.n, time.time() - t0), flush=True)
    rng = np.random.default_rng(7)
    pick = rng.choice(a.n, size=SAMPLE, replace=False)
    ka = keys_of(a.subset(np.sort(pick)))
    freq = np.bincount(b.residues, minlength=20).astype(np.float64)
    order = np.argsort(-freq)                    # commonest first
    alpha = np.arange(20)
    ranked = np.empty(20, dtype=np.int64)
    safe = list(range(4, 16))
    unsafe = [0, 1, 2, 3, 16, 17, 18, 19]
    for rank, r in enumerate(order):
        ranked[r] = safe[rank] if rank < 12 else unsafe[rank - 12]
    print("residue frequencies (alphabet order):", " ".join("%s %.3f" % (c, f) for c, f in zip(synth.AA if hasattr(synth, "AA") else "ACDEFGHIKLMNPQRSTVWY", freq / freq.sum())))
    for name, code in (("alphabet-order codes", alpha), ("frequency-ranked codes", ranked)):
        tests, hits, alias = count(ka, S, code)
        scale = a.n / SAMPLE
        print("%s: %d queries, %d variant tests, %d hits, %d alias-only positives = %.1f %% of (hits + aliases); "
              "per 10M queries: %.2e hits, %.2e aliases" % (name, SAMPLE, tests, hits, alias,
                                                            100.0 * alias / max(1, hits + alias), hits * scale, alias * scale), flush=True)
    sub = ka[:max(1, SAMPLE // 3)]
    tests, hits, false, false4 = count_shared_word(sub, S)
    scale = a.n / len(sub)
    print("every entry under the query's own blanked hash (alphabet-order codes): %d queries, %d tests, %d hits, %d false "
          "positives (one entry's sum bit with another's difference bit, or the +- 16 alias) = %.1f %% of the positives; "
          "per 10M queries: %.2e hits, %.2e false" % (len(sub), tests, hits, false, 100.0 * false / max(1, hits + false),
                                                      hits * scale, false * scale), flush=True)
    print("... with FOUR forms of two dwords each (a + b, a - b, a alone, b alone; a test still looks at eight bits): "
          "%d false positives = %.1f %% of the positives; per 10M queries: %.2e" %
          (false4, 100.0 * false4 / max(1, hits + false4), false4 * scale), flush=True)
    print("total %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
