"""The sum / difference coding of the pair rows (compairr_amd/csrc/kernels_rows.h: pair_entry_bits,
pair_bits, pair_answers), restated in Python and checked exhaustively over the codes:

* no false negative: an entry (a, b) is answered by every query that has the same hash, one of
  the two residues in common and asks for the other (substitutions; with the "gap" code also the
  deletion answers, with the insertion pairs' (r, r) reading the insertion answers);
* through BOTH halves of the test an entry answers exactly the question it stands for and its
  one alias modulo 32, both codes + 16 (a false positive that takes a second neighbour of the
  query -- the walk behind the filter rejects it); in particular a query that is itself in the
  filter gets no positive from its own entry other than the identity, which the kernel masks.

This is host logic: the device functions are exercised by the GPU parity tests."""

import itertools

import numpy as np
import pytest

MASK32 = 0xFFFFFFFF


def ror(x, n):
    n &= 31
    return ((x >> n) | (x << (32 - n))) & MASK32


def bitrev(x):
    return int("{:032b}".format(x)[::-1], 2)


def amounts(W):
    wl, wh = W & MASK32, (W >> 32) & MASK32
    return [wl & 31, wh & 31, (wl >> 8) & 31, (wh >> 8) & 31]


def entry_word(W, a, b):
    """the eight dwords an entry (W, a, b) sets: sums in 0..3, differences in 4..7"""
    am = amounts(W)
    s, d = a + b, a - b
    return [1 << ((am[k] + s) & 31) for k in range(4)] + [1 << ((am[k] + d) & 31) for k in range(4)]


def halves(word, W):
    am = amounts(W)
    xs = xd = MASK32
    for k in range(4):
        xs &= ror(word[k], am[k])
        xd &= ror(word[4 + k], am[k])
    return xs, xd


def answers(xs, xd, ra, rb):
    a1 = ror(xs, rb) & ror(xd, -rb)
    a2 = ror(xs, ra) & ror(bitrev(xd), 31 - ra)
    return a1, a2


@pytest.mark.parametrize("A", [4, 20])
def test_entry_is_found_by_its_neighbours_and_by_nobody_else(A):
    rng = np.random.default_rng(7)
    NONE, GAP = A, A + 1
    codes = list(range(A)) + [NONE, GAP]
    for a, b in itertools.product(codes, codes):
        W = int(rng.integers(0, 1 << 63))
        xs, xd = halves(entry_word(W, a, b), W)
        for ra, rb in itertools.product(codes, codes):
            a1, a2 = answers(xs, xd, ra, rb)
            for v in codes:
                # "v at the first position" (second = rb) is this entry iff (v, rb) == (a, b) -- or
                # its alias: sum and difference agree modulo 32 also for (a + 16, b + 16)
                same = (v == a and rb == b) or ((v - a) % 32 == 16 and (rb - b) % 32 == 16)
                assert ((a1 >> v) & 1) == (1 if same else 0), (a, b, ra, rb, v)
                # "v at the second position" (first = ra) iff (ra, v) == (a, b), or the alias
                same = (ra == a and v == b) or ((ra - a) % 32 == 16 and (v - b) % 32 == 16)
                assert ((a2 >> v) & 1) == (1 if same else 0), (a, b, ra, rb, v)
            # the query's own entry: nothing but the identity bits
            if (ra, rb) == (a, b):
                assert a1 & ((1 << (A + 2)) - 1) == 1 << a and a2 & ((1 << (A + 2)) - 1) == 1 << b


def test_words_with_many_entries_have_no_false_negative():
    rng = np.random.default_rng(11)
    A = 20
    for _ in range(200):
        W = int(rng.integers(0, 1 << 63))
        ents = [(int(rng.integers(0, A + 2)), int(rng.integers(0, A + 2))) for _ in range(int(rng.integers(1, 24)))]
        word = [0] * 8
        for a, b in ents:
            word = [x | y for x, y in zip(word, entry_word(W, a, b))]
        xs, xd = halves(word, W)
        for a, b in ents:
            for ra in range(A + 2):
                a1, a2 = answers(xs, xd, ra, b)
                assert (a1 >> a) & 1                      # found from the side of its second residue
            for rb in range(A + 2):
                a1, a2 = answers(xs, xd, a, rb)
                assert (a2 >> b) & 1                      # ... and of its first
