/*
 * kernels_rows.h -- probe kernel, variant 2: the ROW filter.
 *
 * Same path as kernels.h / kernels_sliced.h (variant enumeration -> Zobrist
 * hash -> Bloom test -> hash-table walk -> exact verify -> matrix accumulate;
 * reference: overlap.cc:253-284, variants.cc:260-428, bloompat.h:40-58,
 * overlap.cc:168-251), but the Bloom test of the A-1 substitution variants of
 * one position (a "row", variants.cc:280-293) costs ONE filter-word read
 * instead of A-1:
 *
 *   Every set-2 sequence t is entered into the filter once per position p, under
 *   the hash of t WITH POSITION p BLANKED,  W = H(t) ^ Z[p][t[p]],  together with
 *   the residue it has there (code t[p] < A), and once more under H(t) itself
 *   with code A ("the sequence as it is").  An entry (W, code) sets eight bits of
 *   the 256-bit filter word addressed by the high half of W: bit
 *   (a_k + code) mod 32 of dword k, a_0..a_3 being the low five bits of the four
 *   bytes of W's low half, a_4..a_7 four 5-bit fields of its high half.
 *
 *   A query q reads, for position p, the word of W = H(q) ^ Z[p][q[p]] -- the same
 *   W as every t that differs from q at most at p -- rotates dword k right by a_k
 *   and ANDs the eight: bit v of the result is the Bloom answer for "q with v at
 *   p" (8 bits per entry, block = one word, bloompat.h:22-58 geometry; at 16
 *   filter bits per entry a false-positive rate of ~6e-4).  All A answers of the
 *   row come out of one 32-byte LDS read and ~20 instructions.
 *
 * With d = 1 (with or without -i) the filter holds PAIR rows instead: one entry per pair
 * of positions, both blanked, the two residues coded as their sum and their difference --
 * one word read answers the 2 (A - 1) substitution variants of two positions
 * (pair_entry_bits / pair_bits / pair_answers below; round 3).  With -i the same words
 * answer the indel variants:
 *   - every set-2 sequence t is also entered once per GAP position ip: u = t with a blank
 *     in front of ip, the pair of u that holds the blank, the blank coded as A + 1 ("gap").
 *     A query's pair (p, p + 1) with both blanked IS (q without p) with a blank at p and
 *     q[p + 1] blanked, and (q without p + 1) with a blank at p + 1 and q[p] blanked: the
 *     word read for the substitution pair answers both deletion variants
 *     (variants.cc:301-325) in bit A + 1 of its two sides -- no read of their own;
 *   - t = q with v in front of g has the pair (v, q[g]) at (g, g + 1), t' = q with w in
 *     front of g + 1 the pair (q[g], w), and blanked they are the same string: ONE word
 *     under the rolling hash of "q with a gap at g and q[g] blanked" answers the insertions
 *     in front of g and in front of g + 1 (variants.cc:329-353).
 * L + 1 reads per query of length L for all its d = 1 variants with -i, L / 2 without.
 *
 * A double substitution (p, q) (d = 2: single rows) is read as
 * row p of "q already substituted at q" (variants.cc:370-399): A-1 reads instead
 * of (A-1)^2 probes.  Bloom positives are queued with their full variant hash
 * exactly as in the other kernels, so everything behind the filter
 * (resolve_kernel: table walk, check_variant, score) is unchanged.
 *
 * Slices: as in variant 1 the filter is cut into class-keyed slices that a
 * workgroup stages in LDS (layout.h), with one refinement the blanking makes
 * possible: the entry of (t, p) is filed under t's class key WITHOUT the term of
 * position p when p is a class position.  All A variants of a class-position
 * row therefore live in ONE slice of that position's "class part" of the filter,
 * and the layout (query_layout.hip) lists those rows as items grouped by that
 * slice: no substitution probe of d = 1 goes to HBM.
 */
#ifndef COMPAIRR_AMD_KERNELS_ROWS_H
#define COMPAIRR_AMD_KERNELS_ROWS_H

#include "kernels_sliced.h"
#include <type_traits>

namespace cmpr {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4 lds_u128_t;
typedef __attribute__((address_space(3))) u32x4 lds_u128_w_t;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glob_void_t;

__device__ __forceinline__ u32x4 lds_u128(uint32_t byte_addr)
{
  return *(lds_u128_t *)(uintptr_t)byte_addr;
}

/* one filter word: 8 dwords */
struct RowWord {
  u32x4 a, b;
};

/* the eight rotation amounts of an entry / a test: the low five bits of the four
   bytes of W's low half and four 5-bit fields of the low 20 bits of its high
   half (the word address is taken from the top of the high half) */
/* bit c of the result <-> "an entry with code c may be present under this W":
   each dword rotated right by its five hash bits, the eight ANDed */
__device__ __forceinline__ uint32_t row_bits(const RowWord &w, uint64_t Wk)
{
  const uint32_t wl = (uint32_t)Wk, wh = (uint32_t)(Wk >> 32);
  const uint32_t x0 = __builtin_amdgcn_alignbit(w.a.x, w.a.x, wl);
  const uint32_t x1 = __builtin_amdgcn_alignbit(w.a.y, w.a.y, wl >> 8);
  const uint32_t x2 = __builtin_amdgcn_alignbit(w.a.z, w.a.z, wl >> 16);
  const uint32_t x3 = __builtin_amdgcn_alignbit(w.a.w, w.a.w, wl >> 24);
  const uint32_t x4 = __builtin_amdgcn_alignbit(w.b.x, w.b.x, wh);
  const uint32_t x5 = __builtin_amdgcn_alignbit(w.b.y, w.b.y, wh >> 5);
  const uint32_t x6 = __builtin_amdgcn_alignbit(w.b.z, w.b.z, wh >> 10);
  const uint32_t x7 = __builtin_amdgcn_alignbit(w.b.w, w.b.w, wh >> 15);
  return (x0 & x1 & x2 & x3) & (x4 & x5 & x6 & x7);
}

/* the eight bits of entry (W, code), as the four 64-bit quarters of the word */
__host__ __device__ inline void row_entry_bits(uint64_t Wk, uint32_t code, uint64_t q[4])
{
  const uint32_t wl = (uint32_t)Wk, wh = (uint32_t)(Wk >> 32);
  q[0] = (1ull << ((wl + code) & 31u)) | (1ull << (32u + (((wl >> 8) + code) & 31u)));
  q[1] = (1ull << (((wl >> 16) + code) & 31u)) | (1ull << (32u + (((wl >> 24) + code) & 31u)));
  q[2] = (1ull << ((wh + code) & 31u)) | (1ull << (32u + (((wh >> 5) + code) & 31u)));
  q[3] = (1ull << (((wh >> 10) + code) & 31u)) | (1ull << (32u + (((wh >> 15) + code) & 31u)));
}

/* ---- pair rows (d = 1 without -i): TWO positions blanked per entry ----
 *
 * With single substitutions only, a set-2 sequence t is entered once per PAIR of positions
 * (p, p + 1), p even, under the hash with BOTH blanked,
 *   W = H(t) ^ Z[p][a] ^ Z[p+1][b],   a = t[p], b = t[p+1] (b = A: t ends at p),
 * and sets bit (a_k + a + b) mod 32 of dwords 0..3 and bit (a_k + a - b) mod 32 of dwords
 * 4..7 -- sum and difference of the two codes.  A query q with (ra, rb) at the pair reads
 * the word of its own pair-blanked hash ONCE, rotates and ANDs each half (xs, xd: bit i <->
 * "an entry with a + b = i" / "a - b = i" may be present) and gets from it
 *   "q with v at p"      (b = rb):  xs[v + rb] & xd[v - rb]   = (xs ror rb) & (xd rol rb),
 *   "q with w at p + 1"  (a = ra):  xs[ra + w] & xd[ra - w]   = (xs ror ra) & (rev(xd) ror (31 - ra)),
 * all 2 (A - 1) substitution variants of the two positions out of one 32-byte read, each
 * still a test of eight filter bits.  (Sum and difference: a test that asked for "a = v"
 * and "b = rb" separately would pass for every v that ANY neighbour of q carries, q itself
 * included -- half the test would be true for every query that is in set 2.)
 */
__host__ __device__ inline void pair_entry_bits(uint64_t Wk, uint32_t a, uint32_t b, uint64_t q[4])
{
  /* rotation amounts: the low five bits of wl, wh, wl >> 8, wh >> 8 -- the same four for
     the sum dwords 0..3 and the difference dwords 4..7 (two shifts per test instead of
     six; the halves are different dwords, their contents independent all the same) */
  const uint32_t wl = (uint32_t)Wk, wh = (uint32_t)(Wk >> 32);
  const uint32_t s = a + b, d = a - b;
  q[0] = (1ull << ((wl + s) & 31u)) | (1ull << (32u + ((wh + s) & 31u)));
  q[1] = (1ull << (((wl >> 8) + s) & 31u)) | (1ull << (32u + (((wh >> 8) + s) & 31u)));
  q[2] = (1ull << ((wl + d) & 31u)) | (1ull << (32u + ((wh + d) & 31u)));
  q[3] = (1ull << (((wl >> 8) + d) & 31u)) | (1ull << (32u + (((wh >> 8) + d) & 31u)));
}

/* the two halves of a pair word, rotated and ANDed (xs: sums, xd: differences) */
__device__ __forceinline__ void pair_bits(const RowWord &w, uint64_t Wk, uint32_t &xs, uint32_t &xd)
{
  const uint32_t wl = (uint32_t)Wk, wh = (uint32_t)(Wk >> 32);
  const uint32_t wl8 = wl >> 8, wh8 = wh >> 8;
  const uint32_t x0 = __builtin_amdgcn_alignbit(w.a.x, w.a.x, wl);
  const uint32_t x1 = __builtin_amdgcn_alignbit(w.a.y, w.a.y, wh);
  const uint32_t x2 = __builtin_amdgcn_alignbit(w.a.z, w.a.z, wl8);
  const uint32_t x3 = __builtin_amdgcn_alignbit(w.a.w, w.a.w, wh8);
  const uint32_t x4 = __builtin_amdgcn_alignbit(w.b.x, w.b.x, wl);
  const uint32_t x5 = __builtin_amdgcn_alignbit(w.b.y, w.b.y, wh);
  const uint32_t x6 = __builtin_amdgcn_alignbit(w.b.z, w.b.z, wl8);
  const uint32_t x7 = __builtin_amdgcn_alignbit(w.b.w, w.b.w, wh8);
  xs = x0 & x1 & x2 & x3;
  xd = x4 & x5 & x6 & x7;
}

/* (1 << n) - 1, n <= 31, and a sign-extended bit field: one instruction each */
__device__ __forceinline__ uint32_t bfm_u32(uint32_t n)
{
  uint32_t r;
  asm("v_bfm_b32 %0, %1, 0" : "=v"(r) : "v"(n));
  return r;
}
__device__ __forceinline__ int bfe_i32(uint32_t x, int off, int width)
{
  return __builtin_amdgcn_sbfe((int)x, (uint32_t)off, (uint32_t)width);
}

/* the answers of a pair: bit v of ansA <-> "v at the first position", bit w of ansB <->
   "w at the second" (own residues and everything above A still in: the caller masks) */
__device__ __forceinline__ void pair_answers(uint32_t xs, uint32_t xd, uint32_t ra, uint32_t rb,
                                             uint32_t &ansA, uint32_t &ansB)
{
  ansA = __builtin_amdgcn_alignbit(xs, xs, rb) & __builtin_amdgcn_alignbit(xd, xd, 0u - rb);
  const uint32_t rev = __builtin_bitreverse32(xd);
  ansB = __builtin_amdgcn_alignbit(xs, xs, ra) & __builtin_amdgcn_alignbit(rev, rev, 31u - ra);
}

/* word of W inside a slice of `nwords` 32-byte words (any count up to 2^13, not only
   powers of two: the filter is sized to the entry count): the top 16 bits of W scaled
   to the slice -- a 24-bit multiply, which the vector unit issues at full rate
   (v_mul_hi_u32 takes four issue slots: tools/calib.hip) */
__host__ __device__ inline uint32_t row_word(uint64_t Wk, uint32_t nwords)
{
  return ((uint32_t)(Wk >> 48) * nwords) >> 16;
}

/* ------------------------------------------------------------------ */
/* filter build                                                         */
/* ------------------------------------------------------------------ */

/* One thread per set-2 sequence: L + 1 entries (bloom_set, bloompat.h:50-53,
   once per blanked position and once for the sequence as it is). */
static __global__ void __launch_bounds__(BLOCK_THREADS)
build_rows_kernel(const BuildParams B)
{
  const uint64_t i = (uint64_t)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if (i >= B.n)
    return;
  const uint64_t b = B.off[i];
  const uint32_t L = (uint32_t)(B.off[i + 1] - b);
  const uint8_t *s = B.res + b;
  uint64_t h = 0;
  if (B.use_genes) {
    const uint64_t *vk = B.zob + (uint64_t)B.A * B.zpos;
    h = vk[B.v[i]] ^ vk[B.n_v + B.j[i]];
  }
  for (uint32_t p = 0; p < L; p++)
    h ^= B.zob[B.A * p + s[p]];
  const SliceGeom &g = B.geom;
  bool heavy = false;
  const uint32_t ck = class_key_of(g.ctab, g, B.A, B.use_genes != 0, s, L,
                                   B.use_genes ? B.v[i] : 0u, B.use_genes ? B.j[i] : 0u, &heavy);
  const uint32_t nwords = g.rw_words;
  unsigned long long *words = (unsigned long long *)B.bloom;
  auto enter = [&](uint64_t Wk, uint32_t code, uint32_t slice) {
    if (B.count) {                            /* (the geometry pass: entries per slice) */
      atomicAdd(B.count + slice, 1u);
      return;
    }
    const uint64_t w = (uint64_t)paged_slice(g, slice, Wk) * nwords + row_word(Wk, nwords);
    uint64_t q[4];
    row_entry_bits(Wk, code, q);
#pragma unroll
    for (int k = 0; k < 4; k++)
      atomicOr(words + 4 * w + k, (unsigned long long)q[k]);
  };
  enter(h, B.A, row_slice(g, ck, -1));
  if (B.pairs) {
    /* pair rows: one entry per pair of positions, filed under the class key without the
       terms of the class positions inside the pair, in the class part of the first of them */
    auto enter_pair = [&](uint64_t Wk, uint32_t a, uint32_t b2, uint32_t slice) {
      if (B.count) {
        atomicAdd(B.count + slice, 1u);
        return;
      }
      const uint64_t w = (uint64_t)paged_slice(g, slice, Wk) * nwords + row_word(Wk, nwords);
      uint64_t q[4];
      pair_entry_bits(Wk, a, b2, q);
#pragma unroll
      for (int k = 0; k < 4; k++)
        atomicOr(words + 4 * w + k, (unsigned long long)q[k]);
    };
    for (uint32_t p = 0; p < L; p += 2) {
      const uint32_t a = s[p], b2 = p + 1 < L ? s[p + 1] : B.A;
      uint32_t key = ck;
      int ci = -1;
      if (heavy)
        for (uint32_t k = 0; k < g.k; k++) {
          const uint32_t mk = class_pos(L, k, g.c0);
          if ((mk & ~1u) == p) {
            key ^= g.ctab[g.off_cr + k * B.A + s[mk]];
            if (ci < 0)
              ci = (int)k;
          }
        }
      if (ci >= 0)
        key ^= pair_part_terms(g.ctab, g, B.A, L, p, [&](uint32_t pos) -> uint32_t { return s[pos]; });
      uint64_t Wk = h ^ B.zob[B.A * p + a];
      if (p + 1 < L)
        Wk ^= B.zob[B.A * (p + 1) + b2];
      enter_pair(Wk, a, b2, row_slice(g, key, ci));
    }
    if (!B.indels)
      return;
    /* -i: the gap pairs.  u = t with a blank in front of position ip (length L + 1), ip =
       0 .. L; the pair of u that holds the blank, with the blank coded as A + 1 ("gap") and
       BOTH positions blanked in the hash -- what a query one residue longer finds under the
       hash of its own pair when the residue it has there is the one t lacks (kernels_rows.h
       probe: deletion answers).  Filed like a pair of a sequence of length L + 1: class
       residues at the positions of that length, taken around the gap. */
    uint64_t hg = 0;                               /* hash of u: blank at 0 = everything shifted by one */
    if (B.use_genes) {
      const uint64_t *vk = B.zob + (uint64_t)B.A * B.zpos;
      hg = vk[B.v[i]] ^ vk[B.n_v + B.j[i]];
    }
    for (uint32_t p = 0; p < L; p++)
      hg ^= B.zob[B.A * (p + 1) + s[p]];
    uint32_t base = ck;                            /* class key without the class residues */
    if (heavy)
      for (uint32_t k = 0; k < g.k; k++)
        base ^= g.ctab[g.off_cr + k * B.A + s[class_pos(L, k, g.c0)]];
    for (uint32_t ip = 0; ip <= L; ip++) {
      if (ip > 0)                                  /* t[ip - 1] moves from position ip to ip - 1 */
        hg ^= B.zob[B.A * ip + s[ip - 1]] ^ B.zob[B.A * (ip - 1) + s[ip - 1]];
      const uint32_t p0 = ip & ~1u;
      uint32_t a, b2;
      uint64_t Wk = hg;
      if (ip == p0) {                              /* (gap, u[ip + 1] = t[ip]) */
        a = B.A + 1u;
        b2 = ip < L ? (uint32_t)s[ip] : B.A;
        if (ip < L)
          Wk ^= B.zob[B.A * (ip + 1) + s[ip]];
      } else {                                     /* (u[ip - 1] = t[ip - 1], gap) */
        a = s[ip - 1];
        b2 = B.A + 1u;
        Wk ^= B.zob[B.A * (ip - 1) + s[ip - 1]];
      }
      uint32_t key = base;
      int ci = -1;
      if (heavy)
        for (uint32_t k = 0; k < g.k; k++) {
          const uint32_t mk = class_pos(L + 1, k, g.c0);
          if ((mk & ~1u) == p0) {
            if (ci < 0)
              ci = (int)k;
          } else {
            key ^= g.ctab[g.off_cr + k * B.A + s[mk < ip ? mk : mk - 1]];
          }
        }
      if (ci >= 0)
        key ^= pair_part_terms(g.ctab, g, B.A, L + 1, p0,
                               [&](uint32_t pos) -> uint32_t { return s[pos < ip ? pos : pos - 1]; });
      enter_pair(Wk, a, b2, row_slice(g, key, ci));
    }
    return;
  }
  for (uint32_t p = 0; p < L; p++) {
    uint32_t key = ck;
    int ci = -1;                               /* first class residue at p, if any */
    if (heavy)
      for (uint32_t k = 0; k < g.k; k++)
        if (class_pos(L, k, g.c0) == p) {
          key ^= g.ctab[g.off_cr + k * B.A + s[p]];
          if (ci < 0)
            ci = (int)k;
        }
    enter(h ^ B.zob[B.A * p + s[p]], s[p], row_slice(g, key, ci));
  }
}

/* ------------------------------------------------------------------ */
/* probe kernel                                                         */
/* ------------------------------------------------------------------ */

/* rows per 64-bit mask register / rows per block / bits of a packed residue */
template <int A> struct RowCfg {
  static constexpr int RPW = A == 20 ? 3 : 16;
  static constexpr int RB = A == 20 ? 3 : 4;       /* one mask register; 3 x (2 + 8) registers
                                                      of keys and filter words in flight */
  static constexpr uint32_t RBITS = A == 20 ? 5u : 2u;
};

/* keys per position of the kernel's LDS copy of the Zobrist table: d = 2 stores a row twice
   (rotated reads); pair rows add a ZERO key for code A -- the residue code the layout pads
   a query with behind its end (query_layout.hip fill_tiles_kernel), so that a pair that
   hangs over the end hashes, and reads, without a test */
__host__ __device__ constexpr uint32_t zs_of(int A, int D, bool pairs)
{
  return D == 2 ? 2u * (uint32_t)A : (pairs ? (uint32_t)A + 1u : (uint32_t)A);
}

/* ------------------------------------------------------------------ */
/* the per-wave queue of this kernel holds ENTRIES, not variants        */
/* ------------------------------------------------------------------ */
/*
 * A lane that finds positives in a block of rows does not turn them into variants on
 * the spot -- that loop ran for the whole wave whenever ONE lane had a positive bit,
 * ~45 instructions for one or two busy lanes, on nearly every block.  It queues one
 * ENTRY: the base hash, where the rows are, and the mask of positive variants.  When
 * 64 entries are queued, all 64 lanes take one each, pop ONE variant from it (hash,
 * kind | position | residue), the 64 variants leave as one block (positives buffer /
 * inline resolve, flush_or_resolve) and the entries that still hold bits are queued
 * again.  The expansion thus runs at full lane utilisation, once per 64 variants.
 *
 *   kind      hash            ca                                  cb, m
 *   K_ROWS    query hash h    K_ROWS | p0 << 3 | rpack << 19      mask of RB rows x A residues, bit A j + v,
 *                                                                 low half in cb, high half in m[0..27];
 *                                                                 rpack (the query's own residues of the RB
 *                                                                 rows, RBITS each) continues in m[28..31]
 *   K_INSROWS hash of q with  K_INSROWS | g << 3 | q[g] << 19     cb = residues v in front of g, m = residues w
 *             a gap at g and                                      in front of g + 1 (-i, pair rows)
 *             q[g] blanked
 *   K_PAIR    query hash h    K_PAIR | p << 3 | ra << 19          cb = mask of residues v at p, m = mask of
 *                             | rb << 24                          residues w at p + 1; ra, rb = the query's own
 *   K_SUB,    the row's       kind | p << 3                       m = mask of residues v (one row: an item,
 *   K_INS     blanked hash                                        or an insertion row)
 *   others    variant hash    as it leaves (pack_a)               cb as it leaves; one variant
 * (a deletion answer -- bit A + 1 of a K_PAIR mask -- leaves as K_DEL, its hash worked out in the drain
 * from the query's first-deleted hash and its residues)
 */
constexpr uint32_t K_ROWS = 5;
constexpr uint32_t K_INSROWS = 6;        /* a pair of insertion rows of one query (-i) */
constexpr uint32_t K_PAIR = 7;           /* a pair of substitution rows (pair rows: d = 1); with -i also its deletion answers */

template <int A, int D, bool GENES, bool INLINE, bool PAIRS>
__device__ __forceinline__ void drain_round(SProber &W, uint32_t zl_addr, int n, bool last = false)
{
  constexpr uint32_t ZS = zs_of(A, D, PAIRS);
  constexpr uint32_t RBITS = RowCfg<A>::RBITS, RMASK = (1u << RBITS) - 1u;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const int first = W.qn - n;
  const bool act = (int)W.lane < n;
  const int e = first + (int)W.lane;
  uint64_t B = 0;
  uint32_t slot = POS_NULL_SLOT, ca = 0, cb = 0, m = 0;
  if (act) {
    B = W.q.hash[e];
    slot = W.q.slot[e];
    ca = W.q.ca[e];
    cb = W.q.cb[e];
    m = W.q.m[e];
  }
  const uint32_t kind = ca & 7u, p0 = (ca >> 3) & 0xffffu;
  uint64_t hv = B;
  uint32_t oca = ca, ocb = cb, ncb = cb, nm = m;
  bool more = false;
  bool lazy_del = false;                  /* a deletion variant whose hash is worked out here */
  uint32_t lazy_p = 0;
  if (kind == K_ROWS) {
    uint64_t mk = ((uint64_t)(m & 0x0fffffffu) << 32) | cb;
    const uint32_t rpack = (ca >> 19) | ((m >> 28) << 13);
    const uint32_t idx = (uint32_t)__ffsll((unsigned long long)mk) - 1u;
    const uint32_t j = idx / (uint32_t)A, v = idx - j * (uint32_t)A;
    const uint32_t p = p0 + j, r = (rpack >> (RBITS * j)) & RMASK;
    const uint32_t za = zl_addr + ZS * p * 8u;
    hv = B ^ lds_u64(za + r * 8u) ^ lds_u64(za + v * 8u);
    oca = pack_a(K_SUB, p, v);
    ocb = 0;
    mk &= mk - 1ull;
    more = mk != 0;
    ncb = (uint32_t)mk;
    nm = (m & 0xf0000000u) | (uint32_t)(mk >> 32);
  } else if (kind == K_PAIR) {
    const uint32_t second = cb == 0u ? 1u : 0u;
    const uint32_t bitsv = second ? m : cb;
    const uint32_t v = (uint32_t)__ffs((int)bitsv) - 1u;
    const uint32_t p = p0 + second, r = second ? (ca >> 24) & 31u : (ca >> 19) & 31u;
    const uint32_t za = zl_addr + ZS * p * 8u;
    hv = B ^ lds_u64(za + r * 8u) ^ lds_u64(za + v * 8u);
    oca = pack_a(K_SUB, p, v);
    ocb = 0;
    ncb = second ? 0u : cb & (cb - 1u);
    nm = second ? m & (m - 1u) : m;
    more = (ncb | nm) != 0u;
    if (v == (uint32_t)A + 1u) {          /* -i: "q without p" */
      oca = pack_a(K_DEL, p, 0);
      lazy_del = act;
      lazy_p = p;
    }
  } else if (kind == K_INSROWS) {
    /* a pair of insertion rows: B = the hash of q with a gap at g and q[g] blanked; cb = the
       residues v that may stand in front of g (then q[g] follows at g + 1), m = the residues
       w in front of g + 1 (q[g] stays at g) */
    const uint32_t second = cb == 0u ? 1u : 0u;
    const uint32_t bitsv = second ? m : cb;
    const uint32_t v = (uint32_t)__ffs((int)bitsv) - 1u;
    const uint32_t rg = (ca >> 19) & 31u;
    const uint32_t z0 = zl_addr + ZS * p0 * 8u, z1 = z0 + ZS * 8u;
    hv = B ^ lds_u64(z0 + (second ? rg : v) * 8u) ^ lds_u64(z1 + (second ? v : rg) * 8u);
    oca = pack_a(K_INS, p0 + second, v);
    ocb = 0;
    ncb = second ? 0u : cb & (cb - 1u);
    nm = second ? m & (m - 1u) : m;
    more = (ncb | nm) != 0u;
  } else if (kind == K_SUB || kind == K_INS) {
    const uint32_t v = (uint32_t)__ffs((int)m) - 1u;
    hv = B ^ lds_u64(zl_addr + (ZS * p0 + v) * 8u);
    oca = pack_a(kind, p0, v);
    ocb = 0;
    nm = m & (m - 1u);
    more = nm != 0;
  }
  if (lazy_del) {
    /* "the query without position lazy_p", from its first-deleted hash and its residues
       (zobrist_hash_delete_first + the rolling update, variants.cc:301-325) */
    const ProbeParams &P = W.P;
    const QueryRec *qr = P.qrec + slot;
    uint64_t hd;
    if (P.rec_tiles) {
      /* record tiles: zobrist_hash_delete_first (zobrist.cc:90-104) from the query's record -- once per deletion
         variant that passed the filter, not once per query */
      hd = 0;
      if (GENES) {
        const uint64_t *gk = P.zob + (size_t)A * P.zpos;
        hd = gk[qr->v] ^ gk[P.n_v + qr->j];
      }
      for (uint32_t y = 1; y < qr->len; y++)
        hd ^= lds_u64(zl_addr + (ZS * (y - 1u) + ((qr->res[y >> 2] >> ((y & 3u) * 8u)) & 0xffu)) * 8u);
    } else {
      hd = P.qhdel[slot];
    }
    const uint32_t *far = P.qres + P.tiles[slot >> 6].res_base + (slot & 63u);
    auto res_of = [&](uint32_t y) -> uint32_t {
      const uint32_t w = y < 36u ? qr->res[y >> 2] : far[(size_t)(y >> 2) * WAVE];
      return (w >> ((y & 3u) * 8u)) & 0xffu;
    };
    uint32_t prev = res_of(0);
#pragma unroll 1
    for (uint32_t y = 1; y <= lazy_p; y++) {
      const uint32_t r = res_of(y);
      if (r != prev)
        hd ^= lds_u64(zl_addr + (ZS * (y - 1u) + prev) * 8u) ^ lds_u64(zl_addr + (ZS * (y - 1u) + r) * 8u);
      prev = r;
    }
    hv = hd;
  }
  /* the variants take the places of their entries and leave as one block */
  if (act) {
    W.q.hash[e] = hv;
    W.q.ca[e] = oca;
    W.q.cb[e] = ocb;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  W.qn = first;
  if (!CMPR_DBG(W.P, DBG_SKIP_RESOLVE))
    flush_or_resolve<GENES, INLINE>(W, first, n, last);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  /* entries with variants left queue up again */
  more = more && act;
  const uint64_t mm = __ballot(more);
  if (mm) {
    if (more) {
      const int x = W.qn + (int)rank_below(mm);
      W.q.hash[x] = B;
      W.q.slot[x] = slot;
      W.q.ca[x] = ca;
      W.q.cb[x] = ncb;
      W.q.m[x] = nm;
    }
    W.qn += __popcll(mm);
  }
}

/* queue one entry per lane with `pos` (nbits = its positive variants, for the counters) */
template <int A, int D, bool GENES, bool INLINE, bool PAIRS>
__device__ __forceinline__ void q_push(SProber &W, uint32_t zl_addr, bool pos, uint64_t B, uint32_t ca,
                                       uint32_t cb, uint32_t m, uint32_t nbits)
{
  const uint64_t mm = __ballot(pos);
  if (mm) {
    if (pos) {
      const int e = W.qn + (int)rank_below(mm);
      W.q.hash[e] = B;
      W.q.slot[e] = W.qslot;
      W.q.ca[e] = ca;
      W.q.cb[e] = cb;
      W.q.m[e] = m;
      W.st.bloom_pos += nbits;
    }
    W.qn += __popcll(mm);
    while (W.qn >= WAVE)
      drain_round<A, D, GENES, INLINE, PAIRS>(W, zl_addr, WAVE);
  }
}

/*
 * Chunk::pass of this kernel:
 *   0      the sequence itself + every substitution row (pair) that holds no class
 *          position of a split ("heavy") tile (+ the double substitutions); with -i
 *          the pairs also answer the tile's deletion variants, and its insertion pairs
 *          run on the same staged slice (the class keys of variant 2 have no length term
 *          with -i, so an indel variant stays in its query's slice unless it moves or
 *          touches a class residue -- pairs that hold one are items, pairs in front of
 *          one are read where they lie)
 *   3 + i  items: the substitution row of class position i of the heavy queries (pair
 *          rows: the pair that holds it, with -i also the insertion pair that does),
 *          blocks of 64 grouped by the slice the row is filed under (class part i of
 *          the filter), ITEM_BLOCKS per claim
 * TileDesc::slice is the slice the tile's rows of this pass are filed under;
 * for a staged chunk it is the slice in LDS.
 *
 * Workgroup = NW waves around a ring of RING slice buffers in LDS.  Chunks are
 * dealt to the workgroups statically (workgroup b: chunks b, b + G, ... of the
 * list, heaviest first).  A wave that finds no tile to work on stages the next
 * chunk: it copies slice and tile references into the next free buffer with
 * LDS-DMA (global_load_lds: no registers, asynchronous) and publishes it; waves
 * claim tiles from the published buffers in order, one tile (or ITEM_BLOCKS item
 * blocks) at a time with an LDS atomic, and a buffer is free again when its tiles
 * are finished.  Nothing in the
 * steady state is a workgroup barrier: staging overlaps the rows, several chunks
 * are copied at once, and no wave idles at the end of a chunk while another still
 * works on it.
 *
 * LDS: [RING slices, rw_words x 32 B each][ZS x zpos Zobrist keys]
 *      [NW WaveQueues][CR tables][heavy bitmap (-i)]
 *      [RING ring slots][RING x chunk_cap tile refs]
 */
constexpr uint32_t RING = 4;
constexpr uint32_t RING_END = 0xffffffffu;
/* blocks of 64 items a wave takes with one claim: an item is one row, and claiming,
   descriptor and address arithmetic of a unit cost ~3 rows' worth of instructions */
constexpr uint32_t ITEM_BLOCKS = 4;

/* -DCMPR_PHASE_TIMING (tools/phase_timing.sh): every wave sums the shader cycles
   (s_memtime) it spends per phase and adds them to ProbeParams::stats[8 + phase];
   a diagnostic build, never the shipped one */
#ifdef CMPR_PHASE_TIMING
#define PT_DECL unsigned long long pt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pt_t = __builtin_amdgcn_s_memtime();
#define PT_MARK(i) do { const unsigned long long pt_n = __builtin_amdgcn_s_memtime(); pt_acc[i] += pt_n - pt_t; pt_t = pt_n; } while (0)
#define PT_FLUSH do { if (lane == 0) for (int pt_k = 0; pt_k < 8; pt_k++) atomicAdd(P.stats + 8 + pt_k, pt_acc[pt_k]); } while (0)
#else
#define PT_DECL
#define PT_MARK(i) do { } while (0)
#define PT_FLUSH do { } while (0)
#endif
enum { PT_CLAIM = 0, PT_TILE_DATA = 1, PT_ROWS = 2, PT_EMIT = 3, PT_OTHER = 4, PT_LOADER_WAIT = 5,
       PT_LOADER_DMA = 6, PT_TAIL = 7 };

/* one buffer of the ring */
struct RingSlot {
  /* tag << 32 | ntiles << 16 | tiles handed out: claimed with ONE 64-bit LDS add,
     so that a claim names the tenant it belongs to; tag 0 = nothing yet, tags
     grow by one per tenant, RING_END = no more chunks */
  unsigned long long claim;
  uint32_t done;             /* tiles of the tenant finished */
  uint32_t slice, pass;
  uint32_t first;            /* class-row chunk: its first item (blocks of 64 items, no tile refs) */
  uint32_t pad[2];           /* ([0] of slot 0: the workgroup's staging counter; [0] of slot 1: buffers whose
                                chain of tenants has ended; [1]: item chunk: its blocks) */
  uint32_t next_chunk;       /* the chunk of the tenant after this one (P.deal: reserved when this one was staged) */
  uint32_t page;             /* the page of its slice the tenant holds | e of the slice << 4 (layout.h SliceGeom; 0: no pages) */
};


/* (16 waves = 1024 lanes is the largest workgroup there is: five waves per SIMD would take two
   workgroups per CU, and two rings of slices do not fit the LDS) */
template <int A, int D, bool INDELS, bool GENES, int NW, bool INLINE, bool WIDE = false>
__global__ void __launch_bounds__(NW * WAVE, 4)
probe_rows_kernel(const ProbeParams P)
{
  /* INLINE = false: the fast form, Bloom positives go to the positives buffer for
     resolve_kernel.  INLINE = true: positives are walked and verified here -- the
     deferred_resolve = 0 mode, and the redo pass after a launch whose positives
     did not fit the buffer (it does nothing unless that launch was flagged). */
  if (INLINE && P.redo && *(volatile unsigned long long *)P.overflow == 0ull)
    return;
  constexpr uint32_t NT = NW * WAVE;
  constexpr uint32_t MCR = kernel_class_res(A, WIDE);  /* (WIDE: layout.h) */
  constexpr bool PAIRS = D == 1;                  /* the filter holds pair rows (build_rows_kernel) */
  constexpr uint32_t ZS = zs_of(A, D, PAIRS);
  /* pair rows: what stands behind the end of a query, also in residue dwords that were never
     loaded (query_layout.hip pads the loaded ones) */
  constexpr uint32_t PADW = PAIRS ? (uint32_t)A * 0x01010101u : 0u;
  constexpr int RPW = RowCfg<A>::RPW;
  constexpr int RB = RowCfg<A>::RB;
  constexpr uint32_t RBITS = RowCfg<A>::RBITS;
  constexpr uint32_t RMASK = (1u << RBITS) - 1u;
  constexpr uint32_t AMASK = (1u << A) - 1u;

  extern __shared__ __align__(16) unsigned char smem[];
  if ((uint32_t)(uintptr_t)smem != 0u)
    __builtin_trap();                       /* the slices are read at absolute LDS addresses */
  const uint32_t nwords = P.geom.rw_words;
  const uint32_t slice_bytes = nwords * ROW_WORD_BYTES;     /* a multiple of 1 KiB */
  const unsigned char *filter = (const unsigned char *)P.bloom;
  const uint32_t zl_addr = RING * slice_bytes;              /* LDS address of zl */
  uint64_t *zl = (uint64_t *)(smem + zl_addr);
  const uint32_t nz = ZS * P.zpos;
  /* (no copy of the matrix in LDS: the fast form scores nothing -- resolve_kernel does --
     and the form that resolves inline, the rare path, adds to the matrix where it lies;
     the 16 KiB go to the slices) */
  WaveQueue *queues = (WaveQueue *)(zl + nz);
  uint32_t *cr_lds = (uint32_t *)(queues + NW);
  uint32_t *hv_lds = cr_lds + MAX_CLASS_RES * A;
  RingSlot *ring = (RingSlot *)(hv_lds + (INDELS ? HEAVY_WORDS : 0u));
  TileRef *tref_lds = (TileRef *)(ring + RING);             /* RING x chunk_cap */
  const uint32_t chunk_cap = P.chunk_cap;
  /* record tiles (ProbeParams::rec_tiles: amino acids, d = 1 without -i): a tile's lengths and residues come from
     the queries' 64-byte records and the hashes are worked out here; the gene keys lie behind the tile references */
  constexpr bool REC_OK = PAIRS && D == 1;
  const bool rec_tiles = REC_OK && P.rec_tiles != 0u;
  /* (rec_tiles == 2, every sequence within 28 residues: the query's hash came with the record, in the last two
     residue words -- nothing is hashed here but, with -i, the insert-first hash) */
  const bool rec_hash = rec_tiles && P.rec_tiles == 2u;
  uint64_t *gk_lds = (uint64_t *)(tref_lds + RING * chunk_cap);
  /* (-i: the class tables in front of the class-residue rows -- CL | CV | CJ -- behind the gene keys: a record
     tile works the query's class key out itself) */
  uint32_t *cb_lds = (uint32_t *)(gk_lds + (GENES ? P.n_v + P.n_j_keys : 0u));

  for (uint32_t i = threadIdx.x; i < nz; i += NT) {
    const uint32_t r = i % ZS;
    zl[i] = (PAIRS && r == (uint32_t)A) ? 0ull : P.zob[(i / ZS) * A + r % A];
  }
  for (uint32_t i = threadIdx.x; i < MAX_CLASS_RES * (uint32_t)A; i += NT)
    cr_lds[i] = P.geom.ctab[P.geom.off_cr + i];
  if (INDELS)
    for (uint32_t i = threadIdx.x; i < HEAVY_WORDS; i += NT)
      hv_lds[i] = P.geom.ctab[P.geom.off_hv + i];
  for (uint32_t i = threadIdx.x; i < RING * (uint32_t)(sizeof(RingSlot) / 4); i += NT)
    ((uint32_t *)ring)[i] = 0;
  if (REC_OK && rec_tiles && GENES)
    for (uint32_t i = threadIdx.x; i < P.n_v + P.n_j_keys; i += NT)
      gk_lds[i] = P.zob[(size_t)A * P.zpos + i];
  if (REC_OK && INDELS && rec_tiles)
    for (uint32_t i = threadIdx.x; i < P.geom.off_cr; i += NT)
      cb_lds[i] = P.geom.ctab[i];

  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x / WAVE;
  const uint32_t KH = P.geom.k;
  const uint32_t smask = P.geom.smask;
  SProber W{P, (const uint64_t *)smem, queues[wave], nullptr,
            lane, 0u, 0u, 0u, smask, 0u, 0, {0ull, 0u, 0u, 0u}, 0ull};
  claim_pos_block(W);
  unsigned long long reads = 0;                   /* filter words read by this lane */
  __syncthreads();                                /* tables and ring slots are in place */

  const uint32_t G = gridDim.x;
  const bool have_chunks = blockIdx.x < P.nchunks;
  PT_DECL

  /* ---------------- staging: any wave without a tile to work on ---------------- */
  /* Chunk number T of this workgroup (its chunks: blockIdx.x + (T - 1) G) goes into
     buffer (T - 1) % RING once that buffer's tenant T - RING is finished.  The
     wave that wins the compare-and-swap on the issue counter copies slice and tile
     references with LDS-DMA (no registers, nothing it issues depends on an earlier
     load of its own), waits for them and publishes the tenant.  Several waves can
     be staging different chunks at the same time, so the copies of up to RING
     chunks overlap -- and a workgroup short of tiles turns more of its waves into
     loaders by itself. */
  /* Which chunk is tenant T of this workgroup: with a static deal chunk blockIdx.x + (T - 1) G of
     the list (heaviest first); with P.deal only the first RING are, and every staging reserves
     the chunk of the tenant that will follow in the same buffer from a counter (one per DEAL_GROUPS-th
     of the grid, each over its share of the rest of the list) -- the list is then worked off in its
     order by whoever is free, and the launch ends within one light
     chunk for every workgroup.  A buffer whose next chunk lies behind the end of the list gets the
     end marker; the launch is over for a workgroup when all RING buffers carry one. */
  uint32_t *issue = (uint32_t *)&ring[0].pad[0];            /* next tag to stage, starts at 1 */
  uint32_t *ended = (uint32_t *)&ring[1].pad[0];            /* buffers with an end marker */
  const uint32_t nchunks = P.nchunks;
  auto try_stage = [&]() -> bool {
    const uint32_t T = *(volatile uint32_t *)issue + 1u;     /* (stored as T - 1: zero-initialised) */
    if (*(volatile uint32_t *)ended >= RING)
      return false;                           /* everything, the end markers included, is issued */
    const uint32_t b = (T - 1u) % RING;
    volatile RingSlot *rs = ring + b;
    const unsigned long long cl = rs->claim;
    const uint32_t hi = (uint32_t)(cl >> 32);
    if (hi == RING_END) {
      /* this buffer is through: the tag passes to the next one */
      uint32_t won = 0;
      if (lane == 0)
        won = atomicCAS(issue, T - 1u, T) == T - 1u ? 1u : 0u;
      return __builtin_amdgcn_readfirstlane(won) != 0u;
    }
    if (hi != (T > RING ? T - RING : 0u) || rs->done != (uint32_t)((cl >> 16) & 0xffffu))
      return false;                           /* the buffer's tenant is not finished (or not even there) */
    uint32_t won = 0;
    if (lane == 0)
      won = atomicCAS(issue, T - 1u, T) == T - 1u ? 1u : 0u;
    if (!__builtin_amdgcn_readfirstlane(won))
      return false;
    const uint32_t idx = (P.deal && T > RING) ? rs->next_chunk : blockIdx.x + (T - 1u) * G;
    const uint32_t grp = blockIdx.x % DEAL_GROUPS;
    if (idx >= nchunks) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      rs->claim = ((unsigned long long)RING_END << 32) | ((unsigned long long)(T & 0xffffu) << 16);
      if (lane == 0)
        atomicAdd(ended, 1u);
      return true;
    }
    uint32_t reserved = 0;
    if (P.deal && lane == 0)
      reserved = atomicAdd((unsigned int *)(P.deal_ctr + grp * DEAL_STRIDE), 1u);
    const Chunk ck = P.chunks[idx];
    /* (opaque: or the per-lane source pointers are hoisted out of the tile loop and
       live -- spilled -- across it for the sake of this rare path) */
    uint32_t l16 = lane * 16u;
    asm volatile("" : "+v"(l16));
    const unsigned char *src = filter + (size_t)ck.slice * slice_bytes + l16;
    const uint32_t dst = b * slice_bytes;
    /* one wave-instruction copies 1 KiB: lane l its 16 bytes to (uniform base) +
       16 l; lanes past the end of a slice that is no whole KiB stay out of it */
    for (uint32_t off = 0; off < slice_bytes; off += 1024u)
      if (off + l16 < slice_bytes)
        __builtin_amdgcn_global_load_lds((glob_void_t *)(src + off),
                                         (lds_void_t *)(uintptr_t)(dst + off), 16, 0, 0);
    const unsigned char *tsrc = (const unsigned char *)(P.tile_refs + ck.first_tile) + l16;
    const uint32_t tdst = (uint32_t)(uintptr_t)(tref_lds + b * chunk_cap);
    /* (a class-row chunk has no tile references: its "tiles" are blocks of 64 items) */
    const uint32_t cpass = ck.pass & ((1u << CHUNK_PAGE_SHIFT) - 1u);     /* (above: the page, layout.h) */
    const uint32_t tbytes = cpass >= 3 ? 0u : ck.ntiles * (uint32_t)sizeof(TileRef);
    for (uint32_t off = 0; off < tbytes; off += 1024u)
      if (off + l16 < tbytes)
        __builtin_amdgcn_global_load_lds((glob_void_t *)(tsrc + off),
                                         (lds_void_t *)(uintptr_t)(tdst + off), 16, 0, 0);
    rs->done = 0;
    rs->slice = ck.slice;
    rs->pass = cpass;
    rs->page = ((ck.pass >> CHUNK_PAGE_SHIFT) & 7u) | (((ck.pass >> CHUNK_PAGE_E_SHIFT) & 3u) << 4);
    rs->first = ck.first_tile;
    rs->pad[1] = ck.ntiles;
    /* what is claimed: a tile, or ITEM_BLOCKS blocks of an item chunk */
    const uint32_t units = cpass >= 3 ? (ck.ntiles + ITEM_BLOCKS - 1u) / ITEM_BLOCKS : ck.ntiles;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         /* the copies have landed */
    if (P.deal)
      rs->next_chunk = RING * G + grp + DEAL_GROUPS * __builtin_amdgcn_readfirstlane(reserved);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    rs->claim = ((unsigned long long)T << 32) | ((unsigned long long)units << 16);
    return true;
  };

  /* what a tile needs per lane before its first row: the query's hash, its length and
     the residues of its first 24 positions; an item block -- per item the row's
     blanked hash, the query's slot (~0: padding) and its residue | position << 8 |
     kind << 24 */
  constexpr uint32_t TDW = 6;                 /* residue dwords that travel with a tile: 24 positions */
  struct TileData {
    uint64_t a;
    uint32_t b, c;
    uint32_t r0, r1, r2, r3, r4, r5;
    uint32_t r6, r7;               /* record tiles: positions 24 .. 31 (never live in the other instantiations) */
  };
  auto load_tile_data = [&](uint32_t len, uint32_t nvalid, uint32_t res_base, uint32_t t,
                            uint32_t tpass) -> TileData {
    TileData x;
    x.a = 0;
    x.b = x.c = 0;
    x.r0 = x.r1 = x.r2 = x.r3 = x.r4 = x.r5 = PADW;
    if constexpr (REC_OK)
      x.r6 = x.r7 = PADW;
    const bool valid = lane < nvalid;
    if (tpass >= 3) {
      /* a block of 64 class-row items, from item res_base on */
      const ItemRec it = P.items[res_base + lane];
      x.a = it.w;
      x.b = it.main;
      x.c = it.rp;
    } else if (REC_OK && rec_tiles) {
      /* the query's record as scatter_kernel left it: {cnt, v, j | rep, len, res 0-7 | res 8-23 | res 24-35, orig};
         gene numbers travel in `a` until the hash is worked out (tile_from_record) */
      if (valid) {
        const u32x4 *rq = (const u32x4 *)(P.qrec + (size_t)t * WAVE + lane);
        const u32x4 q0 = rq[0], q1 = rq[1], q2 = rq[2], q3 = rq[3];
        x.a = (uint64_t)q0.z | ((uint64_t)q0.w << 32);
        x.b = q1.y;
        x.r0 = q1.z; x.r1 = q1.w;
        x.r2 = q2.x; x.r3 = q2.y; x.r4 = q2.z; x.r5 = q2.w;
        x.r6 = q3.x; x.r7 = q3.y;
        x.c = q3.z;                              /* (rec_hash: the hash's high word; its low one is r7) */
      }
    } else {
      const uint32_t slot = t * WAVE + lane;
      if (valid) {
        x.a = P.qgh[slot];
        x.b = P.qlen[slot];
      }
      /* the residues of its first 24 positions: the rows then issue no load of
         their own (vmcnt retires in order -- a wait for one would also wait for
         the next tile's data, requested before) */
      const uint32_t nd = (len + 3u) >> 2;
      const uint32_t *qp = P.qres + res_base + lane;
      if (nd > 0) x.r0 = qp[0];
      if (nd > 1) x.r1 = qp[WAVE];
      if (nd > 2) x.r2 = qp[2 * WAVE];
      if (nd > 3) x.r3 = qp[3 * WAVE];
      if (nd > 4) x.r4 = qp[4 * WAVE];
      if (nd > 5) x.r5 = qp[5 * WAVE];
    }
    return x;
  };

  /* ---------------- the compute waves ---------------- */
  /* A wave works through the tenants in tag order: tag T lives in buffer
     (T - 1) % RING.  A claim is one 64-bit LDS add on the buffer's claim word; it
     returns the tenant's tag, its tile count and the tile handed out, all of the
     same instant. */
  uint32_t my_tag = 1;
  struct Claim {
    bool ok;
    uint32_t b, k;
  };
  /* blocking = false: give up instead of waiting for the loader */
  auto claim_tile = [&](bool blocking) -> Claim {
    Claim c;
    c.ok = false;
    c.b = c.k = 0;
    for (;;) {
      const uint32_t b = (my_tag - 1u) % RING;
      const unsigned long long cl = ((volatile RingSlot *)ring)[b].claim;
      const uint32_t hi = (uint32_t)(cl >> 32);
      if (hi == RING_END) {
        if (*(volatile uint32_t *)ended >= RING)
          return c;                           /* the end: every buffer carries the marker */
        my_tag++;                             /* nothing more comes through this buffer */
        continue;
      }
      if (hi < my_tag) {                      /* not published yet */
        if (!blocking)
          return c;
        PT_MARK(PT_CLAIM);
        const bool staged_one = try_stage();  /* nothing to work on: stage a chunk */
        PT_MARK(PT_LOADER_DMA);
        if (!staged_one)
          __builtin_amdgcn_s_sleep(1);
        continue;
      }
      if (hi > my_tag) {                      /* came and went */
        my_tag++;
        continue;
      }
      unsigned long long v = 0;
      if (lane == 0)
        v = atomicAdd((unsigned long long *)&ring[b].claim, 1ull);
      const uint32_t vlo = __builtin_amdgcn_readfirstlane((uint32_t)v);
      const uint32_t tag = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
      if (tag == RING_END)
        continue;                             /* (the add touched only its count bits) */
      const uint32_t nt = (vlo >> 16) & 0xffffu, k = vlo & 0xffffu;
      if (k < nt) {                           /* a tile of tenant `tag` (>= my_tag) */
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        c.ok = true;
        c.b = b;
        c.k = k;
        return c;
      }
      if (tag == my_tag)
        my_tag++;                             /* it has run out */
    }
  };

  const bool compute_wave = have_chunks;
  bool block_phase = compute_wave;
  Claim cur_c, nxt_c;
  cur_c.ok = nxt_c.ok = false;
  cur_c.b = cur_c.k = nxt_c.b = nxt_c.k = 0;
  TileData cur, nxt;
  cur.a = nxt.a = 0;
  cur.b = cur.c = nxt.b = nxt.c = 0;
  cur.r0 = cur.r1 = cur.r2 = cur.r3 = cur.r4 = cur.r5 = 0;
  nxt.r0 = nxt.r1 = nxt.r2 = nxt.r3 = nxt.r4 = nxt.r5 = 0;
  cur.r6 = cur.r7 = nxt.r6 = nxt.r7 = 0;

  /* What is known of the tile at hand, all wave-uniform: `staged` = it came from a
     chunk (slice in LDS at sbase; pass and cslice are the chunk's), else one of the
     tiles whose slice holds too few queries to be worth staging (claimed one at a
     time, the filter read where it lies). */
  struct Unit {
    bool staged;
    uint32_t sbase, pass, cslice;
    uint32_t t, L, nvalid, K, tslice, tres, tpass;
    uint32_t nblk;                  /* item unit: its blocks of 64 (1 .. ITEM_BLOCKS) */
  };
  /* a tile of a chunk: its reference is in LDS; a block of a class-row chunk: 64
     consecutive items */
  auto unit_of = [&](const Claim &cc) -> Unit {
    Unit u;
    u.staged = true;
    u.sbase = cc.b * slice_bytes;
    u.pass = __builtin_amdgcn_readfirstlane(ring[cc.b].pass);
    u.cslice = __builtin_amdgcn_readfirstlane(ring[cc.b].slice);
    u.nblk = 1;
    if (u.pass >= 3) {
      u.t = 0;
      u.L = 0;
      u.nvalid = WAVE;
      u.K = 0;
      u.tslice = u.cslice;
      u.tres = __builtin_amdgcn_readfirstlane(ring[cc.b].first) + cc.k * (ITEM_BLOCKS * WAVE);
      u.tpass = u.pass;
      const uint32_t left = __builtin_amdgcn_readfirstlane(ring[cc.b].pad[1]) - cc.k * ITEM_BLOCKS;
      u.nblk = left < ITEM_BLOCKS ? left : ITEM_BLOCKS;
    } else {
      const TileRef *tr = tref_lds + cc.b * chunk_cap + cc.k;
      const TileDesc td = tr->td;
      u.t = __builtin_amdgcn_readfirstlane(tr->t);
      u.L = __builtin_amdgcn_readfirstlane(td.len);
      u.nvalid = __builtin_amdgcn_readfirstlane(td.nvalid);
      u.K = __builtin_amdgcn_readfirstlane(td.k);
      u.tslice = __builtin_amdgcn_readfirstlane(td.slice);
      u.tres = __builtin_amdgcn_readfirstlane(td.res_base);
      /* the indel passes reuse the tiles of pass 0 and are named by their chunk */
      u.tpass = u.pass ? u.pass : __builtin_amdgcn_readfirstlane(td.pass);
    }
    return u;
  };
  /* The next thing to do when nothing was claimed ahead: a tile of a chunk (waiting
     for the loader if need be), and when the chunks are through, the unstaged tiles.
     Its data is requested here and awaited at once -- the rare way in. */
  auto unit_small = [&](uint32_t t) -> Unit {
    const TileDesc td = P.tiles[t];
    Unit u;
    u.staged = false;
    u.sbase = u.pass = u.cslice = 0;
    u.nblk = 1;
    u.t = t;
    u.L = __builtin_amdgcn_readfirstlane(td.len);
    u.nvalid = __builtin_amdgcn_readfirstlane(td.nvalid);
    u.K = __builtin_amdgcn_readfirstlane(td.k);
    u.tslice = __builtin_amdgcn_readfirstlane(td.slice);
    u.tres = __builtin_amdgcn_readfirstlane(td.res_base);
    u.tpass = __builtin_amdgcn_readfirstlane(td.pass);
    return u;
  };
  uint32_t small_t = 0;                   /* the unstaged tile at hand (cur_c.ok = false) */
  auto next_unit = [&](TileData &d) -> bool {
    if (block_phase) {
      cur_c = claim_tile(true);
      if (cur_c.ok) {
        const Unit u = unit_of(cur_c);
        d = load_tile_data(u.L, u.nvalid, u.tres, u.t, u.tpass);
        return true;
      }
      block_phase = false;
    }
    cur_c.ok = false;
    /* (no such tiles: not even the claim -- 4096 waves asking one counter at the end
       of the launch wait ~30 ns each for one another) */
    if (P.nsmall == 0)
      return false;
    uint32_t i = 0;
    if (lane == 0)
      i = atomicAdd(P.tile_counter + 1, 1u);
    i = __builtin_amdgcn_readfirstlane(i);
    if (i >= P.nsmall)
      return false;
    small_t = P.small_tiles[i];
    const Unit u = unit_small(small_t);
    d = load_tile_data(u.L, u.nvalid, u.tres, u.t, u.tpass);
    return true;
  };

  PT_MARK(PT_OTHER);
  bool have = next_unit(cur);
  PT_MARK(PT_CLAIM);
  /* The loop's shape is deliberate: the next tile is claimed and its data requested
     at the TOP of the body, the registers change hands at the BOTTOM.  With the
     hand-over at the top the compiler rotates the loop, the copy of the loaded
     registers lands right behind the loads and waits for them -- nothing overlaps. */
  while (have) {
    /* (the descriptor is read again rather than carried round the loop: scalar
       registers are scarce) */
    const Unit un = cur_c.ok ? unit_of(cur_c) : unit_small(small_t);
    const bool staged = un.staged;
    const uint32_t sbase = un.sbase, pass = un.pass, cslice = un.cslice;
    const uint32_t t = un.t;
    (void)pass;
    (void)cslice;
    /* the next tile, if one can be had without waiting: its data is in flight
       while this one is worked on */
    nxt_c.ok = false;
    if (staged) {
      nxt_c = claim_tile(false);
      if (nxt_c.ok) {
        const Unit n = unit_of(nxt_c);
        nxt = load_tile_data(n.L, n.nvalid, n.tres, n.t, n.tpass);
      }
    }
    {
      const uint32_t L = un.L;
      const uint32_t nvalid = un.nvalid;
      const uint32_t K = un.K;       /* 0: light tile */
      const uint32_t tslice = un.tslice;
      const uint32_t tres = un.tres;
      const uint32_t tpass_real = un.tpass;
      /* (ablation builds: a skipped tile gets a pass number no branch below takes) */
      const uint32_t tpass = (CMPR_DBG(P, DBG_SKIP_TILES) ||
                              (CMPR_DBG(P, DBG_SKIP_CLASS_TILES) && tpass_real >= 3) ||
                              (CMPR_DBG(P, DBG_SKIP_MAIN_TILES) && tpass_real < 3)) ? 0xffu : tpass_real;
      PT_MARK(PT_TILE_DATA);
      [[maybe_unused]] uint64_t rec_hins = 0;
      [[maybe_unused]] uint32_t rec_ck = 0;
      if constexpr (REC_OK) {
        if (rec_tiles && tpass_real < 3u) {
          /* tile_from_record: what fill_tiles_kernel wrote for the other layouts -- the residues padded with code A
             behind the query's end (its key is zero: a pair that hangs over the end hashes without a test), the
             query's Zobrist hash (zobrist_hash, zobrist.cc:74-88; db_hash, db.cc:903-916) -- from the record */
          const bool v_ = lane < nvalid;
          const uint32_t len_ = v_ ? cur.b : 0u;
          const uint64_t h_rec = ((uint64_t)cur.c << 32) | cur.r7;     /* (rec_hash; r7 is padding from here on) */
          cur.c = 0;
          auto pad_ = [&](uint32_t d, uint32_t w) -> uint32_t {
            const int n = (int)len_ - (int)(4u * w);
            const uint32_t mk = n <= 0 ? 0u : n >= 4 ? 0xffffffffu : (1u << (8 * n)) - 1u;
            return (d & mk) | (PADW & ~mk);
          };
          cur.r0 = pad_(cur.r0, 0); cur.r1 = pad_(cur.r1, 1); cur.r2 = pad_(cur.r2, 2); cur.r3 = pad_(cur.r3, 3);
          cur.r4 = pad_(cur.r4, 4); cur.r5 = pad_(cur.r5, 5); cur.r6 = pad_(cur.r6, 6); cur.r7 = pad_(cur.r7, 7);
          uint64_t hq = 0;
          const uint32_t gv_ = (uint32_t)cur.a, gj_ = (uint32_t)(cur.a >> 32);
          if (GENES && v_ && (INDELS || !rec_hash))
            hq = gk_lds[gv_] ^ gk_lds[P.n_v + gj_];
          uint64_t hi_ = hq;                           /* (-i) zobrist_hash_insert_first, zobrist.cc:122-136 */
          const uint32_t rw_[8] = {cur.r0, cur.r1, cur.r2, cur.r3, cur.r4, cur.r5, cur.r6, cur.r7};
          if (!rec_hash) {
#pragma unroll
          for (uint32_t w = 0; w < 8; w++)
            if (4u * w < L) {                          /* (wave-uniform: the tile's longest query) */
#pragma unroll
              for (uint32_t k = 0; k < 4; k++)
                if (4u * w + k < L) {                  /* (the table ends three positions behind the longest) */
                  const uint32_t r_ = (rw_[w] >> (8u * k)) & 0xffu;
                  hq ^= lds_u64(zl_addr + (ZS * (4u * w + k) + r_) * 8u);
                  if constexpr (INDELS)
                    hi_ ^= lds_u64(zl_addr + (ZS * (4u * w + k + 1u) + r_) * 8u);
                }
            }
          } else {
            /* the hash came with the record (keys_kernel's, gene keys included) */
            if constexpr (INDELS) {
#pragma unroll
              for (uint32_t w = 0; w < 7; w++)         /* (rec_hash: 28 residues at most) */
                if (4u * w < L) {
#pragma unroll
                  for (uint32_t k = 0; k < 4; k++)
                    if (4u * w + k < L) {
                      const uint32_t r_ = (rw_[w] >> (8u * k)) & 0xffu;
                      hi_ ^= lds_u64(zl_addr + (ZS * (4u * w + k + 1u) + r_) * 8u);
                    }
                }
            }
            hq = h_rec;
          }
          cur.a = v_ ? hq : 0ull;
          cur.b = len_;
          if constexpr (INDELS) {
            rec_hins = v_ ? hi_ : 0ull;
            /* the query's class key (class_key_of, layout.h): base of its (V, J) class -- no length term with -i,
               the table says -- and, where that class is split, its class residues (wave-uniform positions) */
            uint32_t ck_ = cb_lds[len_];
            if (GENES)
              ck_ ^= cb_lds[P.geom.off_cv + gv_] ^ cb_lds[P.geom.off_cj + gj_];
            const uint32_t hb_ = ck_ >> (32 - HEAVY_BUCKETS_LOG2);
            if (KH > 0 && ((hv_lds[hb_ >> 5] >> (hb_ & 31u)) & 1u) && len_ > 0u) {
#pragma unroll
              for (uint32_t i = 0; i < MCR; i++)
                if (i < KH) {
                  const uint32_t pp_ = __builtin_amdgcn_readfirstlane(class_pos(L, i, P.geom.c0));
                  uint32_t x_ = rw_[0];
#pragma unroll
                  for (uint32_t w = 1; w < 8; w++)
                    x_ = (pp_ >> 2) == w ? rw_[w] : x_;
                  ck_ ^= cr_lds[i * A + ((x_ >> ((pp_ & 3u) * 8u)) & 0xffu)];
                }
            }
            rec_ck = v_ ? ck_ : 0u;
          }
        }
      }
      /* the tile's counters (nvar further down), 32 bits wide -- a lane's variants of
         one tile fit -- and added to the 64-bit ones once per tile */
      uint32_t treads = 0;
      const uint32_t *qr = P.qres + tres + lane;
      const bool class_tile = tpass_real >= 3;
      /* Pages (layout.h SliceGeom): the staged buffer holds ONE page of the unit's slice; a probe counts in
         this pass iff its hash names that page -- (hash & pgm) == pgv -- and the unit comes by once per page.
         No pages (nearly every slice): pgm = pgv = 0, every probe counts.  What a unit stands for (its variant
         count) is counted in the pass of page 0. */
      uint32_t pgm = 0, pgv = 0;
      if (PAIRS && staged) {
        const uint32_t pg = __builtin_amdgcn_readfirstlane(ring[cur_c.b].page);
        pgm = ((1u << (pg >> 4)) - 1u) << PAGE_HASH_SHIFT;
        pgv = (pg & 7u) << PAGE_HASH_SHIFT;
      }
      const bool page0 = pgv == 0u;
      auto on_page = [&](uint64_t Wk) -> uint32_t {       /* all ones / zero */
        return (((uint32_t)Wk & pgm) == pgv) ? ~0u : 0u;
      };
      /* a class-row item carries the row's blanked hash, the query's slot in pass 0
         (~0: padding behind the items of a slice) and its residue | position << 8
         (query_layout.hip) */
      const bool valid = class_tile ? cur.b != 0xffffffffu : lane < nvalid;
      [[maybe_unused]] const uint32_t vmask = valid ? ~0u : 0u;
      W.qslot = class_tile ? cur.b : t * WAVE + lane;
      const uint64_t cW = cur.a;
      const uint32_t Ll = (valid && !class_tile) ? cur.b : 0u;
      const unsigned char *own_glob = filter + (size_t)tslice * slice_bytes;

      auto woff_of = [&](uint64_t Wk) -> uint32_t {
        /* row_word() x 32 bytes.  (The word number is kept opaque: the compiler would turn
           (x >> 16) << 5 into (x >> 11) & ~31, which cannot merge with the base add that
           follows -- v_lshl_add_u32 does shift and add in one instruction.) */
        uint32_t w = __umul24((uint32_t)(Wk >> 48), nwords) >> 16;
        asm("" : "+v"(w));
        return w << 5;
      };
      auto word_lds = [&](uint32_t wo) -> RowWord {
        RowWord w;
        w.a = lds_u128(sbase + wo);
        w.b = lds_u128(sbase + wo + 16u);
        return w;
      };
      auto word_glob = [&](const unsigned char *base, uint32_t wo) -> RowWord {
        RowWord w;
        w.a = *(const u32x4 *)(base + wo);
        w.b = *(const u32x4 *)(base + wo + 16u);
        return w;
      };
      /* the tile's slice: the staged copy, or where it lies */
      auto fetch_own = [&](uint64_t Wk) -> RowWord {
        const uint32_t wo = woff_of(Wk);
        treads += valid ? 1u : 0u;
        if (staged)
          return word_lds(wo);
        return word_glob(own_glob, wo);
      };
      /* a slice of the lane's own */
      auto fetch_at = [&](uint64_t Wk, uint32_t slice, bool in_lds) -> RowWord {
        const uint32_t wo = woff_of(Wk);
        treads += valid ? 1u : 0u;
        RowWord w;
        if (in_lds)
          w = word_lds(wo);
        else
          w = word_glob(filter + (size_t)slice * slice_bytes, wo);
        return w;
      };
      auto res_at = [&](uint32_t p) -> uint32_t {
        return (qr[(p >> 2) * WAVE] >> ((p & 3u) * 8)) & 0xffu;
      };

      /* ---- the query's hash (db_hash, db.cc:903-916 / zobrist.cc:74-88) was
              computed when the set was laid out, like the reference's
              seqinfo hash; with -i also the two shifted hashes of the rolling
              indel enumeration (zobrist.cc:90-104, 122-136) ---- */
      const uint64_t h = cur.a;

      /* class positions of this length (wave-uniform), as a bit set */
      uint32_t m[MCR];
#pragma unroll
      for (uint32_t i = 0; i < MCR; i++)
        m[i] = P.geom.c0 + i;                 /* (c0 + i) % L when all of them lie inside: no division */
      if (L < P.geom.c0 + MCR) {              /* wave-uniform, rare: a tile of very short sequences */
#pragma unroll
        for (uint32_t i = 0; i < MCR; i++)    /* (the division runs on the vector unit: back to a scalar) */
          m[i] = __builtin_amdgcn_readfirstlane(class_pos(L, i, P.geom.c0));
      }
      uint64_t cpos_lo = 0, cpos_hi = 0;
      bool cpos_far = false;
#pragma unroll
      for (uint32_t i = 0; i < MCR; i++)
        if (i < K) {
          if (m[i] < 64u)
            cpos_lo |= 1ull << m[i];
          else if (m[i] < 128u)
            cpos_hi |= 1ull << (m[i] - 64u);
          else
            cpos_far = true;
        }
      auto is_class_pos = [&](uint32_t p) -> bool {
        if (p < 64u)
          return ((cpos_lo >> p) & 1ull) != 0;
        if (p < 128u)
          return ((cpos_hi >> (p - 64u)) & 1ull) != 0;
        bool c = false;
        if (cpos_far) {
#pragma unroll
          for (uint32_t i = 0; i < MCR; i++)
            c = c || (i < K && m[i] == p);
        }
        return c;
      };
      /* class-key terms of residue r at position p (zero when p is no class position) */
      auto class_term = [&](uint32_t p, uint32_t r) -> uint32_t {
        uint32_t dk = 0;
#pragma unroll
        for (uint32_t i = 0; i < MCR; i++)
          if (i < K && m[i] == p)
            dk ^= cr_lds[i * A + r];
        return dk;
      };

      uint32_t nvar = 0;

      /* queue the positives of a block of substitution rows: bit A * j + v of
         (m0, m1) <-> residue v at row j of the block; rpack holds the lanes' own
         residues of those rows */
      /* queue the positives of a block of substitution rows: bit A * j + v of m0 <->
         residue v at row j of the block; rpack holds the lanes' own residues of those
         rows (one entry per lane with a positive: q_push) */
      static_assert(RB <= RPW && RB * A <= 60 && RB * RBITS <= 17, "a block of rows is one entry");
      auto emit_sub_rows = [&](uint64_t m0, uint32_t p0, uint32_t rpack) {
        if (CMPR_DBG(P, DBG_SKIP_EMIT))
          m0 = 0;
        q_push<A, D, GENES, INLINE, PAIRS>(W, zl_addr, m0 != 0, h, K_ROWS | (p0 << 3) | (rpack << 19), (uint32_t)m0,
                                    (uint32_t)(m0 >> 32) | ((rpack >> 13) << 28),
                                    (uint32_t)__popcll((unsigned long long)m0));
      };

      if (tpass == 0) {
        /* ---- the unchanged sequence (variants.cc:260-268) ---- */
        {
          const RowWord w = fetch_own(h);
          const bool hit = ((row_bits(w, h) >> A) & 1u) != 0 && on_page(h) != 0u;
          q_push<A, D, GENES, INLINE, PAIRS>(W, zl_addr, valid && hit, h, pack_a(K_SAME, 0, 0), 0, 1u, 1u);
          nvar += 1;
        }

        if constexpr (PAIRS) {
          /* ---- single substitutions (variants.cc:280-293), pair rows: one word read per
                  PAIR of positions answers the 2 (A - 1) variants of both (pair_bits), PB
                  pairs per block.  A block is branch-free -- pairs past the end of the
                  tile, past the end of a shorter query of it and the class pairs of a heavy
                  tile (left to their items) are computed and masked. ---- */
          constexpr int PB = A == 20 ? 3 : 2;            /* pairs per block */
          constexpr uint32_t PP = 2u * PB;               /* positions per block: 6 or 4, both divide 24 */
          const uint32_t nd = (L + 3u) >> 2;
          uint32_t s0 = cur.r0, s1 = cur.r1, s2 = cur.r2, s3 = cur.r3, s4 = cur.r4, s5 = cur.r5;
          uint32_t nrows = 0;                     /* words read (wave-uniform) */
          const uint32_t zlast = P.zpos - 1u;
          constexpr uint32_t GAPBIT = 1u << (A + 1);
          const uint32_t delmask = Ll > 1u ? AMASK | GAPBIT : AMASK;     /* (-i) */
          uint32_t carry = 31u;                   /* (-i) the residue in front of the block: none yet */
          uint32_t ndel = 0;                      /* (-i) deletion variants these rows stand for */
          auto pair_rows = [&](auto staged_c) {
          constexpr bool STAGED = decltype(staged_c)::value;
          for (uint32_t p0 = 0; p0 < (CMPR_DBG(P, DBG_SKIP_LDS_ROWS) ? 0u : L); p0 += PP) {
            if (p0 && p0 % (4u * TDW) == 0) {
              const uint32_t w0 = p0 >> 2;
              if (REC_OK && rec_tiles) {            /* (record tiles: positions 24 .. 31 came with the record) */
                s0 = cur.r6;
                s1 = cur.r7;
                s2 = s3 = s4 = s5 = PADW;
              } else {
              s0 = qr[w0 * WAVE];
              s1 = w0 + 1u < nd ? qr[(w0 + 1u) * WAVE] : PADW;
              s2 = w0 + 2u < nd ? qr[(w0 + 2u) * WAVE] : PADW;
              s3 = w0 + 3u < nd ? qr[(w0 + 3u) * WAVE] : PADW;
              s4 = w0 + 4u < nd ? qr[(w0 + 4u) * WAVE] : PADW;
              s5 = w0 + 5u < nd ? qr[(w0 + 5u) * WAVE] : PADW;
              }
            }
            const uint64_t rr = ((uint64_t)s1 << 32) | s0;
            if constexpr (PP == 4) {
              s0 = s1; s1 = s2; s2 = s3; s3 = s4; s4 = s5; s5 = PADW;
            } else {
              static_assert(PP == 4 || PP == 6, "six bytes of the shift register per block");
              s0 = __builtin_amdgcn_alignbit(s2, s1, 16);
              s1 = __builtin_amdgcn_alignbit(s3, s2, 16);
              s2 = __builtin_amdgcn_alignbit(s4, s3, 16);
              s3 = __builtin_amdgcn_alignbit(s5, s4, 16);
              s4 = __builtin_amdgcn_alignbit(PADW, s5, 16);
              s5 = PADW;
            }
            /* class positions among the positions of the block (wave-uniform bits) */
            uint32_t cbits = 0;
            if (K) {
              if (p0 + PP <= 64u)
                cbits = (uint32_t)(cpos_lo >> p0);
              else
                for (uint32_t j = 0; j < PP; j++)
                  cbits |= is_class_pos(p0 + j) ? (1u << j) : 0u;
            }
            cbits = __builtin_amdgcn_readfirstlane(cbits);
            uint64_t Wk[PB];
            uint32_t ra[PB], rb[PB], wo[PB];
            /* which positions of the block the lane's own query still has (tiles may mix
               lengths): bit i <-> p0 + i < Ll */
            const int left = (int)Ll - (int)p0;
            const uint32_t lv = bfm_u32((uint32_t)(left < 0 ? 0 : left > 31 ? 31 : left));
            /* the pairs' own keys, all 2 PB reads in flight together (a query that ends inside
               or in front of the pair carries code A there, whose key is zero) ... */
#pragma unroll
            for (int j = 0; j < PB; j++) {
              const uint32_t p = p0 + 2u * (uint32_t)j;
              ra[j] = (uint32_t)(rr >> (16 * j)) & 31u;
              rb[j] = (uint32_t)(rr >> (16 * j + 8)) & 31u;
              uint32_t zrow = zl_addr + ZS * 8u * min(p, zlast);
              asm("" : "+s"(zrow));
              uint32_t zrow2 = zl_addr + ZS * 8u * min(p + 1u, zlast);
              asm("" : "+s"(zrow2));
              Wk[j] = h ^ lds_u64(zrow + ra[j] * 8u) ^ lds_u64(zrow2 + rb[j] * 8u);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < PB; j++)
              wo[j] = woff_of(Wk[j]);
            /* ... then the filter words, one pair ahead of the pair being tested */
            RowWord wc = STAGED ? word_lds(wo[0]) : word_glob(own_glob, wo[0]);
            uint32_t xa[PB], xb[PB];
#pragma unroll
            for (int j = 0; j < PB; j++) {
              RowWord wn = wc;
              if (j + 1 < PB)
                wn = STAGED ? word_lds(wo[j + 1]) : word_glob(own_glob, wo[j + 1]);
              __builtin_amdgcn_sched_barrier(0);
              /* (a class pair of a heavy tile: nothing of it counts here -- a scalar select) */
              const uint32_t am = ((cbits >> (2 * j)) & 3u) ? 0u : (INDELS ? AMASK | GAPBIT : AMASK);
              uint32_t xs, xd;
              pair_bits(wc, Wk[j], xs, xd);
              uint32_t a1, a2;
              pair_answers(xs, xd, ra[j], rb[j], a1, a2);
              /* own residue out; the lane's own length as a mask of all ones or none */
              a1 &= ~(1u << ra[j]);
              a2 &= ~(1u << rb[j]);
              if constexpr (INDELS) {
                /* -i: bit A + 1 ("gap") of either side answers "q without that position" (the
                   gap pairs of build_rows_kernel); the variant exists once per run of equal
                   residues, at its first position (variants.cc:301-325), and not for a query
                   of one residue */
                const uint32_t before = j == 0 ? carry : rb[j - 1];
                const uint32_t ta = (ra[j] != before ? delmask : AMASK) & am & (uint32_t)bfe_i32(lv, 2 * j, 1);
                const uint32_t tb = (rb[j] != ra[j] ? delmask : AMASK) & am & (uint32_t)bfe_i32(lv, 2 * j + 1, 1);
                const uint32_t pm = on_page(Wk[j]);
                xa[j] = a1 & ta & pm;
                xb[j] = a2 & tb & pm;
                ndel += (ta >> (A + 1)) + (tb >> (A + 1));
              } else {
                const uint32_t pm = on_page(Wk[j]);
                xa[j] = a1 & am & (uint32_t)bfe_i32(lv, 2 * j, 1) & pm;
                xb[j] = a2 & am & (uint32_t)bfe_i32(lv, 2 * j + 1, 1) & pm;
              }
              wc = wn;
            }
            carry = rb[PB - 1];
            nrows += (uint32_t)PB;
            PT_MARK(PT_ROWS);
            {
              uint32_t any = 0;
#pragma unroll
              for (int j = 0; j < PB; j++)
                any |= xa[j] | xb[j];
              if (CMPR_DBG(P, DBG_SKIP_EMIT))
                any = 0;
              if (__ballot(any != 0u)) {
#pragma unroll
                for (int j = 0; j < PB; j++)
                  q_push<A, D, GENES, INLINE, PAIRS>(W, zl_addr, (xa[j] | xb[j]) != 0u && any != 0u, h,
                                              K_PAIR | ((p0 + 2u * (uint32_t)j) << 3) | (ra[j] << 19) | (rb[j] << 24),
                                              xa[j], xb[j], (uint32_t)__popc(xa[j]) + (uint32_t)__popc(xb[j]));
              }
            }
            PT_MARK(PT_EMIT);
          }
          };
          if (staged)
            pair_rows(std::true_type{});
          else
            pair_rows(std::false_type{});
          /* the variants these rows stand for: A - 1 per position of the lane's query that is
             not in a class pair (wave-uniform positions: mixed tiles hold no wrapped ones) */
          uint32_t ncls = 0;
#pragma unroll
          for (uint32_t i = 0; i < MCR; i++)
            if (i < K) {
              const uint32_t pp = m[i] & ~1u;
              bool fresh = true;
#pragma unroll
              for (uint32_t k = 0; k < MCR; k++)
                if (k < i && (m[k] & ~1u) == pp)
                  fresh = false;
              if (fresh)
                ncls += (pp < Ll ? 1u : 0u) + (pp + 1u < Ll ? 1u : 0u);
            }
          nvar += (Ll - ncls) * (uint32_t)(A - 1) + ndel;
          treads += valid ? nrows : 0u;
        }

        if (D >= 1 && !PAIRS) {
          /* ---- single substitutions (variants.cc:280-293), a row per position, RB
                  rows per block.  A block is branch-free, so that its 2 RB LDS
                  reads (own key, filter word) are in flight together: rows past
                  the end of the tile and the class positions of a heavy tile
                  (left to their own pass) are computed and masked. ---- */
          const uint32_t nd = (L + 3u) >> 2;
          /* the residues of positions p0 .. p0 + RB - 1 lie in two dwords (RB = 6:
             p0 % 4 is 0 or 2; RB = 8: aligned); requested one block ahead */
          /* the residues travel with the tile's data, 24 positions at a time: six
             dwords used as one shift register, the block's residues always in the low
             bytes of the first (no indexing, no load inside the rows; sequences longer
             than 24 refill it where their residues lie) */
          uint32_t s0 = cur.r0, s1 = cur.r1, s2 = cur.r2, s3 = cur.r3, s4 = cur.r4, s5 = cur.r5;
          uint32_t nlive = 0;                     /* rows of this lane that count (x A - 1 variants) */
          uint32_t nrows = 0;                     /* rows read (wave-uniform) */
          auto sub_rows = [&](auto staged_c) {
          constexpr bool STAGED = decltype(staged_c)::value;
          for (uint32_t p0 = 0; p0 < (CMPR_DBG(P, DBG_SKIP_LDS_ROWS) ? 0u : L); p0 += RB) {
            if (p0 && p0 % (4u * TDW) == 0) {
              const uint32_t w0 = p0 >> 2;
              s0 = qr[w0 * WAVE];
              s1 = w0 + 1u < nd ? qr[(w0 + 1u) * WAVE] : 0u;
              s2 = w0 + 2u < nd ? qr[(w0 + 2u) * WAVE] : 0u;
              s3 = w0 + 3u < nd ? qr[(w0 + 3u) * WAVE] : 0u;
              s4 = w0 + 4u < nd ? qr[(w0 + 4u) * WAVE] : 0u;
              s5 = w0 + 5u < nd ? qr[(w0 + 5u) * WAVE] : 0u;
            }
            const uint64_t rr = ((uint64_t)s1 << 32) | s0;
            if constexpr (RB == 4) {
              s0 = s1; s1 = s2; s2 = s3; s3 = s4; s4 = s5; s5 = 0;
            } else {
              constexpr uint32_t SH = 8u * (RB & 3);
              s0 = __builtin_amdgcn_alignbit(s1, s0, SH);
              s1 = __builtin_amdgcn_alignbit(s2, s1, SH);
              s2 = __builtin_amdgcn_alignbit(s3, s2, SH);
              s3 = __builtin_amdgcn_alignbit(s4, s3, SH);
              s4 = __builtin_amdgcn_alignbit(s5, s4, SH);
              s5 >>= SH;
            }
            /* class positions among the rows of the block (wave-uniform bits) */
            uint32_t cbits = 0;
            if (K) {
              if (p0 + RB <= 64u)
                cbits = (uint32_t)(cpos_lo >> p0);
              else
                for (uint32_t j = 0; j < (uint32_t)RB; j++)
                  cbits |= is_class_pos(p0 + j) ? (1u << j) : 0u;
            }
            /* (wave-uniform, but the bit set may live in vector registers: the rows test it
               with scalar instructions) */
            cbits = __builtin_amdgcn_readfirstlane(cbits);
            uint64_t m0 = 0, m1 = 0;
            uint32_t rpack = 0;
            uint64_t Wk[RB];
            uint32_t rj[RB], wo[RB];
            /* the rows' own keys, all RB reads in flight together ... */
#pragma unroll
            for (int j = 0; j < RB; j++) {
              rj[j] = (uint32_t)(rr >> (8 * j)) & 31u;
              rpack |= (rj[j] & RMASK) << (RBITS * j);
              /* (the row's base address as ONE scalar: residue << 3 + base is one instruction) */
              uint32_t zrow = zl_addr + ZS * 8u * (p0 + (uint32_t)j);
              asm("" : "+s"(zrow));
              Wk[j] = lds_u64(zrow + rj[j] * 8u);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < RB; j++) {
              Wk[j] ^= h;
              wo[j] = woff_of(Wk[j]);
            }
            /* ... then the filter words, one row ahead of the row being tested */
            RowWord wc = STAGED ? word_lds(wo[0]) : word_glob(own_glob, wo[0]);
#pragma unroll
            for (int j = 0; j < RB; j++) {
              const uint32_t p = p0 + (uint32_t)j;
              RowWord wn = wc;
              if (j + 1 < RB)
                wn = STAGED ? word_lds(wo[j + 1]) : word_glob(own_glob, wo[j + 1]);
              __builtin_amdgcn_sched_barrier(0);
              /* (a class position of a heavy tile: no residue of the row counts -- a scalar
                 select, so that the lane's own length is the only vector test of the row) */
              const uint32_t am = ((cbits >> j) & 1u) ? 0u : AMASK;
              const uint32_t bits = row_bits(wc, Wk[j]);
              uint32_t x = bits & am & ~(1u << rj[j]);
              const bool live = p < Ll;
              x = live ? x : 0u;
              nlive += (live && am) ? 1u : 0u;
              if (j < RPW)
                m0 |= (uint64_t)x << (A * j);
              else
                m1 |= (uint64_t)x << (A * (j - RPW));
              wc = wn;
            }
            nrows += (uint32_t)RB;
            PT_MARK(PT_ROWS);
            emit_sub_rows(m0 | m1, p0, rpack);
            PT_MARK(PT_EMIT);
          }
          };
          if (staged)
            sub_rows(std::true_type{});
          else
            sub_rows(std::false_type{});
          nvar += nlive * (uint32_t)(A - 1);
          treads += valid ? nrows : 0u;
        }

        if (D >= 2) {
          /* ---- double substitutions (variants.cc:370-399).  For the pair of
                  positions (b, e) -- b blanked, e substituted by w -- one word read
                  per w answers every replacement at b.  When exactly one of the two
                  is a class position it takes the role of b (its term then drops
                  out of the slice key); rows that change the class go to the
                  filter where it lies. ---- */
          const uint32_t qck = K ? P.qck[W.qslot] : 0u;      /* the query's class key */
          for (uint32_t pa = 0; pa + 1 < L; pa++) {
            const bool ca_cls = is_class_pos(pa);
            for (uint32_t pb = pa + 1; pb < L; pb++) {
              const bool cb_cls = is_class_pos(pb);
              const bool swap = cb_cls && !ca_cls;               /* wave-uniform */
              const uint32_t b = swap ? pb : pa, e = swap ? pa : pb;
              const bool b_cls = swap ? cb_cls : ca_cls, e_cls = swap ? ca_cls : cb_cls;
              const uint32_t rb = res_at(b), re = res_at(e);
              const uint32_t ea = zl_addr + (ZS * e + re) * 8u;
              const uint64_t hb = h ^ lds_u64(zl_addr + (ZS * b + rb) * 8u) ^ lds_u64(ea);
              const bool own = !b_cls && !e_cls;
              const uint32_t key_b = b_cls ? qck ^ class_term(b, rb) : qck;
              int ci_b = -1;                           /* class part of a row blanked at b */
#pragma unroll
              for (uint32_t i = 0; i < MCR; i++)
                if (b_cls && ci_b < 0 && i < K && m[i] == b)
                  ci_b = (int)i;
              const uint32_t te_own = e_cls ? class_term(e, re) : 0u;
              const bool live = pb < Ll;                          /* pb < Ll implies pa < Ll */
              nvar += live ? (uint32_t)((A - 1) * (A - 1)) : 0u;
              /* Exactly one class position (it is b): every row of the pair lies in ONE slice of b's class part,
                 the slice the query's ITEM of that position is grouped by -- the item reads them there, from LDS
                 (below, passes >= 3; round 5).  Here they were 37 % of a query's reads, each a random 32 bytes of
                 the filter in HBM: 24.2M sequences against themselves, 230 ms.  Left to this loop: queries whose
                 class positions wrap or that do not fit a record (36 residues), lane by lane. */
              bool rd = live;
              if (b_cls && e_cls && CMPR_DBG(P, DBG_SKIP_HBM_ROWS))
                continue;                    /* (ablation: the rows of two class positions, read where the filter lies) */
              if (b_cls && !e_cls) {
                rd = live && !(Ll >= P.geom.c0 + K && Ll <= 36u);
                if (!__ballot(rd))
                  continue;
              }
              for (uint32_t k0 = 1; k0 < (uint32_t)A; k0 += RB) {
                uint64_t m0 = 0, m1 = 0;
#pragma unroll
                for (int j = 0; j < RB; j++) {
                  const uint32_t k = k0 + (uint32_t)j;
                  if (k < (uint32_t)A) {
                    const uint64_t Wk = hb ^ lds_u64(ea + 8u * k);     /* e <- (re + k) mod A */
                    RowWord w;
                    if (own) {
                      w = fetch_own(Wk);
                    } else {
                      uint32_t wres = re + k;
                      wres = wres >= (uint32_t)A ? wres - (uint32_t)A : wres;
                      const uint32_t key = key_b ^ (e_cls ? te_own ^ class_term(e, wres) : 0u);
                      const uint32_t sl = row_slice(P.geom, key, ci_b);
                      w = fetch_at(Wk, sl, staged && sl == tslice);
                    }
                    uint32_t x = row_bits(w, Wk) & AMASK & ~(1u << rb);
                    x = rd ? x : 0u;
                    if (j < RPW)
                      m0 |= (uint64_t)x << (A * j);
                    else
                      m1 |= (uint64_t)x << (A * (j - RPW));
                  }
                }
                while (__ballot((m0 | m1) != 0)) {
                  const bool pos = (m0 | m1) != 0;
                  const bool first = m0 != 0;
                  const uint64_t mm = first ? m0 : m1;
                  const uint32_t idx = pos ? (uint32_t)__ffsll((unsigned long long)mm) - 1u : 0u;
                  uint32_t j = idx / (uint32_t)A;
                  const uint32_t v = idx - j * (uint32_t)A;
                  j += (pos && !first) ? (uint32_t)RPW : 0u;
                  const uint32_t k = k0 + j;
                  uint32_t wres = re + k;
                  wres = wres >= (uint32_t)A ? wres - (uint32_t)A : wres;
                  const uint64_t hv = hb ^ lds_u64(ea + 8u * k) ^ lds_u64(zl_addr + (ZS * b + v) * 8u);
                  /* (position, residue) pairs in increasing position order */
                  const uint32_t p1 = swap ? e : b, r1 = swap ? wres : v;
                  const uint32_t p2 = swap ? b : e, r2 = swap ? v : wres;
                  q_push<A, D, GENES, INLINE, PAIRS>(W, zl_addr, pos, hv, pack_a(K_SUB2, p1, r1), p2 | (r2 << 24), 1u, 1u);
                  if (first)
                    m0 &= m0 - 1ull;
                  else
                    m1 &= m1 - 1ull;
                }
              }
            }
          }
        }
      } else if (tpass >= 3 && tpass != 0xffu) {
        /* ---- items: rows and variants that are not filed under the slice staged for
                their tile, 64 of them per block, grouped by the slice they ARE filed
                under (query_layout.hip for_each_item): the substitution row of a class
                position (K_SUB) and the insertion row blanked at one (K_INS), in that
                position's class part; a deletion variant (K_DEL), a whole sequence
                looked up by its code-A entry in the main part ---- */
        if (D >= 1) {
          /* the unit's further blocks: requested together now, worked on after the
             first (whose data came with the claim) */
          const uint32_t nblk = un.nblk;
          /* (d = 2: a block is ~230 word reads per item -- its successor's 16 bytes are fetched when it is
             their turn, not kept in twelve registers meanwhile) */
          constexpr bool AHEAD = D < 2;
          uint64_t ea[ITEM_BLOCKS - 1];
          uint32_t eb[ITEM_BLOCKS - 1], ec[ITEM_BLOCKS - 1];
#pragma unroll
          for (uint32_t r = 1; r < ITEM_BLOCKS; r++) {
            ea[r - 1] = 0;
            eb[r - 1] = 0xffffffffu;
            ec[r - 1] = 0;
            if (AHEAD && r < nblk) {
              const ItemRec it = P.items[tres + r * WAVE + lane];
              ea[r - 1] = it.w;
              eb[r - 1] = it.main;
              ec[r - 1] = it.rp;
            }
          }
          uint64_t iw = cW;
          uint32_t im = cur.b, ic = cur.c;
          for (uint32_t r = 0; r < nblk; r++) {
            const bool ival = im != 0xffffffffu;          /* (~0: padding behind the items of a slice) */
            W.qslot = im;
            treads += ival ? 1u : 0u;
            const uint32_t wo = woff_of(iw);
            const RowWord w = staged ? word_lds(wo) : word_glob(own_glob, wo);
            if constexpr (PAIRS) {
              /* a class pair (query_layout.hip for_each_item): pair-blanked hash, own residues
                 (second = A: the query ends with the first), first position; with -i its two
                 deletion answers, and the insertion pairs (K_INS): hash of q with a gap at g
                 and q[g] blanked, q[g] (A: the gap is at the end), q[g - 1] (31: none), g */
              const uint32_t ira = ic & 31u, f2 = (ic >> 5) & 31u, p = (ic >> 10) & 0x3fffu;
              const bool is_ins = INDELS && ((ic >> 24) & 7u) == K_INS;
              const uint32_t irb = is_ins ? ira : f2;
              uint32_t xs, xd, a1, a2;
              pair_bits(w, iw, xs, xd);
              pair_answers(xs, xd, ira, irb, a1, a2);
              constexpr uint32_t GAPBIT = 1u << (A + 1);
              const uint32_t vm = ival ? AMASK : 0u;
              uint32_t m1 = vm & ~(1u << ira), m2 = (irb < (uint32_t)A ? vm : 0u) & ~(1u << irb);
              uint32_t nv = (uint32_t)(A - 1) * (irb < (uint32_t)A ? 2u : 1u);
              if constexpr (INDELS) {
                if (is_ins) {
                  /* in front of g: v != q[g - 1]; in front of g + 1 (if q has a position g): w != q[g] */
                  m1 = vm & ~(1u << f2);
                  nv = (p == 0u ? (uint32_t)A : (uint32_t)(A - 1)) + (ira < (uint32_t)A ? (uint32_t)(A - 1) : 0u);
                } else {
                  const uint32_t d1 = (ic & ITEM_DEL_COUNTS) ? 1u : 0u, d2 = (ic & ITEM_DEL2_COUNTS) ? 1u : 0u;
                  m1 |= ival && d1 ? GAPBIT : 0u;
                  m2 |= ival && d2 ? GAPBIT : 0u;
                  nv += d1 + d2;
                }
              }
              a1 &= m1 & on_page(iw);
              a2 &= m2 & on_page(iw);
              nvar += ival ? nv : 0u;
              if (CMPR_DBG(P, DBG_SKIP_EMIT))
                a1 = a2 = 0;
              const bool pos = (a1 | a2) != 0u;
              if (__ballot(pos)) {
                /* (a substitution pair's entry carries the query's hash, like the pairs of a tile) */
                uint64_t hq = iw;
                if (pos && !is_ins) {
                  hq = iw ^ lds_u64(zl_addr + (ZS * p + ira) * 8u);
                  if (irb < (uint32_t)A)
                    hq ^= lds_u64(zl_addr + (ZS * (p + 1u) + irb) * 8u);
                }
                const uint32_t eca = is_ins ? K_INSROWS | (p << 3) | (ira << 19)
                                            : K_PAIR | (p << 3) | (ira << 19) | (irb << 24);
                q_push<A, D, GENES, INLINE, PAIRS>(W, zl_addr, pos, hq, eca, a1, a2,
                                                   (uint32_t)__popc(a1) + (uint32_t)__popc(a2));
              }
              iw = ea[0];
              im = eb[0];
              ic = ec[0];
#pragma unroll
              for (uint32_t q = 0; q + 2 < ITEM_BLOCKS; q++) {
                ea[q] = ea[q + 1];
                eb[q] = eb[q + 1];
                ec[q] = ec[q + 1];
              }
              continue;
            }
            const uint32_t icr = ic & 0xffu;
            const uint32_t kind = (ic >> 24) & 7u, p = (ic >> 8) & 0xffffu;
            const uint32_t bits = row_bits(w, iw);
            uint32_t x = (kind == K_DEL ? (bits >> A) & 1u : bits & AMASK & ~(1u << icr)) & (ival ? ~0u : 0u);
            nvar += !ival ? 0u : kind == K_DEL ? 1u : (icr == 31u ? (uint32_t)A : (uint32_t)(A - 1));
            /* one entry per item with positives: a row (K_SUB, K_INS: blanked hash + mask of
               residues) or the deletion variant itself */
            if (CMPR_DBG(P, DBG_SKIP_EMIT))
              x = 0;
            q_push<A, D, GENES, INLINE, PAIRS>(W, zl_addr, x != 0, iw, pack_a(kind, p, 0), 0, x,
                                        (uint32_t)__popc(x));
            if constexpr (D >= 2) {
              /* ---- d = 2: the double substitutions with THIS class position blanked and one other, non-class
                      position e substituted (variants.cc:370-399) -- every one of their rows lies in the slice
                      staged for this block (the item's own: the key loses the blanked position's term, and e
                      carries none).  iw = the query's hash with the class position blanked; the query's residues
                      come with its record.  The variants were counted by the tile (main pass). ---- */
              const uint32_t KI = P.geom.k;
              const QueryRec *rq = P.qrec + (ival ? im : 0u);
              const uint32_t Lq = ival ? rq->len : 0u;
              const bool ext = ival && kind == K_SUB && Lq >= P.geom.c0 + KI && Lq <= 36u;
              if (__ballot(ext)) {
                uint32_t lmax = ext ? Lq : 0u;            /* the longest query among the lanes that take part */
                for (int off = 32; off > 0; off >>= 1)
                  lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, off, WAVE));
                lmax = __builtin_amdgcn_readfirstlane(lmax);
                /* (the record's residue dwords one after the other, the next one in flight while four
                   positions are worked on: nine registers of them spilled) */
                uint32_t dcur = ext ? rq->res[0] : 0u, dnext = 0;
                for (uint32_t e = 0; e < lmax; e++) {
                  if ((e & 3u) == 0u) {
                    if (e)
                      dcur = dnext;
                    dnext = (ext && e + 4u < 36u) ? rq->res[(e >> 2) + 1u] : 0u;
                  }
                  if (e >= P.geom.c0 && e < P.geom.c0 + KI)
                    continue;                      /* a class position: both of them class -- the tile reads those */
                  const uint32_t re = (dcur >> ((e & 3u) * 8u)) & 31u;
                  const bool lv2 = ext && e < Lq;
                  const uint32_t ea2 = zl_addr + (ZS * e + (lv2 ? re : 0u)) * 8u;
                  const uint64_t hb2 = iw ^ lds_u64(ea2);
                  for (uint32_t k0 = 1; k0 < (uint32_t)A; k0 += RB) {
                    uint64_t m0 = 0, m1 = 0;
#pragma unroll
                    for (int j = 0; j < RB; j++) {
                      const uint32_t k = k0 + (uint32_t)j;
                      if (k < (uint32_t)A) {
                        const uint64_t Wk = hb2 ^ lds_u64(ea2 + 8u * k);     /* e <- (re + k) mod A */
                        const uint32_t wo2 = woff_of(Wk);
                        treads += lv2 ? 1u : 0u;
                        const RowWord w2 = staged ? word_lds(wo2) : word_glob(own_glob, wo2);
                        uint32_t xb = row_bits(w2, Wk) & AMASK & ~(1u << icr);
                        xb = lv2 ? xb : 0u;
                        if (CMPR_DBG(P, DBG_SKIP_EMIT))
                          xb = 0;
                        if (j < RPW)
                          m0 |= (uint64_t)xb << (A * j);
                        else
                          m1 |= (uint64_t)xb << (A * (j - RPW));
                      }
                    }
                    while (__ballot((m0 | m1) != 0)) {
                      const bool pos = (m0 | m1) != 0;
                      const bool first = m0 != 0;
                      const uint64_t mm = first ? m0 : m1;
                      const uint32_t idx = pos ? (uint32_t)__ffsll((unsigned long long)mm) - 1u : 0u;
                      uint32_t j = idx / (uint32_t)A;
                      const uint32_t v = idx - j * (uint32_t)A;
                      j += (pos && !first) ? (uint32_t)RPW : 0u;
                      const uint32_t k = k0 + j;
                      uint32_t wres = re + k;
                      wres = wres >= (uint32_t)A ? wres - (uint32_t)A : wres;
                      const uint64_t hv = hb2 ^ lds_u64(ea2 + 8u * k) ^ lds_u64(zl_addr + (ZS * p + v) * 8u);
                      /* (position, residue) pairs in increasing position order */
                      const bool cfirst = p < e;
                      const uint32_t p1 = cfirst ? p : e, r1 = cfirst ? v : wres;
                      const uint32_t p2 = cfirst ? e : p, r2 = cfirst ? wres : v;
                      q_push<A, D, GENES, INLINE, PAIRS>(W, zl_addr, pos, hv, pack_a(K_SUB2, p1, r1), p2 | (r2 << 24), 1u, 1u);
                      if (first)
                        m0 &= m0 - 1ull;
                      else
                        m1 &= m1 - 1ull;
                    }
                  }
                }
              }
            }
            /* next block's data moves up */
            if constexpr (AHEAD) {
              iw = ea[0];
              im = eb[0];
              ic = ec[0];
#pragma unroll
              for (uint32_t q = 0; q + 2 < ITEM_BLOCKS; q++) {
                ea[q] = ea[q + 1];
                eb[q] = eb[q + 1];
                ec[q] = ec[q + 1];
              }
            } else if (r + 1u < nblk) {
              const ItemRec it = P.items[tres + (r + 1u) * WAVE + lane];
              iw = it.w;
              im = it.main;
              ic = it.rp;
            }
          }
        }
      }

      if (INDELS && tpass == 0) {
        /* the shifted hash that seeds the rolling gap hashes of the insertion rows (the
           deletion variants were answered with the substitution rows, above) */
        uint64_t h_ins = 0;
        if (rec_tiles)
          h_ins = rec_hins;
        else if (valid)
          h_ins = P.qhins[W.qslot];
        /* Indel variants change the length, hence the class.  t = the variant:
             base(t) = base(q) ^ CL[L] ^ CL[L'],  split iff heavy(base(t)),
           and its rows are filed under base(t) ^ (class residues of t other than
           the blanked position).  Most of them fall into the slice staged for
           this pass; what does not is an item of a later pass (query_layout.hip
           for_each_item) or, rarely, read where it lies. */
        const uint32_t cl_L = P.geom.ctab[L];
        const uint32_t qck = rec_tiles ? rec_ck : P.qck[W.qslot];     /* the query's class key */
        /* residue of a wave-uniform position: from the tile's data (a select chain kept
           opaque, or the compiler turns it into an indexed read of a scratch copy) */
        auto res_reg = [&](uint32_t pp) -> uint32_t {
          if (pp >= 4u * TDW) {
            if (REC_OK && rec_tiles)               /* (record tiles: positions 24 .. 31 came with the record) */
              return ((pp >= 28u ? cur.r7 : cur.r6) >> ((pp & 3u) * 8u)) & 0xffu;
            return res_at(pp);
          }
          const uint32_t wq = pp >> 2;
          uint32_t x = cur.r0;
          asm volatile("" : "+v"(x));
          x = wq == 1u ? cur.r1 : x;
          asm volatile("" : "+v"(x));
          x = wq == 2u ? cur.r2 : x;
          asm volatile("" : "+v"(x));
          x = wq == 3u ? cur.r3 : x;
          asm volatile("" : "+v"(x));
          x = wq == 4u ? cur.r4 : x;
          asm volatile("" : "+v"(x));
          x = wq == 5u ? cur.r5 : x;
          return (x >> ((pp & 3u) * 8u)) & 0xffu;
        };
        uint32_t cbase = 0;                      /* XOR_i CR[i][q[m_i]], heavy tiles */
#pragma unroll
        for (uint32_t i = 0; i < MCR; i++)
          if (i < K && L > 0)
            cbase ^= cr_lds[i * A + res_reg(m[i])];
        const uint32_t base_q = qck ^ cbase;
        auto heavy_of = [&](uint32_t bs) -> uint32_t {
          const uint32_t b = bs >> (32 - HEAVY_BUCKETS_LOG2);
          return (KH > 0 && ((hv_lds[b >> 5] >> (b & 31u)) & 1u)) ? ~0u : 0u;
        };
        /* the tile's residue dwords as a shift register (as in the substitution rows) */
        uint32_t s0 = cur.r0, s1 = cur.r1, s2 = cur.r2, s3 = cur.r3, s4 = cur.r4, s5 = cur.r5;
        const uint32_t nd = (L + 3u) >> 2;

        /* ---- insertions (variants.cc:329-353), pair rows: t = q with v in front of
                position g has the pair (v, q[g]) at (g, g + 1), t' = q with w in front of
                g + 1 the pair (q[g], w) there, and with both positions blanked the two are
                the same string -- q with a gap at g and q[g] blanked -- so ONE word, under
                W2(g) = gap hash of g ^ Z[g+1][q[g]], answers the insertions in front of g
                (first residue varies) and in front of g + 1 (second varies).  PB pairs per
                block: the rolling keys are read together, the filter words one pair ahead.
                (Behind its end a query carries code A, whose keys are zero: the gap hash
                rolls on unchanged, and q[g] = A is "the sequence ends here".) ---- */
        if (!CMPR_DBG(P, DBG_SKIP_INS_ROWS)) {
          /* (two pairs per block, also for amino acids: four keys per pair are in flight here,
             and with three pairs the tile's registers no longer fit) */
          constexpr int PB = 2;
          constexpr uint32_t PP = 2u * PB;
          s0 = cur.r0; s1 = cur.r1; s2 = cur.r2; s3 = cur.r3; s4 = cur.r4; s5 = cur.r5;
          const uint32_t base_t = base_q ^ cl_L ^ P.geom.ctab[L + 1];
          const uint32_t hvy = heavy_of(base_t);
          uint32_t mi[MCR], lo[MCR], hi[MCR];
#pragma unroll
          for (uint32_t i = 0; i < MCR; i++) {
            mi[i] = __builtin_amdgcn_readfirstlane(class_pos(L + 1, i, P.geom.c0));
            lo[i] = hi[i] = 0;
            if (i < KH) {
              if (mi[i] < L)
                lo[i] = cr_lds[i * A + res_reg(mi[i])] & hvy;      /* t[mi] = q[mi],     mi in front of the pair */
              if (mi[i] >= 1)
                hi[i] = cr_lds[i * A + res_reg(mi[i] - 1)] & hvy;  /* t[mi] = q[mi - 1], mi behind the pair */
            }
          }
          /* behind the last class position of the variants nothing moves a class residue: the
             pairs of such a block lie in the staged slice, lane for lane (and in a light tile
             all pairs do) -- no key arithmetic there.  Unless the class positions WRAP (a tile
             of sequences shorter than c0 + K: (c0 + i) mod L and mod L + 1 are different
             positions): then every pair of the variants is keyed on its own. */
          const bool unwrapped = L >= P.geom.c0 + KH;
          uint32_t mi_max = 0;
#pragma unroll
          for (uint32_t i = 0; i < MCR; i++)
            if (i < KH)
              mi_max = mi[i] > mi_max ? mi[i] : mi_max;
          const uint32_t zlast = P.zpos - 1u;
          uint64_t hg = h_ins;                    /* hash of q with a gap at 0 (zobrist_hash_insert_first) */
          uint32_t carry = 31u;                   /* q[g0 - 1]: none in front of the first block */
          auto ins_block = [&](uint32_t g0, auto inner_c) {
            constexpr bool INNER = decltype(inner_c)::value;
            const uint64_t rr = ((uint64_t)s1 << 32) | s0;   /* residues of positions g0 .. g0 + 7 */
            if constexpr (PP == 4) {
              s0 = s1; s1 = s2; s2 = s3; s3 = s4; s4 = s5; s5 = PADW;
            } else {
              s0 = __builtin_amdgcn_alignbit(s2, s1, 16);
              s1 = __builtin_amdgcn_alignbit(s3, s2, 16);
              s2 = __builtin_amdgcn_alignbit(s4, s3, 16);
              s3 = __builtin_amdgcn_alignbit(s5, s4, 16);
              s4 = __builtin_amdgcn_alignbit(PADW, s5, 16);
              s5 = PADW;
            }
            /* which gap positions of the block the lane's query has: bit i <-> g0 + i <= Ll */
            const int left = (int)Ll + 1 - (int)g0;
            const uint32_t lv = valid ? bfm_u32((uint32_t)(left < 0 ? 0 : left > 31 ? 31 : left)) : 0u;
            uint32_t rg[PB], rbefore[PB];
            uint64_t k1[PB], k2[PB], k3[PB], k4[PB];
#pragma unroll
            for (int j = 0; j < PB; j++) {
              const uint32_t g = g0 + 2u * (uint32_t)j;
              rg[j] = (uint32_t)(rr >> (16 * j)) & 31u;                    /* q[g] */
              const uint32_t rn = (uint32_t)(rr >> (16 * j + 8)) & 31u;     /* q[g + 1] */
              rbefore[j] = j == 0 ? carry : (uint32_t)(rr >> (16 * j - 8)) & 31u;
              const uint32_t gs = __builtin_amdgcn_readfirstlane(g);
              uint32_t z0 = zl_addr + ZS * 8u * min(gs, zlast), z1 = zl_addr + ZS * 8u * min(gs + 1u, zlast),
                       z2 = zl_addr + ZS * 8u * min(gs + 2u, zlast);
              asm("" : "+s"(z0));
              asm("" : "+s"(z1));
              asm("" : "+s"(z2));
              k1[j] = lds_u64(z1 + rg[j] * 8u);        /* q[g] one position up: blanked for the pair */
              k2[j] = lds_u64(z0 + rg[j] * 8u);        /* ... and where it comes back to, two gaps on */
              k3[j] = lds_u64(z2 + rn * 8u);
              k4[j] = lds_u64(z1 + rn * 8u);
            }
            carry = (uint32_t)(rr >> (8 * (PP - 1))) & 31u;
            __builtin_amdgcn_sched_barrier(0);
            uint64_t hrow[PB];
            uint32_t wor[PB], slr[PB];
            bool itp[PB];
            bool any_glob = false;
#pragma unroll
            for (int j = 0; j < PB; j++) {
              const uint32_t g = g0 + 2u * (uint32_t)j;
              hrow[j] = hg ^ k1[j];
              hg ^= k1[j] ^ k2[j] ^ k3[j] ^ k4[j];     /* the gap hash of g + 2 */
              wor[j] = woff_of(hrow[j]);
              itp[j] = false;
              slr[j] = tslice;                         /* (the slice as the class keys name it; the staged buffer
                                                          holds one page of it: cslice is where that page lies) */
              if constexpr (!INNER) {
                if (K != 0u && (!unwrapped || g0 <= mi_max)) {          /* wave-uniform: see mi_max */
                  uint32_t key = base_t;
                  bool inside = false;                  /* a class residue of the variants inside the pair */
#pragma unroll
                  for (uint32_t i = 0; i < MCR; i++)
                    if (i < KH) {
                      if ((mi[i] & ~1u) == g)
                        inside = true;
                      else
                        key ^= mi[i] < g ? lo[i] : hi[i];
                    }
                  /* a pair that holds a class position of a split variant class is an item of
                     that position's class part (query_layout.hip) */
                  itp[j] = hvy && inside;
                  slr[j] = row_slice(P.geom, key, -1);
                  any_glob = any_glob || (g <= L && __ballot(valid && !itp[j] && slr[j] != tslice) != 0);
                }
              }
            }
            RowWord wc = word_lds(wor[0]);
            uint32_t xa[PB], xb[PB];
#pragma unroll
            for (int j = 0; j < PB; j++) {
              RowWord wn = wc;
              if (j + 1 < PB)
                wn = word_lds(wor[j + 1]);
              __builtin_amdgcn_sched_barrier(0);
              RowWord w = wc;
              if constexpr (!INNER) {
                if (any_glob) {                                /* rare: a lane's pair lies in another slice */
                  if (valid && !itp[j] && slr[j] != tslice)
                    w = word_glob(filter + (size_t)paged_slice(P.geom, slr[j], hrow[j]) * slice_bytes, wor[j]);
                }
              }
              uint32_t xs, xd, a1, a2;
              pair_bits(w, hrow[j], xs, xd);
              pair_answers(xs, xd, rg[j], rg[j], a1, a2);
              /* in front of g: v != q[g - 1]; in front of g + 1: w != q[g] */
              a1 &= AMASK & ~(1u << rbefore[j]) & (uint32_t)bfe_i32(lv, 2 * j, 1);
              a2 &= AMASK & ~(1u << rg[j]) & (uint32_t)bfe_i32(lv, 2 * j + 1, 1);
              const bool here = INNER || !itp[j];
              /* (pages: a pair of the staged slice counts in the pass of its page; one read where it lies --
                 from the page its hash names -- in the pass of page 0) */
              uint32_t pm = on_page(hrow[j]);
              if constexpr (!INNER)
                pm = slr[j] != tslice ? (page0 ? ~0u : 0u) : pm;
              xa[j] = here ? a1 & pm : 0u;
              xb[j] = here ? a2 & pm : 0u;
              wc = wn;
            }
            treads += valid ? (uint32_t)PB : 0u;
            {
              uint32_t any = 0;
#pragma unroll
              for (int j = 0; j < PB; j++)
                any |= xa[j] | xb[j];
              if (CMPR_DBG(P, DBG_SKIP_EMIT))
                any = 0;
              if (__ballot(any != 0u)) {
#pragma unroll
                for (int j = 0; j < PB; j++)
                  q_push<A, D, GENES, INLINE, PAIRS>(W, zl_addr, (xa[j] | xb[j]) != 0u && any != 0u, hrow[j],
                                              K_INSROWS | ((g0 + 2u * (uint32_t)j) << 3) | (rg[j] << 19),
                                              xa[j], xb[j], (uint32_t)__popc(xa[j]) + (uint32_t)__popc(xb[j]));
              }
            }
          };
          for (uint32_t g0 = 0; g0 <= L; g0 += PP) {
            if (g0 && g0 % (4u * TDW) == 0 && REC_OK && rec_tiles) {
              s0 = cur.r6;
              s1 = cur.r7;
              s2 = s3 = s4 = s5 = PADW;
            } else if (g0 && g0 % (4u * TDW) == 0) {
              const uint32_t w0 = g0 >> 2;
              s0 = w0 < nd ? qr[w0 * WAVE] : PADW;
              s1 = w0 + 1u < nd ? qr[(w0 + 1u) * WAVE] : PADW;
              s2 = w0 + 2u < nd ? qr[(w0 + 2u) * WAVE] : PADW;
              s3 = w0 + 3u < nd ? qr[(w0 + 3u) * WAVE] : PADW;
              s4 = w0 + 4u < nd ? qr[(w0 + 4u) * WAVE] : PADW;
              s5 = w0 + 5u < nd ? qr[(w0 + 5u) * WAVE] : PADW;
            }
            if (K == 0u || (unwrapped && g0 > mi_max))
              ins_block(g0, std::true_type{});
            else
              ins_block(g0, std::false_type{});
          }
          /* the variants these pairs stand for: A in front of position 0, A - 1 in front of
             every other up to the lane's own length -- but for the class pairs of a split
             variant class (items) */
          uint32_t nins = valid ? (uint32_t)A + Ll * (uint32_t)(A - 1) : 0u;
#pragma unroll
          for (uint32_t i = 0; i < MCR; i++)
            if (i < KH && K != 0u) {
              const uint32_t gq = mi[i] & ~1u;
              bool fresh = true;
#pragma unroll
              for (uint32_t k = 0; k < MCR; k++)
                if (k < i && (mi[k] & ~1u) == gq)
                  fresh = false;
              if (fresh && hvy && valid)
                nins -= (gq <= Ll ? (gq == 0u ? (uint32_t)A : (uint32_t)(A - 1)) : 0u) +
                        (gq + 1u <= Ll ? (uint32_t)(A - 1) : 0u);
            }
          nvar += nins;
        }
      }

      W.st.variants += (valid && page0) ? (uint64_t)nvar : 0ull;
      reads += treads;
    }
    if (staged) {
      /* this tile's reads of the slice are done: one step towards handing the
         buffer back to the loader */
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0)
        atomicAdd(&ring[cur_c.b].done, 1u);
    }
    PT_MARK(PT_OTHER);
    /* hand-over: the tile claimed ahead, else a fresh claim */
#ifdef CMPR_PHASE_TIMING
    pt_acc[PT_LOADER_WAIT] += nxt_c.ok ? 1u : 0u;      /* (diagnostic: units whose data was requested ahead) */
#endif
    if (nxt_c.ok) {
      cur_c = nxt_c;
      cur = nxt;
      have = true;
    } else {
      have = next_unit(cur);
    }
    PT_MARK(PT_CLAIM);
  }

  /* leftovers: fewer than 64 entries at a time, until every entry is empty; then the
     block of the positives buffer that was claimed ahead goes back as a block of nulls */
  /* (the rounds this takes = the most variants any entry still holds; their blocks of the
     positives buffer are claimed with ONE atomic now, not one per round, each waiting
     for the answer of the one before: at the end of the kernel every wave is here) */
  {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint32_t bits = 0;
    if ((int)lane < W.qn) {
      const uint32_t ca = W.q.ca[lane], kind = ca & 7u, m = W.q.m[lane];
      bits = (kind == K_ROWS || kind == K_INSROWS) ? (uint32_t)__popc(W.q.cb[lane]) + (uint32_t)__popc(m & 0x0fffffffu)
             : kind == K_PAIR ? (uint32_t)__popc(W.q.cb[lane]) + (uint32_t)__popc(m)
                            : (kind == K_SUB || kind == K_INS) ? (uint32_t)__popc(m) : 1u;
    }
    for (int off = 32; off > 0; off >>= 1)
      bits = max(bits, (uint32_t)__shfl_xor((int)bits, off, WAVE));
    const uint32_t rounds = __builtin_amdgcn_readfirstlane(bits);
    unsigned long long extra = 0;
    const uint32_t xseg = pos_segment_of(P, W.nclaims);
    if (P.pos_buf != nullptr && rounds > 1u && lane == 0)
      extra = atomicAdd(P.pos_ctr + (size_t)xseg * POS_CTR_STRIDE, (unsigned long long)(rounds - 1u) * WAVE);
    for (uint32_t r = 0; r < rounds; r++) {
      if (r >= 1u)                                                  /* (lane 0's is the one read) */
        W.held = (extra + (unsigned long long)(r - 1u) * WAVE) | ((unsigned long long)xseg << 48);
      drain_round<A, D, GENES, INLINE, PAIRS>(W, zl_addr, W.qn < WAVE ? W.qn : WAVE, true);
    }
    if (rounds == 0u) {               /* the block claimed ahead goes back as a block of nulls */
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      flush_or_resolve<GENES, INLINE>(W, 0, 0, true);
    }
  }
  PT_MARK(PT_TAIL);
  PT_FLUSH;

  {
    unsigned long long s[STAT_COUNT] = {W.st.variants, W.st.bloom_pos,
                                        W.st.hash_eq, W.st.matches, reads};
#pragma unroll
    for (int k = 0; k < STAT_COUNT; k++) {
      unsigned long long x = s[k];
      for (int off = 32; off > 0; off >>= 1)
        x += __shfl_down(x, off, WAVE);
      if (lane == 0 && x)
        atomicAdd(stats_dst(P) + k, x);
    }
  }
}

}  // namespace cmpr
#endif
