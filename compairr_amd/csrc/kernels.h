/*
 * kernels.h -- hand-written HIP kernels for gfx950 (MI355X, wave64).
 *
 * Hot path of CompAIRR's --matrix command, rebuilt for the GPU:
 *   index build  : hash every set-2 sequence, insert into hash table + Bloom
 *                  (db_hash db.cc:903-916, hash_insert overlap.cc:63-128)
 *   probe kernel : per query, enumerate every variant within distance d,
 *                  Zobrist-hash it incrementally, test the Bloom filter, and
 *                  for the few survivors walk the hash table, verify the hit
 *                  exactly and add the score to the repertoire matrix
 *                  (process_variants overlap.cc:253-284, generate_variants
 *                  variants.cc:260-428, find_variant_matches overlap.cc:168-251,
 *                  check_variant variants.cc:166-240, compute_score
 *                  overlap.cc:144-166).
 *
 * Mapping: one query per LANE, 64 equal-length queries per wave ("tile").
 * The variant loops (position p, replacement residue v) are then wave-uniform:
 * the key of the replacement residue is a scalar (SGPR) operand, only the key
 * of the query's own residue is a per-lane LDS lookup, and nothing diverges
 * until a Bloom probe comes back positive.  Positives (~1 % of probes) are
 * compacted with a wavefront ballot + prefix count into a per-wave LDS queue
 * and verified 64 at a time, so the rare, long-latency hash-table walks never
 * stall the probe stream.
 */
#ifndef COMPAIRR_AMD_KERNELS_H
#define COMPAIRR_AMD_KERNELS_H

#include <hip/hip_runtime.h>
#include "layout.h"

namespace cmpr {

/* ------------------------------------------------------------------ */
/* small device helpers                                                 */
/* ------------------------------------------------------------------ */

__device__ __forceinline__ uint32_t lane_id()
{
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

/* number of set bits of m below this lane */
__device__ __forceinline__ uint32_t rank_below(uint64_t m)
{
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                   __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__host__ __device__ __forceinline__ uint64_t table_key(uint64_t h)
{
  return h == EMPTY_KEY ? (h ^ 1ull) : h;
}

/* 1u << ((a >> 8 * BYTE) & 31) in ONE instruction: the shifter takes its amount
   from a byte of the register (SDWA operand select) and ignores all but 5 bits */
template <int BYTE>
__device__ __forceinline__ uint32_t one_shl_byte(uint32_t a)
{
  uint32_t d;
  const uint32_t one = 1u;
  /* not convergent: plain per-lane arithmetic (lets loops around it unroll) */
  [[clang::noconvergent]] {
    if (BYTE == 1)
      asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 "
          "src1_sel:DWORD" : "=v"(d) : "v"(a), "v"(one));
    else if (BYTE == 2)
      asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 "
          "src1_sel:DWORD" : "=v"(d) : "v"(a), "v"(one));
    else
      asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 "
          "src1_sel:DWORD" : "=v"(d) : "v"(a), "v"(one));
  }
  return d;
}

/* Bit pattern of a hash, computed, not looked up: the low 5 bits of the four
   bytes of the hash's LOW dword pick two bits of the low dword of the filter
   word (bytes 0, 1) and two of its high dword (bytes 2, 3) -- four shifts whose
   amounts are bytes of the register as it lies (4 instructions).  (The
   reference reads one of 1024 precomputed 8-bit patterns, bloompat.h:45-48; at
   the >= 32 filter bits per key this build uses, 4 bits from a 2^20 pattern
   space give a lower false-positive rate than 8 bits from a 2^10 space, and
   cost two random LDS reads less per probe.) */
__device__ __forceinline__ uint64_t pattern_of(uint64_t h)
{
  const uint32_t a = (uint32_t)h;
  const uint32_t lo = (1u << (a & 31u)) | one_shl_byte<1>(a);
  const uint32_t hi = one_shl_byte<2>(a) | one_shl_byte<3>(a);
  return ((uint64_t)hi << 32) | lo;
}

/* the same pattern as the pieces the probe's test wants (kernels_sliced.h bloom_miss) */
struct BloomPat {
  uint32_t lo;        /* both bits of the word's low dword  */
  uint32_t s2, s3;    /* the two bits of its high dword     */
};
__device__ __forceinline__ BloomPat pattern_fields(uint64_t h)
{
  const uint32_t a = (uint32_t)h;
  BloomPat p;
  p.lo = (1u << (a & 31u)) | one_shl_byte<1>(a);
  p.s2 = one_shl_byte<2>(a);
  p.s3 = one_shl_byte<3>(a);
  return p;
}

/* Byte offset of a hash's filter word before masking to the filter (or slice)
   size: the HIGH dword as it lies (one AND per probe, no shift) */
__device__ __forceinline__ uint32_t bloom_off(uint64_t h)
{
  return (uint32_t)(h >> 32);
}

__host__ __device__ __forceinline__ uint32_t table_home(uint64_t key, uint64_t mask)
{
  /* high half, independent of the Bloom address bits (hashtable.h:36-41); the
     chain starts on a 4-slot boundary, so that the four slots a walk reads per
     round are one aligned 64-byte piece of memory, not two.  (The open-addressing table
     is only built for cmpr_count_duplicates on a set that is not the resident reference.) */
  return (uint32_t)((key >> 32) & mask & ~3ull);
}

/* The record table of set 2 (ref_index.hip): bucket of a key -- its high half, as the
   reference takes the table position from it (hashtable.h:36-41) -- and the 15 bits of
   the key a record carries beside its length (RefRec::len): a record of the right bucket
   with another tag is not looked at further. */
__host__ __device__ __forceinline__ uint32_t dir_bucket(uint64_t key, uint32_t mask)
{
  return (uint32_t)(key >> 32) & mask;
}
__host__ __device__ __forceinline__ uint32_t dir_tag(uint64_t key)
{
  return (uint32_t)key & 0x7fffu;
}
/* What a lookup of bucket `b` does with the record it finds in a slot: the walk starts at
   slot b and goes on slot by slot.  WALK_MINE: a record of this bucket -- look at it (if its
   tag is the key's); the walk goes on iff walk_more(). */
__host__ __device__ __forceinline__ bool walk_ends(uint32_t idx, uint32_t len, uint32_t home, uint32_t b)
{
  /* empty slot; a record of a later bucket (the records lie in bucket order); the last of this bucket's */
  return idx == REC_EMPTY || home > b || (home == b && !(len & REC_MORE));
}

/* ------------------------------------------------------------------ */
/* index build                                                          */
/* ------------------------------------------------------------------ */

struct BuildParams {
  const uint64_t *zob;
  uint32_t        A;
  uint32_t        zpos;
  uint32_t        n_v;
  uint32_t        use_genes;
  const uint8_t  *res;
  const uint64_t *off;
  const uint32_t *v;
  const uint32_t *j;
  uint64_t        n;
  Slot           *table;           /* NULL: no table (the resident reference has its record directory) */
  uint64_t        slot_mask;
  uint64_t       *bloom;           /* NULL: build the table only            */
  uint32_t        bloom_byte_mask;
  uint32_t        sliced;          /* 1: class-keyed slices (layout.h)      */
  uint32_t        indels;          /* row filter: also the gap entries (kernels_rows.h) */
  uint32_t        pairs;           /* row filter: pair rows (d = 1 without -i, kernels_rows.h) */
  uint32_t        pad;
  uint32_t       *count;           /* build_rows_kernel: non-NULL = count the entries per slice instead of filing them */
  SliceGeom       geom;
};

/* One thread per set-2 sequence: Zobrist hash (zobrist.cc:74-88), claim the
   first free slot of the probe chain with a 64-bit CAS (hash_insert,
   overlap.cc:63-128: every entry is inserted, duplicates included), clear the
   pattern bits in the Bloom word (bloom_set, bloompat.h:50-53). */
static __global__ void __launch_bounds__(BLOCK_THREADS)
build_index_kernel(const BuildParams B)
{
  const uint64_t i = (uint64_t)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if (i >= B.n)
    return;
  const uint64_t b = B.off[i];
  const uint32_t L = (uint32_t)(B.off[i + 1] - b);
  uint64_t h = 0;
  if (B.use_genes) {
    const uint64_t *vk = B.zob + (uint64_t)B.A * B.zpos;
    h = vk[B.v[i]] ^ vk[B.n_v + B.j[i]];
  }
  for (uint32_t p = 0; p < L; p++)
    h ^= B.zob[B.A * p + B.res[b + p]];

  if (B.table) {
    const uint64_t key = table_key(h);
    uint64_t slot = table_home(key, B.slot_mask);
    for (;;) {
      unsigned long long prev =
          atomicCAS((unsigned long long *)&B.table[slot].key,
                    (unsigned long long)EMPTY_KEY, (unsigned long long)key);
      if (prev == EMPTY_KEY)
        break;
      slot = (slot + 1) & B.slot_mask;
    }
    B.table[slot].val = (uint32_t)i;
  }

  if (!B.bloom)
    return;                            /* table only: duplicate counting, row filter */
  uint64_t boff = bloom_off(h) & B.bloom_byte_mask;
  if (B.sliced) {
    const uint32_t slice = class_key_of(B.geom.ctab, B.geom, B.A, B.use_genes != 0,
                                        B.res + b, L, B.use_genes ? B.v[i] : 0u,
                                        B.use_genes ? B.j[i] : 0u, nullptr) & B.geom.smask;
    boff = ((uint64_t)slice << (B.geom.words_log2 + 3)) +
           (bloom_off(h) & (((1u << B.geom.words_log2) - 1u) << 3));
  }
  const uint64_t pat = pattern_of(h);
  atomicAnd((unsigned long long *)((char *)B.bloom + boff),
            (unsigned long long)~pat);
}

/* Exact duplicates inside one set: entry i counts when an entry j < i of the
   same repertoire has the same sequence (and V/J unless -g) -- what hash_insert
   reports while indexing (overlap.cc:76-115) and check_duplicates() sums
   (overlap.cc:579-605).  The reference finds j because it inserts in input
   order; here the table is built in parallel, so every same-key entry of the
   probe chain is inspected and "earlier" is decided by the index. */
struct DupParams {
  const uint64_t *zob;
  uint32_t        A, zpos, n_v, use_genes;
  const uint8_t  *res;
  const uint64_t *off;
  const uint32_t *v, *j, *rep;
  uint64_t        n;
  const Slot     *table;           /* a set of its own: open-addressing table of sequence numbers */
  uint64_t        slot_mask;
  const unsigned char *rec;        /* the resident reference: its record table */
  uint32_t        dir_mask;
  unsigned long long *count;
};

static __global__ void __launch_bounds__(BLOCK_THREADS)
count_duplicates_kernel(const DupParams B)
{
  const uint64_t i = (uint64_t)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  bool dup = false;
  if (i < B.n) {
    const uint64_t b = B.off[i];
    const uint32_t L = (uint32_t)(B.off[i + 1] - b);
    uint64_t h = 0;
    if (B.use_genes) {
      const uint64_t *vk = B.zob + (uint64_t)B.A * B.zpos;
      h = vk[B.v[i]] ^ vk[B.n_v + B.j[i]];
    }
    for (uint32_t p = 0; p < L; p++)
      h ^= B.zob[B.A * p + B.res[b + p]];
    const uint64_t key = table_key(h);
    uint64_t slot = table_home(key, B.slot_mask);
    /* the resident reference: the records of the key's bucket, one after the other */
    const uint32_t bk = dir_bucket(key, B.dir_mask);
    uint32_t piece = bk;
    bool ended = false;
    for (;;) {
      uint64_t k, o;
      if (B.rec) {
        if (ended)
          break;
        const RefRec *r = (const RefRec *)B.rec + piece;
        const uint32_t ri = r->idx, rl = r->len, rh = r->home;
        ended = walk_ends(ri, rl, rh, bk);
        if (ri == REC_EMPTY)
          break;
        k = (rh == bk && (rl >> REC_TAG_SHIFT) == dir_tag(key)) ? key : ~key;
        o = ri;
        piece++;
      } else {
        const Slot sl = B.table[slot];
        k = sl.key;
        o = sl.val;
        if (k == EMPTY_KEY)
          break;
        slot = (slot + 1) & B.slot_mask;
      }
      if (k == key) {
        if (o < i && B.rep[o] == B.rep[i] &&
            (!B.use_genes || (B.v[o] == B.v[i] && B.j[o] == B.j[i]))) {
          const uint64_t ob = B.off[o];
          if ((uint32_t)(B.off[o + 1] - ob) == L) {
            bool same = true;
            for (uint32_t p = 0; p < L && same; p++)
              same = B.res[ob + p] == B.res[b + p];
            if (same) {
              dup = true;
              break;
            }
          }
        }
      }
    }
  }
  const uint64_t m = __ballot(dup);
  if (m && lane_id() == 0)
    atomicAdd(B.count, (unsigned long long)__popcll(m));
}

/* SoA -> header + residues per set-2 sequence (layout.h RefRec) */
struct PackParams {
  const uint64_t *off, *cnt;
  const uint32_t *v, *j, *rep, *voff;
  const uint32_t *tag;             /* dir_tag of the sequence's key */
  const uint32_t *home;            /* its bucket */
  const uint32_t *more;            /* != 0: the next slot holds another record of the bucket */
  uint32_t       *bmap;            /* one bit per bucket: set for the buckets that hold a record */
  uint32_t        packed;          /* 1: nucleotides, two bits per residue (layout.h REC_RES_NT) */
  const uint8_t  *res;
  uint64_t        n;
  unsigned char  *out;
};

static __global__ void __launch_bounds__(BLOCK_THREADS)
pack_records_kernel(const PackParams B)
{
  const uint64_t i = (uint64_t)blockIdx.x * BLOCK_THREADS + threadIdx.x;
  if (i >= B.n)
    return;
  const uint64_t b = B.off[i];
  RefRec r;
  r.idx = (uint32_t)i;
  const uint32_t L = (uint32_t)(B.off[i + 1] - b);
  r.len = L | (B.more[i] ? REC_MORE : 0u) | (B.tag[i] << REC_TAG_SHIFT);
  r.cnt = B.cnt ? B.cnt[i] : 1ull;
  r.v = B.v ? B.v[i] : 0u;
  r.j = B.j ? B.j[i] : 0u;
  r.rep = B.rep[i];
  r.home = B.home[i];
  if (B.packed) {
    for (uint32_t p = 0; p < REC_RES; p++)
      r.res[p] = 0;
    for (uint32_t p = 0; p < L && p < REC_RES_NT; p++)
      r.res[p >> 2] |= (unsigned char)((B.res[b + p] & 3u) << ((p & 3u) * 2u));
  } else {
    for (uint32_t p = 0; p < REC_RES; p++)
      r.res[p] = p < L ? B.res[b + p] : (unsigned char)0xff;
  }
  ((RefRec *)B.out)[B.voff[i]] = r;
  if (B.bmap && !B.more[i])                        /* (once per bucket: by its last record) */
    atomicOr(B.bmap + (r.home >> 5), 1u << (r.home & 31u));
}

/* ------------------------------------------------------------------ */
/* probe kernel                                                         */
/* ------------------------------------------------------------------ */

constexpr int QCAP = 2 * WAVE;      /* per-wave queue of Bloom positives   */

struct WaveQueue {
  uint64_t hash[QCAP];
  uint32_t slot[QCAP];   /* query: tile * 64 + lane                          */
  uint32_t ca[QCAP];     /* kind | p1 << 3 | r1 << 24                        */
  uint32_t cb[QCAP];     /* p2 | r2 << 24                                    */
  uint32_t m[QCAP];      /* variant 2 (kernels_rows.h q_push): the entry's mask of positive variants */
};

__device__ __forceinline__ uint32_t pack_a(uint32_t kind, uint32_t p1, uint32_t r1)
{
  return kind | (p1 << 3) | (r1 << 24);
}

/* residue p of query `lane` of tile `td` */
__device__ __forceinline__ uint32_t query_residue(const ProbeParams &P,
                                                  const TileDesc &td,
                                                  uint32_t lane, uint32_t p)
{
  uint32_t w = P.qres[td.res_base + (uint64_t)(p >> 2) * WAVE + lane];
  return (w >> ((p & 3) * 8)) & 0xffu;
}

/* Exact test that the query with the variant applied IS the hit sequence --
   restates check_variant (variants.cc:166-240) over the tile layout.  Written
   without early exits: position x of the hit is compared with the residue the
   variant must have there, all loads of the loop are independent and stay in
   flight together (an early-exit loop costs one memory round trip per residue). */
__device__ bool variant_matches_hit(const ProbeParams &P, const TileDesc &td,
                                    uint32_t lane, uint32_t L, uint32_t ca, uint32_t cb,
                                    const RefRec &rec, const unsigned char *t, const QueryRec *qrec = nullptr)
{
  /* (qrec: the query's residues from its record -- ProbeParams::rec_tiles, no position-major copy exists) */
  const uint32_t kind = ca & 7u, p1 = (ca >> 3) & 0xffffu, r1 = ca >> 24;
  const uint32_t p2 = cb & 0xffffu, r2 = cb >> 24;
  const uint32_t M = rec.len;
  /* expected length of the hit */
  const uint32_t want = kind == K_DEL ? L - 1 : (kind == K_INS ? L + 1 : L);
  if (M != want)
    return false;
  uint32_t bad = 0;
#pragma unroll 4
  for (uint32_t x = 0; x < M; x++) {
    /* query position that lands on x, or "the new residue" */
    uint32_t qp = x;
    bool is_new = false;
    uint32_t newres = r1;
    if (kind == K_DEL) {
      qp = x < p1 ? x : x + 1;
    } else if (kind == K_INS) {
      is_new = x == p1;
      qp = x < p1 ? x : x - 1;
    } else if (kind == K_SUB) {
      is_new = x == p1;
    } else if (kind == K_SUB2) {
      is_new = x == p1 || x == p2;
      newres = x == p1 ? r1 : r2;
    }
    if (qp >= L)
      qp = L ? L - 1 : 0;                       /* only when is_new; keeps the load in range */
    const uint32_t expect = is_new ? newres
                            : qrec   ? (qrec->res[qp >> 2] >> ((qp & 3u) * 8u)) & 0xffu
                                     : query_residue(P, td, lane, qp);
    bad |= expect ^ (uint32_t)t[x];
  }
  return bad == 0;
}

struct LaneStats {
  uint64_t variants;
  uint32_t bloom_pos, hash_eq, matches;
};

/* One verified (query, hit) pair: list it (pairs mode) or add its score to the
   matrix cell (overlap.cc:218-245). */
__device__ __forceinline__ void score_match(const ProbeParams &P, uint32_t qs, uint32_t hit,
                                            uint64_t cell, unsigned long long f,
                                            unsigned long long g, unsigned long long *mat_lds)
{
  if (P.pair_count) {
    /* pairs mode (overlap.cc:232-245) */
    const unsigned long long k = atomicAdd(P.pair_count, 1ull);
    if (k < P.pair_cap) {
      P.pair_q[k] = P.qrec[qs].orig;
      P.pair_h[k] = hit;
    }
  } else if (P.score == 1 /* ratio */ && !P.ignore_counts) {
    unsafeAtomicAdd(P.matrix_f64 + cell, (double)f / (double)g);
  } else {
    unsigned long long sc = 1;
    if (!P.ignore_counts) {
      switch (P.score) {
      case 2: case 6: sc = f < g ? f : g; break;       /* min, Jaccard */
      case 3:         sc = f > g ? f : g; break;       /* max          */
      case 4:         sc = f + g;         break;       /* 2 x mean     */
      default:        sc = f * g;         break;       /* product, MH  */
      }
    }
    if (mat_lds)
      atomicAdd(mat_lds + cell, sc);
    else
      atomicAdd(P.matrix + cell, sc);
  }
}

/* Lane e of the wave resolves queue entry `e`: the records of the variant hash's bucket
   (find_variant_matches, overlap.cc:168-251, over the record directory), verify, score,
   accumulate. */
template <bool GENES>
__device__ void resolve_one(const ProbeParams &P, uint64_t hash, uint32_t qs, uint32_t ca,
                            uint32_t cb, unsigned long long *mat_lds, LaneStats &st)
{
  const uint64_t key = table_key(hash);
  const uint32_t bk = dir_bucket(key, P.dir_mask);
  const TileDesc td = P.tiles[qs >> 6];
  const uint32_t ql = qs & 63u;
  /* (from the query's record: the per-slot arrays of genes, repertoire and count are laid out only for the
     kernels whose tiles read them -- variants 0 and 1) */
  const QueryRec *const qrec = P.qrec + qs;
  const uint32_t q_v = GENES ? qrec->v : 0u, q_j = GENES ? qrec->j : 0u;
  const uint32_t q_rep = qrec->rep;
  const uint32_t q_len = P.rec_tiles ? qrec->len : (uint32_t)P.qlen[qs];      /* own length (tiles may mix lengths) */
  const unsigned long long q_cnt = P.ignore_counts ? 1ull : qrec->cnt;
  for (uint32_t piece = bk;; piece++) {
    RefRec rec = *((const RefRec *)P.rec2 + piece);
    if (rec.idx == REC_EMPTY)
      break;
    const bool last = walk_ends(rec.idx, rec.len, rec.home, bk);
    if (rec.home == bk && (rec.len >> REC_TAG_SHIFT) == dir_tag(key)) {
      rec.len &= 0xffffu;
      st.hash_eq++;
      bool ok = true;
      if (GENES)
        ok = (q_v == rec.v) && (q_j == rec.j);
      /* (the residues where the set lies: this is the rare path, and they are all there) */
      if (ok && variant_matches_hit(P, td, ql, q_len, ca, cb, rec, P.res2 + P.off2[rec.idx],
                                    P.rec_tiles ? qrec : nullptr)) {
        st.matches++;
        score_match(P, qs, rec.idx, (uint64_t)P.R2 * q_rep + rec.rep, q_cnt, rec.cnt, mat_lds);
      }
    }
    if (last)
      break;
  }
}

template <bool GENES>
__device__ __forceinline__ void resolve_entry(const ProbeParams &P, const WaveQueue &q, int e,
                                              unsigned long long *mat_lds, LaneStats &st)
{
  resolve_one<GENES>(P, q.hash[e], q.slot[e], q.ca[e], q.cb[e], mat_lds, st);
}

/* ---- deferred mode: resolve_kernel -------------------------------------- */

/* (query, variant, set-2 sequence) triples whose table key equals the variant
   hash, waiting for verification -- one queue per wave, in LDS */
struct CandQueue {
  uint32_t slot[QCAP], ca[QCAP], cb[QCAP];
  uint32_t tagb[QCAP];                 /* the key's bucket | its tag << 17 ... (see verify_candidate) */
  uint32_t piece[QCAP];                /* the table slot to look at */
  uint32_t bucket[QCAP];
};

/* n low bytes set (n <= 0: none, n >= 4: all) */
__device__ __forceinline__ uint32_t low_bytes(int n)
{
  return n <= 0 ? 0u : (n >= 4 ? 0xffffffffu : ((1u << (8 * n)) - 1u));
}

/* check_variant (variants.cc:166-240) on packed words: positions [16c, 16c+16)
   of the hit (`r`, four residues per dword) against what the variant of the
   query must have there.  q[0..5] are the query's dwords 4c-1 .. 4c+4 (the
   neighbours feed the one-residue shift of a deletion / insertion).  Returns
   non-zero if any position below M differs. */
__device__ __forceinline__ uint32_t block_mismatch(uint32_t c, const uint32_t q[6],
                                                  const uint32_t r[4], uint32_t kind,
                                                  uint32_t p1, uint32_t r1, uint32_t p2,
                                                  uint32_t r2, uint32_t M)
{
  uint32_t bad = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int pos0 = (int)(16 * c) + 4 * j;
    const uint32_t prv = q[j], cur = q[j + 1], nxt = q[j + 2];
    uint32_t e = cur;
    if (kind == K_DEL || kind == K_INS) {
      /* positions below p1 come straight from the query, the others from the
         query shifted by one residue */
      const uint32_t sh = kind == K_DEL ? (cur >> 8) | (nxt << 24) : (cur << 8) | (prv >> 24);
      const uint32_t m = low_bytes((int)p1 - pos0);
      e = (cur & m) | (sh & ~m);
    }
    if (kind == K_SUB || kind == K_INS || kind == K_SUB2) {
      const uint32_t d = p1 - (uint32_t)pos0;
      if (d < 4u)
        e = (e & ~(0xffu << (8 * d))) | (r1 << (8 * d));
    }
    if (kind == K_SUB2) {
      const uint32_t d = p2 - (uint32_t)pos0;
      if (d < 4u)
        e = (e & ~(0xffu << (8 * d))) | (r2 << (8 * d));
    }
    bad |= (e ^ r[j]) & low_bytes((int)M - pos0);
  }
  return bad;
}

/* Verification + scoring of one candidate per lane: the table slot `piece` (a record with its
   first 32 residues) and, when that is a record of the candidate's key, the query's record with
   its first 36 -- one 64-byte line each; sequences longer than 32 take their further residues
   from where set 2 lies.  Returns true when the walk of bucket `bk` goes on in the next slot. */
template <bool GENES, bool EAGER = false>
__device__ __forceinline__ bool verify_candidate(const ProbeParams &P, uint32_t qs, uint32_t ca,
                                                 uint32_t cb, uint32_t tag, uint32_t bk, uint32_t piece,
                                                 unsigned long long *mat_lds, LaneStats &st)
{
  const uint4 *rp = (const uint4 *)P.rec2 + (size_t)piece * 4;
  const uint4 h0 = rp[0], h1 = rp[1], t0 = rp[2], t1 = rp[3];
  /* the query's record (layout.h QueryRec): count, genes, repertoire, length and
     its first 36 residues in ONE 64-byte piece, like the hit's */
  const uint4 *qp = (const uint4 *)P.qrec + (size_t)qs * 4;
  /* EAGER (the first look of the filterless d = 0 kernel: lane = query, the 64 records one run of memory):
     requested together with the slot, one round trip instead of two */
  uint4 a0, a1, a2, a3;
  if (EAGER) {
    a0 = qp[0]; a1 = qp[1]; a2 = qp[2]; a3 = qp[3];
  }
  if (CMPR_DBG(P, DBG_RES_NO_VERIFY)) {
    st.hash_eq += (h0.x ^ h1.x ^ t0.x ^ t1.x ^ qp[0].x ^ qp[1].x ^ qp[2].x ^ qp[3].x) == 0x12345u ? 1u : 0u;
    return false;
  }
  RefRec rec;
  rec.cnt = ((unsigned long long)h0.y << 32) | h0.x;
  rec.idx = h0.z; rec.len = h0.w; rec.v = h1.x; rec.j = h1.y; rec.rep = h1.z; rec.home = h1.w;
  const bool more = !walk_ends(rec.idx, rec.len, rec.home, bk);
  if (rec.idx == REC_EMPTY || rec.home != bk || (rec.len >> REC_TAG_SHIFT) != tag)
    return more;                                 /* nothing here / another bucket's / another key's */
  st.hash_eq++;
  /* (the query's record is asked for only now: a false positive of the filter -- on skewed data up to
     half of the positives -- costs one memory line, not two) */
  if (!EAGER) {
    a0 = qp[0]; a1 = qp[1]; a2 = qp[2]; a3 = qp[3];
  }
  uint32_t q[10];
  q[0] = 0;
  q[1] = a1.z; q[2] = a1.w; q[3] = a2.x; q[4] = a2.y; q[5] = a2.z; q[6] = a2.w;
  q[7] = a3.x; q[8] = a3.y; q[9] = a3.z;
  const uint32_t q_v = GENES ? a0.z : 0u, q_j = GENES ? a0.w : 0u;
  const uint32_t q_rep = a1.x;
  const uint32_t L = a1.y;
  const unsigned long long q_cnt = ((unsigned long long)a0.y << 32) | a0.x;
  const uint32_t kind = ca & 7u, p1 = (ca >> 3) & 0xffffu, r1 = ca >> 24;
  const uint32_t p2 = cb & 0xffffu, r2 = cb >> 24;
  const uint32_t M = rec.len & 0xffffu;
  const uint32_t want = kind == K_DEL ? L - 1 : (kind == K_INS ? L + 1 : L);
  bool ok = M == want;
  if (GENES)
    ok = ok && (q_v == rec.v) && (q_j == rec.j);
  /* the record's residues: a byte each, or (nucleotides) two bits each -- 16 positions per dword, spread to
     the four byte-dwords block_mismatch compares */
  const uint32_t pk[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
  auto spread = [](uint32_t x, uint32_t out[4]) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t b8 = (x >> (8 * k)) & 0xffu;
      out[k] = (b8 & 3u) | ((b8 & 0xcu) << 6) | ((b8 & 0x30u) << 12) | ((b8 & 0xc0u) << 18);
    }
  };
  uint32_t r0[4] = {t0.x, t0.y, t0.z, t0.w}, r1w[4] = {t1.x, t1.y, t1.z, t1.w};
  const bool packed = P.rec_packed != 0u;
  if (packed) {
    spread(pk[0], r0);
    spread(pk[1], r1w);
  }
  uint32_t bad = block_mismatch(0, q, r0, kind, p1, r1, p2, r2, M) |
                 block_mismatch(1, q + 4, r1w, kind, p1, r1, p2, r2, M);
  if (ok && bad == 0 && M > REC_RES) {
    /* residues past the 32nd: nucleotides up to the 128th from the record, everything else from set 2 as
       it lies; the query's from its tile (rare for amino acids -- a CDR3 is shorter) */
    const uint8_t *hr = P.res2 + ((!packed || M > REC_RES_NT) ? P.off2[rec.idx] : 0ull);
    const uint32_t *qr = P.qres + P.tiles[qs >> 6].res_base + (qs & 63u);
    for (uint32_t c = 2; 16 * c < M; c++) {
      uint32_t rr[4] = {0, 0, 0, 0};
      if (packed && c < 8u) {
        uint32_t x = pk[0];
#pragma unroll
        for (uint32_t k = 1; k < 8; k++)
          x = c == k ? pk[k] : x;
        spread(x, rr);
      } else
      for (uint32_t x = 0; x < 16u && 16u * c + x < M; x++)
        rr[x >> 2] |= (uint32_t)hr[16u * c + x] << (8u * (x & 3u));
      uint32_t qq[6];
#pragma unroll
      for (int k = 0; k < 6; k++)
        qq[k] = qr[(size_t)(4 * c - 1 + k) * WAVE];
      bad |= block_mismatch(c, qq, rr, kind, p1, r1, p2, r2, M);
    }
  }
  if (ok && bad == 0) {
    st.matches++;
    score_match(P, qs, rec.idx, (uint64_t)P.R2 * q_rep + rec.rep, q_cnt, rec.cnt, mat_lds);
  }
  return more;
}

/* One round of the candidate queue: lanes 0 .. n-1 take the n entries on top (n = 64 but
   for the last rounds), look at one table slot each, and queue up again for the next slot
   while their walk goes on.  Returns the new fill (wave-uniform). */
template <bool GENES>
__device__ __forceinline__ int verify_round(const ProbeParams &P, CandQueue &cq, int qn, int n, uint32_t lane,
                                            unsigned long long *mat_lds, LaneStats &st)
{
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const int base = qn - n;
  const bool mine = (int)lane < n;
  uint32_t qs = 0, ca = 0, cb = 0, tag = 0, bk = 0, piece = 0;
  bool more = false;
  if (mine) {
    const int x = base + (int)lane;
    qs = cq.slot[x]; ca = cq.ca[x]; cb = cq.cb[x]; tag = cq.tagb[x]; bk = cq.bucket[x]; piece = cq.piece[x];
    more = verify_candidate<GENES>(P, qs, ca, cb, tag, bk, piece, mat_lds, st);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const uint64_t mm = __ballot(more);
  if (more) {
    const int x = base + (int)rank_below(mm);
    cq.slot[x] = qs; cq.ca[x] = ca; cq.cb[x] = cb; cq.tagb[x] = tag; cq.bucket[x] = bk; cq.piece[x] = piece + 1u;
  }
  return base + __popcll(mm);
}

/* destinations of a workgroup's results (ProbeParams::part) */
__device__ __forceinline__ unsigned long long *stats_dst(const ProbeParams &P)
{
  return P.part ? P.part + (size_t)(blockIdx.x % NPART) * P.part_stride + (P.part_stride - STAT_COUNT)
                : P.stats;
}
__device__ __forceinline__ unsigned long long *matrix_dst(const ProbeParams &P)
{
  return P.part ? P.part + (size_t)(blockIdx.x % NPART) * P.part_stride : P.matrix;
}

/* Sums the NPART partial results into the matrix and the counters and clears them
   for the next launch: one workgroup per cell / counter, one thread per slot (all
   NPART loads in flight at once -- a thread walking the slots one after the other
   took 36 us, a quarter of a step at 8 GPUs).  `overwrite`: the cells are written,
   not added to (the matrix was not cleared before the launch).  Also clears the
   counter block of the NEXT launch (`next_ctr`, n64 words).  `sticky` (launches that
   run without a redo pass): set, and left set until the host has seen it, when the
   positives of this launch did not fit their buffer -- its result is then invalid. */
static __global__ void __launch_bounds__(NPART)
reduce_partials_kernel(const ProbeParams P, uint32_t cells, uint32_t overwrite,
                       unsigned long long *next_ctr, uint32_t n64, unsigned long long *usage,
                       unsigned long long *sticky)
{
  static_assert(NPART == 2 * WAVE, "two waves per workgroup");
  __shared__ unsigned long long half[2];
  const uint32_t i = blockIdx.x, s = threadIdx.x;
  if (sticky && P.overflow && i == 0 && s == 0 && *(volatile unsigned long long *)P.overflow != 0ull)
    *sticky = 1ull;
  /* how full the fullest segment of the positives buffer got (the host's margin check) */
  if (usage && P.pos_ctr && i == 0)
    for (uint32_t g = s; g < P.pos_segments; g += NPART)
      atomicMax(usage, P.pos_ctr[(size_t)g * POS_CTR_STRIDE]);
  for (uint32_t k = i * NPART + s; k < n64; k += gridDim.x * NPART)
    next_ctr[k] = 0;
  unsigned long long x = P.part[(size_t)s * P.part_stride + i];
  if (x)
    P.part[(size_t)s * P.part_stride + i] = 0;
  for (int off = 32; off > 0; off >>= 1)
    x += __shfl_down(x, off, WAVE);
  if ((s & (WAVE - 1)) == 0)
    half[s / WAVE] = x;
  __syncthreads();
  if (s != 0)
    return;
  const unsigned long long sum = half[0] + half[1];
  if (i < cells) {
    if (overwrite)
      P.matrix[i] = sum;
    else if (sum)
      P.matrix[i] += sum;
  } else if (sum && i >= P.part_stride - STAT_COUNT) {
    P.stats[i - (P.part_stride - STAT_COUNT)] += sum;
  }
}

/* Second kernel of the deferred mode.  One lane per queued Bloom positive: the slot its
   variant hash's bucket names in the record table (find_variant_matches, overlap.cc:168-251)
   and the query's record, requested together; verified and scored where the slot holds a
   record of that key; a walk that goes on (displaced records in front, more records of the
   bucket) queues up again, so that every further round trip is made by 64 busy lanes too.
   One random memory line per positive for the hit, one (near its tile's others) for the query. */
template <bool GENES>
__global__ void __launch_bounds__(BLOCK_THREADS, 4)      /* <= 128 VGPRs: 4 waves per SIMD */
resolve_kernel(const ProbeParams P)
{
  extern __shared__ __align__(16) unsigned char smem[];
  CandQueue &cq = ((CandQueue *)smem)[threadIdx.x / WAVE];
  unsigned long long *mat_lds =
      (unsigned long long *)(smem + (BLOCK_THREADS / WAVE) * sizeof(CandQueue));
  const uint32_t cells = P.lds_matrix ? P.R1 * P.R2 : 0u;   /* LDS copy only: <= 2048 cells */
  if (P.lds_matrix) {
    for (uint32_t i = threadIdx.x; i < cells; i += BLOCK_THREADS)
      mat_lds[i] = 0;
    __syncthreads();
  } else {
    /* (no LDS copy: the workgroup's partial slot in HBM when the matrix is kept there, else the matrix) */
    mat_lds = P.part && P.part_cells ? matrix_dst(P) : nullptr;
  }
  /* workgroup b of the grid (a multiple of pos_segments) works on segment
     b % pos_segments, together with the gridDim.x / pos_segments - 1 others */
  /* a probe launch whose positives did not all fit is redone with inline resolve
     (kernels_rows.h): what did fit is not to be counted twice */
  const bool skip = P.overflow && *(volatile unsigned long long *)P.overflow != 0ull;
  const uint32_t seg = blockIdx.x & (P.pos_segments - 1);
  const uint32_t seg_block = blockIdx.x / P.pos_segments;
  const uint32_t seg_blocks = gridDim.x / P.pos_segments;
  const unsigned long long *ctr = P.pos_ctr + (size_t)seg * POS_CTR_STRIDE;
  const PosEntry *pos = P.pos_buf + (size_t)seg * (P.pos_cap + WAVE);
  unsigned long long n = ctr[0];
  const unsigned long long lim = ~ctr[1];
  if (lim < n)
    n = lim;                                  /* claims past the capacity were not written */
  if (skip)
    n = 0;
  const uint32_t lane = lane_id();
  LaneStats st{0ull, 0u, 0u, 0u};
  int qn = 0;
  for (unsigned long long base = (unsigned long long)seg_block * BLOCK_THREADS +
                                 (threadIdx.x & ~(WAVE - 1));
       base < n; base += (unsigned long long)seg_blocks * BLOCK_THREADS) {
    bool active = base + lane < n;
    PosEntry e{};
    if (active)
      e = pos[base + lane];
    active = active && e.slot != POS_NULL_SLOT;      /* padding of a block of 64 */
    const uint64_t key = table_key(e.hash);
    uint32_t bk = dir_bucket(key, P.dir_mask);
    if (CMPR_DBG(P, DBG_RES_SEQ_REC))
      bk = (uint32_t)(base + lane) & P.dir_mask;
    if (CMPR_DBG(P, DBG_RES_SEQ_QREC))
      e.slot = (uint32_t)((base + lane) % 8000000ull);
    /* a positive whose bucket holds a record is a candidate: the bucket's slot is where the walk starts.
       (The filter's false positives name an empty bucket two times out of three; with single rows at d = 2
       six of seven positives are false.) */
    if (P.bmap && active)
      active = ((P.bmap[bk >> 5] >> (bk & 31u)) & 1u) != 0u;
    const uint64_t mm = __ballot(active);
    if (mm) {
      if (active) {
        const int x = qn + (int)rank_below(mm);
        cq.slot[x] = e.slot;
        cq.ca[x] = e.ca;
        cq.cb[x] = e.cb;
        cq.tagb[x] = dir_tag(key);
        cq.bucket[x] = bk;
        cq.piece[x] = bk;
      }
      qn += __popcll(mm);
      while (qn >= WAVE)
        qn = verify_round<GENES>(P, cq, qn, WAVE, lane, mat_lds, st);
    }
  }
  while (qn > 0)
    qn = verify_round<GENES>(P, cq, qn, qn < WAVE ? qn : WAVE, lane, mat_lds, st);
  {
    unsigned long long sum[2] = {st.hash_eq, st.matches};
#pragma unroll
    for (int k = 0; k < 2; k++) {
      unsigned long long x = sum[k];
      for (int off = 32; off > 0; off >>= 1)
        x += __shfl_down(x, off, WAVE);
      if (lane == 0 && x)
        atomicAdd(stats_dst(P) + (k == 0 ? STAT_HASH_EQ : STAT_MATCHES), x);
    }
  }
  if (P.lds_matrix) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cells; i += BLOCK_THREADS) {
      const unsigned long long x = mat_lds[i];
      if (x)
        atomicAdd(matrix_dst(P) + i, x);
    }
  }
}

/* per-wave probing state */
struct Prober {
  const ProbeParams  &P;
  WaveQueue          &q;
  unsigned long long *mat_lds;
  uint32_t            lane;
  uint32_t            qslot;     /* tile * 64 + lane                         */
  int                 qn;        /* queue fill, wave-uniform, < 64 on entry  */
  LaneStats           st;
};

template <bool GENES>
__device__ __forceinline__ void drain_full(Prober &W)
{
  /* entries [qn-64, qn) -- all 64 lanes busy */
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  W.qn -= WAVE;
  resolve_entry<GENES>(W.P, W.q, W.qn + (int)W.lane, W.mat_lds, W.st);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

/* One Bloom probe per lane (bloom_get, bloompat.h:55-58) + queue push of the
   positives.  `live` masks lanes whose variant does not exist (padding lanes,
   replacement == original residue, repeated deletion). */
template <bool GENES>
__device__ __forceinline__ void probe(Prober &W, uint64_t hv, bool live,
                                      uint32_t ca, uint32_t cb)
{
  const uint32_t boff = bloom_off(hv) & W.P.bloom_byte_mask;
  const uint64_t word = *(const uint64_t *)((const char *)W.P.bloom + boff);
  const uint64_t pat = pattern_of(hv);
  const bool pos = live && ((word & pat) == 0);
  W.st.variants += live ? 1ull : 0ull;
  const uint64_t m = __ballot(pos);
  if (m) {
    if (pos) {
      const int e = W.qn + (int)rank_below(m);
      W.q.hash[e] = hv;
      W.q.slot[e] = W.qslot;
      W.q.ca[e] = ca;
      W.q.cb[e] = cb;
      W.st.bloom_pos++;
    }
    W.qn += __popcll(m);
    if (W.qn >= WAVE)
      drain_full<GENES>(W);
  }
}

/*
 * A: alphabet (20 aa / 4 nt); D: differences 0..2; INDELS: -i (D == 1);
 * GENES: V/J hashed and compared (no -g).
 *
 * LDS: [A * zpos Zobrist position keys][R1 * R2 matrix (optional)]
 *      [one WaveQueue per wave]
 *
 * The replacement-residue loops are deliberately NOT unrolled: each probe()
 * site carries an inlined queue drain, and memory-level parallelism comes
 * from the 16-32 resident waves per CU (one 64-lane gather in flight each),
 * which already exceeds what the memory system can retire.
 */
template <int A, int D, bool INDELS, bool GENES>
__global__ void __launch_bounds__(BLOCK_THREADS)
probe_kernel(const ProbeParams P)
{
  extern __shared__ __align__(16) unsigned char smem[];
  uint64_t *zl = (uint64_t *)smem;
  const uint32_t nz = (uint32_t)A * P.zpos;
  unsigned long long *mat_all = (unsigned long long *)(zl + nz);
  const uint32_t cells = P.lds_matrix ? P.R1 * P.R2 : 0u;   /* LDS copy only: <= 2048 cells */
  WaveQueue *queues = (WaveQueue *)(mat_all + (P.lds_matrix ? cells : 0));

  if (D != 0)                            /* (d = 0: the layout has hashed the queries) */
    for (uint32_t i = threadIdx.x; i < nz; i += BLOCK_THREADS)
      zl[i] = P.zob[i];
  if (P.lds_matrix)
    for (uint32_t i = threadIdx.x; i < cells; i += BLOCK_THREADS)
      mat_all[i] = 0;
  __syncthreads();

  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x / WAVE;
  /* (no LDS copy of the matrix: the workgroup's partial slot in HBM when the matrix is kept there -- a
     self-comparison sends every identity pair to a diagonal cell --, else the matrix itself) */
  Prober W{P, queues[wave], P.lds_matrix ? mat_all : (P.part && P.part_cells ? matrix_dst(P) : nullptr),
           lane, 0u, 0, {0ull, 0u, 0u, 0u}};
  const uint64_t *zs = P.zob;            /* wave-uniform lookups: scalar loads */
  const uint64_t *gene_keys = P.zob + nz;

  /* d = 0: every tile is the same work, dealt out statically (one counter for 4 x 10^5 tiles
     would serialise the launch); the wave's queue holds the walks that go on (CandQueue) */
  static_assert(sizeof(CandQueue) <= sizeof(WaveQueue), "the d = 0 kernel keeps a CandQueue where the others keep a WaveQueue");
  CandQueue &cq = *(CandQueue *)&queues[wave];
  int cqn = 0;
  if (D == 0) {
    /* ---- d = 0: the unchanged sequence is the only variant (generate_variants_0,
            variants.cc:260-268), and it is looked up where its bucket lies, no filter word in
            front: one test per query, and the word would cost the memory line the table's slot
            costs.  Lane = query: the layout has left its hash in qgh, its record lies next to its
            neighbours' (requested together with the slot), the slot is the one random line
            (find_variant_matches, overlap.cc:168-251).  The bucket's own slot is looked at here,
            all lanes busy; a walk that goes on (a displaced record of an earlier bucket in front,
            more records of this bucket behind) queues up with the next slot like resolve_kernel's,
            so that every further round trip is made by 64 busy lanes too (a wave that walked its 64
            buckets to the end of the longest waited 6-10 round trips per tile).  The next tile's
            hashes are on their way while this tile's slots are. ---- */
    const uint32_t step = gridDim.x * WAVES_PER_BLOCK;
    uint32_t t = (blockIdx.x * WAVES_PER_BLOCK + wave) * P.work_step + P.work_first;
    uint64_t hn = 0;
    uint32_t nvn = 0;
    if (t < P.ntiles) {
      hn = P.qgh[(size_t)(t + P.first_tile) * WAVE + lane];
      nvn = P.tiles[t + P.first_tile].nvalid;
    }
    const uint32_t ca = pack_a(K_SAME, 0, 0);
    while (t < P.ntiles) {
      const uint64_t h = hn;
      const bool valid = lane < nvn;
      const uint32_t qslot = (t + P.first_tile) * WAVE + lane;
      t += step * P.work_step;
      if (t < P.ntiles) {
        hn = P.qgh[(size_t)(t + P.first_tile) * WAVE + lane];
        nvn = P.tiles[t + P.first_tile].nvalid;
      }
      const uint64_t key = table_key(h);
      const uint32_t bk = dir_bucket(key, P.dir_mask), tag = dir_tag(key);
      W.st.variants += valid ? 1ull : 0ull;
      W.st.bloom_pos += valid ? 1u : 0u;
      bool more = false;
      if (valid)
        more = verify_candidate<GENES, true>(P, qslot, ca, 0u, tag, bk, bk, W.mat_lds, W.st);
      const uint64_t mm = __ballot(more);
      if (mm) {
        if (more) {
          const int x = cqn + (int)rank_below(mm);
          cq.slot[x] = qslot; cq.ca[x] = ca; cq.cb[x] = 0u; cq.tagb[x] = tag; cq.bucket[x] = bk;
          cq.piece[x] = bk + 1u;
        }
        cqn += __popcll(mm);
        while (cqn >= WAVE)
          cqn = verify_round<GENES>(P, cq, cqn, WAVE, lane, W.mat_lds, W.st);
      }
    }
    while (cqn > 0)
      cqn = verify_round<GENES>(P, cq, cqn, cqn < WAVE ? cqn : WAVE, lane, W.mat_lds, W.st);
  }
  for (; D != 0;) {
    uint32_t t = 0;
    if (lane == 0)
      t = atomicAdd(P.tile_counter, 1u);
    t = __builtin_amdgcn_readfirstlane(t) * P.work_step + P.work_first;
    if (t >= P.ntiles)
      break;
    t += P.first_tile;
    const TileDesc td = P.tiles[t];
    const uint32_t L = __builtin_amdgcn_readfirstlane(td.len);
    const uint32_t nvalid = __builtin_amdgcn_readfirstlane(td.nvalid);
    const uint32_t *qr = P.qres + td.res_base + lane;   /* + (p / 4) * 64 */
    const bool valid = lane < nvalid;
    W.qslot = t * WAVE + lane;

    /* ---- hash of the query itself (zobrist_hash, zobrist.cc:74-88), and with
            -i the two shifted hashes that seed the rolling indel enumeration
            (zobrist_hash_delete_first :90-104, _insert_first :122-136) ---- */
    uint64_t h = 0;
    if (GENES)
      h = gene_keys[P.qv[W.qslot]] ^ gene_keys[P.n_v + P.qj[W.qslot]];
    uint64_t hdel = h, hins = h;
    {
      uint32_t w = 0;
      for (uint32_t p = 0; p < L; p++) {
        if ((p & 3u) == 0)
          w = qr[(p >> 2) * WAVE];
        const uint32_t r = (w >> ((p & 3u) * 8)) & 0xffu;
        h ^= zl[A * p + r];
        if (INDELS) {
          hins ^= zl[A * (p + 1) + r];
          if (p > 0)
            hdel ^= zl[A * (p - 1) + r];
        }
      }
    }

    /* ---- the unchanged sequence (generate_variants_0, variants.cc:260-268) */
    probe<GENES>(W, h, valid, pack_a(K_SAME, 0, 0), 0);

    if (D >= 1) {
      /* ---- single substitutions (variants.cc:280-293) ---- */
      uint32_t w = 0;
      for (uint32_t p = 0; p < L; p++) {
        if ((p & 3u) == 0)
          w = qr[(p >> 2) * WAVE];
        const uint32_t r = (w >> ((p & 3u) * 8)) & 0xffu;
        const uint64_t h1 = h ^ zl[A * p + r];
        const uint64_t *zrow = zs + A * p;
#pragma unroll 1
        for (uint32_t v = 0; v < (uint32_t)A; v++)
          probe<GENES>(W, h1 ^ zrow[v], valid && v != r, pack_a(K_SUB, p, v), 0);
      }
    }

    if (INDELS) {
      /* ---- deletions: one per run of equal residues, rolling hash
              (variants.cc:301-325; none when L == 1) ---- */
      if (L > 1) {
        uint32_t w = 0, gone = 0;
        uint64_t hd = hdel;
#pragma unroll 1
        for (uint32_t p = 0; p < L; p++) {
          if ((p & 3u) == 0)
            w = qr[(p >> 2) * WAVE];
          const uint32_t r = (w >> ((p & 3u) * 8)) & 0xffu;
          const bool fresh = (p == 0) || (r != gone);
          if (p > 0 && fresh)
            hd ^= zl[A * (p - 1) + gone] ^ zl[A * (p - 1) + r];
          probe<GENES>(W, hd, valid && fresh, pack_a(K_DEL, p, 0), 0);
          gone = r;
        }
      }
      /* ---- insertions: every residue in front of position 0, then behind
              each position every residue that differs from it
              (variants.cc:329-353) ---- */
      {
        uint64_t hi = hins;
        uint32_t w = 0, r = 0xffu;
        for (uint32_t ip = 0; ip <= L; ip++) {
          if (ip > 0) {
            const uint32_t p = ip - 1;
            if ((p & 3u) == 0)
              w = qr[(p >> 2) * WAVE];
            r = (w >> ((p & 3u) * 8)) & 0xffu;
            hi ^= zl[A * p + r] ^ zl[A * ip + r];
          }
          const uint64_t *zrow = zs + A * ip;
#pragma unroll 1
          for (uint32_t v = 0; v < (uint32_t)A; v++)
            probe<GENES>(W, hi ^ zrow[v], valid && v != r, pack_a(K_INS, ip, v), 0);
        }
      }
    }

    if (D >= 2) {
      /* ---- double substitutions p < q (variants.cc:370-399) ---- */
      for (uint32_t p = 0; p + 1 < L; p++) {
        const uint32_t rp = (qr[(p >> 2) * WAVE] >> ((p & 3u) * 8)) & 0xffu;
        const uint64_t hp = h ^ zl[A * p + rp];
        for (uint32_t v = 0; v < (uint32_t)A; v++) {
          const bool pv = valid && v != rp;
          const uint64_t hpv = hp ^ zs[A * p + v];
          const uint32_t ca = pack_a(K_SUB2, p, v);
          uint32_t w = 0;
          for (uint32_t qq = p + 1; qq < L; qq++) {
            if ((qq & 3u) == 0 || qq == p + 1)
              w = qr[(qq >> 2) * WAVE];
            const uint32_t rq = (w >> ((qq & 3u) * 8)) & 0xffu;
            const uint64_t hq = hpv ^ zl[A * qq + rq];
            const uint64_t *zrow = zs + A * qq;
#pragma unroll 1
            for (uint32_t x = 0; x < (uint32_t)A; x++)
              probe<GENES>(W, hq ^ zrow[x], pv && x != rq, ca, qq | (x << 24));
          }
        }
      }
    }
  }

  /* leftovers: fewer than 64 entries */
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (D != 0 && (int)lane < W.qn)
    resolve_entry<GENES>(P, W.q, (int)lane, W.mat_lds, W.st);

  /* statistics: wave reduction, one atomic per counter per wave */
  {
    unsigned long long s[STAT_COUNT] = {W.st.variants, W.st.bloom_pos,
                                        W.st.hash_eq, W.st.matches};
#pragma unroll
    for (int k = 0; k < STAT_COUNT; k++) {
      unsigned long long x = s[k];
      for (int off = 32; off > 0; off >>= 1)
        x += __shfl_down(x, off, WAVE);
      if (lane == 0 && x)
        atomicAdd(stats_dst(P) + k, x);
    }
  }

  if (P.lds_matrix) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cells; i += BLOCK_THREADS) {
      const unsigned long long x = mat_all[i];
      if (x)
        atomicAdd(matrix_dst(P) + i, x);
    }
  }
}

}  // namespace cmpr
#endif
