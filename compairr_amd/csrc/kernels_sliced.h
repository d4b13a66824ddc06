/*
 * kernels_sliced.h -- probe kernel, variant 1: Bloom slices staged in LDS.
 *
 * Same path as kernels.h (variant enumeration -> Zobrist hash -> Bloom probe
 * -> hash-table walk -> exact verify -> matrix accumulate; reference:
 * overlap.cc:253-284, variants.cc:260-428, bloompat.h:40-58, overlap.cc:168-251),
 * but the Bloom filter is laid out so that the probes of one query land in a
 * 32 KiB slice chosen by the query's class key (layout.h, "Sliced Bloom
 * layout").  A workgroup takes a chunk of tiles that share a slice, copies the
 * slice HBM -> LDS, and its waves answer every class-preserving probe (for
 * d = 1 substitutions: all but the k class positions, ~93 % at k = 1) from
 * LDS.  Class-changing variants (substitution at a class position, insertions,
 * deletions) compute their own slice and probe the filter in HBM.
 *
 * Probing is done a ROW at a time (one position, all A replacement residues),
 * in two phases:
 *   1. fully unrolled, branch-free: A hashes, their A filter words (LDS reads
 *      or HBM loads, all in flight together) and A computed bit patterns are
 *      reduced to one per-lane bit mask of Bloom-positive residues;
 *   2. a short loop pops the set bits (usually 0-3 per lane) and compacts the
 *      positives of the wave into the LDS queue with a ballot + prefix count.
 * The replacement-residue keys of a position are fetched once per row into one
 * VGPR pair (lane r holds the key of residue r) and broadcast with v_readlane.
 */
#ifndef COMPAIRR_AMD_KERNELS_SLICED_H
#define COMPAIRR_AMD_KERNELS_SLICED_H

#include "kernels.h"

namespace cmpr {

__device__ __forceinline__ uint64_t readlane64(uint64_t x, uint32_t l)
{
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, (int)l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), (int)l);
  return ((uint64_t)hi << 32) | lo;
}

/* per-wave state of the sliced kernel */
struct SProber {
  const ProbeParams  &P;
  const uint64_t     *slice_lds;
  WaveQueue          &q;
  unsigned long long *mat_lds;
  uint32_t            lane;
  uint32_t            qslot;
  uint32_t            wmask_bytes; /* (slice_words - 1) << 3                  */
  uint32_t            slice_shift; /* log2(bytes per slice)                   */
  uint32_t            smask;
  uint32_t            tile_slice;
  int                 qn;
  LaneStats           st;
  unsigned long long  held;        /* lane 0: the block of the positives buffer claimed ahead: its first entry,
                                      and in bits 48.. the segment it lies in */
  uint32_t            nclaims = 0; /* blocks claimed so far (wave-uniform): the segments are taken in turn */
};

/* `n` queue entries starting at `first` leave the wave: in deferred mode they
   are appended to the global positives buffer with ONE atomic claim (the walks
   happen later, in resolve_kernel); if the buffer is full, or in inline mode,
   lane k resolves entry first + k right here. */
/* The rare path (positives buffer full, or deferred_resolve = 0): the same walk,
   verification and scoring as resolve_one (kernels.h), one slot and one residue
   at a time with nothing unrolled -- inlined at every push site, the fast form
   cost the probe loops ~40 registers. */
template <bool GENES>
__device__ __forceinline__ void resolve_inline_slow(SProber &W, int first, int n)
{
  const ProbeParams &P = W.P;
  if ((int)W.lane >= n)
    return;
  const int e = first + (int)W.lane;
  const uint64_t key = table_key(W.q.hash[e]);
  const uint32_t qs = W.q.slot[e], ca = W.q.ca[e], cb = W.q.cb[e];
  const uint32_t bk = dir_bucket(key, P.dir_mask);
#pragma unroll 1
  for (uint32_t piece = bk;; piece++) {
    const RefRec *rec = (const RefRec *)P.rec2 + piece;
    const uint32_t ri = rec->idx, rl = rec->len, rh = rec->home;
    if (ri == REC_EMPTY)
      break;
    const bool last = walk_ends(ri, rl, rh, bk);
    if (rh == bk && (rl >> REC_TAG_SHIFT) == dir_tag(key)) {
      const uint8_t *rp = P.res2 + P.off2[ri];       /* (the rare path: the residues where the set lies) */
      W.st.hash_eq++;
      bool ok = true;
      if (GENES)
        ok = P.qrec[qs].v == rec->v && P.qrec[qs].j == rec->j;
      if (ok) {
        const uint32_t kind = ca & 7u, p1 = (ca >> 3) & 0xffffu, r1 = ca >> 24;
        const uint32_t p2 = cb & 0xffffu, r2 = cb >> 24;
        /* (record tiles -- layout.h ProbeParams::rec_tiles --: length and residues from the query's record) */
        const QueryRec *const qrec = P.qrec + qs;
        const uint32_t L = P.rec_tiles ? qrec->len : (uint32_t)P.qlen[qs], M = rl & 0xffffu;
        ok = M == (kind == K_DEL ? L - 1 : (kind == K_INS ? L + 1 : L));
        const uint32_t *qr = P.rec_tiles ? nullptr : P.qres + P.tiles[qs >> 6].res_base + (qs & 63u);
#pragma unroll 1
        for (uint32_t x = 0; ok && x < M; x++) {
          /* check_variant (variants.cc:166-240): residue the variant has at x */
          uint32_t want;
          if ((kind == K_SUB || kind == K_INS || kind == K_SUB2) && x == p1)
            want = r1;
          else if (kind == K_SUB2 && x == p2)
            want = r2;
          else {
            const uint32_t qp = kind == K_DEL ? (x < p1 ? x : x + 1)
                                              : (kind == K_INS ? (x < p1 ? x : x - 1) : x);
            want = ((P.rec_tiles ? qrec->res[qp >> 2] : qr[(size_t)(qp >> 2) * WAVE]) >> ((qp & 3u) * 8)) & 0xffu;
          }
          ok = want == (uint32_t)rp[x];
        }
        if (ok) {
          W.st.matches++;
          score_match(P, qs, ri, (uint64_t)P.R2 * P.qrec[qs].rep + rec->rep,
                      P.ignore_counts ? 1ull : P.qrec[qs].cnt, rec->cnt, W.mat_lds);
        }
      }
    }
    if (last)
      break;
  }
}

/* FALLBACK = false (the fast form of kernels_rows.h): entries that do not fit are
   dropped and the launch is flagged (ProbeParams::overflow); resolve_kernel then
   does nothing and a second launch of the kernel's FALLBACK = true form redoes the
   whole step resolving inline, so that capacity is never a limit either way. */
/* The positives buffer is claimed one block of 64 entries AHEAD: the claim (a
   device-scope atomic, 1-2 us until its answer is back) is issued when the
   previous block is written and waited for only when the next is, so a flush costs
   no round trip.  The last flush of a wave (`last`) claims nothing and pads its
   block with null entries (POS_NULL_SLOT). */
/* The segment of a wave's n-th block.  A workgroup used to append to ONE segment (b % segments); on skewed data --
   the cdr3 law: a few slices hold most of the neighbours -- the workgroups of those slices overfilled theirs while
   the others stayed empty, and the step fell back to its redo pass (10M x 10M d = 1 -i: 21 ms for 1.25; round 4).
   Now a wave's blocks go round the segments, starting at its workgroup's.  */
__device__ __forceinline__ uint32_t pos_segment_of(const ProbeParams &P, uint32_t nclaims)
{
  return (blockIdx.x + nclaims) & (P.pos_segments - 1);
}

__device__ __forceinline__ void claim_pos_block(SProber &W)
{
  const ProbeParams &P = W.P;
  if (P.pos_buf == nullptr)
    return;
  const uint32_t seg = pos_segment_of(P, W.nclaims);
  W.nclaims++;
  if (W.lane == 0)
    W.held = atomicAdd(P.pos_ctr + (size_t)seg * POS_CTR_STRIDE, (unsigned long long)WAVE) |
             ((unsigned long long)seg << 48);
}

template <bool GENES, bool FALLBACK = true>
__device__ __forceinline__ void flush_or_resolve(SProber &W, int first, int n, bool last = false)
{
  const ProbeParams &P = W.P;
  bool inline_resolve = FALLBACK && P.pos_buf == nullptr;
  if (!inline_resolve) {
    const uint32_t held_hi = __builtin_amdgcn_readfirstlane((uint32_t)(W.held >> 32));
    const uint32_t seg = held_hi >> 16;                    /* (the block claimed ahead names its segment) */
    unsigned long long *ctr = P.pos_ctr + (size_t)seg * POS_CTR_STRIDE;
    const unsigned long long base =
        ((unsigned long long)(held_hi & 0xffffu) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)W.held);
    if (!last)
      claim_pos_block(W);
    if (base < P.pos_cap) {                   /* the segment has 64 entries of slack */
      PosEntry e;
      e.hash = 0;
      e.slot = POS_NULL_SLOT;
      e.ca = e.cb = e.qbase = 0;
      if ((int)W.lane < n) {
        e.hash = W.q.hash[first + W.lane];
        e.slot = W.q.slot[first + W.lane];
        e.ca = W.q.ca[first + W.lane];
        e.cb = W.q.cb[first + W.lane];
      }
      P.pos_buf[(size_t)seg * (P.pos_cap + WAVE) + base + W.lane] = e;
    } else {
      if (W.lane == 0) {
        atomicMax(ctr + 1, ~base);
        if (!FALLBACK && n > 0)               /* (an empty block that does not fit loses nothing) */
          atomicOr(P.overflow, 1ull);
      }
      inline_resolve = FALLBACK;
    }
  }
  if (FALLBACK && inline_resolve)
    resolve_inline_slow<GENES>(W, first, n);
}

template <bool GENES, bool FALLBACK = true>
__device__ __forceinline__ void s_push(SProber &W, bool pos, uint64_t hv,
                                       uint32_t ca, uint32_t cb)
{
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
    if (W.qn >= WAVE) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      W.qn -= WAVE;
      if (!CMPR_DBG(W.P, DBG_SKIP_RESOLVE))
        flush_or_resolve<GENES, FALLBACK>(W, W.qn, WAVE);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
}

/* (a & b) | c in one instruction */
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t b, uint32_t c)
{
  uint32_t d;
  [[clang::noconvergent]] {
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  }
  return d;
}

/* non-zero = absent.  Inverted polarity, bloompat.h:55-58: present iff
   (word & pattern) == 0.  With the 4 shifts of pattern_fields this is 6
   instructions: shift-or (low dword pattern), (s2 | s3) & high dword as one
   3-input bit operation, and-or */
__device__ __forceinline__ uint32_t bloom_miss(uint64_t word, const BloomPat &p)
{
  const uint32_t y = (p.s2 | p.s3) & (uint32_t)(word >> 32);
  return and_or((uint32_t)word, p.lo, y);
}

__device__ __forceinline__ bool bloom_hit(uint64_t word, const BloomPat &p)
{
  return bloom_miss(word, p) == 0;
}

/* The dynamic LDS of the probe kernel starts at LDS address 0 (there is no
   static __shared__ in front of it; checked at kernel entry), so LDS reads can
   be addressed absolutely: no base add per read, constant offsets go into the
   instruction. */
typedef __attribute__((address_space(3))) const uint64_t lds_u64_t;
__device__ __forceinline__ uint64_t lds_u64(uint32_t byte_addr, int index = 0)
{
  return ((lds_u64_t *)(uintptr_t)byte_addr)[index];
}
/* the same, never paired with a neighbour into ds_read2_b64: two ds_read_b64 move
   their 2 x 512 bytes in 2 LDS cycles each, one ds_read2_b64 takes 8
   (MI355X_MICROARCH.md, LDS table) */
__device__ __forceinline__ uint64_t lds_u64_single(uint32_t byte_addr, int index)
{
  return ((volatile lds_u64_t *)(uintptr_t)byte_addr)[index];
}

/* phase 1, class-preserving row: filter words from the LDS copy of the slice */
template <int A>
__device__ __forceinline__ uint32_t row_lds(const SProber &W, uint64_t h1, uint64_t zrow)
{
  uint32_t mask = 0;
#pragma unroll
  for (int v = 0; v < A; v++) {
    const uint64_t hv = h1 ^ readlane64(zrow, v);
    const uint32_t woff = bloom_off(hv) & W.wmask_bytes;
    const uint64_t word = lds_u64(woff);
    mask |= bloom_hit(word, pattern_fields(hv)) ? (1u << v) : 0u;
  }
  return mask;
}

/* phase 1, class-preserving row over the OTHER residues of a position: the lane
   probes residue (r + k) mod A for k = 1 .. A-1, r being its own residue there
   (never the original: one probe less per row, nothing to mask afterwards).
   `zaddr` is the LDS address of the lane's own key zl[ZS * p + r]; key k is a
   read at constant offset 8 k from it (20 consecutive keys over the lanes: no
   bank conflict), which replaces the two cross-lane reads per probe of
   row_lds.  Bit k - 1 of the result <-> residue (r + k) mod A. */
template <int A>
__device__ __forceinline__ uint32_t row_lds_others(const SProber &W, uint64_t h1, uint32_t zaddr)
{
  /* software-pipelined by one probe: the filter word of probe k + 1 is
     requested before probe k is tested, so two LDS reads are in flight */
  uint32_t mask = 0;
  uint64_t hv = h1 ^ lds_u64_single(zaddr, 1);
  uint64_t word = lds_u64(bloom_off(hv) & W.wmask_bytes);
#pragma unroll
  for (int k = 1; k < A; k++) {
    uint64_t hv_n = 0, word_n = 0;
    if (k + 1 < A) {
      hv_n = h1 ^ lds_u64_single(zaddr, k + 1);
      word_n = lds_u64(bloom_off(hv_n) & W.wmask_bytes);
    }
    mask |= bloom_hit(word, pattern_fields(hv)) ? (1u << (k - 1)) : 0u;
    hv = hv_n;
    word = word_n;
  }
  return mask;
}

/* phase 2 of row_lds_others */
template <bool GENES, int A, bool RES_IN_A = true>
__device__ __forceinline__ void emit_row_others(SProber &W, uint32_t mask, uint64_t h1,
                                                uint32_t zaddr, uint32_t r, uint32_t ca,
                                                uint32_t cb = 0)
{
  if (CMPR_DBG(W.P, DBG_SKIP_EMIT))
    mask = 0;
  while (__ballot(mask != 0)) {
    const bool pos = mask != 0;
    const uint32_t k = pos ? (uint32_t)__ffs((int)mask) : 1u;
    const uint64_t hv = h1 ^ lds_u64(zaddr + 8u * k);
    uint32_t v = r + k;
    v = v >= (uint32_t)A ? v - (uint32_t)A : v;
    s_push<GENES>(W, pos, hv, RES_IN_A ? (ca | (v << 24)) : ca,
                  RES_IN_A ? cb : (cb | (v << 24)));
    mask &= mask - 1u;
  }
}

/* phase 1, class-changing row: slice of residue v = own slice ^ dk_lane ^ crow[v]
   (crow: lane v holds the class-key term of replacement residue v; it applies
   only to lanes whose variant class is split: crow_enable = all ones / zero) */
template <int A>
__device__ __forceinline__ uint32_t row_hbm(const SProber &W, uint64_t h1, uint64_t zrow,
                                            uint32_t dk_lane, uint32_t crow,
                                            uint32_t crow_enable = ~0u)
{
  /* batches of at most 10 loads in flight: enough to cover the latency, and the
     register footprint stays under the 128 VGPRs that 4 waves per SIMD allow */
  constexpr int B = A > 10 ? 10 : A;
  uint32_t mask = 0;
#pragma unroll
  for (int v0 = 0; v0 < A; v0 += B) {
    uint64_t word[B];
#pragma unroll
    for (int k = 0; k < B; k++) {
      const int v = v0 + k;
      const uint64_t hv = h1 ^ readlane64(zrow, v);
      const uint32_t cv = (uint32_t)__builtin_amdgcn_readlane((int)crow, v);
      const uint32_t vslice = (W.tile_slice ^ dk_lane ^ (cv & crow_enable)) & W.smask;
      const uint32_t woff = bloom_off(hv) & W.wmask_bytes;
      const uint64_t off = ((uint64_t)vslice << W.slice_shift) + woff;
      word[k] = *(const uint64_t *)((const char *)W.P.bloom + off);
    }
#pragma unroll
    for (int k = 0; k < B; k++) {
      /* the hash is recomputed (2 readlane + 2 xor) rather than kept live
         across the loads */
      const uint64_t hv = h1 ^ readlane64(zrow, v0 + k);
      mask |= bloom_hit(word[k], pattern_fields(hv)) ? (1u << (v0 + k)) : 0u;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  return mask;
}

/* phase 2: pop the set bits of `mask`, queue the positives.  zl_row = LDS keys
   of this position (A entries); the residue goes into ca (RES_IN_A) or cb. */
template <bool GENES, bool RES_IN_A>
__device__ __forceinline__ void emit_row(SProber &W, uint32_t mask, uint64_t h1,
                                         const uint64_t *zl_row, uint32_t ca, uint32_t cb)
{
  if (CMPR_DBG(W.P, DBG_SKIP_EMIT))
    mask = 0;
  while (__ballot(mask != 0)) {
    const bool pos = mask != 0;
    const uint32_t v = pos ? (uint32_t)__ffs((int)mask) - 1u : 0u;
    const uint64_t hv = h1 ^ zl_row[v];
    s_push<GENES>(W, pos, hv, RES_IN_A ? (ca | (v << 24)) : ca,
                  RES_IN_A ? cb : (cb | (v << 24)));
    mask &= mask - 1u;
  }
}

/* one probe, class-preserving: filter word from the staged slice */
__device__ __forceinline__ bool probe_one_lds(const SProber &W, uint64_t hv)
{
  const uint32_t woff = bloom_off(hv) & W.wmask_bytes;
  const uint64_t word = *(const uint64_t *)((const char *)W.slice_lds + woff);
  return bloom_hit(word, pattern_fields(hv));
}

/* one probe of the tile's own slice: LDS copy when staged, else where it lies */
__device__ __forceinline__ bool probe_one_own(const SProber &W, uint64_t hv, bool staged)
{
  const uint32_t woff = bloom_off(hv) & W.wmask_bytes;
  uint64_t word;
  if (staged)
    word = *(const uint64_t *)((const char *)W.slice_lds + woff);
  else
    word = *(const uint64_t *)((const char *)W.P.bloom +
                               ((uint64_t)W.tile_slice << W.slice_shift) + woff);
  return bloom_hit(word, pattern_fields(hv));
}

/* address of the filter word of a class-changing variant (dk = class-key delta) */
__device__ __forceinline__ const uint64_t *hbm_word(const SProber &W, uint64_t hv, uint32_t dk)
{
  const uint32_t vslice = (W.tile_slice ^ dk) & W.smask;
  const uint32_t woff = bloom_off(hv) & W.wmask_bytes;
  return (const uint64_t *)((const char *)W.P.bloom + ((uint64_t)vslice << W.slice_shift) + woff);
}

/* nucleotides: the three OTHER residues of a position, all three filter words in flight
   together (one wait per position, not one per probe: with a wait behind every read the
   wave stood still for an LDS round trip three times per position -- the vector unit of
   the d = 2 kernels ran at a third of its issue rate).  `eaddr` = LDS address of the
   lane's entry of the delta table (ze: key k at + 8 k); bit k - 1 of the result <->
   residue (r + k) & 3. */
__device__ __forceinline__ uint32_t probe3_lds(const SProber &W, uint64_t hbase, uint32_t eaddr)
{
  uint64_t hv[3], word[3];
#pragma unroll
  for (int k = 0; k < 3; k++)
    hv[k] = hbase ^ lds_u64(eaddr, k + 1);
#pragma unroll
  for (int k = 0; k < 3; k++)
    word[k] = lds_u64(bloom_off(hv[k]) & W.wmask_bytes);
  __builtin_amdgcn_sched_barrier(0);
  BloomPat pat[3];
#pragma unroll
  for (int k = 0; k < 3; k++)
    pat[k] = pattern_fields(hv[k]);
  __builtin_amdgcn_sched_barrier(0);
  uint32_t b3 = 0;
#pragma unroll
  for (int k = 0; k < 3; k++)
    b3 |= bloom_hit(word[k], pat[k]) ? (1u << k) : 0u;
  return b3;
}

/* The residues of a tile, streamed one dword (4 positions x 64 lanes) ahead of
   use so that the HBM/L2 latency of the next dword hides behind the rows of
   the current one. */
struct ResStream {
  const uint32_t *qr;      /* + (p / 4) * 64 */
  uint32_t        L;
  uint32_t        cur, nxt;
  __device__ __forceinline__ void start(const uint32_t *q, uint32_t len)
  {
    qr = q;
    L = len;
    cur = 0;
    nxt = len ? q[0] : 0u;
  }
  /* residue at position p; positions must be visited in increasing order */
  __device__ __forceinline__ uint32_t at(uint32_t p)
  {
    if ((p & 3u) == 0) {
      cur = nxt;
      if (p + 4 < L)
        nxt = qr[((p >> 2) + 1) * WAVE];
    }
    return (cur >> ((p & 3u) * 8)) & 0xffu;
  }
};

/*
 * LDS: [2^w-word Bloom slice][A * zpos Zobrist keys][R1 * R2 matrix (optional)]
 *      [NW WaveQueues][CR tables][heavy bitmap]
 *      [chunk broadcast][tile descriptors of the chunk]
 */
template <int A, int D, bool INDELS, bool GENES, int NW>
__global__ void __launch_bounds__(NW * WAVE, 4)
probe_sliced_kernel(const ProbeParams P)
{
  constexpr uint32_t NT = NW * WAVE;
  constexpr uint32_t MCR = kernel_class_res(A, false);   /* unrolled class-residue loops */
  constexpr uint32_t NTB = 10;                 /* nucleotide positions per block (3 mask bits each) */
  extern __shared__ __align__(16) unsigned char smem[];
  /* the Bloom slice sits at LDS address 0: its reads need no base add */
  uint64_t *slice_lds = (uint64_t *)smem;
  const uint32_t slice_words = 1u << P.geom.words_log2;
  uint64_t *zl = slice_lds + slice_words;
  /* amino acids: every row of keys is stored twice in a line, so that
     "residue (r + k) mod A" is entry r + k (row_lds_others) */
  constexpr uint32_t ZS = zrow_stride(A);
  const uint32_t nz = ZS * P.zpos;
  if ((uint32_t)(uintptr_t)smem != 0u)
    __builtin_trap();                       /* lds_u64 addresses the LDS absolutely */
  /* nucleotides: per position and residue r the keys {Z[r], Z[r]^Z[r+1], Z[r]^Z[r+2],
     Z[r]^Z[r+3]} (residues mod 4): the hash of "r replaced by (r + k) & 3" is the
     current hash ^ entry k, a read at constant offset from the lane's own entry */
  constexpr uint32_t ZE = zdelta_entries(A);
  uint64_t *ze = zl + nz;
  const uint32_t nze = ZE * P.zpos;
  unsigned long long *mat_all = (unsigned long long *)(ze + nze);
  const uint32_t cells = P.lds_matrix ? P.R1 * P.R2 : 0u;   /* LDS copy only: <= 2048 cells */
  WaveQueue *queues = (WaveQueue *)(mat_all + (P.lds_matrix ? cells : 0));
  uint32_t *cr_lds = (uint32_t *)(queues + NW);
  uint32_t *hv_lds = cr_lds + MAX_CLASS_RES * A;     /* heavy-class bitmap */
  uint32_t *bcast = hv_lds + HEAVY_WORDS;
  TileRef *tref_lds = (TileRef *)(bcast + 4);           /* chunk_cap entries */

  for (uint32_t i = threadIdx.x; i < nz; i += NT)
    zl[i] = P.zob[(i / ZS) * A + (i % ZS) % A];
  for (uint32_t i = threadIdx.x; i < nze; i += NT) {
    const uint32_t pos = i / 16u, r = (i / 4u) & 3u, k = i & 3u;
    const uint64_t own = P.zob[pos * 4u + r];
    ze[i] = k ? own ^ P.zob[pos * 4u + ((r + k) & 3u)] : own;
  }
  if (P.lds_matrix)
    for (uint32_t i = threadIdx.x; i < cells; i += NT)
      mat_all[i] = 0;
  for (uint32_t i = threadIdx.x; i < MAX_CLASS_RES * (uint32_t)A; i += NT)
    cr_lds[i] = P.geom.ctab[P.geom.off_cr + i];
  if (INDELS)
    for (uint32_t i = threadIdx.x; i < HEAVY_WORDS; i += NT)
      hv_lds[i] = P.geom.ctab[P.geom.off_hv + i];

  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x / WAVE;
  const uint32_t KH = P.geom.k;           /* class residues of heavy classes */
  SProber W{P, slice_lds, queues[wave], P.lds_matrix ? mat_all : nullptr,
            lane, 0u, (slice_words - 1u) << 3, P.geom.words_log2 + 3u, P.geom.smask,
            0u, 0, {0ull, 0u, 0u, 0u}, 0ull};
  claim_pos_block(W);
  const uint32_t zl_addr = slice_words * 8u;       /* LDS address of zl */
  const uint32_t ze_addr = zl_addr + nz * 8u;      /* ... and of ze */
  const uint32_t zlane = lane < (uint32_t)A ? lane : 0u;   /* lane r <-> residue r */

  /* Two kinds of work.  Block phase: chunks (several tiles that need the same
     slice) are claimed by the whole workgroup, which stages the slice in LDS.
     Wave phase: tiles whose slice holds too few queries to be worth a
     workgroup (many slices, few queries) are claimed one at a time by single
     waves and probe their slice where it lies, in HBM / L2. */
  bool block_phase = true;
  for (;;) {
    Chunk ck;
    ck.slice = 0;
    ck.first_tile = 0;
    ck.ntiles = 0;
    ck.pass = 0;
    if (block_phase) {
      /* ---- next chunk: tiles of one slice; stage that slice into LDS ---- */
      __syncthreads();                     /* everyone is done with the old slice */
      if (threadIdx.x == 0) {
        bcast[0] = atomicAdd(P.tile_counter, 1u);
        bcast[1] = 0;                      /* tiles of the chunk handed out so far */
      }
      __syncthreads();
      const uint32_t item = bcast[0];
      if (item >= P.nchunks) {
        block_phase = false;               /* the same for every thread of the block */
      } else {
        ck = P.chunks[item];
        const uint64_t *src = P.bloom + ((uint64_t)ck.slice << P.geom.words_log2);
        for (uint32_t i = threadIdx.x; i < slice_words; i += NT)
          slice_lds[i] = src[i];
        /* (an item chunk -- pass >= 3 -- has no tile references: its "tiles" are
           blocks of 64 items from item ck.first_tile on) */
        for (uint32_t i = threadIdx.x; (ck.pass & 0xffu) < 3 && i < ck.ntiles; i += NT)
          tref_lds[i] = P.tile_refs[ck.first_tile + i];
        __syncthreads();
      }
    }
    const bool staged = block_phase;       /* own-slice probes: LDS or HBM */
    /* a main chunk may hand out, behind its tiles, the item blocks of its slice (the
       slice is staged once for both; idle waves of a chunk with one or two long tiles
       get work) */
    uint32_t ride_first = 0, ride_blocks = 0;
    if (D >= 2 && block_phase && (ck.pass & CHUNK_WITH_ITEMS)) {
      ride_first = P.slice_items[2 * ck.slice];
      ride_blocks = P.slice_items[2 * ck.slice + 1];
    }
    const uint32_t chunk_pass = ck.pass & 0xffu;      /* 0 main, 1 insertions, 2 deletions, >= 3 items */

    bool all_done = false;
    for (;;) {
      uint32_t t;
      TileDesc td;
      uint32_t item0 = 0;                  /* item tile: its first item */
      uint32_t pass = chunk_pass;          /* (of this tile) */
      if (block_phase) {
        uint32_t tk = 0;
        if (lane == 0)
          tk = atomicAdd(&bcast[1], 1u);
        tk = __builtin_amdgcn_readfirstlane(tk);
        if (tk >= ck.ntiles + ride_blocks)
          break;
        pass = chunk_pass;
        /* an item block: one of an item chunk, or one riding along behind the tiles of a
           main chunk */
        const bool riding = D >= 2 && tk >= ck.ntiles;
        if (riding || (D >= 2 && pass >= 3)) {
          item0 = riding ? ride_first + (tk - ck.ntiles) * WAVE : ck.first_tile + tk * WAVE;
          pass = riding ? 3u : pass;
          t = 0;
          td.len = 0;
          td.nvalid = WAVE;
          td.res_base = 0;
          td.pass = pass;
          td.slice = ck.slice;
          td.k = P.geom.k;
        } else {
          t = __builtin_amdgcn_readfirstlane(tref_lds[tk].t);
          td = tref_lds[tk].td;
        }
      } else {
        uint32_t i = P.nsmall;
        if (lane == 0 && P.nsmall)             /* (none: not even the claim) */
          i = atomicAdd(P.tile_counter + 1, 1u);
        i = __builtin_amdgcn_readfirstlane(i);
        if (i >= P.nsmall) {
          all_done = true;
          break;
        }
        t = P.small_tiles[i];
        td = P.tiles[t];
      }
      if (CMPR_DBG(P, DBG_SKIP_TILES))
        continue;                          /* measures claiming + staging alone */
      /* An item tile (pass >= 3; nucleotides, d = 2: the double substitutions that put
         residue + k on class position c0 + i, staged here is the slice THAT variant
         lives in): 64 queries from anywhere, each lane with its own slot and its own
         residues; L = the longest of them. */
      const bool item_tile = D >= 2 && staged && pass >= 3;
      uint32_t islot = 0xffffffffu;
      uint64_t ihash = 0;                  /* the item's query: its hash and its */
      ResPack ipk{};                       /* residues, 2 bits each              */
      uint32_t icrp = 0;
      if (item_tile) {
        const ItemRec it = P.items[item0 + lane];
        islot = it.main;
        ihash = it.w;
        ipk = P.cpk[item0 + lane];
        icrp = it.rp;
      }
      const uint32_t nvalid = __builtin_amdgcn_readfirstlane(td.nvalid);
      const uint32_t K = __builtin_amdgcn_readfirstlane(td.k);   /* 0: light tile */
      W.tile_slice = __builtin_amdgcn_readfirstlane(td.slice);
      const bool valid = item_tile ? islot != 0xffffffffu : lane < nvalid;
      const uint32_t *qr = P.qres + td.res_base + lane;       /* (item tiles: unused) */
      const uint32_t vmask = valid ? ~0u : 0u;
      W.qslot = item_tile ? islot : t * WAVE + lane;
      /* own length: tiles without -i may mix lengths (L = the longest one) */
      const uint32_t Ll = valid ? (uint32_t)P.qlen[W.qslot] : 0u;
      uint32_t Lmax = Ll;
      if (item_tile)
        for (int off = 32; off > 0; off >>= 1)
          Lmax = max(Lmax, (uint32_t)__shfl_xor((int)Lmax, off, WAVE));
      const uint32_t L = item_tile ? __builtin_amdgcn_readfirstlane(Lmax)
                                   : __builtin_amdgcn_readfirstlane(td.len);
      auto res_at = [&](uint32_t p) -> uint32_t {
        return (qr[(p >> 2) * WAVE] >> ((p & 3u) * 8)) & 0xffu;
      };

      /* ---- query hash (zobrist.cc:74-88) and, with -i, the two shifted
              hashes of the rolling indel enumeration (:90-104, :122-136) ---- */
      ResStream rs;
      rs.start(qr, item_tile ? 0u : L);
      uint64_t h = 0;
      if (GENES && !item_tile)
        h = P.qgh[W.qslot];
      uint64_t hdel = h, hins = h;
      if (item_tile) {
        h = ihash;
      } else {
        for (uint32_t p = 0; p < L; p++) {
          const uint32_t r = rs.at(p);
          h ^= p < Ll ? zl[ZS * p + r] : 0ull;
          if (INDELS) {
            hins ^= zl[ZS * (p + 1) + r];
            if (p > 0)
              hdel ^= zl[ZS * (p - 1) + r];
          }
        }
      }

      /* class positions of this length (wave-uniform) */
      uint32_t m[MCR];
#pragma unroll
      for (uint32_t i = 0; i < MCR; i++)
        m[i] = class_pos(L, i, P.geom.c0);
      /* ... as a bit set over the positions (two scalar registers): the test sits
         in the innermost loops, where eight compares per position made the
         scalar unit the bottleneck of the nucleotide kernels */
      uint64_t cpos_lo = 0, cpos_hi = 0;
#pragma unroll
      for (uint32_t i = 0; i < MCR; i++)
        if (i < K) {
          if (m[i] < 64u)
            cpos_lo |= 1ull << m[i];
          else if (m[i] < 128u)
            cpos_hi |= 1ull << (m[i] - 64u);
        }
      const bool cpos_far = L > 128u;          /* (positions >= 128: the compare loop) */
      auto is_class_pos = [&](uint32_t p) -> bool {
        if (!cpos_far)
          return ((p < 64u ? cpos_lo >> p : cpos_hi >> (p - 64u)) & 1ull) != 0;
        bool c = false;
#pragma unroll
        for (uint32_t i = 0; i < MCR; i++)
          c = c || (i < K && m[i] == p);
        return c;
      };
      /* class-position flags of the NTB positions from p0 on (bit j: position p0 + j) */
      auto class_bits = [&](uint32_t p0) -> uint32_t {
        if (cpos_far) {
          uint32_t b = 0;
          for (uint32_t j = 0; j < NTB; j++)
            b |= is_class_pos(p0 + j) ? (1u << j) : 0u;
          return b;
        }
        uint64_t x;
        if (p0 == 0)
          x = cpos_lo;
        else if (p0 < 64u)
          x = (cpos_lo >> p0) | (cpos_hi << (64u - p0));
        else
          x = p0 < 128u ? cpos_hi >> (p0 - 64u) : 0ull;
        return (uint32_t)x & ((1u << NTB) - 1u);
      };
      auto class_bits8 = [&](uint32_t p0) -> uint32_t { return class_bits(p0) & 0xffu; };
      /* class-key terms of position p: of residue `r` (per lane) and, as a
         row, of every replacement residue (lane v holds the term of v) */
      auto class_terms = [&](uint32_t p, uint32_t r, uint32_t &crow) -> uint32_t {
        uint32_t dk = 0;
        crow = 0;
#pragma unroll
        for (uint32_t i = 0; i < MCR; i++)
          if (i < K && m[i] == p) {
            dk ^= cr_lds[i * A + r];
            crow ^= cr_lds[i * A + zlane];
          }
        return dk;
      };

      uint64_t nvar = pass == 0 ? 1 : 0;

      /* ---- the unchanged sequence (variants.cc:260-268) ---- */
      if (pass == 0)
        s_push<GENES>(W, valid && probe_one_own(W, h, staged), h, pack_a(K_SAME, 0, 0), 0);

      if (D >= 1 && pass == 0) {
        /* ---- single substitutions (variants.cc:280-293) ---- */
        nvar += (uint64_t)(A - 1) * Ll;
        rs.start(qr, L);
        if constexpr (A == 4) {
          /* Nucleotides: a row has only 3 variants, so NTB positions are probed
             per block -- the three OTHER residues of each lane, (r + k) & 3, so
             that no probe is wasted on the original -- and their positives
             share one 30-bit mask and one compaction loop. */
          for (uint32_t p0 = 0; p0 < L; p0 += NTB) {
            uint32_t mask = 0;
            const uint32_t n = L - p0 < NTB ? L - p0 : NTB;
            const uint32_t cbits = class_bits(p0);
            for (uint32_t jj = 0; jj < n; jj++) {
              const uint32_t p = p0 + jj;
              const uint32_t r = rs.at(p);
              uint32_t b3 = 0;
              if (staged && !((cbits >> jj) & 1u)) {
                const uint32_t eaddr = ze_addr + (16u * p + 4u * r) * 8u;
                b3 = probe3_lds(W, h, eaddr);
              } else if (!((cbits >> jj) & 1u)) {
                const uint64_t *zp = zl + 4 * p;
                const uint64_t h1 = h ^ zp[r];
#pragma unroll
                for (uint32_t k = 1; k <= 3; k++)
                  b3 |= probe_one_own(W, h1 ^ zp[(r + k) & 3u], false) ? (1u << (k - 1)) : 0u;
              } else {
                const uint64_t *zp = zl + 4 * p;
                const uint64_t h1 = h ^ zp[r];
                uint32_t crow;
                const uint32_t dk_r = class_terms(p, r, crow);
                uint64_t word[3];
#pragma unroll
                for (uint32_t k = 1; k <= 3; k++) {
                  const uint32_t v = (r + k) & 3u;
                  const uint32_t dk_v = class_terms(p, v, crow);
                  word[k - 1] = *hbm_word(W, h1 ^ zp[v], dk_r ^ dk_v);
                }
#pragma unroll
                for (uint32_t k = 1; k <= 3; k++)
                  b3 |= bloom_hit(word[k - 1], pattern_fields(h1 ^ zp[(r + k) & 3u])) ? (1u << (k - 1)) : 0u;
              }
              mask |= (p < Ll ? b3 : 0u) << (3u * jj);
            }
            if (!valid || CMPR_DBG(P, DBG_SKIP_EMIT))
              mask = 0;
            while (__ballot(mask != 0)) {
              const bool pos = mask != 0;
              const uint32_t b = pos ? (uint32_t)__ffs((int)mask) - 1u : 0u;
              const uint32_t p = p0 + b / 3u;
              const uint32_t r = res_at(p < L ? p : 0);
              const uint32_t k = b % 3u + 1u;
              const uint64_t hv = h ^ ze[16u * p + 4u * r + k];
              s_push<GENES>(W, pos, hv, pack_a(K_SUB, p, (r + k) & 3u), 0);
              mask &= mask - 1u;
            }
          }
        } else
        for (uint32_t p = 0; p < L; p++) {
          const uint32_t r = rs.at(p);
          if (staged && !is_class_pos(p)) {
            const uint32_t zaddr = zl_addr + (ZS * p + r) * 8u;
            const uint64_t h1 = h ^ lds_u64(zaddr);
            uint32_t mask = 0;
            if (!CMPR_DBG(P, DBG_SKIP_LDS_ROWS))
              mask = row_lds_others<A>(W, h1, zaddr);
            mask &= p < Ll ? ~0u : 0u;                    /* not past the lane's own end */
            emit_row_others<GENES, A>(W, mask, h1, zaddr, r, pack_a(K_SUB, p, 0));
            continue;
          }
          const uint64_t h1 = h ^ zl[ZS * p + r];
          const uint64_t zrow = zl[ZS * p + zlane];
          uint32_t mask = 0;
          if (!is_class_pos(p)) {
            if (!CMPR_DBG(P, DBG_SKIP_LDS_ROWS))
              mask = staged ? row_lds<A>(W, h1, zrow) : row_hbm<A>(W, h1, zrow, 0u, 0u);
          } else if (!CMPR_DBG(P, DBG_SKIP_HBM_ROWS)) {
            uint32_t crow;
            const uint32_t dk = class_terms(p, r, crow);
            mask = row_hbm<A>(W, h1, zrow, dk, crow);
          }
          mask &= (p < Ll ? ~0u : 0u) & ~(1u << r);      /* not past the lane's own end, not the original residue */
          emit_row<GENES, true>(W, mask, h1, zl + ZS * p, pack_a(K_SUB, p, 0), 0);
        }
      }

      if (INDELS) {
        /* Indel variants change the length, hence the class:
             ckey' = base ^ CL[L] ^ CL[L'] ^ (heavy(base') ? class residues of the variant : 0)
           while the tile's slice is ckey = base ^ (K ? class residues of the query : 0). */
        /* Most insertion variants of a tile fall into ONE other slice, the
           "sibling" own ^ CL[L] ^ CL[L+1] (all of them when neither class is
           split; for split classes those rows that leave the class residues in
           place), and most deletion variants into own ^ CL[L] ^ CL[L-1].  The
           host schedules the indel rows as separate passes over the tiles of one
           (slice, length) group with that sibling slice staged in LDS; a lane
           whose variant lands elsewhere probes the filter in HBM. */
        const bool do_del = pass == 2;
        const bool do_ins = pass == 1;
        const uint32_t sibling = ck.slice;     /* the slice staged for this pass */
        const uint32_t cl_L = P.geom.ctab[L];
        uint32_t base = cl_L;
        if (GENES)
          base ^= P.geom.ctab[P.geom.off_cv + P.qv[W.qslot]] ^
                  P.geom.ctab[P.geom.off_cj + P.qj[W.qslot]];
        auto heavy_of = [&](uint32_t bs) -> uint32_t {
          const uint32_t b = bs >> (32 - HEAVY_BUCKETS_LOG2);
          return (KH > 0 && ((hv_lds[b >> 5] >> (b & 31u)) & 1u)) ? ~0u : 0u;
        };
        uint32_t cbase = 0;                      /* XOR_i CR[i][s[m_i(L)]], heavy tiles */
#pragma unroll
        for (uint32_t i = 0; i < MCR; i++)
          if (i < K && L > 0)
            cbase ^= cr_lds[i * A + res_at(m[i])];

        /* ---- deletions (variants.cc:301-325): u = s without position p,
                u[x] = x < p ? s[x] : s[x + 1]; one per run of equal residues.
                Blocks of up to 32 positions: phase 1 rolls the hash and gathers
                the filter words, phase 2 replays the roll for the positives. */
        if (L > 1 && do_del && !CMPR_DBG(P, DBG_SKIP_DEL_ROWS)) {
          const uint32_t dlen = cl_L ^ P.geom.ctab[L - 1];
          const uint32_t hv = heavy_of(base ^ dlen);          /* is the variant's class split? */
          const uint32_t dl = dlen ^ cbase;
          uint32_t md[MCR], lo[MCR], hi[MCR];
#pragma unroll
          for (uint32_t i = 0; i < MCR; i++) {
            md[i] = class_pos(L - 1, i, P.geom.c0);
            lo[i] = hi[i] = 0;
            if (i < KH) {
              lo[i] = cr_lds[i * A + res_at(md[i])] & hv;
              hi[i] = cr_lds[i * A + res_at(md[i] + 1)] & hv;
            }
          }
          uint64_t hd = hdel;
          uint32_t gone = 0;
          for (uint32_t p0 = 0; p0 < L; p0 += 32) {
            const uint32_t pe = p0 + 32 < L ? p0 + 32 : L;
            const uint64_t hd0 = hd;
            const uint32_t gone0 = gone;
            uint32_t mask = 0, w = 0;
#pragma unroll 4
            for (uint32_t p = p0; p < pe; p++) {
              if ((p & 3u) == 0 || p == p0)
                w = qr[(p >> 2) * WAVE];
              const uint32_t r = (w >> ((p & 3u) * 8)) & 0xffu;
              const bool fresh = (p == 0) || (r != gone);
              if (p > 0 && fresh)
                hd ^= zl[ZS * (p - 1) + gone] ^ zl[ZS * (p - 1) + r];
              uint32_t dk = dl;
#pragma unroll
              for (uint32_t i = 0; i < MCR; i++)
                dk ^= md[i] < p ? lo[i] : hi[i];
              const uint32_t vslice = valid ? (W.tile_slice ^ dk) & W.smask : sibling;
              const uint32_t woff = bloom_off(hd) & W.wmask_bytes;
              uint64_t word;
              if (vslice == sibling)
                word = *(const uint64_t *)((const char *)slice_lds + woff);
              else
                word = *(const uint64_t *)((const char *)P.bloom +
                                           ((uint64_t)vslice << W.slice_shift) + woff);
              const BloomPat pat = pattern_fields(hd);
              nvar += fresh ? 1u : 0u;
              mask |= (fresh && bloom_hit(word, pat)) ? (1u << (p - p0)) : 0u;
              gone = r;
            }
            mask &= vmask;
            if (__ballot(mask != 0)) {
              uint64_t hr = hd0;
              uint32_t g = gone0;
#pragma unroll 1
              for (uint32_t p = p0; p < pe; p++) {
                const uint32_t r = res_at(p);
                if (p > 0 && r != g)
                  hr ^= zl[ZS * (p - 1) + g] ^ zl[ZS * (p - 1) + r];
                s_push<GENES>(W, (mask >> (p - p0)) & 1u, hr, pack_a(K_DEL, p, 0), 0);
                g = r;
              }
            }
          }
        }

        /* ---- insertions (variants.cc:329-353): u = s with v put in front of
                position ip, u[x] = x < ip ? s[x] : x == ip ? v : s[x - 1] ---- */
        if (do_ins && !CMPR_DBG(P, DBG_SKIP_INS_ROWS)) {
          nvar += (uint64_t)A + (uint64_t)(A - 1) * L;
          const uint32_t dlen = cl_L ^ P.geom.ctab[L + 1];
          const uint32_t hv = heavy_of(base ^ dlen);
          const uint32_t dl = dlen ^ cbase;
          uint32_t mi[MCR], lo[MCR], hi[MCR];
#pragma unroll
          for (uint32_t i = 0; i < MCR; i++) {
            mi[i] = class_pos(L + 1, i, P.geom.c0);
            lo[i] = hi[i] = 0;
            if (i < KH) {
              if (mi[i] < L)
                lo[i] = cr_lds[i * A + res_at(mi[i])] & hv;
              if (mi[i] >= 1)
                hi[i] = cr_lds[i * A + res_at(mi[i] - 1)] & hv;
            }
          }
          uint64_t hi_hash = hins;
          uint32_t r = 0xffu;
          rs.start(qr, L);
          for (uint32_t ip = 0; ip <= L; ip++) {
            if (ip > 0) {
              const uint32_t p = ip - 1;
              r = rs.at(p);
              hi_hash ^= zl[ZS * p + r] ^ zl[ZS * ip + r];
            }
            uint32_t dk0 = dl, crow = 0;
            bool v_on_class_pos = false;
#pragma unroll
            for (uint32_t i = 0; i < MCR; i++)
              if (i < KH) {
                if (mi[i] == ip) {
                  crow ^= cr_lds[i * A + zlane];          /* u[mi] = v */
                  v_on_class_pos = true;
                } else {
                  dk0 ^= mi[i] < ip ? lo[i] : hi[i];
                }
              }
            const uint64_t zrow = zl[ZS * ip + zlane];
            /* rows that put v on a class position spread over up to A slices */
            /* (idle lanes of a partial tile count as in LDS: they must not drag the
               wave through the HBM row) */
            const bool in_lds = !valid || (!(v_on_class_pos && hv) &&
                                           ((W.tile_slice ^ dk0) & W.smask) == sibling);
            uint32_t mask = 0;          /* bit v: residue v */
            uint32_t mask_o = 0;        /* bit k - 1: residue (r + k) mod A (row_lds_others) */
            const uint32_t zaddr = zl_addr + (ZS * ip + (ip > 0 ? r : 0u)) * 8u;
            if (__ballot(in_lds) && !CMPR_DBG(P, DBG_SKIP_LDS_ROWS)) {
              if (A != 4 && ip > 0) {          /* (nucleotide rows are not stored twice) */
                const uint32_t ml = row_lds_others<A>(W, hi_hash, zaddr);
                mask_o = in_lds ? ml : 0u;
              } else {
                const uint32_t ml = row_lds<A>(W, hi_hash, zrow);
                mask = in_lds ? ml : 0u;
              }
            }
            if (__ballot(!in_lds) && !CMPR_DBG(P, DBG_SKIP_HBM_ROWS)) {
              if (!in_lds)
                mask = row_hbm<A>(W, hi_hash, zrow, dk0, crow, hv);
            }
            mask &= vmask;
            mask_o &= vmask;
            if (ip > 0)
              mask &= ~(1u << r);                         /* v != s[ip - 1] */
            emit_row_others<GENES, A>(W, mask_o, hi_hash, zaddr, r, pack_a(K_INS, ip, 0));
            emit_row<GENES, true>(W, mask, hi_hash, zl + ZS * ip, pack_a(K_INS, ip, 0), 0);
          }
        }
      }

      if (D >= 2 && pass == 0) {
        /* ---- double substitutions p < q (variants.cc:370-399) ---- */
        nvar += (uint64_t)(A - 1) * (A - 1) * ((uint64_t)Ll * (Ll ? Ll - 1 : 0) / 2);
        if constexpr (A == 4) {
          /* Nucleotides: both substitutions enumerate the three OTHER residues of
             the lane ((r + k) & 3), so none of the 9 (instead of 16) probes per
             position pair is masked; the second position runs in blocks of NTB
             that share one 30-bit mask and one compaction loop. */
          /* split queries that hold every class position unwrapped: their pairs with a
             class position are items of a later pass (query_layout.hip for_each_item) */
          const bool cls_items = P.sub2_items && K > 0 && L >= P.geom.c0 + K;
          /* (an item holds RESPACK_MAX residues: longer queries of such a tile keep those pairs) */
          const bool lane_items = cls_items && Ll <= RESPACK_MAX;
          const bool all_items = cls_items && __ballot(valid && Ll > RESPACK_MAX) == 0;
          for (uint32_t p = 0; p + 1 < L; p++) {
            const uint32_t rp = res_at(p);
            const bool cp = is_class_pos(p);
            if (cp && all_items)
              continue;
            uint32_t crow_unused;
            const uint32_t dk_rp = cp ? class_terms(p, rp, crow_unused) : 0u;
            const bool fast = staged && !cp;       /* wave-uniform */
#pragma unroll 1
            for (uint32_t kp = 1; kp <= 3; kp++) {
              const uint32_t vp = (rp + kp) & 3u;
              const uint64_t hpv = h ^ ze[16u * p + 4u * rp + kp];
              const uint32_t ca = pack_a(K_SUB2, p, vp);
              const uint32_t dk_pv = cp ? (dk_rp ^ class_terms(p, vp, crow_unused)) : 0u;
              /* second position in blocks of 8 aligned positions = two residue dwords
                 per lane, loaded together (one wait per 24 probes, not one per position) */
              for (uint32_t q0 = (p + 1) & ~7u; q0 < L; q0 += 8) {
                uint32_t mask = 0;
                const uint32_t cbits = class_bits8(q0);
                const uint32_t w0 = qr[(q0 >> 2) * WAVE];
                const uint32_t w1 = q0 + 4 < L ? qr[((q0 >> 2) + 1) * WAVE] : 0u;
#pragma unroll
                for (uint32_t jj = 0; jj < 8; jj++) {
                  const uint32_t qq = q0 + jj;
                  if (qq <= p || qq >= L)
                    continue;                              /* wave-uniform */
                  const uint32_t rq = ((jj < 4 ? w0 : w1) >> ((jj & 3u) * 8)) & 0xffu;
                  const bool cq = ((cbits >> jj) & 1u) != 0;
                  uint32_t b3 = 0;
                  if (cq && all_items)
                    continue;                              /* wave-uniform */
                  if (fast && !cq) {
                    const uint32_t eaddr = ze_addr + (16u * qq + 4u * rq) * 8u;
                    if (!CMPR_DBG(P, DBG_SKIP_LDS_ROWS))
                      b3 = probe3_lds(W, hpv, eaddr);
                  } else if (!cp && !cq) {
                    const uint64_t *zq = zl + 4 * qq;
                    const uint64_t hq = hpv ^ zq[rq];
#pragma unroll
                    for (uint32_t k = 1; k <= 3; k++)
                      b3 |= probe_one_own(W, hq ^ zq[(rq + k) & 3u], false) ? (1u << (k - 1)) : 0u;
                  } else if (!CMPR_DBG(P, DBG_SKIP_HBM_ROWS)) {
                    const uint64_t *zq = zl + 4 * qq;
                    const uint64_t hq = hpv ^ zq[rq];
                    const uint32_t dk_rq = class_terms(qq, rq, crow_unused);
                    uint64_t word[3];
#pragma unroll
                    for (uint32_t k = 1; k <= 3; k++) {
                      const uint32_t x = (rq + k) & 3u;
                      word[k - 1] = *hbm_word(W, hq ^ zq[x],
                                              dk_pv ^ dk_rq ^ class_terms(qq, x, crow_unused));
                    }
#pragma unroll
                    for (uint32_t k = 1; k <= 3; k++)
                      b3 |= bloom_hit(word[k - 1], pattern_fields(hq ^ zq[(rq + k) & 3u]))
                                ? (1u << (k - 1)) : 0u;
                  }
                  if ((cp || cq) && lane_items)
                    b3 = 0;                                /* this lane's pair is an item */
                  mask |= (qq < Ll ? b3 : 0u) << (3u * jj);       /* qq < Ll implies p < Ll */
                }
                if (!valid || CMPR_DBG(P, DBG_SKIP_EMIT))
                  mask = 0;
                while (__ballot(mask != 0)) {
                  const bool pos = mask != 0;
                  const uint32_t b = pos ? (uint32_t)__ffs((int)mask) - 1u : 0u;
                  const uint32_t qq = q0 + b / 3u;
                  const uint32_t rq = res_at(qq < L ? qq : 0);
                  const uint32_t k = b % 3u + 1u;
                  const uint64_t hv = hpv ^ ze[16u * qq + 4u * rq + k];
                  s_push<GENES>(W, pos, hv, ca, qq | (((rq + k) & 3u) << 24));
                  mask &= mask - 1u;
                }
              }
            }
          }
        } else
        for (uint32_t p = 0; p + 1 < L; p++) {
          const uint32_t rp = res_at(p);
          const uint64_t hp = h ^ zl[ZS * p + rp];
          const uint64_t zrow_p = zl[ZS * p + zlane];
          uint32_t crow_p;
          const uint32_t dk_p = class_terms(p, rp, crow_p);
          const bool cp = is_class_pos(p);
#pragma unroll 1
          for (uint32_t v = 0; v < (uint32_t)A; v++) {
            const uint32_t pv = (valid && v != rp) ? ~0u : 0u;
            const uint64_t hpv = hp ^ readlane64(zrow_p, v);
            const uint32_t ca = pack_a(K_SUB2, p, v);
            const uint32_t dk_pv = dk_p ^ (uint32_t)__builtin_amdgcn_readlane((int)crow_p, (int)v);
            uint32_t w = 0;
            for (uint32_t qq = p + 1; qq < L; qq++) {
              if ((qq & 3u) == 0 || qq == p + 1)
                w = qr[(qq >> 2) * WAVE];
              const uint32_t rq = (w >> ((qq & 3u) * 8)) & 0xffu;
              const uint64_t hq = hpv ^ zl[ZS * qq + rq];
              const uint64_t zrow_q = zl[ZS * qq + zlane];
              uint32_t mask;
              if (staged && !cp && !is_class_pos(qq)) {
                const uint32_t zaddr = zl_addr + (ZS * qq + rq) * 8u;
                mask = row_lds_others<A>(W, hq, zaddr) & pv & (qq < Ll ? ~0u : 0u);
                emit_row_others<GENES, A, false>(W, mask, hq, zaddr, rq, ca, qq);
                continue;
              }
              if (!cp && !is_class_pos(qq)) {
                mask = row_hbm<A>(W, hq, zrow_q, 0u, 0u);
              } else {
                uint32_t crow_q;
                const uint32_t dk_q = class_terms(qq, rq, crow_q);
                mask = row_hbm<A>(W, hq, zrow_q, dk_pv ^ dk_q, crow_q);
              }
              mask &= pv & (qq < Ll ? ~0u : 0u) & ~(1u << rq);
              emit_row<GENES, false>(W, mask, hq, zl + ZS * qq, ca, qq);
            }
          }
        }
      }

      if constexpr (D >= 2 && A == 4) {
        if (item_tile) {
          /* ---- items (see item_tile above): position p = c0 + i of every lane takes
                  residue + kp; the second position runs over ALL the others.  A second
                  position that is no class position leaves the variant in the staged
                  slice: the same 9-probe blocks as the main pass.  Two class positions:
                  handled once, from the lower one, where the filter lies. ---- */
          /* which class position, which residue step: the same for the 64 items of a
             block, and written in each of them (residue step | position << 8) */
          const uint64_t have = __ballot(valid);
          const uint32_t crp1 = have ? (uint32_t)__builtin_amdgcn_readlane(
                                           (int)icrp, (int)__builtin_ctzll(have)) : 0u;
          const uint32_t kp = crp1 & 0xffu;
          const uint32_t p = (crp1 >> 8) & 0xffffu;
          /* residues of 16 positions from x0 (a multiple of 16, wave-uniform) */
          auto pk16 = [&](uint32_t x0) -> uint32_t {
            static_assert(RESPACK_MAX == 96, "six words");
            return x0 < 16u ? ipk.w[0] : x0 < 32u ? ipk.w[1] : x0 < 48u ? ipk.w[2]
                 : x0 < 64u ? ipk.w[3] : x0 < 80u ? ipk.w[4] : ipk.w[5];
          };
          const uint32_t rp = (pk16(p & ~15u) >> ((p & 15u) * 2u)) & 3u;
          const uint32_t vp = (rp + kp) & 3u;
          const uint64_t hpv = h ^ ze[16u * p + 4u * rp + kp];
          uint32_t crow_unused;
          for (uint32_t q0 = 0; q0 < (have ? L : 0u); q0 += 8) {
            uint32_t mask = 0;
            const uint32_t cbits = class_bits8(q0);
            const uint32_t w8 = pk16(q0 & ~15u) >> ((q0 & 8u) * 2u);     /* 8 positions x 2 bits */
#pragma unroll
            for (uint32_t jj = 0; jj < 8; jj++) {
              const uint32_t qq = q0 + jj;
              if (qq == p || qq >= L)
                continue;                                  /* wave-uniform */
              const uint32_t rq = (w8 >> (2u * jj)) & 3u;
              const bool cq = ((cbits >> jj) & 1u) != 0;
              uint32_t b3 = 0;
              if (!cq) {
                const uint32_t eaddr = ze_addr + (16u * qq + 4u * rq) * 8u;
                b3 = probe3_lds(W, hpv, eaddr);
              } else if (qq > p) {
                /* (W.tile_slice is the staged slice = the query's own ^ the change at p) */
                const uint64_t *zq = zl + 4 * qq;
                const uint64_t hq = hpv ^ zq[rq];
                const uint32_t dk_rq = class_terms(qq, rq, crow_unused);
                uint64_t word[3];
#pragma unroll
                for (uint32_t k = 1; k <= 3; k++) {
                  const uint32_t x = (rq + k) & 3u;
                  word[k - 1] = *hbm_word(W, hq ^ zq[x], dk_rq ^ class_terms(qq, x, crow_unused));
                }
#pragma unroll
                for (uint32_t k = 1; k <= 3; k++)
                  b3 |= bloom_hit(word[k - 1], pattern_fields(hq ^ zq[(rq + k) & 3u]))
                            ? (1u << (k - 1)) : 0u;
              }
              mask |= (qq < Ll ? b3 : 0u) << (3u * jj);
            }
            if (!valid || CMPR_DBG(P, DBG_SKIP_EMIT))
              mask = 0;
            while (__ballot(mask != 0)) {
              const bool pos = mask != 0;
              const uint32_t b = pos ? (uint32_t)__ffs((int)mask) - 1u : 0u;
              const uint32_t qq = q0 + b / 3u;
              const uint32_t rq = (w8 >> (2u * (b / 3u))) & 3u;
              const uint32_t k = b % 3u + 1u;
              const uint32_t vq = (rq + k) & 3u;
              const uint64_t hv = hpv ^ ze[16u * qq + 4u * rq + k];
              /* (the lower position first, as everywhere) */
              const bool lower = qq < p;
              s_push<GENES>(W, pos, hv, lower ? pack_a(K_SUB2, qq, vq) : pack_a(K_SUB2, p, vp),
                            lower ? (p | (vp << 24)) : (qq | (vq << 24)));
              mask &= mask - 1u;
            }
          }
        }
      }

      W.st.variants += valid ? nvar : 0ull;
    }
    if (all_done)
      break;
  }

  /* leftovers: fewer than 64 entries */
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  flush_or_resolve<GENES>(W, 0, W.qn, true);

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
    for (uint32_t i = threadIdx.x; i < cells; i += NT) {
      const unsigned long long x = mat_all[i];
      if (x)
        atomicAdd(matrix_dst(P) + i, x);
    }
  }
}

}  // namespace cmpr
#endif
