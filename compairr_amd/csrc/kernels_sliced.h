/*
 * kernels_sliced.h -- probe kernel, variant 1: Bloom slices staged in LDS.
 *
 * Same path as kernels.h (variant enumeration -> Zobrist hash -> Bloom probe
 * -> hash-table walk -> exact verify -> matrix accumulate; reference:
 * overlap.cc:253-284, variants.cc:260-428, bloompat.h:40-58, overlap.cc:168-251),
 * but the Bloom filter is laid out so that the probes of one query land in a
 * 16 KiB slice chosen by the query's class key (layout.h, "Sliced Bloom
 * layout").  A workgroup takes a chunk of tiles that share a slice, copies the
 * slice HBM -> LDS with coalesced 16-byte loads, and its four waves answer
 * every class-preserving probe (for d = 1 substitutions: all but the k class
 * positions, i.e. ~93 % at k = 1) from LDS.  Class-changing variants
 * (substitution at a class position, insertions, deletions) compute their own
 * slice and probe the filter in HBM, exactly as variant 0 does.
 *
 * The replacement-residue keys of a position are fetched once per position
 * into one VGPR pair (lane r holds the key of residue r) and broadcast with
 * v_readlane, so the inner loop has no memory access besides the two LDS
 * reads of the probe itself (filter word, bit pattern).
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
  const uint64_t     *pat_lds;
  const uint64_t     *slice_lds;
  const uint32_t     *cr_lds;      /* CR[MAX_CLASS_RES][A]                    */
  WaveQueue          &q;
  unsigned long long *mat_lds;
  uint32_t            lane;
  uint32_t            qslot;
  uint32_t            wmask_bytes; /* (slice_words - 1) << 3                  */
  uint32_t            tile_slice;
  int                 qn;
  LaneStats           st;
};

template <bool GENES>
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
      resolve_entry<GENES>(W.P, W.q, W.qn + (int)W.lane, W.mat_lds, W.st);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
}

/* class-preserving variant: filter word from the LDS copy of the slice */
template <bool GENES>
__device__ __forceinline__ void probe_lds(SProber &W, uint64_t hv, bool live,
                                          uint32_t ca, uint32_t cb)
{
  const uint32_t woff = ((uint32_t)(hv >> (PATTERN_BITS - 3))) & W.wmask_bytes;
  const uint64_t word = *(const uint64_t *)((const char *)W.slice_lds + woff);
  const uint64_t pat = W.pat_lds[(uint32_t)hv & (PATTERN_COUNT - 1)];
  W.st.variants += live ? 1ull : 0ull;
  s_push<GENES>(W, live && ((word & pat) == 0), hv, ca, cb);
}

/* class-changing variant: `dk` = XOR of the class-key terms that differ */
template <bool GENES>
__device__ __forceinline__ void probe_hbm(SProber &W, uint64_t hv, uint32_t dk,
                                          bool live, uint32_t ca, uint32_t cb)
{
  const uint32_t vslice = (W.tile_slice ^ dk) & W.P.geom.smask;
  const uint32_t woff = ((uint32_t)(hv >> (PATTERN_BITS - 3))) & W.wmask_bytes;
  const uint64_t off = ((uint64_t)vslice << (W.P.geom.words_log2 + 3)) + woff;
  const uint64_t word = *(const uint64_t *)((const char *)W.P.bloom + off);
  const uint64_t pat = W.pat_lds[(uint32_t)hv & (PATTERN_COUNT - 1)];
  W.st.variants += live ? 1ull : 0ull;
  s_push<GENES>(W, live && ((word & pat) == 0), hv, ca, cb);
}

/*
 * LDS: [A * zpos Zobrist keys][1024 patterns][R1 * R2 matrix (optional)]
 *      [4 WaveQueues][2^w-word Bloom slice][CR tables][chunk broadcast]
 */
template <int A, int D, bool INDELS, bool GENES>
__global__ void __launch_bounds__(BLOCK_THREADS)
probe_sliced_kernel(const ProbeParams P)
{
  extern __shared__ __align__(16) unsigned char smem[];
  uint64_t *zl = (uint64_t *)smem;
  const uint32_t nz = (uint32_t)A * P.zpos;
  uint64_t *pat_lds = zl + nz;
  unsigned long long *mat_all = (unsigned long long *)(pat_lds + PATTERN_COUNT);
  const uint32_t cells = P.R1 * P.R2;
  WaveQueue *queues = (WaveQueue *)(mat_all + (P.lds_matrix ? cells : 0));
  uint64_t *slice_lds = (uint64_t *)(queues + WAVES_PER_BLOCK);
  const uint32_t slice_words = 1u << P.geom.words_log2;
  uint32_t *cr_lds = (uint32_t *)(slice_lds + slice_words);
  uint32_t *bcast = cr_lds + MAX_CLASS_RES * A;

  for (uint32_t i = threadIdx.x; i < nz; i += BLOCK_THREADS)
    zl[i] = P.zob[i];
  for (uint32_t i = threadIdx.x; i < PATTERN_COUNT; i += BLOCK_THREADS)
    pat_lds[i] = P.patterns[i];
  if (P.lds_matrix)
    for (uint32_t i = threadIdx.x; i < cells; i += BLOCK_THREADS)
      mat_all[i] = 0;
  for (uint32_t i = threadIdx.x; i < MAX_CLASS_RES * (uint32_t)A; i += BLOCK_THREADS)
    cr_lds[i] = P.geom.ctab[P.geom.off_cr + i];

  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x / WAVE;
  const uint32_t K = P.geom.k;
  SProber W{P, pat_lds, slice_lds, cr_lds, queues[wave],
            P.lds_matrix ? mat_all : nullptr, lane, 0u,
            (slice_words - 1u) << 3, 0u, 0, {0ull, 0u, 0u, 0u}};
  const uint64_t *gene_keys = P.zob + nz;
  const uint32_t zlane = lane < (uint32_t)A ? lane : 0u;   /* lane r <-> residue r */

  for (;;) {
    /* ---- next chunk: tiles of one slice; stage that slice into LDS ---- */
    __syncthreads();                       /* everyone is done with the old slice */
    if (threadIdx.x == 0)
      bcast[0] = atomicAdd(P.tile_counter, 1u);
    __syncthreads();
    const uint32_t item = bcast[0];
    if (item >= P.nchunks)
      break;
    const Chunk ck = P.chunks[item];
    {
      const uint64_t *src = P.bloom + ((uint64_t)ck.slice << P.geom.words_log2);
      for (uint32_t i = threadIdx.x; i < slice_words; i += BLOCK_THREADS)
        slice_lds[i] = src[i];
    }
    __syncthreads();
    W.tile_slice = ck.slice;

    for (uint32_t t = ck.first_tile + wave; t < ck.first_tile + ck.ntiles;
         t += WAVES_PER_BLOCK) {
      const TileDesc td = P.tiles[t];
      const uint32_t L = __builtin_amdgcn_readfirstlane(td.len);
      const uint32_t nvalid = __builtin_amdgcn_readfirstlane(td.nvalid);
      const uint32_t *qr = P.qres + td.res_base + lane;
      const bool valid = lane < nvalid;
      W.qslot = t * WAVE + lane;
      auto res_at = [&](uint32_t p) -> uint32_t {
        return (qr[(p >> 2) * WAVE] >> ((p & 3u) * 8)) & 0xffu;
      };

      /* ---- query hash (zobrist.cc:74-88) and, with -i, the two shifted
              hashes of the rolling indel enumeration (:90-104, :122-136) ---- */
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

      /* class positions of this length (wave-uniform) */
      uint32_t m[MAX_CLASS_RES];
#pragma unroll
      for (uint32_t i = 0; i < MAX_CLASS_RES; i++)
        m[i] = class_pos(L, i);
      auto is_class_pos = [&](uint32_t p) -> bool {
        bool c = false;
#pragma unroll
        for (uint32_t i = 0; i < MAX_CLASS_RES; i++)
          c = c || (i < K && m[i] == p);
        return c;
      };
      /* class-key change of replacing residue r by v at position p */
      auto sub_delta = [&](uint32_t p, uint32_t r, uint32_t v) -> uint32_t {
        uint32_t dk = 0;
#pragma unroll
        for (uint32_t i = 0; i < MAX_CLASS_RES; i++)
          if (i < K && m[i] == p)
            dk ^= cr_lds[i * A + r] ^ cr_lds[i * A + v];
        return dk;
      };

      /* ---- the unchanged sequence ---- */
      probe_lds<GENES>(W, h, valid, pack_a(K_SAME, 0, 0), 0);

      if (D >= 1) {
        /* ---- single substitutions (variants.cc:280-293) ---- */
        uint32_t w = 0;
        for (uint32_t p = 0; p < L; p++) {
          if ((p & 3u) == 0)
            w = qr[(p >> 2) * WAVE];
          const uint32_t r = (w >> ((p & 3u) * 8)) & 0xffu;
          const uint64_t h1 = h ^ zl[A * p + r];
          const uint64_t zrow = zl[A * p + zlane];
          if (!is_class_pos(p)) {
#pragma unroll 2
            for (uint32_t v = 0; v < (uint32_t)A; v++)
              probe_lds<GENES>(W, h1 ^ readlane64(zrow, v), valid && v != r,
                               pack_a(K_SUB, p, v), 0);
          } else {
#pragma unroll 1
            for (uint32_t v = 0; v < (uint32_t)A; v++)
              probe_hbm<GENES>(W, h1 ^ readlane64(zrow, v), sub_delta(p, r, v),
                               valid && v != r, pack_a(K_SUB, p, v), 0);
          }
        }
      }

      if (INDELS) {
        /* Indel variants change the length, hence the class: their slice is
           tile_slice ^ (CL[L] ^ CL[L'] ^ old class residues ^ new class residues). */
        const uint32_t cl_L = P.geom.ctab[L];
        uint32_t cbase = 0;                      /* XOR_i CR[i][s[m_i(L)]] */
#pragma unroll
        for (uint32_t i = 0; i < MAX_CLASS_RES; i++)
          if (i < K && L > 0)
            cbase ^= cr_lds[i * A + res_at(m[i])];

        /* ---- deletions (variants.cc:301-325): u = s without position p,
                u[x] = x < p ? s[x] : s[x + 1], length L - 1 ---- */
        if (L > 1) {
          const uint32_t dl = cl_L ^ P.geom.ctab[L - 1] ^ cbase;
          uint32_t md[MAX_CLASS_RES], lo[MAX_CLASS_RES], hi[MAX_CLASS_RES];
#pragma unroll
          for (uint32_t i = 0; i < MAX_CLASS_RES; i++) {
            md[i] = class_pos(L - 1, i);
            lo[i] = hi[i] = 0;
            if (i < K) {
              lo[i] = cr_lds[i * A + res_at(md[i])];
              hi[i] = cr_lds[i * A + res_at(md[i] + 1)];
            }
          }
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
            uint32_t dk = dl;
#pragma unroll
            for (uint32_t i = 0; i < MAX_CLASS_RES; i++)
              dk ^= md[i] < p ? lo[i] : hi[i];
            probe_hbm<GENES>(W, hd, dk, valid && fresh, pack_a(K_DEL, p, 0), 0);
            gone = r;
          }
        }
        /* ---- insertions (variants.cc:329-353): u = s with v put in front of
                position ip, u[x] = x < ip ? s[x] : x == ip ? v : s[x - 1] ---- */
        {
          const uint32_t dl = cl_L ^ P.geom.ctab[L + 1] ^ cbase;
          uint32_t mi[MAX_CLASS_RES], lo[MAX_CLASS_RES], hi[MAX_CLASS_RES];
#pragma unroll
          for (uint32_t i = 0; i < MAX_CLASS_RES; i++) {
            mi[i] = class_pos(L + 1, i);
            lo[i] = hi[i] = 0;
            if (i < K) {
              if (mi[i] < L)
                lo[i] = cr_lds[i * A + res_at(mi[i])];
              if (mi[i] >= 1)
                hi[i] = cr_lds[i * A + res_at(mi[i] - 1)];
            }
          }
          uint64_t hi_hash = hins;
          uint32_t w = 0, r = 0xffu;
          for (uint32_t ip = 0; ip <= L; ip++) {
            if (ip > 0) {
              const uint32_t p = ip - 1;
              if ((p & 3u) == 0)
                w = qr[(p >> 2) * WAVE];
              r = (w >> ((p & 3u) * 8)) & 0xffu;
              hi_hash ^= zl[A * p + r] ^ zl[A * ip + r];
            }
            uint32_t dk0 = dl;
#pragma unroll
            for (uint32_t i = 0; i < MAX_CLASS_RES; i++)
              if (i < K && mi[i] != ip)
                dk0 ^= mi[i] < ip ? lo[i] : hi[i];
            const uint64_t zrow = zl[A * ip + zlane];
#pragma unroll 1
            for (uint32_t v = 0; v < (uint32_t)A; v++) {
              uint32_t dk = dk0;
#pragma unroll
              for (uint32_t i = 0; i < MAX_CLASS_RES; i++)
                if (i < K && mi[i] == ip)
                  dk ^= cr_lds[i * A + v];
              probe_hbm<GENES>(W, hi_hash ^ readlane64(zrow, v), dk, valid && v != r,
                               pack_a(K_INS, ip, v), 0);
            }
          }
        }
      }

      if (D >= 2) {
        /* ---- double substitutions p < q (variants.cc:370-399) ---- */
        for (uint32_t p = 0; p + 1 < L; p++) {
          const uint32_t rp = res_at(p);
          const uint64_t hp = h ^ zl[A * p + rp];
          const uint64_t zrow_p = zl[A * p + zlane];
          const bool cp = is_class_pos(p);
          for (uint32_t v = 0; v < (uint32_t)A; v++) {
            const bool pv = valid && v != rp;
            const uint64_t hpv = hp ^ readlane64(zrow_p, v);
            const uint32_t ca = pack_a(K_SUB2, p, v);
            const uint32_t dkp = cp ? sub_delta(p, rp, v) : 0u;
            uint32_t w = 0;
            for (uint32_t qq = p + 1; qq < L; qq++) {
              if ((qq & 3u) == 0 || qq == p + 1)
                w = qr[(qq >> 2) * WAVE];
              const uint32_t rq = (w >> ((qq & 3u) * 8)) & 0xffu;
              const uint64_t hq = hpv ^ zl[A * qq + rq];
              const uint64_t zrow_q = zl[A * qq + zlane];
              if (!cp && !is_class_pos(qq)) {
#pragma unroll 2
                for (uint32_t x = 0; x < (uint32_t)A; x++)
                  probe_lds<GENES>(W, hq ^ readlane64(zrow_q, x), pv && x != rq, ca,
                                   qq | (x << 24));
              } else {
#pragma unroll 1
                for (uint32_t x = 0; x < (uint32_t)A; x++)
                  probe_hbm<GENES>(W, hq ^ readlane64(zrow_q, x),
                                   dkp ^ sub_delta(qq, rq, x), pv && x != rq, ca,
                                   qq | (x << 24));
              }
            }
          }
        }
      }
    }
  }

  /* leftovers: fewer than 64 entries */
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if ((int)lane < W.qn)
    resolve_entry<GENES>(P, W.q, (int)lane, W.mat_lds, W.st);

  {
    unsigned long long s[STAT_COUNT] = {W.st.variants, W.st.bloom_pos,
                                        W.st.hash_eq, W.st.matches};
#pragma unroll
    for (int k = 0; k < STAT_COUNT; k++) {
      unsigned long long x = s[k];
      for (int off = 32; off > 0; off >>= 1)
        x += __shfl_down(x, off, WAVE);
      if (lane == 0 && x)
        atomicAdd(P.stats + k, x);
    }
  }

  if (P.lds_matrix) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cells; i += BLOCK_THREADS) {
      const unsigned long long x = mat_all[i];
      if (x)
        atomicAdd(P.matrix + i, x);
    }
  }
}

}  // namespace cmpr
#endif
