/*
 * kernels_pairs2.h -- probe kernel for nucleotides, d = 2, on PAIR rows (round 4).
 *
 * Same path as the other probe kernels (variant enumeration -> Zobrist hash -> Bloom test ->
 * hash-table walk -> exact verify -> matrix accumulate; reference: overlap.cc:253-284,
 * variants.cc:357-400 for the double substitutions, bloompat.h:40-58), over the pair-row
 * filter of kernels_rows.h: every set-2 sequence t is entered once per pair of positions
 * (p, p + 1), p even, under its hash with BOTH blanked, the two residues coded as sum and
 * difference (build_rows_kernel, pair_entry_bits).
 *
 * What a word of that filter answers for d = 2:
 *   - both substitutions inside one pair: the word of the query's own pair-blanked hash holds
 *     xs[v + w] & xd[v - w] for every (v, w) -- the 9 doubles, the 6 singles and the identity
 *     of the pair out of ONE read;
 *   - substitutions in two different pairs (j1 < j2): the query with position e of pair j2
 *     replaced (hash h2) is, for the filter, just another sequence: the word of ITS pair j1
 *     (both blanked) answers the 6 single substitutions of pair j1 -- one read per
 *     (j1, e, replacement), 6 variants each.
 * A 45-nucleotide query: 23 + 6 * C(23, 2) = 1541 word reads for its 9046 variants, where
 * the per-variant filter (kernels_sliced.h) probes 8910 times.
 *
 * Work decomposition.  The filter of a 100M-sequence reference set is gigabytes, its slices
 * many and the queries per slice few (one or two tiles), while a tile's work is large
 * (~1500 reads x 64 lanes): so a WORKGROUP, not a wave, takes a tile -- its waves split the
 * tile's pairs j2 ("units", claimed with an LDS atomic), all on the one staged slice -- and
 * the next chunk's slice is copied into the other half of a double buffer (LDS-DMA) while
 * this one is worked on: one workgroup barrier per chunk.
 *
 * Class positions (layout.h): a pair that holds a class position of a split ("heavy") query
 * is filed in that pair's class part of the filter, under the class key without the pair's
 * terms.  For a double substitution with ONE position in such a class pair c, c takes the role
 * of the blanked pair (the other position is substituted in the hash): all those reads of a
 * query go to one slice of c's class part -- an ITEM (query_layout.hip, one per heavy query
 * and class pair, grouped by that slice, 64 to a block).  Both positions in two different class
 * pairs: the read goes where the filter lies (18 of ~1500 per query).
 */
#ifndef COMPAIRR_AMD_KERNELS_PAIRS2_H
#define COMPAIRR_AMD_KERNELS_PAIRS2_H

#include "kernels_rows.h"

namespace cmpr {

/* queue entries of this kernel (kernels_rows.h q_push: hash, slot, ca, cb, m):
 *   K2_CROSS  hash = h2 (the query with position e replaced by ne)
 *             ca = kind | j1 << 3 | qa << 10 | qb << 13 | e << 16 | ne << 24   (qa, qb: the query's own
 *                  residues of pair j1; qb = 4: the query ends with the pair's first position)
 *             cb = residues v that may stand at 2 j1, m = residues w at 2 j1 + 1
 *   K2_SAME   hash = h (the query); ca = kind | j << 3 | qa << 10 | qb << 13;
 *             cb = bit 5 v + w <-> "v at 2 j, w at 2 j + 1" (w = 4: no second position)
 *   others    a finished variant (pack_a), as it leaves
 */
constexpr uint32_t K2_CROSS = 5, K2_SAME = 6;
constexpr uint32_t P2_NONE = 4;                  /* "no residue": the second position of a pair behind the end */
constexpr uint32_t P2_PZ = 20;                   /* pair-blank keys per pair: (qa, qb) -> qa | qb << 2; 16 + qa: no second position */
__device__ __forceinline__ uint32_t p2_pz_index(uint32_t qa, uint32_t qb)
{
  return qb >= 4u ? 16u + qa : qa | (qb << 2);
}

/* LDS tables: ze[16 p + 4 r + k] = Z[p][r] ^ Z[p][(r + k) & 3] (k = 0: Z[p][r] itself);
   pz[P2_PZ j + (a | b << 2)] = Z[2j][a] ^ Z[2j+1][b], pz[P2_PZ j + 16 + a] = Z[2j][a] alone */
struct P2Tables {
  uint32_t ze_addr, pz_addr;
};

template <bool GENES>
__device__ __forceinline__ void p2_drain_round(SProber &W, const P2Tables &T, int n, bool last = false)
{
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
  const uint32_t kind = ca & 7u;
  uint64_t hv = B;
  uint32_t oca = ca, ocb = cb, ncb = cb, nm = m;
  bool more = false;
  if (kind == K2_CROSS) {
    const uint32_t j1 = (ca >> 3) & 127u, qa = (ca >> 10) & 7u, qb = (ca >> 13) & 7u;
    const uint32_t pe = (ca >> 16) & 255u, ne = (ca >> 24) & 3u;
    const uint32_t second = cb == 0u ? 1u : 0u;
    const uint32_t bitsv = second ? m : cb;
    const uint32_t v = (uint32_t)__ffs((int)bitsv) - 1u;
    const uint32_t p = 2u * j1 + second, r = second ? qb : qa;
    hv = B ^ lds_u64(T.ze_addr + (16u * p + 4u * r + ((v - r) & 3u)) * 8u);
    /* (position, residue) pairs in increasing position order */
    if (p < pe) {
      oca = pack_a(K_SUB2, p, v);
      ocb = pe | (ne << 24);
    } else {
      oca = pack_a(K_SUB2, pe, ne);
      ocb = p | (v << 24);
    }
    ncb = second ? 0u : cb & (cb - 1u);
    nm = second ? m & (m - 1u) : m;
    more = (ncb | nm) != 0u;
  } else if (kind == K2_SAME) {
    const uint32_t j = (ca >> 3) & 127u, qa = (ca >> 10) & 7u, qb = (ca >> 13) & 7u;
    const uint32_t b = (uint32_t)__ffs((int)cb) - 1u;
    const uint32_t v = b / 5u, w = b - 5u * v;
    const uint32_t p = 2u * j;
    const bool ch1 = v != qa, ch2 = qb != P2_NONE && w != qb;
    if (ch1)
      hv ^= lds_u64(T.ze_addr + (16u * p + 4u * qa + ((v - qa) & 3u)) * 8u);
    if (ch2)
      hv ^= lds_u64(T.ze_addr + (16u * (p + 1u) + 4u * qb + ((w - qb) & 3u)) * 8u);
    if (ch1 && ch2) {
      oca = pack_a(K_SUB2, p, v);
      ocb = (p + 1u) | (w << 24);
    } else if (ch1) {
      oca = pack_a(K_SUB, p, v);
      ocb = 0;
    } else {
      oca = pack_a(K_SUB, p + 1u, w);
      ocb = 0;
    }
    ncb = cb & (cb - 1u);
    more = ncb != 0u;
  }
  if (act) {
    W.q.hash[e] = hv;
    W.q.ca[e] = oca;
    W.q.cb[e] = ocb;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  W.qn = first;
  flush_or_resolve<GENES, true>(W, first, n, last);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
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

template <bool GENES>
__device__ __forceinline__ void p2_push(SProber &W, const P2Tables &T, bool pos, uint64_t B, uint32_t ca,
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
    /* (inlined at every push site; as a function called from there -- its state passed by value --
       the kernel was 13 % slower: round 4) */
    while (W.qn >= WAVE)
      p2_drain_round<GENES>(W, T, WAVE);
  }
}

/* word wq of a packed query (eight pairs) */
__device__ __forceinline__ uint32_t p2_pair_word(const ResPack &pk, uint32_t wq)
{
  uint32_t w = pk.w[0];
  asm volatile("" : "+v"(w));
  w = wq == 1u ? pk.w[1] : w;
  asm volatile("" : "+v"(w));
  w = wq == 2u ? pk.w[2] : w;
  asm volatile("" : "+v"(w));
  w = wq == 3u ? pk.w[3] : w;
  asm volatile("" : "+v"(w));
  w = wq == 4u ? pk.w[4] : w;
  asm volatile("" : "+v"(w));
  w = wq == 5u ? pk.w[5] : w;
  return w;
}

/* residues of pair j of a packed query: qa | qb << 2 (four bits at 4 j) */
__device__ __forceinline__ uint32_t p2_pair_of(const ResPack &pk, uint32_t j)
{
  static_assert(RESPACK_MAX == 96, "six words, eight pairs each");
  /* (a select chain kept opaque, or the compiler turns it into an indexed read of a scratch copy) */
  const uint32_t wq = j >> 3;
  uint32_t w = pk.w[0];
  asm volatile("" : "+v"(w));
  w = wq == 1u ? pk.w[1] : w;
  asm volatile("" : "+v"(w));
  w = wq == 2u ? pk.w[2] : w;
  asm volatile("" : "+v"(w));
  w = wq == 3u ? pk.w[3] : w;
  asm volatile("" : "+v"(w));
  w = wq == 4u ? pk.w[4] : w;
  asm volatile("" : "+v"(w));
  w = wq == 5u ? pk.w[5] : w;
  return (w >> ((j & 7u) * 4u)) & 15u;
}

/*
 * LDS: [1 or 2 slice buffers, rw_words x 32 B each][ze: 16 x zpos][pz: P2_PZ x ceil(zpos / 2)]
 *      [NW WaveQueues][CR tables][2 x {chunk descriptor, unit counter}][2 x chunk_cap tile refs]
 */
struct P2Slot {
  uint32_t slice, first, ntiles, pass;
  uint32_t next_unit, units, pad[2];
};

template <bool GENES, int NW>
__global__ void __launch_bounds__(NW * WAVE, 4)
probe_pairs2_kernel(const ProbeParams P)
{
  constexpr uint32_t A = 4;
  constexpr uint32_t NT = NW * WAVE;
  constexpr uint32_t MCR = kernel_class_res(A, false);
  extern __shared__ __align__(16) unsigned char smem[];
  if ((uint32_t)(uintptr_t)smem != 0u)
    __builtin_trap();                       /* the slices are read at absolute LDS addresses */
  const uint32_t nwords = P.geom.rw_words;
  const uint32_t slice_bytes = nwords * ROW_WORD_BYTES;
  const unsigned char *filter = (const unsigned char *)P.bloom;
  const uint32_t npairs_max = (P.zpos + 1u) / 2u;
  const uint32_t nbuf = __builtin_amdgcn_readfirstlane(P.geom.nbuf == 1u ? 1u : 2u);   /* slice buffers (layout.h SliceGeom::nbuf) */
  P2Tables T;
  T.ze_addr = nbuf * slice_bytes;
  T.pz_addr = T.ze_addr + 16u * P.zpos * 8u;
  uint64_t *ze = (uint64_t *)(smem + T.ze_addr);
  uint64_t *pz = (uint64_t *)(smem + T.pz_addr);
  /* (the positives that do not fit their buffer are resolved here, by the slow inline form:
     its scores go to an LDS copy of the matrix like everyone else's, kernels_sliced.h) */
  unsigned long long *mat_all = (unsigned long long *)(pz + P2_PZ * npairs_max);
  const uint32_t cells = P.lds_matrix ? P.R1 * P.R2 : 0u;
  WaveQueue *queues = (WaveQueue *)(mat_all + cells);
  uint32_t *cr_lds = (uint32_t *)(queues + NW);
  P2Slot *slots = (P2Slot *)(cr_lds + MAX_CLASS_RES * A);
  TileRef *tref_lds = (TileRef *)(slots + 2);
  const uint32_t chunk_cap = P.chunk_cap;

  for (uint32_t i = threadIdx.x; i < 16u * P.zpos; i += NT) {
    const uint32_t pos = i / 16u, r = (i / 4u) & 3u, k = i & 3u;
    const uint64_t own = P.zob[pos * 4u + r];
    ze[i] = k ? own ^ P.zob[pos * 4u + ((r + k) & 3u)] : own;
  }
  for (uint32_t i = threadIdx.x; i < P2_PZ * npairs_max; i += NT) {
    const uint32_t j = i / P2_PZ, ab = i % P2_PZ, a = ab & 3u, b = ab < 16u ? ab >> 2 : P2_NONE;
    uint64_t x = 0;
    if (2u * j < P.zpos)
      x = P.zob[(2u * j) * 4u + a];
    if (2u * j + 1u < P.zpos && b < 4u)
      x ^= P.zob[(2u * j + 1u) * 4u + b];
    pz[i] = x;
  }
  for (uint32_t i = threadIdx.x; i < MAX_CLASS_RES * A; i += NT)
    cr_lds[i] = P.geom.ctab[P.geom.off_cr + i];
  for (uint32_t i = threadIdx.x; i < cells; i += NT)
    mat_all[i] = 0;

  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x / WAVE;
  SProber W{P, (const uint64_t *)smem, queues[wave], P.lds_matrix ? mat_all : nullptr,
            lane, 0u, 0u, 0u, P.geom.smask, 0u, 0, {0ull, 0u, 0u, 0u}, 0ull};
  claim_pos_block(W);
  unsigned long long reads = 0;

  const uint32_t G = gridDim.x;
  const uint32_t my_chunks = blockIdx.x < P.nchunks ? (P.nchunks - blockIdx.x + G - 1u) / G : 0u;

  /* Slice and tile references of chunk number Tn of this workgroup into buffer Tn & 1, with
     LDS-DMA (1 KiB per wave instruction, no registers).  ONE wave issues the copies of a chunk
     (waves take turns): loads retire in order, so whoever issued them waits for them at its next
     own load -- one wave in sixteen, once per chunk -- while the others work on. */
  auto stage = [&](uint32_t Tn) {
    if (wave != (Tn % NW))
      return;
    const Chunk ck = P.chunks[blockIdx.x + Tn * G];
    const uint32_t b = Tn & (nbuf - 1u);
    uint32_t l16 = lane * 16u;
    const unsigned char *src = filter + (size_t)ck.slice * slice_bytes + l16;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(b * slice_bytes);
    for (uint32_t off = 0; off < slice_bytes; off += 1024u)
      if (off + l16 < slice_bytes)
        __builtin_amdgcn_global_load_lds((glob_void_t *)(src + off), (lds_void_t *)(uintptr_t)(dst + off), 16, 0, 0);
    const uint32_t cpass = ck.pass & 0xffu;
    const uint32_t upt = cpass >= 3u ? 1u : (ck.pass >> 16) & 0xffu;    /* units per tile: pairs of the longest */
    const uint32_t tbytes = cpass >= 3u ? 0u : ck.ntiles * (uint32_t)sizeof(TileRef);
    const unsigned char *tsrc = (const unsigned char *)(P.tile_refs + ck.first_tile) + l16;
    const uint32_t tdst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(tref_lds + b * chunk_cap));
    for (uint32_t off = 0; off < tbytes; off += 1024u)
      if (off + l16 < tbytes)
        __builtin_amdgcn_global_load_lds((glob_void_t *)(tsrc + off), (lds_void_t *)(uintptr_t)(tdst + off), 16, 0, 0);
    if (lane == 0) {
      slots[b].slice = ck.slice;
      slots[b].first = ck.first_tile;
      slots[b].ntiles = ck.ntiles;
      slots[b].pass = cpass;
      slots[b].next_unit = 0;
      /* a tile's units: its pairs j2 (the chunk's longest tile's count for all: surplus units return
         at once); an item chunk's: its blocks of 64 items */
      slots[b].units = ck.ntiles * upt;
      slots[b].pad[0] = upt;
    }
  };

  if (my_chunks && nbuf == 2u)
    stage(0);

  /* class positions (wave-uniform per tile) */
  const uint32_t KH = P.geom.k;

  for (uint32_t Tn = 0; Tn < my_chunks; Tn++) {
    if (nbuf == 1u) {
      __syncthreads();                                     /* chunk Tn - 1 is finished: its buffer is free */
      stage(Tn);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       /* this wave's share of the copies has landed */
    __syncthreads();                                       /* ... everyone's; and chunk Tn - 1 is finished */
    if (nbuf == 2u && Tn + 1u < my_chunks)
      stage(Tn + 1u);
    const uint32_t b = Tn & (nbuf - 1u);
    const uint32_t sbase = b * slice_bytes;
    const uint32_t cslice = slots[b].slice, cpass = slots[b].pass, cfirst = slots[b].first;
    const uint32_t units = slots[b].units, upt = slots[b].pad[0];
    const TileRef *trefs = tref_lds + b * chunk_cap;

    auto woff_of = [&](uint64_t Wk) -> uint32_t {
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
    auto word_glob = [&](uint32_t slice, uint32_t wo) -> RowWord {
      const unsigned char *base = filter + (size_t)slice * slice_bytes;
      RowWord w;
      w.a = *(const u32x4 *)(base + wo);
      w.b = *(const u32x4 *)(base + wo + 16u);
      return w;
    };

    for (;;) {
      uint32_t u = 0;
      if (lane == 0)
        u = atomicAdd(&slots[b].next_unit, 1u);
      u = __builtin_amdgcn_readfirstlane(u);
      if (u >= units)
        break;

      /* ---- what the unit is: pair j2 of a tile, or a block of 64 items ---- */
      const bool item_unit = cpass >= 3u;
      uint32_t t = 0, L = 0, nvalid = WAVE, K = 0, j2 = 0;
      if (!item_unit) {
        const uint32_t tk = u / upt;
        j2 = upt - 1u - (u - tk * upt);                    /* the long units first */
        const TileRef *tr = trefs + tk;
        t = __builtin_amdgcn_readfirstlane(tr->t);
        L = __builtin_amdgcn_readfirstlane(tr->td.len);
        nvalid = __builtin_amdgcn_readfirstlane(tr->td.nvalid);
        K = __builtin_amdgcn_readfirstlane(tr->td.k);
        if (2u * j2 >= L)
          continue;                                        /* (no such pair in this tile) */
      }
      /* the lane's query */
      uint64_t h = 0;
      uint32_t Ll = 0, slot = POS_NULL_SLOT, irp = 0;
      uint64_t iw = 0;
      ResPack pk{};
      if (item_unit) {
        const ItemRec it = P.items[cfirst + u * WAVE + lane];
        slot = it.main;
        iw = it.w;
        irp = it.rp;
        if (slot != POS_NULL_SLOT) {
          pk = P.cpk[cfirst + u * WAVE + lane];
          Ll = P.qlen[slot];
        }
      } else {
        if (lane < nvalid) {
          slot = t * WAVE + lane;
          h = P.qgh[slot];
          Ll = P.qlen[slot];
          pk = P.qpk[slot];
        }
      }
      const bool valid = slot != POS_NULL_SLOT;
      W.qslot = slot;
      uint32_t nvar = 0, treads = 0;

      if (!item_unit) {
        /* class pairs of this tile (wave-uniform: a tile of a heavy class has one length, or
           holds every class position unwrapped) */
        uint64_t cls = 0;                                  /* bit j: pair j holds a class position */
        if (K) {
#pragma unroll
          for (uint32_t i = 0; i < MCR; i++)
            if (i < K)
              cls |= 1ull << (class_pos(L, i, P.geom.c0) >> 1);
        }
        /* ---- the sequence itself: the code-A entry under its hash (variants.cc:260-268) ---- */
        if (j2 == 0u) {
          const RowWord w = word_lds(woff_of(h));
          const bool hit = ((row_bits(w, h) >> A) & 1u) != 0;
          p2_push<GENES>(W, T, valid && hit, h, pack_a(K_SAME, 0, 0), 0, 1u, 1u);
          nvar += 1u;
          treads += valid ? 1u : 0u;
        }
        if ((cls >> j2) & 1ull) {                          /* an item does this pair */
          W.st.variants += valid ? (uint64_t)nvar : 0ull;
          reads += treads;
          continue;
        }
        const uint32_t ab2 = p2_pair_of(pk, j2);
        const uint32_t qa2 = ab2 & 3u, p2a = 2u * j2;
        const bool has_a = valid && p2a < Ll, has_b = valid && p2a + 1u < Ll;
        const uint32_t qb2 = has_b ? ab2 >> 2 : P2_NONE;

        /* ---- both positions in pair j2: one word, 15 variants ---- */
        {
          const uint64_t W0 = h ^ lds_u64(T.pz_addr + (P2_PZ * j2 + p2_pz_index(qa2, qb2)) * 8u);
          const RowWord w = word_lds(woff_of(W0));
          uint32_t xs, xd;
          pair_bits(w, W0, xs, xd);
          const uint32_t rev = __builtin_bitreverse32(xd);
          uint32_t mask = 0;
          if (qb2 == P2_NONE) {
            /* the query ends with 2 j2: "v there" <-> entry (v, A): sum v + 4, difference v - 4 */
            const uint32_t a1 = __builtin_amdgcn_alignbit(xs, xs, 4u) & __builtin_amdgcn_alignbit(xd, xd, 28u);
#pragma unroll
            for (uint32_t v = 0; v < 4u; v++)
              mask |= ((a1 >> v) & 1u) << (5u * v + 4u);
            mask &= ~(1u << (5u * qa2 + 4u));
            nvar += has_a ? 3u : 0u;
          } else {
#pragma unroll
            for (uint32_t v = 0; v < 4u; v++) {
              /* bit w of x <-> "v at 2 j2 and w at 2 j2 + 1": xs[v + w] & xd[v - w] */
              const uint32_t x = __builtin_amdgcn_alignbit(xs, xs, v) & __builtin_amdgcn_alignbit(rev, rev, 31u - v);
              mask |= (x & 15u) << (5u * v);
            }
            mask &= ~(1u << (5u * qa2 + qb2));             /* not the pair as it is */
            nvar += has_b ? 15u : 0u;
          }
          mask = has_a ? mask : 0u;
          treads += has_a ? 1u : 0u;
          p2_push<GENES>(W, T, mask != 0u, h, K2_SAME | (j2 << 3) | (qa2 << 10) | (qb2 << 13), mask, 0u,
                         (uint32_t)__popc(mask));
        }

        /* ---- one position in pair j2, the other in a pair j1 < j2 that is no class pair: the
                query with e replaced reads the word of its pair j1, both blanked.  The six
                (e, replacement) hashes are kept; per pair j1 ONE key read, then six independent
                word reads in flight together ---- */
        uint64_t h2[6];
        const uint32_t lmA = has_a ? 15u : 0u, lmB = has_b ? 15u : 0u;   /* the lane has the position replaced */
#pragma unroll
        for (uint32_t c = 0; c < 6u; c++) {
          const uint32_t which = c / 3u, k = c - 3u * which + 1u;
          const uint32_t qe = which ? (ab2 >> 2) : qa2;
          h2[c] = h ^ lds_u64(T.ze_addr + (16u * (p2a + which) + 4u * qe + k) * 8u);
        }
        const uint32_t ncomb = p2a + 1u < L ? 6u : 3u;     /* (wave-uniform: the tile's longest query) */
        uint32_t n1 = 0;                                   /* pairs j1 read (wave-uniform) */
        uint32_t cur = 0;
        for (uint32_t j1 = 0; j1 < j2; j1++) {
          if ((j1 & 7u) == 0u) {
            cur = p2_pair_word(pk, j1 >> 3);
          }
          const uint32_t ab1 = cur & 15u;
          cur >>= 4;
          if ((cls >> j1) & 1ull)
            continue;                                      /* (an item: class pair j1 blanked, e replaced) */
          n1++;
          const uint32_t qa1 = ab1 & 3u, qb1 = ab1 >> 2;   /* (j1 < j2 <= the lane's last pair: both exist when e does) */
          uint32_t zrow = T.pz_addr + P2_PZ * 8u * j1;
          asm("" : "+s"(zrow));
          const uint64_t K1 = lds_u64(zrow + ab1 * 8u);
          const uint32_t own1 = ~(1u << qa1), own2 = ~(1u << qb1);
          /* NR of the six at a time (all six when the pair has both positions): their words in
             flight together, then the tests */
          auto group = [&](auto lo_c, auto nr_c) {
            constexpr uint32_t LO = decltype(lo_c)::value, NR = decltype(nr_c)::value;
            uint64_t Wk[NR];
            RowWord w[NR];
#pragma unroll
            for (uint32_t c = 0; c < NR; c++) {
              Wk[c] = h2[LO + c] ^ K1;
              w[c] = word_lds(woff_of(Wk[c]));
            }
            __builtin_amdgcn_sched_barrier(0);
            uint32_t x1[NR], x2[NR], any = 0;
#pragma unroll
            for (uint32_t c = 0; c < NR; c++) {
              const uint32_t lmask = (LO + c) >= 3u ? lmB : lmA;
              uint32_t xs, xd, a1, a2;
              pair_bits(w[c], Wk[c], xs, xd);
              pair_answers(xs, xd, qa1, qb1, a1, a2);
              x1[c] = a1 & own1 & lmask;
              x2[c] = a2 & own2 & lmask;
              any |= x1[c] | x2[c];
            }
            if (__ballot(any != 0u)) {
#pragma unroll
              for (uint32_t c = 0; c < NR; c++) {
                const uint32_t which = (LO + c) / 3u, k = (LO + c) - 3u * which + 1u;
                const uint32_t qe = which ? (ab2 >> 2) : qa2;
                p2_push<GENES>(W, T, (x1[c] | x2[c]) != 0u, h2[LO + c],
                               K2_CROSS | (j1 << 3) | (qa1 << 10) | (qb1 << 13) | ((p2a + which) << 16) |
                                   (((qe + k) & 3u) << 24),
                               x1[c], x2[c], (uint32_t)__popc(x1[c]) + (uint32_t)__popc(x2[c]));
              }
            }
          };
          /* (six in flight were tried: the registers they take are spilled elsewhere -- slower) */
          group(std::integral_constant<uint32_t, 0>{}, std::integral_constant<uint32_t, 3>{});
          if (ncomb == 6u)
            group(std::integral_constant<uint32_t, 3>{}, std::integral_constant<uint32_t, 3>{});
        }
        {
          const uint32_t npos2 = (has_a ? 1u : 0u) + (has_b ? 1u : 0u);
          treads += n1 * 3u * npos2;
          nvar += n1 * 18u * npos2;
        }
      } else {
        /* ---- an item: class pair jc of a heavy query, both positions blanked (iw = that hash);
                staged is the slice of jc's class part the query's rows of that pair lie in ---- */
        const uint32_t qa = irp & 31u, qbr = (irp >> 5) & 31u, pc = (irp >> 10) & 0x3fffu;
        const uint32_t jc = pc >> 1;
        const uint32_t qb = qbr >= A ? P2_NONE : qbr;
        const uint32_t Lq = Ll;
        /* the query's hash: the item carries the pair-blanked one */
        h = iw ^ lds_u64(T.pz_addr + (P2_PZ * jc + p2_pz_index(qa & 3u, qb)) * 8u);
        const bool has_b = valid && pc + 1u < Lq;
        /* ---- both positions (or the one) in the class pair ---- */
        {
          const RowWord w = word_lds(woff_of(iw));
          uint32_t xs, xd;
          pair_bits(w, iw, xs, xd);
          const uint32_t rev = __builtin_bitreverse32(xd);
          uint32_t mask = 0;
          if (qb == P2_NONE) {
            const uint32_t a1 = __builtin_amdgcn_alignbit(xs, xs, 4u) & __builtin_amdgcn_alignbit(xd, xd, 28u);
#pragma unroll
            for (uint32_t v = 0; v < 4u; v++)
              mask |= ((a1 >> v) & 1u) << (5u * v + 4u);
            mask &= ~(1u << (5u * qa + 4u));
            nvar += valid ? 3u : 0u;
          } else {
#pragma unroll
            for (uint32_t v = 0; v < 4u; v++) {
              const uint32_t x = __builtin_amdgcn_alignbit(xs, xs, v) & __builtin_amdgcn_alignbit(rev, rev, 31u - v);
              mask |= (x & 15u) << (5u * v);
            }
            mask &= ~(1u << (5u * qa + qb));
            nvar += valid ? 15u : 0u;
          }
          mask = valid ? mask : 0u;
          treads += valid ? 1u : 0u;
          p2_push<GENES>(W, T, mask != 0u, h, K2_SAME | (jc << 3) | (qa << 10) | (qb << 13), mask, 0u,
                         (uint32_t)__popc(mask));
        }
        /* The positions that key the class part of this lane's query (per lane: the items of a
           block come from queries of any length): its class positions and, when the parts are
           keyed by them (layout.h pair_part_terms), the PAIR_EXTRA positions behind. */
        const uint32_t KX = KH + PAIR_EXTRA <= MAX_CLASS_RES ? KH + PAIR_EXTRA : KH;
        uint32_t cpos[MCR];
        uint64_t cls = 0;                                  /* pairs that hold a class position */
#pragma unroll
        for (uint32_t i = 0; i < MCR; i++) {
          cpos[i] = 0xffffffffu;
          if (i < KX && valid) {
            cpos[i] = class_pos(Lq, i, P.geom.c0);
            if (i < KH)
              cls |= 1ull << (cpos[i] >> 1);
          }
        }
        /* longest query of the block (wave-uniform loop bound) */
        uint32_t Lmax = valid ? Lq : 0u;
        for (int off = 32; off > 0; off >>= 1)
          Lmax = max(Lmax, (uint32_t)__shfl_xor((int)Lmax, off, WAVE));
        Lmax = __builtin_amdgcn_readfirstlane(Lmax);
        int ci_c = -1;                                      /* the class part the pair's rows lie in */
#pragma unroll
        for (uint32_t i = 0; i < MCR; i++)
          if (i < KH && (cpos[i] >> 1) == jc && ci_c < 0)
            ci_c = (int)i;
        /* ---- the other position e anywhere outside the pair: class pair blanked, e replaced.
                q[e] keys nothing of the part (no class position, none of the extra ones): the
                staged slice -- three reads per e, in flight together.  Else the replacement
                changes the part's key: read where the filter lies; and when e lies in another
                class pair, only from the LOWER of the two (each pair of positions once).
                Fast path: every query of the block holds the keyed positions unwrapped (they
                are then the same positions for all, and jc is) -- the tests are scalar. ---- */
        const bool uniform = __ballot(valid && Lq < P.geom.c0 + KX) == 0ull;
        const uint32_t jcu = __builtin_amdgcn_readfirstlane(__ballot(valid) ? (uint32_t)__builtin_amdgcn_readlane(
                                 (int)jc, (int)__builtin_ctzll(__ballot(valid) | (1ull << 63))) : 0u);
        const uint32_t own1 = ~(1u << qa), own2 = qb == P2_NONE ? 0u : ~(1u << qb);
        const uint32_t rb_code = qb == P2_NONE ? A : qb;
        uint32_t nhere = 0;                                 /* positions answered from the staged slice (this lane) */
        uint32_t cur = 0;
        for (uint32_t e = 0; e < Lmax; e++) {
          if ((e & 15u) == 0u)
            cur = p2_pair_word(pk, e >> 4);                /* (sixteen positions, two bits each) */
          const uint32_t qe = cur & 3u;
          cur >>= 2;
          const uint32_t je = e >> 1;
          bool keyed, e_cls;
          if (uniform) {
            if (je == jcu)
              continue;
            bool kk = false, cc = false;                   /* scalar */
#pragma unroll
            for (uint32_t i = 0; i < MCR; i++) {
              kk = kk || (i < KX && P.geom.c0 + i == e);
              cc = cc || (i < KH && ((P.geom.c0 + i) >> 1) == je);
            }
            keyed = kk;
            e_cls = cc;
          } else {
            keyed = false;
#pragma unroll
            for (uint32_t i = 0; i < MCR; i++)
              keyed = keyed || (i < KX && cpos[i] == e);
            e_cls = ((cls >> je) & 1ull) != 0;
          }
          const bool live = valid && e < Lq && je != jc;
          const bool here = live && !keyed && !e_cls;
          const bool far = live && (keyed || e_cls) && (!e_cls || je > jc);
          const uint32_t lmask = (here || far) ? 15u : 0u;
          const bool any_far = __ballot(far) != 0ull;
          if (!any_far && __ballot(here) == 0ull)
            continue;
          uint32_t key_e = 0;                               /* the part's key terms of q[e] */
          if (any_far) {
#pragma unroll
            for (uint32_t i = 0; i < MCR; i++)
              if (i < KX && cpos[i] == e)
                key_e ^= cr_lds[i * A + qe];
          }
          uint64_t Wk[3];
          RowWord w[3];
          uint32_t wo[3];
#pragma unroll
          for (uint32_t c = 0; c < 3u; c++) {
            Wk[c] = iw ^ lds_u64(T.ze_addr + (16u * e + 4u * qe + c + 1u) * 8u);
            wo[c] = woff_of(Wk[c]);
            w[c] = word_lds(wo[c]);
          }
          if (any_far) {
#pragma unroll
            for (uint32_t c = 0; c < 3u; c++)
              if (far) {
                const uint32_t ne = (qe + c + 1u) & 3u;
                uint32_t key_n = 0;
#pragma unroll
                for (uint32_t i = 0; i < MCR; i++)
                  if (i < KX && cpos[i] == e)
                    key_n ^= cr_lds[i * A + ne];
                /* the part's slice of this lane's pair: the staged one with e's term exchanged */
                const uint32_t kslice = cslice - (P.geom.smask + 1u + (uint32_t)ci_c * (P.geom.cmask + 1u));
                w[c] = word_glob(row_slice(P.geom, kslice ^ key_e ^ key_n, ci_c), wo[c]);
              }
          }
          __builtin_amdgcn_sched_barrier(0);
          uint32_t x1[3], x2[3], any = 0;
#pragma unroll
          for (uint32_t c = 0; c < 3u; c++) {
            uint32_t xs, xd, a1, a2;
            pair_bits(w[c], Wk[c], xs, xd);
            pair_answers(xs, xd, qa, rb_code, a1, a2);
            x1[c] = a1 & 15u & own1 & lmask;
            x2[c] = a2 & 15u & own2 & lmask;
            any |= x1[c] | x2[c];
          }
          if (__ballot(any != 0u)) {
            const uint64_t he = h ^ iw;                    /* (h2 = the query with e replaced = Wk ^ the pair's keys) */
#pragma unroll
            for (uint32_t c = 0; c < 3u; c++)
              p2_push<GENES>(W, T, (x1[c] | x2[c]) != 0u, Wk[c] ^ he,
                             K2_CROSS | (jc << 3) | (qa << 10) | (qb << 13) | (e << 16) | (((qe + c + 1u) & 3u) << 24),
                             x1[c], x2[c], (uint32_t)__popc(x1[c]) + (uint32_t)__popc(x2[c]));
          }
          nhere += (here || far) ? 1u : 0u;
        }
        treads += 3u * nhere;
        nvar += 3u * nhere * (has_b ? 6u : 3u);
      }
      W.st.variants += valid ? (uint64_t)nvar : 0ull;
      reads += treads;
    }
  }

  /* leftovers: fewer than 64 entries at a time, until every entry is empty (the rounds this takes
     = the most variants any entry still holds; their blocks of the positives buffer are claimed
     with one atomic); then the block claimed ahead goes back as a block of nulls */
  {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint32_t bits = 0;
    if ((int)lane < W.qn) {
      const uint32_t ca = W.q.ca[lane], kind = ca & 7u;
      bits = kind == K2_CROSS ? (uint32_t)__popc(W.q.cb[lane]) + (uint32_t)__popc(W.q.m[lane])
             : kind == K2_SAME ? (uint32_t)__popc(W.q.cb[lane]) : 1u;
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
      p2_drain_round<GENES>(W, T, W.qn < WAVE ? W.qn : WAVE, true);
    }
    if (rounds == 0u) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      flush_or_resolve<GENES, true>(W, 0, 0, true);
    }
  }

  {
    unsigned long long s[STAT_COUNT] = {W.st.variants, W.st.bloom_pos, W.st.hash_eq, W.st.matches, reads};
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
