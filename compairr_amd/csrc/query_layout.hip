/*
 * query_layout.hip -- cmpr_set_queries on the device.
 *
 * The reference hands its per-query loop the parsed set as it lies in memory
 * (overlap.cc:906-938, db accessors db.cc:964-997).  The kernels here want the
 * queries grouped by filter slice and cut into 64-query tiles (layout.h); this
 * file does that regrouping with kernels, so that from cmpr_set_view to "resident
 * in HBM" the host only copies the caller's arrays:
 *
 *   validate_*_kernel   what scan_view checked on the host: monotone offsets,
 *                       ranges of residue / gene / repertoire numbers, counts >= 1;
 *                       longest sequence; per-repertoire count totals
 *   keys_kernel         per query: class key -> slice -> group (slice, heavy,
 *                       length) of every pass, Zobrist hash (db_hash,
 *                       db.cc:903-916, variant 2), group histogram (atomics)
 *   slices_kernel<0>    per slice: tiles / chunks / residue words it needs
 *   (scan)              hipCUB exclusive scan over the slices
 *   slices_kernel<1>    per slice: tile descriptors, first slot of every group,
 *                       chunk descriptors
 *   place_kernel        per query: claims a slot of its group (atomic) and writes
 *                       its fields and residues there (position-major tiles)
 *   place_class_kernel  variant 2: the compact class-row passes (kernels_rows.h)
 *   sibling_*_kernel    -i: tiles regrouped by the slice their insertion /
 *                       deletion variants fall into
 *   chunk order         heaviest chunks first (hipCUB radix sort)
 *
 * Inside a group the queries land in the order their atomics complete, not in
 * input order: the matrix is a sum of exact integers, the pairs list is
 * unordered (README.md:163), so no result depends on it.
 */
#include "context.h"
#include "kernels_sliced.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstring>

using namespace cmpr;

namespace {

constexpr uint32_t MAXP = 1 + MAX_CLASS_RES;     /* passes with a layout of their own */

enum : uint32_t { VERR_OFFSETS = 1, VERR_LONG = 2, VERR_REP = 3, VERR_GENE = 4, VERR_COUNT = 5,
                  VERR_RESIDUE = 6 };

struct SliceTot {
  uint32_t tiles, chunks, small, list;
  unsigned long long res;              /* residue dwords (pass 0) / class slots */
};
struct SliceTotSum {
  __host__ __device__ SliceTot operator()(const SliceTot &a, const SliceTot &b) const
  {
    SliceTot r;
    r.tiles = a.tiles + b.tiles;
    r.chunks = a.chunks + b.chunks;
    r.small = a.small + b.small;
    r.list = a.list + b.list;
    r.res = a.res + b.res;
    return r;
  }
};

/* everything the layout kernels need (passed by value) */
constexpr uint32_t MAX_GROUPS = 32;          /* item groups: >= MAX_CLASS_RES + 1, >= 8 class positions x 3 */
static_assert(MAX_GROUPS >= MAX_CLASS_RES + 1 && MAX_GROUPS >= 8 * 3, "item groups");

struct QL {
  /* the caller's set, uploaded as it is */
  const uint8_t  *res;
  const uint64_t *off;
  const uint32_t *v, *j, *rep;
  const uint64_t *cnt;
  uint64_t        n;
  uint32_t        n_rep, n_v_max, n_j_max;
  uint32_t        A, zpos, n_v, longest, per_slice;
  uint32_t        genes, counts, existence, indels, differences, sliced, rows;
  const uint64_t *zob;
  SliceGeom       geom;
  uint32_t        npass;             /* 1 + class-row passes (variant 2) */
  uint32_t        min_mixed;         /* lengths >= this share tiles; ~0: none do */
  uint32_t        chunk_tiles, small_max, class_unstaged;
  uint64_t        nbuckets;          /* (slice, heavy) buckets */
  uint64_t        nslices;
  /* per pass */
  uint32_t *cnt_g[MAXP], *base_g[MAXP], *fill_g[MAXP], *grp[MAXP];
  SliceTot *tot[MAXP], *pre[MAXP];
  uint32_t  tile0[MAXP], chunk0[MAXP], list0[MAXP], small0[MAXP];
  unsigned long long res0[MAXP];
  uint32_t *tfirst_g;                /* pass 0: first tile of every group */
  /* per query */
  uint64_t *h_tmp, *hins_tmp, *hdel_tmp;
  uint32_t *ck_tmp, *slot_of;
  /* outputs */
  TileDesc *tiles;
  Chunk    *chunks;
  TileRef  *tile_refs;
  uint32_t *small_tiles, *chunk_work;
  uint32_t *qres, *qv, *qj, *qrep, *qorig, *qck;
  uint64_t *qgh, *qhins, *qhdel, *qcnt;
  uint16_t *qlen;
  QueryRec *qrec;
  /* variant 2: rows that cannot be answered from the slice staged for their tile
     become flat ITEMS, grouped by the slice they are filed under and padded to
     whole blocks of 64 per slice (kernels_rows.h, passes >= 3).  Item group g < K:
     class part g of the filter -- the substitution row of class position g of every
     split query and, with -i, its insertion row blanked at class position g of
     the variant.  Group K (with -i): deletion variants whose slice of the main
     part is not the one staged for their tile's deletion pass. */
  uint32_t  ngroups;
  uint32_t  goff[MAX_GROUPS];             /* first counter of the group */
  uint32_t  gslices[MAX_GROUPS];          /* slices of the group */
  uint32_t  gslice0[MAX_GROUPS];          /* its first slice in the filter */
  /* variant 1, nucleotides, d = 2: group (i, k) = the double substitutions whose one
     position is class position i with its residue advanced by k (1 .. 3): that
     variant lives in another slice, the same for every second position that is no
     class position -- one item per (split query, i, k), grouped by that slice
     (kernels_sliced.h, passes >= 3) */
  uint32_t  sub2_items;
  uint2    *slice_items;                  /* sub2: per slice (first item, blocks of 64) when the slice's first
                                             main chunk takes them along (pass | CHUNK_WITH_ITEMS) */
  uint32_t  nitem_slices;                 /* counters in all */
  uint32_t  cblocks;                      /* blocks of 64 items per chunk at most */
  uint32_t *ccnt, *cbase, *cfill, *cnch, *cchpre;   /* [nitem_slices] */
  uint32_t  cchunk0;                      /* first item chunk in the chunk list */
  uint64_t *cw;
  uint32_t *cmain;
  uint32_t *crp;
  cmpr::ResPack *cpk;
  /* validation / statistics */
  uint32_t           *verr;          /* [0] first error kind, [1] longest */
  double             *rep_total;
  unsigned long long *alg_bytes;
};

/* ---- validation ---------------------------------------------------------- */

__global__ void __launch_bounds__(256)
validate_seq_kernel(const QL Q)
{
  extern __shared__ double tot_lds[];          /* n_rep doubles when they fit */
  const bool lds_tot = Q.n_rep <= 2048;
  if (lds_tot) {
    for (uint32_t r = threadIdx.x; r < Q.n_rep; r += 256)
      tot_lds[r] = 0.0;
    __syncthreads();
  }
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t err = 0, L = 0;
  if (i < Q.n) {
    const uint64_t a = Q.off[i], b = Q.off[i + 1];
    if (b < a)
      err = VERR_OFFSETS;
    else if (b - a > 0xffffu)
      err = VERR_LONG;
    else
      L = (uint32_t)(b - a);
    const uint32_t rp = Q.rep[i];
    if (!err && rp >= Q.n_rep)
      err = VERR_REP;
    if (!err && Q.genes && (Q.v[i] >= Q.n_v_max || Q.j[i] >= Q.n_j_max))
      err = VERR_GENE;
    if (!err && Q.counts && Q.cnt[i] < 1)
      err = VERR_COUNT;
    if (!err) {
      const double x = Q.counts ? (double)Q.cnt[i] : 1.0;
      if (lds_tot)
        unsafeAtomicAdd(tot_lds + rp, x);
      else
        unsafeAtomicAdd(Q.rep_total + rp, x);
    }
  }
  if (err)
    atomicCAS(Q.verr, 0u, err);
  /* longest: one atomic per wave */
  uint32_t m = L;
  for (int o = 32; o > 0; o >>= 1)
    m = max(m, (uint32_t)__shfl_down((int)m, o, WAVE));
  if ((threadIdx.x & 63) == 0 && m)
    atomicMax(Q.verr + 1, m);
  if (lds_tot) {
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < Q.n_rep; r += 256)
      if (tot_lds[r] != 0.0)
        unsafeAtomicAdd(Q.rep_total + r, tot_lds[r]);
  }
}

/* every residue code < A (16 bytes per thread) */
__global__ void __launch_bounds__(256)
validate_res_kernel(const uint8_t *res, uint64_t total, uint32_t A, uint32_t *verr)
{
  const uint64_t k = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
  if (k >= total)
    return;
  bool bad = false;
  if (k + 16 <= total && ((uintptr_t)(res + k) & 15u) == 0) {
    const uint4 w = *(const uint4 *)(res + k);
    const uint32_t d[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int b = 0; b < 4; b++)
        bad = bad || ((d[q] >> (8 * b)) & 0xffu) >= A;
  } else {
    for (uint64_t x = k; x < total && x < k + 16; x++)
      bad = bad || res[x] >= A;
  }
  if (bad)
    atomicCAS(verr, 0u, (uint32_t)VERR_RESIDUE);
}

/* ---- keys ---------------------------------------------------------------- */

/* number of variants the reference enumerates for one query
   (generate_variants, variants.cc:260-428) */
__device__ inline uint64_t variants_of(const QL &Q, const uint8_t *s, uint32_t L)
{
  const uint64_t A = Q.A;
  uint64_t n = 1;
  if (Q.differences >= 1) {
    n += (A - 1) * L;
    if (Q.indels) {
      if (L > 1) {
        uint64_t runs = 1;
        for (uint32_t p = 1; p < L; p++)
          runs += s[p] != s[p - 1];
        n += runs;
      }
      n += A + (A - 1) * (uint64_t)L;
    }
  }
  if (Q.differences >= 2)
    n += (A - 1) * (A - 1) * (uint64_t)L * (L ? L - 1 : 0) / 2;
  return n;
}

/* item kinds = the variant kinds of layout.h */
enum : uint32_t { ITEM_SUB = K_SUB, ITEM_INS = K_INS, ITEM_DEL = K_DEL, ITEM_SUB2 = K_SUB2 };

__device__ inline uint32_t class_base_of(const QL &Q, uint64_t i)
{
  const uint32_t L = (uint32_t)(Q.off[i + 1] - Q.off[i]);
  return class_base(Q.geom.ctab, Q.geom, Q.genes != 0, L, Q.genes ? Q.v[i] : 0u,
                    Q.genes ? Q.j[i] : 0u);
}

/* The items of query i (see QL): f(counter index, the row's / variant's hash,
   excluded residue | position << 8 | kind << 24).  Called once to count (HASH =
   false: the hash argument is not computed) and once to place, so both see the
   same items. */
template <bool HASH, typename F>
__device__ inline void for_each_item(const QL &Q, uint64_t i, uint32_t ck, bool heavy, F f)
{
  const SliceGeom &g = Q.geom;
  const uint64_t b = Q.off[i];
  const uint32_t L = (uint32_t)(Q.off[i + 1] - b);
  const uint8_t *s = Q.res + b;
  const uint32_t K = g.k, A = Q.A;
  if (L == 0 || K == 0 || Q.differences < 1)
    return;
  if (Q.sub2_items) {
    /* (only queries that hold every class position unwrapped -- position c0 + i is then
       the same for all of them -- and whose residues fit an item (RESPACK_MAX): the
       others keep their probes where the filter lies) */
    if (heavy && L >= g.c0 + K && L <= RESPACK_MAX)
      for (uint32_t ci = 0; ci < K; ci++) {
        const uint32_t pos = g.c0 + ci, r = s[pos];
        for (uint32_t kp = 1; kp < A; kp++) {
          const uint32_t v = (r + kp) % A;
          const uint32_t key = ck ^ g.ctab[g.off_cr + ci * A + r] ^ g.ctab[g.off_cr + ci * A + v];
          /* (counters slice-major: the blocks of all groups of a slice lie side by side
             and make one chunk -- the slice is staged once for them) */
          f((key & g.smask) * Q.ngroups + (ci * (A - 1) + kp - 1), 0ull, kp | (pos << 8) | (ITEM_SUB2 << 24));
        }
      }
    return;
  }
  /* ---- substitution rows at class positions: class part ci, key without the terms
          of that position; a position that carries several class residues is
          handled by the first of them ---- */
  if (heavy) {
    const uint64_t h = HASH ? Q.h_tmp[i] : 0ull;
    for (uint32_t ci = 0; ci < K; ci++) {
      const uint32_t pos = class_pos(L, ci, g.c0);
      bool first = true;
      uint32_t key = ck;
      for (uint32_t k = 0; k < K; k++)
        if (class_pos(L, k, g.c0) == pos) {
          if (k < ci)
            first = false;
          key ^= g.ctab[g.off_cr + k * A + s[pos]];
        }
      if (first)
        f(Q.goff[ci] + (key & g.cmask), HASH ? h ^ Q.zob[A * pos + s[pos]] : 0ull,
          (uint32_t)s[pos] | (pos << 8) | (ITEM_SUB << 24));
    }
  }
  if (!Q.indels)
    return;
  /* (length, V, J) key of the query: its class key without the class residues */
  uint32_t base = ck;
  if (heavy)
    for (uint32_t k = 0; k < K; k++)
      base ^= g.ctab[g.off_cr + k * A + s[class_pos(L, k, g.c0)]];
  const uint32_t base_i = base ^ g.ctab[L] ^ g.ctab[L + 1];          /* insertion variants */
  const bool heavy_i = class_is_heavy(g.ctab, g, base_i);
  const uint32_t dlen = L > 1 ? g.ctab[L] ^ g.ctab[L - 1] : 0u;
  const uint32_t base_d = base ^ dlen;                               /* deletion variants */
  const bool heavy_d = L > 1 && class_is_heavy(g.ctab, g, base_d);
  const uint32_t sibling = (ck ^ dlen) & g.smask;
  /* One walk over the positions.  With P(x) = XOR_{y<x} Z[y][q[y]], P+(x) and P-(x)
     the same over Z[y+1] / Z[y-1], and the query's two shifted hashes from the keys
     kernel:   gap at ip:  P(ip) ^ hins ^ P+(ip);   q without p:  P(p) ^ hdel ^ P-(p+1). */
  const uint64_t hins = HASH ? Q.hins_tmp[i] : 0ull, hdel = HASH ? Q.hdel_tmp[i] : 0ull;
  uint64_t P0 = 0, Pp = 0, Pm = 0;
  for (uint32_t x = 0; x <= L; x++) {
    /* ---- insertion row blanked at x, if x is a class position of the variant t
            (length L + 1, t[y] = y < x ? q[y] : q[y - 1] around the gap) and t's
            class is split: class part of that position ---- */
    if (heavy_i) {
      int ci = -1;
      uint32_t key = base_i;
      for (uint32_t k = 0; k < K; k++) {
        const uint32_t mk = class_pos(L + 1, k, g.c0);
        if (mk == x) {
          if (ci < 0)
            ci = (int)k;
        } else {
          key ^= g.ctab[g.off_cr + k * A + s[mk < x ? mk : mk - 1]];
        }
      }
      if (ci >= 0)
        f(Q.goff[ci] + (key & g.cmask), P0 ^ hins ^ Pp,
          (x > 0 ? (uint32_t)s[x - 1] : 31u) | (x << 8) | (ITEM_INS << 24));
    }
    if (x == L)
      break;
    /* ---- deletion variant t = q without x (one per run of equal residues) whose
            slice of the main part is not the sibling staged for the tile's
            deletion pass ---- */
    if (L > 1 && (x == 0 || s[x] != s[x - 1])) {
      uint32_t key = base_d;
      if (heavy_d)
        for (uint32_t k = 0; k < K; k++) {
          const uint32_t mk = class_pos(L - 1, k, g.c0);
          key ^= g.ctab[g.off_cr + k * A + s[mk < x ? mk : mk + 1]];
        }
      if ((key & g.smask) != sibling) {
        uint64_t w = 0;
        if (HASH)
          w = P0 ^ hdel ^ (x > 0 ? Pm ^ Q.zob[A * (x - 1) + s[x]] : 0ull);
        f(Q.goff[K] + (key & g.smask), w, 31u | (x << 8) | (ITEM_DEL << 24));
      }
    }
    if (HASH) {
      P0 ^= Q.zob[A * x + s[x]];
      Pp ^= Q.zob[A * (x + 1) + s[x]];
      if (x > 0)
        Pm ^= Q.zob[A * (x - 1) + s[x]];
    }
  }
}

__global__ void __launch_bounds__(256)
keys_kernel(const QL Q)
{
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned long long alg = 0;
  if (i < Q.n) {
    const uint64_t b = Q.off[i];
    const uint32_t L = (uint32_t)(Q.off[i + 1] - b);
    const uint8_t *s = Q.res + b;
    const uint32_t vg = Q.genes ? Q.v[i] : 0u, jg = Q.genes ? Q.j[i] : 0u;
    uint32_t ck = 0;
    bool heavy = false;
    if (Q.sliced)
      ck = class_key_of(Q.geom.ctab, Q.geom, Q.A, Q.genes != 0, s, L, vg, jg, &heavy);
    if (Q.rows) {
      /* zobrist_hash (zobrist.cc:74-88) and, with -i, the two shifted hashes that
         seed the rolling indel enumeration (:90-104, :122-136) */
      uint64_t h = 0;
      if (Q.genes) {
        const uint64_t *gk = Q.zob + (uint64_t)Q.A * Q.zpos;
        h = gk[vg] ^ gk[Q.n_v + jg];
      }
      uint64_t hins = h, hdel = h;
      for (uint32_t p = 0; p < L; p++) {
        const uint32_t r = s[p];
        h ^= Q.zob[Q.A * p + r];
        if (Q.indels) {
          hins ^= Q.zob[Q.A * (p + 1) + r];
          if (p > 0)
            hdel ^= Q.zob[Q.A * (p - 1) + r];
        }
      }
      Q.h_tmp[i] = h;
      if (Q.indels) {
        Q.hins_tmp[i] = hins;
        Q.hdel_tmp[i] = hdel;
      }
      Q.ck_tmp[i] = ck;
    } else if (Q.sub2_items) {
      Q.ck_tmp[i] = ck;
    }
    const uint64_t gl = Q.longest - L;
    {
      const uint64_t bucket = Q.sliced ? 2 * (uint64_t)(ck & Q.geom.smask) + (heavy ? 1 : 0) : 0;
      const uint32_t g = (uint32_t)(bucket * Q.per_slice + gl);
      Q.grp[0][i] = g;
      atomicAdd(Q.cnt_g[0] + g, 1u);
    }
    if (Q.ngroups)
      for_each_item<false>(Q, i, ck, heavy, [&](uint32_t k, uint64_t, uint32_t) { atomicAdd(Q.ccnt + k, 1u); });
    alg = (uint64_t)L + 20 + 8 * variants_of(Q, s, L);
  }
  for (int o = 32; o > 0; o >>= 1)
    alg += __shfl_down(alg, o, WAVE);
  if ((threadIdx.x & 63) == 0 && alg)
    atomicAdd(Q.alg_bytes, alg);
}

/* ---- slices: tiles, chunks ------------------------------------------------ */

/* One thread per slice and pass.  WRITE = 0: what the slice needs (counted);
   WRITE = 1: the same walk, writing at the slice's exclusive prefix. */
template <int WRITE>
__global__ void __launch_bounds__(256)
slices_kernel(const QL Q, uint32_t pi)
{
  const uint64_t sl = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (sl >= Q.nslices)
    return;
  const uint32_t pass = pi == 0 ? 0u : 2u + pi;          /* class-row pass i = pass 3 + i */
  const uint32_t *cnt = Q.cnt_g[pi];
  SliceTot at;
  at.tiles = at.chunks = at.small = at.list = 0;
  at.res = 0;
  if (WRITE)
    at = Q.pre[pi][sl];
  const uint32_t tile_base = Q.tile0[pi] + at.tiles;
  uint32_t ntiles = 0;
  unsigned long long res = 0;
  uint64_t work_lens = 0;
  for (uint32_t hv = 0; hv < (Q.sliced ? 2u : 1u); hv++) {
    const uint64_t bucket = Q.sliced ? 2 * sl + hv : 0;
    const uint32_t tile_k = hv ? Q.geom.k : 0u;
    const uint32_t *c = cnt + bucket * Q.per_slice;
    /* lengths >= min_mixed share tiles: longest first, cut every 64 */
    uint64_t n_long = 0;
    uint32_t gl_end = 0;
    for (uint32_t gl = 0; gl < Q.per_slice; gl++) {
      const uint32_t L = Q.longest - gl;
      if (L < Q.min_mixed)
        break;
      gl_end = gl + 1;
      if (WRITE) {
        const uint64_t g = bucket * Q.per_slice + gl;
        Q.base_g[pi][g] = (uint32_t)((uint64_t)(tile_base + ntiles) * WAVE + n_long);
        if (pi == 0)
          Q.tfirst_g[g] = tile_base + ntiles + (uint32_t)(n_long / WAVE);
      }
      n_long += c[gl];
    }
    {
      const uint32_t nt = (uint32_t)((n_long + WAVE - 1) / WAVE);
      uint32_t gl = 0;
      uint64_t seen = 0;
      for (uint32_t k = 0; k < nt; k++) {
        while (seen + c[gl] <= (uint64_t)k * WAVE) {
          seen += c[gl];
          gl++;
        }
        const uint32_t L = Q.longest - gl;            /* longest query of the tile */
        const uint32_t nv = (uint32_t)min((uint64_t)WAVE, n_long - (uint64_t)k * WAVE);
        if (WRITE) {
          TileDesc td;
          td.len = L;
          td.nvalid = nv;
          td.res_base = (uint32_t)(Q.res0[pi] + at.res + res);
          td.pass = pass;
          td.slice = (uint32_t)sl;
          td.k = tile_k;
          Q.tiles[tile_base + ntiles + k] = td;
        }
        res += pi == 0 ? (unsigned long long)((L + 3) / 4) * WAVE : (unsigned long long)WAVE;
        work_lens += (uint64_t)(L + 1) * nv;
      }
      ntiles += nt;
    }
    /* shorter lengths: one tile group per length */
    for (uint32_t gl = gl_end; gl < Q.per_slice; gl++) {
      const uint32_t L = Q.longest - gl;
      const uint32_t n = c[gl];
      const uint32_t nt = (n + WAVE - 1) / WAVE;
      if (WRITE) {
        const uint64_t g = bucket * Q.per_slice + gl;
        Q.base_g[pi][g] = (tile_base + ntiles) * WAVE;
        if (pi == 0)
          Q.tfirst_g[g] = tile_base + ntiles;
      }
      for (uint32_t k = 0; k < nt; k++) {
        const uint32_t nv = min((uint32_t)WAVE, n - k * WAVE);
        if (WRITE) {
          TileDesc td;
          td.len = L;
          td.nvalid = nv;
          td.res_base = (uint32_t)(Q.res0[pi] + at.res + res);
          td.pass = pass;
          td.slice = (uint32_t)sl;
          td.k = tile_k;
          Q.tiles[tile_base + ntiles + k] = td;
        }
        res += pi == 0 ? (unsigned long long)((L + 3) / 4) * WAVE : (unsigned long long)WAVE;
        work_lens += (uint64_t)(L + 1) * nv;
      }
      ntiles += nt;
    }
  }
  /* chunks of the slice (sliced kernels only).  A slice with very few query
     tiles is not worth a workgroup + a staged copy: its tiles go to the list
     that single waves work through, probing the slice in HBM / L2. */
  uint32_t nchunks = 0, nsmall = 0, nlist = 0;
  if (Q.sliced && ntiles) {
    /* (variant 2 with -i: the deletion and insertion rows of a tile follow its
       substitution rows in the same unit, on the same staged slice -- its class keys
       have no length term -- so there is one chunk per slice, not one per pass) */
    const uint32_t reps = 1u;
    if (!Q.indels && (ntiles <= Q.small_max || (pi > 0 && Q.class_unstaged))) {
      nsmall = ntiles;
      if (WRITE)
        for (uint32_t t = 0; t < ntiles; t++)
          Q.small_tiles[Q.small0[pi] + at.small + t] = tile_base + t;
    } else {
      const uint32_t nc1 = (ntiles + Q.chunk_tiles - 1) / Q.chunk_tiles;
      nchunks = nc1 * reps;
      nlist = ntiles;
      if (WRITE) {
        for (uint32_t t = 0; t < ntiles; t++) {
          TileRef r;
          r.td = Q.tiles[tile_base + t];
          r.t = tile_base + t;
          r.pad = 0;
          Q.tile_refs[Q.list0[pi] + at.list + t] = r;
        }
        for (uint32_t k = 0; k < nc1; k++)
          for (uint32_t rp = 0; rp < reps; rp++) {
            Chunk ck;
            ck.slice = (uint32_t)sl;
            ck.first_tile = Q.list0[pi] + at.list + k * Q.chunk_tiles;
            ck.ntiles = min(Q.chunk_tiles, ntiles - k * Q.chunk_tiles);
            ck.pass = pass + rp;
            if (Q.sub2_items && pi == 0 && k == 0)
              ck.pass |= CHUNK_WITH_ITEMS;      /* (the slice's item blocks ride along) */
            Q.chunks[Q.chunk0[pi] + at.chunks + k * reps + rp] = ck;
          }
      }
    }
  }
  if (!WRITE) {
    SliceTot t;
    t.tiles = ntiles;
    t.chunks = nchunks;
    t.small = nsmall;
    t.list = nlist;
    t.res = res;
    Q.tot[pi][sl] = t;
  }
  (void)work_lens;
}

/* ---- placement ------------------------------------------------------------ */

__global__ void __launch_bounds__(256)
place_kernel(const QL Q)
{
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= Q.n)
    return;
  const uint32_t g = Q.grp[0][i];
  const uint32_t slot = Q.base_g[0][g] + atomicAdd(Q.fill_g[0] + g, 1u);
  const uint32_t tile = slot / WAVE, lane = slot % WAVE;
  const uint64_t b = Q.off[i];
  const uint32_t L = (uint32_t)(Q.off[i + 1] - b);
  Q.slot_of[i] = slot;
  Q.qlen[slot] = (uint16_t)L;
  Q.qorig[slot] = (uint32_t)i;
  Q.qrep[slot] = Q.existence ? (uint32_t)i : Q.rep[i];    /* -x: the row is the sequence itself */
  if (Q.genes) {
    Q.qv[slot] = Q.v[i];
    Q.qj[slot] = Q.j[i];
  }
  if (Q.counts)
    Q.qcnt[slot] = Q.cnt[i];
  if (Q.rows) {
    Q.qgh[slot] = Q.h_tmp[i];
    Q.qck[slot] = Q.ck_tmp[i];
    if (Q.indels) {
      Q.qhins[slot] = Q.hins_tmp[i];
      Q.qhdel[slot] = Q.hdel_tmp[i];
    }
  } else if (Q.genes) {
    const uint64_t *gk = Q.zob + (uint64_t)Q.A * Q.zpos;
    Q.qgh[slot] = gk[Q.v[i]] ^ gk[Q.n_v + Q.j[i]];
  }
  /* residues four to a dword, position-major / lane-minor (layout.h TileDesc), and
     the first nine dwords once more in the query's record */
  uint32_t *dst = Q.qres + Q.tiles[tile].res_base + lane;
  const uint8_t *s = Q.res + b;
  QueryRec qr;
  qr.cnt = Q.counts ? Q.cnt[i] : 1ull;
  qr.v = Q.genes ? Q.v[i] : 0u;
  qr.j = Q.genes ? Q.j[i] : 0u;
  qr.rep = Q.existence ? (uint32_t)i : Q.rep[i];
  qr.len = L;
  qr.pad = 0;
#pragma unroll
  for (uint32_t w = 0; w < 9; w++)
    qr.res[w] = 0;
  for (uint32_t w = 0; 4 * w < L; w++) {
    uint32_t d = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)
      if (4 * w + k < L)
        d |= (uint32_t)s[4 * w + k] << (8 * k);
    dst[(size_t)w * WAVE] = d;
    if (w < 9)
      qr.res[w] = d;
  }
  Q.qrec[slot] = qr;
}

/* variant 2, class rows: per (class part, slice) the items padded to whole blocks
   of 64, and the chunks (at most cblocks blocks each) they make */
__global__ void __launch_bounds__(256)
class_pad_kernel(const QL Q, uint32_t *padded)
{
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= (uint64_t)Q.nitem_slices)
    return;
  uint32_t blocks = (Q.ccnt[k] + WAVE - 1) / WAVE;
  padded[k] = blocks * WAVE;
  if (Q.sub2_items) {
    /* one chunk list per slice, written by its first group */
    if (k % Q.ngroups != 0) {
      Q.cnch[k] = 0;
      return;
    }
    for (uint32_t g = 1; g < Q.ngroups; g++)
      blocks += (Q.ccnt[k + g] + WAVE - 1) / WAVE;
    if (Q.tot[0][k / Q.ngroups].chunks > 0)
      blocks = 0;                            /* (ride along with the slice's first main chunk) */
  }
  Q.cnch[k] = (blocks + Q.cblocks - 1) / Q.cblocks;
}

__global__ void __launch_bounds__(256)
class_chunks_kernel(const QL Q)
{
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= (uint64_t)Q.nitem_slices)
    return;
  uint32_t gi = 0;
  for (uint32_t x = 1; x < Q.ngroups; x++)
    if (k >= Q.goff[x])
      gi = x;
  uint32_t blocks = (Q.ccnt[k] + WAVE - 1) / WAVE;
  if (Q.sub2_items) {
    if (k % Q.ngroups != 0)
      return;
    for (uint32_t g = 1; g < Q.ngroups; g++)
      blocks += (Q.ccnt[k + g] + WAVE - 1) / WAVE;
    gi = 0;                                  /* (which group a block belongs to is in its items) */
    const uint32_t sl = (uint32_t)(k / Q.ngroups);
    if (Q.tot[0][sl].chunks > 0) {           /* the slice's first main chunk takes them along */
      Q.slice_items[sl] = make_uint2(Q.cbase[k], blocks);
      return;
    }
    Q.slice_items[sl] = make_uint2(0u, 0u);
  }
  const uint32_t nc = (blocks + Q.cblocks - 1) / Q.cblocks;
  for (uint32_t q = 0; q < nc; q++) {
    Chunk ck;
    ck.slice = Q.sub2_items ? (uint32_t)(k / Q.ngroups) : Q.gslice0[gi] + (uint32_t)(k - Q.goff[gi]);
    ck.first_tile = Q.cbase[k] + q * Q.cblocks * WAVE;      /* first item */
    ck.ntiles = min(Q.cblocks, blocks - q * Q.cblocks);     /* blocks of 64 items */
    ck.pass = 3 + gi;
    Q.chunks[Q.cchunk0 + Q.cchpre[k] + q] = ck;
  }
}

/* an item: the row's (variant's) hash, what to exclude / where / what kind, and the
   query's slot in pass 0 */
__global__ void __launch_bounds__(256)
place_items_kernel(const QL Q)
{
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= Q.n)
    return;
  const uint32_t slot = Q.slot_of[i];
  const uint32_t ck = Q.ck_tmp[i];
  const bool heavy = Q.geom.k > 0 && class_is_heavy(Q.geom.ctab, Q.geom, class_base_of(Q, i));
  uint64_t hq = 0;
  cmpr::ResPack pk{};
  if (Q.sub2_items && heavy) {
    const uint64_t b = Q.off[i];
    const uint32_t L = (uint32_t)(Q.off[i + 1] - b);
    if (L <= RESPACK_MAX) {
      if (Q.genes) {
        const uint64_t *gk = Q.zob + (uint64_t)Q.A * Q.zpos;
        hq = gk[Q.v[i]] ^ gk[Q.n_v + Q.j[i]];
      }
      for (uint32_t x = 0; x < L; x++) {
        const uint32_t r = Q.res[b + x];
        hq ^= Q.zob[Q.A * x + r];
        pk.w[x >> 4] |= (r & 3u) << ((x & 15u) * 2u);
      }
    }
  }
  for_each_item<true>(Q, i, ck, heavy, [&](uint32_t k, uint64_t w, uint32_t crp) {
    const uint32_t item = Q.cbase[k] + atomicAdd(Q.cfill + k, 1u);
    if (Q.sub2_items) {
      w = hq;                                   /* the query's hash and residues travel with the item */
      Q.cpk[item] = pk;
    }
    Q.cw[item] = w;
    Q.cmain[item] = slot;
    Q.crp[item] = crp;
  });
}

/* ---- -i: tiles regrouped by the slice their indel variants fall into ------- */

/* main-pass group g -> (sibling slice of pass ip, its tile count) */
__device__ inline bool sibling_of(const QL &Q, uint64_t g, uint32_t ip, uint32_t &sib, uint32_t &nt)
{
  const uint64_t bucket = g / Q.per_slice;
  const uint32_t gl = (uint32_t)(g % Q.per_slice);
  const uint32_t L = Q.longest - gl;
  nt = (Q.cnt_g[0][g] + WAVE - 1) / WAVE;
  if (nt == 0 || (ip == 2 && L < 2))
    return false;
  const uint32_t dlen = Q.geom.ctab[L] ^ Q.geom.ctab[ip == 1 ? L + 1 : L - 1];
  sib = ((uint32_t)(bucket / 2) ^ dlen) & Q.geom.smask;
  return true;
}

__global__ void __launch_bounds__(256)
sibling_count_kernel(const QL Q, uint64_t G, uint32_t *sib_cnt /* [2][nslices] */)
{
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= G)
    return;
  for (uint32_t ip = 1; ip <= 2; ip++) {
    uint32_t sib, nt;
    if (sibling_of(Q, g, ip, sib, nt))
      atomicAdd(sib_cnt + (uint64_t)(ip - 1) * Q.nslices + sib, nt);
  }
}

/* per (pass, sibling slice): chunks needed */
__global__ void __launch_bounds__(256)
sibling_chunks_kernel(const uint32_t *sib_cnt, uint64_t n, uint32_t chunk_tiles, uint32_t *nchunks)
{
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (k < n)
    nchunks[k] = (sib_cnt[k] + chunk_tiles - 1) / chunk_tiles;
}

__global__ void __launch_bounds__(256)
sibling_fill_kernel(const QL Q, uint64_t G, const uint32_t *list_pre, uint32_t *sib_fill,
                    uint32_t list_base)
{
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= G)
    return;
  for (uint32_t ip = 1; ip <= 2; ip++) {
    uint32_t sib, nt;
    if (!sibling_of(Q, g, ip, sib, nt))
      continue;
    const uint64_t k = (uint64_t)(ip - 1) * Q.nslices + sib;
    const uint32_t at = list_base + list_pre[k] + atomicAdd(sib_fill + k, nt);
    for (uint32_t t = 0; t < nt; t++) {
      TileRef r;
      r.t = Q.tfirst_g[g] + t;
      r.td = Q.tiles[r.t];
      r.pad = 0;
      Q.tile_refs[at + t] = r;
    }
  }
}

__global__ void __launch_bounds__(256)
sibling_write_chunks_kernel(const QL Q, const uint32_t *sib_cnt, const uint32_t *list_pre,
                            const uint32_t *chunk_pre, uint32_t list_base, uint32_t chunk_base)
{
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= 2 * Q.nslices)
    return;
  const uint32_t n = sib_cnt[k];
  const uint32_t nc = (n + Q.chunk_tiles - 1) / Q.chunk_tiles;
  for (uint32_t q = 0; q < nc; q++) {
    Chunk ck;
    ck.slice = (uint32_t)(k % Q.nslices);
    ck.first_tile = list_base + list_pre[k] + q * Q.chunk_tiles;
    ck.ntiles = min(Q.chunk_tiles, n - q * Q.chunk_tiles);
    ck.pass = 1 + (uint32_t)(k / Q.nslices);
    Q.chunks[chunk_base + chunk_pre[k] + q] = ck;
  }
}

/* ---- chunk order: heaviest first ------------------------------------------ */

__global__ void __launch_bounds__(256)
chunk_work_kernel(const QL Q, uint32_t nchunks, uint32_t *work, uint32_t *idx)
{
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nchunks)
    return;
  const Chunk ck = Q.chunks[k];
  const uint32_t cpass = ck.pass & ~CHUNK_WITH_ITEMS;
  uint64_t w = cpass >= 3 ? (uint64_t)ck.ntiles * WAVE * (Q.sub2_items ? 48 : 2) : 0;
  for (uint32_t t = 0; cpass < 3 && t < ck.ntiles; t++) {
    const TileDesc td = Q.tile_refs[ck.first_tile + t].td;
    w += (uint64_t)(cpass == 0 ? td.len + 1 : cpass == 1 ? td.len + 2 : 2) * td.nvalid;
  }
  if (ck.pass & CHUNK_WITH_ITEMS)
    w += (uint64_t)Q.slice_items[ck.slice].y * WAVE * 48;
  work[k] = (uint32_t)min(w, (uint64_t)0xffffffffu);
  idx[k] = k;
}

__global__ void __launch_bounds__(256)
small_len_kernel(const QL Q, uint32_t nsmall, uint32_t *len)
{
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k < nsmall)
    len[k] = Q.tiles[Q.small_tiles[k]].len;
}

__global__ void __launch_bounds__(256)
gather_chunks_kernel(const Chunk *src, const uint32_t *idx, uint32_t n, Chunk *dst)
{
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k < n)
    dst[k] = src[idx[k]];
}

/* Work sharding (tunables work_shard_index / _count): which context of `step` takes
   the work filed under a slice in a pass.  By slice, not by position in the work
   list: WHICH queries a slice holds is a property of the input, how they are cut
   into tiles and chunks is decided by atomics and differs from one layout run (one
   GPU) to the next. */
__host__ __device__ inline uint32_t work_owner(uint32_t slice, uint32_t pass, uint32_t step)
{
  return (uint32_t)((((uint64_t)slice * 2654435761u + pass * 40503u) >> 7) % step);
}

__global__ void __launch_bounds__(256)
chunk_mine_kernel(const Chunk *chunks, const uint32_t *idx, uint32_t n, uint32_t first, uint32_t step,
                  unsigned char *flag)
{
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k < n) {
    const Chunk ck = chunks[idx[k]];
    flag[k] = work_owner(ck.slice, ck.pass & ~CHUNK_WITH_ITEMS, step) == first ? 1 : 0;
  }
}

__global__ void __launch_bounds__(256)
small_mine_kernel(const QL Q, const uint32_t *small, uint32_t n, uint32_t first, uint32_t step,
                  unsigned char *flag)
{
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k < n) {
    const TileDesc td = Q.tiles[small[k]];
    flag[k] = work_owner(td.slice, td.pass, step) == first ? 1 : 0;
  }
}

template <typename T>
struct Tmp {
  DevBuf<T> b;
  ~Tmp() { b.release(); }
};

inline uint32_t blocks_for(uint64_t n)
{
  return (uint32_t)std::max<uint64_t>(1, (n + 255) / 256);
}

const char *verr_message(uint32_t e)
{
  switch (e) {
  case VERR_OFFSETS: return "offsets not monotone";
  case VERR_LONG:    return "sequence longer than 65535 residues";
  case VERR_REP:     return "repertoire number out of range";
  case VERR_GENE:    return "gene number out of range";
  case VERR_COUNT:   return "duplicate_count must be >= 1";
  default:           return "residue code out of range";
  }
}

}  // namespace

/* Upload + validation of a set view, shared with cmpr_set_reference
   (ref_index): raw arrays into the given device buffers, errors as the host
   scan reported them, longest sequence and per-repertoire count totals. */
int cmpr_upload_and_validate(cmpr_context *c, const cmpr_set_view *s, DevBuf<uint8_t> &res,
                             DevBuf<uint64_t> &off, DevBuf<uint32_t> &v, DevBuf<uint32_t> &j,
                             DevBuf<uint32_t> &rep, DevBuf<uint64_t> &cnt, uint32_t &longest,
                             std::vector<double> &rep_total)
{
  int rc;
  const uint64_t total = s->n ? s->offsets[s->n] : 0;
  static const uint64_t zero_off[1] = {0};
  if ((rc = dev_upload(c, res, s->residues, (size_t)total))) return rc;
  if ((rc = dev_upload(c, off, s->n ? s->offsets : zero_off, (size_t)s->n + 1))) return rc;
  if ((rc = dev_upload(c, rep, s->repertoire, (size_t)s->n))) return rc;
  if (!c->opt.ignore_genes) {
    if ((rc = dev_upload(c, v, s->v_gene, (size_t)s->n))) return rc;
    if ((rc = dev_upload(c, j, s->j_gene, (size_t)s->n))) return rc;
  } else {
    v.release();
    j.release();
  }
  if (!c->opt.ignore_counts) {
    if ((rc = dev_upload(c, cnt, s->count, (size_t)s->n))) return rc;
  } else {
    cnt.release();
  }
  longest = 0;
  rep_total.assign(s->n_repertoires, 0.0);
  if (s->n == 0)
    return CMPR_OK;
  Tmp<uint32_t> verr;
  Tmp<double> tot;
  if ((rc = dev_alloc(c, verr.b, 2))) return rc;
  if ((rc = dev_alloc(c, tot.b, s->n_repertoires))) return rc;
  HIP_TRY(c, hipMemsetAsync(verr.b.p, 0, 2 * sizeof(uint32_t), c->stream));
  HIP_TRY(c, hipMemsetAsync(tot.b.p, 0, s->n_repertoires * sizeof(double), c->stream));
  QL Q;
  memset(&Q, 0, sizeof Q);
  Q.res = res.p; Q.off = off.p; Q.v = v.p; Q.j = j.p; Q.rep = rep.p; Q.cnt = cnt.p;
  Q.n = s->n;
  Q.n_rep = s->n_repertoires;
  Q.n_v_max = c->opt.n_v_genes;
  Q.n_j_max = c->opt.n_j_genes;
  Q.genes = c->opt.ignore_genes ? 0 : 1;
  Q.counts = c->opt.ignore_counts ? 0 : 1;
  Q.verr = verr.b.p;
  Q.rep_total = tot.b.p;
  const size_t lds = s->n_repertoires <= 2048 ? s->n_repertoires * sizeof(double) : 0;
  hipLaunchKernelGGL(validate_seq_kernel, dim3(blocks_for(s->n)), dim3(256), lds, c->stream, Q);
  HIP_TRY(c, hipGetLastError());
  uint32_t hv[2] = {0, 0};
  HIP_TRY(c, hipMemcpyAsync(hv, verr.b.p, sizeof hv, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (hv[0])
    return fail(c, hv[0] == VERR_LONG ? CMPR_EUNSUPPORTED : CMPR_EINVAL, verr_message(hv[0]));
  /* the offsets are monotone: the residues of the set are [0, offsets[n]) */
  if (total) {
    hipLaunchKernelGGL(validate_res_kernel, dim3(blocks_for((total + 15) / 16)), dim3(256), 0,
                       c->stream, res.p, total, (uint32_t)c->opt.alphabet_size, verr.b.p);
    HIP_TRY(c, hipGetLastError());
  }
  HIP_TRY(c, hipMemcpyAsync(hv, verr.b.p, sizeof hv, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(rep_total.data(), tot.b.p, s->n_repertoires * sizeof(double),
                            hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (hv[0])
    return fail(c, CMPR_EINVAL, verr_message(hv[0]));
  longest = hv[1];
  return CMPR_OK;
}

/* the entries of `list` (n of them) whose flag -- set by `mark` -- is 1, order kept */
template <typename Mark>
static int select_mine(cmpr_context *c, const uint32_t *list, uint32_t n, Tmp<uint32_t> &sel, uint64_t &count,
                       Mark mark)
{
  int rc;
  Tmp<unsigned char> flag;
  Tmp<uint32_t> nsel;
  if ((rc = dev_alloc(c, flag.b, (size_t)n))) return rc;
  if ((rc = dev_alloc(c, sel.b, (size_t)n))) return rc;
  if ((rc = dev_alloc(c, nsel.b, 1))) return rc;
  mark(flag.b.p);
  HIP_TRY(c, hipGetLastError());
  size_t sb = 0;
  (void)hipcub::DeviceSelect::Flagged(nullptr, sb, list, flag.b.p, sel.b.p, nsel.b.p, (int)n, c->stream);
  Tmp<char> st;
  if ((rc = dev_alloc(c, st.b, sb))) return rc;
  HIP_TRY(c, hipcub::DeviceSelect::Flagged(st.b.p, sb, list, flag.b.p, sel.b.p, nsel.b.p, (int)n, c->stream));
  uint32_t h = 0;
  HIP_TRY(c, hipMemcpyAsync(&h, nsel.b.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  count = h;
  return CMPR_OK;
}

int cmpr_layout_queries(cmpr_context *c, const cmpr_set_view *s)
{
  int rc;
  /* ---- the caller's arrays, as they are; validation ---- */
  Tmp<uint8_t> res;
  Tmp<uint64_t> off, cnt;
  Tmp<uint32_t> v, j, rep;
  uint32_t longest = 0;
  if ((rc = cmpr_upload_and_validate(c, s, res.b, off.b, v.b, j.b, rep.b, cnt.b, longest, c->tot1)))
    return rc;
  if (longest + EXTRA_POSITIONS > c->zpos)
    return fail(c, CMPR_EINVAL,
                "query longer than the longest_query given to cmpr_set_reference");
  c->n1 = s->n;
  c->R1 = c->opt.existence ? (uint32_t)s->n : s->n_repertoires;

  /* exact integer accumulation needs every cell < 2^64; a cell is at most
     (sum of counts of its row repertoire) x (sum of counts of its column one) */
  {
    double m1 = 0, m2 = 0;
    for (double x : c->tot1) m1 = std::max(m1, x);
    for (double x : c->tot2) m2 = std::max(m2, x);
    c->max_cell_bound = m1 * m2;
    if (!is_f64_score(c->opt) && c->max_cell_bound >= 18446744073709551616.0 / 2)
      return fail(c, CMPR_EUNSUPPORTED,
                  "duplicate counts too large for exact 64-bit accumulation");
  }

  const uint32_t A = (uint32_t)c->opt.alphabet_size;
  const uint64_t nslices = c->sliced ? (uint64_t)c->geom.smask + 1 : 1;
  const uint64_t nbuckets = c->sliced ? 2 * nslices : 1;
  const uint64_t per_slice = (uint64_t)longest + 1;
  if (nbuckets * per_slice >= 0xffffffffull)
    return fail(c, CMPR_EUNSUPPORTED, "too many (slice, length) groups");
  const uint64_t G = nbuckets * per_slice;
  /* class-row passes of variant 2 (flat items, not tiles) */
  /* item groups of variant 2 (flat items, not tiles; see QL) */
  const uint32_t npass = 1;                  /* passes laid out as tiles */
  uint32_t ngroups = 0;
  uint32_t goff[MAX_GROUPS] = {0}, gslices[MAX_GROUPS] = {0}, gslice0[MAX_GROUPS] = {0};
  const bool sub2_items = !c->rows && c->sliced && c->geom.k > 0 && c->opt.differences == 2 &&
                          c->opt.alphabet_size == 4 && !c->opt.indels &&
                          c->sub2_items != 0;
  uint64_t ncs = 0;                          /* counters over all groups */
  if (c->rows && c->geom.k > 0 && c->opt.differences >= 1) {
    for (uint32_t g = 0; g < c->geom.k; g++) {
      goff[g] = (uint32_t)ncs;
      gslices[g] = c->geom.cmask + 1;
      gslice0[g] = row_slice(c->geom, 0, (int)g);
      ncs += gslices[g];
    }
    ngroups = c->geom.k;
    if (c->opt.indels) {
      goff[ngroups] = (uint32_t)ncs;
      gslices[ngroups] = c->geom.smask + 1;
      gslice0[ngroups] = 0;
      ncs += gslices[ngroups];
      ngroups++;
    }
  }
  if (sub2_items) {
    ngroups = c->geom.k * 3;
    for (uint32_t g = 0; g < ngroups; g++) {
      goff[g] = (uint32_t)ncs;
      gslices[g] = c->geom.smask + 1;
      gslice0[g] = 0;
      ncs += gslices[g];
    }
  }
  if (ncs >= 0x7fffffffull)
    return fail(c, CMPR_EUNSUPPORTED, "too many item slices");
  c->npasses = 1 + ngroups;
  const uint64_t chunk_tiles =
      c->chunk_tiles > 0 ? (uint64_t)c->chunk_tiles : c->rows ? 64 : 8 * (uint64_t)c->waves_per_block;
  c->chunk_cap = (uint32_t)chunk_tiles;
  /* Without -i, queries long enough to contain all class positions unwrapped
     (len >= c0 + K) may share a tile with queries of other lengths: inside a
     (slice, heavy) group they are laid out longest first and cut every 64,
     whatever their lengths; the kernel masks each lane by its own length.
     This keeps the padding small when there are many slices and lengths
     (100M references x 49 nucleotide lengths = 1.6M groups).  Shorter queries,
     and all queries with -i (the indel passes stage one sibling slice per
     length), keep one tile group per length. */
  /* (variant 2 mixes lengths with -i too: its class keys then carry no length term,
     ref_index.hip, and its indel rows mask every lane by its own length) */
  const bool mixed_ok = c->sliced && (!c->opt.indels || c->rows);

  QL Q;
  memset(&Q, 0, sizeof Q);
  Q.res = res.b.p; Q.off = off.b.p; Q.v = v.b.p; Q.j = j.b.p; Q.rep = rep.b.p; Q.cnt = cnt.b.p;
  Q.n = s->n;
  Q.n_rep = s->n_repertoires;
  Q.A = A;
  Q.zpos = c->zpos;
  Q.n_v = c->opt.ignore_genes ? 0 : c->opt.n_v_genes;
  Q.longest = longest;
  Q.per_slice = (uint32_t)per_slice;
  Q.genes = c->opt.ignore_genes ? 0 : 1;
  Q.counts = c->opt.ignore_counts ? 0 : 1;
  Q.existence = c->opt.existence ? 1 : 0;
  Q.indels = c->opt.indels ? 1 : 0;
  Q.differences = (uint32_t)c->opt.differences;
  Q.sliced = c->sliced ? 1 : 0;
  Q.rows = c->rows ? 1 : 0;
  Q.zob = c->zob.p;
  Q.geom = c->geom;
  Q.npass = npass;
  Q.min_mixed = mixed_ok ? c->geom.c0 + c->geom.k + (c->opt.indels ? 1u : 0u) : 0xffffffffu;
  Q.chunk_tiles = (uint32_t)chunk_tiles;
  Q.small_max = (uint32_t)c->small_slice_tiles;
  Q.class_unstaged = c->class_rows_unstaged ? 1u : 0u;
  Q.nbuckets = nbuckets;
  Q.nslices = nslices;
  Q.ngroups = ngroups;
  Q.sub2_items = sub2_items ? 1u : 0u;
  c->sub2_active = sub2_items;
  for (uint32_t g = 0; g < MAX_GROUPS; g++) {
    Q.goff[g] = goff[g];
    Q.gslices[g] = gslices[g];
    Q.gslice0[g] = gslice0[g];
  }
  Q.nitem_slices = (uint32_t)ncs;
  Q.cblocks = 4096;                          /* (a claim word counts tiles in 16 bits) */

  /* ---- scratch: group counters, per-query keys ---- */
  Tmp<uint32_t> gcnt, gbase, gfill, grp, tfirst, ck_tmp, slot_of;
  Tmp<uint64_t> h_tmp, hins_tmp, hdel_tmp;
  Tmp<SliceTot> tot, pre;
  Tmp<unsigned long long> alg;
  Tmp<uint32_t> ccnt, cbase, cfill, cnch, cchpre, cpad;
  if (ngroups) {
    if ((rc = dev_alloc(c, ccnt.b, (size_t)ncs))) return rc;
    if ((rc = dev_alloc(c, cbase.b, (size_t)ncs))) return rc;
    if ((rc = dev_alloc(c, cfill.b, (size_t)ncs))) return rc;
    if ((rc = dev_alloc(c, cnch.b, (size_t)ncs))) return rc;
    if ((rc = dev_alloc(c, cchpre.b, (size_t)ncs))) return rc;
    if ((rc = dev_alloc(c, cpad.b, (size_t)ncs))) return rc;
    HIP_TRY(c, hipMemsetAsync(ccnt.b.p, 0, (size_t)ncs * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(cfill.b.p, 0, (size_t)ncs * sizeof(uint32_t), c->stream));
    Q.ccnt = ccnt.b.p;
    Q.cbase = cbase.b.p;
    Q.cfill = cfill.b.p;
    Q.cnch = cnch.b.p;
    Q.cchpre = cchpre.b.p;
  }
  if ((rc = dev_alloc(c, gcnt.b, (size_t)G * npass))) return rc;
  if ((rc = dev_alloc(c, gbase.b, (size_t)G * npass))) return rc;
  if ((rc = dev_alloc(c, gfill.b, (size_t)G * npass))) return rc;
  if ((rc = dev_alloc(c, grp.b, (size_t)s->n * npass))) return rc;
  if ((rc = dev_alloc(c, tfirst.b, (size_t)G))) return rc;
  if ((rc = dev_alloc(c, slot_of.b, (size_t)s->n))) return rc;
  if ((rc = dev_alloc(c, tot.b, (size_t)nslices * npass))) return rc;
  if ((rc = dev_alloc(c, pre.b, (size_t)nslices * npass))) return rc;
  if ((rc = dev_alloc(c, alg.b, 1))) return rc;
  if (sub2_items)
    if ((rc = dev_alloc(c, ck_tmp.b, (size_t)s->n))) return rc;
  if (c->rows) {
    if ((rc = dev_alloc(c, h_tmp.b, (size_t)s->n))) return rc;
    if ((rc = dev_alloc(c, ck_tmp.b, (size_t)s->n))) return rc;
    if (c->opt.indels) {
      if ((rc = dev_alloc(c, hins_tmp.b, (size_t)s->n))) return rc;
      if ((rc = dev_alloc(c, hdel_tmp.b, (size_t)s->n))) return rc;
    }
  }
  HIP_TRY(c, hipMemsetAsync(gcnt.b.p, 0, (size_t)G * npass * sizeof(uint32_t), c->stream));
  HIP_TRY(c, hipMemsetAsync(gfill.b.p, 0, (size_t)G * npass * sizeof(uint32_t), c->stream));
  HIP_TRY(c, hipMemsetAsync(alg.b.p, 0, sizeof(unsigned long long), c->stream));
  for (uint32_t pi = 0; pi < npass; pi++) {
    Q.cnt_g[pi] = gcnt.b.p + (size_t)G * pi;
    Q.base_g[pi] = gbase.b.p + (size_t)G * pi;
    Q.fill_g[pi] = gfill.b.p + (size_t)G * pi;
    Q.grp[pi] = grp.b.p + (size_t)s->n * pi;
    Q.tot[pi] = tot.b.p + (size_t)nslices * pi;
    Q.pre[pi] = pre.b.p + (size_t)nslices * pi;
  }
  Q.tfirst_g = tfirst.b.p;
  Q.h_tmp = h_tmp.b.p;
  Q.hins_tmp = hins_tmp.b.p;
  Q.hdel_tmp = hdel_tmp.b.p;
  Q.ck_tmp = ck_tmp.b.p;
  Q.slot_of = slot_of.b.p;
  Q.alg_bytes = alg.b.p;

  if (s->n) {
    hipLaunchKernelGGL(keys_kernel, dim3(blocks_for(s->n)), dim3(256), 0, c->stream, Q);
    HIP_TRY(c, hipGetLastError());
  }

  /* ---- per slice: what it needs; exclusive scan; totals ---- */
  Tmp<char> cub_tmp;
  size_t cub_bytes = 0;
  {
    SliceTot zero;
    memset(&zero, 0, sizeof zero);
    (void)hipcub::DeviceScan::ExclusiveScan(nullptr, cub_bytes, Q.tot[0], Q.pre[0], SliceTotSum(), zero,
                                      (int)nslices, c->stream);
    if ((rc = dev_alloc(c, cub_tmp.b, cub_bytes))) return rc;
    for (uint32_t pi = 0; pi < npass; pi++) {
      hipLaunchKernelGGL(slices_kernel<0>, dim3(blocks_for(nslices)), dim3(256), 0, c->stream, Q, pi);
      HIP_TRY(c, hipGetLastError());
      size_t b = cub_bytes;
      HIP_TRY(c, hipcub::DeviceScan::ExclusiveScan(cub_tmp.b.p, b, Q.tot[pi], Q.pre[pi],
                                                   SliceTotSum(), zero, (int)nslices, c->stream));
    }
  }
  std::vector<SliceTot> last_tot(npass), last_pre(npass);
  for (uint32_t pi = 0; pi < npass; pi++) {
    HIP_TRY(c, hipMemcpyAsync(&last_tot[pi], Q.tot[pi] + (nslices - 1), sizeof(SliceTot),
                              hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&last_pre[pi], Q.pre[pi] + (nslices - 1), sizeof(SliceTot),
                              hipMemcpyDeviceToHost, c->stream));
  }
  unsigned long long alg_bytes = 0;
  HIP_TRY(c, hipMemcpyAsync(&alg_bytes, alg.b.p, sizeof alg_bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->algorithmic_bytes = alg_bytes;

  uint64_t ntiles = 0, nchunks = 0, nlist = 0, nsmall = 0, res_words = 0, cslots = 0;
  for (uint32_t pi = 0; pi < npass; pi++) {
    const SliceTotSum add;
    const SliceTot t = add(last_pre[pi], last_tot[pi]);
    Q.tile0[pi] = (uint32_t)ntiles;
    Q.chunk0[pi] = (uint32_t)nchunks;
    Q.list0[pi] = (uint32_t)nlist;
    Q.small0[pi] = (uint32_t)nsmall;
    Q.res0[pi] = pi == 0 ? 0ull : cslots;
    ntiles += t.tiles;
    nchunks += t.chunks;
    nlist += t.list;
    nsmall += t.small;
    if (pi == 0)
      res_words = t.res;
    else
      cslots += t.res;
    if (pi == 0)
      c->nmain_tiles = t.tiles;
  }
  /* ---- class rows: items per (class part, slice), padded to blocks of 64 ---- */
  uint64_t class_chunks = 0;
  if (ngroups) {
    hipLaunchKernelGGL(class_pad_kernel, dim3(blocks_for(ncs)), dim3(256), 0, c->stream, Q, cpad.b.p);
    HIP_TRY(c, hipGetLastError());
    size_t b2 = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b2, cpad.b.p, cbase.b.p, (int)ncs, c->stream);
    Tmp<char> t2;
    if ((rc = dev_alloc(c, t2.b, b2))) return rc;
    size_t bb = b2;
    HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(t2.b.p, bb, cpad.b.p, cbase.b.p, (int)ncs, c->stream));
    bb = b2;
    HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(t2.b.p, bb, cnch.b.p, cchpre.b.p, (int)ncs, c->stream));
    uint32_t last[4];
    HIP_TRY(c, hipMemcpyAsync(&last[0], cpad.b.p + (ncs - 1), 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&last[1], cbase.b.p + (ncs - 1), 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&last[2], cnch.b.p + (ncs - 1), 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&last[3], cchpre.b.p + (ncs - 1), 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    cslots = (uint64_t)last[0] + last[1];
    class_chunks = (uint64_t)last[2] + last[3];
  }
  if (ntiles * WAVE >= 0xffffffffull)
    return fail(c, CMPR_EUNSUPPORTED, "too many query tiles");
  if (res_words + 9 * WAVE >= 0xffffffffull || cslots + WAVE >= 0xffffffffull)
    return fail(c, CMPR_EUNSUPPORTED, "query set too large for 32-bit residue positions");
  const uint64_t main_chunks = nchunks, main_list = nlist;

  /* ---- -i: sibling lists of the two indel passes ---- */
  Tmp<uint32_t> sib_cnt, sib_nch, sib_lpre, sib_cpre, sib_fill;
  uint64_t sib_chunks = 0, sib_list = 0;
  /* variant 1: the indel passes list the tiles by sibling slice; variant 2: the same
     tiles, the same slice, three chunks per chunk (slices_kernel) */
  const bool indel_passes = c->sliced && c->opt.indels && !c->rows;
  if (indel_passes) {
    const uint64_t n2s = 2 * nslices;
    if ((rc = dev_alloc(c, sib_cnt.b, (size_t)n2s))) return rc;
    if ((rc = dev_alloc(c, sib_nch.b, (size_t)n2s))) return rc;
    if ((rc = dev_alloc(c, sib_lpre.b, (size_t)n2s))) return rc;
    if ((rc = dev_alloc(c, sib_cpre.b, (size_t)n2s))) return rc;
    if ((rc = dev_alloc(c, sib_fill.b, (size_t)n2s))) return rc;
    HIP_TRY(c, hipMemsetAsync(sib_cnt.b.p, 0, n2s * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(sib_fill.b.p, 0, n2s * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(sibling_count_kernel, dim3(blocks_for(G)), dim3(256), 0, c->stream, Q, G,
                       sib_cnt.b.p);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(sibling_chunks_kernel, dim3(blocks_for(n2s)), dim3(256), 0, c->stream,
                       sib_cnt.b.p, n2s, (uint32_t)chunk_tiles, sib_nch.b.p);
    HIP_TRY(c, hipGetLastError());
    size_t b2 = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b2, sib_cnt.b.p, sib_lpre.b.p, (int)n2s, c->stream);
    Tmp<char> t2;
    if ((rc = dev_alloc(c, t2.b, b2))) return rc;
    size_t bb = b2;
    HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(t2.b.p, bb, sib_cnt.b.p, sib_lpre.b.p, (int)n2s, c->stream));
    bb = b2;
    HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(t2.b.p, bb, sib_nch.b.p, sib_cpre.b.p, (int)n2s, c->stream));
    uint32_t lc[2], lp[2], cc[2];
    HIP_TRY(c, hipMemcpyAsync(&lc[0], sib_cnt.b.p + (n2s - 1), 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&lp[0], sib_lpre.b.p + (n2s - 1), 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&cc[0], sib_nch.b.p + (n2s - 1), 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&cc[1], sib_cpre.b.p + (n2s - 1), 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    sib_list = (uint64_t)lc[0] + lp[0];
    sib_chunks = (uint64_t)cc[0] + cc[1];
    nchunks += sib_chunks;
    nlist += sib_list;
  }
  Q.cchunk0 = (uint32_t)nchunks;
  nchunks += class_chunks;
  if (nchunks >= 0xffffffffull || nlist >= 0xffffffffull)
    return fail(c, CMPR_EUNSUPPORTED, "too many chunks");

  /* ---- the resident layout ---- */
  const size_t slots = (size_t)c->nmain_tiles * WAVE;
  c->ntiles = (uint32_t)ntiles;
  c->nchunks = (uint32_t)nchunks;
  c->nsmall = (uint32_t)nsmall;
  Tmp<Chunk> chunks_unsorted;
  if ((rc = dev_alloc(c, c->tiles, (size_t)ntiles))) return rc;
  if ((rc = dev_alloc(c, chunks_unsorted.b, (size_t)nchunks))) return rc;
  if ((rc = dev_alloc(c, c->chunks, (size_t)nchunks))) return rc;
  if ((rc = dev_alloc(c, c->tile_refs, (size_t)nlist))) return rc;
  if ((rc = dev_alloc(c, c->small_tiles, (size_t)nsmall))) return rc;
  /* + 9 rows: verify_candidate reads nine dwords per query whatever its length */
  if ((rc = dev_alloc(c, c->qres, (size_t)res_words + 9 * WAVE))) return rc;
  if ((rc = dev_alloc(c, c->qrep, slots))) return rc;
  if ((rc = dev_alloc(c, c->qlen, slots))) return rc;
  if ((rc = dev_alloc(c, c->qorig, slots))) return rc;
  if ((rc = dev_alloc(c, c->qrec, slots))) return rc;
  HIP_TRY(c, hipMemsetAsync(c->qrec.p, 0, std::max<size_t>(slots, 1) * sizeof(QueryRec), c->stream));
  HIP_TRY(c, hipMemsetAsync(c->qres.p, 0, ((size_t)res_words + 9 * WAVE) * sizeof(uint32_t), c->stream));
  HIP_TRY(c, hipMemsetAsync(c->qrep.p, 0, std::max<size_t>(slots, 1) * sizeof(uint32_t), c->stream));
  HIP_TRY(c, hipMemsetAsync(c->qlen.p, 0, std::max<size_t>(slots, 1) * sizeof(uint16_t), c->stream));
  HIP_TRY(c, hipMemsetAsync(c->qorig.p, 0, std::max<size_t>(slots, 1) * sizeof(uint32_t), c->stream));
  c->qv.release(); c->qj.release(); c->qgh.release(); c->qcnt.release(); c->qck.release();
  c->qhins.release(); c->qhdel.release(); c->cw.release(); c->cmain.release(); c->crp.release(); c->cpk.release(); c->slice_items.release();
  if (!c->opt.ignore_genes) {
    if ((rc = dev_alloc(c, c->qv, slots))) return rc;
    if ((rc = dev_alloc(c, c->qj, slots))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->qv.p, 0, std::max<size_t>(slots, 1) * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->qj.p, 0, std::max<size_t>(slots, 1) * sizeof(uint32_t), c->stream));
  }
  if (!c->opt.ignore_genes || c->rows) {
    if ((rc = dev_alloc(c, c->qgh, slots))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->qgh.p, 0, std::max<size_t>(slots, 1) * sizeof(uint64_t), c->stream));
  }
  if (!c->opt.ignore_counts) {
    if ((rc = dev_alloc(c, c->qcnt, slots))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->qcnt.p, 0, std::max<size_t>(slots, 1) * sizeof(uint64_t), c->stream));
  }
  if (c->rows) {
    if ((rc = dev_alloc(c, c->qck, slots))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->qck.p, 0, std::max<size_t>(slots, 1) * sizeof(uint32_t), c->stream));
    if (c->opt.indels) {
      if ((rc = dev_alloc(c, c->qhins, slots))) return rc;
      if ((rc = dev_alloc(c, c->qhdel, slots))) return rc;
      HIP_TRY(c, hipMemsetAsync(c->qhins.p, 0, std::max<size_t>(slots, 1) * sizeof(uint64_t), c->stream));
      HIP_TRY(c, hipMemsetAsync(c->qhdel.p, 0, std::max<size_t>(slots, 1) * sizeof(uint64_t), c->stream));
    }
  }
  if (ngroups) {
    /* (+ 64: a block read past the last item stays inside) */
    const size_t ni = (size_t)cslots + WAVE;
    if ((rc = dev_alloc(c, c->cw, ni))) return rc;
    if ((rc = dev_alloc(c, c->cmain, ni))) return rc;
    if ((rc = dev_alloc(c, c->crp, ni))) return rc;
    if (sub2_items && (rc = dev_alloc(c, c->cpk, ni))) return rc;
    if (sub2_items) {
      if ((rc = dev_alloc(c, c->slice_items, 2 * (size_t)nslices))) return rc;
      HIP_TRY(c, hipMemsetAsync(c->slice_items.p, 0, 2 * (size_t)nslices * sizeof(uint32_t), c->stream));
    }
    HIP_TRY(c, hipMemsetAsync(c->cw.p, 0, ni * sizeof(uint64_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->cmain.p, 0xff, ni * sizeof(uint32_t), c->stream));   /* ~0: padding */
    HIP_TRY(c, hipMemsetAsync(c->crp.p, 0, ni * sizeof(uint32_t), c->stream));
  }
  Q.tiles = c->tiles.p;
  Q.chunks = chunks_unsorted.b.p;
  Q.tile_refs = c->tile_refs.p;
  Q.small_tiles = c->small_tiles.p;
  Q.cpk = c->cpk.p;
  Q.slice_items = (uint2 *)c->slice_items.p;
  Q.qres = c->qres.p; Q.qv = c->qv.p; Q.qj = c->qj.p; Q.qrep = c->qrep.p;
  Q.qorig = c->qorig.p; Q.qck = c->qck.p; Q.qgh = c->qgh.p; Q.qhins = c->qhins.p;
  Q.qhdel = c->qhdel.p; Q.qcnt = c->qcnt.p; Q.qlen = c->qlen.p; Q.qrec = c->qrec.p;
  Q.cw = c->cw.p; Q.cmain = c->cmain.p; Q.crp = c->crp.p;

  for (uint32_t pi = 0; pi < npass; pi++) {
    hipLaunchKernelGGL(slices_kernel<1>, dim3(blocks_for(nslices)), dim3(256), 0, c->stream, Q, pi);
    HIP_TRY(c, hipGetLastError());
  }
  if (s->n) {
    hipLaunchKernelGGL(place_kernel, dim3(blocks_for(s->n)), dim3(256), 0, c->stream, Q);
    HIP_TRY(c, hipGetLastError());
    if (ngroups) {
      hipLaunchKernelGGL(place_items_kernel, dim3(blocks_for(s->n)), dim3(256), 0, c->stream, Q);
      HIP_TRY(c, hipGetLastError());
    }
  }
  if (ngroups) {
    hipLaunchKernelGGL(class_chunks_kernel, dim3(blocks_for(ncs)), dim3(256), 0, c->stream, Q);
    HIP_TRY(c, hipGetLastError());
  }
  if (indel_passes) {
    hipLaunchKernelGGL(sibling_fill_kernel, dim3(blocks_for(G)), dim3(256), 0, c->stream, Q, G,
                       sib_lpre.b.p, sib_fill.b.p, (uint32_t)main_list);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(sibling_write_chunks_kernel, dim3(blocks_for(2 * nslices)), dim3(256), 0,
                       c->stream, Q, sib_cnt.b.p, sib_lpre.b.p, sib_cpre.b.p, (uint32_t)main_list,
                       (uint32_t)main_chunks);
    HIP_TRY(c, hipGetLastError());
  }

  const uint32_t wstep = (uint32_t)c->work_shard_count;
  const uint32_t wfirst = (uint32_t)c->work_shard_index;
  if (wfirst >= wstep)
    return fail(c, CMPR_EINVAL, "work_shard_index must be below work_shard_count");
  if (wstep > 1 && !c->sliced)
    return fail(c, CMPR_EUNSUPPORTED, "work shards need a sliced layout (kernel variant 1 or 2)");
  /* ---- heaviest chunks first: the tail of the launch is made of light ones;
          single-wave tiles longest first ---- */
  if (nchunks) {
    Tmp<uint32_t> wk, wk2, ix, ix2;
    if ((rc = dev_alloc(c, wk.b, (size_t)nchunks))) return rc;
    if ((rc = dev_alloc(c, wk2.b, (size_t)nchunks))) return rc;
    if ((rc = dev_alloc(c, ix.b, (size_t)nchunks))) return rc;
    if ((rc = dev_alloc(c, ix2.b, (size_t)nchunks))) return rc;
    hipLaunchKernelGGL(chunk_work_kernel, dim3(blocks_for(nchunks)), dim3(256), 0, c->stream, Q,
                       (uint32_t)nchunks, wk.b.p, ix.b.p);
    HIP_TRY(c, hipGetLastError());
    size_t sb = 0;
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, sb, wk.b.p, wk2.b.p, ix.b.p, ix2.b.p,
                                                 (int)nchunks, 0, 32, c->stream);
    Tmp<char> st;
    if ((rc = dev_alloc(c, st.b, sb))) return rc;
    HIP_TRY(c, hipcub::DeviceRadixSort::SortPairsDescending(st.b.p, sb, wk.b.p, wk2.b.p, ix.b.p,
                                                            ix2.b.p, (int)nchunks, 0, 32, c->stream));
    uint64_t mine = nchunks;
    const uint32_t *order = ix2.b.p;
    Tmp<uint32_t> sel;
    if (wstep > 1) {
      /* the context's share of the sorted list */
      if ((rc = select_mine(c, ix2.b.p, (uint32_t)nchunks, sel, mine, [&](unsigned char *flag) {
            hipLaunchKernelGGL(chunk_mine_kernel, dim3(blocks_for(nchunks)), dim3(256), 0, c->stream,
                               chunks_unsorted.b.p, ix2.b.p, (uint32_t)nchunks, wfirst, wstep, flag);
          })))
        return rc;
      order = sel.b.p;
    }
    if (mine)
      hipLaunchKernelGGL(gather_chunks_kernel, dim3(blocks_for(mine)), dim3(256), 0, c->stream,
                         chunks_unsorted.b.p, order, (uint32_t)mine, c->chunks.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->nchunks = (uint32_t)mine;
  }
  if (nsmall) {
    Tmp<uint32_t> ln, ln2, out;
    if ((rc = dev_alloc(c, ln.b, (size_t)nsmall))) return rc;
    if ((rc = dev_alloc(c, ln2.b, (size_t)nsmall))) return rc;
    if ((rc = dev_alloc(c, out.b, (size_t)nsmall))) return rc;
    hipLaunchKernelGGL(small_len_kernel, dim3(blocks_for(nsmall)), dim3(256), 0, c->stream, Q,
                       (uint32_t)nsmall, ln.b.p);
    HIP_TRY(c, hipGetLastError());
    size_t sb = 0;
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, sb, ln.b.p, ln2.b.p, c->small_tiles.p,
                                                 out.b.p, (int)nsmall, 0, 17, c->stream);
    Tmp<char> st;
    if ((rc = dev_alloc(c, st.b, sb))) return rc;
    HIP_TRY(c, hipcub::DeviceRadixSort::SortPairsDescending(st.b.p, sb, ln.b.p, ln2.b.p,
                                                            c->small_tiles.p, out.b.p, (int)nsmall,
                                                            0, 17, c->stream));
    uint64_t mine = nsmall;
    const uint32_t *keep = out.b.p;
    Tmp<uint32_t> sel;
    if (wstep > 1) {
      if ((rc = select_mine(c, out.b.p, (uint32_t)nsmall, sel, mine, [&](unsigned char *flag) {
            hipLaunchKernelGGL(small_mine_kernel, dim3(blocks_for(nsmall)), dim3(256), 0, c->stream, Q,
                               out.b.p, (uint32_t)nsmall, wfirst, wstep, flag);
          })))
        return rc;
      keep = sel.b.p;
    }
    if (mine)
      HIP_TRY(c, hipMemcpyAsync(c->small_tiles.p, keep, (size_t)mine * sizeof(uint32_t),
                                hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->nsmall = (uint32_t)mine;
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return CMPR_OK;
}
