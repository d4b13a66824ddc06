/*
 * query_layout.hip -- cmpr_set_queries on the device.
 *
 * The reference hands its per-query loop the parsed set as it lies in memory
 * (overlap.cc:906-938, db accessors db.cc:964-997).  The kernels here want the
 * queries grouped by filter slice and cut into 64-query tiles (layout.h); this
 * file does that regrouping with kernels, so that from cmpr_set_view to "resident
 * in HBM" the host only copies the caller's arrays:
 *
 *   (upload)            the caller's arrays in a few query ranges on a copy stream;
 *                       the keys kernel of a range runs while the next is copied
 *   keys_kernel         per query: what scan_view checked on the host (monotone
 *                       offsets, ranges of residue / gene / repertoire numbers,
 *                       counts >= 1, per-repertoire count totals), then class key ->
 *                       slice -> group (slice, heavy, length), Zobrist hash (db_hash,
 *                       db.cc:903-916, variant 2), and its rank in the group -- the
 *                       return value of the histogram's atomic
 *   slices_kernel<0>    per slice: tiles / chunks / residue words it needs
 *   (scan)              hipCUB exclusive scan over the slices
 *   slices_kernel<1>    per slice: tile descriptors, first slot of every group,
 *                       chunk descriptors
 *   scatter_kernel      per query: ONE 64-byte record (layout.h QueryRec) + 32 bytes of
 *                       hashes to slot = group base + rank: whole memory lines, the only
 *                       scattered writes of the layout
 *   fill_tiles_kernel   per tile (one wave): reads those records in slot order and
 *                       writes every per-slot array and the position-major residues
 *                       coalesced, padding lanes included (no memset of the layout)
 *                       ... and, variant 2 / sub2, its flat items (kernels_rows.h), 16 bytes each
 *   sibling_*_kernel    -i: tiles regrouped by the slice their insertion /
 *                       deletion variants fall into
 *   chunk order         heaviest chunks first (hipCUB radix sort)
 *
 * Inside a group the queries land in the order their atomics complete, not in
 * input order: the matrix is a sum of exact integers, the pairs list is
 * unordered (README.md:163), so no result depends on it.
 *
 * Work shards (tunables work_shard_index / _count; one context per GPU): a context
 * lays out only what it works on -- the queries whose slice it owns, the items filed
 * under slices it owns, and, in tiles of a pseudo-slice that no chunk lists, the
 * queries those items belong to (their records are what a verification reads).
 *
 * Temporaries live in two arenas the context keeps from call to call, the resident
 * arrays are reallocated only when they have to grow, and the host waits for the
 * device twice: for the sizes of the layout, and at the end.
 */
#include "context.h"
#include "kernels_sliced.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

using namespace cmpr;

namespace {

constexpr uint32_t MAXP = 1 + MAX_CLASS_RES;     /* passes with a layout of their own */
/* work shards: queries laid out only because an item of theirs is worked on here live in
   tiles of 2^FOREIGN_SLICES_LOG2 pseudo-slices behind the real ones (no chunk lists them) */
constexpr uint32_t FOREIGN_SLICES_LOG2 = 12;

enum : uint32_t { VERR_OFFSETS = 1, VERR_LONG = 2, VERR_REP = 3, VERR_GENE = 4, VERR_COUNT = 5,
                  VERR_RESIDUE = 6, VERR_TOO_LONG = 7, VERR_OFFSETS0 = 8 };

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

/* per slot, next to its QueryRec: hashes and class key (scatter_kernel -> fill_tiles_kernel) */
struct QAux {
  uint64_t h, hins, hdel;
  uint32_t ck;
  uint32_t src;                      /* the query's number in the arrays at hand (QL::res / off ...) */
};

struct QL {
  /* the caller's set, uploaded as it is */
  const uint8_t  *res;
  const uint64_t *off;
  const uint32_t *v, *j, *rep;
  const uint64_t *cnt;
  const uint32_t *orig;              /* routed records: the query's number in the caller's WHOLE set (NULL: i) */
  uint64_t        n, total;          /* sequences; residues in all (offsets[n]) */
  uint32_t        n_rep, n_v_max, n_j_max;
  uint32_t        A, zpos, n_v, longest, per_slice;   /* longest: the longest a query may be (zpos - 3) */
  uint32_t        genes, counts, existence, indels, differences, sliced, rows;
  uint32_t        pairs;             /* variant 2: pair rows (kernels_rows.h) */
  uint32_t        pairs2;            /* ... probed by kernels_pairs2.h (nucleotides, d = 2): residues packed per slot
                                        (qpk) and per item (cpk) */
  cmpr::ResPack  *qpk;
  const uint64_t *zob;
  SliceGeom       geom;
  uint32_t        npass;             /* 1 + class-row passes (variant 2) */
  uint32_t        min_mixed;         /* lengths >= this share tiles; ~0: none do */
  uint32_t        chunk_tiles, small_max, class_unstaged;
  uint64_t        nbuckets;          /* (slice, heavy) buckets */
  uint64_t        nslices;           /* slices laid out as tiles: the real ones + the foreign pseudo-slices */
  uint64_t        nslices_real;
  uint32_t        wfirst, wstep;     /* this context places what work shard wfirst of wstep works on (1: all) */
  uint32_t        direct;            /* d = 0 on the un-sliced kernel: the query's hash is computed here (qgh), the
                                        queries are grouped by pseudo-slice (bits of the hash & pmask) and length */
  uint32_t        rec_hash;          /* record tiles, every sequence within 28 residues: scatter_kernel leaves the
                                        query's Zobrist hash in the record's last two residue words (bytes 28 .. 35) */
  uint32_t        pmask;
  uint32_t        dbg;               /* -DCMPR_ABLATION builds: LDBG_* (timing experiments, results become wrong) */
  uint32_t        route;             /* 1: cmpr_route_queries -- the keys kernel names the contexts a query goes
                                        to (dest_lo / dest_hi, dest_cnt) instead of ranking it in its group */
  uint32_t        alg_step, alg_first; /* routed records (alg_step > 1): the algorithmic bytes of the queries whose
                                          slice context alg_first of alg_step works on */
  unsigned long long *dest_cnt;      /* route: records per destination */
  /* per pass */
  uint32_t *cnt_g[MAXP], *base_g[MAXP], *grp[MAXP];
  uint32_t *rank;                    /* per query: its rank in its group */
  SliceTot *tot[MAXP], *pre[MAXP];
  uint32_t  tile0[MAXP], chunk0[MAXP], list0[MAXP], small0[MAXP];
  unsigned long long res0[MAXP];
  uint32_t *tfirst_g;                /* pass 0: first tile of every group */
  /* per query */
  uint64_t *h_tmp, *hins_tmp, *hdel_tmp;
  uint32_t *ck_tmp;
  struct QAux *aux;                  /* per slot: what the record has no room for */
  /* outputs */
  TileDesc *tiles;
  Chunk    *chunks;
  TileRef  *tile_refs;
  uint32_t *small_tiles, *chunk_work;
  uint32_t *qres, *qv, *qj, *qck;
  uint64_t *qgh, *qhins, *qhdel;
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
  /* The item counters are few (one per slice of a class part: 2048 for 10M amino-acid queries = 64
     lines of memory) and every query adds to one of them: 10^7 atomics on 64 lines took 0.7 ms in
     keys_kernel and again when the items were placed.  With item_reps > 1 copies, batch (i >> 8) of the
     queries counts in copy (i >> 8) mod item_reps; item_replicas_kernel sums the copies into ccnt
     and gives each copy its share of the (class part, slice)'s items (rbase). */
  uint32_t *ccnt_r, *cfill_r, *rbase;               /* [item_reps][nitem_slices] */
  uint32_t  item_reps;
  /* item_wg (round 6; few counters -- <= ITEM_WG_MAX --, no routing): the items are counted per WORKGROUP in
     LDS, nothing per query goes to memory.  keys_kernel and scatter_kernel are launched with the same grid
     over the same query range, so workgroup w sees the same queries in both: keys_kernel counts w's items per
     counter in LDS and claims a run for them with ONE answered atomic per non-empty counter (on padded
     counters, one per 64 bytes: ccnt_r is then [nitem_slices][ITEM_PAD]), leaving the run's start in
     rbase[row][k] (row = row0 of the launch + w); scatter_kernel loads its row into LDS and hands out the
     places with LDS atomics.  (Round 5: 32 replicas of the counters in memory, one atomic per query in each
     of the two kernels: 0.30 ms of keys_kernel's 1.23 and an answered atomic + two reads of scatter_kernel's
     seven scattered operations per query.) */
  uint32_t  item_wg;
  uint32_t  recompute;                    /* fill_tiles_kernel works the hashes and the class key out from the
                                             record (no QAux is scattered): every query fits its record */
  uint32_t  total_on_device;              /* the residue total is off[n], read where the offsets lie */
  uint32_t  zob_lds, zob_words;           /* keys_kernel keeps the Zobrist keys (zob_words of them, gene keys
                                             included) in LDS; scatter_kernel those of the positions */
  uint32_t  zob_pos_words;
  uint32_t  own_np;                       /* 16-byte pieces of a thread's own copy of its residues (own_load): 3 .. 7 */
  uint32_t  ctab_lds;                     /* ... and both the class tables in front of the heavy bitmap
                                             (geom.off_hv words) */
  /* group_wg: few (slice, length) groups -- the direct layout of d = 0, whose groups are pseudo-slices x lengths,
     a thousand counters that every query's answered atomic met in memory (0.29 of the 0.5 ms a 1M-query set
     took from device arrays to the matrix) --: ranked per workgroup in LDS like the items (item_wg): keys_kernel
     ranks a query among its workgroup's queries of the group and claims a run of the group for them once,
     gbase_w[row][g]; scatter_kernel adds it to the group's base. */
  uint32_t  group_wg, G;
  uint32_t *gbase_w;
  uint32_t  cchunk0;                      /* first item chunk in the chunk list */
  cmpr::ItemRec *items;
  cmpr::ResPack *cpk;
  /* validation / statistics */
  uint32_t           *verr;          /* [0] first error kind, [1] longest */
  double             *rep_total;
  unsigned long long *alg_bytes;
};

constexpr uint32_t ITEM_WG_MAX = 16384;      /* counters a workgroup keeps in LDS (at most 64 KiB) */
constexpr uint32_t ITEM_PAD = 16;            /* ... and claims runs from: one counter per 64 bytes */
constexpr uint32_t GROUP_WG_MAX = 8192;      /* (slice, length) groups a workgroup ranks its queries in, in LDS */

/* keys_kernel with parts left out (the "debug" tunable's bits 16.., -DCMPR_ABLATION builds only) */
enum : uint32_t { LDBG_NO_RANK = 1u << 16, LDBG_NO_ITEM_COUNT = 1u << 17, LDBG_NO_HASH = 1u << 18,
                  LDBG_NO_TOTALS = 1u << 19, LDBG_NO_CLASSKEY = 1u << 20, LDBG_NO_TMP_WRITES = 1u << 21,
                  /* scatter_kernel: no record written | no items at all | the record to slot i (sequential) |
                     items worked out, not written | no group base read (slot = i) */
                  LDBG_S_NO_REC = 1u << 22, LDBG_S_NO_ITEMS = 1u << 23, LDBG_S_SEQ_REC = 1u << 24,
                  LDBG_S_NO_ITEM_WRITE = 1u << 25, LDBG_S_NO_BASE = 1u << 26 };
#ifdef CMPR_ABLATION
#define LDBG(Q, bit) (((Q).dbg & (bit)) != 0)
#else
#define LDBG(Q, bit) false
#endif

/* the residues of the set: offsets[n] (read on the device when the host never saw the offsets) */
__device__ inline uint64_t total_of(const QL &Q)
{
  return Q.total_on_device ? Q.off[Q.n] : Q.total;
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

/* the direct layout of d = 0 (QL::direct): a query's work shard is named by these bits of its pseudo-slice key */
constexpr uint32_t DIRECT_OWNER_MASK = 1023u;

/* does this context work on what is filed under `slice` in `pass`? */
__device__ inline bool owned(const QL &Q, uint32_t slice, uint32_t pass)
{
  return Q.wstep <= 1u || work_owner(slice, pass, Q.wstep) == Q.wfirst;
}

/* the context that works on the items counted by counter k (see QL: item groups) */
__device__ inline uint32_t item_owner(const QL &Q, uint32_t k, uint32_t step)
{
  if (Q.sub2_items)            /* (ride along with the slice's main chunk, or chunks of pass 3: one owner) */
    return work_owner(k / Q.ngroups, 0u, step);
  uint32_t gi = 0;
  for (uint32_t x = 1; x < Q.ngroups; x++)
    if (k >= Q.goff[x])
      gi = x;
  return work_owner(Q.gslice0[gi] + (k - Q.goff[gi]), 3u + gi, step);
}

/* ... on the items counted by counter k? */
__device__ inline bool item_owned(const QL &Q, uint32_t k)
{
  return Q.wstep <= 1u || item_owner(Q, k, Q.wstep) == Q.wfirst;
}

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
  /* (every gridDim.x-th batch of 256: see keys_kernel) */
  uint32_t err = 0, L = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < Q.n && !err; i += (uint64_t)gridDim.x * 256) {
    const uint64_t a = Q.off[i], b = Q.off[i + 1];
    if (b < a)
      err = VERR_OFFSETS;
    else if (b - a > 0xffffu)
      err = VERR_LONG;
    else
      L = max(L, (uint32_t)(b - a));
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

/* A thread's OWN copy of its query's residues (round 6): the up to three aligned 16-byte pieces that hold
   them go to the thread's words of LDS (OWN_DW each: an odd stride, so that the lanes' byte reads fall on
   different banks) and are read there a byte at a time -- no barrier, no load that another thread's
   address depends on: a wave runs on as soon as ITS loads are back.  (Rounds 4-5 staged a batch's residues
   for the whole workgroup: two barriers and two dependent scalar loads per batch of
   256 queries.)  Sequences that do not fit the three pieces, or a set whose residues are not 16-byte
   aligned: read where they lie.  The caller has checked b <= e <= total. */
constexpr uint32_t OWN_DW = 13;           /* three pieces + 1: the CDR3 amino-acid modes */
constexpr int OWN_NP_MAX = 7;                /* the general mode: sequences of up to 97 residues in a thread's LDS words */
/* (threads of a workgroup of keys_kernel / scatter_kernel: 256, 512 or 1024 -- the more share one copy of the
   tables (Zobrist keys, class tables, item counters: 22 KiB at 10M queries, 54 KiB with -i), the more waves
   fit a CU beside them; chosen per call, cmpr_layout_queries) */
/* ... in two steps, so that the caller can ask for the NEXT query's fields between them: the loads of the
   pieces (own_load), and their way into LDS (own_commit: waits for them) */
/* (NP: the pieces an instantiation has registers for -- 3 for the CDR3 amino-acid modes, 7 for the general one;
   np <= NP: the pieces the set's longest sequence needs, QL::own_np, and 4 np + 1 words of LDS per thread) */
template <int NP>
struct OwnPieces {
  uint4 w[NP];
  bool  fits;
};

template <int NP>
__device__ inline OwnPieces<NP> own_load(const QL &Q, uint64_t b, uint64_t e, uint64_t total, uint32_t np)
{
  OwnPieces<NP> o;
  const uint64_t a0 = b & ~15ull;
  o.fits = ((uintptr_t)Q.res & 15u) == 0 && e - a0 <= 16u * np;
#pragma unroll
  for (uint32_t k = 0; k < (uint32_t)NP; k++) {
    o.w[k] = make_uint4(0u, 0u, 0u, 0u);
    const uint64_t at = a0 + 16u * k;
    if (k < np && o.fits && at < e) {
      if (at + 16u <= total) {
        o.w[k] = *(const uint4 *)(Q.res + at);
      } else {                                   /* (the last bytes of the set) */
        uint64_t lo = 0, hi = 0;                 /* (no array: an index the compiler cannot see is scratch) */
        for (uint32_t x = 0; x < 16u && at + x < total; x++) {
          const uint64_t r = Q.res[at + x];
          if (x < 8u)
            lo |= r << (8u * x);
          else
            hi |= r << (8u * (x - 8u));
        }
        o.w[k] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
      }
    }
  }
  return o;
}

template <int NP>
__device__ inline const uint8_t *own_commit(const QL &Q, const OwnPieces<NP> &o, uint64_t b, uint64_t e, uint32_t *mine)
{
  if (!o.fits)
    return Q.res + b;
  const uint64_t a0 = b & ~15ull;
#pragma unroll
  for (uint32_t k = 0; k < (uint32_t)NP; k++)
    if (a0 + 16u * k < e) {
      mine[4 * k + 0] = o.w[k].x;
      mine[4 * k + 1] = o.w[k].y;
      mine[4 * k + 2] = o.w[k].z;
      mine[4 * k + 3] = o.w[k].w;
    }
  return (const uint8_t *)mine + (b - a0);
}

/* zobrist_hash (zobrist.cc:74-88) and, with -i, the two shifted hashes that seed the rolling indel
   enumeration (:90-104, :122-136); `z` = the Zobrist keys, in memory or the workgroup's copy in LDS */
template <typename ZP>
__device__ inline void hashes_of(ZP z, uint32_t A, bool indels, const uint8_t *s, uint32_t L, uint64_t &h,
                                 uint64_t &hins, uint64_t &hdel)
{
  for (uint32_t p = 0; p < L; p++) {
    const uint32_t r = s[p];
    h ^= z[A * p + r];
    if (indels) {
      hins ^= z[A * (p + 1) + r];
      if (p > 0)
        hdel ^= z[A * (p - 1) + r];
    }
  }
}

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
enum : uint32_t { ITEM_SUB = K_SUB, ITEM_INS = K_INS, ITEM_SUB2 = K_SUB2, ITEM_PAIR = 7 /* kernels_rows.h K_PAIR */ };
/* flag of an ITEM_SUB item, above its kind: the row's deletion answer counts (layout.h ITEM_DEL_COUNTS) */

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
__device__ inline void for_each_item(const QL &Q, const uint32_t *ct, const uint64_t *zt, uint64_t h_of_q,
                                     uint64_t hins_of_q, const uint8_t *s, uint32_t L, uint32_t ck, bool heavy, F f)
{
  /* (h_of_q, hins_of_q: the query's hash and, with -i, its insert-first hash -- asked only with HASH) */
  /* (s: the query's residues -- where the set lies, or the thread's copy in LDS; ct, zt: the class tables in
     front of the heavy bitmap and the Zobrist keys -- where they lie, or the workgroup's copies in LDS: a
     lookup in memory with 64 different addresses per wave is the dear kind, round 6) */
  const SliceGeom &g = Q.geom;
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
          const uint32_t key = ck ^ ct[g.off_cr + ci * A + r] ^ ct[g.off_cr + ci * A + v];
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
  if (Q.pairs) {
    /* pair rows (kernels_rows.h): one item per PAIR of positions (p, p + 1), p even, that
       holds a class position -- class part of the first class residue inside, key without
       the terms of every class residue inside, the hash with both positions blanked.  With
       -i the pair also answers "q without p" / "q without p + 1": the flags say which of the
       two deletion variants exist (the first position of a run of equal residues,
       variants.cc:301-325; none for a query of one residue). */
    if (heavy) {
      const uint64_t h = HASH ? h_of_q : 0ull;
      for (uint32_t ci = 0; ci < K; ci++) {
        const uint32_t p = class_pos(L, ci, g.c0) & ~1u;
        bool first = true;
        uint32_t key = ck;
        for (uint32_t k = 0; k < K; k++) {
          const uint32_t mk = class_pos(L, k, g.c0);
          if ((mk & ~1u) == p) {
            if (k < ci)
              first = false;
            key ^= ct[g.off_cr + k * A + s[mk]];
          }
        }
        if (first) {
          const uint32_t ra = s[p], rb = p + 1 < L ? (uint32_t)s[p + 1] : A;
          uint64_t w = 0;
          if (HASH) {
            w = h ^ zt[A * p + ra];
            if (p + 1 < L)
              w ^= zt[A * (p + 1) + rb];
          }
          uint32_t fl = 0;
          if (Q.indels && L > 1) {
            if (p == 0 || s[p] != s[p - 1])
              fl |= ITEM_DEL_COUNTS;
            if (p + 1 < L && s[p + 1] != s[p])
              fl |= ITEM_DEL2_COUNTS;
          }
          key ^= pair_part_terms(ct, g, A, L, p, [&](uint32_t pos) -> uint32_t { return s[pos]; });
          f(Q.goff[ci] + (key & g.cmask), w, ra | (rb << 5) | (p << 10) | (ITEM_PAIR << 24) | fl);
        }
      }
    }
    if (!Q.indels)
      return;
    /* -i, insertion pairs: the variants t = q with v in front of g (pair (v, q[g]) at (g,
       g + 1)) and t' = q with w in front of g + 1 (pair (q[g], w)) share the word under the
       hash of "q with a gap at g and q[g] blanked"; the pair is an item when it holds a class
       position of the variants (length L + 1) and their class is split -- filed under the
       class residues of the variants outside the pair, taken around the gap */
    uint32_t base = ck;
    if (heavy)
      for (uint32_t k = 0; k < K; k++)
        base ^= ct[g.off_cr + k * A + s[class_pos(L, k, g.c0)]];
    const uint32_t base_i = base ^ ct[L] ^ ct[L + 1];
    if (!class_is_heavy(g.ctab, g, base_i))
      return;
    const uint64_t hins = HASH ? hins_of_q : 0ull;
    uint64_t P0 = 0, Pp = 0;                     /* XOR_{y<x} Z[y][q[y]],  XOR_{y<x} Z[y+1][q[y]] */
    /* (the pairs that hold a class position of the variants, as a bit set: every other pair is passed by
       without a table lookup -- the loop asked the class tables K times per pair of every query, round 6) */
    uint64_t cand = 0;
    bool cand_far = false;
    for (uint32_t k = 0; k < K; k++) {
      const uint32_t mk = class_pos(L + 1, k, g.c0);
      if (mk < 64u)
        cand |= 1ull << (mk & ~1u);
      else
        cand_far = true;
    }
    for (uint32_t x = 0; x <= L; x++) {
      if ((x & 1u) == 0 && (x < 64u ? ((cand >> x) & 1ull) != 0 : cand_far)) {
        int ci = -1;
        uint32_t key = base_i;
        for (uint32_t k = 0; k < K; k++) {
          const uint32_t mk = class_pos(L + 1, k, g.c0);
          if ((mk & ~1u) == x) {
            if (ci < 0)
              ci = (int)k;
          } else {
            key ^= ct[g.off_cr + k * A + s[mk < x ? mk : mk - 1]];
          }
        }
        if (ci >= 0) {
          key ^= pair_part_terms(ct, g, A, L + 1, x,
                                 [&](uint32_t pos) -> uint32_t { return s[pos < x ? pos : pos - 1]; });
          uint64_t w = P0 ^ hins ^ Pp;           /* gap at x */
          if (HASH && x < L)
            w ^= zt[A * (x + 1) + s[x]];      /* ... and q[x], one position up, blanked */
          f(Q.goff[ci] + (key & g.cmask), w,
            (x < L ? (uint32_t)s[x] : A) | ((x > 0 ? (uint32_t)s[x - 1] : 31u) << 5) | (x << 10) | (ITEM_INS << 24));
        }
      }
      if (x == L)
        break;
      if (HASH) {
        P0 ^= zt[A * x + s[x]];
        Pp ^= zt[A * (x + 1) + s[x]];
      }
    }
    return;
  }
  if (heavy) {
    const uint64_t h = HASH ? h_of_q : 0ull;
    for (uint32_t ci = 0; ci < K; ci++) {
      const uint32_t pos = class_pos(L, ci, g.c0);
      bool first = true;
      uint32_t key = ck;
      for (uint32_t k = 0; k < K; k++)
        if (class_pos(L, k, g.c0) == pos) {
          if (k < ci)
            first = false;
          key ^= ct[g.off_cr + k * A + s[pos]];
        }
      /* (single rows: d = 2 -- with -i, which goes with d = 1 only, the filter holds pair rows) */
      if (first)
        f(Q.goff[ci] + (key & g.cmask), HASH ? h ^ zt[A * pos + s[pos]] : 0ull,
          (uint32_t)s[pos] | (pos << 8) | (ITEM_SUB << 24));
    }
  }
}

/* One thread per query of [q0, q1): first what a host pass over the set would check,
   then -- for a sound query -- its keys. */
/* What an instantiation of keys_kernel / scatter_kernel may take for granted about the flags of QL (the host
   picks the instantiation by them): MODE 1 = amino acids or nucleotides on pair rows without -i (the headline
   workload), 2 = pair rows with -i, 3 = any layout of sequences that fit three 16-byte pieces (CDR3 amino acids:
   d = 0, d = 2, work shards), 0 = anything.  The branches of the other layouts -- the insertion pairs'
   rolling hashes, residue packs, routing -- then cost no registers: scatter_kernel needs 80 for six waves per
   SIMD and took 90 with every path compiled in. */
template <int MODE>
__device__ inline void layout_mode_assume(const QL &Q)
{
  if (MODE == 3) {                           /* any layout, sequences that fit three 16-byte pieces */
    __builtin_assume(Q.own_np == 3);
    return;
  }
  if (MODE != 0) {
    __builtin_assume(Q.rows != 0);
    __builtin_assume(Q.pairs != 0);
    __builtin_assume(Q.sliced != 0);
    __builtin_assume(Q.pairs2 == 0);
    __builtin_assume(Q.sub2_items == 0);
    __builtin_assume(Q.direct == 0);
    __builtin_assume(Q.route == 0);
    __builtin_assume(Q.differences == 1);
    __builtin_assume(Q.recompute != 0);
    __builtin_assume(Q.wstep <= 1);
    __builtin_assume(Q.dbg == 0);
    __builtin_assume(Q.group_wg == 0);
    __builtin_assume(Q.own_np == 3);
  }
  if (MODE == 1)
    __builtin_assume(Q.indels == 0);
  if (MODE == 2)
    __builtin_assume(Q.indels != 0);
}

template <uint32_t LAYOUT_WG, int MODE>
__global__ void __launch_bounds__(LAYOUT_WG, (LAYOUT_WG == 1024 || MODE == 0) ? 4 : 6)
keys_kernel(const QL Q, uint64_t q0, uint64_t q1, uint32_t row0)
{
  layout_mode_assume<MODE>(Q);
  /* dynamic LDS: n_rep doubles when they fit | the Zobrist keys when they fit (zob_lds) | (item_wg) the item
     counters | OWN_DW words per thread */
  extern __shared__ double tot_lds[];
  __shared__ uint32_t dest_lds[64];            /* route: this workgroup's records per destination */
  const bool lds_tot = Q.n_rep <= 2048;
  uint64_t *const zl = (uint64_t *)(tot_lds + (lds_tot ? Q.n_rep : 0u));
  uint32_t *const ctl = (uint32_t *)(zl + (Q.zob_lds ? Q.zob_words : 0u));
  uint32_t *const ihist = ctl + (Q.ctab_lds ? Q.geom.off_hv : 0u);
  uint32_t *const ghist = ihist + (Q.item_wg ? Q.nitem_slices : 0u);
  constexpr int NP = MODE == 0 ? OWN_NP_MAX : 3;
  const uint32_t own_np = MODE == 0 ? Q.own_np : 3u, own_dw = 4u * own_np + 1u;
  uint32_t *const mine = ghist + (Q.group_wg ? Q.G : 0u) + threadIdx.x * own_dw;
  const uint32_t *const ct = Q.ctab_lds ? ctl : Q.geom.ctab;
  if (lds_tot) {
    for (uint32_t r = threadIdx.x; r < Q.n_rep; r += LAYOUT_WG)
      tot_lds[r] = 0.0;
  }
  if (Q.zob_lds)
    for (uint32_t k = threadIdx.x; k < Q.zob_words; k += LAYOUT_WG)
      zl[k] = Q.zob[k];
  if (Q.ctab_lds)
    for (uint32_t k = threadIdx.x; k < Q.geom.off_hv; k += LAYOUT_WG)
      ctl[k] = Q.geom.ctab[k];
  if (Q.item_wg)
    for (uint32_t k = threadIdx.x; k < Q.nitem_slices; k += LAYOUT_WG)
      ihist[k] = 0;
  if (Q.group_wg)
    for (uint32_t k = threadIdx.x; k < Q.G; k += LAYOUT_WG)
      ghist[k] = 0;
  if (threadIdx.x < 64)
    dest_lds[threadIdx.x] = 0;
  /* (a set whose offsets the host never saw: what cmpr_layout_queries checks of them) */
  if (Q.total_on_device && blockIdx.x == 0 && threadIdx.x == 0 && q0 == 0 && Q.off[0] != 0)
    atomicCAS(Q.verr, 0u, (uint32_t)VERR_OFFSETS0);
  const uint64_t total = total_of(Q);
  __syncthreads();
  /* (a workgroup takes every gridDim.x-th batch of 256 queries: its sums leave it once, not once
     per batch -- 39 000 workgroups adding to the same few words and the repertoire totals' one
     line took 2.4 ms per 10M queries where the kernel's own work takes a fifth of that) */
  unsigned long long alg = 0;
  uint32_t err_all = 0, Lmax = 0;
  /* The loop is pipelined by hand (round 6: a wave of this kernel waits four fifths of its time, every load and
     the rank's atomic one behind the other): the fields of the thread's NEXT query are asked for while this
     one's residue pieces are on their way, and the rank of a query -- the answer of its atomic -- is written
     an iteration later, when the wait for newer loads has covered it. */
  struct Fields {
    uint64_t b, e, cn;
    uint32_t rp, vg, jg;
  };
  auto fetch = [&](uint64_t i) -> Fields {
    Fields f;
    f.b = f.e = 0;
    f.cn = 1;
    f.rp = f.vg = f.jg = 0;
    if (i < q1) {
      f.b = Q.off[i];
      f.e = Q.off[i + 1];
      f.rp = Q.rep[i];
      if (Q.genes) {
        f.vg = Q.v[i];
        f.jg = Q.j[i];
      }
      if (Q.counts)
        f.cn = Q.cnt[i];
    }
    return f;
  };
  const uint64_t stride = (uint64_t)gridDim.x * LAYOUT_WG;
  uint64_t i = q0 + (uint64_t)blockIdx.x * LAYOUT_WG + threadIdx.x;
  Fields cur = fetch(i);
  uint64_t pend_i = ~0ull;
  uint32_t pend_rank = 0;
  for (; i < q1; i += stride) {
    uint32_t err = 0;
    const uint64_t b = cur.b, e = cur.e;
    uint32_t L = 0;
    if (e < b || e > total)
      err = VERR_OFFSETS;
    else if (e - b > 0xffffu)
      err = VERR_LONG;
    else if (e - b > Q.longest)
      err = VERR_TOO_LONG;
    else
      L = (uint32_t)(e - b);
    const uint32_t rp = cur.rp;
    if (!err && rp >= Q.n_rep)
      err = VERR_REP;
    if (!err && Q.genes && (cur.vg >= Q.n_v_max || cur.jg >= Q.n_j_max))
      err = VERR_GENE;
    if (!err && Q.counts && cur.cn < 1)
      err = VERR_COUNT;
    OwnPieces<NP> pieces;
    pieces.fits = false;
    if (!err)
      pieces = own_load<NP>(Q, b, e, total, own_np);
    const Fields nxt = fetch(i + stride);
    const uint8_t *s = err ? Q.res : own_commit(Q, pieces, b, e, mine);
    if (pend_i != ~0ull) {
      Q.rank[pend_i] = pend_rank;
      pend_i = ~0ull;
    }
    const uint64_t cnt_i = cur.cn;
    const uint32_t vg = cur.vg, jg = cur.jg;
    cur = nxt;
    if (!err) {
      bool bad = false;
      for (uint32_t p = 0; p < L; p++)
        bad = bad || s[p] >= Q.A;
      if (bad)
        err = VERR_RESIDUE;
    }
    uint32_t grp_out = 0xffffffffu;               /* (one store of the query's group, at the end) */
    if (err && !err_all)
      err_all = err;
    if (!err) {
      Lmax = max(Lmax, L);
      const double x = Q.counts ? (double)cnt_i : 1.0;
      if (lds_tot)
        unsafeAtomicAdd(tot_lds + rp, x);
      else
        unsafeAtomicAdd(Q.rep_total + rp, x);
      uint32_t ck = 0;
      bool heavy = false;
      if (Q.sliced && !LDBG(Q, LDBG_NO_CLASSKEY)) {
        /* (class_key_of, layout.h, with the tables in front of the heavy bitmap read from LDS) */
        ck = class_base(ct, Q.geom, Q.genes != 0, L, vg, jg);
        heavy = Q.geom.k > 0 && class_is_heavy(Q.geom.ctab, Q.geom, ck);
        if (heavy && L > 0)
          for (uint32_t k = 0; k < Q.geom.k; k++)
            ck ^= ct[Q.geom.off_cr + k * Q.A + s[class_pos(L, k, Q.geom.c0)]];
      } else if (Q.sliced)
        ck = (uint32_t)i * 2654435761u;
      if (Q.rows || Q.direct) {
        /* zobrist_hash (zobrist.cc:74-88) and, with -i, the two shifted hashes that
           seed the rolling indel enumeration (:90-104, :122-136) */
        uint64_t h = 0;
        if (Q.genes) {
          if (Q.zob_lds) {
            const uint64_t *gk = zl + Q.A * Q.zpos;
            h = gk[vg] ^ gk[Q.n_v + jg];
          } else {
            const uint64_t *gk = Q.zob + (uint64_t)Q.A * Q.zpos;
            h = gk[vg] ^ gk[Q.n_v + jg];
          }
        }
        uint64_t hins = h, hdel = h;
        if (Q.zob_lds)
          hashes_of((const uint64_t *)zl, Q.A, Q.indels != 0, s, LDBG(Q, LDBG_NO_HASH) ? 0u : L, h, hins, hdel);
        else
          hashes_of(Q.zob, Q.A, Q.indels != 0, s, LDBG(Q, LDBG_NO_HASH) ? 0u : L, h, hins, hdel);
        if (Q.direct) {
          /* d = 0 without a filter: no slices to group by -- the pseudo-slice only spreads the group
             counters (one per length would take every query's atomic) and names the work shard */
          ck = (uint32_t)(h >> 40);
          Q.h_tmp[i] = h;
        } else if (!LDBG(Q, LDBG_NO_TMP_WRITES)) {
          Q.h_tmp[i] = h;
          if (Q.indels) {
            Q.hins_tmp[i] = hins;
            Q.hdel_tmp[i] = hdel;
          }
          Q.ck_tmp[i] = ck;
        }
      } else if (Q.sub2_items) {
        Q.ck_tmp[i] = ck;
      }
      if (Q.route) {
        /* cmpr_route_queries: the contexts this query's record goes to -- the one that works
           on its slice and those that work on one of its items (verification reads the
           record where the positive is resolved) */
        unsigned long long mask = 1ull << work_owner(Q.sliced ? (ck & Q.geom.smask) : (ck & DIRECT_OWNER_MASK), 0u, Q.wstep);
        if (Q.ngroups)
          for_each_item<false>(Q, ct, Q.zob, 0ull, 0ull, s, L, ck, heavy, [&](uint32_t k, uint64_t, uint32_t) {
            mask |= 1ull << item_owner(Q, k, Q.wstep);
          });
        if (Q.route == 2u)
          mask = Q.wstep >= 64u ? ~0ull : (1ull << Q.wstep) - 1ull;       /* (a layout every context holds in full) */
        grp_out = (uint32_t)mask;
        Q.rank[i] = (uint32_t)(mask >> 32);
        for (unsigned long long m = mask; m; m &= m - 1ull)
          atomicAdd(&dest_lds[__builtin_ctzll(m)], 1u);
      }
      /* the items of the query that this context works on */
      bool any_item = false;
      if (Q.ngroups && !Q.route && !LDBG(Q, LDBG_NO_ITEM_COUNT))
        for_each_item<false>(Q, ct, Q.zob, 0ull, 0ull, s, L, ck, heavy, [&](uint32_t k, uint64_t, uint32_t) {
          if (item_owned(Q, k)) {
            if (Q.item_wg)
              atomicAdd(&ihist[k], 1u);
            else
              atomicAdd(Q.ccnt_r + (size_t)((uint32_t)(i >> 8) % Q.item_reps) * Q.nitem_slices + k, 1u);
            any_item = true;
          }
        });
      /* its place: a tile of its slice if this context works on that slice; else, if
         one of its items is worked on here, a tile of the foreign pseudo-slice (no
         chunk lists it: the record is all that is needed); else none */
      const uint64_t gl = Q.longest - L;
      const uint32_t slice = Q.sliced ? (ck & Q.geom.smask) : (ck & Q.pmask);
      uint64_t bucket = ~0ull;
      /* (whose work a query of the direct layout is does not depend on how many pseudo-slices this call
         groups its queries by: the context that routes a record and the one that receives it agree) */
      if (owned(Q, Q.sliced ? slice : (ck & DIRECT_OWNER_MASK), 0u))
        bucket = Q.sliced ? 2 * (uint64_t)slice + (heavy ? 1 : 0) : slice;
      else if (any_item)        /* (spread by query number: one counter per length would serialise
                                   millions of atomics -- 27 ms per 10M queries at two shards, round 3) */
        bucket = 2 * (Q.nslices_real + (((uint32_t)i * 2654435761u) >> (32 - FOREIGN_SLICES_LOG2)));
      if (bucket != ~0ull && !Q.route) {
        const uint32_t g = (uint32_t)(bucket * Q.per_slice + gl);
        grp_out = g;
        /* (its top bit: the query's class is split -- scatter_kernel need not ask the tables again) */
        pend_rank = (LDBG(Q, LDBG_NO_RANK) ? 0u : Q.group_wg ? atomicAdd(&ghist[g], 1u) : atomicAdd(Q.cnt_g[0] + g, 1u)) |
                    (heavy ? 0x80000000u : 0u);
        pend_i = i;                              /* (written when the next iteration has waited for newer loads) */
      }
      /* (the same key as owned() above: ADVICE r5) */
      if (Q.alg_step <= 1u || work_owner(Q.sliced ? slice : (ck & DIRECT_OWNER_MASK), 0u, Q.alg_step) == Q.alg_first)
        alg += LDBG(Q, LDBG_NO_TOTALS) ? 0ull : (uint64_t)L + 20 + 8 * variants_of(Q, s, L);
    }
    Q.grp[0][i] = grp_out;
  }
  if (pend_i != ~0ull)
    Q.rank[pend_i] = pend_rank;
  if (err_all)
    atomicCAS(Q.verr, 0u, err_all);
  /* longest: one atomic per wave */
  uint32_t m = Lmax;
  for (int o = 32; o > 0; o >>= 1)
    m = max(m, (uint32_t)__shfl_down((int)m, o, WAVE));
  if ((threadIdx.x & 63) == 0 && m)
    atomicMax(Q.verr + 1, m);
  for (int o = 32; o > 0; o >>= 1)
    alg += __shfl_down(alg, o, WAVE);
  if ((threadIdx.x & 63) == 0 && alg)
    atomicAdd(Q.alg_bytes, alg);
  __syncthreads();
  if (lds_tot) {
    for (uint32_t r = threadIdx.x; r < Q.n_rep; r += LAYOUT_WG)
      if (tot_lds[r] != 0.0)
        unsafeAtomicAdd(Q.rep_total + r, tot_lds[r]);
  }
  if (Q.route && threadIdx.x < 64 && dest_lds[threadIdx.x])
    atomicAdd(Q.dest_cnt + threadIdx.x, (unsigned long long)dest_lds[threadIdx.x]);
  /* group_wg: a run of every group this workgroup met, for its queries of the group (QL::group_wg) */
  if (Q.group_wg) {
    uint32_t *const row = Q.gbase_w + (size_t)(row0 + blockIdx.x) * Q.G;
    for (uint32_t g = threadIdx.x; g < Q.G; g += LAYOUT_WG) {
      const uint32_t cn = ghist[g];
      if (cn)
        row[g] = atomicAdd(Q.cnt_g[0] + g, cn);
    }
  }
  /* item_wg: a run for this workgroup's items of every counter it met (QL::item_wg) */
  if (Q.item_wg) {
    uint32_t *const row = Q.rbase + (size_t)(row0 + blockIdx.x) * Q.nitem_slices;
    for (uint32_t k = threadIdx.x; k < Q.nitem_slices; k += LAYOUT_WG) {
      const uint32_t cn = ihist[k];
      if (cn)
        row[k] = atomicAdd(Q.ccnt_r + (size_t)k * ITEM_PAD, cn);
    }
  }
}

/* ---- narrowed upload: back to the caller's types -------------------------- */

/* The host narrowed what it could before the copy (lengths instead of 64-bit offsets,
   16-bit gene / repertoire numbers, 32-bit counts: 12 bytes per query instead of 28,
   cmpr_layout_queries); a range of them becomes the caller's arrays again here.  The
   offsets are an exclusive scan of the lengths (hipCUB, started at the range's first
   offset). */
struct Len16To64 {
  __host__ __device__ unsigned long long operator()(const uint16_t &x) const { return x; }
};

__global__ void __launch_bounds__(256)
widen_kernel(const uint16_t *v16, const uint16_t *j16, const uint16_t *rep16, const uint32_t *cnt32,
             uint32_t *v, uint32_t *j, uint32_t *rep, uint64_t *cnt, uint64_t q0, uint64_t q1)
{
  const uint64_t i = q0 + (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= q1)
    return;
  rep[i] = rep16[i];
  if (v) {
    v[i] = v16[i];
    j[i] = j16[i];
  }
  if (cnt)
    cnt[i] = cnt32[i];
}

/* ---- slices: tiles, chunks ------------------------------------------------ */

/* One thread per slice and pass.  WRITE = 0: what the slice needs (counted);
   WRITE = 1: the same walk, writing at the slice's exclusive prefix. */
constexpr uint32_t SLICES_WG = 128;
template <int WRITE>
__global__ void __launch_bounds__(SLICES_WG)
slices_kernel(const QL Q, uint32_t pi, uint32_t in_lds)
{
  /* (the group counters of the workgroup's SLICES_WG slices lie side by side: one coalesced copy to LDS, and
     the loops below -- every step of which waited for a counter from memory -- read them there; round 6) */
  extern __shared__ uint32_t cnt_lds[];
  const uint32_t per_sl = (Q.sliced ? 2u : 1u) * Q.per_slice;
  const uint64_t sl0 = (uint64_t)blockIdx.x * SLICES_WG;
  if (in_lds) {
    const uint64_t first = sl0 * per_sl, end = min((sl0 + SLICES_WG) * (uint64_t)per_sl, Q.nslices * (uint64_t)per_sl);
    for (uint64_t k = first + threadIdx.x; k < end; k += SLICES_WG)
      cnt_lds[k - first] = Q.cnt_g[pi][k];
    __syncthreads();
  }
  const uint64_t sl = sl0 + threadIdx.x;
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
    const uint64_t bucket = Q.sliced ? 2 * sl + hv : sl;
    const uint32_t tile_k = hv ? Q.geom.k : 0u;
    /* (the bucket's counters: in the workgroup's copy, or where they lie) */
    const uint32_t *c = in_lds ? cnt_lds + (bucket * Q.per_slice - sl0 * per_sl) : cnt + bucket * Q.per_slice;
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
  if (Q.sliced && ntiles && sl < Q.nslices_real) {       /* (the foreign pseudo-slices: tiles, no work) */
    /* (variant 2 with -i: the deletion and insertion rows of a tile follow its
       substitution rows in the same unit, on the same staged slice -- its class keys
       have no length term -- so there is one chunk per slice, not one per pass) */
    /* (a slice with pages -- layout.h SliceGeom -- is staged once per page: a chunk per page and run of
       tiles, the page and the slice's e in the chunk's pass word) */
    const uint32_t pt = (Q.rows && Q.geom.page_tab) ? Q.geom.page_tab[sl] : 0u;
    const uint32_t pe = pt & 15u;
    const uint32_t reps = 1u << pe;
    if (!Q.indels && pe == 0u && (ntiles <= Q.small_max || (pi > 0 && Q.class_unstaged))) {
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
            ck.slice = page_slice(Q.geom, (uint32_t)sl, pt, rp);
            ck.first_tile = Q.list0[pi] + at.list + k * Q.chunk_tiles;
            ck.ntiles = min(Q.chunk_tiles, ntiles - k * Q.chunk_tiles);
            ck.pass = pass | (rp << CHUNK_PAGE_SHIFT) | (pe << CHUNK_PAGE_E_SHIFT);
            if (Q.sub2_items && pi == 0 && k == 0)
              ck.pass |= CHUNK_WITH_ITEMS;      /* (the slice's item blocks ride along) */
            if (Q.pairs2) {
              /* kernels_pairs2.h hands a tile out pair by pair: the pairs of the chunk's longest tile */
              uint32_t lmax = 0;
              for (uint32_t t = 0; t < ck.ntiles; t++)
                lmax = max(lmax, Q.tiles[tile_base + k * Q.chunk_tiles + t].len);
              ck.pass |= ((lmax + 1u) / 2u) << 16;
            }
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

/* One thread per query: its 64-byte record to slot = group base + rank -- a whole piece of memory, the only
   scattered write per query -- and its items (variant 2 / sub2: the flat items of kernels_rows.h, 16 bytes
   each; a pass of their own over the queries until round 5).  The records leave the wave TRANSPOSED (round 6):
   a lane's sixteen dwords go through its words of LDS, and four adjacent lanes write the four 16-byte pieces of
   one record -- 64 contiguous bytes, one request to the memory system where a lane writing its own record in
   four instructions made four (6.5 write requests per query in all, 0.23 ms of the kernel's 1.0 even with every
   record sent to consecutive slots).  (Sets with sequences beyond the record's 36 residues, and the residue
   packs of kernels_pairs2.h, also leave a QAux beside the record: QL::recompute.) */
/* (six waves per SIMD -- three workgroups of 512 per CU beside the shared tables -- is worth more than the
   registers a seventh loop-carried field would like: at 109 registers, four waves per SIMD, the kernel took
   0.63 ms per 10M queries where it takes 0.53) */
template <uint32_t LAYOUT_WG, int MODE>
__global__ void __launch_bounds__(LAYOUT_WG, (LAYOUT_WG == 1024 || MODE == 0) ? 4 : 6)
scatter_kernel(const QL Q, uint64_t q0, uint64_t q1, uint32_t row0)
{
  layout_mode_assume<MODE>(Q);
  /* dynamic LDS: (item_wg) where this workgroup's items of counter k go next | SCAT_DW words per thread */
  extern __shared__ uint64_t scat_lds[];
  uint64_t *const zl = scat_lds;
  uint32_t *const ctl = (uint32_t *)(zl + (Q.zob_lds ? Q.zob_pos_words : 0u));
  uint32_t *const ibase = ctl + (Q.ctab_lds ? Q.geom.off_hv : 0u);
  uint32_t *const gb = ibase + (Q.item_wg ? Q.nitem_slices : 0u);        /* group_wg: first slot of this workgroup's
                                                                          queries of group g */
  constexpr int NP = MODE == 0 ? OWN_NP_MAX : 3;
  const uint32_t own_np = MODE == 0 ? Q.own_np : 3u, own_dw = 4u * own_np + 1u;
  uint32_t *const wave_words = gb + (Q.group_wg ? Q.G : 0u) + (threadIdx.x & ~63u) * own_dw;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t *const mine = wave_words + lane * own_dw;
  const uint32_t *const ct = Q.ctab_lds ? ctl : Q.geom.ctab;
  const uint64_t *const zt = Q.zob_lds ? zl : Q.zob;
  if (Q.zob_lds)                             /* (the keys of the positions; the gene keys are not asked here) */
    for (uint32_t k = threadIdx.x; k < Q.zob_pos_words; k += LAYOUT_WG)
      zl[k] = Q.zob[k];
  if (Q.ctab_lds)
    for (uint32_t k = threadIdx.x; k < Q.geom.off_hv; k += LAYOUT_WG)
      ctl[k] = Q.geom.ctab[k];
  if (Q.group_wg) {
    /* (a group this workgroup did not meet holds whatever the arena held -- nobody asks) */
    const uint32_t *const row = Q.gbase_w + (size_t)(row0 + blockIdx.x) * Q.G;
    for (uint32_t g = threadIdx.x; g < Q.G; g += LAYOUT_WG)
      gb[g] = Q.base_g[0][g] + row[g];
  }
  if (Q.zob_lds || Q.ctab_lds || Q.group_wg)
    __syncthreads();
  if (Q.item_wg) {
    /* (the run keys_kernel's workgroup of the same number claimed for the same queries; a counter it did not
       meet holds whatever the arena held -- nobody asks) */
    const uint32_t *const row = Q.rbase + (size_t)(row0 + blockIdx.x) * Q.nitem_slices;
    for (uint32_t k = threadIdx.x; k < Q.nitem_slices; k += LAYOUT_WG)
      ibase[k] = Q.cbase[k] + row[k];
    __syncthreads();
  }
  const uint64_t total = total_of(Q);
  /* (the same queries as keys_kernel's workgroup of this number: QL::item_wg; a wave stays together to the
     end of its last batch: the records are written by the wave, not by the lane) */
  /* (pipelined by hand like keys_kernel: the NEXT query's fields are asked for while this one's group base and
     residue pieces are on their way) */
  struct Fields {                            /* (what the loads of an iteration depend on; the rest is asked
                                                for with them, not ahead: registers -- 80 for six waves per SIMD) */
    uint64_t b, e;
    uint32_t g, rk;
  };
  const bool want_h = Q.ngroups != 0 && Q.rows, want_ck = Q.ngroups != 0 && (Q.rows || Q.sub2_items);
  auto fetch = [&](uint64_t i) -> Fields {
    Fields f;
    f.b = f.e = 0;
    f.g = 0xffffffffu;
    f.rk = 0;
    if (i < q1) {
      f.g = Q.grp[0][i];
      f.rk = Q.rank[i];
      f.b = Q.off[i];
      f.e = Q.off[i + 1];
    }
    return f;
  };
  const uint64_t stride = (uint64_t)gridDim.x * LAYOUT_WG;
  uint64_t i0 = q0 + (uint64_t)blockIdx.x * LAYOUT_WG + (threadIdx.x & ~63u);
  Fields cur = fetch(i0 + lane);
  for (; i0 < q1; i0 += stride) {
    const uint64_t i = i0 + lane;
    uint32_t slot = 0xffffffffu;
    QueryRec qr;
    qr.cnt = 0;
    qr.v = qr.j = qr.rep = qr.len = qr.orig = 0;
    const uint32_t g = cur.g;                    /* (all ones: not worked on by this context, no item of it either --
                                                    or behind the end of the range) */
    const uint64_t b = cur.b, e = cur.e;
    const uint32_t rk = cur.rk;                  /* (rank in the group | class split << 31: keys_kernel) */
    uint32_t gbase = 0, ck = 0;
    uint64_t h_q = 0, hins_q = 0;
    OwnPieces<NP> pieces;
    pieces.fits = false;
    if (g != 0xffffffffu) {
      gbase = LDBG(Q, LDBG_S_NO_BASE) ? 0u : Q.group_wg ? gb[g] : Q.base_g[0][g];
      pieces = own_load<NP>(Q, b, e, total, own_np);
      qr.cnt = Q.counts ? Q.cnt[i] : 1ull;
      qr.v = Q.genes ? Q.v[i] : 0u;
      qr.j = Q.genes ? Q.j[i] : 0u;
      qr.orig = Q.orig ? Q.orig[i] : (uint32_t)i;
      qr.rep = Q.existence ? qr.orig : Q.rep[i];      /* -x: the row is the sequence itself */
      if (want_h) {
        h_q = Q.h_tmp[i];
        if (Q.indels)
          hins_q = Q.hins_tmp[i];
      }
      if (want_ck)
        ck = Q.ck_tmp[i];
      if (Q.rec_hash && !want_h)
        h_q = Q.h_tmp[i];
    }
    cur = fetch(i + stride);
    if (g != 0xffffffffu) {
      slot = LDBG(Q, LDBG_S_NO_BASE) ? (uint32_t)i : gbase + (rk & 0x7fffffffu);
      const uint32_t L = (uint32_t)(e - b);
      const uint8_t *s = own_commit(Q, pieces, b, e, mine);
      qr.len = L;
      if (!Q.recompute) {
        QAux a;
        a.h = a.hins = a.hdel = 0;
        a.ck = 0;
        a.src = (uint32_t)i;
        if (Q.rows) {
          a.h = Q.h_tmp[i];
          a.ck = Q.ck_tmp[i];
          if (Q.indels) {
            a.hins = Q.hins_tmp[i];
            a.hdel = Q.hdel_tmp[i];
          }
        } else if (Q.direct) {
          a.h = Q.h_tmp[i];
        } else if (Q.genes) {
          const uint64_t *gk = Q.zob + (uint64_t)Q.A * Q.zpos;
          a.h = gk[Q.v[i]] ^ gk[Q.n_v + Q.j[i]];
        }
        Q.aux[slot] = a;
      }
      if (Q.ngroups != 0 && !LDBG(Q, LDBG_S_NO_ITEMS)) {
        /* ---- the query's items: the row's (variant's) hash, what to exclude / where / what kind, and
                the query's slot in pass 0 ---- */
        const bool heavy = (rk >> 31) != 0;
        uint64_t hq = 0;
        cmpr::ResPack pk{};
        if (Q.pairs2 && heavy)
          for (uint32_t x = 0; x < L && x < RESPACK_MAX; x++)
            pk.w[x >> 4] |= ((uint32_t)s[x] & 3u) << ((x & 15u) * 2u);
        if (Q.sub2_items && heavy && L <= RESPACK_MAX) {
          if (Q.genes) {
            const uint64_t *gk = Q.zob + (uint64_t)Q.A * Q.zpos;
            hq = gk[qr.v] ^ gk[Q.n_v + qr.j];
          }
          for (uint32_t x = 0; x < L; x++) {
            const uint32_t r = s[x];
            hq ^= Q.zob[Q.A * x + r];
            pk.w[x >> 4] |= (r & 3u) << ((x & 15u) * 2u);
          }
        }
        for_each_item<true>(Q, ct, zt, h_q, hins_q, s, L, ck, heavy, [&](uint32_t k, uint64_t w, uint32_t crp) {
          if (!item_owned(Q, k))
            return;
          uint32_t item;
          if (Q.item_wg) {
            item = atomicAdd(&ibase[k], 1u);
          } else {
            const size_t rk = (size_t)((uint32_t)(i >> 8) % Q.item_reps) * Q.nitem_slices + k;
            item = Q.cbase[k] + Q.rbase[rk] + atomicAdd(Q.cfill_r + rk, 1u);
          }
          if (Q.sub2_items) {
            w = hq;                                   /* the query's hash and residues travel with the item */
            Q.cpk[item] = pk;
          } else if (Q.pairs2) {
            Q.cpk[item] = pk;                         /* (beside the pair-blanked hash) */
          }
          ItemRec it;
          it.w = w;
          it.main = slot;
          it.rp = crp;
          if (!LDBG(Q, LDBG_S_NO_ITEM_WRITE))
            Q.items[item] = it;
        });
      }
    }
    /* (the record's residue dwords last: nine registers that need not live through the items) */
    if (slot != 0xffffffffu) {
      const uint32_t L = (uint32_t)(e - b);
      if (pieces.fits) {
        /* nine dwords from ten aligned ones (the thread's copy keeps the set's byte phase) */
        const uint32_t o = (uint32_t)(b & 15u), w0 = o >> 2, sh = o & 3u;
        uint32_t lo = mine[w0];
#pragma unroll
        for (uint32_t w = 0; w < 9; w++) {
          const uint32_t hi = mine[w0 + w + 1u];             /* (at most the thread's 13th word; masked below) */
          const uint32_t d = __builtin_amdgcn_alignbyte(hi, lo, sh);
          const int n = (int)L - (int)(4u * w);
          qr.res[w] = d & (n <= 0 ? 0u : n >= 4 ? 0xffffffffu : (1u << (8 * n)) - 1u);
          lo = hi;
        }
      } else {
        const uint8_t *s = Q.res + b;
#pragma unroll
        for (uint32_t w = 0; w < 9; w++) {
          uint32_t d = 0;
#pragma unroll
          for (uint32_t k = 0; k < 4; k++)
            if (4 * w + k < L)
              d |= (uint32_t)s[4 * w + k] << (8 * k);
          qr.res[w] = d;
        }
      }
      if (Q.rec_hash) {
        /* record tiles, every sequence within 28 residues: the hash keys_kernel worked out rides in the residue
           words no sequence of the set reaches (bytes 28 .. 35) -- probe_rows_kernel does not hash the tile again */
        qr.res[7] = (uint32_t)h_q;
        qr.res[8] = (uint32_t)(h_q >> 32);
      }
    } else {
#pragma unroll
      for (uint32_t w = 0; w < 9; w++)
        qr.res[w] = 0;
    }
    /* ---- the wave's records, transposed through LDS: half a wave at a time -- 32 records of 17 words fit the
            wave's 64 x 13 words, whose residue pieces have been read ---- */
    if (LDBG(Q, LDBG_S_NO_REC))
      continue;
    static_assert(sizeof(QueryRec) == 64, "the record is four 16-byte pieces");
    static_assert(32 * 17 <= 64 * OWN_DW, "half a wave's records in the wave's words (13 per lane at least)");
    const uint32_t out_slot = LDBG(Q, LDBG_S_SEQ_REC) && slot != 0xffffffffu ? (uint32_t)i : slot;
#pragma unroll
    for (uint32_t half = 0; half < 2; half++) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if ((lane >> 5) == half) {
        uint32_t *const dst = wave_words + (lane & 31u) * 17u;
        const uint32_t *q = (const uint32_t *)&qr;
#pragma unroll
        for (uint32_t w = 0; w < 16; w++)
          dst[w] = q[w];
        dst[16] = out_slot;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (uint32_t r = 0; r < 2; r++) {
        const uint32_t *src = wave_words + (16u * r + (lane >> 2)) * 17u;
        const uint32_t part = lane & 3u;
        const uint32_t sl = src[16];
        const uint4 v = make_uint4(src[4 * part], src[4 * part + 1], src[4 * part + 2], src[4 * part + 3]);
        if (sl != 0xffffffffu)
          *(uint4 *)((char *)(Q.qrec + sl) + 16u * part) = v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

/* One wave per tile, one lane per slot: the records scatter_kernel left, read in slot
   order, become the per-slot arrays and the tile's position-major residues (layout.h
   TileDesc) -- every write coalesced, the padding lanes written too (zeros), so that
   nothing of the layout needs a memset. */
__global__ void __launch_bounds__(256)
fill_tiles_kernel(const QL Q, uint32_t ntiles)
{
  /* dynamic LDS (recompute): the Zobrist keys and the class tables in front of the heavy bitmap, when they fit
     (QL::zob_lds, ctab_lds: a lookup in memory with 64 addresses per wave is the dear kind) */
  extern __shared__ uint64_t fill_lds[];
  __shared__ uint32_t rl[256 * 9];           /* recompute: a lane's residues, read a byte at a time */
  const bool tabs = Q.recompute != 0;
  uint64_t *const zl = fill_lds;
  uint32_t *const ctl = (uint32_t *)(zl + (tabs && Q.zob_lds ? Q.zob_words : 0u));
  if (tabs && Q.zob_lds)
    for (uint32_t k = threadIdx.x; k < Q.zob_words; k += 256)
      zl[k] = Q.zob[k];
  if (tabs && Q.ctab_lds)
    for (uint32_t k = threadIdx.x; k < Q.geom.off_hv; k += 256)
      ctl[k] = Q.geom.ctab[k];
  if (tabs && (Q.zob_lds || Q.ctab_lds))
    __syncthreads();
  const uint64_t *const zt = tabs && Q.zob_lds ? zl : Q.zob;
  const uint32_t *const ct = tabs && Q.ctab_lds ? ctl : Q.geom.ctab;
  const uint32_t lane = threadIdx.x % WAVE;
  /* (a wave per tile, the grid what is resident: the tables are copied once per workgroup) */
  for (uint32_t t = blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE; t < ntiles; t += gridDim.x * (256 / WAVE)) {
  const TileDesc td = Q.tiles[t];
  const uint32_t slot = t * WAVE + lane;
  const bool valid = lane < td.nvalid;
  QueryRec qr;
  QAux a;
  a.h = a.hins = a.hdel = 0;
  a.ck = a.src = 0;
  if (valid) {
    qr = Q.qrec[slot];
    if (!Q.recompute)
      a = Q.aux[slot];
  } else {
    qr.cnt = 0;
    qr.v = qr.j = qr.rep = qr.len = qr.orig = 0;
#pragma unroll
    for (uint32_t w = 0; w < 9; w++)
      qr.res[w] = 0;
    Q.qrec[slot] = qr;
  }
  if (Q.recompute && valid) {
    /* what keys_kernel worked out for this query, once more from its record (every residue is in it):
       the hashes and the class key are not scattered beside the record (32 bytes per query to a random
       place; round 6) -- zobrist_hash and the two shifted hashes, zobrist.cc:74-136 */
    uint32_t *const mine = rl + threadIdx.x * 9;
#pragma unroll
    for (uint32_t w = 0; w < 9; w++)
      mine[w] = qr.res[w];
    const uint8_t *s = (const uint8_t *)mine;
    const uint32_t L = qr.len;
    uint64_t h = 0;
    if (Q.genes) {
      const uint64_t *gk = zt + (uint64_t)Q.A * Q.zpos;
      h = gk[qr.v] ^ gk[Q.n_v + qr.j];
    }
    if (Q.rows || Q.direct) {
      uint64_t hins = h, hdel = h;
      hashes_of(zt, Q.A, Q.indels != 0, s, L, h, hins, hdel);
      a.hins = hins;
      a.hdel = hdel;
      if (Q.rows) {
        /* (class_key_of, layout.h, with the tables in front of the heavy bitmap read from LDS) */
        uint32_t ck = class_base(ct, Q.geom, Q.genes != 0, L, qr.v, qr.j);
        if (Q.geom.k > 0 && class_is_heavy(Q.geom.ctab, Q.geom, ck) && L > 0)
          for (uint32_t k = 0; k < Q.geom.k; k++)
            ck ^= ct[Q.geom.off_cr + k * Q.A + s[class_pos(L, k, Q.geom.c0)]];
        a.ck = ck;
      }
    }
    a.h = h;
  }
  Q.qlen[slot] = (uint16_t)qr.len;
  if (Q.genes && !Q.rows) {                  /* (variants 0 and 1: their tiles hash / key by the genes) */
    Q.qv[slot] = qr.v;
    Q.qj[slot] = qr.j;
  }
  if (Q.rows) {
    Q.qgh[slot] = a.h;
    Q.qck[slot] = a.ck;
    if (Q.indels) {
      Q.qhins[slot] = a.hins;
      Q.qhdel[slot] = a.hdel;
    }
  } else if (Q.genes || Q.direct) {
    Q.qgh[slot] = a.h;
  }
  if (Q.pairs2) {
    /* kernels_pairs2.h reads a query's residues two bits each, one 24-byte piece per slot */
    cmpr::ResPack pk{};
    if (valid) {
      /* (16 residues of the pack = one 16-byte piece of the set when it is aligned: a load per piece, not per
         residue -- 5.5 ms per 12.5M nucleotide queries were spent here, a byte and a wait at a time) */
      const uint64_t b0 = Q.off[a.src];
      const uint32_t n = min(qr.len, (uint32_t)RESPACK_MAX);
      const uint64_t total = total_of(Q);
#pragma unroll
      for (uint32_t k = 0; k < RESPACK_MAX / 16; k++) {
        uint32_t word = 0;
        if (16u * k < n) {
          const uint64_t at = b0 + 16u * k;
          if (at + 16u <= total) {
            /* (two dwords at a time: the set's residues are at any byte phase) */
            const uint8_t *s = Q.res + at;
            uint64_t lo, hi;
            __builtin_memcpy(&lo, s, 8);
            __builtin_memcpy(&hi, s + 8, 8);
#pragma unroll
            for (uint32_t x = 0; x < 8; x++) {
              if (16u * k + x < n)
                word |= ((uint32_t)(lo >> (8u * x)) & 3u) << (x * 2u);
              if (16u * k + 8u + x < n)
                word |= ((uint32_t)(hi >> (8u * x)) & 3u) << ((8u + x) * 2u);
            }
          } else {
            for (uint32_t x = 0; x < 16u && 16u * k + x < n; x++)
              word |= ((uint32_t)Q.res[at + x] & 3u) << (x * 2u);
          }
        }
        pk.w[k] = word;
      }
    }
    Q.qpk[slot] = pk;
  }
  /* residues four to a dword, position-major / lane-minor; the first nine dwords are in
     the record, longer sequences fetch the rest where the caller's residues lie */
  uint32_t *dst = Q.qres + td.res_base + lane;
  const uint32_t nd = (td.len + 3u) >> 2;
  /* (pair rows: behind its end a query carries code A, whose key in the kernel's table is
     zero -- kernels_rows.h zs_of; lanes without a query are all padding) */
  const uint32_t padw = Q.pairs ? Q.A * 0x01010101u : 0u;
  auto padded = [&](uint32_t d, uint32_t w) -> uint32_t {
    const int n = (int)qr.len - (int)(4u * w);            /* residues of the query in this dword */
    const uint32_t m = n <= 0 ? 0u : n >= 4 ? 0xffffffffu : (1u << (8 * n)) - 1u;
    return (d & m) | (padw & ~m);
  };
#pragma unroll
  for (uint32_t w = 0; w < 9; w++)
    if (w < nd)
      dst[(size_t)w * WAVE] = padded(qr.res[w], w);
  if (nd > 9) {
    const uint8_t *s = valid ? Q.res + Q.off[a.src] : Q.res;
    const uint32_t L = qr.len;
    for (uint32_t w = 9; w < nd; w++) {
      uint32_t d = 0;
#pragma unroll
      for (uint32_t k = 0; k < 4; k++)
        if (4 * w + k < L)
          d |= (uint32_t)s[4 * w + k] << (8 * k);
      dst[(size_t)w * WAVE] = padded(d, w);
    }
  }
  }
}

/* the copies of the item counters summed, and where each copy's items start inside their
   (class part, slice) (QL::ccnt_r) */
__global__ void __launch_bounds__(256)
item_replicas_kernel(const QL Q)
{
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= (uint64_t)Q.nitem_slices)
    return;
  if (Q.item_wg) {                           /* (counted per workgroup: the padded counters hold the sums) */
    Q.ccnt[k] = Q.ccnt_r[(size_t)k * ITEM_PAD];
    return;
  }
  uint32_t run = 0;
  for (uint32_t r = 0; r < Q.item_reps; r++) {
    const size_t rk = (size_t)r * Q.nitem_slices + k;
    Q.rbase[rk] = run;
    run += Q.ccnt_r[rk];
  }
  Q.ccnt[k] = run;
}

/* What the host waits for between the two halves of the layout, in ONE piece of device memory (one copy, one
   wait: until round 6 six small copies, ~20 us of stream time each): the layout's totals, the first error of
   the validation, the algorithmic bytes, the sums of the item and sibling lists, the count totals per
   repertoire behind it. */
struct SizesBlock {
  SliceTot last_tot, last_pre;
  unsigned long long alg, sums[4];
  uint32_t verr[2];
};

/* One workgroup: the two exclusive sums over the item counters (padded items -> first item, chunks -> first
   chunk; 32-bit prefixes, 64-bit totals) that four hipCUB calls made, then the SizesBlock. */
__global__ void __launch_bounds__(1024)
sizes_kernel(const QL Q, const uint32_t *cpad, uint64_t ncs, unsigned long long *sums, SizesBlock *blk)
{
  typedef hipcub::BlockScan<uint32_t, 1024> Scan;
  __shared__ typename Scan::TempStorage tmp;
  unsigned long long run0 = 0, run1 = 0;
  for (uint64_t base = 0; base < ncs; base += 1024) {
    const uint64_t k = base + threadIdx.x;
    const uint32_t a = k < ncs ? cpad[k] : 0u, b = k < ncs ? Q.cnch[k] : 0u;
    uint32_t ea, eb, ta, tb;
    Scan(tmp).ExclusiveSum(a, ea, ta);
    __syncthreads();
    Scan(tmp).ExclusiveSum(b, eb, tb);
    __syncthreads();
    if (k < ncs) {
      Q.cbase[k] = (uint32_t)run0 + ea;
      Q.cchpre[k] = (uint32_t)run1 + eb;
    }
    run0 += ta;
    run1 += tb;
  }
  if (threadIdx.x == 0) {
    blk->last_tot = Q.tot[0][Q.nslices - 1];
    blk->last_pre = Q.pre[0][Q.nslices - 1];
    blk->alg = *Q.alg_bytes;
    blk->sums[0] = run0;
    blk->sums[1] = run1;
    blk->sums[2] = sums[2];
    blk->sums[3] = sums[3];
    blk->verr[0] = Q.verr[0];
    blk->verr[1] = Q.verr[1];
  }
  double *rt = (double *)(blk + 1);
  for (uint32_t r = threadIdx.x; r < Q.n_rep; r += 1024)
    rt[r] = Q.rep_total[r];
}

/* the padding of the item lists: behind the items of counter k up to its whole blocks of 64, and one block
   behind the last list (a block read past the end stays inside) -- all ones = "no item" (round 6: a memset of the
   whole array, 160 MB per 10M queries, did this before every call) */
__global__ void __launch_bounds__(256)
item_padding_kernel(const QL Q, const uint32_t *padded, uint32_t total)
{
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  ItemRec none;
  memset(&none, 0xff, sizeof none);
  if (k < (uint64_t)Q.nitem_slices)
    for (uint32_t x = Q.cbase[k] + Q.ccnt[k]; x < Q.cbase[k] + padded[k]; x++)
      Q.items[x] = none;
  if (k < WAVE)
    Q.items[total + k] = none;
}

/* the slice of the filter the items of counter k are filed under, and its pages (layout.h SliceGeom):
   returns e (0: no pages) */
__device__ inline uint32_t item_slice_pages(const QL &Q, uint64_t k, uint32_t *slice, uint32_t *pt_out)
{
  if (Q.sub2_items || !Q.rows)
    return 0u;
  uint32_t gi = 0;
  for (uint32_t x = 1; x < Q.ngroups; x++)
    if (k >= Q.goff[x])
      gi = x;
  const uint32_t ls = Q.gslice0[gi] + (uint32_t)(k - Q.goff[gi]);
  const uint32_t pt = Q.geom.page_tab ? Q.geom.page_tab[ls] : 0u;
  if (slice)
    *slice = ls;
  if (pt_out)
    *pt_out = pt;
  return pt & 15u;
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
  Q.cnch[k] = ((blocks + Q.cblocks - 1) / Q.cblocks) << item_slice_pages(Q, k, nullptr, nullptr);
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
  uint32_t ls = 0, pt = 0;
  const uint32_t pe = item_slice_pages(Q, k, &ls, &pt);
  for (uint32_t q = 0; q < nc; q++)
    for (uint32_t rp = 0; rp < (1u << pe); rp++) {
      Chunk ck;
      ck.slice = Q.sub2_items ? (uint32_t)(k / Q.ngroups) : page_slice(Q.geom, ls, pt, rp);
      ck.first_tile = Q.cbase[k] + q * Q.cblocks * WAVE;      /* first item */
      ck.ntiles = min(Q.cblocks, blocks - q * Q.cblocks);     /* blocks of 64 items */
      ck.pass = (3 + gi) | (rp << CHUNK_PAGE_SHIFT) | (pe << CHUNK_PAGE_E_SHIFT);
      Q.chunks[Q.cchunk0 + Q.cchpre[k] + (q << pe) + rp] = ck;
    }
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

__device__ inline uint32_t chunk_work_of(const QL &Q, const Chunk &ck);

__global__ void __launch_bounds__(256)
chunk_work_kernel(const QL Q, uint32_t nchunks, uint32_t *work, uint32_t *idx)
{
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nchunks)
    return;
  const uint32_t w = chunk_work_of(Q, Q.chunks[k]);
  work[k] = w;
  if (idx)
    idx[k] = k;
  else                                        /* (chunk_order_kernel's classes: the heaviest chunk) */
    atomicMax(work + nchunks, w);
}

/* The chunk order in ONE launch (round 6; up to ORDER_MAX chunks): heaviest first is what the probe kernels' deal
   wants, and it wants it roughly -- the tail of a launch is made of the light chunks -- so a counting sort by
   1024 classes of weight (linear up to the heaviest chunk) does what a 32-bit radix sort of (weight, index) pairs
   did in eight launches (0.05 ms of a 2 ms query set).  The weights and their maximum (work[nchunks]) come from
   chunk_work_kernel; then one workgroup: the classes' histogram in LDS, its prefix sums from the heavy end, the
   chunks to their places. */
constexpr uint32_t ORDER_MAX = 1u << 18;
constexpr uint32_t ORDER_CLASSES = 1024;
__global__ void __launch_bounds__(1024)
chunk_order_kernel(const QL Q, uint32_t nchunks, uint32_t *work, Chunk *out)
{
  __shared__ uint32_t hist[ORDER_CLASSES];
  __shared__ uint32_t wmax;
  typedef hipcub::BlockScan<uint32_t, 1024> Scan;
  __shared__ typename Scan::TempStorage tmp;
  hist[threadIdx.x] = 0;
  if (threadIdx.x == 0)
    wmax = max(work[nchunks], 1u);
  __syncthreads();
  const uint64_t top = wmax;
  auto cls = [&](uint32_t w) -> uint32_t {          /* class 0 = the heaviest */
    return (ORDER_CLASSES - 1u) - (uint32_t)(((uint64_t)w * (ORDER_CLASSES - 1u)) / top);
  };
  for (uint32_t k = threadIdx.x; k < nchunks; k += 1024)
    atomicAdd(&hist[cls(work[k])], 1u);
  __syncthreads();
  uint32_t first = 0;
  Scan(tmp).ExclusiveSum(hist[threadIdx.x], first);
  __syncthreads();
  hist[threadIdx.x] = first;
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < nchunks; k += 1024)
    out[atomicAdd(&hist[cls(work[k])], 1u)] = Q.chunks[k];
}

__device__ inline uint32_t chunk_work_of(const QL &Q, const Chunk &ck)
{
  const uint32_t cpass = ck.pass & 0xffu;
  /* (variant 2 at d = 2 on single rows: an item reads ~12 x 19 words of its slice, kernels_rows.h) */
  uint64_t w = cpass >= 3 ? (uint64_t)ck.ntiles * WAVE * (Q.sub2_items ? 48 : Q.pairs2 ? 5 : (Q.rows && Q.differences == 2) ? 230 : 2) : 0;
  for (uint32_t t = 0; cpass < 3 && t < ck.ntiles; t++) {
    const TileDesc td = Q.tile_refs[ck.first_tile + t].td;
    /* (variant 2: a tile costs its wave the same however full it is; ~10 rows' worth per unit for claim,
       descriptor, tile data and the identity probe.  With -i a position is a pair row plus an insertion pair) */
    if (Q.rows)
      w += (uint64_t)((Q.indels ? 2u : 1u) * td.len + 10u) * WAVE;
    else
      w += (uint64_t)(cpass == 0 ? td.len + 1 : cpass == 1 ? td.len + 2 : 2) * td.nvalid;
  }
  if (ck.pass & CHUNK_WITH_ITEMS)
    w += (uint64_t)Q.slice_items[ck.slice].y * WAVE * 48;
  return (uint32_t)min(w, (uint64_t)0xffffffffu);
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

__global__ void __launch_bounds__(256)
chunk_mine_kernel(const Chunk *chunks, const uint32_t *idx, uint32_t n, uint32_t first, uint32_t step,
                  unsigned char *flag)
{
  const uint32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k < n) {
    const Chunk ck = chunks[idx[k]];
    flag[k] = work_owner(ck.slice, ck.pass & 0xffu, step) == first ? 1 : 0;
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


/* ---- routed records (cmpr_route_queries / cmpr_route_pack / cmpr_set_queries_routed) ------ */

/* One record = the query as the layout needs it: layout.h QueryRec (count, genes, repertoire,
   length, residues 0..35, its number in the caller's WHOLE set) and, for sets with longer
   sequences, the further residues four to a dword -- record_bytes each, a multiple of 16. */
__host__ __device__ inline uint32_t route_record_bytes(uint32_t longest)
{
  return (uint32_t)sizeof(QueryRec) + (longest > 36u ? (longest - 36u + 15u) / 16u * 16u : 0u);
}

/* One thread per query of the share: its record to every destination named by the keys
   kernel, at (first record of the destination) + (a position claimed per workgroup with one
   atomic per destination). */
__global__ void __launch_bounds__(256)
route_pack_kernel(const QL Q, const uint32_t *mask_lo, const uint32_t *mask_hi, uint64_t first_index,
                  const unsigned long long *dest_base, unsigned long long *dest_fill,
                  unsigned char *out, uint32_t record_bytes)
{
  __shared__ uint32_t cnt_lds[64];
  __shared__ unsigned long long base_lds[64];
  if (threadIdx.x < 64)
    cnt_lds[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned long long mask = 0;
  if (i < Q.n)
    mask = ((unsigned long long)mask_hi[i] << 32) | mask_lo[i];
  uint32_t at[4] = {0, 0, 0, 0};        /* position inside the workgroup's run, first four destinations
                                           (a query with more claims the others one by one, below) */
  {
    uint32_t k = 0;
    for (unsigned long long m = mask; m && k < 4; m &= m - 1ull, k++)
      at[k] = atomicAdd(&cnt_lds[__builtin_ctzll(m)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const uint32_t d = threadIdx.x;
    base_lds[d] = cnt_lds[d] ? dest_base[d] + atomicAdd(dest_fill + d, (unsigned long long)cnt_lds[d]) : 0ull;
  }
  __syncthreads();
  if (!mask)
    return;
  const uint64_t b = Q.off[i];
  const uint32_t L = (uint32_t)(Q.off[i + 1] - b);
  const uint8_t *s = Q.res + b;
  QueryRec qr;
  qr.cnt = Q.counts ? Q.cnt[i] : 1ull;
  qr.v = Q.genes ? Q.v[i] : 0u;
  qr.j = Q.genes ? Q.j[i] : 0u;
  qr.rep = Q.rep[i];
  qr.len = L;
  qr.orig = (uint32_t)(first_index + i);
#pragma unroll
  for (uint32_t w = 0; w < 9; w++) {
    uint32_t d = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)
      if (4 * w + k < L)
        d |= (uint32_t)s[4 * w + k] << (8 * k);
    qr.res[w] = d;
  }
  const uint32_t more = (record_bytes - (uint32_t)sizeof(QueryRec)) / 4u;     /* dwords behind the header */
  uint32_t k = 0;
  for (unsigned long long m = mask; m; m &= m - 1ull, k++) {
    const uint32_t d = (uint32_t)__builtin_ctzll(m);
    unsigned long long pos;
    if (k < 4)
      pos = base_lds[d] + at[k];
    else            /* (more than four destinations: the layouts every context holds in full) */
      pos = dest_base[d] + atomicAdd(dest_fill + d, 1ull);
    unsigned char *o = out + (size_t)pos * record_bytes;
    *(QueryRec *)o = qr;
    uint32_t *x = (uint32_t *)(o + sizeof(QueryRec));
    for (uint32_t w = 0; w < more; w++) {
      uint32_t dw = 0;
      for (uint32_t q = 0; q < 4; q++)
        if (36u + 4u * w + q < L)
          dw |= (uint32_t)s[36u + 4u * w + q] << (8 * q);
      x[w] = dw;
    }
  }
}

/* records -> lengths (what the offsets are scanned from); a length that cannot be is named */
__global__ void __launch_bounds__(256)
unpack_len_kernel(const unsigned char *rec, uint64_t n, uint32_t record_bytes, uint32_t longest,
                  uint16_t *len16, uint32_t *verr)
{
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i > n)
    return;
  uint32_t L = 0;
  if (i < n) {
    L = ((const QueryRec *)(rec + (size_t)i * record_bytes))->len;
    if (L > longest) {
      atomicCAS(verr, 0u, (uint32_t)VERR_TOO_LONG);
      L = 0;
    }
  }
  len16[i] = (uint16_t)L;                    /* ([n] = 0: the scan's last output is the total) */
}

/* records -> the arrays the layout kernels read (the caller's types) */
__global__ void __launch_bounds__(256)
unpack_kernel(const unsigned char *rec, uint64_t n, uint32_t record_bytes, const uint64_t *off,
              uint8_t *res, uint32_t *v, uint32_t *j, uint32_t *rep, uint64_t *cnt, uint32_t *orig)
{
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n)
    return;
  const unsigned char *r = rec + (size_t)i * record_bytes;
  const QueryRec qr = *(const QueryRec *)r;
  rep[i] = qr.rep;
  orig[i] = qr.orig;
  if (v) {
    v[i] = qr.v;
    j[i] = qr.j;
  }
  if (cnt)
    cnt[i] = qr.cnt;
  const uint64_t b = off[i];
  const uint32_t L = (uint32_t)(off[i + 1] - b);
  const uint32_t *more = (const uint32_t *)(r + sizeof(QueryRec));
  for (uint32_t p = 0; p < L; p++) {
    const uint32_t w = p < 36u ? qr.res[p >> 2] : more[(p - 36u) >> 2];
    res[b + p] = (uint8_t)(w >> ((p & 3u) * 8u));
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

/* ... of a kernel whose workgroups loop over the batches and leave their sums once */
inline uint32_t blocks_looping(uint64_t n)
{
  return std::min<uint32_t>(blocks_for(n), 2048u);
}

const char *verr_message(uint32_t e)
{
  switch (e) {
  case VERR_OFFSETS: return "offsets not monotone";
  case VERR_LONG:    return "sequence longer than 65535 residues";
  case VERR_REP:     return "repertoire number out of range";
  case VERR_GENE:    return "gene number out of range";
  case VERR_COUNT:   return "duplicate_count must be >= 1";
  case VERR_OFFSETS0: return "offsets[0] must be 0";
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
                             std::vector<double> &rep_total, bool on_device, uint64_t total_dev)
{
  int rc;
  /* (on_device: the view's arrays are device memory -- a copy inside the device; its
     offsets[n] was fetched by the caller) */
  const uint64_t total = on_device ? total_dev : (s->n ? s->offsets[s->n] : 0);
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  static const uint64_t zero_off[1] = {0};
  auto put = [&](auto &buf, const auto *from, size_t count, bool host_src = false) -> int {
    int r = dev_alloc(c, buf, count);
    if (r)
      return r;
    if (count)
      HIP_TRY(c, hipMemcpyAsync(buf.p, from, count * sizeof(*from), host_src ? hipMemcpyHostToDevice : kind,
                                c->stream));
    return CMPR_OK;
  };
  if ((rc = put(res, s->residues, (size_t)total))) return rc;
  if (s->n) {
    if ((rc = put(off, s->offsets, (size_t)s->n + 1))) return rc;
  } else if ((rc = put(off, zero_off, 1, true))) {
    return rc;
  }
  if ((rc = put(rep, s->repertoire, (size_t)s->n))) return rc;
  if (!c->opt.ignore_genes) {
    if ((rc = put(v, s->v_gene, (size_t)s->n))) return rc;
    if ((rc = put(j, s->j_gene, (size_t)s->n))) return rc;
  } else {
    v.release();
    j.release();
  }
  if (!c->opt.ignore_counts) {
    if ((rc = put(cnt, s->count, (size_t)s->n))) return rc;
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
  hipLaunchKernelGGL(validate_seq_kernel, dim3(blocks_looping(s->n)), dim3(256), lds, c->stream, Q);
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

/* cutting an arena: offsets first (sizes known), pointers once it is large enough */
struct ArenaCut {
  size_t used = 0;
  size_t take(size_t bytes)
  {
    const size_t at = used;
    used += (bytes + 255) & ~(size_t)255;
    return at;
  }
};

static int arena_fit(cmpr_context *c, DevArena &A, size_t bytes)
{
  if (A.base && A.cap >= bytes)
    return CMPR_OK;
  A.release();
  /* (a block cmpr_warm_up_sized reserved on this device while the caller still read its input) */
  size_t got = 0;
  if (char *p = cmpr_take_reserved_device(c->device, bytes, &got)) {
    A.base = p;
    A.cap = got;
    return CMPR_OK;
  }
  const size_t want = bytes + bytes / 16 + 4096;
  HIP_TRY(c, hipMalloc((void **)&A.base, want));
  A.cap = want;
  return CMPR_OK;
}

struct U32To64 {
  __host__ __device__ unsigned long long operator()(const uint32_t &x) const { return x; }
};

/* sum of n uint32 as 64 bits into *out (the 32-bit prefix sums beside it may not wrap
   unnoticed: ADVICE r2) */
static hipError_t sum64(void *tmp, size_t &bytes, const uint32_t *in, unsigned long long *out, int n,
                        hipStream_t st)
{
  hipcub::TransformInputIterator<unsigned long long, U32To64, const uint32_t *> it(in, U32To64());
  return hipcub::DeviceReduce::Sum(tmp, bytes, it, out, n, st);
}

int cmpr_layout_queries(cmpr_context *c, const LayoutSource &src)
{
  int rc;
  const auto t_begin = std::chrono::steady_clock::now();
  /* whatever way this function is left early: no copy from the caller's arrays (which may be
     freed on return) or from the pinned staging area (which the next call may reallocate) is
     still in flight, and no kernel still reads the arenas */
  struct Quiesce {
    cmpr_context *c;
    bool ok = false;
    ~Quiesce()
    {
      if (!ok) {
        (void)hipStreamSynchronize(c->copy_stream);
        (void)hipStreamSynchronize(c->stream);
      }
    }
  } quiesce{c};
  const cmpr_set_view *const s = src.view;
  const bool from_host = src.kind == LayoutSource::HOST;
  const bool from_records = src.kind == LayoutSource::RECORDS;
  const bool routing = src.route;                          /* cmpr_route_queries: key the share, count per destination */
  const uint32_t A = (uint32_t)c->opt.alphabet_size;
  const uint32_t Lcap = c->zpos - EXTRA_POSITIONS;        /* the longest a query may be */
  const uint64_t n = from_records ? src.nrec : s->n;
  const uint32_t n_rep = from_records ? src.n_rep : s->n_repertoires;
  const uint32_t record_bytes = route_record_bytes(Lcap);
  c->route.valid = false;
  uint64_t total = 0;                                     /* residues in all */
  if (from_host) {
    total = n ? s->offsets[n] : 0;
  } else if (from_records) {
    total = n * (uint64_t)Lcap;                           /* (an upper bound sizes the arena; the scan gives the sum) */
  }
  /* (device arrays: the offsets stay where they are -- the keys kernel reads offsets[n] there, checks
     offsets[0] and every sequence's length; until round 6 two blocking copies fetched the two ends first) */
  const bool total_on_device = src.kind == LayoutSource::DEVICE && n > 0;
  if (n > 0x7fffffffull)                                  /* (hipCUB item counts are int) */
    return fail(c, CMPR_EUNSUPPORTED, "more than 2^31-1 sequences in one set");
  if (!total_on_device && total > 0xffffull * n)
    return fail(c, CMPR_EINVAL, verr_message(VERR_OFFSETS));
  /* the caller's arrays on the device: copied there (host), unpacked there (records), or
     where the caller has them (device; an empty set still needs its one offset) */
  const bool soa_in_arena = src.kind != LayoutSource::DEVICE || n == 0;
  if (!routing) {
    c->n1 = n;
    c->R1 = c->opt.existence ? (uint32_t)(from_records ? src.n_total : n) : n_rep;
    c->routed = from_records;
  }

  const uint32_t wstep = (uint32_t)c->work_shard_count;
  const uint32_t wfirst = (uint32_t)c->work_shard_index;
  if (wfirst >= wstep)
    return fail(c, CMPR_EINVAL, "work_shard_index must be below work_shard_count");
  /* d = 0 on the un-sliced kernel (no filter: kernels.h probe_kernel, D == 0): the hash is computed here and
     the queries are grouped by length and by PSEUDO-SLICE -- bits of the hash: one group counter per length
     would take every query's atomic, one thread would write every tile (slices_kernel), and a work shard is
     named by bits of the same key.  One per 32 768 queries (the part-filled last tile of each group is the
     price: ~3 % of the slots). */
  const bool direct = !c->sliced && c->opt.differences == 0;
  if (wstep > 1 && !c->sliced && !direct)
    return fail(c, CMPR_EUNSUPPORTED, "work shards need a sliced layout (kernel variant 1 or 2), or d = 0");
  uint64_t pseudo = 1;
  if (direct) {
    if (c->direct_slices_log2 >= 0)
      pseudo = 1ull << c->direct_slices_log2;
    else {
      while (pseudo < 1024 && pseudo * 32768 < n)
        pseudo <<= 1;
    }
  }
  /* variant 1 with -i lists every tile under two sibling slices of other owners: there
     every context lays out everything and takes its share of the chunk list */
  const bool indel_passes = c->sliced && c->opt.indels && !c->rows;
  const bool place_mine = wstep > 1 && !indel_passes;

  const uint64_t nslices_real = c->sliced ? (uint64_t)c->geom.smask + 1 : pseudo;
  const uint64_t nslices = nslices_real + (place_mine && c->sliced ? (1ull << FOREIGN_SLICES_LOG2) : 0);   /* + the foreign pseudo-slices */
  const uint64_t nbuckets = c->sliced ? 2 * nslices : nslices;
  const uint64_t per_slice = (uint64_t)Lcap + 1;
  if (nbuckets * per_slice >= 0x7fffffffull)
    return fail(c, CMPR_EUNSUPPORTED, "too many (slice, length) groups");
  const uint64_t G = nbuckets * per_slice;
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
  const uint64_t n2s = indel_passes ? 2 * nslices : 0;

  /* ---- the query ranges keys_kernel and scatter_kernel are launched over (host arrays: up to four, a
          range's keys run under the copy of the next; few and large: every copy call costs ~20 us of host
          time, and the last range -- the smallest -- is what the upload does not hide) ---- */
  struct Range { uint64_t q0, q1; uint32_t grid, row0; };
  std::vector<Range> ranges;
  /* few item counters: counted per workgroup in LDS (QL::item_wg) */
  const bool item_wg = ncs > 0 && ncs <= ITEM_WG_MAX && !routing && c->item_wg != 0;
  /* the Zobrist keys (gene keys behind them) in keys_kernel's LDS when they are few: CDR3 lengths, not 1 000-
     nucleotide sequences */
  const uint64_t zob_words = (uint64_t)A * c->zpos + (c->opt.ignore_genes ? 0 : c->opt.n_v_genes + c->opt.n_j_genes);
  const bool zob_lds = c->layout_zob_lds != 0 && zob_words * sizeof(uint64_t) <= 12288;
  const uint64_t zob_pos_words = (uint64_t)A * c->zpos;
  const bool ctab_lds = c->layout_zob_lds != 0 && c->sliced && c->geom.off_hv <= 2048;
  const size_t ctab_bytes = ctab_lds ? (size_t)c->geom.off_hv * sizeof(uint32_t) : 0;
  /* the 16-byte pieces that hold the longest sequence at any byte phase (own_load) */
  const uint32_t own_np_want = std::min<uint32_t>((uint32_t)OWN_NP_MAX, std::max<uint32_t>(3u, (Lcap + 15u + 15u) / 16u));
  /* few groups (the direct layout of d = 0): ranked per workgroup in LDS (QL::group_wg) */
  const bool group_wg = G <= GROUP_WG_MAX && !routing && c->item_wg != 0;
  const size_t keys_shared = (n_rep <= 2048 ? n_rep * sizeof(double) : 0) + (zob_lds ? zob_words * sizeof(uint64_t) : 0) +
                             ctab_bytes + (item_wg ? (size_t)ncs * sizeof(uint32_t) : 0) + (group_wg ? G * sizeof(uint32_t) : 0);
  const size_t scatter_shared = (zob_lds ? zob_pos_words * sizeof(uint64_t) : 0) + ctab_bytes +
                                (item_wg ? (size_t)ncs * sizeof(uint32_t) : 0) + (group_wg ? G * sizeof(uint32_t) : 0);
  typedef void (*LayoutFn)(const QL, uint64_t, uint64_t, uint32_t);
  static const uint32_t wg_sizes[3] = {256, 512, 1024};
  static const LayoutFn keys_all[4][3] = {{keys_kernel<256, 0>, keys_kernel<512, 0>, keys_kernel<1024, 0>},
                                          {keys_kernel<256, 1>, keys_kernel<512, 1>, keys_kernel<1024, 1>},
                                          {keys_kernel<256, 2>, keys_kernel<512, 2>, keys_kernel<1024, 2>},
                                          {keys_kernel<256, 3>, keys_kernel<512, 3>, keys_kernel<1024, 3>}};
  static const LayoutFn scatter_all[4][3] = {{scatter_kernel<256, 0>, scatter_kernel<512, 0>, scatter_kernel<1024, 0>},
                                             {scatter_kernel<256, 1>, scatter_kernel<512, 1>, scatter_kernel<1024, 1>},
                                             {scatter_kernel<256, 2>, scatter_kernel<512, 2>, scatter_kernel<1024, 2>},
                                             {scatter_kernel<256, 3>, scatter_kernel<512, 3>, scatter_kernel<1024, 3>}};
  /* (layout_mode_assume: what the instantiation may take for granted) */
  const bool recompute_on = Lcap <= 36u && !c->d2pairs && c->layout_recompute != 0;
  const int lmode = (c->rows && pair_rows(c) && !c->d2pairs && !sub2_items && !direct && !routing && wstep <= 1 &&
                     c->opt.differences == 1 && recompute_on && c->debug == 0 && !group_wg)
                        ? (c->opt.indels ? 2 : 1)
                        : own_np_want == 3u ? 3 : 0;
  const LayoutFn *const keys_fns = keys_all[lmode], *const scatter_fns = scatter_all[lmode];
  const uint32_t own_np = lmode ? 3u : own_np_want;
  /* Both kernels loop over their queries: the grid is what is RESIDENT at once (LDS and registers decide), no
     more -- with 2 048 workgroups of which 6 or 7 per CU fit, the eighth ran alone behind the others, a second
     round at a seventh of the occupancy for as long as the first (round 6: keys 0.80 -> .., scatter 1.01 -> ..).
     The same grid for both (QL::item_wg). */
  uint32_t resident = 2048;
  uint32_t LAYOUT_WG = 256;
  size_t keys_lds = 0, scatter_lds = 0;
  LayoutFn keys_fn = keys_fns[0], scatter_fn = scatter_fns[0];
  {
    /* the workgroup size that leaves the most waves per CU resident (both kernels: the same grid) */
    uint32_t best_waves = 0;
    int bk = 0, bs = 0;
    for (int w = 0; w < 3; w++) {
      const size_t own = (size_t)wg_sizes[w] * (4 * own_np + 1) * sizeof(uint32_t);
      const size_t kl = keys_shared + own, sl = scatter_shared + own;
      if (kl > 160 * 1024 - 512 || sl > 160 * 1024 - 512)
        continue;
      if ((rc = raise_lds_limit(c, (const void *)keys_fns[w], kl))) return rc;
      if ((rc = raise_lds_limit(c, (const void *)scatter_fns[w], sl))) return rc;
      const int ok = occupancy_of(c, (const void *)keys_fns[w], (int)wg_sizes[w], kl);
      const int os = occupancy_of(c, (const void *)scatter_fns[w], (int)wg_sizes[w], sl);
      const uint32_t waves = (uint32_t)std::max(0, std::min(ok, os)) * wg_sizes[w] / WAVE;
      if (waves >= best_waves && waves > 0) {      /* (a tie: the larger workgroup -- fewer runs of items to claim) */
        best_waves = waves;
        LAYOUT_WG = wg_sizes[w];
        keys_lds = kl;
        scatter_lds = sl;
        keys_fn = keys_fns[w];
        scatter_fn = scatter_fns[w];
        resident = (uint32_t)c->cus * (uint32_t)std::min(ok, os);
        bk = ok;
        bs = os;
      }
    }
    if (best_waves == 0)
      return fail(c, CMPR_EUNSUPPORTED, "the layout kernels' tables do not fit the 160 KiB LDS");
    if (getenv("COMPAIRR_HIP_DEBUG"))
      fprintf(stderr, "compairr_hip: layout kernels: workgroups of %u; keys %zu B of LDS, %d per CU; scatter %zu B, %d; "
                      "grid %u; %llu item counters in %u groups (%s)\n", LAYOUT_WG, keys_lds, bk, scatter_lds, bs,
              resident, (unsigned long long)ncs, ngroups, item_wg ? "per workgroup in LDS" : "replicas in memory");
  }
  {
    const uint64_t min_range = 1u << 18;
    const uint64_t nr = from_host ? std::max<uint64_t>(1, std::min<uint64_t>(4, n / min_range)) : 1;
    static const uint32_t cut4[5] = {0, 30, 60, 86, 100};
    uint32_t row = 0;
    for (uint64_t r = 0; r < nr; r++) {
      Range g;
      g.q0 = nr == 4 ? n * cut4[r] / 100 : n * r / nr;
      g.q1 = nr == 4 ? n * cut4[r + 1] / 100 : n * (r + 1) / nr;
      g.grid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, (g.q1 - g.q0 + LAYOUT_WG - 1) / LAYOUT_WG), resident);
      g.row0 = row;
      row += g.grid;
      ranges.push_back(g);
    }
  }
  const uint64_t item_rows = (uint64_t)ranges.back().row0 + ranges.back().grid;

  /* ---- arena A: everything whose size the host knows now ---- */
  ArenaCut cut;
  /* (the counters that start at zero lie side by side: one memset) */
  const size_t o_gcnt = cut.take(G * npass * sizeof(uint32_t));
  const uint32_t item_reps = item_wg ? ITEM_PAD : ncs > 0 && ncs <= 65536 ? 32u : 1u;
  const size_t o_ccnt = cut.take(ncs * sizeof(uint32_t));
  const size_t o_cfill = cut.take(ncs * sizeof(uint32_t));
  const size_t o_ccnt_r = cut.take(ncs * item_reps * sizeof(uint32_t));
  const size_t o_cfill_r = cut.take(item_wg ? 0 : ncs * item_reps * sizeof(uint32_t));
  const size_t o_sibcnt = cut.take(n2s * sizeof(uint32_t));
  const size_t o_sibfill = cut.take(n2s * sizeof(uint32_t));
  const size_t o_alg = cut.take(sizeof(unsigned long long));
  const size_t o_verr = cut.take(2 * sizeof(uint32_t));
  const size_t o_reptot = cut.take((size_t)n_rep * sizeof(double));
  const size_t o_dest = cut.take(2 * 64 * sizeof(unsigned long long));     /* route: counts | fill */
  const size_t o_sums = cut.take(8 * sizeof(unsigned long long));
  const size_t zero_bytes = cut.used;
  const size_t blk_bytes = sizeof(SizesBlock) + (size_t)n_rep * sizeof(double);
  const size_t o_blk = cut.take(blk_bytes);
  const size_t o_res = cut.take(soa_in_arena ? (size_t)total + 16 : 0);
  const size_t o_off = cut.take(soa_in_arena ? (size_t)(n + 1) * sizeof(uint64_t) : 0);
  const size_t o_v = cut.take(c->opt.ignore_genes || !soa_in_arena ? 0 : (size_t)n * sizeof(uint32_t));
  const size_t o_j = cut.take(c->opt.ignore_genes || !soa_in_arena ? 0 : (size_t)n * sizeof(uint32_t));
  const size_t o_rep = cut.take(soa_in_arena ? (size_t)n * sizeof(uint32_t) : 0);
  const size_t o_cnt = cut.take(c->opt.ignore_counts || !soa_in_arena ? 0 : (size_t)n * sizeof(uint64_t));
  const size_t o_orig = cut.take(from_records ? (size_t)n * sizeof(uint32_t) : 0);
  /* narrowed upload (below): lengths, 16-bit ids, 32-bit counts as they arrive */
  const unsigned hw = std::thread::hardware_concurrency();
  const bool narrow_fits = from_host && n_rep <= 65536 &&
                           (c->opt.ignore_genes || (c->opt.n_v_genes <= 65536 && c->opt.n_j_genes <= 65536));
  bool narrow = narrow_fits && (c->narrow_upload == 1 || (c->narrow_upload < 0 && n >= (1u << 20) && hw >= 8));
  const size_t o_len16 = cut.take(narrow || from_records ? (size_t)(n + 1) * sizeof(uint16_t) : 0);
  const size_t o_rep16 = cut.take(narrow ? (size_t)n * sizeof(uint16_t) : 0);
  const size_t o_v16 = cut.take(narrow && !c->opt.ignore_genes ? (size_t)n * sizeof(uint16_t) : 0);
  const size_t o_j16 = cut.take(narrow && !c->opt.ignore_genes ? (size_t)n * sizeof(uint16_t) : 0);
  const size_t o_cnt32 = cut.take(narrow && !c->opt.ignore_counts ? (size_t)n * sizeof(uint32_t) : 0);
  const size_t o_rbase = cut.take(ncs * (item_wg ? item_rows : (uint64_t)item_reps) * sizeof(uint32_t));
  const size_t o_gbase_w = cut.take(group_wg ? (size_t)G * item_rows * sizeof(uint32_t) : 0);
  const size_t o_gbase = cut.take(G * npass * sizeof(uint32_t));
  const size_t o_grp = cut.take((size_t)n * sizeof(uint32_t));
  const size_t o_rank = cut.take((size_t)n * sizeof(uint32_t));
  const size_t o_tfirst = cut.take(G * sizeof(uint32_t));
  const size_t o_tot = cut.take(nslices * sizeof(SliceTot));
  const size_t o_pre = cut.take(nslices * sizeof(SliceTot));
  const bool need_ck = c->rows || sub2_items;
  const size_t o_h = cut.take(c->rows || direct ? (size_t)n * sizeof(uint64_t) : 0);
  const size_t o_ck = cut.take(need_ck ? (size_t)n * sizeof(uint32_t) : 0);
  const size_t o_hins = cut.take(c->rows && c->opt.indels ? (size_t)n * sizeof(uint64_t) : 0);
  const size_t o_hdel = cut.take(c->rows && c->opt.indels ? (size_t)n * sizeof(uint64_t) : 0);
  const size_t o_cbase = cut.take(ncs * sizeof(uint32_t));
  const size_t o_cnch = cut.take(ncs * sizeof(uint32_t));
  const size_t o_cchpre = cut.take(ncs * sizeof(uint32_t));
  const size_t o_cpad = cut.take(ncs * sizeof(uint32_t));
  const size_t o_sibnch = cut.take(n2s * sizeof(uint32_t));
  const size_t o_siblpre = cut.take(n2s * sizeof(uint32_t));
  const size_t o_sibcpre = cut.take(n2s * sizeof(uint32_t));
  /* hipCUB temporary storage: the largest any of the scans / sums below asks for */
  size_t cub_bytes = 0;
  {
    SliceTot zero;
    memset(&zero, 0, sizeof zero);
    size_t b = 0;
    (void)hipcub::DeviceScan::ExclusiveScan(nullptr, b, (SliceTot *)nullptr, (SliceTot *)nullptr, SliceTotSum(),
                                            zero, (int)nslices, c->stream);
    cub_bytes = std::max(cub_bytes, b);
    const int big = (int)std::max<uint64_t>(std::max<uint64_t>(ncs, n2s), 1);
    b = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (uint32_t *)nullptr, (uint32_t *)nullptr, big, c->stream);
    cub_bytes = std::max(cub_bytes, b);
    b = 0;
    (void)sum64(nullptr, b, nullptr, nullptr, big, c->stream);
    cub_bytes = std::max(cub_bytes, b);
    if (narrow || from_records) {
      hipcub::TransformInputIterator<unsigned long long, Len16To64, const uint16_t *> it(nullptr, Len16To64());
      b = 0;
      (void)hipcub::DeviceScan::ExclusiveScan(nullptr, b, it, (unsigned long long *)nullptr, hipcub::Sum(),
                                              0ull, (int)(n + 1), c->stream);
      cub_bytes = std::max(cub_bytes, b);
    }
  }
  const size_t o_cub = cut.take(cub_bytes + 256);
  if ((rc = arena_fit(c, c->arena_a, cut.used))) return rc;
  const auto t_arena_a = std::chrono::steady_clock::now();
  char *const base = c->arena_a.base;
  auto at = [&](size_t off) -> char * { return base + off; };
  HIP_TRY(c, hipMemsetAsync(base, 0, zero_bytes, c->stream));

  QL Q;
  memset(&Q, 0, sizeof Q);
  if (soa_in_arena) {
    Q.res = (const uint8_t *)at(o_res);
    Q.off = (const uint64_t *)at(o_off);
    Q.v = c->opt.ignore_genes ? nullptr : (const uint32_t *)at(o_v);
    Q.j = c->opt.ignore_genes ? nullptr : (const uint32_t *)at(o_j);
    Q.rep = (const uint32_t *)at(o_rep);
    Q.cnt = c->opt.ignore_counts ? nullptr : (const uint64_t *)at(o_cnt);
  } else {                                   /* the caller's device arrays, read where they lie */
    Q.res = s->residues;
    Q.off = s->offsets;
    Q.v = c->opt.ignore_genes ? nullptr : s->v_gene;
    Q.j = c->opt.ignore_genes ? nullptr : s->j_gene;
    Q.rep = s->repertoire;
    Q.cnt = c->opt.ignore_counts ? nullptr : s->count;
  }
  Q.orig = from_records ? (const uint32_t *)at(o_orig) : nullptr;
  Q.n = n;
  Q.total = total;
  Q.n_rep = n_rep;
  Q.alg_step = from_records ? (uint32_t)c->work_shard_count : 1u;
  Q.alg_first = (uint32_t)c->work_shard_index;
  Q.dest_cnt = (unsigned long long *)at(o_dest);
  Q.n_v_max = c->opt.n_v_genes;
  Q.n_j_max = c->opt.n_j_genes;
  Q.A = A;
  Q.zpos = c->zpos;
  Q.n_v = c->opt.ignore_genes ? 0 : c->opt.n_v_genes;
  Q.longest = Lcap;
  Q.per_slice = (uint32_t)per_slice;
  Q.genes = c->opt.ignore_genes ? 0 : 1;
  Q.counts = c->opt.ignore_counts ? 0 : 1;
  Q.existence = c->opt.existence ? 1 : 0;
  Q.indels = c->opt.indels ? 1 : 0;
  Q.pairs = pair_rows(c) ? 1 : 0;
  Q.pairs2 = c->d2pairs ? 1u : 0u;
  Q.differences = (uint32_t)c->opt.differences;
  Q.dbg = (uint32_t)c->debug;
  Q.sliced = c->sliced ? 1 : 0;
  Q.rows = c->rows ? 1 : 0;
  Q.zob = c->zob.p;
  Q.geom = c->geom;
  Q.npass = npass;
  Q.min_mixed = mixed_ok ? c->geom.c0 + c->geom.k + (c->opt.indels ? 1u : 0u) : 0xffffffffu;
  Q.chunk_tiles = (uint32_t)chunk_tiles;
  /* (kernels_pairs2.h works on staged chunks only) */
  Q.small_max = c->d2pairs ? 0u : (uint32_t)c->small_slice_tiles;
  Q.class_unstaged = c->class_rows_unstaged && !c->d2pairs ? 1u : 0u;
  Q.nbuckets = nbuckets;
  Q.nslices = nslices;
  Q.nslices_real = nslices_real;
  Q.wfirst = place_mine ? wfirst : 0u;
  Q.wstep = place_mine ? wstep : 1u;
  Q.direct = direct ? 1u : 0u;
  /* (the row filter at d = 1: layout.h ProbeParams::rec_tiles; decided here because scatter_kernel writes the
     records, applied -- c->rec_tiles -- where fill_tiles_kernel would run) */
  const bool rows_rec = c->rows && pair_rows(c) && !c->d2pairs && c->opt.differences == 1 && A == 20 &&
                        Lcap <= (c->opt.indels ? 31u : 32u) && c->record_tiles != 0 && !sub2_items;
  /* (d = 0 was tried on records too -- the hash in the record, no qgh, no fill_tiles_kernel: 0.23 ms of layout less
     per 10M queries, but probe_kernel<A, 0> reading a 64-byte record per lane a tile ahead instead of 8 coalesced
     bytes took 0.11 ms more and lost a wave per SIMD to the registers: per set -5 %, per launch +30 %; not kept,
     profiles/r06/extra/d0_record_tiles.txt) */
  const bool rec_hash = rows_rec && recompute_on && Lcap <= 28u && c->record_tiles != 2;
  Q.rec_hash = rec_hash ? 1u : 0u;
  Q.pmask = (uint32_t)(pseudo - 1);
  if (routing) {                             /* (a layout every context holds in full: every record to everyone) */
    Q.route = place_mine || wstep == 1 ? 1u : 2u;
    Q.wstep = wstep;
  }
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
  Q.cnt_g[0] = (uint32_t *)at(o_gcnt);
  Q.base_g[0] = (uint32_t *)at(o_gbase);
  Q.grp[0] = (uint32_t *)at(o_grp);
  Q.rank = (uint32_t *)at(o_rank);
  Q.tot[0] = (SliceTot *)at(o_tot);
  Q.pre[0] = (SliceTot *)at(o_pre);
  Q.tfirst_g = (uint32_t *)at(o_tfirst);
  Q.h_tmp = (uint64_t *)at(o_h);
  Q.hins_tmp = (uint64_t *)at(o_hins);
  Q.hdel_tmp = (uint64_t *)at(o_hdel);
  Q.ck_tmp = (uint32_t *)at(o_ck);
  Q.alg_bytes = (unsigned long long *)at(o_alg);
  Q.verr = (uint32_t *)at(o_verr);
  Q.rep_total = (double *)at(o_reptot);
  Q.ccnt = (uint32_t *)at(o_ccnt);
  Q.cbase = (uint32_t *)at(o_cbase);
  Q.cfill = (uint32_t *)at(o_cfill);
  Q.ccnt_r = (uint32_t *)at(o_ccnt_r);
  Q.cfill_r = (uint32_t *)at(o_cfill_r);
  Q.rbase = (uint32_t *)at(o_rbase);
  Q.item_reps = item_reps;
  Q.item_wg = item_wg ? 1u : 0u;
  Q.group_wg = group_wg ? 1u : 0u;
  Q.G = (uint32_t)G;
  Q.gbase_w = (uint32_t *)at(o_gbase_w);
  /* every query fits its 64-byte record (36 residues): the tiles' hashes and class keys are worked out from
     the records, nothing else is scattered (kernels_pairs2.h's residue packs are built where the set lies) */
  Q.recompute = recompute_on ? 1u : 0u;
  Q.total_on_device = total_on_device ? 1u : 0u;
  Q.cnch = (uint32_t *)at(o_cnch);
  Q.cchpre = (uint32_t *)at(o_cchpre);
  uint32_t *const cpad = (uint32_t *)at(o_cpad);
  uint32_t *const sib_cnt = (uint32_t *)at(o_sibcnt), *const sib_fill = (uint32_t *)at(o_sibfill);
  uint32_t *const sib_nch = (uint32_t *)at(o_sibnch), *const sib_lpre = (uint32_t *)at(o_siblpre);
  uint32_t *const sib_cpre = (uint32_t *)at(o_sibcpre);
  unsigned long long *const sums = (unsigned long long *)at(o_sums);
  void *const cub_tmp = at(o_cub);

  /* ---- the caller's arrays, a range of queries at a time on the copy stream; the keys
          kernel of a range (validation included) runs while the next is copied.

          NARROWED (large sets, >= 8 host threads): the link carries 43 bytes per query of
          this ABI -- 7.9 ms per 10M at the ~54 GB/s it gives pageable memory -- of which 28
          are 64-bit offsets and counts and 32-bit ids that fit 2 + 2 + 2 + 2 + 4.  Host
          threads narrow range r + 1 into pinned memory while range r is on the link; a scan
          and a small kernel restore the caller's types on the device.  A value that does
          not fit (a count >= 2^32, an id >= 2^16, offsets that are no CDR3 lengths) sends
          the whole set the wide way, where the keys kernel names the error if it is one. ---- */
  double upload_ms = 0;
  Q.zob_lds = zob_lds ? 1u : 0u;
  Q.zob_words = (uint32_t)zob_words;
  Q.zob_pos_words = (uint32_t)zob_pos_words;
  Q.own_np = own_np;
  Q.ctab_lds = ctab_lds ? 1u : 0u;
  /* HIP events around the big kernels when the caller asks for their times (tunable "layout_timing";
     an event record is a packet the stream waits ~4.5 us for) */
  const bool timing = c->layout_timing != 0 && !from_host && c->ev_layout[0] != nullptr;
  c->layout_marks = 0;
#define LAYOUT_MARK(k)                                                      \
  do {                                                                      \
    if (timing) {                                                           \
      HIP_TRY(c, hipEventRecord(c->ev_layout[k], c->stream));               \
      c->layout_marks |= 1u << (k);                                         \
    }                                                                       \
  } while (0)
  if (!from_host) {
    static const uint64_t zero_off1[1] = {0};
    const size_t lds = keys_lds;
    if (n == 0) {
      HIP_TRY(c, hipMemcpyAsync(at(o_off), zero_off1, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    } else {
      if (from_records) {
        /* records -> lengths -> offsets (scan) -> the arrays the keys kernel reads */
        const unsigned char *rec = (const unsigned char *)src.d_records;
        hipLaunchKernelGGL(unpack_len_kernel, dim3(blocks_for(n + 1)), dim3(256), 0, c->stream, rec, n,
                           record_bytes, Lcap, (uint16_t *)at(o_len16), Q.verr);
        HIP_TRY(c, hipGetLastError());
        hipcub::TransformInputIterator<unsigned long long, Len16To64, const uint16_t *> it(
            (const uint16_t *)at(o_len16), Len16To64());
        size_t b = cub_bytes;
        HIP_TRY(c, hipcub::DeviceScan::ExclusiveScan(cub_tmp, b, it, (unsigned long long *)at(o_off), hipcub::Sum(),
                                                     0ull, (int)(n + 1), c->stream));
        hipLaunchKernelGGL(unpack_kernel, dim3(blocks_for(n)), dim3(256), 0, c->stream, rec, n, record_bytes,
                           (const uint64_t *)at(o_off), (uint8_t *)at(o_res),
                           c->opt.ignore_genes ? nullptr : (uint32_t *)at(o_v),
                           c->opt.ignore_genes ? nullptr : (uint32_t *)at(o_j), (uint32_t *)at(o_rep),
                           c->opt.ignore_counts ? nullptr : (uint64_t *)at(o_cnt), (uint32_t *)at(o_orig));
        HIP_TRY(c, hipGetLastError());
      }
      LAYOUT_MARK(0);
      hipLaunchKernelGGL(keys_fn, dim3(ranges[0].grid), dim3(LAYOUT_WG), lds, c->stream, Q, (uint64_t)0, n, 0u);
      HIP_TRY(c, hipGetLastError());
      LAYOUT_MARK(1);
    }
  } else
  for (int attempt = 0;; attempt++) {
    /* (few, large ranges: every copy call costs ~20 us of host time, and the keys kernel of
       the last range is what the upload does not hide) */
    const uint64_t nranges = ranges.size();
    const size_t lds = keys_lds;
    HIP_TRY(c, hipEventRecord(c->ev_copy[0], c->stream));            /* (the memset above) */
    HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, c->ev_copy[0], 0));
    static const uint64_t zero_off[1] = {0};
    if (n == 0)
      HIP_TRY(c, hipMemcpyAsync(at(o_off), zero_off, sizeof(uint64_t), hipMemcpyHostToDevice, c->copy_stream));
    uint64_t rq[5] = {0, 0, 0, 0, 0};
    for (uint64_t r = 0; r < nranges; r++) {
      rq[r] = ranges[r].q0;
      rq[r + 1] = ranges[r].q1;
    }

    /* the narrowing threads: thread t takes the t-th part of every range, range by range */
    /* (a thread per >= 128k queries: starting 32 threads costs more than a 1M-query share's narrowing) */
    const unsigned T = narrow ? std::max(1u, std::min(std::min(32u, hw), (unsigned)(n >> 17) + 1u)) : 0u;
    uint16_t *h_len16 = nullptr, *h_rep16 = nullptr, *h_v16 = nullptr, *h_j16 = nullptr;
    uint32_t *h_cnt32 = nullptr;
    std::atomic<uint32_t> done[4];
    std::atomic<uint32_t> misfit(0);
    for (auto &d : done)
      d.store(0);
    std::vector<std::thread> workers;
    struct Join {
      std::vector<std::thread> &w;
      ~Join() { for (auto &t : w) if (t.joinable()) t.join(); }
    } join{workers};
    if (narrow) {
      const size_t need = (size_t)(n + 1) * 2 + (size_t)n * (2 + 2 + 2 + 4) + 64;
      if (c->stage_host_bytes < need) {
        if (c->stage_host)
          (void)hipHostFree(c->stage_host);
        c->stage_host = nullptr;
        c->stage_host_bytes = 0;
        size_t got = 0;
        if (void *p = cmpr_take_reserved_host(need, &got)) {       /* (cmpr_warm_up_sized page-locked it ahead) */
          c->stage_host = p;
          c->stage_host_bytes = got;
        } else {
          HIP_TRY(c, hipHostMalloc(&c->stage_host, need + need / 16, hipHostMallocDefault));
          c->stage_host_bytes = need + need / 16;
        }
      }
      char *hp = (char *)c->stage_host;
      h_len16 = (uint16_t *)hp; hp += (((size_t)(n + 1) * 2) + 15) & ~(size_t)15;
      h_rep16 = (uint16_t *)hp; hp += ((size_t)n * 2 + 15) & ~(size_t)15;
      h_v16 = (uint16_t *)hp;   hp += ((size_t)n * 2 + 15) & ~(size_t)15;
      h_j16 = (uint16_t *)hp;   hp += ((size_t)n * 2 + 15) & ~(size_t)15;
      h_cnt32 = (uint32_t *)hp;
      h_len16[n] = 0;
      const bool genes = !c->opt.ignore_genes, counts = !c->opt.ignore_counts;
      for (unsigned t = 0; t < T; t++)
        workers.emplace_back([=, &done, &misfit]() {
          for (uint64_t r = 0; r < nranges; r++) {
            const uint64_t a = rq[r] + (rq[r + 1] - rq[r]) * t / T, b = rq[r] + (rq[r + 1] - rq[r]) * (t + 1) / T;
            uint32_t bad = 0;
            for (uint64_t i = a; i < b; i++) {
              const uint64_t o0 = s->offsets[i], o1 = s->offsets[i + 1];
              const uint64_t len = o1 - o0;
              bad |= (o1 < o0) | (len > 0xffffu);
              h_len16[i] = (uint16_t)len;
              const uint32_t rp = s->repertoire[i];
              bad |= rp > 0xffffu;
              h_rep16[i] = (uint16_t)rp;
            }
            if (genes)
              for (uint64_t i = a; i < b; i++) {
                const uint32_t vv = s->v_gene[i], jj = s->j_gene[i];
                bad |= (vv > 0xffffu) | (jj > 0xffffu);
                h_v16[i] = (uint16_t)vv;
                h_j16[i] = (uint16_t)jj;
              }
            if (counts)
              for (uint64_t i = a; i < b; i++) {
                const uint64_t x = s->count[i];
                bad |= x > 0xffffffffull;
                h_cnt32[i] = (uint32_t)x;
              }
            if (bad)
              misfit.store(1);
            done[r].fetch_add(1, std::memory_order_release);
          }
        });
    }

    bool redo_wide = false;
    for (uint64_t r = 0; r < nranges && n; r++) {
      const uint64_t q0 = rq[r], q1 = rq[r + 1];
      const uint64_t r0 = s->offsets[q0], r1 = s->offsets[q1];
      if (r1 < r0 || r1 > total)
        return fail(c, CMPR_EINVAL, verr_message(VERR_OFFSETS));
      auto t0 = std::chrono::steady_clock::now();
      hipStream_t cs = c->copy_stream;
      if (r1 > r0)
        HIP_TRY(c, hipMemcpyAsync(at(o_res) + r0, s->residues + r0, (size_t)(r1 - r0), hipMemcpyHostToDevice, cs));
      if (narrow) {
        upload_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        while (done[r].load(std::memory_order_acquire) < T)
          std::this_thread::yield();
        if (misfit.load()) {
          redo_wide = true;
          break;
        }
        t0 = std::chrono::steady_clock::now();
        HIP_TRY(c, hipMemcpyAsync(at(o_len16) + q0 * 2, h_len16 + q0, (size_t)(q1 - q0 + 1) * 2, hipMemcpyHostToDevice, cs));
        HIP_TRY(c, hipMemcpyAsync(at(o_rep16) + q0 * 2, h_rep16 + q0, (size_t)(q1 - q0) * 2, hipMemcpyHostToDevice, cs));
        if (!c->opt.ignore_genes) {
          HIP_TRY(c, hipMemcpyAsync(at(o_v16) + q0 * 2, h_v16 + q0, (size_t)(q1 - q0) * 2, hipMemcpyHostToDevice, cs));
          HIP_TRY(c, hipMemcpyAsync(at(o_j16) + q0 * 2, h_j16 + q0, (size_t)(q1 - q0) * 2, hipMemcpyHostToDevice, cs));
        }
        if (!c->opt.ignore_counts)
          HIP_TRY(c, hipMemcpyAsync(at(o_cnt32) + q0 * 4, h_cnt32 + q0, (size_t)(q1 - q0) * 4, hipMemcpyHostToDevice, cs));
      } else {
        HIP_TRY(c, hipMemcpyAsync(at(o_off) + q0 * sizeof(uint64_t), s->offsets + q0,
                                  (size_t)(q1 - q0 + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, cs));
        HIP_TRY(c, hipMemcpyAsync(at(o_rep) + q0 * sizeof(uint32_t), s->repertoire + q0,
                                  (size_t)(q1 - q0) * sizeof(uint32_t), hipMemcpyHostToDevice, cs));
        if (!c->opt.ignore_genes) {
          HIP_TRY(c, hipMemcpyAsync(at(o_v) + q0 * sizeof(uint32_t), s->v_gene + q0,
                                    (size_t)(q1 - q0) * sizeof(uint32_t), hipMemcpyHostToDevice, cs));
          HIP_TRY(c, hipMemcpyAsync(at(o_j) + q0 * sizeof(uint32_t), s->j_gene + q0,
                                    (size_t)(q1 - q0) * sizeof(uint32_t), hipMemcpyHostToDevice, cs));
        }
        if (!c->opt.ignore_counts)
          HIP_TRY(c, hipMemcpyAsync(at(o_cnt) + q0 * sizeof(uint64_t), s->count + q0,
                                    (size_t)(q1 - q0) * sizeof(uint64_t), hipMemcpyHostToDevice, cs));
      }
      hipEvent_t ev = c->ev_copy[r % cmpr_context::NCOPY_EV];
      HIP_TRY(c, hipEventRecord(ev, cs));
      upload_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      HIP_TRY(c, hipStreamWaitEvent(c->stream, ev, 0));
      if (narrow) {
        /* back to offsets (scan of the lengths from the range's first offset: its last
           output is the next range's first offset) and to the caller's widths */
        hipcub::TransformInputIterator<unsigned long long, Len16To64, const uint16_t *> it(
            (const uint16_t *)at(o_len16) + q0, Len16To64());
        size_t b = cub_bytes;
        HIP_TRY(c, hipcub::DeviceScan::ExclusiveScan(cub_tmp, b, it, (unsigned long long *)at(o_off) + q0,
                                                     hipcub::Sum(), (unsigned long long)r0, (int)(q1 - q0 + 1),
                                                     c->stream));
        hipLaunchKernelGGL(widen_kernel, dim3(blocks_for(q1 - q0)), dim3(256), 0, c->stream,
                           (const uint16_t *)at(o_v16), (const uint16_t *)at(o_j16), (const uint16_t *)at(o_rep16),
                           (const uint32_t *)at(o_cnt32),
                           c->opt.ignore_genes ? nullptr : (uint32_t *)at(o_v),
                           c->opt.ignore_genes ? nullptr : (uint32_t *)at(o_j), (uint32_t *)at(o_rep),
                           c->opt.ignore_counts ? nullptr : (uint64_t *)at(o_cnt), q0, q1);
        HIP_TRY(c, hipGetLastError());
      }
      hipLaunchKernelGGL(keys_fn, dim3(ranges[r].grid), dim3(LAYOUT_WG), lds, c->stream, Q, q0, q1, ranges[r].row0);
      HIP_TRY(c, hipGetLastError());
    }
    if (n == 0) {
      HIP_TRY(c, hipEventRecord(c->ev_copy[0], c->copy_stream));
      HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_copy[0], 0));
    }
    if (!redo_wide)
      break;
    /* something does not fit the narrow types: once more, the caller's arrays as they are */
    for (auto &t : workers)
      t.join();
    HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemsetAsync(base, 0, zero_bytes, c->stream));
    narrow = false;
    upload_ms = 0;
    if (attempt)
      return fail(c, CMPR_ESTATE, "upload did not settle");
  }
  const auto t_uploaded = std::chrono::steady_clock::now();

  if (routing) {
    /* cmpr_route_queries ends here: records per destination, the totals of the share, and
       the keyed share left in arena A for cmpr_route_pack */
    uint32_t hv[2] = {0, 0};
    unsigned long long counts[64];
    std::vector<double> tot(n_rep, 0.0);
    HIP_TRY(c, hipMemcpyAsync(hv, Q.verr, sizeof hv, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(counts, Q.dest_cnt, sizeof counts, hipMemcpyDeviceToHost, c->stream));
    if (n_rep)
      HIP_TRY(c, hipMemcpyAsync(tot.data(), Q.rep_total, n_rep * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (hv[0] == VERR_TOO_LONG)
      return fail(c, CMPR_EINVAL, "query longer than the longest_query given to cmpr_set_reference");
    if (hv[0])
      return fail(c, hv[0] == VERR_LONG ? CMPR_EUNSUPPORTED : CMPR_EINVAL, verr_message(hv[0]));
    RouteState &R = c->route;
    R.n = n;
    R.first_index = src.first_index;
    R.n_dest = wstep;
    R.record_bytes = record_bytes;
    R.total_records = 0;
    for (uint32_t d = 0; d < 64; d++) {
      R.counts[d] = d < wstep ? counts[d] : 0;
      R.total_records += R.counts[d];
    }
    R.rep_totals = tot;
    R.res = Q.res; R.off = Q.off; R.v = Q.v; R.j = Q.j; R.rep = Q.rep; R.cnt = Q.cnt;
    R.mask_lo = Q.grp[0];
    R.mask_hi = Q.rank;
    R.dest = Q.dest_cnt;
    R.valid = true;
    c->layout_upload_ms = upload_ms;
    c->layout_total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    c->layout_tail_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_uploaded).count();
    if (getenv("COMPAIRR_HIP_DEBUG"))
      fprintf(stderr, "compairr_hip: route_queries %.2f ms = arena A %.2f + upload (keys hidden) %.2f + counts %.2f  "
                      "(copy calls %.2f; %llu queries, %llu records)\n", c->layout_total_ms,
              std::chrono::duration<double, std::milli>(t_arena_a - t_begin).count(),
              std::chrono::duration<double, std::milli>(t_uploaded - t_arena_a).count(), c->layout_tail_ms, upload_ms,
              (unsigned long long)n, (unsigned long long)R.total_records);
    quiesce.ok = true;
    return CMPR_OK;
  }

  /* (slices_kernel: the counters of a workgroup's slices in LDS when they fit 48 KiB) */
  const size_t slices_lds_want = (size_t)SLICES_WG * (c->sliced ? 2 : 1) * per_slice * sizeof(uint32_t);
  const size_t slices_lds = slices_lds_want <= 48 * 1024 ? slices_lds_want : 0;
  /* ---- per slice: what it needs; exclusive scan; items per (group, slice) padded to
          blocks of 64; -i (variant 1): the sibling lists; all sizes in ONE round trip ---- */
  {
    SliceTot zero;
    memset(&zero, 0, sizeof zero);
    hipLaunchKernelGGL(slices_kernel<0>, dim3((uint32_t)((nslices + SLICES_WG - 1) / SLICES_WG)), dim3(SLICES_WG),
                       slices_lds, c->stream, Q, 0u, slices_lds ? 1u : 0u);
    HIP_TRY(c, hipGetLastError());
    size_t b = cub_bytes;
    HIP_TRY(c, hipcub::DeviceScan::ExclusiveScan(cub_tmp, b, Q.tot[0], Q.pre[0], SliceTotSum(), zero,
                                                 (int)nslices, c->stream));
  }
  if (ngroups) {
    hipLaunchKernelGGL(item_replicas_kernel, dim3(blocks_for(ncs)), dim3(256), 0, c->stream, Q);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(class_pad_kernel, dim3(blocks_for(ncs)), dim3(256), 0, c->stream, Q, cpad);
    HIP_TRY(c, hipGetLastError());
    /* (their two prefix sums and totals: sizes_kernel, below) */
  }
  if (indel_passes) {
    hipLaunchKernelGGL(sibling_count_kernel, dim3(blocks_for(G)), dim3(256), 0, c->stream, Q, G, sib_cnt);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(sibling_chunks_kernel, dim3(blocks_for(n2s)), dim3(256), 0, c->stream,
                       sib_cnt, n2s, (uint32_t)chunk_tiles, sib_nch);
    HIP_TRY(c, hipGetLastError());
    size_t b = cub_bytes;
    HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(cub_tmp, b, sib_cnt, sib_lpre, (int)n2s, c->stream));
    b = cub_bytes;
    HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(cub_tmp, b, sib_nch, sib_cpre, (int)n2s, c->stream));
    b = cub_bytes;
    HIP_TRY(c, sum64(cub_tmp, b, sib_cnt, sums + 2, (int)n2s, c->stream));
    b = cub_bytes;
    HIP_TRY(c, sum64(cub_tmp, b, sib_nch, sums + 3, (int)n2s, c->stream));
  }
  /* the item lists' prefix sums, and everything the host waits for in one block, one copy (pinned) */
  hipLaunchKernelGGL(sizes_kernel, dim3(1), dim3(1024), 0, c->stream, Q, cpad, ngroups ? ncs : (uint64_t)0, sums,
                     (SizesBlock *)at(o_blk));
  HIP_TRY(c, hipGetLastError());
  if (c->h_sizes_bytes < blk_bytes) {
    if (c->h_sizes)
      (void)hipHostFree(c->h_sizes);
    c->h_sizes = nullptr;
    c->h_sizes_bytes = 0;
    HIP_TRY(c, hipHostMalloc(&c->h_sizes, blk_bytes + 65536, hipHostMallocDefault));
    c->h_sizes_bytes = blk_bytes + 65536;
  }
  HIP_TRY(c, hipMemcpyAsync(c->h_sizes, at(o_blk), blk_bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const SizesBlock hb = *(const SizesBlock *)c->h_sizes;
  const SliceTot last_tot = hb.last_tot, last_pre = hb.last_pre;
  const uint32_t hv[2] = {hb.verr[0], hb.verr[1]};
  const unsigned long long alg_bytes = hb.alg;
  const unsigned long long hsums[4] = {hb.sums[0], hb.sums[1], hb.sums[2], hb.sums[3]};
  {
    const double *rt = (const double *)((const SizesBlock *)c->h_sizes + 1);
    c->tot1.assign(rt, rt + n_rep);
  }
  const auto t_sizes = std::chrono::steady_clock::now();
  if (hv[0] == VERR_TOO_LONG)
    return fail(c, CMPR_EINVAL, "query longer than the longest_query given to cmpr_set_reference");
  if (hv[0])
    return fail(c, hv[0] == VERR_LONG ? CMPR_EUNSUPPORTED : CMPR_EINVAL, verr_message(hv[0]));
  c->algorithmic_bytes = alg_bytes;
  /* (routed records: the totals of the WHOLE query set, when the caller has them, bound the
     cells of the summed matrix; this context saw only its share) */
  if (from_records && src.totals)
    c->tot1.assign(src.totals, src.totals + n_rep);

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

  uint64_t ntiles = 0, nchunks = 0, nlist = 0, nsmall = 0, res_words = 0;
  {
    const SliceTotSum add;
    const SliceTot t = add(last_pre, last_tot);
    Q.tile0[0] = Q.chunk0[0] = Q.list0[0] = Q.small0[0] = 0;
    Q.res0[0] = 0;
    ntiles = t.tiles;
    nchunks = t.chunks;
    nlist = t.list;
    nsmall = t.small;
    res_words = t.res;
    c->nmain_tiles = t.tiles;
  }
  const uint64_t cslots = hsums[0], class_chunks = hsums[1];
  if (ntiles * WAVE >= 0xffffffffull)
    return fail(c, CMPR_EUNSUPPORTED, "too many query tiles");
  if (res_words + 9 * WAVE >= 0xffffffffull || cslots + WAVE >= 0xffffffffull)
    return fail(c, CMPR_EUNSUPPORTED, "query set too large for 32-bit residue positions");
  const uint64_t main_chunks = nchunks, main_list = nlist;
  const uint64_t sib_list = hsums[2], sib_chunks = hsums[3];
  nchunks += sib_chunks;
  nlist += sib_list;
  Q.cchunk0 = (uint32_t)nchunks;
  nchunks += class_chunks;
  if (nchunks >= 0x7fffffffull || nlist >= 0xffffffffull)
    return fail(c, CMPR_EUNSUPPORTED, "too many chunks");

  /* ---- arena B: the temporaries sized by the device ---- */
  const size_t slots = (size_t)ntiles * WAVE;
  size_t sort_bytes = 0, small_sort_bytes = 0;
  if (nchunks)
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, sort_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                                       (uint32_t *)nullptr, (uint32_t *)nullptr, (int)nchunks, 0, 32,
                                                       c->stream);
  if (nsmall)
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, small_sort_bytes, (uint32_t *)nullptr,
                                                       (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                                       (int)nsmall, 0, 17, c->stream);
  ArenaCut cb;
  const size_t o_aux = cb.take(std::max<size_t>(Q.recompute ? 1 : slots, 1) * sizeof(QAux));
  const size_t o_chunks_u = cb.take((size_t)nchunks * sizeof(Chunk));
  const size_t o_wk = cb.take(((size_t)nchunks + 1) * sizeof(uint32_t));
  const size_t o_wk2 = cb.take((size_t)nchunks * sizeof(uint32_t));
  const size_t o_ix = cb.take((size_t)nchunks * sizeof(uint32_t));
  const size_t o_ix2 = cb.take((size_t)nchunks * sizeof(uint32_t));
  const size_t o_sort = cb.take(sort_bytes + 256);
  const size_t o_ln = cb.take((size_t)nsmall * sizeof(uint32_t));
  const size_t o_ln2 = cb.take((size_t)nsmall * sizeof(uint32_t));
  const size_t o_sout = cb.take((size_t)nsmall * sizeof(uint32_t));
  const size_t o_ssort = cb.take(small_sort_bytes + 256);
  if ((rc = arena_fit(c, c->arena_b, cb.used))) return rc;
  char *const bb = c->arena_b.base;
  Chunk *const chunks_unsorted = (Chunk *)(bb + o_chunks_u);
  Q.aux = (QAux *)(bb + o_aux);

  /* ---- the resident layout (reallocated only when it has to grow; written in full by
          fill_tiles_kernel, padding included) ---- */
  c->ntiles = (uint32_t)ntiles;
  c->nchunks = (uint32_t)nchunks;
  c->nsmall = (uint32_t)nsmall;
  if ((rc = dev_reserve(c, c->tiles, (size_t)ntiles))) return rc;
  if ((rc = dev_reserve(c, c->chunks, (size_t)nchunks))) return rc;
  if ((rc = dev_reserve(c, c->tile_refs, (size_t)nlist))) return rc;
  if ((rc = dev_reserve(c, c->small_tiles, (size_t)nsmall))) return rc;
  /* + 9 rows: slack behind the last tile for readers that fetch nine dwords per query */
  if ((rc = dev_reserve(c, c->qres, (size_t)res_words + 9 * WAVE))) return rc;
  if ((rc = dev_reserve(c, c->qlen, slots))) return rc;
  if ((rc = dev_reserve(c, c->qrec, slots))) return rc;
  HIP_TRY(c, hipMemsetAsync(c->qres.p + res_words, 0, 9 * WAVE * sizeof(uint32_t), c->stream));
  if (!c->opt.ignore_genes && !c->rows) {
    if ((rc = dev_reserve(c, c->qv, slots))) return rc;
    if ((rc = dev_reserve(c, c->qj, slots))) return rc;
  } else {
    c->qv.release();
    c->qj.release();
  }
  if (!c->opt.ignore_genes || c->rows || (!c->sliced && c->opt.differences == 0)) {
    if ((rc = dev_reserve(c, c->qgh, slots))) return rc;
  } else {
    c->qgh.release();
  }
  if (c->rows) {
    if ((rc = dev_reserve(c, c->qck, slots))) return rc;
  } else {
    c->qck.release();
  }
  if (c->d2pairs) {
    if ((rc = dev_reserve(c, c->qpk, slots))) return rc;
  } else {
    c->qpk.release();
  }
  if (c->rows && c->opt.indels) {
    if ((rc = dev_reserve(c, c->qhins, slots))) return rc;
    if ((rc = dev_reserve(c, c->qhdel, slots))) return rc;
  } else {
    c->qhins.release();
    c->qhdel.release();
  }
  bool pad_items = false;
  if (ngroups) {
    /* (+ 64: a block read past the last item stays inside; all ones = padding) */
    const size_t ni = (size_t)cslots + WAVE;
    if ((rc = dev_reserve(c, c->items, ni))) return rc;
    pad_items = true;                      /* (item_padding_kernel, once Q.items is known) */
    if (sub2_items) {
      if ((rc = dev_reserve(c, c->cpk, ni))) return rc;
      if ((rc = dev_reserve(c, c->slice_items, 2 * (size_t)nslices_real))) return rc;
      HIP_TRY(c, hipMemsetAsync(c->slice_items.p, 0, 2 * (size_t)nslices_real * sizeof(uint32_t), c->stream));
    } else if (c->d2pairs) {
      if ((rc = dev_reserve(c, c->cpk, ni))) return rc;
      c->slice_items.release();
    } else {
      c->cpk.release();
      c->slice_items.release();
    }
  } else {
    c->items.release();
    c->cpk.release();
    c->slice_items.release();
  }
  const auto t_reserved = std::chrono::steady_clock::now();
  Q.tiles = c->tiles.p;
  Q.chunks = chunks_unsorted;
  Q.tile_refs = c->tile_refs.p;
  Q.small_tiles = c->small_tiles.p;
  Q.cpk = c->cpk.p;
  Q.qpk = c->qpk.p;
  Q.slice_items = (uint2 *)c->slice_items.p;
  Q.qres = c->qres.p; Q.qv = c->qv.p; Q.qj = c->qj.p;
  Q.qck = c->qck.p; Q.qgh = c->qgh.p; Q.qhins = c->qhins.p;
  Q.qhdel = c->qhdel.p; Q.qlen = c->qlen.p; Q.qrec = c->qrec.p;
  Q.items = c->items.p;

  if (pad_items) {
    hipLaunchKernelGGL(item_padding_kernel, dim3(blocks_for(std::max<uint64_t>(ncs, WAVE))), dim3(256), 0, c->stream, Q,
                       cpad, (uint32_t)cslots);
    HIP_TRY(c, hipGetLastError());
  }
  hipLaunchKernelGGL(slices_kernel<1>, dim3((uint32_t)((nslices + SLICES_WG - 1) / SLICES_WG)), dim3(SLICES_WG),
                     slices_lds, c->stream, Q, 0u, slices_lds ? 1u : 0u);
  HIP_TRY(c, hipGetLastError());
  LAYOUT_MARK(2);
  if (n) {
    /* (the grid of every range's keys_kernel once more: QL::item_wg) */
    for (const Range &g : ranges) {
      hipLaunchKernelGGL(scatter_fn, dim3(g.grid), dim3(LAYOUT_WG), scatter_lds, c->stream, Q, g.q0, g.q1, g.row0);
      HIP_TRY(c, hipGetLastError());
    }
  }
  LAYOUT_MARK(3);
  /* record tiles (layout.h ProbeParams::rec_tiles): amino acids on pair rows at d = 1 without -i, every sequence of
     both sets within 32 residues (a hit is verified from its 64-byte slot and the query's record alone) -- the probe
     kernel reads the records, no per-slot array is written */
  /* (with -i a hit is one residue longer or shorter than the query: 31) */
  /* (d = 0: QL::direct == 2 -- scatter_kernel has stored the hashes) */
  c->rec_tiles = rows_rec;
  c->rec_hash = c->rec_tiles && rec_hash;
  if (ntiles && !c->rec_tiles) {
    const size_t fill_lds = Q.recompute ? (zob_lds ? zob_words * sizeof(uint64_t) : 0) + ctab_bytes : 0;
    const int fo = occupancy_of(c, (const void *)fill_tiles_kernel, 256, fill_lds);
    const uint32_t fgrid = (uint32_t)std::min<uint64_t>((ntiles + 3) / 4, (uint64_t)c->cus * (uint64_t)std::max(fo, 1));
    hipLaunchKernelGGL(fill_tiles_kernel, dim3(fgrid), dim3(256), fill_lds, c->stream, Q, (uint32_t)ntiles);
    HIP_TRY(c, hipGetLastError());
  }
  LAYOUT_MARK(4);
  if (ngroups) {
    hipLaunchKernelGGL(class_chunks_kernel, dim3(blocks_for(ncs)), dim3(256), 0, c->stream, Q);
    HIP_TRY(c, hipGetLastError());
  }
  if (indel_passes) {
    hipLaunchKernelGGL(sibling_fill_kernel, dim3(blocks_for(G)), dim3(256), 0, c->stream, Q, G,
                       sib_lpre, sib_fill, (uint32_t)main_list);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(sibling_write_chunks_kernel, dim3(blocks_for(2 * nslices)), dim3(256), 0,
                       c->stream, Q, sib_cnt, sib_lpre, sib_cpre, (uint32_t)main_list,
                       (uint32_t)main_chunks);
    HIP_TRY(c, hipGetLastError());
  }

  /* what is laid out is what this context works on -- except for variant 1 with -i,
     where it takes its share of the full lists */
  const bool select_share = wstep > 1 && !place_mine;
  /* ---- heaviest chunks first: the tail of the launch is made of light ones;
          single-wave tiles longest first ---- */
  if (nchunks && nchunks <= ORDER_MAX && !select_share) {
    uint32_t *wk = (uint32_t *)(bb + o_wk);
    HIP_TRY(c, hipMemsetAsync(wk + nchunks, 0, sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(chunk_work_kernel, dim3(blocks_for(nchunks)), dim3(256), 0, c->stream, Q,
                       (uint32_t)nchunks, wk, (uint32_t *)nullptr);
    hipLaunchKernelGGL(chunk_order_kernel, dim3(1), dim3(1024), 0, c->stream, Q, (uint32_t)nchunks, wk, c->chunks.p);
    HIP_TRY(c, hipGetLastError());
    c->nchunks = (uint32_t)nchunks;
  } else if (nchunks) {
    uint32_t *wk = (uint32_t *)(bb + o_wk), *wk2 = (uint32_t *)(bb + o_wk2);
    uint32_t *ix = (uint32_t *)(bb + o_ix), *ix2 = (uint32_t *)(bb + o_ix2);
    hipLaunchKernelGGL(chunk_work_kernel, dim3(blocks_for(nchunks)), dim3(256), 0, c->stream, Q,
                       (uint32_t)nchunks, wk, ix);
    HIP_TRY(c, hipGetLastError());
    size_t sb = sort_bytes;
    HIP_TRY(c, hipcub::DeviceRadixSort::SortPairsDescending(bb + o_sort, sb, wk, wk2, ix, ix2, (int)nchunks, 0, 32,
                                                            c->stream));
    uint64_t mine = nchunks;
    const uint32_t *order = ix2;
    Tmp<uint32_t> sel;
    if (select_share) {
      /* the context's share of the sorted list */
      if ((rc = select_mine(c, ix2, (uint32_t)nchunks, sel, mine, [&](unsigned char *flag) {
            hipLaunchKernelGGL(chunk_mine_kernel, dim3(blocks_for(nchunks)), dim3(256), 0, c->stream,
                               chunks_unsorted, ix2, (uint32_t)nchunks, wfirst, wstep, flag);
          })))
        return rc;
      order = sel.b.p;
    }
    if (mine)
      hipLaunchKernelGGL(gather_chunks_kernel, dim3(blocks_for(mine)), dim3(256), 0, c->stream,
                         chunks_unsorted, order, (uint32_t)mine, c->chunks.p);
    HIP_TRY(c, hipGetLastError());
    if (select_share)
      HIP_TRY(c, hipStreamSynchronize(c->stream));       /* (`sel` is freed on return) */
    c->nchunks = (uint32_t)mine;
  }
  if (nsmall) {
    uint32_t *ln = (uint32_t *)(bb + o_ln), *ln2 = (uint32_t *)(bb + o_ln2), *out = (uint32_t *)(bb + o_sout);
    hipLaunchKernelGGL(small_len_kernel, dim3(blocks_for(nsmall)), dim3(256), 0, c->stream, Q,
                       (uint32_t)nsmall, ln);
    HIP_TRY(c, hipGetLastError());
    size_t sb = small_sort_bytes;
    HIP_TRY(c, hipcub::DeviceRadixSort::SortPairsDescending(bb + o_ssort, sb, ln, ln2, c->small_tiles.p, out,
                                                            (int)nsmall, 0, 17, c->stream));
    uint64_t mine = nsmall;
    const uint32_t *keep = out;
    Tmp<uint32_t> sel;
    if (select_share) {
      if ((rc = select_mine(c, out, (uint32_t)nsmall, sel, mine, [&](unsigned char *flag) {
            hipLaunchKernelGGL(small_mine_kernel, dim3(blocks_for(nsmall)), dim3(256), 0, c->stream, Q,
                               out, (uint32_t)nsmall, wfirst, wstep, flag);
          })))
        return rc;
      keep = sel.b.p;
    }
    if (mine)
      HIP_TRY(c, hipMemcpyAsync(c->small_tiles.p, keep, (size_t)mine * sizeof(uint32_t),
                                hipMemcpyDeviceToDevice, c->stream));
    if (select_share)
      HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->nsmall = (uint32_t)mine;
  }
  LAYOUT_MARK(5);
  if (src.finish && (rc = src.finish()))
    return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const auto t_end = std::chrono::steady_clock::now();
#undef LAYOUT_MARK
  for (float &x : c->layout_kernel_ms)
    x = 0.0f;
  if (c->layout_marks == 0x3fu) {
    /* keys | sizes (per-slice needs, scans, the host's round trip) | slices + scatter | tiles | chunk order */
    for (int k = 0; k < 5; k++)
      (void)hipEventElapsedTime(&c->layout_kernel_ms[k], c->ev_layout[k], c->ev_layout[k + 1]);
  }
  c->layout_upload_ms = upload_ms;
  c->layout_tail_ms = std::chrono::duration<double, std::milli>(t_end - t_uploaded).count();
  c->layout_total_ms = std::chrono::duration<double, std::milli>(t_end - t_begin).count();
  if (getenv("COMPAIRR_HIP_DEBUG")) {
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double, std::milli>(b - a).count();
    };
    fprintf(stderr, "compairr_hip: set_queries %.2f ms = arena A %.2f + upload (keys hidden) %.2f + sizes %.2f + "
                    "arena B / resident %.2f + placement %.2f  (copy calls %.2f; %zu + %zu MiB of temporaries)\n",
            c->layout_total_ms, ms(t_begin, t_arena_a), ms(t_arena_a, t_uploaded), ms(t_uploaded, t_sizes),
            ms(t_sizes, t_reserved), ms(t_reserved, t_end), upload_ms, cut.used >> 20, cb.used >> 20);
  }
  quiesce.ok = true;
  return CMPR_OK;
}


/* cmpr_route_pack: the records of the share keyed by cmpr_route_queries, grouped by
   destination (destination d's run starts at the sum of the counts below d), into the
   caller's device buffer. */
int cmpr_route_pack_impl(cmpr_context *c, void *d_send, uint64_t capacity_bytes)
{
  RouteState &R = c->route;
  if (!R.valid)
    return fail(c, CMPR_ESTATE, "cmpr_route_queries must be called first");
  if (capacity_bytes < R.total_records * R.record_bytes)
    return fail(c, CMPR_EINVAL, "cmpr_route_pack: the buffer is smaller than the records counted");
  if (R.total_records && !d_send)
    return fail(c, CMPR_EINVAL, "cmpr_route_pack: d_send is NULL");
  HIP_TRY(c, hipSetDevice(c->device));
  R.valid = false;                          /* (the share is packed once) */
  if (R.n == 0)
    return CMPR_OK;
  unsigned long long base[128];
  unsigned long long at = 0;
  for (uint32_t d = 0; d < 64; d++) {
    base[d] = at;                           /* first record of the destination */
    at += R.counts[d];
    base[64 + d] = 0;                       /* fill cursors */
  }
  HIP_TRY(c, hipMemcpyAsync(R.dest, base, sizeof base, hipMemcpyHostToDevice, c->stream));
  QL Q;
  memset(&Q, 0, sizeof Q);
  Q.res = R.res; Q.off = R.off; Q.v = R.v; Q.j = R.j; Q.rep = R.rep; Q.cnt = R.cnt;
  Q.n = R.n;
  Q.genes = c->opt.ignore_genes ? 0 : 1;
  Q.counts = c->opt.ignore_counts ? 0 : 1;
  hipLaunchKernelGGL(route_pack_kernel, dim3(blocks_for(R.n)), dim3(256), 0, c->stream, Q, R.mask_lo, R.mask_hi,
                     R.first_index, R.dest, R.dest + 64, (unsigned char *)d_send, R.record_bytes);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return CMPR_OK;
}

void cmpr_touch_layout_kernels()
{
  hipFuncAttributes fa;
  (void)hipFuncGetAttributes(&fa, (const void *)fill_tiles_kernel);
}
