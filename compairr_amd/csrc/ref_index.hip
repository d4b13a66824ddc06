/*
 * ref_index.hip -- cmpr_set_reference on the device: upload of set 2, its
 * validation, the geometry of the sliced filter (class positions, heavy classes,
 * number of class residues), record positions, hash table, filter and records.
 *
 * What the reference does serially before its per-query loop (db_hash,
 * db.cc:903-916; hash_init / bloom_init / hash_insert, overlap.cc:861-873,
 * hashtable.cc:31-54) is one upload and a handful of kernels here; the host
 * touches nothing per sequence, only histograms of a few thousand counters.
 */
#include "context.h"
#include "kernels_rows.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>

using namespace cmpr;

namespace {

inline uint32_t blocks_for(uint64_t n)
{
  return (uint32_t)std::max<uint64_t>(1, (n + 255) / 256);
}

template <typename T>
struct Tmp {
  DevBuf<T> b;
  ~Tmp() { b.release(); }
};

/* lengths: one counter per length (LDS copy per workgroup when they fit) */
__global__ void __launch_bounds__(256)
length_hist_kernel(const uint64_t *off, uint64_t n, uint32_t nh, uint32_t *hist)
{
  extern __shared__ uint32_t h_lds[];
  const bool lds = nh <= 8192;
  if (lds) {
    for (uint32_t k = threadIdx.x; k < nh; k += 256)
      h_lds[k] = 0;
    __syncthreads();
  }
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    const uint32_t L = (uint32_t)(off[i + 1] - off[i]);
    atomicAdd((lds ? h_lds : hist) + min(L, nh - 1), 1u);
  }
  if (lds) {
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < nh; k += 256)
      if (h_lds[k])
        atomicAdd(hist + k, h_lds[k]);
  }
}

/* residue counts per position over a sample of the set (every stride-th sequence) */
__global__ void __launch_bounds__(256)
residue_sample_kernel(const uint8_t *res, const uint64_t *off, uint64_t n, uint64_t stride,
                      uint32_t npos, uint32_t A, uint32_t *cnt)
{
  /* (counted in LDS, a workgroup's sums added once: 3 x 10^6 atomics on 300 addresses of HBM took 2 ms) */
  extern __shared__ uint32_t c_lds[];
  const uint32_t nc = npos * A;
  const bool lds = nc <= 8192;
  if (lds) {
    for (uint32_t k = threadIdx.x; k < nc; k += 256)
      c_lds[k] = 0;
    __syncthreads();
  }
  for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k * stride < n; k += (uint64_t)gridDim.x * 256) {
    const uint64_t i = k * stride;
    const uint64_t b = off[i];
    const uint32_t L = (uint32_t)min((uint64_t)npos, off[i + 1] - b);
    for (uint32_t p = 0; p < L; p++)
      atomicAdd((lds ? c_lds : cnt) + (size_t)p * A + res[b + p], 1u);
  }
  if (lds) {
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < nc; k += 256)
      if (c_lds[k])
        atomicAdd(cnt + k, c_lds[k]);
  }
}

/* (length, V, J) class key of every sequence and the population of its bucket */
__global__ void __launch_bounds__(256)
class_base_kernel(const uint64_t *off, const uint32_t *v, const uint32_t *j, uint64_t n,
                  SliceGeom g, uint32_t genes, uint32_t *base_of, uint32_t *bucket)
{
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n)
    return;
  const uint32_t L = (uint32_t)(off[i + 1] - off[i]);
  const uint32_t b = class_base(g.ctab, g, genes != 0, L, genes ? v[i] : 0u, genes ? j[i] : 0u);
  base_of[i] = b;
  atomicAdd(bucket + (b >> (32 - HEAVY_BUCKETS_LOG2)), 1u);
}

/* sequences per slice if the heavy classes are split by k class residues */
__global__ void __launch_bounds__(256)
slice_population_kernel(const uint8_t *res, const uint64_t *off, const uint32_t *base_of, uint64_t n,
                        SliceGeom g, uint32_t A, uint32_t k, uint32_t *pop)
{
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n)
    return;
  const uint64_t b = off[i];
  const uint32_t L = (uint32_t)(off[i + 1] - b);
  uint32_t ck = base_of[i];
  if (L > 0 && class_is_heavy(g.ctab, g, ck))
    for (uint32_t r = 0; r < k; r++)
      ck ^= g.ctab[g.off_cr + r * A + res[b + class_pos(L, r, g.c0)]];
  atomicAdd(pop + (ck & g.smask), 1u);
}

/* ---- the record table (round 5): set 2 as an ORDERED linear-probing table of whole records ----
   The reference keeps an open-addressing table of (hash, sequence number) pairs and looks the
   sequence up behind it (hashtable.h:22-72, overlap.cc:168-251): two dependent random reads per
   hit, three with the query's record.  Here the 64-byte record of a sequence (layout.h RefRec)
   IS the table entry: bucket b of the sequence's key names the slot its record would lie in,
   a record that finds the slot taken lies in the next free one, and the records are placed in
   the order of their buckets -- record k of the bucket-sorted list at
       p_k = max(b_k, p_{k-1} + 1) = k + max_{j <= k} (b_j - j),
   a prefix maximum -- so that a lookup of bucket b starts at slot b, skips what was displaced
   there from earlier buckets (home < b), finds its own records side by side and stops at the first
   slot that is empty or belongs to a later bucket: ONE random memory line per hit at the load this
   table is built for (<= 0.35), the rare further slots next to it. */
__global__ void __launch_bounds__(256)
ref_keys_kernel(const BuildParams B, uint32_t bucket_mask, uint32_t *bucket, uint32_t *tag, uint32_t *seq)
{
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
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
  const uint64_t key = table_key(h);
  bucket[i] = dir_bucket(key, bucket_mask);
  tag[i] = dir_tag(key);
  seq[i] = (uint32_t)i;
}

/* b_k - k of the bucket-sorted list (both < 2^31) */
__global__ void __launch_bounds__(256)
bucket_lead_kernel(const uint32_t *sorted_bucket, uint64_t n, int32_t *lead)
{
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (k < n)
    lead[k] = (int32_t)sorted_bucket[k] - (int32_t)k;
}

struct MaxI32 {
  __host__ __device__ int32_t operator()(const int32_t &a, const int32_t &b) const { return a > b ? a : b; }
};

/* slot of the k-th record of the list, and "the next slot holds a record of the same bucket" */
__global__ void __launch_bounds__(256)
record_slots_kernel(const uint32_t *sorted_bucket, const int32_t *lead_max, const uint32_t *perm, uint64_t n,
                    uint32_t *slot_of_seq, uint32_t *more_of_seq, uint32_t *last_slot)
{
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n)
    return;
  const uint32_t p = (uint32_t)((int64_t)k + lead_max[k]);
  const uint32_t i = perm[k];
  slot_of_seq[i] = p;
  more_of_seq[i] = (k + 1 < n && sorted_bucket[k + 1] == sorted_bucket[k]) ? 1u : 0u;
  if (k == n - 1)
    *last_slot = p;
}

struct WidenU32 {
  __host__ __device__ unsigned long long operator()(const uint32_t &x) const { return x; }
};

}  // namespace

int cmpr_build_reference(cmpr_context *c, const cmpr_set_view *s, uint32_t longest_query, bool on_device)
{
  if (!c)
    return CMPR_EINVAL;
  std::string why;
  int rc = validate_view(c->opt, s, why, on_device);
  if (rc)
    return fail(c, rc, why);
  HIP_TRY(c, hipSetDevice(c->device));
  /* launches that still read the old index on a caller's stream */
  if (c->events_valid)
    HIP_TRY(c, hipEventSynchronize(c->ev_k1));
  c->have_ref = false;
  c->have_q = false;
  if (s->n > 0x7fffffffull)                               /* (hipCUB item counts are int) */
    return fail(c, CMPR_EUNSUPPORTED, "more than 2^31-1 sequences in one set");
  /* residues in all: offsets[n], which a device view keeps on the device */
  uint64_t residues2 = 0;
  if (s->n && !on_device) {
    residues2 = s->offsets[s->n];
  } else if (s->n) {
    uint64_t ends[2] = {0, 0};
    HIP_TRY(c, hipMemcpy(&ends[0], s->offsets, sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(&ends[1], s->offsets + s->n, sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (ends[0] != 0)
      return fail(c, CMPR_EINVAL, "offsets[0] must be 0");
    if (ends[1] > 0xffffull * s->n)
      return fail(c, CMPR_EINVAL, "offsets not monotone");
    residues2 = ends[1];
  }

  /* ---- the set as the caller has it, validated on the device ---- */
  uint32_t longest = 0;
  if ((rc = cmpr_upload_and_validate(c, s, c->res2, c->off2, c->v2, c->j2, c->rep2, c->cnt2, longest,
                                     c->tot2, on_device, residues2)))
    return rc;
  c->longest2 = longest;
  c->n2 = s->n;
  c->R2 = s->n_repertoires;

  /* Zobrist table for max(longest1, longest2) + 3 positions (overlap.cc:840) */
  const uint32_t A = (uint32_t)c->opt.alphabet_size;
  c->zpos = std::max(longest, longest_query) + EXTRA_POSITIONS;
  const uint32_t n_v = c->opt.ignore_genes ? 0 : c->opt.n_v_genes;
  const uint32_t n_j = c->opt.ignore_genes ? 0 : c->opt.n_j_genes;
  {
    std::vector<uint64_t> z((size_t)A * c->zpos + n_v + n_j);
    SplitMix64 rng(0x636f6d7061697272ull);   /* "compairr" */
    for (auto &x : z)
      x = rng.next();
    rc = dev_upload(c, c->zob, z.data(), z.size());
    if (rc)
      return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));   /* host vectors go away */
  }

  /* table: smallest power of two with fill <= 70 % (hash_init, hashtable.cc:31-54);
     Bloom: one byte per slot (bloom_init(tablesize), overlap.cc:863) */
  c->slots = 1;
  while (FILL_PERCENT * c->slots < 100 * s->n)
    c->slots <<= 1;
  uint64_t bloom_bytes = std::max<uint64_t>(c->slots, 8);
  /* results do not depend on the table size (every hit is verified), only the
     length of the probe chains does: HBM is plentiful, round trips are not */
  c->slots = std::max<uint64_t>(c->slots << c->table_log2_delta, 4);
  /* Kernel variant: the row filter (2) for amino acids -- one word read answers
     the 19 substitutions of a position; nucleotides at d = 1 (3 per position, and L + 1
     entries per sequence to pay for them) keep the per-variant filter (1). */
  /* Nucleotides at d = 2 (round 4): pair rows probed by whole workgroups per tile
     (kernels_pairs2.h) -- 1541 word reads per 45-nucleotide query where the per-variant
     filter is probed 8910 times.  Needs the query's residues packed into RESPACK_MAX positions. */
  const bool d2p_possible = A == 4 && c->opt.differences == 2 && !c->opt.indels &&
                            std::max(longest, longest_query) <= RESPACK_MAX;
  /* d = 0 (round 5, late): no filter at all -- one test per query, and the filter's word would cost the
     memory line the table's slot costs: the un-sliced kernel (0) looks the query up where its bucket lies
     (kernels.h probe_kernel, D == 0), lane = query, no positives buffer, no second kernel. */
  int64_t variant = c->variant >= 0 ? c->variant
                    : c->opt.differences == 0 ? 0
                    : (A == 20 && c->opt.differences >= 1) || (d2p_possible && c->d2_pairs != 0) ? 2 : 1;
  c->d2pairs = variant == 2 && d2p_possible && c->d2_pairs != 0;
  /* The staged layouts keep a slice, the Zobrist tables and the wave queues in
     LDS; with very long sequences (Zobrist tables of more than ~100 KiB) that
     no longer fits and the un-sliced filter is probed where it lies (variant 0). */
  c->sliced = variant >= 1;
  c->rows = variant == 2;
  int64_t swl = c->slice_words_log2;
  if (swl < 0)
    swl = SLICE_WORDS_LOG2;
  /* variant 2: the largest slice in 32-byte words -- 40 KiB by default (a ring of
     two slices + tables + the queues of 16 waves in one workgroup per CU), a power
     of two on request */
  uint64_t row_max_words = c->slice_words_log2 < 0
      ? MAX_ROW_SLICE_WORDS : std::min<uint64_t>(1ull << c->slice_words_log2, MAX_ROW_SLICE_WORDS);
  if (c->sliced) {
    const size_t zrow = c->rows ? 2 * (size_t)A : (size_t)(zrow_stride((int)A) + zdelta_entries((int)A));
    /* everything but the slice(s), with the fewest waves a workgroup may have */
    /* (variant 2 keeps no copy of the matrix in LDS) */
    const size_t fixed = zrow * c->zpos * sizeof(uint64_t) +
                         4 * sizeof(WaveQueue) + (c->rows ? 0 : 2048 * sizeof(unsigned long long)) +
                         MAX_CLASS_RES * A * sizeof(uint32_t) + HEAVY_WORDS * sizeof(uint32_t) + 16 +
                         64 * sizeof(TileRef) + (c->rows ? RING * (sizeof(RingSlot) + 64 * sizeof(TileRef)) : 0);
    size_t need_d2 = 0;
    if (c->rows && c->d2pairs) {
      /* kernels_pairs2.h: two slice buffers beside its tables, a matrix copy and 16 wave queues */
      const size_t fixed2 = (16 * (size_t)c->zpos + 20 * ((size_t)c->zpos + 1) / 2) * sizeof(uint64_t) +
                            2048 * sizeof(unsigned long long) + 16 * sizeof(WaveQueue) +
                            MAX_CLASS_RES * A * sizeof(uint32_t) + 2 * (64 + 64 * sizeof(TileRef)) + 1024;
      const size_t room = fixed2 < 160 * 1024 ? 160 * 1024 - fixed2 : 0;
      const uint64_t nbuf = c->d2_buffers == 1 ? 1 : 2;
      c->geom.nbuf = (uint32_t)nbuf;
      uint64_t w = room / (nbuf * ROW_WORD_BYTES);
      if (w >= 64)
        w -= w % 32;                                     /* whole KiB: LDS-DMA pieces */
      if (w < 1) {
        c->d2pairs = false;                              /* (Zobrist tables too large: single rows, or variant 1) */
      } else {
        row_max_words = c->slice_words_log2 < 0 ? w : std::min<uint64_t>(1ull << c->slice_words_log2, w);
        need_d2 = fixed2 + nbuf * (size_t)row_max_words * ROW_WORD_BYTES;
      }
    }
    if (c->rows && c->slice_words_log2 < 0 && !c->d2pairs) {
      /* the default slice leaves room for the queues of 16 waves; long sequences: a
         smaller slice next to the bigger Zobrist table */
      const size_t fixed16 = fixed + 12 * sizeof(WaveQueue);
      const size_t room = fixed16 < 160 * 1024 ? 160 * 1024 - fixed16 : 0;
      row_max_words = std::min<uint64_t>(row_max_words, room / (RING * ROW_WORD_BYTES));
      row_max_words -= row_max_words % 32;               /* whole KiB: LDS-DMA pieces */
    }
    const size_t need = c->rows && c->d2pairs ? need_d2
                        : fixed + (c->rows ? RING * (size_t)row_max_words * ROW_WORD_BYTES
                                           : ((size_t)8 << swl));
    if (need > 160 * 1024 || (c->rows && row_max_words < 1)) {
      c->sliced = false;
      c->rows = false;
    }
  }
  if (!c->rows)
    c->d2pairs = false;
  /* row filter: L + 1 entries per sequence, and with -i its L + 1 gap entries (kernels_rows.h) */
  /* (what the filter is SIZED for, also with pair rows, which enter fewer -- entries_filed:
     their eight-bit tests then see next to no false positive, and a slice holds more sequences) */
  const uint64_t entries = (residues2 + s->n) * (c->rows && c->opt.indels ? 2 : 1);
  if (c->rows) {
    /* 2 bytes of filter per entry (16 entries per 32-byte word: every dword of a
       word then has ~40 % of its bits set and a test of eight of them passes by
       chance ~6e-4 of the time -- the optimum of a Bloom filter at 16 bits per
       entry), x 2^delta */
    bloom_bytes = std::max<uint64_t>(entries * (uint64_t)c->row_filter_x16 / 16, ROW_WORD_BYTES);
    if (c->d2pairs)      /* sized for the entries filed: one per pair of positions and one per sequence */
      bloom_bytes = std::max<uint64_t>(((residues2 + 1) / 2 + 2 * s->n) * (uint64_t)c->row_filter_x16 / 16, ROW_WORD_BYTES);
    const int64_t delta = c->bloom_log2_delta == -100 ? 0 : c->bloom_log2_delta;
    if (delta > 0)
      bloom_bytes <<= delta;
    else if (delta < 0)
      bloom_bytes = std::max<uint64_t>(bloom_bytes >> (-delta), ROW_WORD_BYTES);
    /* S slices (a power of two: sibling slices are XORs of slice numbers) of
       rw_words <= row_max_words words each */
    uint64_t S = 1;
    while (S * row_max_words * ROW_WORD_BYTES * 17 / 16 < bloom_bytes)    /* (up to 6 % denser) */
      S <<= 1;
    uint64_t words = (bloom_bytes + S * ROW_WORD_BYTES - 1) / (S * ROW_WORD_BYTES);
    if (words >= 64)
      words = (words + 31) / 32 * 32;                  /* whole KiB: LDS-DMA pieces */
    words = std::max<uint64_t>(1, std::min<uint64_t>(words, row_max_words));
    /* d = 2 on single rows: the slices are as large as the LDS takes them, whatever the entry count asks for.
       S is a power of two, so a slice is between half full and full of what it may hold; the room left is
       free here -- a tile is ~2000 word reads per lane, the copy of its slice nothing beside them -- and a
       query's 38 000 variant tests meet a filter up to twice as sparse: 24.2M sequences against themselves,
       370 -> 640 words per slice, positives 1.09 x 10^9 -> 2 x 10^8 for 1.55 x 10^8 pairs, step 130 -> 95 ms
       (round 5).  At d = 1 the copies are what the kernel waits for: the slices stay as small as they can be. */
    if (c->opt.differences == 2 && !c->d2pairs && c->slice_words_log2 < 0 && c->fill_slices != 0 && S > 1)
      words = row_max_words;
    if (S > (1ull << 31))
      return fail(c, CMPR_EUNSUPPORTED, "row filter with more than 2^31 slices");
    c->geom.rw_words = (uint32_t)words;
    c->geom.words_log2 = 0;
    c->bloom_words = S * words * (ROW_WORD_BYTES / 8);       /* 8-byte units; + the class parts, below */
    c->geom.smask = (uint32_t)(S - 1);
    /* class parts (layout.h row_slice): each holds one entry per split sequence,
       the main part L + 1 - K per sequence: S n / entries slices, a power of two */
    /* (with -i a class part holds two entries per split sequence: blank row and gap row) */
    /* (kernels_pairs2.h: a third of a query's reads go to the class parts -- they are sized for
       their one entry per sequence at the filter's own density, 2 bytes per entry, not for the
       single rows' count: 0.26 % of false positives per test became 2e-5, round 4) */
    const uint64_t per_part = c->d2pairs ? bloom_bytes / 2 : entries;
    uint64_t Sc = 1;
    while (Sc < S && Sc * per_part < S * std::max<uint64_t>(s->n, 1) * (c->opt.indels ? 2 : 1))
      Sc <<= 1;
    c->geom.cmask = (uint32_t)(Sc - 1);
  } else {
    /* The LDS-staged layout pays nothing for a sparser filter (a slice is 32 KiB
       whatever the total), so it takes 4 bytes per table slot: with the 2^20
       pattern space that leaves almost only true positives for the table walk. */
    const int64_t delta = c->bloom_log2_delta == -100 ? (c->sliced ? 2 : 0)
                                                      : c->bloom_log2_delta;
    if (delta > 0)
      bloom_bytes <<= delta;
    else if (delta < 0)
      bloom_bytes = std::max<uint64_t>(bloom_bytes >> (-delta), 8);
    if (bloom_bytes > (1ull << 32))
      return fail(c, CMPR_EUNSUPPORTED, "Bloom filter larger than 4 GiB");
    c->bloom_words = bloom_bytes / 8;
    c->geom.rw_words = 0;
  }


  /* ---- variants 1, 2: cut the filter into class-keyed slices (layout.h) ---- */
  if (c->sliced) {
    SliceGeom &g = c->geom;
    if (!c->rows) {
      uint32_t wl = 0;
      while ((1ull << (wl + 1)) <= c->bloom_words && wl + 1 <= (uint32_t)swl)
        wl++;
      g.words_log2 = wl;
      g.smask = (uint32_t)(c->bloom_words >> wl) - 1;
    }
    g.ncl = c->zpos + 1;
    g.off_cv = g.ncl;
    g.off_cj = g.off_cv + n_v;
    g.off_cr = g.off_cj + n_j;
    g.off_hv = g.off_cr + MAX_CLASS_RES * A;
    c->ctab.assign((size_t)g.off_hv + HEAVY_WORDS, 0);
    SplitMix64 crng(0x736c69636573ull);     /* "slices" */
    for (size_t i = 0; i < g.off_hv; i++)
      c->ctab[i] = (uint32_t)(crng.next() >> 32);
    /* With -i the class key carries no length term: an insertion / deletion variant
       then stays in the class of its query (up to the class residues it shifts), so
       the indel rows are answered from the query's own slice and its tiles may mix
       lengths. */
    if (c->opt.indels)
      for (uint32_t L = 0; L < g.ncl; L++)
        c->ctab[L] = 0;
    /* the tables go to the device now (heavy bitmap still empty): the kernels below
       read them there */
    if ((rc = dev_upload(c, c->d_ctab, c->ctab.data(), c->ctab.size()))) return rc;
    g.ctab = c->d_ctab.p;
    const bool genes = !c->opt.ignore_genes;
    const uint64_t S = (uint64_t)g.smask + 1;
    const double slice_bits = (double)(64ull << g.words_log2);
    /* at least 12 filter bits per key in the fullest slice: fill <= 0.28 with 4
       bits per key, false-positive rate <= 6e-3 there and far less elsewhere.
       Row filter: sequences per slice at 24 entries per word (1.5 x the average). */
    const uint64_t entries_filed = !pair_rows(c) ? entries
                                   : (residues2 + 1) / 2 + s->n + (c->opt.indels ? residues2 + s->n : 0);
    const double slice_cap = c->rows
        ? (double)g.rw_words * 24.0 / std::max(1.0, (double)entries_filed / (double)std::max<uint64_t>(s->n, 1))
        : slice_bits / 12.0;
    g.k = 0;
    /* Class positions c0 .. c0+K-1.  They must exist in almost every sequence
       (<= 5th-percentile length of set 2) and be informative (a conserved
       position splits nothing); the closer to the start, the more insertion /
       deletion variants keep their class residues in place.  So: the first
       window of max_class_res positions whose residue entropy in set 2 is at
       least 70 % of the maximum. */
    g.c0 = 0;
    if (s->n > 0) {
      const uint32_t nh = longest + 2;
      Tmp<uint32_t> d_hist;
      if ((rc = dev_alloc(c, d_hist.b, nh))) return rc;
      HIP_TRY(c, hipMemsetAsync(d_hist.b.p, 0, nh * sizeof(uint32_t), c->stream));
      hipLaunchKernelGGL(length_hist_kernel, dim3(std::min<uint32_t>(1024u, blocks_for(s->n))), dim3(256),
                         nh <= 8192 ? nh * sizeof(uint32_t) : 0, c->stream, c->off2.p, s->n, nh,
                         d_hist.b.p);
      HIP_TRY(c, hipGetLastError());
      std::vector<uint32_t> hist(nh);
      HIP_TRY(c, hipMemcpyAsync(hist.data(), d_hist.b.p, nh * sizeof(uint32_t), hipMemcpyDeviceToHost,
                                c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      uint64_t acc = 0;
      uint32_t l5 = longest;
      for (uint32_t L = 0; L <= longest; L++) {
        acc += hist[L];
        if (acc * 20 >= s->n) {
          l5 = L;
          break;
        }
      }
      const uint32_t mcr = max_class_res(A);
      if (l5 > mcr) {
        const uint32_t npos = l5;
        const uint64_t stride = std::max<uint64_t>(1, s->n / 200000);   /* a sample is enough */
        Tmp<uint32_t> d_cnt;
        if ((rc = dev_alloc(c, d_cnt.b, (size_t)npos * A))) return rc;
        HIP_TRY(c, hipMemsetAsync(d_cnt.b.p, 0, (size_t)npos * A * sizeof(uint32_t), c->stream));
        hipLaunchKernelGGL(residue_sample_kernel,
                           dim3(std::min<uint32_t>(64u, blocks_for((s->n + stride - 1) / stride))), dim3(256),
                           (size_t)npos * A <= 8192 ? (size_t)npos * A * sizeof(uint32_t) : 0, c->stream,
                           c->res2.p, c->off2.p, s->n, stride, npos, A, d_cnt.b.p);
        HIP_TRY(c, hipGetLastError());
        std::vector<uint32_t> cnt((size_t)npos * A);
        HIP_TRY(c, hipMemcpyAsync(cnt.data(), d_cnt.b.p, cnt.size() * sizeof(uint32_t),
                                  hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        std::vector<double> ent(npos, 0.0);
        for (uint32_t p = 0; p < npos; p++) {
          double tot = 0;
          for (uint32_t r = 0; r < A; r++)
            tot += (double)cnt[(size_t)p * A + r];
          for (uint32_t r = 0; r < A && tot > 0; r++) {
            const double q = (double)cnt[(size_t)p * A + r] / tot;
            if (q > 0)
              ent[p] -= q * std::log2(q);
          }
        }
        const double need = 0.7 * std::log2((double)A);
        uint32_t best = (l5 - mcr) / 2;
        /* (pair rows: an even anchor first -- two class positions then share one pair, one item
           per heavy query instead of two) */
        const uint32_t step = pair_rows(c) ? 2u : 1u;
        bool found = false;
        for (uint32_t first = 0; first < step && !found; first++)
          for (uint32_t c0 = first; c0 + mcr <= npos; c0 += step) {
            bool ok = true;
            for (uint32_t i = 0; i < mcr; i++)
              ok = ok && ent[c0 + i] >= need;
            if (ok) {
              best = c0;
              found = true;
              break;
            }
          }
        if (!found && pair_rows(c))
          best &= ~1u;
        g.c0 = best;
      }
    }
    if (c->class_anchor >= 0)
      g.c0 = (uint32_t)c->class_anchor;
    if (S > 1 && s->n > 0) {
      /* population of every (length, V, J) class bucket */
      const size_t nb = (size_t)1 << HEAVY_BUCKETS_LOG2;
      Tmp<uint32_t> d_bucket, d_base, d_pop;
      if ((rc = dev_alloc(c, d_bucket.b, nb))) return rc;
      if ((rc = dev_alloc(c, d_base.b, (size_t)s->n))) return rc;
      HIP_TRY(c, hipMemsetAsync(d_bucket.b.p, 0, nb * sizeof(uint32_t), c->stream));
      hipLaunchKernelGGL(class_base_kernel, dim3(blocks_for(s->n)), dim3(256), 0, c->stream, c->off2.p,
                         c->v2.p, c->j2.p, s->n, g, genes ? 1u : 0u, d_base.b.p, d_bucket.b.p);
      HIP_TRY(c, hipGetLastError());
      std::vector<uint32_t> bucket(nb);
      HIP_TRY(c, hipMemcpyAsync(bucket.data(), d_bucket.b.p, nb * sizeof(uint32_t),
                                hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      /* heavy = would take more than half of a slice's budget on its own */
      /* (row filter: an eighth -- its 8-bit tests are sensitive to an overfull slice,
         and a split class costs its queries one cheap class row per class residue) */
      const double thr = c->heavy_threshold >= 0 ? (double)c->heavy_threshold
                                                 : slice_cap / (c->rows ? 8 : 2);
      bool any_heavy = false;
      for (uint32_t b = 0; b < bucket.size(); b++)
        if ((double)bucket[b] > thr) {
          c->ctab[g.off_hv + (b >> 5)] |= 1u << (b & 31);
          any_heavy = true;
        }
      HIP_TRY(c, hipMemcpyAsync(c->d_ctab.p + g.off_hv, c->ctab.data() + g.off_hv,
                                HEAVY_WORDS * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
      /* amino acids: a fourth class residue only where a kernel unrolls for it and gains by it
         (layout.h kernel_class_res: the wide forms of probe_rows_kernel, d = 1.  The single rows of
         d = 2 pay for every class position with an item per query and a row of double substitutions:
         24.2M sequences against themselves, 351 ms with four where three take 233) */
      const uint32_t kmax = (c->rows && c->opt.differences == 1) ? max_class_res(A) : kernel_class_res(A, false);
      if (c->class_residues >= 0) {
        g.k = std::min<uint32_t>((uint32_t)c->class_residues, kmax);
      } else if (any_heavy) {
        /* K = fewest class residues that bring the fullest slice under the cap;
           every one costs the heavy queries one more row elsewhere */
        if ((rc = dev_alloc(c, d_pop.b, (size_t)S))) return rc;
        std::vector<uint32_t> pop((size_t)S);
        double best_max = -1;
        uint32_t best_k = 1;
        for (uint32_t k = 1; k <= kmax; k++) {
          HIP_TRY(c, hipMemsetAsync(d_pop.b.p, 0, (size_t)S * sizeof(uint32_t), c->stream));
          hipLaunchKernelGGL(slice_population_kernel, dim3(blocks_for(s->n)), dim3(256), 0, c->stream,
                             c->res2.p, c->off2.p, d_base.b.p, s->n, g, A, k, d_pop.b.p);
          HIP_TRY(c, hipGetLastError());
          HIP_TRY(c, hipMemcpyAsync(pop.data(), d_pop.b.p, (size_t)S * sizeof(uint32_t),
                                    hipMemcpyDeviceToHost, c->stream));
          HIP_TRY(c, hipStreamSynchronize(c->stream));
          const double mx = *std::max_element(pop.begin(), pop.end());
          if (best_max < 0 || mx < best_max) {
            best_max = mx;
            best_k = k;
          }
          if (mx <= slice_cap)
            break;
          /* (the wide kernels cost a little everywhere and 12 bytes of scratch with -i: the last
             narrow K stands unless it leaves the fullest slice more than a quarter over its budget --
             uniform 10M x 10M with -i: 694 sequences for 627, stays at three; the cdr3 law: 1 775, and
             3 559 for 376 on 24.2M sequences: four, where the positives fall from 3.1 x 10^8 to 7.9 x 10^7) */
          if (k == kernel_class_res(A, false) && k < kmax && mx <= 1.25 * slice_cap)
            break;
        }
        g.k = best_k;
      }
      HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
  }
  c->geom.page_tab = nullptr;
  c->geom.nsl = 0;
  c->geom.pad_pages = 0;
  c->page_tab.release();
  c->page_slices = 0;
  if (c->rows) {        /* main part + one class part per class residue */
    const uint64_t nsl = (uint64_t)c->geom.smask + 1 + (uint64_t)c->geom.k * (c->geom.cmask + 1);
    c->geom.nsl = (uint32_t)nsl;
    /* ---- pages (layout.h SliceGeom): the slices that would hold more entries than their words are good
            for -- the class residues split a big class as evenly as its residues are distributed, and on
            skewed data that is not evenly: 24.2M sequences, d = 1 -i, the fullest slice at 1.7 x its budget
            with FOUR class residues and 45 % of the step's positives false (round 4) -- get 2^e pages.
            One counting pass of the kernel that files the entries (same code, BuildParams::count). ---- */
    if (c->opt.differences == 1 && !c->d2pairs && c->slice_pages != 0 && s->n > 0 && nsl < (1ull << 27)) {
      Tmp<uint32_t> d_cnt;
      if ((rc = dev_alloc(c, d_cnt.b, (size_t)nsl))) return rc;
      HIP_TRY(c, hipMemsetAsync(d_cnt.b.p, 0, (size_t)nsl * sizeof(uint32_t), c->stream));
      BuildParams Bc{};
      Bc.zob = c->zob.p;
      Bc.A = A;
      Bc.zpos = c->zpos;
      Bc.n_v = n_v;
      Bc.use_genes = c->opt.ignore_genes ? 0u : 1u;
      Bc.res = c->res2.p;
      Bc.off = c->off2.p;
      Bc.v = c->v2.p;
      Bc.j = c->j2.p;
      Bc.n = s->n;
      Bc.sliced = 1u;
      Bc.indels = c->opt.indels ? 1u : 0u;
      Bc.pairs = 1u;
      Bc.geom = c->geom;
      Bc.count = d_cnt.b.p;
      hipLaunchKernelGGL(build_rows_kernel, dim3((uint32_t)((s->n + BLOCK_THREADS - 1) / BLOCK_THREADS)),
                         dim3(BLOCK_THREADS), 0, c->stream, Bc);
      HIP_TRY(c, hipGetLastError());
      std::vector<uint32_t> cnt((size_t)nsl);
      HIP_TRY(c, hipMemcpyAsync(cnt.data(), d_cnt.b.p, (size_t)nsl * sizeof(uint32_t), hipMemcpyDeviceToHost,
                                c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      /* a slice's budget: 20 entries per word (the filter is sized for 16 single-row entries per word on
         average; pair rows file fewer).  The class residues are chosen so that the fullest slice stays under
         24 per word, and on skewed data a hump of slices ends up just under that: at 20 .. 24 entries per
         word an eight-bit test passes by chance 2-6 times per thousand, and those few slices made 40 % of
         the positives of 10M x 10M on the cdr3 law with -i (13.2M -> 7.8M with this budget; round 5) */
      const uint64_t budget = c->page_budget > 0 ? (uint64_t)c->page_budget : (uint64_t)c->geom.rw_words * 20;
      const uint32_t emax = (uint32_t)std::min<int64_t>(c->slice_pages < 0 ? PAGE_E_MAX : c->slice_pages, PAGE_E_MAX);
      std::vector<uint32_t> tab((size_t)nsl);
      uint64_t ovf = 0, paged = 0;
      uint32_t fullest = 0;
      for (uint64_t k = 0; k < nsl; k++) {
        uint32_t e = 0;
        while (e < emax && (uint64_t)cnt[k] > (budget << e))
          e++;
        tab[k] = e | ((uint32_t)ovf << 4);
        ovf += (1u << e) - 1u;
        paged += e ? 1 : 0;
        fullest = std::max(fullest, cnt[k]);
      }
      if (getenv("COMPAIRR_HIP_DEBUG"))
        fprintf(stderr, "compairr_hip: pages: %llu of %llu slices over their budget of %llu entries (fullest %u), "
                        "%llu overflow pages\n", (unsigned long long)paged, (unsigned long long)nsl,
                (unsigned long long)budget, fullest, (unsigned long long)ovf);
      if (ovf > 0 && ovf < (1ull << 27)) {
        if ((rc = dev_upload(c, c->page_tab, tab.data(), tab.size()))) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->stream));         /* (the host vector goes away) */
        c->geom.page_tab = c->page_tab.p;
        c->page_slices = ovf;
      }
    }
    c->bloom_words = (nsl + c->page_slices) * c->geom.rw_words * (ROW_WORD_BYTES / 8);
  }
  if (c->slots > (1ull << 30))
    return fail(c, CMPR_EUNSUPPORTED, "reference set too large for a 32-bit record table");
  /* d = 0 on the un-sliced kernel looks every query up where its bucket lies (kernels.h probe_kernel, D == 0):
     no filter is read, none is built (it was sized, cleared and filled for nothing) */
  const bool no_filter = !c->sliced && c->opt.differences == 0;
  if (no_filter)
    c->bloom_words = 1;
  if ((rc = dev_alloc(c, c->bloom, (size_t)c->bloom_words))) return rc;
  /* inverted polarity (bloompat.cc:54-57) for variants 0, 1; the row filter sets bits */
  HIP_TRY(c, hipMemsetAsync(c->bloom.p, c->rows ? 0 : 0xff, c->bloom_words * sizeof(uint64_t), c->stream));

  BuildParams B{};
  B.zob = c->zob.p;
  B.A = A;
  B.zpos = c->zpos;
  B.n_v = n_v;
  B.use_genes = c->opt.ignore_genes ? 0u : 1u;
  B.res = c->res2.p;
  B.off = c->off2.p;
  B.v = c->v2.p;
  B.j = c->j2.p;
  B.n = s->n;
  Tmp<uint32_t> tags, buckets, more;
  {
    /* The record table (see ref_keys_kernel): bucket and tag of every sequence, a stable radix sort of
       the sequence numbers by bucket, the prefix maximum that places the records, the table itself. */
    Tmp<uint32_t> seq, bkt_sorted, perm, last;
    Tmp<int32_t> lead, lead_max;
    Tmp<char> tmp;
    uint64_t nslots = c->slots + 1;             /* (a lookup may start at any bucket; one empty slot behind the last) */
    if ((rc = dev_alloc(c, c->voff2, std::max<size_t>((size_t)s->n, 1)))) return rc;
    if (s->n) {
      const size_t n = (size_t)s->n;
      if ((rc = dev_alloc(c, buckets.b, n))) return rc;
      if ((rc = dev_alloc(c, seq.b, n))) return rc;
      if ((rc = dev_alloc(c, tags.b, n))) return rc;
      if ((rc = dev_alloc(c, more.b, n))) return rc;
      if ((rc = dev_alloc(c, bkt_sorted.b, n))) return rc;
      if ((rc = dev_alloc(c, perm.b, n))) return rc;
      if ((rc = dev_alloc(c, last.b, 1))) return rc;
      hipLaunchKernelGGL(ref_keys_kernel, dim3(blocks_for(s->n)), dim3(256), 0, c->stream, B,
                         (uint32_t)(c->slots - 1), buckets.b.p, tags.b.p, seq.b.p);
      HIP_TRY(c, hipGetLastError());
      int bits = 0;
      while ((1ull << bits) < c->slots)
        bits++;
      size_t tb = 0;
      (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, buckets.b.p, bkt_sorted.b.p, seq.b.p, perm.b.p, (int)s->n,
                                               0, std::max(bits, 1), c->stream);
      if ((rc = dev_alloc(c, tmp.b, tb))) return rc;
      HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(tmp.b.p, tb, buckets.b.p, bkt_sorted.b.p, seq.b.p, perm.b.p,
                                                    (int)s->n, 0, std::max(bits, 1), c->stream));
      if ((rc = dev_alloc(c, lead.b, n))) return rc;
      if ((rc = dev_alloc(c, lead_max.b, n))) return rc;
      hipLaunchKernelGGL(bucket_lead_kernel, dim3(blocks_for(s->n)), dim3(256), 0, c->stream, bkt_sorted.b.p,
                         s->n, lead.b.p);
      HIP_TRY(c, hipGetLastError());
      size_t tb1 = 0;
      (void)hipcub::DeviceScan::InclusiveScan(nullptr, tb1, lead.b.p, lead_max.b.p, MaxI32(), (int)s->n, c->stream);
      if (tb1 > tb) {
        if ((rc = dev_alloc(c, tmp.b, tb1))) return rc;
        tb = tb1;
      }
      HIP_TRY(c, hipcub::DeviceScan::InclusiveScan(tmp.b.p, tb1, lead.b.p, lead_max.b.p, MaxI32(), (int)s->n,
                                                   c->stream));
      hipLaunchKernelGGL(record_slots_kernel, dim3(blocks_for(s->n)), dim3(256), 0, c->stream, bkt_sorted.b.p,
                         lead_max.b.p, perm.b.p, s->n, c->voff2.p, more.b.p, last.b.p);
      HIP_TRY(c, hipGetLastError());
      uint32_t last_slot = 0;
      HIP_TRY(c, hipMemcpyAsync(&last_slot, last.b.p, sizeof last_slot, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      nslots = std::max<uint64_t>(nslots, (uint64_t)last_slot + 2);
    }
    if (nslots >> 32)
      return fail(c, CMPR_EUNSUPPORTED, "reference set too large for 32-bit record positions");
    if ((rc = dev_alloc(c, c->rec2, (size_t)nslots * sizeof(RefRec)))) return rc;
    if ((rc = dev_alloc(c, c->bmap2, (size_t)(c->slots / 32 + 1)))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->bmap2.p, 0, (size_t)(c->slots / 32 + 1) * sizeof(uint32_t), c->stream));
    /* (all ones: RefRec::idx of an empty slot) */
    HIP_TRY(c, hipMemsetAsync(c->rec2.p, 0xff, (size_t)nslots * sizeof(RefRec), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));     /* (the temporaries go away) */
  }
  if (s->n) {
    B.bloom = c->rows ? nullptr : c->bloom.p;
    B.bloom_byte_mask = (uint32_t)((c->bloom_words - 1) << 3);
    B.sliced = c->sliced ? 1u : 0u;
    B.indels = c->rows && c->opt.indels ? 1u : 0u;
    B.pairs = pair_rows(c) ? 1u : 0u;
    B.geom = c->geom;
    const uint32_t grid = (uint32_t)((s->n + BLOCK_THREADS - 1) / BLOCK_THREADS);
    if (c->rows) {
      B.bloom = c->bloom.p;
      hipLaunchKernelGGL(build_rows_kernel, dim3(grid), dim3(BLOCK_THREADS), 0, c->stream, B);
    } else if (!no_filter) {
      /* (variants 0, 1: the per-variant filter; no table) */
      hipLaunchKernelGGL(build_index_kernel, dim3(grid), dim3(BLOCK_THREADS), 0, c->stream, B);
    }
    HIP_TRY(c, hipGetLastError());
  }
  if (s->n) {
    PackParams K{};
    K.voff = c->voff2.p;
    K.res = c->res2.p;
    K.off = c->off2.p;
    K.cnt = c->opt.ignore_counts ? nullptr : c->cnt2.p;
    K.v = c->opt.ignore_genes ? nullptr : c->v2.p;
    K.j = c->opt.ignore_genes ? nullptr : c->j2.p;
    K.rep = c->rep2.p;
    K.n = s->n;
    K.tag = tags.b.p;
    K.home = buckets.b.p;
    K.more = more.b.p;
    K.bmap = c->bmap2.p;
    K.packed = A == 4 ? 1u : 0u;
    K.out = c->rec2.p;
    const uint32_t grid = (uint32_t)((s->n + BLOCK_THREADS - 1) / BLOCK_THREADS);
    hipLaunchKernelGGL(pack_records_kernel, dim3(grid), dim3(BLOCK_THREADS), 0, c->stream, K);
    HIP_TRY(c, hipGetLastError());
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->have_ref = true;
  return CMPR_OK;
}

void cmpr_touch_index_kernels()
{
  hipFuncAttributes fa;
  (void)hipFuncGetAttributes(&fa, (const void *)length_hist_kernel);
}
