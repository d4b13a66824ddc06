/*
 * context.h -- private to libcompairr_hip.so: the context behind the C ABI and
 * the helpers its translation units share (compairr_hip.hip: the entry points;
 * query_layout.hip: the device-side layout of set 1; ref_index.hip: the index
 * of set 2).
 */
#ifndef COMPAIRR_AMD_CONTEXT_H
#define COMPAIRR_AMD_CONTEXT_H

#include "../../include/compairr_hip.h"
#include "kernels.h"

#include "select.h"

#include <functional>
#include <map>
#include <set>
#include <string>
#include <vector>

/* deterministic table contents; the result does not depend on them
   (check_variant makes matches hash-independent, variants.cc:166-240) */
struct SplitMix64 {
  uint64_t s;
  explicit SplitMix64(uint64_t seed) : s(seed) {}
  uint64_t next()
  {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  }
};

template <typename T>
struct DevBuf {
  T     *p = nullptr;
  size_t n = 0;
  size_t cap = 0;          /* elements allocated (>= n; dev_reserve keeps an allocation that is large enough) */
  void release()
  {
    if (p)
      (void)hipFree(p);
    p = nullptr;
    n = 0;
    cap = 0;
  }
};

/* device memory for the temporaries of a call, kept from call to call and grown when a
   call needs more (cmpr_set_queries: no hipMalloc / hipFree in the steady state) */
struct DevArena {
  char  *base = nullptr;
  size_t cap = 0;
  void release()
  {
    if (base)
      (void)hipFree(base);
    base = nullptr;
    cap = 0;
  }
};

using cmpr::Chunk;
using cmpr::PosEntry;
using cmpr::SliceGeom;
using cmpr::Slot;
using cmpr::TileDesc;

/* Everything about a step that does not change from launch to launch (kernels, grid,
   LDS, kernel arguments): worked out once after cmpr_set_queries, not per call. */
struct StepPlan {
  bool     valid = false;
  bool     will_launch = false, reduce_writes = false, deferred = false, redo_kind = false;
  size_t   cells = 0, ctr_blk = 0;
  cmpr::ProbeParams P{};            /* the per-launch fields are filled in at enqueue */
  cmpr::ProbeFn fn = nullptr, fn2 = nullptr, rfn = nullptr;
  uint32_t grid = 0, nw = 0, rgrid = 0;
  size_t   lds = 0, rlds = 0;
};

/* where cmpr_layout_queries takes the query set from */
struct LayoutSource {
  enum Kind { HOST, DEVICE, RECORDS } kind = HOST;
  const cmpr_set_view *view = nullptr;   /* HOST: the caller's arrays; DEVICE: device pointers on the context's device */
  /* RECORDS (cmpr_set_queries_routed): route_record_bytes() each, in device memory */
  const void   *d_records = nullptr;
  uint64_t      nrec = 0;
  uint32_t      n_rep = 0;
  uint64_t      n_total = 0;             /* sequences of the WHOLE query set (-x: the matrix has that many rows) */
  const double *totals = nullptr;        /* count totals per repertoire of the whole set, or NULL */
  /* cmpr_route_queries: key the share, count the records per destination, keep it for cmpr_route_pack */
  bool          route = false;
  uint64_t      first_index = 0;
  /* what the caller has to queue on the context's stream behind the layout (its per-launch scratch): run just
     before the layout's last wait, so that the call waits once at its end, not twice */
  std::function<int()> finish;
};

/* the keyed share cmpr_route_queries leaves in arena A for cmpr_route_pack */
struct RouteState {
  bool     valid = false;
  uint64_t n = 0, first_index = 0, total_records = 0;
  uint32_t n_dest = 0, record_bytes = 0;
  unsigned long long counts[64] = {};
  std::vector<double> rep_totals;
  const uint8_t  *res = nullptr;
  const uint64_t *off = nullptr, *cnt = nullptr;
  const uint32_t *v = nullptr, *j = nullptr, *rep = nullptr;
  const uint32_t *mask_lo = nullptr, *mask_hi = nullptr;
  unsigned long long *dest = nullptr;    /* [64] counts | [64] fill cursors */
};

struct cmpr_context {
  cmpr_options opt{};
  int          device = 0;
  int          cus = 256;
  hipStream_t  stream = nullptr;
  hipEvent_t   ev_stop = nullptr;
  /* kernel-time events of the last TIME_RING calls (cmpr_get_kernel_times), so
     that a caller can time many launches without synchronising after each */
  static const uint32_t TIME_RING = 64;
  hipEvent_t   ring_k0[TIME_RING] = {}, ring_km[TIME_RING] = {}, ring_k1[TIME_RING] = {};
  hipEvent_t   ring_start[TIME_RING] = {};   /* start event of the call in this slot: its own k0, or the
                                                previous call's k1 when the two ran back to back */
  hipEvent_t   ring_mid[TIME_RING] = {};     /* end of its probe kernel: its km */
  uint64_t     calls = 0;            /* overlap launches so far */
  hipEvent_t   ev_k0 = nullptr, ev_km = nullptr, ev_k1 = nullptr;   /* = the ring entry of the last call */
  bool         events_valid = false;
  bool         stop_is_k1 = false;   /* the last call ended with the kernels (ev_k1), no copy behind them */
  std::string  err;

  /* tunables */
  int64_t blocks_per_cu = 8;
  int64_t variant = -1;           /* 0: one global Bloom; 1: LDS-staged slices; 2: LDS-staged
                                     row filter (kernels_rows.h); -1: by alphabet */
  int64_t bloom_log2_delta = -100; /* -100: default (0 for variant 0, +2 sliced) */
  int64_t row_filter_x16 = 32;    /* variant 2: filter bytes per entry, in sixteenths (32 = 2 bytes: 16 bits per
                                     entry, eight of them set) */
  int64_t slice_pages = -1;       /* variant 2, d = 1: overfull slices get up to 2^this pages; -1 = up to 8, 0 = none */
  int64_t page_budget = 0;        /* entries a slice may hold before it gets pages; 0 = 24 per word (tests: small) */
  int64_t bucket_bitmap = -1;     /* resolve_kernel asks the bucket bitmap before it reads a slot of the record table:
                                     -1 = where most positives are false (d = 2), 0 never, 1 always */
  int64_t direct_slices_log2 = -1; /* d = 0 on variant 0: pseudo-slices of the query layout (-1: by the number of queries) */
  int64_t fill_slices = 1;        /* variant 2 at d = 2 (single rows): the slices take all the words their LDS buffer holds */
  int64_t class_residues = -1;    /* -1: choose from the data                  */
  int64_t class_anchor = -1;      /* c0, -1: from set 2's lengths              */
  int64_t heavy_threshold = -1;   /* class population above which it is split;
                                     -1: from the slice size, 0: every class   */
  int64_t chunk_tiles = 0;        /* 0: 8 x waves_per_block                    */
  int64_t waves_per_block = 8;
  bool    sub2_active = false;    /* ... and the query layout in effect lists them */
  int64_t sub2_items = -1;        /* variant 1, nt, d = 2: class-changing double substitutions as items
                                     (-1 = default = on; 0: probed where the filter lies) */
  int64_t work_shard_index = 0;   /* this context works on every work_shard_count-th work   */
  int64_t work_shard_count = 1;   /* item (chunk / small tile / tile) of the step, from here */
  int64_t small_slice_tiles = 0;  /* slices with <= this many tiles are not staged (wave phase): never pays since round 2 */
  int64_t class_rows_unstaged = 0; /* variant 2: class-row tiles read the filter where it lies */
  int64_t host_threads = 0;       /* threads of the host-side layout passes; set in cmpr_create */
  int64_t table_log2_delta = 1;   /* directory buckets = 2^delta x the 70 % rule of hashtable.cc:24 */
  int64_t deferred_resolve = 1;   /* Bloom positives walked by a second kernel   */
  int64_t chunk_deal = 1;         /* variant 2: 1 = beyond a workgroup's first four, chunks are handed out by a counter
                                     in list order (heaviest first); 0 = all of them dealt statically */
  int64_t pos_capacity = 0;       /* entries of the positives buffer; 0 = auto   */
  int64_t pos_grow = -1;          /* the buffer grows to what a launch showed: -1 = when its size was automatic,
                                     1 = also from a given pos_capacity (tests), 0 = never */
  int64_t resolve_blocks_per_cu = 5; /* what resolve_kernel's registers allow      */
  int64_t pos_segments = 64;      /* independently claimed parts of that buffer  */
  bool    waves_per_block_forced = false;
  int64_t debug = 0;              /* ablation switches (layout.h DBG_*), -DCMPR_ABLATION builds only */
  int64_t slice_words_log2 = -1;  /* -1: 12 (8-byte words, variant 1) / 11 (16-byte words, variant 2) */
  uint32_t chunk_cap = 0;         /* tiles per chunk in effect since cmpr_set_queries */

  /* sliced Bloom layout (variant 1) */
  bool                  sliced = false;   /* variant 1 or 2 */
  bool                  rows = false;     /* variant 2 */
  uint32_t              npasses = 1;      /* variant 2: 1 + class-row passes */
  SliceGeom             geom{};
  std::vector<uint32_t> ctab;     /* host copy of the class tables             */
  DevBuf<uint32_t>      d_ctab;
  DevBuf<uint32_t>      page_tab;         /* variant 2, d = 1: pages of the overfull slices (layout.h SliceGeom) */
  uint64_t              page_slices = 0;  /* overflow pages behind the regular slices */
  DevBuf<Chunk>         chunks;
  DevBuf<cmpr::TileRef> tile_refs;    /* what the chunks list */
  DevBuf<uint32_t>      small_tiles;
  uint32_t              nsmall = 0;
  uint32_t              nchunks = 0;

  /* Zobrist + patterns */
  uint32_t          zpos = 0;
  DevBuf<uint64_t>  zob;

  /* set 2 + index */
  bool              have_ref = false;
  uint64_t          n2 = 0;
  uint32_t          R2 = 0, longest2 = 0;
  DevBuf<uint8_t>   res2;
  DevBuf<uint64_t>  off2, cnt2, bloom;
  DevBuf<uint32_t>  v2, j2, rep2;
  DevBuf<uint32_t>  bmap2;          /* one bit per bucket of the record table: it holds a record */
  DevBuf<unsigned char> rec2;      /* the record table (layout.h RefRec; ref_index.hip) */
  DevBuf<uint32_t>  voff2;          /* slot of sequence i in it */
  uint64_t          slots = 0, bloom_words = 0;     /* slots: buckets of the record table (a power of two) */

  /* set 1 tiles */
  bool              have_q = false;
  bool              routed = false;  /* the resident queries are this context's share (cmpr_set_queries_routed) */
  RouteState        route;
  uint64_t          n1 = 0;
  uint32_t          R1 = 0, ntiles = 0;
  DevBuf<TileDesc>  tiles;
  uint32_t          nmain_tiles = 0;     /* tiles of pass 0: slot = tile * 64 + lane */
  DevBuf<uint32_t>  qres, qv, qj;    /* qv, qj: variants 0 and 1 (their tiles read them) */
  DevBuf<uint64_t>  qgh;            /* per slot: V key ^ J key (variants 0, 1: one load, not
                                       two dependent ones) / the query's Zobrist hash
                                       (variant 2; db_hash, db.cc:903-916) */
  DevBuf<uint64_t>  qhins, qhdel;   /* variant 2 with -i: the two shifted hashes
                                       (zobrist.cc:90-104, 122-136) */
  DevBuf<uint16_t>  qlen;
  DevBuf<uint32_t>  qck;
  DevBuf<cmpr::QueryRec> qrec;     /* per slot: what verification reads, 64 bytes */
  /* variant 2, class rows: per item the row's blanked hash, the query's slot in
     pass 0 (~0: padding) and its residue at the class position | position << 8 */
  DevBuf<cmpr::ItemRec> items;
  DevBuf<uint32_t>  slice_items;     /* sub2 items: per slice {first item, blocks} (layout.h CHUNK_WITH_ITEMS) */
  DevBuf<cmpr::ResPack> cpk;             /* sub2 items: the query's residues, 2 bits each */
  DevBuf<cmpr::ResPack> qpk;             /* nucleotides, d = 2 on pair rows: per slot, the query's residues */
  bool              d2pairs = false;    /* ... that kernel is in use (kernels_pairs2.h; decided with the index) */
  int64_t           d2_pairs = -1;      /* tunable: -1 auto, 0 off, 1 on */
  int64_t           d2_buffers = 1;     /* tunable: slice buffers of that kernel (layout.h SliceGeom::nbuf; cfg5: 51 ms with one, 60 with two) */
  uint64_t          algorithmic_bytes = 0;
  double            max_cell_bound = 0;   /* max_i total1[i] * max_j total2[j] */
  std::vector<double> tot1, tot2;

  /* per-launch scratch */
  DevBuf<unsigned long long> matrix;
  DevBuf<double>             matrix_f64;
  DevBuf<PosEntry>           pos_buf;      /* deferred resolve: queued Bloom positives */
  DevBuf<unsigned long long> pos_ctr;      /* per segment: [0] claimed, [1] ~first claim that did not fit */
  unsigned long long        *ctr_cur = nullptr, *ctr_other = nullptr;   /* this launch's block / the next one's */
  bool                       ctr_clean = false;        /* ctr_other and `part` are all zero */
  /* The redo launch of variant 2 (enqueue_overlap) is dropped once a finished launch on
     these sets has shown that the fullest segment of the positives buffer stays clear
     of its capacity by more than launches can differ (which wave ends up with which
     tile only moves the number of part-filled 64-entry blocks): */
  unsigned long long        *d_usage = nullptr;        /* [0] fullest segment of a launch (reduce kernel);
                                                          [1] sticky: a launch WITHOUT redo pass overflowed */
  unsigned long long        *h_usage = nullptr;        /* pinned copy of both */
  hipEvent_t                 ev_usage = nullptr;
  bool                       usage_pending = false, never_overflows = false;
  /* the margin argument holds for one static deal of the chunks: the grid and the waves
     per workgroup the usage was measured with / the shortcut was established for */
  uint32_t                   usage_grid = 0, usage_nw = 0, safe_grid = 0, safe_nw = 0;
  bool                       last_without_redo = false;   /* the last launch relied on never_overflows */
  bool                       force_no_redo = false;       /* test only (tunable "assume_never_overflows") */
  /* a cmpr_overlap_matrix_device launch without redo pass that cmpr_get_stats has not
     looked at yet / one such was found overflowed by a synchronous call's own check */
  bool                       async_unchecked = false, async_overflowed = false;
  hipStream_t                last_stream = nullptr;       /* stream of the last enqueue (launches of one */
  bool                       have_last_stream = false;    /* context are ordered one after the other)   */
  DevArena                   arena_a, arena_b;            /* temporaries of cmpr_set_queries */
  void                      *stage_host = nullptr;        /* pinned staging of the narrowed upload */
  size_t                     stage_host_bytes = 0;
  int64_t                    narrow_upload = -1;          /* tunable: -1 auto, 0 off, 1 on */
  hipStream_t                copy_stream = nullptr;       /* uploads of cmpr_set_queries */
  static const uint32_t      NCOPY_EV = 4;
  hipEvent_t                 ev_copy[NCOPY_EV] = {};
  /* of the last cmpr_set_queries: host time inside the copies, from the last copy to the
     end (the device work the upload did not hide), and in all */
  double                     layout_upload_ms = 0, layout_tail_ms = 0, layout_total_ms = 0;
  /* tunables of the query layout (query_layout.hip): item counters per workgroup in LDS, hashes worked out
     from the records by fill_tiles_kernel (both 1 = default; 0 = round 5's form, kept for A/B and the
     parity suite), and HIP events around its big kernels (tunable "layout_timing", default 0) */
  int64_t                    item_wg = 1, layout_recompute = 1, layout_timing = 0;
  int64_t                    layout_zob_lds = 1;          /* keys_kernel keeps the Zobrist keys in LDS when they fit */
  int64_t                    record_tiles = 1;            /* tunable: 1 = where the layout allows (layout.h rec_tiles), the hash in
                                                             the record where the sequences leave room; 2 = never the hash (rows) */
  bool                       rec_tiles = false;           /* the resident layout has no per-slot arrays: the probe
                                                             kernel reads the queries' records (layout.h ProbeParams) */
  bool                       rec_hash = false;            /* ... and the query's hash lies in the record (layout.h rec_tiles == 2) */
  /* what the runtime said about a kernel with a given dynamic LDS size (asked once: a plan is made per query
     set): workgroups per CU, and whether its LDS limit has been raised */
  std::map<std::pair<const void *, size_t>, int> occupancy_seen;
  std::set<std::pair<const void *, size_t>>      lds_raised;
  void                      *h_sizes = nullptr;           /* pinned: the layout's SizesBlock arrives here */
  size_t                     h_sizes_bytes = 0;
  hipEvent_t                 ev_layout[6] = {};
  uint32_t                   layout_marks = 0;
  float                      layout_kernel_ms[5] = {};    /* keys | sizes, slices | scatter | tiles | chunk order */
  StepPlan                   plan;
  unsigned long long        *d_stats = nullptr;        /* inside pos_ctr's allocation */
  uint32_t                  *d_tile_counter = nullptr; /* likewise */
  unsigned long long        *d_overflow = nullptr, *d_stats2 = nullptr;   /* redo pass (kernels_rows.h) */
  uint32_t                  *d_tile_counter2 = nullptr;
  unsigned long long        *d_deal = nullptr;         /* chunk counters of probe_rows_kernel (DEAL_WORDS) */
  DevBuf<unsigned long long> part;           /* NPART x part_stride partial results */
  uint32_t                   part_stride = 0;
  const void                *attr_fn2 = nullptr;
  size_t                     attr_lds2 = 0;
  uint64_t                   pos_cap = 0;
  uint32_t                   launches = 0;
  const void                *attr_fn = nullptr;   /* kernel whose LDS limit is raised */
  size_t                     attr_lds = 0;
  /* pairs mode, set only while cmpr_overlap_pairs runs */
  uint32_t           *pair_q = nullptr, *pair_h = nullptr;
  unsigned long long *pair_count = nullptr;
  uint64_t            pair_cap = 0;
};


/* What cmpr_warm_up_sized() reserved before any context existed (ABI v5): the pinned staging buffer of the
   narrowed upload and one block of device memory per device, taken over by the first context that needs them
   (query_layout.hip), released by the first cmpr_destroy() if nobody did.  Process-wide, under a mutex. */
struct WarmReservation {
  void  *host = nullptr;
  size_t host_bytes = 0;
  char  *dev = nullptr;
  size_t dev_bytes = 0;
  int    device = -1;
};
/* take what fits (nullptr: nothing reserved, or too small) */
void *cmpr_take_reserved_host(size_t need, size_t *got_bytes);
char *cmpr_take_reserved_device(int device, size_t need, size_t *got_bytes);
void cmpr_release_reservations();

/* message of a failed cmpr_create() (no context yet), per thread */
std::string &cmpr_create_error();

inline int fail(cmpr_context *c, int code, const std::string &msg)
{
  if (c)
    c->err = msg;
  else
    cmpr_create_error() = msg;
  return code;
}

#define HIP_TRY(c, call)                                                        \
  do {                                                                          \
    hipError_t e_ = (call);                                                     \
    if (e_ != hipSuccess)                                                       \
      return fail((c), e_ == hipErrorOutOfMemory ? CMPR_ENOMEM : CMPR_EDEVICE,  \
                  std::string(#call) + ": " + hipGetErrorString(e_));           \
  } while (0)

template <typename T>
int dev_alloc(cmpr_context *c, DevBuf<T> &b, size_t n)
{
  b.release();
  if (n == 0)
    n = 1;
  HIP_TRY(c, hipMalloc((void **)&b.p, n * sizeof(T)));
  b.n = n;
  b.cap = n;
  return CMPR_OK;
}

/* like dev_alloc, but an allocation that is large enough is kept (contents undefined) */
template <typename T>
int dev_reserve(cmpr_context *c, DevBuf<T> &b, size_t n)
{
  if (n == 0)
    n = 1;
  if (b.p && b.cap >= n) {
    b.n = n;
    return CMPR_OK;
  }
  b.release();
  const size_t want = n + n / 16;
  HIP_TRY(c, hipMalloc((void **)&b.p, want * sizeof(T)));
  b.n = n;
  b.cap = want;
  return CMPR_OK;
}

template <typename T>
int dev_upload(cmpr_context *c, DevBuf<T> &b, const T *src, size_t n)
{
  int rc = dev_alloc(c, b, n);
  if (rc)
    return rc;
  if (n)
    HIP_TRY(c, hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
  return CMPR_OK;
}

/* hipFuncSetAttribute(MaxDynamicSharedMemorySize) / hipOccupancyMaxActiveBlocksPerMultiprocessor, remembered */
inline int raise_lds_limit(cmpr_context *c, const void *fn, size_t lds)
{
  if (lds <= 48 * 1024 || c->lds_raised.count({fn, lds}))
    return CMPR_OK;
  HIP_TRY(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  c->lds_raised.insert({fn, lds});
  return CMPR_OK;
}

inline int occupancy_of(cmpr_context *c, const void *fn, int threads, size_t lds)
{
  auto it = c->occupancy_seen.find({fn, lds});
  if (it != c->occupancy_seen.end())
    return it->second;
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, threads, lds) != hipSuccess)
    occ = 0;
  (void)hipGetLastError();
  c->occupancy_seen[{fn, lds}] = occ;
  return occ;
}

inline bool is_f64_score(const cmpr_options &o)
{
  return o.score == CMPR_SCORE_RATIO && !o.ignore_counts;
}

/* pointer / size sanity of a set view (no pass over the data); on_device: the arrays are
   device memory, nothing of them is read here */
int validate_view(const cmpr_options &o, const cmpr_set_view *s, std::string &why, bool on_device = false);

/* query_layout.hip: upload set 1, validate it and lay it out for the kernels,
   all on the device (cmpr_set_queries) */
int cmpr_layout_queries(cmpr_context *c, const LayoutSource &src);

/* query_layout.hip: the records of a keyed share into the caller's buffer (cmpr_route_pack) */
int cmpr_route_pack_impl(cmpr_context *c, void *d_send, uint64_t capacity_bytes);

/* query_layout.hip: the caller's arrays into the given device buffers, validated
   there (what a host pass over the set would check); longest sequence and count
   total per repertoire */
int cmpr_upload_and_validate(cmpr_context *c, const cmpr_set_view *s, DevBuf<uint8_t> &res,
                             DevBuf<uint64_t> &off, DevBuf<uint32_t> &v, DevBuf<uint32_t> &j,
                             DevBuf<uint32_t> &rep, DevBuf<uint64_t> &cnt, uint32_t &longest,
                             std::vector<double> &rep_total, bool on_device = false, uint64_t total_dev = 0);

/* ref_index.hip: cmpr_set_reference (on_device: the view holds device pointers) */
int cmpr_build_reference(cmpr_context *c, const cmpr_set_view *s, uint32_t longest_query, bool on_device);

/* variant 2, d = 1 (with or without -i): the filter holds pair rows (kernels_rows.h); so does
   that of nucleotides at d = 2 when kernels_pairs2.h probes it */
inline bool pair_rows(const cmpr_context *c)
{
  return c->rows && (c->opt.differences == 1 || c->d2pairs);
}

#endif
