/*
 * compairr_hip.hip -- host side of libcompairr_hip.so: the C ABI declared in
 * include/compairr_hip.h over the gfx950 kernels in kernels.h.
 *
 * No CPU fallback lives here: every entry point either runs on the HIP device
 * or fails with an error code.
 */
#include "context.h"
#include "kernels_sliced.h"
#include "kernels_rows.h"
#include "select.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

using namespace cmpr;

std::string &cmpr_create_error()
{
  static thread_local std::string e;
  return e;
}

/* u64s behind the segment counters of the positives buffer: statistics, cursors,
   overflow flag, and the statistics + cursors of the redo pass */
static constexpr size_t CTR_TAIL = 2 * (STAT_COUNT + 1) + 1;

int validate_view(const cmpr_options &o, const cmpr_set_view *s, std::string &why)
{
  if (!s) { why = "set view is NULL"; return CMPR_EINVAL; }
  if (s->n_repertoires == 0 && s->n > 0) { why = "n_repertoires is 0"; return CMPR_EINVAL; }
  if (s->n >= 0xffffffc0ull) { why = "more than 2^32-64 sequences in one set"; return CMPR_EUNSUPPORTED; }
  if (s->n == 0)
    return CMPR_OK;
  if (!s->offsets || !s->repertoire) { why = "offsets/repertoire is NULL"; return CMPR_EINVAL; }
  if (!s->residues && s->offsets[s->n] > 0) { why = "residues is NULL"; return CMPR_EINVAL; }
  if (!o.ignore_genes && (!s->v_gene || !s->j_gene)) { why = "v_gene/j_gene is NULL without ignore_genes"; return CMPR_EINVAL; }
  if (!o.ignore_counts && !s->count) { why = "count is NULL without ignore_counts"; return CMPR_EINVAL; }
  if (s->offsets[0] != 0) { why = "offsets[0] must be 0"; return CMPR_EINVAL; }
  return CMPR_OK;
}

namespace {

int validate_options(const cmpr_options *o, std::string &why)
{
  if (!o) { why = "options is NULL"; return CMPR_EINVAL; }
  if (o->alphabet_size != 20 && o->alphabet_size != 4) {
    why = "alphabet_size must be 20 or 4"; return CMPR_EINVAL;
  }
  if (o->differences < 0) {
    why = "Differences specified with -d or -differences cannot be negative.";
    return CMPR_EINVAL;
  }
  if (o->indels && o->differences != 1) {
    why = "Indels are only allowed when d=1"; return CMPR_EINVAL;
  }
  if (o->differences > 2) {
    why = "d > 2 (the reference's all-against-all path, overlap.cc:286-359) is "
          "not part of the GPU hot path";
    return CMPR_EUNSUPPORTED;
  }
  if (o->score < CMPR_SCORE_PRODUCT || o->score > CMPR_SCORE_JACCARD) {
    why = "unknown score"; return CMPR_EINVAL;
  }
  if (o->existence && (o->score == CMPR_SCORE_MH || o->score == CMPR_SCORE_JACCARD)) {
    why = "The Morisita-Horn / Jaccard index is only allowed when computing repertoire overlap";
    return CMPR_EINVAL;
  }
  if (o->differences > 0 &&
      (o->score == CMPR_SCORE_MH || o->score == CMPR_SCORE_JACCARD)) {
    why = "The Morisita-Horn / Jaccard index is not defined when d>0";
    return CMPR_EINVAL;
  }
  for (int k = 0; k < 6; k++)
    if (o->reserved[k]) { why = "reserved option fields must be zero"; return CMPR_EINVAL; }
  return CMPR_OK;
}


/* Runs fn(t, begin, end) on `threads` host threads over [0, n) cut into equal
   contiguous ranges (thread t gets range t): the per-sequence passes of the
   layout code are independent or become so with per-thread histograms. */
template <typename F>
void parallel_ranges(uint64_t n, unsigned threads, F fn)
{
  if (threads <= 1 || n < 65536) {
    fn(0u, (uint64_t)0, n);
    return;
  }
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < threads; t++)
    pool.emplace_back([=]() { fn(t, n * t / threads, n * (t + 1) / threads); });
  for (std::thread &th : pool)
    th.join();
}

/* residue codes, gene and repertoire numbers in range; lengths >= 1 */
int scan_view(const cmpr_options &o, const cmpr_set_view *s, uint32_t &longest,
              std::vector<double> &rep_total, std::string &why, unsigned threads)
{
  longest = 0;
  rep_total.assign(s->n_repertoires, 0.0);
  if (s->n < 65536)
    threads = 1;
  threads = std::max(1u, threads);
  std::vector<int> rcs(threads, CMPR_OK);
  std::vector<const char *> whys(threads, "");
  std::vector<uint32_t> longs(threads, 0);
  std::vector<std::vector<double> > tots(threads, std::vector<double>(s->n_repertoires, 0.0));
  const uint8_t A = (uint8_t)o.alphabet_size;
  parallel_ranges(s->n, threads, [&](unsigned t, uint64_t b, uint64_t e) {
    auto bad = [&](int rc, const char *w) { rcs[t] = rc; whys[t] = w; };
    for (uint64_t i = b; i < e; i++) {
      if (s->offsets[i + 1] < s->offsets[i]) return bad(CMPR_EINVAL, "offsets not monotone");
      const uint64_t L = s->offsets[i + 1] - s->offsets[i];
      if (L > 0xffffu) return bad(CMPR_EUNSUPPORTED, "sequence longer than 65535 residues");
      longs[t] = std::max<uint32_t>(longs[t], (uint32_t)L);
      if (s->repertoire[i] >= s->n_repertoires) return bad(CMPR_EINVAL, "repertoire number out of range");
      if (!o.ignore_genes && (s->v_gene[i] >= o.n_v_genes || s->j_gene[i] >= o.n_j_genes))
        return bad(CMPR_EINVAL, "gene number out of range");
      if (!o.ignore_counts && s->count[i] < 1) return bad(CMPR_EINVAL, "duplicate_count must be >= 1");
      tots[t][s->repertoire[i]] += o.ignore_counts ? 1.0 : (double)s->count[i];
    }
    /* the residues of this range of sequences (offsets are monotone here) */
    for (uint64_t k = s->offsets[b]; k < s->offsets[e]; k++)
      if (s->residues[k] >= A) return bad(CMPR_EINVAL, "residue code out of range");
  });
  for (unsigned t = 0; t < threads; t++) {
    if (rcs[t] != CMPR_OK) { why = whys[t]; return rcs[t]; }
    longest = std::max(longest, longs[t]);
    for (uint32_t r = 0; r < s->n_repertoires; r++)
      rep_total[r] += tots[t][r];
  }
  return CMPR_OK;
}

/* number of variants the reference enumerates for one query
   (generate_variants, variants.cc:260-428) */
uint64_t variants_of(const cmpr_options &o, const uint8_t *s, uint32_t L)
{
  const uint64_t A = (uint64_t)o.alphabet_size;
  uint64_t n = 1;
  if (o.differences >= 1) {
    n += (A - 1) * L;
    if (o.indels) {
      if (L > 1) {
        uint64_t runs = 1;
        for (uint32_t p = 1; p < L; p++)
          runs += s[p] != s[p - 1];
        n += runs;
      }
      n += A + (A - 1) * (uint64_t)L;
    }
  }
  if (o.differences >= 2)
    n += (A - 1) * (A - 1) * (uint64_t)L * (L ? L - 1 : 0) / 2;
  return n;
}

ProbeFn select_sliced_kernel(const cmpr_options &o, int nw)
{
  const int A = o.alphabet_size, D = o.differences;
  const bool i = o.indels != 0, g = !o.ignore_genes;
  switch (nw) {
  case 4:  return select_probe_v1_nw4(A, D, i, g);
  case 16: return select_probe_v1_nw16(A, D, i, g);
  default: return select_probe_v1_nw8(A, D, i, g);
  }
}

ProbeFn select_rows_kernel(const cmpr_options &o, int nw, bool inline_resolve)
{
  const int A = o.alphabet_size, D = o.differences;
  const bool i = o.indels != 0, g = !o.ignore_genes;
  if (inline_resolve)
    switch (nw) {
    case 4:  return select_probe_v2_inline_nw4(A, D, i, g);
    case 16: return select_probe_v2_inline_nw16(A, D, i, g);
    default: return select_probe_v2_inline_nw8(A, D, i, g);
    }
  switch (nw) {
  case 4:  return select_probe_v2_nw4(A, D, i, g);
  case 16: return select_probe_v2_nw16(A, D, i, g);
  default: return select_probe_v2_nw8(A, D, i, g);
  }
}

ProbeFn select_kernel(const cmpr_options &o)
{
  return select_probe_v0(o.alphabet_size, o.differences, o.indels != 0, !o.ignore_genes);
}


}  // namespace

/* ------------------------------------------------------------------ */

extern "C" int cmpr_abi_version(void)
{
  return CMPR_ABI_VERSION;
}

extern "C" int cmpr_create(const cmpr_options *options, cmpr_context **out)
{
  if (!out)
    return fail(nullptr, CMPR_EINVAL, "out is NULL");
  *out = nullptr;
  std::string why;
  int rc = validate_options(options, why);
  if (rc)
    return fail(nullptr, rc, why);

  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, CMPR_EDEVICE,
                std::string("no HIP device available: ") + hipGetErrorString(e));

  cmpr_context *c = new (std::nothrow) cmpr_context();
  if (!c)
    return fail(nullptr, CMPR_ENOMEM, "out of host memory");
  c->opt = *options;
  if (options->device >= 0) {
    if (options->device >= ndev) {
      delete c;
      return fail(nullptr, CMPR_EINVAL, "device ordinal out of range");
    }
    c->device = options->device;
  } else {
    (void)hipGetDevice(&c->device);
  }
#define CREATE_TRY(call)                                                        \
  do {                                                                          \
    hipError_t e_ = (call);                                                     \
    if (e_ != hipSuccess) {                                                     \
      std::string m = std::string(#call) + ": " + hipGetErrorString(e_);        \
      cmpr_destroy(c);                                                          \
      return fail(nullptr, CMPR_EDEVICE, m);                                    \
    }                                                                           \
  } while (0)
  CREATE_TRY(hipSetDevice(c->device));
  hipDeviceProp_t prop;
  CREATE_TRY(hipGetDeviceProperties(&prop, c->device));
  c->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  CREATE_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  c->host_threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  CREATE_TRY(hipEventCreate(&c->ev_start));
  for (uint32_t i = 0; i < cmpr_context::TIME_RING; i++) {
    CREATE_TRY(hipEventCreate(&c->ring_k0[i]));
    CREATE_TRY(hipEventCreate(&c->ring_km[i]));
    CREATE_TRY(hipEventCreate(&c->ring_k1[i]));
  }
  c->ev_k0 = c->ring_k0[0];
  c->ev_km = c->ring_km[0];
  c->ev_k1 = c->ring_k1[0];
  CREATE_TRY(hipEventCreate(&c->ev_stop));
#undef CREATE_TRY
  /* environment overrides of the tunables (for the CLI, which has no flag) */
  if (const char *e = getenv("COMPAIRR_HIP_VARIANT")) {
    int v = atoi(e);
    if (v >= 0 && v <= 2)
      c->variant = v;
  }
  if (const char *e = getenv("COMPAIRR_HIP_SLICE_WORDS_LOG2")) {
    int v = atoi(e);
    if (v >= 1 && v <= 13)
      c->slice_words_log2 = v;
  }
  if (const char *e = getenv("COMPAIRR_HIP_CLASS_RESIDUES")) {
    int v = atoi(e);
    if (v >= -1 && v <= (int)max_class_res((uint32_t)c->opt.alphabet_size))
      c->class_residues = v;
  }
  *out = c;
  return CMPR_OK;
}

extern "C" void cmpr_destroy(cmpr_context *c)
{
  if (!c)
    return;
  (void)hipSetDevice(c->device);
  if (c->stream)
    (void)hipStreamSynchronize(c->stream);
  c->zob.release();
  c->res2.release(); c->off2.release(); c->cnt2.release(); c->table.release();
  c->bloom.release(); c->v2.release(); c->j2.release(); c->rep2.release();
  c->rec2.release(); c->voff2.release();
  c->tiles.release(); c->qres.release(); c->qv.release(); c->qj.release(); c->qgh.release();
  c->qrep.release(); c->qcnt.release(); c->qlen.release(); c->qorig.release(); c->qck.release();
  c->qhins.release(); c->qhdel.release(); c->cw.release(); c->cmain.release(); c->crp.release(); c->qrec.release();
  c->matrix.release(); c->matrix_f64.release();
  c->pos_buf.release(); c->pos_ctr.release(); c->d_ctab.release(); c->chunks.release(); c->tile_refs.release(); c->small_tiles.release();
  if (c->ev_start) (void)hipEventDestroy(c->ev_start);
  for (uint32_t i = 0; i < cmpr_context::TIME_RING; i++) {
    if (c->ring_k0[i]) (void)hipEventDestroy(c->ring_k0[i]);
    if (c->ring_km[i]) (void)hipEventDestroy(c->ring_km[i]);
    if (c->ring_k1[i]) (void)hipEventDestroy(c->ring_k1[i]);
  }
  if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" const char *cmpr_last_error(const cmpr_context *c)
{
  return c ? c->err.c_str() : cmpr_create_error().c_str();
}

extern "C" uint32_t cmpr_rows(const cmpr_context *c) { return c ? c->R1 : 0; }
extern "C" uint32_t cmpr_cols(const cmpr_context *c) { return c ? c->R2 : 0; }

extern "C" int cmpr_set_tunable(cmpr_context *c, const char *name, int64_t value)
{
  if (!c || !name)
    return CMPR_EINVAL;
  std::string n(name);
  if (n == "blocks_per_cu") {
    if (value < 1 || value > 16)
      return fail(c, CMPR_EINVAL, "blocks_per_cu must be 1..16");
    c->blocks_per_cu = value;
  } else if (n == "variant") {
    if (value < -1 || value > 2)
      return fail(c, CMPR_EINVAL, "variant must be -1 (default), 0, 1 or 2");
    if (c->have_ref)
      return fail(c, CMPR_ESTATE, "set variant before cmpr_set_reference");
    c->variant = value;
  } else if (n == "class_residues") {
    if (value < -1 || value > (int64_t)max_class_res((uint32_t)c->opt.alphabet_size))
      return fail(c, CMPR_EINVAL, "class_residues must be -1..3 (amino acids) / -1..8 (nucleotides)");
    if (c->have_ref)
      return fail(c, CMPR_ESTATE, "set class_residues before cmpr_set_reference");
    c->class_residues = value;
  } else if (n == "class_anchor") {
    if (value < -1 || value > 65535)
      return fail(c, CMPR_EINVAL, "class_anchor must be -1..65535");
    if (c->have_ref)
      return fail(c, CMPR_ESTATE, "set class_anchor before cmpr_set_reference");
    c->class_anchor = value;
  } else if (n == "heavy_threshold") {
    if (value < -1)
      return fail(c, CMPR_EINVAL, "heavy_threshold must be >= -1");
    if (c->have_ref)
      return fail(c, CMPR_ESTATE, "set heavy_threshold before cmpr_set_reference");
    c->heavy_threshold = value;
  } else if (n == "slice_words_log2") {
    if (value < -1 || value == 0 || value > 13)
      return fail(c, CMPR_EINVAL, "slice_words_log2 must be -1 (default) or 1..13");
    if (c->have_ref)
      return fail(c, CMPR_ESTATE, "set slice_words_log2 before cmpr_set_reference");
    c->slice_words_log2 = value;
  } else if (n == "chunk_tiles") {
    if (value < 0 || value > 512)
      return fail(c, CMPR_EINVAL, "chunk_tiles must be 0..512");
    if (c->have_q)
      return fail(c, CMPR_ESTATE, "set chunk_tiles before cmpr_set_queries");
    c->chunk_tiles = value;
  } else if (n == "debug") {
#ifdef CMPR_ABLATION
    c->debug = value;
#else
    return fail(c, CMPR_EINVAL, "the debug switches exist only in a -DCMPR_ABLATION build");
#endif
  } else if (n == "host_threads") {
    if (value < 1 || value > 256)
      return fail(c, CMPR_EINVAL, "host_threads must be 1..256");
    c->host_threads = value;
  } else if (n == "table_log2_delta") {
    if (value < 0 || value > 3)
      return fail(c, CMPR_EINVAL, "table_log2_delta must be 0..3");
    if (c->have_ref)
      return fail(c, CMPR_ESTATE, "set table_log2_delta before cmpr_set_reference");
    c->table_log2_delta = value;
  } else if (n == "deferred_resolve") {
    if (value < 0 || value > 1)
      return fail(c, CMPR_EINVAL, "deferred_resolve must be 0 or 1");
    c->deferred_resolve = value;
  } else if (n == "resolve_blocks_per_cu") {
    if (value < 1 || value > 8)
      return fail(c, CMPR_EINVAL, "resolve_blocks_per_cu must be 1..8");
    c->resolve_blocks_per_cu = value;
  } else if (n == "pos_segments") {
    if (value < 1 || value > 256 || (value & (value - 1)))
      return fail(c, CMPR_EINVAL, "pos_segments must be a power of two, 1..256");
    if (c->have_q)
      return fail(c, CMPR_ESTATE, "set pos_segments before cmpr_set_queries");
    c->pos_segments = value;
  } else if (n == "pos_capacity") {
    if (value < 0)
      return fail(c, CMPR_EINVAL, "pos_capacity must be >= 0");
    if (c->have_q)
      return fail(c, CMPR_ESTATE, "set pos_capacity before cmpr_set_queries");
    c->pos_capacity = value;
  } else if (n == "small_slice_tiles") {
    if (value < 0 || value > 64)
      return fail(c, CMPR_EINVAL, "small_slice_tiles must be 0..64");
    if (c->have_q)
      return fail(c, CMPR_ESTATE, "set small_slice_tiles before cmpr_set_queries");
    c->small_slice_tiles = value;
  } else if (n == "class_rows_unstaged") {
    if (value < 0 || value > 1)
      return fail(c, CMPR_EINVAL, "class_rows_unstaged must be 0 or 1");
    if (c->have_q)
      return fail(c, CMPR_ESTATE, "set class_rows_unstaged before cmpr_set_queries");
    c->class_rows_unstaged = value;
  } else if (n == "waves_per_block") {
    if (value != 4 && value != 8 && value != 16)
      return fail(c, CMPR_EINVAL, "waves_per_block must be 4, 8 or 16");
    if (c->have_q)
      return fail(c, CMPR_ESTATE, "set waves_per_block before cmpr_set_queries");
    c->waves_per_block = value;
    c->waves_per_block_forced = true;
  } else if (n == "bloom_bits_log2_delta") {
    if (value < -4 || value > 4)
      return fail(c, CMPR_EINVAL, "bloom_bits_log2_delta must be -4..4");
    if (c->have_ref)
      return fail(c, CMPR_ESTATE, "set bloom_bits_log2_delta before cmpr_set_reference");
    c->bloom_log2_delta = value;
  } else {
    return fail(c, CMPR_EINVAL, "unknown tunable: " + n);
  }
  return CMPR_OK;
}

extern "C" int cmpr_get_tunable(cmpr_context *c, const char *name, int64_t *value)
{
  if (!c || !name || !value)
    return CMPR_EINVAL;
  std::string n(name);
  if (n == "variant") *value = c->have_ref ? (c->rows ? 2 : c->sliced ? 1 : 0) : c->variant;
  else if (n == "blocks_per_cu") *value = c->blocks_per_cu;
  else if (n == "bloom_bits_log2_delta") {
    *value = 0;
    for (uint64_t b = std::max<uint64_t>(c->slots, 8); b < c->bloom_words * 8; b <<= 1)
      (*value)++;
  }
  else if (n == "class_residues") *value = c->sliced && c->have_ref ? (int64_t)c->geom.k : c->class_residues;
  else if (n == "slice_words_log2") *value = c->sliced && c->have_ref ? (int64_t)c->geom.words_log2 : c->slice_words_log2;
  else if (n == "slice_bytes") *value = !c->sliced ? 0 : c->rows ? (int64_t)c->geom.rw_words * ROW_WORD_BYTES : (int64_t)8 << c->geom.words_log2;
  else if (n == "passes") *value = c->npasses;
  else if (n == "chunk_tiles") *value = c->chunk_tiles > 0 ? c->chunk_tiles : 8 * c->waves_per_block;
  else if (n == "waves_per_block") *value = c->waves_per_block;
  else if (n == "debug") *value = c->debug;
  else if (n == "heavy_threshold") *value = c->heavy_threshold;
  else if (n == "class_anchor") *value = c->sliced && c->have_ref ? (int64_t)c->geom.c0 : c->class_anchor;
  else if (n == "heavy_buckets") {
    *value = 0;
    if (c->sliced && c->have_ref)
      for (uint32_t w = 0; w < HEAVY_WORDS; w++)
        *value += __builtin_popcount(c->ctab[c->geom.off_hv + w]);
  }
  else if (n == "slices") *value = c->sliced ? (int64_t)c->geom.smask + 1 : 1;
  else if (n == "tiles") *value = c->ntiles;
  else if (n == "chunks") *value = c->nchunks;
  else if (n == "small_tiles") *value = c->nsmall;
  else if (n == "small_slice_tiles") *value = c->small_slice_tiles;
  else if (n == "class_rows_unstaged") *value = c->class_rows_unstaged;
  else if (n == "deferred_resolve") *value = c->deferred_resolve;
  else if (n == "table_log2_delta") *value = c->table_log2_delta;
  else if (n == "host_threads") *value = c->host_threads;
  else if (n == "pos_capacity") *value = (int64_t)(c->pos_cap * c->pos_segments);
  else if (n == "pos_segments") *value = c->pos_segments;
  else if (n == "resolve_blocks_per_cu") *value = c->resolve_blocks_per_cu;
  else if (n == "query_slots") *value = (int64_t)c->ntiles * WAVE;
#ifdef CMPR_PHASE_TIMING
  else if (n.size() == 3 && n[0] == 'p' && n[1] == 't' && n[2] >= '0' && n[2] <= '7') {
    /* diagnostic build: wave cycles of phase n of the last launch (kernels_rows.h PT_*) */
    unsigned long long x = 0;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(&x, c->d_stats + 8 + (n[2] - '0'), sizeof x, hipMemcpyDeviceToHost));
    *value = (int64_t)x;
  }
#endif
  else
    return fail(c, CMPR_EINVAL, "unknown tunable: " + n);
  return CMPR_OK;
}

/* ------------------------------------------------------------------ */
/* set 2: upload + index build on the device                            */
/* ------------------------------------------------------------------ */

static int cmpr_set_reference_impl(cmpr_context *c, const cmpr_set_view *s,
                                  uint32_t longest_query)
{
  if (!c)
    return CMPR_EINVAL;
  std::string why;
  int rc = validate_view(c->opt, s, why);
  if (rc)
    return fail(c, rc, why);
  HIP_TRY(c, hipSetDevice(c->device));
  c->have_ref = false;
  c->have_q = false;

  uint32_t longest = 0;
  rc = scan_view(c->opt, s, longest, c->tot2, why, (unsigned)c->host_threads);
  if (rc)
    return fail(c, rc, why);
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

  /* records */
  const uint64_t total = s->n ? s->offsets[s->n] : 0;
  static const uint64_t zero_off[1] = {0};
  if ((rc = dev_upload(c, c->res2, s->residues, (size_t)total))) return rc;
  if ((rc = dev_upload(c, c->off2, s->n ? s->offsets : zero_off, (size_t)s->n + 1))) return rc;
  if ((rc = dev_upload(c, c->rep2, s->repertoire, (size_t)s->n))) return rc;
  if (!c->opt.ignore_genes) {
    if ((rc = dev_upload(c, c->v2, s->v_gene, (size_t)s->n))) return rc;
    if ((rc = dev_upload(c, c->j2, s->j_gene, (size_t)s->n))) return rc;
  } else {
    c->v2.release();
    c->j2.release();
  }
  if (!c->opt.ignore_counts) {
    if ((rc = dev_upload(c, c->cnt2, s->count, (size_t)s->n))) return rc;
  } else {
    c->cnt2.release();
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
     the 19 substitutions of a position; nucleotides (3 per position, and L + 1
     entries per sequence to pay for them) keep the per-variant filter (1). */
  int64_t variant = c->variant >= 0 ? c->variant : (A == 20 ? 2 : 1);
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
    const size_t fixed = zrow * c->zpos * sizeof(uint64_t) +
                         4 * sizeof(WaveQueue) + 2048 * sizeof(unsigned long long) +
                         MAX_CLASS_RES * A * sizeof(uint32_t) + HEAVY_WORDS * sizeof(uint32_t) + 16 +
                         64 * sizeof(TileRef) + (c->rows ? RING * (sizeof(RingSlot) + 64 * sizeof(TileRef)) : 0);
    if (c->rows && c->slice_words_log2 < 0) {
      /* the default slice leaves room for the queues of 16 waves; long sequences: a
         smaller slice next to the bigger Zobrist table */
      const size_t fixed16 = fixed + 12 * sizeof(WaveQueue);
      const size_t room = fixed16 < 160 * 1024 ? 160 * 1024 - fixed16 : 0;
      row_max_words = std::min<uint64_t>(row_max_words, room / (RING * ROW_WORD_BYTES));
      row_max_words -= row_max_words % 32;               /* whole KiB: LDS-DMA pieces */
    }
    const size_t need = fixed + (c->rows ? RING * (size_t)row_max_words * ROW_WORD_BYTES
                                         : ((size_t)8 << swl));
    if (need > 160 * 1024 || (c->rows && row_max_words < 1)) {
      c->sliced = false;
      c->rows = false;
    }
  }
  const uint64_t entries = (s->n ? s->offsets[s->n] : 0) + s->n;   /* row filter: L + 1 per sequence */
  if (c->rows) {
    /* 2 bytes of filter per entry (16 entries per 32-byte word: every dword of a
       word then has ~40 % of its bits set and a test of eight of them passes by
       chance ~6e-4 of the time -- the optimum of a Bloom filter at 16 bits per
       entry), x 2^delta */
    bloom_bytes = std::max<uint64_t>(entries * 2, ROW_WORD_BYTES);
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
    if (S > (1ull << 31))
      return fail(c, CMPR_EUNSUPPORTED, "row filter with more than 2^31 slices");
    c->geom.rw_words = (uint32_t)words;
    c->geom.words_log2 = 0;
    c->bloom_words = S * words * (ROW_WORD_BYTES / 8);       /* 8-byte units; + the class parts, below */
    c->geom.smask = (uint32_t)(S - 1);
    /* class parts (layout.h row_slice): each holds one entry per split sequence,
       the main part L + 1 - K per sequence: S n / entries slices, a power of two */
    uint64_t Sc = 1;
    while (Sc < S && Sc * entries < S * std::max<uint64_t>(s->n, 1))
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
    const bool genes = !c->opt.ignore_genes;
    const uint64_t S = (uint64_t)g.smask + 1;
    const double slice_bits = (double)(64ull << g.words_log2);
    /* at least 12 filter bits per key in the fullest slice: fill <= 0.28 with 4
       bits per key, false-positive rate <= 6e-3 there and far less elsewhere.
       Row filter: sequences per slice at 24 entries per word (1.5 x the average). */
    const double slice_cap = c->rows
        ? (double)g.rw_words * 24.0 / std::max(1.0, (double)entries / (double)std::max<uint64_t>(s->n, 1))
        : slice_bits / 12.0;
    g.k = 0;
    /* Class positions c0 .. c0+K-1.  They must exist in almost every sequence
       (<= 5th-percentile length of set 2) and be informative (a conserved
       position splits nothing); the closer to the start, the more insertion /
       deletion variants keep their class residues in place.  So: the first
       window of max_class_res positions whose residue entropy in set 2 is at
       least 70 % of the maximum. */
    {
      std::vector<uint64_t> hist((size_t)longest + 2, 0);
      for (uint64_t i = 0; i < s->n; i++)
        hist[s->offsets[i + 1] - s->offsets[i]]++;
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
      g.c0 = 0;
      if (l5 > mcr && s->n > 0) {
        const uint32_t npos = l5;
        std::vector<uint64_t> cnt((size_t)npos * A, 0);
        const uint64_t stride = std::max<uint64_t>(1, s->n / 200000);   /* a sample is enough */
        for (uint64_t i = 0; i < s->n; i += stride) {
          const uint64_t b = s->offsets[i];
          const uint32_t L = (uint32_t)std::min<uint64_t>(s->offsets[i + 1] - b, npos);
          for (uint32_t p = 0; p < L; p++)
            cnt[(size_t)p * A + s->residues[b + p]]++;
        }
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
        for (uint32_t c0 = 0; c0 + mcr <= npos; c0++) {
          bool ok = true;
          for (uint32_t i = 0; i < mcr; i++)
            ok = ok && ent[c0 + i] >= need;
          if (ok) {
            best = c0;
            break;
          }
        }
        g.c0 = best;
      }
      if (c->class_anchor >= 0)
        g.c0 = (uint32_t)c->class_anchor;
    }
    if (S > 1 && s->n > 0) {
      /* population of every (length, V, J) class bucket */
      const unsigned T = s->n < 65536 ? 1u : (unsigned)std::max<int64_t>(1, c->host_threads);
      std::vector<uint32_t> bucket((size_t)1 << HEAVY_BUCKETS_LOG2, 0);
      std::vector<uint32_t> base_of((size_t)s->n);
      {
        std::vector<std::vector<uint32_t> > part(T, std::vector<uint32_t>(bucket.size(), 0));
        parallel_ranges(s->n, T, [&](unsigned t, uint64_t lo, uint64_t hi) {
          for (uint64_t i = lo; i < hi; i++) {
            const uint32_t L = (uint32_t)(s->offsets[i + 1] - s->offsets[i]);
            const uint32_t b = class_base(c->ctab.data(), g, genes, L, genes ? s->v_gene[i] : 0,
                                          genes ? s->j_gene[i] : 0);
            base_of[i] = b;
            part[t][b >> (32 - HEAVY_BUCKETS_LOG2)]++;
          }
        });
        for (unsigned t = 0; t < T; t++)
          for (size_t b = 0; b < bucket.size(); b++)
            bucket[b] += part[t][b];
      }
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
      if (c->class_residues >= 0) {
        g.k = (uint32_t)c->class_residues;
      } else if (any_heavy) {
        /* K = fewest class residues that bring the fullest slice under the cap;
           every one costs the heavy queries one HBM-probed row */
        double best_max = -1;
        uint32_t best_k = 1;
        std::vector<uint32_t> pop;
        for (uint32_t k = 1; k <= max_class_res(A); k++) {
          pop.assign((size_t)S, 0);
          {
            std::vector<std::vector<uint32_t> > part(T, std::vector<uint32_t>((size_t)S, 0));
            parallel_ranges(s->n, T, [&](unsigned t, uint64_t lo, uint64_t hi) {
              for (uint64_t i = lo; i < hi; i++) {
                const uint64_t b = s->offsets[i];
                const uint32_t L = (uint32_t)(s->offsets[i + 1] - b);
                uint32_t ck = base_of[i];
                if (L > 0 && class_is_heavy(c->ctab.data(), g, ck))
                  for (uint32_t r = 0; r < k; r++)
                    ck ^= c->ctab[g.off_cr + r * A + s->residues[b + class_pos(L, r, g.c0)]];
                part[t][ck & g.smask]++;
              }
            });
            for (unsigned t = 0; t < T; t++)
              for (size_t x = 0; x < (size_t)S; x++)
                pop[x] += part[t][x];
          }
          const double mx = *std::max_element(pop.begin(), pop.end());
          if (best_max < 0 || mx < best_max) {
            best_max = mx;
            best_k = k;
          }
          if (mx <= slice_cap)
            break;
        }
        g.k = best_k;
      }
    }
    if ((rc = dev_upload(c, c->d_ctab, c->ctab.data(), c->ctab.size()))) return rc;
    g.ctab = c->d_ctab.p;
  }
  if (c->rows)          /* main part + one class part per class residue */
    c->bloom_words = ((uint64_t)c->geom.smask + 1 + (uint64_t)c->geom.k * (c->geom.cmask + 1)) *
                     c->geom.rw_words * (ROW_WORD_BYTES / 8);
  if ((rc = dev_alloc(c, c->table, (size_t)c->slots))) return rc;
  if ((rc = dev_alloc(c, c->bloom, (size_t)c->bloom_words))) return rc;
  HIP_TRY(c, hipMemsetAsync(c->table.p, 0xff, c->slots * sizeof(Slot), c->stream));
  /* inverted polarity (bloompat.cc:54-57) for variants 0, 1; the row filter sets bits */
  HIP_TRY(c, hipMemsetAsync(c->bloom.p, c->rows ? 0 : 0xff, c->bloom_words * sizeof(uint64_t), c->stream));

  {
    /* positions in the verification stream (layout.h RefRec) */
    std::vector<uint32_t> voff((size_t)s->n + 1);
    uint64_t units = 0;
    for (uint64_t i = 0; i < s->n; i++) {
      /* a record of at most 64 bytes never straddles a 64-byte boundary, and no
         header does: one memory request fetches it */
      const uint64_t u = rec_units((uint32_t)(s->offsets[i + 1] - s->offsets[i]));
      const uint64_t room = 4 - (units & 3);
      if (std::min<uint64_t>(u, 4) > room)
        units += room;
      voff[i] = (uint32_t)units;
      units += u;
    }
    units += 8;                 /* verify_candidate reads 64 bytes whatever the length */
    if (units >> 32)
      return fail(c, CMPR_EUNSUPPORTED, "reference set too large for 32-bit record positions");
    if ((rc = dev_upload(c, c->voff2, voff.data(), std::max<size_t>((size_t)s->n, 1)))) return rc;
    if ((rc = dev_alloc(c, c->rec2, std::max<size_t>((size_t)units * REC_UNIT, REC_UNIT)))) return rc;
  }
  if (s->n) {
    BuildParams B{};
    B.zob = c->zob.p;
    B.A = A;
    B.zpos = c->zpos;
    B.n_v = n_v;
    B.use_genes = c->opt.ignore_genes ? 0u : 1u;
    B.voff = c->voff2.p;
    B.res = c->res2.p;
    B.off = c->off2.p;
    B.v = c->v2.p;
    B.j = c->j2.p;
    B.n = s->n;
    B.table = c->table.p;
    B.slot_mask = c->slots - 1;
    B.bloom = c->rows ? nullptr : c->bloom.p;
    B.bloom_byte_mask = (uint32_t)((c->bloom_words - 1) << 3);
    B.sliced = c->sliced ? 1u : 0u;
    B.geom = c->geom;
    const uint32_t grid = (uint32_t)((s->n + BLOCK_THREADS - 1) / BLOCK_THREADS);
    hipLaunchKernelGGL(build_index_kernel, dim3(grid), dim3(BLOCK_THREADS), 0,
                       c->stream, B);
    HIP_TRY(c, hipGetLastError());
    if (c->rows) {
      B.bloom = c->bloom.p;
      hipLaunchKernelGGL(build_rows_kernel, dim3(grid), dim3(BLOCK_THREADS), 0, c->stream, B);
      HIP_TRY(c, hipGetLastError());
    }
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
    K.out = c->rec2.p;
    const uint32_t grid = (uint32_t)((s->n + BLOCK_THREADS - 1) / BLOCK_THREADS);
    hipLaunchKernelGGL(pack_records_kernel, dim3(grid), dim3(BLOCK_THREADS), 0, c->stream, K);
    HIP_TRY(c, hipGetLastError());
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->have_ref = true;
  return CMPR_OK;
}

static int cmpr_count_duplicates_impl(cmpr_context *c, const cmpr_set_view *s, uint64_t *out)
{
  if (!c || !out)
    return CMPR_EINVAL;
  HIP_TRY(c, hipSetDevice(c->device));
  *out = 0;
  const uint32_t A = (uint32_t)c->opt.alphabet_size;
  const uint32_t n_v = c->opt.ignore_genes ? 0 : c->opt.n_v_genes;
  DevBuf<unsigned long long> d_count;
  DevBuf<uint8_t> res;
  DevBuf<uint64_t> off;
  DevBuf<uint32_t> v, j, rep;
  DevBuf<Slot> table;
  struct Cleanup {
    DevBuf<unsigned long long> &a; DevBuf<uint8_t> &b; DevBuf<uint64_t> &c1;
    DevBuf<uint32_t> &d1, &d2, &d3; DevBuf<Slot> &e;
    ~Cleanup() { a.release(); b.release(); c1.release();
                 d1.release(); d2.release(); d3.release(); e.release(); }
  } cleanup{d_count, res, off, v, j, rep, table};
  int rc;
  if ((rc = dev_alloc(c, d_count, 1))) return rc;
  HIP_TRY(c, hipMemsetAsync(d_count.p, 0, sizeof(unsigned long long), c->stream));

  DupParams D{};
  D.A = A;
  D.n_v = n_v;
  D.use_genes = c->opt.ignore_genes ? 0u : 1u;
  D.count = d_count.p;
  if (!s) {
    if (!c->have_ref)
      return fail(c, CMPR_ESTATE, "cmpr_set_reference must be called first");
    D.zob = c->zob.p;
    D.zpos = c->zpos;
    D.res = c->res2.p; D.off = c->off2.p; D.v = c->v2.p; D.j = c->j2.p; D.rep = c->rep2.p;
    D.n = c->n2;
    D.table = c->table.p; D.slot_mask = c->slots - 1;
    D.rec = c->rec2.p;
  } else {
    std::string why;
    if ((rc = validate_view(c->opt, s, why)))
      return fail(c, rc, why);
    uint32_t longest = 0;
    std::vector<double> tot;
    if ((rc = scan_view(c->opt, s, longest, tot, why, (unsigned)c->host_threads)))
      return fail(c, rc, why);
    /* own Zobrist keys when no reference set is resident or it is too short */
    DevBuf<uint64_t> zob_own;
    struct Z { DevBuf<uint64_t> &z; ~Z() { z.release(); } } zclean{zob_own};
    uint32_t zpos = c->zpos;
    const uint64_t *zob = c->zob.p;
    if (!c->have_ref || longest + EXTRA_POSITIONS > c->zpos) {
      zpos = longest + EXTRA_POSITIONS;
      const uint32_t n_j = c->opt.ignore_genes ? 0 : c->opt.n_j_genes;
      std::vector<uint64_t> z((size_t)A * zpos + n_v + n_j);
      SplitMix64 rng(0x6475706c69636174ull);
      for (auto &x : z)
        x = rng.next();
      if ((rc = dev_upload(c, zob_own, z.data(), z.size()))) return rc;
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      zob = zob_own.p;
    }
    const uint64_t total = s->n ? s->offsets[s->n] : 0;
    static const uint64_t zero_off[1] = {0};
    if ((rc = dev_upload(c, res, s->residues, (size_t)total))) return rc;
    if ((rc = dev_upload(c, off, s->n ? s->offsets : zero_off, (size_t)s->n + 1))) return rc;
    if ((rc = dev_upload(c, rep, s->repertoire, (size_t)s->n))) return rc;
    if (!c->opt.ignore_genes) {
      if ((rc = dev_upload(c, v, s->v_gene, (size_t)s->n))) return rc;
      if ((rc = dev_upload(c, j, s->j_gene, (size_t)s->n))) return rc;
    }
    uint64_t slots = 1;
    while (FILL_PERCENT * slots < 100 * s->n)
      slots <<= 1;
    slots = std::max<uint64_t>(slots, 4);         /* chains start on 4-slot boundaries */
    if ((rc = dev_alloc(c, table, (size_t)slots))) return rc;
    HIP_TRY(c, hipMemsetAsync(table.p, 0xff, slots * sizeof(Slot), c->stream));
    if (s->n) {
      BuildParams B{};
      B.zob = zob; B.A = A; B.zpos = zpos; B.n_v = n_v; B.use_genes = D.use_genes;
      B.res = res.p; B.off = off.p; B.v = v.p; B.j = j.p; B.n = s->n;
      B.table = table.p; B.slot_mask = slots - 1;
      B.bloom = nullptr; B.bloom_byte_mask = 0; B.sliced = 0;   /* table only */
      const uint32_t grid = (uint32_t)((s->n + BLOCK_THREADS - 1) / BLOCK_THREADS);
      hipLaunchKernelGGL(build_index_kernel, dim3(grid), dim3(BLOCK_THREADS), 0, c->stream, B);
      HIP_TRY(c, hipGetLastError());
    }
    D.zob = zob; D.zpos = zpos;
    D.res = res.p; D.off = off.p; D.v = v.p; D.j = j.p; D.rep = rep.p;
    D.n = s->n;
    D.table = table.p; D.slot_mask = slots - 1;
    if (D.n) {
      const uint32_t grid = (uint32_t)((D.n + BLOCK_THREADS - 1) / BLOCK_THREADS);
      hipLaunchKernelGGL(count_duplicates_kernel, dim3(grid), dim3(BLOCK_THREADS), 0, c->stream, D);
      HIP_TRY(c, hipGetLastError());
    }
    unsigned long long hc = 0;
    HIP_TRY(c, hipMemcpyAsync(&hc, d_count.p, sizeof hc, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *out = hc;
    return CMPR_OK;
  }
  if (D.n) {
    const uint32_t grid = (uint32_t)((D.n + BLOCK_THREADS - 1) / BLOCK_THREADS);
    hipLaunchKernelGGL(count_duplicates_kernel, dim3(grid), dim3(BLOCK_THREADS), 0, c->stream, D);
    HIP_TRY(c, hipGetLastError());
  }
  unsigned long long hc = 0;
  HIP_TRY(c, hipMemcpyAsync(&hc, d_count.p, sizeof hc, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  *out = hc;
  return CMPR_OK;
}

/* ------------------------------------------------------------------ */
/* set 1: sort by length, cut into 64-query tiles, upload               */
/* ------------------------------------------------------------------ */

static int cmpr_set_queries_impl(cmpr_context *c, const cmpr_set_view *s)
{
  if (!c)
    return CMPR_EINVAL;
  if (!c->have_ref)
    return fail(c, CMPR_ESTATE, "cmpr_set_reference must be called first");
  std::string why;
  int rc = validate_view(c->opt, s, why);
  if (rc)
    return fail(c, rc, why);
  HIP_TRY(c, hipSetDevice(c->device));
  c->have_q = false;

  /* upload, validation, grouping by slice, tiles, chunks: all on the device
     (query_layout.hip) */
  if ((rc = cmpr_layout_queries(c, s)))
    return rc;

  const size_t cells = (size_t)c->R1 * c->R2;
  if ((rc = dev_alloc(c, c->matrix, cells))) return rc;
  if (is_f64_score(c->opt)) {
    if ((rc = dev_alloc(c, c->matrix_f64, cells))) return rc;
  } else {
    c->matrix_f64.release();
  }
  /* positives buffer of the deferred resolve: a capacity, not a limit -- what
     does not fit is resolved inline by the probe kernel */
  {
    const uint64_t total = c->pos_capacity > 0 ? (uint64_t)c->pos_capacity
                                               : std::max<uint64_t>(1u << 20, 4 * c->n1);
    const uint64_t S = (uint64_t)c->pos_segments;
    c->pos_cap = (total + S - 1) / S;                     /* per segment */
    if ((rc = dev_alloc(c, c->pos_buf, S * (c->pos_cap + WAVE)))) return rc;
    /* one block that is zeroed per launch with ONE memset: the segment counters,
       then the statistics, the two work cursors, the overflow flag, and the
       statistics + cursors of a redo pass (kernels_rows.h) */
    if ((rc = dev_alloc(c, c->pos_ctr, S * POS_CTR_STRIDE + CTR_TAIL))) return rc;
    c->d_stats = c->pos_ctr.p + S * POS_CTR_STRIDE;
    c->d_tile_counter = (uint32_t *)(c->d_stats + STAT_COUNT);
    c->d_overflow = c->d_stats + STAT_COUNT + 1;
    c->d_stats2 = c->d_overflow + 1;
    c->d_tile_counter2 = (uint32_t *)(c->d_stats2 + STAT_COUNT);
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->have_q = true;
  return CMPR_OK;
}

/* ------------------------------------------------------------------ */
/* the per-query loop                                                    */
/* ------------------------------------------------------------------ */

namespace {

/* enqueue: zero outputs, run the probe kernel over every tile.  `d_out` is the
   integer matrix to fill (ours or the caller's). */
int enqueue_overlap(cmpr_context *c, unsigned long long *d_out, hipStream_t st)
{
  const size_t cells = (size_t)c->R1 * c->R2;
  HIP_TRY(c, hipEventRecord(c->ev_start, st));
  if (cells)
    HIP_TRY(c, hipMemsetAsync(d_out, 0, cells * sizeof(unsigned long long), st));
  if (cells && is_f64_score(c->opt))
    HIP_TRY(c, hipMemsetAsync(c->matrix_f64.p, 0, cells * sizeof(double), st));
  const bool deferred = c->sliced && c->deferred_resolve;
  HIP_TRY(c, hipMemsetAsync(c->pos_ctr.p, 0, ((size_t)c->pos_segments * POS_CTR_STRIDE +
                                              CTR_TAIL) * sizeof(unsigned long long), st));
  c->launches = 0;
  bool launched = false;

  {
    const uint32_t slot = (uint32_t)(c->calls % cmpr_context::TIME_RING);
    c->ev_k0 = c->ring_k0[slot];
    c->ev_km = c->ring_km[slot];
    c->ev_k1 = c->ring_k1[slot];
    c->calls++;
  }
  HIP_TRY(c, hipEventRecord(c->ev_k0, st));
  if (c->ntiles > 0 && cells > 0) {
    const uint32_t A = (uint32_t)c->opt.alphabet_size;
    ProbeParams P{};
    P.zob = c->zob.p;
    P.zpos = c->zpos;
    P.n_v = c->opt.ignore_genes ? 0 : c->opt.n_v_genes;
    P.bloom = c->bloom.p;
    P.bloom_byte_mask = (uint32_t)((c->bloom_words - 1) << 3);
    P.table = c->table.p;
    P.slot_mask = c->slots - 1;
    P.res2 = c->res2.p;
    P.off2 = c->off2.p;
    P.v2 = c->v2.p;
    P.j2 = c->j2.p;
    P.rep2 = c->rep2.p;
    P.cnt2 = c->cnt2.p;
    P.rec2 = c->rec2.p;
    P.tiles = c->tiles.p;
    P.qres = c->qres.p;
    P.qv = c->qv.p;
    P.qgh = c->qgh.p;
    P.qj = c->qj.p;
    P.qrep = c->qrep.p;
    P.qcnt = c->qcnt.p;
    P.qlen = c->qlen.p;
    P.qorig = c->qorig.p;
    P.qck = c->qck.p;
    P.qrec = c->qrec.p;
    P.qhins = c->qhins.p;
    P.qhdel = c->qhdel.p;
    P.cw = c->cw.p;
    P.cmain = c->cmain.p;
    P.crp = c->crp.p;
    P.pair_q = c->pair_q;
    P.pair_h = c->pair_h;
    P.pair_count = c->pair_count;
    P.pair_cap = c->pair_cap;
    P.ntiles = c->ntiles;
    P.first_tile = 0;
    P.matrix = d_out;
    P.matrix_f64 = c->matrix_f64.p;
    P.R1 = c->R1;
    P.R2 = c->R2;
    P.score = c->opt.score;
    P.ignore_counts = c->opt.ignore_counts;
    P.lds_matrix = (cells <= 2048 && !is_f64_score(c->opt)) ? 1 : 0;
    P.tile_counter = c->d_tile_counter;
    P.stats = c->d_stats;
    P.geom = c->geom;
    P.chunks = c->chunks.p;
    P.tile_refs = c->tile_refs.p;
    P.small_tiles = c->small_tiles.p;
    P.nsmall = c->nsmall;
    P.nchunks = c->nchunks;
    P.debug = (uint32_t)c->debug;
    if (deferred) {
      P.pos_buf = c->pos_buf.p;
      P.pos_ctr = c->pos_ctr.p;
      P.pos_cap = c->pos_cap;
      P.pos_segments = (uint32_t)c->pos_segments;
    }

    /* variant 1: workgroups of 8 waves share one staged slice; when the chunks are
       short (many slices, few queries each) 4 waves keep more of them busy.
       variant 2: one workgroup per CU, 15 compute waves + the loader wave around a
       ring of two slices */
    int nw = c->sliced ? (int)c->waves_per_block : WAVES_PER_BLOCK;
    if (c->rows && !c->waves_per_block_forced)
      nw = 16;
    if (c->sliced && !c->rows && !c->waves_per_block_forced && c->nchunks > 0 &&
        (uint64_t)(c->ntiles - c->nsmall) < 6ull * c->nchunks)
      nw = 4;
    auto lds_for = [&](int waves) -> size_t {
      const size_t zrow = c->rows ? (c->opt.differences == 2 ? 2 * (size_t)A : (size_t)A)
                                  : c->sliced ? (size_t)(zrow_stride((int)A) + zdelta_entries((int)A))
                                              : (size_t)A;
      size_t b = zrow * c->zpos * sizeof(uint64_t) +
                 (P.lds_matrix ? cells * sizeof(unsigned long long) : 0) +
                 (size_t)waves * sizeof(WaveQueue);
      if (c->rows)
        b += RING * (size_t)c->geom.rw_words * ROW_WORD_BYTES + MAX_CLASS_RES * A * sizeof(uint32_t) +
             (c->opt.indels ? HEAVY_WORDS * sizeof(uint32_t) : 0) +
             RING * (sizeof(RingSlot) + (size_t)c->chunk_cap * sizeof(TileRef));
      else if (c->sliced)
        b += ((size_t)1 << c->geom.words_log2) * sizeof(uint64_t) +
             MAX_CLASS_RES * A * sizeof(uint32_t) + HEAVY_WORDS * sizeof(uint32_t) + 16 +
             (size_t)c->chunk_cap * sizeof(TileRef);
      return b;
    };
    size_t lds = lds_for(nw);
    while (lds > 160 * 1024 && c->sliced && nw > 4) {
      nw /= 2;                             /* long sequences: fewer wave queues */
      lds = lds_for(nw);
    }
    if (lds > 160 * 1024)
      return fail(c, CMPR_EUNSUPPORTED,
                  "sequences too long: Zobrist table does not fit the 160 KiB LDS");
    P.chunk_cap = c->chunk_cap;
    /* variant 2: the fast form hands its Bloom positives to resolve_kernel; the form
       that resolves inline is deferred_resolve = 0 and the redo pass below */
    ProbeFn fn = c->rows ? select_rows_kernel(c->opt, nw, !deferred)
                         : c->sliced ? select_sliced_kernel(c->opt, nw) : select_kernel(c->opt);
    if (c->rows && deferred)
      P.overflow = c->d_overflow;
    if (lds > 48 * 1024 && (c->attr_fn != (const void *)fn || c->attr_lds < lds)) {
      /* once per kernel and size, not once per launch */
      HIP_TRY(c, hipFuncSetAttribute((const void *)fn,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      c->attr_fn = (const void *)fn;
      c->attr_lds = lds;
    }
    /* resident workgroups per CU: LDS- and wave-limited, at most the tunable */
    uint64_t per_cu = std::min<uint64_t>((160 * 1024) / lds, 32 / (uint64_t)nw);
    if (c->rows) {
      /* variant 2 deals its chunks out statically over the workgroups of the grid:
         exactly as many as are resident at once (registers count too) */
      int occ = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)fn, nw * WAVE, lds) ==
              hipSuccess && occ > 0)
        per_cu = std::min<uint64_t>(per_cu, (uint64_t)occ);
    }
    per_cu = std::max<uint64_t>(1, std::min<uint64_t>(per_cu, (uint64_t)c->blocks_per_cu));
    uint64_t grid = (uint64_t)c->cus * per_cu;
    grid = std::min<uint64_t>(grid, c->sliced ? (uint64_t)c->nchunks + ((uint64_t)c->nsmall + nw - 1) / nw
                                              : ((uint64_t)c->ntiles + WAVES_PER_BLOCK - 1) /
                                                    WAVES_PER_BLOCK);
    grid = std::max<uint64_t>(grid, 1);
    hipLaunchKernelGGL(fn, dim3((uint32_t)grid), dim3((uint32_t)nw * WAVE), lds, st, P);
    HIP_TRY(c, hipGetLastError());
    c->launches = 1;
    launched = true;
    HIP_TRY(c, hipEventRecord(c->ev_km, st));
    if (deferred && !(c->debug & DBG_SKIP_RESOLVE)) {
      const size_t rlds = (BLOCK_THREADS / WAVE) * sizeof(CandQueue) +
                          (P.lds_matrix ? cells * sizeof(unsigned long long) : 0);
      /* 5 waves/SIMD fit its registers; a multiple of the segment count */
      uint32_t rgrid = (uint32_t)c->cus * (uint32_t)c->resolve_blocks_per_cu;
      rgrid = std::max<uint32_t>(1, rgrid / P.pos_segments) * P.pos_segments;
      hipLaunchKernelGGL(select_resolve(!c->opt.ignore_genes), dim3(rgrid), dim3(BLOCK_THREADS),
                         rlds, st, P);
      HIP_TRY(c, hipGetLastError());
      c->launches = 2;
      if (c->rows) {
        /* Redo pass: if the positives of the fast launch did not fit their buffer
           (flag set: resolve_kernel then did nothing), the same step with every
           positive resolved inline; otherwise its workgroups return at once.
           Capacity is therefore never a limit, and nothing here waits for the host. */
        ProbeFn fn2 = select_rows_kernel(c->opt, nw, true);
        if (lds > 48 * 1024 && (c->attr_fn2 != (const void *)fn2 || c->attr_lds2 < lds)) {
          HIP_TRY(c, hipFuncSetAttribute((const void *)fn2,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          c->attr_fn2 = (const void *)fn2;
          c->attr_lds2 = lds;
        }
        ProbeParams P2 = P;
        P2.pos_buf = nullptr;
        P2.redo = 1;
        P2.stats = c->d_stats2;
        P2.tile_counter = c->d_tile_counter2;
        hipLaunchKernelGGL(fn2, dim3((uint32_t)grid), dim3((uint32_t)nw * WAVE), lds, st, P2);
        HIP_TRY(c, hipGetLastError());
        c->launches = 3;
      }
    }
  }
  if (!launched)
    HIP_TRY(c, hipEventRecord(c->ev_km, st));
  HIP_TRY(c, hipEventRecord(c->ev_k1, st));
  return CMPR_OK;
}

int check_ready(cmpr_context *c)
{
  if (!c)
    return CMPR_EINVAL;
  if (!c->have_ref || !c->have_q)
    return fail(c, CMPR_ESTATE, "cmpr_set_reference and cmpr_set_queries must be called first");
  HIP_TRY(c, hipSetDevice(c->device));
  return CMPR_OK;
}

}  // namespace

extern "C" int cmpr_overlap_matrix_device(cmpr_context *c, void *d_matrix, void *stream)
{
  int rc = check_ready(c);
  if (rc)
    return rc;
  if (!d_matrix)
    return fail(c, CMPR_EINVAL, "d_matrix is NULL");
  if (is_f64_score(c->opt))
    return fail(c, CMPR_EINVAL, "ratio score needs cmpr_overlap_matrix_f64");
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  rc = enqueue_overlap(c, (unsigned long long *)d_matrix, st);
  if (rc)
    return rc;
  HIP_TRY(c, hipEventRecord(c->ev_stop, st));
  c->events_valid = true;
  if (!stream)
    HIP_TRY(c, hipStreamSynchronize(st));
  return CMPR_OK;
}

extern "C" int cmpr_overlap_matrix(cmpr_context *c, uint64_t *out)
{
  int rc = check_ready(c);
  if (rc)
    return rc;
  if (!out && (size_t)c->R1 * c->R2 > 0)
    return fail(c, CMPR_EINVAL, "matrix_out is NULL");
  if (is_f64_score(c->opt))
    return fail(c, CMPR_EINVAL, "ratio score needs cmpr_overlap_matrix_f64");
  rc = enqueue_overlap(c, c->matrix.p, c->stream);
  if (rc)
    return rc;
  const size_t cells = (size_t)c->R1 * c->R2;
  if (cells)
    HIP_TRY(c, hipMemcpyAsync(out, c->matrix.p, cells * sizeof(uint64_t),
                              hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipEventRecord(c->ev_stop, c->stream));
  c->events_valid = true;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return CMPR_OK;
}

static int cmpr_overlap_matrix_f64_impl(cmpr_context *c, double *out)
{
  int rc = check_ready(c);
  if (rc)
    return rc;
  if (!out && (size_t)c->R1 * c->R2 > 0)
    return fail(c, CMPR_EINVAL, "matrix_out is NULL");
  rc = enqueue_overlap(c, c->matrix.p, c->stream);
  if (rc)
    return rc;
  const size_t cells = (size_t)c->R1 * c->R2;
  std::vector<unsigned long long> tmp(cells);
  if (cells) {
    if (is_f64_score(c->opt))
      HIP_TRY(c, hipMemcpyAsync(out, c->matrix_f64.p, cells * sizeof(double),
                                hipMemcpyDeviceToHost, c->stream));
    else
      HIP_TRY(c, hipMemcpyAsync(tmp.data(), c->matrix.p, cells * sizeof(uint64_t),
                                hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(c, hipEventRecord(c->ev_stop, c->stream));
  c->events_valid = true;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (!is_f64_score(c->opt)) {
    const bool mean = c->opt.score == CMPR_SCORE_MEAN && !c->opt.ignore_counts;
    for (size_t k = 0; k < cells; k++)
      out[k] = mean ? (double)tmp[k] / 2 : (double)tmp[k];
  }
  return CMPR_OK;
}

extern "C" int cmpr_overlap_pairs(cmpr_context *c, uint64_t capacity, uint32_t *query_out,
                                  uint32_t *hit_out, uint64_t *count_out)
{
  int rc = check_ready(c);
  if (rc)
    return rc;
  if (!count_out || (capacity && (!query_out || !hit_out)))
    return fail(c, CMPR_EINVAL, "cmpr_overlap_pairs: NULL output");
  DevBuf<uint32_t> dq, dh;
  DevBuf<unsigned long long> dn;
  struct Cleanup {
    cmpr_context *c; DevBuf<uint32_t> &a, &b; DevBuf<unsigned long long> &n;
    ~Cleanup() { a.release(); b.release(); n.release();
                 c->pair_q = c->pair_h = nullptr; c->pair_count = nullptr; c->pair_cap = 0; }
  } cleanup{c, dq, dh, dn};
  if ((rc = dev_alloc(c, dq, (size_t)capacity))) return rc;
  if ((rc = dev_alloc(c, dh, (size_t)capacity))) return rc;
  if ((rc = dev_alloc(c, dn, 1))) return rc;
  HIP_TRY(c, hipMemsetAsync(dn.p, 0, sizeof(unsigned long long), c->stream));
  c->pair_q = dq.p;
  c->pair_h = dh.p;
  c->pair_count = dn.p;
  c->pair_cap = capacity;
  rc = enqueue_overlap(c, c->matrix.p, c->stream);
  if (rc)
    return rc;
  unsigned long long n = 0;
  HIP_TRY(c, hipMemcpyAsync(&n, dn.p, sizeof n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipEventRecord(c->ev_stop, c->stream));
  c->events_valid = true;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const uint64_t have = std::min<uint64_t>(n, capacity);
  if (have) {
    HIP_TRY(c, hipMemcpy(query_out, dq.p, have * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(hit_out, dh.p, have * sizeof(uint32_t), hipMemcpyDeviceToHost));
  }
  *count_out = n;
  return CMPR_OK;
}

extern "C" int cmpr_get_kernel_times(cmpr_context *c, uint32_t max, double *kernel_ms,
                                     double *probe_ms, uint32_t *count_out)
{
  if (!c || !count_out)
    return CMPR_EINVAL;
  *count_out = 0;
  if (!c->events_valid)
    return fail(c, CMPR_ESTATE, "no overlap call has been made");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipEventSynchronize(c->ev_k1));
  uint64_t n = std::min<uint64_t>(std::min<uint64_t>(max, c->calls), cmpr_context::TIME_RING);
  for (uint64_t k = 0; k < n; k++) {
    const uint32_t slot = (uint32_t)((c->calls - n + k) % cmpr_context::TIME_RING);
    float a = 0, b = 0;
    HIP_TRY(c, hipEventElapsedTime(&a, c->ring_k0[slot], c->ring_k1[slot]));
    HIP_TRY(c, hipEventElapsedTime(&b, c->ring_k0[slot], c->ring_km[slot]));
    if (kernel_ms)
      kernel_ms[k] = a;
    if (probe_ms)
      probe_ms[k] = b;
  }
  *count_out = (uint32_t)n;
  return CMPR_OK;
}

extern "C" int cmpr_get_stats(cmpr_context *c, cmpr_stats *out)
{
  if (!c || !out)
    return CMPR_EINVAL;
  if (!c->events_valid)
    return fail(c, CMPR_ESTATE, "no overlap call has been made");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipEventSynchronize(c->ev_stop));
  unsigned long long st[STAT_COUNT], ovf = 0;
  HIP_TRY(c, hipMemcpy(st, c->d_stats, sizeof st, hipMemcpyDeviceToHost));
  HIP_TRY(c, hipMemcpy(&ovf, c->d_overflow, sizeof ovf, hipMemcpyDeviceToHost));
  if (ovf)       /* the positives buffer overflowed: the redo pass did the step */
    HIP_TRY(c, hipMemcpy(st, c->d_stats2, sizeof st, hipMemcpyDeviceToHost));
  float k_ms = 0, t_ms = 0;
  HIP_TRY(c, hipEventElapsedTime(&k_ms, c->ev_k0, c->ev_k1));
  float p_ms = 0;
  HIP_TRY(c, hipEventElapsedTime(&p_ms, c->ev_k0, c->ev_km));
  HIP_TRY(c, hipEventElapsedTime(&t_ms, c->ev_start, c->ev_stop));
  memset(out, 0, sizeof *out);
  out->queries = c->n1;
  out->variants = st[STAT_VARIANTS];
  out->bloom_positive = st[STAT_BLOOM_POS];
  out->hash_equal = st[STAT_HASH_EQ];
  out->matches = st[STAT_MATCHES];
  out->filter_reads = st[STAT_READS];
  out->algorithmic_bytes = c->algorithmic_bytes;
  out->kernel_ms = k_ms;
  out->probe_ms = p_ms;
  out->total_ms = t_ms;
  out->kernel_launches = c->launches;
  return CMPR_OK;
}

extern "C" int cmpr_set_reference(cmpr_context *c, const cmpr_set_view *s,
                                  uint32_t longest_query)
{
  /* the header promises CMPR_ENOMEM, not an exception across the C boundary */
  try {
    return cmpr_set_reference_impl(c, s, longest_query);
  } catch (const std::bad_alloc &) {
    return fail(c, CMPR_ENOMEM, "out of host memory");
  }
}

extern "C" int cmpr_count_duplicates(cmpr_context *c, const cmpr_set_view *s, uint64_t *out)
{
  /* the header promises CMPR_ENOMEM, not an exception across the C boundary */
  try {
    return cmpr_count_duplicates_impl(c, s, out);
  } catch (const std::bad_alloc &) {
    return fail(c, CMPR_ENOMEM, "out of host memory");
  }
}

extern "C" int cmpr_set_queries(cmpr_context *c, const cmpr_set_view *s)
{
  /* the header promises CMPR_ENOMEM, not an exception across the C boundary */
  try {
    return cmpr_set_queries_impl(c, s);
  } catch (const std::bad_alloc &) {
    return fail(c, CMPR_ENOMEM, "out of host memory");
  }
}

extern "C" int cmpr_overlap_matrix_f64(cmpr_context *c, double *out)
{
  /* the header promises CMPR_ENOMEM, not an exception across the C boundary */
  try {
    return cmpr_overlap_matrix_f64_impl(c, out);
  } catch (const std::bad_alloc &) {
    return fail(c, CMPR_ENOMEM, "out of host memory");
  }
}
