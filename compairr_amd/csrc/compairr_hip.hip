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
#include "kernels_pairs2.h"
#include "select.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <mutex>
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
/* one counter block: the segment counters of the positives buffer, the tail above, the chunk counters */
static inline size_t ctr_block_words(uint64_t segments)
{
  return (size_t)segments * POS_CTR_STRIDE + CTR_TAIL + DEAL_WORDS;
}
namespace {
void use_counter_block(cmpr_context *c, int which);
void invalidate_plan(cmpr_context *c);
int make_plan(cmpr_context *c);
}

int validate_view(const cmpr_options &o, const cmpr_set_view *s, std::string &why, bool on_device)
{
  if (!s) { why = "set view is NULL"; return CMPR_EINVAL; }
  if (s->n_repertoires == 0 && s->n > 0) { why = "n_repertoires is 0"; return CMPR_EINVAL; }
  if (s->n >= 0xffffffc0ull) { why = "more than 2^32-64 sequences in one set"; return CMPR_EUNSUPPORTED; }
  if (s->n == 0)
    return CMPR_OK;
  if (!s->offsets || !s->repertoire) { why = "offsets/repertoire is NULL"; return CMPR_EINVAL; }
  if (!o.ignore_genes && (!s->v_gene || !s->j_gene)) { why = "v_gene/j_gene is NULL without ignore_genes"; return CMPR_EINVAL; }
  if (!o.ignore_counts && !s->count) { why = "count is NULL without ignore_counts"; return CMPR_EINVAL; }
  if (on_device) {               /* (offsets[0] and offsets[n] are fetched and checked by the caller) */
    if (!s->residues) { why = "residues is NULL"; return CMPR_EINVAL; }
    return CMPR_OK;
  }
  if (!s->residues && s->offsets[s->n] > 0) { why = "residues is NULL"; return CMPR_EINVAL; }
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

ProbeFn select_rows_kernel(const cmpr_options &o, int nw, bool inline_resolve, bool wide)
{
  const int A = o.alphabet_size, D = o.differences;
  const bool i = o.indels != 0, g = !o.ignore_genes;
  if (wide) {                 /* four amino-acid class residues (layout.h kernel_class_res) */
    if (inline_resolve)
      switch (nw) {
      case 4:  return select_probe_v2_wide_inline_nw4(A, D, i, g);
      case 16: return select_probe_v2_wide_inline_nw16(A, D, i, g);
      default: return select_probe_v2_wide_inline_nw8(A, D, i, g);
      }
    switch (nw) {
    case 4:  return select_probe_v2_wide_nw4(A, D, i, g);
    case 16: return select_probe_v2_wide_nw16(A, D, i, g);
    default: return select_probe_v2_wide_nw8(A, D, i, g);
    }
  }
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
  /* k0 and km only ever feed hipEventElapsedTime: no system-scope fence behind them (a
     fenced record holds the stream up for ~5 us; k1 is also what the host waits on) */
  const unsigned tflags = getenv("COMPAIRR_HIP_EVENT_FENCE") ? hipEventDefault : hipEventDisableSystemFence;
  for (uint32_t i = 0; i < cmpr_context::TIME_RING; i++) {
    CREATE_TRY(hipEventCreateWithFlags(&c->ring_k0[i], tflags));
    CREATE_TRY(hipEventCreateWithFlags(&c->ring_km[i], tflags));
    CREATE_TRY(hipEventCreate(&c->ring_k1[i]));
  }
  c->ev_k0 = c->ring_k0[0];
  c->ev_km = c->ring_km[0];
  c->ev_k1 = c->ring_k1[0];
  CREATE_TRY(hipEventCreate(&c->ev_stop));
  CREATE_TRY(hipEventCreateWithFlags(&c->ev_usage, hipEventDisableTiming));
  CREATE_TRY(hipMalloc((void **)&c->d_usage, 2 * sizeof(unsigned long long)));
  CREATE_TRY(hipMemset(c->d_usage, 0, 2 * sizeof(unsigned long long)));
  CREATE_TRY(hipHostMalloc((void **)&c->h_usage, 2 * sizeof(unsigned long long), hipHostMallocDefault));
  c->h_usage[0] = c->h_usage[1] = 0;
  CREATE_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  for (uint32_t i = 0; i < cmpr_context::NCOPY_EV; i++)
    CREATE_TRY(hipEventCreateWithFlags(&c->ev_copy[i], hipEventDisableTiming));
  for (hipEvent_t &e : c->ev_layout)
    CREATE_TRY(hipEventCreateWithFlags(&e, tflags));

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
  cmpr_release_reservations();              /* (what cmpr_warm_up_sized reserved and nobody took) */
  if (c->stream)
    (void)hipStreamSynchronize(c->stream);
  invalidate_plan(c);
  c->zob.release();
  c->res2.release(); c->off2.release(); c->cnt2.release();
  c->page_tab.release();
  c->bloom.release(); c->v2.release(); c->j2.release(); c->rep2.release();
  c->rec2.release(); c->voff2.release(); c->bmap2.release();
  c->tiles.release(); c->qres.release(); c->qv.release(); c->qj.release(); c->qgh.release();
  c->qlen.release(); c->qck.release();
  c->qhins.release(); c->qhdel.release(); c->items.release(); c->cpk.release(); c->qpk.release(); c->slice_items.release(); c->qrec.release();
  c->matrix.release(); c->matrix_f64.release();
  c->pos_buf.release(); c->pos_ctr.release(); c->d_ctab.release(); c->chunks.release(); c->tile_refs.release(); c->small_tiles.release();
  for (uint32_t i = 0; i < cmpr_context::TIME_RING; i++) {
    if (c->ring_k0[i]) (void)hipEventDestroy(c->ring_k0[i]);
    if (c->ring_km[i]) (void)hipEventDestroy(c->ring_km[i]);
    if (c->ring_k1[i]) (void)hipEventDestroy(c->ring_k1[i]);
  }
  if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
  if (c->ev_usage) (void)hipEventDestroy(c->ev_usage);
  if (c->d_usage) (void)hipFree(c->d_usage);
  if (c->h_usage) (void)hipHostFree(c->h_usage);
  c->arena_a.release();
  c->arena_b.release();
  if (c->stage_host) (void)hipHostFree(c->stage_host);
  if (c->h_sizes) (void)hipHostFree(c->h_sizes);
  for (uint32_t i = 0; i < cmpr_context::NCOPY_EV; i++)
    if (c->ev_copy[i]) (void)hipEventDestroy(c->ev_copy[i]);
  for (hipEvent_t e : c->ev_layout)
    if (e) (void)hipEventDestroy(e);
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" const char *cmpr_last_error(const cmpr_context *c)
{
  return c ? c->err.c_str() : cmpr_create_error().c_str();
}

extern "C" uint32_t cmpr_rows(const cmpr_context *c) { return c ? c->R1 : 0; }
extern "C" uint32_t cmpr_cols(const cmpr_context *c) { return c ? c->R2 : 0; }

static int set_tunable_value(cmpr_context *c, const std::string &n, int64_t value, bool &affects_plan);

extern "C" int cmpr_set_tunable(cmpr_context *c, const char *name, int64_t value)
{
  if (!c || !name)
    return CMPR_EINVAL;
  std::string n(name);
  /* A tunable that was set: the cached step (kernels, grid) is worked out again and the
     no-redo shortcut of variant 2 has to be earned again -- only behind a successful change
     (a refused name or value leaves the context as it was), and not for the knobs no
     launch depends on. */
  bool plan = true;
  const int rc = set_tunable_value(c, n, value, plan);
  if (rc == CMPR_OK && plan) {
    invalidate_plan(c);
    c->usage_pending = c->never_overflows = false;
  }
  return rc;
}

static int set_tunable_value(cmpr_context *c, const std::string &n, int64_t value, bool &affects_plan)
{
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
      return fail(c, CMPR_EINVAL, "class_residues must be -1..4 (amino acids; four: variant 2, d = 1, else three) / -1..8 (nucleotides)");
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
    affects_plan = false;
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
  } else if (n == "d2_pairs") {
    if (value < -1 || value > 1)
      return fail(c, CMPR_EINVAL, "d2_pairs must be -1 (auto), 0 or 1");
    if (c->have_ref)
      return fail(c, CMPR_ESTATE, "set d2_pairs before cmpr_set_reference");
    c->d2_pairs = value;
  } else if (n == "d2_buffers") {
    if (value < 1 || value > 2)
      return fail(c, CMPR_EINVAL, "d2_buffers must be 1 or 2");
    if (c->have_ref)
      return fail(c, CMPR_ESTATE, "set d2_buffers before cmpr_set_reference");
    c->d2_buffers = value;
  } else if (n == "chunk_deal") {
    if (value < 0 || value > 1)
      return fail(c, CMPR_EINVAL, "chunk_deal must be 0 (static) or 1 (by a counter)");
    c->chunk_deal = value;
  } else if (n == "narrow_upload") {
    if (value < -1 || value > 1)
      return fail(c, CMPR_EINVAL, "narrow_upload must be -1 (auto), 0 or 1");
    c->narrow_upload = value;
    affects_plan = false;
  } else if (n == "item_wg" || n == "layout_recompute" || n == "layout_timing" || n == "layout_zob_lds" ||
             n == "record_tiles") {
    if (value < 0 || value > (n == "record_tiles" ? 2 : 1))
      return fail(c, CMPR_EINVAL, n + (n == "record_tiles" ? " must be 0, 1 or 2" : " must be 0 or 1"));
    if (n == "record_tiles" && c->have_q)
      return fail(c, CMPR_ESTATE, "set record_tiles before cmpr_set_queries");
    (n == "item_wg" ? c->item_wg : n == "layout_recompute" ? c->layout_recompute :
     n == "layout_zob_lds" ? c->layout_zob_lds : n == "record_tiles" ? c->record_tiles : c->layout_timing) = value;
    affects_plan = false;
  } else if (n == "assume_never_overflows") {
    /* TEST ONLY: the next launch runs without redo pass as if the margin had been
       shown (tests/test_gpu_parity.py forces an overflow behind it) */
    c->force_no_redo = value != 0;
    c->usage_pending = false;               /* (a measurement in flight would overrule the pretence) */
    affects_plan = false;
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
  } else if (n == "pos_grow") {
    if (value < -1 || value > 1)
      return fail(c, CMPR_EINVAL, "pos_grow must be -1 (auto), 0 or 1");
    c->pos_grow = value;
  } else if (n == "pos_capacity") {
    if (value < 0)
      return fail(c, CMPR_EINVAL, "pos_capacity must be >= 0");
    if (c->have_q)
      return fail(c, CMPR_ESTATE, "set pos_capacity before cmpr_set_queries");
    c->pos_capacity = value;
  } else if (n == "work_shard_count" || n == "work_shard_index") {
    if (value < 0 || value > 65535 || (n == "work_shard_count" && value < 1))
      return fail(c, CMPR_EINVAL, "work_shard_count must be 1..65535, work_shard_index 0..count-1");
    if (c->have_q)
      return fail(c, CMPR_ESTATE, "set the work shard before cmpr_set_queries");
    (n == "work_shard_count" ? c->work_shard_count : c->work_shard_index) = value;
  } else if (n == "small_slice_tiles") {
    if (value < 0 || value > 64)
      return fail(c, CMPR_EINVAL, "small_slice_tiles must be 0..64");
    if (c->have_q)
      return fail(c, CMPR_ESTATE, "set small_slice_tiles before cmpr_set_queries");
    c->small_slice_tiles = value;
  } else if (n == "sub2_items") {
    if (value < -1 || value > 1)
      return fail(c, CMPR_EINVAL, "sub2_items must be -1 (default), 0 or 1");
    if (c->have_q)
      return fail(c, CMPR_ESTATE, "set sub2_items before cmpr_set_queries");
    c->sub2_items = value;
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
  } else if (n == "slice_pages") {
    if (value < -1 || value > (int64_t)PAGE_E_MAX)
      return fail(c, CMPR_EINVAL, "slice_pages must be -1..3");
    if (c->have_ref)
      return fail(c, CMPR_ESTATE, "set slice_pages before cmpr_set_reference");
    c->slice_pages = value;
  } else if (n == "page_budget") {
    if (value < 0 || value > (1 << 24))
      return fail(c, CMPR_EINVAL, "page_budget must be 0..2^24");
    if (c->have_ref)
      return fail(c, CMPR_ESTATE, "set page_budget before cmpr_set_reference");
    c->page_budget = value;
  } else if (n == "bucket_bitmap") {
    if (value < -1 || value > 1)
      return fail(c, CMPR_EINVAL, "bucket_bitmap must be -1 (auto), 0 or 1");
    c->bucket_bitmap = value;
  } else if (n == "fill_slices") {
    if (value < 0 || value > 1)
      return fail(c, CMPR_EINVAL, "fill_slices must be 0 or 1");
    if (c->have_ref)
      return fail(c, CMPR_ESTATE, "set fill_slices before cmpr_set_reference");
    c->fill_slices = value;
  } else if (n == "direct_slices_log2") {
    if (value < -1 || value > 12)
      return fail(c, CMPR_EINVAL, "direct_slices_log2 must be -1 (auto) or 0..12");
    c->direct_slices_log2 = value;
  } else if (n == "row_filter_x16") {
    if (value < 8 || value > 128)
      return fail(c, CMPR_EINVAL, "row_filter_x16 must be 8..128 (sixteenths of a byte per entry)");
    if (c->have_ref)
      return fail(c, CMPR_ESTATE, "set row_filter_x16 before cmpr_set_reference");
    c->row_filter_x16 = value;
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
  else if (n == "waves_per_block") *value = c->plan.valid ? (int64_t)c->plan.nw : c->waves_per_block;
  else if (n == "narrow_upload") *value = c->narrow_upload;
  else if (n == "d2_buffers") *value = c->have_ref && c->d2pairs ? (int64_t)c->geom.nbuf : c->d2_buffers;
  else if (n == "d2_pairs") *value = c->have_ref ? (c->d2pairs ? 1 : 0) : c->d2_pairs;
  else if (n == "chunk_deal") *value = c->chunk_deal;
  else if (n == "layout_upload_us") *value = (int64_t)(c->layout_upload_ms * 1e3);
  else if (n == "layout_tail_us") *value = (int64_t)(c->layout_tail_ms * 1e3);
  else if (n == "layout_total_us") *value = (int64_t)(c->layout_total_ms * 1e3);
  else if (n == "item_wg") *value = c->item_wg;
  else if (n == "layout_recompute") *value = c->layout_recompute;
  else if (n == "layout_timing") *value = c->layout_timing;
  else if (n == "record_tiles") *value = c->have_q ? (c->rec_tiles ? (c->rec_hash ? 1 : 2) : 0) : c->record_tiles;
  else if (n == "layout_keys_us") *value = (int64_t)(c->layout_kernel_ms[0] * 1e3);
  else if (n == "layout_sizes_us") *value = (int64_t)(c->layout_kernel_ms[1] * 1e3);
  else if (n == "layout_scatter_us") *value = (int64_t)(c->layout_kernel_ms[2] * 1e3);
  else if (n == "layout_tiles_us") *value = (int64_t)(c->layout_kernel_ms[3] * 1e3);
  else if (n == "layout_order_us") *value = (int64_t)(c->layout_kernel_ms[4] * 1e3);
  else if (n == "never_overflows") *value = c->never_overflows ? 1 : 0;
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
  else if (n == "sub2_items") *value = c->sub2_items;
  else if (n == "work_shard_count") *value = c->work_shard_count;
  else if (n == "work_shard_index") *value = c->work_shard_index;
  else if (n == "class_rows_unstaged") *value = c->class_rows_unstaged;
  else if (n == "deferred_resolve") *value = c->deferred_resolve;
  else if (n == "table_log2_delta") *value = c->table_log2_delta;
  else if (n == "row_filter_x16") *value = c->row_filter_x16;
  else if (n == "fill_slices") *value = c->fill_slices;
  else if (n == "direct_slices_log2") *value = c->direct_slices_log2;
  else if (n == "bucket_bitmap") *value = c->bucket_bitmap;
  else if (n == "slice_pages") *value = c->slice_pages;
  else if (n == "page_budget") *value = c->page_budget;
  else if (n == "page_slices") *value = (int64_t)c->page_slices;
  else if (n == "host_threads") *value = c->host_threads;
  else if (n == "pos_capacity") *value = (int64_t)(c->pos_cap * c->pos_segments);
  else if (n == "pos_segments") *value = c->pos_segments;
  else if (n == "pos_grow") *value = c->pos_grow;
  else if (n == "resolve_blocks_per_cu") *value = c->resolve_blocks_per_cu;
  else if (n == "query_slots") *value = (int64_t)c->ntiles * WAVE;
  else if (n == "items") *value = (int64_t)c->items.n;       /* variant 2 / sub2: item slots, padding included */
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
    D.rec = c->rec2.p;
    D.dir_mask = (uint32_t)(c->slots - 1);
  } else {
    std::string why;
    if ((rc = validate_view(c->opt, s, why)))
      return fail(c, rc, why);
    /* upload + validation on the device (query_layout.hip) */
    uint32_t longest = 0;
    std::vector<double> tot;
    DevBuf<uint64_t> cnt_tmp;
    struct Ct { DevBuf<uint64_t> &z; ~Ct() { z.release(); } } ctclean{cnt_tmp};
    if ((rc = cmpr_upload_and_validate(c, s, res, off, v, j, rep, cnt_tmp, longest, tot)))
      return rc;
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

/* what every way of handing over queries does first (cmpr_set_queries, _device, _routed,
   cmpr_route_queries): the old set, and what was left unasked about it, goes */
static int retire_queries(cmpr_context *c)
{
  HIP_TRY(c, hipSetDevice(c->device));
  /* launches on the old set that may still be running on a caller's stream read what the
     layout is about to overwrite */
  if (c->events_valid)
    HIP_TRY(c, hipEventSynchronize(c->ev_k1));
  /* asynchronous launches on the OLD set that nobody has asked about (cmpr_get_stats): the
     question is answered here, loudly, rather than carried over to the new set -- a sticky
     overflow word must not fail the first cmpr_get_stats on the new one, nor be lost */
  if (c->async_unchecked || c->async_overflowed) {
    unsigned long long sticky = 0;
    HIP_TRY(c, hipMemcpy(&sticky, c->d_usage + 1, sizeof sticky, hipMemcpyDeviceToHost));
    const bool bad = sticky != 0 || c->async_overflowed;
    if (sticky)
      HIP_TRY(c, hipMemset(c->d_usage + 1, 0, sizeof sticky));
    c->async_unchecked = c->async_overflowed = false;
    if (bad)
      return fail(c, CMPR_ESTATE, "positives buffer overflowed in a launch without redo pass on the previous "
                                  "query set: its result was invalid (cmpr_get_stats was not asked); repeat "
                                  "the call to set the new queries");
  }
  c->have_q = false;
  c->usage_pending = c->never_overflows = false;
  c->last_without_redo = false;
  c->events_valid = false;
  c->calls = 0;
  invalidate_plan(c);
  return CMPR_OK;
}

static int cmpr_set_queries_impl(cmpr_context *c, const LayoutSource &src_in)
{
  LayoutSource src = src_in;
  if (!c)
    return CMPR_EINVAL;
  if (!c->have_ref)
    return fail(c, CMPR_ESTATE, "cmpr_set_reference must be called first");
  int rc;
  if (src.kind != LayoutSource::RECORDS) {
    std::string why;
    rc = validate_view(c->opt, src.view, why, src.kind == LayoutSource::DEVICE);
    if (rc)
      return fail(c, rc, why);
  } else {
    if (src.nrec && !src.d_records)
      return fail(c, CMPR_EINVAL, "d_records is NULL");
    if (src.nrec && src.n_rep == 0)
      return fail(c, CMPR_EINVAL, "n_repertoires is 0");
    if (src.nrec >= 0xffffffc0ull || src.n_total > 0xffffffffull)
      return fail(c, CMPR_EUNSUPPORTED, "more than 2^32-64 sequences in one set");
  }
  if ((rc = retire_queries(c)))
    return rc;

  /* the per-launch scratch, queued behind the layout's kernels (LayoutSource::finish) */
  src.finish = [c]() -> int {
  int rc;
  /* (kept from call to call when large enough: no hipMalloc / hipFree in the steady state) */
  const size_t cells = (size_t)c->R1 * c->R2;
  if ((rc = dev_reserve(c, c->matrix, cells))) return rc;
  if (is_f64_score(c->opt)) {
    if ((rc = dev_reserve(c, c->matrix_f64, cells))) return rc;
  } else {
    c->matrix_f64.release();
  }
  /* positives buffer of the deferred resolve: a capacity, not a limit -- what
     does not fit is resolved inline by the probe kernel */
  {
    /* (a work shard queues its share of the positives) */
    const uint64_t per_query = c->rows && c->opt.differences == 2 ? 32 : 4;
    /* (routed records: what is resident IS the share) */
    const uint64_t total = c->pos_capacity > 0 ? (uint64_t)c->pos_capacity
                                               : (1u << 20) + per_query * c->n1 /
                                                     (c->routed ? 1ull : (uint64_t)c->work_shard_count);
    const uint64_t S = (uint64_t)c->pos_segments;
    c->pos_cap = (total + S - 1) / S;                     /* per segment */
    if ((rc = dev_reserve(c, c->pos_buf, S * (c->pos_cap + WAVE)))) return rc;
    /* one block that is zeroed per launch with ONE memset: the segment counters,
       then the statistics, the two work cursors, the overflow flag, and the
       statistics + cursors of a redo pass (kernels_rows.h) */
    /* (two of them: a launch uses one and clears the other for the next launch --
       reduce_partials_kernel -- so that a step starts without a memset) */
    if ((rc = dev_reserve(c, c->pos_ctr, 2 * ctr_block_words(S)))) return rc;
    c->ctr_clean = false;
    use_counter_block(c, 0);
    /* partial results of the workgroups (ProbeParams::part); cleared here, and by
       reduce_partials_kernel after every launch */
    const size_t cells = (size_t)c->R1 * c->R2;
    c->part_stride = (uint32_t)((cells <= PART_CELLS_MAX && !is_f64_score(c->opt) ? cells : 0) + STAT_COUNT);
    if ((rc = dev_reserve(c, c->part, (size_t)NPART * c->part_stride))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->part.p, 0, (size_t)NPART * c->part_stride * sizeof(unsigned long long),
                              c->stream));
  }
  return CMPR_OK;
  };

  /* upload, validation, grouping by slice, tiles, chunks: all on the device
     (query_layout.hip); returns with the stream drained */
  if ((rc = cmpr_layout_queries(c, src)))
    return rc;
  if ((rc = make_plan(c))) return rc;
  c->have_q = true;
  return CMPR_OK;
}

/* ------------------------------------------------------------------ */
/* the per-query loop                                                    */
/* ------------------------------------------------------------------ */

namespace {

/* the counter block a launch works with (block 0 or 1 of pos_ctr's allocation) */
void use_counter_block(cmpr_context *c, int which)
{
  const size_t blk = ctr_block_words((uint64_t)c->pos_segments);
  c->ctr_cur = c->pos_ctr.p + (size_t)which * blk;
  c->ctr_other = c->pos_ctr.p + (size_t)(1 - which) * blk;
  c->d_stats = c->ctr_cur + (size_t)c->pos_segments * POS_CTR_STRIDE;
  c->d_tile_counter = (uint32_t *)(c->d_stats + STAT_COUNT);
  c->d_overflow = c->d_stats + STAT_COUNT + 1;
  c->d_stats2 = c->d_overflow + 1;
  c->d_tile_counter2 = (uint32_t *)(c->d_stats2 + STAT_COUNT);
  c->d_deal = c->ctr_cur + blk - DEAL_WORDS;
}

/* the cached plan is stale (sets or tunables changed) */
void invalidate_plan(cmpr_context *c)
{
  c->plan.valid = false;
}

/* What a step launches, worked out once per (sets, tunables): kernels, grid, LDS and
   every kernel argument that does not change from launch to launch. */
int make_plan(cmpr_context *c)
{
  int rc;
  invalidate_plan(c);
  StepPlan &S = c->plan;
  S = StepPlan();
  const size_t cells = (size_t)c->R1 * c->R2;
  S.cells = cells;
  S.ctr_blk = ctr_block_words((uint64_t)c->pos_segments);
  S.will_launch = c->ntiles > 0 && cells > 0;
  S.deferred = c->sliced && c->deferred_resolve;
  /* (variant 2 resolving inline adds to the matrix where it lies: cleared before, not
     written by the reduce kernel) */
  const bool rows_inline = c->rows && !c->d2pairs && !S.deferred;
  S.reduce_writes = S.will_launch && cells <= 2048 && !is_f64_score(c->opt) && !rows_inline;
  /* (kernels_pairs2.h resolves what does not fit its buffer inline: no redo launch) */
  S.redo_kind = c->rows && !c->d2pairs && S.deferred && !(c->debug & DBG_SKIP_RESOLVE);
  S.nw = c->sliced ? (uint32_t)c->waves_per_block : WAVES_PER_BLOCK;
  if (!S.will_launch) {
    S.valid = true;
    return CMPR_OK;
  }
  const uint32_t A = (uint32_t)c->opt.alphabet_size;
  ProbeParams &P = S.P;
  P.zob = c->zob.p;
  P.zpos = c->zpos;
  P.n_v = c->opt.ignore_genes ? 0 : c->opt.n_v_genes;
  P.n_j_keys = c->opt.ignore_genes ? 0 : c->opt.n_j_genes;
  P.rec_tiles = c->rec_tiles ? (c->rec_hash ? 2u : 1u) : 0u;
  P.bloom = c->bloom.p;
  P.bloom_byte_mask = (uint32_t)((c->bloom_words - 1) << 3);
  P.dir_mask = (uint32_t)(c->slots - 1);
  /* (asked before a slot is read where most positives are false -- d = 2: six of seven with single rows, 49 of 50
     with the nucleotide pair rows: resolve 33 -> 27 ms on the 24.2M self-comparison, 4.7 -> 3.7 ms at cfg5; where
     most are true -- d <= 1 -- the question is one more dependent read per hit: +4 .. +15 %) */
  const bool ask_bitmap = c->bucket_bitmap < 0 ? c->opt.differences >= 2 : c->bucket_bitmap != 0;
  P.bmap = ask_bitmap ? c->bmap2.p : nullptr;
  P.rec_packed = c->opt.alphabet_size == 4 ? 1u : 0u;
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
  P.qlen = c->qlen.p;
  P.qck = c->qck.p;
  P.qrec = c->qrec.p;
  P.qhins = c->qhins.p;
  P.qhdel = c->qhdel.p;
  P.items = c->items.p;
  P.cpk = c->cpk.p;
  P.qpk = c->qpk.p;
  P.slice_items = c->slice_items.p;
  P.ntiles = c->ntiles;
  P.first_tile = 0;
  P.sub2_items = c->sub2_active ? 1u : 0u;
  P.part = c->part.p;
  P.part_stride = c->part_stride;
  P.part_cells = c->part_stride - STAT_COUNT;
  /* (variant 0 at d >= 1 lays every query out and takes every count-th tile; at d = 0 -- and in the sliced
     layouts -- what is laid out is what this context works on) */
  const bool direct = !c->sliced && c->opt.differences == 0;
  P.work_first = direct ? 0u : (uint32_t)c->work_shard_index;
  P.work_step = direct ? 1u : (uint32_t)c->work_shard_count;
  P.matrix_f64 = c->matrix_f64.p;
  P.R1 = c->R1;
  P.R2 = c->R2;
  P.score = c->opt.score;
  P.ignore_counts = c->opt.ignore_counts;
  P.lds_matrix = (cells <= 2048 && !is_f64_score(c->opt) && !rows_inline) ? 1 : 0;
  P.geom = c->geom;
  P.chunks = c->chunks.p;
  P.tile_refs = c->tile_refs.p;
  P.small_tiles = c->small_tiles.p;
  P.nsmall = c->nsmall;
  P.nchunks = c->nchunks;
  P.debug = (uint32_t)c->debug;
  if (S.deferred) {
    P.pos_buf = c->pos_buf.p;
    P.pos_cap = c->pos_cap;
    P.pos_segments = (uint32_t)c->pos_segments;
  }

  /* variant 1: workgroups of 8 waves share one staged slice; when the chunks are
     short (many slices, few queries each) 4 waves keep more of them busy.
     variant 2: one workgroup of 16 waves per CU around a ring of slices */
  int nw = (int)S.nw;
  if (c->rows && !c->waves_per_block_forced)
    nw = 16;
  if (c->sliced && !c->rows && !c->waves_per_block_forced && c->nchunks > 0 &&
      (uint64_t)(c->ntiles - c->nsmall) < 6ull * c->nchunks)
    nw = 4;
  auto lds_for = [&](int waves) -> size_t {
    const size_t zrow = c->rows ? (size_t)zs_of((int)A, c->opt.differences == 2 ? 2 : 1, pair_rows(c))
                                : c->sliced ? (size_t)(zrow_stride((int)A) + zdelta_entries((int)A))
                                            : (size_t)A;
    size_t b = zrow * c->zpos * sizeof(uint64_t) +
               (P.lds_matrix && !c->rows ? cells * sizeof(unsigned long long) : 0) +
               (size_t)waves * sizeof(WaveQueue);
    if (c->rows)
      b += RING * (size_t)c->geom.rw_words * ROW_WORD_BYTES + MAX_CLASS_RES * A * sizeof(uint32_t) +
           (c->opt.indels ? HEAVY_WORDS * sizeof(uint32_t) : 0) +
           RING * (sizeof(RingSlot) + (size_t)c->chunk_cap * sizeof(TileRef)) +
           (c->rec_tiles && !c->opt.ignore_genes ? ((size_t)c->opt.n_v_genes + c->opt.n_j_genes) * sizeof(uint64_t) : 0) +
           (c->rec_tiles && c->opt.indels ? (size_t)c->geom.off_cr * sizeof(uint32_t) : 0);
    else if (c->sliced)
      b += ((size_t)1 << c->geom.words_log2) * sizeof(uint64_t) +
           MAX_CLASS_RES * A * sizeof(uint32_t) + HEAVY_WORDS * sizeof(uint32_t) + 16 +
           (size_t)c->chunk_cap * sizeof(TileRef);
    return b;
  };
  if (c->d2pairs) {
    /* kernels_pairs2.h: one workgroup of 16 waves per CU, two slice buffers */
    nw = 16;
    const size_t npairs = ((size_t)c->zpos + 1) / 2;
    const size_t lds2 = (size_t)c->geom.nbuf * c->geom.rw_words * ROW_WORD_BYTES +
                        (16 * (size_t)c->zpos + P2_PZ * npairs) * sizeof(uint64_t) +
                        (P.lds_matrix ? cells * sizeof(unsigned long long) : 0) + 16 * sizeof(WaveQueue) +
                        MAX_CLASS_RES * A * sizeof(uint32_t) + 2 * sizeof(P2Slot) +
                        2 * (size_t)c->chunk_cap * sizeof(TileRef);
    if (lds2 > 160 * 1024)
      return fail(c, CMPR_EUNSUPPORTED, "the pair-row kernel of d = 2 does not fit the 160 KiB LDS");
    ProbeFn fn = select_probe_pairs2(!c->opt.ignore_genes);
    if ((rc = raise_lds_limit(c, (const void *)fn, lds2))) return rc;
    P.chunk_cap = c->chunk_cap;
    uint64_t grid = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)c->cus, (uint64_t)c->nchunks));
    S.fn = fn;
    S.grid = (uint32_t)grid;
    S.nw = 16;
    S.lds = lds2;
    if (S.deferred) {
      S.rlds = (BLOCK_THREADS / WAVE) * sizeof(CandQueue) +
               (P.lds_matrix ? cells * sizeof(unsigned long long) : 0);
      uint32_t rgrid = (uint32_t)c->cus * (uint32_t)c->resolve_blocks_per_cu;
      S.rgrid = std::max<uint32_t>(1, rgrid / P.pos_segments) * P.pos_segments;
      S.rfn = select_resolve(!c->opt.ignore_genes);
    }
    S.valid = true;
    return CMPR_OK;
  }
  size_t lds = lds_for(nw);
  while (lds > 160 * 1024 && c->sliced && nw > 4) {
    nw /= 2;                             /* long sequences: fewer wave queues */
    lds = lds_for(nw);
  }
  if (lds > 160 * 1024)
    return fail(c, CMPR_EUNSUPPORTED,
                "sequences too long: Zobrist table does not fit the 160 KiB LDS");
  P.chunk_cap = c->chunk_cap;
  P.deal = c->rows ? (uint32_t)c->chunk_deal : 0u;
  /* variant 2: the fast form hands its Bloom positives to resolve_kernel; the form
     that resolves inline is deferred_resolve = 0 and the redo pass */
  /* a layout with four amino-acid class residues runs on the wide instantiations */
  const bool wide = c->rows && A != 4 && c->geom.k > kernel_class_res((uint32_t)A, false);
  ProbeFn fn = c->rows ? select_rows_kernel(c->opt, nw, !S.deferred, wide)
                       : c->sliced ? select_sliced_kernel(c->opt, nw) : select_kernel(c->opt);
  if (fn == nullptr)
    return fail(c, CMPR_EUNSUPPORTED, "no kernel for this layout (class residues)");
  if ((rc = raise_lds_limit(c, (const void *)fn, lds))) return rc;
  /* resident workgroups per CU: LDS- and wave-limited, at most the tunable */
  uint64_t per_cu = std::min<uint64_t>((160 * 1024) / lds, 32 / (uint64_t)nw);
  if (c->rows || direct) {
    /* variant 2 deals its chunks out statically over the workgroups of the grid:
       exactly as many as are resident at once (registers count too); so does the d = 0
       kernel its tiles (a second round of workgroups would start when the first is through) */
    const int occ = occupancy_of(c, (const void *)fn, nw * WAVE, lds);
    if (occ > 0)
      per_cu = std::min<uint64_t>(per_cu, (uint64_t)occ);
  }
  per_cu = std::max<uint64_t>(1, std::min<uint64_t>(per_cu, (uint64_t)c->blocks_per_cu));
  uint64_t grid = (uint64_t)c->cus * per_cu;
  grid = std::min<uint64_t>(grid, c->sliced ? (uint64_t)c->nchunks + ((uint64_t)c->nsmall + nw - 1) / nw
                                            : ((uint64_t)c->ntiles + WAVES_PER_BLOCK - 1) /
                                                  WAVES_PER_BLOCK);
  grid = std::max<uint64_t>(grid, 1);
  S.fn = fn;
  S.grid = (uint32_t)grid;
  S.nw = (uint32_t)nw;
  S.lds = lds;
  if (S.deferred) {
    S.rlds = (BLOCK_THREADS / WAVE) * sizeof(CandQueue) +
             (P.lds_matrix ? cells * sizeof(unsigned long long) : 0);
    /* 5 waves/SIMD fit its registers; a multiple of the segment count */
    uint32_t rgrid = (uint32_t)c->cus * (uint32_t)c->resolve_blocks_per_cu;
    S.rgrid = std::max<uint32_t>(1, rgrid / P.pos_segments) * P.pos_segments;
    S.rfn = select_resolve(!c->opt.ignore_genes);
  }
  if (S.redo_kind) {
    /* the redo pass (issue_step) */
    S.fn2 = select_rows_kernel(c->opt, nw, true, wide);
    if ((rc = raise_lds_limit(c, (const void *)S.fn2, lds))) return rc;
  }
  S.valid = true;
  return CMPR_OK;
}

/* what changes from launch to launch */
struct StepArgs {
  unsigned long long *d_out;
  bool needs_clear;          /* counter blocks and partial slots are not known to be zero */
  bool track_usage;          /* measure how full the positives buffer gets */
  bool with_redo;            /* variant 2: enqueue the redo pass */
};

/* The memsets, launches and the mid-step event of one step on `st`.  The counter block in
   use was chosen by the caller (use_counter_block). */
int issue_step(cmpr_context *c, const StepArgs &a, hipStream_t st, hipEvent_t ev_km)
{
  const StepPlan &S = c->plan;
  const size_t cells = S.cells;
  /* A step normally starts without a memset: its counter block was cleared by the
     previous launch's reduce kernel, and with the matrix privatised in LDS the reduce
     kernel WRITES the cells.  Otherwise (first launch, a launch that failed half-way,
     matrix too large for LDS, nothing to launch) everything is cleared here. */
  if (cells && !S.reduce_writes)
    HIP_TRY(c, hipMemsetAsync(a.d_out, 0, cells * sizeof(unsigned long long), st));
  if (cells && is_f64_score(c->opt))
    HIP_TRY(c, hipMemsetAsync(c->matrix_f64.p, 0, cells * sizeof(double), st));
  if (a.needs_clear) {
    HIP_TRY(c, hipMemsetAsync(c->pos_ctr.p, 0, 2 * S.ctr_blk * sizeof(unsigned long long), st));
    HIP_TRY(c, hipMemsetAsync(c->part.p, 0, (size_t)NPART * c->part_stride * sizeof(unsigned long long), st));
  }
  if (!S.will_launch) {                       /* (nobody will clear the other block) */
    HIP_TRY(c, hipMemsetAsync(c->ctr_other, 0, S.ctr_blk * sizeof(unsigned long long), st));
    if (ev_km)
      HIP_TRY(c, hipEventRecord(ev_km, st));
    c->launches = 0;
    return CMPR_OK;
  }
  ProbeParams P = S.P;
  P.matrix = a.d_out;
  P.tile_counter = c->d_tile_counter;
  P.deal_ctr = c->d_deal;
  P.stats = c->d_stats;
  P.pair_q = c->pair_q;
  P.pair_h = c->pair_h;
  P.pair_count = c->pair_count;
  P.pair_cap = c->pair_cap;
  if (S.deferred)
    P.pos_ctr = c->ctr_cur;
  if (c->rows && S.deferred)
    P.overflow = c->d_overflow;
  /* a launch of variant 2 without redo pass leaves word of an overflow behind */
  unsigned long long *sticky = (S.redo_kind && !a.with_redo) ? c->d_usage + 1 : nullptr;
  hipLaunchKernelGGL(S.fn, dim3(S.grid), dim3(S.nw * WAVE), S.lds, st, P);
  HIP_TRY(c, hipGetLastError());
  c->launches = 1;
  if (ev_km)
    HIP_TRY(c, hipEventRecord(ev_km, st));
  if (a.track_usage)
    HIP_TRY(c, hipMemsetAsync(c->d_usage, 0, sizeof(unsigned long long), st));
  auto reduce_partials = [&]() {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(P.part_stride), dim3(NPART), 0, st, P,
                       P.part_cells, S.reduce_writes ? 1u : 0u, c->ctr_other,
                       (uint32_t)S.ctr_blk, a.track_usage ? c->d_usage : nullptr, sticky);
  };
  const bool resolve_pass = S.deferred && !(c->debug & DBG_SKIP_RESOLVE);
  if (!resolve_pass) {
    reduce_partials();
    HIP_TRY(c, hipGetLastError());
    return CMPR_OK;
  }
  hipLaunchKernelGGL(S.rfn, dim3(S.rgrid), dim3(BLOCK_THREADS), S.rlds, st, P);
  reduce_partials();
  HIP_TRY(c, hipGetLastError());
  c->launches = 2;
  if (a.track_usage) {
    HIP_TRY(c, hipMemcpyAsync(c->h_usage, c->d_usage, sizeof(unsigned long long),
                              hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipEventRecord(c->ev_usage, st));
    c->usage_pending = true;
    c->usage_grid = S.grid;
    c->usage_nw = S.nw;
  }
  if (a.with_redo) {
    /* Redo pass: if the positives of the fast launch did not fit their buffer
       (flag set: resolve_kernel then did nothing), the same step with every
       positive resolved inline; otherwise its workgroups return at once.
       Capacity is therefore never a limit, and nothing here waits for the host. */
    ProbeParams P2 = P;
    P2.pos_buf = nullptr;
    P2.part = nullptr;                  /* (straight into matrix and stats2) */
    P2.redo = 1;
    P2.stats = c->d_stats2;
    P2.tile_counter = c->d_tile_counter2;
      P2.deal_ctr = c->d_deal + DEAL_WORDS / 2;
    hipLaunchKernelGGL(S.fn2, dim3(S.grid), dim3(S.nw * WAVE), S.lds, st, P2);
    HIP_TRY(c, hipGetLastError());
    c->launches = 3;
  }
  return CMPR_OK;
}

/* enqueue one step on `st`: `d_out` is the integer matrix to fill (ours or the caller's) */
int enqueue_overlap(cmpr_context *c, unsigned long long *d_out, hipStream_t st)
{
  int rc;
  if (!c->plan.valid && (rc = make_plan(c)))
    return rc;
  const StepPlan &S = c->plan;
  /* launches of one context are ordered one after the other, whatever streams the
     caller hands in: each uses state (counter blocks, positives buffer, partial
     slots) the previous one leaves behind */
  const bool same_stream = c->have_last_stream && c->last_stream == st;
  if (c->have_last_stream && c->last_stream != st && c->events_valid)
    HIP_TRY(c, hipStreamWaitEvent(st, c->ev_k1, 0));
  c->last_stream = st;
  c->have_last_stream = true;
  /* (event records are packets the stream waits for, ~4.5 us each: three per step, and
     only two when the previous launch of this context is still running on the same
     stream -- this one then starts where that one ends, and its end event IS this
     launch's start) */
  const bool back_to_back = c->events_valid && c->stop_is_k1 && same_stream &&
                            hipEventQuery(c->ev_k1) == hipErrorNotReady;
  (void)hipGetLastError();
  {
    hipEvent_t prev_end = c->ev_k1;
    const uint32_t slot = (uint32_t)(c->calls % cmpr_context::TIME_RING);
    c->ev_k0 = back_to_back ? prev_end : c->ring_k0[slot];
    c->ev_km = c->ring_km[slot];
    c->ev_k1 = c->ring_k1[slot];
    c->ring_start[slot] = c->ev_k0;
    c->calls++;
  }
  StepArgs a;
  a.d_out = d_out;
  a.needs_clear = !c->ctr_clean;
  int which = 0;
  if (c->ctr_clean)
    which = c->ctr_cur == c->pos_ctr.p ? 1 : 0;
  use_counter_block(c, which);
  c->ctr_clean = false;                     /* until this launch is through */

  /* The redo launch of variant 2 is needed until a finished launch on these sets, with
     this grid, has shown the margin.  The argument holds only while the chunks are dealt
     statically (no unstaged tiles claimed through a global counter: those move whole
     tiles, not part-filled blocks, between the segments from launch to launch). */
  const bool static_deal = c->nsmall == 0;
  if (c->force_no_redo && S.redo_kind) {      /* test only */
    c->never_overflows = true;
    c->safe_grid = S.grid;
    c->safe_nw = S.nw;
  }
  c->force_no_redo = false;
  if (c->never_overflows && (!S.redo_kind || c->safe_grid != S.grid || c->safe_nw != S.nw))
    c->never_overflows = false;
  if (c->usage_pending && (c->usage_grid != S.grid || c->usage_nw != S.nw))
    c->usage_pending = false;                 /* measured with another deal of the chunks */
  if (S.redo_kind && S.will_launch && c->usage_pending && hipEventQuery(c->ev_usage) == hipSuccess) {
    c->usage_pending = false;
    /* (a wave's blocks go round the segments -- kernels_sliced.h pos_segment_of --, so a segment hears
       from every wave of the grid) */
    const uint64_t waves_per_segment = (uint64_t)S.grid * (uint64_t)S.nw;
    /* (which chunks a workgroup gets may differ from launch to launch -- chunk_deal --: a wave's k-th
       block goes to segment (its workgroup + k) mod S whatever it works on, so a segment receives
       1/S of all blocks, give or take one block per wave; the number of blocks is that of the
       positives / 64, give or take two per wave) */
    const uint64_t margin = 2 * WAVE * (waves_per_segment + 1) + 4 * WAVE * ((uint64_t)S.grid * S.nw / S.P.pos_segments + 1);
    c->never_overflows = static_deal && *c->h_usage + margin <= c->pos_cap;
    c->safe_grid = S.grid;
    c->safe_nw = S.nw;
    /* The fullest segment did not fit: that launch was redone resolving inline (correct, and several times
       slower).  The buffer's size was a guess (4 positives per query); now that the number is known the
       buffer grows to it -- once per query set, behind the launches in flight (24.2M sequences against
       themselves, d = 1 -i: 3.1 x 10^8 positives where 9.8 x 10^7 were provided for; round 4). */
    if ((c->pos_grow > 0 || (c->pos_grow < 0 && c->pos_capacity == 0)) && *c->h_usage > c->pos_cap) {
      const uint64_t Sg = (uint64_t)c->pos_segments;
      const uint64_t want = *c->h_usage + *c->h_usage / 4 + margin;
      if (Sg * (want + WAVE) * sizeof(PosEntry) <= (96ull << 30)) {
        /* (launches of a context are ordered one behind the other, also across streams -- the wait for
           the previous one is already queued on `st`) */
        HIP_TRY(c, hipStreamSynchronize(st));
        if ((rc = dev_reserve(c, c->pos_buf, (size_t)(Sg * (want + WAVE)))))
          return rc;
        c->pos_cap = want;
        c->plan.P.pos_cap = want;
        c->plan.P.pos_buf = c->pos_buf.p;
      }
    }
  }
  a.track_usage = S.redo_kind && S.will_launch && static_deal && !c->never_overflows && !c->usage_pending;
  a.with_redo = S.redo_kind && !c->never_overflows;
  c->last_without_redo = S.redo_kind && S.will_launch && !a.with_redo;

  if (!back_to_back)
    HIP_TRY(c, hipEventRecord(c->ev_k0, st));
  c->ring_mid[(uint32_t)((c->calls - 1) % cmpr_context::TIME_RING)] = c->ev_km;
  if ((rc = issue_step(c, a, st, c->ev_km)))
    return rc;
  HIP_TRY(c, hipEventRecord(c->ev_k1, st));
  c->ctr_clean = true;
  return CMPR_OK;
}

/* Before the stream is synchronised: fetch the word a launch without redo pass leaves
   behind when its positives did not fit (reduce_partials_kernel). */
int fetch_overflow_word(cmpr_context *c, hipStream_t st)
{
  c->h_usage[1] = 0;
  if (c->last_without_redo)
    HIP_TRY(c, hipMemcpyAsync(c->h_usage + 1, c->d_usage + 1, sizeof(unsigned long long),
                              hipMemcpyDeviceToHost, st));
  return CMPR_OK;
}

/* After the synchronisation: true = the launch overflowed and had no redo pass; its
   result is invalid, the shortcut is withdrawn and the caller repeats the step (the
   repeat carries the redo pass). */
bool overflowed_without_redo(cmpr_context *c, hipStream_t st)
{
  if (!c->last_without_redo || c->h_usage[1] == 0)
    return false;
  /* (the word may also stem from an asynchronous launch before this one that nobody has
     asked about yet: cmpr_get_stats still has to report that one) */
  if (c->async_unchecked)
    c->async_overflowed = true;
  (void)hipMemsetAsync(c->d_usage + 1, 0, sizeof(unsigned long long), st);
  c->never_overflows = c->usage_pending = false;
  return true;
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
  if (!stream) {
    /* synchronous use: checked like the other synchronous entry points -- a launch that
       relied on the no-redo shortcut and overflowed is repeated with the redo pass */
    for (int attempt = 0;; attempt++) {
      rc = enqueue_overlap(c, (unsigned long long *)d_matrix, c->stream);
      if (rc)
        return rc;
      if ((rc = fetch_overflow_word(c, c->stream)))
        return rc;
      HIP_TRY(c, hipEventRecord(c->ev_stop, c->stream));
      c->stop_is_k1 = false;
      c->events_valid = true;
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      if (!overflowed_without_redo(c, c->stream))
        return CMPR_OK;
      if (attempt)
        return fail(c, CMPR_ESTATE, "positives buffer overflowed twice without redo pass");
    }
  }
  hipStream_t st = (hipStream_t)stream;
  rc = enqueue_overlap(c, (unsigned long long *)d_matrix, st);
  if (rc)
    return rc;
  c->stop_is_k1 = true;                     /* (nothing follows the kernels on this path) */
  c->events_valid = true;
  if (c->last_without_redo)
    c->async_unchecked = true;              /* until cmpr_get_stats has looked */
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
  const size_t cells = (size_t)c->R1 * c->R2;
  for (int attempt = 0;; attempt++) {
    rc = enqueue_overlap(c, c->matrix.p, c->stream);
    if (rc)
      return rc;
    if (cells)
      HIP_TRY(c, hipMemcpyAsync(out, c->matrix.p, cells * sizeof(uint64_t),
                                hipMemcpyDeviceToHost, c->stream));
    if ((rc = fetch_overflow_word(c, c->stream)))
      return rc;
    HIP_TRY(c, hipEventRecord(c->ev_stop, c->stream));
    c->stop_is_k1 = false;
    c->events_valid = true;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    /* a launch that relied on the no-redo shortcut and overflowed all the same: once
       more, now with the redo pass behind it */
    if (!overflowed_without_redo(c, c->stream))
      break;
    if (attempt)
      return fail(c, CMPR_ESTATE, "positives buffer overflowed twice without redo pass");
  }
  return CMPR_OK;
}

static int cmpr_overlap_matrix_f64_impl(cmpr_context *c, double *out)
{
  int rc = check_ready(c);
  if (rc)
    return rc;
  if (!out && (size_t)c->R1 * c->R2 > 0)
    return fail(c, CMPR_EINVAL, "matrix_out is NULL");
  const size_t cells = (size_t)c->R1 * c->R2;
  std::vector<unsigned long long> tmp(cells);
  for (int attempt = 0;; attempt++) {
    rc = enqueue_overlap(c, c->matrix.p, c->stream);
    if (rc)
      return rc;
    if (cells) {
      if (is_f64_score(c->opt))
        HIP_TRY(c, hipMemcpyAsync(out, c->matrix_f64.p, cells * sizeof(double),
                                  hipMemcpyDeviceToHost, c->stream));
      else
        HIP_TRY(c, hipMemcpyAsync(tmp.data(), c->matrix.p, cells * sizeof(uint64_t),
                                  hipMemcpyDeviceToHost, c->stream));
    }
    if ((rc = fetch_overflow_word(c, c->stream)))
      return rc;
    HIP_TRY(c, hipEventRecord(c->ev_stop, c->stream));
    c->stop_is_k1 = false;
    c->events_valid = true;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!overflowed_without_redo(c, c->stream))
      break;
    if (attempt)
      return fail(c, CMPR_ESTATE, "positives buffer overflowed twice without redo pass");
  }
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
  c->pair_q = dq.p;
  c->pair_h = dh.p;
  c->pair_count = dn.p;
  c->pair_cap = capacity;
  unsigned long long n = 0;
  for (int attempt = 0;; attempt++) {
    HIP_TRY(c, hipMemsetAsync(dn.p, 0, sizeof(unsigned long long), c->stream));
    rc = enqueue_overlap(c, c->matrix.p, c->stream);
    if (rc)
      return rc;
    HIP_TRY(c, hipMemcpyAsync(&n, dn.p, sizeof n, hipMemcpyDeviceToHost, c->stream));
    if ((rc = fetch_overflow_word(c, c->stream)))
      return rc;
    HIP_TRY(c, hipEventRecord(c->ev_stop, c->stream));
    c->stop_is_k1 = false;
    c->events_valid = true;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (!overflowed_without_redo(c, c->stream))
      break;
    if (attempt)
      return fail(c, CMPR_ESTATE, "positives buffer overflowed twice without redo pass");
  }
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
  /* (one less than the ring holds: the start of the oldest call may be the end event of the
     call before it, whose ring entry the newest call has just taken) */
  uint64_t n = std::min<uint64_t>(std::min<uint64_t>(max, c->calls), cmpr_context::TIME_RING - 1);
  for (uint64_t k = 0; k < n; k++) {
    const uint32_t slot = (uint32_t)((c->calls - n + k) % cmpr_context::TIME_RING);
    float a = 0, b = 0;
    HIP_TRY(c, hipEventElapsedTime(&a, c->ring_start[slot], c->ring_k1[slot]));
    HIP_TRY(c, hipEventElapsedTime(&b, c->ring_start[slot], c->ring_mid[slot]));
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
  hipEvent_t ev_end = c->stop_is_k1 ? c->ev_k1 : c->ev_stop;
  HIP_TRY(c, hipEventSynchronize(ev_end));
  unsigned long long st[STAT_COUNT], ovf = 0, sticky = 0;
  HIP_TRY(c, hipMemcpy(st, c->d_stats, sizeof st, hipMemcpyDeviceToHost));
  HIP_TRY(c, hipMemcpy(&ovf, c->d_overflow, sizeof ovf, hipMemcpyDeviceToHost));
  HIP_TRY(c, hipMemcpy(&sticky, c->d_usage + 1, sizeof sticky, hipMemcpyDeviceToHost));
  const bool async_overflowed = c->async_overflowed;
  c->async_unchecked = c->async_overflowed = false;
  if (sticky || async_overflowed) {
    /* One of the launches since the last check (cmpr_overlap_matrix_device does not
       synchronise, so it cannot look itself) ran without redo pass and overflowed its
       positives buffer: the matrix it left is incomplete.  The shortcut is withdrawn;
       the next launches carry the redo pass again. */
    if (sticky)
      HIP_TRY(c, hipMemset(c->d_usage + 1, 0, sizeof sticky));
    c->never_overflows = c->usage_pending = false;
    return fail(c, CMPR_ESTATE, "positives buffer overflowed in a launch without redo pass: result invalid, "
                                "repeat the call");
  }
  if (ovf)       /* the positives buffer overflowed: the redo pass did the step */
    HIP_TRY(c, hipMemcpy(st, c->d_stats2, sizeof st, hipMemcpyDeviceToHost));
  float k_ms = 0, t_ms = 0;
  HIP_TRY(c, hipEventElapsedTime(&k_ms, c->ev_k0, c->ev_k1));
  float p_ms = 0;
  HIP_TRY(c, hipEventElapsedTime(&p_ms, c->ev_k0, c->ev_km));
  HIP_TRY(c, hipEventElapsedTime(&t_ms, c->ev_k0, ev_end));
  memset(out, 0, sizeof *out);
  out->queries = c->n1;
  out->variants = st[STAT_VARIANTS];
  out->bloom_positive = st[STAT_BLOOM_POS];
  out->hash_equal = st[STAT_HASH_EQ];
  out->matches = st[STAT_MATCHES];
  out->filter_reads = st[STAT_READS];
  /* (a work shard does its share of every query's variants) */
  out->algorithmic_bytes = c->routed ? c->algorithmic_bytes : c->algorithmic_bytes / (uint64_t)c->work_shard_count;
  out->kernel_ms = k_ms;
  out->probe_ms = p_ms;
  out->total_ms = t_ms;
  out->kernel_launches = c->launches;
  return CMPR_OK;
}

static int set_reference_guarded(cmpr_context *c, const cmpr_set_view *s, uint32_t longest_query, bool on_device)
{
  /* the header promises CMPR_ENOMEM, not an exception across the C boundary */
  try {
    if (c)
      invalidate_plan(c);
    return cmpr_build_reference(c, s, longest_query, on_device);
  } catch (const std::bad_alloc &) {
    return fail(c, CMPR_ENOMEM, "out of host memory");
  }
}

extern "C" int cmpr_set_reference(cmpr_context *c, const cmpr_set_view *s, uint32_t longest_query)
{
  return set_reference_guarded(c, s, longest_query, false);
}

extern "C" int cmpr_set_reference_device(cmpr_context *c, const cmpr_set_view *d_set2, uint32_t longest_query)
{
  return set_reference_guarded(c, d_set2, longest_query, true);
}

/* query_layout.hip / ref_index.hip: one attribute query of a kernel of theirs (loads the code object) */
void cmpr_touch_layout_kernels();
void cmpr_touch_index_kernels();

extern "C" int cmpr_warm_up(const cmpr_options *o)
{
  if (!o)
    return CMPR_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return CMPR_EDEVICE;
  if (o->device >= ndev)
    return CMPR_EINVAL;
  if (o->device >= 0 && hipSetDevice(o->device) != hipSuccess)
    return CMPR_EDEVICE;
  (void)hipFree(nullptr);                              /* the device's context */
  /* the code objects a step of these options runs (each translation unit is one, loaded at first use) */
  hipFuncAttributes fa;
  (void)hipFuncGetAttributes(&fa, (const void *)reduce_partials_kernel);
  cmpr_touch_index_kernels();
  cmpr_touch_layout_kernels();
  const int A = o->alphabet_size, D = o->differences;
  const bool i = o->indels != 0, g = !o->ignore_genes;
  if ((A == 20 || A == 4) && D >= 0 && D <= 2 && (!i || D == 1)) {
    ProbeFn fns[3] = {select_resolve(g), nullptr, nullptr};
    if (A == 20 && D >= 1) {
      fns[1] = select_probe_v2_nw16(A, D, i, g);
      fns[2] = select_probe_v2_inline_nw16(A, D, i, g);
    } else if (A == 4 && D == 2) {
      fns[1] = select_probe_pairs2(g);
    } else if (D == 0) {
      /* d = 0 runs on variant 0, no filter: probe -> reduce, no resolve_kernel */
      fns[0] = nullptr;
      fns[1] = select_probe_v0(A, D, i, g);
    } else {
      fns[1] = select_probe_v1_nw8(A, D, i, g);
      fns[2] = select_probe_v1_nw4(A, D, i, g);
    }
    for (ProbeFn f : fns)
      if (f)
        (void)hipFuncGetAttributes(&fa, (const void *)f);
  }
  (void)hipGetLastError();
  return CMPR_OK;
}

/* ---- cmpr_warm_up_sized: the reservations (context.h WarmReservation) ---- */
namespace {
std::mutex g_warm_mu;
WarmReservation g_warm;
}

void *cmpr_take_reserved_host(size_t need, size_t *got_bytes)
{
  std::lock_guard<std::mutex> lock(g_warm_mu);
  if (!g_warm.host || g_warm.host_bytes < need)
    return nullptr;
  void *p = g_warm.host;
  *got_bytes = g_warm.host_bytes;
  g_warm.host = nullptr;
  g_warm.host_bytes = 0;
  return p;
}

char *cmpr_take_reserved_device(int device, size_t need, size_t *got_bytes)
{
  std::lock_guard<std::mutex> lock(g_warm_mu);
  if (!g_warm.dev || g_warm.device != device || g_warm.dev_bytes < need)
    return nullptr;
  char *p = g_warm.dev;
  *got_bytes = g_warm.dev_bytes;
  g_warm.dev = nullptr;
  g_warm.dev_bytes = 0;
  return p;
}

void cmpr_release_reservations()
{
  std::lock_guard<std::mutex> lock(g_warm_mu);
  if (g_warm.host)
    (void)hipHostFree(g_warm.host);
  if (g_warm.dev)
    (void)hipFree(g_warm.dev);
  g_warm = WarmReservation();
}

extern "C" int cmpr_warm_up_sized(const cmpr_options *o, uint64_t n_queries_hint, uint64_t n_refs_hint,
                                  uint64_t residue_bytes_hint)
{
  int rc = cmpr_warm_up(o);
  if (rc)
    return rc;
  (void)n_refs_hint;                      /* (the index build's temporaries come and go: nothing to keep for it) */
  int device = o->device;
  if (device < 0 && hipGetDevice(&device) != hipSuccess)
    return CMPR_EDEVICE;
  /* what the first cmpr_set_queries of a set of that size allocates: the pinned staging buffer of the narrowed
     upload (lengths, 16-bit ids, 32-bit counts: 12 bytes per query -- page-locking 100+ MB is what a cold layout
     call waited for) and arena A (the caller's arrays on the device + ~60 bytes of temporaries per query) */
  void *host = nullptr;
  size_t host_bytes = 0;
  if (n_queries_hint >= (1u << 20)) {
    const size_t need = (size_t)(n_queries_hint + 1) * 2 + (size_t)n_queries_hint * (2 + 2 + 2 + 4) + 64;
    host_bytes = need + need / 16;
    if (hipHostMalloc(&host, host_bytes, hipHostMallocDefault) != hipSuccess) {
      host = nullptr;
      host_bytes = 0;
    }
  }
  char *dev = nullptr;
  size_t dev_bytes = 0;
  if (n_queries_hint) {
    dev_bytes = (size_t)n_queries_hint * 96 + (size_t)residue_bytes_hint + (64u << 20);
    if (hipMalloc((void **)&dev, dev_bytes) != hipSuccess) {
      dev = nullptr;
      dev_bytes = 0;
    }
  }
  (void)hipGetLastError();
  std::lock_guard<std::mutex> lock(g_warm_mu);
  if (g_warm.host)
    (void)hipHostFree(g_warm.host);
  if (g_warm.dev)
    (void)hipFree(g_warm.dev);
  g_warm.host = host;
  g_warm.host_bytes = host_bytes;
  g_warm.dev = dev;
  g_warm.dev_bytes = dev_bytes;
  g_warm.device = device;
  return CMPR_OK;
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

static int set_queries_guarded(cmpr_context *c, const LayoutSource &src)
{
  /* the header promises CMPR_ENOMEM, not an exception across the C boundary */
  try {
    return cmpr_set_queries_impl(c, src);
  } catch (const std::bad_alloc &) {
    return fail(c, CMPR_ENOMEM, "out of host memory");
  }
}

extern "C" int cmpr_set_queries(cmpr_context *c, const cmpr_set_view *s)
{
  LayoutSource src;
  src.kind = LayoutSource::HOST;
  src.view = s;
  return set_queries_guarded(c, src);
}

extern "C" int cmpr_set_queries_device(cmpr_context *c, const cmpr_set_view *d_set1)
{
  LayoutSource src;
  src.kind = LayoutSource::DEVICE;
  src.view = d_set1;
  return set_queries_guarded(c, src);
}

extern "C" int cmpr_set_queries_routed(cmpr_context *c, const void *d_records, uint64_t n_records,
                                       uint32_t n_repertoires, uint64_t n_total, const double *rep_totals)
{
  LayoutSource src;
  src.kind = LayoutSource::RECORDS;
  src.d_records = d_records;
  src.nrec = n_records;
  src.n_rep = n_repertoires;
  src.n_total = n_total;
  src.totals = rep_totals;
  return set_queries_guarded(c, src);
}

extern "C" int cmpr_route_queries(cmpr_context *c, const cmpr_set_view *share, uint64_t first_index,
                                  uint32_t n_dest, uint64_t *counts_out, uint32_t *record_bytes_out,
                                  double *rep_totals_out)
{
  if (!c)
    return CMPR_EINVAL;
  try {
    if (!c->have_ref)
      return fail(c, CMPR_ESTATE, "cmpr_set_reference must be called first");
    if (!counts_out || !record_bytes_out)
      return fail(c, CMPR_EINVAL, "cmpr_route_queries: NULL output");
    if (n_dest == 0 || n_dest > 64 || (int64_t)n_dest != c->work_shard_count)
      return fail(c, CMPR_EINVAL, "cmpr_route_queries: n_dest must be the work_shard_count tunable (1..64)");
    std::string why;
    int rc = validate_view(c->opt, share, why);
    if (rc)
      return fail(c, rc, why);
    if (first_index + share->n > 0xffffffffull)
      return fail(c, CMPR_EUNSUPPORTED, "more than 2^32 sequences in the whole query set");
    if ((rc = retire_queries(c)))
      return rc;
    LayoutSource src;
    src.kind = LayoutSource::HOST;
    src.view = share;
    src.route = true;
    src.first_index = first_index;
    if ((rc = cmpr_layout_queries(c, src)))
      return rc;
    for (uint32_t d = 0; d < n_dest; d++)
      counts_out[d] = c->route.counts[d];
    *record_bytes_out = c->route.record_bytes;
    if (rep_totals_out)
      for (uint32_t r = 0; r < share->n_repertoires; r++)
        rep_totals_out[r] = c->route.rep_totals[r];
    return CMPR_OK;
  } catch (const std::bad_alloc &) {
    return fail(c, CMPR_ENOMEM, "out of host memory");
  }
}

extern "C" int cmpr_route_pack(cmpr_context *c, void *d_send, uint64_t capacity_bytes)
{
  if (!c)
    return CMPR_EINVAL;
  return cmpr_route_pack_impl(c, d_send, capacity_bytes);
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
