/*
 * compairr_hip.h -- C ABI of libcompairr_hip.so, the MI355X (gfx950) drop-in
 * for CompAIRR's repertoire-overlap hot path.
 *
 * The reference has no FFI layer; its narrowest seam for this path is the
 * launch of sim_thread inside overlap() (/root/reference/src/overlap.cc:926-936):
 * everything before it (AIRR-TSV parse, db accessors) is the input, everything
 * after it (matrix dump, overlap.cc:944-1039) the output.  The entry points
 * below are what a binding at that seam needs; each cites the reference
 * interface it replaces.  Plain pointers and sizes only, no C++/torch types.
 *
 * Threading: a context is used from one host thread at a time; distinct
 * contexts are independent (no process-global state, unlike the file-static
 * state at overlap.cc:24-53).
 *
 * Ownership: the caller owns every array it passes in and the output matrix;
 * arrays may be freed as soon as the call that received them returns.  The
 * library owns all device memory; nothing is retained after cmpr_destroy().
 *
 * Errors: every int-returning function returns 0 on success and a non-zero
 * CMPR_E* code otherwise; cmpr_last_error() then gives the message.  The
 * reference convention (fatal(): "\nError: %s\n" + exit(1), util.cc:84-88) is
 * applied by the host program, not here.
 */
#ifndef COMPAIRR_HIP_H
#define COMPAIRR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMPR_ABI_VERSION 5

enum {
  CMPR_OK          = 0,
  CMPR_EINVAL      = 1,   /* illegal option / argument combination          */
  CMPR_ENOMEM      = 2,   /* host or device allocation failed               */
  CMPR_EDEVICE     = 3,   /* HIP runtime error (message has the HIP string) */
  CMPR_EUNSUPPORTED= 4,   /* legal for the reference, not on this path      */
  CMPR_ESTATE      = 5    /* calls out of order                             */
};

/* -s / --score, same numbering as the reference enum (compairr.h:125-135) */
enum {
  CMPR_SCORE_PRODUCT = 0,
  CMPR_SCORE_RATIO   = 1,
  CMPR_SCORE_MIN     = 2,
  CMPR_SCORE_MAX     = 3,
  CMPR_SCORE_MEAN    = 4,
  CMPR_SCORE_MH      = 5,
  CMPR_SCORE_JACCARD = 6
};

/*
 * The process-global opt_* flags the loop reads (compairr.h:139-162), as a POD.
 */
typedef struct cmpr_options {
  int32_t  differences;     /* opt_differences, 0..2 on this path             */
  int32_t  indels;          /* opt_indels (requires differences == 1)         */
  int32_t  ignore_genes;    /* opt_ignore_genes                               */
  int32_t  ignore_counts;   /* opt_ignore_counts                              */
  int32_t  score;           /* opt_score_int, CMPR_SCORE_*                    */
  int32_t  alphabet_size;   /* alphabet_size: 20 (aa) or 4 (-n)               */
  uint32_t n_v_genes;       /* db_get_v_gene_count() (db.cc:1018)             */
  uint32_t n_j_genes;       /* db_get_j_gene_count() (db.cc:1023)             */
  int32_t  device;          /* HIP device ordinal; -1 = current device        */
  int32_t  existence;       /* opt_existence (-x): matrix rows are the set-1
                               SEQUENCES, in input order (overlap.cc:226)       */
  int32_t  reserved[6];     /* must be zero                                   */
} cmpr_options;

/*
 * What the loop reads of a repertoire set through db_getsequence /
 * db_getsequencelen / db_get_v_gene / db_get_j_gene / db_get_count /
 * db_get_repertoire_id_no (db.cc:964-997), as structure-of-arrays in HOST
 * memory.  Residues are the reference's codes (map_aa / map_nt, db.cc:33-71),
 * not ASCII.
 */
typedef struct cmpr_set_view {
  uint64_t        n;              /* number of sequences                      */
  const uint8_t  *residues;       /* offsets[n] residue codes, concatenated   */
  const uint64_t *offsets;        /* n + 1 entries, offsets[0] == 0           */
  const uint32_t *v_gene;         /* n; may be NULL when ignore_genes         */
  const uint32_t *j_gene;         /* n; may be NULL when ignore_genes         */
  const uint32_t *repertoire;     /* n; values < n_repertoires                */
  const uint64_t *count;          /* n; >= 1; may be NULL when ignore_counts  */
  uint32_t        n_repertoires;
  uint32_t        reserved;
} cmpr_set_view;

/* Work and timing of the last cmpr_overlap_* call. */
typedef struct cmpr_stats {
  uint64_t queries;            /* set-1 sequences processed                   */
  uint64_t variants;           /* variant hashes held against the Bloom filter
                                  (executed tests, counted by the kernel)      */
  uint64_t bloom_positive;     /* probes that passed the Bloom filter         */
  uint64_t hash_equal;         /* hash slots equal to a variant hash          */
  uint64_t matches;            /* verified (query, hit) pairs                 */
  uint64_t algorithmic_bytes;  /* sum over queries of (L + 20) + 8 * V(L)     */
  double   kernel_ms;          /* HIP-event time of probe + resolve kernels   */
  double   total_ms;           /* first launch -> matrix ready on the stream  */
  uint32_t kernel_launches;
  uint32_t reserved;
  double   probe_ms;           /* HIP-event time of the probe kernel alone    */
  uint64_t filter_reads;       /* filter words read for those tests: one per
                                  variant (kernel variants 0, 1), one per ROW of
                                  up to alphabet_size variants (variant 2)      */
} cmpr_stats;

typedef struct cmpr_context cmpr_context;

/* ABI version of the loaded library (== CMPR_ABI_VERSION it was built with). */
int cmpr_abi_version(void);

/*
 * Create a context on one device.  Replaces the option globals + the
 * zobrist_init() call (overlap.cc:840, zobrist.cc:28-67).
 */
int cmpr_create(const cmpr_options *options, cmpr_context **out);

/* Frees device memory and the context (bloom_exit/hash_exit/zobrist_exit,
   overlap.cc:1044-1049).  NULL is a no-op. */
void cmpr_destroy(cmpr_context *ctx);

/* Message of the last failed call on this context ("" if none).  With
   ctx == NULL: message of the last failed cmpr_create() on this thread. */
const char *cmpr_last_error(const cmpr_context *ctx);

/*
 * Upload set 2 and build its index on the device: per-sequence Zobrist hash
 * (db_hash, db.cc:903-916), hash table + Bloom filter inserts (hash_init,
 * bloom_init, hash_insert loop, overlap.cc:861-873).  Every entry is inserted,
 * duplicates included.  `longest_query` is the longest set-1 sequence that
 * will be submitted (the Zobrist table needs max(longest1, longest2) + 3
 * positions, overlap.cc:840); pass 0 to size it from this set alone, in which
 * case later longer queries make cmpr_set_queries() fail with CMPR_EINVAL.
 *
 * Footprint: the record table holds one 64-byte slot per BUCKET, buckets = 2^table_log2_delta x the
 * smallest power of two >= n / 0.7 (hashtable.cc:24, 36-38) -- 183 to 366 bytes per reference sequence at
 * the default delta of 1 (10M: 2 GiB; 100M: 32 GiB) --, the row filter 2 bytes per residue position + 2 per
 * sequence (twice that with -i), the set itself ~45 bytes per sequence.  At most 2^30 buckets: sets beyond
 * ~375M sequences (187M at delta 1 ... 750M at delta 0) return CMPR_EUNSUPPORTED; "table_log2_delta" = 0
 * halves the table.
 */
int cmpr_set_reference(cmpr_context *ctx, const cmpr_set_view *set2,
                       uint32_t longest_query);

/*
 * Upload set 1 (the queries) and lay it out for the kernel.  After this call
 * the queries are resident in HBM; cmpr_overlap_* may be called repeatedly.
 * Passing the same view as set 2 gives the reference's one-file mode
 * (overlap.cc:799-825).  The context keeps its device allocations from call to
 * call (a second set of similar size costs no allocation).  With the work-shard
 * tunables set, only what this context works on is laid out (the queries are still
 * uploaded and keyed in full; cmpr_route_queries below divides that too).
 */
int cmpr_set_queries(cmpr_context *ctx, const cmpr_set_view *set1);

/*
 * The same two calls for sets that already live in HBM (a parser that writes to the
 * device, a previous stage of a pipeline, another library): every pointer of the view is
 * a DEVICE pointer on the context's device, same types and meaning as above.  Nothing
 * crosses PCIe but two offsets (offsets[0], offsets[n]) and the layout's sizes; the set is
 * validated on the device like a host set.  The reference set is copied (the library keeps
 * its own arrays); the query arrays are read where they lie during the call and may be
 * freed or overwritten once it has returned.  (The reference reads both sets where db.cc
 * left them, db.cc:964-997 -- this is that, for a caller whose "where" is the GPU.)
 */
int cmpr_set_reference_device(cmpr_context *ctx, const cmpr_set_view *d_set2,
                              uint32_t longest_query);
int cmpr_set_queries_device(cmpr_context *ctx, const cmpr_set_view *d_set1);

/*
 * Multi-GPU layout of ONE query set over N contexts (one per GPU, tunables
 * work_shard_count = N, work_shard_index = 0..N-1, the same reference set in each).
 * The reference hands chunks of 1000 queries to its threads (overlap.cc:421-433) and
 * sums their private matrices (overlap.cc:510-527); here the contexts divide the WORK of a
 * step by filter slice, and the queries go where their work is:
 *
 *   1. every context is given ANY share of the set (e.g. the i-th N-th of it):
 *      cmpr_route_queries() uploads and keys that share and says how many records go to
 *      each of the n_dest contexts (counts_out[n_dest]; a query goes to the context that
 *      works on its slice and to those that work on one of its class-position items --
 *      1 to 2 destinations on CDR3 amino acids), the size of a record (*record_bytes_out,
 *      a multiple of 16) and, if rep_totals_out is not NULL, the duplicate_count total
 *      per repertoire of the share (n_repertoires doubles; summed over the shares they
 *      bound the cells of the final matrix);
 *   2. cmpr_route_pack() writes those records into a DEVICE buffer of the caller, grouped
 *      by destination (destination d's run starts at record counts[0] + .. + counts[d-1]);
 *   3. the caller moves the runs to their destinations -- one all-to-all over xGMI (RCCL)
 *      or, inside one process, device-to-device copies;
 *   4. every context receives with cmpr_set_queries_routed(): n_records records in device
 *      memory (in any order), the number of repertoires and of sequences of the WHOLE set,
 *      and optionally the summed totals of step 1 (NULL: this context's own share bounds
 *      its own matrix only).
 * After step 4 the context holds exactly what it works on; cmpr_overlap_* gives its part of
 * the matrix, the parts add up to the whole (one sum-reduce), pair lists and -x rows carry
 * the sequence numbers of the whole set (first_index + position in the share).
 * Uploads and layout work divide by N; no context ever sees the whole set.
 */
int cmpr_route_queries(cmpr_context *ctx, const cmpr_set_view *share, uint64_t first_index,
                       uint32_t n_dest, uint64_t *counts_out, uint32_t *record_bytes_out,
                       double *rep_totals_out);
int cmpr_route_pack(cmpr_context *ctx, void *d_send, uint64_t capacity_bytes);
int cmpr_set_queries_routed(cmpr_context *ctx, const void *d_records, uint64_t n_records,
                            uint32_t n_repertoires, uint64_t n_total, const double *rep_totals);

/*
 * The per-query loop (sim_thread / process_variants / find_variant_matches,
 * overlap.cc:376-538, 253-284, 168-251).  Writes the R1 x R2 matrix,
 * row = set-1 repertoire number (with options.existence: R1 = number of set-1
 * sequences, row = sequence number), column = set-2 repertoire number
 * (overlap.cc:222-226), as exact integer sums:
 *   product, MH : sum of count1 * count2
 *   min, Jaccard: sum of min;  max: sum of max;  -f: number of pairs
 *   mean        : sum of (count1 + count2), i.e. TWICE the reference's cell
 * The matrix is overwritten, not accumulated into.  Host-memory output.
 * CMPR_SCORE_RATIO is not an integer sum: use cmpr_overlap_matrix_f64().
 */
int cmpr_overlap_matrix(cmpr_context *ctx, uint64_t *matrix_out);

/* Same loop, double-precision cells; the only form that supports ratio
   (order-dependent rounding, as in the threaded reference, overlap.cc:510-527). */
int cmpr_overlap_matrix_f64(cmpr_context *ctx, double *matrix_out);

/*
 * Same loop, result left in DEVICE memory: `d_matrix` points to
 * R1 * R2 uint64 on the context's device (e.g. a buffer a collective library
 * will reduce across GPUs).  The kernels are enqueued on `stream`
 * (a hipStream_t, NULL = the context's own stream) and the call returns
 * without synchronising when `stream` is not NULL.
 *
 * Launches of one context are ordered one after the other whatever streams they are
 * given (each uses state the previous one leaves behind; the library inserts the
 * event wait when the stream changes).
 *
 * Because this entry point does not wait, it cannot look at the one condition that
 * can invalidate a launch after the fact: amino acids with d >= 1 drop their redo
 * pass once a finished launch on the same sets has shown that the positives buffer
 * has room to spare; should a later launch overflow all the same, its matrix is
 * incomplete.  The device records that; the next cmpr_get_stats() then fails with
 * CMPR_ESTATE ("... overflowed in a launch without redo pass"), withdraws the
 * shortcut, and later launches carry the redo pass again.  Callers of this entry
 * point must therefore call cmpr_get_stats() before trusting a series of launches.
 * The synchronous entry points (cmpr_overlap_matrix, _f64, _pairs) check by
 * themselves and repeat the step with the redo pass: they never return such a result
 * (and a synchronous call that comes between such a launch and the question does not
 * take the answer away: cmpr_get_stats() still fails once).
 */
int cmpr_overlap_matrix_device(cmpr_context *ctx, void *d_matrix, void *stream);

/*
 * The matching (query, hit) pairs themselves -- what the reference appends to
 * its pairs list in find_variant_matches (overlap.cc:232-245) and prints with
 * -p/--pairs (overlap.cc:455-507).  Runs the same loop; up to `capacity` pairs are
 * written (query_out[k] = index into the set given to cmpr_set_queries,
 * hit_out[k] = index into the reference set; order unspecified, as in the
 * reference, README.md:163), *count_out = number of pairs found.  When
 * *count_out > capacity the arrays hold an arbitrary subset: call again with a
 * larger capacity (cmpr_stats.matches of a previous cmpr_overlap_matrix() call is
 * the exact number).  capacity == 0 with NULL arrays only counts.
 */
int cmpr_overlap_pairs(cmpr_context *ctx, uint64_t capacity, uint32_t *query_out,
                       uint32_t *hit_out, uint64_t *count_out);

/*
 * Exact duplicates inside one set: the number the reference reports as
 * "Warning: N duplicates detected in repertoire set K" -- sequences that repeat
 * an earlier one of the same repertoire with the same V, J (unless
 * ignore_genes) and residues (hash_insert's return value summed,
 * overlap.cc:76-115, 579-605, 865-873).  set == NULL: the resident reference
 * set (needs cmpr_set_reference); otherwise `set` is uploaded and indexed in a
 * temporary table that is freed before returning.  Does not disturb the
 * resident sets.
 */
int cmpr_count_duplicates(cmpr_context *ctx, const cmpr_set_view *set, uint64_t *out);

/*
 * (ABI v4)  What a process pays once before its first launch, asked for early: the HIP
 * runtime, the device's context, the code objects of the kernels the given options will run.
 * The reference has no counterpart -- its threads start in microseconds (overlap.cc:926-936);
 * a GPU process needs ~0.4 s here, which a caller can spend while it still reads its input
 * (compairr_amd/host/overlap_host.cc does, on a thread of its own).  Only `differences`,
 * `indels`, `alphabet_size`, `ignore_genes` and `device` of `options` are looked at.  Creates
 * nothing the caller has to free.  Thread-safe against every other entry point.
 */
int cmpr_warm_up(const cmpr_options *options);

/*
 * (ABI v5)  cmpr_warm_up() for a caller that knows roughly how large its sets will be -- the host program
 * knows both file sizes before it parses a line (compairr_amd/host/overlap_host.cc) --: additionally
 * reserves what the first cmpr_set_queries() of a set of `n_queries_hint` sequences with
 * `residue_bytes_hint` residues would otherwise allocate while the caller waits (the page-locked staging
 * buffer of its upload, 12 bytes per query, and the device arena of the layout's temporaries).  The first
 * context that needs them takes them over; what nobody took is released by the first cmpr_destroy().
 * `n_refs_hint` is accepted for symmetry (the index build keeps nothing that could be reserved).  Hints
 * are hints: too small, and the call allocates as before; too large, and memory is held until taken or
 * released.  The reference has no counterpart (overlap.cc:840-887 allocates in microseconds).
 */
int cmpr_warm_up_sized(const cmpr_options *options, uint64_t n_queries_hint, uint64_t n_refs_hint,
                       uint64_t residue_bytes_hint);

/* Statistics of the last overlap call (synchronises the context's events). */
int cmpr_get_stats(cmpr_context *ctx, cmpr_stats *out);

/*
 * HIP-event kernel times of the last `max` (at most 63) cmpr_overlap_* calls,
 * oldest first: kernel_ms[k] = probe + resolve kernels, probe_ms[k] = the probe
 * kernel alone (either may be NULL).  Lets a caller queue many launches on a
 * stream without synchronising after each (the reference has one timed region,
 * "Analysing:", overlap.cc:906-938; this is its per-launch counterpart).
 */
int cmpr_get_kernel_times(cmpr_context *ctx, uint32_t max, double *kernel_ms,
                          double *probe_ms, uint32_t *count_out);

/* Sizes, for callers that allocate the matrix. */
uint32_t cmpr_rows(const cmpr_context *ctx);      /* R1, after set_queries   */
uint32_t cmpr_cols(const cmpr_context *ctx);      /* R2, after set_reference */

/* Tuning knobs (unknown names -> CMPR_EINVAL).  Results never depend on them.
     "variant"               0: one Bloom filter probed in HBM -- at d = 0 no filter at
                             all, the query looked up where its bucket lies (default
                             for d = 0); 1: class-keyed
                             32 KiB slices staged in LDS, one filter word per
                             variant (default for nucleotides at d >= 1); 2: the same
                             slices over the row filter, one filter word per
                             position (default for amino acids at d >= 1); -1: default
     "blocks_per_cu"         resident workgroups per CU the grid is sized for
     "bloom_bits_log2_delta" filter bytes = hash-table slots << delta
                             (default 0 for variant 0, +2 for variant 1)
     "class_residues"        -1 (default: from the data) or 0..4 amino acids / 0..8 nucleotides (four
                             amino-acid residues: variant 2 at d = 1, on kernel instantiations of
                             their own; anywhere else the value is clamped to three)
     "slice_words_log2"      log2 of the (largest) slice in filter words: 64-bit
                             words, default 12 (variant 1); 256-bit words,
                             -1 = default: sized to the LDS (<= 640 words), or
                             a power of two 1..13 (variant 2)
     "chunk_tiles"           tiles per workgroup work item (default 8 x waves)
     "waves_per_block"       4, 8 or 16 waves per workgroup (variants 1, 2;
                             default 8 / 16)
     "small_slice_tiles"     slices with at most this many tiles are probed where
                             they lie instead of being staged (default 0)
     "work_shard_count",     this context does the work filed under every
     "work_shard_index"      count-th share of the filter slices, share `index`
                             (default 1, 0: everything).  N contexts -- e.g. one
                             per GPU -- with the same sets and index 0..N-1
                             produce matrices, pair lists and counters that add
                             up to the unsharded ones (variants 1, 2).
     "sub2_items"            nucleotides, d = 2, variant 1: the double substitutions
                             that put a new residue on a class position are grouped
                             by the slice they land in and probed there (1, and
                             the default -1), or probed where the filter lies (0)
     "d2_pairs"              nucleotides, d = 2: -1 (default) / 1: pair rows probed by a workgroup per
                             tile (kernels_pairs2.h; sequences of at most 96 residues), 0: off
     "d2_buffers"            that kernel's slice buffers: 1 (default; slices twice the size: fuller
                             tiles) or 2 (the next slice is copied while this one is worked on)
     "chunk_deal"            variant 2: 1 (default): beyond a workgroup's first four, chunks are handed out
                             by counters in list order (heaviest first) -- on skewed data (the cdr3 law,
                             d = 1 -i) the probe kernel takes 1.5 ms where a static deal takes 2.5; 0: static
     "table_log2_delta"      buckets of the record table = 2^delta x the 70 % rule of hashtable.cc:24 (default 1:
                             at most 0.35 full -- one memory line per looked-up sequence); 0..3
     "bucket_bitmap"         resolve_kernel asks a bitmap (one bit per bucket: it holds a record) before it
                             reads a slot of the record table: -1 (default) where most Bloom positives are
                             false (d = 2), 0 never, 1 always
     "slice_pages"           variant 2, d = 1: a slice of the row filter that holds more than "page_budget"
                             entries is spread over up to 2^this pages by hash bits, its tiles worked on once
                             per page (-1 = default: 3; 0: no pages); "page_budget": entries, 0 = default
                             (20 per 32-byte word)
     "fill_slices"           variant 2, d = 2 on single rows: 1 (default) the slices take all the words their
                             LDS buffer holds (a sparser filter for the 38 000 tests of a query), 0 as many
                             as the entries ask for
     "row_filter_x16"        variant 2: bytes of filter per entry in sixteenths (default 32 = 2 bytes)
     "direct_slices_log2"    d = 0 (variant 0, the default there: no filter, every query looked up where its
                             bucket lies): the layout groups the queries by length and by 2^this pseudo-slices
                             -- bits of the hash --; -1 (default) one per 32 768 queries
     "pos_grow"              the positives buffer grows to what a launch showed when it overflowed:
                             -1 (default) when its size was automatic, 1 also from a given
                             "pos_capacity", 0 never
     "deferred_resolve", "resolve_blocks_per_cu", "pos_segments", "pos_capacity",
     "heavy_threshold", "class_anchor", "class_rows_unstaged",
     "host_threads"          see compairr_amd/csrc/compairr_hip.hip
     "assume_never_overflows" TEST ONLY: the next launch runs without redo pass as if
                             the margin had been shown
   "variant", "bloom_bits_log2_delta", "class_residues", "slice_words_log2", "d2_pairs", "d2_buffers",
   "table_log2_delta", "slice_pages", "page_budget", "fill_slices" and "row_filter_x16" must be set before
   cmpr_set_reference(); "chunk_tiles", "waves_per_block",
   "small_slice_tiles", "direct_slices_log2" and the work shard before cmpr_set_queries().  ("debug" exists
   only in a -DCMPR_ABLATION build of the library.) */
int cmpr_set_tunable(cmpr_context *ctx, const char *name, int64_t value);

/* Current value of a tunable (for the data-dependent ones, the value in effect
   after cmpr_set_reference / cmpr_set_queries), plus the read-only names
   "slices", "tiles", "chunks", "query_slots" (tiles x 64, padding included),
   "never_overflows" (1: the redo pass is currently dropped) and, of the last
   cmpr_set_queries* / cmpr_route_queries in microseconds, "layout_total_us",
   "layout_upload_us" (host time inside the copy calls) and "layout_tail_us" (from the
   last copy to the end: the device work the upload did not hide). */
int cmpr_get_tunable(cmpr_context *ctx, const char *name, int64_t *value);

#ifdef __cplusplus
}
#endif
#endif
