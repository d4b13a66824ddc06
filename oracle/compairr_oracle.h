/*
 * compairr_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the repertoire-overlap hot path of CompAIRR
 * 1.13.0 (reference: /root/reference/src/overlap.cc:253-284,168-251,376-538,
 * variants.cc:260-428, zobrist.cc:28-136, bloompat.{h,cc}, hashtable.{h,cc}).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (libcompairr_hip.so, the compairr CLI) never
 * links or calls it.
 *
 * Parity pin: checked byte-for-byte against the reference program compiled
 * from /root/reference (oracle/_ref/compairr) on the reference's own
 * test/ *.tsv golden files and on the generated vectors under tests/golden/
 * (tests/test_oracle_golden.py).
 */
#ifndef COMPAIRR_ORACLE_H
#define COMPAIRR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* same numbering as the reference's score enum (compairr.h:125-135) */
enum {
  ORACLE_SCORE_PRODUCT = 0,
  ORACLE_SCORE_RATIO   = 1,
  ORACLE_SCORE_MIN     = 2,
  ORACLE_SCORE_MAX     = 3,
  ORACLE_SCORE_MEAN    = 4,
  ORACLE_SCORE_MH      = 5,
  ORACLE_SCORE_JACCARD = 6
};

/* What the per-query loop reads through db_get* (db.cc:964-997), as SoA. */
typedef struct oracle_set {
  uint64_t        n;             /* sequences                                */
  const uint8_t  *residues;      /* residue codes 0..A-1, concatenated       */
  const uint64_t *offsets;       /* n+1 entries; seq i = [offsets[i], offsets[i+1]) */
  const uint32_t *v_gene;        /* n, global V numbering                    */
  const uint32_t *j_gene;        /* n, global J numbering                    */
  const uint32_t *repertoire;    /* n, per-set first-appearance numbering    */
  const uint64_t *count;         /* n, duplicate_count (>= 1)                */
  uint32_t        n_repertoires;
} oracle_set;

typedef struct oracle_opts {
  int32_t  differences;    /* -d                                            */
  int32_t  indels;         /* -i (only with d == 1)                         */
  int32_t  ignore_genes;   /* -g                                            */
  int32_t  ignore_counts;  /* -f                                            */
  int32_t  score;          /* -s, ORACLE_SCORE_*                            */
  int32_t  alphabet_size;  /* 20 (aa) or 4 (nt)                             */
  int32_t  threads;        /* -t                                            */
  uint32_t n_v_genes;
  uint32_t n_j_genes;
  int32_t  existence;      /* -x: rows are set-1 sequences (overlap.cc:226)  */
} oracle_opts;

typedef struct oracle_stats {
  uint64_t variants;         /* variants generated == Bloom probes           */
  uint64_t bloom_positive;   /* probes that passed the Bloom filter          */
  uint64_t slots_visited;    /* occupied hash slots inspected                */
  uint64_t hash_equal;       /* slots whose 64-bit hash equalled the variant */
  uint64_t matches;          /* verified (seed, hit) pairs                   */
  uint64_t dup_set1;         /* check_duplicates(set 1), two-set mode only   */
  uint64_t dup_set2;         /* duplicates seen while indexing set 2         */
  double   seconds_index;    /* hashing + hash_insert of set 2               */
  double   seconds_analysis; /* the "Analysing:" region                      */
} oracle_stats;

/*
 * matrix: caller-allocated double[rows * set2->n_repertoires], rows =
 * set1->n_repertoires (or set1->n with -x); row = set-1 repertoire number (or
 * sequence number with -x), column = set-2 repertoire number; zeroed by
 * the callee (overlap.cc:882-887).  set2 may alias set1 (one-file mode).
 * Returns 0, or -1 on an illegal option combination / allocation failure.
 */
int oracle_overlap(const oracle_opts *opts,
                   const oracle_set  *set1,
                   const oracle_set  *set2,
                   double            *matrix,
                   oracle_stats      *stats);

/*
 * The (seed, hit) pairs the loop finds -- what the reference writes with
 * -p/--pairs (overlap.cc:232-245, 455-507).  Single-threaded.  Up to `capacity`
 * pairs are stored (seed = index into set1, hit = index into set2, in the order
 * found); *count is the total number found.  Returns 0 or -1.
 */
int oracle_pairs(const oracle_opts *opts, const oracle_set *set1, const oracle_set *set2,
                 uint64_t capacity, uint32_t *seed_out, uint32_t *hit_out, uint64_t *count);

/*
 * Independent O(N*M) evaluation of the pair definition (SURVEY.md section 8a,
 * "Semantic summary"); shares no code with the hashing path above.  Small
 * inputs only.
 */
int oracle_bruteforce(const oracle_opts *opts,
                      const oracle_set  *set1,
                      const oracle_set  *set2,
                      double            *matrix);

/* The value printed for a cell (overlap.cc:540-577): MH / Jaccard / raw sum. */
double oracle_cell_value(const oracle_opts *opts, double cell,
                         double count1, double sq_count1,
                         double count2, double sq_count2);

#ifdef __cplusplus
}
#endif
#endif
