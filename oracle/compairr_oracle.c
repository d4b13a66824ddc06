/*
 * compairr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (own code, plain C11 + pthreads) of the CompAIRR 1.13.0
 * repertoire-overlap hot path.  Every function cites the reference lines it
 * follows.  See compairr_oracle.h for who may use this file and how its parity
 * is pinned.
 */
#define _GNU_SOURCE
#include "compairr_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ */
/* constants of the reference                                          */
/* ------------------------------------------------------------------ */

#define EXTRA_POSITIONS 3      /* MAX_INSERTS, compairr.h:111            */
#define MAX_HASHED_D    2      /* MAXDIFF_HASH, compairr.h:113           */
#define FILL_PERCENT    70     /* HASHFILLPCT, hashtable.cc:24           */
#define PAT_BITS        10     /* BLOOM_PATTERN_SHIFT, bloompat.h:22     */
#define PAT_COUNT       (1u << PAT_BITS)
#define PAT_K           8      /* bits per pattern, bloompat.cc:38       */
#define QUERY_CHUNK     1000   /* CHUNK, overlap.cc:60                   */

static double now_seconds(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------ */
/* Zobrist keys (zobrist.cc:28-67, zobrist.h:24-27)                    */
/* ------------------------------------------------------------------ */

typedef struct {
  uint64_t *pos;     /* pos[A * p + residue]                            */
  uint64_t *vkey;    /* per V gene                                      */
  uint64_t *jkey;    /* per J gene                                      */
  int       A;
  int       genes;   /* 0 when -g                                       */
} zob_t;

/* one key = four 31-bit glibc random() draws folded by <<16 ^ (zobrist.cc:52-63) */
static uint64_t draw_key(void)
{
  uint64_t z = (uint64_t)random();
  for (int k = 0; k < 3; k++)
    z = (z << 16) ^ (uint64_t)random();
  return z;
}

static int zob_init(zob_t *z, int A, unsigned positions,
                    uint32_t n_v, uint32_t n_j, int use_genes)
{
  uint64_t total = (uint64_t)A * positions + n_v + n_j;
  z->pos = (uint64_t *)malloc((total ? total : 1) * sizeof(uint64_t));
  if (!z->pos)
    return -1;
  for (uint64_t i = 0; i < total; i++)
    z->pos[i] = draw_key();
  z->vkey = z->pos + (uint64_t)A * positions;
  z->jkey = z->vkey + n_v;
  z->A = A;
  z->genes = use_genes;
  return 0;
}

static inline uint64_t zk(const zob_t *z, unsigned p, unsigned r)
{
  return z->pos[(unsigned)z->A * p + r];
}

static inline uint64_t zob_gene_part(const zob_t *z, uint32_t v, uint32_t j)
{
  return z->genes ? (z->vkey[v] ^ z->jkey[j]) : 0;   /* zobrist.cc:83-84 */
}

/* zobrist_hash (zobrist.cc:74-88); `shift` = +1 reproduces
   zobrist_hash_insert_first (:122-136), skip = 1 / shift = -1 reproduces
   zobrist_hash_delete_first (:90-104). */
static uint64_t zob_hash_shifted(const zob_t *z, const uint8_t *s, unsigned len,
                                 uint32_t v, uint32_t j, unsigned skip, int shift)
{
  uint64_t h = zob_gene_part(z, v, j);
  for (unsigned p = skip; p < len; p++)
    h ^= zk(z, (unsigned)((int)p + shift), s[p]);
  return h;
}

/* ------------------------------------------------------------------ */
/* Blocked Bloom filter, inverted polarity (bloompat.cc, bloompat.h)   */
/* ------------------------------------------------------------------ */

typedef struct {
  uint64_t *words;
  uint64_t  mask;                 /* word index mask                    */
  uint64_t  pat[PAT_COUNT];
} bloom_t;

static int bloom_make(bloom_t *b, uint64_t bytes)
{
  if (bytes < 8)
    bytes = 8;                                    /* bloompat.cc:63     */
  b->words = (uint64_t *)malloc(bytes);
  if (!b->words)
    return -1;
  memset(b->words, 0xff, bytes);                  /* bloom_zap, :54-57  */
  b->mask = (bytes >> 3) - 1;
  for (unsigned i = 0; i < PAT_COUNT; i++) {      /* :36-52             */
    uint64_t p = 0;
    for (int k = 0; k < PAT_K; k++) {
      uint64_t bit;
      do
        bit = 1ULL << (random() & 63);
      while (p & bit);
      p |= bit;
    }
    b->pat[i] = p;
  }
  return 0;
}

static inline uint64_t *bloom_word(bloom_t *b, uint64_t h)
{
  return b->words + ((h >> PAT_BITS) & b->mask);   /* bloompat.h:40-43   */
}

static inline void bloom_insert(bloom_t *b, uint64_t h)
{
  *bloom_word(b, h) &= ~b->pat[h & (PAT_COUNT - 1)];   /* :50-53         */
}

static inline int bloom_maybe(bloom_t *b, uint64_t h)
{
  return !(*bloom_word(b, h) & b->pat[h & (PAT_COUNT - 1)]);   /* :55-58 */
}

/* ------------------------------------------------------------------ */
/* Linear-probing hash table (hashtable.cc:31-54, hashtable.h:36-72)   */
/* ------------------------------------------------------------------ */

typedef struct {
  uint64_t  slots;
  uint64_t  mask;
  uint64_t *key;          /* full 64-bit hash                           */
  uint64_t *val;          /* sequence number in set 2                   */
  uint8_t  *used;         /* one bit per slot                           */
} table_t;

static int table_make(table_t *t, uint64_t n)
{
  t->slots = 1;
  while (FILL_PERCENT * t->slots < 100 * n)
    t->slots <<= 1;
  t->mask = t->slots - 1;
  t->used = (uint8_t *)calloc((t->slots + 63) / 8, 1);
  t->key  = (uint64_t *)malloc(t->slots * sizeof(uint64_t));
  t->val  = (uint64_t *)malloc(t->slots * sizeof(uint64_t));
  return (t->used && t->key && t->val) ? 0 : -1;
}

static void table_free(table_t *t)
{
  free(t->used);
  free(t->key);
  free(t->val);
}

static inline uint64_t table_home(const table_t *t, uint64_t h)
{
  return (h >> 32) & t->mask;                      /* hashtable.h:36-41 */
}

static inline int table_used(const table_t *t, uint64_t s)
{
  return t->used[s >> 3] & (1u << (s & 7));
}

/* ------------------------------------------------------------------ */
/* set accessors                                                        */
/* ------------------------------------------------------------------ */

static inline const uint8_t *seq_of(const oracle_set *s, uint64_t i)
{
  return s->residues + s->offsets[i];
}

static inline unsigned len_of(const oracle_set *s, uint64_t i)
{
  return (unsigned)(s->offsets[i + 1] - s->offsets[i]);
}

/* ------------------------------------------------------------------ */
/* index build with duplicate counting (overlap.cc:63-128)             */
/* ------------------------------------------------------------------ */

static int index_insert(const oracle_set *s, const uint64_t *hashes,
                        table_t *t, bloom_t *b, uint64_t i, int ignore_genes)
{
  int dup = 0;
  uint64_t h = hashes[i];
  uint64_t slot = table_home(t, h);
  while (table_used(t, slot)) {
    if ((!b || bloom_maybe(b, h)) && t->key[slot] == h) {
      uint64_t o = t->val[slot];
      if (s->repertoire[i] == s->repertoire[o] &&
          (ignore_genes ||
           (s->v_gene[i] == s->v_gene[o] && s->j_gene[i] == s->j_gene[o])) &&
          len_of(s, i) == len_of(s, o) &&
          memcmp(seq_of(s, i), seq_of(s, o), len_of(s, i)) == 0)
        dup = 1;
    }
    slot = (slot + 1) & t->mask;
  }
  t->used[slot >> 3] |= (uint8_t)(1u << (slot & 7));
  t->key[slot] = h;
  t->val[slot] = i;
  if (b)
    bloom_insert(b, h);
  return dup;
}

/* ------------------------------------------------------------------ */
/* variants (variants.h:65-73, variants.cc:53-107, 260-428)            */
/* ------------------------------------------------------------------ */

enum { K_SAME, K_SUB, K_DEL, K_INS, K_SUB2 };

typedef struct {
  uint64_t hash;
  uint32_t kind;
  uint32_t p1, p2;
  uint8_t  r1, r2;
} variant_t;

static uint64_t variant_bound(uint64_t L, int A, int d, int indels)
{
  uint64_t n = 1;
  if (d >= 1) {
    n += L * (uint64_t)(A - 1);
    if (indels)
      n += L + (L + 1) * (uint64_t)(A - 1) + 1;
  }
  if (d >= 2)
    n += L * (L - (L ? 1 : 0)) / 2 * (uint64_t)(A - 1) * (uint64_t)(A - 1);
  return n;
}

static inline void emit(variant_t *list, unsigned *n, uint64_t h, uint32_t kind,
                        uint32_t p1, uint8_t r1, uint32_t p2, uint8_t r2)
{
  variant_t *x = list + (*n)++;
  x->hash = h;
  x->kind = kind;
  x->p1 = p1;
  x->r1 = r1;
  x->p2 = p2;
  x->r2 = r2;
}

static unsigned enumerate_variants(const zob_t *z, const oracle_opts *o,
                                   uint64_t h0, const uint8_t *s, unsigned L,
                                   uint32_t v, uint32_t j, variant_t *list)
{
  const unsigned A = (unsigned)o->alphabet_size;
  unsigned n = 0;

  emit(list, &n, h0, K_SAME, 0, 0, 0, 0);            /* variants.cc:260-268 */

  if (o->differences >= 1) {
    /* single substitutions, position-major, original residue skipped
       (variants.cc:280-293) */
    for (unsigned p = 0; p < L; p++) {
      uint64_t without = h0 ^ zk(z, p, s[p]);
      for (unsigned r = 0; r < A; r++)
        if (r != s[p])
          emit(list, &n, without ^ zk(z, p, r), K_SUB, p, (uint8_t)r, 0, 0);
    }

    if (o->indels) {
      /* deletions: one per run of equal residues, rolling hash
         (variants.cc:301-325) */
      if (L > 1) {
        uint64_t h = zob_hash_shifted(z, s, L, v, j, 1, -1);
        emit(list, &n, h, K_DEL, 0, 0, 0, 0);
        uint8_t gone = s[0];
        for (unsigned p = 1; p < L; p++) {
          if (s[p] != gone) {
            h ^= zk(z, p - 1, gone) ^ zk(z, p - 1, s[p]);
            emit(list, &n, h, K_DEL, p, 0, 0, 0);
            gone = s[p];
          }
        }
      }
      /* insertions: all residues before position 0, then after each
         position every residue different from it (variants.cc:329-353) */
      uint64_t h = zob_hash_shifted(z, s, L, v, j, 0, +1);
      for (unsigned r = 0; r < A; r++)
        emit(list, &n, h ^ zk(z, 0, r), K_INS, 0, (uint8_t)r, 0, 0);
      for (unsigned p = 0; p < L; p++) {
        h ^= zk(z, p, s[p]) ^ zk(z, p + 1, s[p]);
        for (unsigned r = 0; r < A; r++)
          if (r != s[p])
            emit(list, &n, h ^ zk(z, p + 1, r), K_INS, p + 1, (uint8_t)r, 0, 0);
      }
    }
  }

  if (o->differences >= 2) {
    /* double substitutions p < q (variants.cc:370-399) */
    for (unsigned p = 0; p < L; p++) {
      uint64_t hp = h0 ^ zk(z, p, s[p]);
      for (unsigned r = 0; r < A; r++) {
        if (r == s[p])
          continue;
        uint64_t hpr = hp ^ zk(z, p, r);
        for (unsigned q = p + 1; q < L; q++) {
          uint64_t hq = hpr ^ zk(z, q, s[q]);
          for (unsigned w = 0; w < A; w++)
            if (w != s[q])
              emit(list, &n, hq ^ zk(z, q, w), K_SUB2, p, (uint8_t)r, q, (uint8_t)w);
        }
      }
    }
  }
  return n;
}

/* exact verification of a hash-equal candidate (variants.cc:166-240) */
static int variant_is(const uint8_t *s, unsigned L, const variant_t *x,
                      const uint8_t *t, unsigned M)
{
  switch (x->kind) {
  case K_SAME:
    return L == M && memcmp(s, t, L) == 0;
  case K_SUB:
    return L == M && t[x->p1] == x->r1 &&
           memcmp(s, t, x->p1) == 0 &&
           memcmp(s + x->p1 + 1, t + x->p1 + 1, L - x->p1 - 1) == 0;
  case K_DEL:
    return L - 1 == M &&
           memcmp(s, t, x->p1) == 0 &&
           memcmp(s + x->p1 + 1, t + x->p1, L - x->p1 - 1) == 0;
  case K_INS:
    return L + 1 == M && t[x->p1] == x->r1 &&
           memcmp(s, t, x->p1) == 0 &&
           memcmp(s + x->p1, t + x->p1 + 1, L - x->p1) == 0;
  case K_SUB2:
    return L == M && t[x->p1] == x->r1 && t[x->p2] == x->r2 &&
           memcmp(s, t, x->p1) == 0 &&
           memcmp(s + x->p1 + 1, t + x->p1 + 1, x->p2 - x->p1 - 1) == 0 &&
           memcmp(s + x->p2 + 1, t + x->p2 + 1, L - x->p2 - 1) == 0;
  }
  return 0;
}

/* compute_score (overlap.cc:144-166) */
static inline double pair_score(const oracle_opts *o, uint64_t a, uint64_t b)
{
  if (o->ignore_counts)
    return 1.0;
  switch (o->score) {
  case ORACLE_SCORE_MH:
  case ORACLE_SCORE_PRODUCT:
    return (double)a * (double)b;
  case ORACLE_SCORE_RATIO:
    return (double)a / (double)b;
  case ORACLE_SCORE_JACCARD:
  case ORACLE_SCORE_MIN:
    return (double)(a < b ? a : b);
  case ORACLE_SCORE_MAX:
    return (double)(a > b ? a : b);
  case ORACLE_SCORE_MEAN:
    return ((double)a + (double)b) / 2;
  }
  return 0.0;
}

/* ------------------------------------------------------------------ */
/* the per-query loop (overlap.cc:168-251, 253-284, 286-374, 376-538)  */
/* ------------------------------------------------------------------ */

typedef struct {
  const oracle_opts *o;
  const oracle_set  *s1, *s2;
  const zob_t       *z;
  bloom_t           *bloom;
  const table_t     *table;
  const uint64_t    *hash1;
  double            *matrix;       /* global                            */
  uint64_t           maxvar;
  pthread_mutex_t    lock;
  uint64_t           cursor;       /* next unclaimed query              */
  oracle_stats       st;
  int                failed;
  /* optional pair sink (overlap.cc:232-245) */
  uint32_t          *pair_q, *pair_h;
  uint64_t           pair_cap, pair_count;
  int                want_pairs;
} job_t;

static void chain_walk(job_t *J, uint64_t q, const variant_t *x, double *M,
                       oracle_stats *st)
{
  const oracle_set *A = J->s1, *B = J->s2;
  const table_t *t = J->table;
  uint64_t slot = table_home(t, x->hash);
  while (table_used(t, slot)) {
    st->slots_visited++;
    if (t->key[slot] == x->hash) {
      uint64_t hit = t->val[slot];
      st->hash_equal++;
      if (J->o->ignore_genes ||
          (A->v_gene[q] == B->v_gene[hit] && A->j_gene[q] == B->j_gene[hit])) {
        if (variant_is(seq_of(A, q), len_of(A, q), x,
                       seq_of(B, hit), len_of(B, hit))) {
          /* overlap.cc:218-228: row = repertoire of the seed, or the seed with -x */
          const uint64_t row = J->o->existence ? q : A->repertoire[q];
          M[(uint64_t)B->n_repertoires * row + B->repertoire[hit]] +=
              pair_score(J->o, A->count[q], B->count[hit]);
          st->matches++;
          if (J->want_pairs) {
            pthread_mutex_lock(&J->lock);
            if (J->pair_count < J->pair_cap) {
              J->pair_q[J->pair_count] = (uint32_t)q;
              J->pair_h[J->pair_count] = (uint32_t)hit;
            }
            J->pair_count++;
            pthread_mutex_unlock(&J->lock);
          }
        }
      }
    }
    slot = (slot + 1) & t->mask;
  }
}

static void query_hashed(job_t *J, uint64_t q, variant_t *list, double *M,
                         oracle_stats *st)
{
  const oracle_set *A = J->s1;
  unsigned n = enumerate_variants(J->z, J->o, J->hash1[q], seq_of(A, q),
                                  len_of(A, q), A->v_gene[q], A->j_gene[q], list);
  st->variants += n;
  for (unsigned k = 0; k < n; k++)
    if (bloom_maybe(J->bloom, list[k].hash)) {
      st->bloom_positive++;
      chain_walk(J, q, list + k, M, st);
    }
}

/* d > 2: equal-length Hamming scan over all of set 2 (overlap.cc:286-359,
   util.cc:172-184) */
static void query_scan(job_t *J, uint64_t q, double *M, oracle_stats *st)
{
  const oracle_set *A = J->s1, *B = J->s2;
  const uint8_t *s = seq_of(A, q);
  unsigned L = len_of(A, q);
  for (uint64_t hit = 0; hit < B->n; hit++) {
    if (!J->o->ignore_genes &&
        !(A->v_gene[q] == B->v_gene[hit] && A->j_gene[q] == B->j_gene[hit]))
      continue;
    if (len_of(B, hit) != L)
      continue;
    const uint8_t *t = seq_of(B, hit);
    int64_t diffs = 0;
    for (unsigned p = 0; p < L && diffs <= J->o->differences; p++)
      diffs += s[p] != t[p];
    if (diffs <= J->o->differences) {
      if (J->want_pairs) {
        pthread_mutex_lock(&J->lock);
        if (J->pair_count < J->pair_cap) {
          J->pair_q[J->pair_count] = (uint32_t)q;
          J->pair_h[J->pair_count] = (uint32_t)hit;
        }
        J->pair_count++;
        pthread_mutex_unlock(&J->lock);
      }
      const uint64_t row = J->o->existence ? q : A->repertoire[q];
      M[(uint64_t)B->n_repertoires * row + B->repertoire[hit]] +=
          pair_score(J->o, A->count[q], B->count[hit]);
      st->matches++;
    }
  }
}

static void *worker(void *arg)
{
  job_t *J = (job_t *)arg;
  const uint64_t cells = (J->o->existence ? J->s1->n : (uint64_t)J->s1->n_repertoires) *
                         J->s2->n_repertoires;
  const int hashed = J->o->differences <= MAX_HASHED_D;
  oracle_stats st;
  memset(&st, 0, sizeof st);

  variant_t *list = NULL;
  if (hashed) {
    list = (variant_t *)malloc((J->maxvar ? J->maxvar : 1) * sizeof(variant_t));
    if (!list) {
      J->failed = 1;
      return NULL;
    }
  }
  /* private matrix per thread when threaded (overlap.cc:393-416) */
  double *M = J->matrix;
  if (J->o->threads > 1) {
    M = (double *)calloc(cells ? cells : 1, sizeof(double));
    if (!M) {
      free(list);
      J->failed = 1;
      return NULL;
    }
  }

  for (;;) {
    /* dynamic chunks of 1000 queries from a shared cursor (overlap.cc:418-433) */
    pthread_mutex_lock(&J->lock);
    uint64_t first = J->cursor;
    uint64_t last = first + QUERY_CHUNK;
    if (last > J->s1->n)
      last = J->s1->n;
    J->cursor = last;
    pthread_mutex_unlock(&J->lock);
    if (first >= last)
      break;
    for (uint64_t q = first; q < last; q++) {
      if (hashed)
        query_hashed(J, q, list, M, &st);
      else
        query_scan(J, q, M, &st);
    }
  }

  /* merge (overlap.cc:510-527) */
  pthread_mutex_lock(&J->lock);
  if (M != J->matrix)
    for (uint64_t k = 0; k < cells; k++)
      J->matrix[k] += M[k];
  J->st.variants += st.variants;
  J->st.bloom_positive += st.bloom_positive;
  J->st.slots_visited += st.slots_visited;
  J->st.hash_equal += st.hash_equal;
  J->st.matches += st.matches;
  pthread_mutex_unlock(&J->lock);

  if (M != J->matrix)
    free(M);
  free(list);
  return NULL;
}

static unsigned longest_of(const oracle_set *s)
{
  unsigned m = 0;
  for (uint64_t i = 0; i < s->n; i++)
    if (len_of(s, i) > m)
      m = len_of(s, i);
  return m;
}

static uint64_t *hash_all(const zob_t *z, const oracle_set *s)
{
  /* db_hash (db.cc:903-916) */
  uint64_t *h = (uint64_t *)malloc((s->n ? s->n : 1) * sizeof(uint64_t));
  if (!h)
    return NULL;
  for (uint64_t i = 0; i < s->n; i++)
    h[i] = zob_hash_shifted(z, seq_of(s, i), len_of(s, i),
                            s->v_gene[i], s->j_gene[i], 0, 0);
  return h;
}

static int run_overlap(const oracle_opts *o, const oracle_set *s1, const oracle_set *s2,
                       double *matrix, oracle_stats *stats, uint64_t pair_cap,
                       uint32_t *pair_q, uint32_t *pair_h, uint64_t *pair_count);

int oracle_overlap(const oracle_opts *o, const oracle_set *s1,
                   const oracle_set *s2, double *matrix, oracle_stats *stats)
{
  return run_overlap(o, s1, s2, matrix, stats, 0, NULL, NULL, NULL);
}

int oracle_pairs(const oracle_opts *o, const oracle_set *s1, const oracle_set *s2,
                 uint64_t capacity, uint32_t *seed_out, uint32_t *hit_out, uint64_t *count)
{
  const uint64_t cells = (o->existence ? s1->n : (uint64_t)s1->n_repertoires) * s2->n_repertoires;
  double *m = (double *)malloc((cells ? cells : 1) * sizeof(double));
  if (!m || !count)
    return -1;
  oracle_opts one = *o;
  one.threads = 1;
  int rc = run_overlap(&one, s1, s2, m, NULL, capacity, seed_out, hit_out, count);
  free(m);
  return rc;
}

static int run_overlap(const oracle_opts *o, const oracle_set *s1, const oracle_set *s2,
                       double *matrix, oracle_stats *stats, uint64_t pair_cap,
                       uint32_t *pair_q, uint32_t *pair_h, uint64_t *pair_count)
{
  if (o->differences < 0 || (o->indels && o->differences != 1) ||
      (o->alphabet_size != 20 && o->alphabet_size != 4) ||
      o->threads < 1 || o->threads > 256 ||
      o->score < 0 || o->score > ORACLE_SCORE_JACCARD)
    return -1;

  const int same = (s1 == s2);
  const uint64_t cells = (o->existence ? s1->n : (uint64_t)s1->n_repertoires) * s2->n_repertoires;
  for (uint64_t k = 0; k < cells; k++)
    matrix[k] = 0;

  job_t J;
  memset(&J, 0, sizeof J);
  J.o = o;
  J.s1 = s1;
  J.s2 = s2;
  J.matrix = matrix;
  J.want_pairs = pair_count != NULL;
  J.pair_cap = pair_cap;
  J.pair_q = pair_q;
  J.pair_h = pair_h;
  pthread_mutex_init(&J.lock, NULL);

  zob_t z;
  memset(&z, 0, sizeof z);
  bloom_t *bloom = NULL;
  table_t table;
  memset(&table, 0, sizeof table);
  uint64_t *h1 = NULL, *h2 = NULL;
  int rc = 0;

  double t0 = now_seconds();
  if (o->differences <= MAX_HASHED_D) {
    unsigned l1 = longest_of(s1), l2 = longest_of(s2);
    unsigned longest = l1 > l2 ? l1 : l2;

    srandom(1);                                         /* compairr.cc:747 */
    if (zob_init(&z, o->alphabet_size, longest + EXTRA_POSITIONS,
                 o->n_v_genes, o->n_j_genes, !o->ignore_genes))
      return -1;                                        /* overlap.cc:840  */

    h1 = hash_all(&z, s1);
    if (!h1) { rc = -1; goto done; }
    if (!same) {
      /* check_duplicates(set 1) on a throw-away table (overlap.cc:579-605) */
      table_t tmp;
      if (table_make(&tmp, s1->n)) { table_free(&tmp); rc = -1; goto done; }
      for (uint64_t i = 0; i < s1->n; i++)
        J.st.dup_set1 += (uint64_t)index_insert(s1, h1, &tmp, NULL, i,
                                                o->ignore_genes);
      table_free(&tmp);
      h2 = hash_all(&z, s2);
      if (!h2) { rc = -1; goto done; }
    } else {
      h2 = h1;
    }

    /* overlap.cc:861-873 */
    if (table_make(&table, s2->n)) { rc = -1; goto done; }
    bloom = (bloom_t *)malloc(sizeof(bloom_t));
    if (!bloom || bloom_make(bloom, table.slots)) { rc = -1; goto done; }
    for (uint64_t i = 0; i < s2->n; i++)
      J.st.dup_set2 += (uint64_t)index_insert(s2, h2, &table, bloom, i,
                                              o->ignore_genes);

    J.z = &z;
    J.bloom = bloom;
    J.table = &table;
    J.hash1 = h1;
    J.maxvar = variant_bound(l1, o->alphabet_size, o->differences, o->indels);
  }
  double t1 = now_seconds();

  /* overlap.cc:926-936 */
  if (o->threads == 1) {
    worker(&J);
  } else {
    pthread_t *th = (pthread_t *)malloc((size_t)o->threads * sizeof(pthread_t));
    if (!th) { rc = -1; goto done; }
    for (int t = 0; t < o->threads; t++)
      pthread_create(th + t, NULL, worker, &J);
    for (int t = 0; t < o->threads; t++)
      pthread_join(th[t], NULL);
    free(th);
  }
  double t2 = now_seconds();
  if (J.failed)
    rc = -1;

  J.st.seconds_index = t1 - t0;
  J.st.seconds_analysis = t2 - t1;
  if (stats)
    *stats = J.st;
  if (pair_count)
    *pair_count = J.pair_count;

done:
  if (bloom) {
    free(bloom->words);
    free(bloom);
  }
  table_free(&table);
  if (h2 != h1)
    free(h2);
  free(h1);
  free(z.pos);
  pthread_mutex_destroy(&J.lock);
  return rc;
}

/* ------------------------------------------------------------------ */
/* independent brute force over the pair definition                    */
/* ------------------------------------------------------------------ */

/* is t equal to s with exactly one residue removed? (|s| == |t| + 1) */
static int one_removed(const uint8_t *s, unsigned L, const uint8_t *t)
{
  unsigned p = 0;
  while (p < L - 1 && s[p] == t[p])
    p++;
  for (; p < L - 1; p++)
    if (s[p + 1] != t[p])
      return 0;
  return 1;
}

int oracle_bruteforce(const oracle_opts *o, const oracle_set *A,
                      const oracle_set *B, double *matrix)
{
  const uint64_t cells = (o->existence ? A->n : (uint64_t)A->n_repertoires) * B->n_repertoires;
  for (uint64_t k = 0; k < cells; k++)
    matrix[k] = 0;
  for (uint64_t q = 0; q < A->n; q++) {
    const uint8_t *s = seq_of(A, q);
    unsigned L = len_of(A, q);
    for (uint64_t h = 0; h < B->n; h++) {
      if (!o->ignore_genes &&
          (A->v_gene[q] != B->v_gene[h] || A->j_gene[q] != B->j_gene[h]))
        continue;
      const uint8_t *t = seq_of(B, h);
      unsigned M = len_of(B, h);
      int ok = 0;
      if (L == M) {
        int64_t diffs = 0;
        for (unsigned p = 0; p < L; p++)
          diffs += s[p] != t[p];
        ok = diffs <= o->differences;
      } else if (o->indels && o->differences == 1) {
        if (L == M + 1)
          ok = one_removed(s, L, t);
        else if (M == L + 1)
          ok = one_removed(t, M, s);
      }
      if (ok)
        matrix[(uint64_t)B->n_repertoires * (o->existence ? q : A->repertoire[q]) + B->repertoire[h]] +=
            pair_score(o, A->count[q], B->count[h]);
    }
  }
  return 0;
}

/* show_matrix_value (overlap.cc:540-577) */
double oracle_cell_value(const oracle_opts *o, double cell,
                         double count1, double sq1, double count2, double sq2)
{
  if (o->score == ORACLE_SCORE_MH) {
    double lx = sq1 / count1 / count1;
    double ly = sq2 / count2 / count2;
    double xy = 1.0 * count1 * count2;
    return (2.0 * cell) / ((lx + ly) * xy);
  }
  if (o->score == ORACLE_SCORE_JACCARD)
    return cell / (count1 + count2 - cell);
  return cell;
}
