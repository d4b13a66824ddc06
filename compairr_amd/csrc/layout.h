/*
 * layout.h -- device data layout shared by the kernels and the host side of
 * libcompairr_hip.so.  gfx950 only; see DESIGN.md "Data layout in HBM".
 */
#ifndef COMPAIRR_AMD_LAYOUT_H
#define COMPAIRR_AMD_LAYOUT_H

#include <stdint.h>

namespace cmpr {

constexpr int      WAVE              = 64;
constexpr int      BLOCK_THREADS     = 256;
constexpr int      WAVES_PER_BLOCK   = BLOCK_THREADS / WAVE;

/* Zobrist table: A * zpos position keys, then n_v V keys, then n_j J keys.
   zpos = longest + 3 (insertion variants index position L, reference
   MAX_INSERTS, compairr.h:111). */
constexpr uint32_t EXTRA_POSITIONS   = 3;

/* Blocked Bloom filter, one 64-bit word per block, inverted polarity (a cleared
   bit = present) so that the membership test is (word & pattern) == 0 -- the
   block geometry of bloompat.h:22-58.  The reference sets 8 bits per key taken
   from ONE table of 1024 patterns, which floors its false-positive rate at
   (keys per word) / 1024; this build sets 4 bits per key computed directly from
   20 hash bits (kernels.h pattern_of): a 2^20 pattern space and no table. */
constexpr uint32_t PATTERN_BITS      = 10;     /* per dword of the word: hash bits [0,5)
                                                  and [8,13) / [16,21) and [24,29); the
                                                  word address is taken from bit 35 up */
constexpr uint32_t PATTERN_K         = 4;

/* Open-addressing table: 64-bit key (the sequence hash) and 32-bit payload (the
   set-2 sequence number) side by side in one 16-byte slot, so that a probe
   reads key and payload with one access; an all-ones key marks a free slot
   (the reference keeps three arrays: hash_values, hash_data and an occupancy
   bitmap, hashtable.h:22-29,48-72). */
struct Slot {
  uint64_t key;
  uint32_t val;
  uint32_t pad;
};
constexpr uint64_t EMPTY_KEY         = ~0ull;
constexpr uint32_t FILL_PERCENT      = 70;     /* hashtable.cc:24 */

/* variant kinds, mutation_kind_enum of variants.h:56-63 */
enum : uint32_t { K_SAME = 0, K_SUB = 1, K_DEL = 2, K_INS = 3, K_SUB2 = 4 };

/* One tile = 64 queries of equal length, one per lane.  Residues are stored
   four to a dword, position-major and lane-minor:
     qres[res_base + (p / 4) * 64 + lane]  byte (p % 4)
   so that one wave instruction reads four positions of all 64 queries as one
   256-byte coalesced access. */
/* up to 96 nucleotides, 2 bits each (sub2 items of variant 1) */
constexpr uint32_t RESPACK_MAX = 96;
struct alignas(8) ResPack {
  uint32_t w[RESPACK_MAX / 16];
};

struct TileDesc {
  uint32_t len;       /* residues of the longest query of the tile          */
  uint32_t nvalid;    /* lanes 0..nvalid-1 hold queries, the rest is padding */
  uint32_t res_base;  /* dword offset of the tile's residues in qres         */
  uint32_t pass;      /* 0 (variant 1: the indel passes 1, 2 reuse the tiles of
                         pass 0 and are named by their Chunk)                 */
  uint32_t slice;     /* filter slice the tile's probes go to (sliced modes)  */
  uint32_t k;         /* class residues of the tile's queries: 0 (light) or K  */
};

/* Sliced Bloom layout (kernel variant 1).  The filter is cut into S = 2^s
   slices of 2^w words; a sequence's slice is NOT taken from its Zobrist hash
   but from a "class key" that is invariant under most single edits:
       base  = CL[len] ^ CV[v] ^ CJ[j]
       ckey  = base ^ (heavy(base) ? CR[0][seq[m_0]] ^ ... ^ CR[K-1][seq[m_K-1]] : 0)
       slice = ckey & (S - 1),   m_i = (c0 + i) % len
   heavy(base) is one bit of a 65536-bucket bitmap filled from set 2: a
   (len, V, J) class that alone would overfill a slice is split over up to
   20^K slices by K "class residues" at the fixed positions c0 .. c0+K-1 (c0
   is picked from set 2's length distribution so that they lie inside almost
   every sequence); all other ("light") classes are not split.  Hence every substitution variant
   of a light query, and every substitution variant of a heavy query that does
   not touch a class position, lives in the query's own slice.  Queries are
   grouped by slice, a workgroup stages that slice into LDS once and answers
   those probes from LDS; only variants that change the class key go to the
   filter in HBM.  The word inside a slice and the bit pattern still come from
   the Zobrist hash. */
constexpr uint32_t MAX_CLASS_RES      = 8;    /* table rows; K <= 4 for aa (20^4 splits),
                                                 K <= 8 for nt (4^8 splits)            */
/* the most class residues a layout may use, and the most a kernel instantiation unrolls its
   class-residue loops for.  Amino acids: three, and four in the WIDE instantiations of
   probe_rows_kernel -- taken only when three leave the fullest slice well over its budget (the
   (V, J) classes of -i on skewed data: ref_index.hip); the wide form of the -i kernel pays
   for its fourth residue with 12 bytes of scratch, so the narrow one stays the default. */
__host__ __device__ constexpr uint32_t max_class_res(uint32_t A) { return A == 4 ? 8u : 4u; }
__host__ __device__ constexpr uint32_t kernel_class_res(uint32_t A, bool wide) { return A == 4 ? 8u : wide ? 4u : 3u; }
constexpr uint32_t SLICE_WORDS_LOG2   = 12;    /* 4096 words = 32 KiB per slice */
constexpr uint32_t ROW_WORD_BYTES      = 32;   /* variant 2: filter word = 8 dwords, one bit of each per entry */
constexpr uint32_t MAX_ROW_SLICE_WORDS = 640;  /* variant 2: at most 20 KiB per slice (a ring of 4 in LDS) */
constexpr uint32_t HEAVY_BUCKETS_LOG2 = 16;
constexpr uint32_t HEAVY_WORDS        = (1u << HEAVY_BUCKETS_LOG2) / 32;

struct SliceGeom {
  uint32_t smask;          /* S - 1                                          */
  uint32_t words_log2;     /* w                                              */
  uint32_t k;              /* K: class residues of heavy classes               */
  uint32_t ncl;            /* entries of CL (lengths 0..ncl-1)               */
  /* class tables, u32 each:
     CL[ncl] | CV[n_v] | CJ[n_j] | CR[MAX_CLASS_RES][A] | heavy bitmap[HEAVY_WORDS] */
  const uint32_t *ctab;
  uint32_t off_cv, off_cj, off_cr, off_hv;
  uint32_t c0;             /* first class position                          */
  uint32_t rw_words;       /* variant 2 (kernels_rows.h): 32-byte words per
                              slice, any count <= MAX_ROW_SLICE_WORDS          */
  uint32_t cmask;          /* variant 2: slices per class part - 1 (row_slice)  */
  uint32_t nbuf;           /* kernels_pairs2.h: slice buffers in LDS -- 2: the next chunk's slice is copied
                              while this one is worked on; 1: slices twice the size (more queries per slice:
                              fuller tiles), staged between two barriers */
  /* Pages (variant 2, d = 1; round 5).  A slice that holds more entries than its words are good for -- skewed
     data: the commonest combination of class residues of a big (V, J) class -- is given 2^e pages of rw_words
     words each: an entry goes to the page its hash names (page_of), page 0 where the slice lies, the others in
     an overflow area behind the nsl regular slices.  page_tab[s] = e | first overflow slice << 4 (device
     memory; NULL: no slice has pages).  A tile of such a slice is worked on once per page (a chunk per page:
     Chunk::pass bits), each variant counting in the pass of its own page. */
  const uint32_t *page_tab;
  uint32_t nsl;            /* regular slices: main part + class parts */
  uint32_t pad_pages;
};
constexpr uint32_t PAGE_E_MAX = 3;            /* at most 8 pages per slice */
constexpr uint32_t PAGE_HASH_SHIFT = 20;      /* page = bits 20.. of the entry's hash: used neither by the word
                                                 index (top 16) nor by the rotation amounts (kernels_rows.h) */
constexpr uint32_t CHUNK_PAGE_SHIFT = 20;     /* Chunk::pass (variant 2): bits 20..22 the page the chunk stages, */
constexpr uint32_t CHUNK_PAGE_E_SHIFT = 24;   /* bits 24..25 e of its slice (0: the slice has no pages) */
__host__ __device__ inline uint32_t page_of(uint64_t Wk, uint32_t e)
{
  return ((uint32_t)Wk >> PAGE_HASH_SHIFT) & ((1u << e) - 1u);
}
/* where page `pg` of slice `s` lies, in slices */
__host__ __device__ inline uint32_t page_slice(const SliceGeom &g, uint32_t s, uint32_t pt, uint32_t pg)
{
  return pg == 0u ? s : g.nsl + (pt >> 4) + pg - 1u;
}
/* ... of an entry / a probe with hash Wk filed under slice s */
__host__ __device__ inline uint32_t paged_slice(const SliceGeom &g, uint32_t s, uint64_t Wk)
{
  if (g.page_tab == nullptr)
    return s;
  const uint32_t pt = g.page_tab[s];
  return page_slice(g, s, pt, page_of(Wk, pt & 15u));
}

/* Variant 2 files the entries of a row under the class key WITHOUT the terms of
   the blanked position.  Rows whose blanked position is class position i of a
   split class live in "class part" i of the filter, a run of cmask + 1 slices
   behind the smask + 1 slices of the main part: those rows are 1 / (L + 1) of the
   entries, and a pass over them stages only its own part. */
__host__ __device__ inline uint32_t row_slice(const SliceGeom &g, uint32_t key, int class_index)
{
  return class_index < 0 ? (key & g.smask)
                         : g.smask + 1u + (uint32_t)class_index * (g.cmask + 1u) + (key & g.cmask);
}

__host__ __device__ inline uint32_t class_pos(uint32_t len, uint32_t i, uint32_t c0)
{
  /* (no division where the position does not wrap -- every sequence of at least c0 + K residues: an integer
     modulo is ~40 instructions on this chip, and the layout kernels ask a dozen times per query) */
  const uint32_t x = c0 + i;
  return x < len ? x : len ? x % len : 0;
}

/* Pair rows: the class PART of a pair that holds class positions is keyed by what is left of
   the class residues -- one of three when a pair swallows two -- and that is too coarse for
   the big (V, J) classes of -i (no length term): 3 x 10^5 sequences twenty ways overfill
   their slices.  So the slices of the class parts (only they) are keyed by PAIR_EXTRA more
   residues, at the positions behind the class positions, those inside the pair left out
   like the class positions inside it.  `res_at(pos)` = the residue the sequence at hand
   (a set-2 sequence, a query, or an indel variant of either) has at pos. */
constexpr uint32_t PAIR_EXTRA = 3;
template <typename F>
__host__ __device__ inline uint32_t pair_part_terms(const uint32_t *t, const SliceGeom &g, uint32_t A,
                                                    uint32_t len, uint32_t pair_pos, F res_at)
{
  uint32_t x = 0;
  if (g.k + PAIR_EXTRA > MAX_CLASS_RES || len == 0)
    return 0;
  for (uint32_t e = 0; e < PAIR_EXTRA; e++) {
    const uint32_t pos = class_pos(len, g.k + e, g.c0);
    if ((pos & ~1u) != pair_pos)
      x ^= t[g.off_cr + (g.k + e) * A + res_at(pos)];
  }
  return x;
}

/* `t` = the class tables (host vector or device pointer, same layout) */
__host__ __device__ inline uint32_t class_base(const uint32_t *t, const SliceGeom &g,
                                               bool genes, uint32_t L, uint32_t v, uint32_t j)
{
  uint32_t b = t[L];
  if (genes)
    b ^= t[g.off_cv + v] ^ t[g.off_cj + j];
  return b;
}

__host__ __device__ inline bool class_is_heavy(const uint32_t *t, const SliceGeom &g,
                                               uint32_t base)
{
  const uint32_t b = base >> (32 - HEAVY_BUCKETS_LOG2);
  return (t[g.off_hv + (b >> 5)] >> (b & 31u)) & 1u;
}

__host__ __device__ inline uint32_t class_key_of(const uint32_t *t, const SliceGeom &g,
                                                 uint32_t A, bool genes, const uint8_t *s,
                                                 uint32_t L, uint32_t v, uint32_t j,
                                                 bool *heavy_out)
{
  uint32_t ck = class_base(t, g, genes, L, v, j);
  const bool heavy = g.k > 0 && class_is_heavy(t, g, ck);
  if (heavy && L > 0)
    for (uint32_t i = 0; i < g.k; i++)
      ck ^= t[g.off_cr + i * A + s[class_pos(L, i, g.c0)]];
  if (heavy_out)
    *heavy_out = heavy;
  return ck;
}

/* tile descriptor + tile number: the unit the chunks list, staged in LDS per chunk */
struct TileRef {
  TileDesc td;
  uint32_t t;            /* tile number: slot = t * 64 + lane                */
  uint32_t pad;
};

/* one block-level work item of the sliced kernel: tiles of one slice */
struct Chunk {
  uint32_t slice;        /* slice to stage in LDS                            */
  uint32_t first_tile;   /* index into ProbeParams::tile_refs                */
  uint32_t ntiles;
  uint32_t pass;         /* 0: all rows of the tiles' own slice; variant 1 with -i:
                            1 insertion / 2 deletion rows with the sibling slice
                            (own ^ CL[L] ^ CL[L+-1]) staged; variant 2: 3 + g = a
                            chunk of items of group g (first_tile = first item,
                            ntiles = blocks of 64 items)                        */
};

/* Chunk::pass bit: a main chunk of variant 1 that also hands out the item blocks of its
   slice (ProbeParams::slice_items), behind its tiles */
constexpr uint32_t CHUNK_WITH_ITEMS = 0x100u;
constexpr uint32_t NPART = 128;              /* partial-result slots (ProbeParams::part) */
constexpr uint32_t PART_CELLS_MAX = 65536;   /* largest matrix kept in the slots (x NPART x 8 bytes = 64 MiB) */
constexpr uint32_t POS_CTR_STRIDE = 16;      /* u64s: one 128-byte line per segment */
/* probe_rows_kernel's chunk counters (ProbeParams::deal_ctr): the workgroups of group blockIdx % DEAL_GROUPS
   share one, on a line of its own (one counter for the whole grid: 18 000 answered atomics on one address
   took as long as the launch); the second half belongs to the redo launch */
constexpr uint32_t DEAL_GROUPS = 8, DEAL_STRIDE = 8, DEAL_WORDS = 2 * DEAL_GROUPS * DEAL_STRIDE;

/* entries per position of the sliced kernel's LDS copy of the Zobrist table:
   amino-acid rows are stored twice in a line (kernels_sliced.h row_lds_others) */
__host__ __device__ constexpr uint32_t zrow_stride(int A)
{
  return A == 4 ? 4u : 2u * (uint32_t)A;
}

/* entries per position of the nucleotide kernels' second LDS table (kernels_sliced.h):
   4 residues x {own key, 3 replacement deltas} */
__host__ __device__ constexpr uint32_t zdelta_entries(int A)
{
  return A == 4 ? 16u : 0u;
}

/* One item (kernels_rows.h passes >= 3; kernels_sliced.h sub2 items): the row's blanked
   hash (sub2: the query's hash), the query's slot (~0: padding behind the items of a
   slice) and its residue | position << 8 | kind << 24 -- 16 bytes, one load per lane */
constexpr uint32_t ITEM_DEL_COUNTS = 1u << 27;   /* ItemRec::rp of a pair item with -i: "q without its first
                                                    position" is a variant (first of a run of equal
                                                    residues), answered by bit A + 1 of the same word
                                                    (kernels_rows.h) */
constexpr uint32_t ITEM_DEL2_COUNTS = 1u << 28;  /* a pair item (pair rows): the same for its second position */
struct alignas(16) ItemRec {
  uint64_t w;
  uint32_t main;
  uint32_t rp;
};

/* A Bloom-positive variant waiting for its hash-table walk */
struct PosEntry {
  uint64_t hash;
  uint32_t slot;     /* query: tile * 64 + lane                         */
  uint32_t ca, cb;   /* variant: kind | p1 << 3 | r1 << 24 ; p2 | r2 << 24 */
  uint32_t qbase;    /* (unused: the verification looks a long query's tile up itself) */
};
/* The positives buffer is written in blocks of 64 entries; the entries a wave
   has no positive for carry this slot */
constexpr uint32_t POS_NULL_SLOT = 0xffffffffu;

/* One set-2 sequence as the verification step reads it, and one slot of the record table
   (ref_index.hip): a 32-byte header and the first 32 residues, one byte each -- ONE 64-byte
   memory line per looked-up CDR3.  A longer sequence has its further residues where the set
   lies (ProbeParams::res2 / off2). */
constexpr uint32_t REC_RES = 32;              /* residues inside the record (amino acids: a byte each) */
constexpr uint32_t REC_RES_NT = 128;          /* ... nucleotides: two bits each, four to a byte (a CDR3 of 45 nucleotides is
                                                 verified from the slot alone, like one of 15 amino acids) */
constexpr uint32_t REC_EMPTY = 0xffffffffu;   /* RefRec::idx of an empty slot */
constexpr uint32_t REC_MORE = 1u << 16;       /* RefRec::len: the next slot holds another record of this bucket */
constexpr uint32_t REC_TAG_SHIFT = 17;        /* RefRec::len bits 17..31: 15 bits of the key (kernels.h dir_tag) */
struct RefRec {
  uint64_t cnt;       /* duplicate_count (1 with -f)                        */
  uint32_t idx;       /* the sequence's number in set 2 (REC_EMPTY: nothing here) */
  uint32_t len;       /* bits 0..15 the length, REC_MORE, tag              */
  uint32_t v, j;      /* 0 with -g                                          */
  uint32_t rep;
  uint32_t home;      /* bucket of the sequence's key: the slot its record lies in unless displaced */
  uint8_t  res[REC_RES];
};
static_assert(sizeof(RefRec) == 64, "a record is one 64-byte line");
constexpr uint32_t REC_UNIT = 16;

/* One set-1 sequence as the verification step reads it (resolve_kernel): 64 bytes
   per query slot -- a verified candidate costs one request for the hit's record
   and one for the query's. */
struct QueryRec {
  uint64_t cnt;       /* duplicate_count (1 with -f)                        */
  uint32_t v, j;      /* 0 with -g                                          */
  uint32_t rep;       /* matrix row                                         */
  uint32_t len;
  uint32_t res[9];    /* residues 0..35, four to a dword                    */
  uint32_t orig;      /* the query's index in the caller's set 1            */
};

/* per-launch kernel arguments */
struct ProbeParams {
  /* Zobrist */
  const uint64_t *zob;
  uint32_t        zpos;
  uint32_t        n_v;
  /* Bloom */
  const uint64_t *bloom;
  uint32_t        bloom_byte_mask;   /* (words - 1) << 3                     */
  uint32_t        deal;              /* probe_rows_kernel: != 0: the chunks beyond a workgroup's first RING are handed
                                        out by counters (deal_ctr), in list order: heaviest first */
  /* record table of set 2 (ref_index.hip): the records of hash bucket b lie from slot b of rec2 on */
  uint32_t        dir_mask;         /* buckets - 1 */
  uint32_t        rec_packed;       /* 1: nucleotides -- a record holds its first REC_RES_NT residues, two bits each */
  const uint32_t *bmap;             /* one bit per bucket: it holds a record (NULL: not consulted).  A filter's false
                                       positive names an empty bucket two times out of three; the bits of 2^25 buckets
                                       are 4 MiB -- answered from cache, where the table's slot is a line of HBM */
  /* set 2 records */
  const uint8_t  *res2;
  const uint64_t *off2;
  const uint32_t *v2;
  const uint32_t *j2;
  const uint32_t *rep2;
  const uint64_t *cnt2;
  const unsigned char *rec2;        /* the record table: RefRec slots */
  /* set 1 tiles */
  const TileDesc *tiles;
  const uint32_t *qres;
  const uint32_t *qv;
  const uint32_t *qj;
  const uint64_t *qgh;             /* per slot: V key ^ J key (NULL with -g); variant 2:
                                      the query's Zobrist hash                   */
  const uint64_t *qhins, *qhdel;   /* variant 2 with -i: the shifted hashes that seed
                                      the rolling indel enumeration              */
  const ItemRec  *items;           /* variant 2 class rows / variant 1 sub2: the flat items       */
  const ResPack  *cpk;             /* variant 1 sub2 items: the query's residues, 2 bits each
                                      (<= RESPACK_MAX positions); ItemRec::w then holds the query's hash.
                                      kernels_pairs2.h items: the same residues beside the pair-blanked hash */
  const ResPack  *qpk;             /* kernels_pairs2.h: per slot, the query's residues, 2 bits each */
  const uint16_t *qlen;            /* per slot: own length (<= tile len)         */
  uint32_t n_j_keys;               /* J-gene keys behind the V-gene keys of the Zobrist table (0 with -g) */
  uint32_t rec_tiles;              /* (round 6) amino acids, d = 1 without -i, every sequence within the record's 36
                                      residues: NO per-slot arrays are laid out -- probe_rows_kernel takes a tile's
                                      lengths and residues from the 64-byte QueryRecs scatter_kernel wrote and works
                                      the hashes out itself (the Zobrist keys are in its LDS anyway): fill_tiles_kernel,
                                      0.29 ms per 10M queries of reading those records and writing them out again
                                      position-major, does not run.  2: every sequence within 28 residues -- the query's
                                      Zobrist hash rides in the record's last two residue words (bytes 28 .. 35, where
                                      scatter_kernel leaves what keys_kernel worked out): the kernel does not hash */
  /* (repertoire, count and the query's number in the caller's set are read from its QueryRec, by the
     kernels that resolve a positive -- no per-slot arrays of them since round 6: 16 bytes per slot less for
     the layout to write) */
  const uint32_t *qck;             /* per slot: class key (variant 2)             */
  const QueryRec *qrec;            /* per slot: what verification reads            */
  uint32_t        ntiles;
  uint32_t        first_tile;
  /* Where a workgroup leaves its counters and its LDS copy of the matrix: slot
     blockIdx % NPART of `part` (part_stride u64 each: the matrix cells, then STAT_COUNT
     counters), summed into matrix / stats by reduce_partials_kernel.  Thousands of
     workgroups adding to the same 16 memory lines serialise (~8 ns per atomic and
     line: 150 us of a 260 us resolve kernel); NULL = add to matrix / stats directly. */
  unsigned long long *part;
  uint32_t        part_stride;
  uint32_t        sub2_items;  /* variant 1, nt, d = 2: class-position pairs are items (passes >= 3) */
  const uint32_t *slice_items; /* ... per slice {first item, blocks} riding along with its first main chunk */
  uint32_t        work_first, work_step;   /* variant 0: this launch takes tiles work_first + k work_step */
  /* output */
  unsigned long long *matrix;      /* R1 * R2 integer sums                   */
  double             *matrix_f64;  /* ratio score only                       */
  uint32_t        R1, R2;
  int32_t         score;
  int32_t         ignore_counts;
  int32_t         lds_matrix;      /* 1: privatise the matrix in LDS          */
  uint32_t        part_cells;      /* cells kept in the partial slots (`part`): all of them when the matrix is
                                      privatised in LDS; a matrix too large for LDS but of at most PART_CELLS_MAX
                                      cells is added to the workgroup's slot in HBM instead of to `matrix` itself --
                                      a self-comparison sends every identity pair to one of R diagonal cells, and
                                      10^5 atomics on one address serialise (24.2M sequences, 120 repertoires:
                                      resolve 3.6 ms; round 4) */
  /* deferred resolve: Bloom positives are appended here by the probe kernel and
     walked / verified / scored by resolve_kernel at full occupancy */
  PosEntry           *pos_buf;      /* NULL: resolve inline in the probe kernel  */
  unsigned long long *pos_ctr;      /* per segment, POS_CTR_STRIDE apart: [0]
                                       entries claimed (may exceed pos_cap), [1]
                                       ~(first claim that did not fit), 0 = none */
  uint64_t            pos_cap;      /* entries per segment (+ 64 of slack)       */
  uint32_t            pos_segments; /* power of two; workgroup b appends to
                                       segment b % pos_segments: claims are
                                       same-address atomics, which serialise    */
  uint32_t            redo;         /* 1: this launch is the redo pass (see overflow) */
  unsigned long long *overflow;     /* set by a fast-form probe launch (kernels_rows.h)
                                       whose positives did not fit: resolve_kernel
                                       skips, the redo launch does the step inline;
                                       NULL for kernels that resolve inline instead  */
  unsigned long long *deal_ctr;      /* DEAL_GROUPS counters, DEAL_STRIDE words apart (see `deal`) */
  /* pairs mode (cmpr_overlap_pairs): matches are listed, not scored */
  uint32_t           *pair_q, *pair_h;
  unsigned long long *pair_count;   /* NULL: matrix mode                        */
  uint64_t            pair_cap;
  /* sliced mode */
  SliceGeom       geom;
  const Chunk    *chunks;
  const uint32_t *small_tiles;     /* tiles handled by single waves, unstaged      */
  uint32_t        nsmall;
  uint32_t        pad3;
  const TileRef  *tile_refs;       /* chunk c covers tile_refs[first .. first+n)       */
  uint32_t        nchunks;
  uint32_t        chunk_cap;       /* tiles per chunk at most (variant 2: LDS tile refs per ring slot) */
  uint32_t        debug;           /* ablation switches, 0 in production      */
  /* work distribution + statistics */
  uint32_t           *tile_counter;  /* [0] chunks, [1] small tiles                  */
  unsigned long long *stats;       /* [variants, bloom+, hash==, matches]     */
};

/* ProbeParams::debug bits: timing experiments only, results become wrong.  They
   exist only in a library built with -DCMPR_ABLATION (make ablation); in the
   shipped build CMPR_DBG() is the constant false and the "debug" tunable is
   rejected. */
enum : uint32_t { DBG_SKIP_HBM_ROWS = 1, DBG_SKIP_EMIT = 2, DBG_SKIP_RESOLVE = 4,
                  DBG_SKIP_LDS_ROWS = 8, DBG_SKIP_INS_ROWS = 16, DBG_SKIP_DEL_ROWS = 32,
                  DBG_SKIP_TILES = 64, DBG_SKIP_CLASS_TILES = 128, DBG_SKIP_MAIN_TILES = 256,
                  /* resolve_kernel: the directory entry / the record / the query's record read at consecutive
                     places instead of where they are -- what does the randomness of each access cost? */
                  DBG_RES_SEQ_DIR = 512, DBG_RES_SEQ_REC = 1024, DBG_RES_SEQ_QREC = 2048, DBG_RES_NO_VERIFY = 4096 };
#ifdef CMPR_ABLATION
#define CMPR_DBG(P, bit) (((P).debug & (bit)) != 0)
#else
#define CMPR_DBG(P, bit) false
#endif

/* STAT_VARIANTS counts the variant tests the kernel executed (one per variant
   hash held against the filter); STAT_READS the filter words read for them */
#ifdef CMPR_PHASE_TIMING
enum { STAT_VARIANTS = 0, STAT_BLOOM_POS = 1, STAT_HASH_EQ = 2, STAT_MATCHES = 3,
       STAT_READS = 4, STAT_COUNT = 16 };      /* [8..15]: per-phase wave cycles (kernels_rows.h) */
#else
enum { STAT_VARIANTS = 0, STAT_BLOOM_POS = 1, STAT_HASH_EQ = 2, STAT_MATCHES = 3,
       STAT_READS = 4, STAT_COUNT = 5 };
#endif

}  // namespace cmpr
#endif
