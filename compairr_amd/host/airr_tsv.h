/*
 * airr_tsv.h -- AIRR rearrangement TSV reader producing the structure-of-arrays
 * view the GPU path consumes (include/compairr_hip.h: cmpr_set_view).
 *
 * Input rules follow the reference reader (/root/reference/src/db.cc:172-296
 * header, :298-706 line, :708-901 file); storage is SoA instead of the
 * reference's 56-byte seqinfo_s records (db.cc:77-88).
 */
#ifndef COMPAIRR_AMD_AIRR_TSV_H
#define COMPAIRR_AMD_AIRR_TSV_H

#include <stdint.h>
#include <cstdio>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "options.h"

namespace cmprhost {

/* first-appearance numbering of strings (repertoire ids per set; V and J gene
   names shared by both sets, db.cc:121-125) */
struct Interner {
  std::map<std::string, uint32_t> index;
  std::vector<std::string>        names;
  uint32_t intern(const char *s);
};

struct GeneTables {
  Interner v, j;
};

/* A vector of scalars whose resize() leaves new elements uninitialised (std::vector value-initialises
   them: 430 MB of zeros written by ONE thread per 10M-sequence file, a quarter of the time the threads of
   the reader then need to fill them in). */
template <typename T>
struct default_init_allocator : std::allocator<T> {
  template <typename U> struct rebind { typedef default_init_allocator<U> other; };
  default_init_allocator() {}
  template <typename U> default_init_allocator(const default_init_allocator<U> &) {}
  template <typename U> void construct(U *p) { ::new (static_cast<void *>(p)) U; }
  template <typename U, typename A1> void construct(U *p, const A1 &a1) { ::new (static_cast<void *>(p)) U(a1); }
};
template <typename T>
struct pod_vector : std::vector<T, default_init_allocator<T> > {};

struct RepertoireSet {
  /* per sequence */
  pod_vector<uint8_t>  residues;     /* codes, concatenated               */
  pod_vector<uint64_t> offsets;      /* n + 1                             */
  pod_vector<uint32_t> v_gene, j_gene, repertoire;
  pod_vector<uint64_t> count;
  std::vector<std::string> sequence_id;   /* kept only when asked for (-x, -p) */
  std::vector<std::string> keep;          /* -k columns, tab-joined (-p) */
  /* per set */
  Interner  repertoires;
  uint64_t  ignored_unknown = 0, ignored_empty = 0;
  uint64_t  total_count = 0;
  uint32_t  longest = 0, shortest = 0xffffffffu;

  uint64_t size() const { return repertoire.size(); }
  uint64_t residue_count() const { return residues.size(); }
};

/* Reads `filename` ("-" = standard input) into `out`.  Errors are reported the
   way the reference does: message on `log`, exit status 1. */
void read_airr_tsv(const char *filename, const Options &opt, GeneTables &genes,
                   const char *default_repertoire_id, FILE *log,
                   RepertoireSet &out, bool require_sequence_id = false,
                   bool keep_sequence_id = false);

}  // namespace cmprhost
#endif
