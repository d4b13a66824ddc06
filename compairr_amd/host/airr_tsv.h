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

struct RepertoireSet {
  /* per sequence */
  std::vector<uint8_t>  residues;     /* codes, concatenated               */
  std::vector<uint64_t> offsets;      /* n + 1                             */
  std::vector<uint32_t> v_gene, j_gene, repertoire;
  std::vector<uint64_t> count;
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
