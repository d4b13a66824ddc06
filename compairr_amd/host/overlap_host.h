/*
 * overlap_host.h -- host driver of the --matrix command: read the sets, hand
 * the per-query loop to a backend, print the matrix.  Mirrors what
 * overlap() does around the loop (/root/reference/src/overlap.cc:607-1079).
 */
#ifndef COMPAIRR_AMD_OVERLAP_HOST_H
#define COMPAIRR_AMD_OVERLAP_HOST_H

#include <vector>

#include "airr_tsv.h"
#include "options.h"

namespace cmprhost {

struct BackendReport {
  double   seconds_index = 0;     /* set-2 upload + index build               */
  double   seconds_queries = 0;   /* set-1 upload + layout                    */
  double   seconds_analysis = 0;  /* the per-query loop (reference: "Analysing:") */
  double   kernel_ms = 0;
  uint64_t variants = 0, bloom_positive = 0, hash_equal = 0, matches = 0;
  uint64_t dup_set1 = 0, dup_set2 = 0;   /* exact duplicates (overlap.cc:850,872) */
  uint64_t algorithmic_bytes = 0;
  std::string device_name;
};

/* (seed, hit) pairs for -p/--pairs: indices into set 1 / set 2 */
struct PairList {
  std::vector<uint32_t> seed, hit;
};

/* The seam where the reference launches sim_thread (overlap.cc:926-936). */
class OverlapBackend {
public:
  virtual ~OverlapBackend() {}
  virtual const char *name() const = 0;
  /* Called on a thread of its own while the input files are being read: whatever of the
     backend's start-up does not depend on the data (the HIP runtime and the device context
     take ~0.4 s).  Must not fail loudly -- overlap() reports what is wrong. */
  virtual void prewarm(const Options &) {}
  /* Fills cells[R1 * R2] with the values the reference keeps in
     repertoire_matrix (doubles; row = set-1 repertoire number, column =
     set-2 repertoire number).  `same` = one-file mode (set2 is set1).
     Returns false and sets `error` on failure. */
  virtual bool overlap(const Options &opt, const GeneTables &genes,
                       const RepertoireSet &set1, const RepertoireSet &set2,
                       bool same, std::vector<double> &cells,
                       BackendReport &report, std::string &error,
                       PairList *pairs /* nullptr unless -p */) = 0;
};

/* Whole program: parse argv, run, print.  Returns the exit status. */
int compairr_main(int argc, char **argv, OverlapBackend &backend);

}  // namespace cmprhost
#endif
