/*
 * oracle_cli_main.cc -- TEST BINARY (tests/bin/compairr_oracle_cli), never
 * shipped: the product's host code (option parsing, AIRR-TSV reader, matrix
 * printer) with the per-query loop served by the CPU oracle instead of the
 * GPU.  It lets the CPU test suite check the host logic and the oracle
 * against the golden vectors without a GPU.
 */
#include <string.h>

#include "compairr_oracle.h"
#include "overlap_host.h"

using namespace cmprhost;

namespace {

oracle_set view_of(const RepertoireSet &s)
{
  oracle_set v;
  memset(&v, 0, sizeof v);
  v.n = s.size();
  v.residues = s.residues.data();
  v.offsets = s.offsets.data();
  v.v_gene = s.v_gene.data();
  v.j_gene = s.j_gene.data();
  v.repertoire = s.repertoire.data();
  v.count = s.count.data();
  v.n_repertoires = (uint32_t)s.repertoires.names.size();
  return v;
}

class OracleBackend : public OverlapBackend {
public:
  const char *name() const override { return "CPU oracle (test only)"; }
  bool overlap(const Options &o, const GeneTables &genes, const RepertoireSet &set1,
               const RepertoireSet &set2, bool same, std::vector<double> &cells,
               BackendReport &rep, std::string &error, PairList *pairs) override
  {
    oracle_opts oo;
    memset(&oo, 0, sizeof oo);
    oo.differences = (int32_t)o.differences;
    oo.indels = o.indels;
    oo.ignore_genes = o.ignore_genes;
    oo.ignore_counts = o.ignore_counts;
    oo.score = (int32_t)o.score;
    oo.alphabet_size = o.alphabet_size;
    oo.threads = (int32_t)o.threads;
    oo.n_v_genes = (uint32_t)genes.v.names.size();
    oo.n_j_genes = (uint32_t)genes.j.names.size();
    oo.existence = o.existence;
    const oracle_set v1 = view_of(set1);
    const oracle_set v2 = view_of(set2);
    oracle_stats st;
    memset(&st, 0, sizeof st);
    if (oracle_overlap(&oo, &v1, same ? &v1 : &v2, cells.data(), &st)) {
      error = "oracle_overlap failed";
      return false;
    }
    rep.seconds_index = st.seconds_index;
    rep.seconds_analysis = st.seconds_analysis;
    rep.variants = st.variants;
    rep.bloom_positive = st.bloom_positive;
    rep.hash_equal = st.hash_equal;
    rep.matches = st.matches;
    rep.dup_set1 = st.dup_set1;
    rep.dup_set2 = st.dup_set2;
    if (pairs) {
      uint64_t n = 0;
      pairs->seed.resize(st.matches);
      pairs->hit.resize(st.matches);
      if (oracle_pairs(&oo, &v1, same ? &v1 : &v2, st.matches, pairs->seed.data(),
                       pairs->hit.data(), &n) || n != st.matches) {
        error = "oracle_pairs failed";
        return false;
      }
    }
    return true;
  }
};

}  // namespace

int main(int argc, char **argv)
{
  OracleBackend backend;
  return compairr_main(argc, argv, backend);
}
