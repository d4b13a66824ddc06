/*
 * hip_backend.cc -- see hip_backend.h.
 */
#include "hip_backend.h"

#include <dlfcn.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <chrono>

#include "compairr_hip.h"

namespace cmprhost {

namespace {

struct Api {
  void *handle = nullptr;
  int (*abi_version)(void) = nullptr;
  int (*create)(const cmpr_options *, cmpr_context **) = nullptr;
  void (*destroy)(cmpr_context *) = nullptr;
  const char *(*last_error)(const cmpr_context *) = nullptr;
  int (*set_reference)(cmpr_context *, const cmpr_set_view *, uint32_t) = nullptr;
  int (*set_queries)(cmpr_context *, const cmpr_set_view *) = nullptr;
  int (*overlap_matrix)(cmpr_context *, uint64_t *) = nullptr;
  int (*overlap_matrix_f64)(cmpr_context *, double *) = nullptr;
  int (*get_stats)(cmpr_context *, cmpr_stats *) = nullptr;
  int (*count_duplicates)(cmpr_context *, const cmpr_set_view *, uint64_t *) = nullptr;
  int (*overlap_pairs)(cmpr_context *, uint64_t, uint32_t *, uint32_t *, uint64_t *) = nullptr;
};

template <typename F>
bool bind(void *h, const char *name, F &fn, std::string &error)
{
  fn = reinterpret_cast<F>(dlsym(h, name));
  if (!fn) {
    error = std::string("libcompairr_hip.so lacks symbol ") + name;
    return false;
  }
  return true;
}

cmpr_set_view view_of(const RepertoireSet &s)
{
  cmpr_set_view v;
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

double since(std::chrono::steady_clock::time_point t0)
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

class HipBackend : public OverlapBackend {
public:
  explicit HipBackend(const Api &api) : api_(api) {}
  ~HipBackend() override
  {
    if (api_.handle)
      dlclose(api_.handle);
  }
  const char *name() const override { return "HIP gfx950 (libcompairr_hip.so)"; }

  bool overlap(const Options &o, const GeneTables &genes, const RepertoireSet &set1,
               const RepertoireSet &set2, bool same, std::vector<double> &cells,
               BackendReport &rep, std::string &error, PairList *pairs) override
  {
    pairs_ = pairs;
    same_ = same;
    cmpr_options co;
    memset(&co, 0, sizeof co);
    co.differences = (int32_t)std::min<int64_t>(o.differences, INT32_MAX);
    co.indels = o.indels;
    co.ignore_genes = o.ignore_genes;
    co.ignore_counts = o.ignore_counts;
    co.score = (int32_t)o.score;
    co.alphabet_size = o.alphabet_size;
    co.n_v_genes = (uint32_t)genes.v.names.size();
    co.n_j_genes = (uint32_t)genes.j.names.size();
    co.device = (int32_t)o.device;
    co.existence = o.existence;

    cmpr_context *ctx = nullptr;
    if (api_.create(&co, &ctx)) {
      error = api_.last_error(nullptr);
      return false;
    }
    bool ok = run(ctx, set1, set2, cells, rep, error);
    api_.destroy(ctx);
    return ok;
  }

private:
  bool run(cmpr_context *ctx, const RepertoireSet &set1, const RepertoireSet &set2,
           std::vector<double> &cells, BackendReport &rep, std::string &error)
  {
    const cmpr_set_view v1 = view_of(set1), v2 = view_of(set2);
    auto t0 = std::chrono::steady_clock::now();
    if (api_.set_reference(ctx, &v2, set1.longest)) {
      error = api_.last_error(ctx);
      return false;
    }
    /* duplicate warnings: set 2 from the resident index; set 1 only when it is a
       different file (check_duplicates, overlap.cc:846-851) */
    uint64_t dups = 0;
    if (api_.count_duplicates(ctx, nullptr, &dups)) {
      error = api_.last_error(ctx);
      return false;
    }
    rep.dup_set2 = dups;
    if (!same_) {
      if (api_.count_duplicates(ctx, &v1, &dups)) {
        error = api_.last_error(ctx);
        return false;
      }
      rep.dup_set1 = dups;
    }
    rep.seconds_index = since(t0);
    t0 = std::chrono::steady_clock::now();
    if (api_.set_queries(ctx, &v1)) {
      error = api_.last_error(ctx);
      return false;
    }
    rep.seconds_queries = since(t0);
    t0 = std::chrono::steady_clock::now();
    if (api_.overlap_matrix_f64(ctx, cells.data())) {
      error = api_.last_error(ctx);
      return false;
    }
    rep.seconds_analysis = since(t0);
    cmpr_stats st;
    if (api_.get_stats(ctx, &st) == 0) {
      rep.kernel_ms = st.kernel_ms;
      rep.variants = st.variants;
      rep.bloom_positive = st.bloom_positive;
      rep.hash_equal = st.hash_equal;
      rep.matches = st.matches;
      rep.algorithmic_bytes = st.algorithmic_bytes;
    }
    rep.device_name = "HIP device";
    if (pairs_) {
      /* the matrix pass counted the pairs exactly; list them with a second pass */
      uint64_t n = rep.matches, got = 0;
      pairs_->seed.resize(n);
      pairs_->hit.resize(n);
      if (api_.overlap_pairs(ctx, n, pairs_->seed.data(), pairs_->hit.data(), &got)) {
        error = api_.last_error(ctx);
        return false;
      }
      if (got != n) {
        error = "pair count changed between passes";
        return false;
      }
    }
    return true;
  }

  Api api_;
  bool same_ = false;
  PairList *pairs_ = nullptr;
};

}  // namespace

OverlapBackend *make_hip_backend(const char *argv0, std::string &error)
{
  std::vector<std::string> candidates;
  if (const char *env = getenv("COMPAIRR_HIP_LIB"))
    candidates.push_back(env);
  char exe[PATH_MAX];
  ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
  if (n > 0) {
    exe[n] = 0;
    std::string dir(exe);
    dir = dir.substr(0, dir.find_last_of('/'));
    candidates.push_back(dir + "/../compairr_amd/lib/libcompairr_hip.so");
    candidates.push_back(dir + "/libcompairr_hip.so");
  }
  (void)argv0;
  candidates.push_back("libcompairr_hip.so");

  Api api;
  std::string tried;
  for (const std::string &path : candidates) {
    api.handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (api.handle)
      break;
    tried += "\n  " + path + ": " + dlerror();
  }
  if (!api.handle) {
    error = "Unable to load the HIP library libcompairr_hip.so (no CPU fallback exists):" + tried;
    return nullptr;
  }
  if (!bind(api.handle, "cmpr_abi_version", api.abi_version, error) ||
      !bind(api.handle, "cmpr_create", api.create, error) ||
      !bind(api.handle, "cmpr_destroy", api.destroy, error) ||
      !bind(api.handle, "cmpr_last_error", api.last_error, error) ||
      !bind(api.handle, "cmpr_set_reference", api.set_reference, error) ||
      !bind(api.handle, "cmpr_set_queries", api.set_queries, error) ||
      !bind(api.handle, "cmpr_overlap_matrix", api.overlap_matrix, error) ||
      !bind(api.handle, "cmpr_overlap_matrix_f64", api.overlap_matrix_f64, error) ||
      !bind(api.handle, "cmpr_get_stats", api.get_stats, error) ||
      !bind(api.handle, "cmpr_count_duplicates", api.count_duplicates, error) ||
      !bind(api.handle, "cmpr_overlap_pairs", api.overlap_pairs, error)) {
    dlclose(api.handle);
    return nullptr;
  }
  if (api.abi_version() != CMPR_ABI_VERSION) {
    error = "libcompairr_hip.so has a different ABI version";
    dlclose(api.handle);
    return nullptr;
  }
  return new HipBackend(api);
}

}  // namespace cmprhost
