/*
 * hip_backend.cc -- see hip_backend.h.
 */
#include "hip_backend.h"

#include <dlfcn.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <thread>

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
  int (*warm_up)(const cmpr_options *) = nullptr;
  int (*warm_up_sized)(const cmpr_options *, uint64_t, uint64_t, uint64_t) = nullptr;
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

  void prewarm(const Options &o) override
  {
    cmpr_options co;
    memset(&co, 0, sizeof co);
    co.differences = (int32_t)std::min<int64_t>(o.differences, INT32_MAX);
    co.indels = o.indels;
    co.ignore_genes = o.ignore_genes;
    co.alphabet_size = o.alphabet_size;
    if (o.devices.empty()) {
      co.device = (int32_t)o.device;
      /* how large the query set will be, from the size of its file (the program knows it before it parses a
         line): an AIRR line of the columns this program reads is ~35 bytes and up, ~15 of them residues --
         a hint: the library reserves its page-locked upload buffer and layout arena meanwhile (ABI v5) */
      uint64_t nq = 0, nres = 0;
      struct stat st;
      if (o.input1 && strcmp(o.input1, "-") != 0 && stat(o.input1, &st) == 0 && S_ISREG(st.st_mode)) {
        nq = (uint64_t)st.st_size / 34;
        nres = (uint64_t)st.st_size / 2;
      }
      (void)api_.warm_up_sized(&co, nq, 0, nres);
      return;
    }
    /* (--devices: one thread per device, as overlap() then creates their contexts side by side -- warmed one
       after the other, eight devices and a small input would wait for the start-up they were to hide) */
    std::vector<std::thread> th;
    for (size_t g = 0; g < o.devices.size(); g++) {
      cmpr_options cg = co;
      cg.device = o.devices[g];
      th.emplace_back([this, cg]() { (void)api_.warm_up(&cg); });
    }
    for (auto &t : th)
      t.join();
  }

  bool overlap(const Options &o, const GeneTables &genes, const RepertoireSet &set1,
               const RepertoireSet &set2, bool same, std::vector<double> &cells,
               BackendReport &rep, std::string &error, PairList *pairs) override
  {
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

    const cmpr_set_view v1 = view_of(set1), v2 = view_of(set2);
    if (o.devices.size() <= 1) {
      if (o.devices.size() == 1)
        co.device = o.devices[0];
      Shard sh;
      sh.first = 0;
      sh.view = v1;
      sh.cells = &cells;
      sh.pairs = pairs;
      return run(co, sh, v1, v2, set1.longest, same, true, rep, error);
    }

    /* ---- several GPUs: the queries are cut into one contiguous
            shard per listed device, every device builds the whole set-2 index,
            one host thread drives each; the matrices are integer-valued sums
            (exact in double below 2^53), so adding them gives the same cells as
            one device.  -x rows belong to the queries: they are placed, not
            added.  Pairs are concatenated. ---- */
    const size_t G = o.devices.size();
    const uint64_t n = set1.size();
    const size_t R2 = set2.repertoires.names.size();
    std::vector<Shard> shards(G);
    std::vector<std::vector<double> > shard_cells(G);
    std::vector<std::vector<uint64_t> > shard_offsets(G);
    std::vector<PairList> shard_pairs(G);
    std::vector<BackendReport> reports(G);
    std::vector<std::string> errors(G);
    std::vector<char> ok(G, 0);
    /* balance by residues: variants per query grow with the length */
    const uint64_t total_res = set1.residue_count();
    uint64_t at = 0;
    for (size_t g = 0; g < G; g++) {
      uint64_t end = n;
      if (g + 1 < G) {
        const uint64_t want = total_res / G * (g + 1);
        end = (uint64_t)(std::upper_bound(set1.offsets.begin(), set1.offsets.end(), want) -
                         set1.offsets.begin());
        end = std::min<uint64_t>(std::max<uint64_t>(end, at), n);
      }
      Shard &sh = shards[g];
      sh.first = at;
      shard_offsets[g].resize(end - at + 1);
      for (uint64_t i = at; i <= end; i++)
        shard_offsets[g][i - at] = set1.offsets[i] - set1.offsets[at];
      sh.view = v1;
      sh.view.n = end - at;
      sh.view.residues = v1.residues + set1.offsets[at];
      sh.view.offsets = shard_offsets[g].data();
      sh.view.v_gene = v1.v_gene + at;
      sh.view.j_gene = v1.j_gene + at;
      sh.view.repertoire = v1.repertoire + at;
      sh.view.count = v1.count + at;
      shard_cells[g].assign(o.existence ? (size_t)(end - at) * R2 : cells.size(), 0.0);
      sh.cells = &shard_cells[g];
      sh.pairs = pairs ? &shard_pairs[g] : nullptr;
      at = end;
    }
    /* empty shards (fewer queries than devices) have nothing to do; the first
       shard that runs also counts the duplicates */
    size_t first_busy = G;
    for (size_t g = 0; g < G; g++)
      if (shards[g].view.n > 0 && first_busy == G)
        first_busy = g;
    if (first_busy == G)
      first_busy = 0;                      /* empty set 1: shard 0 reports the index only */
    std::vector<std::thread> threads;
    for (size_t g = 0; g < G; g++) {
      if (shards[g].view.n == 0 && g != first_busy) {
        ok[g] = 1;
        continue;
      }
      threads.emplace_back([&, g]() {
        cmpr_options cg = co;
        cg.device = o.devices[g];
        ok[g] = run(cg, shards[g], v1, v2, set1.longest, same, g == first_busy, reports[g],
                    errors[g]);
      });
    }
    for (std::thread &t : threads)
      t.join();
    for (size_t g = 0; g < G; g++)
      if (!ok[g]) {
        error = "device " + std::to_string(o.devices[g]) + ": " + errors[g];
        return false;
      }
    rep = reports[first_busy];
    for (size_t g = 0; g < G; g++) {
      if (o.existence)
        std::copy(shard_cells[g].begin(), shard_cells[g].end(),
                  cells.begin() + (size_t)shards[g].first * R2);
      else
        for (size_t k = 0; k < cells.size(); k++)
          cells[k] += shard_cells[g][k];
      if (g != first_busy) {
        rep.seconds_index = std::max(rep.seconds_index, reports[g].seconds_index);
        rep.seconds_queries = std::max(rep.seconds_queries, reports[g].seconds_queries);
        rep.seconds_analysis = std::max(rep.seconds_analysis, reports[g].seconds_analysis);
        rep.kernel_ms = std::max(rep.kernel_ms, reports[g].kernel_ms);
        rep.variants += reports[g].variants;
        rep.bloom_positive += reports[g].bloom_positive;
        rep.hash_equal += reports[g].hash_equal;
        rep.matches += reports[g].matches;
        rep.algorithmic_bytes += reports[g].algorithmic_bytes;
      }
      if (pairs) {
        for (size_t k = 0; k < shard_pairs[g].seed.size(); k++) {
          pairs->seed.push_back(shard_pairs[g].seed[k] + (uint32_t)shards[g].first);
          pairs->hit.push_back(shard_pairs[g].hit[k]);
        }
      }
    }
    rep.device_name = std::to_string(G) + " HIP devices";
    return true;
  }

private:
  /* the part of set 1 one device works on */
  struct Shard {
    uint64_t first = 0;                 /* number of its first query in set 1 */
    cmpr_set_view view;
    std::vector<double> *cells = nullptr;
    PairList *pairs = nullptr;
  };

  bool run(const cmpr_options &co, const Shard &sh, const cmpr_set_view &v1_all,
           const cmpr_set_view &v2, uint32_t longest_query, bool same, bool count_dups,
           BackendReport &rep, std::string &error) const
  {
    cmpr_context *ctx = nullptr;
    if (api_.create(&co, &ctx)) {
      error = api_.last_error(nullptr);
      return false;
    }
    const bool ok = run_on(ctx, sh, v1_all, v2, longest_query, same, count_dups, rep, error);
    api_.destroy(ctx);
    return ok;
  }

  bool run_on(cmpr_context *ctx, const Shard &sh, const cmpr_set_view &v1_all,
              const cmpr_set_view &v2, uint32_t longest_query, bool same, bool count_dups,
              BackendReport &rep, std::string &error) const
  {
    auto t0 = std::chrono::steady_clock::now();
    if (api_.set_reference(ctx, &v2, longest_query)) {
      error = api_.last_error(ctx);
      return false;
    }
    /* duplicate warnings: set 2 from the resident index; set 1 only when it is a
       different file (check_duplicates, overlap.cc:846-851) */
    if (count_dups) {
      uint64_t dups = 0;
      if (api_.count_duplicates(ctx, nullptr, &dups)) {
        error = api_.last_error(ctx);
        return false;
      }
      rep.dup_set2 = dups;
      if (!same) {
        if (api_.count_duplicates(ctx, &v1_all, &dups)) {
          error = api_.last_error(ctx);
          return false;
        }
        rep.dup_set1 = dups;
      }
    }
    rep.seconds_index = since(t0);
    t0 = std::chrono::steady_clock::now();
    if (api_.set_queries(ctx, &sh.view)) {
      error = api_.last_error(ctx);
      return false;
    }
    rep.seconds_queries = since(t0);
    t0 = std::chrono::steady_clock::now();
    if (api_.overlap_matrix_f64(ctx, sh.cells->data())) {
      error = api_.last_error(ctx);
      return false;
    }
    rep.seconds_analysis = since(t0);
    /* (the pairs pass below sizes its arrays from these counters, and a failing
       cmpr_get_stats means the matrix itself is not to be trusted: fatal either way) */
    cmpr_stats st;
    if (api_.get_stats(ctx, &st)) {
      error = api_.last_error(ctx);
      return false;
    }
    rep.kernel_ms = st.kernel_ms;
    rep.variants = st.variants;
    rep.bloom_positive = st.bloom_positive;
    rep.hash_equal = st.hash_equal;
    rep.matches = st.matches;
    rep.algorithmic_bytes = st.algorithmic_bytes;
    rep.device_name = "HIP device";
    if (sh.pairs) {
      /* the matrix pass counted the pairs exactly; list them with a second pass */
      uint64_t n = rep.matches, got = 0;
      sh.pairs->seed.resize(n);
      sh.pairs->hit.resize(n);
      if (api_.overlap_pairs(ctx, n, sh.pairs->seed.data(), sh.pairs->hit.data(), &got)) {
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
      !bind(api.handle, "cmpr_overlap_pairs", api.overlap_pairs, error) ||
      !bind(api.handle, "cmpr_warm_up", api.warm_up, error) ||
      !bind(api.handle, "cmpr_warm_up_sized", api.warm_up_sized, error)) {
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
