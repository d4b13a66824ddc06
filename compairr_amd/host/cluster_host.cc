/*
 * cluster_host.cc -- see cluster_host.h.  Own code; the behaviour it keeps is
 * cited per block from /root/reference/src/cluster.cc and variants.cc.
 */
#include "cluster_host.h"

#include <string.h>

#include <algorithm>
#include <chrono>
#include <thread>
#include <string>
#include <tuple>
#include <vector>

namespace cmprhost {

namespace {

const uint32_t NO_CLUSTER = 0xffffffffu;   /* cluster.cc:24 */

/* Where generate_variants() (variants.cc:402-428) lists the variant of `s`
   that equals `t`: unchanged, then substitutions by (position, residue), then
   deletions by position, insertions by (position, residue) (variants.cc:270-355)
   and double substitutions by (i, v, j, w) (variants.cc:357-400).  The
   reference appends a seed's hits in that order (cluster.cc:156-161), and the
   order decides where a sequence lands inside its cluster.  Each (s, t) pair
   within the distance has exactly one listed variant: deletions are listed at
   the first position of a run of equal residues (variants.cc:301-325), and an
   inserted residue is the first of its run in the result (variants.cc:329-353). */
typedef std::tuple<uint32_t, uint32_t, uint32_t, uint32_t, uint32_t> VariantRank;

VariantRank rank_of(const uint8_t *s, uint32_t L, const uint8_t *t, uint32_t M)
{
  if (L == M) {
    uint32_t n = 0, p[2] = {0, 0};
    for (uint32_t x = 0; x < L && n < 2; x++)
      if (s[x] != t[x])
        p[n++] = x;
    if (n == 0)
      return VariantRank(0, 0, 0, 0, 0);
    if (n == 1)
      return VariantRank(1, p[0], t[p[0]], 0, 0);
    return VariantRank(4, p[0], t[p[0]], p[1], t[p[1]]);
  }
  if (M + 1 == L) {
    uint32_t x = 0;
    while (x < M && s[x] == t[x])
      x++;                                    /* x = L - 1 if t is a prefix of s */
    uint32_t pos = x;
    while (pos > 0 && s[pos - 1] == s[x])
      pos--;
    return VariantRank(2, pos, 0, 0, 0);
  }
  /* M == L + 1 */
  uint32_t x = 0;
  while (x < L && s[x] == t[x])
    x++;
  const uint32_t v = t[x];
  uint32_t pos = x;
  while (pos > 0 && t[pos - 1] == v)
    pos--;
  return VariantRank(3, pos, v, 0, 0);
}

}  // namespace

int run_cluster(const Options &o, OverlapBackend &backend, FILE *log, FILE *out)
{
  fprintf(log, "Immune receptor repertoire clustering\n\n");       /* cluster.cc:227 */

  GeneTables genes;
  RepertoireSet set;
  /* (the backend's data-independent start-up runs while the file is read: overlap_host.h prewarm) */
  reader_thread_active(true);          /* (an error exit meanwhile leaves through _exit: options.h) */
  std::thread warm([&]() { backend.prewarm(o); });
  auto t0 = std::chrono::steady_clock::now();
  read_airr_tsv(o.input1, o, genes, "1", log, set, false, true);   /* cluster.cc:232 */
  warm.join();
  reader_thread_active(false);
  fprintf(log, "Reading sequences: %.9lfs\n",
          std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  const uint64_t n = set.size();
  fprintf(log, "\n");
  fprintf(log, "Unique V genes:    %lu\n", (unsigned long)genes.v.names.size());
  fprintf(log, "Unique J genes:    %lu\n", (unsigned long)genes.j.names.size());
  fprintf(log, "\n");

  /* ---- building the network (cluster.cc:225-274): every (seed, hit != seed)
          within the distance, from the backend's pairs mode ---- */
  Options bo = o;
  bo.matrix = true;
  bo.cluster = false;
  bo.ignore_counts = true;        /* scores are not used */
  bo.score = SCORE_PRODUCT;
  const size_t R = set.repertoires.names.size();
  std::vector<double> cells(R * R, 0.0);
  BackendReport rep;
  std::string error;
  PairList pairs;
  if (n > 0 && !backend.overlap(bo, genes, set, set, true, cells, rep, error, &pairs)) {
    fprintf(stderr, "\nError: %s\n", error.c_str());
    return 1;
  }
  fprintf(log, "Hashing sequences: 100%% (%.9lfs)\n", rep.seconds_index);
  fprintf(log, "Building network:  100%% (%.9lfs)\n", rep.seconds_queries + rep.seconds_analysis);
  if (!rep.device_name.empty())
    fprintf(log, "GPU kernel:        %.3f ms, %lu variants, %lu pairs\n", rep.kernel_ms,
            (unsigned long)rep.variants, (unsigned long)rep.matches);

  /* adjacency lists in the reference's order: by listed variant, then by hit
     number (equal sequences sit in the probe chain in insertion order,
     cluster.cc:60-73, 94-135) */
  t0 = std::chrono::steady_clock::now();
  std::vector<uint64_t> start(n + 1, 0);
  for (size_t k = 0; k < pairs.seed.size(); k++)
    if (pairs.seed[k] != pairs.hit[k])
      start[pairs.seed[k] + 1]++;
  for (uint64_t i = 0; i < n; i++)
    start[i + 1] += start[i];
  std::vector<uint32_t> network(start[n]);
  {
    std::vector<uint64_t> fill(start.begin(), start.end() - 1);
    for (size_t k = 0; k < pairs.seed.size(); k++)
      if (pairs.seed[k] != pairs.hit[k])
        network[fill[pairs.seed[k]]++] = pairs.hit[k];
  }
  pairs.seed.clear();
  pairs.seed.shrink_to_fit();
  pairs.hit.clear();
  pairs.hit.shrink_to_fit();
  {
    std::vector<std::pair<VariantRank, uint32_t> > keyed;
    for (uint64_t seed = 0; seed < n; seed++) {
      const uint64_t a = start[seed], b = start[seed + 1];
      if (b - a < 2)
        continue;
      const uint8_t *s = set.residues.data() + set.offsets[seed];
      const uint32_t L = (uint32_t)(set.offsets[seed + 1] - set.offsets[seed]);
      keyed.clear();
      for (uint64_t k = a; k < b; k++) {
        const uint32_t hit = network[k];
        keyed.push_back(std::make_pair(
            rank_of(s, L, set.residues.data() + set.offsets[hit],
                    (uint32_t)(set.offsets[hit + 1] - set.offsets[hit])),
            hit));
      }
      std::sort(keyed.begin(), keyed.end());
      for (uint64_t k = a; k < b; k++)
        network[k] = keyed[k - a].second;
    }
  }

  /* ---- clustering (cluster.cc:200-223, 276-410): breadth-first sweep; the
          members of a cluster form a chain in the order they were reached ---- */
  std::vector<uint32_t> clusterid(n, NO_CLUSTER), next(n, NO_CLUSTER);
  struct ClusterInfo { uint32_t seed, size; };
  std::vector<ClusterInfo> clusters;
  for (uint64_t seed = 0; seed < n; seed++) {
    if (clusterid[seed] != NO_CLUSTER)
      continue;
    const uint32_t id = (uint32_t)clusters.size();
    clusterid[seed] = id;
    uint32_t tail = (uint32_t)seed, size = 0;
    for (uint32_t cur = (uint32_t)seed; cur != NO_CLUSTER; cur = next[cur]) {
      size++;
      for (uint64_t k = start[cur]; k < start[cur + 1]; k++) {
        const uint32_t hit = network[k];
        if (clusterid[hit] == NO_CLUSTER) {
          clusterid[hit] = id;
          next[tail] = hit;
          tail = hit;
        }
      }
    }
    clusters.push_back(ClusterInfo{(uint32_t)seed, size});
  }
  /* largest first (cluster.cc:50-58, 414); glibc's qsort is a merge sort, so
     equal sizes stay in the order the clusters were found */
  std::stable_sort(clusters.begin(), clusters.end(),
                   [](const ClusterInfo &x, const ClusterInfo &y) { return x.size > y.size; });
  fprintf(log, "Clustering:        100%% (%.9lfs)\n",
          std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());

  /* ---- output (cluster.cc:419-452) ---- */
  const char *letters = o.nucleotides ? "acgt" : "ACDEFGHIKLMNPQRSTVWY";   /* db.cc:73-74 */
  fprintf(out, "#cluster_no\tcluster_size\trepertoire_id\tsequence_id\t"
               "duplicate_count\tv_call\tj_call\t%s\n", o.seq_header);
  for (size_t i = 0; i < clusters.size(); i++)
    for (uint32_t a = clusters[i].seed; a != NO_CLUSTER; a = next[a]) {
      fprintf(out, "%u\t%u\t", (unsigned)(i + 1), clusters[i].size);
      fprintf(out, "%s\t%s\t%lu\t%s\t%s\t", set.repertoires.names[set.repertoire[a]].c_str(),
              a < set.sequence_id.size() ? set.sequence_id[a].c_str() : "",
              (unsigned long)set.count[a], genes.v.names[set.v_gene[a]].c_str(),
              genes.j.names[set.j_gene[a]].c_str());
      for (uint64_t p = set.offsets[a]; p < set.offsets[a + 1]; p++)
        fputc(letters[set.residues[p]], out);
      fputc('\n', out);
    }
  fprintf(log, "\n");
  fprintf(log, "Clusters:          %u\n", (unsigned)clusters.size());
  return 0;
}

}  // namespace cmprhost
