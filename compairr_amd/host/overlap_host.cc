/*
 * overlap_host.cc -- see overlap_host.h.  Own code; the behaviour it keeps is
 * cited per block from /root/reference/src/overlap.cc and compairr.cc.
 */
#include "overlap_host.h"

#include "cluster_host.h"

#include <math.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <thread>

namespace cmprhost {

namespace {

FILE *open_output(const char *name)
{
  /* "-" = standard output (util.cc:156-170) */
  if (strcmp(name, "-") == 0) {
    int fd = dup(STDOUT_FILENO);
    return fd < 0 ? nullptr : fdopen(fd, "w");
  }
  return fopen(name, "w");
}

void log_time(FILE *log, const char *prompt)
{
  char buf[100];
  time_t now = time(nullptr);
  size_t n = strftime(buf, sizeof buf, "%a %b %d %T %Z %Y", localtime(&now));
  fprintf(log, "%s%s\n", prompt, n ? buf : "?");
}

struct RepertoireTotals {
  std::vector<uint64_t> size, count;
  std::vector<double>   sq_count;
  std::vector<uint32_t> order;      /* display order: strcmp on the ids */
};

/* overlap.cc:627-667 */
void totals_of(const RepertoireSet &s, RepertoireTotals &t)
{
  const size_t R = s.repertoires.names.size();
  t.size.assign(R, 0);
  t.count.assign(R, 0);
  t.sq_count.assign(R, 0.0);
  for (uint64_t i = 0; i < s.size(); i++) {
    const uint32_t r = s.repertoire[i];
    const uint64_t c = s.count[i];
    t.size[r]++;
    t.count[r] += c;
    t.sq_count[r] += (double)(uint64_t)(c * c);
  }
  t.order.resize(R);
  for (uint32_t i = 0; i < R; i++)
    t.order[i] = i;
  const std::vector<std::string> &names = s.repertoires.names;
  std::sort(t.order.begin(), t.order.end(), [&names](uint32_t a, uint32_t b) {
    return strcmp(names[a].c_str(), names[b].c_str()) < 0;
  });
}

/* overlap.cc:669-697 */
void log_repertoires(FILE *log, const RepertoireSet &s, const RepertoireTotals &t)
{
  const size_t R = t.order.size();
  uint64_t sum_size = 0, sum_count = 0;
  for (size_t i = 0; i < R; i++) {
    sum_size += t.size[i];
    sum_count += t.count[i];
  }
  int w1 = std::max(1, (int)(1 + floor(log10((double)R))));
  int w2 = std::max(9, (int)(1 + floor(log10((double)sum_size))));
  int w3 = std::max(5, (int)(1 + floor(log10((double)sum_count))));
  if (R == 0) w1 = 1;
  if (sum_size == 0) w2 = 9;
  if (sum_count == 0) w3 = 5;
  fprintf(log, "Repertoires in set:\n");
  fprintf(log, "%*s %*s %*s %s\n", w1, "#", w2, "Sequences", w3, "Count", "Repertoire ID");
  for (size_t i = 0; i < R; i++) {
    const uint32_t r = t.order[i];
    fprintf(log, "%*u %*lu %*lu %s\n", w1, (unsigned)(i + 1), w2,
            (unsigned long)t.size[r], w3, (unsigned long)t.count[r],
            s.repertoires.names[r].c_str());
  }
  fprintf(log, "\n");
}

/* show_matrix_value, overlap.cc:540-577 */
double cell_value(const Options &o, double cell, const RepertoireTotals &t1,
                  uint32_t s, const RepertoireTotals &t2, uint32_t t)
{
  if (o.score == SCORE_MH) {
    double lx = t1.sq_count[s] / t1.count[s] / t1.count[s];
    double ly = t2.sq_count[t] / t2.count[t] / t2.count[t];
    double xy = 1.0 * t1.count[s] * t2.count[t];
    return (2.0 * cell) / ((lx + ly) * xy);
  }
  if (o.score == SCORE_JACCARD) {
    double sa = (double)t1.count[s], sb = (double)t2.count[t];
    return cell / (sa + sb - cell);
  }
  return cell;
}

}  // namespace

/* COMPAIRR_HOST_TIMING=1: where the host program's wall-clock time goes, on stderr */
static void host_mark(const char *what)
{
  static const bool on = getenv("COMPAIRR_HOST_TIMING") != nullptr;
  static const auto t_first = std::chrono::steady_clock::now();
  if (on)
    fprintf(stderr, "[host %8.3f ms] %s\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_first).count(), what);
}

int compairr_main(int argc, char **argv, OverlapBackend &backend)
{
  host_mark("main");
  Options o;
  parse_command_line(argc, argv, o);

  FILE *log = stderr;
  if (o.log) {
    log = fopen(o.log, "w");
    if (!log)
      fatal("Unable to open log file for writing.");
  }
  FILE *out = open_output(o.output);
  if (!out)
    fatal("Unable to open output file for writing.");
  FILE *pairsfile = nullptr;
  if (o.pairs) {
    pairsfile = open_output(o.pairs);
    if (!pairsfile)
      fatal("Unable to open pairs file for writing.");
  }

  if (o.version || o.help) {
    print_header(log);
    if (o.help)
      print_usage(stderr);
    fclose(out);
    if (log != stderr)
      fclose(log);
    return 0;
  }

  print_header(log);
  log_time(log, "Start time:        ");
  print_options(log, o, backend.name());
  fprintf(log, "\n");

  if (o.cluster) {
    /* compairr.cc:776-777 */
    const int rc = run_cluster(o, backend, log, out);
    if (rc == 0) {
      fprintf(log, "\n");
      log_time(log, "End time:          ");
    }
    fclose(out);
    if (log != stderr)
      fclose(log);
    return rc;
  }

  /* ---- read (overlap.cc:611-825) ----
     The reference reads file 1, then file 2, then starts its threads.  Here three things run side by
     side: file 1 on this thread, file 2 on a thread of its own (its log lines are kept and written
     where the reference writes them; an error in it is acted on only after file 1 has been read
     without one -- options.h DeferredExit), and the backend's data-independent start-up. */
  GeneTables genes;
  RepertoireSet set1, set2_storage;
  RepertoireTotals tot1, tot2_storage;
  const bool same = !(o.input2 && strcmp(o.input1, o.input2));

  /* (while a helper thread runs, an error exit leaves through _exit: no static destructor of the HIP runtime
     under the feet of a thread that is just starting it -- options.h reader_thread_active) */
  reader_thread_active(true);
  std::thread warm([&]() { backend.prewarm(o); });
  struct Joiner {
    std::thread &t;
    ~Joiner() { if (t.joinable()) { t.join(); reader_thread_active(false); } }
  } warm_joiner{warm};

  GeneTables genes2;                      /* file 2's genes in ITS first-appearance order, merged below */
  DeferredExit exit2;
  char *log2_text = nullptr;
  size_t log2_len = 0;
  FILE *log2 = nullptr;
  double seconds2 = 0;
  std::thread reader2;
  /* (standard input can be read once, by one reader; it is file 1's when both name it) */
  const bool concurrent = !same && strcmp(o.input2, "-") != 0 && (log2 = open_memstream(&log2_text, &log2_len)) != nullptr;
  if (concurrent) {
    reader_thread_active(true);
    reader2 = std::thread([&]() {
      g_deferred_exit = &exit2;
      const auto r0 = std::chrono::steady_clock::now();
      try {
        read_airr_tsv(o.input2, o, genes2, "2", log2, set2_storage, false, o.pairs != nullptr);
      } catch (const ReaderAbort &) {
      }
      seconds2 = std::chrono::duration<double>(std::chrono::steady_clock::now() - r0).count();
      g_deferred_exit = nullptr;
    });
  }

  fprintf(log, "Immune receptor repertoire set 1\n\n");
  auto t0 = std::chrono::steady_clock::now();
  read_airr_tsv(o.input1, o, genes, "1", log, set1, o.existence, o.pairs != nullptr);
  auto t1 = std::chrono::steady_clock::now();
  fprintf(log, "Reading sequences: %.9lfs\n\n",
          std::chrono::duration<double>(t1 - t0).count());
  host_mark("file 1 read");
  totals_of(set1, tot1);
  log_repertoires(log, set1, tot1);
  host_mark("file 1 totals");

  if (o.existence && set1.repertoires.names.size() > 1)
    fatal("Multiple repertoires are not allowed in the first file specified on the command "
          "line with the -x or --existence command.");      /* overlap.cc:699-703 */

  fprintf(log, "Immune receptor repertoire set 2\n\n");
  if (!same) {
    if (concurrent) {
      reader2.join();
      reader_thread_active(false);
      fclose(log2);
      if (log2_text)
        fwrite(log2_text, 1, log2_len, log);
      free(log2_text);
      if (exit2.failed)
        /* (the HIP start-up may still be running on `warm`: exit_with_message leaves through _exit while a
           helper thread is active -- no static destructor of the runtime under a thread that initialises it;
           ADVICE r5) */
        exit_with_message(stderr, false, exit2.text);
      /* file 2's genes join file 1's in the order they first appeared (db.cc:121-125) */
      std::vector<uint32_t> mv(genes2.v.names.size()), mj(genes2.j.names.size());
      for (size_t k = 0; k < mv.size(); k++)
        mv[k] = genes.v.intern(genes2.v.names[k].c_str());
      for (size_t k = 0; k < mj.size(); k++)
        mj[k] = genes.j.intern(genes2.j.names[k].c_str());
      for (uint32_t &x : set2_storage.v_gene)
        x = mv[x];
      for (uint32_t &x : set2_storage.j_gene)
        x = mj[x];
    } else {
      t0 = std::chrono::steady_clock::now();
      read_airr_tsv(o.input2, o, genes, "2", log, set2_storage, false, o.pairs != nullptr);
      t1 = std::chrono::steady_clock::now();
      seconds2 = std::chrono::duration<double>(t1 - t0).count();
    }
    fprintf(log, "Reading sequences: %.9lfs\n\n", seconds2);
    totals_of(set2_storage, tot2_storage);
    if (set2_storage.repertoires.names.empty())
      fatal("Repertoire set missing repertoire_id.");
    log_repertoires(log, set2_storage, tot2_storage);
  } else {
    fprintf(log, "Set 2 is identical to set 1\n\n");
    if (set1.repertoires.names.empty())
      fatal("Repertoire set is missing repertoire_id.");
  }
  const RepertoireSet &set2 = same ? set1 : set2_storage;
  const RepertoireTotals &tot2 = same ? tot1 : tot2_storage;

  host_mark("both sets ready");
  fprintf(log, "Unique V genes:    %lu\n", (unsigned long)genes.v.names.size());
  fprintf(log, "Unique J genes:    %lu\n", (unsigned long)genes.j.names.size());

  /* ---- the per-query loop, on the backend (overlap.cc:840-938) ---- */
  const size_t R2 = set2.repertoires.names.size();
  /* -x: one matrix row per set-1 sequence, in input order (overlap.cc:889-899) */
  const size_t R1 = o.existence ? (size_t)set1.size() : set1.repertoires.names.size();
  std::vector<double> cells(R1 * R2, 0.0);
  BackendReport rep;
  std::string error;
  PairList pairs;
  if (warm.joinable()) {
    warm.join();
    reader_thread_active(false);
  }
  host_mark("backend warm");
  if (!backend.overlap(o, genes, set1, set2, same, cells, rep, error,
                       o.pairs ? &pairs : nullptr)) {
    fprintf(stderr, "\nError: %s\n", error.c_str());
    return 1;
  }
  host_mark("backend done");
  /* the reference's duplicate warnings (overlap.cc:846-851, 872-873) */
  if (rep.dup_set1 > 0)
    fprintf(log, "Warning: %lu duplicates detected in repertoire set 1\n",
            (unsigned long)rep.dup_set1);
  fprintf(log, "Hashing sequences: 100%% (%.9lfs)\n", rep.seconds_index);
  if (rep.dup_set2 > 0)
    fprintf(log, "Warning: %lu duplicates detected in repertoire set 2\n",
            (unsigned long)rep.dup_set2);
  fprintf(log, "Query layout:      100%% (%.9lfs)\n", rep.seconds_queries);
  fprintf(log, "Analysing:         100%% (%.9lfs)\n", rep.seconds_analysis);
  if (!rep.device_name.empty()) {
    const double s = rep.kernel_ms * 1e-3;
    fprintf(log, "GPU:               %s\n", rep.device_name.c_str());
    fprintf(log, "GPU kernel:        %.3f ms, %.3e query sequences/s, "
                 "%.1f GB/s algorithmic\n",
            rep.kernel_ms, s > 0 ? set1.size() / s : 0.0,
            s > 0 ? rep.algorithmic_bytes / s * 1e-9 : 0.0);
    fprintf(log, "GPU work:          %lu variants, %lu Bloom positives, "
                 "%lu hash matches, %lu pairs\n",
            (unsigned long)rep.variants, (unsigned long)rep.bloom_positive,
            (unsigned long)rep.hash_equal, (unsigned long)rep.matches);
  }

  /* ---- pairs (overlap.cc:908-925 header, 455-507 lines; order unspecified,
          README.md:163) ---- */
  if (pairsfile) {
    const char *letters = o.nucleotides ? "acgt" : "ACDEFGHIKLMNPQRSTVWY";   /* db.cc:73-75 */
    fprintf(pairsfile, "#repertoire_id_1\tsequence_id_1\tduplicate_count_1\tv_call_1\tj_call_1\t%s_1",
            o.seq_header);
    for (const std::string &k : o.keep_names)
      fprintf(pairsfile, "\t%s_1", k.c_str());
    fprintf(pairsfile, "\trepertoire_id_2\tsequence_id_2\tduplicate_count_2\tv_call_2\tj_call_2\t%s_2",
            o.seq_header);
    for (const std::string &k : o.keep_names)
      fprintf(pairsfile, "\t%s_2", k.c_str());
    if (o.distance)
      fprintf(pairsfile, "\tdistance");
    fprintf(pairsfile, "\n");
    auto put = [&](const RepertoireSet &s, uint32_t i) {
      fprintf(pairsfile, "%s\t%s\t%lu\t%s\t%s\t", s.repertoires.names[s.repertoire[i]].c_str(),
              i < s.sequence_id.size() ? s.sequence_id[i].c_str() : "",
              (unsigned long)s.count[i], genes.v.names[s.v_gene[i]].c_str(),
              genes.j.names[s.j_gene[i]].c_str());
      for (uint64_t p = s.offsets[i]; p < s.offsets[i + 1]; p++)
        fputc(letters[s.residues[p]], pairsfile);
      if (!o.keep_names.empty())
        fprintf(pairsfile, "\t%s", i < s.keep.size() ? s.keep[i].c_str() : "");
    };
    for (size_t k = 0; k < pairs.seed.size(); k++) {
      const uint32_t a = pairs.seed[k], b = pairs.hit[k];
      put(set1, a);
      fputc('\t', pairsfile);
      put(set2, b);
      if (o.distance) {
        /* Hamming distance for equal lengths, else one indel (overlap.cc:491-502) */
        long dist = 1;
        const uint64_t la = set1.offsets[a + 1] - set1.offsets[a];
        const uint64_t lb = set2.offsets[b + 1] - set2.offsets[b];
        if (la == lb) {
          dist = 0;
          for (uint64_t p = 0; p < la; p++)
            dist += set1.residues[set1.offsets[a] + p] != set2.residues[set2.offsets[b] + p];
        }
        fprintf(pairsfile, "\t%ld", dist);
      }
      fputc('\n', pairsfile);
    }
    fclose(pairsfile);
  }

  /* ---- print (overlap.cc:944-1039): rows/columns in strcmp order of the
          ids, every value "\t%.10lg" ---- */
  t0 = std::chrono::steady_clock::now();
  if (o.no_matrix) {
    /* --no-matrix: nothing is printed (overlap.cc:944-946) */
  } else if (o.existence) {
    /* overlap.cc:971-991, 1017-1037: rows are the sequences, by sequence_id */
    if (o.alternative) {
      fprintf(out, "#sequence_id_1\trepertoire_id_2\tmatches\n");
      for (size_t i = 0; i < R1; i++)
        for (size_t j = 0; j < R2; j++) {
          const uint32_t t = tot2.order[j];
          fprintf(out, "%s\t%s\t%.10lg\n", set1.sequence_id[i].c_str(),
                  set2.repertoires.names[t].c_str(), cells[R2 * i + t]);
        }
    } else {
      fprintf(out, "#");
      for (size_t j = 0; j < R2; j++)
        fprintf(out, "\t%s", set2.repertoires.names[tot2.order[j]].c_str());
      fprintf(out, "\n");
      for (size_t i = 0; i < R1; i++) {
        fprintf(out, "%s", set1.sequence_id[i].c_str());
        for (size_t j = 0; j < R2; j++)
          fprintf(out, "\t%.10lg", cells[R2 * i + tot2.order[j]]);
        fprintf(out, "\n");
      }
    }
  } else if (o.alternative) {
    fprintf(out, "#repertoire_id_1\trepertoire_id_2\tmatches\n");
    for (size_t i = 0; i < R1; i++) {
      const uint32_t s = tot1.order[i];
      for (size_t j = 0; j < R2; j++) {
        const uint32_t t = tot2.order[j];
        fprintf(out, "%s\t%s", set1.repertoires.names[s].c_str(),
                set2.repertoires.names[t].c_str());
        fprintf(out, "\t%.10lg", cell_value(o, cells[R2 * s + t], tot1, s, tot2, t));
        fprintf(out, "\n");
      }
    }
  } else {
    fprintf(out, "#");
    for (size_t j = 0; j < R2; j++)
      fprintf(out, "\t%s", set2.repertoires.names[tot2.order[j]].c_str());
    fprintf(out, "\n");
    for (size_t i = 0; i < R1; i++) {
      const uint32_t s = tot1.order[i];
      fprintf(out, "%s", set1.repertoires.names[s].c_str());
      for (size_t j = 0; j < R2; j++) {
        const uint32_t t = tot2.order[j];
        fprintf(out, "\t%.10lg", cell_value(o, cells[R2 * s + t], tot1, s, tot2, t));
      }
      fprintf(out, "\n");
    }
  }
  t1 = std::chrono::steady_clock::now();
  fprintf(log, "Writing results:   100%% (%.9lfs)\n\n",
          std::chrono::duration<double>(t1 - t0).count());
  log_time(log, "End time:          ");

  fclose(out);
  if (log != stderr)
    fclose(log);
  return 0;
}

}  // namespace cmprhost
