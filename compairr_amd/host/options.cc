/*
 * options.cc -- see options.h.  Behaviour follows
 * /root/reference/src/compairr.cc:292-706 (args_init), :200-246 (args_show),
 * :248-283 (args_usage); the code is this repository's own.
 */
#include <atomic>
#include <unistd.h>
#include "options.h"

#include <getopt.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

namespace cmprhost {

static const char *const kScoreNames[SCORE_END] = {
    "Product", "Ratio", "Min", "Max", "Mean", "MH", "Jaccard"};

static const char *const kScoreDescr[SCORE_END] = {
    "Sum of products of counts", "Sum of ratios of counts",
    "Sum of minimum of counts",  "Sum of maximum of counts",
    "Sum of mean of counts",     "Morisita-Horn index",
    "Jaccard index"};

const char *score_description(int64_t s)
{
  return (s >= 0 && s < SCORE_END) ? kScoreDescr[s] : "?";
}

thread_local DeferredExit *g_deferred_exit = nullptr;
static std::atomic<int> g_reader_threads(0);

void reader_thread_active(bool on)
{
  g_reader_threads += on ? 1 : -1;
}

void exit_with_message(FILE *stream, bool stream_is_log, const std::string &text)
{
  if (g_deferred_exit) {
    /* (a message for the log joins what the reader thread has logged so far, in its place) */
    g_deferred_exit->failed = true;
    g_deferred_exit->to_log = stream_is_log;
    if (stream_is_log)
      fputs(text.c_str(), stream);
    else
      g_deferred_exit->text = text;
    throw ReaderAbort();
  }
  fputs(text.c_str(), stream);
  if (g_reader_threads.load() > 0) {
    fflush(NULL);
    _exit(1);
  }
  exit(1);
}

void fatal(const char *msg)
{
  /* util.cc:84-88 */
  exit_with_message(stderr, false, std::string("\nError: ") + msg + "\n");
}

void print_header(FILE *f)
{
  fprintf(f, "CompAIRR-MI355X 0.1 (hot path of CompAIRR 1.13.0, HIP/gfx950) - "
             "Comparison of Adaptive Immune Receptor Repertoires\n");
  fprintf(f, "\n");
}

void print_usage(FILE *f)
{
  fprintf(f, "Usage: compairr [OPTIONS] TSVFILE1 [TSVFILE2]\n\n");
  fprintf(f, "Commands:\n");
  fprintf(f, " -h, --help                  display this help and exit\n");
  fprintf(f, " -v, --version               display version information\n");
  fprintf(f, " -m, --matrix                compute overlap matrix between two sets\n");
  fprintf(f, " -x, --existence             check existence of sequences in repertoires\n");
  fprintf(f, " -c, --cluster               cluster sequences in one repertoire\n");
  fprintf(f, " -z, --deduplicate           (not available in the MI355X build)\n");
  fprintf(f, "\nGeneral options:\n");
  fprintf(f, " -d, --differences INTEGER   number of differences accepted (0*, 1, 2)\n");
  fprintf(f, " -i, --indels                allow insertions or deletions when d=1\n");
  fprintf(f, " -f, --ignore-counts         ignore duplicate_count information\n");
  fprintf(f, " -g, --ignore-genes          ignore V and J gene information\n");
  fprintf(f, " -n, --nucleotides           compare nucleotides, not amino acids\n");
  fprintf(f, " -s, --score STRING          MH, Jaccard, product*, ratio, min, max, or mean\n");
  fprintf(f, " -t, --threads INTEGER       number of host threads to use (1*-256)\n");
  fprintf(f, " -u, --ignore-unknown        ignore sequences with unknown symbols\n");
  fprintf(f, " -e, --ignore-empty          ignore empty sequences\n");
  fprintf(f, "     --device INTEGER        HIP device to run on (current device*)\n");
  fprintf(f, "     --devices LIST          comma-separated HIP devices; queries are sharded over them\n");
  fprintf(f, "\nInput/output options:\n");
  fprintf(f, " -a, --alternative           output results in three-column format, not matrix\n");
  fprintf(f, "     --cdr3                  use the cdr3(_aa) column instead of junction(_aa)\n");
  fprintf(f, "     --distance              include sequence distance in pairs file\n");
  fprintf(f, " -k, --keep-columns STRING   comma-separated columns to copy to pairs file\n");
  fprintf(f, " -l, --log FILENAME          log to file (stderr*)\n");
  fprintf(f, " -o, --output FILENAME       output results to file (stdout*)\n");
  fprintf(f, "     --no-matrix             do not keep or output any matrix\n");
  fprintf(f, " -p, --pairs FILENAME        output matching pairs to file (none*)\n");
  fprintf(f, "\n                             * default value\n\n");
}

void print_options(FILE *f, const Options &o, const char *backend_name)
{
  if (o.cluster) {
    fprintf(f, "Command:           Cluster (-c)\n");
    fprintf(f, "Repertoire:        %s\n", o.input1);
  } else if (o.existence) {
    fprintf(f, "Command:           Existence (-x)\n");
    fprintf(f, "Repertoire:        %s\n", o.input1);
    fprintf(f, "Repertoire set:    %s\n", o.input2);
  } else {
    fprintf(f, "Command:           Overlap (-m)\n");
    fprintf(f, "Repertoire set 1:  %s\n", o.input1);
    fprintf(f, "Repertoire set 2:  %s\n", o.input2 ? o.input2 : "(same as set 1)");
  }
  fprintf(f, "Nucleotides (n):   %s\n", o.nucleotides ? "Yes" : "No");
  fprintf(f, "Differences (d):   %ld\n", (long)o.differences);
  fprintf(f, "Indels (i):        %s\n", o.indels ? "Yes" : "No");
  fprintf(f, "Ignore counts (f): %s\n", o.ignore_counts ? "Yes" : "No");
  fprintf(f, "Ignore genes (g):  %s\n", o.ignore_genes ? "Yes" : "No");
  fprintf(f, "Ign. unknown (u):  %s\n", o.ignore_unknown ? "Yes" : "No");
  fprintf(f, "Ignore empty (e):  %s\n", o.ignore_empty ? "Yes" : "No");
  fprintf(f, "Use cdr3 column:   %s\n", o.cdr3 ? "Yes" : "No");
  fprintf(f, "Threads (t):       %ld\n", (long)o.threads);
  fprintf(f, "Output file (o):   %s\n", o.no_matrix ? "(none)" : o.output);
  if (!o.cluster) {
    fprintf(f, "Output format (a): %s\n", o.alternative ? "Column" : "Matrix");
    fprintf(f, "Score (s):         %s\n", score_description(o.score));
    fprintf(f, "Pairs file (p):    %s\n", o.pairs ? o.pairs : "(none)");
    fprintf(f, "Keep columns:      %s\n", o.keep_columns ? o.keep_columns : "");
  }
  fprintf(f, "Log file (l):      %s\n", o.log ? o.log : "(stderr)");
  fprintf(f, "Backend:           %s\n", backend_name);
}

static int64_t numeric_argument(const char *str, const char *option)
{
  /* args_long, compairr.cc:175-185 */
  char *end = nullptr;
  int64_t v = strtol(str, &end, 10);
  if (*end) {
    fprintf(stderr, "\nInvalid numeric argument for option %s\n", option);
    exit(1);
  }
  return v;
}

void parse_command_line(int argc, char **argv, Options &o)
{
  static const char short_options[] = "acd:efghik:l:mno:p:s:t:uvxz";
  enum { LONG_CDR3 = 1000, LONG_DISTANCE, LONG_NO_MATRIX, LONG_DEVICE, LONG_DEVICES };
  static const struct option long_options[] = {
      {"alternative", no_argument, nullptr, 'a'},
      {"cdr3", no_argument, nullptr, LONG_CDR3},
      {"cluster", no_argument, nullptr, 'c'},
      {"differences", required_argument, nullptr, 'd'},
      {"distance", no_argument, nullptr, LONG_DISTANCE},
      {"ignore-empty", no_argument, nullptr, 'e'},
      {"ignore-counts", no_argument, nullptr, 'f'},
      {"ignore-genes", no_argument, nullptr, 'g'},
      {"help", no_argument, nullptr, 'h'},
      {"indels", no_argument, nullptr, 'i'},
      {"keep-columns", required_argument, nullptr, 'k'},
      {"log", required_argument, nullptr, 'l'},
      {"matrix", no_argument, nullptr, 'm'},
      {"nucleotides", no_argument, nullptr, 'n'},
      {"no-matrix", no_argument, nullptr, LONG_NO_MATRIX},
      {"output", required_argument, nullptr, 'o'},
      {"pairs", required_argument, nullptr, 'p'},
      {"score", required_argument, nullptr, 's'},
      {"summands", required_argument, nullptr, 's'},
      {"threads", required_argument, nullptr, 't'},
      {"ignore-unknown", no_argument, nullptr, 'u'},
      {"version", no_argument, nullptr, 'v'},
      {"existence", no_argument, nullptr, 'x'},
      {"deduplicate", no_argument, nullptr, 'z'},
      {"device", required_argument, nullptr, LONG_DEVICE},
      {"devices", required_argument, nullptr, LONG_DEVICES},
      {nullptr, 0, nullptr, 0}};

  bool seen[26] = {false};
  opterr = 1;
  optind = 1;
  int c;
  while ((c = getopt_long(argc, argv, short_options, long_options, nullptr)) != -1) {
    if (c >= 'a' && c <= 'z') {
      /* every lowercase option at most once (compairr.cc:403-423) */
      if (seen[c - 'a']) {
        const char *name = "?";
        for (const struct option *lo = long_options; lo->name; lo++)
          if (lo->val == c) {
            name = lo->name;
            break;
          }
        fprintf(stderr, "Error: Option -%c or --%s specified more than once.\n", c, name);
        exit(1);
      }
      seen[c - 'a'] = true;
    }
    switch (c) {
    case 'a': o.alternative = true; break;
    case 'c': o.cluster = true; break;
    case 'd': o.differences = numeric_argument(optarg, "-d or --differences"); break;
    case 'e': o.ignore_empty = true; break;
    case 'f': o.ignore_counts = true; break;
    case 'g': o.ignore_genes = true; break;
    case 'h': o.help = true; break;
    case 'i': o.indels = true; break;
    case 'k': o.keep_columns = optarg; break;
    case 'l': o.log = optarg; break;
    case 'm': o.matrix = true; break;
    case 'n': o.nucleotides = true; break;
    case 'o': o.output = optarg; break;
    case 'p': o.pairs = optarg; break;
    case 's': o.score_string = optarg; break;
    case 't': o.threads = numeric_argument(optarg, "-t or --threads"); break;
    case 'u': o.ignore_unknown = true; break;
    case 'v': o.version = true; break;
    case 'x': o.existence = true; break;
    case 'z': o.deduplicate = true; break;
    case LONG_CDR3: o.cdr3 = true; break;
    case LONG_DISTANCE: o.distance = true; break;
    case LONG_NO_MATRIX: o.no_matrix = true; break;
    case LONG_DEVICE: o.device = numeric_argument(optarg, "--device"); break;
    case LONG_DEVICES: {
      /* comma-separated device ordinals; the same ordinal may be listed twice */
      o.devices.clear();
      std::string cur;
      for (const char *q = optarg;; q++) {
        if (*q == ',' || *q == 0) {
          o.devices.push_back((int)numeric_argument(cur.c_str(), "--devices"));
          if (cur.empty() || o.devices.back() < 0)
            fatal("Argument to --devices must be a comma-separated list of device numbers");
          cur.clear();
          if (*q == 0)
            break;
        } else {
          cur.push_back(*q);
        }
      }
      break;
    }
    default:
      print_header(stderr);
      print_usage(stderr);
      exit(1);
    }
  }

  /* exactly one command (compairr.cc:561-565) */
  int commands = o.help + o.version + o.matrix + o.cluster + o.existence + o.deduplicate;
  if (commands == 0)
    fatal("Please specify a command (--help, --version, --matrix, --existence, --cluster, or --deduplicate)");
  if (commands > 1)
    fatal("Please specify just one command (--help, --version, --matrix, --existence, --cluster, or --deduplicate)");

  if (o.help || o.version) {
    if (optind != argc)
      fatal("Incorrect number of arguments");
    return;
  }

  /* commands outside the hot path this build replaces */
  if (o.deduplicate)
    fatal("The -z / --deduplicate command is not part of the MI355X build (only -m / --matrix is).");

  if (o.cluster) {
    /* compairr.cc:601-611 */
    if (optind + 1 == argc)
      o.input1 = argv[optind];
    else
      fatal("Incorrect number of arguments. One input file must be specified.");
  } else if (o.existence) {
    /* compairr.cc:589-600 */
    if (optind + 2 == argc) {
      o.input1 = argv[optind];
      o.input2 = argv[optind + 1];
    } else {
      fatal("Incorrect number of arguments. Two input files must be specified.");
    }
  } else if (optind + 2 == argc) {
    o.input1 = argv[optind];
    o.input2 = argv[optind + 1];
  } else if (optind + 1 == argc) {
    o.input1 = argv[optind];
    o.input2 = nullptr;
  } else {
    fatal("Incorrect number of arguments. One or two input files must be specified.");
  }

  if (o.keep_columns) {
    /* compairr.cc:620-626, parse_keep_columns :114-173 */
    if (!o.pairs)
      fatal("Option --keep-columns only allowed with --pairs options.");
    bool ok = true;
    std::string cur;
    for (const char *q = o.keep_columns;; q++) {
      const char ch = *q;
      if (ch == ',' || ch == 0) {
        if (cur.empty()) {
          ok = false;
          break;
        }
        o.keep_names.push_back(cur);
        cur.clear();
        if (ch == 0)
          break;
      } else if ((ch >= 'A' && ch <= 'Z') || (ch >= 'a' && ch <= 'z') ||
                 (ch >= '0' && ch <= '9') || ch == '_') {
        cur.push_back(ch);
      } else {
        ok = false;
        break;
      }
    }
    if (!ok)
      fatal("Illegal list of columns with --keep-columns option. It must be a comma-separated "
            "list of column names. Allowed symbols: A-Z, a-z, _, and 0-9.");
  }

  if (o.threads < 1 || o.threads > 256) {
    fprintf(stderr, "\nError: Illegal number of threads specified with "
                    "-t or --threads, must be in the range 1 to %u.\n", 256u);
    exit(1);
  }
  if (o.differences < 0)
    fatal("Differences specified with -d or -differences cannot be negative.");
  if (o.indels && o.differences != 1)
    fatal("Indels are only allowed when d=1");

  if (o.cluster) {
    /* compairr.cc:642-650 */
    if (o.pairs)
      fatal("Option -p or --pairs is not allowed with -c or --cluster");
    if (o.alternative)
      fatal("Option -a or --alternative is not allowed with -c or --cluster");
    if (o.score_string)
      fatal("Option -s or --score is not allowed with -c or --cluster");
  }

  if (o.score_string) {
    o.score = -1;
    for (int i = 0; i < SCORE_END; i++)
      if (strcasecmp(o.score_string, kScoreNames[i]) == 0) {
        o.score = i;
        break;
      }
    if (o.score < 0)
      fatal("Argument to -s or --score must be MH, Jaccard, product, ratio, min, max or mean");
  }
  if (!o.matrix) {
    /* compairr.cc:667-677 */
    if (o.score == SCORE_MH)
      fatal("The Morisita-Horn index is only allowed when computing repertoire overlap");
    if (o.score == SCORE_JACCARD)
      fatal("The Jaccard index is only allowed when computing repertoire overlap");
  }
  if (o.differences > 0) {
    if (o.score == SCORE_MH)
      fatal("The Morisita-Horn index is not defined when d>0");
    if (o.score == SCORE_JACCARD)
      fatal("The Jaccard index is not defined when d>0");
  }
  /* d > 2 is legal here; the backend decides (the HIP path rejects it). */

  o.alphabet_size = o.nucleotides ? 4 : 20;
  o.seq_header = o.cdr3 ? (o.nucleotides ? "cdr3" : "cdr3_aa")
                        : (o.nucleotides ? "junction" : "junction_aa");
}

}  // namespace cmprhost
