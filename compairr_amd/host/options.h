/*
 * options.h -- command line of the MI355X build of `compairr`.
 *
 * Same option table, defaults and validation rules as the reference program
 * (/root/reference/src/compairr.cc:292-706) for the commands built on the
 * per-query loop (--matrix, --existence, --cluster); --deduplicate is
 * recognised and rejected with a clear message instead of being silently
 * ignored.
 */
#ifndef COMPAIRR_AMD_OPTIONS_H
#define COMPAIRR_AMD_OPTIONS_H

#include <stdint.h>
#include <cstdio>
#include <string>
#include <vector>

namespace cmprhost {

enum Score {
  SCORE_PRODUCT = 0, SCORE_RATIO, SCORE_MIN, SCORE_MAX, SCORE_MEAN, SCORE_MH,
  SCORE_JACCARD, SCORE_END
};

struct Options {
  bool alternative = false;      /* -a */
  bool cdr3 = false;             /* --cdr3 */
  bool cluster = false;          /* -c */
  bool deduplicate = false;      /* -z (not on this path) */
  bool distance = false;         /* --distance */
  bool existence = false;        /* -x */
  bool help = false;             /* -h */
  bool ignore_counts = false;    /* -f */
  bool ignore_empty = false;     /* -e */
  bool ignore_genes = false;     /* -g */
  bool ignore_unknown = false;   /* -u */
  bool indels = false;           /* -i */
  bool matrix = false;           /* -m */
  bool nucleotides = false;      /* -n */
  bool no_matrix = false;        /* --no-matrix */
  bool version = false;          /* -v */
  const char *keep_columns = nullptr;  /* -k */
  std::vector<std::string> keep_names; /* parsed -k list */
  const char *log = nullptr;           /* -l */
  const char *output = "-";            /* -o */
  const char *pairs = nullptr;         /* -p */
  const char *score_string = nullptr;  /* -s */
  int64_t differences = 0;       /* -d */
  int64_t score = SCORE_PRODUCT;
  int64_t threads = 1;           /* -t (host threads; the loop runs on the GPU) */
  int64_t device = -1;           /* --device N: HIP device ordinal (addition) */
  std::vector<int> devices;      /* --devices A,B,..: one shard of the queries per
                                    listed device (addition) */

  const char *input1 = nullptr;
  const char *input2 = nullptr;

  int alphabet_size = 20;
  const char *seq_header = "junction_aa";
};

/* Parses argv exactly as the reference does (getopt_long, each lowercase
   option at most once, one command, argument counts, value checks).  Prints
   the reference's messages and exits with status 1 on any error. */
void parse_command_line(int argc, char **argv, Options &opt);

void print_usage(FILE *f);
void print_header(FILE *f);
void print_options(FILE *f, const Options &opt, const char *backend_name);

const char *score_description(int64_t score);

[[noreturn]] void fatal(const char *msg);

/* The second input file is read on a thread of its own while the first is being read
   (overlap_host.cc).  The reference reads them one after the other, so an error in file 1
   is what it reports even when file 2 is broken too: the reader thread does not print and
   exit, it keeps its message (`text` as it is to be written, to stderr or to the log) and
   unwinds; the main thread acts on it once file 1 has been read without error. */
struct DeferredExit {
  bool        failed = false;
  bool        to_log = false;      /* message goes to the log stream, not stderr */
  std::string text;
};
struct ReaderAbort {};
extern thread_local DeferredExit *g_deferred_exit;     /* non-NULL on a reader thread */
/* a reader thread is running: the main thread leaves with _exit (no static destructors under its feet) */
void reader_thread_active(bool on);
/* error exit of the file reader: `text` to `stream`, status 1 (or deferred, see above) */
[[noreturn]] void exit_with_message(FILE *stream, bool stream_is_log, const std::string &text);

}  // namespace cmprhost
#endif
