/*
 * airr_tsv.cc -- see airr_tsv.h.  Own implementation of the input rules of
 * /root/reference/src/db.cc (cited per rule below).
 *
 * The reference reads line by line on one thread (getline + strsep + three
 * std::map lookups per line, db.cc:708-901: 4-5 s per 10M-line file).  Here the
 * file is read into memory once, cut at line ends into one range per `-t`
 * thread, parsed in parallel into per-range buffers with range-local string
 * tables, and merged in file order, so that repertoire / V / J numbers are the
 * same first-appearance numbers the reference assigns (db.cc:510-520, 592-631)
 * and the first error in file order is the one reported.
 */
#include "airr_tsv.h"

#include <chrono>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <new>
#include <thread>
#include <unordered_map>

namespace cmprhost {

uint32_t Interner::intern(const char *s)
{
  auto it = index.find(s);
  if (it != index.end())
    return it->second;
  uint32_t no = (uint32_t)names.size();
  names.push_back(s);
  index.insert(std::make_pair(names.back(), no));
  return no;
}

namespace {

/* residue codes: position in "ACDEFGHIKLMNPQRSTVWY" / "ACGT" (U = T), either
   case (map_aa, map_nt: db.cc:33-71) */
struct ResidueMaps {
  signed char aa[256], nt[256];
  ResidueMaps()
  {
    memset(aa, -1, sizeof aa);
    memset(nt, -1, sizeof nt);
    const char *a = "ACDEFGHIKLMNPQRSTVWY";
    for (int i = 0; a[i]; i++) {
      aa[(unsigned char)a[i]] = (signed char)i;
      aa[(unsigned char)(a[i] | 0x20)] = (signed char)i;
    }
    const char *n = "ACGT";
    for (int i = 0; n[i]; i++) {
      nt[(unsigned char)n[i]] = (signed char)i;
      nt[(unsigned char)(n[i] | 0x20)] = (signed char)i;
    }
    nt[(unsigned char)'U'] = nt[(unsigned char)'u'] = 3;
  }
};
const ResidueMaps kMaps;

struct Columns {
  std::vector<int> keep;       /* -k columns, 0 = absent from the header */
  bool store_ids = false;      /* keep sequence_id (empty when absent)   */
  int repertoire_id = 0, sequence_id = 0, duplicate_count = 0, v_call = 0,
      j_call = 0, junction = 0, junction_aa = 0, cdr3 = 0, cdr3_aa = 0;
};

/* splits `line` in place on tabs (strsep semantics: empty fields are fields) */
void split_tabs(char *line, std::vector<char *> &fields)
{
  fields.clear();
  char *p = line;
  for (;;) {
    fields.push_back(p);
    char *t = strchr(p, '\t');
    if (!t)
      break;
    *t = 0;
    p = t + 1;
  }
}

/* db.cc:172-296 */
void parse_header(char *line, const Options &o, Columns &c, FILE *log, bool need_id)
{
  std::vector<char *> f;
  split_tabs(line, f);
  c.keep.assign(o.keep_names.size(), 0);
  for (size_t k = 0; k < f.size(); k++) {
    const int i = (int)k + 1;
    const char *t = f[k];
    if (!strcmp(t, "repertoire_id")) c.repertoire_id = i;
    else if (!strcmp(t, "sequence_id")) c.sequence_id = i;
    else if (!strcmp(t, "duplicate_count")) c.duplicate_count = i;
    else if (!strcmp(t, "v_call")) c.v_call = i;
    else if (!strcmp(t, "j_call")) c.j_call = i;
    else if (!strcmp(t, "junction")) c.junction = i;
    else if (!strcmp(t, "junction_aa")) c.junction_aa = i;
    else if (!strcmp(t, "cdr3")) c.cdr3 = i;
    else if (!strcmp(t, "cdr3_aa")) c.cdr3_aa = i;
    for (size_t kk = 0; kk < o.keep_names.size(); kk++)
      if (o.keep_names[kk] == t)
        c.keep[kk] = i;
  }
  const int seqcol = o.cdr3 ? (o.nucleotides ? c.cdr3 : c.cdr3_aa)
                            : (o.nucleotides ? c.junction : c.junction_aa);
  const bool missing = (!c.sequence_id && need_id) ||
                       (!c.duplicate_count && !o.ignore_counts) ||
                       (!c.v_call && !o.ignore_genes) ||
                       (!c.j_call && !o.ignore_genes) || !seqcol;
  if (missing) {
    fprintf(log, "\nMissing essential column(s) in header of AIRR TSV input file:");
    if (need_id && !c.sequence_id) fprintf(log, " sequence_id");
    if (!o.ignore_counts && !c.duplicate_count) fprintf(log, " duplicate_count");
    if (!o.ignore_genes) {
      if (!c.v_call) fprintf(log, " v_call");
      if (!c.j_call) fprintf(log, " j_call");
    }
    if (!seqcol) fprintf(log, " %s", o.seq_header);
    exit_with_message(log, true, "\n");
  }
  /* db.cc:283-295 */
  bool any_missing = false;
  for (int col : c.keep)
    any_missing = any_missing || col < 1;
  if (any_missing) {
    fprintf(log, "\nWarning: missing column(s) to keep in header:");
    for (size_t kk = 0; kk < c.keep.size(); kk++)
      if (c.keep[kk] < 1)
        fprintf(log, " %s", o.keep_names[kk].c_str());
    fprintf(log, "\n");
  }
}

inline const char *field(const std::vector<char *> &f, int col)
{
  return (col >= 1 && (size_t)col <= f.size()) ? f[col - 1] : nullptr;
}

/* string table local to one range of the file */
struct LocalNames {
  std::unordered_map<std::string, uint32_t> index;
  std::vector<std::string> names;           /* first-appearance order */
  /* (the value seen last is the commonest next one -- a file is mostly runs of one repertoire, and the gene
     of a line is often the gene of the line before: no map look-up then) */
  uint32_t last_no = 0xffffffffu;
  uint32_t intern(const char *s)
  {
    if (last_no != 0xffffffffu && names[last_no] == s)
      return last_no;
    const uint32_t no = intern_slow(s);
    last_no = no;
    return no;
  }
  uint32_t intern_slow(const char *s)
  {
    auto it = index.find(s);
    if (it != index.end())
      return it->second;
    uint32_t no = (uint32_t)names.size();
    names.push_back(s);
    index.emplace(names.back(), no);
    return no;
  }
};

/* Transparent huge pages for the reader's large buffers (file text, parsed arrays): two 10M-sequence files are
   ~2.5 GB of fresh memory, and faulting that in 4 KiB at a time -- 600 000 faults that 64 threads take under one lock --
   was most of the "read" phase (measured apart: 2.5 GiB touched by 64 threads in 210-280 ms, in 9 ms on 2-MiB pages;
   tools/dev/exit_cost.c).  Where the system offers them on request (transparent_hugepage = madvise) the 2-MiB-aligned
   inside of a fresh allocation is asked for huge pages before it is touched; elsewhere this is a no-op. */
static void advise_huge(void *p, size_t bytes)
{
#ifdef MADV_HUGEPAGE
  static const bool off = getenv("COMPAIRR_NO_HUGEPAGES") != nullptr;      /* (measurement aid) */
  if (off)
    return;
  const uintptr_t H = (uintptr_t)2 << 20;
  const uintptr_t a = ((uintptr_t)p + H - 1) & ~(H - 1), e = ((uintptr_t)p + bytes) & ~(H - 1);
  if (p && e > a)
    (void)madvise((void *)a, (size_t)(e - a), MADV_HUGEPAGE);
#else
  (void)p;
  (void)bytes;
#endif
}
template <typename V>
static void reserve_huge(V &v, size_t n)
{
  v.reserve(n);
  advise_huge((void *)v.data(), v.capacity() * sizeof(typename V::value_type));
}

/* A range's parsed arrays (RangeResult) live in ONE mapping per range, 2-MiB aligned and asked for huge pages: six
   vectors of a megabyte or two each would sit in mappings of their own, too small and too oddly placed for a huge
   page (64 ranges per file: 0.85 GiB in 4-KiB faults for two 10M-sequence files).  A bump allocator; what does not fit
   (a vector that outgrows the estimate) comes from malloc. */
struct HugeArena {
  char  *map = nullptr, *base = nullptr;
  size_t map_bytes = 0, cap = 0, used = 0;
  HugeArena() {}
  HugeArena(const HugeArena &) = delete;
  HugeArena &operator=(const HugeArena &) = delete;
  ~HugeArena()
  {
    if (map)
      munmap(map, map_bytes);
  }
  void init(size_t bytes)
  {
    const size_t H = (size_t)2 << 20;
    const size_t want = (bytes + H - 1) & ~(H - 1);
    void *m = mmap(nullptr, want + H, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED)
      return;                                  /* (everything from malloc then) */
    map = (char *)m;
    map_bytes = want + H;
    base = (char *)(((uintptr_t)map + H - 1) & ~(uintptr_t)(H - 1));
    cap = want;
    advise_huge(base, cap);
  }
  void *take(size_t n)
  {
    n = (n + 63) & ~(size_t)63;
    if (!base || n > cap - used)
      return nullptr;
    void *p = base + used;
    used += n;
    return p;
  }
  bool owns(const void *p) const { return base && (const char *)p >= base && (const char *)p < base + cap; }
};
template <typename T> struct ArenaAlloc {
  typedef T value_type;
  HugeArena *a;
  ArenaAlloc(HugeArena *x = nullptr) : a(x) {}
  template <typename U> ArenaAlloc(const ArenaAlloc<U> &o) : a(o.a) {}
  T *allocate(size_t n)
  {
    void *p = a ? a->take(n * sizeof(T)) : nullptr;
    if (!p && !(p = malloc(n * sizeof(T) + (n == 0))))
      throw std::bad_alloc();
    return (T *)p;
  }
  void deallocate(T *p, size_t)
  {
    if (!(a && a->owns(p)))
      free(p);
  }
  template <typename U> bool operator==(const ArenaAlloc<U> &o) const { return a == o.a; }
  template <typename U> bool operator!=(const ArenaAlloc<U> &o) const { return a != o.a; }
};
template <typename T> using ArenaVec = std::vector<T, ArenaAlloc<T> >;

/* what one thread produces from its range */
struct RangeResult {
  HugeArena             arena;                /* (first: it outlives the vectors that live in it) */
  ArenaVec<uint8_t>     residues;
  ArenaVec<uint32_t>    lengths, v, j, rep;   /* range-local string numbers */
  ArenaVec<uint64_t>    count;
  std::vector<std::string> ids;              /* sequence_id, when kept */
  std::vector<std::string> keep;             /* -k columns, tab-joined */
  LocalNames            reps, vs, js;
  uint64_t              ignored_unknown = 0, ignored_empty = 0;
  bool                  failed = false;
  std::string           error;                /* the message for the log */
  RangeResult()
      : residues(ArenaAlloc<uint8_t>(&arena)), lengths(ArenaAlloc<uint32_t>(&arena)), v(ArenaAlloc<uint32_t>(&arena)),
        j(ArenaAlloc<uint32_t>(&arena)), rep(ArenaAlloc<uint32_t>(&arena)), count(ArenaAlloc<uint64_t>(&arena))
  {
  }
  RangeResult(const RangeResult &) = delete;
  RangeResult &operator=(const RangeResult &) = delete;
};

void fail_line(RangeResult &r, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void fail_line(RangeResult &r, const char *fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  r.failed = true;
  r.error = buf;
}

/* One data line (db.cc:298-706).  Returns false after recording an error. */
bool parse_line(char *line, uint64_t lineno, const Options &o, const Columns &c,
                const char *default_rep, RangeResult &d, std::vector<char *> &f, bool need_id)
{
  split_tabs(line, f);
  const char *repertoire_id = field(f, c.repertoire_id);
  const char *duplicate_count = field(f, c.duplicate_count);
  const char *v_call = field(f, c.v_call);
  const char *j_call = field(f, c.j_call);
  const int seqcol = o.cdr3 ? (o.nucleotides ? c.cdr3 : c.cdr3_aa)
                            : (o.nucleotides ? c.junction : c.junction_aa);
  const char *seq = field(f, seqcol);
  if (!seq) {
    /* the reference dereferences a null pointer here (db.cc:384-398); report
       it the way it reports an empty value (db.cc:653-668) */
    fail_line(d, "\n\nError: missing or empty %s value on line %lu\n", o.seq_header,
              (unsigned long)lineno);
    return false;
  }

  /* residues (db.cc:410-486) */
  const signed char *map = o.nucleotides ? kMaps.nt : kMaps.aa;
  const size_t start = d.residues.size();
  bool drop = false;
  for (const char *q = seq; *q; q++) {
    const unsigned char ch = (unsigned char)*q;
    const signed char m = map[ch];
    if (m >= 0) {
      d.residues.push_back((uint8_t)m);
    } else if (ch >= 32 && ch <= 126) {
      if (o.ignore_unknown) {
        drop = true;
        d.ignored_unknown++;
      } else {
        fail_line(d, "\n\nError: Illegal character '%c' in sequence on line %lu. "
                     "Use -u to ignore.\n", ch, (unsigned long)lineno);
        return false;
      }
    } else {
      fail_line(d, "\n\nError: Illegal character (ascii no %d) in sequence on line %lu\n",
                ch, (unsigned long)lineno);
      return false;
    }
  }
  const uint32_t len = (uint32_t)(d.residues.size() - start);
  if (len == 0) {
    if (o.ignore_empty) {
      drop = true;
      d.ignored_empty++;
    } else {
      fail_line(d, "\n\nError: Empty sequence in sequence on line %lu. Use -e to ignore.\n",
                (unsigned long)lineno);
      return false;
    }
  }
  if (drop) {
    d.residues.resize(start);
    return true;
  }

  /* repertoire_id: default when the column or the field is absent (db.cc:505-520) */
  const uint32_t rep = d.reps.intern(repertoire_id ? repertoire_id : default_rep);

  /* sequence_id (db.cc:523-542): required by -x for the first file */
  const char *sequence_id = field(f, c.sequence_id);
  if (need_id && !(sequence_id && *sequence_id)) {
    fail_line(d, "\n\nError: missing or empty sequence_id value on line %lu\n",
              (unsigned long)lineno);
    return false;
  }

  /* duplicate_count (db.cc:545-571) */
  uint64_t count = 1;
  if (duplicate_count && *duplicate_count) {
    char *end = nullptr;
    long v = strtol(duplicate_count, &end, 10);
    if (end && *end == 0 && v >= 1) {
      count = (uint64_t)v;
    } else {
      fail_line(d, "\n\nError: Illegal duplicate_count on line %lu: %s\n",
                (unsigned long)lineno, duplicate_count);
      return false;
    }
  } else if (!o.ignore_counts) {
    fail_line(d, "\n\nError: missing or empty duplicate_count on line %lu\n",
              (unsigned long)lineno);
    return false;
  }

  /* v_call, j_call (db.cc:578-631): required unless -g, interned either way */
  if (!o.ignore_genes && !(v_call && *v_call)) {
    fail_line(d, "\n\nError: missing or empty v_call value on line %lu\n", (unsigned long)lineno);
    return false;
  }
  if (!o.ignore_genes && !(j_call && *j_call)) {
    /* the reference interns v_call before it looks at j_call; the tables of a
       failing run are never used, so only the message order matters */
    fail_line(d, "\n\nError: missing or empty j_call value on line %lu\n", (unsigned long)lineno);
    return false;
  }
  if (need_id || c.store_ids)
    d.ids.push_back(sequence_id ? sequence_id : "");
  if (!c.keep.empty()) {
    /* db.cc:671-701: the kept columns, tab separated, absent ones empty */
    std::string k;
    for (size_t kk = 0; kk < c.keep.size(); kk++) {
      if (kk)
        k.push_back('\t');
      const char *val = field(f, c.keep[kk]);
      if (val)
        k += val;
    }
    d.keep.push_back(k);
  }
  d.lengths.push_back(len);
  d.rep.push_back(rep);
  d.count.push_back(count);
  d.v.push_back(d.vs.intern(v_call ? v_call : ""));
  d.j.push_back(d.js.intern(j_call ? j_call : ""));
  return true;
}


/* parses the lines of text[begin, end) (ends at a line end or at EOF) */
void parse_range(char *text, size_t begin, size_t end, uint64_t first_lineno,
                 const Options &o, const Columns &c, const char *default_rep,
                 RangeResult &out, bool need_id, size_t nlines)
{
  /* (the lines of the range were counted for the error messages' sake: room for one sequence each, no vector
     grows -- and copies itself -- while the range is parsed) */
  const size_t nres_guess = nlines ? (end - begin) / 6 + 64 : 0;
  out.arena.init(nlines * (4 * sizeof(uint32_t) + sizeof(uint64_t)) + nres_guess + 6 * 64);
  out.lengths.reserve(nlines);
  out.v.reserve(nlines);
  out.j.reserve(nlines);
  out.rep.reserve(nlines);
  out.count.reserve(nlines);
  out.residues.reserve(nres_guess);
  std::vector<char *> fields;
  uint64_t lineno = first_lineno;
  size_t pos = begin;
  while (pos < end) {
    char *line = text + pos;
    char *nl = (char *)memchr(line, '\n', end - pos);
    size_t n = nl ? (size_t)(nl - line) : end - pos;
    pos += n + (nl ? 1 : 0);
    /* LF, then CR of DOS files (db.cc:765-775, 824-836) */
    line[n] = 0;
    if (n > 0 && line[n - 1] == '\r')
      line[--n] = 0;
    if (!parse_line(line, lineno, o, c, default_rep, out, fields, need_id))
      return;
    lineno++;
  }
}

/* the file's bytes and a closing NUL (the last line may lack its LF) */
struct FileText {
  char  *p = nullptr;
  size_t size = 0;            /* bytes of the file */
  ~FileText() { free(p); }
  char *data() { return p; }
};

/* A regular file: one allocation of its size, not value-initialised (a vector that doubles copies a 370 MB
   file twice over and clears it first), filled by one or two readers (pread).  Anything else (a pipe,
   standard input): read to its end into a growing buffer. */
bool read_whole_file(const char *filename, size_t threads, FileText &text)
{
  int fd = strcmp(filename, "-") == 0 ? dup(STDIN_FILENO) : open(filename, O_RDONLY);
  if (fd < 0)
    return false;
  struct stat sb;
  if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0 && lseek(fd, 0, SEEK_CUR) == 0) {
    const size_t n = (size_t)sb.st_size;
    text.p = (char *)malloc(n + 1);
    if (text.p) {
      advise_huge(text.p, n + 1);
      /* (two readers at most: with the second file being read beside this one, more of them only meet in the
         page cache's locks -- 64 readers per file took 0.40-0.56 s for what one takes 0.35-0.45 s, measured) */
      size_t parts = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(threads, 2), n / (4 << 20) + 1));
      if (const char *e = getenv("COMPAIRR_READ_PARTS"))      /* (measurement aid: readers of a file) */
        parts = std::max<size_t>(1, std::min<size_t>(parts, (size_t)atoi(e)));
      std::vector<size_t> got(parts, 0);
      std::vector<std::thread> pool;
      for (size_t r = 0; r < parts; r++)
        pool.emplace_back([&, r]() {
          size_t at = n * r / parts;
          const size_t end = n * (r + 1) / parts;
          while (at < end) {
            const ssize_t k = pread(fd, text.p + at, end - at, (off_t)at);
            if (k <= 0)
              break;
            at += (size_t)k;
          }
          got[r] = at - n * r / parts;
        });
      for (auto &t : pool)
        t.join();
      size_t used = 0;
      bool whole = true;
      for (size_t r = 0; r < parts; r++) {
        whole = whole && got[r] == n * (r + 1) / parts - n * r / parts;
        used += got[r];
      }
      if (whole) {
        close(fd);
        text.size = used;
        text.p[used] = 0;
        return true;
      }
      /* (the file shrank under our feet: read it the plain way) */
      free(text.p);
      text.p = nullptr;
    }
  }
  size_t cap = 1 << 20, used = 0;
  text.p = (char *)malloc(cap);
  while (text.p) {
    if (used + 1 >= cap) {
      cap *= 2;
      char *q = (char *)realloc(text.p, cap);
      if (!q)
        break;
      text.p = q;
    }
    const ssize_t k = read(fd, text.p + used, cap - used - 1);
    if (k <= 0)
      break;
    used += (size_t)k;
  }
  close(fd);
  if (!text.p)
    return false;
  text.size = used;
  text.p[used] = 0;
  return true;
}

}  // namespace

/* COMPAIRR_HOST_TIMING=1: the reader's phases, on stderr */
static void reader_mark(const char *file, const char *what, std::chrono::steady_clock::time_point t0)
{
  static const bool on = getenv("COMPAIRR_HOST_TIMING") != nullptr;
  if (on)
    fprintf(stderr, "[reader %s] %-12s %8.3f ms\n", file, what,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
}

void read_airr_tsv(const char *filename, const Options &o, GeneTables &genes,
                   const char *default_rep, FILE *log, RepertoireSet &d, bool need_id,
                   bool keep_id)
{
  const auto t_begin = std::chrono::steady_clock::now();
  FileText text;
  if (!read_whole_file(filename, (size_t)std::max<int64_t>(1, o.threads), text)) {
    exit_with_message(log, true, std::string("\nError: Unable to open input data file (") + filename + ").\n");
  }
  reader_mark(filename, "file read", t_begin);
  const size_t size = text.size;
  if (size == 0)
    fatal("Unable to read from the input file");   /* db.cc:758-759 */

  d = RepertoireSet();
  d.offsets.push_back(0);

  /* leading comment lines, then the header (db.cc:781-797) */
  Columns cols;
  size_t pos = 0;
  uint64_t lineno = 0;
  bool have_header = false;
  while (pos < size && !have_header) {
    char *line = text.data() + pos;
    char *nl = (char *)memchr(line, '\n', size - pos);
    size_t n = nl ? (size_t)(nl - line) : size - pos;
    pos += n + (nl ? 1 : 0);
    lineno++;
    line[n] = 0;
    if (n > 0 && line[n - 1] == '\r')
      line[--n] = 0;
    if (line[0] == '#' || line[0] == '@')
      continue;
    parse_header(line, o, cols, log, need_id);
    cols.store_ids = keep_id;
    have_header = true;
  }

  /* cut the data lines into one range per thread, at line ends */
  const size_t threads = (size_t)std::max<int64_t>(1, o.threads);
  const size_t body = size - pos;
  size_t nranges = std::min<size_t>(threads, body / (1 << 16) + 1);
  std::vector<size_t> cut(nranges + 1);
  cut[0] = pos;
  cut[nranges] = size;
  for (size_t r = 1; r < nranges; r++) {
    size_t c = pos + body * r / nranges;
    if (c < cut[r - 1])
      c = cut[r - 1];
    const char *nl = (const char *)memchr(text.data() + c, '\n', size - c);
    cut[r] = nl ? (size_t)(nl - text.data()) + 1 : size;
  }
  /* line number of the first line of every range (error messages) */
  std::vector<uint64_t> first_line(nranges, lineno + 1);
  std::vector<uint64_t> lines(nranges, 0);
  {
    std::vector<std::thread> pool;
    for (size_t r = 0; r < nranges; r++)
      pool.emplace_back([&, r]() {
        uint64_t k = 0;
        const char *p = text.data() + cut[r], *e = text.data() + cut[r + 1];
        while (p < e && (p = (const char *)memchr(p, '\n', e - p)) != nullptr) {
          k++;
          p++;
        }
        lines[r] = k;
      });
    for (auto &t : pool)
      t.join();
    for (size_t r = 1; r < nranges; r++)
      first_line[r] = first_line[r - 1] + lines[r - 1];
  }

  reader_mark(filename, "lines counted", t_begin);
  std::vector<RangeResult> part(nranges);
  {
    std::vector<std::thread> pool;
    for (size_t r = 0; r < nranges; r++)
      pool.emplace_back([&, r]() {
        parse_range(text.data(), cut[r], cut[r + 1], first_line[r], o, cols, default_rep,
                    part[r], need_id, (size_t)lines[r] + 1);
      });
    for (auto &t : pool)
      t.join();
  }

  reader_mark(filename, "parsed", t_begin);
  /* the first error in file order is the one the serial reference would hit */
  for (size_t r = 0; r < nranges; r++)
    if (part[r].failed) {
      exit_with_message(log, true, part[r].error);
    }

  /* merge in file order: global first-appearance numbering.  The names are few and merged serially, in
     file order; the per-sequence arrays are copied by one thread per range to the places a prefix sum of
     the ranges' sizes gives them (10M sequences: a serial merge took as long as the parsing). */
  std::vector<size_t> n0(nranges + 1, 0), r0(nranges + 1, 0);
  for (size_t r = 0; r < nranges; r++) {
    n0[r + 1] = n0[r] + part[r].lengths.size();
    r0[r + 1] = r0[r] + part[r].residues.size();
  }
  const size_t n = n0[nranges], nres = r0[nranges];
  std::vector<std::vector<uint32_t> > mr(nranges), mv(nranges), mj(nranges);
  for (size_t r = 0; r < nranges; r++) {
    const RangeResult &p = part[r];
    /* a sequence interns its repertoire, then V, then J; the three tables are
       independent, so mapping each range-local table in its own first-appearance
       order reproduces the global order */
    mr[r].resize(p.reps.names.size());
    for (size_t k = 0; k < mr[r].size(); k++)
      mr[r][k] = d.repertoires.intern(p.reps.names[k].c_str());
    mv[r].resize(p.vs.names.size());
    for (size_t k = 0; k < mv[r].size(); k++)
      mv[r][k] = genes.v.intern(p.vs.names[k].c_str());
    mj[r].resize(p.js.names.size());
    for (size_t k = 0; k < mj[r].size(); k++)
      mj[r][k] = genes.j.intern(p.js.names[k].c_str());
    d.sequence_id.insert(d.sequence_id.end(), p.ids.begin(), p.ids.end());
    d.keep.insert(d.keep.end(), p.keep.begin(), p.keep.end());
    d.ignored_unknown += p.ignored_unknown;
    d.ignored_empty += p.ignored_empty;
  }
  reserve_huge(d.residues, nres);          /* (asked for huge pages before the resize touches -- zero-fills -- them) */
  reserve_huge(d.offsets, n + 1);
  reserve_huge(d.v_gene, n);
  reserve_huge(d.j_gene, n);
  reserve_huge(d.repertoire, n);
  reserve_huge(d.count, n);
  d.residues.resize(nres);
  d.offsets.resize(n + 1);
  d.offsets[0] = 0;
  d.v_gene.resize(n);
  d.j_gene.resize(n);
  d.repertoire.resize(n);
  d.count.resize(n);
  {
    std::vector<uint64_t> tot(nranges, 0);
    std::vector<uint32_t> lmax(nranges, 0), lmin(nranges, 0xffffffffu);
    std::vector<std::thread> pool;
    for (size_t r = 0; r < nranges; r++)
      pool.emplace_back([&, r]() {
        const RangeResult &p = part[r];
        if (!p.residues.empty())
          memcpy(d.residues.data() + r0[r], p.residues.data(), p.residues.size());
        uint64_t off = r0[r];
        for (size_t k = 0; k < p.lengths.size(); k++) {
          const uint32_t len = p.lengths[k];
          const size_t i = n0[r] + k;
          off += len;
          d.offsets[i + 1] = off;
          d.repertoire[i] = mr[r][p.rep[k]];
          d.v_gene[i] = mv[r][p.v[k]];
          d.j_gene[i] = mj[r][p.j[k]];
          d.count[i] = p.count[k];
          tot[r] += p.count[k];
          if (len > lmax[r]) lmax[r] = len;
          if (len < lmin[r]) lmin[r] = len;
        }
      });
    for (auto &t : pool)
      t.join();
    for (size_t r = 0; r < nranges; r++) {
      d.total_count += tot[r];
      if (lmax[r] > d.longest) d.longest = lmax[r];
      if (lmin[r] < d.shortest) d.shortest = lmin[r];
    }
  }

  reader_mark(filename, "merged", t_begin);
  /* db.cc:847-887 */
  if (d.ignored_unknown > 0)
    fprintf(log, "%lu sequences with unknown symbols ignored.\n", (unsigned long)d.ignored_unknown);
  if (d.ignored_empty > 0)
    fprintf(log, "%lu empty sequences ignored.\n", (unsigned long)d.ignored_empty);
  fprintf(log, "Repertoires:       %lu\n", (unsigned long)d.repertoires.names.size());
  fprintf(log, "Sequences:         %lu\n", (unsigned long)d.size());
  fprintf(log, "Residues:          %lu\n", (unsigned long)d.residue_count());
  if (d.size() > 0) {
    fprintf(log, "Shortest:          %u\n", d.shortest);
    fprintf(log, "Longest:           %u\n", d.longest);
    fprintf(log, "Average length:    %.1lf\n", 1.0 * d.residue_count() / d.size());
  } else {
    fprintf(log, "Shortest:          -\nLongest:           -\nAverage length:    -\n");
  }
  fprintf(log, "Total dupl. count: %lu\n", (unsigned long)d.total_count);
}

}  // namespace cmprhost
