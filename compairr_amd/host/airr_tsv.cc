/*
 * airr_tsv.cc -- see airr_tsv.h.  Own implementation of the input rules of
 * /root/reference/src/db.cc (cited per rule below).
 */
#include "airr_tsv.h"

#include <stdlib.h>
#include <string.h>
#include <unistd.h>

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
void parse_header(char *line, const Options &o, Columns &c, FILE *log)
{
  std::vector<char *> f;
  split_tabs(line, f);
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
  }
  const int seqcol = o.cdr3 ? (o.nucleotides ? c.cdr3 : c.cdr3_aa)
                            : (o.nucleotides ? c.junction : c.junction_aa);
  const bool missing = (!c.duplicate_count && !o.ignore_counts) ||
                       (!c.v_call && !o.ignore_genes) ||
                       (!c.j_call && !o.ignore_genes) || !seqcol;
  if (missing) {
    fprintf(log, "\nMissing essential column(s) in header of AIRR TSV input file:");
    if (!o.ignore_counts && !c.duplicate_count) fprintf(log, " duplicate_count");
    if (!o.ignore_genes) {
      if (!c.v_call) fprintf(log, " v_call");
      if (!c.j_call) fprintf(log, " j_call");
    }
    if (!seqcol) fprintf(log, " %s", o.seq_header);
    fprintf(log, "\n");
    exit(1);
  }
}

inline const char *field(const std::vector<char *> &f, int col)
{
  return (col >= 1 && (size_t)col <= f.size()) ? f[col - 1] : nullptr;
}

/* db.cc:298-706 */
void parse_line(char *line, uint64_t lineno, const Options &o, const Columns &c,
                GeneTables &genes, const char *default_rep, FILE *log,
                RepertoireSet &d, std::vector<char *> &f)
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
    fprintf(log, "\n\nError: missing or empty %s value on line %lu\n",
            o.seq_header, (unsigned long)lineno);
    exit(1);
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
        fprintf(log, "\n\nError: Illegal character '%c' in sequence on line %lu. "
                     "Use -u to ignore.\n", ch, (unsigned long)lineno);
        exit(1);
      }
    } else {
      fprintf(log, "\n\nError: Illegal character (ascii no %d) in sequence on line %lu\n",
              ch, (unsigned long)lineno);
      exit(1);
    }
  }
  const uint32_t len = (uint32_t)(d.residues.size() - start);
  if (len == 0) {
    if (o.ignore_empty) {
      drop = true;
      d.ignored_empty++;
    } else {
      fprintf(log, "\n\nError: Empty sequence in sequence on line %lu. Use -e to ignore.\n",
              (unsigned long)lineno);
      exit(1);
    }
  }
  if (drop) {
    d.residues.resize(start);
    return;
  }

  /* repertoire_id: default when the column or the field is absent (db.cc:505-520) */
  const uint32_t rep = d.repertoires.intern(repertoire_id ? repertoire_id : default_rep);

  /* duplicate_count (db.cc:545-571) */
  uint64_t count = 1;
  if (duplicate_count && *duplicate_count) {
    char *end = nullptr;
    long v = strtol(duplicate_count, &end, 10);
    if (end && *end == 0 && v >= 1) {
      count = (uint64_t)v;
    } else {
      fprintf(log, "\n\nError: Illegal duplicate_count on line %lu: %s\n",
              (unsigned long)lineno, duplicate_count);
      exit(1);
    }
  } else if (!o.ignore_counts) {
    fprintf(log, "\n\nError: missing or empty duplicate_count on line %lu\n",
            (unsigned long)lineno);
    exit(1);
  }

  /* v_call, j_call (db.cc:578-631): required unless -g, interned either way */
  if (!o.ignore_genes && !(v_call && *v_call)) {
    fprintf(log, "\n\nError: missing or empty v_call value on line %lu\n", (unsigned long)lineno);
    exit(1);
  }
  const uint32_t vno = genes.v.intern(v_call ? v_call : "");
  if (!o.ignore_genes && !(j_call && *j_call)) {
    fprintf(log, "\n\nError: missing or empty j_call value on line %lu\n", (unsigned long)lineno);
    exit(1);
  }
  const uint32_t jno = genes.j.intern(j_call ? j_call : "");

  d.offsets.push_back(d.residues.size());
  d.repertoire.push_back(rep);
  d.count.push_back(count);
  d.v_gene.push_back(vno);
  d.j_gene.push_back(jno);
  d.total_count += count;
  if (len > d.longest) d.longest = len;
  if (len < d.shortest) d.shortest = len;
}

}  // namespace

void read_airr_tsv(const char *filename, const Options &o, GeneTables &genes,
                   const char *default_rep, FILE *log, RepertoireSet &d)
{
  FILE *fp = nullptr;
  if (strcmp(filename, "-") == 0) {
    int fd = dup(STDIN_FILENO);
    fp = fd < 0 ? nullptr : fdopen(fd, "rb");
  } else {
    fp = fopen(filename, "rb");
  }
  if (!fp) {
    fprintf(log, "\nError: Unable to open input data file (%s).\n", filename);
    exit(1);
  }

  d = RepertoireSet();
  d.offsets.push_back(0);

  char *line = nullptr;
  size_t cap = 0;
  uint64_t lineno = 0;
  bool have_header = false;
  bool any = false;
  Columns cols;
  std::vector<char *> fields;
  ssize_t n;
  while ((n = getline(&line, &cap, fp)) >= 0) {
    any = true;
    lineno++;
    /* LF, then CR of DOS files (db.cc:765-775, 824-836) */
    if (n > 0 && line[n - 1] == '\n') line[--n] = 0;
    if (n > 0 && line[n - 1] == '\r') line[--n] = 0;
    if (!have_header) {
      /* leading comment lines (db.cc:781-790) */
      if (line[0] == '#' || line[0] == '@')
        continue;
      parse_header(line, o, cols, log);
      have_header = true;
    } else {
      parse_line(line, lineno, o, cols, genes, default_rep, log, d, fields);
    }
  }
  free(line);
  fclose(fp);
  if (!any)
    fatal("Unable to read from the input file");   /* db.cc:758-759 */

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
