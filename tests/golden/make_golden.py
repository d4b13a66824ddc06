#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs oracle/_ref/compairr (the unmodified reference program compiled from
/root/reference by oracle/Makefile) on small inputs and records, per case, the
argv, the input files, the exit status and the matrix it wrote to stdout.
Inputs are (a) the reference's own test data files test/{seta,setb,setc}.tsv
(data, copied verbatim), (b) hand-written edge-case TSVs defined below and
(c) seeded random sets from compairr_amd.synth.

Run from the repository root in the build container (needs /root/reference):
    python tests/golden/make_golden.py
The GPU box only ever reads the committed results.
"""

import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from compairr_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "compairr")
REF_TEST = "/root/reference/test"
INPUTS = os.path.join(HERE, "inputs")
EXPECTED = os.path.join(HERE, "expected")

HDR = "repertoire_id\tsequence_id\tduplicate_count\tv_call\tj_call\tjunction\tjunction_aa\n"


def rows(lines, header=HDR, eol="\n"):
    return header.replace("\n", eol) + "".join(l + eol for l in lines)


EDGE_FILES = {
    # CRLF line ends (db.cc:832-836)
    "crlf.tsv": rows(["X1\ta\t2\tV1\tJ1\ttgtgct\tCASSL", "X2\tb\t3\tV1\tJ1\ttgtgca\tCASSV"], eol="\r\n"),
    # lower-case residues, U in nucleotides (db.cc:33-71)
    "lower.tsv": rows(["X1\ta\t2\tV1\tJ1\tugugcu\tcassl", "X1\tb\t5\tV1\tJ1\tTGTGCT\tCASSL",
                       "X2\tc\t7\tV1\tJ1\ttgugca\tcAsSv"]),
    # leading comment lines (db.cc:781-790)
    "comments.tsv": "# a comment\n@another\n#\tthird\n" + rows(
        ["X1\ta\t2\tV1\tJ1\ttgtgct\tCASSL", "X2\tb\t3\tV1\tJ1\ttgtgca\tCASSV"]),
    # no repertoire_id column -> id "1"/"2" (overlap.cc:619,715; db.cc:505-508)
    "norep.tsv": rows(["a\t2\tV1\tJ1\tCASSL", "b\t3\tV1\tJ1\tCASSV", "c\t4\tV2\tJ1\tCASSL"],
                      header="sequence_id\tduplicate_count\tv_call\tj_call\tjunction_aa\n"),
    # reordered + extra columns
    "reorder.tsv": rows(["CASSL\tfoo\tJ1\t2\tX1\tV1\ta", "CASSV\tbar\tJ1\t3\tX2\tV1\tb",
                         "CASSV\tbaz\tJ1\t11\tX1\tV1\tc"],
                        header="junction_aa\textra\tj_call\tduplicate_count\trepertoire_id\tv_call\tsequence_id\n"),
    # cdr3 / cdr3_aa columns (--cdr3)
    "cdr3.tsv": rows(["X1\ta\t2\tV1\tJ1\tgctagc\tASSL\tCASSLF", "X2\tb\t3\tV1\tJ1\tgctagt\tASSV\tCASSVF",
                      "X2\tc\t5\tV1\tJ1\tgctagc\tASSL\tCASSLW"],
                     header="repertoire_id\tsequence_id\tduplicate_count\tv_call\tj_call\tcdr3\tcdr3_aa\tjunction_aa\n"),
    # illegal symbols and empty sequences (-u, -e; db.cc:442-486)
    "unknown.tsv": rows(["X1\ta\t2\tV1\tJ1\ttgt\tCASSL", "X1\tb\t3\tV1\tJ1\ttgn\tCAXSL",
                         "X2\tc\t4\tV1\tJ1\ttgt\tCAS*L", "X2\td\t5\tV1\tJ1\t\t",
                         "X2\te\t6\tV1\tJ1\ttgt\tCASSL", "X3\tf\t6\tV1\tJ1\ttgt\tCASSV"]),
    # duplicates inside a repertoire, large counts (counts multiply)
    "dups.tsv": rows(["X1\ta\t3000000\tV1\tJ1\ttgt\tCASSL", "X1\tb\t5\tV1\tJ1\ttgt\tCASSL",
                      "X1\tc\t7\tV1\tJ1\ttgt\tCASSV", "X2\td\t4000000\tV1\tJ1\ttgt\tCASSL"]),
    # display order is strcmp on ids: R1 < R10 < R2 (overlap.cc:130-142)
    "order.tsv": rows(["R2\ta\t2\tV1\tJ1\ttgt\tCASSL", "R10\tb\t3\tV1\tJ1\ttgt\tCASSL",
                       "R1\tc\t5\tV1\tJ1\ttgt\tCASSV", "R10\td\t7\tV1\tJ1\ttgt\tCASSV"]),
    # indel corner cases: homopolymers, length 1, both ends (variants.cc:301-353)
    "indel_a.tsv": rows(["P\ta\t1\tV1\tJ1\ta\tA", "P\tb\t2\tV1\tJ1\taa\tAA", "P\tc\t3\tV1\tJ1\taaa\tAAA",
                         "P\td\t5\tV1\tJ1\taaaa\tAAAA", "Q\te\t7\tV1\tJ1\tca\tCA", "Q\tf\t11\tV1\tJ1\tac\tAC",
                         "Q\tg\t13\tV1\tJ1\tc\tC", "Q\th\t17\tV1\tJ1\tcac\tCAC", "Q\ti\t19\tV1\tJ1\taca\tACA"]),
    "indel_b.tsv": rows(["S\ta\t23\tV1\tJ1\taa\tAA", "S\tb\t29\tV1\tJ1\ta\tA", "T\tc\t31\tV1\tJ1\tcaa\tCAA",
                         "T\td\t37\tV1\tJ1\taac\tAAC", "T\te\t41\tV1\tJ1\tcc\tCC", "S\tf\t43\tV1\tJ1\taaaaa\tAAAAA",
                         "T\tg\t47\tV2\tJ1\taa\tAA", "S\th\t53\tV1\tJ2\tac\tAC"]),
    # empty duplicate_count is legal only with -f (db.cc:545-571)
    "nocount.tsv": rows(["X1\ta\t\tV1\tJ1\ttgt\tCASSL", "X2\tb\t\tV1\tJ1\ttgt\tCASSL", "X2\tc\t\tV1\tJ1\ttgt\tCASSV"]),
    # no sequence_id column: fatal for the first file of -x (db.cc:229)
    "noid.tsv": rows(["X1\t2\tV1\tJ1\tCASSL", "X1\t3\tV1\tJ1\tCASSV"],
                     header="repertoire_id\tduplicate_count\tv_call\tj_call\tjunction_aa\n"),
    # no gene columns at all: legal with -g (db.cc:231-232)
    "nogenes.tsv": rows(["X1\ta\t2\tCASSL", "X2\tb\t3\tCASSV", "X2\tc\t5\tCASSL"],
                        header="repertoire_id\tsequence_id\tduplicate_count\tjunction_aa\n"),
}


def random_files():
    """Seeded random inputs (c): aa/nt, several repertoires, planted neighbours."""
    out = {}
    for name, kw in {
        "rand_aa_a.tsv": dict(n=700, seed=11, prefix="A", n_repertoires=5, pool_size=200),
        "rand_aa_b.tsv": dict(n=600, seed=12, prefix="B", n_repertoires=4, pool_size=200),
        "rand_nt_a.tsv": dict(n=300, seed=21, prefix="A", n_repertoires=3, pool_size=80, nucleotides=True),
        "rand_nt_b.tsv": dict(n=250, seed=22, prefix="B", n_repertoires=3, pool_size=80, nucleotides=True),
    }.items():
        n = kw.pop("n")
        seed = kw.pop("seed")
        out[name] = (synth.make_set(n, seed, pool_seed=777, **kw), kw.get("nucleotides", False))
    for name, kw in {
        "tiny_aa_a.tsv": dict(n=120, seed=31, alphabet_size=20, letters=2, prefix="A"),
        "tiny_aa_b.tsv": dict(n=100, seed=32, alphabet_size=20, letters=2, prefix="B"),
        "tiny_nt_a.tsv": dict(n=150, seed=41, alphabet_size=4, letters=2, max_len=7, prefix="A"),
        "tiny_nt_b.tsv": dict(n=130, seed=42, alphabet_size=4, letters=3, max_len=7, prefix="B"),
        # dense neighbourhoods for -c: clusters of many sizes, long breadth-first chains
        "clus_aa.tsv": dict(n=1500, seed=61, alphabet_size=20, letters=3, min_len=3, max_len=8,
                            prefix="C"),
        "clus_nt.tsv": dict(n=1200, seed=62, alphabet_size=4, letters=4, min_len=4, max_len=9,
                            n_v=3, prefix="C"),
    }.items():
        n = kw.pop("n")
        seed = kw.pop("seed")
        out[name] = (synth.tiny_set(n, seed, **kw), kw["alphabet_size"] == 4)
    # single-repertoire query files for -x, sharing the pool of the rand_* sets
    out["rand_x_aa.tsv"] = (synth.make_set(400, 51, pool_seed=777, prefix="Q", n_repertoires=1,
                                           pool_size=200), False)
    out["rand_x_nt.tsv"] = (synth.make_set(200, 52, pool_seed=777, prefix="Q", n_repertoires=1,
                                           pool_size=80, nucleotides=True), True)
    return out


def late_file(n, seed, prefix):
    """A file of ~450 KB whose repertoire ids, V and J genes make their FIRST
    appearance spread over the whole file, i.e. in different 64-KiB ranges of a
    threaded reader: first-appearance numbering (db.cc:510-520, 592-631) defines
    the matrix layout, so a reader that cuts the file into ranges has to merge
    them in file order."""
    import numpy as np
    s = synth.make_set(n, seed, pool_seed=4242, prefix=prefix, n_repertoires=14, pool_size=3500)
    rng = np.random.default_rng(seed)
    # a sequence may only appear once "its" repertoire / genes are unlocked: rank
    # every row by the latest of its three ranks, then order rows by that rank
    rank = np.maximum.reduce([s.repertoire.astype(np.int64) * n // 14,
                              (s.v_gene.astype(np.int64) % 20) * n // 20,
                              s.j_gene.astype(np.int64) * n // 13])
    order = np.argsort(rank + rng.integers(0, n // 40, size=n), kind="stable")
    lines = []
    for k, i in enumerate(order):
        seq = "".join(s.alphabet[c] for c in s.residues[int(s.offsets[i]):int(s.offsets[i + 1])])
        lines.append("%s%02d\t%s%d\t%d\t%s\t%s\t%s" % (
            prefix, int(s.repertoire[i]), prefix.lower(), k, int(s.count[i]),
            s.v_names[int(s.v_gene[i])], s.j_names[int(s.j_gene[i])], seq))
    return rows(lines, header="repertoire_id\tsequence_id\tduplicate_count\tv_call\tj_call\tjunction_aa\n")


def cases():
    c = []

    def add(name, args, files, note="", cmd="-m", pairs=False):
        c.append({"name": name, "args": args, "files": files, "note": note, "cmd": cmd,
                  "pairs": pairs})

    # (a) the reference's own test + README examples + the survey's KAT table
    add("ref_test_sh", "-d 1 -i", ["seta.tsv", "setb.tsv"], "test/test.sh:9, test/expected.tsv")
    for k, (args, files) in enumerate([
        ("-d 1", ["seta.tsv", "setb.tsv"]), ("", ["seta.tsv", "setb.tsv"]),
        ("-d 1 -a", ["seta.tsv", "setb.tsv"]), ("-d 1 -g -n", ["seta.tsv", "setb.tsv"]),
        ("-d 2 -n -g", ["seta.tsv", "setb.tsv"]), ("-d 1", ["setb.tsv"]),
        ("-d 1 -s ratio -t 4", ["setb.tsv"]), ("-s jaccard", ["seta.tsv", "setb.tsv"]),
        ("-s MH", ["setb.tsv"]), ("-d 3", ["seta.tsv", "setb.tsv"]),
        ("-d 1", ["setb.tsv", "setb.tsv"]), ("-d 1 -i", ["seta.tsv", "setc.tsv"]),
        ("-d 2", ["setb.tsv", "setc.tsv"]), ("-d 1 -f", ["seta.tsv", "setb.tsv"]),
    ]):
        add("ref_kat_%02d" % k, args, files)

    # (b) edge inputs
    add("edge_crlf", "-d 1", ["crlf.tsv", "lower.tsv"])
    add("edge_lower_aa", "-d 1", ["lower.tsv"])
    add("edge_lower_nt", "-d 1 -n", ["lower.tsv", "crlf.tsv"])
    add("edge_comments", "-d 1 -i", ["comments.tsv", "crlf.tsv"])
    add("edge_norep", "-d 1", ["norep.tsv", "crlf.tsv"])
    add("edge_norep_both", "-d 1", ["norep.tsv", "norep.tsv"])
    add("edge_norep_two", "", ["crlf.tsv", "norep.tsv"])
    add("edge_reorder", "-d 1 -s min", ["reorder.tsv", "crlf.tsv"])
    add("edge_cdr3_aa", "-d 1 --cdr3", ["cdr3.tsv"])
    add("edge_cdr3_nt", "-d 1 --cdr3 -n", ["cdr3.tsv"])
    add("edge_cdr3_junction", "-d 1", ["cdr3.tsv"])
    add("edge_unknown_u_e", "-d 1 -u -e", ["unknown.tsv"])
    add("edge_unknown_u_e_nt", "-d 1 -u -e -n", ["unknown.tsv", "crlf.tsv"])
    add("edge_dups", "", ["dups.tsv"])
    add("edge_dups_d1", "-d 1", ["dups.tsv", "dups.tsv"])
    add("edge_dups_mh", "-s mh", ["dups.tsv"])
    add("edge_order", "-d 1", ["order.tsv"])
    add("edge_order_a", "-d 1 -a", ["order.tsv", "dups.tsv"])
    add("edge_nocount_f", "-d 1 -f", ["nocount.tsv"])
    add("edge_nogenes_g", "-d 1 -g", ["nogenes.tsv", "crlf.tsv"])
    for score in ("product", "min", "max", "mean", "ratio"):
        add("edge_indel_%s" % score, "-d 1 -i -s %s" % score, ["indel_a.tsv", "indel_b.tsv"])
    add("edge_indel_self", "-d 1 -i", ["indel_a.tsv"])
    add("edge_indel_nt", "-d 1 -i -n", ["indel_a.tsv", "indel_b.tsv"])
    add("edge_indel_g", "-d 1 -i -g", ["indel_b.tsv", "indel_a.tsv"])
    add("edge_indel_f", "-d 1 -i -f", ["indel_a.tsv", "indel_b.tsv"])
    add("edge_indel_d2", "-d 2", ["indel_a.tsv", "indel_b.tsv"])
    add("edge_indel_d0", "", ["indel_a.tsv", "indel_b.tsv"])

    # errors: only the exit status is compared
    add("err_illegal_char", "-d 1", ["unknown.tsv"], "exit 1")
    add("err_empty_seq", "-d 1 -u", ["unknown.tsv"], "exit 1")
    add("err_nocount", "-d 1", ["nocount.tsv"], "exit 1")
    add("err_nogenes", "-d 1", ["nogenes.tsv"], "exit 1")
    add("err_indel_d2", "-d 2 -i", ["seta.tsv", "setb.tsv"], "exit 1")
    add("err_mh_d1", "-d 1 -s mh", ["seta.tsv", "setb.tsv"], "exit 1")
    add("err_twice", "-d 1 -d 1", ["seta.tsv", "setb.tsv"], "exit 1")
    add("err_score", "-s bogus", ["seta.tsv", "setb.tsv"], "exit 1")
    add("err_missing_file", "-d 1", ["seta.tsv", "does_not_exist.tsv"], "exit 1")

    # (c) random sets x option matrix
    for fam, nt in (("rand_aa", False), ("rand_nt", True), ("tiny_aa", False), ("tiny_nt", True)):
        a, b = "%s_a.tsv" % fam, "%s_b.tsv" % fam
        n = " -n" if nt else ""
        for d in ("", "-d 1", "-d 1 -i", "-d 2"):
            tag = d.replace("-", "").replace(" ", "") or "d0"
            for g in ("", " -g"):
                add("%s_%s%s" % (fam, tag, "_g" if g else ""), (d + g + n).strip(), [a, b])
            add("%s_%s_f" % (fam, tag), (d + " -f" + n).strip(), [a, b])
            add("%s_%s_self" % (fam, tag), (d + n).strip(), [a])
        for score in ("min", "max", "mean"):
            add("%s_d1_%s" % (fam, score), ("-d 1 -s %s%s" % (score, n)).strip(), [a, b])
        add("%s_jaccard" % fam, ("-s jaccard" + n).strip(), [a, b])
        add("%s_mh" % fam, ("-s MH" + n).strip(), [a, b])
        add("%s_mh_self" % fam, ("-s MH -a" + n).strip(), [b])
        add("%s_d1_t3" % fam, ("-d 1 -i -t 3" + n).strip(), [a, b])
    add("tiny_aa_d3", "-d 3", ["tiny_aa_a.tsv", "tiny_aa_b.tsv"], "d > 2: reference's brute-force path")

    # (d) -x / --existence: rows are the sequences of the first file
    add("x_readme_ex2", "-d 1 -f", ["setc.tsv", "setb.tsv"], "README.md:557-561", cmd="-x")
    add("x_ref_a", "-d 1 -a", ["setc.tsv", "setb.tsv"], cmd="-x")
    add("x_ref_min_i", "-d 1 -i -s min", ["setc.tsv", "setb.tsv"], cmd="-x")
    add("x_ref_nt", "-d 2 -n -g", ["setc.tsv", "seta.tsv"], cmd="-x")
    add("x_same_file", "-d 1", ["setc.tsv", "setc.tsv"], cmd="-x")
    for fam, nt in (("aa", False), ("nt", True)):
        q, r = "rand_x_%s.tsv" % fam, "rand_%s_b.tsv" % fam
        n = " -n" if nt else ""
        for d in ("", "-d 1", "-d 1 -i", "-d 2"):
            tag = d.replace("-", "").replace(" ", "") or "d0"
            add("x_%s_%s" % (fam, tag), (d + n).strip(), [q, r], cmd="-x")
        add("x_%s_d1_g_f" % fam, ("-d 1 -g -f" + n).strip(), [q, r], cmd="-x")
        add("x_%s_d1_a_max" % fam, ("-d 1 -a -s max" + n).strip(), [q, r], cmd="-x")
        add("x_%s_d1_mean_t3" % fam, ("-d 1 -i -s mean -t 3" + n).strip(), [q, r], cmd="-x")
    # (e) -p / --pairs (+ -k, --distance, --no-matrix): the pairs file, lines sorted
    #     ("the order of the lines is unspecified", README.md:163)
    add("p_readme_ex1", "-d 1", ["seta.tsv", "setb.tsv"], "README.md:440-444", pairs=True)
    add("p_ref_indel_dist_keep", "-d 1 -i --distance -k sequence,productive,nosuchcolumn",
        ["seta.tsv", "setb.tsv"], pairs=True)
    add("p_ref_nt_nomatrix", "-d 1 -n -g --no-matrix", ["seta.tsv", "setb.tsv"], pairs=True)
    add("p_x_readme_ex2", "-d 1 -f", ["setc.tsv", "setb.tsv"], "README.md:568-573", cmd="-x",
        pairs=True)
    add("p_self", "-d 1", ["setb.tsv"], pairs=True)
    add("p_dups_d0", "", ["dups.tsv"], pairs=True)
    add("p_noid_g", "-d 1 -g", ["nogenes.tsv", "crlf.tsv"], pairs=True)
    add("p_noid_set1", "-d 1", ["noid.tsv", "crlf.tsv"], "no sequence_id column: empty ids",
        pairs=True)
    for fam, nt in (("rand_aa", False), ("rand_nt", True), ("tiny_aa", False), ("tiny_nt", True)):
        a, b = "%s_a.tsv" % fam, "%s_b.tsv" % fam
        n = " -n" if nt else ""
        add("p_%s_d1i_dist" % fam, ("-d 1 -i --distance" + n).strip(), [a, b], pairs=True)
        add("p_%s_d2_dist" % fam, ("-d 2 --distance" + n).strip(), [a, b], pairs=True)
        add("p_%s_d0_self" % fam, n.strip(), [a], pairs=True)
    add("p_x_aa_d1i", "-d 1 -i --distance", ["rand_x_aa.tsv", "rand_aa_b.tsv"], cmd="-x", pairs=True)
    # (f) -c / --cluster: single-linkage clusters of ONE file; the order of the lines is the
    #     reference's breadth-first order (cluster.cc:200-223, 376-452)
    add("c_ref_seta_d1", "-d 1", ["seta.tsv"], cmd="-c")
    add("c_ref_setb_d1i", "-d 1 -i", ["setb.tsv"], cmd="-c")
    add("c_ref_setb_d2_nt_g", "-d 2 -n -g", ["setb.tsv"], cmd="-c")
    add("c_dups_d0", "", ["dups.tsv"], cmd="-c")
    add("c_indel_a_d1i", "-d 1 -i", ["indel_a.tsv"], cmd="-c")
    add("c_noid_g", "-d 1 -g", ["nogenes.tsv"], "no sequence_id / gene columns", cmd="-c")
    add("c_cdr3", "-d 1 --cdr3", ["cdr3.tsv"], cmd="-c")
    for fam, nt in (("rand_aa_a", False), ("rand_nt_a", True), ("tiny_aa_a", False),
                    ("tiny_nt_b", True), ("clus_aa", False), ("clus_nt", True)):
        n = " -n" if nt else ""
        for d in ("", "-d 1", "-d 1 -i", "-d 2"):
            tag = d.replace("-", "").replace(" ", "") or "d0"
            add("c_%s_%s" % (fam, tag), (d + n).strip(), [fam + ".tsv"], cmd="-c")
        add("c_%s_d1i_g_t3" % fam, ("-d 1 -i -g -t 3" + n).strip(), [fam + ".tsv"], cmd="-c")
        add("c_%s_d2_g" % fam, ("-d 2 -g" + n).strip(), [fam + ".tsv"], cmd="-c")
    # (g) files of several 64-KiB reader ranges, ids and genes appearing late (db.cc:510-520):
    #     the threaded reader's merge, at -t 1 / 3 / 8
    for t in (1, 3, 8):
        add("late_aa_d1i_t%d" % t, "-d 1 -i -t %d" % t, ["late_a.tsv", "late_b.tsv"])
    add("late_aa_d1_self_t8", "-d 1 -t 8", ["late_b.tsv"])
    add("late_aa_d0_g_t3", "-g -t 3 -s min", ["late_b.tsv", "late_a.tsv"])
    add("err_c_two_files", "-d 1", ["seta.tsv", "setb.tsv"], "exit 1", cmd="-c")
    add("err_c_pairs", "-d 1", ["seta.tsv"], "exit 1", cmd="-c", pairs=True)
    add("err_c_alternative", "-d 1 -a", ["seta.tsv"], "exit 1", cmd="-c")
    add("err_c_score", "-d 1 -s min", ["seta.tsv"], "exit 1", cmd="-c")
    add("err_p_keep_without_pairs", "-d 1 -k sequence", ["seta.tsv", "setb.tsv"], "exit 1")
    add("err_p_keep_bad_list", "-d 1 -k a,,b", ["seta.tsv", "setb.tsv"], "exit 1", pairs=True)
    add("err_x_multi_rep", "-d 1", ["seta.tsv", "setb.tsv"], "exit 1", cmd="-x")
    add("err_x_one_file", "-d 1", ["setc.tsv"], "exit 1", cmd="-x")
    add("err_x_mh", "-s MH", ["setc.tsv", "setb.tsv"], "exit 1", cmd="-x")
    add("err_x_no_sequence_id", "-d 1", ["noid.tsv", "crlf.tsv"], "exit 1", cmd="-x")
    return c


def main():
    if not os.path.exists(REF):
        sys.exit("build the reference first: make -C oracle ref")
    shutil.rmtree(INPUTS, ignore_errors=True)
    shutil.rmtree(EXPECTED, ignore_errors=True)
    os.makedirs(INPUTS)
    os.makedirs(EXPECTED)
    for f in ("seta.tsv", "setb.tsv", "setc.tsv"):
        shutil.copy(os.path.join(REF_TEST, f), os.path.join(INPUTS, f))
    for name, text in EDGE_FILES.items():
        with open(os.path.join(INPUTS, name), "w", newline="") as fh:
            fh.write(text)
    for name, (s, nt) in random_files().items():
        # both junction and junction_aa columns are not needed: one file per alphabet
        s.write_tsv(os.path.join(INPUTS, name), nucleotides=nt)

    for name, (n, seed, prefix) in {"late_a.tsv": (11000, 71, "A"), "late_b.tsv": (10000, 72, "B")}.items():
        with open(os.path.join(INPUTS, name), "w", newline="") as fh:
            fh.write(late_file(n, seed, prefix))

    manifest = []
    for case in cases():
        logf = os.path.join(HERE, "_log.tmp")
        argv = [REF, case["cmd"]] + case["files"] + case["args"].split() + ["-l", logf]
        pairsf = os.path.join(HERE, "_pairs.tmp")
        if case["pairs"]:
            argv += ["-p", pairsf]
        p = subprocess.run(argv, cwd=INPUTS, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        if case["pairs"] and p.returncode == 0:
            lines = open(pairsf, "rb").read().splitlines(keepends=True)
            with open(os.path.join(EXPECTED, case["name"] + ".pairs.tsv"), "wb") as fh:
                fh.writelines(lines[:1] + sorted(lines[1:]))
        if os.path.exists(pairsf):
            os.remove(pairsf)
        entry = dict(case)
        entry["exit"] = p.returncode
        # the duplicate warnings of the log are part of the contract
        entry["warnings"] = [l.rstrip("\n") for l in open(logf, errors="replace")
                             if l.startswith("Warning:")] if os.path.exists(logf) else []
        if os.path.exists(logf):
            os.remove(logf)
        if p.returncode == 0:
            with open(os.path.join(EXPECTED, case["name"] + ".tsv"), "wb") as fh:
                fh.write(p.stdout)
        elif not case["name"].startswith("err_"):
            sys.exit("reference failed on %s: %s" % (case["name"], argv))
        manifest.append(entry)
    with open(os.path.join(HERE, "manifest.json"), "w") as fh:
        json.dump(manifest, fh, indent=1)
    print("%d cases written" % len(manifest))


if __name__ == "__main__":
    main()
