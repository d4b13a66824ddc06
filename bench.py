#!/usr/bin/env python3
"""bench.py -- query sequences/sec through the --matrix hot path on MI355X.

One "step" = one QUERY SET through the hot path: the set's arrays (the SoA of
include/compairr_hip.h, already resident in HBM when the timed region starts) are laid
out for the kernels (cmpr_set_queries_device: keys, regrouping by filter slice, tiles,
items -- per-query-set GPU work that lies between "first kernel launch" and "matrix
ready" of SURVEY 8d) and then passed once over the reference index (variant
enumeration -> Zobrist hash -> Bloom test -> record-table walk -> verify -> matrix
accumulate), plus -- at N > 1 -- the RCCL all-reduce of the repertoire matrix.
`value` = queries x steps / wall time of exactly that.  `--step resident` times the
launches alone over one laid-out set (rounds 1-5's headline; reported by every run as
`value_resident_step`).  Default workload = BASELINE.json configs[2]: synthetic
10M-vs-10M CDR3aa, d = 1, substitutions only.

N > 1: one rank per GPU -- launched by torch.distributed.run, or, when `python3 bench.py --gpus N` is typed
without a launcher (no RANK in the environment), by bench.py itself: it starts the same
torch.distributed.run command as a CHILD process before anything has touched the GPU and relays
rank 0's line and the exit code (compairr_amd.dist.spawn_ranks).  The reference set
(record table + filter) is replicated on every GPU and the query set is sharded:
  --scaling strong (default)  the SAME seeded 10M queries, total work fixed
                              (BASELINE configs[3]).  --shard-by queries (default, the
                              north star's split): N contiguous query shards, the way
                              overlap.cc:421-433 hands out query chunks; a rank lays out
                              and works on its N-th of the set, no data-path collective
                              but the matrix reduce.  --shard-by work: a rank takes the
                              work filed under its share of the filter slices (library
                              tunables work_shard_index / _count) and the queries go
                              where their work is: every rank uploads and keys a
                              contiguous N-th of them, one all-to-all over xGMI moves
                              the records, every rank lays out what it received
                              (compairr_amd.dist.exchange_queries; --layout replicated:
                              every rank uploads and keys all of them) -- the shorter
                              resident step, the dearer query set.
  --scaling weak              every rank its own 10M-query shard.
The only collective of the default split is one all-reduce (sum, int64) of the R1 x R2
matrix per step; it runs on a second stream, ordered by events, so that the all-reduce of a
step overlaps the work of the next (two matrices).  In strong mode the reduced matrix is
the N = 1 matrix; it is compared, at every N, with the matrix the reference binary printed
for the same workload (tests/golden/full_size.json, "parity_vs_reference_full_size"); a
mismatch there or on the CPU sample ends the run with exit status 1.

Prints ONE JSON line on rank 0 (see the contract in the task description), with
`roofline` (the dominant kernel against its binding unit, from the committed
rocprofv3 counters and the live HIP-event time) and `cpu_baseline` (the
reference program timed on this box's host cores on a bounded sample of the same
workload).
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def recorded_workload(args):
    """The reference binary's own matrix for this workload, if tests/golden/full_size.json
    (written in the build container by tests/golden/make_full_size.py) holds it: the parity
    anchor of every run at every N."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _full_size
    if args.scaling != "strong":
        return None
    return _full_size.by_bench_args(args.refs, args.queries, args.differences, args.indels,
                                    args.nucleotides, args.ignore_genes, args.self_cmp, args.law, args.repertoires)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--refs", type=int, default=10_000_000, help="set-2 sequences")
    p.add_argument("--queries", type=int, default=10_000_000,
                   help="set-1 sequences (strong: in total; weak: per GPU)")
    p.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    p.add_argument("--shard-by", choices=["work", "queries"], default="queries",
                   help="strong scaling at N > 1: what the ranks divide (see the module docstring)")
    p.add_argument("--step", choices=["query-set", "resident"], default="query-set",
                   help="what one timed step is: a query set from its device arrays to the matrix (layout + "
                        "launch; the default), or one launch over the resident layout (rounds 1-5)")
    p.add_argument("--layout", choices=["routed", "replicated"], default="routed",
                   help="--shard-by work under torch.distributed.run: every rank uploads and keys its N-th of "
                        "the queries and the records move to the ranks that work on them (default), or every "
                        "rank uploads and keys all queries (round 3)")
    p.add_argument("--law", choices=["uniform", "cdr3"], default="uniform",
                   help="generator law (compairr_amd/synth.py): the BASELINE one (uniform residues) or the "
                        "robustness workload (conserved ends, skewed composition, Zipf clone sizes)")
    p.add_argument("--repertoires", type=int, default=16, help="repertoires per set")
    p.add_argument("--differences", "-d", type=int, default=1)
    p.add_argument("--indels", action="store_true")
    p.add_argument("--nucleotides", action="store_true")
    p.add_argument("--ignore-genes", action="store_true")
    p.add_argument("--cpu-sample", type=int, default=0,
                   help="queries in the CPU-baseline sample (0 = auto, -1 = skip)")
    p.add_argument("--cpu-refs", type=int, default=0,
                   help="reference sequences in the CPU-baseline sample (0 = all of them).  The reference binary "
                        "reads its sets from TSV files: 100M reference sequences take it > 10 minutes of parsing "
                        "and indexing before its per-query loop starts -- cfg5's baseline is timed against the "
                        "first N of them (the loop's cost per query does not depend on their number but for cache "
                        "effects: one hash-table lookup per variant)")
    p.add_argument("--cpu-kind", choices=["auto", "reference", "port"], default="auto",
                   help="CPU baseline: the reference binary (oracle/_ref, fed TSV files) or the oracle "
                        "port (oracle/liboracle.so, sets in memory); auto = the binary when present")
    p.add_argument("--self", dest="self_cmp", action="store_true",
                   help="one-file mode: the queries are the reference set itself")
    p.add_argument("--tunable", action="append", default=[], metavar="NAME=VALUE")
    p.add_argument("--skip-host-layout", action="store_true",
                   help="do not measure cmpr_set_queries from HOST buffers (value_incl_layout becomes null): under "
                        "rocprofv3 --kernel-trace --stats every launch of the layout's kernels is then a full-size one "
                        "of a timed step, and the trace's averages are the bench line's per-kernel times")
    p.add_argument("--launcher", choices=["auto", "always", "never"], default="auto",
                   help="auto: `--gpus N` with N > 1 and no RANK in the environment starts its N ranks itself "
                        "(torch.distributed.run as a child process, before anything touches the GPU); always: "
                        "also at N = 1 (the RCCL path at world size 1); never: expect a launcher around it")
    return p.parse_args()


def _opt_argv(args):
    argv = ["-d", str(args.differences)]
    if args.indels:
        argv.append("-i")
    if args.nucleotides:
        argv.append("-n")
    if args.ignore_genes:
        argv.append("-g")
    return argv


def cpu_baseline(ref, queries, opt, sample, args):
    """The reference algorithm on this box's host cores (rank 0, N = 1 only), on a
    bounded sample: the first `sample` queries of the workload against the full
    reference set.

    kind "reference": the unmodified reference program (oracle/_ref/compairr,
    compiled from /root/reference by oracle/Makefile and shipped as a binary)
    is fed the sample as AIRR TSV files and its own 'Analysing:' log time is
    taken -- exactly the per-query loop, overlap.cc:906-938.
    kind "port": the oracle port (oracle/liboracle.so) when the binary is absent.
    Returns (dict, integer matrix, sample set, print format, reference set used)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle
    cores = os.cpu_count() or 1
    threads = max(1, min(cores, 256))           # the reference caps -t at 256
    q = queries.subset(slice(0, sample))
    if args.cpu_refs and args.cpu_refs < ref.n:
        ref = ref.subset(slice(0, args.cpu_refs))        # (the parity check below runs on the same subset)
    exe = os.path.join(ROOT, "oracle", "_ref", "compairr")
    if os.path.exists(exe) and args.cpu_kind != "port":
        import re
        import subprocess
        import tempfile
        with tempfile.TemporaryDirectory(prefix="cmpr_bench_") as tmp:
            fq, fr = os.path.join(tmp, "q.tsv"), os.path.join(tmp, "r.tsv")
            q.write_tsv_fast(fq, args.nucleotides)
            ref.write_tsv_fast(fr, args.nucleotides)
            log, out = os.path.join(tmp, "log"), os.path.join(tmp, "out.tsv")
            t0 = time.time()
            p = subprocess.run([exe, "-m", fq, fr, "-t", str(threads), "-l", log, "-o", out]
                               + _opt_argv(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            wall = time.time() - t0
            if p.returncode == 0:
                text = open(log).read()
                sec = float(re.search(r"Analysing:\s+100% \(([0-9.]+)s\)", text).group(1))
                rows = [l.rstrip("\n").split("\t") for l in open(out)]
                cols = rows[0][1:]
                cell = {(r[0], cid): float(x) for r in rows[1:] for cid, x in zip(cols, r[1:])}
                m = np.zeros((q.n_repertoires, ref.n_repertoires))
                for i, a in enumerate(q.repertoire_ids):
                    for j, b in enumerate(ref.repertoire_ids):
                        m[i, j] = cell.get((a, b), 0.0)
                return ({"value": q.n / sec, "unit": "query sequences/s", "cores": threads,
                         "kind": "reference", "reference_sequences_in_sample": int(ref.n),
                         "sample": "CompAIRR 1.13.0 binary (oracle/_ref), -t %d, first %d queries "
                                   "of the workload vs all %d reference sequences; its own "
                                   "'Analysing:' time %.2f s (whole run incl. TSV parse %.1f s)"
                                   % (threads, q.n, ref.n, sec, wall)},
                        m, q, "%.10g", ref)
    m, st = _oracle.overlap(q, ref, opt, threads=threads)
    rate = q.n / st.seconds_analysis if st.seconds_analysis > 0 else 0.0
    return ({"value": rate, "unit": "query sequences/s", "cores": threads, "kind": "port",
             "sample": "first %d queries of the workload vs all %d reference sequences; "
                       "per-query loop only (reference 'Analysing:' region) %.2f s, "
                       "index build %.2f s excluded" % (q.n, ref.n, st.seconds_analysis,
                                                        st.seconds_index)},
            _oracle.integer_cells(m, opt).astype(np.float64), q, None, ref)


def roofline(workload, st, probe_ms, kernel_ms, kernel_name):
    """The probe kernel against the unit that binds it.

    Per-launch work of each unit at N = 1 comes from the committed rocprofv3 counter
    summary of THIS workload (profiles/roofline_inputs.json, one entry per workload,
    written by tools/pmc_summary.py: wave-level VALU instructions, LDS-array cycles, HBM
    bytes per the guide's FETCH_SIZE / WRITE_SIZE recipe), scaled by this rank's share
    of the step (live filter reads / the profile's: 1 at N = 1, ~1/N for a shard);
    the time is the live HIP-event time of the kernel.  Peaks, all from
    /opt/skills/guides/MI355X_MICROARCH.md: HBM 8 TB/s; LDS one array cycle per CU and
    clock; VALU = SIMDs x nominal clock / 2 (a wave64 instruction issues over 2 cycles per
    SIMD).  `bound` is the unit with the highest utilisation, `frac` that utilisation.
    (`frac_mix_priced`: the same vector-instruction rate against a peak priced with this
    repository's own per-class issue costs -- tools/calib.hip, weighted over the kernel's
    STATIC instruction listing by tools/isa_mix.py -- rounds 3-5's `frac`; secondary.)
    `counters_stale` says the library sources have changed since the counters were taken.
    `algorithmic_equiv` keeps SURVEY 8d's figure (8 bytes per variant, as the reference
    reads its filter): not a physical rate -- this kernel answers a row of variants with one
    LDS read."""
    t = probe_ms * 1e-3
    out = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
           "traffic": None, "kernel": kernel_name,
           "kernel_ms": probe_ms, "step_kernels_ms": kernel_ms,
           "resolve_kernel_ms": kernel_ms - probe_ms,
           "algorithmic_bytes_per_launch": st.algorithmic_bytes,
           "algorithmic_equiv": {"GB/s": st.algorithmic_bytes / (kernel_ms * 1e-3) / 1e9,
                                 "of_hbm_peak": st.algorithmic_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "note": "SURVEY 8d bytes (8 B per variant) / probe + resolve time; NOT a rate of this "
                                         "kernel: it exceeds 1 because a row of variants is answered by one LDS read, "
                                         "not by 8 bytes of HBM each"},
           "variants_per_launch": st.variants, "filter_reads_per_launch": st.filter_reads,
           "bloom_positive_per_launch": st.bloom_positive, "pairs_per_launch": st.matches}
    path = os.path.join(ROOT, "profiles", "roofline_inputs.json")
    try:
        with open(path) as fh:
            inp = json.load(fh)
    except Exception:
        inp = None
    k = (inp or {}).get("workloads", {}).get(workload)
    if not k:
        out["note"] = "no committed counter summary for this workload: utilisation not priced"
        return out
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from csrc_hash import csrc_hash
    out["counters_stale"] = bool(inp.get("csrc_sha256") != csrc_hash(ROOT))
    cal = inp["calibration"]
    # this rank's share of the profiled (N = 1) launch: by filter reads, else by variant tests
    share = 1.0
    if k.get("filter_reads") and st.filter_reads:
        share = st.filter_reads / k["filter_reads"]
    elif k.get("variants") and st.variants:
        share = st.variants / k["variants"]
    units = {}
    mix_priced = None
    if k.get("valu_insts"):
        # the guide's flat peak: a wave64 instruction issues over 2 cycles per SIMD (>= 2 waves per SIMD)
        peak = cal["simds"] * cal["nominal_clock_hz"] / 2.0
        units["valu"] = (k["valu_insts"] * share / t, peak, "wave-instructions/s")
        cyc = k.get("mix_cycles_per_valu_inst")
        if cyc:
            mix_priced = units["valu"][0] / (cal["simds"] * cal["nominal_clock_hz"] / cyc)
    if k.get("lds_active_cycles"):
        peak = cal["cus"] * cal["nominal_clock_hz"]        # (both units at the nominal clock)
        units["lds"] = (k["lds_active_cycles"] * share / t, peak, "LDS-array cycles/s")
    if k.get("hbm_bytes"):
        units["hbm"] = (k["hbm_bytes"] * share / t / 1e9, HBM_PEAK_GBS, "GB/s")
        out["traffic"] = k["hbm_bytes"] * share
    if not units:
        return out
    best = max(units, key=lambda u: units[u][0] / units[u][1])
    a, p, unit = units[best]
    out.update({"bound": best, "achieved": a, "peak": p, "unit": unit, "frac": a / p,
                "frac_mix_priced": mix_priced,
                "utilisation": {u: v[0] / v[1] for u, v in units.items()},
                "share_of_the_profiled_launch": share,
                "valu_issue_cycles_per_instruction_static_mix": k.get("mix_cycles_per_valu_inst"),
                # SQ_ACTIVE_INST_* / SQ_BUSY_CU_CYCLES of the committed profile run (not live):
                # vector instructions per CU cycle -- 1.0 = one per SIMD every 4 cycles
                "busy_fraction_from_counters": k.get("busy_fraction_from_counters"),
                "counters_from": k.get("source"),
                "note": "achieved = per-launch work of the binding unit (committed rocprofv3 PMC summary "
                        "of this workload x this rank's share) / live HIP-event time of the probe kernel; "
                        "peaks from MI355X_MICROARCH.md"})
    return out


def layout_rooflines(workload, n, residues, slots, items, indels, times):
    """The layout's three big kernels against the HBM roofline (they move bytes and nothing else):
    achieved = ALGORITHMIC bytes of the kernel -- what its contract makes it read and write once, DESIGN.md
    section 3 -- / its live HIP-event time; traffic = the committed FETCH_SIZE / WRITE_SIZE counters of the same
    kernel when profiles/roofline_inputs.json holds them.
      keys_kernel     per query: reads its residues, offset 8, v 4, j 4, repertoire 4, count 8; writes group 4,
                      rank 4, hash 8 (24 with -i), class key 4; one 4-byte counter read-modify-write (8)
      scatter_kernel  per query: the same reads + group, rank, hash(es), class key, group base 4; writes the
                      64-byte record; per item 16
      fill_tiles      per slot (padding lanes included): reads the 64-byte record; writes residues (position-
                      major, ~the query's length rounded up to 4), length 2, hash 8 (24 with -i), class key 4
                      (amino acids on the row filter: genes, repertoire and count stay in the record)"""
    hb = 24 if indels else 8
    alg = {"keys": residues + n * (28 + 8 + hb + 4 + 8),
           "scatter": residues + n * (28 + 8 + hb + 4 + 4 + 64) + items * 16,
           "tiles": slots * (64 + 2 + hb + 4) + (residues + 2 * n) * slots // max(n, 1)}
    path = os.path.join(ROOT, "profiles", "roofline_inputs.json")
    try:
        with open(path) as fh:
            prof = json.load(fh).get("workloads", {}).get(workload, {}).get("layout_kernels", {})
    except Exception:
        prof = {}
    out = {}
    for k, name in (("keys", "keys_kernel"), ("scatter", "scatter_kernel"), ("tiles", "fill_tiles_kernel")):
        ms = times.get(k)
        if not ms:
            continue
        a = alg[k] / (ms * 1e-3) / 1e9
        out[k] = {"bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                  "traffic": (prof.get(name) or {}).get("hbm_bytes"), "kernel": name, "kernel_ms": ms,
                  "algorithmic_bytes_per_launch": alg[k]}
    return out


LAYOUT_PARTS = ("keys", "sizes", "scatter", "tiles", "order")


def layout_kernel_times(h):
    """HIP-event times (ms) of the last cmpr_set_queries_device, recorded by the library on its own stream
    (tunable layout_timing): keys_kernel | per-slice sizes, scans, the host's round trip, slices_kernel |
    scatter_kernel | fill_tiles_kernel | item chunks and the chunk order."""
    return {k: h.get_tunable("layout_%s_us" % k) / 1e3 for k in LAYOUT_PARTS}


def start_ranks(args):
    """`python3 bench.py --gpus N` as the driver types it for N = 1, at N > 1: start the N ranks as a child
    (compairr_amd.dist.spawn_ranks -> torch.distributed.run), relay rank 0's JSON line and the child's exit
    code.  Nothing here initialises the GPU (torch.cuda.device_count() does not on this image), and this
    process is never replaced by another program."""
    import torch
    from compairr_amd.dist import spawn_ranks
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.exit("bench.py: --gpus %d but this box shows %d HIP device%s"
                 % (args.gpus, have, "" if have == 1 else "s"))
    argv = [a for a in sys.argv[1:]]
    # the child ranks must not start ranks of their own
    argv += ["--launcher", "never"]
    sys.exit(spawn_ranks(os.path.abspath(__file__), argv, args.gpus))


def main():
    args = parse_args()
    if "RANK" not in os.environ and args.launcher != "never" and (args.gpus > 1 or args.launcher == "always"):
        start_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist
    from compairr_amd import HipOverlap, Options, synth
    from compairr_amd.dist import exchange_queries, shard_bounds

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run (RANK set) the RCCL path is taken even at world
    # size 1, so that `--gpus 1` launched that way exercises the same code as N > 1
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (--launcher never: start it with torch.distributed.run)"
                 % (args.gpus, world))

    opt = Options(differences=args.differences, indels=args.indels,
                  nucleotides=args.nucleotides, ignore_genes=args.ignore_genes,
                  n_v_genes=synth.N_V, n_j_genes=synth.N_J, device=local_rank)

    # ---- synthetic workload (seeded; identical reference set on every rank) ----
    t0 = time.time()
    law = dict(law=args.law) if args.law != "uniform" else {}
    if args.repertoires != 16:
        law["n_repertoires"] = args.repertoires
    ref = synth.make_set(args.refs, 2, prefix="B", nucleotides=args.nucleotides,
                         pool_size=args.refs // 4, **law)
    strong = args.scaling == "strong"
    if args.self_cmp:
        full = ref
        args.queries = ref.n
    else:
        full = synth.make_set(args.queries, 1 + (0 if strong else 1000 * rank), prefix="A",
                              nucleotides=args.nucleotides, pool_size=args.refs // 4, **law)
    by_work = strong and use_dist and args.shard_by == "work"
    routed = by_work and args.layout == "routed"
    if strong and world > 1 and not by_work:
        lo, hi = shard_bounds(full.n, rank, world)
        qry = full.subset(slice(lo, hi))      # keeps the full set's repertoire numbering
    else:
        qry = full
    t_gen = time.time() - t0

    # ---- resident in HBM before the timed region ----
    h = HipOverlap(opt)
    for kv in args.tunable:
        k, v = kv.split("=")
        h.set_tunable(k, int(v))
    if by_work:
        h.set_tunable("work_shard_count", world)
        h.set_tunable("work_shard_index", rank)
    t0 = time.time()
    h.set_reference(ref, full.longest)
    t_index = time.time() - t0

    if routed:
        # this rank's share of the caller's set: a contiguous N-th (any split would do)
        lo, hi = shard_bounds(full.n, rank, world)
        share = full.subset(slice(lo, hi))

        def lay_out():
            return exchange_queries(h, share, lo, full.n, rank, world, device="cuda")
    else:
        def lay_out():
            h.set_queries(qry)
            return None

    # The first call of a context also allocates (device arenas, resident layout, positives
    # buffer); a further call on the same context -- the steady state of a service that works
    # through query sets -- reuses all of it.  Both are reported; at N > 1 the time of a
    # layout is that of the slowest rank (barrier in front, maximum behind).
    def timed_layout():
        if use_dist:
            dist.barrier()
        t = time.perf_counter()
        m = lay_out()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        if use_dist:
            x = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(x, op=dist.ReduceOp.MAX)
            dt = float(x.item())
        return dt, m
    skip_host = args.skip_host_layout and not routed
    t_layout_first, _ = (None, None) if skip_host else timed_layout()
    t_layout, moved = (None, None) if skip_host else timed_layout()
    layout_ms = None if skip_host else {
                 "total": t_layout * 1e3,
                 "library_call_total": h.get_tunable("layout_total_us") / 1e3,
                 "host_time_in_copy_calls": h.get_tunable("layout_upload_us") / 1e3,
                 "after_last_copy": h.get_tunable("layout_tail_us") / 1e3,
                 "first_call_incl_allocations": t_layout_first * 1e3,
                 "how": ("routed: upload + keys of this rank's N-th, all-to-all of the records, layout of "
                         "what was received; slowest rank" if routed else
                         "cmpr_set_queries (host buffers) on this rank's queries; slowest rank"),
                 "exchange": moved}
    # ---- this rank's query set as DEVICE arrays: what a timed step starts from ----
    per_set = args.step == "query-set"
    view = keep = None
    if not routed:
        view, keep = h.device_view(qry)
        h.set_tunable("layout_timing", 1)                  # (events around the layout's kernels: ~25 us per call)
        h.set_queries_device(view)                         # (warm: allocations kept)
        torch.cuda.synchronize()
    R1, R2 = h.shape
    layout = h.layout()
    h_items = h.get_tunable("items") if layout["variant"] == 2 else 0
    rec_tiles = bool(h.get_tunable("record_tiles"))    # no per-slot arrays: the probe kernel reads the records
    layout["record_tiles"] = int(rec_tiles)
    # every rank uses the same R1 x R2 (16 x 16 for the synthetic law).  Two matrices:
    # step i fills one while the all-reduce of step i - 1 still works on the other
    mats = [torch.zeros(R1 * R2, dtype=torch.int64, device="cuda") for _ in range(2)]
    # one explicit stream for the launches, one for the collective: nothing in a step is
    # ordered by the host but the layout's own two waits (its sizes, its end), and the
    # all-reduce of a step (latency-bound: 2 KiB) overlaps the work of the next
    stream = torch.cuda.Stream()
    comm = torch.cuda.Stream()
    filled = [torch.cuda.Event() for _ in range(2)]      # the kernels have written matrix b
    reduced = [torch.cuda.Event() for _ in range(2)]     # ... and its all-reduce is through
    for e in reduced:
        e.record(stream)
    nstep = [0]
    layout_kernels = []                                  # per step: HIP-event times of the layout's kernels

    def launch():
        b = nstep[0] & 1
        nstep[0] += 1
        if use_dist:
            stream.wait_event(reduced[b])      # (the all-reduce of two steps ago)
        h.overlap_matrix_device(mats[b].data_ptr(), stream.cuda_stream)
        if use_dist:
            # RCCL sum over xGMI, R1*R2 int64 (executed at world size 1 too: `torchrun
            # --nproc-per-node 1` exercises this very code)
            filled[b].record(stream)
            with torch.cuda.stream(comm):
                comm.wait_event(filled[b])
                dist.all_reduce(mats[b], op=dist.ReduceOp.SUM)
                reduced[b].record(comm)

    def step(record=False):
        if per_set:
            # the query set from its device arrays to "resident" (keys, regrouping, tiles, items; the
            # library's own stream, the call returns when the layout is complete), ...
            if routed:
                lay_out()
            else:
                h.set_queries_device(view)
            if record:
                layout_kernels.append(layout_kernel_times(h))
        launch()                               # ... then once over the reference index

    def timed(fn, k):
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()               # (every stream: the last all-reduce included)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        el = torch.tensor([dt], dtype=torch.float64, device="cuda")
        if use_dist:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item())

    with torch.cuda.stream(stream):
        for k in range(args.warmup):
            step()
            if k == 0:
                # (a positives buffer that the first launch overflowed grows when a later launch finds that
                # launch finished: let that be the second warm-up launch, not the first timed step)
                torch.cuda.synchronize()
        elapsed = timed(lambda: step(True), args.steps)
        matrix = mats[(nstep[0] - 1) & 1]
        st = h.stats()
        # HIP events recorded by the library on the kernels' stream, one set per step of
        # the timed region (ring of the last 64 launches: no synchronisation inside the loop)
        kernel_ms, probe_ms = h.kernel_times(args.steps)
        # the launches alone over the resident layout (the headline of rounds 1-5)
        for _ in range(2):
            launch()
        resident_steps = max(args.steps, 10)
        elapsed_resident = elapsed if not per_set else timed(launch, resident_steps)
        if per_set:
            matrix_resident = mats[(nstep[0] - 1) & 1]
            same_resident = bool(torch.equal(matrix_resident, matrix))
        else:
            same_resident = True

    total_queries = args.queries if strong else args.queries * world
    value = total_queries * args.steps / elapsed
    k_avg_ms = float(np.mean(kernel_ms))
    p_avg_ms = float(np.mean(probe_ms))

    result_matrix = matrix.cpu().numpy().astype(np.uint64).reshape(R1, R2)
    checksum = synth.checksum(result_matrix)

    # One more rate of the same workload (one GPU): the launch as a synchronous call that ends with
    # the matrix in host memory, and the layout call on its own.
    device_soa = None
    if world == 1 and not use_dist and view is not None:
        torch.cuda.synchronize()
        t = time.perf_counter()
        h.set_queries_device(view)
        torch.cuda.synchronize()
        t_dev = time.perf_counter() - t
        first = h.overlap_matrix()
        t = time.perf_counter()
        for _ in range(10):
            got = h.overlap_matrix()
        t_sync = (time.perf_counter() - t) / 10
        device_soa = {"set_queries_device_ms": t_dev * 1e3,
                      "value_from_device_soa": total_queries / (t_dev + elapsed_resident / resident_steps
                                                                if per_set else t_dev + elapsed / args.steps),
                      "step_ms_incl_d2h": t_sync * 1e3,
                      "value_incl_d2h": total_queries / t_sync,
                      "same_matrix": bool(np.array_equal(first, result_matrix) and
                                          np.array_equal(got, result_matrix))}
    del keep

    failed = []
    if rank == 0:
        wl = workload_name(args)
        baseline = None
        parity = None
        if world == 1 and args.cpu_sample >= 0:
            # ~10-30 s of CPU work: single-core rates of the reference loop
            per_q = {0: 3e7, 1: 2.3e5, 2: 2.6e3}[args.differences] / (2 if args.indels else 1)
            sample = args.cpu_sample or int(min(args.queries, max(1000, per_q * 20)))
            baseline, want, q, fmt, ref_used = cpu_baseline(ref, qry, opt, sample, args)
            # the same sample on the GPU must give the same cells: bit for bit against
            # the port, digit for digit (the reference prints %.10lg) against the binary
            if ref_used is not ref:
                h.set_reference(ref_used, full.longest)    # (a bounded reference sample: --cpu-refs)
            h.set_queries(q)
            got = h.overlap_matrix().astype(np.float64)
            if fmt:
                got = np.vectorize(lambda x: float(fmt % x))(got)
            parity = bool(np.array_equal(got, want))
            if not parity:
                print("PARITY FAILURE: HIP matrix differs from the CPU %s on the sample"
                      % baseline["kind"], file=sys.stderr)
                failed.append("parity_on_cpu_sample")
        rec = recorded_workload(args)
        parity_full = None
        if rec is not None:
            import _full_size
            why = _full_size.mismatch(rec, result_matrix)
            parity_full = why is None
            if why:
                print("PARITY FAILURE: the (reduced) matrix differs from the reference's "
                      "(tests/golden/full_size.json %s): %s" % (rec["name"], why), file=sys.stderr)
                failed.append("parity_vs_reference_full_size")
        if not same_resident:
            print("PARITY FAILURE: the launches over the resident layout give another matrix than the "
                  "timed steps", file=sys.stderr)
            failed.append("same_matrix_resident")
        if device_soa and not device_soa["same_matrix"]:
            print("PARITY FAILURE: the synchronous call gives another matrix than the timed steps",
                  file=sys.stderr)
            failed.append("same_matrix_sync")
        # every kernel that is a tenth of the step or more against its roofline; `roofline` = the one that
        # takes the longest (the dominant kernel of the timed step)
        step_ms = dict({k: float(np.mean([x[k] for x in layout_kernels])) for k in LAYOUT_PARTS}
                       if layout_kernels else {})
        if rec_tiles:
            step_ms.pop("tiles", None)             # (record tiles: fill_tiles_kernel does not run -- an empty interval)
        roofs = layout_rooflines(wl, int(st.queries), int(qry.offsets[-1]) if qry.n else 0,
                                 int(layout["query_slots"]), int(h_items), args.indels, step_ms) if per_set else {}
        roofs["probe"] = roofline(wl, st, p_avg_ms, k_avg_ms,
                                  "probe_pairs2_kernel" if layout.get("d2_pairs") else
                                  {0: "probe_kernel", 1: "probe_sliced_kernel", 2: "probe_rows_kernel"}[layout["variant"]])
        dominant = max(roofs, key=lambda k: roofs[k]["kernel_ms"])
        roof = dict(roofs[dominant], dominant_of=sorted(roofs))
        out = {
            "metric": "query sequences/sec for --matrix d=%d%s, %s-vs-%s %s" % (
                args.differences, " --indels" if args.indels else "",
                human(args.queries), human(args.refs),
                "nucleotide" if args.nucleotides else "CDR3aa"),
            "value": value,
            "step": ("one query set: cmpr_set_queries_device (layout from device arrays) + one launch over the "
                     "reference index" + (" + all-reduce" if use_dist else "")) if per_set and not routed else
                    ("one query set: route + all-to-all + layout of what arrived + one launch + all-reduce"
                     if per_set else "one launch over the resident layout"),
            "unit": "query sequences/s",
            "n_gpus": world,
            "ranks_seen": dist.get_world_size() if use_dist else 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": wl,
                       "queries_total": total_queries, "queries_this_gpu": int(st.queries),
                       "reference_sequences": args.refs,
                       "repertoires": [int(R1), int(R2)],
                       "sharding": ("%s scaling: %s over %d GPUs, reference index "
                                    "replicated, one RCCL all-reduce of the matrix per step"
                                    % (args.scaling,
                                       "the step's work items (filter slice + the tiles that probe it) "
                                       "dealt by slice, the queries routed to the GPUs that work on them"
                                       if routed else
                                       "the step's work items dealt by slice, all queries uploaded to every GPU"
                                       if by_work
                                       else "queries sharded", world)) if world > 1 else "single GPU",
                       "matrix_checksum": checksum,
                       "layout": layout,
                       "setup_seconds": {"generate": round(t_gen, 2), "index_build+upload": round(t_index, 3),
                                         "query_layout+upload": None if skip_host else round(t_layout, 4),
                                         "query_layout+upload_first_call": None if skip_host else round(t_layout_first, 4)},
                       # cmpr_set_queries, warm context: upload of the caller's arrays in ranges
                       # (the keys kernel of a range runs under the copy of the next), then what
                       # the upload cannot hide (sizes, scatter, tiles, items, chunk order)
                       "query_layout_ms": layout_ms},
            # from cmpr_set_view in host memory to the matrix: upload + device-side layout
            # of the queries (once per query set) + one step
            # (warm: a context that has laid out a set of this size before and keeps its
            #  allocations; cold: the first call of a context, allocations included -- the
            #  definition BENCH_r01 / r02 used for "value_incl_layout")
            "value_incl_layout": None if skip_host else total_queries / (t_layout + elapsed_resident / resident_steps),
            "value_incl_layout_cold": None if skip_host else
                                      total_queries / (t_layout_first + elapsed_resident / resident_steps),
            # the launches alone over one laid-out query set (what rounds 1-5 reported as `value`)
            "value_resident_step": total_queries * resident_steps / elapsed_resident,
            "resident_step_ms": elapsed_resident / resident_steps * 1e3,
            "resident_steps_same_matrix": same_resident,
            # HIP-event times inside the timed steps (means): the layout's kernels, then probe and resolve
            "step_kernels_ms": dict(
                {k: float(np.mean([x[k] for x in layout_kernels])) for k in LAYOUT_PARTS
                 if not (rec_tiles and k == "tiles")} if layout_kernels else {},
                probe=p_avg_ms, resolve=k_avg_ms - p_avg_ms),
            # from DEVICE arrays to the matrix (no PCIe), and the synchronous step that ends
            # with the matrix on the host
            "value_from_device_soa": device_soa and device_soa["value_from_device_soa"],
            "step_ms_incl_d2h": device_soa and device_soa["step_ms_incl_d2h"],
            "device_resident_inputs": device_soa,
            # the whole result of the timed steps against the matrix the reference binary
            # printed for this very workload (tests/golden/full_size.json), at every N
            "parity_vs_reference_full_size": parity_full,
            "roofline": roof,
            "roofline_kernels": roofs,
            "cpu_baseline": baseline,
            "parity_on_cpu_sample": parity,
        }
        print(json.dumps(out), flush=True)
    h.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        # a number whose matrix is not the reference's is no number: the run fails
        sys.exit(1)


def human(n):
    if n % 1_000_000 == 0:
        return "%dM" % (n // 1_000_000)
    if n % 1000 == 0:
        return "%dk" % (n // 1000)
    return str(n)


def workload_name(args):
    return "synthetic %s-vs-%s %s, d=%d%s%s" % (
        human(args.queries), human(args.refs), "nucleotide" if args.nucleotides else "CDR3aa",
        args.differences, " --indels" if args.indels else " substitutions only" if args.differences else "",
        " --ignore-genes" if args.ignore_genes else ", V/J matched") + (
        " (self comparison)" if getattr(args, "self_cmp", False) else "") + (
        " [law %s]" % args.law if getattr(args, "law", "uniform") != "uniform" else "") + (
        " [%d repertoires]" % args.repertoires if getattr(args, "repertoires", 16) != 16 else "")


if __name__ == "__main__":
    main()
