#!/usr/bin/env python3
"""bench.py -- query sequences/sec through the --matrix hot path on MI355X.

One "step" = one pass of the hot path (variant enumeration -> Zobrist hash ->
Bloom probe -> hash-table walk -> verify -> matrix accumulate) over one batch of
synthetic queries that is already resident in HBM, plus -- at N > 1 -- the RCCL
all-reduce of the repertoire matrix.  Default workload = BASELINE.json
configs[2]: synthetic 10M-vs-10M CDR3aa, d = 1, substitutions only.

N > 1 (launched by torch.distributed.run, one rank per GPU): the reference set
(hash table + Bloom) is replicated, every rank holds its own 10M-query shard
(rank 0's shard is the N = 1 workload; weak scaling: the job's query set is
N x 10M), and the only collective is one all-reduce of the R1 x R2 matrix.

Prints ONE JSON line on rank 0 (see the contract in the task description), with
`roofline` (algorithmic bytes / HIP-event kernel time vs 8 TB/s HBM peak) and
`cpu_baseline` (the reference algorithm timed on this box's host cores on a
bounded sample of the same workload).
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


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--refs", type=int, default=10_000_000, help="set-2 sequences")
    p.add_argument("--queries", type=int, default=10_000_000, help="set-1 sequences per GPU")
    p.add_argument("--differences", "-d", type=int, default=1)
    p.add_argument("--indels", action="store_true")
    p.add_argument("--nucleotides", action="store_true")
    p.add_argument("--ignore-genes", action="store_true")
    p.add_argument("--cpu-sample", type=int, default=0,
                   help="queries in the CPU-baseline sample (0 = auto, -1 = skip)")
    p.add_argument("--self", dest="self_cmp", action="store_true",
                   help="one-file mode: the queries are the reference set itself")
    p.add_argument("--tunable", action="append", default=[], metavar="NAME=VALUE")
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
    Returns (dict, integer matrix, sample set)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle
    cores = os.cpu_count() or 1
    threads = max(1, min(cores, 256))           # the reference caps -t at 256
    q = queries.subset(slice(0, sample))
    exe = os.path.join(ROOT, "oracle", "_ref", "compairr")
    if os.path.exists(exe):
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
                         "kind": "reference",
                         "sample": "CompAIRR 1.13.0 binary (oracle/_ref), -t %d, first %d queries "
                                   "of the workload vs all %d reference sequences; its own "
                                   "'Analysing:' time %.2f s (whole run incl. TSV parse %.1f s)"
                                   % (threads, q.n, ref.n, sec, wall)},
                        m, q, "%.10g")
    m, st = _oracle.overlap(q, ref, opt, threads=threads)
    rate = q.n / st.seconds_analysis if st.seconds_analysis > 0 else 0.0
    return ({"value": rate, "unit": "query sequences/s", "cores": threads, "kind": "port",
             "sample": "first %d queries of the workload vs all %d reference sequences; "
                       "per-query loop only (reference 'Analysing:' region) %.2f s, "
                       "index build %.2f s excluded" % (q.n, ref.n, st.seconds_analysis,
                                                        st.seconds_index)},
            _oracle.integer_cells(m, opt).astype(np.float64), q, None)


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist
    from compairr_amd import HipOverlap, Options, synth
    from compairr_amd.dist import allreduce_matrix

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run (RANK set) the RCCL path is taken even at world
    # size 1, so that `--gpus 1` launched that way exercises the same code as N > 1
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1 and args.gpus == 1, \
        "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world)

    opt = Options(differences=args.differences, indels=args.indels,
                  nucleotides=args.nucleotides, ignore_genes=args.ignore_genes,
                  n_v_genes=synth.N_V, n_j_genes=synth.N_J, device=local_rank)

    # ---- synthetic workload (seeded; identical reference set on every rank) ----
    t0 = time.time()
    ref = synth.make_set(args.refs, 2, prefix="B", nucleotides=args.nucleotides,
                         pool_size=args.refs // 4)
    qry = ref if args.self_cmp else synth.make_set(
        args.queries, 1 + 1000 * rank, prefix="A", nucleotides=args.nucleotides,
        pool_size=args.refs // 4)
    if args.self_cmp:
        args.queries = ref.n
    t_gen = time.time() - t0

    # ---- resident in HBM before the timed region ----
    h = HipOverlap(opt)
    for kv in args.tunable:
        k, v = kv.split("=")
        h.set_tunable(k, int(v))
    t0 = time.time()
    h.set_reference(ref, qry.longest)
    t_index = time.time() - t0
    t0 = time.time()
    h.set_queries(qry)
    t_layout = time.time() - t0
    R1, R2 = h.shape
    layout = h.layout()
    # every rank must use the same R1 x R2 (16 x 16 for the synthetic law)
    matrix = torch.zeros(R1 * R2, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream()

    def step():
        h.overlap_matrix_device(matrix.data_ptr(), stream.cuda_stream)
        allreduce_matrix(matrix)               # RCCL sum over xGMI, R1*R2 int64 (no-op at N=1)

    kernel_ms, probe_ms = [], []
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    st = h.stats()
    # HIP events recorded by the library on the kernels' stream, one set per step of
    # the timed region (ring of the last 64 launches: no synchronisation inside the loop)
    kernel_ms, probe_ms = h.kernel_times(args.steps)

    total_queries = args.queries * world
    value = total_queries * args.steps / elapsed
    k_avg_ms = float(np.mean(kernel_ms))
    achieved = st.algorithmic_bytes / (k_avg_ms * 1e-3) / 1e9

    result_matrix = matrix.cpu().numpy().astype(np.uint64).reshape(R1, R2)

    if rank == 0:
        baseline = None
        parity = None
        if world == 1 and args.cpu_sample >= 0:
            # ~10-30 s of CPU work: single-core rates of the reference loop
            per_q = {0: 3e7, 1: 2.3e5, 2: 2.6e3}[args.differences] / (2 if args.indels else 1)
            sample = args.cpu_sample or int(min(args.queries, max(1000, per_q * 20)))
            baseline, want, q, fmt = cpu_baseline(ref, qry, opt, sample, args)
            # the same sample on the GPU must give the same cells: bit for bit against
            # the port, digit for digit (the reference prints %.10lg) against the binary
            h.set_queries(q)
            got = h.overlap_matrix().astype(np.float64)
            if fmt:
                got = np.vectorize(lambda x: float(fmt % x))(got)
            parity = bool(np.array_equal(got, want))
            if not parity:
                print("PARITY FAILURE: HIP matrix differs from the CPU %s on the sample"
                      % baseline["kind"], file=sys.stderr)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as fh:
                    t = json.load(fh)
                if t.get("workload") == workload_name(args):
                    traffic = t.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "query sequences/sec for --matrix d=%d%s, %s-vs-%s %s" % (
                args.differences, " --indels" if args.indels else "",
                human(args.queries), human(args.refs),
                "nucleotide" if args.nucleotides else "CDR3aa"),
            "value": value,
            "unit": "query sequences/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": workload_name(args),
                       "queries_per_gpu": args.queries, "reference_sequences": args.refs,
                       "repertoires": [int(R1), int(R2)],
                       "sharding": "queries sharded per GPU, reference index replicated, "
                                   "one RCCL all-reduce of the matrix" if world > 1 else "single GPU",
                       "matrix_checksum": synth.checksum(result_matrix),
                       "layout": layout,
                       "setup_seconds": {"generate": round(t_gen, 2), "index_build+upload": round(t_index, 3),
                                         "query_layout+upload": round(t_layout, 3)}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "probe_sliced_kernel" if layout.get("variant") == 1
                         else "probe_kernel",
                         # probe + resolve kernels (the figure `achieved` is priced on)
                         "kernel_ms": k_avg_ms,
                         "probe_kernel_ms": float(np.mean(probe_ms)),
                         "resolve_kernel_ms": k_avg_ms - float(np.mean(probe_ms)),
                         "algorithmic_bytes_per_launch": st.algorithmic_bytes,
                         "variants_per_launch": st.variants,
                         "bloom_positive_per_launch": st.bloom_positive,
                         "pairs_per_launch": st.matches,
                         "note": "achieved = SURVEY 8d algorithmic bytes (8 B per variant, as the "
                                 "reference reads its filter) / HIP-event time of the probe + "
                                 "resolve kernels; it can exceed the HBM peak because the filter "
                                 "words come from LDS; `traffic` is the measured HBM bytes"},
            "cpu_baseline": baseline,
            "parity_on_cpu_sample": parity,
        }
        print(json.dumps(out), flush=True)
    h.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


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
        " (self comparison)" if getattr(args, "self_cmp", False) else "")


if __name__ == "__main__":
    main()
