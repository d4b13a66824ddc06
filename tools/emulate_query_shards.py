#!/usr/bin/env python3
"""One rank's share of `bench.py --gpus N` (the default split: N contiguous query shards, the
reference index on every GPU, nothing on the links but the matrix reduce), measured on ONE GPU:
for every N and every rank r the r-th N-th of the query set is laid out from its device arrays
and passed once over the index -- the timed step of bench.py -- `steps` times; the slowest rank
is the step of the job.  The matrices of the N shards are summed and held against the matrix the
reference binary printed for the whole workload (tests/golden/full_size.json), at every N.

No more than one GPU was available to the builder: the numbers are what ONE rank would need, the
all-reduce of the 2 KiB matrix (latency: a few tens of microseconds over xGMI, overlapped with the
next step by bench.py's second stream) is NOT in them.

usage (GPU box): python3 tools/emulate_query_shards.py [--n 1,2,4,8] [--indels] [--steps 10] [--out file.json]
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", default="1,2,4,8")
    ap.add_argument("--queries", type=int, default=10_000_000)
    ap.add_argument("--refs", type=int, default=10_000_000)
    ap.add_argument("--differences", "-d", type=int, default=1)
    ap.add_argument("--indels", action="store_true")
    ap.add_argument("--nucleotides", action="store_true")
    ap.add_argument("--ignore-genes", action="store_true")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--ranks", type=int, default=0, help="ranks measured per N (0: all of them)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    import _full_size
    from compairr_amd import HipOverlap, Options, synth
    from compairr_amd.dist import shard_bounds

    ref = synth.make_set(args.refs, 2, prefix="B", nucleotides=args.nucleotides, pool_size=args.refs // 4)
    full = synth.make_set(args.queries, 1, prefix="A", nucleotides=args.nucleotides, pool_size=args.refs // 4)
    opt = Options(differences=args.differences, indels=args.indels, nucleotides=args.nucleotides,
                  ignore_genes=args.ignore_genes, n_v_genes=synth.N_V, n_j_genes=synth.N_J, device=0)
    rec = _full_size.by_bench_args(args.refs, args.queries, args.differences, args.indels, args.nucleotides,
                                   args.ignore_genes, False)
    h = HipOverlap(opt)
    h.set_reference(ref, full.longest)
    h.set_tunable("layout_timing", 1)
    stream = torch.cuda.Stream()
    rows = []
    base = None
    for N in [int(x) for x in args.n.split(",")]:
        total = None
        worst = None
        ranks = range(N) if not args.ranks else range(min(N, args.ranks))
        for r in ranks:
            lo, hi = shard_bounds(full.n, r, N)
            share = full.subset(slice(lo, hi))
            view, keep = h.device_view(share)
            h.set_queries_device(view)
            R1, R2 = h.shape
            mat = torch.zeros(R1 * R2, dtype=torch.int64, device="cuda")
            with torch.cuda.stream(stream):
                for _ in range(3):                                   # warm: allocations, the positives buffer's size
                    h.set_queries_device(view)
                    h.overlap_matrix_device(mat.data_ptr(), stream.cuda_stream)
                    torch.cuda.synchronize()
                t0 = time.perf_counter()
                lay = 0.0
                for _ in range(args.steps):
                    t = time.perf_counter()
                    h.set_queries_device(view)
                    lay += time.perf_counter() - t
                    h.overlap_matrix_device(mat.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / args.steps
            kms, pms = h.kernel_times(args.steps)
            one = {"rank": r, "queries": share.n, "step_ms": dt * 1e3, "layout_ms": lay / args.steps * 1e3,
                   "launch_kernels_ms": float(np.mean(kms)), "probe_ms": float(np.mean(pms))}
            if worst is None or one["step_ms"] > worst["step_ms"]:
                worst = one
            m = mat.cpu().numpy().astype(np.uint64).reshape(R1, R2)
            total = m if total is None else total + m
            del keep
        parity = None
        if rec is not None and not args.ranks:
            parity = _full_size.mismatch(rec, total) is None
        if base is None:
            base = worst["step_ms"]
        row = {"n_gpus": N, "slowest_rank": worst, "queries_per_second": args.queries / (worst["step_ms"] * 1e-3),
               "speedup_vs_first": base / worst["step_ms"], "sum_of_shards_is_the_reference_matrix": parity}
        rows.append(row)
        print(json.dumps(row), flush=True)
    if args.out:
        with open(args.out, "w") as fh:
            json.dump({"what": __doc__.split("\n\n")[0], "argv": sys.argv[1:], "rows": rows}, fh, indent=1)
    h.close()


if __name__ == "__main__":
    main()
