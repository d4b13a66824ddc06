#!/usr/bin/env python3
"""Per-rank cost of the routed multi-GPU layout, emulated on ONE GPU: N contexts on this
device stand for the N ranks of `bench.py --gpus N` (each with the reference index, each given
a contiguous N-th of the 10M queries); the records change hands with device-to-device copies
instead of the all-to-all over xGMI, whose time is ESTIMATED from the bytes that would cross
the links.  Prints one line per N: the slowest rank's route + pack, receive + layout, step.

usage (GPU box): python tools/emulate_routed.py [--n 1,2,4,8] [--indels] [--queries 10000000]
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", default="1,2,4,8")
    ap.add_argument("--queries", type=int, default=10_000_000)
    ap.add_argument("--refs", type=int, default=10_000_000)
    ap.add_argument("--indels", action="store_true")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--link-gbs", type=float, default=48.0,
                    help="what one xGMI link gives a point-to-point copy, GB/s (estimate of the exchange)")
    args = ap.parse_args()
    import torch
    from compairr_amd import HipOverlap, Options, synth
    from compairr_amd.dist import shard_bounds

    ref = synth.make_set(args.refs, 2, prefix="B", pool_size=args.refs // 4)
    full = synth.make_set(args.queries, 1, prefix="A", pool_size=args.refs // 4)
    opt = Options(differences=1, indels=args.indels, n_v_genes=synth.N_V, n_j_genes=synth.N_J, device=0)
    want = None
    for N in [int(x) for x in args.n.split(",")]:
        hs = []
        for r in range(N):
            h = HipOverlap(opt)
            h.set_tunable("work_shard_count", N)
            h.set_tunable("work_shard_index", r)
            h.set_reference(ref, full.longest)
            hs.append(h)
        shares = []
        for r in range(N):
            lo, hi = shard_bounds(full.n, r, N)
            shares.append((lo, full.subset(slice(lo, hi))))
        best = None
        for rep in range(4):                      # warm contexts; the best round counts (the caching allocator's
                                                  # first sizes and the arenas' growth fall into the early ones)
            t_route, t_recv, sends = [], [], []
            totals, rb = np.zeros(full.n_repertoires), 0
            for h, (first, share) in zip(hs, shares):
                torch.cuda.synchronize()
                t = time.perf_counter()
                counts, rb, tot = h.route_queries(share, first, N)
                buf = torch.empty(max(int(counts.sum()), 1) * rb, dtype=torch.uint8, device="cuda")
                h.route_pack(buf.data_ptr(), int(counts.sum()) * rb)
                torch.cuda.synchronize()
                t_route.append(time.perf_counter() - t)
                totals += tot
                sends.append((counts, buf))
            cross = []
            for d, h in enumerate(hs):
                runs = []
                for counts, buf in sends:
                    start = int(counts[:d].sum()) * rb
                    runs.append(buf[start:start + int(counts[d]) * rb])
                recv = torch.cat(runs)
                # bytes that arrive over the links (everything but the rank's own run), 7 links in parallel
                cross.append(sum(int(c[d]) for k, (c, _) in enumerate(sends) if k != d) * rb)
                torch.cuda.synchronize()
                t = time.perf_counter()
                h.set_queries_routed(recv.data_ptr(), recv.numel() // rb, full.n_repertoires, full.n, totals)
                torch.cuda.synchronize()
                t_recv.append(time.perf_counter() - t)
            now = (max(t_route), max(t_recv), max(cross), sum(int(c.sum()) for c, _ in sends))
            if best is None or now[0] + now[1] < best[0] + best[1]:
                best = now
        step_ms, probe_ms, mats = [], [], []
        for h in hs:
            mats.append(h.overlap_matrix())
            t = torch.zeros(mats[-1].size, dtype=torch.int64, device="cuda")
            s = torch.cuda.Stream()
            for _ in range(5):
                h.overlap_matrix_device(t.data_ptr(), s.cuda_stream)
            s.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                h.overlap_matrix_device(t.data_ptr(), s.cuda_stream)
            s.synchronize()
            step_ms.append((time.perf_counter() - t0) / args.steps * 1e3)
            k, p = h.kernel_times(args.steps)
            probe_ms.append(float(np.mean(p)))
        total = sum(mats)
        if want is None:
            want = total
        ok = bool(np.array_equal(total, want))
        links = max(1, min(7, N - 1))
        xch_ms = best[2] / links / (args.link_gbs * 1e9) * 1e3 if N > 1 else 0.0
        layout_ms = (best[0] + best[1]) * 1e3 + xch_ms
        out = {"N": N, "route_pack_ms": best[0] * 1e3, "receive_layout_ms": best[1] * 1e3,
               "exchange_ms_estimated": xch_ms, "layout_ms": layout_ms,
               "records_total": best[3], "step_ms": max(step_ms), "probe_ms": max(probe_ms),
               "value_resident": full.n / (max(step_ms) * 1e-3),
               "value_incl_layout": full.n / (layout_ms * 1e-3 + max(step_ms) * 1e-3),
               "matrices_add_up": ok, "checksum": synth.checksum(total)}
        print(json.dumps(out), flush=True)
        for h in hs:
            h.close()


if __name__ == "__main__":
    main()
