#!/usr/bin/env python3
"""cmpr_set_queries of the bench workload, a few times over: wall time per call (what
bench.py reports as query_layout+upload).  Run under
  rocprofv3 --kernel-trace --memory-copy-trace --stats -- python3 tools/layout_trace.py
for the per-kernel / per-copy shares."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from compairr_amd import HipOverlap, Options, synth  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--n", type=int, default=10_000_000)
p.add_argument("--reps", type=int, default=4)
p.add_argument("--indels", action="store_true")
p.add_argument("--device", action="store_true", help="cmpr_set_queries_device on a device-resident copy")
p.add_argument("--tunable", action="append", default=[])
a = p.parse_args()
ref = synth.make_set(a.n, 2, prefix="B", pool_size=a.n // 4)
qry = synth.make_set(a.n, 1, prefix="A", pool_size=a.n // 4)
opt = Options(differences=1, indels=a.indels, n_v_genes=synth.N_V, n_j_genes=synth.N_J, device=0)
with HipOverlap(opt) as h:
    for kv in a.tunable:
        k, v = kv.split("=")
        h.set_tunable(k, int(v))
    t0 = time.perf_counter()
    h.set_reference(ref, qry.longest)
    print("set_reference %.2f ms" % ((time.perf_counter() - t0) * 1e3))
    dv = keep = None
    if a.device:
        dv, keep = HipOverlap.device_view(qry)
    for r in range(a.reps):
        t0 = time.perf_counter()
        if a.device:
            h.set_queries_device(dv)
        else:
            h.set_queries(qry)
        print("set_queries %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    m = h.overlap_matrix()
    print("checksum", synth.checksum(m))
