#!/usr/bin/env python3
"""Where a wave's cycles go inside probe_rows_kernel: runs the bench workload on the
diagnostic library (make timing: -DCMPR_PHASE_TIMING, s_memtime stamps around the
phases) and prints the per-phase share.  usage: tools/phase_timing.py [tunable=value ...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["COMPAIRR_HIP_LIB"] = os.path.join(ROOT, "compairr_amd", "lib", "libcompairr_hip_timing.so")
sys.path.insert(0, ROOT)
from compairr_amd import HipOverlap, Options, synth

n = int(os.environ.get("PT_N", "10000000"))
opt = Options(differences=1, indels="--indels" in sys.argv, n_v_genes=synth.N_V, n_j_genes=synth.N_J)
ref = synth.make_set(n, 2, prefix="B", pool_size=n // 4)
qry = synth.make_set(n, 1, prefix="A", pool_size=n // 4)
h = HipOverlap(opt)
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("=")
        h.set_tunable(k, int(v))
h.set_reference(ref, qry.longest)
h.set_queries(qry)
for _ in range(3):
    h.overlap_matrix()
st = h.stats()
names = ["claim", "tile_data", "rows", "emit", "other", "loader_wait", "loader_dma", "tail"]
vals = [h.get_tunable("pt%d" % k) for k in range(8)]
tot = sum(vals) or 1
print("probe %.3f ms kernel %.3f ms tiles %d chunks %d" % (st.probe_ms, st.kernel_ms, h.get_tunable("tiles"), h.get_tunable("chunks")))
# (slot 5 is no phase: it counts the units whose data was requested one unit ahead)
ahead = vals[5]
vals[5] = 0
tot = sum(vals) or 1
for nme, v in zip(names, vals):
    if nme != "loader_wait":
        print("  %-12s %12d cycles  %5.1f %%" % (nme, v, 100.0 * v / tot))
print("  units whose data was requested ahead: %d" % ahead)
