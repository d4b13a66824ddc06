#!/bin/bash
# usage (GPU box): tools/r04_keys_pmc.sh TAG -- counters of the layout kernels of cmpr_set_queries_device at 10M
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r04_keys}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_GDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_REQ_sum" \
           "TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $O/p$i -o p --output-format csv -- python3 $R/tools/layout_trace.py --device --reps 2 > $O/p$i.log 2>&1
done
python3 - $O <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[-2].split("::")[-1] if "(" in r["Kernel_Name"] else r["Kernel_Name"]
        k = r["Kernel_Name"][:60]
        if any(x in k for x in ("keys_kernel", "place_items", "scatter_kernel", "fill_tiles")):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k)
    for c, xs in sorted(v.items()):
        print("   %-40s %.4g (n=%d)" % (c, sum(xs) / len(xs), len(xs)))
PY
