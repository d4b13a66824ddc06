#!/bin/bash
# usage (GPU box): tools/pmc_extra.sh <tag> <bench.py args...>  -- issue-side counters of the probe kernel
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
i=0
for ctrs in "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
            "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" \
            "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
            "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs -d $O/pmcx_$i -o p --output-format csv -- \
      python3 $R/bench.py --cpu-sample -1 --steps 3 --warmup 1 "$@" > $O/pmcx_$i.log 2>&1
done
python3 - $O <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/pmcx_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "probe_rows_kernel" in k and "true>" not in k.split("(")[0][-8:] or "resolve_kernel" in k:
            acc[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k)
    for c,vals in sorted(v.items()):
        print("   %-28s %.4g (n=%d)" % (c, sum(vals)/len(vals), len(vals)))
PY
