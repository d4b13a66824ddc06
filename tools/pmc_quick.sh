#!/bin/bash
# usage (GPU box): tools/pmc_quick.sh <tag> <bench args...>  -- three PMC passes over the issue-side counters
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
            "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $ctrs -d $O/pmc_$i -o p --output-format csv -- \
      python3 $R/bench.py --cpu-sample -1 --steps 3 --warmup 1 "$@" > $O/pmc_$i.log 2>&1
done
python3 - $O <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in acc.items():
    if "probe" in k or "resolve" in k:
        print(k)
        for c,v in sorted(d.items()): print("   %-24s %.4g" % (c, sum(v)/len(v)))
PY
rm -rf $O/pmc_*/
