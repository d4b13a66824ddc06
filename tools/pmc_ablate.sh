#!/bin/bash
# usage (GPU box): tools/pmc_ablate.sh <tag> "debug=64" ...  -- VALU / scalar instruction counts of the
# probe kernel with parts of it switched off (the -DCMPR_ABLATION library)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
export COMPAIRR_HIP_LIB=$R/compairr_amd/lib/libcompairr_hip_ablation.so
cd /tmp && export TMPDIR=/tmp
i=0
for t in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU -d $O/ab_$i -o p --output-format csv -- \
      python3 $R/bench.py --cpu-sample -1 --steps 3 --warmup 1 $EXTRA --tunable $t > $O/ab_$i.log 2>&1
  python3 - $O/ab_$i "$t" <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(list); dur=[]
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "probe_rows_kernel" in k and "false>" in k:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-12s" % sys.argv[2], "  ".join("%s %.4g" % (c, sum(v)/len(v)) for c,v in sorted(acc.items())))
PY
done
