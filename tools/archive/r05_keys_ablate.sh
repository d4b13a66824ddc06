#!/bin/bash
# usage (GPU box): tools/r05_keys_ablate.sh <tag>  -- keys_kernel / scatter_kernel with parts left out (-DCMPR_ABLATION library)
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r05k}
O=$R/gpurun_out/$tag; mkdir -p $O
export COMPAIRR_HIP_LIB=$R/compairr_amd/lib/libcompairr_hip_ablation.so
cd /tmp && export TMPDIR=/tmp
for dbg in 0 65536 131072 262144 524288 1048576 2097152 4128768; do
  rm -rf $O/prof
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- \
      python3 $R/bench.py --cpu-sample -1 --steps 2 --warmup 1 --tunable debug=$dbg > $O/b.json 2> $O/b.err
  f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
  python3 - $f $dbg <<'PY'
import csv,sys
out=[]
for r in csv.DictReader(open(sys.argv[1])):
    for k in ("keys_kernel","scatter_kernel","fill_tiles"):
        if k in r["Name"]:
            out.append("%s max %.1f us (avg %.1f x %s)" % (k, float(r["MaxNs"])/1e3, float(r["AverageNs"])/1e3, r["Calls"]))
print("debug=%s: " % sys.argv[2] + " | ".join(out))
PY
done
rm -rf $O/prof
