#!/bin/bash
# usage (GPU box): tools/r04_skew.sh <tag>  -- the cdr3-law workloads (skewed positives) under capacity settings
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r04s}
O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
for v in "" "--tunable pos_capacity=400000000"; do
  for w in "--law cdr3 --indels" "--law cdr3"; do
  timeout 900 python3 bench.py $w --cpu-sample -1 $v > $O/x.json 2> $O/x.err
  python3 - $O/x.json "$w $v" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("%s: step %.4f ms probe %.4f rest %.4f positives %s pairs %s parity %s" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], r["bloom_positive_per_launch"], r["pairs_per_launch"], d["parity_vs_reference_full_size"]))
PY
  done
done
