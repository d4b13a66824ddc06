#!/bin/bash
# usage (GPU box): tools/r05_kstats.sh <tag> <bench args...>  -- rocprofv3 kernel stats of one bench run, top kernels printed
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o p --output-format csv -- \
    python3 $R/bench.py --cpu-sample -1 "$@" > $O/stats_bench.json 2> $O/stats.err
f=$(find $O/prof_stats -name '*kernel_stats.csv' | head -1)
cp $f $O/kernel_stats.csv
python3 - $O/kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%-70s calls %4s avg %10.1f us  max %10.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["MaxNs"])/1e3))
PY
find $O -name '*agent_info*' -delete; find $O/prof_stats -name '*.csv' -size +8M -delete
