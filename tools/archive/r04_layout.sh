#!/bin/bash
# usage (GPU box): tools/r04_layout.sh TAG -- per-kernel shares of cmpr_set_queries_device at 10M
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r04_layout}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/dev -o p --output-format csv -- python3 $R/tools/layout_trace.py --device --reps 6 > $O/dev.log 2>&1
cat $O/dev.log | tail -9
python3 - $O/dev <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:28]:
        print("%-70s calls %5s total_us %10.1f avg_us %9.1f" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
