#!/bin/bash
# usage (GPU box): tools/r04_layout2.sh TAG -- layout parity tests, per-kernel layout trace, bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r04_layout2}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "invalid or device_resident or routed or golden or tiny_adversarial" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 300 python3 bench.py --cpu-sample -1 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print({k: j[k] for k in ('ms_per_step','value_incl_layout','value_incl_layout_cold','value_from_device_soa','device_resident_inputs')}); print(j['config']['query_layout_ms']); print(j['config']['setup_seconds'])"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/dev -o p --output-format csv -- python3 $R/tools/layout_trace.py --device --reps 6 > $O/dev.log 2>&1
grep set_ $O/dev.log | tail -4
python3 - $O/dev <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print("%-70s calls %5s total_us %10.1f avg_us %9.1f" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
