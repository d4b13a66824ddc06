#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02y}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rb in 5 2 1; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/tr_$rb -o p --output-format csv -- \
    python3 $R/bench.py --cpu-sample -1 --steps 10 --warmup 3 --tunable work_shard_count=8 --tunable work_shard_index=0 --tunable resolve_blocks_per_cu=$rb > $O/tr_$rb.json 2> $O/tr_$rb.err
  echo "resolve_blocks_per_cu=$rb"
  grep -h "probe_rows_kernel\|resolve_kernel\|fillBuffer" $O/tr_$rb/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
done
python3 - $O/tr_5 <<'PY'
import csv,glob,sys
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
last=rows[-14:]
t0=last[0][0]
for s,e,n in last:
    print("%9.1f us  +%8.1f us  %s" % ((s-t0)/1e3, (e-s)/1e3, n))
PY
