#!/bin/bash
# usage (GPU box): tools/step_trace.sh <tag> <bench.py args...>  -- start / duration of every kernel of the
# last two steps (rocprofv3 --kernel-trace): the gaps between the kernels of a step
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/tr -o p --output-format csv -- \
  python3 $R/bench.py --cpu-sample -1 --steps 10 --warmup 3 "$@" > $O/bench.json 2> $O/err.log
python3 - $O/tr <<'PY'
import csv,glob,sys
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
rows.sort()
names=[n for _,_,n in rows]
idx=[i for i,n in enumerate(names) if "probe_" in n and "true>" not in n]
last=rows[idx[-2]:idx[-1]+4] if len(idx)>=2 else rows[-12:]
t0=last[0][0]; prev=None
for s,e,n in last:
    print("%9.1f us  +%8.1f us  gap %6.1f  %s" % ((s-t0)/1e3, (e-s)/1e3, 0 if prev is None else (s-prev)/1e3, n))
    prev=e
PY
