#!/bin/bash
# round-3 starting point on one MI355X: calibration by instruction class, the bench line,
# the shares of cmpr_set_queries, the emulated work shards and a step trace at 1/8 of the work
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r03a}
O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
tools/bin/calib > $O/calibration.json 2> $O/calib.err
timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err
tools/emulate_work_shards.sh ${tag}_w > $O/work_shards.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/layout_prof -o p --output-format csv -- \
    python3 $R/tools/layout_trace.py > $O/layout_trace.txt 2> $O/layout_trace.err
cd $R
tools/step_trace.sh ${tag}_tr8 --tunable work_shard_count=8 --tunable work_shard_index=0 > $O/step_trace8.txt 2>&1
tools/step_trace.sh ${tag}_tr1 > $O/step_trace1.txt 2>&1
find $O -name "*.csv" -size +20M -delete
ls -la $O
