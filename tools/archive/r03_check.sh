#!/bin/bash
# usage (GPU box): tools/r03_check.sh <tag>  -- GPU suite, bench, layout shares, step traces
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r03c}
O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
COMPAIRR_HIP_DEBUG=1 timeout 600 python3 bench.py --cpu-sample -1 > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
COMPAIRR_HIP_EVENT_FENCE=1 timeout 600 python3 bench.py --cpu-sample -1 > $O/bench_fence.json 2>/dev/null
timeout 600 python3 bench.py --cpu-sample -1 --tunable step_graph=0 > $O/bench_nograph.json 2>/dev/null
python3 - $O <<'PY'
import json,sys
for f in ("bench","bench_fence","bench_nograph"):
    try:
        d=json.loads(open(sys.argv[1]+"/%s.json"%f).read().strip().splitlines()[-1])
        print(f, "step %.4f probe %.4f resolve %.4f incl_layout %.3g setup %s" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["resolve_kernel_ms"], d["value_incl_layout"], d["config"]["setup_seconds"]))
    except Exception as e: print(f,"ERR",e)
PY
tools/emulate_work_shards.sh ${tag}_w > $O/work_shards.txt 2>&1; cat $O/work_shards.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/layout_prof -o p --output-format csv -- \
    python3 $R/tools/layout_trace.py > $O/layout_trace.txt 2> $O/layout_trace.err
cat $O/layout_trace.txt
cd $R
tools/step_trace.sh ${tag}_tr8 --tunable work_shard_count=8 --tunable work_shard_index=0 > $O/step_trace8.txt 2>&1; cat $O/step_trace8.txt
tools/bin/calib > $O/calibration.json 2> $O/calib.err
find $O $R/gpurun_out/${tag}_tr8 -name "*.csv" -size +20M -delete
