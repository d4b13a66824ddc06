#!/bin/bash
# usage (GPU box): tools/r04_first.sh <tag>  -- GPU suite, bench, emulated work shards with their layout times
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r04a}
O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
COMPAIRR_HIP_DEBUG=1 timeout 600 python3 bench.py --cpu-sample -1 > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
for n in 1 2 4 8; do
  COMPAIRR_HIP_DEBUG=1 timeout 600 python3 bench.py --steps 20 --warmup 5 --cpu-sample -1 --tunable work_shard_count=$n --tunable work_shard_index=0 > $O/w$n.json 2> $O/w$n.err
  python3 - $O/w$n.json $n <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("work shard 1/%s: step %.3f ms probe %.3f resolve %.3f chunks %s layout %s" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["config"]["layout"]["chunks"], d["config"]["query_layout_ms"]))
PY
done
