#!/bin/bash
# usage (GPU box): tools/r04_full.sh <tag>  -- whole GPU suite, default bench, cfg5, emulations
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r04m}
O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
COMPAIRR_HIP_DEBUG=1 timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python3 - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("default bench: step %.4f ms probe %.4f rest %.4f parity_full %s parity_cpu %s incl_layout %.3g cold %.3g dev %s cpu %s" % (d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["parity_vs_reference_full_size"], d["parity_on_cpu_sample"], d["value_incl_layout"], d["value_incl_layout_cold"], d["device_resident_inputs"], d["cpu_baseline"]))
PY
tools/r04_cfg5.sh ${tag} "d2_pairs=1"
for n in 1 2 4 8; do
  timeout 600 python3 bench.py --steps 30 --warmup 5 --cpu-sample -1 --tunable work_shard_count=$n --tunable work_shard_index=0 > $O/w$n.json 2> $O/w$n.err
  python3 - $O/w$n.json $n <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("work shard 1/%s: step %.4f ms probe %.4f rest %.4f" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"]))
PY
done
timeout 900 python3 tools/emulate_routed.py > $O/routed.txt 2> $O/routed.err; cat $O/routed.txt
