#!/bin/bash
# round 6: quick A/B of the bench line (and a parity subset) -- usage: tools/r06_ab.sh <tag> [pytest -k expression]
tag=$1; K=${2:-"synthetic_aa or device_resident or tiny_adversarial or ragged or long_sequences"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > $O/pytest_subset.log 2>&1
tail -3 $O/pytest_subset.log
timeout 600 python3 bench.py --cpu-sample -1 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout 600 python3 bench.py --cpu-sample -1 --indels > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 600 python3 bench.py --cpu-sample -1 --refs 1000000 --queries 1000000 -d 0 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
timeout 600 python3 bench.py --cpu-sample -1 --law cdr3 > $O/bench_cdr3.json 2> $O/bench_cdr3.err
for f in $O/bench_*.json; do echo $f; python3 - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("  value %.3e  ms/step %.3f  resident %.3f ms  kernels %s  parity %s" % (d["value"], d["ms_per_step"], d["resident_step_ms"],
          {k: round(v,3) for k,v in d["step_kernels_ms"].items()}, d["parity_vs_reference_full_size"]))
    print("  set_queries_device_ms %.3f  host layout %.2f ms" % (d["device_resident_inputs"]["set_queries_device_ms"], d["config"]["query_layout_ms"]["total"]))
except Exception as e:
    print("  FAILED", e); print(open(sys.argv[1].replace(".json",".err")).read()[-2000:])
PY
done
