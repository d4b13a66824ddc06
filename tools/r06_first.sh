#!/bin/bash
# round 6, first GPU call: parity subset on the new layout, bench A/B (round 5's layout form by tunable), kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06a; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu \
   -k "synthetic_aa or synthetic_nt or device_resident or tiny_adversarial or broken_second or ragged or long_sequences or routed_queries_add_up or work_shards or existence or pairs_list" \
   > $O/pytest_subset.log 2>&1
tail -3 $O/pytest_subset.log
timeout 600 python3 bench.py > $O/bench_new.json 2> $O/bench_new.err
timeout 600 python3 bench.py --cpu-sample -1 --tunable item_wg=0 --tunable layout_recompute=0 > $O/bench_r5layout.json 2> $O/bench_r5layout.err
timeout 600 python3 bench.py --cpu-sample -1 --tunable item_wg=0 > $O/bench_noitemwg.json 2> $O/bench_noitemwg.err
timeout 600 python3 bench.py --cpu-sample -1 --indels > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 600 python3 bench.py --cpu-sample -1 --indels --tunable item_wg=0 --tunable layout_recompute=0 > $O/bench_cfg4_r5layout.json 2> $O/bench_cfg4_r5layout.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o p --output-format csv -- \
    python3 $R/bench.py --cpu-sample -1 --steps 5 --warmup 2 > $O/stats_bench.json 2> $O/stats.err
find $O -name '*agent_info*' -delete; find $O -name '*.csv' -size +8M -delete
for f in $O/bench_*.json; do echo $f; python3 - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("  value %.3e  ms/step %.3f  resident %.3f ms  kernels %s  parity %s %s" % (d["value"], d["ms_per_step"], d["resident_step_ms"],
          {k: round(v,3) for k,v in d["step_kernels_ms"].items()}, d["parity_vs_reference_full_size"], d["parity_on_cpu_sample"]))
    print("  set_queries_device_ms %.3f  host layout %.2f ms" % (d["device_resident_inputs"]["set_queries_device_ms"], d["config"]["query_layout_ms"]["total"]))
except Exception as e:
    print("  FAILED", e); print(open(sys.argv[1].replace(".json",".err")).read()[-2000:])
PY
done
f=$(find $O/prof_stats -name '*kernel_stats.csv' | head -1); head -25 $f
