#!/bin/bash
# usage (GPU box): tools/r04_reprofile.sh -- the rocprofv3 passes of the four BASELINE workloads, their summary, then
# the four bench lines against it (one call: the counters and the lines come from the same sources and box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r04_suite gpurun_out/r04_extra
tools/profile_all.sh r04 > /dev/null 2>&1
tools/r04_summarise.sh > /dev/null 2>&1
for w in cfg3 cfg2 cfg4 cfg5; do mkdir -p gpurun_out/r04_$w; done
timeout 900 python3 bench.py > gpurun_out/r04_cfg3/bench.json 2> gpurun_out/r04_cfg3/bench.err
timeout 900 python3 bench.py --refs 1000000 --queries 1000000 --differences 0 > gpurun_out/r04_cfg2/bench.json 2> gpurun_out/r04_cfg2/bench.err
timeout 900 python3 bench.py --indels > gpurun_out/r04_cfg4/bench.json 2> gpurun_out/r04_cfg4/bench.err
timeout 1500 python3 bench.py --nucleotides --ignore-genes --differences 2 --refs 100000000 --queries 12500000 --steps 2 --warmup 1 --cpu-kind port > gpurun_out/r04_cfg5/bench.json 2> gpurun_out/r04_cfg5/bench.err
for w in cfg3 cfg2 cfg4 cfg5; do python3 - gpurun_out/r04_$w/bench.json $w <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(sys.argv[2], "step %.4f probe %.4f rest %.4f frac %.3f bound %s stale %s parity %s/%s" % (d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], r["frac"], r["bound"], r.get("counters_stale"), d.get("parity_vs_reference_full_size"), d.get("parity_on_cpu_sample")))
PY
done
echo "elapsed $SECONDS s"
[ $SECONDS -gt 800 ] && exit 0
O=gpurun_out/r04_extra
P="--law cdr3 --refs 24200000 --queries 24200000 --self --repertoires 120"
timeout 1200 python3 bench.py $P --differences 2 --steps 3 --warmup 2 --cpu-sample 2000 > $O/pub_d2.json 2> $O/pub_d2.err
python3 - $O/pub_d2.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("pub d=2 step %.2f probe %.2f rest %.2f K %s" % (d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["config"]["layout"]["class_residues"]))
PY
echo "elapsed $SECONDS s"
[ $SECONDS -gt 900 ] && exit 0
timeout 150 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tiny_adversarial and 2-False" > gpurun_out/r04_suite/pytest_k.txt 2>&1; tail -2 gpurun_out/r04_suite/pytest_k.txt
