#!/bin/bash
# usage (GPU box), after tools/r04_profile.sh + tools/r04_summarise.sh (profiles/roofline_inputs.json then carries the counters
# of these very sources): the bench lines of the four BASELINE workloads once more, the published shape at d = 2 behind a
# second warm-up launch (the positives buffer grows after the first), the routed emulation, and 4 + 4 minutes of the fuzzers
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for w in cfg3 cfg2 cfg4 cfg5; do mkdir -p gpurun_out/r04_$w; done
timeout 900 python3 bench.py > gpurun_out/r04_cfg3/bench.json 2> gpurun_out/r04_cfg3/bench.err
timeout 900 python3 bench.py --refs 1000000 --queries 1000000 --differences 0 > gpurun_out/r04_cfg2/bench.json 2> gpurun_out/r04_cfg2/bench.err
timeout 900 python3 bench.py --indels > gpurun_out/r04_cfg4/bench.json 2> gpurun_out/r04_cfg4/bench.err
timeout 1500 python3 bench.py --nucleotides --ignore-genes --differences 2 --refs 100000000 --queries 12500000 --steps 2 --warmup 1 --cpu-kind port > gpurun_out/r04_cfg5/bench.json 2> gpurun_out/r04_cfg5/bench.err
for w in cfg3 cfg2 cfg4 cfg5; do python3 - gpurun_out/r04_$w/bench.json $w <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(sys.argv[2], "step %.4f probe %.4f rest %.4f frac %.3f bound %s stale %s util %s parity %s/%s" % (d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], r["frac"], r["bound"], r.get("counters_stale"), r.get("utilisation"), d.get("parity_vs_reference_full_size"), d.get("parity_on_cpu_sample")))
PY
done
O=gpurun_out/r04_extra; mkdir -p $O
P="--law cdr3 --refs 24200000 --queries 24200000 --self --repertoires 120"
timeout 2400 python3 bench.py $P --differences 2 --steps 3 --warmup 2 --cpu-sample 2000 > $O/pub_d2.json 2> $O/pub_d2.err
python3 - $O/pub_d2.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("pub d=2 step %.2f probe %.2f rest %.2f value %.3g parity cpu %s ref %.3g" % (d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["value"], d["parity_on_cpu_sample"], (d.get("cpu_baseline") or {}).get("value", 0)))
PY
line() { python3 - $1 "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]; l = d["config"]["layout"]; c = d.get("cpu_baseline") or {}
print("%s: step %.4f (probe %.4f rest %.4f) value %.3g K=%s slices=%s positives=%s pairs=%s parity %s/%s ref %.3g" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["value"], l["class_residues"], l["slices"], r["bloom_positive_per_launch"], r["pairs_per_launch"], d["parity_vs_reference_full_size"], d["parity_on_cpu_sample"], c.get("value", 0)))
PY
}
timeout 900 python3 bench.py --law cdr3 --indels > $O/cdr3_d1i.json 2> $O/cdr3_d1i.err; line $O/cdr3_d1i.json "cdr3 d=1 -i"
timeout 1200 python3 bench.py $P --differences 1 --indels > $O/pub_d1i.json 2> $O/pub_d1i.err; line $O/pub_d1i.json "pub d=1 -i"
timeout 900 python3 tools/emulate_routed.py > $O/routed.txt 2> $O/routed.err; cat $O/routed.txt
timeout 400 python3 tests/fuzz_gpu.py --seconds 150 --seed 40404 > $O/fuzz_lib.txt 2>&1; tail -2 $O/fuzz_lib.txt
timeout 400 python3 tests/fuzz_cli_gpu.py --seconds 150 --seed 50505 > $O/fuzz_cli.txt 2>&1; tail -2 $O/fuzz_cli.txt
