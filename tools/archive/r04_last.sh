#!/bin/bash
# usage (GPU box): tools/r04_last.sh -- the published shape at d = 2 once more, and the filter geometry on the cdr3 law with -i
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r04_extra; mkdir -p $O
P="--law cdr3 --refs 24200000 --queries 24200000 --self --repertoires 120"
timeout 2400 python3 bench.py $P --differences 2 --steps 3 --warmup 2 --cpu-sample 2000 > $O/pub_d2.json 2> $O/pub_d2.err
python3 - $O/pub_d2.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("pub d=2 step %.2f probe %.2f rest %.2f value %.3g parity cpu %s ref %.3g" % (d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["value"], d["parity_on_cpu_sample"], (d.get("cpu_baseline") or {}).get("value", 0)))
PY
line() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']; l = j['config']['layout']; print('$1', 'step', round(j['ms_per_step'],4), 'probe', round(r['kernel_ms'],4), 'rest', round(r.get('resolve_kernel_ms') or 0,4), 'K', l['class_residues'], 'slices', l['slices'], 'chunks', l['chunks'], 'pos', r['bloom_positive_per_launch'], j['config']['matrix_checksum'][:8])"; }
for t in "" "--tunable bloom_bits_log2_delta=1" "--tunable bloom_bits_log2_delta=-1" "--tunable heavy_threshold=8"; do
  timeout 300 python3 bench.py --law cdr3 --indels --cpu-sample -1 --steps 10 --warmup 3 $t 2>/dev/null | line "cdr3 -i [$t]"
done
for t in "--tunable bloom_bits_log2_delta=1"; do
  timeout 300 python3 bench.py --law cdr3 --cpu-sample -1 --steps 10 --warmup 3 $t 2>/dev/null | line "cdr3 [$t]"
done
