#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r04p}
O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grows or many_repertoires or existence or overflow or repeated or shortcut or pairs_list or scores" > $O/pytest_a.txt 2>&1; tail -5 $O/pytest_a.txt
line() {
python3 - $1 "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(sys.argv[2], "FAILED", e); sys.exit(0)
r=d["roofline"]; l=d["config"]["layout"]; c=d.get("cpu_baseline") or {}
print("%s: step %.4f ms (probe %.4f, rest %.4f) value %.3g incl_layout %.3g | positives=%s pairs=%s | parity full %s cpu %s | ref %.3g q/s" % (
    sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["value"], d["value_incl_layout"], r["bloom_positive_per_launch"], r["pairs_per_launch"], d["parity_vs_reference_full_size"], d["parity_on_cpu_sample"], c.get("value", 0)))
PY
}
P="--law cdr3 --refs 24200000 --queries 24200000 --self --repertoires 120"
timeout 1200 python3 bench.py $P --differences 0 > $O/pub_d0.json 2> $O/pub_d0.err; line $O/pub_d0.json "pub d=0"
timeout 1200 python3 bench.py $P --differences 1 > $O/pub_d1.json 2> $O/pub_d1.err; line $O/pub_d1.json "pub d=1"
timeout 1200 python3 bench.py $P --differences 1 --indels > $O/pub_d1i.json 2> $O/pub_d1i.err; line $O/pub_d1i.json "pub d=1 -i"
timeout 2400 python3 bench.py $P --differences 2 --steps 3 --warmup 2 --cpu-sample 2000 > $O/pub_d2.json 2> $O/pub_d2.err; line $O/pub_d2.json "pub d=2"
timeout 900 python3 bench.py --cpu-sample -1 > $O/cfg3.json 2> $O/cfg3.err; line $O/cfg3.json "cfg3"
timeout 1500 tools/e2e_cli.sh > $O/e2e.txt 2>&1; grep -E "wall|identical|GPU kernel" $O/e2e.txt
