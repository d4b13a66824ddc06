#!/bin/bash
# usage (GPU box): tools/nt_quick.sh <tag>  -- the nucleotide workloads that fit a quick run
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/$1; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py --cpu-sample -1 "$@" > $O/$name.json 2>$O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print("%-14s step %.3f probe %.3f resolve %.3f pos %.3g chk %s" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], r["bloom_positive_per_launch"], d["config"]["matrix_checksum"][:8]))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
run nt_d1_10m   --nucleotides --ignore-genes --steps 10 --warmup 2
run nt_d2_2m_5m --nucleotides --ignore-genes --differences 2 --refs 5000000 --queries 2000000 --steps 5 --warmup 2
[ "$2" = "full" ] && run cfg5 --nucleotides --ignore-genes --differences 2 --refs 100000000 --queries 12500000 --steps 2 --warmup 1
true
