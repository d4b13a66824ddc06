#!/bin/bash
# usage (GPU box): tools/r05_direct_sweep.sh -- 1M x 1M and 10M x 10M at d = 0 (the filterless kernel) against the
# table's load (table_log2_delta) and the grid (blocks_per_cu)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05_direct_sweep; mkdir -p $O
line() {
python3 - $1 "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    r=d["roofline"]
    print("%s: step %.4f ms (probe %.4f, rest %.4f) hash_eq/pairs=%s parity %s" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], r["pairs_per_launch"], d["parity_vs_reference_full_size"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
B="timeout 600 python3 bench.py --cpu-sample -1 --differences 0"
for n in 1000000 10000000; do
for t in 0 1 2 3; do
$B --refs $n --queries $n --tunable table_log2_delta=$t > $O/n${n}_t$t.json 2> $O/n${n}_t$t.err; line $O/n${n}_t$t.json "n=$n table_log2_delta=$t"
done
for b in 1 2 8; do
$B --refs $n --queries $n --tunable blocks_per_cu=$b > $O/n${n}_b$b.json 2> $O/n${n}_b$b.err; line $O/n${n}_b$b.json "n=$n blocks_per_cu=$b"
done
done
echo "elapsed $SECONDS s"
