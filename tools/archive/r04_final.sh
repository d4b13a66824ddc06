#!/bin/bash
# usage (GPU box): tools/r04_final.sh <tag>  -- suite, bench lines of DESIGN section 5/6, emulations, whole program
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r04z}
O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
line() {
python3 - $1 "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(sys.argv[2], "FAILED", e); sys.exit(0)
r=d["roofline"]; l=d["config"]["layout"]; c=d.get("cpu_baseline") or {}
print("%s: step %.4f ms (probe %.4f, rest %.4f) value %.3g incl_layout %.3g cold %.3g dev %s d2h %s | K=%s slices=%s tiles=%s chunks=%s positives=%s pairs=%s | parity full %s cpu %s | ref %.3g q/s x%s (%s)" % (
    sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["value"], d["value_incl_layout"], d["value_incl_layout_cold"], d.get("value_from_device_soa"), d.get("step_ms_incl_d2h"),
    l["class_residues"], l["slices"], l["tiles"], l["chunks"], r["bloom_positive_per_launch"], r["pairs_per_launch"], d["parity_vs_reference_full_size"], d["parity_on_cpu_sample"], c.get("value", 0), c.get("cores"), c.get("kind")))
PY
}
COMPAIRR_HIP_DEBUG=1 timeout 900 python3 bench.py > $O/cfg3.json 2> $O/cfg3.err; line $O/cfg3.json "cfg3"; grep set_queries $O/cfg3.err | tail -3
timeout 900 python3 bench.py --refs 1000000 --queries 1000000 --differences 0 > $O/cfg2.json 2> $O/cfg2.err; line $O/cfg2.json "cfg2"
timeout 900 python3 bench.py --indels > $O/cfg4.json 2> $O/cfg4.err; line $O/cfg4.json "cfg4"
timeout 900 python3 bench.py --self > $O/self.json 2> $O/self.err; line $O/self.json "self10m"
timeout 900 python3 bench.py --nucleotides --ignore-genes --cpu-sample -1 > $O/nt1.json 2> $O/nt1.err; line $O/nt1.json "nt d=1 10Mx10M"
timeout 900 python3 bench.py --law cdr3 > $O/cdr3_d1.json 2> $O/cdr3_d1.err; line $O/cdr3_d1.json "cdr3 d=1"
timeout 900 python3 bench.py --law cdr3 --indels > $O/cdr3_d1i.json 2> $O/cdr3_d1i.err; line $O/cdr3_d1i.json "cdr3 d=1 -i"
P="--law cdr3 --refs 24200000 --queries 24200000 --self --repertoires 120"
timeout 1200 python3 bench.py $P --differences 0 > $O/pub_d0.json 2> $O/pub_d0.err; line $O/pub_d0.json "pub d=0"
timeout 1200 python3 bench.py $P --differences 1 > $O/pub_d1.json 2> $O/pub_d1.err; line $O/pub_d1.json "pub d=1"
timeout 1200 python3 bench.py $P --differences 1 --indels > $O/pub_d1i.json 2> $O/pub_d1i.err; line $O/pub_d1i.json "pub d=1 -i"
timeout 2400 python3 bench.py $P --differences 2 --steps 2 --warmup 1 --cpu-sample 2000 > $O/pub_d2.json 2> $O/pub_d2.err; line $O/pub_d2.json "pub d=2"
for n in 1 2 4 8; do
  timeout 600 python3 bench.py --steps 30 --warmup 5 --cpu-sample -1 --tunable work_shard_count=$n --tunable work_shard_index=0 > $O/w$n.json 2> $O/w$n.err
  python3 - $O/w$n.json $n <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("work shard 1/%s: step %.4f ms probe %.4f rest %.4f layout %.2f" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["config"]["query_layout_ms"]["total"]))
PY
done
COMPAIRR_HIP_DEBUG=1 timeout 900 python3 tools/emulate_routed.py > $O/routed.txt 2> $O/routed.err; cat $O/routed.txt; grep "route_queries\|set_queries" $O/routed.err | tail -6
timeout 1500 tools/e2e_cli.sh > $O/e2e.txt 2>&1; cat $O/e2e.txt
