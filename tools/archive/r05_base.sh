#!/bin/bash
# usage (GPU box): tools/r05_base.sh <tag> [quick]  -- the workloads the round-4 verdict names, one line each
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r05a}
O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
line() {
python3 - $1 "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
except Exception as e:
    print(sys.argv[2], "FAILED", e); sys.exit(0)
r=d["roofline"]; c=d.get("cpu_baseline") or {}; q=d["config"]["query_layout_ms"]
print("%s: step %.4f ms (probe %.4f, rest %.4f) value %.3g | layout host %.2f ms dev %s ms | incl_layout %.3g from_dev %s | positives=%s pairs=%s | parity full %s cpu %s" % (
    sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["value"], q["total"],
    (d.get("device_resident_inputs") or {}).get("set_queries_device_ms"), d["value_incl_layout"], d.get("value_from_device_soa"),
    r["bloom_positive_per_launch"], r["pairs_per_launch"], d["parity_vs_reference_full_size"], d["parity_on_cpu_sample"]))
PY
}
B="timeout 1200 python3 bench.py --cpu-sample -1"
$B > $O/cfg3.json 2> $O/cfg3.err; line $O/cfg3.json "cfg3"
$B --indels > $O/cfg4.json 2> $O/cfg4.err; line $O/cfg4.json "cfg4"
$B --self > $O/self.json 2> $O/self.err; line $O/self.json "self 10M d=1"
$B --law cdr3 > $O/cdr3.json 2> $O/cdr3.err; line $O/cdr3.json "cdr3 d=1"
$B --law cdr3 --indels > $O/cdr3i.json 2> $O/cdr3i.err; line $O/cdr3i.json "cdr3 d=1 -i"
if [ "$2" != "quick" ]; then
P="--law cdr3 --refs 24200000 --queries 24200000 --self --repertoires 120"
$B $P --differences 0 > $O/pub_d0.json 2> $O/pub_d0.err; line $O/pub_d0.json "pub d=0"
$B $P --differences 1 > $O/pub_d1.json 2> $O/pub_d1.err; line $O/pub_d1.json "pub d=1"
$B $P --differences 1 --indels > $O/pub_d1i.json 2> $O/pub_d1i.err; line $O/pub_d1i.json "pub d=1 -i"
fi
