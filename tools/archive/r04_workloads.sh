#!/bin/bash
# usage (GPU box): tools/r04_workloads.sh <tag>  -- the robustness workload (cdr3 law) and the shape of the
# reference's published benchmark (README.md:726-755: 24.2M sequences, 120 repertoires, self-vs-self)
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r04w}
O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
line() {
python3 - $1 "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(sys.argv[2], "FAILED", e); sys.exit(0)
r=d["roofline"]; l=d["config"]["layout"]; c=d.get("cpu_baseline") or {}
print("%s: step %.4f ms (probe %.4f, rest %.4f) value %.3g q/s incl_layout %.3g | K=%s slices=%s tiles=%s chunks=%s positives=%s pairs=%s | parity full %s cpu-sample %s | reference %.3g q/s on %s threads" % (
    sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["value"], d["value_incl_layout"], l["class_residues"], l["slices"], l["tiles"], l["chunks"],
    r["bloom_positive_per_launch"], r["pairs_per_launch"], d["parity_vs_reference_full_size"], d["parity_on_cpu_sample"], c.get("value", 0), c.get("cores")))
PY
}
timeout 900 python3 bench.py --law cdr3 > $O/cdr3_d1.json 2> $O/cdr3_d1.err; line $O/cdr3_d1.json "10M x 10M cdr3 law d=1"
timeout 900 python3 bench.py --law cdr3 --indels > $O/cdr3_d1i.json 2> $O/cdr3_d1i.err; line $O/cdr3_d1i.json "10M x 10M cdr3 law d=1 -i"
P="--law cdr3 --refs 24200000 --queries 24200000 --self --repertoires 120"
timeout 1200 python3 bench.py $P --differences 0 > $O/pub_d0.json 2> $O/pub_d0.err; line $O/pub_d0.json "24.2M self, 120 repertoires, d=0"
timeout 1200 python3 bench.py $P --differences 1 > $O/pub_d1.json 2> $O/pub_d1.err; line $O/pub_d1.json "24.2M self, 120 repertoires, d=1"
timeout 1200 python3 bench.py $P --differences 1 --indels > $O/pub_d1i.json 2> $O/pub_d1i.err; line $O/pub_d1i.json "24.2M self, 120 repertoires, d=1 -i"
timeout 2400 python3 bench.py $P --differences 2 --steps 2 --warmup 1 --cpu-sample 2000 > $O/pub_d2.json 2> $O/pub_d2.err; line $O/pub_d2.json "24.2M self, 120 repertoires, d=2"
tail -3 $O/pub_d2.err
