#!/bin/bash
# usage (GPU box): tools/r05_direct.sh [quick] -- d = 0 without a filter (variant 0, the default there) against the
# sliced filter (variant 1, the default of rounds 2-5): parity tests of everything d = 0 touches, then 1M x 1M,
# the 10M self-comparison and the 24.2M published shape both ways
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05_direct; mkdir -p $O
if [ "$1" != "quick" ]; then
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "not full_size" > $O/parity.txt 2>&1; tail -5 $O/parity.txt
fi
line() {
python3 - $1 "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    r=d["roofline"]; q=d["config"]["query_layout_ms"]
    print("%s: step %.4f ms (probe %.4f, rest %.4f) value %.3g | layout host %.2f dev %s | pairs=%s variant=%s slots=%s | parity full %s cpu %s" % (
        sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["value"], q["total"], (d.get("device_resident_inputs") or {}).get("set_queries_device_ms"),
        r["pairs_per_launch"], d["config"]["layout"].get("variant"), d["config"]["layout"].get("query_slots"), d["parity_vs_reference_full_size"], d["parity_on_cpu_sample"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
B="timeout 900 python3 bench.py --cpu-sample -1"
C="--refs 1000000 --queries 1000000 --differences 0"
$B $C > $O/cfg2.json 2> $O/cfg2.err; line $O/cfg2.json "cfg2 direct"
$B $C --tunable variant=1 > $O/cfg2_v1.json 2> $O/cfg2_v1.err; line $O/cfg2_v1.json "cfg2 variant 1"
$B $C --tunable direct_slices_log2=0 > $O/cfg2_p0.json 2> $O/cfg2_p0.err; line $O/cfg2_p0.json "cfg2 direct, one pseudo-slice"
$B $C --tunable direct_slices_log2=8 > $O/cfg2_p8.json 2> $O/cfg2_p8.err; line $O/cfg2_p8.json "cfg2 direct, 256 pseudo-slices"
$B --differences 0 --self > $O/self0.json 2> $O/self0.err; line $O/self0.json "10M self d=0 direct"
$B --differences 0 --self --tunable variant=1 > $O/self0_v1.json 2> $O/self0_v1.err; line $O/self0_v1.json "10M self d=0 variant 1"
$B --differences 0 --nucleotides > $O/nt0.json 2> $O/nt0.err; line $O/nt0.json "10M x 10M nt d=0 direct"
P="--law cdr3 --refs 24200000 --queries 24200000 --self --repertoires 120"
$B $P --differences 0 > $O/pub_d0.json 2> $O/pub_d0.err; line $O/pub_d0.json "pub d=0 direct"
$B $P --differences 0 --tunable variant=1 > $O/pub_d0_v1.json 2> $O/pub_d0_v1.err; line $O/pub_d0_v1.json "pub d=0 variant 1"
echo "elapsed $SECONDS s"
