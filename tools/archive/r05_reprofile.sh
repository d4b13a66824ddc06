#!/bin/bash
# usage (GPU box): tools/r05_reprofile.sh -- calibration, the rocprofv3 passes of the four BASELINE workloads, their
# summary, then the four bench lines against it (one call: the counters and the lines come from the same sources and
# box), then the secondary workloads of DESIGN section 5 and the emulated shards of section 6
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r05_cal gpurun_out/r05_extra
tools/bin/calib > gpurun_out/r05_cal/calibration.json 2> gpurun_out/r05_cal/calib.err; tail -c 200 gpurun_out/r05_cal/calibration.json; echo
tools/profile_all.sh r05 > /dev/null 2>&1
tools/r05_summarise.sh > gpurun_out/r05_cal/summarise.txt 2>&1
for w in cfg3 cfg2 cfg4 cfg5; do mkdir -p gpurun_out/r05_$w; done
timeout 900 python3 bench.py > gpurun_out/r05_cfg3/bench.json 2> gpurun_out/r05_cfg3/bench.err
timeout 900 python3 bench.py --refs 1000000 --queries 1000000 --differences 0 > gpurun_out/r05_cfg2/bench.json 2> gpurun_out/r05_cfg2/bench.err
timeout 900 python3 bench.py --indels > gpurun_out/r05_cfg4/bench.json 2> gpurun_out/r05_cfg4/bench.err
timeout 1800 python3 bench.py --nucleotides --ignore-genes --differences 2 --refs 100000000 --queries 12500000 --steps 2 --warmup 1 --cpu-refs 10000000 --cpu-sample 20000 > gpurun_out/r05_cfg5/bench.json 2> gpurun_out/r05_cfg5/bench.err
for w in cfg3 cfg2 cfg4 cfg5; do python3 - gpurun_out/r05_$w/bench.json $w <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]); r = d["roofline"]
    print(sys.argv[2], "step %.4f probe %.4f rest %.4f frac %.3f guide %s bound %s stale %s parity %s/%s cpu %s %.3g" % (d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], r["frac"] or 0, r.get("frac_guide_peak"), r["bound"], r.get("counters_stale"), d.get("parity_vs_reference_full_size"), d.get("parity_on_cpu_sample"), (d.get("cpu_baseline") or {}).get("kind"), (d.get("cpu_baseline") or {}).get("value", 0)))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
echo "elapsed $SECONDS s"
O=gpurun_out/r05_extra
line() {
python3 - $1 "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    r=d["roofline"]; c=d.get("cpu_baseline") or {}; q=d["config"]["query_layout_ms"]
    print("%s: step %.4f ms (probe %.4f, rest %.4f) value %.3g | layout host %.2f dev %s | positives=%s pairs=%s K=%s slices=%s | parity full %s cpu %s | ref %.3g q/s" % (
        sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["value"], q["total"], (d.get("device_resident_inputs") or {}).get("set_queries_device_ms"),
        r["bloom_positive_per_launch"], r["pairs_per_launch"], d["config"]["layout"].get("class_residues"), d["config"]["layout"].get("slices"), d["parity_vs_reference_full_size"], d["parity_on_cpu_sample"], c.get("value", 0)))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
B="timeout 1500 python3 bench.py"
$B --self > $O/self.json 2> $O/self.err; line $O/self.json "self 10M d=1"
$B --law cdr3 > $O/cdr3_d1.json 2> $O/cdr3_d1.err; line $O/cdr3_d1.json "cdr3 d=1"
$B --law cdr3 --indels > $O/cdr3_d1i.json 2> $O/cdr3_d1i.err; line $O/cdr3_d1i.json "cdr3 d=1 -i"
$B --nucleotides --ignore-genes > $O/nt1.json 2> $O/nt1.err; line $O/nt1.json "nt d=1 -g"
P="--law cdr3 --refs 24200000 --queries 24200000 --self --repertoires 120"
$B $P --differences 0 > $O/pub_d0.json 2> $O/pub_d0.err; line $O/pub_d0.json "pub d=0"
$B $P --differences 1 > $O/pub_d1.json 2> $O/pub_d1.err; line $O/pub_d1.json "pub d=1"
$B $P --differences 1 --indels > $O/pub_d1i.json 2> $O/pub_d1i.err; line $O/pub_d1i.json "pub d=1 -i"
$B $P --differences 2 --steps 3 --warmup 2 --cpu-sample 2000 > $O/pub_d2.json 2> $O/pub_d2.err; line $O/pub_d2.json "pub d=2"
echo "elapsed $SECONDS s"
tools/r05_emulate.sh r05_extra cfg3 cfg4
echo "elapsed $SECONDS s"
tools/e2e_cli.sh > $O/e2e.txt 2>&1; grep -E "wall|identical" $O/e2e.txt
