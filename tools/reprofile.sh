#!/bin/bash
# usage (GPU box): tools/reprofile.sh <rNN>  -- THE entry point that regenerates what profiles/<rNN>/INDEX.md lists, in one
# call so that counters and bench lines come from the same sources and box: calibration, the rocprofv3 passes of the four
# BASELINE workloads (tools/profile_all.sh -> tools/profile_round.sh: kernel stats, then one --pmc pass per counter group,
# never mixed with a trace), their summary (tools/summarise_round.sh -> tools/pmc_summary.py), the four bench lines against
# it, the secondary workloads of DESIGN section 5, the emulated shards of section 6, the command-line program end to end.
R=${GRAFT_REPO_ROOT:-$(pwd)}
t=${1:-r06}
cd $R
mkdir -p gpurun_out/${t}_cal gpurun_out/${t}_extra
tools/bin/calib > gpurun_out/${t}_cal/calibration.json 2> gpurun_out/${t}_cal/calib.err; tail -c 200 gpurun_out/${t}_cal/calibration.json; echo
tools/profile_all.sh $t > /dev/null 2>&1
tools/summarise_round.sh $t > gpurun_out/${t}_cal/summarise.txt 2>&1
tail -6 gpurun_out/${t}_cal/summarise.txt
for w in cfg3 cfg2 cfg4 cfg5; do mkdir -p gpurun_out/${t}_$w; done
timeout 900 python3 bench.py > gpurun_out/${t}_cfg3/bench.json 2> gpurun_out/${t}_cfg3/bench.err
timeout 900 python3 bench.py --refs 1000000 --queries 1000000 --differences 0 > gpurun_out/${t}_cfg2/bench.json 2> gpurun_out/${t}_cfg2/bench.err
timeout 900 python3 bench.py --indels > gpurun_out/${t}_cfg4/bench.json 2> gpurun_out/${t}_cfg4/bench.err
timeout 1800 python3 bench.py --nucleotides --ignore-genes --differences 2 --refs 100000000 --queries 12500000 --steps 2 --warmup 1 --cpu-refs 10000000 --cpu-sample 20000 > gpurun_out/${t}_cfg5/bench.json 2> gpurun_out/${t}_cfg5/bench.err
line() {
python3 - $1 "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    r=d["roofline"]; p=d["roofline_kernels"]["probe"]; c=d.get("cpu_baseline") or {}
    print("%s: %.3f ms per query set = %s | value %.3g (resident step %.4f ms, %.3g) | dominant %s frac %.3f; probe frac %s (%s) stale %s | positives=%s pairs=%s K=%s slices=%s | parity full %s cpu %s | ref %.3g q/s" % (
        sys.argv[2], d["ms_per_step"], {k: round(v, 3) for k, v in d["step_kernels_ms"].items()}, d["value"], d["resident_step_ms"], d["value_resident_step"],
        r["kernel"], r["frac"] or 0, p.get("frac"), p.get("bound"), p.get("counters_stale"),
        p["bloom_positive_per_launch"], p["pairs_per_launch"], d["config"]["layout"].get("class_residues"), d["config"]["layout"].get("slices"),
        d["parity_vs_reference_full_size"], d["parity_on_cpu_sample"], c.get("value", 0)))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for w in cfg3 cfg2 cfg4 cfg5; do line gpurun_out/${t}_$w/bench.json $w; done
echo "elapsed $SECONDS s"
O=gpurun_out/${t}_extra
B="timeout 1500 python3 bench.py"
$B --self > $O/self.json 2> $O/self.err; line $O/self.json "self 10M d=1"
$B --law cdr3 > $O/cdr3_d1.json 2> $O/cdr3_d1.err; line $O/cdr3_d1.json "cdr3 d=1"
$B --law cdr3 --indels > $O/cdr3_d1i.json 2> $O/cdr3_d1i.err; line $O/cdr3_d1i.json "cdr3 d=1 -i"
$B --nucleotides --ignore-genes > $O/nt1.json 2> $O/nt1.err; line $O/nt1.json "nt d=1 -g"
$B --differences 0 > $O/d0_10m.json 2> $O/d0_10m.err; line $O/d0_10m.json "10M d=0"
P="--law cdr3 --refs 24200000 --queries 24200000 --self --repertoires 120"
$B $P --differences 0 > $O/pub_d0.json 2> $O/pub_d0.err; line $O/pub_d0.json "pub d=0"
$B $P --differences 1 > $O/pub_d1.json 2> $O/pub_d1.err; line $O/pub_d1.json "pub d=1"
$B $P --differences 1 --indels > $O/pub_d1i.json 2> $O/pub_d1i.err; line $O/pub_d1i.json "pub d=1 -i"
$B $P --differences 2 --steps 3 --warmup 2 --cpu-sample 2000 > $O/pub_d2.json 2> $O/pub_d2.err; line $O/pub_d2.json "pub d=2"
echo "elapsed $SECONDS s"
python3 tools/emulate_query_shards.py --out $O/cfg3_query_shards.json > $O/cfg3_query_shards.txt 2>&1; tail -4 $O/cfg3_query_shards.txt | cut -c1-300
python3 tools/emulate_query_shards.py --indels --out $O/cfg4_query_shards.json > $O/cfg4_query_shards.txt 2>&1; tail -4 $O/cfg4_query_shards.txt | cut -c1-300
tools/emulate_work_shards.sh $O > $O/work_shards.txt 2>&1; tail -12 $O/work_shards.txt
echo "elapsed $SECONDS s"
tools/e2e_cli.sh > $O/e2e.txt 2>&1; grep -E "wall|identical" $O/e2e.txt
echo "elapsed $SECONDS s"
