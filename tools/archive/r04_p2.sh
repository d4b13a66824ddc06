#!/bin/bash
# usage (GPU box): tools/r04_p2.sh <tag>  -- pair-row d = 2 kernel: parity, then timing
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r04f}
O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "(tiny_adversarial_sets and True-2-False) or synthetic_nt or (routed and nt_d2) or (work_shards and nt_d2) or sep_reduce or (pairs_list and 2-False) or (existence and 2-False) or long_sequences or items_next" > $O/pytest_a.txt 2>&1; tail -25 $O/pytest_a.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size_matches_reference and cfg5_sub" > $O/pytest_c.txt 2>&1; tail -5 $O/pytest_c.txt
if [ "$QUICK" != "" ]; then
for v in "d2_pairs=1" "variant=1"; do
  timeout 900 python3 bench.py --nucleotides --ignore-genes --differences 2 --refs 20000000 --queries 2500000 --steps 3 --warmup 1 --cpu-sample -1 --tunable $v > $O/mid_$v.json 2> $O/mid_$v.err
  tail -2 $O/mid_$v.err
  python3 - $O/mid_$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("2.5M x 20M step %.3f ms probe %.3f rest %.3f checksum %s layout %s reads %s pos %s" % (d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["config"]["matrix_checksum"], d["config"]["layout"], r["filter_reads_per_launch"], r["bloom_positive_per_launch"]))
PY
done
else
for v in "d2_pairs=1" "variant=1"; do
  timeout 900 python3 bench.py --nucleotides --ignore-genes --differences 2 --refs 100000000 --queries 12500000 --steps 2 --warmup 1 --cpu-sample -1 --tunable $v > $O/cfg5_$v.json 2> $O/cfg5_$v.err
  tail -2 $O/cfg5_$v.err
  python3 - $O/cfg5_$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("cfg5 step %.3f ms probe %.3f rest %.3f checksum %s layout %s reads %s pos %s" % (d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["config"]["matrix_checksum"], d["config"]["layout"], r["filter_reads_per_launch"], r["bloom_positive_per_launch"]))
PY
done
fi
if [ "$PMC" != "" ]; then
cd /tmp && export TMPDIR=/tmp
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $ctrs -d $O/pmc_$i -o p --output-format csv -- \
      python3 $R/bench.py --nucleotides --ignore-genes --differences 2 --refs 20000000 --queries 2500000 --cpu-sample -1 --steps 2 --warmup 1 --tunable d2_pairs=1 > $O/pmc_$i.log 2>&1
done
python3 - $O <<'PY'
import csv,glob,sys,collections
tot=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pairs2" in r["Kernel_Name"] or "resolve" in r["Kernel_Name"]:
            tot[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in tot.items():
    print(k, {c: sum(x)/len(x) for c,x in v.items()})
PY
find $O -name '*agent_info*' -delete; find $O -name '*.csv' -size +8M -delete
fi
