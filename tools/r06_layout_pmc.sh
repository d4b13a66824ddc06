#!/bin/bash
# usage (GPU box): tools/r06_layout_pmc.sh <tag> [bench args] -- counters of the layout kernels (separate --pmc passes)
tag=${1:-r06pmc}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
            "SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_WAVES" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES" \
            "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" \
            "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" \
            "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCC_EA0_ATOMIC_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $ctrs -d $O/pmc_$i -o p --output-format csv -- \
      python3 $R/bench.py --cpu-sample -1 --steps 3 --warmup 1 "$@" > $O/pmc_$i.log 2>&1
done
python3 - $O <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
dur=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[n].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,d in acc.items():
    if any(w in k for w in ("keys_kernel","scatter_kernel","fill_tiles")):
        print(k[:70], "max %.1f us" % max(dur[k]))
        for c,v in sorted(d.items()): print("   %-32s max %.4g  mean %.4g  (%d)" % (c, max(v), sum(v)/len(v), len(v)))
PY
rm -rf $O/pmc_*/
