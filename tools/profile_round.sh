#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh <tag> <bench.py args...>
# 1. rocprofv3 --kernel-trace --stats of the bench command (per-kernel durations);
# 2. one rocprofv3 --pmc pass per counter group (FETCH_SIZE and WRITE_SIZE each alone,
#    as MI355X_MICROARCH.md prescribes; PMC passes are never mixed with traces).
# Everything lands in gpurun_out/<tag>/; tools/pmc_summary.py turns it into profiles/.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
python3 $R/tools/csrc_hash.py > $O/csrc_sha256.txt
cd /tmp && export TMPDIR=/tmp
if [ "$STATS_ONLY" != 1 ]; then timeout 900 python3 $R/bench.py "$@" > $O/bench.json 2> $O/bench.err; fi
if [ "$STATS_ONLY" = 1 ]; then rm -rf $O/prof_stats; fi
# (--skip-host-layout: every launch of the layout's kernels in the trace is a full-size one of a step -- the trace's
#  averages are then the bench line's per-kernel times)
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o p --output-format csv -- \
    python3 $R/bench.py --cpu-sample -1 --skip-host-layout "$@" > $O/stats_bench.json 2> $O/stats.err
if [ "$STATS_ONLY" = 1 ]; then find $O -name '*agent_info*' -delete; find $O -name '*.csv' -size +8M -delete; exit 0; fi
i=0
# PMC_GROUPS=essential: the six groups the roofline needs (long workloads)
CTR_GROUPS=("FETCH_SIZE" "WRITE_SIZE"
        "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
        "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY"
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES")
if [ "$PMC_GROUPS" != "essential" ]; then
  CTR_GROUPS+=("TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"
           "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU")
fi
for ctrs in "${CTR_GROUPS[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $ctrs -d $O/pmc_$i -o p --output-format csv -- \
      python3 $R/bench.py --cpu-sample -1 --steps 3 --warmup 1 "$@" > $O/pmc_$i.log 2>&1
done
find $O -name '*agent_info*' -delete; find $O -name '*.csv' -size +8M -delete
ls $O
