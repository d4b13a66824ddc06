#!/bin/bash
# usage (on the GPU box): tools/pmc_run.sh <out subdir of gpurun_out> <bench.py args...>
# One rocprofv3 --pmc pass per counter group (PMC passes are never mixed with traces),
# then the per-kernel averages.
out=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS" \
            "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
            "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs -d $R/gpurun_out/$out/pmc_$i -o p --output-format csv -- \
      python3 $R/bench.py --cpu-sample -1 --steps 3 --warmup 1 "$@" > $R/gpurun_out/$out/log_$i.txt 2>&1
done
python3 $R/tools/pmc_quick.py $R/gpurun_out/$out | grep -E "probe|resolve"
