cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for ctrs in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_ATOMIC_sum" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs -d $R/gpurun_out/self/pmc_$i -o p --output-format csv -- python3 $R/bench.py --cpu-sample -1 --self --steps 3 --warmup 1 > $R/gpurun_out/self/log_$i.txt 2>&1
done
python3 $R/tools/pmc_quick.py $R/gpurun_out/self
