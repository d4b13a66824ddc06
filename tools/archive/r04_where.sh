#!/bin/bash
# usage (GPU box): tools/r04_where.sh TAG -- phase timing and ablation counters of probe_rows_kernel on cfg3
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
T=${1:-r04_where}
mkdir -p gpurun_out/$T
timeout 300 python3 tools/phase_timing.py > gpurun_out/$T/phase.txt 2>&1; cat gpurun_out/$T/phase.txt
timeout 1500 tools/pmc_ablate.sh $T debug=0 debug=2 debug=8 debug=10 debug=64 debug=128 debug=256 > gpurun_out/$T/ablate.txt 2>&1; cat gpurun_out/$T/ablate.txt
for d in 0 2 8 10 64 128 256; do
  COMPAIRR_HIP_LIB=$R/compairr_amd/lib/libcompairr_hip_ablation.so timeout 300 python3 bench.py --cpu-sample -1 --steps 10 --warmup 2 --tunable debug=$d 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('debug=$d', j['ms_per_step'], j.get('kernel_ms'))" | tee -a gpurun_out/$T/times.txt
done
