#!/bin/bash
# usage (GPU box): tools/r04_ab.sh TAG [libs...] -- parity subset on the default library, then the d = 1 bench lines per library, twice
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
T=${1:-r04_ab}; shift
LIBS=${@:-libcompairr_hip.so libcompairr_hip_base.so}
mkdir -p gpurun_out/$T
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tiny_adversarial or golden or long_seq or repeated or overflow or routed or full_size or work_shards" > gpurun_out/$T/pytest.txt 2>&1; tail -3 gpurun_out/$T/pytest.txt
line() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']; print('$1', 'step', round(j['ms_per_step'],4), 'probe', round(r['kernel_ms'],4), 'rest', round(r.get('resolve_kernel_ms') or 0,4), 'K', j['config']['layout']['class_residues'], 'pos', r['bloom_positive_per_launch'], j['config']['matrix_checksum'][:8])"; }
for rep in 1 2; do
for lib in $LIBS; do
  for w in "" "--indels" "--law cdr3" "--law cdr3 --indels" "--tunable work_shard_count=8 --tunable work_shard_index=0"; do
    COMPAIRR_HIP_LIB=$R/compairr_amd/lib/$lib timeout 600 python3 bench.py --cpu-sample -1 --steps 20 --warmup 5 $w 2>/dev/null | line "$lib $w" | tee -a gpurun_out/$T/times.txt
  done
done
P="--law cdr3 --refs 24200000 --queries 24200000 --self --repertoires 120"
for lib in $LIBS; do
  for w in "--differences 1" "--differences 1 --indels"; do
    COMPAIRR_HIP_LIB=$R/compairr_amd/lib/$lib timeout 900 python3 bench.py $P $w --cpu-sample -1 --steps 5 --warmup 2 2>/dev/null | line "$lib pub $w" | tee -a gpurun_out/$T/times.txt
  done
done
done
