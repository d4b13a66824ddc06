#!/bin/bash
# usage (GPU box): tools/r04_deal.sh TAG -- parity subset, then the step with chunk_deal = 0 / 1 at 1, 1/2, 1/4, 1/8 of the work
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
T=${1:-r04_deal}
mkdir -p gpurun_out/$T
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tiny_adversarial or golden or repeated or long_seq or work_shards or overflow or routed" > gpurun_out/$T/pytest.txt 2>&1; tail -3 gpurun_out/$T/pytest.txt
line() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']; print('$1', 'step', round(j['ms_per_step'],4), 'probe', round(r['kernel_ms'],4), 'rest', round(r.get('resolve_kernel_ms') or 0,4), j['config']['matrix_checksum'][:8])"; }
for a in 0 1; do
  for n in 1 2 4 8; do
    timeout 300 python3 bench.py --cpu-sample -1 --steps 20 --warmup 5 --tunable chunk_deal=$a --tunable work_shard_count=$n --tunable work_shard_index=0 2>/dev/null | line "deal=$a 1/$n" | tee -a gpurun_out/$T/times.txt
  done
  for w in "--indels" "--law cdr3" "--law cdr3 --indels"; do
    timeout 300 python3 bench.py --cpu-sample -1 --steps 20 --warmup 5 $w --tunable chunk_deal=$a 2>/dev/null | line "deal=$a $w" | tee -a gpurun_out/$T/times.txt
  done
done
