#!/bin/bash
# per-rank step of an N-GPU strong-scaling run (--shard-by work), emulated on one GPU
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-r02w}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "work_shards or debug_switches" 2>&1 | tail -3
for n in 1 2 4 8; do
  python bench.py --steps 20 --warmup 5 --cpu-sample -1 --tunable work_shard_count=$n --tunable work_shard_index=0 > $O/w$n.json 2> $O/w$n.err
  python - $O/w$n.json $n <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("work shard 1/%s: step %.3f ms probe %.3f resolve %.3f chunks %s" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["config"]["layout"]["chunks"]))
PY
done
