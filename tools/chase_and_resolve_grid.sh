#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
tools/bin/chase
for rb in 5 3 2 1; do
  python bench.py --steps 20 --warmup 5 --cpu-sample -1 --tunable work_shard_count=8 --tunable work_shard_index=0 --tunable resolve_blocks_per_cu=$rb > gpurun_out/rb$rb.json 2>/dev/null
  python - gpurun_out/rb$rb.json $rb <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("1/8 shard, resolve_blocks_per_cu=%s: step %.3f probe %.3f resolve %.3f" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"]))
PY
done
for rb in 8 3; do
  python bench.py --steps 20 --warmup 5 --cpu-sample -1 --tunable resolve_blocks_per_cu=$rb > gpurun_out/rbf$rb.json 2>/dev/null
  python - gpurun_out/rbf$rb.json $rb <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("full, resolve_blocks_per_cu=%s: step %.3f probe %.3f resolve %.3f" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"]))
PY
done
