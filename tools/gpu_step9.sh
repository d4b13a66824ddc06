#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-r02x}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "work_shards" 2>&1 | grep -v "^$" | tail -40
for q in 1000 100000; do
  python bench.py --steps 20 --warmup 5 --cpu-sample -1 --queries $q --scaling weak > $O/q$q.json 2> $O/q$q.err
  python - $O/q$q.json $q <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("queries %s: step %.3f ms probe %.3f resolve %.3f layout %s" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["config"]["layout"]))
PY
done
