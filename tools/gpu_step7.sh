#!/bin/bash
# per-rank step of an 8-GPU strong-scaling run, emulated on one GPU: 1/8 of the queries
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-r02s}; mkdir -p $O
for q in 10000000 5000000 2500000 1250000; do
  python bench.py --steps 20 --warmup 5 --queries $q --scaling weak > $O/q$q.json 2> $O/q$q.err
  python - $O/q$q.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("queries %9d step %.3f ms probe %.3f resolve %.3f" % (d["config"]["queries_per_gpu"], d["ms_per_step"], r.get("probe_kernel_ms",0), r.get("resolve_kernel_ms",0)))
PY
done
