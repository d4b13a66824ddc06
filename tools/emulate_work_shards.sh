#!/bin/bash
# per-rank cost of an N-GPU strong-scaling run by WORK shards (`bench.py --shard-by work --layout replicated`: every rank
# holds the whole query set in HBM, lays out and works on what is filed under its share of the filter slices), one rank
# emulated on one GPU: the query set per rank (layout + launch) and the launch alone over the resident layout.
# usage (GPU box): tools/emulate_work_shards.sh <gpurun_out subdir> [bench args]
cd "$(dirname "$0")/.." || exit 1
O=${1:-gpurun_out/work_shards}; shift; mkdir -p $O
for n in 1 2 4 8; do
  python3 bench.py --steps 20 --warmup 5 --cpu-sample -1 --tunable work_shard_count=$n --tunable work_shard_index=0 "$@" > $O/w$n.json 2> $O/w$n.err
  python3 - $O/w$n.json $n <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["step_kernels_ms"]
print("work shard 1/%s: query set %.3f ms %s | launch alone (resident) %.4f ms | chunks %s" % (
    sys.argv[2], d["ms_per_step"], {a: round(b, 3) for a, b in k.items()}, d["resident_step_ms"], d["config"]["layout"]["chunks"]))
PY
done
