#!/bin/bash
# usage (GPU box): tools/r04_cfg5.sh <tag> [tunable sets...]  -- cfg5 per-GPU shape under the given tunables
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
for v in "$@"; do
  args=""; for kv in $v; do args="$args --tunable $kv"; done
  name=$(echo $v | tr ' =' '__')
  timeout 900 python3 bench.py --nucleotides --ignore-genes --differences 2 --refs 100000000 --queries 12500000 --steps 2 --warmup 1 --cpu-sample -1 $args > $O/cfg5_$name.json 2> $O/cfg5_$name.err
  tail -1 $O/cfg5_$name.err
  python3 - $O/cfg5_$name.json "$v" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]; l=d["config"]["layout"]
print("cfg5 [%s] step %.3f ms probe %.3f rest %.3f checksum %s K=%s slices=%s tiles=%s chunks=%s reads %s pos %s layout_ms %.1f" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["config"]["matrix_checksum"][:8], l["class_residues"], l["slices"], l["tiles"], l["chunks"], r["filter_reads_per_launch"], r["bloom_positive_per_launch"], d["config"]["query_layout_ms"]["total"]))
PY
done
