#!/bin/bash
# usage: tools/bench_sweep.sh OUTDIR "bench args A" "bench args B" ...   (each: a full bench.py argument string)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/$1; shift; mkdir -p $O
i=0
for cfg in "$@"; do
  i=$((i+1))
  timeout 900 python bench.py --cpu-sample -1 $cfg > $O/run$i.json 2>$O/run$i.err
  python - "$cfg" $O/run$i.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read())
    r=d["roofline"]; l=d["config"]["layout"]
    print("%-78s step %.3f ms probe %.3f resolve %.3f | v=%s K=%s slices=%s tiles=%s chunks=%s small=%s pos=%s pairs=%s setup=%s/%s cs=%s" % (
        sys.argv[1], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], l.get("variant"), l.get("class_residues"), l.get("slices"),
        l.get("tiles"), l.get("chunks"), l.get("small_tiles"), r["bloom_positive_per_launch"], r["pairs_per_launch"], d["config"]["setup_seconds"]["index_build+upload"], d["config"]["setup_seconds"]["query_layout+upload"], d["config"]["matrix_checksum"][:8]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
