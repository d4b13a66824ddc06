#!/bin/bash
# usage: tools/tunable_sweep.sh OUTDIR "tunables A" "tunables B" ...   (each: space-separated NAME=VALUE and/or single-word bench flags such as --indels; flags that take a value need tools/bench_sweep.sh)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/$1; shift; mkdir -p $O
i=0
for cfg in "$@"; do
  i=$((i+1))
  args=""
  for kv in $cfg; do
    case "$kv" in --*) args="$args $kv";; *) args="$args --tunable $kv";; esac
  done
  timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample -1 $args > $O/run$i.json 2>$O/run$i.err
  python - "$cfg" $O/run$i.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read())
    r=d["roofline"]; l=d["config"]["layout"]
    print("%-60s step %.3f ms probe %.3f resolve %.3f | K=%s slices=%s tiles=%s chunks=%s small=%s pos=%s setup=%s cs=%s" % (
        sys.argv[1], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], l.get("class_residues"), l.get("slices"),
        l.get("tiles"), l.get("chunks"), l.get("small_tiles"), r["bloom_positive_per_launch"], d["config"]["setup_seconds"]["query_layout+upload"], d["config"]["matrix_checksum"][:8]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
