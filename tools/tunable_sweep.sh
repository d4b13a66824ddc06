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
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    l=d["config"]["layout"]
    k=" ".join("%s %.3f" % (a, b) for a, b in d["step_kernels_ms"].items())
    print("%-40s set %.3f ms  launch alone %.3f | %s | K=%s slices=%s tiles=%s chunks=%s cs=%s" % (
        sys.argv[1] or "(defaults)", d["ms_per_step"], d["resident_step_ms"], k, l.get("class_residues"), l.get("slices"),
        l.get("tiles"), l.get("chunks"), d["config"]["matrix_checksum"][:8]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
