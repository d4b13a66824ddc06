#!/bin/bash
# usage (GPU box): tools/r05_sweep.sh <tag> "<bench args>" name=v1,v2,... [name=...]  -- one tunable at a time
cd "$(dirname "$0")/.." || exit 1
tag=$1; args=$2; shift 2
O=gpurun_out/$tag; mkdir -p $O
for spec in "$@"; do
  name=${spec%%=*}
  for v in $(echo ${spec#*=} | tr , ' '); do
    timeout 900 python3 bench.py --cpu-sample -1 --steps 20 --warmup 4 $args --tunable $name=$v > $O/s.json 2> $O/s.err
    python3 - $O/s.json "$name=$v" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    r=d["roofline"]; l=d["config"]["layout"]
    print("%s: step %.4f probe %.4f rest %.4f positives %s pairs %s | slices %s slice_bytes %s chunks %s tiles %s | parity %s" % (
        sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], r["bloom_positive_per_launch"], r["pairs_per_launch"],
        l.get("slices"), l.get("slice_bytes"), l.get("chunks"), l.get("tiles"), d["parity_vs_reference_full_size"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json", ".err")).read()[-300:])
PY
  done
done
