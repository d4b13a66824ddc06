#!/bin/bash
# usage (GPU box): tools/r05_resolve_ablate.sh <tag>  -- what does each random access of resolve_kernel cost?
# (-DCMPR_ABLATION library; results of the switched runs are wrong by design)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-r05r}; mkdir -p $O
export COMPAIRR_HIP_LIB=$PWD/compairr_amd/lib/libcompairr_hip_ablation.so
for wl in "--self" "--law cdr3 --refs 24200000 --queries 24200000 --self --repertoires 120"; do
for dbg in 0 512 1024 2048 1536 3072 3584 4096 7680; do
  python3 bench.py --cpu-sample -1 --steps 10 --warmup 3 $wl --tunable debug=$dbg > $O/x.json 2> $O/x.err
  python3 - $O/x.json "$wl debug=$dbg" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    r=d["roofline"]
    print("%s: probe %.4f resolve+reduce %.4f positives %s" % (sys.argv[2], r["kernel_ms"], r["resolve_kernel_ms"], r["bloom_positive_per_launch"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
done
