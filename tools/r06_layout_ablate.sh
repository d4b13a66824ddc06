#!/bin/bash
# usage (GPU box): tools/r06_layout_ablate.sh <tag> [bench args]  -- keys_kernel / scatter_kernel with parts left out
# (-DCMPR_ABLATION library; results become wrong, only the layout's HIP-event times are read)
tag=${1:-r06abl}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag; mkdir -p $O
export COMPAIRR_HIP_LIB=$R/compairr_amd/lib/libcompairr_hip_ablation.so
cd $R
for dbg in 0 65536 131072 262144 524288 1048576 2097152 4128768 4194304 8388608 16777216 33554432 67108864 $((4194304+8388608)) $((4194304+8388608+67108864)); do
  timeout 300 python3 bench.py --cpu-sample -1 --steps 5 --warmup 2 --tunable debug=$dbg "$@" > $O/b.json 2> $O/b.err
  python3 - $O/b.json $dbg <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("debug=%-10s %s" % (sys.argv[2], {k: round(v,3) for k,v in d["step_kernels_ms"].items()}))
except Exception as e:
    print("debug=%s FAILED %s" % (sys.argv[2], e))
PY
done
