#!/bin/bash
# usage (GPU box): tools/r06_layout_ablate.sh <tag> [bench args]  -- keys_kernel / scatter_kernel with parts left out
# (-DCMPR_ABLATION library; results become wrong, only the layout's HIP-event times are read)
tag=${1:-r06abl}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag; mkdir -p $O
export COMPAIRR_HIP_LIB=$R/compairr_amd/lib/libcompairr_hip_ablation.so
cd $R
NOITEMS=$((131072+8388608))
for dbg in 0 65536 $NOITEMS $((NOITEMS+65536)) $((NOITEMS+262144)) $((NOITEMS+524288)) $((NOITEMS+1048576)) $((NOITEMS+2097152)) $((NOITEMS+4128768-131072)) \
           4194304 8388608 16777216 33554432 67108864 $((4194304+8388608)) $((4194304+8388608+67108864)); do
  timeout 300 python3 bench.py --cpu-sample -1 --steps 5 --warmup 2 --tunable debug=$dbg "$@" > $O/b.json 2> $O/b.err
  python3 - $O/b.json $dbg <<'PY'
import json,sys
names={65536:"NO_RANK",131072:"NO_ITEM_COUNT",262144:"NO_HASH",524288:"NO_TOTALS",1048576:"NO_CLASSKEY",2097152:"NO_TMP_WRITES",
       4194304:"S_NO_REC",8388608:"S_NO_ITEMS",16777216:"S_SEQ_REC",33554432:"S_NO_ITEM_WRITE",67108864:"S_NO_BASE"}
d=int(sys.argv[2]); lab="+".join(n for b,n in names.items() if d&b) or "-"
try:
    j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    k=j["step_kernels_ms"]
    print("%-60s keys %.3f scatter %.3f tiles %.3f" % (lab, k["keys"], k["scatter"], k.get("tiles", 0.0)))
except Exception as e:
    print("%-60s FAILED %s" % (lab, e))
PY
done
