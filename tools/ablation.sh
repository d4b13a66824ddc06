#!/bin/bash
# usage (GPU box): tools/ablation.sh OUT "debug=64" "debug=128" ...   -- bench.py on the -DCMPR_ABLATION library
cd "$(dirname "$0")/.." || exit 1
export COMPAIRR_HIP_LIB=$PWD/compairr_amd/lib/libcompairr_hip_ablation.so
bash tools/gpu_sweep.sh "$@"
