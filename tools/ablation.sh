#!/bin/bash
# usage (GPU box): tools/ablation.sh OUT "bench args incl. --tunable debug=N" ...   -- bench.py on the
# -DCMPR_ABLATION library (make ablation); debug bits: layout.h DBG_*
cd "$(dirname "$0")/.." || exit 1
export COMPAIRR_HIP_LIB=$PWD/compairr_amd/lib/libcompairr_hip_ablation.so
bash tools/bench_sweep.sh "$@"
