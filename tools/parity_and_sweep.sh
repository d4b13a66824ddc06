#!/bin/bash
# usage (GPU box): tools/parity_and_sweep.sh TAG -- the fast parity subset, then the headline bench under a few layouts
cd "$(dirname "$0")/.." || exit 1
T=${1:-r02f}
O=gpurun_out/$T; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tiny_adversarial or synthetic_aa or synthetic_nt or ragged or scores or existence or pairs_list or many_repertoires or duplicate or errors or repeatable or long_seq" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
bash tools/tunable_sweep.sh $T "" "class_residues=2" "class_rows_unstaged=1" "class_residues=2 class_rows_unstaged=1" "heavy_threshold=200" "--indels"
