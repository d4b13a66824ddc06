#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-r02c}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tiny_adversarial or synthetic_aa or synthetic_nt or ragged or scores or existence or pairs_list or many_repertoires or duplicate or errors or repeatable or long_seq" > $O/pytest.log 2>&1
tail -25 $O/pytest.log
bash tools/gpu_sweep.sh ${1:-r02c} "" "class_residues=2" "variant=1" "--indels" "--indels variant=1"
