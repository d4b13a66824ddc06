#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
T=${1:-r02o}
O=gpurun_out/$T; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tiny_adversarial or synthetic_aa or synthetic_nt or ragged or scores or existence or pairs_list or long_seq" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
bash tools/gpu_sweep.sh $T "" "--indels" "--self" "--differences 0 --queries 1000000 --refs 1000000"
bash tools/profile_round.sh $T/prof --steps 20 --warmup 5 > $O/profile.log 2>&1
cp profiles/r02/calibration.json $O/prof/ 2>/dev/null
python3 tools/pmc_summary.py $O/prof $O/prof/summary v8 "synthetic 10M-vs-10M CDR3aa, d=1 substitutions only, V/J matched" > $O/summary.log 2>&1
tail -5 $O/summary.log
cat $O/prof/bench.json | head -c 2500
