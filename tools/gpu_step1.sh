#!/bin/bash
# first GPU step of round 2: calibration + variant-2 parity + bench
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r02a; mkdir -p $O
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $O/calib tools/calib.hip && $O/calib > $O/calibration.json 2>$O/calib.err
cat $O/calibration.json
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tiny_adversarial or synthetic_aa or synthetic_nt or ragged or scores or existence or pairs_list or many_repertoires" > $O/pytest.log 2>&1
tail -15 $O/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_v2.json 2>$O/bench_v2.err; tail -3 $O/bench_v2.err; cat $O/bench_v2.json | head -c 3000
timeout 600 python bench.py --steps 10 --warmup 3 --tunable variant=1 --cpu-sample -1 > $O/bench_v1.json 2>$O/bench_v1.err; cat $O/bench_v1.json | head -c 1500
