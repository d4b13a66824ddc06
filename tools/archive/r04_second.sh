#!/bin/bash
# usage (GPU box): tools/r04_second.sh <tag>  -- new-API tests first, then the whole GPU suite, bench, routed emulation
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r04b}
O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "routed or device_resident or work_shards or two_streams" > $O/pytest_new.txt 2>&1; tail -15 $O/pytest_new.txt
timeout 600 python -m pytest tests/test_bench_gpu.py -x -q -m gpu > $O/pytest_bench.txt 2>&1; tail -15 $O/pytest_bench.txt
COMPAIRR_HIP_DEBUG=1 timeout 600 python3 bench.py --cpu-sample -1 > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err; tail -c 1500 $O/bench.json; echo
timeout 900 python3 tools/emulate_routed.py > $O/routed.txt 2> $O/routed.err; cat $O/routed.txt; tail -3 $O/routed.err
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
