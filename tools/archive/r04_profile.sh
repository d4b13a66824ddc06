#!/bin/bash
# usage (GPU box): tools/r04_profile.sh  -- calibration, then the rocprofv3 passes of the four BASELINE workloads
# (the GPU suite runs in tools/r04_final.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r04_suite
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04_suite/pytest.txt 2>&1; tail -4 gpurun_out/r04_suite/pytest.txt
mkdir -p gpurun_out/r04_cal; tools/bin/calib > gpurun_out/r04_cal/calibration.json 2> gpurun_out/r04_cal/calib.err; tail -c 300 gpurun_out/r04_cal/calibration.json
tools/profile_all.sh r04
