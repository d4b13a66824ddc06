#!/bin/bash
# usage (GPU box): tools/r04_suite.sh TAG  -- the whole GPU suite, output under gpurun_out/TAG
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
T=${1:-r04_suite}
mkdir -p gpurun_out/$T
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/$T/pytest.txt 2>&1; tail -8 gpurun_out/$T/pytest.txt
