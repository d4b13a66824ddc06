#!/bin/bash
# usage (GPU box): tools/profile_all.sh <round tag, e.g. r03>  -- bench line, kernel stats and PMC passes of the
# four BASELINE workloads that run on one GPU (cfg2, cfg3, cfg4, cfg5's per-GPU shape)
R=${GRAFT_REPO_ROOT:-$(pwd)}
t=$1
cd $R
tools/profile_round.sh ${t}_cfg3 > /dev/null 2>&1
tools/profile_round.sh ${t}_cfg2 --refs 1000000 --queries 1000000 --differences 0 > /dev/null 2>&1
tools/profile_round.sh ${t}_cfg4 --indels > /dev/null 2>&1
PMC_GROUPS=essential tools/profile_round.sh ${t}_cfg5 --nucleotides --ignore-genes --differences 2 --refs 100000000 --cpu-refs 10000000 --cpu-sample 20000 \
    --queries 12500000 --steps 2 --warmup 1 > /dev/null 2>&1
for w in cfg3 cfg2 cfg4 cfg5; do echo "== $w"; tail -c 600 gpurun_out/${t}_$w/bench.json; echo; tail -2 gpurun_out/${t}_$w/bench.err; done
