#!/bin/bash
# usage (GPU box): tools/restats.sh <rNN>  -- only the rocprofv3 --kernel-trace --stats pass of the four workloads again
# (tools/profile_round.sh with STATS_ONLY=1), e.g. after bench.py changed but the library did not
R=${GRAFT_REPO_ROOT:-$(pwd)}
t=${1:-r06}
cd $R
export STATS_ONLY=1
tools/profile_round.sh ${t}_cfg3 > /dev/null 2>&1
tools/profile_round.sh ${t}_cfg2 --refs 1000000 --queries 1000000 --differences 0 > /dev/null 2>&1
tools/profile_round.sh ${t}_cfg4 --indels > /dev/null 2>&1
tools/profile_round.sh ${t}_cfg5 --nucleotides --ignore-genes --differences 2 --refs 100000000 --queries 12500000 --steps 2 --warmup 1 > /dev/null 2>&1
for w in cfg3 cfg2 cfg4 cfg5; do echo "== $w"; ls gpurun_out/${t}_$w/prof_stats/* | head -3; tail -c 300 gpurun_out/${t}_$w/stats_bench.json; echo; tail -2 gpurun_out/${t}_$w/stats.err; done
