#!/bin/bash
# usage (GPU box): tools/r06_cli_cold.sh -- where the first query layout of the command-line program goes (COMPAIRR_HIP_DEBUG)
N=${1:-10000000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=$(mktemp -d /tmp/e2e.XXXXXX)
python3 - <<PY
import sys
sys.path.insert(0, "$R")
from compairr_amd import synth
a = synth.make_set($N, 1, prefix="A", pool_size=$N // 4)
b = synth.make_set($N, 2, prefix="B", pool_size=$N // 4)
a.write_tsv_fast("$T/a.tsv"); b.write_tsv_fast("$T/b.tsv")
PY
for rep in 1 2 3; do
  t0=$(date +%s%N)
  COMPAIRR_HIP_DEBUG=1 COMPAIRR_HOST_TIMING=1 $R/bin/compairr -m $T/a.tsv $T/b.tsv -d 1 -t 64 -o $T/ours.out -l $T/ours.log 2> $T/err.txt
  echo "ours -d 1: $(( ($(date +%s%N) - t0) / 1000000 )) ms wall"
  grep -E "set_queries|host .*backend|layout kernels|set_reference|index" $T/err.txt | head -12
  grep -E "Hashing|Query layout|Analysing" $T/ours.log
done
rm -rf $T
