#!/bin/bash
# usage (GPU box): tools/r05_reader_ab.sh -- the host program's wall clock on two 10M-sequence files, readers per file varied
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=$(mktemp -d /tmp/e2e.XXXXXX)
python3 - <<PY
import sys
sys.path.insert(0, "$R")
from compairr_amd import synth
a = synth.make_set(10000000, 1, prefix="A", pool_size=2500000)
b = synth.make_set(10000000, 2, prefix="B", pool_size=2500000)
a.write_tsv_fast("$T/a.tsv"); b.write_tsv_fast("$T/b.tsv")
PY
for rep in 1 2 3; do
for parts in 64 8 1; do
for t in 64 16; do
  t0=$(date +%s%N)
  COMPAIRR_READ_PARTS=$parts COMPAIRR_HOST_TIMING=1 $R/bin/compairr -m $T/a.tsv $T/b.tsv -d 1 -t $t -o $T/o.out -l $T/o.log 2> $T/err.txt
  w=$(( ($(date +%s%N) - t0) / 1000000 ))
  echo "readers $parts -t $t: $w ms wall | $(grep 'file 1 read' $T/err.txt | tr -s ' ') | $(grep 'backend done' $T/err.txt | tr -s ' ')"
done; done; done
rm -rf $T
