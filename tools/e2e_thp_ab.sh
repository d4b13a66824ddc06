#!/bin/bash
# usage (GPU box): tools/e2e_thp_ab.sh [N]  -- bin/compairr on two N-sequence files with and without the reader's
# huge-page advice (COMPAIRR_NO_HUGEPAGES=1), alternating, same box: wall clock and the host's phase marks
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
for rep in 1 2 3 4 5 6; do
  for mode in huge plain; do
    if [ $mode = plain ]; then export COMPAIRR_NO_HUGEPAGES=1; else unset COMPAIRR_NO_HUGEPAGES; fi
    t0=$(date +%s%N)
    COMPAIRR_HOST_TIMING=1 $R/bin/compairr -m $T/a.tsv $T/b.tsv -d 1 -t 64 -o $T/o.$mode -l $T/l.$mode 2> $T/e.$mode
    w=$(( ($(date +%s%N) - t0) / 1000000 ))
    echo "$mode wall $w ms | $(grep -E 'both sets ready|backend done' $T/e.$mode | sed 's/\[host *//; s/ ms\]//' | tr '\n' ';') $(grep -E 'Hashing|Query layout' $T/l.$mode | sed 's/.*(//; s/s)//' | tr '\n' ' ')"
  done
done
cmp $T/o.huge $T/o.plain && echo "outputs identical"
rm -rf $T
