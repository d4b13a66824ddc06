#!/bin/bash
# End-to-end comparison on the GPU box: this build's bin/compairr vs the reference binary
# (oracle/_ref/compairr) on the same two synthetic 10M-sequence AIRR TSV files.
# usage: tools/e2e_cli.sh [N]   (N sequences per set, default 10000000)
N=${1:-10000000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=$(mktemp -d /tmp/e2e.XXXXXX)
python3 - <<PY
import sys, time
sys.path.insert(0, "$R")
from compairr_amd import synth
t = time.time()
a = synth.make_set($N, 1, prefix="A", pool_size=$N // 4)
b = synth.make_set($N, 2, prefix="B", pool_size=$N // 4)
a.write_tsv_fast("$T/a.tsv"); b.write_tsv_fast("$T/b.tsv")
print("generated + written in %.1f s" % (time.time() - t))
PY
ls -la $T
for args in "-d 1" "-d 1 -i"; do
  t0=$(date +%s%N)
  COMPAIRR_HOST_TIMING=1 $R/bin/compairr -m $T/a.tsv $T/b.tsv $args -t 64 -o $T/ours.out -l $T/ours.log
  echo "ours  $args: $(( ($(date +%s%N) - t0) / 1000000 )) ms wall"
  grep -E "Reading sequences|Hashing|Query layout|Analysing|GPU kernel|Writing" $T/ours.log
  if [ -x $R/oracle/_ref/compairr ]; then
    t0=$(date +%s%N)
    $R/oracle/_ref/compairr -m $T/a.tsv $T/b.tsv $args -t 256 -o $T/ref.out -l $T/ref.log
    echo "ref   $args: $(( ($(date +%s%N) - t0) / 1000000 )) ms wall"
    grep -E "Reading sequences|Hashing sequences|Analysing|Writing" $T/ref.log | sed 's/.*\r//'
    cmp $T/ours.out $T/ref.out && echo "outputs identical"
  fi
done
# the spread of the host program's wall clock: five more runs of each
for args in "-d 1" "-d 1 -i" "-d 0"; do
  w=""
  for rep in 1 2 3 4 5; do
    t0=$(date +%s%N)
    $R/bin/compairr -m $T/a.tsv $T/b.tsv $args -t 64 -o $T/ours2.out -l $T/ours2.log
    w="$w $(( ($(date +%s%N) - t0) / 1000000 ))"
  done
  echo "ours  $args, five more runs:$w ms wall"
done
rm -rf $T
