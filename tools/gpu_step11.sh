#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
bash tools/gpu_sweep2.sh ${1:-r02ss} \
  "--nucleotides --ignore-genes --differences 2 --refs 100000000 --queries 2000000 --steps 2 --warmup 1 --tunable small_slice_tiles=0" \
  "--nucleotides --ignore-genes --differences 2 --refs 100000000 --queries 2000000 --steps 2 --warmup 1 --tunable small_slice_tiles=0 --tunable chunk_tiles=8" \
  "--nucleotides --ignore-genes --differences 2 --refs 100000000 --queries 12500000 --steps 2 --warmup 1 --tunable small_slice_tiles=0" \
  "--nucleotides --ignore-genes --differences 2 --refs 5000000 --queries 2000000 --steps 3 --warmup 1 --tunable small_slice_tiles=0" \
  "--queries 1250000 --scaling weak --tunable small_slice_tiles=0" \
  "--queries 100000 --scaling weak --tunable small_slice_tiles=0"
