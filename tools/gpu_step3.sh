#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
T=${1:-r02e}
bash tools/gpu_sweep.sh $T "heavy_threshold=100" "heavy_threshold=200" "heavy_threshold=400" "heavy_threshold=200 class_residues=2" "bloom_bits_log2_delta=1" "heavy_threshold=200 waves_per_block=4" "heavy_threshold=200 waves_per_block=16" "heavy_threshold=200 chunk_tiles=16"
bash tools/pmc_run.sh $T/pmc 2>&1 | tail -30
