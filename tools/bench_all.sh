#!/bin/bash
# usage (GPU box): tools/bench_all.sh <dir under gpurun_out>  -- the bench lines of the four BASELINE workloads that
# run on one GPU, priced with the committed profiles/roofline_inputs.json
O=gpurun_out/${1:-bench_all}; mkdir -p $O
timeout 600 python bench.py > $O/cfg3_bench.json 2> $O/cfg3.err
timeout 600 python bench.py --refs 1000000 --queries 1000000 --differences 0 > $O/cfg2_bench.json 2> $O/cfg2.err
timeout 600 python bench.py --indels > $O/cfg4_bench.json 2> $O/cfg4.err
timeout 1200 python bench.py --nucleotides --ignore-genes --differences 2 --refs 100000000 --queries 12500000 --steps 2 --warmup 1 --cpu-kind port > $O/cfg5_bench.json 2> $O/cfg5.err
for w in cfg3 cfg2 cfg4 cfg5; do python tools/bench_line.py $w < $O/${w}_bench.json; done
