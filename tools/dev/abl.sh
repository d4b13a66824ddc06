export COMPAIRR_HIP_LIB=$PWD/compairr_amd/lib/libcompairr_hip_ablation.so
for d in 0 16 8 128 256; do
  timeout 200 python bench.py --cpu-sample -1 --steps 5 --warmup 2 --indels --tunable debug=$d 2>/dev/null | python tools/bench_line.py debug=$d
done
