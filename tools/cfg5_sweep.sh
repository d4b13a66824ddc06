#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03q
A="--cpu-sample -1 --nucleotides --ignore-genes --differences 2 --refs 100000000 --queries 12500000 --steps 2 --warmup 1"
run() { name=$1; shift; timeout 400 python bench.py $A "$@" > gpurun_out/r03q/$name.json 2>gpurun_out/r03q/$name.err; python - gpurun_out/r03q/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]["layout"]
    print("%-14s step %.2f probe %.2f resolve %.2f pos %.3g K %s slices %s delta %s chk %s" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], r["bloom_positive_per_launch"], c["class_residues"], c["slices"], c["bloom_bits_log2_delta"], d["config"]["matrix_checksum"][:8]))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
run base
COMPAIRR_HIP_LIB=$PWD/compairr_amd/lib/libcompairr_hip_ablation.so run nohbm --tunable debug=1
run k5 --tunable class_residues=5
run k4 --tunable class_residues=4
run delta0 --tunable bloom_bits_log2_delta=0
