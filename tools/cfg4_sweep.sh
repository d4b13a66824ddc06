#!/bin/sh
# cfg4 (--indels) against a few layout tunables: step / probe / resolve per setting.
#   tools/cfg4_sweep.sh   (on the GPU box; output gpurun_out/cfg4_sweep.txt)
mkdir -p gpurun_out/cfg4_sweep
out=gpurun_out/cfg4_sweep.txt
: > $out
run() {
  tag=$1; shift
  timeout 300 python bench.py --indels --steps 20 --warmup 5 --cpu-sample -1 "$@" > gpurun_out/cfg4_sweep/$tag.json 2>/dev/null
  python - gpurun_out/cfg4_sweep/$tag.json "$tag" >> $out <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]; l = d["config"]["layout"]
    print("%-28s step %.3f probe %.3f resolve %.3f positives %d slices %d chunks %d" % (
        sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], r["bloom_positive_per_launch"], l["slices"], l["chunks"]))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run base
run delta1 --tunable bloom_bits_log2_delta=1
run deltam1 --tunable bloom_bits_log2_delta=-1
run k2 --tunable class_residues=2
run k4 --tunable class_residues=4
run ct32 --tunable chunk_tiles=32
run ct128 --tunable chunk_tiles=128
run delta1_k4 --tunable bloom_bits_log2_delta=1 --tunable class_residues=4
cat $out
