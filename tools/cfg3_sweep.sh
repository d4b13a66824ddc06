#!/bin/sh
# cfg3 against a few layout tunables: step / probe / resolve per setting.
#   tools/cfg3_sweep.sh [extra bench args]   (on the GPU box; output gpurun_out/cfg3_sweep.txt)
mkdir -p gpurun_out/cfg3_sweep
out=gpurun_out/cfg3_sweep.txt
: > $out
run() {
  tag=$1; shift
  timeout 300 python bench.py --steps 20 --warmup 5 --cpu-sample -1 "$@" > gpurun_out/cfg3_sweep/$tag.json 2>/dev/null
  python - gpurun_out/cfg3_sweep/$tag.json "$tag" >> $out <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]; l = d["config"]["layout"]
    print("%-28s step %.4f probe %.4f resolve %.4f positives %d reads %d slices %d chunks %d tiles %d" % (
        sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], r["bloom_positive_per_launch"],
        r["filter_reads_per_launch"], l["slices"], l["chunks"], l["tiles"]))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run base "$@"
run deltam1 --tunable bloom_bits_log2_delta=-1 "$@"
run deltam1_ct128 --tunable bloom_bits_log2_delta=-1 --tunable chunk_tiles=128 "$@"
run deltam2 --tunable bloom_bits_log2_delta=-2 "$@"
run k2 --tunable class_residues=2 "$@"
run k0 --tunable class_residues=0 "$@"
run deltam1_k2 --tunable bloom_bits_log2_delta=-1 --tunable class_residues=2 "$@"
cat $out
