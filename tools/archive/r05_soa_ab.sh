#!/bin/bash
# usage (GPU box): tools/r05_soa_ab.sh -- an EXPERIMENT, not the product: a library built from the tree's sources +
# tools/r05_soa_experiment.patch (-DCMPR_EXP_SOA; exp_lib/, not in the tree): the d = 0 kernel reads the query's side of a
# verification from the tile's position-major arrays (128-byte requests shared by the lanes) instead of from the query's
# 64-byte record (a request per lane).  Parity first, then the shipped library against it on one box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05_soa; mkdir -p $O
COMPAIRR_HIP_LIB=$R/exp_lib/libcompairr_hip_soa.so timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -x \
    -k "tiny_adversarial or synthetic or long_sequences_d0 or mh_and_jaccard or many_repertoires or ragged or work_shards or routed" > $O/parity.txt 2>&1
tail -2 $O/parity.txt
line() {
python3 - $1 "$2" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    r=d["roofline"]
    print("%s: step %.4f ms (probe %.4f, rest %.4f) parity full %s" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["resolve_kernel_ms"], d["parity_vs_reference_full_size"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
B="timeout 600 python3 bench.py --cpu-sample -1 --differences 0"
P="--law cdr3 --refs 24200000 --queries 24200000 --self --repertoires 120"
run() {  # tag lib args...
  tag=$1; lib=$2; shift; shift
  if [ -n "$lib" ]; then export COMPAIRR_HIP_LIB=$R/exp_lib/$lib; else unset COMPAIRR_HIP_LIB; fi
  $B "$@" > $O/$tag.json 2> $O/$tag.err; line $O/$tag.json "$tag"
}
run base_self_d0 "" --self
run soa_self_d0 libcompairr_hip_soa.so --self
run base_d0_10m ""
run soa_d0_10m libcompairr_hip_soa.so
run base_cfg2 "" --refs 1000000 --queries 1000000
run soa_cfg2 libcompairr_hip_soa.so --refs 1000000 --queries 1000000
run base_pub_d0 "" $P
run soa_pub_d0 libcompairr_hip_soa.so $P
echo "elapsed $SECONDS s"
