#!/bin/bash
# usage (GPU box): tools/r05_pin_ab.sh -- an EXPERIMENT, not the product: two libraries built from the tree's sources +
# tools/r05_pin_experiment.patch (exp_lib/, not in the tree):
#   pin        -DCMPR_EXP_PIN: verify_candidate's loads are used in front of the tag test (the compiler cannot sink
#              the slot's residues and -- EAGER, the d = 0 kernel -- the query's record behind it)
#   pin_eager  the same + -DCMPR_EXP_RES_EAGER: resolve_kernel requests the query's record with the slot, too
# against the shipped library, on one box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/r05_pin; mkdir -p $O
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
B="timeout 600 python3 bench.py --cpu-sample -1"
P="--law cdr3 --refs 24200000 --queries 24200000 --self --repertoires 120"
run() {  # tag lib args...
  tag=$1; lib=$2; shift; shift
  if [ -n "$lib" ]; then export COMPAIRR_HIP_LIB=$R/exp_lib/$lib; else unset COMPAIRR_HIP_LIB; fi
  $B "$@" > $O/$tag.json 2> $O/$tag.err; line $O/$tag.json "$tag"
}
run base_self_d0 "" --differences 0 --self
run pin_self_d0 libcompairr_hip_pin.so --differences 0 --self
run pin_cfg2 libcompairr_hip_pin.so --differences 0 --refs 1000000 --queries 1000000
run pin_d0_10m libcompairr_hip_pin.so --differences 0
run pin_pub_d0 libcompairr_hip_pin.so $P --differences 0
run base_pub_d1 "" $P --differences 1
run pin_pub_d1 libcompairr_hip_pin.so $P --differences 1
run pineager_pub_d1 libcompairr_hip_pin_eager.so $P --differences 1
run pin_cfg3 libcompairr_hip_pin.so
run pineager_cfg3 libcompairr_hip_pin_eager.so
run pin_self_d1 libcompairr_hip_pin.so --self
run pineager_self_d1 libcompairr_hip_pin_eager.so --self
run pineager_cdr3i libcompairr_hip_pin_eager.so --law cdr3 --indels
echo "elapsed $SECONDS s"
