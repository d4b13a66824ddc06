#!/bin/sh
# VGPR / SGPR / scratch / occupancy of every kernel, from the compiler's
# -Rpass-analysis=kernel-resource-usage remarks (no GPU needed).
#   tools/kernel_resources.sh            main TU + resolve + variant 2 (8 waves)
#   tools/kernel_resources.sh 1 8        variant 1, 8 waves per workgroup
cd "$(dirname "$0")/.." || exit 1
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -c -Rpass-analysis=kernel-resource-usage -o /dev/null"
run() {
  /opt/rocm/bin/hipcc $FLAGS "$@" 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name)}
       /TotalSGPRs:/ {sg=$(NF-1)} / VGPRs:/ {vg=$(NF-1)} /ScratchSize/ {sc=$(NF-1)}
       /Occupancy/ {oc=$(NF-1); printf "%-70s vgpr=%s sgpr=%s scratch=%s occ=%s\n", name, vg, sg, sc, oc}' |
  sed 's/_ZN4cmpr//; s/ENS_11ProbeParamsE//; s/EvNS_11ProbeParamsE//' | c++filt
}
if [ -n "$1" ]; then
  run -DTU_VARIANT="$1" -DTU_NW="${2:-8}" compairr_amd/csrc/probe_tu.hip
else
  run compairr_amd/csrc/compairr_hip.hip
  run -DTU_VARIANT=9 compairr_amd/csrc/probe_tu.hip
  run -DTU_VARIANT=2 -DTU_NW=8 compairr_amd/csrc/probe_tu.hip
fi
