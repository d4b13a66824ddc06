#!/bin/sh
# VGPR / SGPR / scratch / occupancy of every kernel, from the compiler's
# -Rpass-analysis=kernel-resource-usage remarks (no GPU needed).
cd "$(dirname "$0")/.." || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -c \
    -Rpass-analysis=kernel-resource-usage -o /dev/null compairr_amd/csrc/compairr_hip.hip 2>&1 |
awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name)}
     /TotalSGPRs:/ {sg=$(NF-1)} / VGPRs:/ {vg=$(NF-1)} /ScratchSize/ {sc=$(NF-1)}
     /Occupancy/ {oc=$(NF-1); printf "%-70s vgpr=%s sgpr=%s scratch=%s occ=%s\n", name, vg, sg, sc, oc}' |
sed 's/_ZN4cmpr//; s/ENS_11ProbeParamsE//; s/EvNS_11ProbeParamsE//'
