#!/bin/bash
# usage (build container, after tools/r05_profile.sh came back): tools/r05_summarise.sh
# gpurun_out/r05_cfg{2,3,4,5} + gpurun_out/r05_cal -> profiles/r05/ and profiles/roofline_inputs.json
cd "$(dirname "$0")/.." || exit 1
mkdir -p profiles/r05 /tmp/r05isa
cp gpurun_out/r05_cal/calibration.json profiles/r05/calibration.json
H="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -S --cuda-device-only"
$H -DTU_VARIANT=2 -DTU_NW=16 -DTU_INLINE=0 compairr_amd/csrc/probe_tu.hip -o /tmp/r05isa/v2.s 2>/dev/null
$H -DTU_VARIANT=0 compairr_amd/csrc/probe_tu.hip -o /tmp/r05isa/v0.s 2>/dev/null
$H -DTU_VARIANT=3 compairr_amd/csrc/probe_tu.hip -o /tmp/r05isa/p2.s 2>/dev/null
rm -f profiles/roofline_inputs.json
python3 tools/pmc_summary.py gpurun_out/r05_cfg3 profiles/r05 cfg3 "synthetic 10M-vs-10M CDR3aa, d=1 substitutions only, V/J matched" \
    /tmp/r05isa/v2.s _ZN4cmpr17probe_rows_kernelILi20ELi1ELb0ELb1ELi16ELb0ELb0EEEvNS_11ProbeParamsE > /dev/null
python3 tools/pmc_summary.py gpurun_out/r05_cfg2 profiles/r05 cfg2 "synthetic 1M-vs-1M CDR3aa, d=0, V/J matched" \
    /tmp/r05isa/v0.s _ZN4cmpr12probe_kernelILi20ELi0ELb0ELb1EEEvNS_11ProbeParamsE > /dev/null
python3 tools/pmc_summary.py gpurun_out/r05_cfg4 profiles/r05 cfg4 "synthetic 10M-vs-10M CDR3aa, d=1 --indels, V/J matched" \
    /tmp/r05isa/v2.s _ZN4cmpr17probe_rows_kernelILi20ELi1ELb1ELb1ELi16ELb0ELb0EEEvNS_11ProbeParamsE > /dev/null
python3 tools/pmc_summary.py gpurun_out/r05_cfg5 profiles/r05 cfg5 "synthetic 12500k-vs-100M nucleotide, d=2 substitutions only --ignore-genes" \
    /tmp/r05isa/p2.s _ZN4cmpr19probe_pairs2_kernelILb0ELi16EEEvNS_11ProbeParamsE > /dev/null
python3 - <<'PY'
import json
d=json.load(open("profiles/roofline_inputs.json"))
for k,v in d["workloads"].items():
    print(k, "| kernel_us %.1f valu %.3g lds_active %.3g conflicts %.3g hbm %.3g mix %.2f busy %s" % (
        v["kernel_us_in_profile"], v["valu_insts"] or 0, v["lds_active_cycles"] or 0, v["lds_bank_conflict_cycles"] or 0, v["hbm_bytes"] or 0, v["mix_cycles_per_valu_inst"] or 0, v["busy_fraction_from_counters"]))
PY
ls profiles/r05
