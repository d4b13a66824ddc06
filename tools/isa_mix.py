#!/usr/bin/env python3
"""Issue cost of a kernel's vector-instruction mix, from its ISA and the calibrated
per-class costs (tools/calib.hip -> calibration.json).

usage: tools/isa_mix.py <file.s> <mangled kernel name> <calibration.json> [first_line last_line]

Counts the VALU instructions of the kernel (or of the given line range of its function,
e.g. the main loop) by issue class and prints the mean cycles per wave64 instruction per
SIMD at >= 4 waves per SIMD -- what a SIMD needs to issue this mix back to back.  Classes
not calibrated one by one are priced by their encoding: plain VOP1 / VOP2 / VOPC (_e32)
like v_xor_b32, three-operand VOP3 / SDWA / 64-bit like v_alignbit_b32.
"""
import json
import re
import sys

path, kernel, cal_path = sys.argv[1:4]
lo = int(sys.argv[4]) if len(sys.argv) > 4 else None
hi = int(sys.argv[5]) if len(sys.argv) > 5 else None
cal = json.load(open(cal_path))
cost = {}
for c in cal["valu_classes"]:
    w = c["waves_per_simd"]
    cost[c["class"].split(" ")[0].rstrip(":")] = w["4"]["cycles_per_instr_at_nominal_clock"]
FAST = cost.get("v_xor_b32", 2.2)
SLOW = cost.get("v_alignbit_b32", 4.1)
PAIR = cost.get("pair", 3.2)            # v_cmp_*_e64 / v_cndmask_b32_e64, each
known = {"v_bitop3_b32": cost.get("v_bitop3_b32", FAST), "v_fma_f32": cost.get("v_fma_f32", FAST),
         "v_mov_b32": cost.get("v_mov_b32", FAST), "v_lshrrev_b32": cost.get("v_lshrrev_b32", FAST),
         "v_mul_u32_u24": cost.get("v_mul_u32_u24", SLOW), "v_pk_fma_f32": cost.get("v_pk_fma_f32", SLOW)}

lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
if lo is not None:
    body = body[lo:hi]
n = 0
cycles = 0.0
by = {}
for l in body:
    m = re.match(r"\s+(v_[a-z0-9_]+)", l)
    if not m:
        continue
    op = m.group(1)
    if op.startswith("v_readfirstlane") or op.startswith("v_readlane") or op.startswith("v_writelane"):
        c = FAST
    else:
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
        if base in known and not op.endswith("_sdwa"):
            c = known[base]
        elif op.startswith("v_cmp") and op.endswith("_e64") or op == "v_cndmask_b32_e64":
            c = PAIR
        elif op.endswith("_e32"):
            c = FAST
        else:
            c = SLOW
    n += 1
    cycles += c
    k = "%.1f" % c
    by[k] = by.get(k, 0) + 1
print(json.dumps({"kernel": kernel, "valu_instructions": n, "mean_cycles_per_instruction": cycles / max(n, 1),
                  "by_cost": by, "costs": {"fast": FAST, "slow": SLOW, "cmp/cndmask_e64": PAIR}}))
