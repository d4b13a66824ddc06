#!/usr/bin/env python3
"""Summarise rocprofv3 output of tools/profile_round.sh (kernel stats + separate --pmc
passes) into profiles/<round>/<tag>_pmc_summary.json, profiles/traffic.json and
profiles/roofline_inputs.json (what bench.py prices its `roofline` object on).

usage: tools/pmc_summary.py <gpurun_out/tag dir> <profiles/rNN dir> <tag> "<workload name>" [<kernel.s> <mangled kernel name>]

Kernels are kept apart by their full name (the fast and the redo form of the probe
kernel are different instantiations); "probe" below is the probe kernel with the
longest mean duration, "resolve" likewise.

HBM traffic follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are
in KiB, collected in separate passes; on gfx950 FETCH_SIZE tallies 128-byte
requests at 64 bytes, so the read side is doubled -- for the probe kernel, whose reads are
wide and coalesced; resolve_kernel's random 64-byte lines are taken as counted.
"""
import collections
import csv
import re
import glob
import json
import os
import shutil
import sys

src, dst, tag, workload = sys.argv[1:5]
os.makedirs(dst, exist_ok=True)
root = os.path.dirname(dst.rstrip("/"))

per = collections.defaultdict(lambda: {"dur": [], "ctr": collections.defaultdict(list), "dispatch": None})
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        d = per[name]
        d["ctr"][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (name, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key)
            d["dur"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        d["dispatch"] = {"grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"]),
                         "vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                         "lds": int(r.get("LDS_Block_Size", 0) or 0)}


def pick(word):
    best = None
    for name, d in per.items():
        if word in name and d["dur"]:
            m = sum(d["dur"]) / len(d["dur"])
            if best is None or m > best[1]:
                best = (name, m)
    return best


out = {"workload": workload, "kernels": {}}
for f in glob.glob(os.path.join(src, "prof_stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, "%s_kernel_stats.csv" % tag))
    stats = {r["Name"]: r for r in csv.DictReader(open(f))}
else:
    stats = locals().get("stats", {})

# the step's kernels: the layout's three (round 6: a step is a query set from its device arrays to the matrix),
# then probe and resolve
WORDS = {"probe": "probe", "resolve": "resolve", "keys": "keys_kernel", "scatter": "scatter_kernel",
         "tiles": "fill_tiles_kernel"}
for k in ("keys", "scatter", "tiles", "probe", "resolve"):
    b = pick(WORDS[k])
    if not b:
        continue
    name, mean_us = b
    d = per[name]
    e = {"name": name, "dispatch": d["dispatch"], "pmc_pass_mean_duration_us": mean_us,
         "counters": {c: {"per_launch_mean": sum(v) / len(v), "per_launch_max": max(v), "launches": len(v)}
                      for c, v in d["ctr"].items()}}
    e["longest_launch_us"] = max(d["dur"])
    if name in stats:
        r = stats[name]
        e["stats"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                      "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
    c = e["counters"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        fetch = c["FETCH_SIZE"]["per_launch_mean"] * 1024
        write = c["WRITE_SIZE"]["per_launch_mean"] * 1024
        # the guide's x 2 on FETCH_SIZE is for 128-byte requests tallied at 64: the probe kernel's wide,
        # coalesced reads (slice copies, tile data).  resolve_kernel asks for single random 64-byte lines:
        # its raw figure stands
        # ... and so does the filterless d = 0 kernel's (probe_kernel<A, 0, ..>: one random slot per query;
        # its coalesced reads of the queries' records are then undercounted, not the slots doubled)
        # (the layout kernels stream their inputs 4 to 16 bytes per lane, coalesced: doubled like the probe kernel's;
        #  the guide calls widths other than 16 bytes per lane uncalibrated -- the raw figures stand beside it)
        direct = re.search(r"probe_kernel<\d+, 0,", name) is not None
        e["fetch_factor"] = 1 if (k == "resolve" or direct) else 2
        e["hbm_bytes_per_launch"] = e["fetch_factor"] * fetch + write
        e["raw_fetch_bytes"] = fetch
        e["raw_write_bytes"] = write
        e["hbm_bytes_per_launch_longest"] = (e["fetch_factor"] * c["FETCH_SIZE"]["per_launch_max"] +
                                             c["WRITE_SIZE"]["per_launch_max"]) * 1024
    # issue-side utilisation straight from the counters: SQ_ACTIVE_INST_* count in units
    # of 4 cycles per SIMD, SQ_BUSY_CU_CYCLES in cycles per CU (4 SIMDs): their ratio is
    # the fraction of SIMD cycles in which the unit was executing an instruction
    if "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CU_CYCLES" in c:
        busy = c["SQ_BUSY_CU_CYCLES"]["per_launch_mean"]
        e["busy_fraction"] = {u: c[n]["per_launch_mean"] / busy
                              for u, n in (("valu", "SQ_ACTIVE_INST_VALU"), ("scalar", "SQ_ACTIVE_INST_SCA"),
                                           ("lds", "SQ_ACTIVE_INST_LDS")) if n in c}
    out["kernels"][k] = e

tot = [e.get("hbm_bytes_per_launch") for k, e in out["kernels"].items() if k in ("probe", "resolve")]
if tot and all(x is not None for x in tot):
    out["hbm_bytes_per_step"] = sum(tot)
    out["hbm_bytes_note"] = ("sum over the step's kernels of (f x FETCH_SIZE + WRITE_SIZE) x 1024: f = 2 for the probe "
                             "kernel (the gfx950 correction of MI355X_MICROARCH.md: 128-byte requests tallied at 64), "
                             "f = 1 for resolve_kernel (random 64-byte lines)")
for name in ("bench.json", "stats_bench.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, name)))
with open(os.path.join(dst, "%s_pmc_summary.json" % tag), "w") as fh:
    json.dump(out, fh, indent=1)
root = os.path.dirname(dst.rstrip("/"))
if "hbm_bytes_per_step" in out:
    with open(os.path.join(root, "traffic.json"), "w") as fh:
        json.dump({"workload": workload, "hbm_bytes_per_launch": out["hbm_bytes_per_step"],
                   "per_kernel": {k: d.get("hbm_bytes_per_launch") for k, d in out["kernels"].items()},
                   "source": os.path.join(dst, "%s_pmc_summary.json" % tag),
                   "note": out["hbm_bytes_note"]}, fh, indent=1)

# ---- what bench.py's roofline object is priced on: one entry per workload ----
# (optional 5th / 6th argument: the kernel's ISA file and mangled name, for the issue
#  cost of its instruction mix -- tools/isa_mix.py)
cal_path = os.path.join(dst, "calibration.json")
if "probe" in out["kernels"] and os.path.exists(cal_path):
    cal = json.load(open(cal_path))
    c = out["kernels"]["probe"]["counters"]
    g = lambda n: c[n]["per_launch_mean"] if n in c else None
    cost = {x["class"].split(" ")[0].rstrip(":"): x["waves_per_simd"]["4"]["cycles_per_instr_at_nominal_clock"]
            for x in cal["valu_classes"]}
    f_eff = None
    if g("GRBM_GUI_ACTIVE"):
        # effective shader clock of this kernel: GRBM_GUI_ACTIVE is summed over the 8 XCDs
        f_eff = g("GRBM_GUI_ACTIVE") / 8 / (out["kernels"]["probe"]["pmc_pass_mean_duration_us"] * 1e-6)
    mix = None
    if len(sys.argv) > 6:
        import subprocess
        mix = json.loads(subprocess.check_output([sys.executable, os.path.join(os.path.dirname(__file__), "isa_mix.py"),
                                                  sys.argv[5], sys.argv[6], cal_path]).decode())
    bench = {}
    bp = os.path.join(src, "bench.json")
    if os.path.exists(bp) and os.path.getsize(bp):
        lines = [l for l in open(bp).read().splitlines() if l.startswith("{")]
        if lines:
            bench = json.loads(lines[-1])
    inp_path = os.path.join(root, "roofline_inputs.json")
    try:
        inputs = json.load(open(inp_path))
        assert "workloads" in inputs
    except Exception:
        inputs = {"workloads": {}}
    sha = None
    sp = os.path.join(src, "csrc_sha256.txt")
    if os.path.exists(sp):
        sha = open(sp).read().strip()
    if inputs.get("csrc_sha256") not in (None, sha):
        # counters of another state of the sources: they are not mixed
        inputs["workloads"] = {}
    inputs["csrc_sha256"] = sha
    inputs["calibration"] = {
        "from": cal_path, "cus": cal["cus"], "simds": cal["cus"] * 4, "nominal_clock_hz": cal["clock_mhz"] * 1e6,
        # cycles per wave64 instruction per SIMD at >= 4 waves per SIMD, wall time x nominal clock
        "valu_cycles_fast": cost.get("v_xor_b32"), "valu_cycles_slow": cost.get("v_alignbit_b32"),
        "valu_cycles_by_class": cost,
    }
    inputs["workloads"][workload] = {
        "source": os.path.join(dst, "%s_pmc_summary.json" % tag),
        "kernel": out["kernels"]["probe"]["name"],
        "kernel_us_in_profile": out["kernels"]["probe"]["pmc_pass_mean_duration_us"],
        "valu_insts": g("SQ_INSTS_VALU"), "salu_insts": g("SQ_INSTS_SALU"),
        "lds_active_cycles": g("SQ_LDS_IDX_ACTIVE"),
        "lds_bank_conflict_cycles": g("SQ_LDS_BANK_CONFLICT"),
        "hbm_bytes": out["kernels"]["probe"].get("hbm_bytes_per_launch"),
        "effective_clock_hz": f_eff,
        "busy_fraction_from_counters": out["kernels"]["probe"].get("busy_fraction"),
        "mix_cycles_per_valu_inst": mix["mean_cycles_per_instruction"] if mix else None,
        "mix": mix,
        # work of the profiled (N = 1) launch: a shard's share is priced against these
        "filter_reads": ((bench.get("roofline_kernels") or {}).get("probe") or bench.get("roofline") or {}).get("filter_reads_per_launch"),
        "variants": ((bench.get("roofline_kernels") or {}).get("probe") or bench.get("roofline") or {}).get("variants_per_launch"),
        # the layout's kernels (bench.py layout_rooflines): counted HBM bytes per launch of the 10M-query call
        # (the longest launch of the kernel in the profile: the device path lays the whole set out in one)
        "layout_kernels": {WORDS[k]: {"hbm_bytes": out["kernels"][k].get("hbm_bytes_per_launch_longest"),
                                      "us_in_profile": out["kernels"][k].get("longest_launch_us")}
                           for k in ("keys", "scatter", "tiles") if k in out["kernels"]},
    }
    with open(inp_path, "w") as fh:
        json.dump(inputs, fh, indent=1)
print(json.dumps(out, indent=1)[:3000])
