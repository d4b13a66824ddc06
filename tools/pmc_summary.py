#!/usr/bin/env python3
"""Summarise rocprofv3 output of tools/profile_round.sh (kernel stats + separate --pmc
passes) for the probe and resolve kernels into profiles/<round>/ and profiles/traffic.json.

usage: tools/pmc_summary.py <gpurun_out/tag dir> <profiles/rNN dir> <tag> "<workload name>"

HBM traffic follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are
in KiB, collected in separate passes; on gfx950 FETCH_SIZE tallies 128-byte
requests at 64 bytes, so the read side is doubled.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst, tag, workload = sys.argv[1:5]
os.makedirs(dst, exist_ok=True)
KERNELS = ("probe", "resolve")


def which(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


out = {"workload": workload, "kernels": {}}
for f in glob.glob(os.path.join(src, "prof_stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, "%s_kernel_stats.csv" % tag))
    for r in csv.DictReader(open(f)):
        k = which(r["Name"])
        if k:
            out["kernels"].setdefault(k, {})["stats"] = {
                "name": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}

for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = which(r["Kernel_Name"])
        if not k:
            continue
        acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        out["kernels"].setdefault(k, {})["dispatch"] = {
            "grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"]),
            "vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
            "lds": int(r.get("LDS_Block_Size", 0) or 0)}
    for (k, c), v in acc.items():
        out["kernels"][k].setdefault("counters", {})[c] = {
            "per_launch_mean": sum(v) / len(v), "launches": len(v)}

total = 0.0
have = True
for k, d in out["kernels"].items():
    c = d.get("counters", {})
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        fetch = c["FETCH_SIZE"]["per_launch_mean"] * 1024
        write = c["WRITE_SIZE"]["per_launch_mean"] * 1024
        d["hbm_bytes_per_launch"] = 2 * fetch + write
        d["raw_fetch_bytes"] = fetch
        d["raw_write_bytes"] = write
        total += d["hbm_bytes_per_launch"]
    else:
        have = False
if have and out["kernels"]:
    out["hbm_bytes_per_step"] = total
    out["hbm_bytes_note"] = ("sum over the step's kernels of (2 x FETCH_SIZE + WRITE_SIZE) x 1024: "
                             "FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md")
for name in ("bench.json", "stats_bench.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, name)))
with open(os.path.join(dst, "%s_pmc_summary.json" % tag), "w") as fh:
    json.dump(out, fh, indent=1)
if "hbm_bytes_per_step" in out:
    with open(os.path.join(os.path.dirname(dst.rstrip("/")), "traffic.json"), "w") as fh:
        json.dump({"workload": workload, "hbm_bytes_per_launch": out["hbm_bytes_per_step"],
                   "per_kernel": {k: d.get("hbm_bytes_per_launch") for k, d in out["kernels"].items()},
                   "source": os.path.join(dst, "%s_pmc_summary.json" % tag),
                   "note": out["hbm_bytes_note"]}, fh, indent=1)
print(json.dumps(out, indent=1))
