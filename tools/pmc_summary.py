#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + separate --pmc passes) for the
probe kernel into profiles/<round>/ and profiles/traffic.json.

usage: tools/pmc_summary.py <gpurun_out dir> <profiles/rNN dir> <tag> "<workload name>"

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
out = {"workload": workload, "kernel": None, "counters": {}, "kernel_stats": {}}

for f in glob.glob(os.path.join(src, "prof_stats*", "*", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, "%s_kernel_stats.csv" % tag))
    for r in csv.DictReader(open(f)):
        if "probe" in r["Name"]:
            out["kernel"] = r["Name"]
            out["kernel_stats"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                   "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}

for f in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
    acc = collections.defaultdict(list)
    dur = {}
    for r in csv.DictReader(open(f)):
        if "probe" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            out["dispatch"] = {"grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"]),
                               "vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"])}
    for k, v in acc.items():
        out["counters"][k] = {"per_launch_mean": sum(v) / len(v), "launches": len(v),
                              "kernel_ms_under_pmc": sum(dur.values()) / max(1, len(dur))}

c = out["counters"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    fetch = c["FETCH_SIZE"]["per_launch_mean"] * 1024
    write = c["WRITE_SIZE"]["per_launch_mean"] * 1024
    out["hbm_bytes_per_launch"] = 2 * fetch + write
    out["hbm_bytes_note"] = ("(2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE doubled per the "
                             "gfx950 correction in MI355X_MICROARCH.md; raw FETCH_SIZE bytes %.4g, "
                             "WRITE_SIZE bytes %.4g" % (fetch, write))
with open(os.path.join(dst, "%s_pmc_summary.json" % tag), "w") as fh:
    json.dump(out, fh, indent=1)
if "hbm_bytes_per_launch" in out:
    with open(os.path.join(os.path.dirname(dst.rstrip("/")), "traffic.json"), "w") as fh:
        json.dump({"workload": workload, "hbm_bytes_per_launch": out["hbm_bytes_per_launch"],
                   "source": os.path.join(dst, "%s_pmc_summary.json" % tag),
                   "note": out["hbm_bytes_note"]}, fh, indent=1)
print(json.dumps(out, indent=1))
