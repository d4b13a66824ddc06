#!/usr/bin/env python3
"""Print the per-kernel average of every counter found in rocprofv3 counter_collection CSVs.

usage: tools/pmc_quick.py <dir with pmc_* subdirectories>
"""
import collections
import csv
import glob
import os
import sys

acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "pmc_*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0][:40]
        acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
        acc[(name, "~duration_us")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, c), v in sorted(acc.items()):
    print("%-42s %-28s n=%-4d avg=%.4g" % (k, c, len(v), sum(v) / len(v)))
