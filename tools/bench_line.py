#!/usr/bin/env python3
"""Print a one-line summary of bench.py JSON output read from stdin."""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
line = [l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
r = d["roofline"]
print("%-24s %.3e q/s  step %.4f ms  probe %.4f resolve %.4f  frac %s  positives %.3e pairs %.3e  chk %s" % (
    tag, d["value"], d["ms_per_step"], r["kernel_ms"], r.get("resolve_kernel_ms") or 0.0, r.get("frac"),
    r["bloom_positive_per_launch"], r["pairs_per_launch"], d["config"]["matrix_checksum"][:8]))
