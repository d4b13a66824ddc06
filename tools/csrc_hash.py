#!/usr/bin/env python3
"""sha256 over the sources the HIP library is built from (compairr_amd/csrc/*, include/*.h):
stamps a counter profile with the code it was taken on (bench.py: counters_stale)."""
import glob
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash(root=ROOT):
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "compairr_amd", "csrc", "*")) +
                   glob.glob(os.path.join(root, "include", "*.h")))
    for f in files:
        if os.path.isfile(f):
            h.update(os.path.relpath(f, root).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    sys.stdout.write(csrc_hash() + "\n")
