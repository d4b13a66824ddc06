#!/usr/bin/env python3
"""Randomised whole-program sweep on the GPU box: bin/compairr against the compiled
reference (oracle/_ref/compairr) on random small AIRR TSV files, commands -m / -x / -c
with random options.  Not collected by pytest (run by hand:
`python tests/fuzz_cli_gpu.py --seconds 300`); stops at the first difference."""

import argparse
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from compairr_amd import synth  # noqa: E402

OURS = os.path.join(ROOT, "bin", "compairr")
REF = os.path.join(ROOT, "oracle", "_ref", "compairr")


def run(binary, argv, cwd):
    p = subprocess.run([binary] + argv, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600)
    return p.returncode


def sorted_lines(path):
    with open(path, "rb") as fh:
        lines = fh.read().splitlines(keepends=True)
    return lines[:1] + sorted(lines[1:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=777)
    args = ap.parse_args()
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/compairr is missing (make -C oracle ref, in the build container)")
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    n = 0
    with tempfile.TemporaryDirectory() as tmp:
        while time.time() - t0 < args.seconds:
            nt = bool(rng.integers(0, 2))
            A = 4 if nt else 20
            cmd = ["-m", "-x", "-c"][int(rng.integers(0, 3))]
            letters = int(rng.integers(2, 5))
            kw = dict(alphabet_size=A, letters=letters, max_len=int(rng.integers(3, 10)),
                      n_v=int(rng.integers(1, 4)), n_j=int(rng.integers(1, 3)))
            a = synth.tiny_set(int(rng.integers(1, 1500)), int(rng.integers(1 << 30)),
                               n_repertoires=1 if cmd == "-x" else int(rng.integers(1, 6)),
                               prefix="A", **kw)
            b = synth.tiny_set(int(rng.integers(1, 1500)), int(rng.integers(1 << 30)),
                               n_repertoires=int(rng.integers(1, 6)), prefix="B", **kw)
            a.write_tsv(os.path.join(tmp, "a.tsv"), nucleotides=nt)
            b.write_tsv(os.path.join(tmp, "b.tsv"), nucleotides=nt)
            d = int(rng.integers(0, 3))
            argv = [cmd, "a.tsv"] + ([] if cmd == "-c" or (cmd == "-m" and rng.random() < 0.3)
                                     else ["b.tsv"])
            argv += ["-d", str(d)]
            if d == 1 and rng.random() < 0.5:
                argv.append("-i")
            if nt:
                argv.append("-n")
            if rng.random() < 0.4:
                argv.append("-g")
            if rng.random() < 0.3:
                argv.append("-f")
            if cmd != "-c" and rng.random() < 0.5:
                argv += ["-s", ["product", "min", "max", "mean"][int(rng.integers(0, 4))]]
            if cmd != "-c" and rng.random() < 0.3:
                argv.append("-a")
            pairs = cmd != "-c" and rng.random() < 0.3
            argv += ["-t", str(int(rng.integers(1, 5)))]
            outs = {}
            for name, binary in (("ours", OURS), ("ref", REF)):
                extra = ["-o", name + ".out", "-l", name + ".log"]
                if pairs:
                    extra += ["-p", name + ".pairs"]
                if name == "ours" and rng.random() < 0.3:
                    extra += ["--devices", "0,0,0"]
                rc = run(binary, argv + extra, tmp)
                outs[name] = rc
            ok = outs["ours"] == outs["ref"]
            if ok and outs["ref"] == 0:
                ok = open(os.path.join(tmp, "ours.out"), "rb").read() == \
                     open(os.path.join(tmp, "ref.out"), "rb").read()
                if ok and pairs:
                    ok = sorted_lines(os.path.join(tmp, "ours.pairs")) == \
                         sorted_lines(os.path.join(tmp, "ref.pairs"))
            if not ok:
                keep = os.path.join(ROOT, "gpurun_out", "fuzz_cli_fail")
                os.makedirs(keep, exist_ok=True)
                for f in os.listdir(tmp):
                    os.replace(os.path.join(tmp, f), os.path.join(keep, f))
                print("DIFFERENCE after %d cases: %s (exit ours %d, ref %d); files kept in %s"
                      % (n, " ".join(argv), outs["ours"], outs["ref"], keep))
                sys.exit(1)
            n += 1
    print("%d random command lines, all identical to the reference (%.0f s)" % (n, time.time() - t0))


if __name__ == "__main__":
    main()
