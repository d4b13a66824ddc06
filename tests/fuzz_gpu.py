#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: HIP path (through the C ABI) against the
oracle over random inputs, options and layout tunables.  Not collected by pytest
(run it by hand: `python tests/fuzz_gpu.py --seconds 300`); stops at the first
mismatch and prints the configuration that produced it.  A seeded, time-boxed slice of it
runs in the GPU suite (tests/test_fuzz_gpu.py)."""

import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import _oracle  # noqa: E402
from _routed import routed_contexts  # noqa: E402
from compairr_amd import HipOverlap, Options, synth  # noqa: E402


def random_case(rng):
    nt = bool(rng.integers(0, 2))
    A = 4 if nt else 20
    tiny = rng.random() < 0.5
    n1 = int(rng.integers(1, 30000))
    n2 = int(rng.integers(1, 30000))
    if tiny:
        letters = int(rng.integers(2, 5))
        max_len = int(rng.integers(3, 12))
        nrep = int(rng.integers(1, 5))
        a = synth.tiny_set(min(n1, 3000), int(rng.integers(1 << 30)), alphabet_size=A, letters=letters,
                           max_len=max_len, n_repertoires=nrep, prefix="A")
        b = synth.tiny_set(min(n2, 3000), int(rng.integers(1 << 30)), alphabet_size=A, letters=letters,
                           max_len=max_len, n_repertoires=nrep, prefix="B")
        genes = dict(n_v_genes=2, n_j_genes=2)
    else:
        pool = int(rng.integers(50, 5000))
        ps = int(rng.integers(1 << 30))
        a = synth.make_set(n1, int(rng.integers(1 << 30)), pool_seed=ps, pool_size=pool, prefix="A",
                           nucleotides=nt, n_repertoires=int(rng.integers(1, 40)))
        b = synth.make_set(n2, int(rng.integers(1 << 30)), pool_seed=ps, pool_size=pool, prefix="B",
                           nucleotides=nt, n_repertoires=int(rng.integers(1, 40)))
        genes = dict(n_v_genes=synth.N_V, n_j_genes=synth.N_J)
    d = int(rng.integers(0, 3))
    indels = bool(d == 1 and rng.integers(0, 2))
    if d == 2 and not tiny and not nt:
        a = a.subset(slice(0, min(a.n, 4000)))          # oracle time
    if d == 2 and nt and not tiny:
        a = a.subset(slice(0, min(a.n, 1500)))
    o = Options(differences=d, indels=indels, nucleotides=nt,
                ignore_genes=bool(rng.integers(0, 2)), ignore_counts=bool(rng.integers(0, 2)),
                score=["product", "min", "max", "mean"][int(rng.integers(0, 4))], **genes)
    tun = {}
    if rng.random() < 0.7:
        tun["slice_words_log2"] = int(rng.integers(2, 13))
    if rng.random() < 0.5:
        tun["class_residues"] = int(rng.integers(0, 9 if nt else 5))
    if rng.random() < 0.3:
        tun["heavy_threshold"] = int(rng.integers(0, 50))
    if rng.random() < 0.4:
        # odd and even anchors, also behind the start: pairs that hold one or two class
        # positions, and with -i the pairs in front of them (read where they lie)
        tun["class_anchor"] = int(rng.integers(0, 12))
    if rng.random() < 0.3:
        tun["chunk_tiles"] = int(rng.integers(1, 65))
    if rng.random() < 0.3:
        tun["small_slice_tiles"] = int(rng.integers(0, 65))
    if rng.random() < 0.3:
        tun["waves_per_block"] = [4, 8, 16][int(rng.integers(0, 3))]
    if rng.random() < 0.3:
        tun["pos_capacity"] = int(rng.integers(1, 5000))
    if rng.random() < 0.2:
        tun["pos_segments"] = [1, 2, 64, 256][int(rng.integers(0, 4))]
    if rng.random() < 0.2:
        tun["deferred_resolve"] = 0
    if rng.random() < 0.2:
        tun["table_log2_delta"] = int(rng.integers(0, 3))
    r = rng.random()
    if r < 0.1:
        tun["variant"] = 0
    elif r < 0.3:
        tun["variant"] = 1
    elif r < 0.6:
        tun["variant"] = 2                                # the row filter, also where it is not the default
    if rng.random() < 0.25:
        tun["chunk_deal"] = 0                             # every chunk dealt statically (default: by counters)
    if rng.random() < 0.3:
        tun["bucket_bitmap"] = int(rng.integers(0, 2))    # the bucket bitmap in front of the record table
    if rng.random() < 0.35:
        tun["page_budget"] = int(rng.integers(1, 200))    # overfull slices get pages (variant 2, d = 1)
        if rng.random() < 0.3:
            tun["slice_pages"] = int(rng.integers(0, 4))
    if rng.random() < 0.5:
        tun["direct_slices_log2"] = int(rng.integers(0, 7))   # pseudo-slices of the d = 0 layout (variant 0)
    if rng.random() < 0.2:
        tun["class_rows_unstaged"] = 1
    if rng.random() < 0.4:
        tun["sub2_items"] = 1                             # (takes effect for nucleotides, d = 2, variant 1)
    if rng.random() < 0.3:
        tun["host_threads"] = int(rng.integers(1, 9))
    if rng.random() < 0.3:
        tun["narrow_upload"] = 1                          # (by default only for sets of a million and more)
    # round 6's query layout against round 5's form of it: item counters / group ranks per workgroup in LDS,
    # hashes recomputed by fill_tiles_kernel, tables in LDS -- each switched off now and then
    if rng.random() < 0.2:
        tun["item_wg"] = 0
    if rng.random() < 0.2:
        tun["layout_recompute"] = 0
    if rng.random() < 0.2:
        tun["layout_zob_lds"] = 0
    if rng.random() < 0.4:
        # (amino acids at d = 1: 0 = per-slot arrays after all, 2 = records, hashed by the kernel)
        tun["record_tiles"] = 0 if rng.random() < 0.6 else 2
    same = rng.random() < 0.2
    return a, (a if same else b), o, tun


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=12345)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    n = 0
    while time.time() - t0 < args.seconds:
        a, b, o, tun = random_case(rng)
        want, ost = _oracle.overlap(a, b, o, threads=8)
        want = _oracle.integer_cells(want, o)
        # sometimes as work shards (each context lays out only what it works on): the
        # matrices, counters and pair lists of the shards add up to the whole
        shards = int(rng.integers(2, 6)) if (tun.get("variant", -1) != 0 and rng.random() < 0.2) else 1
        got, nmatch, nvar, npairs = None, 0, 0, 0
        ok = True
        # ... and sometimes the shards get their queries routed (every context keys a share,
        # the records change hands: cmpr_route_queries / _pack / cmpr_set_queries_routed)
        routed = shards > 1 and rng.random() < 0.5
        if routed:
            with HipOverlap(o) as h:
                for k in ["variant"] + [k for k in tun if k != "variant"]:
                    if k in tun:
                        h.set_tunable(k, tun[k])
                h.set_reference(b, a.longest)
                routed = h.get_tunable("variant") != 0
        if routed:
            order = {k: tun[k] for k in ["variant"] + [k for k in tun if k != "variant"] if k in tun}
            hs = routed_contexts(a, b, o, shards, order)
            try:
                for h in hs:
                    m = h.overlap_matrix()
                    st = h.stats()
                    got = m if got is None else got + m
                    nmatch += st.matches
                    nvar += st.variants
                    if n % 4 == 0:
                        npairs += len(h.overlap_pairs())
            finally:
                for h in hs:
                    h.close()
        for index in range(0 if routed else shards):
            with HipOverlap(o) as h:
                order = ["variant"] + [k for k in tun if k != "variant"]
                for k in order:
                    if k in tun:
                        h.set_tunable(k, tun[k])
                if shards > 1:
                    h.set_tunable("work_shard_count", shards)
                    h.set_tunable("work_shard_index", index)
                h.set_reference(b, a.longest)
                if h.get_tunable("variant") == 0 and shards > 1:
                    shards = 1                            # (long sequences fell back to the unsliced kernel)
                    h.set_tunable("work_shard_count", 1)
                    h.set_tunable("work_shard_index", 0)
                # (every third case: the query set handed over as device arrays -- the path bench.py times)
                if n % 3 == 1:
                    view, keep = h.device_view(a)
                    h.set_queries_device(view)
                    del keep
                else:
                    h.set_queries(a)
                m = h.overlap_matrix()
                st = h.stats()
                # repeated launches (the third may run without its redo pass) give the same
                for _ in range(2 if n % 3 == 0 else 0):
                    ok = ok and np.array_equal(h.overlap_matrix(), m)
                if n % 4 == 0:
                    npairs += len(h.overlap_pairs())
            got = m if got is None else got + m
            nmatch += st.matches
            nvar += st.variants
            if shards == 1:
                break
        ok = ok and np.array_equal(got, want) and nmatch == ost.matches and nvar == ost.variants
        if ok and n % 4 == 0:
            ok = npairs == ost.matches
        st = None
        if not ok:
            print("MISMATCH after %d cases: n1=%d n2=%d opt=%s tun=%s shards=%d routed=%s"
                  % (n, a.n, b.n, o, tun, shards, routed))
            sys.exit(1)
        n += 1
    print("%d random cases, all bit-exact (%.0f s)" % (n, time.time() - t0))


if __name__ == "__main__":
    main()
