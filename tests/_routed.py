"""N library contexts on one device standing for the N ranks of a multi-GPU run (test
infrastructure shared by tests/test_gpu_parity.py and tests/fuzz_gpu.py)."""

import numpy as np

from compairr_amd import HipOverlap


def routed_contexts(a, b, o, count, tun, shares=None):
    """`count` contexts on this device, each given a contiguous share of `a` (or `shares`:
    [(first index, subset)]); the records change hands with device-to-device copies -- what
    the all-to-all of compairr_amd.dist.exchange_queries does between GPUs.  Returns the
    contexts, laid out and ready."""
    import torch
    from compairr_amd.dist import shard_bounds
    hs = []
    for index in range(count):
        h = HipOverlap(o)
        for k, v in tun.items():
            h.set_tunable(k, v)
        h.set_tunable("work_shard_count", count)
        h.set_tunable("work_shard_index", index)
        h.set_reference(b, a.longest)
        hs.append(h)
    if shares is None:
        shares = []
        for index in range(count):
            lo, hi = shard_bounds(a.n, index, count)
            shares.append((lo, a.subset(slice(lo, hi))))
    sends, totals, rb = [], np.zeros(a.n_repertoires), 0
    for h, (first, share) in zip(hs, shares):
        counts, rb, tot = h.route_queries(share, first, count)
        totals += tot
        buf = torch.empty(max(int(counts.sum()), 1) * rb, dtype=torch.uint8, device="cuda")
        h.route_pack(buf.data_ptr(), int(counts.sum()) * rb)
        sends.append((counts, buf))
    for d, h in enumerate(hs):
        runs = []
        for counts, buf in sends:
            start = int(counts[:d].sum()) * rb
            runs.append(buf[start:start + int(counts[d]) * rb])
        recv = torch.cat(runs) if runs else torch.empty(0, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        h.set_queries_routed(recv.data_ptr(), recv.numel() // rb, a.n_repertoires, a.n, totals)
    return hs
