"""ctypes binding of the C ABI in ``include/compairr_hip.h``.

There is deliberately no fallback: if ``libcompairr_hip.so`` has not been built
(``make lib`` / ``__graft_entry__.build()``) or no HIP device is present, the
calls fail loudly."""

from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from .sets import RepertoireSet

SCORES = {"product": 0, "ratio": 1, "min": 2, "max": 3, "mean": 4, "mh": 5, "jaccard": 6}

EXPORTS = [
    "cmpr_abi_version", "cmpr_create", "cmpr_destroy", "cmpr_last_error",
    "cmpr_set_reference", "cmpr_set_queries", "cmpr_overlap_matrix",
    "cmpr_overlap_matrix_f64", "cmpr_overlap_matrix_device", "cmpr_get_stats",
    "cmpr_get_kernel_times",
    "cmpr_rows", "cmpr_cols", "cmpr_set_tunable", "cmpr_get_tunable",
    "cmpr_count_duplicates", "cmpr_overlap_pairs",
    "cmpr_set_reference_device", "cmpr_set_queries_device",
    "cmpr_route_queries", "cmpr_route_pack", "cmpr_set_queries_routed",
    "cmpr_warm_up", "cmpr_warm_up_sized",
]


class HipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("libcompairr_hip error %d: %s" % (code, msg))
        self.code = code


class _Options(C.Structure):
    _fields_ = [("differences", C.c_int32), ("indels", C.c_int32),
                ("ignore_genes", C.c_int32), ("ignore_counts", C.c_int32),
                ("score", C.c_int32), ("alphabet_size", C.c_int32),
                ("n_v_genes", C.c_uint32), ("n_j_genes", C.c_uint32),
                ("device", C.c_int32), ("existence", C.c_int32),
                ("reserved", C.c_int32 * 6)]


class _SetView(C.Structure):
    _fields_ = [("n", C.c_uint64), ("residues", C.c_void_p), ("offsets", C.c_void_p),
                ("v_gene", C.c_void_p), ("j_gene", C.c_void_p),
                ("repertoire", C.c_void_p), ("count", C.c_void_p),
                ("n_repertoires", C.c_uint32), ("reserved", C.c_uint32)]


class _Stats(C.Structure):
    _fields_ = [("queries", C.c_uint64), ("variants", C.c_uint64),
                ("bloom_positive", C.c_uint64), ("hash_equal", C.c_uint64),
                ("matches", C.c_uint64), ("algorithmic_bytes", C.c_uint64),
                ("kernel_ms", C.c_double), ("total_ms", C.c_double),
                ("kernel_launches", C.c_uint32), ("reserved", C.c_uint32),
                ("probe_ms", C.c_double), ("filter_reads", C.c_uint64)]


@dataclass
class Options:
    differences: int = 0
    indels: bool = False
    ignore_genes: bool = False
    ignore_counts: bool = False
    score: str = "product"
    nucleotides: bool = False
    n_v_genes: int = 1
    n_j_genes: int = 1
    device: int = -1
    existence: bool = False


@dataclass
class Stats:
    queries: int
    variants: int
    bloom_positive: int
    hash_equal: int
    matches: int
    algorithmic_bytes: int
    kernel_ms: float
    total_ms: float
    kernel_launches: int
    probe_ms: float = 0.0
    filter_reads: int = 0


def library_path() -> str:
    env = os.environ.get("COMPAIRR_HIP_LIB")
    if env:
        return env
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                        "libcompairr_hip.so")


_lib = None


def load_library() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    # torch (used by the tests and bench.py for device buffers, streams and
    # torch.distributed) bundles its own libamdhip64 with the same SONAME; two HIP
    # runtimes in one process do not both see the GPU.  Loading torch first
    # makes this library bind to the runtime torch already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            "%s is missing: build it with `make lib` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback." % path)
    lib = C.CDLL(path)
    lib.cmpr_abi_version.restype = C.c_int
    lib.cmpr_create.argtypes = [C.POINTER(_Options), C.POINTER(C.c_void_p)]
    lib.cmpr_destroy.argtypes = [C.c_void_p]
    lib.cmpr_destroy.restype = None
    lib.cmpr_last_error.argtypes = [C.c_void_p]
    lib.cmpr_last_error.restype = C.c_char_p
    lib.cmpr_set_reference.argtypes = [C.c_void_p, C.POINTER(_SetView), C.c_uint32]
    lib.cmpr_set_queries.argtypes = [C.c_void_p, C.POINTER(_SetView)]
    lib.cmpr_set_reference_device.argtypes = [C.c_void_p, C.POINTER(_SetView), C.c_uint32]
    lib.cmpr_set_queries_device.argtypes = [C.c_void_p, C.POINTER(_SetView)]
    lib.cmpr_route_queries.argtypes = [C.c_void_p, C.POINTER(_SetView), C.c_uint64, C.c_uint32,
                                       C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_void_p]
    lib.cmpr_route_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    lib.cmpr_set_queries_routed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64,
                                            C.c_void_p]
    lib.cmpr_overlap_matrix.argtypes = [C.c_void_p, C.c_void_p]
    lib.cmpr_overlap_matrix_f64.argtypes = [C.c_void_p, C.c_void_p]
    lib.cmpr_overlap_matrix_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.cmpr_get_stats.argtypes = [C.c_void_p, C.POINTER(_Stats)]
    lib.cmpr_get_kernel_times.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_double),
                                          C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    lib.cmpr_overlap_pairs.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                       C.POINTER(C.c_uint64)]
    lib.cmpr_count_duplicates.argtypes = [C.c_void_p, C.POINTER(_SetView), C.POINTER(C.c_uint64)]
    lib.cmpr_rows.argtypes = [C.c_void_p]
    lib.cmpr_rows.restype = C.c_uint32
    lib.cmpr_cols.argtypes = [C.c_void_p]
    lib.cmpr_cols.restype = C.c_uint32
    lib.cmpr_set_tunable.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    lib.cmpr_get_tunable.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]
    lib.cmpr_warm_up.argtypes = [C.POINTER(_Options)]
    lib.cmpr_warm_up_sized.argtypes = [C.POINTER(_Options), C.c_uint64, C.c_uint64, C.c_uint64]
    if lib.cmpr_abi_version() != 5:
        raise RuntimeError("libcompairr_hip.so ABI version mismatch")
    _lib = lib
    return lib


def _view(s: RepertoireSet) -> _SetView:
    v = _SetView()
    v.n = s.n
    v.residues = s.residues.ctypes.data
    v.offsets = s.offsets.ctypes.data
    v.v_gene = s.v_gene.ctypes.data
    v.j_gene = s.j_gene.ctypes.data
    v.repertoire = s.repertoire.ctypes.data
    v.count = s.count.ctypes.data
    v.n_repertoires = s.n_repertoires
    return v


class HipOverlap:
    """One context of the C ABI: create -> set_reference -> set_queries ->
    overlap_matrix*, mirroring the reference's overlap() around sim_thread
    (/root/reference/src/overlap.cc:840-938)."""

    def __init__(self, opt: Options):
        self._lib = load_library()
        o = _Options()
        o.differences = opt.differences
        o.indels = int(opt.indels)
        o.ignore_genes = int(opt.ignore_genes)
        o.ignore_counts = int(opt.ignore_counts)
        o.score = SCORES[opt.score.lower()]
        o.alphabet_size = 4 if opt.nucleotides else 20
        o.n_v_genes = opt.n_v_genes
        o.n_j_genes = opt.n_j_genes
        o.device = opt.device
        o.existence = int(opt.existence)
        self.opt = opt
        self._ctx = C.c_void_p()
        rc = self._lib.cmpr_create(C.byref(o), C.byref(self._ctx))
        if rc:
            raise HipError(rc, self._lib.cmpr_last_error(None).decode())

    def _check(self, rc: int) -> None:
        if rc:
            raise HipError(rc, self._lib.cmpr_last_error(self._ctx).decode())

    def close(self) -> None:
        if self._ctx:
            self._lib.cmpr_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_tunable(self, name: str, value: int) -> None:
        self._check(self._lib.cmpr_set_tunable(self._ctx, name.encode(), value))

    def get_tunable(self, name: str) -> int:
        v = C.c_int64()
        self._check(self._lib.cmpr_get_tunable(self._ctx, name.encode(), C.byref(v)))
        return v.value

    def layout(self) -> dict:
        return {k: self.get_tunable(k) for k in
                ("variant", "slices", "slice_words_log2", "class_residues",
                 "bloom_bits_log2_delta", "waves_per_block", "chunk_tiles",
                 "tiles", "chunks", "small_tiles", "query_slots", "class_anchor", "d2_pairs")}

    def set_reference(self, s: RepertoireSet, longest_query: int = 0) -> None:
        v = _view(s)
        self._check(self._lib.cmpr_set_reference(self._ctx, C.byref(v), longest_query))

    def set_queries(self, s: RepertoireSet) -> None:
        v = _view(s)
        self._check(self._lib.cmpr_set_queries(self._ctx, C.byref(v)))

    # ---- sets that already live in HBM (include/compairr_hip.h: cmpr_set_*_device) ----

    @staticmethod
    def device_view(s: RepertoireSet, device="cuda"):
        """(view of device pointers, the torch tensors that own the memory) of a set copied to
        the GPU with torch -- the stand-in for a caller whose sets live in HBM."""
        import torch
        keep = [torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(device)
                for a in (s.residues, s.offsets, s.v_gene, s.j_gene, s.repertoire, s.count)]
        v = _SetView()
        v.n = s.n
        (v.residues, v.offsets, v.v_gene, v.j_gene, v.repertoire, v.count) = [t.data_ptr() for t in keep]
        v.n_repertoires = s.n_repertoires
        torch.cuda.synchronize()
        return v, keep

    def set_reference_device(self, view: _SetView, longest_query: int = 0) -> None:
        self._check(self._lib.cmpr_set_reference_device(self._ctx, C.byref(view), longest_query))

    def set_queries_device(self, view: _SetView) -> None:
        self._check(self._lib.cmpr_set_queries_device(self._ctx, C.byref(view)))

    # ---- one query set over N contexts (cmpr_route_queries / _pack / cmpr_set_queries_routed) ----

    def route_queries(self, share: RepertoireSet, first_index: int, n_dest: int):
        """Key `share` (its first sequence is number `first_index` of the whole set); returns
        (records per destination, bytes per record, duplicate_count totals per repertoire)."""
        v = _view(share)
        counts = (C.c_uint64 * n_dest)()
        rb = C.c_uint32()
        tot = np.zeros(share.n_repertoires, dtype=np.float64)
        self._check(self._lib.cmpr_route_queries(self._ctx, C.byref(v), first_index, n_dest, counts,
                                                 C.byref(rb), tot.ctypes.data))
        return np.array(list(counts), dtype=np.int64), rb.value, tot

    def route_pack(self, d_send: int, capacity_bytes: int) -> None:
        self._check(self._lib.cmpr_route_pack(self._ctx, C.c_void_p(d_send), capacity_bytes))

    def set_queries_routed(self, d_records: int, n_records: int, n_repertoires: int, n_total: int,
                           rep_totals: Optional[np.ndarray] = None) -> None:
        t = None if rep_totals is None else np.ascontiguousarray(rep_totals, dtype=np.float64)
        self._check(self._lib.cmpr_set_queries_routed(
            self._ctx, C.c_void_p(d_records), n_records, n_repertoires, n_total,
            None if t is None else t.ctypes.data))

    @property
    def shape(self):
        return (self._lib.cmpr_rows(self._ctx), self._lib.cmpr_cols(self._ctx))

    def overlap_matrix(self) -> np.ndarray:
        out = np.zeros(self.shape, dtype=np.uint64)
        self._check(self._lib.cmpr_overlap_matrix(self._ctx, out.ctypes.data))
        return out

    def overlap_matrix_f64(self) -> np.ndarray:
        out = np.zeros(self.shape, dtype=np.float64)
        self._check(self._lib.cmpr_overlap_matrix_f64(self._ctx, out.ctypes.data))
        return out

    def overlap_matrix_device(self, d_matrix: int, stream: Optional[int] = None) -> None:
        self._check(self._lib.cmpr_overlap_matrix_device(
            self._ctx, C.c_void_p(d_matrix), C.c_void_p(stream or 0)))

    def overlap_pairs(self) -> np.ndarray:
        """(query index, hit index) of every matching pair, sorted."""
        n = C.c_uint64()
        self._check(self._lib.cmpr_overlap_pairs(self._ctx, 0, None, None, C.byref(n)))
        q = np.zeros(n.value, dtype=np.uint32)
        h = np.zeros(n.value, dtype=np.uint32)
        self._check(self._lib.cmpr_overlap_pairs(self._ctx, n.value, q.ctypes.data,
                                                 h.ctypes.data, C.byref(n)))
        assert n.value == len(q)
        out = np.stack([q, h], axis=1)
        return out[np.lexsort((out[:, 1], out[:, 0]))]

    def count_duplicates(self, s: Optional[RepertoireSet] = None) -> int:
        """Exact duplicates inside `s` (None: the resident reference set)."""
        out = C.c_uint64()
        v = _view(s) if s is not None else None
        self._check(self._lib.cmpr_count_duplicates(
            self._ctx, C.byref(v) if v is not None else None, C.byref(out)))
        return out.value

    def kernel_times(self, max_calls: int = 64):
        """(kernel_ms[], probe_ms[]) of the last calls, oldest first (HIP events)."""
        n = min(max_calls, 63)
        k = (C.c_double * n)()
        p = (C.c_double * n)()
        got = C.c_uint32(0)
        self._check(self._lib.cmpr_get_kernel_times(self._ctx, n, k, p, C.byref(got)))
        return list(k[:got.value]), list(p[:got.value])

    def stats(self) -> Stats:
        st = _Stats()
        self._check(self._lib.cmpr_get_stats(self._ctx, C.byref(st)))
        return Stats(st.queries, st.variants, st.bloom_positive, st.hash_equal,
                     st.matches, st.algorithmic_bytes, st.kernel_ms, st.total_ms,
                     st.kernel_launches, st.probe_ms, st.filter_reads)


def overlap(set1: RepertoireSet, set2: RepertoireSet, opt: Options):
    """Convenience: the whole path once; returns (cells as float64 in the
    reference's units, Stats)."""
    with HipOverlap(opt) as h:
        h.set_reference(set2, set1.longest)
        h.set_queries(set1)
        m = h.overlap_matrix_f64()
        return m, h.stats()
