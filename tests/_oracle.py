"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module; the product never does."""

from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCORES = {"product": 0, "ratio": 1, "min": 2, "max": 3, "mean": 4, "mh": 5, "jaccard": 6}


class _Set(C.Structure):
    _fields_ = [("n", C.c_uint64), ("residues", C.c_void_p), ("offsets", C.c_void_p),
                ("v_gene", C.c_void_p), ("j_gene", C.c_void_p),
                ("repertoire", C.c_void_p), ("count", C.c_void_p),
                ("n_repertoires", C.c_uint32)]


class _Opts(C.Structure):
    _fields_ = [("differences", C.c_int32), ("indels", C.c_int32),
                ("ignore_genes", C.c_int32), ("ignore_counts", C.c_int32),
                ("score", C.c_int32), ("alphabet_size", C.c_int32),
                ("threads", C.c_int32), ("n_v_genes", C.c_uint32),
                ("n_j_genes", C.c_uint32), ("existence", C.c_int32)]


class _Stats(C.Structure):
    _fields_ = [("variants", C.c_uint64), ("bloom_positive", C.c_uint64),
                ("slots_visited", C.c_uint64), ("hash_equal", C.c_uint64),
                ("matches", C.c_uint64), ("dup_set1", C.c_uint64),
                ("dup_set2", C.c_uint64), ("seconds_index", C.c_double),
                ("seconds_analysis", C.c_double)]


@dataclass
class OracleStats:
    variants: int
    bloom_positive: int
    slots_visited: int
    hash_equal: int
    matches: int
    dup_set1: int
    dup_set2: int
    seconds_index: float
    seconds_analysis: float


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing: run `make oracle`")
        _lib = C.CDLL(path)
        for f in (_lib.oracle_overlap, _lib.oracle_bruteforce):
            f.restype = C.c_int
    return _lib


def _set(s) -> _Set:
    v = _Set()
    v.n = s.n
    v.residues = s.residues.ctypes.data
    v.offsets = s.offsets.ctypes.data
    v.v_gene = s.v_gene.ctypes.data
    v.j_gene = s.j_gene.ctypes.data
    v.repertoire = s.repertoire.ctypes.data
    v.count = s.count.ctypes.data
    v.n_repertoires = s.n_repertoires
    return v


def _opts(opt, threads, n_v, n_j) -> _Opts:
    o = _Opts()
    o.differences = opt.differences
    o.indels = int(opt.indels)
    o.ignore_genes = int(opt.ignore_genes)
    o.ignore_counts = int(opt.ignore_counts)
    o.score = SCORES[opt.score.lower()]
    o.alphabet_size = 4 if opt.nucleotides else 20
    o.threads = threads
    o.n_v_genes = n_v
    o.n_j_genes = n_j
    o.existence = int(getattr(opt, "existence", False))
    return o


def overlap(set1, set2, opt, threads: int = 1):
    """The oracle's matrix (float64, reference cell units) and its statistics.
    `opt` is a compairr_amd.Options."""
    o = _opts(opt, threads, opt.n_v_genes, opt.n_j_genes)
    a = _set(set1)
    b = a if set2 is set1 else _set(set2)
    rows = set1.n if getattr(opt, "existence", False) else set1.n_repertoires
    m = np.zeros((rows, set2.n_repertoires), dtype=np.float64)
    st = _Stats()
    rc = lib().oracle_overlap(C.byref(o), C.byref(a), C.byref(b),
                              C.c_void_p(m.ctypes.data), C.byref(st))
    if rc:
        raise RuntimeError("oracle_overlap failed (%d)" % rc)
    return m, OracleStats(*(getattr(st, f[0]) for f in _Stats._fields_))


def bruteforce(set1, set2, opt):
    o = _opts(opt, 1, opt.n_v_genes, opt.n_j_genes)
    a = _set(set1)
    b = a if set2 is set1 else _set(set2)
    rows = set1.n if getattr(opt, "existence", False) else set1.n_repertoires
    m = np.zeros((rows, set2.n_repertoires), dtype=np.float64)
    rc = lib().oracle_bruteforce(C.byref(o), C.byref(a), C.byref(b),
                                 C.c_void_p(m.ctypes.data))
    if rc:
        raise RuntimeError("oracle_bruteforce failed (%d)" % rc)
    return m


def pairs(set1, set2, opt):
    """Sorted array of (seed, hit) pairs the reference loop finds (-p/--pairs)."""
    o = _opts(opt, 1, opt.n_v_genes, opt.n_j_genes)
    a = _set(set1)
    b = a if set2 is set1 else _set(set2)
    count = C.c_uint64()
    f = lib().oracle_pairs
    f.restype = C.c_int
    rc = f(C.byref(o), C.byref(a), C.byref(b), C.c_uint64(0), None, None, C.byref(count))
    if rc:
        raise RuntimeError("oracle_pairs failed")
    q = np.zeros(count.value, dtype=np.uint32)
    h = np.zeros(count.value, dtype=np.uint32)
    rc = f(C.byref(o), C.byref(a), C.byref(b), C.c_uint64(count.value),
           C.c_void_p(q.ctypes.data), C.c_void_p(h.ctypes.data), C.byref(count))
    if rc:
        raise RuntimeError("oracle_pairs failed")
    out = np.stack([q, h], axis=1)
    return out[np.lexsort((out[:, 1], out[:, 0]))]


def integer_cells(m: np.ndarray, opt) -> np.ndarray:
    """Oracle cells (doubles) -> the exact integer sums the C ABI returns
    (mean is accumulated as a + b, i.e. twice the reference's cell)."""
    if opt.score.lower() == "mean" and not opt.ignore_counts:
        m = m * 2
    r = np.rint(m)
    assert np.array_equal(r, m), "non-integer oracle cell"
    return r.astype(np.uint64)
