#!/usr/bin/env python3
"""Where a fresh process spends its start-up: dlopen of the library, cmpr_warm_up, the first context and index."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t0 = time.perf_counter()
lib = C.CDLL(os.path.join(ROOT, "compairr_amd", "lib", "libcompairr_hip.so"))
t1 = time.perf_counter()


class Opt(C.Structure):
    _fields_ = [("differences", C.c_int32), ("indels", C.c_int32), ("ignore_genes", C.c_int32), ("ignore_counts", C.c_int32),
                ("score", C.c_int32), ("alphabet_size", C.c_int32), ("n_v_genes", C.c_uint32), ("n_j_genes", C.c_uint32),
                ("device", C.c_int32), ("existence", C.c_int32), ("reserved", C.c_int32 * 6)]


o = Opt(differences=1, alphabet_size=20, n_v_genes=60, n_j_genes=13, device=0)
lib.cmpr_warm_up.argtypes = [C.POINTER(Opt)]
rc = lib.cmpr_warm_up(C.byref(o))
t2 = time.perf_counter()
ctx = C.c_void_p()
lib.cmpr_create.argtypes = [C.POINTER(Opt), C.POINTER(C.c_void_p)]
rc2 = lib.cmpr_create(C.byref(o), C.byref(ctx))
t3 = time.perf_counter()
print("dlopen %.3f s, cmpr_warm_up %.3f s (rc %d), cmpr_create %.3f s (rc %d)" % (t1 - t0, t2 - t1, rc, t3 - t2, rc2))
