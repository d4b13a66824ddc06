"""The C-ABI library loads and exports every entry point include/*.h declares
(no compute calls: there is no GPU here); the product fails loudly, never
falls back, when no HIP device exists."""

import ctypes
import os
import re
import subprocess

import pytest

from compairr_amd import hip
from conftest import GOLDEN_INPUTS, ROOT, has_gpu


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "compairr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cmpr_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(hip.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(hip.library_path())
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.cmpr_abi_version() == 5


def test_header_is_plain_c_and_a_c_program_links_the_library(tmp_path):
    """The boundary is a C ABI: the header compiles as C11 (-pedantic), and a C translation unit that names every
    entry point links against the shared library (sizes agree with the ctypes mirror)."""
    calls = "\n".join("  p[%d] = (void *)%s;" % (i, n) for i, n in enumerate(declared_symbols()))
    src = tmp_path / "abi.c"
    src.write_text('#include "compairr_hip.h"\n#include <stdio.h>\nint main(void) {\n  void *p[64];\n%s\n'
                   '  printf("%%d %%zu %%zu %%zu %%d\\n", CMPR_ABI_VERSION, sizeof(cmpr_options), sizeof(cmpr_set_view), '
                   'sizeof(cmpr_stats), p[0] != 0);\n  return 0;\n}\n' % calls)
    lib = hip.library_path()
    exe = tmp_path / "abi"
    p = subprocess.run(["gcc", "-std=c11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-Wno-pedantic",
                        "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), lib,
                        "-Wl,-rpath," + os.path.dirname(lib)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    # (the header alone, strictly: no extension a C compiler would have to forgive)
    q = subprocess.run(["gcc", "-std=c11", "-pedantic-errors", "-Wall", "-Wextra", "-fsyntax-only", "-x", "c",
                        os.path.join(ROOT, "include", "compairr_hip.h")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert q.returncode == 0, q.stderr.decode()[-2000:]
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout.decode().split() == ["5", "64", "64", "88", "1"]


def test_struct_sizes_match_header():
    # cmpr_options: 9 x 4 + 7 x 4; cmpr_set_view: 8 + 6 x 8 + 2 x 4; cmpr_stats
    assert ctypes.sizeof(hip._Options) == 64
    assert ctypes.sizeof(hip._SetView) == 64
    assert ctypes.sizeof(hip._Stats) == 88


def test_invalid_options_rejected_before_any_device_work():
    lib = hip.load_library()
    o = hip._Options()
    o.alphabet_size = 7
    ctx = ctypes.c_void_p()
    assert lib.cmpr_create(ctypes.byref(o), ctypes.byref(ctx)) == 1     # CMPR_EINVAL
    assert b"alphabet_size" in lib.cmpr_last_error(None)
    o.alphabet_size = 20
    o.differences = 1
    o.indels = 1
    o.score = 5                                                          # MH with d > 0
    assert lib.cmpr_create(ctypes.byref(o), ctypes.byref(ctx)) == 1
    o.score = 0
    o.differences = 3                                                    # d > 2: unsupported
    o.indels = 0
    assert lib.cmpr_create(ctypes.byref(o), ctypes.byref(ctx)) == 4


@pytest.mark.skipif(has_gpu(), reason="checks the no-device failure mode")
def test_no_silent_cpu_fallback_library():
    with pytest.raises(hip.HipError) as e:
        hip.HipOverlap(hip.Options(differences=1))
    assert e.value.code == 3                                             # CMPR_EDEVICE


@pytest.mark.skipif(has_gpu(), reason="checks the no-device failure mode")
def test_no_silent_cpu_fallback_cli():
    p = subprocess.run([os.path.join(ROOT, "bin", "compairr"), "-m", "seta.tsv", "setb.tsv",
                        "-d", "1", "-l", os.devnull], cwd=GOLDEN_INPUTS,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and p.stdout == b""
    assert b"Error:" in p.stderr and b"HIP" in p.stderr


def test_missing_library_is_fatal(tmp_path):
    env = dict(os.environ, COMPAIRR_HIP_LIB=str(tmp_path / "nope.so"))
    exe = tmp_path / "compairr"
    exe.write_bytes(open(os.path.join(ROOT, "bin", "compairr"), "rb").read())
    exe.chmod(0o755)
    p = subprocess.run([str(exe), "-m", "seta.tsv"], cwd=GOLDEN_INPUTS, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"no CPU fallback" in p.stderr
