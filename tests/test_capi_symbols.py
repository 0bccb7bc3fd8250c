"""CPU-only: the C-ABI library builds, loads, and exports every symbol the header declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol(built):
    from xivo_amd.lib import ALL_SYMBOLS, lib_path, load_library
    hdr = open(os.path.join(ROOT, "include", "xivo_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(xivo_hip_[A-Za-z0-9_]+)\s*\(", hdr, )))
    assert declared, "no declarations found"
    lib = ctypes.CDLL(lib_path())
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/xivo_hip.h but not exported"
    assert sorted(ALL_SYMBOLS) == declared, (set(ALL_SYMBOLS) ^ set(declared))
    load_library()


def test_strerror_and_tile_query_need_no_gpu(built):
    from xivo_amd.lib import load_library
    lib = load_library()
    assert lib.xivo_hip_strerror(0) == b"ok"
    assert b"positive definite" in lib.xivo_hip_strerror(-3)
    bm, bn = ctypes.c_int(), ctypes.c_int()
    lib.xivo_hip_gemm_tile(250, 250, 0, ctypes.byref(bm), ctypes.byref(bn))
    assert (bm.value, bn.value) == (128, 128)
    lib.xivo_hip_gemm_tile(160, 250, 0, ctypes.byref(bm), ctypes.byref(bn))
    assert (bm.value, bn.value) == (160, 128)
    lib.xivo_hip_gemm_tile(160, 160, 1, ctypes.byref(bm), ctypes.byref(bn))
    assert (bm.value, bn.value) == (128, 128)


def test_no_product_code_touches_the_oracle():
    """The product path must never import / link the oracle or a CPU fallback."""
    bad = []
    for base in ("xivo_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"xivo_oracle|ref_binding|oracle/", txt) and "no reference to the oracle" not in txt:
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad
