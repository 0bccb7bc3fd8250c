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
    for base in ("xivo_amd", "include", "scripts"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"xivo_oracle|ref_binding|oracle/", txt) and "no reference to the oracle" not in txt:
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_create_without_gpu_fails_loudly_not_fatally(built):
    """No CPU fallback: on a box without a GPU the product reports an error status
    (and the Python binding raises) instead of computing something else."""
    import subprocess, sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import ctypes as C\n"
        "from xivo_amd.lib import load_library, Context, XivoHipError\n"
        "lib = load_library()\n"
        "h = C.c_void_p()\n"
        "rc = lib.xivo_hip_create(C.byref(h), 0, 64, 16, 2, 0)\n"
        "if rc == 0:\n"
        "    lib.xivo_hip_destroy(h); print('HAS_GPU')\n"
        "else:\n"
        "    assert rc < 0 and h.value is None\n"
        "    try:\n"
        "        Context(64, 16, 2); print('NO_RAISE')\n"
        "    except XivoHipError as e:\n"
        "        print('RAISED', e.status)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "HAS_GPU" in out.stdout or "RAISED" in out.stdout


def test_invalid_arguments_return_status_codes(built):
    from xivo_amd.lib import load_library
    lib = load_library()
    h = ctypes.c_void_p()
    assert lib.xivo_hip_create(None, 0, 64, 16, 2, 0) == -1
    assert lib.xivo_hip_create(ctypes.byref(h), 0, 0, 16, 2, 0) == -1
    assert lib.xivo_hip_create(ctypes.byref(h), 0, 64, 16 * 30, 2, 0) == -5      # M beyond the solver's register budget
    assert lib.xivo_hip_sync(None) == -1 and lib.xivo_hip_update_joseph(None, 1) == -1
    lib.xivo_hip_destroy(None)   # no-op


def test_host_library_exports_the_batch_estimator(built):
    """libxivo_host.so (C++ side: adapter + xivo::hip::BatchEstimator) loads next to the C ABI and exports the entry
    points xivo_amd/batch.py binds (no compute call without a GPU)."""
    from xivo_amd import batch
    host = batch.load_host_library()
    for name in ("xivo_batch_create", "xivo_batch_destroy", "xivo_batch_imu", "xivo_batch_visual", "xivo_batch_poses",
                 "xivo_batch_book", "xivo_batch_stats", "xivo_batch_ctx", "xivo_host_selftest_update_step"):
        assert hasattr(host, name), name
    # the numpy mirror of struct xivo_batch_cfg (host/batch_estimator.cpp) has the size the C++ side compiled
    assert batch.batch_cfg_dtype.itemsize == host.xivo_batch_cfg_size() == 5656      # (round 6: + the 32 bytes of the control_stepsize fields of xivo_prop_opts)


def test_candidate_comparison_order_matches_the_reference_as_coded():
    """Criteria::CandidateComparison (src/options.cpp:34-61): status first, then Feature::score() = -P(2,2); the score of
    `comparison_score_type` is computed but never used. Host-only entry point: runs without a GPU."""
    import functools
    import numpy as np
    import xivo_oracle as orc
    from xivo_amd.lib import candidate_order, subfilter_dtype
    rng = np.random.default_rng(0)
    nb, n = 3, 17
    f = np.zeros((nb, n), dtype=subfilter_dtype)
    for b in range(nb):
        for i in range(n):
            A = rng.normal(size=(3, 3)); P = A @ A.T * 1e-2
            f[b, i]["P"] = P.T.reshape(-1)
            f[b, i]["status"] = rng.integers(0, 2)
            f[b, i]["outlier_counter"] = rng.choice([0.0, 0.005, 0.5])
            ready = f[b, i]["status"] == orc.FEAT_READY
            ok = rng.random() < 0.8
            f[b, i]["candidate"] = (1 if ok else 0) | (2 if (ok and ready) else 0)
    for strict in (False, True):
        order, cnt, score = candidate_order(f, strict=strict, score_type=2)
        for b in range(nb):
            passing = [i for i in range(n) if f[b, i]["candidate"] & (2 if strict else 1)]
            Pm = lambda i: f[b, i]["P"].reshape(3, 3).T
            cmp = lambda a, c: -1 if orc.candidate_before(f[b, a]["status"], Pm(a), f[b, c]["status"], Pm(c)) else (
                1 if orc.candidate_before(f[b, c]["status"], Pm(c), f[b, a]["status"], Pm(a)) else 0)
            exp = sorted(passing, key=functools.cmp_to_key(cmp))          # stable, like the entry point
            assert cnt[b] == len(passing) and order[b, :cnt[b]].tolist() == exp and (order[b, cnt[b]:] == -1).all()
            for i in range(n):
                assert abs(score[b, i] - orc.candidate_scores(Pm(i), f[b, i]["outlier_counter"])[2]) < 1e-15
    s0 = candidate_order(f, score_type=0)[2]; s1 = candidate_order(f, score_type=1)[2]
    assert abs(s0[1, 3] - orc.candidate_scores(f[1, 3]["P"].reshape(3, 3).T, 0)[0]) < 1e-15
    assert abs(s1[1, 3] - orc.candidate_scores(f[1, 3]["P"].reshape(3, 3).T, 0)[1]) < 1e-15
