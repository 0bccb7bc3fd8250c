#!/usr/bin/env python
"""Generates tests/golden/golden_v7.npz: Estimator::PrinceDormand with control_stepsize = true (src/princedormand.cpp:26-60)
as EXTRACTED (oracle/_ref/libxivo_refx_n203_*.so, oracle/ref/extract_reference.py) - a chain of Estimator::Propagate calls of
different lengths, the state and covariance after every call. The extracted function keeps its configuration AND its current
step `h` in function-local statics (:12-13): the chain runs in a PRIVATE copy of the library (fresh statics) with
refx_pd_control(1, ...) called first. Inputs come from make_golden_v6.prop_case (never stored twice).
Run in the authoring container only:  python tests/golden/make_golden_v7.py"""
import ctypes as C
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

PD_CTL = dict(tolerance=1e-3, attempts=12, min_scale_factor=0.125, max_scale_factor=4.0)
# 2.5 ms: h = 2 ms, then 4 h clipped to the 0.5 ms that are left; 7 ms: the carried step (2 ms after the clip rule) grows 4 x and is
# clipped / halved towards the end of the sample (:53-58); 12 ms; 1 ms: shorter than the carried step (h = min(h, dt), :34)
CHAIN_NS = (2500000, 7000000, 2500000, 12000000, 1000000)


def private_refx(N=203):
    """A RefX on a private copy of the extracted library: its function-local statics are this object's alone."""
    import ref_binding
    src = ref_binding.loadx(N).path if hasattr(ref_binding.loadx(N), "path") else None
    if src is None:
        for v in ("v4", "v3"):
            p = os.path.join(ROOT, "oracle", "_ref", f"libxivo_refx_n{N}_{v}.so")
            if os.path.exists(p) and (v == "v3" or ref_binding._has_avx512()):
                src = p
                break
    d = tempfile.mkdtemp(prefix="refx_private_")
    dst = os.path.join(d, "libxivo_refx_private_%d.so" % os.getpid())
    shutil.copy(src, dst)
    return ref_binding.RefX(dst)


def run_chain(rx, seed, stepsize=0.002):
    import make_golden_v6 as v6
    import xivo_oracle as orc
    c = v6.prop_case(seed)
    rx.lib.refx_pd_control.restype = None
    rx.lib.refx_pd_control(C.c_int(1), C.c_double(PD_CTL["tolerance"]), C.c_int(PD_CTL["attempts"]), C.c_double(PD_CTL["min_scale_factor"]),
                           C.c_double(PD_CTL["max_scale_factor"]))
    X, P = c["X"], c["P"]
    outs = []
    for dt_ns in CHAIN_NS:
        o = rx.propagate("PrinceDormand", X, P, c["gy"], c["ac"], c["sg"], c["sa"], dt_ns, c["Qimu"], c["Qmodel"], c["g"], stepsize=stepsize)
        X = orc.MotionState(o["Rsb"], o["Tsb"], o["Vsb"], X.bg.copy(), X.ba.copy(), X.Rsg.copy())
        P = o["P"]
        outs.append(o)
    return c, outs


if __name__ == "__main__":
    out = {}
    # (the reference prints err / h / s of every controlled step to stdout, :49)
    for seed in (1, 2):
        rx = private_refx(203) if seed == 1 else None
        if rx is None:                       # a second chain needs statics of its own: another private copy
            rx = private_refx(203)
        c, outs = run_chain(rx, seed)
        for i, o in enumerate(outs):
            k = f"pdc_s{seed}_{i}"
            out[k + "_Rsb"] = o["Rsb"]; out[k + "_Tsb"] = o["Tsb"]; out[k + "_Vsb"] = o["Vsb"]
            out[k + "_Pmm"] = o["P"][:23, :23]; out[k + "_Pms_w"] = o["P"][:23, 23:] @ c["w"]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "golden_v7.npz"), **out)
    print("wrote golden_v7.npz with", len(out), "arrays")
