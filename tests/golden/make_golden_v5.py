#!/usr/bin/env python
"""Generates tests/golden/golden_v5.npz from the EXTRACTED reference build compiled with the reference's three
online-calibration defines (oracle/_ref/libxivo_refx_calib_*.so): the motion side of those builds -
Estimator::RK4Step / PrinceDormandStep with ComposeMotion on imu_.Cg() / imu_.Ca() and the dWsb/dCg, dVsb/dCa columns of
ComputeMotionJacobianAt (the text of src/rk4.cpp:35-103, src/princedormand.cpp:85-221, src/estimator.cpp:598-704), and
Estimator::AbsorbError on td / Ca / Cg / the camera intrinsics (src/estimator.cpp:875-890 with IMUState::operator+=,
src/imu.cpp:7-21, as extracted). Stored: the nominal state after the step, the 39 motion rows of P, a weighted column sum
of the rows below them, and the absorbed calibration state.
Run in the authoring container only:  python tests/golden/make_golden_v5.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ref_binding  # noqa: E402
import xivo_oracle as orc  # noqa: E402

N, NM = 228, 39
CAM = dict(model=2, rows=480, cols=640, fx=500.0, fy=510.0, cx=320.0, cy=240.0, d=[0.01, -0.02, 0.1, 0.05, -0.01])


def case(seed):
    rng = np.random.default_rng(seed)
    A = rng.uniform(-1, 1, (N, N)); P = A @ A.T / N * 1e-3 + 1e-6 * np.eye(N)
    X = orc.MotionState(orc.so3_exp(rng.normal(size=3) * 0.3), rng.normal(size=3), rng.normal(size=3), rng.normal(size=3) * 0.01,
                        rng.normal(size=3) * 0.05, orc.so3_exp(np.array([0.02, -0.01, 0.0])))
    Cg = np.eye(3) + 0.02 * rng.normal(size=(3, 3)); Ca = np.triu(np.eye(3) + 0.02 * rng.normal(size=(3, 3)))
    gy = rng.normal(size=3) * 0.5; ac = rng.normal(size=3) * 2 + np.array([0, 0, 9.8]); sg = rng.normal(size=3) * 5; sa = rng.normal(size=3) * 20
    Qimu = np.diag(rng.uniform(1e-5, 1e-3, 12)); g = np.array([0, 0, -9.8])
    err = rng.normal(size=N) * 1e-2
    Rbc = orc.so3_exp(rng.normal(size=3) * 0.1); Tbc = rng.normal(size=3) * 0.1
    w = rng.normal(size=N - NM)
    return dict(P=P, X=X, Cg=Cg, Ca=Ca, gy=gy, ac=ac, sg=sg, sa=sa, Qimu=Qimu, g=g, err=err, Rbc=Rbc, Tbc=Tbc, w=w, td=0.013)


def main():
    x = ref_binding.loadx("calib")
    assert x.N == N and x.calib_slots()[4] == NM
    out = {}
    for seed in (1, 2):
        c = case(seed)
        for method in ("RK4", "PrinceDormand"):
            R, T, V, Pn = x.integrator_step(method, c["X"], c["P"], c["gy"], c["ac"], c["sg"], c["sa"], 0.004, c["Qimu"], c["g"], c["Cg"], c["Ca"])
            k = f"s{seed}_{method}"
            out[k + "_Rsb"] = R; out[k + "_Tsb"] = T; out[k + "_Vsb"] = V
            out[k + "_Ptop"] = Pn[:NM, :]; out[k + "_Pleft_w"] = c["w"] @ Pn[NM:, :NM]
            assert np.array_equal(Pn[NM:, NM:], c["P"][NM:, NM:])       # P_ss is not touched (rk4.cpp:95-102)
        o = x.absorb_motion_calib(c["X"], c["Rbc"], c["Tbc"], c["td"], c["Cg"], c["Ca"], CAM, c["err"])
        k = f"s{seed}_absorb"
        for name in ("Rsb", "Tsb", "Vsb", "bg", "ba", "Rsg", "Rbc", "Tbc", "Cg", "Ca"):
            out[f"{k}_{name}"] = o[name]
        out[k + "_td"] = np.array([o["td"]])
        out[k + "_intr"] = np.array([o["cam"]["fx"], o["cam"]["fy"], o["cam"]["cx"], o["cam"]["cy"]] + list(o["cam"]["d"]))
    path = os.path.join(ROOT, "tests", "golden", "golden_v5.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
