#!/usr/bin/env python
"""Generates tests/golden/golden_v3.npz from the EXTRACTED reference build (round 4): oracle/_ref/libxivo_refx_n203_*.so is
the reference's own text of the member functions (oracle/ref/extract_reference.py cuts them out of /root/reference/src at
build time, oracle/ref/xivo_refx.cpp compiles them) - these vectors are therefore outputs of the reference itself run here:
  * Estimator::MHGating (src/update.cpp:50-116): inlier list, statuses, num_mh_rejected_ for three scenes (regular,
    three outliers, forced relaxation of the threshold);
  * Estimator::FilterUpdate (src/update.cpp:120-153): stacked H (FillJacobianBlock incl. the :675-676 overwrite), err_
    before the absorb, P+, the absorbed state and features;
  * Estimator::RK4Step / PrinceDormandStep (src/rk4.cpp:35-103, src/princedormand.cpp:85-221): state and the motion rows of P.
Inputs are regenerated from seeds (sha256 of their bytes stored, so a drifting generator is noticed).
Run in the authoring container only:  python tests/golden/make_golden_v3.py"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_binding  # noqa: E402
import xivo_oracle as orc  # noqa: E402
from scene_util import scene_arrays, oracle_jacobians, spd  # noqa: E402
from xivo_amd import synth  # noqa: E402

N, NG, NF = 203, 15, 30


def digest(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8).copy()


def gating_scene(seed):
    """inputs of the MHGating cases (shared with tests/test_oracle_pinned.py)"""
    cam = synth.PINHOLE
    lay = orc.Layout(NG, NF)
    sc = synth.g_level(NG, NF, NF, 1, seed=400 + seed, cam=cam)
    _, _, _, xp = scene_arrays(sc, cam)
    if seed == 1:
        xp[0, [2, 7, 11]] += 40.0
    if seed == 2:
        xp[0, 3:] += np.linspace(6, 60, NF - 3)[:, None]
    P = spd(N, 50 + seed) * 1e-4
    Js, inns, _ = oracle_jacobians(sc, cam, lay, xp, 0)
    status = np.full(NF, 3, dtype=np.int32); status[0] = 7
    return Js, inns, P, status


def filter_update_scene():
    cam = synth.RADTAN
    lay = orc.Layout(NG, NF)
    sc = synth.g_level(NG, NF, NF, 1, seed=77, cam=cam)
    _, _, _, xp = scene_arrays(sc, cam)
    P = spd(N, 9) * 1e-4
    Js, inns, _ = oracle_jacobians(sc, cam, lay, xp, 0)
    X = orc.MotionState(sc["Rsb"][0], sc["Tsb"][0], [0.1, -0.2, 0.05], [0.01, 0.0, -0.01], [0.02, 0.01, 0.0], orc.so3_exp([0.01, -0.02, 0.0]))
    return sc, lay, Js, inns, P, X


def step_inputs(k):
    rng = np.random.default_rng(190 + k)
    A = rng.uniform(-1, 1, size=(N, N)); P = A @ A.T / N * 1e-3 + 1e-6 * np.eye(N)
    X = orc.MotionState(orc.so3_exp(rng.normal(size=3) * 0.4), rng.normal(size=3), rng.normal(size=3),
                        rng.normal(size=3) * 0.01, rng.normal(size=3) * 0.05, orc.so3_exp([0.02, -0.01, 0.0]))
    gy, ac = rng.normal(size=3) * (1 + 3 * k), np.array([0.2, -0.1, 9.7]) + rng.normal(size=3)
    sg, sa = rng.normal(size=3) * 5, rng.normal(size=3)
    Qi = np.diag(rng.uniform(1e-6, 1e-3, 12)); gv = np.array([0.0, 0.0, -9.8]); dt = 0.001 * (1 + k)
    return X, P, gy, ac, sg, sa, dt, Qi, gv


def main():
    x = ref_binding.loadx(N)
    out = {"built_from": np.array(open(os.path.join(ROOT, "oracle", "_ref", "refx_MANIFEST.txt")).read())}
    for seed in range(3):
        Js, inns, P, status = gating_scene(seed)
        idx, st_after, nrej, ndes = x.mh_gating(Js, inns, P, 2.25, 5.991, 1.1, 5, status)
        out[f"gate_{seed}_in_sha"] = digest(Js, inns, P)
        out[f"gate_{seed}_inliers"] = idx; out[f"gate_{seed}_status"] = st_after
        out[f"gate_{seed}_nrej"] = np.array([nrej, ndes])
    sc, lay, Js, inns, P, X = filter_update_scene()
    H, err, Pn, Rsb, Tsb, Vsb, bg, ba, Rsg, xs = x.filter_update(Js, inns, sc["ref"][0], sc["sind"][0], 2.25, P, X, sc["Rbc"][0], sc["Tbc"][0], sc["x"][0])
    out["fu_in_sha"] = digest(Js, inns, P, X.Rsb, sc["x"][0])
    nz = np.nonzero(H)
    out["fu_H_nz_rows"], out["fu_H_nz_cols"], out["fu_H_nz_vals"] = nz[0].astype(np.int32), nz[1].astype(np.int32), H[nz]
    out["fu_err"], out["fu_P"] = err, Pn.astype(np.float64)
    out["fu_state"] = np.concatenate([Rsb.reshape(-1), Tsb, Vsb, bg, ba, Rsg.reshape(-1)]); out["fu_x"] = xs
    for k in range(2):
        X, P, gy, ac, sg, sa, dt, Qi, gv = step_inputs(k)
        out[f"step_{k}_in_sha"] = digest(P, X.Rsb, X.Tsb, gy, ac, sg, sa, Qi)
        for method in ("RK4", "PD"):
            R1, T1, V1, P1 = x.integrator_step(method, X, P, gy, ac, sg, sa, dt, Qi, gv)
            assert np.array_equal(P1[23:, 23:], P[23:, 23:])     # the structure block is not touched
            out[f"step_{k}_{method}_state"] = np.concatenate([R1.reshape(-1), T1, V1])
            out[f"step_{k}_{method}_Prows"] = P1[:23, :].copy()
    path = os.path.join(ROOT, "tests", "golden", "golden_v3.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
