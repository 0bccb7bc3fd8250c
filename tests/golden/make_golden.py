#!/usr/bin/env python
"""Generates tests/golden/golden_v1.npz from oracle/_ref (the reference's own
Eigen/Sophus/helpers.cpp/camera arithmetic, built from /root/reference where it
lies). Run in the authoring container only:  python tests/golden/make_golden.py
The fixture stores INPUTS and OUTPUTS, so the tests do not depend on the
generators staying unchanged."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ref_binding  # noqa: E402
import xivo_oracle as orc  # noqa: E402
from xivo_amd import synth  # noqa: E402


def main():
    ref = ref_binding.load()
    g = {}
    # --- a1 UpdateJosephForm at three small sizes
    for tag, (N, F) in {"a": (37, 3), "b": (64, 8), "c": (113, 20)}.items():
        P, H, inn, dR = synth.s_level(N, F, 1, seed={"a": 1, "b": 2, "c": 3}[tag])
        err, Pn = ref.update_joseph(H[0], P[0], inn[0], dR[0])
        g[f"uj_{tag}_P"], g[f"uj_{tag}_H"], g[f"uj_{tag}_inn"], g[f"uj_{tag}_dR"] = P[0], H[0], inn[0], dR[0]
        g[f"uj_{tag}_err"], g[f"uj_{tag}_Pn"] = err, Pn
    # --- cameras
    rng = np.random.default_rng(7)
    cams = {"pinhole": synth.PINHOLE, "equi": synth.EQUI, "radtan": synth.RADTAN, "atan": synth.ATAN}
    xc = rng.uniform(-0.5, 0.5, size=(6, 2))
    xc[5] = [1e-6, -2e-6]  # ATAN singular branch (R < 1e-4)
    g["cam_xc"] = xc
    for name, cam in cams.items():
        out = [ref.camera_project(cam, x) for x in xc]
        g[f"cam_{name}_xp"] = np.array([o[0] for o in out])
        g[f"cam_{name}_J"] = np.array([o[1] for o in out])
    # --- a4 ComputeJacobian + a3 FillJacobianBlock + a6 MH distances, per camera model
    lay = orc.Layout(4, 10)
    g["lay"] = np.array([lay.N, lay.group_begin, lay.n_groups, lay.feature_begin, lay.n_features])
    for name, cam in cams.items():
        sc = synth.g_level(4, 10, 10, 1, seed=11, cam=cam)
        Js, inns, xps = [], [], []
        for i in range(10):
            xcn = sc["Xcn"][0, i]
            xp_pred, _ = ref.camera_project(cam, xcn[:2] / xcn[2])
            xp = xp_pred + sc["pix_noise"][0, i]
            ref_s, s = int(sc["ref"][0, i]), int(sc["sind"][0, i])
            J, inn, _ = ref.compute_jacobian(sc["x"][0, i], xp, sc["gR"][0, ref_s], sc["gT"][0, ref_s], sc["Rsb"][0],
                                             sc["Tsb"][0], sc["Rbc"][0], sc["Tbc"][0], cam, lay, ref_s, s)
            Js.append(J); inns.append(inn); xps.append(xp)
        for k in ("x", "gR", "gT", "Rsb", "Tsb", "Rbc", "Tbc", "ref", "sind"):
            g[f"jac_{name}_{k}"] = sc[k][0]
        g[f"jac_{name}_xp"] = np.array(xps)
        g[f"jac_{name}_J"] = np.array(Js)
        g[f"jac_{name}_inn"] = np.array(inns)
        H = np.zeros((20, lay.N))
        for i in range(10):
            ref.fill_jacobian_block(H, 2 * i, Js[i], lay, int(sc["ref"][0, i]), int(sc["sind"][0, i]))
        g[f"jac_{name}_H"] = H
        A = rng.uniform(-1, 1, size=(lay.N, lay.N))
        P = A @ A.T / lay.N + 1e-3 * np.eye(lay.N)
        g[f"jac_{name}_P"] = P
        g[f"jac_{name}_dist"] = ref.mh_distances(np.array(Js), P, np.array(inns), 2.25)
    # --- a8/a9 OOS: per-observation Jacobians + SlowGivens projection
    sc = synth.g_level(6, 4, 4, 1, seed=21, cam=synth.PINHOLE)
    lay6 = orc.Layout(6, 4)
    g["oos_lay"] = np.array([lay6.N, lay6.group_begin, lay6.n_groups, lay6.feature_begin, lay6.n_features])
    Xs = np.array([0.3, -0.2, 4.0])
    Hf = np.zeros((10, 3)); Hx = np.zeros((10, lay6.N)); r = np.zeros(10)
    obs_xp = rng.uniform(100, 400, size=(5, 2))
    for c in range(5):
        hf, hx, inn = ref.oos_internal(Xs, sc["gR"][0, c], sc["gT"][0, c], sc["Rbc"][0], sc["Tbc"][0], obs_xp[c],
                                       synth.PINHOLE, lay6, c)
        Hf[2 * c:2 * c + 2], Hx[2 * c:2 * c + 2], r[2 * c:2 * c + 2] = hf, hx, inn
    Hxp, rp, A = ref.slow_givens(Hf, Hx, r)
    for k, v in dict(Xs=Xs, gR=sc["gR"][0], gT=sc["gT"][0], Rbc=sc["Rbc"][0], Tbc=sc["Tbc"][0], xp=obs_xp, Hf=Hf, Hx=Hx,
                     r=r, Hxp=Hxp, rp=rp, A=A).items():
        g[f"oos_{k}"] = v
    # --- FullPivLU kernel on a rank-deficient and a full-rank case
    A1 = rng.normal(size=(3, 8)); A2 = A1.copy(); A2[2] = 2 * A2[0] - A2[1]
    for tag, M in (("full", A1), ("def", A2)):
        ker, rank = ref.fullpivlu_kernel(M)
        g[f"lu_{tag}_A"], g[f"lu_{tag}_ker"], g[f"lu_{tag}_rank"] = M, ker, np.array(rank)
    # --- Givens (the unused orthonormal elimination, helpers.cpp:48-75)
    Hf = rng.normal(size=(8, 3)); Hx = rng.normal(size=(8, 5)); x = rng.normal(size=8)
    rows, xo, Hxo, Hfo = ref.Givens(x, Hx, Hf)
    g["giv_Hf"], g["giv_Hx"], g["giv_x"] = Hf, Hx, x
    g["giv_rows"], g["giv_xo"], g["giv_Hxo"], g["giv_Hfo"] = np.array(rows), xo, Hxo, Hfo
    # --- propagation tail (rk4.cpp:89-102, estimator.cpp:590)
    N, nm = 41, 23
    A = rng.uniform(-1, 1, size=(N, N)); P = A @ A.T / N + 1e-3 * np.eye(N)
    FK = rng.normal(size=(nm, nm)); PK = rng.normal(size=(nm, nm)); PK = PK + PK.T
    Q = np.diag(rng.uniform(1e-6, 1e-4, nm))
    g["prop_P"], g["prop_FK"], g["prop_PK"], g["prop_Q"], g["prop_dt"] = P, FK, PK, Q, np.array(0.002)
    g["prop_Pn"] = ref.rk4_cov_tail(P, FK, PK, 0.002, Q)
    # --- RK4Step (rk4.cpp:35-103) incl. ComposeMotion / ComputeMotionJacobianAt, line-faithful driver
    N = 41
    A = rng.uniform(-1, 1, size=(N, N)); P = A @ A.T / N * 1e-3 + 1e-6 * np.eye(N)
    X = orc.MotionState(orc.so3_exp([0.1, -0.2, 0.3]), [0.1, 0.2, 0.3], [0.5, -0.1, 0.2], [0.01, 0.02, -0.01],
                        [0.05, -0.02, 0.03], orc.so3_exp([0.01, 0.02, 0.0]))
    gv = np.array([0.0, 0.0, -9.8]); Qi = np.diag([1e-4] * 3 + [1e-3] * 3 + [1e-6] * 3 + [1e-5] * 3)
    gy, ac = np.array([0.1, 0.2, -0.1]), np.array([0.3, 0.1, 9.7])
    sg, sa = np.array([1.0, -2.0, 0.5]), np.array([0.2, 0.1, -0.3])
    R1, T1, V1, P1 = ref.rk4_step(X, P, gy, ac, sg, sa, 0.002, Qi, gv)
    for k, v in dict(P=P, Rsb=X.Rsb, Tsb=X.Tsb, Vsb=X.Vsb, bg=X.bg, ba=X.ba, Rsg=X.Rsg, g=gv, Qimu=Qi, gyro=gy, accel=ac, sg=sg,
                     sa=sa, dt=np.array(0.002), Rn=R1, Tn=T1, Vn=V1, Pn=P1).items():
        g[f"rk4_{k}"] = v
    # --- Sophus SO3::exp
    w = rng.normal(size=(4, 3)) * 0.5
    g["so3_w"] = w
    g["so3_R"] = np.array([ref.so3_exp(v) for v in w])
    # --- Feature::SubfilterUpdate (feature.cpp:246-297), 4 cameras x 3 features x 3 consecutive frames
    #     (feature 2 gets a wild pixel in frame 1: the ratio > 1 branch)
    rs = np.random.default_rng(2024)
    for name, cam in cams.items():
        sc = synth.g_level(3, 3, 3, 1, seed=31, cam=cam)
        xs, Ps, stats, ics, ocs, xp_frames = [], [], [], [], [], []
        x = sc["x"][0] + rs.normal(size=(3, 3)) * np.array([0.01, 0.01, 0.15])
        g[f"sub_{name}_x0"] = x.copy()
        P = np.array([np.diag([1e-4, 1e-4, 0.25]) for _ in range(3)])
        ic = np.zeros(3, dtype=np.int64); oc = np.zeros(3)
        for fr in range(3):
            xp = np.empty((3, 2))
            for i in range(3):
                Xcn = sc["Xcn"][0, i]
                xp[i] = orc.camera_project(cam, Xcn[:2] / Xcn[2])[0] + rs.normal(size=2) * 0.8
            if fr == 1:
                xp[2] += 25.0
            xp_frames.append(xp)
            st = np.zeros(3, dtype=np.int64)
            for i in range(3):
                r = int(sc["ref"][0, i])
                x[i], P[i], st[i], ic[i], oc[i] = ref.subfilter_update(
                    x[i], P[i], xp[i], sc["Rsb"][0], sc["Tsb"][0], sc["Rbc"][0], sc["Tbc"][0], sc["gR"][0, r], sc["gT"][0, r],
                    cam, 3.5, 5.991, 1, int(ic[i]), float(oc[i]))
            xs.append(x.copy()); Ps.append(P.copy()); stats.append(st.copy()); ics.append(ic.copy()); ocs.append(oc.copy())
        g[f"sub_{name}_xp"] = np.array(xp_frames)
        g[f"sub_{name}_x"] = np.array(xs); g[f"sub_{name}_P"] = np.array(Ps)
        g[f"sub_{name}_status"] = np.array(stats); g[f"sub_{name}_ic"] = np.array(ics); g[f"sub_{name}_oc"] = np.array(ocs)
        for k in ("Rsb", "Tsb", "Rbc", "Tbc", "gR", "gT", "ref"):
            g[f"sub_{name}_{k}"] = sc[k][0]
    out = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out), "bytes,", len(g), "arrays")


if __name__ == "__main__":
    main()
