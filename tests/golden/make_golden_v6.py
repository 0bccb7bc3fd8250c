#!/usr/bin/env python
"""Generates tests/golden/golden_v6.npz from the EXTRACTED reference builds (oracle/_ref/libxivo_refx_*.so: the reference's own
text cut out of /root/reference/src and compiled, oracle/ref/extract_reference.py) - the rows of the path that rounds 1-4
pinned by a retyping only, plus the two reference variants built in round 5:

  lc_*      Feature::ComputeLCJacobian (src/oos.cpp:92-145) under the stacking of Estimator::CloseLoopInternal
            (src/update.cpp:183-196): default build (N = 251), online-calibration build (N = 228: the intrinsics block),
            USE_INVDEPTH build
  inv_*     Feature::ComputeJacobian + FillJacobianBlock, Feature::SubfilterUpdate of the USE_INVDEPTH build
  oos_*     Feature::ComputeOOSJacobian + ComputeOOSJacobianInternal + SlowGivens (src/oos.cpp:8-89), whole-buffer quirk included
  sub_*     Feature::SubfilterUpdate (src/feature.cpp:246-297), default build
  prop_*    Estimator::Propagate (src/estimator.cpp:539-592) through Estimator::RK4 / PrinceDormand (outer loops,
            src/rk4.cpp:5-33, src/princedormand.cpp:7-83) and their steps
  ransac_*  Estimator::OnePointRANSAC (src/update.cpp:213-393), the whole function

The case generators below are imported by tests/test_oracle_pinned.py and the GPU tests, so inputs are never stored twice.
Run in the authoring container only:  python tests/golden/make_golden_v6.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from xivo_amd import synth  # noqa: E402

CAMS = {"pinhole": synth.PINHOLE, "equi": synth.EQUI, "radtan": synth.RADTAN, "atan": synth.ATAN}
CAM_DIM = {"pinhole": 4, "atan": 5, "radtan": 9, "equi": 8}
BUILD_SIZES = {"n251": (8, 60), "invdepth": (8, 60), "calib": (15, 30)}


def to_invdepth(x):
    """(X/Z, Y/Z, log Z) -> (X/Z, Y/Z, 1/Z): the same point in the USE_INVDEPTH parametrisation"""
    x = np.array(x, dtype=np.float64, copy=True)
    x[..., 2] = 1.0 / np.exp(x[..., 2])
    return x


def lc_case(build, cam, seed, n=6):
    """n loop-closure matches of one filter: old feature i (anchored to group ref[i]) re-observed from another group slot.
    Returns (scene, matches list for the oracle, arrays for the extracted wrapper / the device)."""
    ng, nf = BUILD_SIZES[build]
    sc = synth.g_level(ng, nf, nf, 1, seed=seed, cam=cam)
    rng = np.random.default_rng(50 + seed)
    x_all = to_invdepth(sc["x"][0]) if build == "invdepth" else sc["x"][0].copy()
    matches, feat_idx = [], []
    cand = rng.permutation(nf)
    for i in cand:
        if len(matches) == n:
            break
        r = int(sc["ref"][0, i]); g = int((r + 1 + rng.integers(0, ng - 1)) % ng)
        z = np.exp(sc["x"][0, i, 2]); Xc = np.array([sc["x"][0, i, 0] * z, sc["x"][0, i, 1] * z, z])
        Xs = sc["gR"][0, r] @ (sc["Rbc"][0] @ Xc + sc["Tbc"][0]) + sc["gT"][0, r]
        Xcn = sc["Rbc"][0].T @ (sc["gR"][0, g].T @ (Xs - sc["gT"][0, g]) - sc["Tbc"][0])
        if Xcn[2] < 0.5 or abs(Xcn[0] / Xcn[2]) > 0.7 or abs(Xcn[1] / Xcn[2]) > 0.7:
            continue
        xp = np.array([cam["cx"], cam["cy"]]) + rng.normal(0, 60.0, 2)
        matches.append(dict(x=x_all[i], Rsbr=sc["gR"][0, r], Tsbr=sc["gT"][0, r], Rsb=sc["gR"][0, g], Tsb=sc["gT"][0, g], g_sind=g, xp=xp))
        feat_idx.append(int(i))
    assert len(matches) == n
    return sc, x_all, matches, feat_idx


def oos_case(cam, seed, k=5):
    """one out-of-state feature seen from k of 8 in-state groups (N = 251 build)"""
    sc = synth.g_level(8, 60, 60, 1, seed=seed, cam=cam)
    rng = np.random.default_rng(70 + seed)
    i = int(rng.integers(0, 60)); r = int(sc["ref"][0, i])
    gs = [int(g) for g in rng.permutation(8)[:k]]
    obs = [(g, np.array([cam["cx"], cam["cy"]]) + rng.normal(0, 40.0, 2)) for g in gs]
    return sc, i, r, obs


def sub_case(cam, seed, invdepth=False):
    sc = synth.g_level(4, 12, 12, 1, seed=seed, cam=cam)
    rng = np.random.default_rng(90 + seed)
    i = int(rng.integers(0, 12)); r = int(sc["ref"][0, i])
    x = to_invdepth(sc["x"][0, i]) if invdepth else sc["x"][0, i].copy()
    A = rng.uniform(-1, 1, (3, 3)); P = A @ A.T * 0.01 + np.diag([1e-3, 1e-3, 0.05 if not invdepth else 1e-3])
    xp = np.array([cam["cx"], cam["cy"]]) + rng.normal(0, 30.0, 2)
    return dict(x=x, P=P, xp=xp, Rsb=sc["Rsb"][0], Tsb=sc["Tsb"][0], Rbc=sc["Rbc"][0], Tbc=sc["Tbc"][0], Rsbr=sc["gR"][0, r], Tsbr=sc["gT"][0, r],
                init_counter=int(rng.integers(0, 7)), outlier_counter=float(rng.uniform(0, 0.5)))


def prop_case(seed, N=203):
    import xivo_oracle as orc
    rng = np.random.default_rng(seed)
    A = rng.uniform(-1, 1, (N, N)); P = A @ A.T / N * 1e-3 + 1e-6 * np.eye(N)
    X = orc.MotionState(orc.so3_exp(rng.normal(size=3) * 0.3), rng.normal(size=3), rng.normal(size=3), rng.normal(size=3) * 0.01,
                        rng.normal(size=3) * 0.05, orc.so3_exp(np.array([0.02, -0.01, 0.0])))
    gy = rng.normal(size=3) * 0.5; ac = rng.normal(size=3) * 2 + np.array([0, 0, 9.8])
    sg = rng.normal(size=3) * 5; sa = rng.normal(size=3) * 20
    Qimu = np.diag(rng.uniform(1e-5, 1e-3, 12)); Qmodel = np.diag(rng.uniform(1e-10, 1e-8, 23)); g = np.array([0, 0, -9.8])
    # 2.5 ms (one full sub-step + the half-step trick's tail, src/rk4.cpp:19-26), 7 ms (three sub-steps + a remainder)
    return dict(P=P, X=X, gy=gy, ac=ac, sg=sg, sa=sa, Qimu=Qimu, Qmodel=Qmodel, g=g, w=rng.normal(size=N - 23))


def ransac_case(tag):
    """All F features are MH inliers; pixel noise mixes small and large innovations so that the low-innovation set, the
    chi-square rescue and the rejection all occur. 'tmp': the gauge group holds no low-innovation inlier (temporary
    reference group, src/update.cpp:292-301)."""
    cam = {"pin": synth.PINHOLE, "rad": synth.RADTAN, "tmp": synth.EQUI}[tag]
    seed = {"pin": 5, "rad": 6, "tmp": 7}[tag]
    ng, nf = 15, 30                                   # the default build's kMaxGroup / kMaxFeature (N = 203)
    sc = synth.g_level(ng, nf, nf, 1, seed=300 + seed, cam=cam)
    rng = np.random.default_rng(seed)
    N = 23 + 6 * ng + 3 * nf
    A = rng.uniform(-1, 1, size=(N, N)); P = (A @ A.T / N + 1e-3 * np.eye(N)) * 1e-4
    noise = rng.normal(size=(nf, 2)) * np.where(rng.uniform(size=(nf, 1)) < 0.6, 0.5, 4.0)
    return dict(sc=sc, cam=cam, P=P, noise=noise, R=1.0, thresh=1.8, chi2=5.89, gauge=(-1 if tag == "tmp" else 0), N=N, ng=ng, nf=nf)


def ransac_pixels(c, project):
    """measured pixel = predicted pixel (by the checker's own camera model `project(cam, xcn) -> xp`) + the case's noise"""
    sc = c["sc"]
    return np.array([project(c["cam"], sc["Xcn"][0, i][:2] / sc["Xcn"][0, i][2]) for i in range(c["nf"])]) + c["noise"]


def main():
    import ref_binding
    import xivo_oracle as orc
    out = {}
    # ---- loop-closure rows
    for build in ("n251", "calib", "invdepth"):
        lib = ref_binding.loadx({"n251": 251}.get(build, build))
        for name, cam in CAMS.items():
            sc, x_all, matches, _ = lc_case(build, cam, 3)
            H, inn = lib.compute_lc_jacobian([m["x"] for m in matches], [m["Rsbr"] for m in matches], [m["Tsbr"] for m in matches],
                                             [m["Rsb"] for m in matches], [m["Tsb"] for m in matches], [m["g_sind"] for m in matches],
                                             [m["xp"] for m in matches], sc["Rbc"][0], sc["Tbc"][0], cam)
            k = f"lc_{build}_{name}"
            cols = np.nonzero(np.abs(H).sum(0))[0].astype(np.int32)
            out[k + "_cols"] = cols; out[k + "_H"] = H[:, cols]; out[k + "_inn"] = inn; out[k + "_N"] = np.array([lib.N])
    # ---- USE_INVDEPTH: in-state Jacobian + stacked rows, depth sub-filter
    inv = ref_binding.loadx("invdepth")
    assert inv.use_invdepth() and inv.N == 251
    for name, cam in CAMS.items():
        sc = synth.g_level(8, 60, 60, 1, seed=9, cam=cam)
        for i in (0, 17, 59):
            r = int(sc["ref"][0, i]); x = to_invdepth(sc["x"][0, i])
            xp = np.array([cam["cx"], cam["cy"]]) + np.array([13.0 * (i % 5) - 20, 9.0 * (i % 7) - 25])
            J, inn, Hrow = inv.compute_jacobian(x, xp, sc["gR"][0, r], sc["gT"][0, r], sc["Rsb"][0], sc["Tsb"][0], sc["Rbc"][0], sc["Tbc"][0],
                                                cam, r, int(sc["sind"][0, i]))
            k = f"inv_jac_{name}_{i}"
            cols = np.nonzero(np.abs(J).sum(0))[0].astype(np.int32)
            out[k + "_cols"] = cols; out[k + "_J"] = J[:, cols]; out[k + "_inn"] = inn
            hc = np.nonzero(np.abs(Hrow).sum(0))[0].astype(np.int32)
            out[k + "_Hcols"] = hc; out[k + "_H"] = Hrow[:, hc]
        c = sub_case(cam, 4, invdepth=True)
        xs, Ps, st, ic, oc = inv.subfilter_update(c["x"], c["P"], c["xp"], c["Rsb"], c["Tsb"], c["Rbc"], c["Tbc"], c["Rsbr"], c["Tsbr"], cam,
                                                  3.5, 5.991, 5, c["init_counter"], c["outlier_counter"])
        out[f"inv_sub_{name}"] = np.concatenate([xs, Ps.reshape(-1), [st, ic, oc]])
    # ---- OOS rows, sub-filter (default build)
    d251 = ref_binding.loadx(251)
    for name, cam in CAMS.items():
        sc, i, r, obs = oos_case(cam, 2)
        rows, Hx, inn, Xs = d251.compute_oos_jacobian(sc["x"][0, i], sc["gR"][0, r], sc["gT"][0, r], obs, sc["gR"][0], sc["gT"][0], sc["Rbc"][0],
                                                      sc["Tbc"][0], cam, min_obs=5)
        k = f"oos_{name}"
        out[k + "_rows"] = np.array([rows]); out[k + "_Xs"] = Xs
        nz = np.nonzero(np.abs(Hx).sum(1))[0]
        cols = np.nonzero(np.abs(Hx).sum(0))[0].astype(np.int32)
        out[k + "_nzrows"] = nz.astype(np.int32); out[k + "_cols"] = cols; out[k + "_Hx"] = Hx[np.ix_(nz, cols)]; out[k + "_inn"] = inn
        c = sub_case(cam, 4)
        xs, Ps, st, ic, oc = d251.subfilter_update(c["x"], c["P"], c["xp"], c["Rsb"], c["Tsb"], c["Rbc"], c["Tbc"], c["Rsbr"], c["Tsbr"], cam,
                                                   3.5, 5.991, 5, c["init_counter"], c["outlier_counter"])
        out[f"sub_{name}"] = np.concatenate([xs, Ps.reshape(-1), [st, ic, oc]])
    # ---- Propagate with the integrators' outer loops (default build, N = 203)
    d203 = ref_binding.loadx(203)
    for seed in (1, 2):
        c = prop_case(seed)
        for method in ("RK4", "PrinceDormand"):
            for dt_ns in (2500000, 7000000):
                o = d203.propagate(method, c["X"], c["P"], c["gy"], c["ac"], c["sg"], c["sa"], dt_ns, c["Qimu"], c["Qmodel"], c["g"], stepsize=0.002)
                k = f"prop_s{seed}_{method}_{dt_ns}"
                out[k + "_Rsb"] = o["Rsb"]; out[k + "_Tsb"] = o["Tsb"]; out[k + "_Vsb"] = o["Vsb"]
                out[k + "_Pmm"] = o["P"][:23, :23]; out[k + "_Pms_w"] = o["P"][:23, 23:] @ c["w"]; out[k + "_Pleft_w"] = c["w"] @ o["P"][23:, :23]
                out[k + "_last"] = np.concatenate([o["last_gyro"], o["last_accel"]])
                assert np.array_equal(o["P"][23:, 23:], c["P"][23:, 23:])
    # ---- OnePointRANSAC, whole function (default build)
    for tag in ("pin", "rad", "tmp"):
        c = ransac_case(tag)
        sc = c["sc"]
        xp = ransac_pixels(c, lambda cam, xcn: orc.camera_project(cam, xcn)[0])
        X = orc.MotionState(sc["Rsb"][0], sc["Tsb"][0], np.zeros(3), np.zeros(3), np.zeros(3), np.eye(3))
        o = d203.one_point_ransac(X, sc["Rbc"][0], sc["Tbc"][0], c["P"], sc["x"][0], xp, sc["ref"][0], sc["sind"][0], sc["gR"][0], sc["gT"][0],
                                  c["cam"], c["R"], c["thresh"], c["chi2"], gauge_sind=c["gauge"])
        k = f"ransac_{tag}"
        out[k + "_xp"] = xp; out[k + "_keep"] = o["keep"]; out[k + "_status"] = o["status"]; out[k + "_nrej"] = np.array([o["n_rejected"]])
        assert np.array_equal(o["P"], c["P"]) and np.array_equal(o["x"], sc["x"][0])      # RestoreState (src/update.cpp:383)
        # J_ after the final re-computation = J at the original state
        cols = np.nonzero(np.abs(o["J"][3]).sum(0))[0].astype(np.int32)
        out[k + "_J3cols"] = cols; out[k + "_J3"] = o["J"][3][:, cols]
    path = os.path.join(ROOT, "tests", "golden", "golden_v6.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
