"""The C++ adapter (xivo_amd/host/estimator_hip.h) keeps the reference's member-function
surface: ComputeInstateJacobians -> MHGating -> FilterUpdate (src/manager.cpp:72-104).
Reads like the reference flow; checked against the oracle flow over inliers only."""
import ctypes as C
import os

import numpy as np
import pytest

import xivo_oracle as orc
from helpers import rel_fro, TOL_P, TOL_DX
from scene_util import scene_arrays, oracle_jacobians, spd
from xivo_amd import synth
from xivo_amd.lib import Cam, Layout

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("flags", [0, 1])
def test_update_step_through_cpp_adapter(built, flags):
    lib = C.CDLL(os.path.join(ROOT, "xivo_amd", "libxivo_host.so"))
    cam = synth.EQUI
    ng, nf, F = 6, 16, 16
    sc = synth.g_level(ng, nf, F, 1, seed=77, cam=cam)
    lay = orc.Layout(ng, nf)
    poses, groups, feats, xp = scene_arrays(sc, cam)
    feats["xp"][0, [3, 9]] += 60.0; xp[0, [3, 9]] += 60.0
    P = spd(lay.N, 5) * 1e-4
    Pio = np.asfortranarray(P.copy())
    err = np.zeros(lay.N); mask = np.zeros(F, dtype=np.uint8); nrej = C.c_int(); msg = C.create_string_buffer(256)
    clay = Layout(lay.N, lay.group_begin, ng, lay.feature_begin, nf)
    ccam = Cam(); ccam.model, ccam.rows, ccam.cols = cam["model"], cam["rows"], cam["cols"]
    ccam.fx, ccam.fy, ccam.cx, ccam.cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    for i, v in enumerate(cam["d"]):
        ccam.d[i] = v
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.xivo_host_selftest_update_step(C.byref(clay), C.byref(ccam), C.c_uint(flags), F, p(poses), p(groups), p(feats),
                                            p(Pio), C.c_double(2.25), C.c_double(5.991), C.c_double(1.1), 5, p(err),
                                            p(mask), C.byref(nrej), msg, 256)
    assert rc == 0, msg.value
    Js, inns, _ = oracle_jacobians(sc, cam, lay, xp, 0)
    d = orc.mh_distances(Js, P, inns, 2.25)
    m, rej, _ = orc.mh_gate(d, 5.991, 1.1, 5)
    assert np.array_equal(mask.astype(bool), m) and nrej.value == rej and (~m).sum() == 2
    idx = np.nonzero(m)[0]
    H, inn, dR = orc.stack_measurements(Js[idx], inns[idx], sc["ref"][0][idx], sc["sind"][0][idx], lay, 2.25,
                                        fix_group_block=bool(flags & 1))
    e_ref, P_ref, _ = orc.update_joseph(H, P, inn, dR)
    assert rel_fro(np.ascontiguousarray(Pio), P_ref) < TOL_P and rel_fro(err, e_ref) < TOL_DX


@pytest.mark.parametrize("method,visual", [("RK4", False), ("RK4", True), ("PD", False)])
def test_propagate_through_cpp_adapter(built, method, visual):
    """Estimator::Propagate (estimator.cpp:539-592): host stages + device covariance tail."""
    lib = C.CDLL(os.path.join(ROOT, "xivo_amd", "libxivo_host.so"))
    N = 203
    rng = np.random.default_rng(8)
    P = spd(N, 3) * 1e-3
    X = orc.MotionState(orc.so3_exp([0.1, -0.2, 0.3]), [0.1, 0.2, 0.3], [0.5, -0.1, 0.2], [0.01, 0.02, -0.01],
                        [0.05, -0.02, 0.03], orc.so3_exp([0.01, 0.02, 0.0]))
    gv = np.array([0.0, 0.0, -9.8])
    Qi = np.diag([1e-4] * 3 + [1e-3] * 3 + [1e-6] * 3 + [1e-5] * 3); Qm = np.diag(rng.uniform(1e-9, 1e-7, 23))
    last_g, last_a = np.array([0.1, 0.2, -0.1]), np.array([0.3, 0.1, 9.7])
    curr_g, curr_a = np.array([0.12, 0.19, -0.08]), np.array([0.31, 0.12, 9.69])
    slope_g, slope_a = np.array([1.0, -2.0, 0.5]), np.array([0.2, 0.1, -0.3])   # from the previous IMU sample
    dt = 0.005
    if visual:
        sg, sa = slope_g, slope_a
    else:
        sg, sa = (curr_g - last_g) / dt, (curr_a - last_a) / dt
    Xe, Pe = orc.propagate(X, P, last_g, last_a, sg, sa, dt, Qi, Qm, gv, method=method)
    st = np.concatenate([X.Rsb.T.reshape(-1), X.Tsb, X.Vsb, X.bg, X.ba, X.Rsg.T.reshape(-1)]).copy()
    imu = np.concatenate([last_g, last_a, curr_g, curr_a, slope_g, slope_a]).copy()
    Pio = np.asfortranarray(P.copy()); msg = C.create_string_buffer(256)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    Qif, Qmf = np.asfortranarray(Qi), np.asfortranarray(Qm)
    rc = lib.xivo_host_selftest_propagate(N, int(method == "RK4"), int(visual), C.c_double(dt), C.c_double(0.002), p(st), p(Pio),
                                          p(imu), p(Qif), p(Qmf), p(gv), msg, 256)
    assert rc == 0, msg.value
    assert np.abs(st[0:9].reshape(3, 3).T - Xe.Rsb).max() < 1e-13
    assert np.abs(st[9:12] - Xe.Tsb).max() < 1e-14 and np.abs(st[12:15] - Xe.Vsb).max() < 1e-13
    assert rel_fro(np.ascontiguousarray(Pio), Pe) < 1e-12          # accumulated-Phi tail == per-sub-step tails
    assert np.allclose(imu[12:15], sg) and np.allclose(imu[15:18], sa)


def test_one_point_ransac_through_cpp_adapter(built):
    """Estimator::OnePointRANSAC (update.cpp:213-393) as a composition over the device pieces."""
    lib = C.CDLL(os.path.join(ROOT, "xivo_amd", "libxivo_host.so"))
    cam = synth.PINHOLE
    ng, nf, F = 4, 12, 12
    sc = synth.g_level(ng, nf, F, 1, seed=5, cam=cam)
    lay = orc.Layout(ng, nf)
    poses, groups, feats, xp = scene_arrays(sc, cam)
    feats["xp"][0, [2, 7]] += [6.0, -5.0]; xp[0, [2, 7]] += [6.0, -5.0]
    feats["xp"][0, 9] += 90.0; xp[0, 9] += 90.0
    P = spd(lay.N, 1) * 1e-4
    st = dict(Rsb=sc["Rsb"][0], Tsb=sc["Tsb"][0], Vsb=np.zeros(3), bg=np.zeros(3), ba=np.zeros(3), Rbc=sc["Rbc"][0],
              Tbc=sc["Tbc"][0], Rsg=np.eye(3), gR=sc["gR"][0].copy(), gT=sc["gT"][0].copy(), x=sc["x"][0].copy(),
              sind=sc["sind"][0], ref=sc["ref"][0])
    exp = orc.one_point_ransac(st, P, xp[0], cam, lay, 2.25, 5.0, 5.89, 0, list(range(ng)))
    assert (~exp["low"]).sum() == 3 and len(exp["rejected"]) == 2      # the scenario exercises rescue and rejection
    clay = Layout(lay.N, lay.group_begin, ng, lay.feature_begin, nf)
    ccam = Cam(); ccam.model, ccam.rows, ccam.cols = cam["model"], cam["rows"], cam["cols"]
    ccam.fx, ccam.fy, ccam.cx, ccam.cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    kept = np.zeros(F, dtype=np.uint8); chi2 = np.zeros(F); nrej = C.c_int(); rerr = C.c_double(); msg = C.create_string_buffer(256)
    Pf = np.asfortranarray(P)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.xivo_host_selftest_ransac(C.byref(clay), C.byref(ccam), F, p(poses), p(groups), p(feats), p(Pf), C.c_double(2.25),
                                       C.c_double(5.0), C.c_double(5.89), 0, p(kept), p(chi2), C.byref(nrej), C.byref(rerr), msg, 256)
    assert rc == 0, msg.value
    assert sorted(np.nonzero(kept)[0].tolist()) == exp["inliers"] and nrej.value == len(exp["rejected"])
    for i, d in exp["chi2"].items():
        assert abs(chi2[i] - d) < 1e-7 * max(1.0, d)
    assert rerr.value == 0.0          # RestoreState: P_ and the nominal state are bit-identical to the backup


def test_multi_frame_sequence_replay(built):
    """Four camera frames of one filter through the adapter - IMU propagation (Dormand-Prince), Jacobians,
    MH gating, stacking with the FillJacobianBlock quirk, Joseph update, AbsorbError, state feedback -
    against the same flow assembled from the oracle pieces. Parity is asserted on the final P and state."""
    from xivo_amd.lib import group_dtype, feat_dtype
    lib = C.CDLL(os.path.join(ROOT, "xivo_amd", "libxivo_host.so"))
    cam = synth.PINHOLE
    ng, nf, F, T, n_imu, dt = 4, 10, 10, 4, 4, 0.0025
    sc = synth.g_level(ng, nf, F, 1, seed=31, cam=cam)
    lay = orc.Layout(ng, nf); N = lay.N
    rng = np.random.default_rng(4)
    P0 = spd(N, 9) * 1e-4
    gv = np.array([0.0, 0.0, -9.8])
    Rsg = orc.so3_exp([0.01, -0.02, 0.0])
    st = dict(Rsb=sc["Rsb"][0].copy(), Tsb=sc["Tsb"][0].copy(), Vsb=np.array([0.02, -0.01, 0.01]), bg=np.array([1e-3, -2e-3, 1e-3]),
              ba=np.array([0.02, 0.01, -0.01]), Rbc=sc["Rbc"][0].copy(), Tbc=sc["Tbc"][0].copy(), Rsg=Rsg, gR=sc["gR"][0].copy(),
              gT=sc["gT"][0].copy(), x=sc["x"][0].copy(), sind=sc["sind"][0], ref=sc["ref"][0])
    hover = -(st["Rsb"].T @ Rsg @ gv) + st["ba"]
    imu = np.empty((T, n_imu, 6))
    imu[..., :3] = st["bg"] + rng.normal(0, 0.02, size=(T, n_imu, 3))
    imu[..., 3:] = hover + rng.normal(0, 0.05, size=(T, n_imu, 3))
    pix = np.empty((T, F, 2))
    for t in range(T):
        for i in range(F):
            Xcn = sc["Xcn"][0, i]
            pix[t, i] = orc.camera_project(cam, Xcn[:2] / Xcn[2])[0] + rng.normal(0, 1.0, 2)
    pix[1, 3] += 70.0                                    # one gross outlier in frame 1
    Qi = np.diag([1e-4] * 3 + [1e-3] * 3 + [1e-6] * 3 + [1e-5] * 3); Qm = np.diag(rng.uniform(1e-9, 1e-7, 23))

    # ---- oracle flow
    s = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in st.items()}
    P = P0.copy()
    last_g, last_a = imu[0, 0, :3].copy(), imu[0, 0, 3:].copy()
    sg, sa = np.zeros(3), np.zeros(3)
    alive = np.ones(F, dtype=bool)
    n_inl = []

    def prop(dtv, g0, a0):
        nonlocal P
        X = orc.MotionState(s["Rsb"], s["Tsb"], s["Vsb"], s["bg"], s["ba"], s["Rsg"])
        Xn, P = orc.propagate(X, P, g0, a0, sg, sa, dtv, Qi, Qm, gv, method="PD")
        s["Rsb"], s["Tsb"], s["Vsb"] = Xn.Rsb, Xn.Tsb, Xn.Vsb
    for t in range(T):
        for k in range(n_imu):
            cg, ca = imu[t, k, :3], imu[t, k, 3:]
            if t == 0 and k == 0:
                continue
            sg, sa = (cg - last_g) / dt, (ca - last_a) / dt
            prop(dt, last_g, last_a)
            last_g, last_a = cg.copy(), ca.copy()
        prop(0.5 * dt, last_g, last_a)
        last_g, last_a = last_g + sg * 0.5 * dt, last_a + sa * 0.5 * dt
        idx = [i for i in range(F) if alive[i]]
        JI = [orc.compute_jacobian(s["x"][i], pix[t, i], s["gR"][int(s["ref"][i])], s["gT"][int(s["ref"][i])], s["Rsb"], s["Tsb"],
                                   s["Rbc"], s["Tbc"], cam, lay, int(s["ref"][i]), int(s["sind"][i]))[:2] for i in idx]
        Js = np.array([j for j, _ in JI]); inns = np.array([r for _, r in JI])
        m = orc.mh_gate(orc.mh_distances(Js, P, inns, 2.25), 5.991, 1.1, 5)[0] if len(idx) > 5 else np.ones(len(idx), bool)
        keep = [idx[q] for q in range(len(idx)) if m[q]]
        n_inl.append(len(keep))
        H, inn, dR = orc.stack_measurements(Js[m], inns[m], [s["ref"][i] for i in keep], [s["sind"][i] for i in keep], lay, 2.25)
        err, P, _ = orc.update_joseph(H, P, inn, dR)
        orc.absorb_error(s, err, lay, list(range(ng)), keep)
        for q, i in enumerate(idx):
            if not m[q]:
                alive[i] = False
                P = orc.p_zero_rc(P, lay.feature_begin + 3 * int(s["sind"][i]), 3)
    assert n_inl[1] == F - 1 and not alive[3]

    # ---- adapter flow
    _, groups, feats, _ = scene_arrays(sc, cam)
    state30 = np.concatenate([st["Rsb"].T.reshape(-1), st["Tsb"], st["Vsb"], st["bg"], st["ba"], Rsg.T.reshape(-1)]).copy()
    Pio = np.asfortranarray(P0.copy()); inl = np.zeros(T, dtype=np.int32); msg = C.create_string_buffer(256)
    clay = Layout(N, lay.group_begin, ng, lay.feature_begin, nf)
    ccam = Cam(); ccam.model, ccam.rows, ccam.cols = cam["model"], cam["rows"], cam["cols"]
    ccam.fx, ccam.fy, ccam.cx, ccam.cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    g_io = np.ascontiguousarray(groups[0]); f_io = np.ascontiguousarray(feats[0])
    Rbc_cm = np.ascontiguousarray(st["Rbc"].T.reshape(-1)); Tbc = np.ascontiguousarray(st["Tbc"])
    imu_c, pix_c = np.ascontiguousarray(imu), np.ascontiguousarray(pix)
    Qif, Qmf = np.asfortranarray(Qi), np.asfortranarray(Qm)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib.xivo_host_selftest_sequence(C.byref(clay), C.byref(ccam), F, T, n_imu, C.c_double(dt), p(imu_c), p(pix_c), p(state30),
                                         p(g_io), p(f_io), p(Rbc_cm), p(Tbc), p(Pio), p(Qif), p(Qmf), p(gv), 0, C.c_double(2.25),
                                         p(inl), msg, 256)
    assert rc == 0, msg.value
    assert inl.tolist() == n_inl
    assert rel_fro(np.ascontiguousarray(Pio), P) < TOL_P
    assert np.abs(state30[0:9].reshape(3, 3).T - s["Rsb"]).max() < 1e-9
    assert np.abs(state30[9:12] - s["Tsb"]).max() < 1e-9 and np.abs(state30[12:15] - s["Vsb"]).max() < 1e-9
    assert np.abs(state30[15:18] - s["bg"]).max() < 1e-10 and np.abs(state30[18:21] - s["ba"]).max() < 1e-10
    for g_ in range(ng):
        assert np.abs(np.asarray(g_io[g_]["Tsb"]) - s["gT"][g_]).max() < 1e-9
    for i in range(F):
        assert np.abs(np.asarray(f_io[i]["x"]) - s["x"][i]).max() < 1e-8


@pytest.mark.parametrize("use_ransac", [0, 1])
def test_online_calibration_build_of_the_cpp_adapter(built, use_ransac):
    """xivo_amd/host/estimator_hip.{h,cpp} compiled with the reference's three online-calibration defines
    (libxivo_host_calib.so: -DUSE_ONLINE_TEMPORAL_CALIB -DUSE_ONLINE_IMU_CALIB -DUSE_ONLINE_CAMERA_CALIB): the constructor derives
    Index::td / Cg / kCameraBegin / kMotionSize as src/core.h:40-105 numbers them and switches the context
    (xivo_hip_set_calib); Propagate (39-dimensional motion block, device-native), ComputeInstateJacobians with the td / Cg / bg /
    intrinsics blocks in Feature::J_, MHGating on the whole row, [OnePointRANSAC,] FilterUpdate through FillJacobianBlock with
    those blocks, AbsorbError of td / Ca / Cg / intrinsics - the reference's UpdateStep flow, against the oracle."""
    from xivo_amd.lib import calib_dtype, cam_intr, imu_dtype
    lib = C.CDLL(os.path.join(ROOT, "xivo_amd", "libxivo_host_calib.so"))
    cam = synth.RADTAN
    ng, nf, F = 6, 16, 16
    lay = orc.calib_layout(ng, nf, True, True, 9)
    sc = synth.g_level(ng, nf, F, 1, seed=91, cam=cam)
    poses, groups, feats, xp = scene_arrays(sc, cam)
    rng = np.random.default_rng(12)
    if use_ransac:
        xp = xp - sc["pix_noise"] + rng.normal(size=xp.shape) * 0.3
        xp[0, [3, 9]] += [[3.0, -2.5], [-2.8, 3.1]]; xp[0, 12] += 40.0
    else:
        xp[0, [3, 9]] += 60.0
    feats["xp"] = xp
    cal = dict(gyro=rng.normal(size=3) * 0.3, Cg=np.eye(3) + 0.01 * rng.normal(size=(3, 3)), bg=rng.normal(size=3) * 0.01,
               Vsb=rng.normal(size=3) * 0.5, td=0.012)
    Ca = np.triu(np.eye(3) + 0.01 * rng.normal(size=(3, 3)))
    poses[0]["Vsb"], poses[0]["bg"], poses[0]["ba"] = cal["Vsb"], cal["bg"], rng.normal(size=3) * 0.02
    poses[0]["Rsg"] = orc.so3_exp(np.array([0.01, -0.02, 0.0])).T.reshape(-1)
    calib = np.zeros(1, dtype=calib_dtype)
    calib[0]["gyro"], calib[0]["Cg"], calib[0]["td"] = cal["gyro"], cal["Cg"].T.reshape(-1), cal["td"]
    calib[0]["Ca"], calib[0]["intr"] = Ca.T.reshape(-1), cam_intr(cam)
    P = spd(lay.N, 7) * 1e-4
    Pio = np.asfortranarray(P.copy())
    err = np.zeros(lay.N); mask = np.zeros(F, dtype=np.uint8); msg = C.create_string_buffer(256); slots = np.zeros(5, dtype=np.int32)
    clay = Layout(lay.N, lay.group_begin, ng, lay.feature_begin, nf)
    ccam = Cam(); ccam.model, ccam.rows, ccam.cols = cam["model"], cam["rows"], cam["cols"]
    ccam.fx, ccam.fy, ccam.cx, ccam.cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    for i, v in enumerate(cam["d"]):
        ccam.d[i] = v
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    # one IMU sample in front of the update (visual message: slopes as handed in)
    imu = np.zeros(1, dtype=imu_dtype)
    imu[0]["gyro"], imu[0]["accel"] = cal["gyro"], np.array([0.2, -0.1, 9.7])
    imu[0]["slope_gyro"], imu[0]["slope_accel"], imu[0]["dt"] = np.array([1.0, -2.0, 0.5]), np.array([0.2, 0.1, -0.3]), 0.005
    Qi = np.asfortranarray(np.diag([1e-4] * 3 + [1e-3] * 3 + [1e-6] * 3 + [1e-5] * 3)); nm = lay.motion_size
    Qm = np.asfortranarray(np.diag(rng.uniform(1e-9, 1e-7, nm))); gv = np.array([0.0, 0.0, -9.8])
    R1, TH, CHI = 2.25, 2.0, 5.89
    rc = lib.xivo_host_selftest_update_step_calib(C.byref(clay), C.byref(ccam), C.c_uint(0), F, p(poses), p(groups), p(feats), p(calib), p(Pio),
                                                  C.c_double(R1), C.c_double(5.991), C.c_double(1.1), 5, use_ransac, C.c_double(TH),
                                                  C.c_double(CHI), p(err), p(mask), 1, p(imu), p(Qi), p(Qm), p(gv), 0, p(slots), msg, 256)
    assert rc == 0, msg.value
    assert slots.tolist() == [lay.td, lay.Cg, lay.cam_begin, 9, nm] and nm == 39
    # ---- oracle: Propagate, Jacobians at the propagated state, gating [RANSAC], update, absorb
    X = orc.MotionState(sc["Rsb"][0], sc["Tsb"][0], cal["Vsb"], cal["bg"], np.asarray(poses[0]["ba"]).copy(), orc.so3_exp(np.array([0.01, -0.02, 0.0])))
    Xe, Pe = orc.propagate(X, P, imu[0]["gyro"], imu[0]["accel"], imu[0]["slope_gyro"], imu[0]["slope_accel"], 0.005, Qi, Qm, gv, method="RK4",
                           Cg=cal["Cg"], Ca=Ca, layout=lay)[:2]
    gyro_after = imu[0]["gyro"] + imu[0]["slope_gyro"] * 0.005             # last_gyro_ advanced by the visual message (estimator.cpp:573-574)
    cal2 = dict(gyro=gyro_after, Cg=cal["Cg"], bg=cal["bg"], Vsb=Xe.Vsb, td=cal["td"])
    Js, inns = [], []
    for i in range(F):
        r = int(sc["ref"][0][i])
        Ji, ii, _, _ = orc.compute_jacobian(sc["x"][0][i], xp[0][i], sc["gR"][0][r], sc["gT"][0][r], Xe.Rsb, Xe.Tsb, sc["Rbc"][0], sc["Tbc"][0], cam, lay,
                                            r, int(sc["sind"][0][i]), calib=cal2)
        Js.append(Ji); inns.append(ii)
    Js, inns = np.array(Js), np.array(inns)
    m, _, _ = orc.mh_gate(orc.mh_distances(Js, Pe, inns, R1), 5.991, 1.1, 5)
    keep = m.copy()
    if use_ransac:
        idx = np.nonzero(m)[0]
        st = dict(Rsb=Xe.Rsb.copy(), Tsb=Xe.Tsb.copy(), Vsb=Xe.Vsb.copy(), bg=cal["bg"].copy(), ba=np.asarray(poses[0]["ba"]).copy(), Rbc=sc["Rbc"][0].copy(),
                  Tbc=sc["Tbc"][0].copy(), Rsg=X.Rsg.copy(), gR=sc["gR"][0].copy(), gT=sc["gT"][0].copy(), x=sc["x"][0][idx].copy(),
                  sind=sc["sind"][0][idx], ref=sc["ref"][0][idx], td=cal["td"], Cg=cal["Cg"].copy(), Ca=Ca.copy(), cam=dict(cam, d=list(cam["d"])))
        out = orc.one_point_ransac(st, Pe, xp[0][idx], cam, lay, R1, TH, CHI, 0, range(ng), calib_gyro=gyro_after)
        keep = np.zeros(F, dtype=bool); keep[idx[out["inliers"]]] = True
        assert 0 < out["low"].sum() < len(idx)                              # a partial update and a rescue pass really happen
    assert np.array_equal(mask.astype(bool), keep), (mask, keep)
    k = np.nonzero(keep)[0]
    H, inn, dR = orc.stack_measurements(Js[k], inns[k], sc["ref"][0][k], sc["sind"][0][k], lay, R1)
    e_ref, P_ref, _ = orc.update_joseph(H, Pe, inn, dR)
    assert rel_fro(np.ascontiguousarray(Pio), P_ref) < TOL_P and rel_fro(err, e_ref) < TOL_DX
    # AbsorbError: td, Ca (upper triangle, row by row), Cg (row by row), the nine RADTAN intrinsics
    assert abs(calib[0]["td"] - (cal["td"] + e_ref[lay.td])) < 1e-12
    Cg_new = calib[0]["Cg"].reshape(3, 3).T; Ca_new = calib[0]["Ca"].reshape(3, 3).T
    assert np.abs(Cg_new - (cal["Cg"] + e_ref[lay.Cg:lay.Cg + 9].reshape(3, 3))).max() < 1e-12
    dCa = np.zeros((3, 3)); dCa[np.triu_indices(3)] = e_ref[lay.Ca:lay.Ca + 6]
    assert np.abs(Ca_new - (Ca + dCa)).max() < 1e-12
    assert np.abs(calib[0]["intr"] - (cam_intr(cam) + e_ref[lay.cam_begin:lay.cam_begin + 9])).max() < 1e-9
    assert np.abs(np.asarray(poses[0]["Tsb"]) - (Xe.Tsb + e_ref[3:6])).max() < 1e-9
