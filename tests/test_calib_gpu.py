"""Online-calibration builds (xivo_hip_set_calib). Measurement side: the td / Cg / bg / intrinsics blocks of
Feature::ComputeJacobian (src/feature.cpp:592-609, :611-618, :632-651), their stacking by Feature::FillJacobianBlock
(:664-670, :679-683), MH gating on the whole row and the update. Motion side: Estimator::Propagate with the 24 / 38 / 39-
dimensional motion block (ComposeMotion with imu_.Cg() / imu_.Ca(), the dWsb/dCg and dVsb/dCa columns of
ComputeMotionJacobianAt, src/estimator.cpp:598-704) and AbsorbError's retraction of td / Cg / Ca / the intrinsics
(src/estimator.cpp:875-890). All against the oracle, whose calibration code is pinned to the reference's own text
compiled with the three defines (tests/test_oracle_pinned.py)."""
import numpy as np
import pytest

import xivo_oracle as orc
from helpers import rel_fro, TOL_P, TOL_DX
from scene_util import scene_arrays, spd
from xivo_amd import synth
from xivo_amd.lib import Context, calib_dtype, cam_intr, imu_dtype, oos_dtype

pytestmark = pytest.mark.gpu
CAMS = {"pinhole": synth.PINHOLE, "equi": synth.EQUI, "radtan": synth.RADTAN, "atan": synth.ATAN}
CAM_DIM = {"pinhole": 4, "atan": 5, "radtan": 9, "equi": 8}
R_VIS, MH, MULT = 2.25, 5.991, 1.1


_ROWS_FLAGS = {"v": 0}


@pytest.fixture(params=["lead", "dense"])
def rows(request):
    """The two stackings of an online-calibration build: row-pair compressed rows + the leading dense block of the calibration
    columns on the sparse pipeline (round 5, default; last_path 1), or dense rows on the dense pipeline (the round-4 stacking:
    XIVO_HIP_FLAG_DENSE_H on the context)."""
    from xivo_amd.lib import FLAG_DENSE_H
    _ROWS_FLAGS["v"] = FLAG_DENSE_H if request.param == "dense" else 0
    yield 0 if request.param == "dense" else 1
    _ROWS_FLAGS["v"] = 0


def setup(name, temporal, imu, camera, B=3, ng=6, nf=14, seed=3, M_extra=0):
    cam = CAMS[name]
    lay = orc.calib_layout(ng, nf, temporal, imu, CAM_DIM[name] if camera else 0)
    sc = synth.g_level(ng, nf, nf, B, seed=seed, cam=cam)
    poses, groups, feats, xp = scene_arrays(sc, cam)
    rng = np.random.default_rng(seed)
    calib = np.zeros(B, dtype=calib_dtype)
    cals = []
    for b in range(B):
        cal = dict(gyro=rng.normal(size=3) * 0.5, Cg=np.eye(3) + 0.01 * rng.normal(size=(3, 3)), bg=rng.normal(size=3) * 0.01,
                   Vsb=rng.normal(size=3), td=0.01 + 0.005 * b)
        poses[b]["Vsb"], poses[b]["bg"] = cal["Vsb"], cal["bg"]
        calib[b]["gyro"], calib[b]["Cg"], calib[b]["td"] = cal["gyro"], cal["Cg"].T.reshape(-1), cal["td"]
        calib[b]["Ca"], calib[b]["intr"] = np.eye(3).reshape(-1), cam_intr(cam)
        cals.append(cal)
    ctx = Context(lay.N, 2 * nf + M_extra, B, flags=_ROWS_FLAGS["v"])
    ctx.set_layout(lay.N, lay.group_begin, ng, lay.feature_begin, nf, cam)
    ctx.set_calib(lay.td, lay.Cg, lay.cam_begin, lay.cam_dim)
    return cam, lay, sc, poses, groups, feats, xp, calib, cals, ctx


def oracle_rows(sc, cam, lay, xp, cals, b):
    Js, inns, Jcs = [], [], []
    for i in range(sc["x"].shape[1]):
        r = int(sc["ref"][b][i])
        J, inn, _, Jc = orc.compute_jacobian(sc["x"][b][i], xp[b][i], sc["gR"][b][r], sc["gT"][b][r], sc["Rsb"][b], sc["Tsb"][b],
                                            sc["Rbc"][b], sc["Tbc"][b], cam, lay, r, int(sc["sind"][b][i]), calib=cals[b])
        Js.append(J); inns.append(inn); Jcs.append(Jc)
    return np.array(Js), np.array(inns), np.array(Jcs)


@pytest.mark.parametrize("name", list(CAMS))
@pytest.mark.parametrize("temporal,imu,camera", [(True, True, True), (True, False, False), (False, False, True), (True, True, False)])
def test_calibration_jacobian_blocks_and_stacking(built, rows, name, temporal, imu, camera):
    cam, lay, sc, poses, groups, feats, xp, calib, cals, ctx = setup(name, temporal, imu, camera)
    B, F = poses.shape[0], feats.shape[1]
    with ctx:
        ctx.upload_P(np.array([spd(lay.N, 60 + b) * 1e-4 for b in range(B)]))
        ctx.set_scene(poses, groups, feats); ctx.set_calib_state(calib)
        ctx.jacobians_instate()
        J21, inn = ctx.get_jacobians()
        Jc = ctx.get_jacobians_calib(F=F)
        ctx.filter_update(R_VIS, MH, MULT, 5, use_gating=False)          # stacks every present feature
        assert ctx.last_path() == rows
        Hs = [ctx.get_H(b) for b in range(B)]                             # (the dense copy of the rows, rebuilt on demand)
    for b in range(B):
        Js, inns, Jcs = oracle_rows(sc, cam, lay, xp, cals, b)
        assert rel_fro(Jc[b], Jcs) < 1e-12 and np.abs(inn[b] - inns).max() < 1e-9
        if not temporal:
            assert not Jc[b][:, :, :13].any()
        if not camera:
            assert not Jc[b][:, :, 13:].any()
        H, innv, dR = orc.stack_measurements(Js, inns, sc["ref"][b], sc["sind"][b], lay, R_VIS)
        assert Hs[b][0].shape == H.shape and rel_fro(Hs[b][0], H) < 1e-12
        nzc = (np.abs(H).sum(0) > 0).sum()
        assert nzc >= 12 + (13 if temporal and imu else (4 if temporal else 0)) + (CAM_DIM[name] if camera else 0)


@pytest.mark.parametrize("name", ["pinhole", "equi", "radtan", "atan"])
def test_filter_update_with_calibration_columns(built, rows, name):
    """jac -> stack -> MH gating on the WHOLE row (as f->J(), update.cpp:60-70) -> UpdateJosephForm, against the
    reference flow where rejected features are not stacked: masks identical, P 1e-6, dx 1e-8 - the td / Cg / bg / intrinsics
    components of dx included (the host absorbs those, src/estimator.cpp:879-889)."""
    cam, lay, sc, poses, groups, feats, xp, calib, cals, ctx = setup(name, True, True, True, B=3, ng=8, nf=20, seed=11)
    feats["xp"][1, [2, 7]] += 60.0; xp[1, [2, 7]] += 60.0
    B, F = poses.shape[0], feats.shape[1]
    P = np.array([spd(lay.N, 80 + b) * 1e-4 for b in range(B)])
    with ctx:
        ctx.upload_P(P); ctx.set_scene(poses, groups, feats); ctx.set_calib_state(calib)
        ctx.filter_update(R_VIS, MH, MULT, 5, use_gating=True)
        assert ctx.last_path() == rows
        err = ctx.get_err(); Pn = ctx.download_P(); mask, dist = ctx.get_gate(F, B)
        assert (ctx.get_status() == 0).all()
    rejected = 0
    for b in range(B):
        Js, inns, _ = oracle_rows(sc, cam, lay, xp, cals, b)
        d = orc.mh_distances(Js, P[b], inns, R_VIS)
        m, _, _ = orc.mh_gate(d, MH, MULT, 5)
        assert np.array_equal(mask[b].astype(bool), m) and rel_fro(dist[b], d) < 1e-9
        rejected += int((~m).sum())
        idx = np.nonzero(m)[0]
        H, inn, dR = orc.stack_measurements(Js[idx], inns[idx], sc["ref"][b][idx], sc["sind"][b][idx], lay, R_VIS)
        e_ref, P_ref, _ = orc.update_joseph(H, P[b], inn, dR)
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX
        assert np.abs(e_ref[lay.td]) > 0 and np.abs(e_ref[lay.Cg:lay.Cg + 9]).max() > 0 and np.abs(e_ref[lay.cam_begin:lay.cam_begin + 4]).max() > 0
    assert rejected >= 2


def test_stand_alone_gate_and_calibration_off_again(built, rows):
    """xivo_hip_mh_gate on a calibration context gates on the whole row too (then stack + update as separate calls);
    set_calib() switches back to the default build."""
    cam, lay, sc, poses, groups, feats, xp, calib, cals, ctx = setup("equi", True, True, True, B=3, ng=6, nf=14, seed=4)
    feats["xp"][2, [1, 5]] += 50.0; xp[2, [1, 5]] += 50.0
    B, F = 3, feats.shape[1]
    P = np.array([spd(lay.N, 5 + b) * 1e-4 for b in range(B)])
    with ctx:
        ctx.upload_P(P); ctx.set_scene(poses, groups, feats); ctx.set_calib_state(calib)
        ctx.jacobians_instate()
        mask, dist = ctx.mh_gate(R_VIS, MH, MULT, 5)
        ctx.stack(R_VIS); ctx.update_joseph()
        assert ctx.last_path() == rows
        err = ctx.get_err(); Pn = ctx.download_P()
        rej = 0
        for b in range(B):
            Js, inns, _ = oracle_rows(sc, cam, lay, xp, cals, b)
            d = orc.mh_distances(Js, P[b], inns, R_VIS)
            m, _, _ = orc.mh_gate(d, MH, MULT, 5)
            assert np.array_equal(mask[b], m) and rel_fro(dist[b], d) < 1e-9
            rej += int((~m).sum())
            idx = np.nonzero(m)[0]
            H, inn, dR = orc.stack_measurements(Js[idx], inns[idx], sc["ref"][b][idx], sc["sind"][b][idx], lay, R_VIS)
            e_ref, P_ref, _ = orc.update_joseph(H, P[b], inn, dR)
            assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX
        assert rej >= 2
        ctx.set_calib()                                                                         # default build again
        ctx.upload_P(np.array([spd(lay.N, 5 + b) * 1e-4 for b in range(3)]))
        ctx.filter_update(R_VIS, MH, MULT, 5, use_gating=True)
        assert ctx.last_path() == rows                           # compressed rows, sparse pipeline (dense fixture: XIVO_HIP_FLAG_DENSE_H stays forced)


@pytest.mark.parametrize("name", ["radtan", "pinhole"])
def test_indefinite_S_of_a_calibration_stacking_takes_the_ldlt_fallback(built, rows, name):
    """A covariance that lost its definiteness under an online-calibration stacking: the L D L^T fallback forms S from the
    compressed rows AND the leading dense block of the calibration columns (rows == 1) or from the dense rows (rows == 0);
    P+ and dx equal the oracle's as-coded update, the regular filters of the batch are untouched by it."""
    cam, lay, sc, poses, groups, feats, xp, calib, cals, ctx = setup(name, True, True, True, B=4, ng=6, nf=14, seed=21)
    B = 4
    P = np.array([spd(lay.N, 300 + b) * 1e-4 for b in range(B)])
    w, Q = np.linalg.eigh(P[1]); w[-3:] *= -1.0
    P[1] = (Q * w) @ Q.T; P[1] = 0.5 * (P[1] + P[1].T)
    P[3] = -P[3]
    with ctx:
        ctx.upload_P(P); ctx.set_scene(poses, groups, feats); ctx.set_calib_state(calib)
        ctx.filter_update(R_VIS, MH, MULT, 5, use_gating=False)
        assert ctx.last_path() == rows
        st = ctx.get_status(check=False); used = ctx.get_ldlt_used()
        Pn, err = ctx.download_P(), ctx.get_err()
    assert (st == 0).all() and used.tolist() == [0, 1, 0, 1]
    for b in range(B):
        Js, inns, _ = oracle_rows(sc, cam, lay, xp, cals, b)
        H, inn, dR = orc.stack_measurements(Js, inns, sc["ref"][b], sc["sind"][b], lay, R_VIS)
        if b in (1, 3):
            assert np.linalg.eigvalsh(H @ P[b] @ H.T + np.diag(dR)).min() < 0
        e_ref, P_ref, _ = orc.update_joseph(H, P[b], inn, dR)
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX


def test_oos_rows_behind_a_calibration_stacking(built, rows):
    """OOS rows appended behind an online-calibration stacking: whatever form the in-state rows were stacked in, the update
    sees dense rows that carry the calibration columns themselves (re-stacked on demand) + the OOS block, dense pipeline."""
    ng, nf, n_oos, B = 8, 6, 3, 2
    cam, lay, sc, poses, groups, feats, xp, calib, cals, ctx = setup("pinhole", True, True, True, B=B, ng=ng, nf=nf, seed=5, M_extra=3 * 13)
    rng = np.random.default_rng(9)
    oos = np.zeros((B, n_oos), dtype=oos_dtype)
    obs_all = {}
    for b in range(B):
        for o in range(n_oos):
            k = [5, 2, 8][o]
            Xs = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(3, 6)])
            gs = rng.permutation(ng)[:k]
            oos[b, o]["Xs"] = Xs; oos[b, o]["n_obs"] = k
            obs = []
            for q, g in enumerate(gs):
                _, _, inn = orc.oos_jacobian_internal(Xs, sc["gR"][b, g], sc["gT"][b, g], sc["Rbc"][b], sc["Tbc"][b], [0, 0], cam, lay, int(g))
                pix = -inn + rng.normal(0, 1.0, 2)
                oos[b, o]["group_sind"][q] = g; oos[b, o]["xp"][q] = pix
                obs.append((int(g), pix))
            obs_all[b, o] = (Xs, obs)
    P = np.array([spd(lay.N, 40 + b) * 1e-4 for b in range(B)])
    with ctx:
        ctx.upload_P(P); ctx.set_scene(poses, groups, feats); ctx.set_calib_state(calib)
        ctx.jacobians_instate(); mask, _ = ctx.mh_gate(R_VIS, MH, MULT, 5); ctx.stack(R_VIS)
        nrows = ctx.oos_project(oos, 3.5 ** 2)
        ctx.update_joseph()
        assert ctx.last_path() == 0
        got = [ctx.get_H(b) for b in range(B)]
        err = ctx.get_err(); Pn = ctx.download_P()
    assert nrows.tolist() == [7 + 1 + 13] * B and mask.all()
    for b in range(B):
        Js, inns, _ = oracle_rows(sc, cam, lay, xp, cals, b)
        H, inn, dR = orc.stack_measurements(Js, inns, sc["ref"][b], sc["sind"][b], lay, R_VIS)
        for o in range(n_oos):
            Xs, obs = obs_all[b, o]
            Hxp, rp, _ = orc.oos_jacobian(Xs, obs, sc["gR"][b], sc["gT"][b], sc["Rbc"][b], sc["Tbc"][b], cam, lay)
            H = np.vstack([H, Hxp]); inn = np.concatenate([inn, rp]); dR = np.concatenate([dR, np.full(len(rp), 3.5 ** 2)])
        assert got[b][0].shape == H.shape and rel_fro(got[b][0], H) < 1e-10 and rel_fro(got[b][1], inn) < 1e-9
        assert np.abs(H[:2 * nf, lay.td]).max() > 0 and np.abs(got[b][0][:2 * nf, lay.Cg:lay.Cg + 9]).max() > 0
        e_ref, P_ref, _ = orc.update_joseph(H, P[b], inn, dR)
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX


def test_gated_update_on_a_calibration_stacking_gates_the_whole_rows(built, rows):
    """xivo_hip_update_dense_gated after xivo_hip_stack on an online-calibration context: the gate inside the update works on
    (H P) H^T of the WHOLE stacked rows - a stacking with the leading block is re-stacked densely for it (dense pipeline)."""
    cam, lay, sc, poses, groups, feats, xp, calib, cals, ctx = setup("equi", True, True, True, B=3, ng=6, nf=14, seed=14)
    feats["xp"][1, [3, 9]] += 55.0; xp[1, [3, 9]] += 55.0
    B, F = 3, feats.shape[1]
    P = np.array([spd(lay.N, 15 + b) * 1e-4 for b in range(B)])
    with ctx:
        ctx.upload_P(P); ctx.set_scene(poses, groups, feats); ctx.set_calib_state(calib)
        ctx.jacobians_instate()
        ctx.filter_update(R_VIS, MH, MULT, 5, use_gating=False)          # (leaves every present feature in the mask)
        ctx.upload_P(P)
        ctx.stack(R_VIS)
        ctx.update_dense_gated(F, R_VIS, MH, MULT, 5)
        assert ctx.last_path() == 0
        mask, dist = ctx.get_gate(F, B)
        err = ctx.get_err(); Pn = ctx.download_P()
    rej = 0
    for b in range(B):
        Js, inns, _ = oracle_rows(sc, cam, lay, xp, cals, b)
        H, inn, dR = orc.stack_measurements(Js, inns, sc["ref"][b], sc["sind"][b], lay, R_VIS)
        # the gate of the S level sees the rows AS STACKED (FillJacobianBlock's overwrite included), not J()
        d = np.array([inn[2 * f:2 * f + 2] @ np.linalg.solve(H[2 * f:2 * f + 2] @ P[b] @ H[2 * f:2 * f + 2].T + R_VIS * np.eye(2), inn[2 * f:2 * f + 2])
                      for f in range(F)])
        m, _, _ = orc.mh_gate(d, MH, MULT, 5)
        assert np.array_equal(mask[b].astype(bool), m) and rel_fro(dist[b], d) < 1e-9
        rej += int((~m).sum())
        keep = np.repeat(m, 2)
        e_ref, P_ref, _ = orc.update_joseph(H[keep], P[b], inn[keep], dR[keep])
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX
    assert rej >= 2


@pytest.mark.parametrize("name", ["radtan", "equi"])
def test_one_point_ransac_of_a_calibration_build(built, name):
    """Estimator::OnePointRANSAC (src/update.cpp:213-393) on an online-calibration context (round 5): the low-innovation
    set, the partial update on the WHOLE rows J() incl. the td / Cg / bg / intrinsics blocks, AbsorbError of td / Cg / Ca /
    intrinsics, Jacobians at the updated state (with the updated intrinsics), the whole-row chi-square rescue, RestoreState
    of everything incl. the calibration state - against the oracle, whose calibration branch is pinned to the reference's
    own text compiled with the three defines (tests/test_oracle_pinned.py)."""
    B, ng, nf = 8, 5, 14
    cam, lay, sc, poses, groups, feats, xp, calib, cals, ctx = setup(name, True, True, True, B=B, ng=ng, nf=nf, seed=31)
    rng = np.random.default_rng(8)
    xp = xp - sc["pix_noise"] + rng.normal(size=xp.shape) * 0.3
    gauge = np.zeros(B, dtype=np.int32)
    for b in range(B):
        kind = b % 4
        if kind in (0, 2):                                     # a few high-innovation features, one hopeless
            far = rng.choice(nf, size=4, replace=False)
            xp[b, far[:-1]] += rng.choice([-1, 1], size=(3, 2)) * rng.uniform(2.0, 4.0, size=(3, 2))
            xp[b, far[-1]] += 35.0
        elif kind == 3:                                        # nothing is low-innovation: rescue against the prior
            xp[b] += rng.choice([-1, 1], size=(nf, 2)) * rng.uniform(2.5, 3.5, size=(nf, 2))
        gauge[b] = -1 if kind == 2 else int(rng.integers(0, ng))
    feats["xp"] = xp
    P = np.array([spd(lay.N, 400 + b) * 1e-4 for b in range(B)])
    THRESH, CHI2, R1 = 2.0, 5.89, 1.0
    with ctx:
        ctx.upload_P(P); ctx.set_scene(poses, groups, feats); ctx.set_calib_state(calib)
        ctx.jacobians_instate()
        mh_mask, _ = ctx.mh_gate(R1, MH, MULT, 5)
        J0, inn0 = ctx.get_jacobians(); Jc0 = ctx.get_jacobians_calib(F=nf)
        keep, chi, nrej = ctx.one_point_ransac(R1, THRESH, CHI2, gauge=gauge)
        assert np.array_equal(ctx.download_P(), P)                                   # RestoreState
        p2, g2, _ = ctx.get_scene(); c2 = ctx.get_calib_state()
        assert np.array_equal(p2["Rsb"], poses["Rsb"]) and np.array_equal(g2["Tsb"], groups["Tsb"])
        assert np.array_equal(c2["intr"], calib["intr"]) and np.array_equal(c2["Cg"], calib["Cg"]) and np.array_equal(c2["td"], calib["td"])
        J1, inn1 = ctx.get_jacobians(); Jc1 = ctx.get_jacobians_calib(F=nf)
        assert np.array_equal(J0, J1) and np.array_equal(inn0, inn1) and np.array_equal(Jc0, Jc1)
        ctx.stack(R1); ctx.update_joseph()
        Pn, err = ctx.download_P(), ctx.get_err()
    seen = dict(partial=0, prior=0, rejected=0, rescued=0)
    for b in range(B):
        idx = np.nonzero(mh_mask[b])[0]
        st = dict(Rsb=sc["Rsb"][b].copy(), Tsb=sc["Tsb"][b].copy(), Vsb=cals[b]["Vsb"].copy(), bg=cals[b]["bg"].copy(), ba=np.zeros(3),
                  Rbc=sc["Rbc"][b].copy(), Tbc=sc["Tbc"][b].copy(), Rsg=np.eye(3), gR=sc["gR"][b].copy(), gT=sc["gT"][b].copy(),
                  x=sc["x"][b][idx].copy(), sind=sc["sind"][b][idx], ref=sc["ref"][b][idx], td=cals[b]["td"], Cg=cals[b]["Cg"].copy(),
                  Ca=np.eye(3), cam=dict(cam, d=list(cam.get("d", []))))
        out = orc.one_point_ransac(st, P[b], xp[b][idx], cam, lay, R1, THRESH, CHI2, int(gauge[b]), range(ng), calib_gyro=cals[b]["gyro"])
        exp = np.zeros(nf, dtype=bool); exp[idx[out["inliers"]]] = True
        assert np.array_equal(keep[b], exp), (b, keep[b], exp)
        assert nrej[b] == len(out["rejected"])
        for i, d in out["chi2"].items():
            assert abs(chi[b, idx[i]] - d) < 1e-7 * max(1.0, d), (b, i)
        low = out["low"]
        seen["prior"] += (not low.any()); seen["partial"] += (low.any() and not low.all())
        seen["rejected"] += len(out["rejected"]); seen["rescued"] += len(out["inliers"]) - int(low.sum())
        Js, inns, _ = oracle_rows(sc, cam, lay, xp, cals, b)
        kept = np.nonzero(exp)[0]
        H, inn, dR = orc.stack_measurements(Js[kept], inns[kept], sc["ref"][b][kept], sc["sind"][b][kept], lay, R1)
        e_ref, P_ref, _ = orc.update_joseph(H, P[b], inn, dR)
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX
    assert seen["partial"] >= 3 and seen["prior"] >= 1 and seen["rejected"] >= 1 and seen["rescued"] >= 3, seen


# ---- motion side ---------------------------------------------------------------------------------------------------------
def motion_setup(temporal, imu_cal, camera_dim, B=3, ng=4, nf=8, seed=5, cam=synth.RADTAN):
    lay = orc.calib_layout(ng, nf, temporal, imu_cal, camera_dim)
    sc = synth.g_level(ng, nf, nf, B, seed=seed, cam=cam)
    poses, groups, feats, xp = scene_arrays(sc, cam)
    rng = np.random.default_rng(seed + 100)
    calib = np.zeros(B, dtype=calib_dtype)
    st, Cgs, Cas = [], [], []
    for b in range(B):
        X = orc.MotionState(sc["Rsb"][b], sc["Tsb"][b], rng.normal(size=3) * 0.5, rng.normal(size=3) * 0.01,
                            rng.normal(size=3) * 0.05, orc.so3_exp(np.array([0.02, -0.03, 0.0])))
        st.append(X)
        poses[b]["Vsb"] = X.Vsb; poses[b]["bg"] = X.bg; poses[b]["ba"] = X.ba; poses[b]["Rsg"] = X.Rsg.T.reshape(-1)
        Cg = np.eye(3) + 0.02 * rng.normal(size=(3, 3)) if imu_cal else np.eye(3)
        Ca = np.triu(np.eye(3) + 0.02 * rng.normal(size=(3, 3))) if imu_cal else np.eye(3)
        Cgs.append(Cg); Cas.append(Ca)
        calib[b]["gyro"] = rng.normal(size=3) * 0.3; calib[b]["Cg"] = Cg.T.reshape(-1); calib[b]["Ca"] = Ca.T.reshape(-1)
        calib[b]["td"] = 0.004 * (b + 1); calib[b]["intr"] = cam_intr(cam)
    ctx = Context(lay.N, 2 * nf, B)
    ctx.set_layout(lay.N, lay.group_begin, ng, lay.feature_begin, nf, cam)
    ctx.set_calib(lay.td, lay.Cg, lay.cam_begin, lay.cam_dim)
    return lay, sc, poses, groups, feats, calib, st, Cgs, Cas, ctx


@pytest.mark.parametrize("method", ["RK4", "PrinceDormand"])
@pytest.mark.parametrize("temporal,imu_cal,camera_dim,motion", [(True, True, 9, 39), (True, False, 0, 24), (False, True, 0, 38),
                                                                (True, True, 0, 39)])
def test_propagate_with_calibration_columns(built, method, temporal, imu_cal, camera_dim, motion):
    """Two IMU samples per filter (sub-stepping incl. the half-step tail), kMotionSize = 39 / 24 / 38: nominal state and P
    against the oracle run sample by sample."""
    lay, sc, poses, groups, feats, calib, st, Cgs, Cas, ctx = motion_setup(temporal, imu_cal, camera_dim)
    assert lay.motion_size == motion
    B, N, K = poses.shape[0], lay.N, 2
    rng = np.random.default_rng(31)
    P = np.array([spd(N, 70 + b) * 1e-3 for b in range(B)])
    imu = np.zeros((B, K), dtype=imu_dtype)
    imu["gyro"] = rng.normal(size=(B, K, 3)) * 0.3; imu["accel"] = rng.normal(size=(B, K, 3)) + np.array([0, 0, 9.8])
    imu["slope_gyro"] = rng.normal(size=(B, K, 3)) * 5.0; imu["slope_accel"] = rng.normal(size=(B, K, 3)) * 20.0
    imu["dt"] = (0.0047 * (1.0 + 0.1 * np.arange(B)))[:, None]
    Qi = np.diag(rng.uniform(1e-6, 1e-4, 12)); A = rng.normal(size=(motion, motion)) * 1e-4; Qm = A @ A.T
    g = np.array([0.0, 0.0, -9.796])
    with ctx:
        ctx.upload_P(P); ctx.set_scene(poses, groups, feats); ctx.set_calib_state(calib)
        with pytest.raises(RuntimeError):
            ctx.propagate(imu, Qi, np.eye(23), g)            # the 23-dimensional entry refuses a calibration context
        ctx.propagate_calib(imu, Qi, Qm, g, method=method, stepsize=0.002)
        Pn = ctx.download_P()
        pose_d, _, _ = ctx.get_scene()
    for b in range(B):
        Xr, Pr = st[b], P[b]
        for k in range(K):
            Xr, Pr = orc.propagate(Xr, Pr, imu["gyro"][b, k], imu["accel"][b, k], imu["slope_gyro"][b, k], imu["slope_accel"][b, k],
                                   float(imu["dt"][b, k]), Qi, Qm, g, method=method, stepsize=0.002, Cg=Cgs[b], Ca=Cas[b], layout=lay)
        assert rel_fro(Pn[b], Pr) < 1e-11
        assert np.abs(pose_d[b]["Rsb"].reshape(3, 3).T - Xr.Rsb).max() < 1e-12
        assert np.abs(pose_d[b]["Tsb"] - Xr.Tsb).max() < 1e-12 and np.abs(pose_d[b]["Vsb"] - Xr.Vsb).max() < 1e-12
        if imu_cal:   # the Cg / Ca columns are really in the Jacobian the oracle integrated
            F1, _ = orc.motion_jacobian(st[b].Rsb, st[b].bg, st[b].ba, imu["gyro"][b, 0], imu["accel"][b, 0], g, Cgs[b], Cas[b], lay)
            assert np.abs(F1[0:3, lay.Cg:lay.Cg + 9]).max() > 0.01 and np.abs(F1[6:9, lay.Ca:lay.Ca + 6]).max() > 1.0


@pytest.mark.parametrize("temporal,imu_cal,motion", [(True, True, 39), (True, False, 24)])
def test_propagate_calib_step_size_control_as_coded(built, temporal, imu_cal, motion):
    """xivo_prop_opts.control_stepsize on an online-calibration context (kMotionSize 39 / 24): the step-size-controlled branch of
    Estimator::PrinceDormand (src/princedormand.cpp:26-60) as coded - samples of different lengths, two calls, the step carried
    per filter - against the oracle restatement (pinned to the extracted function: tests/test_oracle_pinned.py, golden_v7)."""
    lay, sc, poses, groups, feats, calib, st, Cgs, Cas, ctx = motion_setup(temporal, imu_cal, 0)
    assert lay.motion_size == motion
    B, N = poses.shape[0], lay.N
    rng = np.random.default_rng(77)
    P = np.array([spd(N, 170 + b) * 1e-3 for b in range(B)])
    dts = np.array([0.0025, 0.007, 0.0025, 0.012, 0.001])
    K = len(dts)
    imu = np.zeros((B, K), dtype=imu_dtype)
    imu["gyro"] = rng.normal(size=(B, K, 3)) * 0.3; imu["accel"] = rng.normal(size=(B, K, 3)) + np.array([0, 0, 9.8])
    imu["slope_gyro"] = rng.normal(size=(B, K, 3)) * 5.0; imu["slope_accel"] = rng.normal(size=(B, K, 3)) * 20.0
    imu["dt"] = dts[None, :] * (1.0 + 0.07 * np.arange(B))[:, None]
    Qi = np.diag(rng.uniform(1e-6, 1e-4, 12)); A = rng.normal(size=(motion, motion)) * 1e-4; Qm = A @ A.T
    g = np.array([0.0, 0.0, -9.796])
    ctlo = dict(tolerance=1e-3, attempts=12, min_scale_factor=0.125, max_scale_factor=4.0)
    with ctx:
        ctx.upload_P(P); ctx.set_scene(poses, groups, feats); ctx.set_calib_state(calib)
        ctx.propagate_calib(imu[:, :2], Qi, Qm, g, method="PrinceDormand", stepsize=0.002, pd_control=ctlo)
        ctx.propagate_calib(imu[:, 2:], Qi, Qm, g, method="PrinceDormand", stepsize=0.002, pd_control=ctlo)
        Pn = ctx.download_P()
        pose_d, _, _ = ctx.get_scene()
    for b in range(B):
        Xr, Pr = st[b], P[b]
        ctl = orc.PDControl(stepsize=0.002, **ctlo)
        for k in range(K):
            Xr, Pr = orc.propagate(Xr, Pr, imu["gyro"][b, k], imu["accel"][b, k], imu["slope_gyro"][b, k], imu["slope_accel"][b, k],
                                   float(imu["dt"][b, k]), Qi, Qm, g, method="PrinceDormand", stepsize=0.002, Cg=Cgs[b], Ca=Cas[b], layout=lay,
                                   pd_control=ctl)
        assert len(ctl.steps) > K                                       # (some sample took more than one step)
        assert rel_fro(Pn[b], Pr) < 1e-11
        assert np.abs(pose_d[b]["Rsb"].reshape(3, 3).T - Xr.Rsb).max() < 1e-12
        assert np.abs(pose_d[b]["Tsb"] - Xr.Tsb).max() < 1e-12 and np.abs(pose_d[b]["Vsb"] - Xr.Vsb).max() < 1e-12


def test_absorb_error_retracts_the_calibration_state(built):
    """xivo_hip_absorb_error on an online-calibration context: td, Ca's upper triangle, Cg, the nine radtan intrinsics move by
    their dx components (src/core.h:150-152, src/imu.cpp:7-21, common/camera_autocalib.h:34-47); the Jacobians that follow use
    the filter's own intrinsics."""
    cam = synth.RADTAN
    lay, sc, poses, groups, feats, calib, st, Cgs, Cas, ctx = motion_setup(True, True, 9, cam=cam)
    B, N, F = poses.shape[0], lay.N, feats.shape[1]
    rng = np.random.default_rng(8)
    xp = np.array([[feats[b][i]["xp"] for i in range(F)] for b in range(B)])
    with ctx:
        ctx.upload_P(np.array([spd(N, 90 + b) * 1e-6 for b in range(B)]))
        ctx.set_scene(poses, groups, feats); ctx.set_calib_state(calib)
        ctx.filter_update(R_VIS, MH, MULT, 5, use_gating=False)
        dx = ctx.get_err()
        ctx.absorb_error()
        cal_d = ctx.get_calib_state()
        pose_d, _, feat_d = ctx.get_scene()
        ctx.jacobians_instate()
        J21, inn = ctx.get_jacobians()
    for b in range(B):
        sto = dict(Rsb=sc["Rsb"][b].copy(), Tsb=sc["Tsb"][b].copy(), Vsb=st[b].Vsb.copy(), bg=st[b].bg.copy(), ba=st[b].ba.copy(),
                   Rbc=sc["Rbc"][b].copy(), Tbc=sc["Tbc"][b].copy(), Rsg=st[b].Rsg.copy(), td=float(calib[b]["td"]), Cg=Cgs[b].copy(),
                   Ca=Cas[b].copy(), cam=dict(cam, d=list(cam["d"])), gR=sc["gR"][b].copy(), gT=sc["gT"][b].copy(),
                   x=sc["x"][b].copy(), sind=sc["sind"][b])
        orc.absorb_error(sto, dx[b], lay, range(lay.n_groups), range(F))
        assert abs(cal_d[b]["td"] - sto["td"]) < 1e-15 and abs(dx[b][lay.td]) > 0
        assert np.abs(cal_d[b]["Cg"].reshape(3, 3).T - sto["Cg"]).max() < 1e-15
        assert np.abs(cal_d[b]["Ca"].reshape(3, 3).T - sto["Ca"]).max() < 1e-15
        assert np.abs(np.tril(cal_d[b]["Ca"].reshape(3, 3).T, -1)).max() == 0
        assert np.abs(cal_d[b]["intr"] - cam_intr(sto["cam"])).max() < 1e-12
        assert np.abs(cal_d[b]["intr"] - cam_intr(cam)).max() > 0
        assert np.abs(pose_d[b]["Tsb"] - sto["Tsb"]).max() < 1e-12
        # the next linearisation projects with the filter's retracted intrinsics
        cal = dict(gyro=calib[b]["gyro"], Cg=sto["Cg"], bg=sto["bg"], Vsb=sto["Vsb"], td=sto["td"])
        for i in range(0, F, 3):
            r = int(sc["ref"][b][i])
            J, inn_o, _, _ = orc.compute_jacobian(sto["x"][i], xp[b][i], sto["gR"][r], sto["gT"][r], sto["Rsb"], sto["Tsb"], sto["Rbc"],
                                                  sto["Tbc"], sto["cam"], lay, r, int(sc["sind"][b][i]), calib=cal)
            assert np.abs(inn[b][i] - inn_o).max() < 1e-8


def test_resident_loop_online_calibration(built):
    """Three frames of Propagate -> ComputeInstateJacobians -> MHGating -> FilterUpdate -> AbsorbError of the full
    online-calibration build (kMotionSize 39 + 9 radtan intrinsics), everything resident: P, the nominal state, td / Cg / Ca /
    the intrinsics; only the IMU samples and last_gyro_ cross the boundary. Against the oracle run frame by frame."""
    cam = synth.RADTAN
    lay, sc, poses, groups, feats, calib, st, Cgs, Cas, ctx = motion_setup(True, True, 9, B=2, ng=5, nf=12, seed=9, cam=cam)
    B, N, F, G = poses.shape[0], lay.N, feats.shape[1], lay.n_groups
    rng = np.random.default_rng(17)
    nm = lay.motion_size
    P0 = np.array([spd(N, 40 + b) * 1e-6 for b in range(B)])
    Qi = np.diag(rng.uniform(1e-6, 1e-4, 12)); A = rng.normal(size=(nm, nm)) * 1e-5; Qm = A @ A.T
    g = np.array([0.0, 0.0, -9.796])
    xp = np.array([[feats[b][i]["xp"] for i in range(F)] for b in range(B)])
    frames = []
    for k in range(3):
        imu = np.zeros((B, 2), dtype=imu_dtype)
        imu["gyro"] = rng.normal(size=(B, 2, 3)) * 0.05; imu["accel"] = rng.normal(size=(B, 2, 3)) * 0.1 + np.array([0, 0, 9.8])
        imu["slope_gyro"] = rng.normal(size=(B, 2, 3)); imu["slope_accel"] = rng.normal(size=(B, 2, 3))
        imu["dt"] = 0.0025
        frames.append((imu, rng.normal(size=(B, 3)) * 0.3))
    with ctx:
        ctx.upload_P(P0); ctx.set_scene(poses, groups, feats); ctx.set_calib_state(calib)
        masks = []
        for imu, gyro in frames:
            ctx.set_calib_gyro(gyro)
            ctx.propagate_calib(imu, Qi, Qm, g, method="PrinceDormand", stepsize=0.002)
            ctx.filter_update(R_VIS, MH, MULT, 5, use_gating=True)
            assert (ctx.get_status() == 0).all()
            masks.append(ctx.get_gate(F, B)[0].astype(bool))
            ctx.absorb_error()
        Pn = ctx.download_P(); cal_d = ctx.get_calib_state(); pose_d, grp_d, feat_d = ctx.get_scene()
    for b in range(B):
        s = dict(Rsb=sc["Rsb"][b].copy(), Tsb=sc["Tsb"][b].copy(), Vsb=st[b].Vsb.copy(), bg=st[b].bg.copy(), ba=st[b].ba.copy(),
                 Rbc=sc["Rbc"][b].copy(), Tbc=sc["Tbc"][b].copy(), Rsg=st[b].Rsg.copy(), td=float(calib[b]["td"]), Cg=Cgs[b].copy(),
                 Ca=Cas[b].copy(), cam=dict(cam, d=list(cam["d"])), gR=sc["gR"][b].copy(), gT=sc["gT"][b].copy(), x=sc["x"][b].copy(),
                 sind=sc["sind"][b])
        P = P0[b]
        for k, (imu, gyro) in enumerate(frames):
            X = orc.MotionState(s["Rsb"], s["Tsb"], s["Vsb"], s["bg"], s["ba"], s["Rsg"])
            for j in range(2):
                X, P = orc.propagate(X, P, imu["gyro"][b, j], imu["accel"][b, j], imu["slope_gyro"][b, j], imu["slope_accel"][b, j], 0.0025,
                                     Qi, Qm, g, method="PrinceDormand", stepsize=0.002, Cg=s["Cg"], Ca=s["Ca"], layout=lay)
            s["Rsb"], s["Tsb"], s["Vsb"] = X.Rsb, X.Tsb, X.Vsb
            cal = dict(gyro=gyro[b], Cg=s["Cg"], bg=s["bg"], Vsb=s["Vsb"], td=s["td"])
            Js, inns = [], []
            for i in range(F):
                r = int(sc["ref"][b][i])
                J, inn, _, _ = orc.compute_jacobian(s["x"][i], xp[b][i], s["gR"][r], s["gT"][r], s["Rsb"], s["Tsb"], s["Rbc"], s["Tbc"],
                                                    s["cam"], lay, r, int(sc["sind"][b][i]), calib=cal)
                Js.append(J); inns.append(inn)
            Js, inns = np.array(Js), np.array(inns)
            m, _, _ = orc.mh_gate(orc.mh_distances(Js, P, inns, R_VIS), MH, MULT, 5)
            assert np.array_equal(masks[k][b], m)
            idx = np.nonzero(m)[0]
            H, inn, dR = orc.stack_measurements(Js[idx], inns[idx], sc["ref"][b][idx], sc["sind"][b][idx], lay, R_VIS)
            e, P, _ = orc.update_joseph(H, P, inn, dR)
            orc.absorb_error(s, e, lay, range(G), idx)
        assert rel_fro(Pn[b], P) < TOL_P
        assert np.abs(pose_d[b]["Rsb"].reshape(3, 3).T - s["Rsb"]).max() < 1e-9 and np.abs(pose_d[b]["Tsb"] - s["Tsb"]).max() < 1e-9
        assert np.abs(pose_d[b]["Vsb"] - s["Vsb"]).max() < 1e-9 and np.abs(pose_d[b]["bg"] - s["bg"]).max() < 1e-9
        assert abs(cal_d[b]["td"] - s["td"]) < 1e-9 and np.abs(cal_d[b]["Cg"].reshape(3, 3).T - s["Cg"]).max() < 1e-9
        assert np.abs(cal_d[b]["Ca"].reshape(3, 3).T - s["Ca"]).max() < 1e-9
        assert np.abs(cal_d[b]["intr"] - cam_intr(s["cam"])).max() < 1e-6
        assert np.abs(cal_d[b]["intr"] - cam_intr(cam)).max() > 1e-9 and abs(cal_d[b]["td"] - calib[b]["td"]) > 1e-12
        for i in range(F):
            assert np.abs(feat_d[b][i]["x"] - s["x"][i]).max() < 1e-8
