"""Online-calibration builds, measurement side (xivo_hip_set_calib): the td / Cg / bg / intrinsics blocks of
Feature::ComputeJacobian (src/feature.cpp:592-609, :611-618, :632-651), their stacking by Feature::FillJacobianBlock
(:664-670, :679-683), MH gating on the whole row and the update - against the oracle, whose calibration blocks are pinned
to the reference's own text compiled with the three defines (tests/test_oracle_pinned.py)."""
import numpy as np
import pytest

import xivo_oracle as orc
from helpers import rel_fro, TOL_P, TOL_DX
from scene_util import scene_arrays, spd
from xivo_amd import synth
from xivo_amd.lib import Context, calib_dtype

pytestmark = pytest.mark.gpu
CAMS = {"pinhole": synth.PINHOLE, "equi": synth.EQUI, "radtan": synth.RADTAN, "atan": synth.ATAN}
CAM_DIM = {"pinhole": 4, "atan": 5, "radtan": 9, "equi": 8}
R_VIS, MH, MULT = 2.25, 5.991, 1.1


def setup(name, temporal, imu, camera, B=3, ng=6, nf=14, seed=3):
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
        cals.append(cal)
    ctx = Context(lay.N, 2 * nf, B)
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
def test_calibration_jacobian_blocks_and_stacking(built, name, temporal, imu, camera):
    cam, lay, sc, poses, groups, feats, xp, calib, cals, ctx = setup(name, temporal, imu, camera)
    B, F = poses.shape[0], feats.shape[1]
    with ctx:
        ctx.upload_P(np.array([spd(lay.N, 60 + b) * 1e-4 for b in range(B)]))
        ctx.set_scene(poses, groups, feats); ctx.set_calib_state(calib)
        ctx.jacobians_instate()
        J21, inn = ctx.get_jacobians()
        Jc = ctx.get_jacobians_calib(F=F)
        ctx.filter_update(R_VIS, MH, MULT, 5, use_gating=False)          # stacks every present feature (dense rows)
        Hs = [ctx.get_H(b) for b in range(B)]
        assert ctx.last_path() == 0
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
def test_filter_update_with_calibration_columns(built, name):
    """jac -> stack (dense rows) -> MH gating on the WHOLE row (as f->J(), update.cpp:60-70) -> UpdateJosephForm, against the
    reference flow where rejected features are not stacked: masks identical, P 1e-6, dx 1e-8 - the td / Cg / bg / intrinsics
    components of dx included (the host absorbs those, src/estimator.cpp:879-889)."""
    cam, lay, sc, poses, groups, feats, xp, calib, cals, ctx = setup(name, True, True, True, B=3, ng=8, nf=20, seed=11)
    feats["xp"][1, [2, 7]] += 60.0; xp[1, [2, 7]] += 60.0
    B, F = poses.shape[0], feats.shape[1]
    P = np.array([spd(lay.N, 80 + b) * 1e-4 for b in range(B)])
    with ctx:
        ctx.upload_P(P); ctx.set_scene(poses, groups, feats); ctx.set_calib_state(calib)
        ctx.filter_update(R_VIS, MH, MULT, 5, use_gating=True)
        assert ctx.last_path() == 0
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


def test_calibration_off_again_and_unsupported_entries(built):
    cam, lay, sc, poses, groups, feats, xp, calib, cals, ctx = setup("pinhole", True, True, True)
    with ctx:
        ctx.set_scene(poses, groups, feats); ctx.set_calib_state(calib)
        ctx.jacobians_instate()
        assert ctx.lib.xivo_hip_mh_gate(ctx.h, 3, 2.25, 5.991, 1.1, 5, None, None) == -5      # gate on the compact 21 columns: not in this mode
        ctx.set_calib()                                                                         # default build again
        ctx.upload_P(np.array([spd(lay.N, 5 + b) * 1e-4 for b in range(3)]))
        ctx.filter_update(R_VIS, MH, MULT, 5, use_gating=True)
        assert ctx.last_path() == 1                                                             # compressed rows, sparse pipeline
