"""GPU parity of the layout-faithful ("G-level") path through the C ABI:
Feature::ComputeJacobian, Estimator::MHGating, FilterUpdate stacking incl. the
FillJacobianBlock quirk, OOS null-space rows, P edits and the propagation tail."""
import numpy as np
import pytest

import xivo_oracle as orc
from helpers import rel_fro, TOL_P, TOL_DX
from scene_util import scene_arrays, oracle_jacobians, spd
from xivo_amd import synth
from xivo_amd.lib import Context, FLAG_FIX_GROUP_BLOCK, oos_dtype

pytestmark = pytest.mark.gpu
CAMS = {"pinhole": synth.PINHOLE, "equi": synth.EQUI, "radtan": synth.RADTAN, "atan": synth.ATAN}
R_VIS, MH, MULT = 2.25, 5.991, 1.1


def make(ng, nf, F, B, seed, cam, N=None, flags=0, M_max=None):
    sc = synth.g_level(ng, nf, F, B, seed=seed, cam=cam, N=N)
    lay = orc.Layout(ng, nf, N=sc["N"])
    ctx = Context(lay.N, M_max or 2 * F, B, flags=flags)
    ctx.set_layout(lay.N, lay.group_begin, ng, lay.feature_begin, nf, cam)
    poses, groups, feats, xp = scene_arrays(sc, cam)
    return sc, lay, ctx, poses, groups, feats, xp


@pytest.mark.parametrize("name", list(CAMS))
def test_instate_jacobians(built, name):
    cam = CAMS[name]
    sc, lay, ctx, poses, groups, feats, xp = make(5, 12, 12, 3, 1, cam)
    with ctx:
        ctx.set_scene(poses, groups, feats)
        ctx.jacobians_instate()
        J, inn = ctx.get_jacobians()
    for b in range(3):
        _, inns, blocks = oracle_jacobians(sc, cam, lay, xp, b)
        Jo = np.concatenate([blocks[:, k] for k in range(7)], axis=2)  # [F, 2, 21]
        assert rel_fro(J[b], Jo) < 1e-12
        assert np.abs(inn[b] - inns).max() < 1e-9


def test_mh_gating_distances_mask_and_relaxation(built):
    cam = synth.PINHOLE
    sc, lay, ctx, poses, groups, feats, xp = make(6, 20, 20, 4, 2, cam)
    # outliers: filter 1 gets three wild pixels; filter 2 gets almost all wild (forces relaxation)
    feats["xp"][1, [2, 7, 11]] += 40.0
    xp[1, [2, 7, 11]] += 40.0
    feats["xp"][2, 3:] += np.linspace(6, 60, 17)[:, None]
    xp[2, 3:] += np.linspace(6, 60, 17)[:, None]
    P = np.array([spd(lay.N, 10 + b) * 1e-4 for b in range(4)])
    with ctx:
        ctx.upload_P(P)
        ctx.set_scene(poses, groups, feats)
        ctx.jacobians_instate()
        mask, dist = ctx.mh_gate(R_VIS, MH, MULT, 5)
    for b in range(4):
        Js, inns, _ = oracle_jacobians(sc, cam, lay, xp, b)
        d = orc.mh_distances(Js, P[b], inns, R_VIS)
        m, _, _ = orc.mh_gate(d, MH, MULT, 5)
        assert rel_fro(dist[b], d) < 1e-9
        assert np.array_equal(mask[b], m)
    assert mask[0].all() and (~mask[1]).sum() == 3 and mask[2].sum() >= 5 and (~mask[2]).sum() > 0


@pytest.mark.parametrize("fix", [False, True])
def test_stacking_with_and_without_fill_quirk(built, fix):
    cam = synth.EQUI
    sc, lay, ctx, poses, groups, feats, xp = make(4, 9, 9, 2, 3, cam, flags=FLAG_FIX_GROUP_BLOCK if fix else 0)
    feats["xp"][0, 4] += 80.0; xp[0, 4] += 80.0
    P = np.array([spd(lay.N, 20 + b) * 1e-4 for b in range(2)])
    with ctx:
        ctx.upload_P(P); ctx.set_scene(poses, groups, feats)
        ctx.jacobians_instate()
        mask, _ = ctx.mh_gate(R_VIS, MH, MULT, 5)
        ctx.stack(R_VIS)
        got = [ctx.get_H(b) for b in range(2)]
    assert not mask[0, 4]
    for b in range(2):
        Js, inns, _ = oracle_jacobians(sc, cam, lay, xp, b)
        H, inn, dR = orc.stack_measurements(Js, inns, sc["ref"][b], sc["sind"][b], lay, R_VIS, fix_group_block=fix)
        H, inn, dR = orc.neutralise_rows(H, inn, dR, np.repeat(mask[b], 2))
        assert rel_fro(got[b][0], H) < 1e-12 and np.abs(got[b][1] - inn).max() < 1e-9
        assert np.array_equal(got[b][2], dR)


@pytest.mark.parametrize("flags,path", [(0, 1), (FLAG_FIX_GROUP_BLOCK, 1), (64, 0)])   # 64 = XIVO_HIP_FLAG_DENSE_H
@pytest.mark.parametrize("name,N", [("pinhole", None), ("equi", 251)])
def test_filter_update_equals_reference_over_inliers_only(built, name, N, flags, path):
    """jac -> gate -> stack -> UpdateJosephForm on device == the reference flow where
    rejected features are simply not stacked (update.cpp:105-141); through the compressed rows emitted by the
    stack kernel (with and without the FillJacobianBlock quirk) and through the dense pipeline."""
    cam = CAMS[name]
    ng, nf, F, B = (8, 60, 60, 3) if N else (5, 14, 14, 3)
    sc, lay, ctx, poses, groups, feats, xp = make(ng, nf, F, B, 4, cam, N=N, flags=flags)
    feats["xp"][1, [0, 5]] += 55.0; xp[1, [0, 5]] += 55.0
    P = np.array([spd(lay.N, 30 + b) * 1e-4 for b in range(B)])
    with ctx:
        ctx.upload_P(P); ctx.set_scene(poses, groups, feats)
        ctx.filter_update(R_VIS, MH, MULT, 5, use_gating=True)
        assert ctx.last_path() == path
        err = ctx.get_err(); Pn = ctx.download_P()
        assert (ctx.get_status() == 0).all()
    rejected = 0
    for b in range(B):
        Js, inns, _ = oracle_jacobians(sc, cam, lay, xp, b)
        m, _, _ = orc.mh_gate(orc.mh_distances(Js, P[b], inns, R_VIS), MH, MULT, 5)
        idx = np.nonzero(m)[0]
        rejected += int((~m).sum())
        H, inn, dR = orc.stack_measurements(Js[idx], inns[idx], sc["ref"][b][idx], sc["sind"][b][idx], lay, R_VIS,
                                            fix_group_block=bool(flags & FLAG_FIX_GROUP_BLOCK))
        e_ref, P_ref, _ = orc.update_joseph(H, P[b], inn, dR)
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX
    assert rejected >= 2


def test_oos_rows_match_slow_givens(built):
    cam = synth.PINHOLE
    ng, nf, F, B, n_oos = 8, 6, 6, 2, 3
    sc, lay, ctx, poses, groups, feats, xp = make(ng, nf, F, B, 5, cam, M_max=2 * F + 3 * 13)
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
                hf, hx, inn = orc.oos_jacobian_internal(Xs, sc["gR"][b, g], sc["gT"][b, g], sc["Rbc"][b], sc["Tbc"][b],
                                                        [0, 0], cam, lay, int(g))
                pix = -inn + rng.normal(0, 1.0, 2)
                oos[b, o]["group_sind"][q] = g; oos[b, o]["xp"][q] = pix
                obs.append((int(g), pix))
            obs_all[b, o] = (Xs, obs)
    P = np.array([spd(lay.N, 40 + b) * 1e-4 for b in range(B)])
    with ctx:
        ctx.upload_P(P); ctx.set_scene(poses, groups, feats)
        ctx.jacobians_instate(); ctx.mh_gate(R_VIS, MH, MULT, 5); ctx.stack(R_VIS)
        rows = ctx.oos_project(oos, 3.5 ** 2)
        got = [ctx.get_H(b) for b in range(B)]
        ctx.update_joseph()
        err = ctx.get_err(); Pn = ctx.download_P()
    assert rows.tolist() == [7 + 1 + 13] * B
    for b in range(B):
        Js, inns, _ = oracle_jacobians(sc, cam, lay, xp, b)
        H, inn, dR = orc.stack_measurements(Js, inns, sc["ref"][b], sc["sind"][b], lay, R_VIS)
        for o in range(n_oos):
            Xs, obs = obs_all[b, o]
            Hxp, rp, A = orc.oos_jacobian(Xs, obs, sc["gR"][b], sc["gT"][b], sc["Rbc"][b], sc["Tbc"][b], cam, lay)
            H = np.vstack([H, Hxp]); inn = np.concatenate([inn, rp]); dR = np.concatenate([dR, np.full(len(rp), 3.5 ** 2)])
        assert got[b][0].shape == H.shape
        assert rel_fro(got[b][0], H) < 1e-10 and rel_fro(got[b][1], inn) < 1e-9 and np.allclose(got[b][2], dR)
        e_ref, P_ref, _ = orc.update_joseph(H, P[b], inn, dR)
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX


def test_oos_rows_whole_buffer_mode_as_coded(built):
    """XIVO_HIP_OOS_WHOLE_BUFFER (xivo_hip_oos_project_ex): src/oos.cpp:28 as coded hands SlowGivens the whole 2 kMaxGroup-row
    buffers - every feature contributes 2 kMaxGroup - 3 rows. Against the oracle's FullPivLU kernel of the zero-padded
    buffers: the same rows, inn, diagR; and the update (K, dx, P+) equals the default call's (the extra rows are zero)."""
    cam = synth.PINHOLE
    ng, nf, F, B, n_oos = 8, 6, 6, 2, 3
    sc, lay, ctx, poses, groups, feats, xp = make(ng, nf, F, B, 5, cam, M_max=2 * F + 3 * 13)
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
    out = {}
    for whole in (False, True):
        with make(ng, nf, F, B, 5, cam, M_max=2 * F + 3 * 13)[2] as c2:
            c2.upload_P(P); c2.set_scene(poses, groups, feats)
            c2.jacobians_instate(); c2.mh_gate(R_VIS, MH, MULT, 5); c2.stack(R_VIS)
            rows = c2.oos_project(oos, 3.5 ** 2, whole_buffer=whole)
            got = [c2.get_H(b) for b in range(B)]
            c2.update_joseph()
            out[whole] = (rows.tolist(), got, c2.get_err(), c2.download_P())
    ctx.close()
    assert out[False][0] == [7 + 1 + 13] * B and out[True][0] == [3 * (2 * ng - 3)] * B
    for b in range(B):
        Js, inns, _ = oracle_jacobians(sc, cam, lay, xp, b)
        H, inn, dR = orc.stack_measurements(Js, inns, sc["ref"][b], sc["sind"][b], lay, R_VIS)
        for o in range(n_oos):
            Xs, obs = obs_all[b, o]
            Hxp, rp, A = orc.oos_jacobian(Xs, obs, sc["gR"][b], sc["gT"][b], sc["Rbc"][b], sc["Tbc"][b], cam, lay, whole_buffer_groups=ng)
            assert Hxp.shape[0] == 2 * ng - 3
            H = np.vstack([H, Hxp]); inn = np.concatenate([inn, rp]); dR = np.concatenate([dR, np.full(len(rp), 3.5 ** 2)])
        gH, ginn, gdR = out[True][1][b]
        assert gH.shape == H.shape
        assert rel_fro(gH, H) < 1e-10 and rel_fro(ginn, inn) < 1e-9 and np.array_equal(gdR, dR)
        e_ref, P_ref, _ = orc.update_joseph(H, P[b], inn, dR)
        assert rel_fro(out[True][3][b], P_ref) < TOL_P and rel_fro(out[True][2][b], e_ref) < TOL_DX
        assert rel_fro(out[True][3][b], out[False][3][b]) < 1e-12 and rel_fro(out[True][2][b], out[False][2][b]) < 1e-10


def test_p_edits_and_snapshot(built):
    N, B = 113, 3
    P = np.array([spd(N, 50 + b) for b in range(B)])
    with Context(N, 20, B) as ctx:
        ctx.upload_P(P)
        ctx.snapshot_P()
        ctx.p_zero_rc(1, 29, 6)                     # RemoveGroupFromState (estimator.cpp:757-759)
        ctx.p_copy_rc(2, 35, 0, 3); ctx.p_copy_rc(2, 38, 3, 3)   # AddGroupToState (estimator.cpp:808-816)
        P3 = np.array([[1.0, 0.1, 0.2], [0.1, 2.0, 0.3], [0.2, 0.3, 3.0]])
        ctx.p_zero_rc(0, 53, 3); ctx.p_set_block3(0, 53, P3)     # FillCovarianceBlock (feature.cpp:753-760)
        got = ctx.download_P()
        d = ctx.p_diag(0)
        ctx.restore_P()
        back = ctx.download_P()
    assert np.array_equal(got[1], orc.p_zero_rc(P[1], 29, 6))
    assert np.array_equal(got[2], orc.p_copy_rc(orc.p_copy_rc(P[2], 35, 0, 3), 38, 3, 3))
    exp0 = orc.p_set_block3(orc.p_zero_rc(P[0], 53, 3), 53, P3)
    assert np.array_equal(got[0], exp0) and np.array_equal(d, np.diag(exp0))
    assert np.array_equal(back, P)


def test_propagation_tail(built):
    N, nm, B = 203, 23, 3
    rng = np.random.default_rng(3)
    P = np.array([spd(N, 60 + b) for b in range(B)])
    FK = rng.normal(size=(B, nm, nm)); PK = rng.normal(size=(B, nm, nm)); PK = PK + np.transpose(PK, (0, 2, 1))
    Q = np.diag(rng.uniform(1e-6, 1e-4, nm))
    dt = 0.002
    exp, Phi, Pmm = [], [], []
    for b in range(B):
        Pn, ph = orc.rk4_cov_tail(P[b], FK[b], PK[b], dt, Q)
        exp.append(Pn); Phi.append(ph); Pmm.append(Pn[:nm, :nm])
    with Context(N, 20, B) as ctx:
        ctx.upload_P(P)
        ctx.propagate_cov(np.array(Phi), np.array(Pmm))
        got = ctx.download_P()
    for b in range(B):
        assert rel_fro(got[b], exp[b]) < 1e-14


# 512 = STANDALONE_TAIL with OOS rows stacked: its G = T H^T + K R walks compressed rows that - with mixed stacking - hold the
# in-state rows only: such a call takes dense rows instead (round 4; it used to return a silently wrong P+). 64 = DENSE_H: dense rows
# stacked at once, as-coded sequence. 256 = SYMMETRIC_FORM on the mixed stacking.
@pytest.mark.parametrize("compress,flags", [(False, 0), (True, 0), (True, 64), (True, 512), (False, 512), (True, 256), (True, 8192)])
def test_config3_full_size_instate_plus_oos(built, compress, flags):
    """BASELINE.json config 3 at full size: N=251 (8 groups, 60 in-state features -> 120 rows)
    + 20 OOS features seen from k=5 groups (7 projected rows each -> 140 rows), M=260.
    compress: the 140 OOS rows (non-zero only over the 6 extrinsics + 48 group columns) are replaced by the 54-row
    triangular factor of their QR decomposition before the update (measurement compression, src/estimator.h:399-402,
    src/helpers.cpp:77-101): M = 174; K, dx, P+ are unchanged to rounding."""
    cam = synth.PINHOLE
    ng, nf, F, B, n_oos, k = 8, 60, 60, 2, 20, 5
    # flags 0 (round 3): mixed stacking - the in-state rows stay row-pair compressed, only the OOS block is dense (the 16
    # spare rows of the allocation are what the 16-row-padded OOS block needs behind row 120)
    sc, lay, ctx, poses, groups, feats, xp = make(ng, nf, F, B, 6, cam, M_max=2 * F + n_oos * (2 * k - 3) + 16, flags=flags)
    assert lay.N == 251
    rng = np.random.default_rng(19)
    oos = np.zeros((B, n_oos), dtype=oos_dtype)
    obs_all = {}
    for b in range(B):
        for o in range(n_oos):
            Xs = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(3, 6)])
            gs = rng.permutation(ng)[:k]
            oos[b, o]["Xs"] = Xs; oos[b, o]["n_obs"] = k
            obs = []
            for q, g in enumerate(gs):
                _, _, inn = orc.oos_jacobian_internal(Xs, sc["gR"][b, g], sc["gT"][b, g], sc["Rbc"][b], sc["Tbc"][b], [0, 0],
                                                      cam, lay, int(g))
                pix = -inn + rng.normal(0, 1.0, 2)
                oos[b, o]["group_sind"][q] = g; oos[b, o]["xp"][q] = pix
                obs.append((int(g), pix))
            obs_all[b, o] = (Xs, obs)
    P = np.array([spd(lay.N, 70 + b) * 1e-4 for b in range(B)])
    with ctx:
        ctx.upload_P(P); ctx.set_scene(poses, groups, feats)
        ctx.jacobians_instate(); ctx.mh_gate(R_VIS, MH, MULT, 5); ctx.stack(R_VIS)
        rows = ctx.oos_project(oos, 3.5 ** 2)
        if compress:
            Hfull = [ctx.get_H(b) for b in range(B)]
            crow = ctx.compress_oos(1.5)
            assert crow.tolist() == [54] * B
            for b in range(B):
                Hc, ic, rc_ = ctx.get_H(b)
                Hf, if_, rf = Hfull[b]
                assert Hc.shape[0] == 174 and np.array_equal(Hc[:120], Hf[:120]) and np.array_equal(ic[:120], if_[:120])
                A, C = Hf[120:], Hc[120:]
                assert np.all(rc_[120:] == 3.5 ** 2)
                # an orthogonal transform of the block: same Gram matrix, same H^T r; upper-trapezoidal over its columns
                assert rel_fro(C.T @ C, A.T @ A) < 1e-12 and rel_fro(C.T @ ic[120:], A.T @ if_[120:]) < 1e-12
                cols = np.nonzero(np.abs(A).sum(0))[0]
                assert len(cols) == 54 and np.array_equal(np.nonzero(np.abs(C).sum(0))[0], cols)
                assert np.count_nonzero(np.tril(C[:, cols], -1)) == 0
                # the retained energy of the residual: ||Q1^T r|| <= ||r||
                assert np.linalg.norm(ic[120:]) <= np.linalg.norm(if_[120:]) * (1 + 1e-12)
        ctx.update_joseph()
        # mixed stacking runs the sparse pipeline - unless a flag asks for the stand-alone tail (512: its G = T H^T walks ALL of H)
        # or for dense rows (64); the symmetric form (256) and the throughput route (8192) need nothing of H behind the solve
        assert ctx.last_path() == (1 if flags in (0, 256, 8192) else 0)
        # (flags 0 at B = 2: the latency route - whitened outputs, product by the tiled kernel; 8192 keeps the form inside the solve kernel)
        assert ctx.last_route() == {0: "sparse_whitened", 64: "dense_ascoded", 512: "dense_whitened", 256: "sparse_symmetric", 8192: "sparse_in_solve"}[flags]
        err = ctx.get_err(); Pn = ctx.download_P()
        assert (ctx.get_status() == 0).all()
    assert rows.tolist() == [140] * B
    for b in range(B):
        Js, inns, _ = oracle_jacobians(sc, cam, lay, xp, b)
        H, inn, dR = orc.stack_measurements(Js, inns, sc["ref"][b], sc["sind"][b], lay, R_VIS)
        for o in range(n_oos):
            Xs, obs = obs_all[b, o]
            Hxp, rp, _ = orc.oos_jacobian(Xs, obs, sc["gR"][b], sc["gT"][b], sc["Rbc"][b], sc["Tbc"][b], cam, lay)
            H = np.vstack([H, Hxp]); inn = np.concatenate([inn, rp]); dR = np.concatenate([dR, np.full(len(rp), 3.5 ** 2)])
        assert H.shape == (260, 251)
        e_ref, P_ref, _ = orc.update_joseph(H, P[b], inn, dR)
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX


@pytest.mark.parametrize("nframes", [3, 52])
def test_resident_frames_with_device_absorb_error(built, nframes):
    """frames of (Jacobians -> MHGating -> stack -> UpdateJosephForm -> AbsorbError)
    with P, dx and the nominal state (pose, groups, features) never leaving the device; only new pixel
    measurements are fed in. Oracle: the same loop on the host (estimator.cpp:875-921 for the retraction).
    52 frames cross State::counter % 50 == 0: the periodic SO3 re-normalisation / Rsg projection of core.h:154-162."""
    cam = synth.RADTAN
    B, ng, nf = 3, 5, 14
    sc, lay, ctx, poses, groups, feats, xp = make(ng, nf, nf, B, 9, cam)
    rng = np.random.default_rng(4)
    extra = dict(Vsb=rng.normal(size=(B, 3)), bg=rng.normal(size=(B, 3)) * 1e-2, ba=rng.normal(size=(B, 3)) * 1e-2,
                 Rsg=np.array([orc.so3_exp(rng.normal(size=3) * 0.05) for _ in range(B)]))
    for b in range(B):
        poses[b]["Vsb"] = extra["Vsb"][b]; poses[b]["bg"] = extra["bg"][b]; poses[b]["ba"] = extra["ba"][b]
        poses[b]["Rsg"] = extra["Rsg"][b].T.reshape(-1)
    feats["xp"][1, 5] += 60.0; xp[1, 5] += 60.0      # one outlier, rejected in every frame
    P0 = np.array([spd(lay.N, 70 + b) * 1e-4 for b in range(B)])
    frames = [xp] + [xp + rng.normal(size=xp.shape) * 0.7 for _ in range(nframes - 1)]
    with ctx:
        ctx.upload_P(P0); ctx.set_scene(poses, groups, feats)
        masks = []
        for k, meas in enumerate(frames):
            if k:                                    # only the pixels change; x / poses stay as the device left them
                _, _, fcur = ctx.get_scene()
                fcur["xp"] = meas
                pcur, gcur, _ = ctx.get_scene()
                ctx.set_scene(pcur, gcur, fcur)
            ctx.filter_update(R_VIS, MH, MULT, 5, True)
            masks.append(ctx.get_gate(nf)[0].copy())
            ctx.absorb_error()
        assert np.abs(ctx.get_err()).max() == 0.0     # err_.setZero()
        P = ctx.download_P()
        pose_d, group_d, feat_d = ctx.get_scene()
    for b in range(B):
        st = dict(Rsb=sc["Rsb"][b].copy(), Tsb=sc["Tsb"][b].copy(), Rbc=sc["Rbc"][b].copy(), Tbc=sc["Tbc"][b].copy(),
                  Vsb=extra["Vsb"][b].copy(), bg=extra["bg"][b].copy(), ba=extra["ba"][b].copy(), Rsg=extra["Rsg"][b].copy(),
                  gR=sc["gR"][b].copy(), gT=sc["gT"][b].copy(), x=sc["x"][b].copy(), sind=sc["sind"][b], ref=sc["ref"][b])
        Pb = P0[b].copy()
        for k, meas in enumerate(frames):
            Js, inns = [], []
            for i in range(nf):
                r = int(st["ref"][i])
                J, inn, _ = orc.compute_jacobian(st["x"][i], meas[b, i], st["gR"][r], st["gT"][r], st["Rsb"], st["Tsb"],
                                                 st["Rbc"], st["Tbc"], cam, lay, r, int(st["sind"][i]))
                Js.append(J); inns.append(inn)
            Js, inns = np.array(Js), np.array(inns)
            m, _, _ = orc.mh_gate(orc.mh_distances(Js, Pb, inns, R_VIS), MH, MULT, 5)
            assert np.array_equal(masks[k][b], m)
            H, inn, dR = orc.stack_measurements(Js[m], inns[m], st["ref"][m], st["sind"][m], lay, R_VIS)
            dx, Pb, _ = orc.update_joseph(H, Pb, inn, dR)
            orc.absorb_error(st, dx, lay, range(ng), np.nonzero(m)[0])
        assert not masks[0][1, 5]
        assert rel_fro(P[b], Pb) < TOL_P
        cmT = lambda v: np.asarray(v).reshape(3, 3).T
        for name in ("Rsb", "Rbc", "Rsg"):
            assert np.abs(cmT(pose_d[b][name]) - st[name]).max() < 1e-9, name
        for name in ("Tsb", "Tbc", "Vsb", "bg", "ba"):
            assert np.abs(pose_d[b][name] - st[name]).max() < 1e-9, name
        for g in range(ng):
            assert np.abs(cmT(group_d[b, g]["Rsb"]) - st["gR"][g]).max() < 1e-9
            assert np.abs(group_d[b, g]["Tsb"] - st["gT"][g]).max() < 1e-9
        assert np.abs(feat_d[b]["x"] - st["x"]).max() < 1e-9


@pytest.mark.parametrize("name", list(CAMS))
def test_subfilter_update_and_candidates(built, name):
    """Feature::SubfilterUpdate + Criteria::Candidate(Strict) + Feature::score for a batch of
    not-yet-in-state features, three consecutive frames on the device vs the oracle (golden-pinned restatement)."""
    from xivo_amd.lib import subfilter_dtype
    cam = CAMS[name]
    B, ng, nfeat = 3, 4, 9
    sc, lay, ctx, poses, groups, feats, xp = make(ng, nfeat, nfeat, B, 17, cam)
    rng = np.random.default_rng(8)
    sub = np.zeros((B, nfeat), dtype=subfilter_dtype)
    x0 = sc["x"] + rng.normal(size=sc["x"].shape) * np.array([0.01, 0.01, 0.15])
    P0 = np.diag([1e-4, 1e-4, 0.25])
    for b in range(B):
        for i in range(nfeat):
            sub[b, i]["x"] = x0[b, i]; sub[b, i]["P"] = P0.T.reshape(-1); sub[b, i]["ref_sind"] = sc["ref"][b, i]
    frames = [xp + rng.normal(size=xp.shape) * 0.8 for _ in range(3)]
    frames[1][0, 4] += 25.0           # wild pixel: ratio > 1 branch
    frames[2][1, 2] -= 40.0
    ref_state = [[dict(x=x0[b, i].copy(), P=P0.copy(), ic=0, oc=0.0, st=0) for i in range(nfeat)] for b in range(B)]
    with ctx:
        ctx.set_scene(poses, groups, feats)
        for fr in range(3):
            sub["xp"] = frames[fr]
            sub = ctx.subfilter_update(sub, ready_steps=1)
            for b in range(B):
                for i in range(nfeat):
                    s = ref_state[b][i]; r = int(sc["ref"][b, i])
                    s["x"], s["P"], s["st"], s["ic"], s["oc"] = orc.subfilter_update(
                        s["x"], s["P"], frames[fr][b, i], sc["Rsb"][b], sc["Tsb"][b], sc["Rbc"][b], sc["Tbc"][b],
                        sc["gR"][b, r], sc["gT"][b, r], cam, 3.5, 5.991, 1, s["ic"], s["oc"])
                    d = sub[b, i]
                    assert np.abs(d["x"] - s["x"]).max() < 1e-10
                    assert rel_fro(d["P"].reshape(3, 3).T, s["P"]) < 1e-9
                    assert d["status"] == s["st"] and d["init_counter"] == s["ic"]
                    assert abs(d["outlier_counter"] - s["oc"]) < 1e-9 * max(1.0, s["oc"])
                    cand, strict = orc.candidate_flags(s["x"], s["st"], s["oc"])
                    assert d["candidate"] == (1 if cand else 0) | (2 if strict else 0)
                    assert abs(d["score"] - orc.feature_score(s["P"])) < 1e-12
    assert any(ref_state[b][i]["oc"] > 0 for b in range(B) for i in range(nfeat))   # the inflated-S branch ran
    assert (sub["status"] == 1).all()


@pytest.mark.parametrize("rows,nx,eff", [(8, 60, -1), (32, 107, -1), (20, 60, 14), (6, 3, -1)])
def test_device_givens_matches_reference_semantics(built, rows, nx, eff):
    """xivo::Givens (src/helpers.cpp:48-75) batched on the device, including what the reference actually codes:
    only the first 3 columns of Hx are rotated (helpers.cpp:64) and the first 3 rows are stripped."""
    rng = np.random.default_rng(rows + nx)
    nb = 5
    x = rng.normal(size=(nb, rows)); Hx = rng.normal(size=(nb, rows, nx)); Hf = rng.normal(size=(nb, rows, 3))
    Hf[1, 3, 0] = 0.0; Hf[1, 4, 0] = 1e-6        # |b| < eps branch of givens()
    with Context(8, 2, 1) as ctx:
        ro, xd, Hxd, Hfd = ctx.givens(x, Hx, Hf, eff)
    for b in range(nb):
        r, xo, Hxo, Hfo = orc.givens_eliminate(x[b], Hx[b], Hf[b], eff)
        assert ro[b] == r
        assert np.abs(xd[b] - xo).max() < 1e-12 and np.abs(Hxd[b] - Hxo).max() < 1e-12 and np.abs(Hfd[b] - Hfo).max() < 1e-12


@pytest.mark.parametrize("rows,nx,eff", [(12, 5, -1), (40, 21, -1), (200, 130, -1), (90, 70, 80)])
def test_device_qr_compression(built, rows, nx, eff):
    """xivo::QR (src/helpers.cpp:78-101): measurement compression, all columns rotated; nx > 64 exercises the
    multi-chunk column ownership."""
    rng = np.random.default_rng(rows * 7 + nx)
    nb = 3
    x = rng.normal(size=(nb, rows)); Hx = rng.normal(size=(nb, rows, nx))
    with Context(8, 2, 1) as ctx:
        ro, xd, Hxd = ctx.qr(x, Hx, eff)
    for b in range(nb):
        r, xo, Hxo = orc.qr_compress(x[b], Hx[b], eff)
        assert ro[b] == r
        assert np.abs(xd[b] - xo).max() < 1e-10 and np.abs(Hxd[b] - Hxo).max() < 1e-10
        assert np.abs(np.tril(Hxd[b][:r], -1)).max() < 1e-10


@pytest.mark.parametrize("method,dt,stepsize,gyro_scale", [("RK4", 0.005, 0.002, 0.3), ("PrinceDormand", 0.0045, 0.002, 0.3),
                                                           ("RK4", 0.003, -1.0, 0.3), ("PrinceDormand", 0.01, 0.002, 0.3),
                                                           ("RK4", 0.1, -1.0, 9.0), ("PrinceDormand", 0.1, 0.05, 9.0)])
def test_device_propagate_state_and_covariance(built, method, dt, stepsize, gyro_scale):
    """Estimator::Propagate entirely on the device (nominal motion state of the resident scene + P): RK4Step /
    PrinceDormandStep with the reference's sub-stepping, ComposeMotion, ComputeMotionJacobianAt, covariance tail and
    + Qmodel, vs the oracle (pinned against the line-faithful Sophus/Eigen RK4Step, golden rk4_*)."""
    from xivo_amd.lib import imu_dtype
    cam = synth.PINHOLE
    B, ng, nf = 4, 3, 6
    sc, lay, ctx, poses, groups, feats, xp = make(ng, nf, nf, B, 21, cam)
    rng = np.random.default_rng(12)
    N = lay.N
    st = []
    for b in range(B):
        X = orc.MotionState(sc["Rsb"][b], sc["Tsb"][b], rng.normal(size=3) * 0.5, rng.normal(size=3) * 0.01,
                            rng.normal(size=3) * 0.05, orc.so3_exp(np.array([0.02, -0.03, 0.0])))
        st.append(X)
        poses[b]["Vsb"] = X.Vsb; poses[b]["bg"] = X.bg; poses[b]["ba"] = X.ba; poses[b]["Rsg"] = X.Rsg.T.reshape(-1)
    P = np.array([spd(N, 50 + b) * 1e-3 for b in range(B)])
    imu = np.zeros(B, dtype=imu_dtype)
    # gyro_scale 9: rotation increments of ~1 rad per stage - the device takes the halve-and-square route of
    # so3_exp_small where the oracle evaluates sin / cos
    imu["gyro"] = rng.normal(size=(B, 3)) * gyro_scale; imu["accel"] = rng.normal(size=(B, 3)) + np.array([0, 0, 9.8])
    imu["slope_gyro"] = rng.normal(size=(B, 3)) * 5.0; imu["slope_accel"] = rng.normal(size=(B, 3)) * 20.0
    imu["dt"] = dt * (1.0 + 0.1 * np.arange(B))        # a different sub-step pattern per filter
    Qi = np.diag(rng.uniform(1e-6, 1e-4, 12)); A = rng.normal(size=(23, 23)) * 1e-4; Qm = A @ A.T
    g = np.array([0.0, 0.0, -9.796])
    with ctx:
        ctx.upload_P(P); ctx.set_scene(poses, groups, feats)
        ctx.propagate(imu, Qi, Qm, g, method=method, stepsize=stepsize)
        Pn = ctx.download_P()
        pose_d, _, _ = ctx.get_scene()
    for b in range(B):
        Xr, Pr = orc.propagate(st[b], P[b], imu["gyro"][b], imu["accel"][b], imu["slope_gyro"][b], imu["slope_accel"][b],
                               float(imu["dt"][b]), Qi, Qm, g, method=method, stepsize=stepsize)
        assert rel_fro(Pn[b], Pr) < 1e-11
        assert np.abs(pose_d[b]["Rsb"].reshape(3, 3).T - Xr.Rsb).max() < 1e-12
        assert np.abs(pose_d[b]["Tsb"] - Xr.Tsb).max() < 1e-12 and np.abs(pose_d[b]["Vsb"] - Xr.Vsb).max() < 1e-12
        assert np.array_equal(pose_d[b]["bg"], st[b].bg) and np.array_equal(pose_d[b]["ba"], st[b].ba)


def test_device_prince_dormand_step_size_control_as_coded(built):
    """xivo_prop_opts.control_stepsize: the step-size-controlled branch of Estimator::PrinceDormand (src/princedormand.cpp:26-60)
    as coded - PrinceDormandStep returns 0 (:216-220), so the step grows by max_scale_factor after every step, clipped / halved
    at the end of a sample (:53-58), started from gyro0 + slope * total_step (:38-39) and carried from sample to sample and
    from call to call (the reference's function-local static). A chain of samples of different lengths, split over two calls
    (three samples in one call, two in the next), per filter a different chain - against the oracle restatement that
    tests/test_oracle_pinned.py pins to the extracted function (golden_v7), and against those stored outputs directly."""
    import test_oracle_pinned as top
    from xivo_amd.lib import imu_dtype
    m7 = top._v7(); m6 = top._v6()
    cam = synth.PINHOLE
    B, ng, nf = 3, 15, 30                                 # the default build's sizes: N = 203, the golden chain's
    sc, lay, ctx, poses, groups, feats, xp = make(ng, nf, nf, B, 21, cam)
    assert lay.N == 203
    chains = [m7.CHAIN_NS, m7.CHAIN_NS, (4000000, 2500000, 9000000, 600000, 3000000)]
    seeds = [1, 2, 2]
    cases = [m6.prop_case(sd) for sd in seeds]
    for b in range(B):
        X = cases[b]["X"]
        poses[b]["Rsb"] = X.Rsb.T.reshape(-1); poses[b]["Tsb"] = X.Tsb
        poses[b]["Vsb"] = X.Vsb; poses[b]["bg"] = X.bg; poses[b]["ba"] = X.ba; poses[b]["Rsg"] = X.Rsg.T.reshape(-1)
    P = np.array([cases[b]["P"] for b in range(B)])
    n = len(m7.CHAIN_NS)
    imu = np.zeros((B, n), dtype=imu_dtype)
    for b in range(B):
        c = cases[b]
        imu["gyro"][b] = c["gy"]; imu["accel"][b] = c["ac"]; imu["slope_gyro"][b] = c["sg"]; imu["slope_accel"][b] = c["sa"]
        imu["dt"][b] = np.array(chains[b]) * 1e-9
    c0 = cases[0]
    with ctx:
        ctx.upload_P(P); ctx.set_scene(poses, groups, feats)
        with pytest.raises(Exception):                                     # RK4 has no such branch
            ctx.propagate(imu[:, :1], c0["Qimu"], c0["Qmodel"], c0["g"], method="RK4", stepsize=0.002, pd_control=m7.PD_CTL)
        ctx.propagate(imu[:, :3], c0["Qimu"], c0["Qmodel"], c0["g"], method="PrinceDormand", stepsize=0.002, pd_control=m7.PD_CTL)
        ctx.propagate(imu[:, 3:], c0["Qimu"], c0["Qmodel"], c0["g"], method="PrinceDormand", stepsize=0.002, pd_control=m7.PD_CTL)
        Pn = ctx.download_P()
        pose_d, _, _ = ctx.get_scene()
    for b in range(B):
        c = cases[b]
        ctl = orc.PDControl(stepsize=0.002, **m7.PD_CTL)
        X = orc.MotionState(c["X"].Rsb.copy(), c["X"].Tsb.copy(), c["X"].Vsb.copy(), c["X"].bg.copy(), c["X"].ba.copy(), c["X"].Rsg.copy())
        Pr = c["P"]
        for dt_ns in chains[b]:
            # (the same IMU sample every call - the chain of the golden file; Qimu / Qmodel / g are the first case's for every filter)
            X, Pr = orc.propagate(X, Pr, c["gy"], c["ac"], c["sg"], c["sa"], dt_ns * 1e-9, c0["Qimu"], c0["Qmodel"], c0["g"],
                                  method="PrinceDormand", stepsize=0.002, pd_control=ctl)[:2]
        assert rel_fro(Pn[b], Pr) < 1e-11
        assert np.abs(pose_d[b]["Rsb"].reshape(3, 3).T - X.Rsb).max() < 1e-12
        assert np.abs(pose_d[b]["Tsb"] - X.Tsb).max() < 1e-12 and np.abs(pose_d[b]["Vsb"] - X.Vsb).max() < 1e-12
    k = f"pdc_s1_{n - 1}"                                                  # filter 0 is the stored chain of seed 1 itself
    assert rel_fro(Pn[0][:23, :23], top.G7[k + "_Pmm"]) < 1e-10 and np.abs(pose_d[0]["Vsb"] - top.G7[k + "_Vsb"]).max() < 1e-11
    assert rel_fro(Pn[0][:23, 23:] @ cases[0]["w"], top.G7[k + "_Pms_w"]) < 1e-10


def test_resident_full_frame_loop(built):
    """The whole per-frame EKF loop with nothing but the IMU sample and the pixels crossing the boundary:
    Propagate -> ComputeInstateJacobians -> MHGating -> FilterUpdate -> AbsorbError, three frames, state and P
    resident on the device (src/estimator.cpp:539-592, src/manager.cpp:72-104, src/estimator.cpp:875-921)."""
    from xivo_amd.lib import imu_dtype
    cam = synth.EQUI
    B, ng, nf = 2, 4, 10
    sc, lay, ctx, poses, groups, feats, xp = make(ng, nf, nf, B, 33, cam)
    rng = np.random.default_rng(6)
    N = lay.N
    ref = []
    for b in range(B):
        X = orc.MotionState(sc["Rsb"][b], sc["Tsb"][b], rng.normal(size=3) * 0.05, rng.normal(size=3) * 1e-3,
                            rng.normal(size=3) * 1e-2, np.eye(3))
        poses[b]["Vsb"] = X.Vsb; poses[b]["bg"] = X.bg; poses[b]["ba"] = X.ba; poses[b]["Rsg"] = X.Rsg.T.reshape(-1)
        ref.append(dict(X=X, P=spd(N, 80 + b) * 1e-4, gR=sc["gR"][b].copy(), gT=sc["gT"][b].copy(), x=sc["x"][b].copy()))
    P0 = np.array([r["P"] for r in ref])
    Qi = np.eye(12) * 1e-6; Qm = np.eye(23) * 1e-8; g = np.array([0.0, 0.0, -9.8])
    frames = []
    for k in range(3):
        imu = np.zeros(B, dtype=imu_dtype)
        imu["gyro"] = rng.normal(size=(B, 3)) * 0.02; imu["accel"] = rng.normal(size=(B, 3)) * 0.05 + np.array([0, 0, 9.8])
        imu["slope_gyro"] = rng.normal(size=(B, 3)); imu["slope_accel"] = rng.normal(size=(B, 3))
        imu["dt"] = 0.005
        frames.append((imu, xp + rng.normal(size=xp.shape) * 0.6))
    with ctx:
        ctx.upload_P(P0); ctx.set_scene(poses, groups, feats)
        for imu, meas in frames:
            ctx.propagate(imu, Qi, Qm, g, method="RK4", stepsize=0.002)
            pcur, gcur, fcur = ctx.get_scene()
            fcur["xp"] = meas
            ctx.set_scene(pcur, gcur, fcur)
            ctx.filter_update(R_VIS, MH, MULT, 5, True)
            ctx.absorb_error()
        Pn = ctx.download_P()
        pose_d, group_d, feat_d = ctx.get_scene()
    for b in range(B):
        r = ref[b]
        for imu, meas in frames:
            r["X"], r["P"] = orc.propagate(r["X"], r["P"], imu["gyro"][b], imu["accel"][b], imu["slope_gyro"][b],
                                           imu["slope_accel"][b], 0.005, Qi, Qm, g, method="RK4", stepsize=0.002)
            Js, inns = [], []
            for i in range(nf):
                rr = int(sc["ref"][b, i])
                J, inn, _ = orc.compute_jacobian(r["x"][i], meas[b, i], r["gR"][rr], r["gT"][rr], r["X"].Rsb, r["X"].Tsb,
                                                 sc["Rbc"][b], sc["Tbc"][b], cam, lay, rr, int(sc["sind"][b, i]))
                Js.append(J); inns.append(inn)
            Js, inns = np.array(Js), np.array(inns)
            m, _, _ = orc.mh_gate(orc.mh_distances(Js, r["P"], inns, R_VIS), MH, MULT, 5)
            H, inn, dR = orc.stack_measurements(Js[m], inns[m], sc["ref"][b][m], sc["sind"][b][m], lay, R_VIS)
            dx, r["P"], _ = orc.update_joseph(H, r["P"], inn, dR)
            stt = dict(Rsb=r["X"].Rsb, Tsb=r["X"].Tsb, Vsb=r["X"].Vsb, bg=r["X"].bg, ba=r["X"].ba, Rbc=sc["Rbc"][b].copy(),
                       Tbc=sc["Tbc"][b].copy(), Rsg=r["X"].Rsg, gR=r["gR"], gT=r["gT"], x=r["x"], sind=sc["sind"][b])
            orc.absorb_error(stt, dx, lay, range(ng), np.nonzero(m)[0])
            r["X"] = orc.MotionState(stt["Rsb"], stt["Tsb"], stt["Vsb"], stt["bg"], stt["ba"], stt["Rsg"])
            sc["Rbc"][b], sc["Tbc"][b] = stt["Rbc"], stt["Tbc"]
        assert rel_fro(Pn[b], r["P"]) < TOL_P
        assert np.abs(pose_d[b]["Rsb"].reshape(3, 3).T - r["X"].Rsb).max() < 1e-9
        assert np.abs(pose_d[b]["Tsb"] - r["X"].Tsb).max() < 1e-9 and np.abs(pose_d[b]["Vsb"] - r["X"].Vsb).max() < 1e-9
        assert np.abs(pose_d[b]["bg"] - r["X"].bg).max() < 1e-10 and np.abs(pose_d[b]["ba"] - r["X"].ba).max() < 1e-10
        assert np.abs(feat_d[b]["x"] - r["x"]).max() < 1e-9


def test_device_propagate_several_imu_samples_in_one_call(built):
    """Five IMU samples per filter in one xivo_hip_propagate call == five Estimator::Propagate calls in a row (each
    adds Qmodel); the cross-covariance tail is applied once with the accumulated transition."""
    from xivo_amd.lib import imu_dtype
    cam = synth.PINHOLE
    B, ng, nf, K = 3, 3, 6, 5
    sc, lay, ctx, poses, groups, feats, xp = make(ng, nf, nf, B, 22, cam)
    rng = np.random.default_rng(13)
    st = []
    for b in range(B):
        X = orc.MotionState(sc["Rsb"][b], sc["Tsb"][b], rng.normal(size=3) * 0.3, rng.normal(size=3) * 0.01,
                            rng.normal(size=3) * 0.05, np.eye(3))
        st.append(X)
        poses[b]["Vsb"] = X.Vsb; poses[b]["bg"] = X.bg; poses[b]["ba"] = X.ba; poses[b]["Rsg"] = X.Rsg.T.reshape(-1)
    P = np.array([spd(lay.N, 60 + b) * 1e-3 for b in range(B)])
    imu = np.zeros((B, K), dtype=imu_dtype)
    imu["gyro"] = rng.normal(size=(B, K, 3)) * 0.3; imu["accel"] = rng.normal(size=(B, K, 3)) + np.array([0, 0, 9.8])
    imu["slope_gyro"] = rng.normal(size=(B, K, 3)) * 5.0; imu["slope_accel"] = rng.normal(size=(B, K, 3)) * 20.0
    imu["dt"] = 0.005
    Qi = np.diag(rng.uniform(1e-6, 1e-4, 12)); A = rng.normal(size=(23, 23)) * 1e-4; Qm = A @ A.T
    g = np.array([0.0, 0.0, -9.796])
    with ctx:
        ctx.upload_P(P); ctx.set_scene(poses, groups, feats)
        ctx.propagate(imu, Qi, Qm, g, method="RK4", stepsize=0.002)
        Pn = ctx.download_P()
        pose_d, _, _ = ctx.get_scene()
    for b in range(B):
        Xr, Pr = st[b], P[b]
        for k in range(K):
            Xr, Pr = orc.propagate(Xr, Pr, imu["gyro"][b, k], imu["accel"][b, k], imu["slope_gyro"][b, k],
                                   imu["slope_accel"][b, k], 0.005, Qi, Qm, g, method="RK4", stepsize=0.002)
        assert rel_fro(Pn[b], Pr) < 1e-11
        assert np.abs(pose_d[b]["Rsb"].reshape(3, 3).T - Xr.Rsb).max() < 1e-12
        assert np.abs(pose_d[b]["Tsb"] - Xr.Tsb).max() < 1e-12 and np.abs(pose_d[b]["Vsb"] - Xr.Vsb).max() < 1e-12


def test_oos_compression_trigger_and_ragged_blocks(built):
    """compression_trigger_ratio: a block with fewer than ratio x (non-zero columns) rows is left alone; filters of one
    call hold different numbers of OOS observations; the compressed update equals the oracle's uncompressed one."""
    cam = synth.EQUI
    ng, nf, F, B, n_oos = 4, 10, 10, 3, 12
    sc, lay, ctx, poses, groups, feats, xp = make(ng, nf, F, B, 12, cam, M_max=2 * F + n_oos * 5)
    rng = np.random.default_rng(3)
    oos = np.zeros((B, n_oos), dtype=oos_dtype)
    obs_all = {}
    kk = {0: 4, 1: 3, 2: 2}                       # observations per OOS feature: 5 / 3 / 1 projected rows each
    for b in range(B):
        for o in range(n_oos):
            k = kk[b]
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
    P = np.array([spd(lay.N, 90 + b) * 1e-4 for b in range(B)])
    with ctx:
        ctx.upload_P(P); ctx.set_scene(poses, groups, feats)
        ctx.jacobians_instate(); ctx.mh_gate(R_VIS, MH, MULT, 5); ctx.stack(R_VIS)
        rows = ctx.oos_project(oos, 3.5 ** 2)
        assert rows.tolist() == [60, 36, 12]
        crow = ctx.compress_oos(1.5)
        # 30 non-zero columns (6 + 24): 60 > 45 compresses to 30 rows; 36 and 12 rows stay (36 < 45)
        assert crow.tolist() == [30, 36, 12]
        ctx.update_joseph()
        err = ctx.get_err(); Pn = ctx.download_P()
        assert (ctx.get_status() == 0).all()
    for b in range(B):
        Js, inns, _ = oracle_jacobians(sc, cam, lay, xp, b)
        H, inn, dR = orc.stack_measurements(Js, inns, sc["ref"][b], sc["sind"][b], lay, R_VIS)
        for o in range(n_oos):
            Xs, obs = obs_all[b, o]
            Hxp, rp, _ = orc.oos_jacobian(Xs, obs, sc["gR"][b], sc["gT"][b], sc["Rbc"][b], sc["Tbc"][b], cam, lay)
            H = np.vstack([H, Hxp]); inn = np.concatenate([inn, rp]); dR = np.concatenate([dR, np.full(len(rp), 3.5 ** 2)])
        e_ref, P_ref, _ = orc.update_joseph(H, P[b], inn, dR)
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX
