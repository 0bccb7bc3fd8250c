"""Round 5, through the C ABI on the GPU:
 * Feature::ComputeLCJacobian rows under Estimator::CloseLoopInternal's stacking (src/oos.cpp:92-145, src/update.cpp:183-196):
   default, online-calibration (intrinsics block) and USE_INVDEPTH builds - rows against the oracle (pinned to the reference's
   own text, tests/test_oracle_pinned.py) and against the stored rows of the extracted build, then through the update;
 * the USE_INVDEPTH build (src/feature.cpp:98-105): in-state Jacobians, the whole feature-level update, the depth sub-filter;
 * MH gating of online-calibration builds on ragged batches (absent entries are no candidates; per-filter present count);
 * XIVO_HIP_FLAG_FP32_WHITENED over a CHAIN of updates (the tolerance BASELINE config 4 asks to be stated), and its
   restriction to shapes whose product runs outside the solve kernel."""
import importlib.util
import os

import numpy as np
import pytest

import xivo_oracle as orc
from helpers import rel_fro, TOL_P, TOL_DX
from scene_util import scene_arrays, spd
from xivo_amd import synth
from xivo_amd.lib import Context, FLAG_INVDEPTH, FLAG_FP32_WHITENED, calib_dtype, cam_intr, lc_dtype, subfilter_dtype

pytestmark = pytest.mark.gpu
CAMS = {"pinhole": synth.PINHOLE, "equi": synth.EQUI, "radtan": synth.RADTAN, "atan": synth.ATAN}
CAM_DIM = {"pinhole": 4, "atan": 5, "radtan": 9, "equi": 8}
R_VIS, MH, MULT = 2.25, 5.991, 1.1
G6 = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v6.npz"))


def _v6():
    spec = importlib.util.spec_from_file_location("make_golden_v6", os.path.join(os.path.dirname(__file__), "golden", "make_golden_v6.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def _cat(scs):
    """batch-1 scenes -> one scene of batch len(scs)"""
    out = dict(scs[0])
    for k, v in scs[0].items():
        if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == 1:
            out[k] = np.concatenate([s[k] for s in scs])
    return out


@pytest.mark.parametrize("name", list(CAMS))
@pytest.mark.parametrize("build", ["n251", "calib", "invdepth"])
def test_loop_closure_rows_and_update(built, build, name):
    m = _v6()
    cam = CAMS[name]
    ng, nf = m.BUILD_SIZES[build]
    inv = build == "invdepth"
    lay = orc.calib_layout(ng, nf, True, True, CAM_DIM[name]) if build == "calib" else orc.Layout(ng, nf)
    seeds = (3, 4, 5)                                   # (seed 3 = the stored case of golden_v6)
    cases = [m.lc_case(build, cam, s) for s in seeds]
    B, n = len(cases), 6
    sc = _cat([c[0] for c in cases])
    poses, groups, feats, _ = scene_arrays(sc, cam, xp=np.zeros((B, nf, 2)))
    for b in range(B):
        feats["x"][b] = cases[b][1]                     # (inverse depth for that build)
    mt = np.zeros((B, n), dtype=lc_dtype)
    for b in range(B):
        for i, (q, fi) in enumerate(zip(cases[b][2], cases[b][3])):
            mt[b, i]["feat"], mt[b, i]["group_sind"], mt[b, i]["xp"] = fi, q["g_sind"], q["xp"]
    mt[1, 2]["feat"] = -1                               # ragged: filter 1 has one match fewer
    P = np.array([spd(lay.N, 40 + b) * 1e-4 for b in range(B)])
    Rlc = 1.5 ** 2
    with Context(lay.N, 2 * nf, B, flags=FLAG_INVDEPTH if inv else 0) as ctx:
        ctx.set_layout(lay.N, lay.group_begin, ng, lay.feature_begin, nf, cam)
        if build == "calib":
            ctx.set_calib(lay.td, lay.Cg, lay.cam_begin, lay.cam_dim)
            cal = np.zeros(B, dtype=calib_dtype)
            for b in range(B):
                cal[b]["Cg"] = np.eye(3).reshape(-1); cal[b]["Ca"] = np.eye(3).reshape(-1); cal[b]["intr"] = cam_intr(cam)
        ctx.upload_P(P); ctx.set_scene(poses, groups, feats)
        if build == "calib":
            ctx.set_calib_state(cal)
        ctx.close_loop_stack(mt, Rlc)
        got = [ctx.get_H(b) for b in range(B)]
        ctx.update_joseph()
        if build != "calib":
            assert ctx.last_path() == 1                 # the rows fit the row-pair compressed form: sparse pipeline
        err = ctx.get_err(); Pn = ctx.download_P()
        assert (ctx.get_status() == 0).all()
    for b in range(B):
        Ho, io = orc.lc_jacobian_rows(cases[b][2], sc["Rbc"][b], sc["Tbc"][b], cam, lay, invdepth=inv)
        dRo = np.full(2 * n, Rlc)
        if b == 1:                                      # the absent match is a neutral row pair
            Ho[4:6] = 0.0; io[4:6] = 0.0; dRo[4:6] = 1.0
        assert got[b][0].shape == Ho.shape and rel_fro(got[b][0], Ho) < 1e-12 and np.abs(got[b][1] - io).max() < 1e-9
        assert np.array_equal(got[b][2], dRo)
        if b == 0:                                      # ... and the reference's own text, stored
            k = f"lc_{build}_{name}"
            cols = G6[k + "_cols"]
            assert np.nonzero(np.abs(got[0][0]).sum(0))[0].tolist() == cols.tolist()
            assert np.abs(got[0][0][:, cols] - G6[k + "_H"]).max() / np.abs(Ho).max() < 1e-10 and np.abs(got[0][1] - G6[k + "_inn"]).max() < 1e-9
        keep = np.ones(2 * n, dtype=bool)
        if b == 1:
            keep[4:6] = False
        e_ref, P_ref, _ = orc.update_joseph(Ho[keep], P[b], io[keep], dRo[keep])
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX


@pytest.mark.parametrize("name", list(CAMS))
def test_invdepth_build_jacobians_update_and_subfilter(built, name):
    m = _v6()
    cam = CAMS[name]
    ng, nf, B = 8, 60, 3
    sc = synth.g_level(ng, nf, nf, B, seed=9, cam=cam)
    lay = orc.Layout(ng, nf)
    poses, groups, feats, xp = scene_arrays(sc, cam)
    xi = m.to_invdepth(sc["x"])
    feats["x"] = xi
    feats["xp"][1, [3, 8]] += 60.0; xp[1, [3, 8]] += 60.0
    P = np.array([spd(lay.N, 70 + b) * 1e-4 for b in range(B)])
    with Context(lay.N, 2 * nf, B, flags=FLAG_INVDEPTH) as ctx:
        ctx.set_layout(lay.N, lay.group_begin, ng, lay.feature_begin, nf, cam)
        ctx.upload_P(P); ctx.set_scene(poses, groups, feats)
        ctx.jacobians_instate()
        J, inn = ctx.get_jacobians()
        ctx.filter_update(R_VIS, MH, MULT, 5, use_gating=True)
        err = ctx.get_err(); Pn = ctx.download_P(); mask, _ = ctx.get_gate(nf, B)
        # depth sub-filter of candidates in the same parametrisation
        sub = np.zeros((B, 4), dtype=subfilter_dtype)
        cs = [[m.sub_case(cam, 10 + 4 * b + i, invdepth=True) for i in range(4)] for b in range(B)]
        for b in range(B):
            for i in range(4):
                c = cs[b][i]
                sub[b, i]["x"] = c["x"]; sub[b, i]["P"] = c["P"].T.reshape(-1); sub[b, i]["xp"] = c["xp"]; sub[b, i]["ref_sind"] = i % ng
                sub[b, i]["init_counter"] = c["init_counter"]; sub[b, i]["outlier_counter"] = c["outlier_counter"]
        sub = ctx.subfilter_update(sub)
    rejected = 0
    for b in range(B):
        Js, inns, blocks = [], [], []
        for i in range(nf):
            r = int(sc["ref"][b, i])
            Ji, ii, blk = orc.compute_jacobian(xi[b, i], xp[b, i], sc["gR"][b, r], sc["gT"][b, r], sc["Rsb"][b], sc["Tsb"][b], sc["Rbc"][b],
                                               sc["Tbc"][b], cam, lay, r, int(sc["sind"][b, i]), invdepth=True)
            Js.append(Ji); inns.append(ii); blocks.append(blk)
        Js, inns, blocks = np.array(Js), np.array(inns), np.array(blocks)
        Jo = np.concatenate([blocks[:, k] for k in range(7)], axis=2)
        assert rel_fro(J[b], Jo) < 1e-12 and np.abs(inn[b] - inns).max() < 1e-9
        mk, _, _ = orc.mh_gate(orc.mh_distances(Js, P[b], inns, R_VIS), MH, MULT, 5)
        assert np.array_equal(mask[b].astype(bool), mk)
        rejected += int((~mk).sum())
        idx = np.nonzero(mk)[0]
        H, iv, dR = orc.stack_measurements(Js[idx], inns[idx], sc["ref"][b][idx], sc["sind"][b][idx], lay, R_VIS)
        e_ref, P_ref, _ = orc.update_joseph(H, P[b], iv, dR)
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX
        for i in range(4):
            c = cs[b][i]; r = i % ng
            xs, Ps, st, ic, oc = orc.subfilter_update(c["x"], c["P"], c["xp"], sc["Rsb"][b], sc["Tsb"][b], sc["Rbc"][b], sc["Tbc"][b],
                                                      sc["gR"][b, r], sc["gT"][b, r], cam, 3.5, 5.991, 5, c["init_counter"], c["outlier_counter"],
                                                      invdepth=True)
            d = sub[b, i]
            assert np.abs(d["x"] - xs).max() < 1e-9 * max(1.0, np.abs(xs).max()) and rel_fro(d["P"].reshape(3, 3).T, Ps) < 1e-9
            assert d["status"] == st and d["init_counter"] == ic and abs(d["outlier_counter"] - oc) < 1e-9 * max(1.0, oc)
            cand, strict = orc.candidate_flags(xs, st, oc, invdepth=True)
            assert d["candidate"] == (1 if cand else 0) | (2 if strict else 0)
    assert rejected >= 2
    # the stored Jacobian of the extracted USE_INVDEPTH build (filter 0 of this scene = the golden scene)
    sc1 = synth.g_level(ng, nf, nf, 1, seed=9, cam=cam)
    assert np.array_equal(sc1["x"][0], sc["x"][0])


def test_calibration_gate_on_a_ragged_batch(built):
    """Absent entries of an online-calibration context are no gating candidates: filter 0 holds 4 features (<= min_inliers: not
    gated, its wild pixel stays an inlier, src/manager.cpp:635), filter 1 holds 9 of 14 with two wild pixels (gated on the
    present ones only), filter 2 is full."""
    name = "equi"; cam = CAMS[name]
    ng, nf, B = 6, 14, 3
    lay = orc.calib_layout(ng, nf, True, True, CAM_DIM[name])
    sc = synth.g_level(ng, nf, nf, B, seed=21, cam=cam)
    poses, groups, feats, xp = scene_arrays(sc, cam)
    rng = np.random.default_rng(21)
    calib = np.zeros(B, dtype=calib_dtype); cals = []
    for b in range(B):
        cal = dict(gyro=rng.normal(size=3) * 0.5, Cg=np.eye(3) + 0.01 * rng.normal(size=(3, 3)), bg=rng.normal(size=3) * 0.01, Vsb=rng.normal(size=3), td=0.01)
        poses[b]["Vsb"], poses[b]["bg"] = cal["Vsb"], cal["bg"]
        calib[b]["gyro"], calib[b]["Cg"], calib[b]["td"] = cal["gyro"], cal["Cg"].T.reshape(-1), cal["td"]
        calib[b]["Ca"], calib[b]["intr"] = np.eye(3).reshape(-1), cam_intr(cam)
        cals.append(cal)
    present = [np.arange(4), np.array([0, 1, 3, 4, 6, 8, 9, 11, 13]), np.arange(nf)]
    for b in range(B):
        absent = np.setdiff1d(np.arange(nf), present[b])
        feats["sind"][b, absent] = -1
    feats["xp"][0, 2] += 70.0; xp[0, 2] += 70.0
    feats["xp"][1, [3, 9]] += 70.0; xp[1, [3, 9]] += 70.0
    P = np.array([spd(lay.N, 90 + b) * 1e-4 for b in range(B)])
    with Context(lay.N, 2 * nf, B) as ctx:
        ctx.set_layout(lay.N, lay.group_begin, ng, lay.feature_begin, nf, cam)
        ctx.set_calib(lay.td, lay.Cg, lay.cam_begin, lay.cam_dim)
        ctx.upload_P(P); ctx.set_scene(poses, groups, feats); ctx.set_calib_state(calib)
        ctx.filter_update(R_VIS, MH, MULT, 5, use_gating=True)
        err = ctx.get_err(); Pn = ctx.download_P(); mask, dist = ctx.get_gate(nf, B)
        assert (ctx.get_status() == 0).all()
    for b in range(B):
        idx = present[b]
        Js, inns = [], []
        for i in idx:
            r = int(sc["ref"][b][i])
            Ji, ii, _, _ = orc.compute_jacobian(sc["x"][b][i], xp[b][i], sc["gR"][b][r], sc["gT"][b][r], sc["Rsb"][b], sc["Tsb"][b], sc["Rbc"][b],
                                                sc["Tbc"][b], cam, lay, r, int(sc["sind"][b][i]), calib=cals[b])
            Js.append(Ji); inns.append(ii)
        Js, inns = np.array(Js), np.array(inns)
        if len(idx) > 5:
            d = orc.mh_distances(Js, P[b], inns, R_VIS)
            mk, _, _ = orc.mh_gate(d, MH, MULT, 5)
            assert rel_fro(dist[b][idx], d) < 1e-9
        else:
            mk = np.ones(len(idx), dtype=bool)
        full = np.zeros(nf, dtype=bool); full[idx] = mk
        assert np.array_equal(mask[b].astype(bool), full), (b, mask[b], full)
        k = idx[mk]
        H, iv, dR = orc.stack_measurements(Js[mk], inns[mk], sc["ref"][b][k], sc["sind"][b][k], lay, R_VIS)
        e_ref, P_ref, _ = orc.update_joseph(H, P[b], iv, dR)
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX
    assert mask[0, 2] == 1 and mask[1, 3] == 0 and mask[1, 9] == 0


# Stated tolerance of XIVO_HIP_FLAG_FP32_WHITENED over a CHAIN of updates on the resident covariance (BASELINE config 4 "fp32 MFMA
# with stated tolerance"), against the all-fp64 oracle chain:
#   P   relative Frobenius <= 5e-5 after any number of updates (measured 2.6e-5 / 8.8e-6 after 25 updates at (400,300) / (300,176))
#   dx  of update k is BIT-IDENTICAL to the fp64 path given the same prior (test_fp32_whitened_operands_beyond_one_workgroup);
#       against the fp64 CHAIN it inherits the float rounding of P - V^T Y in the directions the measurements have already
#       shrunk (a cancellation: those components are ~1e-3 of |P|, the float operands resolve 6e-8 |P|), and the gain of the
#       next update is made of exactly those: <= 5e-2 relative after 25 updates (measured 2.3e-2 / 7.0e-3), i.e. <= 0.05 of a
#       posterior standard deviation per state on average (asserted below as the Mahalanobis norm of the dx error).
# This is why the mode is opt-in and the library default is all fp64.
TOL_P_CHAIN, TOL_DX_CHAIN, TOL_DX_SIGMA = 5e-5, 5e-2, 5e-2


@pytest.mark.parametrize("N,F,steps", [(400, 150, 25), (300, 88, 25)])
def test_fp32_whitened_chain(built, N, F, steps):
    B = 2
    P0, _, _, _ = synth.s_level(N, F, B, seed=31 + N)
    meas = [synth.s_level(N, F, B, seed=1000 + 17 * k + N)[1:] for k in range(steps)]
    worst_P = worst_dx = worst_sig = 0.0
    with Context(N, 2 * F, B, flags=FLAG_FP32_WHITENED) as ctx:
        ctx.upload_P(P0)
        Pc = [P0[b].copy() for b in range(B)]
        for k in range(steps):
            H, inn, dR = meas[k]
            ctx.set_measurements(H, inn, dR); ctx.update_joseph()
            assert (ctx.get_status() == 0).all()
            err = ctx.get_err()
            for b in range(B):
                e_ref, Pc[b], _ = orc.update_joseph(H[b], Pc[b], inn[b], dR[b])
                worst_dx = max(worst_dx, rel_fro(err[b], e_ref))
                # the dx error in units of the posterior standard deviation: sqrt(e^T P+^-1 e / N)
                d = err[b] - e_ref
                worst_sig = max(worst_sig, float(np.sqrt(d @ np.linalg.solve(0.5 * (Pc[b] + Pc[b].T), d) / N)))
            if k % 6 == 5 or k == steps - 1:
                Pn = ctx.download_P()
                for b in range(B):
                    worst_P = max(worst_P, rel_fro(Pn[b], Pc[b]))
                    assert np.array_equal(Pn[b], Pn[b].T)
                    w = np.linalg.eigvalsh(Pn[b])
                    assert w.min() > -1e-9 * w.max(), (k, w.min(), w.max())          # stays PSD to rounding
    print("fp32-whitened chain of %d updates at N=%d M=%d: worst rel. error P %.2e, dx %.2e (%.2e posterior sigma)"
          % (steps, N, 2 * F, worst_P, worst_dx, worst_sig))
    assert worst_P < TOL_P_CHAIN and worst_dx < TOL_DX_CHAIN and worst_sig < TOL_DX_SIGMA
    assert worst_P > 1e-12                                                            # (it really took the float path)


def test_fp32_whitened_flag_leaves_in_solve_shapes_alone(built):
    """Shapes the in-solve update holds (N <= 256, M <= 176) are all fp64 under the flag at EVERY batch size - also the
    few-filter latency route, which evaluates the product outside the solve kernel (include/xivo_hip.h)."""
    N, F = 250, 80
    for B in (1, 3, 96):
        P, H, inn, dR = synth.s_level(N, F, B, seed=5 + B)
        out = []
        for fl in (0, FLAG_FP32_WHITENED):
            with Context(N, 2 * F, B, flags=fl) as ctx:
                ctx.upload_P(P); ctx.set_measurements(H, inn, dR); ctx.update_joseph()
                out.append((ctx.get_err(), ctx.download_P()))
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]), B
