"""Round 6: the one-kernel measurement update (xivo_amd/csrc/fused_update.hip) - Estimator::UpdateJosephForm
(src/estimator.cpp:1257-1288) with the numeric core of Estimator::MHGating (src/update.cpp:60-96) in front of it, for the shapes
one workgroup holds (TUM-VI build 203 / 60, BASELINE config 2 150 / 100) - against the oracle and against the multi-kernel
pipeline it replaces (XIVO_HIP_FLAG_MULTI_KERNEL)."""
import numpy as np
import pytest

import xivo_oracle as orc
from helpers import rel_fro, TOL_P, TOL_DX
from xivo_amd import synth
from xivo_amd.lib import Context, FLAG_PROFILE, FLAG_MULTI_KERNEL

pytestmark = pytest.mark.gpu

# (N, F): M = 2 F. Instantiations: <4,16,64> M <= 64 (slab 64 wide), <4,16,32> the same factor on a state whose 64-wide slab
# does not fit next to S, <7,12,32> M <= 112 on N <= 192
FUSED_SHAPES = [(203, 30), (150, 50), (192, 56), (147, 49), (180, 53), (224, 32), (250, 30), (256, 24), (160, 40), (100, 12), (64, 8), (37, 3), (100, 1)]
# the last three: so few features that the row-pair compression finds more than 12 "common" columns (columns most pairs
# name) - not XIVO's row structure; the one-kernel route declines and the multi-kernel pipeline runs (checked all the same)
NOT_FUSED = {(64, 8), (37, 3), (100, 1)}


def _gated_reference(P, H, inn, dR, mask):
    keep = np.repeat(mask.astype(bool), 2)
    return orc.update_joseph(H[keep], P, inn[keep], dR[keep])[:2]


@pytest.mark.parametrize("N,F", FUSED_SHAPES)
@pytest.mark.parametrize("B", [1, 70])
def test_fused_update_matches_oracle(built, N, F, B):
    """Gate + update in one kernel: masks equal to the oracle's MH gate, P+ 1e-6, dx 1e-8 on the surviving rows, symmetric
    output, for one filter (the reference's own deployment) and for a batch; the stage label says which kernel ran."""
    P, H, inn, dR = synth.s_level(N, F, 8, seed=5 * N + F)
    idx = np.arange(B) % 8
    P, H, inn, dR = P[idx].copy(), H[idx].copy(), inn[idx].copy(), dR[idx].copy()
    if F >= 8 and B > 5:
        inn[5, 4:8] *= 1e4                                      # two features of filter 5 fail the gate
    R = float(dR[0, 0])
    with Context(N, 2 * F, B, flags=FLAG_PROFILE) as ctx:
        ctx.upload_P(P); ctx.set_measurements(H, inn, dR)
        ctx.update_dense_gated(F, R, 5.991, 1.1, 5)
        assert ctx.last_path() == 1
        prof = ctx.profile_get()
        mask, dist = ctx.get_gate(F, B)
        st = ctx.get_status(check=False); used = ctx.get_ldlt_used()
        Pn, err = ctx.download_P(), ctx.get_err()
    if (N, F) not in NOT_FUSED:
        assert prof["trsm_gain"]["kernel"].startswith("fused_update_f64_kernel"), prof
        assert set(k for k, v in prof.items() if v["launches"]) <= {"trsm_gain", "other", "stack_H"}, prof   # (stack_H: the hand-over of the measurements)
    assert (st == 0).all() and used.sum() == 0
    for b in sorted(set([0, min(5, B - 1), B - 1])):
        if F > 5:                                   # (Estimator::OutlierRejection gates only when F > min_required_inliers_, src/manager.cpp:635)
            d_ref = orc.mh_distances(H[b].reshape(F, 2, N), P[b], inn[b].reshape(F, 2), R)
            m_ref = orc.mh_gate(d_ref, 5.991, 1.1, 5)[0]
            assert np.array_equal(mask[b].astype(bool), np.asarray(m_ref).astype(bool))
            assert np.allclose(dist[b], d_ref, rtol=1e-9, atol=0)
            keep = mask[b]
        else:
            keep = np.ones(F, dtype=bool)
        e_ref, P_ref = _gated_reference(P[b], H[b], inn[b], dR[b], keep)
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX
        assert np.array_equal(Pn[b], Pn[b].T)
    if F >= 8 and B > 5:
        assert not mask[5, 2:4].any()


@pytest.mark.parametrize("N,F", [(203, 30), (150, 50), (256, 24)])
def test_fused_update_against_the_multi_kernel_pipeline(built, N, F):
    """The same batch through the one-kernel route and through the five kernels it replaces (XIVO_HIP_FLAG_MULTI_KERNEL):
    identical masks, distances to rounding, P+ / dx far inside the tolerances (both evaluate the whitened Joseph form; the
    summation order of P H^T's private columns and the in-LDS factor differ at the rounding level)."""
    B = 40
    P, H, inn, dR = synth.s_level(N, F, 8, seed=N + 13 * F)
    idx = np.arange(B) % 8
    P, H, inn, dR = P[idx].copy(), H[idx].copy(), inn[idx].copy(), dR[idx].copy()
    inn[3, 4:8] *= 1e4                                          # features 2 and 3 of filter 3 fail the gate
    R = float(dR[0, 0])
    out = []
    for flags in (0, FLAG_MULTI_KERNEL):
        with Context(N, 2 * F, B, flags=flags | FLAG_PROFILE) as ctx:
            ctx.upload_P(P)
            dH = ctx.device_array(np.ascontiguousarray(np.transpose(H, (0, 2, 1))))
            dinn = ctx.device_array(inn); dRd = ctx.device_array(dR)
            ctx.set_measurements_device(dH, dinn, dRd, 2 * F, B)      # device hand-over: no dense copies alive
            ctx.update_dense_gated(F, R, 5.991, 1.1, 5)
            prof = ctx.profile_get()
            mask, dist = ctx.get_gate(F, B)
            out.append((ctx.download_P(), ctx.get_err(), mask.copy(), dist.copy(), prof["trsm_gain"]["kernel"]))
    assert out[0][4].startswith("fused_update") and not out[1][4].startswith("fused_update")
    assert np.array_equal(out[0][2], out[1][2]) and not out[0][2][3, 2:4].any()
    assert np.allclose(out[0][3], out[1][3], rtol=1e-10, atol=0)
    for b in range(B):
        assert rel_fro(out[0][0][b], out[1][0][b]) < 1e-10 and rel_fro(out[0][1][b], out[1][1][b]) < 1e-9


@pytest.mark.parametrize("N,F", [(203, 30), (150, 50), (192, 56)])
def test_fused_update_six_and_nine_private_slots(built, N, F):
    """The kernel walks six private slots per row pair when no pair of the batch uses more (synth.s_level's rows: 12 common + 6
    private columns) and nine otherwise (the in-state rows FillJacobianBlock stacks: group anchor 6 + feature 3): the same
    batch with three more non-zero columns per pair takes the nine-slot instantiation - both against the oracle, the stage
    label says which ran."""
    B = 5
    P, H, inn, dR = synth.s_level(N, F, B, seed=3 * N + F)
    rng = np.random.default_rng(N + F)
    H9 = H.copy()
    for b in range(B):
        for f in range(F):
            free = np.nonzero(H[b, 2 * f] == 0)[0]
            free = free[free >= 21]                             # behind the motion / extrinsics columns every pair names
            cols = rng.choice(free, 3, replace=False)
            for c_ in cols:
                H9[b, 2 * f:2 * f + 2, c_] = rng.normal(size=2) * np.abs(H[b, 2 * f:2 * f + 2]).max()
    for Hx, slots in ((H, 6), (H9, 9)):
        with Context(N, 2 * F, B, flags=FLAG_PROFILE) as ctx:
            ctx.upload_P(P); ctx.set_measurements(Hx, inn, dR); ctx.update_joseph()
            label = ctx.profile_get()["trsm_gain"]["kernel"]
            want = slots if (N, F) != (192, 56) else 9          # (the 32-column slab instantiations exist for nine slots only)
            assert ctx.last_route() == "fused" and label.startswith("fused_update_f64_kernel") and label.endswith(",%d>" % want), label
            assert (ctx.get_status() == 0).all()
            Pn, err = ctx.download_P(), ctx.get_err()
        for b in range(B):
            e_ref, P_ref, _ = orc.update_joseph(Hx[b], P[b], inn[b], dR[b])
            assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX
            assert np.array_equal(Pn[b], Pn[b].T)


def test_fused_update_without_gate_and_chained(built):
    """xivo_hip_update_joseph (no gating) on the one-kernel route, three updates in a row on the resident covariance."""
    N, F, B = 203, 30, 6
    P, H, inn, dR = synth.s_level(N, F, B, seed=21)
    Pref = P.copy()
    with Context(N, 2 * F, B, flags=FLAG_PROFILE) as ctx:
        ctx.upload_P(P)
        for it in range(3):
            _, H, inn, dR = synth.s_level(N, F, B, seed=200 + it)
            ctx.set_measurements(H, inn, dR)
            ctx.update_joseph()
            e_last = []
            for b in range(B):
                e, Pref[b], _ = orc.update_joseph(H[b], Pref[b], inn[b], dR[b])
                e_last.append(e)
        assert ctx.profile_get()["trsm_gain"]["kernel"].startswith("fused_update")
        Pn, err = ctx.download_P(), ctx.get_err()
    for b in range(B):
        assert rel_fro(Pn[b], Pref[b]) < TOL_P and rel_fro(err[b], e_last[b]) < TOL_DX


def test_fused_update_hands_a_broken_factor_to_the_ldlt_fallback(built):
    """An S the Cholesky cannot factor (a covariance that lost its definiteness: three eigenvalues of P flipped on one filter;
    R stays positive - the reference takes sqrt(R), src/estimator.cpp:1283): the one-kernel route reports it, leaves the prior,
    and the pivoted L D L^T fallback updates that filter the reference's way from the P H^T the kernel handed over."""
    N, F, B = 203, 30, 4
    P, H, inn, dR = synth.s_level(N, F, B, seed=77)
    w, Q = np.linalg.eigh(P[2])
    w[-3:] *= -1.0
    P[2] = (Q * w) @ Q.T
    P[2] = 0.5 * (P[2] + P[2].T)
    with Context(N, 2 * F, B, flags=FLAG_PROFILE) as ctx:
        ctx.upload_P(P); ctx.set_measurements(H, inn, dR); ctx.update_joseph()
        assert ctx.profile_get()["trsm_gain"]["kernel"].startswith("fused_update")
        st = ctx.get_status(check=False); used = ctx.get_ldlt_used()
        Pn, err = ctx.download_P(), ctx.get_err()
    assert (st == 0).all() and list(used) == [0, 0, 1, 0]
    for b in range(B):
        e_ref, P_ref, _ = orc.update_joseph(H[b], P[b], inn[b], dR[b])
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX


def test_fused_update_large_batch_every_filter(built):
    """More workgroups than CUs: every filter of a 600-filter batch is updated (checked through the trace of P+ against
    eight distinct references) and the results do not depend on the batch position."""
    N, F, B = 203, 30, 600
    P, H, inn, dR = synth.s_level(N, F, 8, seed=9)
    idx = np.arange(B) % 8
    with Context(N, 2 * F, B) as ctx:
        ctx.upload_P(P[idx]); ctx.set_measurements(H[idx], inn[idx], dR[idx])
        ctx.update_dense_gated(F, float(dR[0, 0]), 5.991, 1.1, 5)
        Pn, err = ctx.download_P(), ctx.get_err()
    for b in range(8, B):
        assert np.array_equal(Pn[b], Pn[b % 8]) and np.array_equal(err[b], err[b % 8])
