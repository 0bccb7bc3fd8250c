"""GPU parity of Estimator::UpdateJosephForm (src/estimator.cpp:1257-1288)
through the C ABI vs the numpy oracle, on the BASELINE.json synthetic sizes."""
import numpy as np
import pytest

import xivo_oracle as orc
from helpers import rel_fro, TOL_P, TOL_DX
from xivo_amd import synth
from xivo_amd.lib import Context, FLAG_DENSE_H, FLAG_THROUGHPUT_ROUTE, FLAG_MULTI_KERNEL

pytestmark = pytest.mark.gpu

CASES = [  # (N, F) -> M = 2F ; BASELINE.json configs + TUM-VI default build + ragged sizes
    (150, 50), (250, 80), (400, 150), (203, 30), (251, 130), (37, 3), (64, 8), (100, 1),
    (100, 192),   # M = 384 = the largest factor the solver is built for (24 blocks); M > N is legal
    (176, 88),    # Mp = Np = 176: the largest single-workgroup triangle of the block-list kernel
    (192, 96),    # first size that falls back to strip tiles / the streamed solve
    (600, 20),    # source too wide for the LDS slab: gather form of the compressed-row kernels
    # states wider than one workgroup / factors beyond the LDS: whitened outputs (V^T, Y^T) from the chunked or streamed
    # solve + one tiled product P - V^T Y (round 3; before: T, G, P+ from three stand-alone kernels)
    (300, 60),    # two 256-column chunks, factor in LDS (10 block rows)
    (300, 88),    # ... eleven block rows: streamed instead of the LDS kernel
    (512, 128),   # M = 256: sixteen block rows, streamed; four 128-column chunks
    (848, 125),   # the reference's largest contemplated build (125 features, 75 groups: src/CMakeLists.txt:27-28)
]


@pytest.mark.parametrize("N,F", CASES)
@pytest.mark.parametrize("kind", ["sparse", "dense", "sparse_forced_dense", "sparse_throughput_route"])
def test_update_joseph_matches_oracle(built, N, F, kind):
    """kind: XIVO row sparsity -> sparse-H pipeline (ell.h); dense H -> as-coded dense pipeline picked
    automatically; the same sparse H with XIVO_HIP_FLAG_DENSE_H -> as-coded dense pipeline. Five filters take the latency
    route where it exists (M <= 224: streamed solve + tiled product); sparse_throughput_route: the same update on the
    kernels sized for thousands of filters (XIVO_HIP_FLAG_THROUGHPUT_ROUTE: one workgroup per filter, update inside the solve)."""
    B = 5
    dense = kind == "dense"
    P, H, inn, dR = synth.s_level(N, F, B, seed=N * 7 + F, dense=dense)
    flags = {"sparse_forced_dense": FLAG_DENSE_H, "sparse_throughput_route": FLAG_THROUGHPUT_ROUTE}.get(kind, 0)
    kind = "sparse" if kind == "sparse_throughput_route" else kind
    with Context(N, 2 * F, B, flags=flags) as ctx:
        ctx.upload_P(P)
        ctx.set_measurements(H, inn, dR)
        ctx.update_joseph()
        # dense H with 2 features on a 37-dim state still fits the compressed form (<= 28 columns)
        assert ctx.last_path() == (1 if kind == "sparse" or (dense and N <= 28) else 0)
        err = ctx.get_err()
        Pn = ctx.download_P()
        assert (ctx.get_status() == 0).all()
    for b in range(B):
        e_ref, P_ref, _ = orc.update_joseph(H[b], P[b], inn[b], dR[b])
        assert rel_fro(Pn[b], P_ref) < TOL_P
        assert rel_fro(err[b], e_ref) < TOL_DX
        assert np.array_equal(Pn[b], Pn[b].T)  # lower triangle mirrored


def test_upload_download_roundtrip_bit_exact(built):
    N, B = 203, 4
    rng = np.random.default_rng(0)
    A = rng.normal(size=(B, N, N))
    P = np.tril(A) + np.transpose(np.tril(A, -1), (0, 2, 1))          # symmetric: comes back bit for bit
    with Context(N, 60, B) as ctx:
        ctx.upload_P(P)
        assert np.array_equal(ctx.download_P(), P)
        ctx.upload_P(P[2:3] * 2, b0=1)
        got = ctx.download_P()
        assert np.array_equal(got[1], P[2] * 2) and np.array_equal(got[0], P[0])
        # not symmetric: the lower triangle of what was uploaded is authoritative (include/xivo_hip.h), mirrored exactly
        ctx.upload_P(A)
        assert np.array_equal(ctx.download_P(), P)


def test_chained_updates_stay_consistent(built):
    """Three successive updates on the resident P (new measurements each time)."""
    N, F, B = 150, 50, 2
    P, H, inn, dR = synth.s_level(N, F, B, seed=11)
    Pref = P.copy()
    with Context(N, 2 * F, B) as ctx:
        ctx.upload_P(P)
        for it in range(3):
            _, H, inn, dR = synth.s_level(N, F, B, seed=100 + it)
            ctx.set_measurements(H, inn, dR)
            ctx.update_joseph()
            for b in range(B):
                _, Pref[b], _ = orc.update_joseph(H[b], Pref[b], inn[b], dR[b])
        Pn = ctx.download_P()
    for b in range(B):
        assert rel_fro(Pn[b], Pref[b]) < TOL_P


def test_smaller_M_after_larger_M(built):
    """Stale rows of a previous, larger measurement set must not leak."""
    N, B = 150, 2
    P, H1, inn1, dR1 = synth.s_level(N, 50, B, seed=1)
    _, H2, inn2, dR2 = synth.s_level(N, 20, B, seed=2)
    with Context(N, 100, B) as ctx:
        ctx.upload_P(P); ctx.set_measurements(H1, inn1, dR1); ctx.update_joseph()
        ctx.upload_P(P); ctx.set_measurements(H2, inn2, dR2); ctx.update_joseph()
        err = ctx.get_err(); Pn = ctx.download_P()
    for b in range(B):
        e_ref, P_ref, _ = orc.update_joseph(H2[b], P[b], inn2[b], dR2[b])
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX


def test_not_spd_is_reported_not_fatal(built):
    from xivo_amd.lib import XivoHipError
    N, F, B = 64, 8, 2
    P, H, inn, dR = synth.s_level(N, F, B, seed=5)
    dR[1, :] = -1e9  # S = HPH^T + R becomes indefinite for filter 1
    from xivo_amd.lib import FLAG_NO_LDLT_FALLBACK      # (default: the pivoted L D L^T fallback updates it, tests/test_robustness_gpu.py)
    with Context(N, 2 * F, B, flags=FLAG_NO_LDLT_FALLBACK) as ctx:
        ctx.upload_P(P); ctx.set_measurements(H, inn, dR); ctx.update_joseph()
        st = ctx.get_status(check=False)
        assert st[0] == 0 and st[1] != 0
        with pytest.raises(XivoHipError):
            ctx.get_status()


def _gating_case(N, F, B, seed):
    """Dense rows small enough that S_f ~ R, plus a few wild innovations per filter."""
    P, H, inn, dR = synth.s_level(N, F, B, seed=seed)
    H *= 0.01
    rng = np.random.default_rng(seed)
    for b in range(B):
        bad = rng.choice(F, size=3 + b, replace=False)
        inn[b].reshape(F, 2)[bad] += rng.choice([-1, 1], size=(len(bad), 2)) * rng.uniform(8, 30, size=(len(bad), 2))
    return P, H, inn, dR


def _oracle_gate(P, H, inn, R, thresh, mult, min_inl):
    F = H.shape[0] // 2
    J = H.reshape(F, 2, -1)
    d = orc.mh_distances(J, P, inn.reshape(F, 2), R)
    return orc.mh_gate(d, thresh, mult, min_inl)[0], d


@pytest.mark.parametrize("flags", [0, FLAG_DENSE_H])
def test_mh_gate_dense_matches_oracle(built, flags):
    N, F, B = 150, 50, 4
    P, H, inn, dR = _gating_case(N, F, B, 21)
    with Context(N, 2 * F, B, flags=flags) as ctx:
        ctx.upload_P(P); ctx.set_measurements(H, inn, dR)
        mask, dist = ctx.mh_gate_dense(F, 2.25, 5.991, 1.1, 5)
        ctx.update_joseph()
        err = ctx.get_err(); Pn = ctx.download_P()
    for b in range(B):
        m, d = _oracle_gate(P[b], H[b], inn[b], 2.25, 5.991, 1.1, 5)
        assert rel_fro(dist[b], d) < 1e-9 and np.array_equal(mask[b], m) and (~m).sum() >= 3
        rows = np.repeat(m, 2)
        e_ref, P_ref, _ = orc.update_joseph(H[b][rows], P[b], inn[b][rows], dR[b][rows])   # rejected rows not stacked
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX


@pytest.mark.parametrize("flags", [0, FLAG_DENSE_H])
@pytest.mark.parametrize("N,F", [(150, 50), (250, 80)])
def test_update_dense_gated_single_pass(built, N, F, flags):
    B = 3
    P, H, inn, dR = _gating_case(N, F, B, 33)
    with Context(N, 2 * F, B, flags=flags) as ctx:
        ctx.upload_P(P); ctx.set_measurements(H, inn, dR)
        ctx.update_dense_gated(F, 2.25, 5.991, 1.1, 5)
        mask, dist = ctx.get_gate(F)
        err = ctx.get_err(); Pn = ctx.download_P()
        assert (ctx.get_status() == 0).all()
    for b in range(B):
        m, d = _oracle_gate(P[b], H[b], inn[b], 2.25, 5.991, 1.1, 5)
        assert rel_fro(dist[b], d) < 1e-9 and np.array_equal(mask[b], m)
        rows = np.repeat(m, 2)
        e_ref, P_ref, _ = orc.update_joseph(H[b][rows], P[b], inn[b][rows], dR[b][rows])
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX


def test_gating_skipped_when_too_few_features(built):
    """F <= min_required_inliers_: OutlierRejection does not gate (src/manager.cpp:635)."""
    N, F, B = 64, 4, 2
    P, H, inn, dR = _gating_case(N, F, B, 5)
    with Context(N, 2 * F, B) as ctx:
        ctx.upload_P(P); ctx.set_measurements(H, inn, dR)
        ctx.update_dense_gated(F, 2.25, 5.991, 1.1, 5)
        err = ctx.get_err(); Pn = ctx.download_P()
    for b in range(B):
        e_ref, P_ref, _ = orc.update_joseph(H[b], P[b], inn[b], dR[b])
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX


def test_mixed_batch_falls_back_to_dense_path(built):
    """One filter with a dense H in the batch: the whole call takes dense rows (the whitened update on dense H P / S)."""
    N, F, B = 150, 50, 4
    P, H, inn, dR = synth.s_level(N, F, B, seed=77)
    H[2] = synth.s_level(N, F, 1, seed=78, dense=True)[1][0]
    with Context(N, 2 * F, B) as ctx:
        ctx.upload_P(P); ctx.set_measurements(H, inn, dR); ctx.update_joseph()
        assert ctx.last_path() == 0
        ctx.upload_P(P); ctx.update_joseph(B=2)          # filters 0..1 alone are sparse
        assert ctx.last_path() == 1
        err = ctx.get_err(0, 2); Pn = ctx.download_P(0, 2)
    for b in range(2):
        e_ref, P_ref, _ = orc.update_joseph(H[b], P[b], inn[b], dR[b])
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX


def test_sparse_path_with_structural_zero_in_common_column(built):
    """A feature whose Jacobian has exact zeros in the shared pose columns, an all-zero row pair in the middle,
    and a pair with 13 private columns (does not fit -> dense fallback for that batch)."""
    N, F, B = 120, 20, 2
    P, H, inn, dR = synth.s_level(N, F, B, seed=5)
    H[:, 6:8, 0:6] = 0.0          # feature 3: no dependence on the sensor pose
    H[:, 10:12, :] = 0.0          # feature 5: empty pair
    with Context(N, 2 * F, B) as ctx:
        ctx.upload_P(P); ctx.set_measurements(H, inn, dR); ctx.update_joseph()
        assert ctx.last_path() == 1
        err = ctx.get_err(); Pn = ctx.download_P()
        for b in range(B):
            e_ref, P_ref, _ = orc.update_joseph(H[b], P[b], inn[b], dR[b])
            assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX
        H2 = H.copy()
        H2[1, 4, 60:73] = 1.0     # 13 extra private columns in one row
        ctx.upload_P(P); ctx.set_measurements(H2, inn, dR); ctx.update_joseph()
        assert ctx.last_path() == 0
        err = ctx.get_err(); Pn = ctx.download_P()
        for b in range(B):
            e_ref, P_ref, _ = orc.update_joseph(H2[b], P[b], inn[b], dR[b])
            assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX


@pytest.mark.parametrize("B", [16, 96])
def test_profile_reports_kernels_bytes_and_path(built, B):
    """What bench.py builds its roofline from: per-stage kernel names (as rocprofv3 spells them, minus spaces),
    algorithmic flops / bytes and which pipeline ran. B = 16 takes the latency route (at most 64 filters: the solve on
    128-column workgroups of the streamed kernel + the tiled product), B = 96 one workgroup per filter."""
    from xivo_amd.lib import FLAG_PROFILE
    N, F = 250, 80
    latency = B <= 64
    P, H, inn, dR = synth.s_level(N, F, B, seed=1)
    for flags, path, hp_kernel in ((FLAG_PROFILE, 1, "ell_tile_kernel<0,12,64,9>"),
                                   (FLAG_PROFILE | FLAG_DENSE_H, 0, "gemm_nt_f64_kernel<5,4,double>")):
        with Context(N, 2 * F, B, flags=flags) as ctx:
            ctx.upload_P(P); ctx.set_measurements(H, inn, dR)
            ctx.update_dense_gated(F, 2.25, 5.991, 1.1, 5)
            prof = ctx.profile_get()
            assert ctx.last_path() == path
        assert prof["gemm_HP"]["kernel"] == hp_kernel
        # sparse pipeline: the solve kernel carries the whole covariance update on the gain in its registers - no stand-alone
        # T / G / P+ kernels (latency route: the whitened outputs leave the solve and one tiled product follows)
        tail = ("gemm_Pnew",) if latency and path == 1 else (() if path == 1 else ("gemm_AP", "gemm_Pnew"))
        stages = ("gemm_HP", "gemm_S", "chol_S", "trsm_gain") + tail
        for st in stages:
            assert prof[st]["launches"] == 1 and prof[st]["ms"] > 0 and prof[st]["kernel"]
            assert prof[st]["bytes_per_launch"] > 0 and prof[st]["flops_per_launch"] > 0
        if path == 1:
            assert prof["gemm_AP"]["launches"] == 0 and prof["gemm_KH_I"]["launches"] == 0
            assert prof["gemm_Pnew"]["launches"] == (1 if latency else 0)
        assert prof["chol_S"]["kernel"].startswith("chol_reg_f64_kernel<10")   # B < 512: the latency kernel
        if path == 1:
            assert prof["trsm_gain"]["kernel"] == ("trsm_stream_f64_kernel<14,1,4>" if latency else "trsm_lds_f64_kernel<10,4>")
        else:
            assert prof["trsm_gain"]["kernel"] == "trsm_lds_f64_kernel<10,0>"


@pytest.mark.parametrize("flags", [0, FLAG_DENSE_H])
def test_ill_conditioned_innovation_covariance(built, flags):
    """cond(S) ~ 1e7 (P with eigenvalues over 8 decades, R = 1e-6): the gain carries ~1e-7 relative error, which
    is exactly what the Joseph correction term is for. Both pipelines must stay inside the tolerances and report
    SPD factors; the sparse pipeline forms that correction on the fp32 MFMA."""
    N, F, B = 150, 40, 3
    rng = np.random.default_rng(3)
    _, H, inn, _ = synth.s_level(N, F, B, seed=5)
    H *= 0.05
    P = np.empty((B, N, N))
    for b in range(B):
        Q, _ = np.linalg.qr(rng.normal(size=(N, N)))
        P[b] = (Q * np.logspace(-8, 0, N)) @ Q.T
        P[b] = 0.5 * (P[b] + P[b].T)
    dR = np.full((B, 2 * F), 1e-6)
    with Context(N, 2 * F, B, flags=flags) as ctx:
        ctx.upload_P(P); ctx.set_measurements(H, inn, dR); ctx.update_joseph()
        assert (ctx.get_status() == 0).all()
        err = ctx.get_err(); Pn = ctx.download_P()
    for b in range(B):
        S = H[b] @ P[b] @ H[b].T + np.diag(dR[b])
        assert np.linalg.cond(S) > 1e6
        e_ref, P_ref, _ = orc.update_joseph(H[b], P[b], inn[b], dR[b])
        assert rel_fro(Pn[b], P_ref) < TOL_P
        assert rel_fro(err[b], e_ref) < 1e-6     # dx inherits cond(S) * eps from ANY solver; 1e-8 is for cond ~ 1e3
        w = np.linalg.eigvalsh(Pn[b])
        assert w.min() > -1e-9 * w.max()


# ---------------------------------------------------------------- XIVO_HIP_FLAG_SYMMETRIC_FORM
SYM_CASES = [(150, 50), (250, 80), (400, 150), (203, 30), (37, 3), (100, 192), (600, 20)]


@pytest.mark.parametrize("N,F", SYM_CASES)
@pytest.mark.parametrize("kind", ["sparse", "dense"])
def test_symmetric_form_matches_the_reference_update(built, N, F, kind):
    """P+ = P - W^T W, dx = W^T L^-1 inn with W = L^-1 (H P): what the Joseph form of estimator.cpp:1276-1287 evaluates to
    for the optimal gain, without the backward solve and the correction product. Same tolerances as the default."""
    from xivo_amd.lib import FLAG_SYMMETRIC_FORM
    B = 4
    P, H, inn, dR = synth.s_level(N, F, B, seed=N * 3 + F, dense=kind == "dense")
    with Context(N, 2 * F, B, flags=FLAG_SYMMETRIC_FORM) as ctx:
        ctx.upload_P(P); ctx.set_measurements(H, inn, dR); ctx.update_joseph()
        err = ctx.get_err(); Pn = ctx.download_P()
        assert (ctx.get_status() == 0).all()
    for b in range(B):
        e_ref, P_ref, _ = orc.update_joseph(H[b], P[b], inn[b], dR[b])
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX
        assert np.array_equal(Pn[b], Pn[b].T)


def test_symmetric_form_gated_ill_conditioned_and_not_spd(built):
    from xivo_amd.lib import FLAG_SYMMETRIC_FORM
    # gating in front of it (the bench's step)
    N, F, B = 250, 80, 3
    P, H, inn, dR = _gating_case(N, F, B, 41)
    with Context(N, 2 * F, B, flags=FLAG_SYMMETRIC_FORM) as ctx:
        ctx.upload_P(P); ctx.set_measurements(H, inn, dR)
        ctx.update_dense_gated(F, 2.25, 5.991, 1.1, 5)
        mask, _ = ctx.get_gate(F)
        err = ctx.get_err(); Pn = ctx.download_P()
    for b in range(B):
        m, _ = _oracle_gate(P[b], H[b], inn[b], 2.25, 5.991, 1.1, 5)
        assert np.array_equal(mask[b], m) and not m.all()
        rows = np.repeat(m, 2)
        e_ref, P_ref, _ = orc.update_joseph(H[b][rows], P[b], inn[b][rows], dR[b][rows])
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX
    # cond(S) ~ 1e7: the error of this form scales with cond(L) = sqrt(cond(S))
    N, F, B = 150, 40, 3
    rng = np.random.default_rng(3)
    _, H, inn, _ = synth.s_level(N, F, B, seed=5)
    H *= 0.05
    P = np.empty((B, N, N))
    for b in range(B):
        Q, _ = np.linalg.qr(rng.normal(size=(N, N)))
        P[b] = (Q * np.logspace(-8, 0, N)) @ Q.T
        P[b] = 0.5 * (P[b] + P[b].T)
    dR = np.full((B, 2 * F), 1e-6)
    dR[1, 7] = -1e12                           # and one filter whose S is not positive definite: prior kept, reported
    from xivo_amd.lib import FLAG_NO_LDLT_FALLBACK
    with Context(N, 2 * F, B, flags=FLAG_SYMMETRIC_FORM | FLAG_NO_LDLT_FALLBACK) as ctx:
        ctx.upload_P(P); ctx.set_measurements(H, inn, dR); ctx.update_joseph()
        st = ctx.get_status(check=False)
        err = ctx.get_err(); Pn = ctx.download_P()
    assert st[1] != 0 and st[0] == 0 and st[2] == 0 and np.array_equal(Pn[1], P[1])
    for b in (0, 2):
        e_ref, P_ref, _ = orc.update_joseph(H[b], P[b], inn[b], dR[b])
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < 1e-6
        w = np.linalg.eigvalsh(Pn[b])
        assert w.min() > -1e-9 * w.max()


@pytest.mark.parametrize("N,F", [(250, 80), (150, 50), (203, 30), (37, 3)])
def test_standalone_tail_flag_matches_the_in_solve_update(built, N, F):
    """XIVO_HIP_FLAG_STANDALONE_TAIL: the covariance update from stand-alone kernels (re-associated expression) instead of
    inside the solve kernel (whitened expression). Two rounding-level re-orderings of the same update: they agree with each
    other and with the oracle far inside the tolerances, dx is identical (the gain is the same solve)."""
    from xivo_amd.lib import FLAG_STANDALONE_TAIL, FLAG_PROFILE
    B = 3
    P, H, inn, dR = synth.s_level(N, F, B, seed=17)
    outs, errs, kern = [], [], []
    for flags in (FLAG_THROUGHPUT_ROUTE | FLAG_MULTI_KERNEL, FLAG_STANDALONE_TAIL):   # (the multi-kernel pipeline's two tails)
        with Context(N, 2 * F, B, flags=flags | FLAG_PROFILE) as ctx:
            ctx.upload_P(P); ctx.set_measurements(H, inn, dR); ctx.update_joseph()
            assert ctx.last_path() == 1
            outs.append(ctx.download_P()); errs.append(ctx.get_err())
            kern.append({k: v["kernel"] for k, v in ctx.profile_get().items() if v["launches"]})
    # (the in-solve whitened form: <NBM,4>, or its ten-wave instantiation <7,4,false,10,3> for seven block rows on <= 160 columns)
    assert (kern[0]["trsm_gain"].endswith(",4>") or kern[0]["trsm_gain"].startswith("trsm_lds_f64_kernel<7,4,")) and "gemm_Pnew" not in kern[0]
    assert kern[1]["trsm_gain"].endswith(",1>") and "gemm_Pnew" in kern[1] and "gemm_KH_I" in kern[1]
    assert rel_fro(errs[0], errs[1]) < 1e-13      # same solve; the in-solve variant sums dx = K inn block by block as the gain appears
    assert rel_fro(outs[0], outs[1]) < 1e-11
    for b in range(B):
        _, P_ref, _ = orc.update_joseph(H[b], P[b], inn[b], dR[b])
        assert rel_fro(outs[0][b], P_ref) < 1e-10 and rel_fro(outs[1][b], P_ref) < 1e-10


# ---------------------------------------------------------------- the two in-solve evaluations of the Joseph expression
@pytest.mark.parametrize("N,F", [(250, 80), (250, 88), (150, 50), (203, 30), (37, 3), (64, 8), (100, 1), (256, 72)])
def test_whitened_form_in_the_solve_kernel_and_on_the_latency_route_agree(built, N, F):
    """P+ = P - (W - D)^T (W + D) (W = L^-1 H P, D = the backward substitution's own residual) - the Joseph expression of
    src/estimator.cpp:1276-1287 for the computed gain with S = L L^T - inside the solve kernel (one workgroup per filter) and
    from the streamed solve's whitened outputs + the tiled product (nine filters without a flag: the latency route): the two
    agree with each other and with the oracle far inside the tolerances. (250, 88): M = 176, eleven block rows - the
    packed-diagonal instantiation. (The multi-kernel pipeline's two executions; the one-kernel route: test_fused_gpu.py.)"""
    from xivo_amd.lib import FLAG_PROFILE
    B = 9                                    # more than one XCD group of 8
    P, H, inn, dR = synth.s_level(N, F, B, seed=23)
    outs, errs, kern, routes = [], [], [], []
    for flags in (FLAG_THROUGHPUT_ROUTE | FLAG_MULTI_KERNEL, FLAG_MULTI_KERNEL):
        with Context(N, 2 * F, B, flags=flags | FLAG_PROFILE) as ctx:
            ctx.upload_P(P); ctx.set_measurements(H, inn, dR); ctx.update_joseph()
            assert ctx.last_path() == 1 and (ctx.get_status() == 0).all()
            outs.append(ctx.download_P()); errs.append(ctx.get_err())
            kern.append(ctx.profile_get()["trsm_gain"]["kernel"]); routes.append(ctx.last_route())
    assert routes == ["sparse_in_solve", "sparse_whitened"]
    assert (kern[0].endswith(",4>") or kern[0].startswith("trsm_lds_f64_kernel<7,4,")) and kern[1].startswith("trsm_stream_f64_kernel<")
    assert rel_fro(errs[0], errs[1]) < 1e-13 and rel_fro(outs[0], outs[1]) < 1e-11
    for b in range(B):
        e_ref, P_ref, _ = orc.update_joseph(H[b], P[b], inn[b], dR[b])
        for o, e in zip(outs, errs):
            assert rel_fro(o[b], P_ref) < 1e-10 and rel_fro(e[b], e_ref) < TOL_DX
            assert np.array_equal(o[b], o[b].T)


@pytest.mark.parametrize("route", [FLAG_THROUGHPUT_ROUTE, 0])
def test_whitened_form_ill_conditioned_and_not_spd(built, route):
    """cond(S) ~ 1e7 plus one filter whose S is indefinite: tolerances as for every other evaluation, the indefinite filter
    keeps its prior bit for bit, the smallest eigenvalue of P+ stays at rounding level. Both routes of the whitened form
    (in-solve update; streamed solve + tiled product for few filters)."""
    N, F, B = 150, 40, 4
    rng = np.random.default_rng(31)
    _, H, inn, _ = synth.s_level(N, F, B, seed=6)
    H *= 0.05
    P = np.empty((B, N, N))
    for b in range(B):
        Q, _ = np.linalg.qr(rng.normal(size=(N, N)))
        P[b] = (Q * np.logspace(-8, 0, N)) @ Q.T
        P[b] = 0.5 * (P[b] + P[b].T)
    dR = np.full((B, 2 * F), 1e-6)
    dR[2, :] = -1e3
    from xivo_amd.lib import FLAG_NO_LDLT_FALLBACK
    with Context(N, 2 * F, B, flags=FLAG_NO_LDLT_FALLBACK | route) as ctx:
        ctx.upload_P(P); ctx.set_measurements(H, inn, dR); ctx.update_joseph()
        st = ctx.get_status(check=False)
        err = ctx.get_err(); Pn = ctx.download_P()
    assert st[2] != 0 and (st[[0, 1, 3]] == 0).all()
    assert np.array_equal(Pn[2], P[2])
    for b in (0, 1, 3):
        e_ref, P_ref, _ = orc.update_joseph(H[b], P[b], inn[b], dR[b])
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < 1e-6
        w = np.linalg.eigvalsh(Pn[b])
        assert w.min() > -1e-9 * w.max()


_CHOL_SNIPPET = r"""
import sys, json, hashlib
sys.path.insert(0, {root!r})
import numpy as np
from xivo_amd import synth
from xivo_amd.lib import Context
out = {{}}
for (N, F) in [(150, 50), (250, 80), (203, 30), (64, 20), (100, 96), (100, 110), (120, 150)]:   # 7, 10, 4, 3, 12 block rows; 14 and 19: the eight-wave register kernel
    B = 520                                   # >= 512: the size class where the pick matters
    P, H, inn, dR = synth.s_level(N, F, 8, seed=41)
    idx = np.arange(B) % 8
    with Context(N, 2 * F, B, flags=65536) as ctx:      # XIVO_HIP_FLAG_MULTI_KERNEL: the pipeline with a stand-alone factorisation
        ctx.upload_P(P[idx]); ctx.set_measurements(H[idx], inn[idx], dR[idx]); ctx.update_joseph()
        Pn = ctx.download_P(); err = ctx.get_err()
    out["%d,%d" % (N, F)] = [hashlib.sha256(Pn.tobytes()).hexdigest(), hashlib.sha256(err.tobytes()).hexdigest()]
print(json.dumps(out))
"""


def test_cholesky_kernels_are_bit_identical(built):
    """The one-wave and the four-wave register Cholesky run the same arithmetic in the same order (factor_invert_diag on
    the matrix pipe, pivot_scale, two accumulators per block product): whichever kernel runs (XIVO_HIP_CHOL_WAVE: the one-wave
    kernel for every size - the one A/B knob the factorisation keeps) and whichever instantiation of the register kernel
    (four waves per factor or eight, three workgroups per CU / two), P+ and dx come out bit for bit the same - across nodes,
    ranks, runs and batch sizes."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for knob in ((), ("XIVO_HIP_CHOL_WAVE",)):
        env = dict(os.environ)
        env.pop("XIVO_HIP_CHOL_WAVE", None)
        for k in knob:
            env[k] = "1"
        r = subprocess.run([sys.executable, "-c", _CHOL_SNIPPET.format(root=root)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert res[0] == res[1], res


@pytest.mark.parametrize("N,F,waves", [(150, 50, 10), (147, 49, 10), (180, 53, 12), (192, 56, 12)])
def test_seven_block_rows_on_a_narrow_state_keep_W_in_registers(built, N, F, waves):
    """Round 5: a factor of seven block rows (M <= 112) on a state of at most 160 / 192 columns takes the ten- / twelve-wave
    instantiation of the solve kernel whose 170 VGPRs per wave hold the right-hand sides AND W (no stash through HBM):
    P+ and dx against the oracle, rejected features included."""
    from xivo_amd.lib import FLAG_PROFILE
    B = 70
    P, H, inn, dR = synth.s_level(N, F, 8, seed=3 * N + F)
    idx = np.arange(B) % 8
    P, H, inn, dR = P[idx].copy(), H[idx].copy(), inn[idx].copy(), dR[idx].copy()
    inn[5, 4:8] *= 1e4                                      # two features of filter 5 fail the gate
    with Context(N, 2 * F, B, flags=FLAG_THROUGHPUT_ROUTE | FLAG_MULTI_KERNEL | FLAG_PROFILE) as ctx:   # (round 6: these shapes default to the one-kernel update)
        ctx.upload_P(P); ctx.set_measurements(H, inn, dR)
        ctx.update_dense_gated(F, float(dR[0, 0]), 5.991, 1.1, 5)
        assert ctx.last_path() == 1
        prof = ctx.profile_get()
        mask, _ = ctx.get_gate(F, B)
        st = ctx.get_status(check=False); used = ctx.get_ldlt_used()
        Pn, err = ctx.download_P(), ctx.get_err()
    assert prof["trsm_gain"]["kernel"] == "trsm_lds_f64_kernel<7,4,%d,3>" % waves
    assert (st == 0).all() and used.sum() == 0
    assert not mask[5, 2:4].any() and mask[0].all()
    for b in (0, 3, 5, 9, 13, B - 1):
        keep = np.repeat(mask[b].astype(bool), 2)
        e_ref, P_ref, _ = orc.update_joseph(H[b][keep], P[b], inn[b][keep], dR[b][keep])
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX
        assert np.array_equal(Pn[b], Pn[b].T)


@pytest.mark.parametrize("N,F", [(150, 50), (250, 80), (37, 3), (400, 150)])
def test_update_matches_the_eigen_driver_directly(built, N, F):
    """Same inputs through the HIP path and through oracle/_ref's EXTRACTED build - the reference's own text of
    Estimator::UpdateJosephForm (src/estimator.cpp:1257-1288) cut out of the reference tree and compiled against its Eigen
    3.3.9 (the prebuilt library travels to the GPU box) - without the numpy oracle in between."""
    try:
        import ref_binding
        ref = ref_binding.loadx(203)      # (UpdateJosephForm's members are dynamic: the N = 203 library serves any N)
    except Exception as e:
        pytest.skip(f"oracle/_ref extracted build not available here: {e}")
    B = 3
    P, H, inn, dR = synth.s_level(N, F, B, seed=7 * N + F)
    with Context(N, 2 * F, B) as ctx:
        ctx.upload_P(P); ctx.set_measurements(H, inn, dR); ctx.update_joseph()
        err = ctx.get_err(); Pn = ctx.download_P()
    for b in range(B):
        e_ref, P_ref = ref.update_joseph(H[b], P[b], inn[b], dR[b])[:2]
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX


def _route_cases():
    """Every route of plan_update (xivo_amd/csrc/capi.hip) on seeded random shapes: (flags, dense H?, filters) -> the route the
    table names for it. Shapes: state dim 24..419, 1..149 features."""
    from xivo_amd.lib import FLAG_DENSE_H, FLAG_SYMMETRIC_FORM, FLAG_STANDALONE_TAIL
    rng = np.random.default_rng(20260924)
    modes = [(0, False, None), (FLAG_THROUGHPUT_ROUTE, False, None), (FLAG_MULTI_KERNEL, False, None),
             (FLAG_MULTI_KERNEL | FLAG_THROUGHPUT_ROUTE, False, None), (FLAG_SYMMETRIC_FORM, False, "sparse_symmetric"),
             (FLAG_STANDALONE_TAIL, False, "sparse_tail"), (FLAG_DENSE_H, False, "dense_ascoded"), (0, True, "dense_whitened"),
             (FLAG_THROUGHPUT_ROUTE, True, "dense_whitened"), (FLAG_SYMMETRIC_FORM, True, "dense_symmetric")]
    cases = []
    for i in range(40):
        N = int(rng.integers(24, 420))
        F = int(rng.integers(1, min(96, max(2, N // 3)) + 1))
        if i % 6 == 5:
            F = int(rng.integers(90, 150))                      # factors beyond the LDS: streamed solve
        B = int(rng.integers(1, 10))
        fl, dense, route = modes[i % len(modes)]
        cases.append((N, F, B, fl, dense, route, 1000 + i))
    return cases


@pytest.mark.parametrize("N,F,B,flags,dense,route,seed", _route_cases())
def test_every_route_of_the_plan(built, N, F, B, flags, dense, route, seed):
    """Seeded random shapes through every route of the one table that selects them (plan_update in capi.hip): the one-kernel
    update, the whitened form inside the solve kernel / from the whitened outputs (chunked, streamed, the latency route), the
    stand-alone tail, the symmetric form, the as-coded dense sequence, the whitened update on dense rows. The route the
    library reports is one the table allows for the flags; P+ 1e-6, dx 1e-8, symmetric output."""
    P, H, inn, dR = synth.s_level(N, F, B, seed=seed, dense=dense)
    with Context(N, 2 * F, B, flags=flags) as ctx:
        ctx.upload_P(P); ctx.set_measurements(H, inn, dR); ctx.update_joseph()
        assert (ctx.get_status() == 0).all() and not ctx.get_ldlt_used().any()
        err = ctx.get_err(); Pn = ctx.download_P()
        got = ctx.last_route()
    if dense and N <= 28:
        route = None                                # (a dense H of at most 28 columns still fits the compressed form)
    if route is not None:
        assert got == route, (got, route)
    elif flags & FLAG_MULTI_KERNEL:
        assert got in ("sparse_in_solve", "sparse_whitened"), got
    else:
        assert got in ("fused", "sparse_in_solve", "sparse_whitened"), got
    if (flags & FLAG_THROUGHPUT_ROUTE) and got == "sparse_whitened":
        assert N > 256 or 2 * F > 176               # only a shape one workgroup does not hold leaves the solve kernel then
    for b in range(B):
        e_ref, P_ref, _ = orc.update_joseph(H[b], P[b], inn[b], dR[b])
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX
        assert np.array_equal(Pn[b], Pn[b].T)


TOL_P_FP32 = 5e-5   # stated tolerance of XIVO_HIP_FLAG_FP32_WHITENED (config 4): relative Frobenius on P+


@pytest.mark.parametrize("N,F,flags", [(400, 150, 0), (300, 60, 0), (512, 64, 0), (251, 100, 0)])
def test_fp32_whitened_operands_beyond_one_workgroup(built, N, F, flags):
    """XIVO_HIP_FLAG_FP32_WHITENED (round 4; BASELINE config 4 "fp32 MFMA with stated tolerance"): for shapes whose product
    runs outside the solve kernel the whitened operands V^T, Y^T leave the fp64 solve as float and P - V^T Y runs on
    v_mfma_f32_16x16x4_f32. dx is untouched (1e-8: everything up to the gain is fp64), P+ within the stated 5e-5 (measured
    ~1e-7: the float rounding of the operands), same status."""
    from xivo_amd.lib import FLAG_FP32_WHITENED
    B = 3
    P, H, inn, dR = synth.s_level(N, F, B, seed=900 + N + F)
    out = {}
    for fl in (flags, flags | FLAG_FP32_WHITENED):
        with Context(N, 2 * F, B, flags=fl) as ctx:
            ctx.upload_P(P); ctx.set_measurements(H, inn, dR); ctx.update_joseph()
            assert (ctx.get_status() == 0).all()
            out[fl] = (ctx.get_err(), ctx.download_P())
    worst = 0.0
    for b in range(B):
        e_ref, P_ref, _ = orc.update_joseph(H[b], P[b], inn[b], dR[b])
        e32, P32 = out[flags | FLAG_FP32_WHITENED][0][b], out[flags | FLAG_FP32_WHITENED][1][b]
        assert rel_fro(e32, e_ref) < TOL_DX and rel_fro(P32, P_ref) < TOL_P_FP32
        assert np.array_equal(e32, out[flags][0][b])                 # dx: the very same fp64 arithmetic
        worst = max(worst, rel_fro(P32, P_ref))
        assert np.array_equal(P32, P32.T)
    assert worst > 1e-12                                             # (it really took the float path)
    print("fp32 whitened operands: worst rel. Frobenius error on P+ = %.2e at N=%d M=%d" % (worst, N, 2 * F))


def test_persistent_slab_kernel_ragged_batch(built):
    """1043 filters (not a multiple of 8, more than two per CU): `ell<S>` runs its persistent form - one workgroup per CU
    walks filters b, b + 256, ... with the next filter's first slab prefetched under the last walk - and the last round is
    ragged. Every filter must equal its twin built from the same source filter bit for bit, and the oracle on a sample;
    gating on (two features per filter pushed out)."""
    N, F, B, U = 250, 80, 1043, 7
    P1, H1, inn1, dR1 = _gating_case(N, F, U, 91)
    idx = np.arange(B) % U
    P, H, inn, dR = P1[idx], H1[idx], inn1[idx], dR1[idx]
    with Context(N, 2 * F, B) as ctx:
        ctx.upload_P(P); ctx.set_measurements(H, inn, dR)
        ctx.update_dense_gated(F, 2.25, 5.991, 1.1, 5)
        assert ctx.last_path() == 1
        err = ctx.get_err(); Pn = ctx.download_P(); mask = ctx.get_gate(F)[0].astype(bool)
        assert (ctx.get_status() == 0).all()
    for b in range(U, B):
        assert np.array_equal(err[b], err[b - U]) and np.array_equal(mask[b], mask[b - U])
    for b in range(U, B, 13):
        assert np.array_equal(Pn[b], Pn[b % U])
    for b in (0, 3, 6):
        m, d = _oracle_gate(P[b], H[b], inn[b], 2.25, 5.991, 1.1, 5)
        assert np.array_equal(mask[b], m) and (~m).sum() >= 2
        rows = np.repeat(m, 2)
        e_ref, P_ref, _ = orc.update_joseph(H[b][rows], P[b], inn[b][rows], dR[b][rows])
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX


@pytest.mark.parametrize("N,F,steps,r_scale", [(100, 20, 120, 1.0), (250, 80, 40, 1.0), (150, 60, 30, 1.0), (250, 80, 30, 1e-3)])
def test_long_chain_of_default_form_updates_stays_psd_and_close(built, N, F, steps, r_scale):
    """The default device form P - (W - D)^T (W + D) is PSD only up to rounding, not by construction as the as-coded
    product (I - KH) P (I - KH)^T + K R K^T is. A long chain on the resident covariance - every update shrinks P along new
    directions, cond(S) grows step by step, no noise is added in between - must stay symmetric, positive semi-definite and
    next to the as-coded chain of the oracle: relative distance 1e-6 at every checkpoint (the per-update tolerance, not
    accumulated), smallest eigenvalue above -1e-12 of the largest. (150, 60): M = 120 rows against N = 150 columns - nearly as many
    measurements as states in every update; r_scale 1e-3: measurement noise a thousand times smaller, cond(S) correspondingly larger.)"""
    B = 2
    P, _, _, _ = synth.s_level(N, F, B, seed=5)
    Pref = P.copy()
    with Context(N, 2 * F, B) as ctx:
        ctx.upload_P(P)
        for it in range(steps):
            _, H, inn, dR = synth.s_level(N, F, B, seed=700 + it)
            dR = dR * r_scale
            ctx.set_measurements(H, inn, dR)
            ctx.update_joseph()
            assert (ctx.get_status() == 0).all()
            for b in range(B):
                _, Pref[b], _ = orc.update_joseph(H[b], Pref[b], inn[b], dR[b])
            if it % 20 == 19 or it == steps - 1:
                Pn = ctx.download_P()
                for b in range(B):
                    assert np.array_equal(Pn[b], Pn[b].T)
                    w = np.linalg.eigvalsh(Pn[b])
                    assert w.min() > -1e-12 * w.max(), (it, w.min(), w.max())
                    assert rel_fro(Pn[b], Pref[b]) < TOL_P, (it, rel_fro(Pn[b], Pref[b]))


def test_gate_in_the_factorisation_prologue_is_bit_identical_to_the_gate_kernel(built):
    """For thousands of filters Estimator::MHGating (src/update.cpp:60-96) rides in the prologue of the Cholesky kernel
    (chol_reg_f64_kernel<..., GATE>: distances from the compact diagonal blocks of S, relaxation, rejected pairs decoupled
    where S is loaded); fewer than 512 filters keep gate_ell_kernel as a launch of its own (and the few-factor instantiation
    of the factorisation): masks, distances, dx and P+ of the same filters must be the same bits either way."""
    from xivo_amd.lib import FLAG_PROFILE
    for (N, F) in [(250, 80), (150, 50), (203, 30)]:
        P, H, inn, dR = synth.s_level(N, F, 8, seed=77)
        inn[:, 4:8] *= 400.0; inn[3, 10:2 * F - 6] *= 900.0      # features the gate throws out (filter 3: nearly all of them - threshold relaxation)
        res = []
        for B in (600, 500):
            idx = np.arange(B) % 8
            with Context(N, 2 * F, B, flags=FLAG_MULTI_KERNEL | FLAG_THROUGHPUT_ROUTE | FLAG_PROFILE) as ctx:
                ctx.upload_P(P[idx]); ctx.set_measurements(H[idx], inn[idx], dR[idx])
                ctx.update_dense_gated(F, 2.25, 5.991, 1.1, 5, B)
                prof = ctx.profile_get()
                mask, dist = ctx.get_gate(F, B)
                res.append((ctx.download_P()[:500], ctx.get_err()[:500], mask[:500].copy(), dist[:500].copy(), ctx.get_status(check=False)[:500]))
            assert (prof["mh_gate"]["launches"] == 0) == (B == 600) and ("+gate" in prof["chol_S"]["kernel"]) == (B == 600), prof
        for x, y in zip(res[0], res[1]):
            assert np.array_equal(x, y)
        assert 0 < res[0][2].sum() < res[0][2].size               # something was rejected, not everything
