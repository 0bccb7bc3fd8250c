"""GPU parity of Estimator::UpdateJosephForm (src/estimator.cpp:1257-1288)
through the C ABI vs the numpy oracle, on the BASELINE.json synthetic sizes."""
import numpy as np
import pytest

import xivo_oracle as orc
from helpers import rel_fro, TOL_P, TOL_DX
from xivo_amd import synth
from xivo_amd.lib import Context, FLAG_FULL_PNEW

pytestmark = pytest.mark.gpu

CASES = [  # (N, F) -> M = 2F ; BASELINE.json configs + TUM-VI default build + ragged sizes
    (150, 50), (250, 80), (400, 150), (203, 30), (251, 130), (37, 3), (64, 8), (100, 1),
]


@pytest.mark.parametrize("N,F", CASES)
@pytest.mark.parametrize("dense", [False, True])
def test_update_joseph_matches_oracle(built, N, F, dense):
    B = 5
    P, H, inn, dR = synth.s_level(N, F, B, seed=N * 7 + F, dense=dense)
    with Context(N, 2 * F, B) as ctx:
        ctx.upload_P(P)
        ctx.set_measurements(H, inn, dR)
        ctx.update_joseph()
        err = ctx.get_err()
        Pn = ctx.download_P()
        assert (ctx.get_status() == 0).all()
    for b in range(B):
        e_ref, P_ref, _ = orc.update_joseph(H[b], P[b], inn[b], dR[b])
        assert rel_fro(Pn[b], P_ref) < TOL_P
        assert rel_fro(err[b], e_ref) < TOL_DX
        assert np.array_equal(Pn[b], Pn[b].T)  # lower triangle mirrored


def test_full_pnew_flag_equals_mirrored(built):
    N, F, B = 150, 50, 3
    P, H, inn, dR = synth.s_level(N, F, B, seed=3)
    outs = []
    for flags in (0, FLAG_FULL_PNEW):
        with Context(N, 2 * F, B, flags=flags) as ctx:
            ctx.upload_P(P); ctx.set_measurements(H, inn, dR); ctx.update_joseph()
            outs.append(ctx.download_P())
    assert rel_fro(outs[0], outs[1]) < 1e-13


def test_upload_download_roundtrip_bit_exact(built):
    N, B = 203, 4
    rng = np.random.default_rng(0)
    P = rng.normal(size=(B, N, N))
    with Context(N, 60, B) as ctx:
        ctx.upload_P(P)
        assert np.array_equal(ctx.download_P(), P)
        ctx.upload_P(P[2:3] * 2, b0=1)
        got = ctx.download_P()
        assert np.array_equal(got[1], P[2] * 2) and np.array_equal(got[0], P[0])


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
    with Context(N, 2 * F, B) as ctx:
        ctx.upload_P(P); ctx.set_measurements(H, inn, dR); ctx.update_joseph()
        st = ctx.get_status(check=False)
        assert st[0] == 0 and st[1] != 0
        with pytest.raises(XivoHipError):
            ctx.get_status()
