"""The literal drop-in call: Estimator::UpdateJosephForm() on members that live in pageable host memory
(/root/reference/src/estimator.cpp:1257-1288, callers src/update.cpp:141 and :332) through the one-call entry
xivo_hip_update_joseph_host - against the oracle, against the six-call sequence it replaces (bit for bit), in place on the
caller's P_ call after call with host edits in between, with the residency modes, for an H_ that does not fit the compressed rows, and through the C++ adapter's
timing harness (what bench.py's `dropin` block runs)."""
import ctypes as C
import os

import numpy as np
import pytest

import xivo_oracle as orc
from helpers import rel_fro, TOL_P, TOL_DX
from xivo_amd import synth
from xivo_amd.lib import Context, HOST_P_RESIDENT, HOST_KEEP_P, FLAG_NO_LDLT_FALLBACK, FLAG_DENSE_H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def six_calls(ctx, P, H, inn, dR):
    ctx.upload_P(P[None])
    ctx.set_measurements(H[None], inn[None], dR[None])
    ctx.update_joseph(1)
    assert (ctx.get_status(0, 1) == 0).all()
    return ctx.get_err(0, 1)[0], ctx.download_P(0, 1)[0]


@pytest.mark.parametrize("N,F", [(250, 80), (203, 30), (150, 50), (64, 8), (251, 60), (100, 21)])
def test_one_call_equals_oracle_and_the_six_call_sequence(built, N, F):
    P, H, inn, dR = synth.s_level(N, F, 2, seed=77 + N)
    M = 2 * F
    with Context(N, M, 1) as ctx:
        for b in range(2):
            e6, P6 = six_calls(ctx, P[b], H[b], inn[b], dR[b])
            Pio = np.asfortranarray(P[b].copy())
            err, rc = ctx.update_joseph_host(H[b], inn[b], dR[b], Pio)
            assert rc == 0 and ctx.last_path() == 1
            e_ref, P_ref, _ = orc.update_joseph(H[b], P[b], inn[b], dR[b])
            assert rel_fro(Pio, P_ref) < TOL_P and rel_fro(err, e_ref) < TOL_DX
            # same compressed rows, same kernels: the same bits as the general entry points
            assert np.array_equal(err, e6) and np.array_equal(np.ascontiguousarray(Pio), P6)


def test_in_place_updates_in_a_row_see_the_hosts_writes(built):
    """One P_ buffer updated call after call and edited by the host in between (RemoveFeatureFromState-style zeroing,
    src/estimator.cpp:778-783): every call must start from what the host holds - no stale copy anywhere."""
    N, F = 203, 30
    P, H, inn, dR = synth.s_level(N, F, 4, seed=5)
    with Context(N, 2 * F, 1) as ctx:
        Pio = np.asfortranarray(P[0].copy())
        Pc = P[0].copy()
        for k in range(4):
            err, _ = ctx.update_joseph_host(H[k], inn[k], dR[k], Pio)
            e_ref, Pc, _ = orc.update_joseph(H[k], Pc, inn[k], dR[k])
            assert rel_fro(Pio, Pc) < TOL_P and rel_fro(err, e_ref) < TOL_DX
            off = 23 + 6 * 15 + 3 * k
            Pio[off:off + 3, :] = 0.0; Pio[:, off:off + 3] = 0.0      # host edit of the authoritative P_
            Pc[off:off + 3, :] = 0.0; Pc[:, off:off + 3] = 0.0


def test_residency_modes(built):
    N, F = 150, 50
    P, H, inn, dR = synth.s_level(N, F, 2, seed=9)
    with Context(N, 2 * F, 1) as ctx:
        e_ref, P1, _ = orc.update_joseph(H[0], P[0], inn[0], dR[0])
        e_ref2, P2, _ = orc.update_joseph(H[1], P1, inn[1], dR[1])
        Pio = np.asfortranarray(P[0].copy())
        # first update: P+ stays on the device, the host array is not written
        err, _ = ctx.update_joseph_host(H[0], inn[0], dR[0], Pio, mode=HOST_KEEP_P)
        assert rel_fro(err, e_ref) < TOL_DX and np.array_equal(Pio, P[0])
        # second update: nothing uploaded (garbage in the host array must not matter), P+ comes back
        Pio[:] = np.nan
        err, _ = ctx.update_joseph_host(H[1], inn[1], dR[1], Pio, mode=HOST_P_RESIDENT)
        assert rel_fro(err, e_ref2) < TOL_DX and rel_fro(Pio, P2) < TOL_P
        # both: no P pointer at all
        ctx.upload_P(P[0][None])
        err, _ = ctx.update_joseph_host(H[0], inn[0], dR[0], None, mode=HOST_P_RESIDENT | HOST_KEEP_P)
        assert rel_fro(err, e_ref) < TOL_DX and rel_fro(ctx.download_P(0, 1)[0], P1) < TOL_P


def test_filter_b_of_a_larger_context_and_changing_row_counts(built):
    N = 150
    P, H, inn, dR = synth.s_level(N, 50, 3, seed=21)
    with Context(N, 100, 3) as ctx:
        ctx.upload_P(P)
        for b, F in ((2, 50), (0, 20), (1, 35)):      # fewer rows than the call before: stale rows must be gone
            Pio = np.asfortranarray(P[b].copy())
            err, _ = ctx.update_joseph_host(H[b][:2 * F], inn[b][:2 * F], dR[b][:2 * F], Pio, b=b)
            e_ref, P_ref, _ = orc.update_joseph(H[b][:2 * F], P[b], inn[b][:2 * F], dR[b][:2 * F])
            assert rel_fro(Pio, P_ref) < TOL_P and rel_fro(err, e_ref) < TOL_DX


@pytest.mark.parametrize("flags", [0, FLAG_DENSE_H])
def test_an_H_without_the_row_structure_takes_the_general_route(built, flags):
    N, M = 120, 40
    rng = np.random.default_rng(3)
    A = rng.uniform(-1, 1, (N, N)); P = A @ A.T / N + 1e-3 * np.eye(N)
    H = rng.uniform(-1, 1, (M, N)); inn = rng.normal(0, 1.5, M); dR = np.full(M, 2.25)
    with Context(N, M, 1, flags=flags) as ctx:
        Pio = np.asfortranarray(P.copy())
        err, _ = ctx.update_joseph_host(H, inn, dR, Pio)
        assert ctx.last_path() == 0
        e_ref, P_ref, _ = orc.update_joseph(H, P, inn, dR)
        assert rel_fro(Pio, P_ref) < TOL_P and rel_fro(err, e_ref) < TOL_DX


def test_not_spd_is_reported_when_the_fallback_is_off(built):
    N, F = 100, 20
    P, H, inn, dR = synth.s_level(N, F, 1, seed=2)
    Pbad = -P[0]
    with Context(N, 2 * F, 1, flags=FLAG_NO_LDLT_FALLBACK) as ctx:
        Pio = np.asfortranarray(Pbad.copy())
        err, rc = ctx.update_joseph_host(H[0], inn[0], dR[0], Pio, check=False)
        assert rc == -3 and np.array_equal(Pio, Pbad)
    with Context(N, 2 * F, 1) as ctx:          # default: updated the reference's way (pivoted L D L^T), status OK
        Pio = np.asfortranarray(Pbad.copy())
        err, rc = ctx.update_joseph_host(H[0], inn[0], dR[0], Pio, check=False)
        e_ref, P_ref, _ = orc.update_joseph(H[0], Pbad, inn[0], dR[0])
        assert rc == 0 and rel_fro(Pio, P_ref) < TOL_P and rel_fro(err, e_ref) < 1e-7


def test_bad_arguments(built):
    with Context(64, 16, 1) as ctx:
        lib = ctx.lib
        a = np.zeros(64 * 64)
        assert lib.xivo_hip_update_joseph_host(ctx.h, 0, 16, None, 16, None, None, None, 64, None, 0) == -1
        assert lib.xivo_hip_update_joseph_host(ctx.h, 1, 16, a.ctypes.data, 16, a.ctypes.data, a.ctypes.data, a.ctypes.data, 64, a.ctypes.data, 0) == -1
        assert lib.xivo_hip_update_joseph_host(ctx.h, 0, 32, a.ctypes.data, 32, a.ctypes.data, a.ctypes.data, a.ctypes.data, 64, a.ctypes.data, 0) == -1


@pytest.mark.parametrize("N,F", [(250, 80), (203, 30)])
def test_adapter_timing_harness(built, N, F):
    """xivo::hip::Estimator::UpdateJosephForm() through libxivo_host.so, members in pageable memory: the three plumbing
    modes give the oracle's result; the per-call wall times are what bench.py reports as `dropin`."""
    lib = C.CDLL(os.path.join(ROOT, "xivo_amd", "libxivo_host.so"))
    P, H, inn, dR = synth.s_level(N, F, 1, seed=31)
    M = 2 * F
    Pf, Hf = np.asfortranarray(P[0]), np.asfortranarray(H[0])
    e_ref, P_ref, _ = orc.update_joseph(H[0], P[0], inn[0], dR[0])
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    res = {}
    for mode in (0, 1, 2):
        n = 60
        ms = np.zeros(n); Pout = np.zeros((N, N), order="F"); err = np.zeros(N); msg = C.create_string_buffer(256)
        rc = lib.xivo_host_time_update_joseph(N, M, p(Pf), p(Hf), p(inn[0].copy()), p(dR[0].copy()), n, mode, C.c_uint(0), p(ms),
                                              p(Pout), p(err), msg, 256)
        assert rc == 0, msg.value
        assert rel_fro(Pout, P_ref) < TOL_P and rel_fro(err, e_ref) < TOL_DX
        res[mode] = (np.median(ms[10:]), Pout.copy(), err.copy())
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])   # one call == six calls, bit for bit
    print(f"\nUpdateJosephForm() wall, N={N} M={M}: one call {res[0][0]:.3f} ms, six calls {res[1][0]:.3f} ms, "
          f"one call + resident prior {res[2][0]:.3f} ms")
