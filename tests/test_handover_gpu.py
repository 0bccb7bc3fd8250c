"""Hand-over of the stacked measurements (Estimator::H_, inn_, diagR_ of src/update.cpp:129-138) to the device path:
the batched dense -> row-pair compression (one launch), the device-pointer entry the bench times per step, the
dense copies rebuilt on demand, odd row counts, filters that do not fit the compressed form."""
import numpy as np
import pytest

import xivo_oracle as orc
from helpers import rel_fro, TOL_P, TOL_DX
from xivo_amd import synth
from xivo_amd.lib import Context

pytestmark = pytest.mark.gpu


def _dev(ctx, arrs):
    return [ctx.device_array(a) for a in arrs]


@pytest.mark.parametrize("N,F", [(250, 80), (150, 50), (64, 8), (203, 30)])
def test_device_entry_equals_host_entry(built, N, F):
    B, M = 6, 2 * F
    P, H, inn, dR = synth.s_level(N, F, B, seed=5 * N + F)
    Hc = np.ascontiguousarray(np.transpose(H, (0, 2, 1)))       # column-major M x N per filter
    outs = []
    for dev in (False, True):
        with Context(N, M, B) as ctx:
            ctx.upload_P(P)
            if dev:
                dH, dinn, dRd = _dev(ctx, [Hc, inn, dR])
                ctx.set_measurements_device(dH, dinn, dRd, M, B)
            else:
                ctx.set_measurements(H, inn, dR)
            ctx.update_joseph()
            assert ctx.last_path() == 1
            outs.append((ctx.download_P(), ctx.get_err()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    for b in range(B):
        e_ref, P_ref, _ = orc.update_joseph(H[b], P[b], inn[b], dR[b])
        assert rel_fro(outs[1][0][b], P_ref) < TOL_P and rel_fro(outs[1][1][b], e_ref) < TOL_DX


def test_device_entry_strided_and_repeated(built):
    """leading dimension > M, filters further apart than one matrix, hand-over repeated with new rows (stale rows of
    a larger M must vanish)."""
    N, B = 100, 4
    with Context(N, 40, B) as ctx:
        for F in (20, 7):
            M, ldh = 2 * F, 2 * F + 6
            P, H, inn, dR = synth.s_level(N, F, B, seed=F)
            buf = np.full((B, N + 3, ldh), 7.0)
            buf[:, :N, :M] = np.transpose(H, (0, 2, 1))
            dH, dinn, dRd = _dev(ctx, [buf, inn, dR])
            ctx.upload_P(P)
            ctx.set_measurements_device(dH, dinn, dRd, M, B, strideH=(N + 3) * ldh, ldh=ldh)
            ctx.update_joseph()
            Pn, err = ctx.download_P(), ctx.get_err()
            for b in range(B):
                e_ref, P_ref, _ = orc.update_joseph(H[b], P[b], inn[b], dR[b])
                assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX
            Hg, ig, rg = ctx.get_H(1)
            assert Hg.shape[0] == M and np.array_equal(Hg, H[1]) and np.array_equal(ig, inn[1]) and np.array_equal(rg, dR[1])


@pytest.mark.parametrize("M", [7, 13, 1])
def test_odd_row_count(built, M):
    N, B = 48, 3
    rng = np.random.default_rng(M)
    A = rng.uniform(-1, 1, (B, N, N))
    P = A @ np.transpose(A, (0, 2, 1)) / N + 1e-3 * np.eye(N)
    H = np.zeros((B, M, N))
    for b in range(B):
        for m in range(M):
            cols = rng.choice(N, 9, replace=False)
            H[b, m, cols] = rng.normal(0, 10, 9)
    inn = rng.normal(0, 1.5, (B, M)); dR = np.full((B, M), 2.25)
    with Context(N, 16, B) as ctx:
        ctx.upload_P(P)
        ctx.set_measurements(H, inn, dR)
        ctx.update_joseph()
        Pn, err = ctx.download_P(), ctx.get_err()
        Hg, _, _ = ctx.get_H(2)
        assert Hg.shape[0] == M and np.array_equal(Hg, H[2])
    for b in range(B):
        e_ref, P_ref, _ = orc.update_joseph(H[b], P[b], inn[b], dR[b])
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX


def test_mixed_batch_one_filter_does_not_fit(built):
    """one dense H among structured ones: the whole call takes the dense pipeline; the structured filters' dense rows
    are rebuilt from their compressed form, the dense filter's were written at hand-over."""
    N, F, B = 120, 20, 5
    P, H, inn, dR = synth.s_level(N, F, B, seed=77)
    _, Hd, _, _ = synth.s_level(N, F, 1, seed=78, dense=True)
    H[3] = Hd[0]
    with Context(N, 2 * F, B) as ctx:
        ctx.upload_P(P)
        ctx.set_measurements(H[:2], inn[:2], dR[:2], b0=0)        # chunked hand-over like bench.py's upload loop
        ctx.set_measurements(H[2:], inn[2:], dR[2:], b0=2)
        ctx.update_joseph()
        assert ctx.last_path() == 0
        Pn, err = ctx.download_P(), ctx.get_err()
        for b in (0, 3, 4):
            Hg, _, _ = ctx.get_H(b)
            assert np.array_equal(Hg, H[b])
    for b in range(B):
        e_ref, P_ref, _ = orc.update_joseph(H[b], P[b], inn[b], dR[b])
        assert rel_fro(Pn[b], P_ref) < TOL_P and rel_fro(err[b], e_ref) < TOL_DX


def test_gated_update_after_device_handover_is_repeatable(built):
    """the bench's step: hand-over + gating + update, twice from the same P: gating neutralises the compressed rows in
    place, the next hand-over must restore them."""
    N, F, B = 250, 80, 4
    M = 2 * F
    P, H, inn, dR = synth.s_level(N, F, B, seed=9)
    H *= 0.01                     # S_f ~ R, so the wild innovations below are rejected
    inn[:, :6] += 25.0
    Hc = np.ascontiguousarray(np.transpose(H, (0, 2, 1)))
    res = []
    with Context(N, M, B) as ctx:
        dH, dinn, dRd = _dev(ctx, [Hc, inn, dR])
        for _ in range(2):
            ctx.upload_P(P)
            ctx.set_measurements_device(dH, dinn, dRd, M, B)
            ctx.update_dense_gated(F, 2.25, 5.991, 1.1, 5)
            mask, _ = ctx.get_gate(F)
            res.append((ctx.download_P(), ctx.get_err(), mask))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][2], res[1][2])
    for b in range(B):
        d = orc.mh_distances(H[b].reshape(F, 2, -1), P[b], inn[b].reshape(F, 2), 2.25)
        assert np.array_equal(res[0][2][b], orc.mh_gate(d, 5.991, 1.1, 5)[0]) and not res[0][2][b].all()
        keep = np.repeat(res[0][2][b], 2)
        e_ref, P_ref, _ = orc.update_joseph(H[b][keep], P[b], inn[b][keep], dR[b][keep])
        assert rel_fro(res[0][0][b], P_ref) < TOL_P and rel_fro(res[0][1][b], e_ref) < TOL_DX


_NOCOMPRESS_SNIPPET = r"""
import sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/oracle"); sys.path.insert(0, {root!r} + "/tests")
import numpy as np
import xivo_oracle as orc
from helpers import rel_fro
from xivo_amd import synth
from xivo_amd.lib import Context
N, F, B = 150, 50, 3
P, H, inn, dR = synth.s_level(N, F, B, seed=9)
with Context(N, 2 * F + 20, B) as ctx:      # allocated rows beyond M: the neutral padding is exercised too
    ctx.upload_P(P); ctx.set_measurements(H, inn, dR); ctx.update_joseph()
    path = ctx.last_path(); Pn = ctx.download_P(); err = ctx.get_err(); Hb, innb, dRb = ctx.get_H(1)
w = 0.0
for b in range(B):
    e_ref, P_ref, _ = orc.update_joseph(H[b], P[b], inn[b], dR[b])
    w = max(w, rel_fro(Pn[b], P_ref), rel_fro(err[b], e_ref))
print(json.dumps(dict(path=path, worst=w, H_equal=bool(np.array_equal(Hb, H[1])), inn_equal=bool(np.array_equal(innb, inn[1])))))
"""


def test_handover_without_compression_takes_the_dense_pipeline(built):
    """When the LDS lists of the compression kernel cannot hold a shape (very wide states) the hand-over keeps every
    filter's dense rows and the update takes the dense pipeline. The shape limit is far away (N > ~2800 at M = 384), so
    the same path is forced here with XIVO_HIP_NO_COMPRESS."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env["XIVO_HIP_NO_COMPRESS"] = "1"
    r = subprocess.run([sys.executable, "-c", _NOCOMPRESS_SNIPPET.format(root=root)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["path"] == 0 and d["worst"] < 1e-9 and d["H_equal"] and d["inn_equal"], d
