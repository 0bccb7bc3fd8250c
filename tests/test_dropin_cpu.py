"""CPU-only: the host-side row-pair compression of the one-call plumbing entry (xivo_hip_update_joseph_host scans H_ on
the host while it stages it) against a plain restatement of the format the batched device hand-over writes
(xivo_amd/csrc/ell.h, meas_compress_kernel): common columns = columns used by more than half of the non-empty row pairs,
ascending, at most 16; the rest of a pair's columns in ascending order in its 12 private slots."""
import ctypes as C

import numpy as np
import pytest

from xivo_amd import synth

CW, PW = 16, 12
W = CW + PW


def restate(H, pairs_clear):
    """H [M, N] -> (idx [pairs_clear, 28], val [pairs_clear, 28, 2], nc, pw, over)"""
    M, N = H.shape
    pairs = (M + 1) // 2
    Hp = np.zeros((2 * pairs, N)); Hp[:M] = H
    nz = [np.nonzero((Hp[2 * p] != 0) | (Hp[2 * p + 1] != 0))[0] for p in range(pairs)]
    ne = sum(len(c) > 0 for c in nz)
    occ = np.zeros(N, dtype=int)
    for c in nz:
        occ[c] += 1
    flagged = [n for n in range(N) if ne > 0 and 2 * occ[n] > ne]
    common = flagged[:CW]
    nc = min(len(flagged), CW)
    slot = {n: t for t, n in enumerate(common)}
    idx = np.zeros((pairs_clear, W), dtype=np.int32); val = np.zeros((pairs_clear, W, 2))
    idx[:, :nc] = common
    pw = over = 0
    for p in range(pairs):
        pos = 0
        for n in nz[p][:W]:
            if n in slot:
                val[p, slot[n]] = Hp[2 * p:2 * p + 2, n]
            else:
                if pos < PW:
                    idx[p, CW + pos] = n; val[p, CW + pos] = Hp[2 * p:2 * p + 2, n]
                pos += 1
        if len(nz[p]) > W:
            pos = PW + 1
        over |= pos > PW
        pw = max(pw, pos)
    return idx, val, nc, pw, int(over)


def run(lib, H, pairs_clear):
    M, N = H.shape
    Hc = np.asfortranarray(H)
    idx = np.full((pairs_clear, W), -7, dtype=np.int32); val = np.full((pairs_clear, W, 2), np.nan)
    nc, pw = C.c_int(-1), C.c_int(-1)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    over = lib.xivo_hip_selftest_host_compress(p(Hc), M, M, N, pairs_clear, p(idx), p(val), C.byref(nc), C.byref(pw))
    return idx, val, nc.value, pw.value, over


@pytest.mark.parametrize("N,F,odd", [(250, 80, False), (203, 30, False), (150, 50, False), (64, 8, False), (100, 21, True), (44, 3, False)])
def test_host_compression_equals_the_format_restatement(built, N, F, odd):
    from xivo_amd.lib import load_library
    lib = load_library()
    _, H, _, _ = synth.s_level(N, F, 3, seed=11 + N)
    for b in range(3):
        Hb = H[b][:-1] if odd else H[b]            # an odd row count: the last pair has one row
        if b == 1:                                 # a rejected (neutralised) feature: an all-zero pair, and a negative zero
            Hb = Hb.copy(); Hb[4:6] = 0.0; Hb[4, 3] = -0.0
        pairs_clear = (Hb.shape[0] + 15) // 16 * 8 + 8   # cleared beyond the rows in use, as the context's M_max asks
        got, want = run(lib, Hb, pairs_clear), restate(Hb, pairs_clear)
        assert got[2:] == want[2:], (got[2:], want[2:])
        assert got[4] == 0
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        # the compressed rows ARE the matrix: rebuilt dense == H
        D = np.zeros_like(Hb)
        for p_ in range((Hb.shape[0] + 1) // 2):
            for t in range(W):
                v = got[1][p_, t]
                if v[0] != 0 or v[1] != 0:
                    D[2 * p_, got[0][p_, t]] = v[0]
                    if 2 * p_ + 1 < Hb.shape[0]:
                        D[2 * p_ + 1, got[0][p_, t]] = v[1]
        assert np.array_equal(D, Hb + 0.0)


def test_rows_that_do_not_fit_are_reported(built):
    from xivo_amd.lib import load_library
    lib = load_library()
    rng = np.random.default_rng(5)
    H = rng.uniform(-1, 1, (20, 120))              # dense rows: 120 non-zero columns per pair
    got, want = run(lib, H, 16), restate(H, 16)
    assert got[4] == 1 and want[4] == 1 and got[2:] == want[2:]
    # 13 private columns in one pair only
    _, Hs, _, _ = synth.s_level(150, 10, 1, seed=3)
    Hs = Hs[0].copy(); Hs[0, 100:113] = 1.0
    got, want = run(lib, Hs, 16), restate(Hs, 16)
    assert got[4] == want[4] and got[2:] == want[2:]
    assert lib.xivo_hip_selftest_host_compress(None, 1, 1, 1, 1, None, None, None, None) == -1


def test_fused_update_tile_tables_cover_every_block_pair_once(built):
    """The orientation tables of the one-kernel update's product phase (csrc/fused_update.hip, kFusedTiles10 / 13): every
    unordered pair of column blocks is formed by exactly one wave, counts as the kernel's dispatch assumes, SIMD loads balanced
    (14 14 14 13 of 55 tiles; 23 23 23 22 of 91). Host code only."""
    import ctypes as C
    from xivo_amd.lib import load_library
    lib = load_library()
    for nwl, want in ((10, [14, 13, 14, 14]), (13, [23, 23, 23, 22])):
        simd = (C.c_int * 4)()
        assert lib.xivo_hip_selftest_fused_tiles(nwl, simd) == 0
        assert sorted(simd) == sorted(want) and sum(simd) == nwl * (nwl + 1) // 2
    assert lib.xivo_hip_selftest_fused_tiles(12, None) == -1
