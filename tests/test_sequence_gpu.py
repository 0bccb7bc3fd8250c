"""Resident-state edits, ragged batches and whole sequences through the C ABI vs the oracle."""
import numpy as np
import pytest

import xivo_oracle as orc
from seq_oracle import OracleBackend
from xivo_amd import formats, pcw, sequence
from xivo_amd import lib as L

pytestmark = pytest.mark.gpu


def rel_fro(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _start(cfg, B, seed):
    rng = np.random.default_rng(seed)
    sims = [pcw.TrajectorySim("lissajous", seed=seed + b) for b in range(B)]
    poses = sequence.initial_poses(cfg, sims, t0=0.3 + 0.1 * seed)
    P0 = []
    for b in range(B):
        A = rng.uniform(-1, 1, size=(cfg.N, cfg.N))
        P = A @ A.T / cfg.N * 1e-3 + 1e-5 * np.eye(cfg.N)
        P0.append(0.5 * (P + P.T))
    return poses, np.array(P0), rng


def _random_ops(cfg, B, rng, n):
    ops = []
    for _ in range(n):
        b = int(rng.integers(B)); k = int(rng.integers(8))
        g = int(rng.integers(cfg.n_groups)); j = int(rng.integers(cfg.n_features))
        if k == L.EDIT_P_ZERO_RC:
            off = int(rng.integers(cfg.N - 6)); ops.append(sequence._op(b, k, off, int(rng.integers(1, 7))))
        elif k == L.EDIT_P_COPY_RC:
            ops.append(sequence._op(b, k, int(rng.integers(23, cfg.N - 6)), int(rng.integers(0, 17)), int(rng.integers(1, 7))))
        elif k == L.EDIT_P_SET_BLOCK3:
            A = rng.normal(size=(3, 3)); ops.append(sequence._op(b, k, int(rng.integers(cfg.N - 3)), v=(A @ A.T).reshape(-1)))
        elif k in (L.EDIT_ADD_GROUP, L.EDIT_REMOVE_GROUP):
            ops.append(sequence._op(b, k, g))
        elif k == L.EDIT_ADD_FEATURE:
            A = rng.normal(size=(3, 3)) * 0.1
            v = np.concatenate([rng.normal(size=3) * 0.3, rng.uniform(0, 400, 2), (A @ A.T).reshape(-1)])
            ops.append(sequence._op(b, k, j, int(rng.integers(cfg.n_features)), g, v=v))
        else:
            ops.append(sequence._op(b, k, j, v=rng.uniform(0, 400, 2)))
    return np.array(ops, dtype=L.edit_dtype)


def test_edit_batch_matches_the_reference_edits_bit_for_bit(built):
    """Random mixes of every XIVO_EDIT_* kind over several filters in one call == the host edits of
    src/estimator.cpp:745-846 / src/feature.cpp:753-760 applied one by one (copies and zero fills: exact)."""
    cfg = sequence.SequenceConfig(n_groups=6, n_features=12)
    B = 5
    poses, P0, rng = _start(cfg, B, 1)
    hb = sequence.HipBackend(cfg, B, poses, P0)
    ob = OracleBackend(cfg, B, poses, P0)
    try:
        for rnd in range(3):
            ops = _random_ops(cfg, B, rng, 60)
            ops = ops[np.argsort(ops["b"], kind="stable")]
            hb.edit(ops); ob.edit(ops)
            Pd = hb.covariance()
            pose_d, group_d, feat_d = hb.scene()
            for b in range(B):
                s = ob.st[b]
                assert np.array_equal(Pd[b], s["P"]), (rnd, b)
                assert np.array_equal(feat_d[b]["sind"], s["sind"])
                on = s["sind"] >= 0
                assert np.array_equal(feat_d[b]["x"][on], s["x"][on]) and np.array_equal(feat_d[b]["ref_sind"][on], s["ref"][on])
                assert np.array_equal(feat_d[b]["xp"][on], s["xp"][on])
                assert np.array_equal(group_d[b]["Rsb"].reshape(-1, 3, 3).transpose(0, 2, 1), s["gR"])
                assert np.array_equal(group_d[b]["Tsb"], s["gT"])
    finally:
        hb.close()


def test_set_pixels_dense_frame_upload(built):
    """xivo_hip_set_pixels: one [B x F x 2] array per frame, NaN pairs leave an entry as it is."""
    cfg = sequence.SequenceConfig(n_groups=4, n_features=8)
    B = 3
    poses, P0, rng = _start(cfg, B, 4)
    hb = sequence.HipBackend(cfg, B, poses, P0)
    try:
        ops = [sequence._op(b, L.EDIT_ADD_FEATURE, j, j, 0, v=np.concatenate([[0.1, 0.2, 0.3], [100.0 + j, 50.0 + b], np.eye(3).reshape(-1)]))
               for b in range(B) for j in range(0, 8, 2)]
        hb.edit(np.array(ops, dtype=L.edit_dtype))
        xp = np.full((B, 8, 2), np.nan)
        xp[:, 0] = [[11.0, 12.0], [21.0, 22.0], [31.0, 32.0]]
        xp[1, 4] = [5.5, np.nan]                 # half a pair is no pair
        xp[2, 6] = [7.0, 8.0]
        xp[0, 1] = [1.0, 1.0]                    # absent entry: the pixel is stored but the entry stays absent
        hb.set_pixels(xp)
        _, _, f = hb.scene()
        assert np.array_equal(f["xp"][:, 0], xp[:, 0]) and np.array_equal(f["xp"][2, 6], [7.0, 8.0])
        assert np.array_equal(f["xp"][1, 4], [104.0, 51.0]) and np.array_equal(f["xp"][0, 2], [102.0, 50.0])
        assert f["sind"][0, 1] == -1 and np.array_equal(f["sind"][:, 0], [0, 0, 0])
        with pytest.raises(L.XivoHipError):
            hb.ctx.set_pixels(np.zeros((B + 1, 8, 2)))
    finally:
        hb.close()


def test_edit_batch_rejects_bad_ops(built):
    cfg = sequence.SequenceConfig(n_groups=4, n_features=8)
    poses, P0, _ = _start(cfg, 2, 2)
    hb = sequence.HipBackend(cfg, 2, poses, P0)
    try:
        bad = [sequence._op(2, L.EDIT_REMOVE_GROUP, 0), sequence._op(0, L.EDIT_ADD_GROUP, 4),
               sequence._op(0, L.EDIT_ADD_FEATURE, 8, 0, 0), sequence._op(0, L.EDIT_ADD_FEATURE, 0, 8, 0),
               sequence._op(0, L.EDIT_P_ZERO_RC, cfg.N - 2, 3), sequence._op(0, 99, 0),
               sequence._op(0, L.EDIT_P_COPY_RC, 0, cfg.N - 1, 2)]
        for o in bad:
            with pytest.raises(L.XivoHipError):
                hb.ctx.edit_batch(cfg.n_features, np.array([o], dtype=L.edit_dtype))
        # ops must be grouped by filter: the raw entry point refuses a descending filter index
        two = np.array([sequence._op(1, L.EDIT_REMOVE_GROUP, 0), sequence._op(0, L.EDIT_REMOVE_GROUP, 0)], dtype=L.edit_dtype)
        rc = hb.ctx.lib.xivo_hip_edit_batch(hb.ctx.h, cfg.n_features, 2, two.ctypes.data)
        assert rc == -1
        hb.ctx.edit_batch(cfg.n_features, np.zeros(0, dtype=L.edit_dtype))      # empty list is fine
    finally:
        hb.close()


def test_ragged_batch_filter_update(built):
    """Filters of one call hold different numbers of features (absent entries, sind = -1): 0, fewer than
    min_inliers (no gating, src/manager.cpp:635), and a full list with gross outliers that gating must reject."""
    cfg = sequence.SequenceConfig(n_groups=5, n_features=14, fix_group_block=False)
    B = 4
    present = [0, 3, 9, 14]
    poses, P0, rng = _start(cfg, B, 3)
    hb = sequence.HipBackend(cfg, B, poses, P0)
    ob = OracleBackend(cfg, B, poses, P0)
    try:
        ops = []
        fx, cx, cy = cfg.cam["fx"], cfg.cam["cx"], cfg.cam["cy"]
        for b in range(B):
            for g in range(2):
                ops.append(sequence._op(b, L.EDIT_ADD_GROUP, g))
            slots = rng.permutation(cfg.n_features)[:present[b]]
            for q, j in enumerate(slots):
                xp = rng.uniform([80, 60], [560, 420])
                x = [(xp[0] - cx) / fx, (xp[1] - cy) / fx, np.log(rng.uniform(1.0, 6.0))]
                noise = rng.normal(size=2) * (400.0 if q % 5 == 4 else 1.0)      # every 5th: gross outlier
                A = rng.normal(size=(3, 3)) * 0.01
                ops.append(sequence._op(b, L.EDIT_ADD_FEATURE, int(j), int(j), q % 2,
                                        v=np.concatenate([x, xp + noise, (A @ A.T + 1e-5 * np.eye(3)).reshape(-1)])))
        ops = np.array(ops, dtype=L.edit_dtype)
        hb.edit(ops); ob.edit(ops)
        md = hb.update(); mo = ob.update()
        assert np.array_equal(md, mo)
        assert md[0].sum() == 0 and md[1].sum() == 3 and md[3].sum() < 14
        Pd = hb.covariance(); Rd, Td = hb.poses(); Ro, To = ob.poses()
        _, _, feat_d = hb.scene()
        assert np.array_equal(Pd[0], ob.st[0]["P"])              # nothing to update: P untouched, bit for bit
        for b in range(B):
            assert rel_fro(Pd[b], ob.st[b]["P"]) < 1e-10
            assert np.abs(Td[b] - To[b]).max() < 1e-10 and np.abs(Rd[b] - Ro[b]).max() < 1e-10
            on = ob.st[b]["sind"] >= 0
            assert np.abs(feat_d[b]["x"][on] - ob.st[b]["x"][on]).max(initial=0) < 1e-10
    finally:
        hb.close()


@pytest.mark.parametrize("fix", [True, False])
def test_sequences_device_vs_oracle(built, fix):
    """Three point-cloud-world sequences for 14 camera frames (224 IMU samples): the device-resident run and the oracle
    run take identical life-cycle decisions (same inlier masks every frame) and end in the same state / covariance."""
    B = 3
    cfg = sequence.SequenceConfig(fix_group_block=fix)
    runs = {}
    for name, factory in (("hip", sequence.HipBackend), ("oracle", OracleBackend)):
        worlds = [pcw.RandomPCW(seed=10 + b) for b in range(B)]
        sims = [pcw.TrajectorySim("trefoil" if b == 1 else "lissajous", seed=200 + b) for b in range(B)]
        runs[name] = sequence.run_pcw(factory, cfg, worlds, sims, total_time=0.56)
    h, o = runs["hip"], runs["oracle"]
    try:
        for bh, bo in zip(h["runner"].books, o["runner"].books):
            assert bh.feat_id == bo.feat_id and bh.feat_ref == bo.feat_ref and bh.group_refs == bo.group_refs
        assert h["runner"].n_rejected == o["runner"].n_rejected
        assert np.abs(h["Tsb"] - o["Tsb"]).max() < 1e-8 and np.abs(h["Wsb"] - o["Wsb"]).max() < 1e-8
        Pd = h["backend"].covariance()
        for b in range(B):
            assert rel_fro(Pd[b], o["backend"].st[b]["P"]) < 1e-6
        assert min(bk.n_instate() for bk in h["runner"].books) > 10
    finally:
        h["backend"].close()


def test_many_sequences_track_ground_truth(built, tmp_path):
    """48 sequences (different worlds, curves, noise seeds) for 2.4 s on one context: every estimate stays finite and
    close to ground truth; the trajectory dump round-trips (src/app/vio.cpp:101-106 format)."""
    B = 48
    cfg = sequence.SequenceConfig()
    worlds = [pcw.RandomPCW(seed=b) for b in range(B)]
    sims = [pcw.TrajectorySim("lissajous" if b % 2 == 0 else "trefoil", rate=0.08 + 0.001 * b, seed=300 + b) for b in range(B)]
    out = sequence.run_pcw(sequence.HipBackend, cfg, worlds, sims, total_time=2.4)
    try:
        assert np.isfinite(out["Tsb"]).all() and np.isfinite(out["Wsb"]).all()
        ate = np.array([formats.ate_rmse(out["Tsb"][:, b], out["gt_Tsb"][:, b], align=False) for b in range(B)])
        print('ATE', np.round(np.sort(ate), 3))
        assert np.median(ate) < 0.08 and ate.max() < 0.5, ate
        P = out["backend"].covariance()
        assert np.isfinite(P).all() and all(np.linalg.eigvalsh(0.5 * (P[b] + P[b].T)).min() > -1e-9 for b in range(0, B, 8))
        formats.write_trajectory(str(tmp_path / "t.txt"), out["ts"], out["Tsb"][:, 0], out["Wsb"][:, 0])
        ts, T, W = formats.read_trajectory(str(tmp_path / "t.txt"))
        assert np.array_equal(ts, out["ts"]) and np.allclose(T, out["Tsb"][:, 0], rtol=1e-8, atol=1e-12)
    finally:
        out["backend"].close()


def test_pyxivo_style_estimator_runs_the_reference_client_loop(built):
    """The loop of scripts/pyxivo_pcw.py:133-163 against xivo_amd.pyxivo.Estimator (one filter): IMU packets first at
    equal stamps, VisualMeasPointCloud with (ids, x, y, depth), accessors in pyxivo's shapes. The batched driver
    (run_pcw, B = 1) issues the same device calls, so both end in the same pose."""
    import os
    from xivo_amd import pyxivo
    here = os.path.dirname(os.path.abspath(__file__))
    cfg = pyxivo.config_from_cfg(pyxivo.load_json_with_comments(os.path.join(here, "golden", "pcw_like_cfg.json")))
    imu = pcw.TrajectorySim("lissajous", seed=41)
    cfg.X0["Vsb"] = imu.vel(0.0)
    vision = pcw.RandomPCW(seed=5)
    K = np.array([[275.0, 0, 320.0], [0, 275.0, 240.0], [0, 0, 1.0]])
    Rbc = pcw.so3_exp(cfg.Wbc)
    est = pyxivo.Estimator(cfg, "", "lissajous", False)
    est.InitWithSimDepths()
    with pytest.raises(NotImplementedError):
        est.VisualMeas(0, "img.png")
    total, imu_dt, vis_dt = 0.8, 0.0025, 0.04
    packets = [(k * imu_dt, 0) for k in range(int(round(total / imu_dt)))] + [(k * vis_dt, 1) for k in range(int(round(total / vis_dt)))]
    packets.sort(key=lambda p: (round(p[0] * 1e9), p[1]))
    try:
        for t, kind in packets:
            ts = int(round(t * 1e9))
            if kind == 0:
                accel, gyro = imu.meas(t)
                est.InertialMeas(ts, gyro[0], gyro[1], gyro[2], accel[0], accel[1], accel[2])
            else:
                Rsb, Tsb = imu.gsb(t)
                ids, meas = vision.generate_measurements(Rsb @ Rbc, Rsb @ cfg.Tbc + Tsb, K, 640, 480, 1.0)   # cfg extrinsics, as read_cfg_data does
                est.VisualMeasPointCloud(ts, ids, meas)
                gsb = est.gsb()                       # pose right after the camera update
        assert gsb.shape == (3, 4) and est.gsc().shape == (3, 4) and est.Pstate().shape == (9, 9) and est.P().shape == (203, 203)
        assert est.VisionInitialized() and est.now() == int(round((total - imu_dt) * 1e9))    # the last message was an IMU one
        n = est.num_instate_features()
        assert 10 < n <= 30 and est.InstateFeatureIDs().shape == (n,) and est.InstateFeaturePositions().shape == (n, 3)
        assert est.InstateFeatureIDs().min() >= 10000 and len(set(est.InstateFeatureIDs())) == n
        assert est.InstateGroupPoses().shape == (est.num_instate_groups(), 7) and est.InstateGroupCovs().shape == (6 * est.num_instate_groups(), 6)
        assert set(est.InstateFeatureRefGroups()) <= set(est.InstateGroupIDs())
        assert np.allclose(np.linalg.norm(est.InstateGroupPoses()[:, :4], axis=1), 1.0, atol=1e-9)
        # in-state landmarks sit on points of the simulated world (the filter's map is consistent with the world)
        Xs = est.InstateFeaturePositions()
        d = np.linalg.norm(Xs[:, None, :] - vision.Xs[None], axis=2).min(axis=1)
        assert np.median(d) < 0.4, d      # (log-depth prior std 0.1: ~10 % of a 3..10 m range along the ray)
        Rt, Tt = imu.gsb(total - vis_dt)
        assert np.linalg.norm(gsb[:, 3] - Tt) < 0.1 and np.abs(gsb[:, :3] - Rt).max() < 0.02
        # same sequence through the batched driver
        sims = [pcw.TrajectorySim("lissajous", seed=41)]; worlds = [pcw.RandomPCW(seed=5)]
        cfg_b = pyxivo.config_from_cfg(pyxivo.load_json_with_comments(os.path.join(here, "golden", "pcw_like_cfg.json")))
        out = sequence.run_pcw(sequence.HipBackend, cfg_b, worlds, sims, total_time=total)
        try:
            # (not bit for bit: the client's integer-nanosecond stamps give dt = 0.0025 to the last ulp only)
            assert np.abs(out["Tsb"][-1, 0] - gsb[:, 3]).max() < 1e-9
        finally:
            out["backend"].close()
    finally:
        est.close()


def test_cpp_batch_estimator_equals_python_runner(built):
    """xivo::hip::BatchEstimator (C++ host side: IMU bookkeeping, slot book-keeping, edit lists) fed the same messages
    as the Python runner takes the same decisions every frame - identical slot books - and ends in the same state (the
    only arithmetic on the host is the feature initialisation, log() of libm vs numpy: last-ulp differences)."""
    B = 4
    cfg = sequence.SequenceConfig()
    mk = lambda: ([pcw.RandomPCW(seed=20 + b) for b in range(B)],
                  [pcw.TrajectorySim("trefoil" if b % 2 else "lissajous", seed=400 + b) for b in range(B)])
    w1, s1 = mk()
    py = sequence.run_pcw(sequence.HipBackend, cfg, w1, s1, total_time=1.0)
    w2, s2 = mk()
    cp = sequence.run_pcw_cpp(cfg, w2, s2, total_time=1.0)
    try:
        assert np.array_equal(py["ts"], cp["ts"])
        for b in range(B):
            fid, fref, gref = cp["estimator"].book(b)
            bk = py["runner"].books[b]
            assert list(fid) == bk.feat_id and list(fref) == bk.feat_ref and list(gref) == bk.group_refs
        st = cp["estimator"].stats()
        assert st["updates"] == py["runner"].n_updates and st["mh_rejected"] == py["runner"].n_rejected
        assert np.abs(py["Tsb"] - cp["Tsb"]).max() < 1e-10 and np.abs(py["Wsb"] - cp["Wsb"]).max() < 1e-10
        assert 0 < st["host_seconds"] < 1.0
    finally:
        py["backend"].close(); cp["estimator"].close()


def test_vectorised_batch_run_tracks_ground_truth(built):
    """256 sequences through the vectorised simulators and xivo::hip::BatchEstimator: every trajectory follows its
    ground truth."""
    cfg = sequence.SequenceConfig()
    out = sequence.run_pcw_batch(cfg, 256, total_time=1.6)
    try:
        ate = np.sqrt(np.mean(np.sum((out["Tsb"] - out["gt_Tsb"]) ** 2, axis=2), axis=0))
        assert np.isfinite(ate).all() and np.median(ate) < 0.05 and ate.max() < 0.3, np.sort(ate)[-5:]
        st = out["estimator"].stats()
        assert st["updates"] >= 256 * 38
    finally:
        out["estimator"].close()


@pytest.mark.parametrize("extra", [[], ["-host", "cpp"], ["-vectorized"], ["-integration_method", "RK4", "-as_coded_group_block"]])
def test_run_pcw_cli(built, extra, tmp_path):
    """scripts/run_pcw.py end to end (the three host sides, both integrators): one JSON report line, sane tracking error."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "scripts", "run_pcw.py"), "-sequences", "6", "-total_time", "0.6"] + extra
    if "-vectorized" not in extra:
        cmd += ["-dump", str(tmp_path)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    assert rep["sequences"] == 6 and rep["frames_per_sequence"] == 15 and rep["updates"] >= 6 * 13
    assert rep["ate_m"]["max"] < 0.2
    if "-vectorized" not in extra:
        ts, T, W = formats.read_trajectory(str(tmp_path / "seq0003.txt"))
        assert len(ts) == 15 and ts[1] == 40_000_000 and np.isfinite(T).all() and np.isfinite(W).all()


def test_cpp_batch_estimator_equals_python_runner_under_rejections(built):
    """Same comparison with a gate far tighter than the innovation statistics (chi-square threshold 0.02 instead of
    5.991; with cfg/pcw.json's process noise the predicted uncertainty, ~11 px, dominates S and the distances of good
    features are ~0.04): MH gating now rejects features every frame (they leave the state and are candidates again at
    once) - the paths a well-tuned run never takes."""
    B = 4
    cfg = sequence.SequenceConfig(MH_thresh=0.02)
    mk = lambda: ([pcw.RandomPCW(seed=60 + b) for b in range(B)],
                  [pcw.TrajectorySim("trefoil" if b % 2 else "lissajous", seed=500 + b) for b in range(B)])
    w1, s1 = mk()
    py = sequence.run_pcw(sequence.HipBackend, cfg, w1, s1, total_time=1.0)
    w2, s2 = mk()
    cp = sequence.run_pcw_cpp(cfg, w2, s2, total_time=1.0)
    try:
        st = cp["estimator"].stats()
        assert py["runner"].n_rejected > 50 and st["mh_rejected"] == py["runner"].n_rejected
        for b in range(B):
            fid, fref, gref = cp["estimator"].book(b)
            bk = py["runner"].books[b]
            assert list(fid) == bk.feat_id and list(fref) == bk.feat_ref and list(gref) == bk.group_refs
        assert np.abs(py["Tsb"] - cp["Tsb"]).max() < 1e-9
    finally:
        py["backend"].close(); cp["estimator"].close()


def test_sequences_with_one_point_ransac_python_and_cpp_hosts_agree(built):
    """cfg use_1pt_RANSAC: Estimator::OutlierRejection runs MH gating and then OnePointRANSAC (src/manager.cpp:629-650).
    Both hosts drive xivo_hip_one_point_ransac between gating and the update; with a threshold of 0.4 px most frames have
    high-innovation features, so the partial update / rescue paths run every frame. Same slot books, same trajectories,
    and the filter still tracks ground truth."""
    B = 3
    cfg = sequence.SequenceConfig(use_1pt_RANSAC=True, ransac_thresh=0.4, ransac_Chi2=5.89)
    mk = lambda: ([pcw.RandomPCW(seed=80 + b) for b in range(B)],
                  [pcw.TrajectorySim("trefoil" if b % 2 else "lissajous", seed=700 + b) for b in range(B)])
    w1, s1 = mk()
    py = sequence.run_pcw(sequence.HipBackend, cfg, w1, s1, total_time=1.0)
    w2, s2 = mk()
    cp = sequence.run_pcw_cpp(cfg, w2, s2, total_time=1.0)
    w3, s3 = mk()
    plain = sequence.run_pcw(sequence.HipBackend, sequence.SequenceConfig(), w3, s3, total_time=1.0)
    try:
        for b in range(B):
            fid, fref, gref = cp["estimator"].book(b)
            bk = py["runner"].books[b]
            assert list(fid) == bk.feat_id and list(fref) == bk.feat_ref and list(gref) == bk.group_refs
        assert np.abs(py["Tsb"] - cp["Tsb"]).max() < 1e-9
        ate = [formats.ate_rmse(py["Tsb"][:, b], py["gt_Tsb"][:, b], align=False) for b in range(B)]
        assert max(ate) < 0.08, ate
        # RANSAC really changed something (features were rejected that plain MH gating keeps)
        assert py["runner"].n_rejected > plain["runner"].n_rejected
    finally:
        py["backend"].close(); cp["estimator"].close(); plain["backend"].close()


def test_sequences_in_the_invdepth_build_python_and_cpp_hosts_agree(built):
    """The reference's USE_INVDEPTH build end to end (src/feature.cpp:98-105, :144-150): features initialised as
    (x, y, 1 / z), XIVO_HIP_FLAG_INVDEPTH on the context; both hosts take the same decisions and follow ground truth.
    initial_std_z is an inverse-depth standard deviation in this build."""
    B = 4
    cfg = sequence.SequenceConfig(use_invdepth=True, initial_std_z=0.05)
    mk = lambda: ([pcw.RandomPCW(seed=60 + b) for b in range(B)],
                  [pcw.TrajectorySim("trefoil" if b % 2 else "lissajous", seed=500 + b) for b in range(B)])
    w1, s1 = mk()
    py = sequence.run_pcw(sequence.HipBackend, cfg, w1, s1, total_time=1.6)
    w2, s2 = mk()
    cp = sequence.run_pcw_cpp(cfg, w2, s2, total_time=1.6)
    try:
        for b in range(B):
            fid, fref, gref = cp["estimator"].book(b)
            bk = py["runner"].books[b]
            assert list(fid) == bk.feat_id and list(fref) == bk.feat_ref and list(gref) == bk.group_refs
        assert np.abs(py["Tsb"] - cp["Tsb"]).max() < 1e-9 and np.abs(py["Wsb"] - cp["Wsb"]).max() < 1e-9
        ate = np.array([formats.ate_rmse(py["Tsb"][:, b], py["gt_Tsb"][:, b], align=False) for b in range(B)])
        assert np.isfinite(ate).all() and ate.max() < 0.3, ate
        _, _, f = py["backend"].ctx.get_scene()
        x2 = f["x"][..., 2][f["sind"] >= 0]
        assert (x2 > 0.05).all() and (x2 < 25.0).all()          # inverse depths of points 0.05 .. 10 m away, not log depths
    finally:
        py["backend"].close(); cp["estimator"].close()
