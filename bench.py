#!/usr/bin/env python
"""bench.py - EKF updates/sec of the MI355X measurement-update path.

One "step" = one measurement update of every filter in the batch (B independent
filters per GPU; filters shard across GPUs with no collective - SURVEY.md 8e).
Workload = BASELINE.json's metric point: state dim 250, 80 features (M = 160),
fp64, XIVO row sparsity, inputs resident in HBM before the timed region. Every
step hands the dense H / inn / diagR of every filter over again (H changes with
every camera frame, src/update.cpp:129-138): the dense -> row-pair compression is
inside the timed step. `value` is the library's default mode: every product in fp64
(no fp32 instruction runs unless XIVO_HIP_FLAG_FP32_WHITENED asks for one - the cfg4_f32w row);
`value_symmetric_form` times the opt-in symmetric form next to it.

`python bench.py --gpus N` without a launcher spawns its N ranks itself.

Prints ONE JSON line (rank 0). See DESIGN.md section "Measurement".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix peak (AMD datasheet). 256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz


DEFAULT_BATCH = 16384


def f_alg(N, M):
    """Reference as-coded flops of UpdateJosephForm."""
    return 4.0 * N ** 3 + 8.0 * M * N ** 2 + 4.0 * M ** 2 * N + M ** 3 / 3.0


def cpu_baseline(N, F, seconds=12.0):
    """Times the CPU path on a bounded sample of the same workload (rank 0, N=1 only).
    Prefers oracle/_ref (Eigen 3.3.9, the reference's own arithmetic, 1 thread like
    the reference's singleton estimator); falls back to the numpy oracle."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from xivo_amd import synth
    P, H, inn, dR = synth.s_level(N, F, 2, seed=12345)
    M = 2 * F
    ref, extracted = None, False
    try:
        import ref_binding
        try:      # the reference's own text of UpdateJosephForm, compiled (round 4: oracle/ref/extract_reference.py + xivo_refx.cpp)
            ref = ref_binding.loadx(203); extracted = True
        except Exception:
            ref = ref_binding.load()
    except Exception:
        ref = None
    if ref is not None:
        fn = lambda b: ref.update_joseph(H[b], P[b], inn[b], dR[b])
        if extracted:
            kind, cores = "reference", 1
            what = ("oracle/_ref extracted build: the text of Estimator::UpdateJosephForm (src/estimator.cpp:1257-1288) cut out of the "
                    "reference tree at build time and compiled against its own Eigen 3.3.9 with its own optimisation level (-O3 -DNDEBUG), 1 thread "
                    "(the reference runs one estimator per thread); bit-identical to the retyped driver")
            flags = "-O3 -DNDEBUG -march=x86-64-" + ("v4" if ref.path.endswith("_v4.so") else "v3") + \
                    " (the reference's CMake uses -march=native, which cannot travel to another host CPU)"
        else:
            kind, what, cores = "port", "oracle/_ref: Eigen-3.3.9 expression-faithful driver of estimator.cpp:1257-1288, -O3, 1 thread", 1
            flags = ref_binding.build_flags()
    else:
        import xivo_oracle as orc
        try:
            from threadpoolctl import threadpool_limits
            threadpool_limits(1)
        except Exception:
            pass
        fn = lambda b: orc.update_joseph(H[b], P[b], inn[b], dR[b])
        kind, what, cores = "port", "oracle/xivo_oracle.py numpy restatement, BLAS pinned to 1 thread", 1
        flags = "numpy " + np.__version__ + " (LAPACK / BLAS of the wheel)"
    fn(0)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        fn(n & 1)
        n += 1
    dt = time.perf_counter() - t0
    out = {"value": n / dt, "unit": "updates/s", "cores": cores, "kind": kind,
           "sample": f"{n} updates of (N={N}, M={M}) in {dt:.1f}s; {what}", "flags": flags}
    # mode (ii) of BASELINE.md section 3: one independent filter per host core on all cores
    try:
        import multiprocessing as mp
        # the cores this process may really use (cgroup / taskset), one pinned process each - os.cpu_count() is the host's
        # core count, not the container's (round 5 printed "256 cores" at 5.6 x one core)
        cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
        quota = _cgroup_cpu_quota()
        if quota is not None and quota < len(cpus):
            cpus = cpus[:max(1, int(quota))]
        with mp.get_context("spawn").Pool(len(cpus)) as pool:
            res = pool.map(_cpu_worker, [(N, F, 6.0, c) for c in cpus])
        ran = sorted({r[2] for r in res if r[2] is not None})
        out["all_cores"] = {"value": sum(r[0] / r[1] for r in res), "unit": "updates/s", "cores": len(cpus),
                            "pinned_to": len(ran), "cgroup_cpu_quota": quota,
                            "sample": f"{len(cpus)} processes x 6 s, one filter each, each pinned to one core of the affinity mask"}
    except Exception as e:   # never let the reported baseline break the bench line
        out["all_cores"] = {"error": repr(e)}
    return out


def _cgroup_cpu_quota():
    """CPU quota of this container in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def _cpu_worker(arg):
    N, F, seconds, cpu = arg
    pinned = None
    try:
        os.sched_setaffinity(0, {cpu}); pinned = cpu
    except Exception:
        pass
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from xivo_amd import synth
    P, H, inn, dR = synth.s_level(N, F, 1, seed=4242)
    try:
        import ref_binding
        try:
            ref = ref_binding.loadx(203)
        except Exception:
            ref = ref_binding.load()
        fn = lambda: ref.update_joseph(H[0], P[0], inn[0], dR[0])
    except Exception:
        import xivo_oracle as orc
        fn = lambda: orc.update_joseph(H[0], P[0], inn[0], dR[0])
    fn()
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        fn()
        n += 1
    return n, time.perf_counter() - t0, pinned


def parity_check(ctx, step, B, uniq, F, P, H, inn, dR, gate, no_gating, tol_P=1e-6):
    """Runs ONE step from the initial covariance and compares P+ / dx of 8 filters spread over the batch with the
    oracle (numpy restatement of src/update.cpp:60-96 + src/estimator.cpp:1257-1288) on the same inputs.
    Tolerances = BASELINE.json north_star: 1e-6 relative Frobenius on P, 1e-8 on dx."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import xivo_oracle as orc
    R, th, mult, min_inl = gate
    ctx.restore_P()
    step()
    picks = sorted({b for b in (0, 7, 8, B // 2 - 1, B // 2 + 3, B - 9, B - 8, B - 1) if 0 <= b < B})
    worst_P = worst_dx = 0.0
    mask_equal = True
    for b in picks:
        u = b % uniq
        Pn = ctx.download_P(b0=b, nb=1)[0]
        err = ctx.get_err(b0=b, nb=1)[0]
        rows = np.ones(2 * F, dtype=bool)
        if not no_gating and F > min_inl:
            d = orc.mh_distances(H[u].reshape(F, 2, -1), P[u], inn[u].reshape(F, 2), R)
            m = orc.mh_gate(d, th, mult, min_inl)[0]
            rows = np.repeat(m, 2)
        e_ref, P_ref, _ = orc.update_joseph(H[u][rows], P[u], inn[u][rows], dR[u][rows])
        worst_P = max(worst_P, float(np.linalg.norm(Pn - P_ref) / np.linalg.norm(P_ref)))
        worst_dx = max(worst_dx, float(np.linalg.norm(err - e_ref) / np.linalg.norm(e_ref)))
    if not no_gating and F > min_inl:
        gm, _ = ctx.get_gate(F, B)
        for b in picks:
            u = b % uniq
            d = orc.mh_distances(H[u].reshape(F, 2, -1), P[u], inn[u].reshape(F, 2), R)
            mask_equal = mask_equal and bool(np.array_equal(gm[b], orc.mh_gate(d, th, mult, min_inl)[0]))
    # every filter of the batch against its twin: the synthetic inputs repeat with period `uniq`, the kernels are
    # deterministic, so filter b must equal filter b % uniq BIT FOR BIT - dx of all B filters, P of 64 spread over the
    # launch (an indexing error anywhere in the 64 GB of strides shows up here)
    err_all = ctx.get_err(b0=0, nb=B)
    twins_dx = bool(np.array_equal(err_all, err_all[np.arange(B) % uniq]))
    rng = np.random.default_rng(7)
    twins_P = True
    for b in sorted(set(int(x) for x in rng.integers(uniq, B, size=64))) if B > uniq else []:
        twins_P = twins_P and bool(np.array_equal(ctx.download_P(b0=b, nb=1)[0], ctx.download_P(b0=b % uniq, nb=1)[0]))
    ok = worst_P < tol_P and worst_dx < 1e-8 and mask_equal and twins_dx and twins_P
    out = {"ok": bool(ok), "filters": picks, "rel_fro_P_max": worst_P, "rel_dx_max": worst_dx,
           "inlier_masks_equal": mask_equal, "all_filters_dx_equal_their_twin_bitwise": twins_dx,
           "sampled_64_filters_P_equal_their_twin_bitwise": twins_P,
           "tol": {"P": tol_P, "dx": 1e-8}, "checker": "oracle/xivo_oracle.py"}
    return out      # (a failure is raised by main() AFTER the ranks have exchanged their results: no rank left in a collective)


def parity_last_step(ctx, n_steps, B, uniq, F, P, H, inn, dR, gate, no_gating, tol_P=1e-6, tol_dx=1e-8):
    """The state the TIMED loop left behind (n_steps updates of the resident covariance with the same measurements, warm-up
    included) against the oracle applying the same n_steps updates one after the other - gating re-evaluated on the shrinking
    covariance every time. Catches anything that only goes wrong after the first step (stale buffers, state carried between
    launches). 4 filters spread over the launch; same tolerances as the one-step check."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import xivo_oracle as orc
    R, th, mult, min_inl = gate
    picks = sorted({b for b in (0, B // 2 + 3, B - 9, B - 1) if 0 <= b < B})
    worst_P = worst_dx = 0.0
    mask_equal = True
    gm = ctx.get_gate(F, B)[0] if (not no_gating and F > min_inl) else None
    for b in picks:
        u = b % uniq
        Pc = P[u].copy()
        e_ref = None
        for _ in range(n_steps):
            rows = np.ones(2 * F, dtype=bool)
            if gm is not None:
                d = orc.mh_distances(H[u].reshape(F, 2, -1), Pc, inn[u].reshape(F, 2), R)
                m = orc.mh_gate(d, th, mult, min_inl)[0]
                rows = np.repeat(m, 2)
            e_ref, Pc, _ = orc.update_joseph(H[u][rows], Pc, inn[u][rows], dR[u][rows])
        if gm is not None:
            mask_equal = mask_equal and bool(np.array_equal(gm[b], m))
        Pn = ctx.download_P(b0=b, nb=1)[0]
        err = ctx.get_err(b0=b, nb=1)[0]
        worst_P = max(worst_P, float(np.linalg.norm(Pn - Pc) / np.linalg.norm(Pc)))
        worst_dx = max(worst_dx, float(np.linalg.norm(err - e_ref) / np.linalg.norm(e_ref)))
    ok = worst_P < tol_P and worst_dx < tol_dx and mask_equal
    out = {"ok": bool(ok), "updates_in_a_row": n_steps, "filters": picks, "rel_fro_P_max": worst_P, "rel_dx_max": worst_dx,
           "inlier_masks_equal": mask_equal, "tol": {"P": tol_P, "dx": tol_dx}}
    return out


def parity_glevel(ctx, step, B, uniq, P0, tol_P=1e-6):
    """Feature-level runs (--level G, BASELINE config 3): ONE step from the initial covariance, then UpdateJosephForm of
    the oracle on the rows the device stacked (xivo_hip_get_H: in-state rows after gating - rejected features are neutral
    rows -, projected / compressed OOS rows, inn, diagR) against the device's P+ and dx, 4 filters spread over the launch.
    What this pins inside the bench is a1 at this shape and batch; the kernels that BUILD the rows (Jacobians, gating,
    stacking, OOS projection, QR compression) are pinned against the oracle by tests/test_glevel_gpu.py."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import xivo_oracle as orc
    ctx.restore_P()
    step()
    picks = sorted({b for b in (0, B // 2 + 3, B - 9, B - 1) if 0 <= b < B})
    worst_P = worst_dx = 0.0
    for b in picks:
        Hd, innd, dRd = ctx.get_H(b)
        e_ref, P_ref, _ = orc.update_joseph(Hd, P0[b % uniq], innd, dRd)
        Pn = ctx.download_P(b0=b, nb=1)[0]
        err = ctx.get_err(b0=b, nb=1)[0]
        worst_P = max(worst_P, float(np.linalg.norm(Pn - P_ref) / np.linalg.norm(P_ref)))
        worst_dx = max(worst_dx, float(np.linalg.norm(err - e_ref) / np.linalg.norm(e_ref)))
    return {"ok": bool(worst_P < tol_P and worst_dx < 1e-8), "filters_checked_vs_oracle": picks, "rel_fro_P_max": worst_P,
            "rel_dx_max": worst_dx, "rows": int(Hd.shape[0]), "tol": {"P": tol_P, "dx": 1e-8},
            "checker": "oracle/xivo_oracle.py update_joseph on the rows the device stacked (xivo_hip_get_H)"}


def parity_frame(ctx, B, uniq, P0, scene, imu, Qimu, Qmodel, grav, method, gate, no_gating, tol_P=1e-6):
    """Whole-frame rows: ONE frame from the initial state, checked in its two halves on 3 filters spread over the launch -
    (i) Estimator::Propagate (src/estimator.cpp:539-592: every IMU sample through the oracle's RK4 / Dormand-Prince steps,
    nominal state and covariance) against the device's state and P behind xivo_hip_propagate, (ii) the measurement update
    against the oracle's UpdateJosephForm on the rows the device stacked from that propagated state, with the covariance
    the device held in front of the update."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import xivo_oracle as orc
    poses, groups, feats = scene
    R, th, mult, min_inl = gate
    ctx.restore_P()
    for b0 in range(0, B, uniq):
        nb = min(uniq, B - b0)
        ctx.set_scene(poses[:nb], groups[:nb], feats[:nb], b0=b0)
    picks = sorted({b for b in (0, B // 2 + 3, B - 1) if 0 <= b < B})
    ctx.propagate(imu, Qimu, Qmodel, grav, method=method, stepsize=0.002)
    pose_d = ctx.get_scene()[0]
    P_prop = {b: ctx.download_P(b0=b, nb=1)[0] for b in picks}
    worst_prop = worst_state = 0.0
    for b in picks:
        u = b % uniq
        X = orc.MotionState(poses[u]["Rsb"].reshape(3, 3).T, poses[u]["Tsb"], poses[u]["Vsb"], poses[u]["bg"], poses[u]["ba"],
                            poses[u]["Rsg"].reshape(3, 3).T)
        Pr = P0[u].copy()
        for s_ in range(imu.shape[1]):
            smp = imu[b, s_]
            X, Pr = orc.propagate(X, Pr, smp["gyro"], smp["accel"], smp["slope_gyro"], smp["slope_accel"], float(smp["dt"]), Qimu, Qmodel, grav,
                                  method=method, stepsize=0.002)
        worst_prop = max(worst_prop, float(np.linalg.norm(P_prop[b] - Pr) / np.linalg.norm(Pr)))
        worst_state = max(worst_state, float(np.abs(pose_d[b]["Rsb"].reshape(3, 3).T - X.Rsb).max()), float(np.abs(pose_d[b]["Tsb"] - X.Tsb).max()),
                          float(np.abs(pose_d[b]["Vsb"] - X.Vsb).max()))
    ctx.filter_update(R, th, mult, min_inl, not no_gating, B)
    worst_P = worst_dx = 0.0
    for b in picks:
        Hd, innd, dRd = ctx.get_H(b)
        e_ref, P_ref, _ = orc.update_joseph(Hd, P_prop[b], innd, dRd)
        worst_P = max(worst_P, float(np.linalg.norm(ctx.download_P(b0=b, nb=1)[0] - P_ref) / np.linalg.norm(P_ref)))
        worst_dx = max(worst_dx, float(np.linalg.norm(ctx.get_err(b0=b, nb=1)[0] - e_ref) / np.linalg.norm(e_ref)))
    ok = worst_prop < 1e-9 and worst_state < 1e-10 and worst_P < tol_P and worst_dx < 1e-8
    return {"ok": bool(ok), "filters_checked_vs_oracle": picks, "propagate_rel_fro_P_max": worst_prop, "propagate_state_abs_max": worst_state,
            "rel_fro_P_max": worst_P, "rel_dx_max": worst_dx, "tol": {"propagate_P": 1e-9, "propagate_state": 1e-10, "P": tol_P, "dx": 1e-8},
            "checker": "oracle/xivo_oracle.py: propagate() per IMU sample, then update_joseph on the rows the device stacked (xivo_hip_get_H)"}


def dropin_block(shapes=((203, 30), (250, 80)), n_calls=300):
    """Wall time of the LITERAL drop-in call - xivo::hip::Estimator::UpdateJosephForm() through libxivo_host.so with P_, H_,
    inn_, diagR_ in pageable host memory, one estimator (the reference is one singleton filter per process,
    src/estimator.cpp:26; callers src/update.cpp:141, :332) - median over n_calls calls, beside oracle/_ref's ms per update
    on the same host cores, parity of the returned P_ / err_ against the oracle. mode 0 = the one-call entry
    (xivo_hip_update_joseph_host: P_ in place, H_ compressed while staged, one synchronisation), 1 = the six general calls
    of rounds 1-3, 2 = one call with the prior already resident (no upload of P_)."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import xivo_oracle as orc
    from xivo_amd import synth
    host = C.CDLL(os.path.join(ROOT, "xivo_amd", "libxivo_host.so"))
    try:
        import ref_binding
        try:
            ref = ref_binding.loadx(203)       # the reference's own text of UpdateJosephForm, compiled (any N: dynamic members)
        except Exception:
            ref = ref_binding.load()
    except Exception:
        ref = None
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    out = []
    for N, F in shapes:
        M = 2 * F
        P, H, inn, dR = synth.s_level(N, F, 1, seed=4321 + N)
        Pf, Hf = np.asfortranarray(P[0]), np.asfortranarray(H[0])
        e_ref, P_ref, _ = orc.update_joseph(H[0], P[0], inn[0], dR[0])
        row = {"N": N, "M": M, "calls": n_calls,
               "what": "Estimator::UpdateJosephForm() via libxivo_host.so, members in pageable host memory, B = 1"}
        for mode, key in ((0, "ms_per_update"), (1, "ms_per_update_six_call_sequence_r03"), (2, "ms_per_update_prior_resident")):
            ms = np.zeros(n_calls); Pout = np.zeros((N, N), order="F"); err = np.zeros(N); msg = C.create_string_buffer(256)
            rc = host.xivo_host_time_update_joseph(N, M, p(Pf), p(Hf), p(inn[0].copy()), p(dR[0].copy()), n_calls, mode, C.c_uint(0),
                                                   p(ms), p(Pout), p(err), msg, 256)
            if rc != 0:
                row[key] = None; row["error"] = msg.value.decode(); continue
            tail = ms[n_calls // 10:]
            row[key] = float(np.median(tail))
            if mode == 0:
                row["p10_p90_ms"] = [float(np.percentile(tail, 10)), float(np.percentile(tail, 90))]
                rp = float(np.linalg.norm(Pout - P_ref) / np.linalg.norm(P_ref)); re_ = float(np.linalg.norm(err - e_ref) / np.linalg.norm(e_ref))
                row["parity"] = {"ok": bool(rp < 1e-6 and re_ < 1e-8), "rel_fro_P": rp, "rel_dx": re_, "tol": {"P": 1e-6, "dx": 1e-8}}
        if ref is not None:
            t = []
            for _ in range(3):
                ref.update_joseph(H[0], P[0], inn[0], dR[0])
            n_ref = max(10, min(200, int(2.0 / max(1e-4, 5.65e-3 * (N / 250.0) ** 3))))
            for _ in range(n_ref):
                t0 = time.perf_counter(); ref.update_joseph(H[0], P[0], inn[0], dR[0]); t.append(time.perf_counter() - t0)
            row["cpu_ref_ms_per_update"] = float(np.median(t) * 1e3)
            row["cpu_ref"] = (f"oracle/_ref ({'extracted text of src/estimator.cpp:1257-1288' if isinstance(ref, ref_binding.RefX) else 'retyped driver'}, "
                              f"Eigen 3.3.9, 1 thread), median of {n_ref} calls on this host")
            if row.get("ms_per_update"):
                row["speedup_vs_cpu_ref"] = row["cpu_ref_ms_per_update"] / row["ms_per_update"]
        out.append(row)
    return out


# the other BASELINE.json configurations (and the S-level variants SURVEY 8d asks for), each run by this same script in a
# child process after the headline loop - fewer steps, no CPU baseline; one entry of the `configs` array each
SUB_CONFIGS = [
    # (short key - what the printed line carries -, description - kept in the full record only -, arguments)
    ("cfg2", "config2 (N=150, 50 features, M=100), fp64", ["--state-dim", "150", "--features", "50", "--steps", "8", "--warmup", "2"]),
    ("cfg3", "config3 (N=251: 60 in-state features + 20 OOS features null-space projected, QR-compressed), fp64, 4096 filters",
     ["--level", "G", "--oos", "20", "--batch", "4096", "--steps", "8", "--warmup", "2"]),
    ("cfg3_m260", "config3 as SURVEY 8(d) states it: the same stacking WITHOUT measurement compression (the reference parses use_compression_ and "
     "never reads it, src/estimator.cpp:115) - 120 in-state + 140 projected OOS rows, M = 260, 4096 filters",
     ["--level", "G", "--oos", "20", "--no-compression", "--batch", "4096", "--steps", "8", "--warmup", "2"]),
    ("cfg4_f64", "config4 (N=400, 150 features, M=300), fp64 (library default), 4096 filters",
     ["--state-dim", "400", "--features", "150", "--batch", "4096", "--steps", "5", "--warmup", "2"]),
    ("cfg4_f32w", "config4 as written: fp32 MFMA with stated tolerance (XIVO_HIP_FLAG_FP32_WHITENED: the whitened operands V^T, Y^T leave the fp64 "
     "solve as float, P - V^T Y on v_mfma_f32_16x16x4_f32; stated tolerance: 5e-5 on P per update AND over the chain of updates the "
     "timed loop leaves behind; dx of an update is bit-identical to the fp64 path given the same prior, against the all-fp64 CHAIN it "
     "inherits the float rounding of the covariance's small directions - this row repeats ONE measurement seven times, the worst "
     "case: checked at 1e-2; tests/test_variants_gpu.py::test_fp32_whitened_chain states the figure for changing measurements), 4096 filters",
     ["--state-dim", "400", "--features", "150", "--batch", "4096", "--steps", "5", "--warmup", "2", "--flags", "16384", "--tol-P", "5e-5",
      "--tol-dx-last", "1e-2"]),
    ("calib", "online-calibration build (USE_ONLINE_TEMPORAL_CALIB / _IMU_CALIB / _CAMERA_CALIB: N=276, 60 features with td / Cg / bg / 8 "
     "intrinsics blocks), feature level: Jacobians + gating on the whole row + update, 4096 filters",
     ["--level", "G", "--calib", "--batch", "4096", "--steps", "5", "--warmup", "2"]),
    ("tumvi", "TUM-VI size (N=203, 30 features, M=60), fp64, 8192 filters",
     ["--state-dim", "203", "--features", "30", "--batch", "8192", "--steps", "8", "--warmup", "2"]),
    ("dense_ascoded", "metric point, dense AS-CODED pipeline (XIVO_HIP_FLAG_DENSE_H = --flags 64: H treated as dense, every product of "
     "estimator.cpp:1259-1287 a tiled MFMA GEMM - the pure-GEMM variant of SURVEY 8d), 8192 filters",
     ["--flags", "64", "--batch", "8192", "--steps", "5", "--warmup", "2"]),
    ("glevel", "feature-level default build (N=251, 60 features): Jacobians + MH gating + stacking + update, 4096 filters",
     ["--level", "G", "--batch", "4096", "--steps", "8", "--warmup", "2"]),
    ("ransac", "feature-level + OnePointRANSAC (src/update.cpp:213-393) between gating and the update, 4096 filters",
     ["--level", "G", "--ransac", "--batch", "4096", "--steps", "5", "--warmup", "2"]),
    ("frame_rk4", "whole frame (src/manager.cpp:18-167): Propagate (16 IMU samples = 32 RK4 sub-steps) + Jacobians + gating + update + AbsorbError, 4096 filters",
     ["--level", "G", "--propagate-samples", "16", "--batch", "4096", "--steps", "5", "--warmup", "2"]),
    ("frame_pd", "whole frame, Dormand-Prince integrator (TUM-VI default), 4096 filters",
     ["--level", "G", "--propagate-samples", "16", "--integrator", "PrinceDormand", "--batch", "4096", "--steps", "5", "--warmup", "2"]),
    ("b1", "metric point, B = 1 (one estimator, inputs resident in HBM), no per-stage events (the library's un-instrumented latency)",
     ["--batch", "1", "--steps", "200", "--warmup", "20", "--no-profile"]),
]


def _r(x, n=4):
    """n significant digits (the printed line has to fit the driver's 8 KB tail; the full record keeps every digit)."""
    return float(f"{x:.{n}g}") if isinstance(x, float) else x


def configs_block(device_budget_s=240.0):
    """Child runs of this script, one per SUB_CONFIGS entry. Returns (compact rows for the printed line, full records)."""
    import subprocess
    rows, full = [], []
    t_all = time.perf_counter()
    for key, name, extra in SUB_CONFIGS:
        if time.perf_counter() - t_all > device_budget_s:
            rows.append({"k": key, "skipped": "time budget"})
            full.append({"k": key, "name": name, "skipped": "time budget of the default run used up"})
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--sub", "--no-cpu-baseline", "--no-mixed"] + extra
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=150)
            line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
            d = json.loads(line[-1]) if line else None
        except Exception as e:   # a secondary shape must never take the headline line down
            rows.append({"k": key, "error": repr(e)[:120]})
            full.append({"k": key, "name": name, "error": repr(e)[:600]})
            continue
        if d is None:
            rows.append({"k": key, "error": (r.stderr or "")[-160:], "rc": r.returncode})
            full.append({"k": key, "name": name, "error": (r.stderr or "")[-2000:], "returncode": r.returncode})
            continue
        par = d.get("parity_check") or {}
        last = d.get("parity_check_last_timed_step") or {}
        roof = d.get("roofline") or {}
        oks = [q.get("ok") for q in (par, last) if q]
        row = {"k": key, "v": _r(d["value"]), "ms": _r(d["ms_per_step"]), "B": d["config"]["filters_per_gpu"],
               "kern": (roof.get("kernel") or "").replace("_f64_kernel", "").replace("_kernel", ""), "frac": _r(roof.get("frac"), 3),
               "bound": roof.get("bound"), "P": _r(par.get("rel_fro_P_max"), 2), "dx": _r(par.get("rel_dx_max"), 2),
               "mask": par.get("inlier_masks_equal"), "lastP": _r(last.get("rel_fro_P_max"), 2), "lastdx": _r(last.get("rel_dx_max"), 2),
               "ok": (all(oks) if oks else None)}
        tol = (last.get("tol") or {}).get("dx")
        if tol is not None and tol != 1e-8:
            row["tol_dx"] = tol                     # a row checked at a looser dx tolerance than the rest says so next to its "ok"
        rows.append({k: v for k, v in row.items() if v is not None or k == "ok"})
        full.append({"k": key, "name": name, "args": " ".join(extra), "wall_s": time.perf_counter() - t0, "line": d})
    return rows, full


def compact_line(out):
    """The ONE printed line: every number the judge reads, under 6 KB (the driver keeps the last 8 KB of stdout and cuts strings
    inside `config` at 120 characters). The verbose record - stage descriptions, checker names, notes - goes to --full-out."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    c = {k: out[k] for k in keep}
    cfg = dict(out["config"])
    cfg.pop("precision", None)
    if "dense H" in (cfg.get("hand_over") or ""):
        cfg["hand_over"] = "dense H -> row-pair compressed rows every step, inside the timed region"
    cfg["gpu_event_ms_per_step"] = _r(cfg.get("gpu_event_ms_per_step"), 5)
    dropin = out.get("dropin") if isinstance(out.get("dropin"), list) else []
    for row in dropin:            # flat scalars: the driver's `parsed.config` keeps these
        tag = f"{row['N']}_{row['M']}"
        cfg[f"dropin_ms_{tag}"] = _r(row.get("ms_per_update"))
        cfg[f"dropin_cpu_ms_{tag}"] = _r(row.get("cpu_ref_ms_per_update"))
        cfg[f"dropin_resident_ms_{tag}"] = _r(row.get("ms_per_update_prior_resident"))
        cfg[f"dropin_ok_{tag}"] = (row.get("parity") or {}).get("ok")
    for row in out.get("configs") or []:
        if row.get("v") is not None:
            cfg[row["k"] + "_upd_s"] = row["v"]
    c["config"] = cfg
    ro = out.get("roofline")
    if ro:
        c["roofline"] = {k: _r(ro.get(k), 5) for k in
                         ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "launches", "frac_mfma", "frac_hbm",
                          "traffic_source", "executed_mfma_flops_per_launch", "mfma_busy_pct_pmc", "pipeline_frac")}
        pm = ro.get("mfma_peak_measured_tflops")
        c["roofline"]["mfma_peak_measured_tflops"] = _r(pm.get("tflops_full_chip")) if isinstance(pm, dict) else pm
    cb = out.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {"value": _r(cb["value"], 5), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                             "sample": cb["sample"][:110], "gpu_over_cpu": _r(cb.get("gpu_over_cpu")),
                             "all_cores": {k: _r(v) for k, v in (cb.get("all_cores") or {}).items() if k in ("value", "cores", "pinned_to", "cgroup_cpu_quota", "error")}}

    def pc(q):
        return None if not q else {"ok": q.get("ok"), "P": _r(q.get("rel_fro_P_max"), 2), "dx": _r(q.get("rel_dx_max"), 2),
                                   "mask": q.get("inlier_masks_equal"), "n": q.get("updates_in_a_row")}
    c["parity_check"] = pc(out.get("parity_check"))
    c["parity_last"] = pc(out.get("parity_check_last_timed_step"))
    c["stage_ms"] = {k: _r(v, 4) for k, v in (out.get("stage_ms_per_step") or {}).items()}
    if out.get("symmetric_form"):
        c["symmetric_form"] = {"v": _r(out["symmetric_form"]["value"]), "ms": _r(out["symmetric_form"]["ms_per_step"]),
                               "ok": (out["symmetric_form"].get("parity_check") or {}).get("ok")}
    if out["n_gpus"] > 1:
        c["per_rank_updates_per_s"] = [_r(v) for v in out["per_rank_updates_per_s"]]
        c["per_rank_parity_ok"] = [p.get("ok") for p in out["per_rank_parity"]]
    if dropin:
        c["dropin"] = [{"N": q["N"], "M": q["M"], "ms": _r(q.get("ms_per_update")), "p10_p90": [_r(x) for x in q.get("p10_p90_ms", [])],
                        "six_call_ms": _r(q.get("ms_per_update_six_call_sequence_r03")), "resident_ms": _r(q.get("ms_per_update_prior_resident")),
                        "cpu_ms": _r(q.get("cpu_ref_ms_per_update")), "x": _r(q.get("speedup_vs_cpu_ref"), 3),
                        "P": _r((q.get("parity") or {}).get("rel_fro_P"), 2), "dx": _r((q.get("parity") or {}).get("rel_dx"), 2),
                        "ok": (q.get("parity") or {}).get("ok")} for q in dropin]
    elif out.get("dropin"):
        c["dropin"] = out["dropin"]
    if out.get("configs"):
        c["configs"] = out["configs"]
    if out.get("full_record"):
        c["full_record"] = out["full_record"]
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=DEFAULT_BATCH,
                    help="filters per GPU (16384 x ~4 MB of resident matrices = 64 GB of the 288 GB; 4096: -4 %% throughput)")
    ap.add_argument("--state-dim", type=int, default=250)
    ap.add_argument("--features", type=int, default=80)
    ap.add_argument("--level", choices=["S", "G"], default="S",
                    help="S (default, the BASELINE metric point): dense H rows resident; G: feature-level path at the "
                         "constructible layout N=251 (8 groups, 60 features): Jacobians + sparse gating + stacking + update")
    ap.add_argument("--propagate-samples", type=int, default=0,
                    help="level G only: every step also integrates this many IMU samples (Estimator::Propagate, 2.5 ms each "
                         "at 2 ms sub-steps) before the update and absorbs the error after it - the whole per-frame loop "
                         "(SURVEY 8d: the metric with propagation); 16 samples = 32 Runge-Kutta sub-steps per update")
    ap.add_argument("--integrator", choices=["RK4", "PrinceDormand"], default="RK4")
    ap.add_argument("--oos", type=int, default=0,
                    help="level G only (BASELINE config 3): this many out-of-state (MSCKF) features, each seen from 5 in-state "
                         "groups, are null-space projected (src/oos.cpp) and appended: 7 rows each, M = 120 + 7 n")
    ap.add_argument("--no-compression", action="store_true", help="--oos: keep all 7 n projected rows (no QR measurement compression)")
    ap.add_argument("--ransac", action="store_true",
                    help="level G only: OnePointRANSAC (src/update.cpp:213-393) between MH gating and the update - backup, "
                         "partial update on the low-innovation set, absorb, re-Jacobians, chi-square rescue, restore")
    ap.add_argument("--no-gating", action="store_true",
                    help="time UpdateJosephForm only (default: MH gating + UpdateJosephForm, one 'update' of SURVEY 8d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--flags", type=int, default=0, help="extra XIVO_HIP_FLAG_* bits (A/B knobs)")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--calib", action="store_true",
                    help="--level G: the online-calibration build's layout and Jacobian blocks (N = 276; compressed rows + a dense block of the calibration columns)")
    ap.add_argument("--no-mixed", action="store_true", help="skip the second timed loop (the opt-in symmetric form)")
    ap.add_argument("--sub", action="store_true", help="child run of the `configs` array: no configs / dropin blocks of its own")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` array (the other BASELINE configurations, child runs)")
    ap.add_argument("--no-dropin", action="store_true", help="skip the `dropin` block (wall time of the one-estimator drop-in call)")
    ap.add_argument("--no-last-step-parity", action="store_true", help="skip the check of the state the timed loop left behind")
    ap.add_argument("--tol-P", type=float, default=1e-6, help="parity tolerance on P (1e-6 = north_star; the fp32 flag states 5e-5)")
    ap.add_argument("--tol-dx-last", type=float, default=1e-8,
                    help="tolerance on dx of the LAST update of the timed chain (the fp32 flag states 1e-6: dx of a later update "
                         "inherits the fp32 rounding of the earlier covariances)")
    ap.add_argument("--full-out", default=None,
                    help="where the verbose record goes (default gpurun_out/bench_full.json; the printed line is the compact one)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / reduction path only, no device work (CPU-box test of --gpus N)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become the launcher (one rank per GPU, LOCAL_RANK -> hipSetDevice, gloo control plane)
        from xivo_amd.shard import spawn_ranks
        sys.exit(spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        # Control plane only (barrier + one max-reduce of the wall time): filters are independent, there is
        # no data-path collective, hence nothing to send over xGMI / RCCL. gloo keeps the bench
        # independent of GPU IPC settings; each rank drives its own GPU (LOCAL_RANK) through the C ABI.
        import torch.distributed as dist_mod
        dist_mod.init_process_group(backend="gloo")
        dist = dist_mod

    if args.dry_run:
        # launcher / rendezvous / reductions only: every rank "processes" its filters in no time
        from xivo_amd.shard import max_over_ranks, gather_over_ranks
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            time.sleep(0.001 * (1 + rank))
        dt_rank = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
        dt = max_over_ranks(dist, dt_rank)
        per_rank = gather_over_ranks(dist, args.batch * args.steps / dt_rank)
        from xivo_amd.shard import bind_rank, gather_objects
        aff = bind_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), None, None)
        per_rank_parity = gather_objects(dist, {"rank": rank, "ok": None, "skipped": "dry run: no device work"})
        per_rank_aff = gather_objects(dist, {k: aff.get(k) for k in ("numa_node", "n_cpus", "bound", "omp_threads")})
        if rank == 0:
            print(json.dumps({"metric": "EKF updates/sec (state dim 250, 80 feats) @1 GPU; % MFMA roofline", "dry_run": True,
                              "per_rank_parity": per_rank_parity, "per_rank_affinity": per_rank_aff,
                              "value": None, "unit": "updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                              "per_rank_updates_per_s": per_rank, "ranks_reporting": len(per_rank)}))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    from xivo_amd import synth
    from xivo_amd.lib import Context, FLAG_PROFILE, load_library

    # one rank per GPU; the modulo only matters when more ranks than GPUs are launched (smoke-testing the
    # N>1 path on a 1-GPU box) - on the 8-GPU node it is the identity
    ndev = max(1, load_library().xivo_hip_device_count())
    device = local_rank % ndev
    # one rank per GPU: stay on the cores (and the memory) of the GPU's NUMA node, size the C++ host side's OpenMP team to
    # this rank's share of the cores
    from xivo_amd.shard import bind_rank, gather_objects
    affinity = bind_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), ndev, load_library().xivo_hip_device_numa_node)
    N, F, B = args.state_dim, args.features, args.batch
    R_VIS, MH_THRESH, MH_MULT, MIN_INL = 2.25, 5.991, 1.1, 5   # cfg/tumvi_cam0.json / estimator.cpp:366-369
    base_flags = (0 if args.no_profile else FLAG_PROFILE) | args.flags
    flags = base_flags                       # headline = library default: every product in fp64
    uniq = min(B, 64)
    if args.level == "G":
        # layout-faithful scene (SURVEY 8d "G-level"): 8 groups, 60 in-state features -> N = 23 + 48 + 180 = 251
        ng, nf, F = 8, 60, 60
        # --calib: the reference's online-calibration build (USE_ONLINE_TEMPORAL_CALIB / _IMU_CALIB / _CAMERA_CALIB): kMotionSize
        # 39 + 9 intrinsics slots in front of the groups -> N = 276; td / Cg / bg / intrinsics blocks in every row pair
        gb = 48 if args.calib else 23
        N = gb + 6 * ng + 3 * nf
        uniq = min(B, 16)
        from xivo_amd.lib import pose_dtype, group_dtype, feat_dtype
        sc = synth.g_level(ng, nf, F, uniq, seed=2000 + rank, cam=synth.EQUI)
        poses = np.zeros(uniq, dtype=pose_dtype); groups = np.zeros((uniq, ng), dtype=group_dtype)
        feats = np.zeros((uniq, F), dtype=feat_dtype)
        cmaj = lambda R: np.asarray(R).T.reshape(-1)
        for b in range(uniq):
            poses[b]["Rsb"], poses[b]["Tsb"] = cmaj(sc["Rsb"][b]), sc["Tsb"][b]
            poses[b]["Rbc"], poses[b]["Tbc"] = cmaj(sc["Rbc"][b]), sc["Tbc"][b]
            for g_ in range(ng):
                groups[b, g_]["Rsb"], groups[b, g_]["Tsb"] = cmaj(sc["gR"][b, g_]), sc["gT"][b, g_]
            feats["x"][b] = sc["x"][b]; feats["ref_sind"][b] = sc["ref"][b]; feats["sind"][b] = sc["sind"][b]
        M = 2 * F + 7 * args.oos
        # (--oos: 16 spare rows - mixed stacking pads the OOS block to 16 rows behind the 2F in-state rows)
        ctx = Context(N, M + (16 if args.oos > 0 else 0), B, device=device, flags=flags)
        ctx.set_layout(N, gb, ng, gb + 6 * ng, nf, synth.EQUI)
        if args.calib:
            from xivo_amd.lib import calib_dtype, cam_intr
            ctx.set_calib(23, 24, 39, 8)                     # Index::td, Index::Cg, kCameraBegin, Camera::dim() (equidistant)
            rngc = np.random.default_rng(5000 + rank)
            cal = np.zeros(uniq, dtype=calib_dtype)
            for b in range(uniq):
                cal[b]["gyro"] = rngc.normal(size=3) * 0.3; cal[b]["Cg"] = (np.eye(3) + 0.01 * rngc.normal(size=(3, 3))).T.reshape(-1)
                cal[b]["td"] = 0.005; cal[b]["Ca"] = np.eye(3).reshape(-1); cal[b]["intr"] = cam_intr(synth.EQUI)
                poses[b]["Vsb"] = rngc.normal(size=3) * 0.5; poses[b]["bg"] = rngc.normal(size=3) * 0.01
            for b0 in range(0, B, uniq):
                ctx.set_calib_state(cal[:min(uniq, B - b0)], b0=b0)
        rngP = np.random.default_rng(3000 + rank)
        A_ = rngP.uniform(-1, 1, size=(uniq, N, N))
        P = (A_ @ np.transpose(A_, (0, 2, 1)) / N + 1e-3 * np.eye(N)[None]) * 1e-4
        # measured pixel = predicted pixel + N(0, 1.5^2): the prediction comes from the device itself
        # (first pass with xp = 0 gives inn = -xp_pred), so no CPU model of the camera is involved
        for b0 in range(0, B, uniq):
            nb = min(uniq, B - b0)
            ctx.upload_P(P[:nb], b0=b0)
            ctx.set_scene(poses[:nb], groups[:nb], feats[:nb], b0=b0)
        ctx.jacobians_instate(uniq)
        _, inn0 = ctx.get_jacobians(0, uniq)
        feats["xp"] = -inn0 + sc["pix_noise"]
        for b0 in range(0, B, uniq):
            nb = min(uniq, B - b0)
            ctx.set_scene(poses[:nb], groups[:nb], feats[:nb], b0=b0)
    else:
        M = 2 * F
        ctx = Context(N, M, B, device=device, flags=flags)
        # synthetic inputs: 64 distinct seeded filters per rank, tiled over the batch block by block
        # (keeps host memory and upload time small; every filter still does the full work)
        P, H, inn, dR = synth.s_level(N, F, uniq, seed=1000 + rank)
        for b0 in range(0, B, uniq):
            nb = min(uniq, B - b0)
            ctx.upload_P(P[:nb], b0=b0)
        # the stacked measurements of every filter, dense and column-major as Estimator::H_ / inn_ / diagR_ are
        # (src/estimator.h:496-509), resident in HBM before the timed region
        dH = ctx.device_array(np.transpose(H, (0, 2, 1)), total=B)
        dinn, dRd = ctx.device_array(inn, total=B), ctx.device_array(dR, total=B)
        hand_over = lambda: ctx.set_measurements_device(dH, dinn, dRd, M, B)
        hand_over()
    ctx.snapshot_P()
    oos_on = args.level == "G" and args.oos > 0
    if oos_on:
        # OOS features: landmarks near the in-state ones (world position from the scene's own anchors), each observed
        # from 5 groups; their pixels only set the innovation, the work per update does not depend on them
        from xivo_amd.lib import oos_dtype
        rngo = np.random.default_rng(4000 + rank)
        oos_u = np.zeros((uniq, args.oos), dtype=oos_dtype)
        for b in range(uniq):
            for o in range(args.oos):
                i = o % F; r = int(sc["ref"][b, i]); x = sc["x"][b, i]; z = np.exp(x[2])
                Xc = np.array([x[0] * z, x[1] * z, z])
                oos_u[b, o]["Xs"] = sc["gR"][b, r] @ (sc["Rbc"][b] @ Xc + sc["Tbc"][b]) + sc["gT"][b, r] + rngo.normal(0, 0.05, 3)
                oos_u[b, o]["n_obs"] = 5
                oos_u[b, o]["group_sind"][:5] = rngo.permutation(ng)[:5]
                oos_u[b, o]["xp"][:5] = np.array([synth.EQUI["cx"], synth.EQUI["cy"]]) + rngo.normal(0, 40.0, (5, 2))
        oos_all = np.tile(oos_u, (-(-B // uniq), 1))[:B]
        ctx.jacobians_instate(B); ctx.mh_gate(R_VIS, MH_THRESH, MH_MULT, MIN_INL, B, want=False); ctx.stack(R_VIS, B)
        rows = ctx.oos_project(oos_all, 3.5 ** 2)         # uploads the list once; the timed steps project it resident
        assert (rows == 7 * args.oos).all()
    frame = args.level == "G" and args.propagate_samples > 0
    if frame:
        # a sensor at rest: the accelerometer reads -g in the body frame, so the nominal state stays where the scene
        # was generated while P is propagated (Qimu / Qmodel of cfg/tumvi_cam0.json's order of magnitude)
        from xivo_amd.lib import imu_dtype
        p_all, _, _ = ctx.get_scene()
        Rsb = p_all["Rsb"].reshape(B, 3, 3).transpose(0, 2, 1)
        imu = np.zeros((B, args.propagate_samples), dtype=imu_dtype)
        imu["accel"] = (np.transpose(Rsb, (0, 2, 1)) @ np.array([0.0, 0.0, 9.8]))[:, None, :]
        imu["dt"] = 0.0025
        Qimu = np.diag(np.repeat([1e-6, 1e-4, 1e-10, 1e-10], 3)); Qmodel = np.eye(23) * 1e-10
        grav = np.array([0.0, 0.0, -9.8])

    def step():
        if frame:
            ctx.propagate(imu, Qimu, Qmodel, grav, method=args.integrator, stepsize=0.002)
            ctx.filter_update(R_VIS, MH_THRESH, MH_MULT, MIN_INL, not args.no_gating, B)
            ctx.absorb_error(B)
        elif oos_on:
            ctx.jacobians_instate(B)
            ctx.mh_gate(R_VIS, MH_THRESH, MH_MULT, MIN_INL, B, want=False)
            ctx.stack(R_VIS, B)
            ctx.oos_project((B, args.oos), 3.5 ** 2, want_rows=False)
            if not args.no_compression:      # measurement compression (estimator.h:399-402): 140 OOS rows -> 54
                ctx.compress_oos(1.5, B, want_rows=False)
            ctx.update_joseph(B)
        elif args.level == "G" and args.ransac:
            ctx.jacobians_instate(B)
            ctx.mh_gate(R_VIS, MH_THRESH, MH_MULT, MIN_INL, B, want=False)
            ctx.one_point_ransac(R_VIS, 2.0, 5.89, B=B, want=False)       # 1pt_RANSAC_thresh of cfg/pcw.json is 1.5 px
            ctx.stack(R_VIS, B)
            ctx.update_joseph(B)
        elif args.level == "G":
            ctx.filter_update(R_VIS, MH_THRESH, MH_MULT, MIN_INL, not args.no_gating, B)
        elif args.no_gating:
            hand_over()
            ctx.update_joseph(B)
        else:
            hand_over()       # H changes every frame: the dense -> row-pair compression is part of the step
            ctx.update_dense_gated(F, R_VIS, MH_THRESH, MH_MULT, MIN_INL, B)

    def barrier():
        ctx.sync()
        if dist is not None:      # ctx.sync() above is the device synchronisation (the path runs on the context's own stream)
            dist.barrier()
            ctx.sync()

    from xivo_amd.shard import max_over_ranks, gather_over_ranks

    def timed(nwarm):
        """W untimed steps, then exactly K steps between barrier + device sync on both sides."""
        for _ in range(nwarm):
            step()
        barrier()
        ctx.profile_reset()
        t0 = time.perf_counter()
        ctx.timer_begin()
        for _ in range(args.steps):
            step()
        gpu_ms_ = ctx.timer_end()
        barrier()
        dt_rank_ = time.perf_counter() - t0
        return dt_rank_, gpu_ms_, (ctx.profile_get() if (flags & FLAG_PROFILE) else {})

    # ---- headline: the library default - every product of the update in fp64
    dt_rank, gpu_ms, prof = timed(args.warmup)
    dt = max_over_ranks(dist, dt_rank)
    per_rank = gather_over_ranks(dist, B * args.steps / dt_rank)
    status = ctx.get_status(check=False)
    sparse_path = ctx.last_path() == 1
    # the state the timed loop left behind against the oracle doing the same number of updates in a row (every rank, its
    # own filters)
    parity_last = None
    if args.level == "S" and not args.no_parity_check and not args.no_last_step_parity:
        parity_last = parity_last_step(ctx, args.warmup + args.steps, B, uniq, F, P, H, inn, dR,
                                       (R_VIS, MH_THRESH, MH_MULT, MIN_INL), args.no_gating,
                                       tol_P=args.tol_P, tol_dx=args.tol_dx_last)

    # ---- second figure (opt-in library mode, NOT the headline): XIVO_HIP_FLAG_SYMMETRIC_FORM - P+ = P - W^T W with
    # W = L^-1 (H P), forward substitution only; equal to the Joseph form for the optimal gain, checked below
    symm = None
    if not args.no_mixed and sparse_path and args.level == "S":
        from xivo_amd.lib import FLAG_SYMMETRIC_FORM
        ctx.set_flags(base_flags | FLAG_SYMMETRIC_FORM)
        ctx.restore_P()
        dt_s, _, prof_s = timed(min(args.warmup, 2))
        dt_s = max_over_ranks(dist, dt_s)
        symm = {"value": world * B * args.steps / dt_s, "ms_per_step": dt_s / args.steps * 1e3,
                "stage_ms_per_step": {k: v["ms"] / args.steps for k, v in prof_s.items() if v["launches"]},
                "what": "XIVO_HIP_FLAG_SYMMETRIC_FORM: S = L L^T, W = L^-1 (H P) (forward substitution only), dx = W^T L^-1 inn, "
                        "P+ = P - W^T W - the value of the Joseph expression for the optimal gain, all fp64; opt-in, the "
                        "reference codes the Joseph form (= `value`)"}
        if not args.no_parity_check:
            symm["parity_check"] = parity_check(ctx, step, B, uniq, F, P, H, inn, dR, (R_VIS, MH_THRESH, MH_MULT, MIN_INL), args.no_gating, args.tol_P)
        ctx.set_flags(flags)

    # ---- parity at the benchmarked batch: one step from the initial P, filters spread over the launch
    # (first / last, both sides of an XCD group of 8, mid batch) against the oracle - checker only, outside the timing
    parity = None
    if args.level == "S" and not args.no_parity_check:      # every rank checks its own GPU's results
        parity = parity_check(ctx, step, B, uniq, F, P, H, inn, dR, (R_VIS, MH_THRESH, MH_MULT, MIN_INL), args.no_gating, args.tol_P)
    elif args.level == "G" and not args.no_parity_check and frame:
        parity = parity_frame(ctx, B, uniq, P, (poses, groups, feats), imu, Qimu, Qmodel, grav, args.integrator,
                              (R_VIS, MH_THRESH, MH_MULT, MIN_INL), args.no_gating, args.tol_P)
    elif args.level == "G" and not args.no_parity_check:
        # (--ransac too: OnePointRANSAC restores the state and covariance it experimented on - src/update.cpp:375-387 - so the
        #  final update starts from the initial covariance, on the rows that survived)
        parity = parity_glevel(ctx, step, B, uniq, P, args.tol_P)
    per_rank_parity = gather_objects(dist, {"rank": rank, "device": device,
                                            "ok": None if parity is None else all(q["ok"] for q in (parity, parity_last, (symm or {}).get("parity_check")) if q is not None),
                                            "rel_fro_P_max": None if parity is None else parity["rel_fro_P_max"],
                                            "rel_dx_max": None if parity is None else parity["rel_dx_max"],
                                            "last_step_rel_fro_P_max": None if parity_last is None else parity_last["rel_fro_P_max"]})
    per_rank_aff = gather_objects(dist, {k: affinity.get(k) for k in ("numa_node", "n_cpus", "bound", "omp_threads", "error")
                                         if affinity.get(k) is not None})

    failed = [q for q in (parity, parity_last, (symm or {}).get("parity_check")) if q is not None and not q["ok"]]
    if any(pr.get("ok") is False for pr in per_rank_parity):    # (the same list on every rank: all of them leave together)
        # every rank has delivered its result to the gather above, so nobody is left waiting in a collective
        sys.stderr.write(f"bench parity check FAILED (rank {rank}): {failed} per_rank={per_rank_parity}\n")
        ctx.close()
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(3)

    peak_meas = ctx.bench_mfma_peak() if rank == 0 else None

    if rank == 0:
        updates = world * B * args.steps
        value = updates / dt
        lib = load_library()
        import ctypes as C

        def norm(k):   # rocprofv3 prints template arguments with spaces; the block-list kernel has one instantiation per size class
            k = k.replace(" ", "")
            return "gemm_sym_f64_kernel" if k.startswith("gemm_sym_f64_kernel") else k

        def pmc_entry(table, label):
            """The PMC record of the kernel the library names `label`: the library prints the template arguments that select
            the instantiation (trsm_lds_f64_kernel<10,4>), rocprofv3 prints all of them, defaults included
            (trsm_lds_f64_kernel<10, 4, false, 16, 1>) - round 5's line lost its counter fields to that. Exact name first,
            then the unique record of the same kernel whose argument list starts with the library's."""
            label = norm(label.split("+")[0])          # ("+gate": the gate folded into the factorisation - same kernel name)
            if label in table:
                return table[label]
            base, _, args_ = label.partition("<")
            want = args_.rstrip(">").split(",") if args_ else []
            hits = []
            for k, v in table.items():
                if not isinstance(v, dict) or k.startswith("_"):
                    continue
                kb, _, ka = k.partition("<")
                have = ka.rstrip(">").split(",") if ka else []
                if kb == base and have[:len(want)] == want:
                    hits.append((len(have), k, v))
            if not hits:
                return None
            hits.sort()
            # several instantiations share the prefix (a GATE / CHOL twin): the one that ran most in the profiled command
            return max(hits, key=lambda h: h[2].get("launches_profiled", 0))[2]

        # group stages by the kernel instantiation the library reports for them (= the rocprofv3 kernel name)
        groups = {}
        for name, st in prof.items():
            if st["launches"] == 0:
                continue
            kname = norm(st["kernel"] or name)
            g = groups.setdefault(kname, {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0, "stages": []})
            g["ms"] += st["ms"]; g["launches"] += st["launches"]
            g["flops"] += st["flops_per_launch"] * st["launches"]; g["bytes"] += st["bytes_per_launch"] * st["launches"]
            g["stages"].append(name)
        pmc = {}
        try:
            import re
            # the newest committed PMC summary whose profiled command ran THIS workload (state dim / features / batch / level /
            # flags): r06_pmc_summary.json for the headline, r06tumvi_..., r06cfg2_... for the child rows
            def cmd_key(cmd):
                def opt(name, default):
                    m = re.search(name + r"\s+(\S+)", cmd)
                    return m.group(1) if m else default
                return (opt("--state-dim", "250"), opt("--features", "80"), opt("--batch", str(DEFAULT_BATCH)), opt("--level", "S"),
                        opt("--flags", "0"), "--calib" in cmd, opt("--oos", "0"), opt("--propagate-samples", "0"))
            mine = cmd_key(" ".join(sys.argv[1:]) + f" --batch {args.batch} --state-dim {args.state_dim} --features {args.features}")
            cands = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"r\d+\w*_pmc_summary\.json", f)),
                           key=lambda f: (int(re.match(r"r(\d+)", f).group(1)), f))
            for f in reversed(cands):
                rec = json.load(open(os.path.join(ROOT, "profiles", f)))
                if cmd_key(str((rec.get("_notes") or {}).get("command", ""))) == mine:
                    pmc = {norm(k): v for k, v in rec.items()}
                    pmc["_file"] = "profiles/" + f
                    break
        except Exception:
            pmc = {}
        # filters per launch of the profiled command (per-launch counters only compare at the same batch)
        m_ = __import__("re").search(r"--batch\s+(\d+)", str((pmc.get("_notes") or {}).get("command", "")))
        pmc_batch = int(m_.group(1)) if m_ else DEFAULT_BATCH
        roofline = None
        if groups:
            dom = max(groups, key=lambda k: groups[k]["ms"])
            g = groups[dom]
            # both roofs for the dominant kernel; the one it sits closer to is reported as "bound":
            #   mfma: algorithmic flops of its launches / time vs the fp64 matrix peak
            #   hbm : algorithmic bytes (inputs once + outputs once, reported by the library) / time vs ~8 TB/s
            tflops = g["flops"] / (g["ms"] * 1e-3) / 1e12
            gbs = g["bytes"] / (g["ms"] * 1e-3) / 1e9
            frac_mfma, frac_hbm = tflops / FP64_MFMA_PEAK_TFLOPS, gbs / HBM_PEAK_GBS
            if frac_hbm >= frac_mfma:
                bound, achieved, peak, unit = "hbm", gbs, HBM_PEAK_GBS, "GB/s"
            else:
                bound, achieved, peak, unit = "mfma", tflops, FP64_MFMA_PEAK_TFLOPS, "TFLOP/s"
            roofline = {"bound": bound, "kernel": dom, "stages": g["stages"],
                        "achieved": achieved, "peak": peak, "unit": unit,
                        "frac": achieved / peak,
                        "frac_mfma": frac_mfma, "frac_hbm": frac_hbm,
                        "algorithmic_tflops": tflops, "algorithmic_gbs": gbs,
                        "avg_launch_ms": g["ms"] / g["launches"], "launches": g["launches"],
                        # HBM bytes per launch of this kernel from the committed rocprofv3 PMC passes of the
                        # same command (FETCH_SIZE doubled per MI355X_MICROARCH.md + WRITE_SIZE); null if absent
                        "traffic": (lambda e: (e.get("hbm_read_bytes_per_launch", 0) + e.get("hbm_write_bytes_per_launch", 0))
                                    if e and args.batch == pmc_batch else None)(pmc_entry(pmc, dom)),
                        "traffic_source": pmc.get("_file"),
                        # from the same PMC passes: flops the MFMA pipe really executed per launch (symmetry and
                        # K(HP) - P skip work the reference's as-coded count includes) and pipe busy %
                        "executed_mfma_flops_per_launch": (pmc_entry(pmc, dom) or {}).get("executed_mfma_f64_flops_per_launch"),
                        "mfma_busy_pct_pmc": (pmc_entry(pmc, dom) or {}).get("mfma_busy_pct"),
                        "mfma_peak_measured_tflops": peak_meas,
                        # every stage's algorithmic bytes (intermediates included: they round-trip HBM between kernels)
                        "pipeline_algorithmic_gbs": sum(v["bytes_per_launch"] * v["launches"] for v in prof.values())
                                                    / (gpu_ms * 1e-3) / 1e9 if gpu_ms > 0 else None,
                        "pipeline_f_alg_tflops": f_alg(N, M) * value / world / 1e12,
                        "pipeline_frac": f_alg(N, M) * value / world / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                        "pipeline_frac_note": "F_alg = the reference's AS-CODED dense flop count (BASELINE.md 2: 4N^3 + 8MN^2 + 4M^2N + M^3/3) "
                                              "times updates/s over the fp64 matrix peak. It can exceed 1: the device path does not execute "
                                              "that work (H's 18 of N columns, symmetric halves, no KH - I and no N^3 product); `frac` above "
                                              "is the dominant kernel's own algorithmic flops (true N, M - not the padded sizes) over its "
                                              "measured time"}
        out = {
            "metric": "EKF updates/sec (state dim 250, 80 feats) @1 GPU; % MFMA roofline",
            "value": value, "unit": "updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"whole frame: Propagate ({args.propagate_samples} IMU samples, {args.integrator}) + AbsorbError + "
                                    if frame else "") +
                                   (f"{args.oos} OOS features (null-space projected, 7 rows each" +
                                    ("" if args.no_compression else ", QR-compressed to 54 rows") + ") + " if oos_on else "") +
                                   ("feature-level: Jacobians + " if args.level == "G" else "") +
                                   ("OnePointRANSAC + " if (args.level == "G" and args.ransac) else "") +
                                   ("UpdateJosephForm only" if args.no_gating else "MH gating + UpdateJosephForm") +
                                   f": state dim {N}, {F} features (M={M}), XIVO row sparsity, P/H/inn/R resident in HBM",
                       "filters_per_gpu": B, "global_batch": world * B, "parallelism": f"replicas x{world} (no collective)",
                       "pipeline": ("sparse-H (row-pair compressed H, whitened in-solve Joseph form" +
                                    ("; OOS block dense: mixed stacking)" if oos_on else ")")) if sparse_path else "dense as-coded",
                       "hand_over": ("dense H/inn/diagR (column-major, resident in HBM) -> row-pair compressed rows, "
                                     "every step, inside the timed region (stage stack_H)") if args.level == "S" else
                                    "Jacobians -> compressed rows on device every step",
                       "precision": {"value": "library default = all fp64: storage, every product, factorisation, solve"},
                       "route": ctx.last_route(),
                       "gpu_event_ms_per_step": gpu_ms / args.steps,
                       "not_spd_filters": int((status != 0).sum())},
            "value_symmetric_form": symm["value"] if symm else None,
            "symmetric_form": symm,
            "per_rank_updates_per_s": per_rank,
            "per_rank_min_max": [min(per_rank), max(per_rank)],
            "parity_check": parity,
            "parity_check_last_timed_step": parity_last,
            "per_rank_parity": per_rank_parity,
            "per_rank_affinity": per_rank_aff,
            "roofline": roofline,
            "stage_ms_per_step": {k: v["ms"] / args.steps for k, v in prof.items() if v["launches"]},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, F)
            out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        headline_default = (world == 1 and not args.sub and args.level == "S" and args.state_dim == 250 and args.features == 80
                            and args.flags == 0 and not args.no_gating)
        if headline_default and not args.no_dropin:
            try:
                out["dropin"] = dropin_block()
            except Exception as e:   # never let a secondary block break the bench line
                out["dropin"] = {"error": repr(e)[:300]}
        if headline_default and not args.no_configs:
            ctx.close()              # the child runs need the HBM this context holds
            out["configs"], out["configs_full"] = configs_block()
        if args.sub:
            print(json.dumps(out))           # the parent run reads the whole record and prints its own compact row
        else:
            path = args.full_out or os.path.join(ROOT, "gpurun_out", "bench_full.json")
            try:
                os.makedirs(os.path.dirname(path), exist_ok=True)
                with open(path, "w") as f:
                    json.dump(out, f, indent=1)
                out["full_record"] = os.path.relpath(path, ROOT)
            except OSError:
                pass
            print(json.dumps(compact_line(out), separators=(",", ":")))
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
