"""Image-free sequence driver over the resident C-ABI path (BASELINE configs 1 / 5 surrogate).

Runs B independent visual-inertial sequences at once on one GPU context: for every camera frame the IMU samples since the
last frame go down in one `xivo_hip_propagate` call, the frame's state edits of all filters in one `xivo_hip_edit_batch`
call and its pixels in one `xivo_hip_set_pixels` call, then `xivo_hip_filter_update` (Jacobians -> MH gating -> stacking -> Joseph update) and `xivo_hip_absorb_error`.
State, covariance, groups and features never leave the device; the host keeps only the slot book-keeping
(`gsel_` / `fsel_` of src/estimator.h:496-503) and decides who enters and leaves the state.

What is mirrored from the reference and what is simplified:
  * IMU bookkeeping (`ImuFeeder`): Estimator::Propagate's last/curr/slope handling, src/estimator.cpp:548-575.
  * message order: IMU before camera at equal timestamps (scripts/pyxivo_pcw.py:121-129).
  * per frame (Estimator::UpdateStep order, src/manager.cpp:30-110): tracker-dropped in-state features are removed
    (ProcessTracks, :152-169), the filter update runs on the tracked in-state features, MH-rejected features are
    removed (src/update.cpp:105-113), groups that lost all their features are discarded (DiscardAffectedGroups),
    then new features enter (SelectAndAddNewFeatures).
  * SIMPLIFIED: a new feature enters the state in the frame it is first seen, with the simulator's depth
    (`InitWithSimDepths`, scripts/pyxivo_pcw.py:139-140, src/manager.cpp:588) and the configured initial std
    (src/estimator.cpp:349-353) - no depth sub-filter warm-up, no gauge features, no reference-group switching, no
    group lifetime cap; its anchor is a group created from the current pose (AddGroupToState).
The numerics of every step are the device path; this file holds no arithmetic of the filter itself.
"""
import numpy as np

from . import lib as L
from .pcw import so3_exp, so3_log


class ImuFeeder:
    """last_/curr_/slope_ bookkeeping of Estimator::Propagate (src/estimator.cpp:548-575) for B filters: turns raw
    (t, gyro, accel) messages and camera timestamps into the xivo_imu_in records of xivo_hip_propagate."""

    def __init__(self, B, t0, gyro0, accel0):
        self.t = np.full(B, float(t0))
        self.last_gyro = np.array(gyro0, dtype=float).reshape(B, 3).copy()
        self.last_accel = np.array(accel0, dtype=float).reshape(B, 3).copy()
        self.slope_gyro = np.zeros((B, 3)); self.slope_accel = np.zeros((B, 3))
        self.pending = []

    def imu(self, t, gyro, accel):
        """one IMU message per filter at time t (visual_meas == false branch, :558-567)"""
        dt = t - self.t
        if np.all(dt <= 0):
            # Estimator::Propagate returns early on dt == 0 and leaves last_ / slope_ untouched (src/estimator.cpp:550-555);
            # a message from the past (dt < 0) is skipped the same way, with a warning
            if np.any(dt < 0):
                import warnings
                warnings.warn("IMU message older than the filter time: skipped")
            return
        # per-filter clocks: a filter whose dt <= 0 returns early like the reference's estimator (its last_ / slope_ / time
        # stay untouched and its record is a zero-length step, which xivo_hip_propagate integrates as the identity)
        go = dt > 0
        rec = np.zeros(self.t.shape[0], dtype=L.imu_dtype)
        gyro = np.broadcast_to(np.asarray(gyro, dtype=float), self.last_gyro.shape)
        accel = np.broadcast_to(np.asarray(accel, dtype=float), self.last_accel.shape)
        safe = np.where(go, dt, 1.0)[:, None]
        self.slope_gyro = np.where(go[:, None], (gyro - self.last_gyro) / safe, self.slope_gyro)
        self.slope_accel = np.where(go[:, None], (accel - self.last_accel) / safe, self.slope_accel)
        rec["gyro"], rec["accel"] = self.last_gyro, self.last_accel
        rec["slope_gyro"], rec["slope_accel"], rec["dt"] = self.slope_gyro, self.slope_accel, np.where(go, dt, 0.0)
        self.last_gyro = np.where(go[:, None], gyro, self.last_gyro)
        self.last_accel = np.where(go[:, None], accel, self.last_accel)
        self.t = np.where(go, t, self.t)
        self.pending.append(rec)

    def visual(self, t):
        """camera message at time t (visual_meas == true branch, :568-575); dt == 0 propagates nothing (:550-555)"""
        dt = np.maximum(t - self.t, 0.0)      # (a filter already at or past t propagates nothing: zero-length record)
        if np.all(dt == 0):
            return
        rec = np.zeros(self.t.shape[0], dtype=L.imu_dtype)
        rec["gyro"], rec["accel"] = self.last_gyro, self.last_accel
        rec["slope_gyro"], rec["slope_accel"], rec["dt"] = self.slope_gyro, self.slope_accel, dt
        self.last_gyro = self.last_gyro + self.slope_gyro * dt[:, None]
        self.last_accel = self.last_accel + self.slope_accel * dt[:, None]
        self.t = np.maximum(self.t, t)
        self.pending.append(rec)

    def take(self):
        """-> [B x K] records since the last take (None if there are none)"""
        if not self.pending:
            return None
        out = np.stack(self.pending, axis=1)
        self.pending = []
        return out


class SequenceConfig:
    """The numbers of cfg/pcw.json the path reads (reference defaults), sizes of the TUM-VI build (src/core.h:95-105)."""

    def __init__(self, **kw):
        self.n_groups, self.n_features = 15, 30
        self.cam = dict(model=L_CAM_PINHOLE, rows=480, cols=640, fx=275.0, fy=275.0, cx=320.0, cy=240.0, d=[])
        self.Wbc, self.Tbc = np.array([-1.57079633, 0.0, 0.0]), np.zeros(3)
        self.gravity = np.array([0.0, 0.0, -9.8])
        self.X0 = None                  # cfg "X" (initial nominal state) when the filter does not start from ground truth
        self.P0 = dict(Wsb=0.001, Tsb=0.001, Vsb=0.5, bg=1e-10, ba=1e-10, Wbc=1e-10, Tbc=1e-10, Wsg=1e-10)
        self.Qmodel = dict(Wsb=0.01, Wbc=0.0, Wsg=0.0)
        self.Qimu = dict(gyro=5e-3, accel=5e-2, gyro_bias=0.0, accel_bias=0.0)
        self.integration_method, self.stepsize = "PrinceDormand", 0.002
        self.visual_meas_std = 1.0
        self.MH_thresh, self.MH_adjust_factor, self.min_inliers = 5.991, 1.1, 5
        self.use_MH_gating = True       # cfg use_MH_gating (src/estimator.cpp:364)
        self.use_1pt_RANSAC = False     # cfg use_1pt_RANSAC (off in cfg/pcw.json, cfg/tumvi_cam0.json)
        self.ransac_thresh, self.ransac_Chi2 = 5.0, 5.89       # 1pt_RANSAC_thresh / 1pt_RANSAC_Chi2 defaults (estimator.cpp:132-134)
        self.initial_std_x = self.initial_std_y = 1.0      # pixels, divided by the focal length (estimator.cpp:351-352)
        self.initial_std_z = 0.10
        self.min_depth, self.max_depth = 0.05, 10.0
        # Feature::FillJacobianBlock as coded drops the group-rotation block (src/feature.cpp:675-676); a
        # sequence tracks markedly worse with it, so the driver asks for the evidently intended row.
        # False = bit-faithful to the reference's stacking.
        self.fix_group_block = True
        self.min_new_features = 3       # open a new group only when at least this many feature slots are free
        # the reference's USE_INVDEPTH build (src/CMakeLists.txt:10): features are (X/Z, Y/Z, 1/Z); initial_std_z is then an
        # inverse-depth standard deviation
        self.use_invdepth = False
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError("unknown option " + k)
            setattr(self, k, v)

    @property
    def N(self):
        return 23 + 6 * self.n_groups + 3 * self.n_features

    def P_init(self):
        """Estimator ctor, src/estimator.cpp:257-304: P_ = identity (kFullSize - unused group / feature slots keep a
        unit diagonal until a remove op zeroes them), the motion blocks scaled by cfg "P" - which are STANDARD
        DEVIATIONS: the whole matrix is then squared (`P_ *= P_`, :304). "Tbc" may be a scalar or a 3-vector (:266-271)."""
        P = np.eye(self.N)
        d = np.ones(23)
        p = self.P0
        d[0:3], d[3:6], d[6:9], d[9:12], d[12:15] = p["Wsb"], p["Tsb"], p["Vsb"], p["bg"], p["ba"]
        d[15:18], d[18:21], d[21:23] = p["Wbc"], np.asarray(p["Tbc"], dtype=float).reshape(-1)[:3], p["Wsg"]
        P[:23, :23] = np.diag(d * d)
        return P

    def focal_length(self):
        """Camera::GetFocalLength() = 0.5 sqrt(fx^2 + fy^2) (src/camera_manager.cpp:56) - what the initial feature
        std in pixels is divided by (src/estimator.cpp:351-352); NOT fx: 0.707 fx for a square pixel."""
        return 0.5 * float(np.hypot(self.cam["fx"], self.cam["fy"]))

    def Qmodel_matrix(self):
        """src/estimator.cpp:313-318: only the Wsb, Wbc and Wsg blocks are read from cfg "Qmodel", then squared"""
        q = self.Qmodel
        d = np.zeros(23)
        d[0:3], d[15:18], d[21:23] = q["Wsb"], q["Wbc"], q["Wsg"]
        return np.diag(d * d)

    def Qimu_matrix(self):
        """src/estimator.cpp:325-330 (std -> variance)"""
        q = self.Qimu
        d = np.concatenate([np.full(3, q["gyro"]), np.full(3, q["accel"]), np.full(3, q["gyro_bias"]),
                            np.full(3, q["accel_bias"])])
        return np.diag(d * d)


L_CAM_PINHOLE = 0


class HipBackend:
    """The product path: every numeric step is a C-ABI call on the resident state (fails loudly without the
    library / a GPU - there is no host fallback)."""

    def __init__(self, cfg, B, poses0, P0, device=0, flags=0):
        self.cfg, self.B, self.F = cfg, B, cfg.n_features
        if cfg.fix_group_block:
            flags |= L.FLAG_FIX_GROUP_BLOCK
        if getattr(cfg, "use_invdepth", False):
            flags |= L.FLAG_INVDEPTH
        self.ctx = L.Context(cfg.N, 2 * cfg.n_features, B, device=device, flags=flags)
        self.ctx.set_layout(cfg.N, 23, cfg.n_groups, 23 + 6 * cfg.n_groups, cfg.n_features, cfg.cam)
        self.ctx.upload_P(P0)
        groups = np.zeros((B, cfg.n_groups), dtype=L.group_dtype)
        groups["Rsb"][:] = np.eye(3).reshape(-1)
        feats = np.zeros((B, self.F), dtype=L.feat_dtype)
        feats["sind"] = -1
        self.ctx.set_scene(poses0, groups, feats)
        self.Qimu, self.Qmodel = cfg.Qimu_matrix(), cfg.Qmodel_matrix()

    def propagate(self, imu):
        self.ctx.propagate(imu, self.Qimu, self.Qmodel, self.cfg.gravity,
                           method="RK4" if self.cfg.integration_method == "RK4" else "PD", stepsize=self.cfg.stepsize)

    def edit(self, ops):
        self.ctx.edit_batch(self.F, ops)

    def set_pixels(self, xp):
        self.ctx.set_pixels(xp)

    def update(self):
        c = self.cfg
        R = c.visual_meas_std ** 2
        if c.use_1pt_RANSAC:
            # Estimator::OutlierRejection with use_1pt_RANSAC (src/manager.cpp:629-650): MH gating, then OnePointRANSAC on
            # its inliers; the update runs on what RANSAC keeps. (No gauge group / previous-frame group list in this
            # simplified life cycle: temporary reference group every time, every slot absorbed.)
            self.ctx.jacobians_instate()
            self.ctx.mh_gate(R, c.MH_thresh, c.MH_adjust_factor, c.min_inliers if c.use_MH_gating else 1 << 30, want=False)
            self.ctx.one_point_ransac(R, c.ransac_thresh, c.ransac_Chi2, want=False)
            self.ctx.stack(R)
            self.ctx.update_joseph()
        else:
            self.ctx.filter_update(R, c.MH_thresh, c.MH_adjust_factor, c.min_inliers, bool(c.use_MH_gating))
        mask, _ = self.ctx.get_gate(self.F)
        # a filter whose S was not positive definite keeps its prior P and absorbs nothing (device side); surfaced here
        self.last_status = self.ctx.get_status(check=False)
        self.n_not_spd = getattr(self, "n_not_spd", 0) + int((self.last_status != 0).sum())
        self.ctx.absorb_error()
        return mask

    def sync(self):
        self.ctx.sync()

    def poses(self):
        p, _, _ = self.ctx.get_scene()
        return p["Rsb"].reshape(-1, 3, 3).transpose(0, 2, 1).copy(), p["Tsb"].copy()

    def covariance(self):
        return self.ctx.download_P()

    def scene(self):
        return self.ctx.get_scene()

    def close(self):
        self.ctx.close()


class _Book:
    """slot book-keeping of one filter: gsel_ / fsel_ and who sits where (src/estimator.h:496-503)"""

    def __init__(self, n_groups, n_features):
        self.group_refs = [-1] * n_groups        # -1: free slot, else number of in-state features anchored there
        self.group_gen = [0] * n_groups          # how many groups have lived in the slot (tells a re-used slot apart)
        self.feat_id = [-1] * n_features         # track id held by feature slot j (-1: free)
        self.feat_ref = [-1] * n_features
        self.id2slot = {}

    def drop_feature(self, j):
        del self.id2slot[self.feat_id[j]]
        self.group_refs[self.feat_ref[j]] -= 1
        self.feat_id[j] = -1; self.feat_ref[j] = -1

    def n_instate(self):
        return len(self.id2slot)


def _op(b, kind, i0=0, i1=0, i2=0, v=()):
    o = np.zeros((), dtype=L.edit_dtype)
    o["b"], o["kind"], o["i0"], o["i1"], o["i2"] = b, kind, i0, i1, i2
    if len(v):
        o["v"][:len(v)] = v
    return o


class SequenceRunner:
    """Drives B sequences frame by frame through a backend (`HipBackend`; tests also run the same decisions against
    an oracle backend). `frame()` consumes the pending IMU records and one camera frame per filter."""

    def __init__(self, backend, cfg, B):
        self.be, self.cfg, self.B = backend, cfg, B
        self.books = [_Book(cfg.n_groups, cfg.n_features) for _ in range(B)]
        self.n_updates = 0
        self.n_rejected = 0
        self.timers = None       # set to {} to accumulate wall seconds per phase (adds a device sync per phase)

    def _tick(self, name, t0):
        if self.timers is None:
            return 0.0
        import time
        if name not in ("host_pre", "host_post") and hasattr(self.be, "sync"):
            self.be.sync()
        t1 = time.perf_counter()
        self.timers[name] = self.timers.get(name, 0.0) + (t1 - t0)
        return t1

    def _discard_empty_groups(self, b, ops):
        bk = self.books[b]
        for g, r in enumerate(bk.group_refs):
            if r == 0:
                ops.append(_op(b, L.EDIT_REMOVE_GROUP, g))
                bk.group_refs[g] = -1

    def frame(self, imu, tracks):
        """imu: [B x K] xivo_imu_in records or None; tracks: per filter (ids [n], xp_and_depths [n x 3])."""
        import time
        cfg, be = self.cfg, self.be
        t0 = time.perf_counter()
        if imu is not None:
            be.propagate(imu)
        t0 = self._tick("propagate", t0) or t0
        # --- before the update: tracker-dropped features leave, tracked ones get their new pixel
        ops = []
        xp = np.full((self.B, cfg.n_features, 2), np.nan)     # the frame's pixels of the tracked in-state features
        for b in range(self.B):
            bk = self.books[b]
            ids, meas = tracks[b]
            pos = {int(i): k for k, i in enumerate(ids)}
            for j in range(cfg.n_features):
                fid = bk.feat_id[j]
                if fid < 0:
                    continue
                if fid in pos:
                    xp[b, j] = meas[pos[fid], :2]
                else:
                    ops.append(_op(b, L.EDIT_REMOVE_FEATURE, j))
                    bk.drop_feature(j)
            self._discard_empty_groups(b, ops)
        ops = np.array(ops, dtype=L.edit_dtype)
        t0 = self._tick("host_pre", t0) or t0
        be.edit(ops)
        be.set_pixels(xp)
        t0 = self._tick("edit", t0) or t0
        # --- measurement update on the tracked in-state features (every filter, ragged)
        mask = be.update()
        t0 = self._tick("update", t0) or t0
        self.n_updates += sum(1 for bk in self.books if bk.n_instate() > 0)
        # --- after the update: MH-rejected features leave, then new features enter with a new group
        ops = []
        fx, fy, cx, cy = cfg.cam["fx"], cfg.cam["fy"], cfg.cam["cx"], cfg.cam["cy"]
        fl = cfg.focal_length()
        std = np.array([cfg.initial_std_x / fl, cfg.initial_std_y / fl, cfg.initial_std_z])
        P3 = np.diag(std * std).T.reshape(-1)
        for b in range(self.B):
            bk = self.books[b]
            for j in range(cfg.n_features):
                if bk.feat_id[j] >= 0 and not mask[b, j]:
                    ops.append(_op(b, L.EDIT_REMOVE_FEATURE, j))
                    bk.drop_feature(j)
                    self.n_rejected += 1
            self._discard_empty_groups(b, ops)
            free = [j for j in range(cfg.n_features) if bk.feat_id[j] < 0]
            gfree = [g for g, r in enumerate(bk.group_refs) if r < 0]
            if not gfree or (len(free) < cfg.min_new_features and bk.n_instate() > 0):
                continue
            ids, meas = tracks[b]
            cand = [k for k in np.argsort(ids, kind="stable")
                    if int(ids[k]) not in bk.id2slot and cfg.min_depth < meas[k, 2] < cfg.max_depth]
            if not cand:
                continue
            g = gfree[0]
            ops.append(_op(b, L.EDIT_ADD_GROUP, g))
            bk.group_refs[g] = 0; bk.group_gen[g] += 1
            for j, k in zip(free, cand):
                x = [(meas[k, 0] - cx) / fx, (meas[k, 1] - cy) / fy, (1.0 / meas[k, 2] if getattr(cfg, "use_invdepth", False) else np.log(meas[k, 2]))]   # Feature::Initialize, feature.cpp:144-150
                ops.append(_op(b, L.EDIT_ADD_FEATURE, j, j, g, v=np.concatenate([x, meas[k, :2], P3])))
                bk.feat_id[j] = int(ids[k]); bk.feat_ref[j] = g; bk.id2slot[int(ids[k])] = j
                bk.group_refs[g] += 1
        ops = np.array(ops, dtype=L.edit_dtype)
        t0 = self._tick("host_post", t0) or t0
        be.edit(ops)
        self._tick("edit", t0)
        return mask


def initial_poses(cfg, sims, t0=0.0):
    """xivo_pose_in records at t0 from the ground truth of each simulator (the reference starts from cfg "X";
    velocity comes from the trajectory, scripts/imu_trajectories.py get_imu_sim init_Vsb)"""
    B = len(sims)
    poses = np.zeros(B, dtype=L.pose_dtype)
    Rbc = so3_exp(cfg.Wbc)
    for b, s in enumerate(sims):
        Rsb, Tsb = s.gsb(t0)
        poses[b]["Rsb"] = Rsb.T.reshape(-1); poses[b]["Tsb"] = Tsb
        poses[b]["Rbc"] = Rbc.T.reshape(-1); poses[b]["Tbc"] = cfg.Tbc
        poses[b]["Vsb"] = s.vel(t0)
        poses[b]["Rsg"] = np.eye(3).reshape(-1)
    return poses


def run_pcw(backend_factory, cfg, worlds, sims, total_time=4.0, imu_dt=0.0025, vision_dt=0.04, noise_vision_std=1.0,
            timers=None):
    """The loop of scripts/pyxivo_pcw.py:117-163 for B = len(sims) sequences at once.
    -> dict(ts [n] ns, Tsb [n x B x 3], Wsb [n x B x 3], gt_Tsb [n x B x 3], runner, backend)"""
    B = len(sims)
    K = np.array([[cfg.cam["fx"], 0, cfg.cam["cx"]], [0, cfg.cam["fy"], cfg.cam["cy"]], [0, 0, 1.0]])
    Rbc = so3_exp(cfg.Wbc)
    poses0 = initial_poses(cfg, sims)
    P0 = np.repeat(cfg.P_init()[None], B, axis=0)
    be = backend_factory(cfg, B, poses0, P0)
    runner = SequenceRunner(be, cfg, B)
    runner.timers = timers
    m0 = [s.meas(0.0) for s in sims]
    feeder = ImuFeeder(B, 0.0, [m[1] for m in m0], [m[0] for m in m0])
    n_imu = int(round(total_time / imu_dt)); every = int(round(vision_dt / imu_dt))
    ts, est_T, est_W, gt_T = [], [], [], []
    for k in range(n_imu):
        t = k * imu_dt
        if k > 0:
            m = [s.meas(t) for s in sims]
            feeder.imu(t, np.array([x[1] for x in m]), np.array([x[0] for x in m]))
        if k % every == 0:
            feeder.visual(t)
            tracks = []
            for b in range(B):
                Rsb, Tsb = sims[b].gsb(t)
                tracks.append(worlds[b].generate_measurements(Rsb @ Rbc, Rsb @ cfg.Tbc + Tsb, K, cfg.cam["cols"],
                                                              cfg.cam["rows"], noise_vision_std))
            runner.frame(feeder.take(), tracks)
            R, T = be.poses()
            ts.append(int(round(t * 1e9))); est_T.append(T); est_W.append(np.array([so3_log(r) for r in R]))
            gt_T.append(np.array([s.gsb(t)[1] for s in sims]))
    return dict(ts=np.array(ts), Tsb=np.array(est_T), Wsb=np.array(est_W), gt_Tsb=np.array(gt_T), runner=runner,
                backend=be)


def run_pcw_cpp(cfg, worlds, sims, total_time=4.0, imu_dt=0.0025, vision_dt=0.04, noise_vision_std=1.0, device=0):
    """run_pcw with the C++ host side: the same messages go to xivo::hip::BatchEstimator (xivo_amd/host/batch_estimator.h)
    through its InertialMeas / VisualMeasPointCloud entry points instead of ImuFeeder + SequenceRunner.
    -> dict(ts, Tsb, Wsb, gt_Tsb, estimator)"""
    from .batch import BatchEstimator
    B = len(sims)
    K = np.array([[cfg.cam["fx"], 0, cfg.cam["cx"]], [0, cfg.cam["fy"], cfg.cam["cy"]], [0, 0, 1.0]])
    Rbc = so3_exp(cfg.Wbc)
    est = BatchEstimator(cfg, B, initial_poses(cfg, sims), cfg.P_init(), device=device)
    n_imu = int(round(total_time / imu_dt)); every = int(round(vision_dt / imu_dt))
    ts, est_T, est_W, gt_T = [], [], [], []
    for k in range(n_imu):
        t = k * imu_dt
        m = [s.meas(t) for s in sims]
        est.InertialMeas(t, np.array([x[1] for x in m]), np.array([x[0] for x in m]))
        if k % every == 0:
            tracks = []
            for b in range(B):
                Rsb, Tsb = sims[b].gsb(t)
                tracks.append(worlds[b].generate_measurements(Rsb @ Rbc, Rsb @ cfg.Tbc + Tsb, K, cfg.cam["cols"],
                                                              cfg.cam["rows"], noise_vision_std))
            est.VisualMeasPointCloud(t, tracks)
            R, T = est.gsb()
            ts.append(int(round(t * 1e9))); est_T.append(T); est_W.append(np.array([so3_log(r) for r in R]))
            gt_T.append(np.array([s.gsb(t)[1] for s in sims]))
    return dict(ts=np.array(ts), Tsb=np.array(est_T), Wsb=np.array(est_W), gt_Tsb=np.array(gt_T), estimator=est)


def run_pcw_batch(cfg, B, total_time=2.0, imu_dt=0.0025, vision_dt=0.04, noise_vision_std=1.0, npts=1000, seed=0, device=0,
                  timers=None):
    """Thousands of sequences end to end: the vectorised simulators of xivo_amd/pcw.py (BatchTrajectorySim, BatchPCW) feed
    xivo::hip::BatchEstimator message by message. -> dict(ts, Tsb [n x B x 3], gt_Tsb, estimator)"""
    import time
    from .batch import BatchEstimator
    from .pcw import BatchPCW, BatchTrajectorySim
    motion = ["lissajous" if b % 2 == 0 else "trefoil" for b in range(B)]
    rate = 0.08 + 0.04 * (np.arange(B) % 7) / 7
    sim = BatchTrajectorySim(motion, rate, seed=seed + 1)
    world = BatchPCW(B, npts=npts, seed=seed)
    K = np.array([[cfg.cam["fx"], 0, cfg.cam["cx"]], [0, cfg.cam["fy"], cfg.cam["cy"]], [0, 0, 1.0]])
    Rbc = so3_exp(cfg.Wbc)
    poses = np.zeros(B, dtype=L.pose_dtype)
    R0, T0 = sim.gsb(0.0)
    poses["Rsb"] = R0.transpose(0, 2, 1).reshape(B, 9); poses["Tsb"] = T0; poses["Vsb"] = sim.vel(0.0)
    poses["Rbc"] = Rbc.T.reshape(-1); poses["Tbc"] = cfg.Tbc; poses["Rsg"] = np.eye(3).reshape(-1)
    est = BatchEstimator(cfg, B, poses, cfg.P_init(), device=device)
    host = est.host
    n_imu = int(round(total_time / imu_dt)); every = int(round(vision_dt / imu_dt))
    ts, est_T, gt_T = [], [], []
    tm = timers if timers is not None else {}
    for k in range(n_imu):
        t = k * imu_dt
        t0 = time.perf_counter()
        accel, gyro = sim.meas(t)
        tm["sim"] = tm.get("sim", 0.0) + time.perf_counter() - t0
        est.InertialMeas(t, gyro, accel)
        if k % every == 0:
            t0 = time.perf_counter()
            Rsb, Tsb = sim.gsb(t)
            off, ids, meas = world.generate(Rsb @ Rbc, np.einsum("bij,j->bi", Rsb, cfg.Tbc) + Tsb, K, cfg.cam["cols"],
                                            cfg.cam["rows"], noise_vision_std)
            t1 = time.perf_counter()
            tm["sim"] = tm.get("sim", 0.0) + t1 - t0
            mask = np.zeros((B, cfg.n_features), dtype=np.uint8)
            if host.xivo_batch_visual(est.h, float(t), off.ctypes.data, ids.ctypes.data, meas.ctypes.data, mask.ctypes.data) != 0:
                raise RuntimeError("VisualMeasPointCloud failed")
            tm["frame"] = tm.get("frame", 0.0) + time.perf_counter() - t1
            ts.append(int(round(t * 1e9))); est_T.append(est.poses()["Tsb"].copy()); gt_T.append(Tsb)
    return dict(ts=np.array(ts), Tsb=np.array(est_T), gt_Tsb=np.array(gt_T), estimator=est)
