"""pyxivo-shaped Python surface over the MI355X path.

`Estimator` has the constructor and the methods of the reference's pybind module that an image-free client uses
(pybind11/pyxivo.cpp:332-398; scripts/pyxivo_pcw.py:133-163 is such a client): `InertialMeas`, `VisualMeasPointCloud`,
`InitWithSimDepths`, the state accessors (`gsb`, `gbc`, `gsc`, `Vsb`, `bg`, `ba`, `Rg`, `P`, `Pstate`, ...) and the
in-state feature / group listings (src/estimator_accessors.cpp). One `Estimator` is one filter on the device (a context
of batch 1); for many sequences at once use `xivo_amd.sequence.run_pcw`, which drives the same calls batched.

Every numeric step is the C ABI (`xivo_amd/sequence.HipBackend`); the life cycle is the simplified one documented in
`xivo_amd/sequence.py`. Image input (`VisualMeas`), the tracker-only modes, loop closure and the viewer are outside the
path this repository implements and raise NotImplementedError.
"""
import json
import re

import numpy as np

from . import lib as L
from . import sequence
from .pcw import so3_exp

_CAM_MODELS = {"pinhole": 0, "atan": 1, "fov": 1, "radtan": 2, "equidistant": 3}   # XIVO_CAM_* of include/xivo_hip.h


def load_json_with_comments(path):
    """cfg/*.json of the reference carry // comments (scripts/utils.py cleanup_and_load_json)"""
    with open(path) as f:
        text = f.read()
    text = re.sub(r"//[^\n]*", "", text)
    return json.loads(text)


def config_from_cfg(cfg):
    """The entries of a reference estimator cfg (cfg/pcw.json, cfg/tumvi_cam0.json schema) that the path reads ->
    SequenceConfig. Missing keys keep the reference defaults."""
    c = sequence.SequenceConfig()
    X = cfg.get("X", {})
    if "Wbc" in X:
        w = np.asarray(X["Wbc"], dtype=float)
        c.Wbc = w if w.size == 3 else None
        if w.size == 9:     # a rotation matrix instead of a rotation vector (scripts/pyxivo_pcw.py:47-50)
            from .pcw import so3_log
            c.Wbc = so3_log(w.reshape(3, 3))
    if "Tbc" in X:
        c.Tbc = np.asarray(X["Tbc"], dtype=float).reshape(3)
    c.X0 = {k: np.asarray(X.get(k, [0, 0, 0]), dtype=float).reshape(-1)[:3] for k in ("Wsb", "Tsb", "Vsb", "bg", "ba", "Wsg")}
    if "gravity" in cfg:
        c.gravity = np.asarray(cfg["gravity"], dtype=float)
    for k in c.P0:
        if k in cfg.get("P", {}):
            v = np.asarray(cfg["P"][k], dtype=float).reshape(-1)
            c.P0[k] = v[:3] if (k == "Tbc" and v.size == 3) else float(v[0])      # "Tbc": scalar or 3-vector (estimator.cpp:266-271)
    for k in c.Qmodel:
        if k in cfg.get("Qmodel", {}):
            c.Qmodel[k] = float(cfg["Qmodel"][k])
    for k in c.Qimu:
        if k in cfg.get("Qimu", {}):
            v = np.asarray(cfg["Qimu"][k], dtype=float).reshape(-1)
            c.Qimu[k] = v if v.size == 3 else float(v[0])
    c.integration_method = cfg.get("integration_method", c.integration_method)
    if c.integration_method in cfg and "stepsize" in cfg[c.integration_method]:
        c.stepsize = float(cfg[c.integration_method]["stepsize"])
    c.visual_meas_std = float(cfg.get("visual_meas_std", c.visual_meas_std))
    c.MH_thresh = float(cfg.get("MH_thresh", c.MH_thresh))
    c.MH_adjust_factor = float(cfg.get("MH_adjust_factor", c.MH_adjust_factor))
    c.min_inliers = int(cfg.get("min_inliers", c.min_inliers))
    c.use_MH_gating = bool(cfg.get("use_MH_gating", c.use_MH_gating))
    c.use_1pt_RANSAC = bool(cfg.get("use_1pt_RANSAC", c.use_1pt_RANSAC))
    c.ransac_thresh = float(cfg.get("1pt_RANSAC_thresh", c.ransac_thresh))
    c.ransac_Chi2 = float(cfg.get("1pt_RANSAC_Chi2", c.ransac_Chi2))
    for k in ("initial_std_x", "initial_std_y", "initial_std_z", "min_depth", "max_depth"):
        if k in cfg:
            setattr(c, k, float(cfg[k]))
    cam = cfg.get("camera_cfg")
    if cam:
        model = _CAM_MODELS[cam["model"]]
        d = []
        if cam["model"] == "equidistant":
            d = list(cam["k0123"])
        elif cam["model"] == "radtan":
            d = [cam["p1"], cam["p2"]] + list(cam["k012"])
        elif cam["model"] in ("atan", "fov"):
            d = [cam["w"]]
        c.cam = dict(model=model, rows=int(cam["rows"]), cols=int(cam["cols"]), fx=float(cam["fx"]), fy=float(cam["fy"]),
                     cx=float(cam["cx"]), cy=float(cam["cy"]), d=d)
    return c


class Estimator:
    """pyxivo.Estimator(cfg, viewer_cfg, name, tracker_only) (pybind11/pyxivo.cpp:30-60)."""

    def __init__(self, cfg, viewer_cfg="", name="", tracker_only=False, device=0):
        if tracker_only:
            raise NotImplementedError("tracker-only mode is outside the EKF update path")
        if isinstance(cfg, sequence.SequenceConfig):
            self.cfg = cfg
        else:
            self.cfg = config_from_cfg(load_json_with_comments(cfg) if isinstance(cfg, str) else dict(cfg))
        if self.cfg.cam["model"] != 0:
            raise NotImplementedError("point-cloud input initialises features with a pinhole un-projection only")
        self.name = name
        c = self.cfg
        X0 = c.X0 or {k: np.zeros(3) for k in ("Wsb", "Tsb", "Vsb", "bg", "ba", "Wsg")}
        pose = np.zeros(1, dtype=L.pose_dtype)
        pose[0]["Rsb"] = so3_exp(X0["Wsb"]).T.reshape(-1); pose[0]["Tsb"] = X0["Tsb"]; pose[0]["Vsb"] = X0["Vsb"]
        pose[0]["bg"] = X0["bg"]; pose[0]["ba"] = X0["ba"]
        pose[0]["Rbc"] = so3_exp(c.Wbc).T.reshape(-1); pose[0]["Tbc"] = c.Tbc
        wsg = np.array([X0["Wsg"][0], X0["Wsg"][1], 0.0])
        pose[0]["Rsg"] = so3_exp(wsg).T.reshape(-1)
        self._be = sequence.HipBackend(c, 1, pose, c.P_init()[None], device=device)
        self._runner = sequence.SequenceRunner(self._be, c, 1)
        self._feeder = None
        self._ts = 0
        self._vision = False
        self._sim_depths = False
        self._group_ids = {}          # group slot -> (id, slot generation); ids handed out like Group::counter_ (src/group.h)
        self._next_group_id = 0

    # ---- messages --------------------------------------------------------------------------------------------
    def InertialMeas(self, ts, wx, wy, wz, ax, ay, az):
        """Estimator::InertialMeas (src/estimator.cpp:443-470): queue one IMU message (ts in ns)"""
        t = ts * 1e-9
        g, a = np.array([[wx, wy, wz]], dtype=float), np.array([[ax, ay, az]], dtype=float)
        if self._feeder is None:
            self._feeder = sequence.ImuFeeder(1, t, g, a)      # the first message only initialises last_gyro_/last_accel_
        else:
            self._feeder.imu(t, g, a)
        self._ts = int(ts)

    def InitWithSimDepths(self):
        self._sim_depths = True

    def VisualMeasPointCloud(self, ts, feature_ids, xp_and_depths):
        """Estimator::VisualMeasPointCloud (src/estimator.cpp:1133-1180): tracks given as ids + (x, y, depth) rows.
        Features enter with the given depth (InitWithSimDepths; without it the reference starts from `initial_z`
        and the depth sub-filter, which this driver does not run)."""
        if not self._sim_depths:
            raise NotImplementedError("call InitWithSimDepths(): depth-less initialisation needs the sub-filter warm-up")
        ids = np.asarray(feature_ids, dtype=np.int64).reshape(-1)
        meas = np.asarray(xp_and_depths, dtype=float).reshape(-1, 3)
        imu = None
        if self._feeder is not None:
            self._feeder.visual(ts * 1e-9)
            imu = self._feeder.take()
        self._runner.frame(imu, [(ids, meas)])
        bk = self._runner.books[0]
        for g, r in enumerate(bk.group_refs):                          # ids for the groups created this frame
            if r >= 0 and self._group_ids.get(g, (None, -1))[1] != bk.group_gen[g]:
                self._group_ids[g] = (self._next_group_id, bk.group_gen[g]); self._next_group_id += 1
            elif r < 0:
                self._group_ids.pop(g, None)
        self._ts = int(ts)
        self._vision = True

    def VisualMeas(self, *a):
        raise NotImplementedError("image input needs the tracker (outside the EKF update path)")

    VisualMeasTrackerOnly = VisualMeasPointCloudTrackerOnly = CloseLoop = VisualMeas

    def Visualize(self):
        pass

    # ---- state ------------------------------------------------------------------------------------------------
    def _flush(self):
        """IMU messages queued since the last camera frame are integrated before the state is read"""
        if self._feeder is not None:
            imu = self._feeder.take()
            if imu is not None:
                self._be.propagate(imu)

    def _scene(self):
        self._flush()
        p, g, f = self._be.scene()
        return p[0], g[0], f[0]

    @staticmethod
    def _R(v):
        return np.asarray(v).reshape(3, 3).T

    def now(self):
        return self._ts

    def VisionInitialized(self):
        return self._vision

    MeasurementUpdateInitialized = VisionInitialized

    def gsb(self):
        p, _, _ = self._scene()
        return np.hstack([self._R(p["Rsb"]), p["Tsb"].reshape(3, 1)])

    def gbc(self):
        p, _, _ = self._scene()
        return np.hstack([self._R(p["Rbc"]), p["Tbc"].reshape(3, 1)])

    def gsc(self):
        p, _, _ = self._scene()
        Rsb, Rbc = self._R(p["Rsb"]), self._R(p["Rbc"])
        return np.hstack([Rsb @ Rbc, (Rsb @ p["Tbc"] + p["Tsb"]).reshape(3, 1)])

    def Vsb(self):
        return self._scene()[0]["Vsb"].copy()

    def bg(self):
        return self._scene()[0]["bg"].copy()

    def ba(self):
        return self._scene()[0]["ba"].copy()

    def Rg(self):
        return self._R(self._scene()[0]["Rsg"])

    def td(self):
        return 0.0

    def Ca(self):
        return np.eye(3)

    def Cg(self):
        return np.eye(3)

    def P(self):
        self._flush()
        return self._be.covariance()[0]

    def Pstate(self):
        return self.P()[:9, :9]          # src/estimator.h:159

    def CameraIntrinsics(self):
        c = self.cfg.cam
        return np.array([c["fx"], c["fy"], c["cx"], c["cy"]])

    def CameraDistortionType(self):
        return "pinhole"

    # ---- in-state features / groups (src/estimator_accessors.cpp), ordered by state slot ------------------------
    def _slots(self):
        bk = self._runner.books[0]
        return [j for j in range(self.cfg.n_features) if bk.feat_id[j] >= 0]

    def num_instate_features(self):
        return len(self._slots())

    def num_instate_groups(self):
        return sum(1 for r in self._runner.books[0].group_refs if r >= 0)

    def num_mh_rejected(self):
        return self._runner.n_rejected

    def num_oneptransac_rejected(self):
        return 0

    def InstateFeatureIDs(self, n_output=None):
        bk = self._runner.books[0]
        return np.array([bk.feat_id[j] for j in self._slots()], dtype=np.int64)[:n_output]

    def InstateFeatureSinds(self, n_output=None):
        return np.array(self._slots(), dtype=np.int64)[:n_output]

    def InstateFeatureRefGroups(self, n_output=None):
        bk = self._runner.books[0]
        return np.array([self._group_ids[bk.feat_ref[j]][0] for j in self._slots()], dtype=np.int64)[:n_output]

    def InstateFeaturexc(self, n_output=None):
        """(x/z, y/z, log z) in the anchor camera frame, Feature::x (src/feature.h:258-262)"""
        _, _, f = self._scene()
        return np.array([f["x"][j] for j in self._slots()]).reshape(-1, 3)[:n_output]

    def InstateFeatureXc(self, n_output=None):
        x = self.InstateFeaturexc(n_output)
        z = 1.0 / x[:, 2] if getattr(self.cfg, "use_invdepth", False) else np.exp(x[:, 2])   # Feature::z (src/feature.cpp:120-126)
        return np.stack([x[:, 0] * z, x[:, 1] * z, z], axis=1)

    def InstateFeaturePositions(self, n_output=None):
        """Feature::Xs (src/feature.cpp:107-112): anchor pose applied to the point in the anchor camera frame"""
        p, g, f = self._scene()
        Rbc, Tbc = self._R(p["Rbc"]), p["Tbc"]
        out = []
        for j in self._slots():
            x = f["x"][j]; z = 1.0 / x[2] if getattr(self.cfg, "use_invdepth", False) else np.exp(x[2])
            Xc = np.array([x[0] * z, x[1] * z, z])
            r = int(f["ref_sind"][j])
            out.append(self._R(g["Rsb"][r]) @ (Rbc @ Xc + Tbc) + g["Tsb"][r])
        return np.array(out).reshape(-1, 3)[:n_output]

    def InstateFeatureMeas(self, n_output=None):
        _, _, f = self._scene()
        return np.array([f["xp"][j] for j in self._slots()]).reshape(-1, 2)[:n_output]

    def InstateFeatureCovs(self, n_output=None):
        """3x3 covariance of each feature's state block (rows of 9, row-major)"""
        P = self.P(); fb = 23 + 6 * self.cfg.n_groups
        return np.array([P[fb + 3 * j:fb + 3 * j + 3, fb + 3 * j:fb + 3 * j + 3].reshape(-1) for j in self._slots()]).reshape(-1, 9)[:n_output]

    def _gslots(self):
        return [g for g, r in enumerate(self._runner.books[0].group_refs) if r >= 0]

    def InstateGroupIDs(self):
        return np.array([self._group_ids[g][0] for g in self._gslots()], dtype=np.int64)

    def InstateGroupSinds(self):
        return np.array(self._gslots(), dtype=np.int64)

    def InstateGroupPoses(self):
        """rows (qx, qy, qz, qw, Tx, Ty, Tz), src/estimator_accessors.cpp:589-618"""
        _, g, _ = self._scene()
        rows = []
        for s in self._gslots():
            R = self._R(g["Rsb"][s])
            w = np.sqrt(max(0.0, 1.0 + np.trace(R))) / 2.0
            q = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (4.0 * w) if w > 1e-8 else np.zeros(3)
            rows.append(np.concatenate([q, [w], g["Tsb"][s]]))
        return np.array(rows).reshape(-1, 7)

    def InstateGroupCovs(self):
        """6x6 covariance blocks of the in-state groups, stacked (src/estimator_accessors.cpp:620-640)"""
        P = self.P()
        return np.vstack([P[23 + 6 * s:29 + 6 * s, 23 + 6 * s:29 + 6 * s] for s in self._gslots()]) if self._gslots() else np.zeros((0, 6))

    def close(self):
        self._be.close()
