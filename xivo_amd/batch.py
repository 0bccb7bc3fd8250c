"""ctypes view of xivo::hip::BatchEstimator (xivo_amd/host/batch_estimator.h): the C++ host side of the sequence loop -
Estimator::InertialMeas / VisualMeasPointCloud semantics for B filters resident on one GPU context."""
import ctypes as C
import os

import numpy as np

from . import lib as L

_HERE = os.path.dirname(os.path.abspath(__file__))
_HOST = None

# struct xivo_batch_cfg (xivo_amd/host/batch_estimator.cpp)
batch_cfg_dtype = np.dtype([
    ("n_groups", "i4"), ("n_features", "i4"),
    ("cam_model", "i4"), ("cam_rows", "i4"), ("cam_cols", "i4"), ("_pad0", "i4"),
    ("fx", "f8"), ("fy", "f8"), ("cx", "f8"), ("cy", "f8"), ("d", "f8", 5),
    ("visual_meas_std", "f8"), ("MH_thresh", "f8"), ("MH_adjust_factor", "f8"),
    ("min_inliers", "i4"), ("min_new_features", "i4"), ("fix_group_block", "i4"), ("disable_MH_gating", "i4"),
    ("initial_std_x", "f8"), ("initial_std_y", "f8"), ("initial_std_z", "f8"), ("min_depth", "f8"), ("max_depth", "f8"),
    ("prop", L.prop_opts_dtype),
    ("use_1pt_RANSAC", "i4"), ("use_invdepth", "i4"), ("ransac_thresh", "f8"), ("ransac_Chi2", "f8")])


def load_host_library():
    """libxivo_host.so (C++ adapter + batch estimator); raises if it has not been built - there is no fallback."""
    global _HOST
    if _HOST is None:
        L.load_library()      # the C ABI first (the host library links against it)
        path = os.path.join(_HERE, "libxivo_host.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path + " not built: python -c 'import __graft_entry__ as g; g.build()'")
        _HOST = C.CDLL(path)
        _HOST.xivo_batch_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        _HOST.xivo_batch_destroy.argtypes = [C.c_void_p]; _HOST.xivo_batch_destroy.restype = None
        _HOST.xivo_batch_imu.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        _HOST.xivo_batch_visual.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _HOST.xivo_batch_poses.argtypes = [C.c_void_p, C.c_void_p]
        _HOST.xivo_batch_book.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _HOST.xivo_batch_stats.argtypes = [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_double)]
        _HOST.xivo_batch_stats.restype = None
        _HOST.xivo_batch_ctx.argtypes = [C.c_void_p]; _HOST.xivo_batch_ctx.restype = C.c_void_p
    return _HOST


class BatchEstimator:
    def __init__(self, cfg, B, poses0, P0, device=0):
        """cfg: xivo_amd.sequence.SequenceConfig; poses0: [B] pose_dtype; P0: [N, N] shared initial covariance"""
        self.host = load_host_library()
        self.cfg, self.B, self.F = cfg, B, cfg.n_features
        c = np.zeros(1, dtype=batch_cfg_dtype)
        c["n_groups"], c["n_features"] = cfg.n_groups, cfg.n_features
        cam = cfg.cam
        c["cam_model"], c["cam_rows"], c["cam_cols"] = cam["model"], cam["rows"], cam["cols"]
        c["fx"], c["fy"], c["cx"], c["cy"] = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
        c["visual_meas_std"], c["MH_thresh"], c["MH_adjust_factor"] = cfg.visual_meas_std, cfg.MH_thresh, cfg.MH_adjust_factor
        c["min_inliers"], c["min_new_features"], c["fix_group_block"] = cfg.min_inliers, cfg.min_new_features, int(cfg.fix_group_block)
        c["disable_MH_gating"] = 0 if getattr(cfg, "use_MH_gating", True) else 1
        c["use_1pt_RANSAC"] = 1 if getattr(cfg, "use_1pt_RANSAC", False) else 0
        c["use_invdepth"] = 1 if getattr(cfg, "use_invdepth", False) else 0
        c["ransac_thresh"], c["ransac_Chi2"] = getattr(cfg, "ransac_thresh", 5.0), getattr(cfg, "ransac_Chi2", 5.89)
        c["initial_std_x"], c["initial_std_y"], c["initial_std_z"] = cfg.initial_std_x, cfg.initial_std_y, cfg.initial_std_z
        c["min_depth"], c["max_depth"] = cfg.min_depth, cfg.max_depth
        c["prop"]["Qimu"] = cfg.Qimu_matrix().T.reshape(-1); c["prop"]["Qmodel"] = cfg.Qmodel_matrix().T.reshape(-1)
        c["prop"]["g"] = cfg.gravity; c["prop"]["method"] = 0 if cfg.integration_method == "RK4" else 1
        c["prop"]["stepsize"] = cfg.stepsize
        poses0 = np.ascontiguousarray(poses0, dtype=L.pose_dtype)
        P0 = np.asfortranarray(np.asarray(P0, dtype=np.float64))
        h = C.c_void_p()
        if self.host.xivo_batch_create(c.ctypes.data, B, device, poses0.ctypes.data, P0.ctypes.data, C.byref(h)) != 0:
            raise RuntimeError("xivo_batch_create failed")
        self.h = h

    def close(self):
        if self.h:
            self.host.xivo_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def InertialMeas(self, t, gyro, accel):
        gyro = np.ascontiguousarray(gyro, dtype=np.float64).reshape(self.B, 3)
        accel = np.ascontiguousarray(accel, dtype=np.float64).reshape(self.B, 3)
        if self.host.xivo_batch_imu(self.h, float(t), gyro.ctypes.data, accel.ctypes.data) != 0:
            raise RuntimeError("InertialMeas failed")

    def VisualMeasPointCloud(self, t, tracks):
        """tracks: per filter (ids [n], xp_and_depths [n x 3]) -> inlier mask [B x F]"""
        off = np.zeros(self.B + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(tr[0]) for tr in tracks])
        ids = np.ascontiguousarray(np.concatenate([np.asarray(tr[0], dtype=np.int64) for tr in tracks])) if off[-1] else np.zeros(1, dtype=np.int64)
        meas = np.ascontiguousarray(np.concatenate([np.asarray(tr[1], dtype=np.float64).reshape(-1, 3) for tr in tracks])) if off[-1] else np.zeros((1, 3))
        mask = np.zeros((self.B, self.F), dtype=np.uint8)
        if self.host.xivo_batch_visual(self.h, float(t), off.ctypes.data, ids.ctypes.data, meas.ctypes.data, mask.ctypes.data) != 0:
            raise RuntimeError("VisualMeasPointCloud failed")
        return mask.astype(bool)

    def poses(self):
        p = np.zeros(self.B, dtype=L.pose_dtype)
        if self.host.xivo_batch_poses(self.h, p.ctypes.data) != 0:
            raise RuntimeError("poses failed")
        return p

    def gsb(self):
        p = self.poses()
        return p["Rsb"].reshape(-1, 3, 3).transpose(0, 2, 1).copy(), p["Tsb"].copy()

    def book(self, b):
        fid = np.zeros(self.F, dtype=np.int64); fref = np.zeros(self.F, dtype=np.int32)
        gref = np.zeros(self.cfg.n_groups, dtype=np.int32)
        self.host.xivo_batch_book(self.h, b, fid.ctypes.data, fref.ctypes.data, gref.ctypes.data)
        return fid, fref, gref

    def stats(self):
        nu, nr, hs = C.c_long(), C.c_long(), C.c_double()
        self.host.xivo_batch_stats(self.h, C.byref(nu), C.byref(nr), C.byref(hs))
        return dict(updates=nu.value, mh_rejected=nr.value, host_seconds=hs.value)
