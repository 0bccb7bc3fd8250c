"""ctypes binding of include/xivo_hip.h (test / bench plumbing only)."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
OOS_MAX_OBS = 16

FLAG_FIX_GROUP_BLOCK = 1
FLAG_PROFILE = 2
FLAG_DENSE_H = 64
FLAG_SYMMETRIC_FORM = 256
FLAG_STANDALONE_TAIL = 512
FLAG_NO_LDLT_FALLBACK = 4096
FLAG_FP32_WHITENED = 16384  # N > 256 / M > 176: V^T, Y^T as float, P - V^T Y on the fp32 MFMA
FLAG_THROUGHPUT_ROUTE = 8192    # every batch size on the kernels sized for thousands of filters (default: <= 64 filters take the latency route)
FLAG_INVDEPTH = 32768           # USE_INVDEPTH build: features are (X/Z, Y/Z, 1/Z) (src/feature.cpp:98-105)
OOS_WHOLE_BUFFER = 1             # xivo_hip_oos_project_ex options
FLAG_MULTI_KERNEL = 65536       # keep shapes the one-kernel update holds (fused_update.hip) on the multi-kernel pipeline
CAM_PINHOLE, CAM_ATAN, CAM_RADTAN, CAM_EQUI = 0, 1, 2, 3


class XivoHipError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"xivo_hip status {status}: {msg}")
        self.status = status


class Layout(C.Structure):
    _fields_ = [("N", C.c_int), ("group_begin", C.c_int), ("n_groups", C.c_int),
                ("feature_begin", C.c_int), ("n_features", C.c_int)]


class Cam(C.Structure):
    _fields_ = [("model", C.c_int), ("rows", C.c_int), ("cols", C.c_int),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("d", C.c_double * 5)]


# numpy dtypes mirroring the C structs (all naturally aligned, no padding surprises:
# sizes are asserted against ctypes below)
pose_dtype = np.dtype([("Rsb", "f8", 9), ("Tsb", "f8", 3), ("Rbc", "f8", 9), ("Tbc", "f8", 3),
                       ("Vsb", "f8", 3), ("bg", "f8", 3), ("ba", "f8", 3), ("Rsg", "f8", 9)])
group_dtype = np.dtype([("Rsb", "f8", 9), ("Tsb", "f8", 3)])
feat_dtype = np.dtype([("x", "f8", 3), ("xp", "f8", 2), ("ref_sind", "i4"), ("sind", "i4")])
oos_dtype = np.dtype([("Xs", "f8", 3), ("n_obs", "i4"), ("group_sind", "i4", OOS_MAX_OBS),
                      ("_pad", "i4"), ("xp", "f8", (OOS_MAX_OBS, 2))])


class _OosC(C.Structure):
    _fields_ = [("Xs", C.c_double * 3), ("n_obs", C.c_int), ("group_sind", C.c_int * OOS_MAX_OBS),
                ("xp", (C.c_double * 2) * OOS_MAX_OBS)]


assert oos_dtype.itemsize == C.sizeof(_OosC), (oos_dtype.itemsize, C.sizeof(_OosC))
lc_dtype = np.dtype([("feat", "i4"), ("group_sind", "i4"), ("xp", "f8", 2)])      # xivo_lc_match
assert lc_dtype.itemsize == 24
imu_dtype = np.dtype([("gyro", "f8", 3), ("accel", "f8", 3), ("slope_gyro", "f8", 3), ("slope_accel", "f8", 3), ("dt", "f8")])
prop_opts_dtype = np.dtype([("Qimu", "f8", 144), ("Qmodel", "f8", 529), ("g", "f8", 3), ("method", "i4"), ("_pad", "i4"),
                            ("stepsize", "f8"), ("control_stepsize", "i4"), ("attempts", "i4"), ("tolerance", "f8"),
                            ("min_scale_factor", "f8"), ("max_scale_factor", "f8")])
assert imu_dtype.itemsize == 104 and prop_opts_dtype.itemsize == (144 + 529 + 3) * 8 + 16 + 32
subfilter_dtype = np.dtype([("x", "f8", 3), ("P", "f8", 9), ("xp", "f8", 2), ("outlier_counter", "f8"), ("score", "f8"),
                            ("ref_sind", "i4"), ("status", "i4"), ("init_counter", "i4"), ("candidate", "i4")])
subfilter_opts_dtype = np.dtype([("Rtri", "f8"), ("MH_thresh", "f8"), ("ready_steps", "i4"), ("_pad", "i4"),
                                 ("min_depth", "f8"), ("max_depth", "f8"), ("max_subfilter_outlier", "f8")])
# xivo_edit_op (include/xivo_hip.h): one resident-state edit of one filter
edit_dtype = np.dtype([("b", "i4"), ("kind", "i4"), ("i0", "i4"), ("i1", "i4"), ("i2", "i4"), ("reserved", "i4"),
                       ("v", "f8", 14)])
assert edit_dtype.itemsize == 136
EDIT_P_ZERO_RC, EDIT_P_COPY_RC, EDIT_P_SET_BLOCK3, EDIT_ADD_GROUP, EDIT_REMOVE_GROUP, EDIT_ADD_FEATURE, \
    EDIT_REMOVE_FEATURE, EDIT_SET_XP = range(8)
assert subfilter_dtype.itemsize == 144 and subfilter_opts_dtype.itemsize == 48
assert feat_dtype.itemsize == 48 and pose_dtype.itemsize == 336 and group_dtype.itemsize == 96


def lib_path():
    # XIVO_HIP_LIBRARY: A/B timing of experimental builds in one process pool (scripts only)
    return os.environ.get("XIVO_HIP_LIBRARY") or os.path.join(_HERE, "libxivo_hip.so")


_LIB = None

_SIGS = {
    "xivo_hip_create": [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint],
    "xivo_hip_sync": [C.c_void_p],
    "xivo_hip_set_flags": [C.c_void_p, C.c_uint],
    "xivo_hip_upload_P": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_int],
    "xivo_hip_download_P": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_int],
    "xivo_hip_snapshot_P": [C.c_void_p],
    "xivo_hip_restore_P": [C.c_void_p],
    "xivo_hip_p_zero_rc": [C.c_void_p, C.c_int, C.c_int, C.c_int],
    "xivo_hip_p_copy_rc": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int],
    "xivo_hip_p_set_block3": [C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "xivo_hip_p_diag": [C.c_void_p, C.c_int, C.c_void_p],
    "xivo_hip_set_measurements": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_int,
                                  C.c_void_p, C.c_long, C.c_void_p, C.c_long],
    "xivo_hip_set_measurements_device": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_int,
                                         C.c_void_p, C.c_long, C.c_void_p, C.c_long],
    "xivo_hip_update_joseph": [C.c_void_p, C.c_int],
    "xivo_hip_get_err": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_long],
    "xivo_hip_get_status": [C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "xivo_hip_mh_gate_dense": [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int,
                               C.c_void_p, C.c_void_p],
    "xivo_hip_update_dense_gated": [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int],
    "xivo_hip_get_gate": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    "xivo_hip_set_layout": [C.c_void_p, C.POINTER(Layout), C.POINTER(Cam)],
    "xivo_hip_set_scene": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    "xivo_hip_jacobians_instate": [C.c_void_p, C.c_int],
    "xivo_hip_get_jacobians": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    "xivo_hip_mh_gate": [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p,
                         C.c_void_p],
    "xivo_hip_stack": [C.c_void_p, C.c_int, C.c_double],
    "xivo_hip_oos_project": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_void_p],
    "xivo_hip_oos_project_ex": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_uint],
    "xivo_hip_close_loop_stack": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_double],
    "xivo_hip_compress_oos": [C.c_void_p, C.c_int, C.c_double, C.c_void_p],
    "xivo_hip_one_point_ransac": [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p],
    "xivo_hip_filter_update": [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int],
    "xivo_hip_last_path": [C.c_void_p],
    "xivo_hip_last_route": [C.c_void_p],
    "xivo_hip_stage_kernel": [C.c_void_p, C.c_int],
    "xivo_hip_stage_bytes": [C.c_void_p, C.c_int],
    "xivo_hip_absorb_error": [C.c_void_p, C.c_int],
    "xivo_hip_propagate": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    "xivo_hip_givens": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                        C.c_void_p],
    "xivo_hip_qr": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p],
    "xivo_hip_subfilter_update": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    "xivo_hip_candidate_order": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    "xivo_hip_edit_batch": [C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "xivo_hip_set_pixels": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p],
    "xivo_hip_get_scene": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    "xivo_hip_get_H": [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p],
    "xivo_hip_propagate_cov": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p],
    "xivo_hip_dev_alloc": [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)],
    "xivo_hip_dev_free": [C.c_void_p, C.c_void_p],
    "xivo_hip_dev_upload": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t],
    "xivo_hip_timer_begin": [C.c_void_p],
    "xivo_hip_timer_end": [C.c_void_p, C.POINTER(C.c_float)],
    "xivo_hip_profile_reset": [C.c_void_p],
    "xivo_hip_profile_get": [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_float),
                             C.POINTER(C.c_int), C.POINTER(C.c_double)],
    "xivo_hip_bench_mfma_peak": [C.c_void_p, C.POINTER(C.c_double)],
    "xivo_hip_get_ldlt_used": [C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "xivo_hip_update_joseph_host": [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_uint],
    "xivo_hip_set_calib": [C.c_void_p, C.c_void_p],
    "xivo_hip_set_calib_state": [C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "xivo_hip_get_jacobians_calib": [C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "xivo_hip_get_calib_state": [C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "xivo_hip_set_calib_gyro": [C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "xivo_hip_propagate_calib": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    "xivo_hip_selftest_fused_tiles": [C.c_int, C.c_void_p],
    "xivo_hip_selftest_host_compress": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
}
HOST_P_RESIDENT, HOST_KEEP_P = 1, 2


class CalibLayout(C.Structure):
    _fields_ = [("td", C.c_int), ("Cg", C.c_int), ("cam_begin", C.c_int), ("cam_dim", C.c_int)]


calib_dtype = np.dtype([("gyro", "f8", 3), ("Cg", "f8", 9), ("td", "f8"), ("Ca", "f8", 9), ("intr", "f8", 9)])
assert calib_dtype.itemsize == 248


def cam_intr(cam):
    """xivo_calib_in::intr of a camera dict: fx fy cx cy, then the distortion parameters in xivo_cam.d's order"""
    d = list(cam.get("d", [])) + [0.0] * 5
    return np.array([cam["fx"], cam["fy"], cam["cx"], cam["cy"]] + d[:5], dtype=np.float64)
# every symbol include/xivo_hip.h declares (tests check the library exports them all)
ALL_SYMBOLS = sorted(list(_SIGS) + ["xivo_hip_destroy", "xivo_hip_strerror", "xivo_hip_gemm_tile", "xivo_hip_device_count",
                                        "xivo_hip_device_numa_node", "xivo_hip_route_name"])


def load_library():
    """Load libxivo_hip.so; raises (no fallback) if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not built - run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the HIP extension is the product; there is no CPU fallback)")
    lib = C.CDLL(path)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.xivo_hip_device_count.argtypes = []
    lib.xivo_hip_device_count.restype = C.c_int
    lib.xivo_hip_device_numa_node.argtypes = [C.c_int]
    lib.xivo_hip_device_numa_node.restype = C.c_int
    lib.xivo_hip_destroy.argtypes = [C.c_void_p]
    lib.xivo_hip_destroy.restype = None
    lib.xivo_hip_strerror.argtypes = [C.c_int]
    lib.xivo_hip_strerror.restype = C.c_char_p
    lib.xivo_hip_route_name.argtypes = [C.c_int]
    lib.xivo_hip_route_name.restype = C.c_char_p
    lib.xivo_hip_gemm_tile.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.xivo_hip_gemm_tile.restype = None
    lib.xivo_hip_stage_kernel.restype = C.c_char_p
    lib.xivo_hip_stage_bytes.restype = C.c_double
    _LIB = lib
    return lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def candidate_order(feats, strict=False, score_type=0):
    """Criteria::CandidateComparison order (src/options.cpp:34-61) of a [nb, n] subfilter_dtype array: returns
    (order [nb, n] padded with -1, count [nb], score [nb, n] of `score_type`). Host arithmetic, no GPU needed."""
    lib = load_library()
    feats = np.ascontiguousarray(feats, dtype=subfilter_dtype)
    nb, n = feats.shape
    order = np.full((nb, n), -1, dtype=np.int32); cnt = np.zeros(nb, dtype=np.int32); score = np.zeros((nb, n))
    rc = lib.xivo_hip_candidate_order(_ptr(feats), nb, n, int(strict), int(score_type), _ptr(order), _ptr(cnt), _ptr(score))
    if rc != 0:
        raise XivoHipError(rc, lib.xivo_hip_strerror(rc).decode())
    return order, cnt, score


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """A batch of `batch` independent filters with state dim N on one GPU.

    Matrices cross this boundary as numpy arrays shaped [batch, cols, rows]
    C-contiguous == column-major [rows x cols] per filter (Eigen's layout); the
    helpers below take/return [batch, rows, cols] arrays and do the transposes.
    """

    def __init__(self, N, M_max, batch, device=0, flags=0):
        self.lib = load_library()
        self.N, self.M_max, self.batch = int(N), int(M_max), int(batch)
        h = C.c_void_p()
        self._check(self.lib.xivo_hip_create(C.byref(h), device, N, M_max, batch, flags))
        self.h = h
        self.flags = flags

    def _check(self, rc):
        if rc != 0:
            raise XivoHipError(rc, self.lib.xivo_hip_strerror(rc).decode())

    def close(self):
        if getattr(self, "h", None):
            for p in getattr(self, "_dev_bufs", []):
                self.lib.xivo_hip_dev_free(self.h, p)
            self._dev_bufs = []
            self.lib.xivo_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def sync(self):
        self._check(self.lib.xivo_hip_sync(self.h))

    def set_flags(self, flags):
        self._check(self.lib.xivo_hip_set_flags(self.h, flags))
        self.flags = flags

    # ---- P ---------------------------------------------------------------
    def upload_P(self, P, b0=0):
        P = np.asarray(P, dtype=np.float64)
        nb = P.shape[0]
        Pc = _f64(np.transpose(P, (0, 2, 1)))  # column-major per filter
        self._check(self.lib.xivo_hip_upload_P(self.h, b0, nb, _ptr(Pc), self.N * self.N, self.N))

    def download_P(self, b0=0, nb=None):
        nb = self.batch - b0 if nb is None else nb
        out = np.empty((nb, self.N, self.N), dtype=np.float64)
        self._check(self.lib.xivo_hip_download_P(self.h, b0, nb, _ptr(out), self.N * self.N, self.N))
        return np.transpose(out, (0, 2, 1)).copy()

    def snapshot_P(self):
        self._check(self.lib.xivo_hip_snapshot_P(self.h))

    def restore_P(self):
        self._check(self.lib.xivo_hip_restore_P(self.h))

    def p_zero_rc(self, b, off, length):
        self._check(self.lib.xivo_hip_p_zero_rc(self.h, b, off, length))

    def p_copy_rc(self, b, dst, src, length):
        self._check(self.lib.xivo_hip_p_copy_rc(self.h, b, dst, src, length))

    def p_set_block3(self, b, off, P3):
        P3c = _f64(np.asarray(P3).T)
        self._check(self.lib.xivo_hip_p_set_block3(self.h, b, off, _ptr(P3c)))

    def p_diag(self, b):
        out = np.empty(self.N)
        self._check(self.lib.xivo_hip_p_diag(self.h, b, _ptr(out)))
        return out

    # ---- S-level ------------------------------------------------------------
    def set_measurements(self, H, inn, diagR, b0=0):
        H = np.asarray(H, dtype=np.float64)
        nb, M, N = H.shape
        assert N == self.N
        Hc = _f64(np.transpose(H, (0, 2, 1)))
        inn = _f64(inn)
        dR = _f64(diagR)
        self._check(self.lib.xivo_hip_set_measurements(self.h, b0, nb, M, _ptr(Hc), M * N, M, _ptr(inn), M,
                                                       _ptr(dR), M))

    def set_measurements_device(self, dH, dinn, dR, M, nb, b0=0, strideH=None, ldh=None):
        """Hand-over of measurements that already live in device memory: dH / dinn / dR are device addresses
        (ints, e.g. torch tensor .data_ptr()) of nb column-major M x N matrices and M-vectors."""
        ldh = M if ldh is None else ldh
        strideH = ldh * self.N if strideH is None else strideH
        self._check(self.lib.xivo_hip_set_measurements_device(self.h, b0, nb, M, C.c_void_p(dH), strideH, ldh,
                                                              C.c_void_p(dinn), M, C.c_void_p(dR), M))

    def device_array(self, a, total=None):
        """Device copy of the numpy array `a` on this context's GPU (bench / tests: inputs that are already resident);
        total = number of leading-axis entries the buffer holds, `a` is repeated to fill it. Returns the address."""
        a = np.ascontiguousarray(a)
        total = a.shape[0] if total is None else total
        per = a.nbytes // a.shape[0]
        p = C.c_void_p()
        self._check(self.lib.xivo_hip_dev_alloc(self.h, per * total, C.byref(p)))
        self._dev_bufs = getattr(self, "_dev_bufs", []) + [p]
        n0 = min(a.shape[0], total)
        self._check(self.lib.xivo_hip_dev_upload(self.h, p, _ptr(a), per * n0, per * total))
        return p.value

    def update_joseph(self, B=None):
        self._check(self.lib.xivo_hip_update_joseph(self.h, self.batch if B is None else B))

    # ---- online-calibration builds (measurement side) ----------------------------
    def set_calib(self, td=-1, Cg=-1, cam_begin=0, cam_dim=0):
        """switch the td / Cg / bg / intrinsics Jacobian blocks on (td = Cg = -1 and cam_dim = 0: off)"""
        if td < 0 and Cg < 0 and cam_dim == 0:
            self._check(self.lib.xivo_hip_set_calib(self.h, None))
        else:
            cl = CalibLayout(td, Cg, cam_begin, cam_dim)
            self._check(self.lib.xivo_hip_set_calib(self.h, C.byref(cl)))

    def set_calib_state(self, calib, b0=0):
        calib = np.ascontiguousarray(calib, dtype=calib_dtype)
        self._check(self.lib.xivo_hip_set_calib_state(self.h, b0, calib.shape[0], _ptr(calib)))

    def set_calib_gyro(self, gyro, b0=0):
        gyro = np.ascontiguousarray(gyro, dtype=np.float64).reshape(-1, 3)
        self._check(self.lib.xivo_hip_set_calib_gyro(self.h, b0, gyro.shape[0], _ptr(gyro)))

    def get_calib_state(self, b0=0, nb=None):
        nb = self.batch - b0 if nb is None else nb
        out = np.zeros(nb, dtype=calib_dtype)
        self._check(self.lib.xivo_hip_get_calib_state(self.h, b0, nb, _ptr(out)))
        return out

    def get_jacobians_calib(self, b0=0, nb=None, F=None):
        nb = self.batch - b0 if nb is None else nb
        out = np.empty((nb, F, 2, 22))
        self._check(self.lib.xivo_hip_get_jacobians_calib(self.h, b0, nb, _ptr(out)))
        return out

    def update_joseph_host(self, H, inn, diagR, P_cm, b=0, mode=0, check=True):
        """The one-call plumbing entry (xivo_hip_update_joseph_host): H [M, N] row-major here (transposed to Eigen's
        column-major), P_cm a column-major N x N float64 array updated IN PLACE. Returns (err, rc)."""
        H = np.asarray(H, dtype=np.float64)
        M, N = H.shape
        assert N == self.N, (N, self.N)
        assert P_cm is None or (P_cm.dtype == np.float64 and P_cm.flags["F_CONTIGUOUS"] and P_cm.shape == (N, N))
        Hc = _f64(H.T)
        inn = _f64(inn); dR = _f64(diagR)
        err = np.empty(self.N)
        rc = self.lib.xivo_hip_update_joseph_host(self.h, b, M, _ptr(Hc), M, _ptr(inn), _ptr(dR),
                                                  None if P_cm is None else _ptr(P_cm), self.N, _ptr(err), mode)
        if check:
            self._check(rc)
        return err, rc

    def get_err(self, b0=0, nb=None):
        nb = self.batch - b0 if nb is None else nb
        out = np.empty((nb, self.N))
        self._check(self.lib.xivo_hip_get_err(self.h, b0, nb, _ptr(out), self.N))
        return out

    def get_status(self, b0=0, nb=None, check=True):
        nb = self.batch - b0 if nb is None else nb
        out = np.zeros(nb, dtype=np.int32)
        rc = self.lib.xivo_hip_get_status(self.h, b0, nb, _ptr(out))
        if check:
            self._check(rc)
        return out

    def get_ldlt_used(self, b0=0, nb=None):
        """1 for every filter whose last update ran the pivoted L D L^T fallback (S not positive definite)"""
        nb = self.batch - b0 if nb is None else nb
        out = np.zeros(nb, dtype=np.int32)
        self._check(self.lib.xivo_hip_get_ldlt_used(self.h, b0, nb, _ptr(out)))
        return out

    def mh_gate_dense(self, F, R, thresh, mult, min_inliers, B=None):
        B = self.batch if B is None else B
        mask = np.zeros((B, F), dtype=np.uint8)
        dist = np.zeros((B, F))
        self._check(self.lib.xivo_hip_mh_gate_dense(self.h, B, F, R, thresh, mult, min_inliers, _ptr(mask),
                                                    _ptr(dist)))
        return mask.astype(bool), dist

    def update_dense_gated(self, F, R, thresh, mult, min_inliers, B=None):
        self._check(self.lib.xivo_hip_update_dense_gated(self.h, self.batch if B is None else B, F, R, thresh, mult,
                                                         min_inliers))

    def get_gate(self, F, B=None):
        B = self.batch if B is None else B
        mask = np.zeros((B, F), dtype=np.uint8)
        dist = np.zeros((B, F))
        self._check(self.lib.xivo_hip_get_gate(self.h, B, F, _ptr(mask), _ptr(dist)))
        return mask.astype(bool), dist

    # ---- G-level ------------------------------------------------------------
    def set_layout(self, N, group_begin, n_groups, feature_begin, n_features, cam):
        lay = Layout(N, group_begin, n_groups, feature_begin, n_features)
        self.layout = lay
        c = Cam()
        c.model, c.rows, c.cols = cam["model"], cam.get("rows", 480), cam.get("cols", 640)
        c.fx, c.fy, c.cx, c.cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
        d = list(cam.get("d", [])) + [0.0] * 5
        for i in range(5):
            c.d[i] = d[i]
        self._check(self.lib.xivo_hip_set_layout(self.h, C.byref(lay), C.byref(c)))

    def set_scene(self, poses, groups, feats, b0=0):
        poses = np.ascontiguousarray(poses, dtype=pose_dtype)
        groups = np.ascontiguousarray(groups, dtype=group_dtype)
        feats = np.ascontiguousarray(feats, dtype=feat_dtype)
        nb, F = feats.shape
        self.F = F
        self._check(self.lib.xivo_hip_set_scene(self.h, b0, nb, F, _ptr(poses), _ptr(groups), _ptr(feats)))

    def jacobians_instate(self, B=None):
        self._check(self.lib.xivo_hip_jacobians_instate(self.h, self.batch if B is None else B))

    def get_jacobians(self, b0=0, nb=None):
        nb = self.batch - b0 if nb is None else nb
        J = np.empty((nb, self.F, 2, 21))
        inn = np.empty((nb, self.F, 2))
        self._check(self.lib.xivo_hip_get_jacobians(self.h, b0, nb, _ptr(J), _ptr(inn)))
        return J, inn

    def mh_gate(self, R, thresh, mult, min_inliers, B=None, want=True):
        B = self.batch if B is None else B
        if not want:      # mask / distances stay on the device (read them with get_gate)
            self._check(self.lib.xivo_hip_mh_gate(self.h, B, R, thresh, mult, min_inliers, None, None))
            return None
        mask = np.zeros((B, self.F), dtype=np.uint8)
        dist = np.zeros((B, self.F))
        self._check(self.lib.xivo_hip_mh_gate(self.h, B, R, thresh, mult, min_inliers, _ptr(mask), _ptr(dist)))
        return mask.astype(bool), dist

    def stack(self, R, B=None):
        self._check(self.lib.xivo_hip_stack(self.h, self.batch if B is None else B, R))

    def oos_project(self, feats, Roos, want_rows=True, whole_buffer=False):
        """feats: [nb, n_oos] oos_dtype, or a (nb, n_oos) tuple to project the resident list of the last call again.
        whole_buffer: XIVO_HIP_OOS_WHOLE_BUFFER - 2 * n_groups - 3 rows per feature, as src/oos.cpp:28 is coded"""
        if isinstance(feats, tuple):
            nb, n_oos = feats
            ptr = None
        else:
            feats = np.ascontiguousarray(feats, dtype=oos_dtype)
            nb, n_oos = feats.shape
            ptr = _ptr(feats)
        rows = np.zeros(nb, dtype=np.int32) if want_rows else None
        self._check(self.lib.xivo_hip_oos_project_ex(self.h, 0, nb, n_oos, ptr, Roos, _ptr(rows) if want_rows else None,
                                                     OOS_WHOLE_BUFFER if whole_buffer else 0))
        return rows

    def close_loop_stack(self, matches, Rlc, b0=0):
        """matches: [nb, n] lc_dtype - Feature::ComputeLCJacobian rows of Estimator::CloseLoopInternal become the staged measurement"""
        matches = np.ascontiguousarray(matches, dtype=lc_dtype)
        nb, n = matches.shape
        self._check(self.lib.xivo_hip_close_loop_stack(self.h, b0, nb, n, _ptr(matches), Rlc))

    def compress_oos(self, trigger_ratio=1.5, B=None, want_rows=True):
        """QR measurement compression of the OOS rows appended by oos_project (estimator.h:399-402)."""
        B = self.batch if B is None else B
        rows = np.zeros(B, dtype=np.int32)
        self._check(self.lib.xivo_hip_compress_oos(self.h, B, trigger_ratio, _ptr(rows) if want_rows else None))
        return rows

    def one_point_ransac(self, R, ransac_thresh, ransac_chi2, gauge=None, absorb_groups=None, B=None, want=True):
        """Estimator::OnePointRANSAC on the resident state (after jacobians_instate + mh_gate). gauge: [B] group slots
        (-1 none); absorb_groups: [B] uint64 masks of instate_groups_ (None: every slot).
        Returns (inlier mask [B, F], chi-square of the rescue test [B, F], rejected per filter [B])."""
        B = self.batch if B is None else B
        F = self.F
        g = None if gauge is None else np.ascontiguousarray(gauge, dtype=np.int32)
        ag = None if absorb_groups is None else np.ascontiguousarray(absorb_groups, dtype=np.uint64)
        if not want:      # results stay on the device (the inlier mask is read by the following stack / absorb)
            self._check(self.lib.xivo_hip_one_point_ransac(self.h, B, R, ransac_thresh, ransac_chi2, None if g is None else _ptr(g),
                                                           None if ag is None else _ptr(ag), None, None, None))
            return None
        mask = np.zeros((B, F), dtype=np.uint8); chi = np.zeros((B, F)); nrej = np.zeros(B, dtype=np.int32)
        self._check(self.lib.xivo_hip_one_point_ransac(self.h, B, R, ransac_thresh, ransac_chi2, None if g is None else _ptr(g),
                                                       None if ag is None else _ptr(ag), _ptr(mask), _ptr(chi), _ptr(nrej)))
        return mask.astype(bool), chi, nrej

    def filter_update(self, R, thresh, mult, min_inliers, use_gating=True, B=None):
        self._check(self.lib.xivo_hip_filter_update(self.h, self.batch if B is None else B, R, thresh, mult,
                                                    min_inliers, int(use_gating)))

    def last_path(self):
        """0: dense rows, 1: sparse-H (row-pair compressed) rows."""
        return int(self.lib.xivo_hip_last_path(self.h))

    def last_route(self):
        """Name of the route the last update pass took (the table of plan_update in capi.hip): fused, sparse_in_solve,
        sparse_whitened, sparse_symmetric, sparse_tail, dense_ascoded, dense_whitened, dense_symmetric."""
        return self.lib.xivo_hip_route_name(int(self.lib.xivo_hip_last_route(self.h))).decode()

    def subfilter_update(self, feats, Rtri=3.5, MH_thresh=5.991, ready_steps=5, min_depth=0.05, max_depth=5.0,
                         max_subfilter_outlier=0.01, b0=0):
        """feats: [nb, n] array of subfilter_dtype (P column-major); returns the updated copy."""
        feats = np.ascontiguousarray(feats, dtype=subfilter_dtype).copy()
        nb, n = feats.shape
        o = np.zeros(1, dtype=subfilter_opts_dtype)
        o["Rtri"], o["MH_thresh"], o["ready_steps"] = Rtri, MH_thresh, ready_steps
        o["min_depth"], o["max_depth"], o["max_subfilter_outlier"] = min_depth, max_depth, max_subfilter_outlier
        self._check(self.lib.xivo_hip_subfilter_update(self.h, b0, nb, n, _ptr(feats), _ptr(o)))
        return feats

    def givens(self, x, Hx, Hf, effective_rows=-1):
        """Batched xivo::Givens. x [nb, rows], Hx [nb, rows, nx], Hf [nb, rows, nf] (row-major numpy views of the
        matrices); returns (rows_out, x, Hx, Hf)."""
        nb, rows, nx = Hx.shape
        nf = Hf.shape[2]
        xd = np.ascontiguousarray(x, dtype=np.float64).copy()
        Hxd = np.ascontiguousarray(np.transpose(Hx, (0, 2, 1)), dtype=np.float64).copy()   # column-major per problem
        Hfd = np.ascontiguousarray(np.transpose(Hf, (0, 2, 1)), dtype=np.float64).copy()
        ro = np.zeros(nb, dtype=np.int32)
        self._check(self.lib.xivo_hip_givens(self.h, nb, rows, nx, nf, _ptr(xd), _ptr(Hxd), _ptr(Hfd), effective_rows, _ptr(ro)))
        return ro, xd, np.transpose(Hxd, (0, 2, 1)).copy(), np.transpose(Hfd, (0, 2, 1)).copy()

    def qr(self, x, Hx, effective_rows=-1):
        nb, rows, nx = Hx.shape
        xd = np.ascontiguousarray(x, dtype=np.float64).copy()
        Hxd = np.ascontiguousarray(np.transpose(Hx, (0, 2, 1)), dtype=np.float64).copy()
        ro = np.zeros(nb, dtype=np.int32)
        self._check(self.lib.xivo_hip_qr(self.h, nb, rows, nx, _ptr(xd), _ptr(Hxd), effective_rows, _ptr(ro)))
        return ro, xd, np.transpose(Hxd, (0, 2, 1)).copy()

    def propagate(self, imu, Qimu, Qmodel, g, method="RK4", stepsize=0.002, b0=0, pd_control=None):
        """imu: [nb] or [nb, n_imu] array of imu_dtype (the samples since the last call, in order); Qimu 12x12,
        Qmodel 23x23 (numpy row-major). pd_control: dict(tolerance, attempts, min_scale_factor, max_scale_factor) switches on the
        step-size-controlled branch of Estimator::PrinceDormand (src/princedormand.cpp:26-60, as coded)."""
        imu = np.ascontiguousarray(imu, dtype=imu_dtype)
        if imu.ndim == 1:
            imu = imu[:, None]
        imu = np.ascontiguousarray(imu)
        o = np.zeros(1, dtype=prop_opts_dtype)
        o["Qimu"] = np.asarray(Qimu, dtype=np.float64).T.reshape(-1)
        o["Qmodel"] = np.asarray(Qmodel, dtype=np.float64).T.reshape(-1)
        o["g"] = g; o["method"] = 0 if method == "RK4" else 1; o["stepsize"] = stepsize
        if pd_control is not None:
            o["control_stepsize"] = 1
            o["tolerance"] = pd_control.get("tolerance", 1e-3); o["attempts"] = pd_control.get("attempts", 12)
            o["min_scale_factor"] = pd_control.get("min_scale_factor", 0.125); o["max_scale_factor"] = pd_control.get("max_scale_factor", 4.0)
        self._check(self.lib.xivo_hip_propagate(self.h, b0, imu.shape[0], imu.shape[1], _ptr(imu), _ptr(o)))

    def propagate_calib(self, imu, Qimu, Qmodel, g, method="RK4", stepsize=0.002, b0=0, pd_control=None):
        """Estimator::Propagate of an online-calibration build (set_calib with td >= 0 or Cg >= 0): Qmodel is
        kMotionSize x kMotionSize (numpy row-major); the resident calibration state supplies Cg / Ca."""
        imu = np.ascontiguousarray(imu, dtype=imu_dtype)
        if imu.ndim == 1:
            imu = imu[:, None]
        imu = np.ascontiguousarray(imu)
        o = np.zeros(1, dtype=prop_opts_dtype)
        o["Qimu"] = np.asarray(Qimu, dtype=np.float64).T.reshape(-1)
        o["g"] = g; o["method"] = 0 if method == "RK4" else 1; o["stepsize"] = stepsize
        if pd_control is not None:
            o["control_stepsize"] = 1
            o["tolerance"] = pd_control.get("tolerance", 1e-3); o["attempts"] = pd_control.get("attempts", 12)
            o["min_scale_factor"] = pd_control.get("min_scale_factor", 0.125); o["max_scale_factor"] = pd_control.get("max_scale_factor", 4.0)
        Qm = np.ascontiguousarray(np.asarray(Qmodel, dtype=np.float64).T)
        self._check(self.lib.xivo_hip_propagate_calib(self.h, b0, imu.shape[0], imu.shape[1], _ptr(imu), _ptr(o), _ptr(Qm)))

    def absorb_error(self, B=None):
        self._check(self.lib.xivo_hip_absorb_error(self.h, self.batch if B is None else B))

    def edit_batch(self, F, ops):
        """ops: array of edit_dtype; sorted by filter here (stable, so the per-filter order is kept)."""
        ops = np.ascontiguousarray(ops, dtype=edit_dtype)
        if ops.size:
            ops = np.ascontiguousarray(ops[np.argsort(ops["b"], kind="stable")])
        self.F = F
        self._check(self.lib.xivo_hip_edit_batch(self.h, F, int(ops.size), _ptr(ops) if ops.size else None))

    def set_pixels(self, xp, b0=0):
        """xp: [nb, F, 2], NaN = leave the entry's pixel as it is"""
        xp = np.ascontiguousarray(xp, dtype=np.float64)
        nb, F = xp.shape[:2]
        self.F = F
        self._check(self.lib.xivo_hip_set_pixels(self.h, b0, nb, F, _ptr(xp)))

    def get_scene(self, b0=0, nb=None):
        nb = self.batch - b0 if nb is None else nb
        poses = np.zeros(nb, dtype=pose_dtype)
        groups = np.zeros((nb, self.layout.n_groups), dtype=group_dtype)
        feats = np.zeros((nb, self.F), dtype=feat_dtype)
        self._check(self.lib.xivo_hip_get_scene(self.h, b0, nb, _ptr(poses), _ptr(groups), _ptr(feats)))
        return poses, groups, feats

    def get_H(self, b):
        M = C.c_int()
        self._check(self.lib.xivo_hip_get_H(self.h, b, C.byref(M), None, 0, None, None))
        M = M.value
        H = np.empty((self.N, M))
        inn = np.empty(M)
        dR = np.empty(M)
        self._check(self.lib.xivo_hip_get_H(self.h, b, None, _ptr(H), M, _ptr(inn), _ptr(dR)))
        return H.T.copy(), inn, dR

    def propagate_cov(self, Phi, Pmm, b0=0):
        Phi = np.asarray(Phi, dtype=np.float64)
        nb, nm, _ = Phi.shape
        Phic = _f64(np.transpose(Phi, (0, 2, 1)))
        Pmmc = _f64(np.transpose(np.asarray(Pmm, dtype=np.float64), (0, 2, 1)))
        self._check(self.lib.xivo_hip_propagate_cov(self.h, b0, nb, nm, _ptr(Phic), _ptr(Pmmc)))

    # ---- timing ---------------------------------------------------------------
    def timer_begin(self):
        self._check(self.lib.xivo_hip_timer_begin(self.h))

    def timer_end(self):
        ms = C.c_float()
        self._check(self.lib.xivo_hip_timer_end(self.h, C.byref(ms)))
        return ms.value

    def profile_reset(self):
        self._check(self.lib.xivo_hip_profile_reset(self.h))

    def profile_get(self):
        n = C.c_int()
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        launches = (C.c_int * 16)()
        flops = (C.c_double * 16)()
        self._check(self.lib.xivo_hip_profile_get(self.h, C.byref(n), names, ms, launches, flops))
        return {names[i].decode(): {"ms": ms[i], "launches": launches[i], "flops_per_launch": flops[i],
                                    "kernel": self.lib.xivo_hip_stage_kernel(self.h, i).decode(),
                                    "bytes_per_launch": self.lib.xivo_hip_stage_bytes(self.h, i)}
                for i in range(n.value)}

    def bench_mfma_peak(self):
        t = (C.c_double * 4)()
        self._check(self.lib.xivo_hip_bench_mfma_peak(self.h, t))
        return {"tflops_full_chip": t[0], "cycles_per_mfma_1wave_per_simd": t[1],
                "clock_ghz_lower_bound_full_chip": t[2], "tflops_1wave_per_simd": t[3]}
