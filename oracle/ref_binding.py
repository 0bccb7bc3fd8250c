"""ctypes loader of oracle/_ref/libxivo_ref_*.so (TEST INFRASTRUCTURE ONLY).

The library is the reference's own arithmetic (Eigen 3.3.9 / Sophus / helpers.cpp /
camera headers compiled from /root/reference where they lie, see oracle/ref/). It is
built in the authoring container and travels prebuilt to the GPU box; load() returns
None-raising errors if it is absent so callers can skip."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class CamCfg(C.Structure):
    _fields_ = [("model", C.c_int), ("rows", C.c_int), ("cols", C.c_int), ("fx", C.c_double), ("fy", C.c_double),
                ("cx", C.c_double), ("cy", C.c_double), ("d", C.c_double * 5)]


def _cam(cam):
    c = CamCfg()
    c.model, c.rows, c.cols = cam["model"], cam.get("rows", 480), cam.get("cols", 640)
    c.fx, c.fy, c.cx, c.cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    d = list(cam.get("d", [])) + [0.0] * 5
    for i in range(5):
        c.d[i] = d[i]
    return c


def _has_avx512():
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        return False
    need = ("avx512f", "avx512bw", "avx512cd", "avx512dq", "avx512vl")
    return all(f in flags for f in need)


def _F(a):  # column-major copy
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Ref:
    def __init__(self, path):
        self.path = path
        self.lib = C.CDLL(path)
        self.lib.ref_slow_givens.restype = C.c_int
        self.lib.ref_Givens.restype = C.c_int
        self.lib.ref_fullpivlu_kernel.restype = C.c_int

    def update_joseph(self, H, P, inn, diagR):
        M, N = H.shape
        Hf, Pf = _F(H), _F(P)
        err = np.empty(N); Pout = np.empty((N, N), order="F")
        self.lib.ref_update_joseph(C.c_int(N), C.c_int(M), _p(Hf), _p(Pf), _p(np.ascontiguousarray(inn, dtype=np.float64)),
                                   _p(np.ascontiguousarray(diagR, dtype=np.float64)), _p(err), _p(Pout))
        return err, np.ascontiguousarray(Pout)

    def mh_distances(self, J, P, inn, R):
        F, _, N = J.shape
        Jc = np.ascontiguousarray(np.transpose(J, (0, 2, 1)), dtype=np.float64)  # each 2 x N col-major
        d = np.empty(F)
        self.lib.ref_mh_distances(C.c_int(F), C.c_int(N), _p(Jc), _p(_F(P)), _p(np.ascontiguousarray(inn, dtype=np.float64)),
                                  C.c_double(R), _p(d))
        return d

    def compute_jacobian(self, x, xp_meas, Rsbr, Tsbr, Rsb, Tsb, Rbc, Tbc, cam, layout, ref_sind, sind):
        N = layout.N
        J = np.empty((2, N), order="F"); inn = np.empty(2); cache = np.empty(9 + 63)
        c = _cam(cam)
        v = lambda a: _p(np.ascontiguousarray(a, dtype=np.float64))
        Rs = [_F(Rsbr), _F(Rsb), _F(Rbc)]
        self.lib.ref_compute_jacobian(v(x), v(xp_meas), _p(Rs[0]), v(Tsbr), _p(Rs[1]), v(Tsb), _p(Rs[2]), v(Tbc),
                                      C.byref(c), C.c_int(N), C.c_int(layout.group_begin), C.c_int(layout.feature_begin),
                                      C.c_int(ref_sind), C.c_int(sind), _p(J), _p(inn), _p(cache))
        names = ["dXcn_dWsb", "dXcn_dTsb", "dXcn_dWbc", "dXcn_dTbc", "dXcn_dWsbr", "dXcn_dTsbr", "dXcn_dXs"]
        cd = dict(Xc=cache[0:3].copy(), Xs=cache[3:6].copy(), Xcn=cache[6:9].copy())
        for i, n in enumerate(names):
            cd[n] = cache[9 + 9 * i:18 + 9 * i].reshape(3, 3).T.copy()
        return np.ascontiguousarray(J), inn, cd

    def subfilter_update(self, x, P, xp_meas, Rsb, Tsb, Rbc, Tbc, Rsbr, Tsbr, cam, Rtri, MH_thresh, ready_steps,
                         init_counter, outlier_counter):
        """Feature::SubfilterUpdate. Returns (x, P, status, init_counter, outlier_counter)."""
        x = np.array(x, dtype=np.float64); Pf = _F(np.array(P, dtype=np.float64))
        ic = C.c_int(init_counter); oc = C.c_double(outlier_counter)
        c = _cam(cam)
        v = lambda a: _p(np.ascontiguousarray(a, dtype=np.float64))
        Rs = [_F(Rsb), _F(Rbc), _F(Rsbr)]
        self.lib.ref_subfilter_update.restype = C.c_int
        st = self.lib.ref_subfilter_update(_p(x), _p(Pf), v(xp_meas), _p(Rs[0]), v(Tsb), _p(Rs[1]), v(Tbc), _p(Rs[2]),
                                           v(Tsbr), C.byref(c), C.c_double(Rtri), C.c_double(MH_thresh),
                                           C.c_int(ready_steps), C.byref(ic), C.byref(oc))
        return x, np.ascontiguousarray(Pf), int(st), ic.value, oc.value

    def fill_jacobian_block(self, H, row, J, layout, ref_sind, sind):
        M, N = H.shape
        Hf = _F(H)
        self.lib.ref_fill_jacobian_block(_p(Hf), C.c_int(M), C.c_int(N), C.c_int(row), _p(_F(J)),
                                         C.c_int(layout.group_begin), C.c_int(layout.feature_begin),
                                         C.c_int(ref_sind), C.c_int(sind))
        H[...] = Hf

    def oos_internal(self, Xs, Rsb, Tsb, Rbc, Tbc, xp_obs, cam, layout, g_sind):
        N = layout.N
        Hf = np.empty((2, 3), order="F"); Hx = np.empty((2, N), order="F"); inn = np.empty(2)
        c = _cam(cam)
        v = lambda a: _p(np.ascontiguousarray(a, dtype=np.float64))
        R1, R2 = _F(Rsb), _F(Rbc)
        self.lib.ref_oos_internal(v(Xs), _p(R1), v(Tsb), _p(R2), v(Tbc), v(xp_obs), C.byref(c), C.c_int(N),
                                  C.c_int(layout.group_begin), C.c_int(g_sind), _p(Hf), _p(Hx), _p(inn))
        return np.ascontiguousarray(Hf), np.ascontiguousarray(Hx), inn

    def slow_givens(self, Hf, Hx, inn):
        R, N = Hx.shape
        Hxo = np.empty((R, N), order="F"); ro = np.empty(R); A = np.empty((R, R), order="F"); ac = C.c_int()
        rows = self.lib.ref_slow_givens(C.c_int(R), C.c_int(N), _p(_F(Hf)), _p(_F(Hx)),
                                        _p(np.ascontiguousarray(inn, dtype=np.float64)), _p(Hxo), _p(ro), _p(A), C.byref(ac))
        Hxo = np.ndarray((rows, N), dtype=np.float64, buffer=Hxo.data, order="F").copy()
        A = np.ndarray((R, ac.value), dtype=np.float64, buffer=A.data, order="F").copy()
        return np.ascontiguousarray(Hxo), ro[:rows].copy(), np.ascontiguousarray(A)

    def givens(self, a, b):
        G = np.empty((2, 2), order="F")
        self.lib.ref_givens(C.c_double(a), C.c_double(b), _p(G))
        return np.ascontiguousarray(G)

    def Givens(self, x, Hx, Hf, effective_rows=-1):
        R, N = Hx.shape
        xf = np.ascontiguousarray(x, dtype=np.float64).copy(); Hxf = _F(Hx).copy(order="F"); Hff = _F(Hf).copy(order="F")
        rows = self.lib.ref_Givens(C.c_int(R), C.c_int(N), _p(xf), _p(Hxf), _p(Hff), C.c_int(effective_rows))
        return rows, xf, np.ascontiguousarray(Hxf), np.ascontiguousarray(Hff)

    def QR(self, x, Hx, effective_rows=-1):
        R, N = Hx.shape
        xf = np.ascontiguousarray(x, dtype=np.float64).copy(); Hxf = _F(Hx).copy(order="F")
        rows = self.lib.ref_QR(C.c_int(R), C.c_int(N), _p(xf), _p(Hxf), C.c_int(effective_rows))
        return rows, xf, np.ascontiguousarray(Hxf)

    def fullpivlu_kernel(self, A):
        r, c = A.shape
        ker = np.empty((c, c), order="F"); rank = C.c_int()
        k = self.lib.ref_fullpivlu_kernel(C.c_int(r), C.c_int(c), _p(_F(A)), _p(ker), C.byref(rank))
        ker = np.ndarray((c, k), dtype=np.float64, buffer=ker.data, order="F").copy()
        return np.ascontiguousarray(ker), rank.value

    def camera_project(self, cam, xc):
        c = _cam(cam)
        xp = np.empty(2); J = np.empty((2, 2), order="F")
        self.lib.ref_camera_project(C.byref(c), _p(np.ascontiguousarray(xc, dtype=np.float64)), _p(xp), _p(J))
        return xp, np.ascontiguousarray(J)

    def rk4_cov_tail(self, P, FK, PK, dt, Qmodel=None):
        N = P.shape[0]; nm = FK.shape[0]
        Pf = _F(P).copy(order="F")
        q = _p(_F(Qmodel)) if Qmodel is not None else None
        Qf = _F(Qmodel) if Qmodel is not None else None
        self.lib.ref_rk4_cov_tail(C.c_int(N), C.c_int(nm), _p(Pf), _p(_F(FK)), _p(_F(PK)), C.c_double(dt),
                                  _p(Qf) if Qf is not None else None)
        return np.ascontiguousarray(Pf)

    def rk4_step(self, X, P, gyro0, accel0, slope_gyro, slope_accel, dt, Qimu, g_vec):
        """X: oracle MotionState. Returns (Rsb, Tsb, Vsb, P_new)."""
        N = P.shape[0]
        st = np.concatenate([_F(X.Rsb).reshape(-1, order="F"), X.Tsb, X.Vsb, X.bg, X.ba, _F(X.Rsg).reshape(-1, order="F")])
        st = np.ascontiguousarray(st, dtype=np.float64)
        Pf = _F(P).copy(order="F")
        v = lambda a: _p(np.ascontiguousarray(a, dtype=np.float64))
        Qf = _F(Qimu)
        keep = [np.ascontiguousarray(a, dtype=np.float64) for a in (gyro0, accel0, slope_gyro, slope_accel, g_vec)]
        self.lib.ref_rk4_step(C.c_int(N), _p(st), _p(Pf), _p(keep[0]), _p(keep[1]), _p(keep[2]), _p(keep[3]), C.c_double(dt),
                              _p(Qf), _p(keep[4]))
        return st[0:9].reshape(3, 3).T.copy(), st[9:12].copy(), st[12:15].copy(), np.ascontiguousarray(Pf)

    def pd_step(self, X, P, gyro0, accel0, slope_gyro, slope_accel, dt, Qimu, g_vec):
        """Estimator::PrinceDormandStep (src/princedormand.cpp:85-221) restated on Sophus / Eigen. Returns (Rsb, Tsb, Vsb, P_new)."""
        return self._step(self.lib.ref_pd_step, X, P, gyro0, accel0, slope_gyro, slope_accel, dt, Qimu, g_vec)

    def _step(self, fn, X, P, gyro0, accel0, slope_gyro, slope_accel, dt, Qimu, g_vec):
        N = P.shape[0]
        st = np.concatenate([_F(X.Rsb).reshape(-1, order="F"), X.Tsb, X.Vsb, X.bg, X.ba, _F(X.Rsg).reshape(-1, order="F")])
        st = np.ascontiguousarray(st, dtype=np.float64)
        Pf = _F(P).copy(order="F")
        Qf = _F(Qimu)
        keep = [np.ascontiguousarray(a, dtype=np.float64) for a in (gyro0, accel0, slope_gyro, slope_accel, g_vec)]
        fn(C.c_int(N), _p(st), _p(Pf), _p(keep[0]), _p(keep[1]), _p(keep[2]), _p(keep[3]), C.c_double(dt), _p(Qf), _p(keep[4]))
        return st[0:9].reshape(3, 3).T.copy(), st[9:12].copy(), st[12:15].copy(), np.ascontiguousarray(Pf)

    def one_point_ransac_core(self, J, inn, P, sind, ref, gauge, layout, R, ransac_thresh):
        """Estimator::OnePointRANSAC (src/update.cpp:238-332): low-innovation set, temporary reference group, zeroing of
        P, stacking of the full rows, UpdateJosephForm. J: [F, 2, N], inn: [F, 2]. Returns (n_low or -1, low mask, err,
        P after zeroing + update)."""
        F, _, N = J.shape
        Jc = np.ascontiguousarray(np.stack([_F(J[i]).reshape(-1, order="F") for i in range(F)]))
        innc = np.ascontiguousarray(inn, dtype=np.float64)
        sind = np.ascontiguousarray(sind, dtype=np.int32); ref = np.ascontiguousarray(ref, dtype=np.int32)
        low = np.zeros(F, dtype=np.int32); err = np.zeros(N); Pout = np.zeros((N, N), order="F")
        self.lib.ref_one_point_ransac_core.restype = C.c_int
        n = self.lib.ref_one_point_ransac_core(C.c_int(N), C.c_int(F), _p(Jc), _p(innc), _p(_F(P)), _p(sind), _p(ref),
                                               C.c_int(int(gauge)), C.c_int(layout.group_begin), C.c_int(layout.feature_begin),
                                               C.c_double(R), C.c_double(ransac_thresh), _p(low), _p(err), _p(Pout))
        return n, low.astype(bool), err, np.ascontiguousarray(Pout)

    def so3_exp(self, w):
        R = np.empty((3, 3), order="F")
        self.lib.ref_so3_exp(_p(np.ascontiguousarray(w, dtype=np.float64)), _p(R))
        return np.ascontiguousarray(R)


class RefX:
    """oracle/_ref/libxivo_refx_n<N>_*.so: the reference's OWN TEXT of UpdateJosephForm, MHGating, FilterUpdate (+
    FillJacobianBlock, AbsorbError), RK4Step / PrinceDormandStep (+ ComposeMotion, ComputeMotionJacobianAt), cut out of
    /root/reference/src at build time (oracle/ref/extract_reference.py) and compiled as member functions of shim classes
    (oracle/ref/xivo_refx.cpp). One library per compile-time state size."""

    def __init__(self, path):
        self.path = path
        self.lib = C.CDLL(path)
        for f in ("refx_full_size", "refx_group_begin", "refx_feature_begin", "refx_mh_gating"):
            getattr(self.lib, f).restype = C.c_int
        self.N = self.lib.refx_full_size()
        self.group_begin, self.feature_begin = self.lib.refx_group_begin(), self.lib.refx_feature_begin()

    def update_joseph(self, H, P, inn, diagR):
        M, N = H.shape
        err = np.empty(N); Pout = np.empty((N, N), order="F")
        self.lib.refx_update_joseph(C.c_int(N), C.c_int(M), _p(_F(H)), _p(_F(P)), _p(np.ascontiguousarray(inn, dtype=np.float64)),
                                    _p(np.ascontiguousarray(diagR, dtype=np.float64)), _p(err), _p(Pout))
        return err, np.ascontiguousarray(Pout)

    def _Jc(self, J):
        F, _, N = J.shape
        assert N == self.N, (N, self.N)
        return np.ascontiguousarray(np.stack([_F(J[i]).reshape(-1, order="F") for i in range(F)]))

    def mh_gating(self, J, inn, P, R, thresh, mult, min_inliers, status=None):
        """Estimator::MHGating as extracted. J [F, 2, N], inn [F, 2]. Returns (inlier indices in order, status after,
        num_mh_rejected_, number of features handed to DestroyFeatures)."""
        F = J.shape[0]
        st = np.full(F, 3, dtype=np.int32) if status is None else np.ascontiguousarray(status, dtype=np.int32).copy()
        idx = np.full(F, -1, dtype=np.int32); nrej = C.c_int(); ndes = C.c_int()
        n = self.lib.refx_mh_gating(C.c_int(F), _p(self._Jc(J)), _p(np.ascontiguousarray(inn, dtype=np.float64)), _p(_F(P)),
                                    C.c_double(R), C.c_double(thresh), C.c_double(mult), C.c_int(min_inliers), _p(st), _p(idx),
                                    C.byref(nrej), C.byref(ndes))
        return idx[:n].copy(), st, nrej.value, ndes.value

    def filter_update(self, J, inn, ref_sind, sind, R, P, X, Rbc, Tbc, x):
        """Estimator::FilterUpdate as extracted (stacking through Feature::FillJacobianBlock, UpdateJosephForm,
        AbsorbError). X: oracle MotionState; x [F, 3]. Returns (H, err before the absorb, P+, Rsb, Tsb, Vsb, bg, ba, Rsg, x+)."""
        F = J.shape[0]
        st = np.ascontiguousarray(np.concatenate([_F(X.Rsb).reshape(-1, order="F"), X.Tsb, X.Vsb, X.bg, X.ba,
                                                  _F(X.Rsg).reshape(-1, order="F")]), dtype=np.float64)
        Pf = _F(P).copy(order="F"); xs = np.ascontiguousarray(x, dtype=np.float64).copy()
        H = np.zeros((2 * F, self.N), order="F"); err = np.zeros(self.N)
        self.lib.refx_filter_update(C.c_int(F), _p(self._Jc(J)), _p(np.ascontiguousarray(inn, dtype=np.float64)),
                                    _p(np.ascontiguousarray(ref_sind, dtype=np.int32)), _p(np.ascontiguousarray(sind, dtype=np.int32)),
                                    C.c_double(R), _p(Pf), _p(st), _p(_F(Rbc)), _p(np.ascontiguousarray(Tbc, dtype=np.float64)), _p(xs),
                                    _p(H), _p(err))
        return (np.ascontiguousarray(H), err, np.ascontiguousarray(Pf), st[0:9].reshape(3, 3).T.copy(), st[9:12].copy(), st[12:15].copy(),
                st[15:18].copy(), st[18:21].copy(), st[21:30].reshape(3, 3).T.copy(), xs)

    def calib_slots(self):
        """(td, Cg, cam_begin, max camera intrinsics, kMotionSize) of this build as the extracted enum Index numbers them"""
        for f in ("refx_index_td", "refx_index_Cg", "refx_camera_begin", "refx_max_camera_intrinsics", "refx_motion_size"):
            getattr(self.lib, f).restype = C.c_int
        return (self.lib.refx_index_td(), self.lib.refx_index_Cg(), self.lib.refx_camera_begin(),
                self.lib.refx_max_camera_intrinsics(), self.lib.refx_motion_size())

    def compute_jacobian(self, x, xp_meas, Rsbr, Tsbr, Rsb, Tsb, Rbc, Tbc, cam, ref_sind, sind, gyro=(0, 0, 0), Cg=None,
                         bg=(0, 0, 0), Vsb=(0, 0, 0), td=0.0):
        """Feature::ComputeJacobian + FillJacobianBlock as extracted (this library's build: default or online calibration).
        Returns (J [2, N], inn [2], the two stacked rows [2, N])."""
        N = self.N
        J = np.zeros((2, N), order="F"); inn = np.zeros(2); Hrow = np.zeros((2, N), order="F")
        v = lambda a: _p(np.ascontiguousarray(a, dtype=np.float64))
        Cgm = np.eye(3) if Cg is None else np.asarray(Cg, float)
        c = _cam(cam)
        keep = [np.ascontiguousarray(a, dtype=np.float64) for a in (x, xp_meas, Tsbr, Tsb, Tbc, gyro, bg, Vsb)]
        mats = [_F(Rsbr), _F(Rsb), _F(Rbc), _F(Cgm)]
        self.lib.refx_compute_jacobian(_p(keep[0]), _p(keep[1]), _p(mats[0]), _p(keep[2]), _p(mats[1]), _p(keep[3]), _p(mats[2]),
                                       _p(keep[4]), _p(keep[5]), _p(mats[3]), _p(keep[6]), _p(keep[7]), C.c_double(td), C.byref(c),
                                       C.c_int(int(ref_sind)), C.c_int(int(sind)), _p(J), _p(inn), _p(Hrow))
        return np.ascontiguousarray(J), inn, np.ascontiguousarray(Hrow)

    def integrator_step(self, method, X, P, gyro0, accel0, slope_gyro, slope_accel, dt, Qimu, g_vec, Cg=None, Ca=None):
        """Estimator::RK4Step / PrinceDormandStep as extracted (imu_.Cg() / imu_.Ca() = Cg / Ca, identity when None; in the
        online-calibration build the motion Jacobian also carries the Cg / Ca columns, src/estimator.cpp:674-688).
        Returns (Rsb, Tsb, Vsb, P_new)."""
        assert P.shape[0] == self.N
        st = np.ascontiguousarray(np.concatenate([_F(X.Rsb).reshape(-1, order="F"), X.Tsb, X.Vsb, X.bg, X.ba,
                                                  _F(X.Rsg).reshape(-1, order="F")]), dtype=np.float64)
        Pf = _F(P).copy(order="F")
        keep = [np.ascontiguousarray(a, dtype=np.float64) for a in (gyro0, accel0, slope_gyro, slope_accel, g_vec)]
        mats = [_F(np.eye(3) if Cg is None else Cg), _F(np.eye(3) if Ca is None else Ca)]
        self.lib.refx_integrator_step_calib(C.c_int(1 if method == "RK4" else 0), _p(st), _p(Pf), _p(keep[0]), _p(keep[1]), _p(keep[2]),
                                            _p(keep[3]), C.c_double(dt), _p(_F(Qimu)), _p(keep[4]), _p(mats[0]), _p(mats[1]))
        return st[0:9].reshape(3, 3).T.copy(), st[9:12].copy(), st[12:15].copy(), np.ascontiguousarray(Pf)

    def index_Ca(self):
        self.lib.refx_index_Ca.restype = C.c_int
        return self.lib.refx_index_Ca()

    def absorb_motion_calib(self, X, Rbc, Tbc, td, Cg, Ca, cam, err):
        """Estimator::AbsorbError(err) as extracted on the motion state + td + IMU calibration + camera intrinsics (no groups /
        features listed). Returns dict(Rsb, Tsb, Vsb, bg, ba, Rsg, Rbc, Tbc, td, Cg, Ca, cam)."""
        st = np.ascontiguousarray(np.concatenate([_F(X.Rsb).reshape(-1, order="F"), X.Tsb, X.Vsb, X.bg, X.ba,
                                                  _F(X.Rsg).reshape(-1, order="F")]), dtype=np.float64)
        Rbo = np.zeros((3, 3), order="F"); Tbo = np.zeros(3)
        tdv = C.c_double(td); Cgf = _F(Cg).copy(order="F"); Caf = _F(Ca).copy(order="F")
        c = _cam(cam)
        keep = [np.ascontiguousarray(Tbc, dtype=np.float64), np.ascontiguousarray(err, dtype=np.float64), _F(Rbc)]
        assert keep[1].shape[0] == self.N
        self.lib.refx_absorb_motion_calib(_p(st), _p(keep[2]), _p(keep[0]), _p(Rbo), _p(Tbo), C.byref(tdv), _p(Cgf), _p(Caf), C.byref(c),
                                          _p(keep[1]))
        camo = dict(cam); camo.update(fx=c.fx, fy=c.fy, cx=c.cx, cy=c.cy, d=[c.d[i] for i in range(len(cam.get("d", [])))])
        return dict(Rsb=st[0:9].reshape(3, 3).T.copy(), Tsb=st[9:12].copy(), Vsb=st[12:15].copy(), bg=st[15:18].copy(),
                    ba=st[18:21].copy(), Rsg=st[21:30].reshape(3, 3).T.copy(), Rbc=np.ascontiguousarray(Rbo), Tbc=Tbo, td=tdv.value,
                    Cg=np.ascontiguousarray(Cgf), Ca=np.ascontiguousarray(Caf), cam=camo)


    # ---- round 5: the remaining rows of the path on the reference's own text -------------------------------------------
    def _state30(self, X):
        return np.ascontiguousarray(np.concatenate([_F(X.Rsb).reshape(-1, order="F"), X.Tsb, X.Vsb, X.bg, X.ba,
                                                    _F(X.Rsg).reshape(-1, order="F")]), dtype=np.float64)

    def max_group(self):
        self.lib.refx_max_group.restype = C.c_int
        return self.lib.refx_max_group()

    def use_invdepth(self):
        self.lib.refx_use_invdepth.restype = C.c_int
        return bool(self.lib.refx_use_invdepth())

    def feature_xs(self, x, Rsbr, Tsbr, Rbc, Tbc):
        """Feature::Xs(gbc) as extracted (src/feature.cpp:107-118; Xc through unproject_logz / unproject_invz per build)."""
        out = np.zeros(3)
        keep = [np.ascontiguousarray(a, dtype=np.float64) for a in (x, Tsbr, Tbc)]
        mats = [_F(Rsbr), _F(Rbc)]
        self.lib.refx_feature_xs(_p(keep[0]), _p(mats[0]), _p(keep[1]), _p(mats[1]), _p(keep[2]), _p(out))
        return out

    def compute_oos_jacobian(self, x, Rsbr, Tsbr, obs, groups_R, groups_T, Rbc, Tbc, cam, min_obs=5, instate=None):
        """Feature::ComputeOOSJacobian (+Internal, SlowGivens) as extracted, whole-buffer quirk of src/oos.cpp:28 included.
        obs = [(g_sind, xp), ...]. Returns (rows, Hx [rows, N], inn [rows], Xs); rows = 0 below min_obs in-state observations."""
        n = len(obs)
        oR = np.ascontiguousarray(np.stack([_F(groups_R[g]).reshape(-1, order="F") for g, _ in obs]))
        oT = np.ascontiguousarray(np.stack([np.asarray(groups_T[g], float) for g, _ in obs]))
        osind = np.ascontiguousarray([g for g, _ in obs], dtype=np.int32)
        oin = np.ascontiguousarray(np.ones(n) if instate is None else instate, dtype=np.int32)
        oxp = np.ascontiguousarray(np.stack([np.asarray(p, float) for _, p in obs]))
        cap = 2 * self.max_group()
        Hx = np.zeros(cap * self.N); inn = np.zeros(cap); Xs = np.zeros(3)
        keep = [np.ascontiguousarray(a, dtype=np.float64) for a in (x, Tsbr, Tbc)]
        mats = [_F(Rsbr), _F(Rbc)]
        c = _cam(cam)
        self.lib.refx_compute_oos_jacobian.restype = C.c_int
        rows = self.lib.refx_compute_oos_jacobian(_p(keep[0]), _p(mats[0]), _p(keep[1]), C.c_int(n), _p(oR), _p(oT), _p(osind), _p(oin), _p(oxp),
                                                  _p(mats[1]), _p(keep[2]), C.byref(c), C.c_int(min_obs), _p(Xs), _p(Hx), _p(inn))
        if rows <= 0:
            return 0, np.zeros((0, self.N)), np.zeros(0), Xs
        return rows, np.ascontiguousarray(Hx[:rows * self.N].reshape(self.N, rows).T), inn[:rows].copy(), Xs

    def compute_lc_jacobian(self, x, Rsbr, Tsbr, obs_R, obs_T, obs_sind, obs_xp, Rbc, Tbc, cam):
        """Feature::ComputeLCJacobian as extracted, one call per match on a zeroed H_ / inn_ (Estimator::CloseLoopInternal,
        src/update.cpp:183-196). Arrays over the n matches. Returns (H [2n, N], inn [2n])."""
        n = len(obs_sind)
        xs = np.ascontiguousarray(x, dtype=np.float64).reshape(n, 3)
        rR = np.ascontiguousarray(np.stack([_F(R).reshape(-1, order="F") for R in Rsbr])); rT = np.ascontiguousarray(Tsbr, dtype=np.float64).reshape(n, 3)
        oR = np.ascontiguousarray(np.stack([_F(R).reshape(-1, order="F") for R in obs_R])); oT = np.ascontiguousarray(obs_T, dtype=np.float64).reshape(n, 3)
        osind = np.ascontiguousarray(obs_sind, dtype=np.int32); oxp = np.ascontiguousarray(obs_xp, dtype=np.float64).reshape(n, 2)
        H = np.zeros((2 * n, self.N), order="F"); inn = np.zeros(2 * n)
        Rb = _F(Rbc); Tb = np.ascontiguousarray(Tbc, dtype=np.float64); c = _cam(cam)
        self.lib.refx_compute_lc_jacobian(C.c_int(n), _p(xs), _p(rR), _p(rT), _p(oR), _p(oT), _p(osind), _p(oxp), _p(Rb), _p(Tb), C.byref(c),
                                          _p(H), _p(inn))
        return np.ascontiguousarray(H), inn

    def subfilter_update(self, x, P, xp_meas, Rsb, Tsb, Rbc, Tbc, Rsbr, Tsbr, cam, Rtri, MH_thresh, ready_steps, init_counter,
                         outlier_counter):
        """Feature::SubfilterUpdate as extracted. Returns (x, P, status 0/1, init_counter, outlier_counter) like Ref's."""
        xs = np.ascontiguousarray(x, dtype=np.float64).copy(); Pf = _F(P).copy(order="F")
        ic = C.c_int(int(init_counter)); oc = C.c_double(float(outlier_counter)); c = _cam(cam)
        keep = [np.ascontiguousarray(a, dtype=np.float64) for a in (xp_meas, Tsb, Tbc, Tsbr)]
        mats = [_F(Rsb), _F(Rbc), _F(Rsbr)]
        self.lib.refx_subfilter_update.restype = C.c_int
        st = self.lib.refx_subfilter_update(_p(xs), _p(Pf), _p(keep[0]), _p(mats[0]), _p(keep[1]), _p(mats[1]), _p(keep[2]), _p(mats[2]),
                                            _p(keep[3]), C.byref(c), C.c_double(Rtri), C.c_double(MH_thresh), C.c_int(ready_steps),
                                            C.byref(ic), C.byref(oc))
        return xs, np.ascontiguousarray(Pf), st, ic.value, oc.value

    def propagate(self, method, X, P, last_gyro, last_accel, slope_gyro, slope_accel, dt_ns, Qimu, Qmodel, g_vec, stepsize=0.002,
                  curr_gyro=None, curr_accel=None, Cg=None, Ca=None):
        """Estimator::Propagate as extracted, with the outer loops Estimator::RK4 / PrinceDormand as extracted. curr_* given: an
        IMU message (slopes computed, :559-568); else a visual message (slopes as handed in, :569-575). dt = dt_ns ns.
        Returns dict(Rsb, Tsb, Vsb, P, last_gyro, last_accel, slope_gyro, slope_accel)."""
        assert P.shape[0] == self.N
        st = self._state30(X); Pf = _F(P).copy(order="F")
        visual = curr_gyro is None
        lg = np.ascontiguousarray(last_gyro, dtype=np.float64).copy(); la = np.ascontiguousarray(last_accel, dtype=np.float64).copy()
        sg = np.ascontiguousarray(slope_gyro, dtype=np.float64).copy(); sa = np.ascontiguousarray(slope_accel, dtype=np.float64).copy()
        cg = np.zeros(3) if visual else np.ascontiguousarray(curr_gyro, dtype=np.float64)
        ca = np.zeros(3) if visual else np.ascontiguousarray(curr_accel, dtype=np.float64)
        gv = np.ascontiguousarray(g_vec, dtype=np.float64)
        mats = [_F(Qimu), _F(Qmodel), _F(np.eye(3) if Cg is None else Cg), _F(np.eye(3) if Ca is None else Ca)]
        self.lib.refx_propagate.restype = C.c_int
        rc = self.lib.refx_propagate(C.c_int(0 if method == "RK4" else 1), C.c_int(1 if visual else 0), _p(st), _p(Pf), _p(lg), _p(la), _p(cg),
                                     _p(ca), _p(sg), _p(sa), C.c_longlong(int(dt_ns)), _p(mats[0]), _p(mats[1]), _p(gv), _p(mats[2]),
                                     _p(mats[3]), C.c_double(stepsize))
        if rc != 0:
            raise RuntimeError("the extracted integrators fixed another step size earlier in this process (function-local static)")
        return dict(Rsb=st[0:9].reshape(3, 3).T.copy(), Tsb=st[9:12].copy(), Vsb=st[12:15].copy(), P=np.ascontiguousarray(Pf),
                    last_gyro=lg, last_accel=la, slope_gyro=sg, slope_accel=sa)

    def one_point_ransac(self, X, Rbc, Tbc, P, x, xp, ref_sind, sind, gR, gT, cam, R, ransac_thresh, ransac_chi2, gauge_sind=-1,
                         absorb_groups=None, in_update=None, status=None, ransac_prob=0.99):
        """Estimator::OnePointRANSAC as extracted - the whole function. gR / gT: kMaxGroup poses by slot. Returns
        dict(keep [F] bool, status [F], n_rejected, P, x, J [F, 2, N]) - P / x as the function leaves them (restored)."""
        F = len(sind); N = self.N; G = self.max_group()
        assert len(gR) == G
        st = self._state30(X); Pf = _F(P).copy(order="F")
        xs = np.ascontiguousarray(x, dtype=np.float64).copy().reshape(F, 3); xps = np.ascontiguousarray(xp, dtype=np.float64).reshape(F, 2)
        stt = np.full(F, 3, dtype=np.int32) if status is None else np.ascontiguousarray(status, dtype=np.int32).copy()
        gRf = np.ascontiguousarray(np.stack([_F(R_).reshape(-1, order="F") for R_ in gR])); gTf = np.ascontiguousarray(gT, dtype=np.float64).reshape(G, 3)
        ab = (1 << G) - 1 if absorb_groups is None else int(absorb_groups)
        iu = np.ascontiguousarray(np.zeros(F) if in_update is None else in_update, dtype=np.int32)
        keep = np.zeros(F, dtype=np.int32); nrej = C.c_int(); J = np.zeros((F, 2 * N)); c = _cam(cam)
        Rb = _F(Rbc); Tb = np.ascontiguousarray(Tbc, dtype=np.float64)
        rs = np.ascontiguousarray(ref_sind, dtype=np.int32); si = np.ascontiguousarray(sind, dtype=np.int32)
        self.lib.refx_one_point_ransac.restype = C.c_int
        self.lib.refx_one_point_ransac(C.c_int(F), _p(xs), _p(xps), _p(rs), _p(si), _p(stt), _p(gRf), _p(gTf), _p(st), _p(Rb), _p(Tb), _p(Pf),
                                       C.c_double(R), C.c_double(ransac_thresh), C.c_double(ransac_prob), C.c_double(ransac_chi2),
                                       C.c_int(int(gauge_sind)), C.c_ulonglong(ab), _p(iu), C.byref(c), _p(keep), C.byref(nrej), _p(J))
        return dict(keep=keep.astype(bool), status=stt, n_rejected=nrej.value, P=np.ascontiguousarray(Pf), x=xs,
                    J=np.stack([J[f].reshape(N, 2).T for f in range(F)]))


_REFX = {}


def loadx(N=203):
    """The extracted-text library compiled for state size N (203 or 251), or N = "calib": the default sizes with the
    reference's three online-calibration defines (N = 228), or N = "invdepth": the 8-group / 60-feature sizes (N = 251) with
    -DUSE_INVDEPTH; raises FileNotFoundError if it was never built."""
    if N in _REFX:
        return _REFX[N]
    for v in (("v4", "v3") if _has_avx512() else ("v3",)):
        name = {"calib": "calib", "invdepth": "n251inv"}.get(N, f"n{N}")
        path = os.path.join(_HERE, "_ref", f"libxivo_refx_{name}_{v}.so")
        if os.path.exists(path):
            _REFX[N] = RefX(path)
            return _REFX[N]
    raise FileNotFoundError(f"oracle/_ref/libxivo_refx_n{N}_*.so not built (needs /root/reference; run oracle/ref/Makefile)")


_REF = None


def load():
    """Returns the Ref wrapper; raises FileNotFoundError if oracle/_ref was never built."""
    global _REF
    if _REF is not None:
        return _REF
    v = "v4" if _has_avx512() else "v3"
    path = os.path.join(_HERE, "_ref", f"libxivo_ref_{v}.so")
    if not os.path.exists(path):
        path = os.path.join(_HERE, "_ref", "libxivo_ref_v3.so")
    if not os.path.exists(path):
        raise FileNotFoundError("oracle/_ref not built (needs /root/reference; run oracle/ref/Makefile)")
    _REF = Ref(path)
    _REF.path = path
    return _REF


def build_flags():
    """The exact compiler line of the loaded oracle/_ref variant (oracle/ref/Makefile), for bench.py's cpu_baseline.flags."""
    ref = load()
    march = "x86-64-v4" if ref.path.endswith("_v4.so") else "x86-64-v3"
    flags = "-std=c++17 -O3 -DNDEBUG -DEIGEN_INITIALIZE_MATRICES_BY_ZERO -DSOPHUS_USE_BASIC_LOGGING -fPIC -shared"
    try:
        for line in open(os.path.join(_HERE, "ref", "Makefile")):
            if line.startswith("FLAGS ="):
                flags = line.split("=", 1)[1].strip().rstrip("\\").strip()
    except OSError:
        pass
    return (f"g++ {flags} -march={march} (Eigen 3.3.9 + Sophus from the reference tree; the reference's own build uses "
            "-O3 -march=native -DNDEBUG, CMakeLists.txt:25-26,32 - a native build cannot travel to another host CPU)")
