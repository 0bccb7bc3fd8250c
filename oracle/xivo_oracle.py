"""CPU oracle: a numpy restatement of XIVO's EKF measurement-update hot path.

TEST INFRASTRUCTURE ONLY. Nothing in the product path (xivo_amd/, include/)
may import, call or link this file; only tests/, __graft_entry__.smoke() and
bench.py's `cpu_baseline` leg use it, and only as the checker.

Every function restates one piece of the reference (paths relative to
/root/reference) in plain float64 numpy, keeping the reference's expression
order where that is observable. Pinning (see tests/test_oracle_pinned.py and
tests/golden/):
  * the restatement is validated against oracle/_ref/libxivo_ref_*.so, which is
    compiled from the reference's OWN sources where they lie (Eigen 3.3.9 LDLT /
    LLT / FullPivLU, Sophus, src/helpers.cpp verbatim, common/project.h and the
    common/camera_*.h headers verbatim) plus an expression-faithful driver
    (oracle/ref/xivo_ref.cpp) for the member functions that cannot be compiled
    (they live in Estimator/Feature TUs that need OpenCV);
  * golden vectors generated from that library are committed under
    tests/golden/ (script: tests/golden/make_golden.py);
  * the reference's own known-answer test `GivensSub`
    (src/test/unittest_givens.cpp:15-37) and its finite-difference Jacobian
    checks (src/test/unittest_jacobians_instate.cpp, ..._oos.cpp) are re-run
    against these functions.
The reference has NO test for UpdateJosephForm / MHGating / FilterUpdate /
Propagate; for those the pin is the Eigen-built driver above, not a reference
golden vector (stated in DESIGN.md).
"""
import math
import numpy as np

# ----------------------------------------------------------------------------
# a16: error-state layout (src/core.h:40-105, default build: no online calib)
# ----------------------------------------------------------------------------
WSB, TSB, VSB, BG, BA, WBC, TBC, WSG = 0, 3, 6, 9, 12, 15, 18, 21
K_MOTION = 23


class Layout:
    """kGroupBegin / kFeatureBegin / kFullSize (src/core.h:87-105) as run-time values."""

    def __init__(self, n_groups, n_features, N=None, group_begin=K_MOTION, td=-1, Cg=-1, cam_begin=0, cam_dim=0):
        # online-calibration builds (src/core.h:49-83): slots of td / Cg (Ca follows Cg) / the camera intrinsics, -1 / 0 = absent;
        # calib_layout() below derives them and group_begin the way the reference's Index enum does
        self.td, self.Cg, self.cam_begin, self.cam_dim = td, Cg, cam_begin, cam_dim
        self.Ca = Cg + 9 if Cg >= 0 else -1
        self.motion_size = Cg + 15 if Cg >= 0 else (td + 1 if td >= 0 else K_MOTION)   # Index::End = kMotionSize
        self.group_begin = group_begin
        self.n_groups = n_groups
        self.feature_begin = group_begin + 6 * n_groups
        self.n_features = n_features
        full = self.feature_begin + 3 * n_features
        self.N = full if N is None else N   # N > full: trailing zero slots (estimator.cpp:757-759)
        assert self.N >= full


def calib_layout(n_groups, n_features, temporal=True, imu=True, camera_dim=9):
    """Layout of an online-calibration build exactly as src/core.h:40-105 numbers it: Index::td = Wsg + 2 (temporal),
    Cg = td + 1 (or Wsg + 2), Ca = Cg + 9, End = kMotionSize; kCameraBegin = kMotionSize, kMaxCameraIntrinsics = 9 with
    USE_ONLINE_CAMERA_CALIB (camera_dim = Camera::dim() of the model: the slots in use), kGroupBegin behind them."""
    nxt = WSG + 2
    td = Cg = -1
    if temporal:
        td = nxt; nxt += 1
    if imu:
        Cg = nxt; nxt += 15
    motion = nxt
    cam_begin = motion
    group_begin = motion + (9 if camera_dim > 0 else 0)
    return Layout(n_groups, n_features, group_begin=group_begin, td=td, Cg=Cg, cam_begin=cam_begin, cam_dim=camera_dim)


# ----------------------------------------------------------------------------
# a1: Estimator::UpdateJosephForm (src/estimator.cpp:1257-1288)
# ----------------------------------------------------------------------------
def ldlt_eigen(A):
    """Eigen 3.3.9 `LDLT<MatrixXd, Lower>::compute` = internal::ldlt_inplace<Lower>::unblocked
    (/root/reference/thirdparty/eigen/Eigen/src/Cholesky/LDLT.h:291-404), the factorisation behind
    `S_.ldlt()` (src/estimator.cpp:1266): diagonal pivoting on the largest remaining |diagonal| (first index wins a tie,
    as maxCoeff), lower triangle only, a zero pivot is left undivided (:364-376). Returns (mat, transpositions):
    strict lower part of mat = L (unit diagonal), diagonal of mat = D."""
    mat = np.array(A, dtype=np.float64)
    n = mat.shape[0]
    tr = np.arange(n)
    if n <= 1:
        return mat, tr
    for k in range(n):
        p = k + int(np.argmax(np.abs(np.diag(mat)[k:])))          # :318-320
        tr[k] = p
        if p != k:                                                  # :323-339, lower triangle only
            mat[[k, p], :k] = mat[[p, k], :k]
            mat[p + 1:, [k, p]] = mat[p + 1:, [p, k]]
            mat[k, k], mat[p, p] = mat[p, p], mat[k, k]
            for i in range(k + 1, p):
                mat[i, k], mat[p, i] = mat[p, i], mat[i, k]
        rs = n - k - 1
        if k > 0:                                                   # :350-356
            temp = np.diag(mat)[:k] * mat[k, :k]
            mat[k, k] -= mat[k, :k] @ temp
            if rs > 0:
                mat[k + 1:, k] -= mat[k + 1:, :k] @ temp
        akk = mat[k, k]
        valid = abs(akk) > 0.0                                      # :362-363
        if k == 0 and not valid:                                    # :365-376: the whole diagonal is zero
            return mat, np.arange(n)
        if rs > 0 and valid:
            mat[k + 1:, k] /= akk                                   # :378-379
    return mat, tr


def ldlt_solve_eigen(mat, tr, B):
    """LDLT::_solve_impl (LDLT.h:561-600): x = P^T L^-T D^+ L^-1 P b with the pseudo-inverse of D at Eigen's tolerance
    1 / highest() (a pivot of exactly zero - or denormal - drops its component instead of dividing)."""
    X = np.array(B, dtype=np.float64)
    n = mat.shape[0]
    for k in range(n):                                              # dst = P b
        if tr[k] != k:
            X[[k, tr[k]]] = X[[tr[k], k]]
    L = np.tril(mat, -1) + np.eye(n)
    for i in range(n):                                              # L^-1 (unit lower)
        X[i] -= L[i, :i] @ X[:i]
    d = np.diag(mat)
    tol = 1.0 / np.finfo(np.float64).max
    for i in range(n):                                              # D^+
        X[i] = X[i] / d[i] if abs(d[i]) > tol else 0.0
    for i in range(n - 1, -1, -1):                                  # L^-T
        X[i] -= L[i + 1:, i] @ X[i + 1:]
    for k in range(n - 1, -1, -1):                                  # P^T
        if tr[k] != k:
            X[[k, tr[k]]] = X[[tr[k], k]]
    return X


def update_joseph(H, P, inn, diagR, solver="lu"):
    """Returns (err, P_new, K_scaled). Expression order as the reference:
    S=(H P) H^T; +R; K^T = S^-1 (H P); err = K inn; A = K H - I;
    P = (A P) A^T; K *= sqrt(R) column-wise; P += K K^T.
    solver = "lu" (LAPACK, any non-singular S) or "ldlt" (the restatement of Eigen's pivoted L D L^T above: what the
    reference runs, and the only one of the two that is defined for a singular S - zero pivots drop their component)."""
    H = np.asarray(H, dtype=np.float64)
    P = np.asarray(P, dtype=np.float64)
    S = (H @ P) @ H.T                                   # :1259
    S[np.diag_indices_from(S)] += diagR                 # :1261-1263
    if solver == "ldlt":
        Kt = ldlt_solve_eigen(*ldlt_eigen(S), H @ P)    # :1266, S_.ldlt().solve(H_ * P_) as Eigen evaluates it
    else:
        Kt = np.linalg.solve(S, H @ P)                  # :1266 (Eigen: pivoted LDL^T)
    K = Kt.T
    err = K @ inn                                       # :1267
    A = K @ H                                           # :1276
    A[np.diag_indices_from(A)] -= 1.0                   # :1277-1279
    Pn = (A @ P) @ A.T                                  # :1280
    Ks = K * np.sqrt(diagR)[None, :]                    # :1282-1286
    Pn = Pn + Ks @ Ks.T                                 # :1287
    return err, Pn, Ks


F_ALG = lambda N, M: 4.0 * N ** 3 + 8.0 * M * N ** 2 + 4.0 * M ** 2 * N + M ** 3 / 3.0


# ----------------------------------------------------------------------------
# a6: Estimator::MHGating numeric core (src/update.cpp:60-96)
# ----------------------------------------------------------------------------
def mh_distances(J, P, inn, R):
    """J: [F, 2, N], inn: [F, 2] -> d[F]; S = J P J^T + R I2; d = r^T S.llt().solve(r)."""
    F = J.shape[0]
    d = np.empty(F)
    for i in range(F):
        S = (J[i] @ P) @ J[i].T                          # :65
        S[0, 0] += R                                     # :66
        S[1, 1] += R                                     # :67
        # Eigen LLT reads the lower triangle only
        l00 = math.sqrt(S[0, 0]); l10 = S[1, 0] / l00; l11 = math.sqrt(S[1, 1] - l10 * l10)
        y0 = inn[i, 0] / l00; y1 = (inn[i, 1] - l10 * y0) / l11
        x1 = y1 / l11; x0 = (y0 - l10 * x1) / l00
        d[i] = inn[i, 0] * x0 + inn[i, 1] * x1           # :68
    return d


def mh_gate(dist, thresh, mult, min_inliers):
    """Threshold-relaxation loop (src/update.cpp:72-96). Returns (mask, num_rejected
    accumulated over relaxations - the reference's counting quirk -, final thresh)."""
    F = len(dist)
    if min_inliers <= 0:                                 # loop body never runs
        return np.zeros(F, dtype=bool), 0, -1.0
    inliers = []
    rejected = 0
    used = thresh
    th = thresh
    mask = np.zeros(F, dtype=bool)
    guard = 0
    while len(inliers) < min_inliers:                    # :73
        inliers = []
        mask[:] = False
        for i in range(F):
            if dist[i] < th:                             # :84
                inliers.append(i); mask[i] = True
            else:
                rejected += 1                            # :87
        used = th
        th *= mult                                       # :94
        guard += 1
        if len(inliers) == F or guard > 4096:
            break
    return mask, rejected, used


# ----------------------------------------------------------------------------
# SO3 helpers (thirdparty/sophus/sophus/so3.hpp hat / exp)
# ----------------------------------------------------------------------------
def hat(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def so3_exp(w):
    """Rodrigues formula (matches Sophus::SO3::exp to rounding; used for test scene
    generation and ComposeMotion, never for parity-critical values)."""
    th = np.linalg.norm(w)
    Wm = hat(w)
    if th < 1e-10:
        return np.eye(3) + Wm + 0.5 * Wm @ Wm
    return np.eye(3) + math.sin(th) / th * Wm + (1 - math.cos(th)) / (th * th) * Wm @ Wm


# ----------------------------------------------------------------------------
# cameras (common/camera_pinhole.h:17-37, camera_equidist.h:23-95,
# camera_radtan.h:23-90, camera_atan.h:22-67) - Project(xc, &jac)
# ----------------------------------------------------------------------------
CAM_PINHOLE, CAM_ATAN, CAM_RADTAN, CAM_EQUI = 0, 1, 2, 3


def camera_project(cam, xc):
    fx, fy, cx, cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    x, y = float(xc[0]), float(xc[1])
    model = cam["model"]
    d = cam.get("d", [])
    if model == CAM_PINHOLE:
        return np.array([fx * x + cx, fy * y + cy]), np.array([[fx, 0.0], [0.0, fy]])
    if model == CAM_EQUI:
        k0, k1, k2, k3 = d[:4]
        n2 = x * x + y * y; n = math.sqrt(n2); n3 = n2 + 1
        th = math.atan2(n, 1.0); phi = math.atan2(y, x)
        th2 = th * th; th3 = th2 * th; th4 = th3 * th; th5 = th3 * th2; th6 = th5 * th
        th7 = th5 * th2; th8 = th7 * th; th9 = th7 * th2
        r = th + k0 * th3 + k1 * th5 + k2 * th7 + k3 * th9
        c, s = math.cos(phi), math.sin(phi)
        xp = np.array([fx * r * c + cx, fy * r * s + cy])
        dphi_dx, dphi_dy = -y / n2, x / n2
        dth_dx, dth_dy = x / n3 / n, y / n3 / n
        dr = 1 + k0 * 3 * th2 + k1 * 5 * th4 + k2 * 7 * th6 + k3 * 9 * th8
        J = np.array([[fx * c * dr * dth_dx - fx * r * s * dphi_dx, fx * c * dr * dth_dy - fx * r * s * dphi_dy],
                      [fy * s * dr * dth_dx + fy * r * c * dphi_dx, fy * s * dr * dth_dy + fy * r * c * dphi_dy]])
        return xp, J
    if model == CAM_RADTAN:
        p1, p2, k1, k2, k3 = d[:5]
        t2 = x * x; t3 = y * y; t4 = k1 * x * 2.0; t5 = k1 * y * 2.0; t6 = p1 * x * 2.0; t7 = p2 * y * 2.0
        t8 = t2 * 3.0; t9 = t3 * 3.0; t10 = t6 * y; t11 = t7 * x; t12 = t2 + t3; t13 = t3 + t8; t14 = t2 + t9
        t15 = t12 * t12; t16 = t12 * t12 * t12; t17 = k1 * t12
        t22 = k2 * t12 * x * 4.0; t23 = k2 * t12 * y * 4.0; t18 = k2 * t15; t19 = k3 * t16
        t20 = p1 * t14; t21 = p2 * t13; t24 = k3 * t15 * x * 6.0; t25 = k3 * t15 * y * 6.0
        t26 = t4 + t22 + t24; t27 = t5 + t23 + t25; t28 = t17 + t18 + t19 + 1.0
        t29 = t28 * x; t30 = t28 * y; t31 = t10 + t21 + t29; t32 = t11 + t20 + t30
        xp = np.array([cx + fx * t31, cy + fy * t32])
        J = np.array([[fx * (t28 + p2 * x * 6.0 + p1 * y * 2.0 + t26 * x), fx * (t6 + t7 + t27 * x)],
                      [fy * (t6 + t7 + t26 * y), fy * (t28 + p2 * x * 2.0 + p1 * y * 6.0 + t27 * y)]])
        return xp, J
    if model == CAM_ATAN:
        w = d[0]
        invw = 1.0 / w; w2 = 2.0 * math.tan(w * 0.5)
        R = math.sqrt(x * x + y * y)
        singular = R < 0.0001 or w == 0
        f = 1.0 if singular else invw * math.atan(w2 * R) / R
        xp = np.array([fx * f * x + cx, fy * f * y + cy])
        if singular:
            J = np.array([[fx, 0.0], [0.0, fy]])
        else:
            a = w2 * R
            df_dR = invw * (1.0 / (1 + a * a) * a - math.atan(a)) / R / R
            dfx, dfy = df_dR * x / R, df_dR * y / R
            J = np.array([[fx * f + fx * x * dfx, fx * x * dfy], [fy * y * dfx, fy * f + fy * y * dfy]])
        return xp, J
    raise ValueError("unknown camera model")


def camera_project_jacc(cam, xc):
    """d(xp)/d(intrinsics) of the USE_ONLINE_CAMERA_CALIB builds, [2, dim] in each model's own parameter order:
    pinhole fx fy cx cy (camera_pinhole.h:31-35); atan + w (camera_atan.h:62-91); radtan + p1 p2 k1 k2 k3
    (camera_radtan.h:78-96); equidistant + k0..k3 (camera_equidist.h:80-94)."""
    fx, fy = cam["fx"], cam["fy"]
    x, y = float(xc[0]), float(xc[1])
    model = cam["model"]
    d = cam.get("d", [])
    if model == CAM_PINHOLE:
        return np.array([[x, 0, 1, 0], [0, y, 0, 1.0]])
    if model == CAM_EQUI:
        k0, k1, k2, k3 = d[:4]
        n = math.sqrt(x * x + y * y)
        th = math.atan2(n, 1.0); phi = math.atan2(y, x)
        th2 = th * th; th3 = th2 * th; th5 = th3 * th2; th7 = th5 * th2; th9 = th7 * th2
        r = th + k0 * th3 + k1 * th5 + k2 * th7 + k3 * th9
        c, s_ = math.cos(phi), math.sin(phi)
        J = np.zeros((2, 8))
        J[0, 0] = r * c; J[0, 2] = 1; J[1, 1] = r * s_; J[1, 3] = 1
        dr_dk = np.array([th3, th5, th7, th9])
        J[0, 4:] = fx * c * dr_dk; J[1, 4:] = fy * s_ * dr_dk
        return J
    if model == CAM_RADTAN:
        p1, p2, k1, k2, k3 = d[:5]
        t2 = x * x; t3 = y * y; t6 = p1 * x * 2.0; t7 = p2 * y * 2.0; t8 = t2 * 3.0; t9 = t3 * 3.0
        t10 = t6 * y; t11 = t7 * x; t12 = t2 + t3; t13 = t3 + t8; t14 = t2 + t9
        t15 = t12 * t12; t16 = t12 * t12 * t12; t17 = k1 * t12; t18 = k2 * t15; t19 = k3 * t16
        t20 = p1 * t14; t21 = p2 * t13; t28 = t17 + t18 + t19 + 1.0
        t29 = t28 * x; t30 = t28 * y; t31 = t10 + t21 + t29; t32 = t11 + t20 + t30
        J = np.zeros((2, 9))
        J[0, 0] = t31; J[0, 2] = 1.0; J[0, 4] = fx * x * y * 2.0; J[0, 5] = fx * t13
        J[0, 6] = fx * t12 * x; J[0, 7] = fx * t15 * x; J[0, 8] = fx * t16 * x
        J[1, 1] = t32; J[1, 3] = 1.0; J[1, 4] = fy * t14; J[1, 5] = fy * x * y * 2.0
        J[1, 6] = fy * t12 * y; J[1, 7] = fy * t15 * y; J[1, 8] = fy * t16 * y
        return J
    if model == CAM_ATAN:
        w = d[0]
        invw = 1.0 / w; w2 = 2.0 * math.tan(w * 0.5)
        R = math.sqrt(x * x + y * y)
        J = np.zeros((2, 5))
        if R < 0.0001 or w == 0:
            J[0, 0] = x; J[0, 2] = 1; J[1, 1] = y; J[1, 3] = 1
            return J
        f = invw * math.atan(w2 * R) / R
        J[0, 0] = f * x; J[0, 2] = 1; J[1, 1] = f * y; J[1, 3] = 1
        df_dinvw = math.atan(w2 * R) / R; dinvw_dw = -invw * invw
        df_datan = invw / R; datan_dw2R = 1 / (1 + (w2 * R) * (w2 * R)); dw2R_dw2 = R
        dw2_dw = (1 / math.cos(w * 0.5)) ** 2
        df_dw = df_dinvw * dinvw_dw + df_datan * datan_dw2R * dw2R_dw2 * dw2_dw
        J[0, 4] = fx * x * df_dw; J[1, 4] = fy * y * df_dw
        return J
    raise ValueError("unknown camera model")


def project(Xc):
    """common/project.h:11-24"""
    X, Y, Z = Xc
    return np.array([X / Z, Y / Z]), np.array([[1 / Z, 0, -X / (Z * Z)], [0, 1 / Z, -Y / (Z * Z)]])


def unproject_logz(x):
    """common/project.h:79-95"""
    z = math.exp(x[2])
    return np.array([x[0] * z, x[1] * z, z]), np.array([[z, 0, x[0] * z], [0, z, x[1] * z], [0, 0, z]])


def unproject_invz(x):
    """common/project.h:52-56: unproject_invz = project_invz (:31-46) applied to (X/Z, Y/Z, 1/Z)"""
    r = x[2]
    return (np.array([x[0] / r, x[1] / r, 1.0 / r]),
            np.array([[1 / r, 0, -x[0] / (r * r)], [0, 1 / r, -x[1] / (r * r)], [0, 0, -1 / (r * r)]]))


def feature_xc(x, invdepth=False):
    """Feature::Xc (src/feature.cpp:98-105): unproject_logz, or unproject_invz in the USE_INVDEPTH build"""
    return unproject_invz(x) if invdepth else unproject_logz(x)


def feature_z(x, invdepth=False):
    """Feature::z (src/feature.cpp:120-126)"""
    return 1.0 / x[2] if invdepth else math.exp(x[2])


def feature_xs(x, Rsbr, Tsbr, Rbc, Tbc, invdepth=False):
    """Feature::Xs(gbc) (src/feature.cpp:107-118): gsc = ref_->gsb() * gbc; Xs = gsc * Xc"""
    Xc, _ = feature_xc(x, invdepth)
    return (Rsbr @ Rbc) @ Xc + (Rsbr @ Tbc + Tsbr)


# ----------------------------------------------------------------------------
# a9b: Feature::ComputeLCJacobian (src/oos.cpp:92-145) under the stacking of
# Estimator::CloseLoopInternal (src/update.cpp:183-196)
# ----------------------------------------------------------------------------
def lc_jacobian_rows(matches, Rbc, Tbc, cam, layout, invdepth=False):
    """matches = [dict(x, Rsbr, Tsbr: the OLD feature's state and anchor-group pose; Rsb, Tsb, g_sind: the observing group;
    xp: the observed pixel)]. Returns (H [2n, N], inn [2n]) - H_.setZero(2n, N) then one ComputeLCJacobian per match."""
    n = len(matches)
    H = np.zeros((2 * n, layout.N)); inn = np.zeros(2 * n)          # update.cpp:184-186
    Rbc_t = Rbc.T
    for i, m in enumerate(matches):
        Rsb_t = np.asarray(m["Rsb"]).T
        goff = layout.group_begin + 6 * int(m["g_sind"])            # :98
        Xb = Rsb_t @ (feature_xs(m["x"], m["Rsbr"], m["Tsbr"], Rbc, Tbc, invdepth) - m["Tsb"])   # :107
        dXb_dTsb = -Rsb_t; dXb_dWsb = hat(Xb)                      # :109-110
        Xcn = Rbc_t @ (Xb - Tbc)                                    # :113
        dXcn_dXb = Rbc_t; dXcn_dTbc = -Rbc_t; dXcn_dWbc = hat(Xcn)  # :114-116
        dXcn_dTsb = dXcn_dXb @ dXb_dTsb                             # :120
        dXcn_dWsb = dXcn_dXb @ dXb_dWsb                             # :121
        xcn, dxcn_dXcn = project(Xcn)                               # :123
        xp, dxp_dxcn = camera_project(cam, xcn)                     # :125-130
        dxp_dXcn = dxp_dxcn @ dxcn_dXcn                             # :132
        st = 2 * i
        H[st:st + 2, goff:goff + 3] = dxp_dXcn @ dXcn_dWsb          # :135
        H[st:st + 2, goff + 3:goff + 6] = dxp_dXcn @ dXcn_dTsb      # :136
        H[st:st + 2, WBC:WBC + 3] = dxp_dXcn @ dXcn_dWbc            # :137
        H[st:st + 2, TBC:TBC + 3] = dxp_dXcn @ dXcn_dTbc            # :138
        if getattr(layout, "cam_dim", 0) > 0:                       # :140-143 (USE_ONLINE_CAMERA_CALIB)
            jacc = camera_project_jacc(cam, xcn)
            H[st:st + 2, layout.cam_begin:layout.cam_begin + layout.cam_dim] = jacc[:, :layout.cam_dim]
        inn[st:st + 2] = np.asarray(m["xp"], dtype=np.float64) - xp  # :145
    return H, inn


# ----------------------------------------------------------------------------
# a4: Feature::ComputeJacobian (src/feature.cpp:542-656)
# ----------------------------------------------------------------------------
def compute_jacobian(x, xp_meas, Rsbr, Tsbr, Rsb, Tsb, Rbc, Tbc, cam, layout, ref_sind, sind,
                     return_cache=False, calib=None, invdepth=False):
    """Returns (J [2 x N], inn [2], blocks [7,2,3]) for one in-state feature.
    calib = dict(gyro, Cg, bg, Vsb, td) with a layout from calib_layout(): the online-calibration builds' blocks as well
    (src/feature.cpp:592-609, :611-618, :632-651); then also returns Jc [2, 22] = [td | Cg 9 | bg 3 | intrinsics 9] last."""
    Rsb_t, Rbc_t = Rsb.T, Rbc.T
    Xc, dXc_dx = feature_xc(x, invdepth)                 # :555 (Xc(&cache_.dXc_dx), feature.cpp:98-105)
    Xbr = Rbc @ Xc + Tbc                                 # :556
    Xs = Rsbr @ Xbr + Tsbr                               # :557
    Xb = Rsb_t @ (Xs - Tsb)                              # :558
    Xcn = Rbc_t @ (Xb - Tbc)                             # :559
    dXbr_dXc = Rbc                                       # :562
    dXbr_dWbc = -Rbc @ hat(Xc)                           # :564
    dXs_dXbr = Rsbr                                      # :567
    dXs_dWsbr = -Rsbr @ hat(Xbr)                         # :569
    dXb_dXs = Rsb_t                                      # :572
    dXb_dTsb = -Rsb_t                                    # :573
    dXb_dWsb = hat(Xb)                                   # :574
    dXcn_dXb = Rbc_t                                     # :577
    dXcn_dTbc = -Rbc_t + dXcn_dXb @ dXb_dXs @ dXs_dXbr @ np.eye(3)          # :578-579
    dXcn_dWbc = hat(Xcn) + dXcn_dXb @ dXb_dXs @ dXs_dXbr @ dXbr_dWbc        # :580-581
    dXcn_dTsb = dXcn_dXb @ dXb_dTsb                      # :584
    dXcn_dWsb = dXcn_dXb @ dXb_dWsb                      # :585
    dXcn_dTsbr = dXcn_dXb @ dXb_dXs @ np.eye(3)          # :586
    dXcn_dWsbr = dXcn_dXb @ dXb_dXs @ dXs_dWsbr          # :587
    dXcn_dXs = dXcn_dXb @ dXb_dXs                        # :588
    dXcn_dx = dXcn_dXs @ dXs_dXbr @ dXbr_dXc @ dXc_dx    # :589
    xcn, dxcn_dXcn = project(Xcn)                        # :611
    xp, dxp_dxcn = camera_project(cam, xcn)              # :617
    dxp_dXcn = dxp_dxcn @ dxcn_dXcn                      # :620
    blocks = np.stack([dxp_dXcn @ dXcn_dWsb, dxp_dXcn @ dXcn_dTsb, dxp_dXcn @ dXcn_dWbc,
                       dxp_dXcn @ dXcn_dTbc, dxp_dXcn @ dXcn_dWsbr, dxp_dXcn @ dXcn_dTsbr,
                       dxp_dXcn @ dXcn_dx])              # :623-645
    goff = layout.group_begin + 6 * ref_sind             # :639
    foff = layout.feature_begin + 3 * sind               # :640
    J = np.zeros((2, layout.N))
    for b, off in enumerate([WSB, TSB, WBC, TBC, goff, goff + 3, foff]):
        J[:, off:off + 3] = blocks[b]
    inn = np.asarray(xp_meas, dtype=np.float64) - xp     # :654
    if calib is not None:
        Jc = np.zeros((2, 22))
        if layout.td >= 0:
            gyro = np.asarray(calib["gyro"], float); Cg = np.asarray(calib["Cg"], float)
            gyro_calib = Cg @ gyro - np.asarray(calib["bg"], float)                             # :593
            dXcn_dtd = -Rbc_t @ (hat(gyro_calib) @ Rsb_t @ (Xs - Tsb) + Rsb_t @ np.asarray(calib["Vsb"], float))   # :594-595
            dXcn_dW = Rbc_t @ hat(Rsb_t @ (Xs - Tsb)) * calib["td"]                             # :598-599 (dAB_dB<3,1>(A) = A)
            J[:, layout.td] = dxp_dXcn @ dXcn_dtd                                               # :632
            Jc[:, 0] = J[:, layout.td]
            if layout.Cg >= 0:
                dW_dCg = np.zeros((3, 9))
                for i in range(3):
                    dW_dCg[i, 3 * i:3 * i + 3] = gyro                                           # :601-604
                J[:, layout.Cg:layout.Cg + 9] = dxp_dXcn @ (dXcn_dW @ dW_dCg)                   # :605, :634
                Jc[:, 1:10] = J[:, layout.Cg:layout.Cg + 9]
            J[:, BG:BG + 3] = dxp_dXcn @ (-dXcn_dW)                                             # :607, :636
            Jc[:, 10:13] = J[:, BG:BG + 3]
        if layout.cam_dim > 0:
            jacc = camera_project_jacc(cam, xcn)                                                # :611-614
            dim = layout.cam_dim
            J[:, layout.cam_begin:layout.cam_begin + dim] = jacc[:, :dim]                       # :647-651
            Jc[:, 13:13 + dim] = jacc[:, :dim]
        return J, inn, blocks, Jc
    if return_cache:
        cache = dict(Xc=Xc, Xbr=Xbr, Xs=Xs, Xb=Xb, Xcn=Xcn, dXcn_dWsb=dXcn_dWsb, dXcn_dTsb=dXcn_dTsb,
                     dXcn_dWbc=dXcn_dWbc, dXcn_dTbc=dXcn_dTbc, dXcn_dWsbr=dXcn_dWsbr,
                     dXcn_dTsbr=dXcn_dTsbr, dXcn_dx=dXcn_dx, dXcn_dXs=dXcn_dXs, xp=xp)
        return J, inn, blocks, cache
    return J, inn, blocks


# ----------------------------------------------------------------------------
# a3: Feature::FillJacobianBlock (src/feature.cpp:658-684), a2: FilterUpdate
# stacking (src/update.cpp:129-138)
# ----------------------------------------------------------------------------
def fill_jacobian_block(H, row, J, layout, ref_sind, sind, fix_group_block=False):
    for off in (WSB, TSB, WBC, TBC):                     # :659-662
        H[row:row + 2, off:off + 3] = J[:, off:off + 3]
    goff = layout.group_begin + 6 * ref_sind             # :672
    foff = layout.feature_begin + 3 * sind               # :673
    H[row:row + 2, goff:goff + 3] = J[:, goff:goff + 3]              # :675
    if fix_group_block:
        H[row:row + 2, goff + 3:goff + 6] = J[:, goff + 3:goff + 6]
    else:
        H[row:row + 2, goff:goff + 3] = J[:, goff + 3:goff + 6]      # :676 (the quirk)
    H[row:row + 2, foff:foff + 3] = J[:, foff:foff + 3]              # :677
    if getattr(layout, "td", -1) >= 0:                               # :664-670 (USE_ONLINE_TEMPORAL_CALIB [+ _IMU_CALIB])
        H[row:row + 2, layout.td] = J[:, layout.td]
        if layout.Cg >= 0:
            H[row:row + 2, layout.Cg:layout.Cg + 9] = J[:, layout.Cg:layout.Cg + 9]
        H[row:row + 2, BG:BG + 3] = J[:, BG:BG + 3]
    if getattr(layout, "cam_dim", 0) > 0:                            # :679-683 (USE_ONLINE_CAMERA_CALIB)
        cb, dim = layout.cam_begin, layout.cam_dim
        H[row:row + 2, cb:cb + dim] = J[:, cb:cb + dim]


def stack_measurements(Js, inns, ref_sinds, sinds, layout, R, fix_group_block=False):
    """Rows for the features given, in the order given (update.cpp:129-138)."""
    F = len(Js)
    H = np.zeros((2 * F, layout.N))                      # :130
    inn = np.zeros(2 * F)
    diagR = np.empty(2 * F)
    for i in range(F):
        fill_jacobian_block(H, 2 * i, Js[i], layout, ref_sinds[i], sinds[i], fix_group_block)   # :135
        inn[2 * i:2 * i + 2] = inns[i]                   # :136
        diagR[2 * i:2 * i + 2] = R                       # :137
    return H, inn, diagR


def neutralise_rows(H, inn, diagR, mask_rows):
    """The device keeps rejected features' rows in place but neutral (H row 0,
    inn 0, diagR 1); algebraically identical to not stacking them."""
    H = H.copy(); inn = inn.copy(); diagR = diagR.copy()
    H[~mask_rows] = 0.0; inn[~mask_rows] = 0.0; diagR[~mask_rows] = 1.0
    return H, inn, diagR


# ----------------------------------------------------------------------------
# Eigen::FullPivLU<MatrixXd>::kernel()
# (thirdparty/eigen/Eigen/src/LU/FullPivLU.h:490-580 computeInPlace, :619-699 kernel)
# ----------------------------------------------------------------------------
def fullpivlu_kernel(Ain):
    lu = np.array(Ain, dtype=np.float64, order="F")
    rows, cols = lu.shape
    size = min(rows, cols)
    colsT = list(range(size))
    nonzero = size
    maxpivot = 0.0
    for k in range(size):
        corner = np.abs(lu[k:, k:])
        # Eigen's visitor walks column-major and keeps the first maximum
        flat = corner.flatten(order="F")
        idx = int(np.argmax(flat))
        big = flat[idx]
        br, bc = idx % (rows - k) + k, idx // (rows - k) + k
        if big == 0.0:
            nonzero = k
            for i in range(k, size):
                colsT[i] = i
            break
        maxpivot = max(maxpivot, big)
        colsT[k] = bc
        if k != br:
            lu[[k, br], :] = lu[[br, k], :]
        if k != bc:
            lu[:, [k, bc]] = lu[:, [bc, k]]
        if k < rows - 1:
            lu[k + 1:, k] /= lu[k, k]
        if k < size - 1:
            lu[k + 1:, k + 1:] -= np.outer(lu[k + 1:, k], lu[k, k + 1:])
    q = list(range(cols))
    for k in range(size):
        q[k], q[colsT[k]] = q[colsT[k]], q[k]
    thr = maxpivot * (np.finfo(np.float64).eps * size)
    piv = [i for i in range(nonzero) if abs(lu[i, i]) > thr]
    rank = len(piv)
    dimker = cols - rank
    if dimker == 0:
        return np.zeros((cols, 1)), rank
    m = np.zeros((rank, cols))
    for i in range(rank):
        m[i, i:] = lu[piv[i], i:]
    for i in range(rank):
        m[i, :i] = 0.0
    for i in range(rank):
        if piv[i] != i:
            m[:, [i, piv[i]]] = m[:, [piv[i], i]]
    for c in range(rank, cols):
        for i in range(rank - 1, -1, -1):
            s = m[i, c] - m[i, i + 1:rank] @ m[i + 1:rank, c]
            m[i, c] = s / m[i, i]
    for i in range(rank - 1, -1, -1):
        if piv[i] != i:
            m[:, [i, piv[i]]] = m[:, [piv[i], i]]
    ker = np.zeros((cols, dimker))
    for i in range(rank):
        ker[q[i], :] = -m[i, rank:]
    for k in range(dimker):
        ker[q[rank + k], k] = 1.0
    return ker, rank


def slow_givens(Hf, Hx):
    """src/helpers.cpp:13-23: A = FullPivLU(Hf^T).kernel(); Hx <- A^T Hx."""
    A, _ = fullpivlu_kernel(Hf.T)
    return A.T @ Hx, A


# a10: givens / Givens (src/helpers.cpp:27-75, G&VL Alg. 5.1.3, eps guard common/alias.h:80)
EPS_ALIAS = float(np.float32(1e-4))


def givens(a, b):
    if abs(b) < EPS_ALIAS:
        c, s = 1.0, 0.0
    elif abs(b) > abs(a):
        t = -a / b; s = 1 / math.sqrt(1 + t * t); c = s * t
    else:
        t = -b / a; c = 1 / math.sqrt(1 + t * t); s = c * t
    return np.array([[c, s], [-s, c]])


def givens_eliminate(x, Hx, Hf, effective_rows=-1):
    x = x.copy(); Hx = Hx.copy(); Hf = Hf.copy()
    rows = Hf.shape[0] if effective_rows == -1 else effective_rows
    cols = Hf.shape[1]
    for c in range(cols):
        for r in range(rows - 2, c - 1, -1):
            Gt = givens(Hf[r, c], Hf[r + 1, c]).T
            Hf[r:r + 2, :] = Gt @ Hf[r:r + 2, :]
            # NB the reference rotates only the first `cols` columns of Hx (helpers.cpp:64)
            Hx[r:r + 2, :cols] = Gt @ Hx[r:r + 2, :cols]
            x[r:r + 2] = Gt @ x[r:r + 2]
    for r in range(rows - cols):
        x[r] = x[r + cols]; Hx[r] = Hx[r + cols]; Hf[r] = Hf[r + cols]
    return rows - cols, x, Hx, Hf


def qr_compress(x, Hx, effective_rows=-1):
    """xivo::QR (src/helpers.cpp:78-101): Givens triangularisation of the first `rows` rows of Hx (all columns
    rotated, unlike Givens above), residual rotated along; returns (rows, x, Hx) - the caller keeps the top block."""
    x = x.copy(); Hx = Hx.copy()
    rows = Hx.shape[0] if effective_rows == -1 else effective_rows
    cols = Hx.shape[1]
    for c in range(cols):
        for r in range(rows - 2, c - 1, -1):
            Gt = givens(Hx[r, c], Hx[r + 1, c]).T
            Hx[r:r + 2, :] = Gt @ Hx[r:r + 2, :]
            x[r:r + 2] = Gt @ x[r:r + 2]
    return rows, x, Hx


# ----------------------------------------------------------------------------
# a8/a9: Feature::ComputeOOSJacobian(+Internal) (src/oos.cpp:8-89)
# ----------------------------------------------------------------------------
def oos_jacobian_internal(Xs, Rsb, Tsb, Rbc, Tbc, xp_obs, cam, layout, g_sind):
    """One observation: returns (Hf 2x3, Hx 2xN, inn 2) (src/oos.cpp:39-89)."""
    Rsb_t, Rbc_t = Rsb.T, Rbc.T
    Xb = Rsb_t @ (Xs - Tsb)                              # :52
    dXb_dXs = Rsb_t; dXb_dTsb = -Rsb_t; dXb_dWsb = hat(Xb)
    Xcn = Rbc_t @ (Xb - Tbc)                             # :58
    dXcn_dXb = Rbc_t; dXcn_dWbc = hat(Xcn); dXcn_dTbc = -Rbc_t
    xcn, dxcn_dXcn = project(Xcn)                        # :66
    xp, dxp_dxcn = camera_project(cam, xcn)              # :68
    dxp_dXcn = dxp_dxcn @ dxcn_dXcn                      # :70
    inn = np.asarray(xp_obs, dtype=np.float64) - xp      # :72
    Hf = dxp_dXcn @ dXcn_dXb @ dXb_dXs                   # :74-75
    Hx = np.zeros((2, layout.N))                         # :77
    goff = layout.group_begin + 6 * g_sind
    Hx[:, goff:goff + 3] = dxp_dXcn @ dXcn_dXb @ dXb_dWsb        # :78-79
    Hx[:, goff + 3:goff + 6] = dxp_dXcn @ dXcn_dXb @ dXb_dTsb    # :80-81
    Hx[:, WBC:WBC + 3] = dxp_dXcn @ dXcn_dWbc            # :82-83
    Hx[:, TBC:TBC + 3] = dxp_dXcn @ dXcn_dTbc            # :84-85
    return Hf, Hx, inn


def oos_jacobian(Xs, obs, groups_R, groups_T, Rbc, Tbc, cam, layout, whole_buffer_groups=0):
    """ComputeOOSJacobian on the 2k live rows (the reference passes the
    whole 2*kMaxGroup buffer, oos.cpp:28 - dead code there; this repo and the
    oracle project only the live rows). obs = [(g_sind, xp), ...].
    Returns (Hx' [(2k-rank) x N], inn' [(2k-rank)], A).
    whole_buffer_groups = kMaxGroup > 0: src/oos.cpp:28 AS CODED - SlowGivens gets the whole 2 kMaxGroup-row buffers
    (src/jac.h:12-16 resizes them and nothing clears them: the rows behind 2k are taken as zero here), so
    2 kMaxGroup - rank rows come back."""
    k = len(obs)
    rows = 2 * whole_buffer_groups if whole_buffer_groups else 2 * k
    assert rows >= 2 * k
    Hf = np.zeros((rows, 3)); Hx = np.zeros((rows, layout.N)); r = np.zeros(rows)
    for c, (g, xp) in enumerate(obs):
        hf, hx, inn = oos_jacobian_internal(Xs, groups_R[g], groups_T[g], Rbc, Tbc, xp, cam, layout, g)
        Hf[2 * c:2 * c + 2] = hf; Hx[2 * c:2 * c + 2] = hx; r[2 * c:2 * c + 2] = inn
    Hxp, A = slow_givens(Hf, Hx)                         # :27-28
    rp = A.T @ r                                         # :29
    return Hxp, rp, A


# ----------------------------------------------------------------------------
# a12/a14: covariance part of RK4Step (src/rk4.cpp:35-103) and
# ComputeMotionJacobianAt (src/estimator.cpp:615-704), default build
# ----------------------------------------------------------------------------
def motion_jacobian(Rsb, bg, ba, gyro, accel, g_vec, Cg=None, Ca=None, layout=None):
    """layout (online-calibration builds): F, G get layout.motion_size rows and, under USE_ONLINE_IMU_CALIB, the blocks
    dWsb/dCg and dVsb/dCa of src/estimator.cpp:626-638, :674-688."""
    Cg = np.eye(3) if Cg is None else Cg
    Ca = np.eye(3) if Ca is None else Ca
    nm = K_MOTION if layout is None else layout.motion_size
    gyro_calib = Cg @ gyro - bg
    accel_calib = Ca @ accel - ba
    F = np.zeros((nm, nm)); G = np.zeros((nm, 12))
    dW_dW = -hat(gyro_calib)
    dV_dW = -Rsb @ hat(accel_calib)
    dV_dba = -Rsb
    dV_dWsg = -Rsb @ hat(g_vec)
    for j in range(3):
        F[WSB + j, BG + j] = -1; F[TSB + j, VSB + j] = 1
        for i in range(3):
            F[WSB + i, WSB + j] = dW_dW[i, j]
            F[VSB + i, WSB + j] = dV_dW[i, j]
            F[VSB + i, BA + j] = dV_dba[i, j]
            if j < 2:
                F[VSB + i, WSG + j] = dV_dWsg[i, j]
    if layout is not None and layout.Cg >= 0:
        # :626-631 dWsb_dCg: row i carries the RAW gyro sample at columns 3 i .. 3 i + 2
        dWsb_dCg = np.zeros((3, 9))
        for i in range(3):
            dWsb_dCg[i, 3 * i:3 * i + 3] = gyro
        # :633-636 dV_dCa = dAB_dA<3,3>(accel) * dAB_dB<3,3>(Rsb) * dA_dAu<3>() with the index conventions of
        # common/rodrigues.h:143-165 (D(p N + n, n M + m) += B(m, p)), :208-227 (D(p N + n, m P + p) = A(n, m)), :26-37
        dV_dRCa = np.zeros((3, 9))
        for n in range(3):
            for m in range(3):
                dV_dRCa[0 * 3 + n, n * 3 + m] += accel[m]
        dRCa_dCafm = np.zeros((9, 9))
        for n in range(3):
            for pp in range(3):
                for m in range(3):
                    dRCa_dCafm[pp * 3 + n, m * 3 + pp] = Rsb[n, m]
        dCafm_dCa = np.zeros((9, 6))
        idx = 0
        for i in range(3):
            for j in range(i, 3):
                dCafm_dCa[i * 3 + j, idx] = 1; idx += 1
        dV_dCa = dV_dRCa @ dRCa_dCafm @ dCafm_dCa
        F[WSB:WSB + 3, layout.Cg:layout.Cg + 9] = dWsb_dCg          # :674-679
        F[VSB:VSB + 3, layout.Ca:layout.Ca + 6] = dV_dCa            # :680-684
    for j in range(3):
        G[WSB + j, j] = -1; G[BG + j, 6 + j] = 1; G[BA + j, 9 + j] = 1
        for i in range(3):
            G[VSB + i, 3 + j] = -Rsb[i, j]
    return F, G


def rk4_cov_tail(P, FK, PK, dt, Qmodel=None):
    """P_mm += PK dt; Phi = I + FK dt; P_ms <- Phi P_ms; P_sm <- P_sm Phi^T
    (src/rk4.cpp:92-102); optionally P_mm += Qmodel (src/estimator.cpp:590)."""
    P = P.copy()
    nm = FK.shape[0]
    Phi = np.eye(nm) + FK * dt
    P[:nm, :nm] = P[:nm, :nm] + PK * dt
    P[:nm, nm:] = Phi @ P[:nm, nm:]
    P[nm:, :nm] = P[nm:, :nm] @ Phi.T
    if Qmodel is not None:
        P[:nm, :nm] += Qmodel
    return P, Phi


def rk4_step_cov(P, Fs, Gs, Qimu, dt):
    """Covariance/transition part of RK4Step given the 4 stage Jacobians
    (F_i, G_i) evaluated along the mean trajectory (src/rk4.cpp:49-96)."""
    h = 0.5 * dt
    nm = Fs[0].shape[0]
    P0 = P[:nm, :nm]
    q = lambda F, G, Pm: F @ Pm + Pm @ F.T + G @ Qimu @ G.T
    FK1 = Fs[0]; PK1 = q(Fs[0], Gs[0], P0)
    FK2 = Fs[1] + Fs[1] @ FK1 * h; PK2 = q(Fs[1], Gs[1], P0 + h * PK1)
    FK3 = Fs[2] + Fs[2] @ FK2 * h; PK3 = q(Fs[2], Gs[2], P0 + h * PK2)
    FK4 = Fs[3] + Fs[3] @ FK3 * dt; PK4 = q(Fs[3], Gs[3], P0 + dt * PK3)
    FK = (FK1 + 2.0 * (FK2 + FK3) + FK4) / 6.0
    PK = (PK1 + 2.0 * (PK2 + PK3) + PK4) / 6.0
    return FK, PK


# ----------------------------------------------------------------------------
# a17: host edits of P_ (src/estimator.cpp:754-846, 1382-1389, 1474-1478;
# src/feature.cpp:753-760)
# ----------------------------------------------------------------------------
def p_zero_rc(P, off, n):
    P = P.copy(); P[off:off + n, :] = 0; P[:, off:off + n] = 0; return P


def p_copy_rc(P, dst, src, n):
    P = P.copy()
    P[dst:dst + n, :] = P[src:src + n, :]               # rows first (estimator.cpp:808-809)
    P[:, dst:dst + n] = P[:, src:src + n]               # then columns (:810-811)
    return P


def p_set_block3(P, off, P3):
    P = P.copy(); P[off:off + 3, off:off + 3] = P3; return P


# ----------------------------------------------------------------------------
# a11-a14: Estimator::Propagate with the RK4 / Prince-Dormand integrators
# (src/estimator.cpp:539-704, src/rk4.cpp:5-103, src/princedormand.cpp:7-221),
# default build (no online IMU calibration: Cg = Ca = I unless given).
# Both integrators are the same stage recursion with different tableaux, incl.
# the reference's own conventions (ComposeMotion is handed a_ij-weighted
# velocities AND the shortened step; RK4's 4th stage samples the IMU at the half
# step, rk4.cpp:77-79).
# ----------------------------------------------------------------------------
class MotionState:
    """The part of `State` (src/core.h:117-180) that Propagate touches."""

    def __init__(self, Rsb, Tsb, Vsb, bg, ba, Rsg):
        self.Rsb, self.Tsb, self.Vsb = np.array(Rsb, float), np.array(Tsb, float), np.array(Vsb, float)
        self.bg, self.ba, self.Rsg = np.array(bg, float), np.array(ba, float), np.array(Rsg, float)

    def copy(self):
        return MotionState(self.Rsb, self.Tsb, self.Vsb, self.bg, self.ba, self.Rsg)


def compose_motion(X, V, gyro, accel, dt, g_vec, Cg=None, Ca=None):
    """src/estimator.cpp:598-613 (in place)."""
    Cg = np.eye(3) if Cg is None else Cg
    Ca = np.eye(3) if Ca is None else Ca
    gyro_calib = Cg @ gyro - X.bg
    accel_calib = Ca @ accel - X.ba
    X.Tsb = X.Tsb + V * dt                                                  # :608
    X.Vsb = X.Vsb + (X.Rsb @ accel_calib + X.Rsg @ g_vec) * dt              # :609
    X.Rsb = X.Rsb @ so3_exp(gyro_calib * dt)                                # :610 (normalize(): no-op to rounding)


RK4_TABLEAU = dict(
    a=[[], [0.5], [0.0, 0.5], [0.0, 0.0, 1.0]],
    c_step=[0.0, 0.5, 0.5, 1.0],
    c_imu=[0.0, 0.5, 0.5, 0.5],        # rk4.cpp:77: the 4th stage re-uses the half-step IMU sample
    b=[1 / 6.0, 2 / 6.0, 2 / 6.0, 1 / 6.0])
PD_TABLEAU = dict(
    a=[[], [2 / 9.0], [1 / 12.0, 3 / 12.0], [55 / 324.0, -75 / 324.0, 200 / 324.0],
       [83 / 330.0, -195 / 330.0, 305 / 330.0, 27 / 330.0],
       [-19 / 28.0, 63 / 28.0, 4 / 28.0, -108 / 28.0, 88 / 28.0],
       [38 / 400.0, 0.0, 240 / 400.0, -243 / 400.0, 330 / 400.0, 35 / 400.0]],
    c_step=[0.0, 2 / 9.0, 3 / 9.0, 5 / 9.0, 6 / 9.0, 1.0, 1.0],
    c_imu=[0.0, 2 / 9.0, 3 / 9.0, 5 / 9.0, 6 / 9.0, 1.0, 1.0],
    b=[0.0862, 0.0, 0.6660, -0.7857, 0.9570, 0.0965, -0.0200])       # princedormand.cpp:195-200


def integrator_step(X, P, gyro0, accel0, slope_gyro, slope_accel, dt, Qimu, g_vec, tab, Cg=None, Ca=None, layout=None):
    """One RK4Step (rk4.cpp:35-103) / PrinceDormandStep (princedormand.cpp:85-221).
    Returns (X_new, P_new). layout: an online-calibration build's (kMotionSize = layout.motion_size)."""
    nm = K_MOTION if layout is None else layout.motion_size
    Pmm = P[:nm, :nm]
    Ks, FKs, PKs = [], [], []
    for i in range(len(tab["b"])):
        X0 = X.copy()
        ga = np.concatenate([gyro0, accel0]) + np.concatenate([slope_gyro, slope_accel]) * (tab["c_imu"][i] * dt)
        if i > 0:
            V = sum(a * K for a, K in zip(tab["a"][i], Ks))
            compose_motion(X0, V, ga[:3], ga[3:], tab["c_step"][i] * dt, g_vec, Cg, Ca)
        F, G = motion_jacobian(X0.Rsb, X0.bg, X0.ba, ga[:3], ga[3:], g_vec, Cg, Ca, layout)
        Ks.append(X0.Vsb.copy())
        if i == 0:
            FK = F.copy(); P0 = Pmm.copy()
        else:
            FK = F + F @ sum(a * f for a, f in zip(tab["a"][i], FKs)) * dt
            P0 = Pmm + sum(a * p for a, p in zip(tab["a"][i], PKs)) * dt
        FKs.append(FK)
        PKs.append(F @ P0 + P0 @ F.T + G @ Qimu @ G.T)
    Kt = sum(b * K for b, K in zip(tab["b"], Ks))
    FK = sum(b * f for b, f in zip(tab["b"], FKs))
    PK = sum(b * p for b, p in zip(tab["b"], PKs))
    Xn = X.copy()
    compose_motion(Xn, Kt, gyro0 + slope_gyro * dt, accel0 + slope_accel * dt, dt, g_vec, Cg, Ca)
    Pn, _ = rk4_cov_tail(P, FK, PK, dt)
    return Xn, Pn


def integrate(X, P, gyro0, accel0, slope_gyro, slope_accel, dt, Qimu, g_vec, tab, stepsize=0.002, Cg=None, Ca=None, layout=None):
    """Fixed sub-stepping with the half-step tail trick (rk4.cpp:13-32, princedormand.cpp:62-81)."""
    if stepsize < 0:
        return integrator_step(X, P, gyro0, accel0, slope_gyro, slope_accel, dt, Qimu, g_vec, tab, Cg, Ca, layout)
    total = 0.0
    gyro, accel = np.array(gyro0, float), np.array(accel0, float)
    while total < dt:
        h = stepsize
        if total + h > dt:
            h = dt - total
        elif total + h + 0.5 * h > dt:
            h = 0.5 * h
        X, P = integrator_step(X, P, gyro, accel, slope_gyro, slope_accel, h, Qimu, g_vec, tab, Cg, Ca, layout)
        gyro = gyro + slope_gyro * h
        accel = accel + slope_accel * h
        total += h
    return X, P


class PDControl:
    """cfg_["PrinceDormand"] of the step-size-controlled branch (src/princedormand.cpp:17-24) plus the function-local static `h`
    that branch carries from one call of Estimator::PrinceDormand to the next (:13, :27-33, :52)."""

    def __init__(self, stepsize=0.002, tolerance=1e-3, attempts=12, min_scale_factor=0.125, max_scale_factor=4.0):
        self.h0 = stepsize; self.h = stepsize
        self.tolerance = tolerance; self.attempts = attempts          # (`attempts` is read and never used, :19)
        self.min_scale_factor = min_scale_factor; self.max_scale_factor = max_scale_factor
        self.steps = []                                               # the step lengths taken (what the reference prints, :49)


def integrate_pd_controlled(X, P, gyro0, accel0, slope_gyro, slope_accel, dt, Qimu, g_vec, ctl, Cg=None, Ca=None, layout=None):
    """Estimator::PrinceDormand with control_stepsize = true (src/princedormand.cpp:26-60) AS CODED: PrinceDormandStep returns 0
    (its error estimate is commented out, :216-220), so every step is followed by scale = max_scale_factor; the step starts from
    gyro0 + slope * total_step (:38-39 - not the running sum of the fixed-step branch), and `h` survives the call."""
    total = 0.0
    h = ctl.h
    if h < 1e-6:
        h = ctl.h0
    h = min(h, dt)
    gyro0, accel0 = np.array(gyro0, float), np.array(accel0, float)
    while total < dt:
        X, P = integrator_step(X, P, gyro0 + slope_gyro * total, accel0 + slope_accel * total, slope_gyro, slope_accel, h, Qimu, g_vec,
                               PD_TABLEAU, Cg, Ca, layout)
        err = 0.0                                                     # :216-220
        ctl.steps.append(h)
        total += h
        if err == 0.0:
            scale = ctl.max_scale_factor
        else:
            scale = 0.8 * math.sqrt(math.sqrt(ctl.tolerance * h / err))
            scale = min(max(scale, ctl.min_scale_factor), ctl.max_scale_factor)
        h *= scale
        if total < dt:
            if total + h > dt:
                h = dt - total
            elif total + h + 0.5 * h > dt:
                h = 0.5 * h
    ctl.h = h
    return X, P


def propagate(X, P, gyro0, accel0, slope_gyro, slope_accel, dt, Qimu, Qmodel, g_vec, method="RK4", stepsize=0.002,
              Cg=None, Ca=None, layout=None, pd_control=None):
    """Estimator::Propagate's integration + P_mm += Qmodel (estimator.cpp:580-590); the IMU
    slope bookkeeping of :556-575 is the caller's. layout / Cg / Ca: online-calibration builds (Qmodel kMotionSize square)."""
    tab = RK4_TABLEAU if method == "RK4" else PD_TABLEAU
    if pd_control is not None:                                        # control_stepsize = true (PDControl carries `h` between calls)
        assert method != "RK4"
        X, P = integrate_pd_controlled(X, P, gyro0, accel0, slope_gyro, slope_accel, dt, Qimu, g_vec, pd_control, Cg, Ca, layout)
    else:
        X, P = integrate(X, P, gyro0, accel0, slope_gyro, slope_accel, dt, Qimu, g_vec, tab, stepsize, Cg, Ca, layout)
    P = P.copy()
    nm = K_MOTION if layout is None else layout.motion_size
    P[:nm, :nm] += Qmodel
    return X, P


# ----------------------------------------------------------------------------
# a15: Estimator::AbsorbError (src/estimator.cpp:875-921) with State::operator+=
# (src/core.h:135-165), SO3xR3::operator+= (src/group.h:25-29), Feature::UpdateState
# (src/feature.h:220). Every kEnforceSO3Freq = 50 calls (State::counter, core.h:120-122,154-162) Rsb and Rbc are
# re-normalised (Sophus SO3::normalize: unit quaternion) and Rsg <- exp(log(Rsg) with z zeroed).
# ----------------------------------------------------------------------------
ENFORCE_SO3_FREQ = 50


def rot_to_quat(R):
    """(w, x, y, z) of a rotation matrix (Shepperd), normalised."""
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s]
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s]
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s]
    q = np.array(q)
    return q / np.linalg.norm(q)


def quat_to_rot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def so3_log_quat(q):
    """Sophus SO3::logAndTheta on the unit quaternion (thirdparty/sophus/sophus/so3.hpp)."""
    w, v = q[0], np.asarray(q[1:])
    n2 = float(v @ v)
    if n2 < 1e-20:
        k = 2.0 / w - 2.0 / 3.0 * n2 / (w ** 3)
    else:
        n = math.sqrt(n2)
        k = (math.pi / n if w > 0 else -math.pi / n) if abs(w) < 1e-10 else 2.0 * math.atan(n / w) / n
    return k * v


def absorb_error(st, err, layout, upd_groups, upd_feats):
    """st: dict with Rsb,Tsb,Vsb,bg,ba,Rbc,Tbc,Rsg, gR [G,3,3], gT [G,3], x [F,3], sind [F] (modified in place);
    st["counter"] (created at 0) is State::counter."""
    st["Rsb"] = st["Rsb"] @ so3_exp(err[WSB:WSB + 3])
    st["Tsb"] = st["Tsb"] + err[TSB:TSB + 3]
    st["Vsb"] = st["Vsb"] + err[VSB:VSB + 3]
    st["bg"] = st["bg"] + err[BG:BG + 3]
    st["ba"] = st["ba"] + err[BA:BA + 3]
    st["Rbc"] = st["Rbc"] @ so3_exp(err[WBC:WBC + 3])
    st["Tbc"] = st["Tbc"] + err[TBC:TBC + 3]
    st["Rsg"] = st["Rsg"] @ so3_exp(np.array([err[WSG], err[WSG + 1], 0.0]))
    if getattr(layout, "td", -1) >= 0:                             # core.h:150-152 (USE_ONLINE_TEMPORAL_CALIB)
        st["td"] = st["td"] + err[layout.td]
    if getattr(layout, "Cg", -1) >= 0:                             # estimator.cpp:879-884 -> IMUState::operator+= (src/imu.cpp:7-21):
        k = layout.Ca                                              # Ca's upper triangle row by row, then Cg row by row
        for i in range(3):
            for j in range(i, 3):
                st["Ca"][i, j] += err[k]; k += 1
        for i in range(3):
            for j in range(3):
                st["Cg"][i, j] += err[layout.Cg + 3 * i + j]
    if getattr(layout, "cam_dim", 0) > 0:                          # estimator.cpp:886-890 -> A_*Camera::UpdateState
        cam = st["cam"]                                            # (common/camera_autocalib.h): fx fy cx cy, then d in its own order
        cam["fx"] += err[layout.cam_begin]; cam["fy"] += err[layout.cam_begin + 1]
        cam["cx"] += err[layout.cam_begin + 2]; cam["cy"] += err[layout.cam_begin + 3]
        for k in range(4, layout.cam_dim):
            cam["d"][k - 4] += err[layout.cam_begin + k]
    st["counter"] = st.get("counter", 0) + 1                       # core.h:154-162
    if st["counter"] % ENFORCE_SO3_FREQ == 0:
        st["Rsb"] = quat_to_rot(rot_to_quat(st["Rsb"]))
        st["Rbc"] = quat_to_rot(rot_to_quat(st["Rbc"]))
        w = so3_log_quat(rot_to_quat(st["Rsg"]))
        st["Rsg"] = so3_exp(np.array([w[0], w[1], 0.0]))
    for g in upd_groups:                                           # estimator.cpp:897-905
        off = layout.group_begin + 6 * g
        st["gR"][g] = st["gR"][g] @ so3_exp(err[off:off + 3])
        st["gT"][g] = st["gT"][g] + err[off + 3:off + 6]
    for i in upd_feats:                                            # :906-912
        off = layout.feature_begin + 3 * int(st["sind"][i])
        st["x"][i] = st["x"][i] + err[off:off + 3]


# ----------------------------------------------------------------------------
# Feature::SubfilterUpdate (src/feature.cpp:246-297) - the 3x3 depth sub-filter of a feature
# that is not in the state yet - and the candidate tests / score that decide who enters the state
# (src/options.cpp:10-33, src/feature.cpp:133-142).
# ----------------------------------------------------------------------------
FEAT_INITIALIZING, FEAT_READY = 0, 1


def subfilter_update(x, P, xp_meas, Rsb, Tsb, Rbc, Tbc, Rsbr, Tsbr, cam, Rtri=3.5, MH_thresh=5.991, ready_steps=5,
                     init_counter=0, outlier_counter=0.0, invdepth=False):
    """Returns (x, P, status, init_counter, outlier_counter). 3x3 matrices row-major [i, j]."""
    x = np.asarray(x, dtype=np.float64); P = np.asarray(P, dtype=np.float64)
    init_counter += 1                                              # :256
    Xc, dXc_dx = feature_xc(x, invdepth)                           # :258 Xc(&dXc_dx), feature.cpp:98-105
    # gtot = (gsb * gbc)^-1 * ref.gsb * gbc  (:260)
    Rsc, Tsc = Rsb @ Rbc, Rsb @ Tbc + Tsb
    Rrc, Trc = Rsbr @ Rbc, Rsbr @ Tbc + Tsbr
    Rtot, Ttot = Rsc.T @ Rrc, Rsc.T @ (Trc - Tsc)
    Xcn = Rtot @ Xc + Ttot                                         # :261
    dxcn_dXcn = np.array([[1 / Xcn[2], 0, -Xcn[0] / Xcn[2] ** 2], [0, 1 / Xcn[2], -Xcn[1] / Xcn[2] ** 2]])
    xp, dxp_dxcn = camera_project(cam, Xcn[:2] / Xcn[2])          # :264-267
    H = ((dxp_dxcn @ dxcn_dXcn) @ Rtot) @ dXc_dx                   # :269 (left to right)
    inn = np.asarray(xp_meas, dtype=np.float64) - xp
    S = H @ P @ H.T
    S[0, 0] += Rtri; S[1, 1] += Rtri                               # :273-275
    ratio = float(inn @ np.linalg.solve(S, inn)) / MH_thresh       # :277 (Eigen: LDLT)
    if ratio > 1:                                                  # :279-285
        S[0, 0] += Rtri * (ratio - 1); S[1, 1] += Rtri * (ratio - 1)
        outlier_counter += math.sqrt(ratio)
    else:
        outlier_counter = 0.0
    K = P @ H.T @ np.linalg.inv(S)                                 # :287
    x = x + K @ inn
    I_KH = np.eye(3) - K @ H
    P = I_KH @ P @ I_KH.T + K * Rtri @ K.T                         # :291 (Rtri, not the inflated S)
    status = FEAT_READY if init_counter > ready_steps else FEAT_INITIALIZING   # :293-297
    return x, P, status, init_counter, outlier_counter


def candidate_flags(x, status, outlier_counter, zmin=0.05, zmax=5.0, max_subfilter_outlier=0.01, invdepth=False):
    """(Criteria::Candidate, Criteria::CandidateStrict), src/options.cpp:10-33."""
    zed = feature_z(x, invdepth)                                   # Feature::z(), feature.cpp:120-126
    ok = outlier_counter < max_subfilter_outlier and zmin < zed < zmax
    return (status in (FEAT_READY, FEAT_INITIALIZING)) and ok, status == FEAT_READY and ok


def candidate_scores(P, outlier_counter):
    """the three `comparison_score_type` values Criteria::CandidateComparison computes (src/options.cpp:42-56):
    DepthUncertainty, CovarianceDiagNorm, CovarianceDiagNormPlusOutlierCount."""
    P = np.asarray(P)
    dn = float(np.linalg.norm(np.diag(P)))
    return -1.0 * float(P[2, 2]), -1.0 * dn, -1.0 * (dn + outlier_counter)


def candidate_before(status1, P1, status2, P2):
    """Criteria::CandidateComparison(f1, f2) as coded (src/options.cpp:58-60): status as integer (READY = 2 >
    INITIALIZING = 1, src/core.h:190-199), then Feature::score() - the score of `comparison_score_type` is unused."""
    s1 = 2 if status1 == FEAT_READY else 1
    s2 = 2 if status2 == FEAT_READY else 1
    return (s1 > s2) or (s1 == s2 and feature_score(P1) > feature_score(P2))


def feature_score(P):
    """Feature::score(), src/feature.cpp:133-142: confidence in depth."""
    return -float(np.asarray(P)[2, 2])


# ----------------------------------------------------------------------------
# a7: Estimator::OnePointRANSAC (src/update.cpp:213-393), numeric core for one filter.
# ----------------------------------------------------------------------------
def one_point_ransac(st, P, xp, cam, layout, R, ransac_thresh, ransac_chi2, gauge_group, instate_groups,
                     in_current_ekf_update=(), calib_gyro=None):
    """st as in absorb_error plus ref [F]. All F features are the MH inliers handed in.
    Returns dict(inliers=sorted feature indices kept, rejected=..., chi2={feature: distance},
    low=low-innovation mask, err=dx of the partial update, P_partial=P after it).
    The hypothesis loop of :238-258 draws k but never uses it: the low-innovation set is
    simply {f : |xp - pred| < ransac_thresh} (pred == the prediction of ComputeJacobian)."""
    import copy
    F = len(xp)
    # online-calibration builds: st also carries td, Cg, Ca and cam (the intrinsics are state: BackupState / AbsorbError /
    # RestoreState move them with X_); calib_gyro = last_gyro_ (update.cpp:348-349 hands it to ComputeJacobian)
    calib_build = getattr(layout, "td", -1) >= 0 or getattr(layout, "cam_dim", 0) > 0

    def jac_all(s):
        out = []
        cal = dict(gyro=calib_gyro, Cg=s["Cg"], bg=s["bg"], Vsb=s["Vsb"], td=s["td"]) if calib_build else None
        cm = s["cam"] if calib_build and getattr(layout, "cam_dim", 0) > 0 else cam
        for i in range(F):
            r = int(s["ref"][i])
            out.append(compute_jacobian(s["x"][i], xp[i], s["gR"][r], s["gT"][r], s["Rsb"], s["Tsb"], s["Rbc"], s["Tbc"],
                                        cm, layout, r, int(s["sind"][i]), calib=cal)[:2])
        return out
    J0 = jac_all(st)
    low = np.array([np.linalg.norm(J0[i][1]) < ransac_thresh for i in range(F)])         # :245-249
    res = dict(low=low, chi2={}, err=None, P_partial=None)
    if low.all():                                                                          # :263-265
        res.update(inliers=list(range(F)), rejected=[])
        return res
    s2 = copy.deepcopy(st)                                                                 # BackupState :283
    P2 = P.copy()
    active_groups = sorted(set(int(g) for g in st["ref"]))
    groups_low = sorted(set(int(st["ref"][i]) for i in range(F) if low[i]))
    if low.any():
        if gauge_group not in groups_low:                                                  # :292-301
            cov = [sum(P2[layout.group_begin + 6 * g + d, layout.group_begin + 6 * g + d] for d in range(6)) for g in groups_low]
            tmpref = groups_low[int(np.argmin(cov))]                                       # FindNewRefGroup, estimator.cpp:1394-1407
            P2 = p_zero_rc(P2, layout.group_begin + 6 * tmpref, 6)
        for i in range(F):                                                                 # :304-310
            if not low[i]:
                P2 = p_zero_rc(P2, layout.feature_begin + 3 * int(st["sind"][i]), 3)
        for g in active_groups:                                                            # :311-317
            if g not in groups_low:
                P2 = p_zero_rc(P2, layout.group_begin + 6 * g, 6)
        idx = [i for i in range(F) if low[i]]
        H = np.vstack([J0[i][0] for i in idx])                                             # :326 full J rows (no FillJacobianBlock)
        inn = np.concatenate([J0[i][1] for i in idx])
        err, P2, _ = update_joseph(H, P2, inn, np.full(len(inn), R))                       # :332
        absorb_error(s2, err, layout, instate_groups, in_current_ekf_update)              # :333
        res["err"], res["P_partial"] = err, P2
    kept, rejected = [i for i in range(F) if low[i]], []
    J1 = jac_all(s2)                                                                       # :348 (at the updated state)
    for i in range(F):
        if not low[i]:
            d = mh_distances(J1[i][0][None], P2, J1[i][1][None], R)[0]                     # :352-356
            res["chi2"][i] = d
            (kept if d < ransac_chi2 else rejected).append(i)
    res.update(inliers=sorted(kept), rejected=rejected)                                    # state restored by the caller's copy
    return res
