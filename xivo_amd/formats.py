"""Data formats on either side of the path: the EuRoC / TUM-VI csv layout `DataLoader` reads
(src/loader.cpp:14-60: `<dir>/data.csv`, one header line, `ts[ns],wx,wy,wz,ax,ay,az` for the IMU and `ts[ns],filename`
for the camera, '#' lines skipped, entries merged in ascending timestamp order) and the trajectory dump of the `vio`
application (src/app/vio.cpp:101-106: one line `ts Tsb(3) Wsb(3)` per message, Wsb = log of the rotation), plus the
absolute trajectory error used to score it (scripts/tum_rgbd_benchmark_tools/evaluate_ate.py: Horn alignment, RMSE of the
translational residual)."""
import os

import numpy as np


def read_imu_csv(imu_dir):
    """-> (ts [n] int64 ns, gyro [n x 3], accel [n x 3]), ascending in time."""
    ts, gyro, accel = [], [], []
    with open(os.path.join(imu_dir, "data.csv")) as f:
        f.readline()                                   # header (loader.cpp:40)
        for line in f.read().split():                  # `is >> line`: whitespace separated tokens
            if line.startswith("#"):
                continue
            c = line.split(",")
            ts.append(int(c[0])); gyro.append([float(v) for v in c[1:4]]); accel.append([float(v) for v in c[4:7]])
    ts = np.array(ts, dtype=np.int64); o = np.argsort(ts, kind="stable")
    return ts[o], np.array(gyro, dtype=float).reshape(-1, 3)[o], np.array(accel, dtype=float).reshape(-1, 3)[o]


def read_cam_csv(image_dir):
    """-> (ts [n] int64 ns, paths [n]) with paths = <image_dir>/data/<filename> (loader.cpp:19-29)."""
    ts, paths = [], []
    with open(os.path.join(image_dir, "data.csv")) as f:
        f.readline()
        for line in f.read().split():
            if line.startswith("#"):
                continue
            c = line.split(",")
            ts.append(int(c[0])); paths.append(os.path.join(image_dir, "data", c[1]))
    ts = np.array(ts, dtype=np.int64); o = np.argsort(ts, kind="stable")
    return ts[o], [paths[i] for i in o]


def write_imu_csv(imu_dir, ts, gyro, accel):
    os.makedirs(imu_dir, exist_ok=True)
    with open(os.path.join(imu_dir, "data.csv"), "w") as f:
        f.write("#timestamp [ns],w_RS_S_x [rad s^-1],w_RS_S_y [rad s^-1],w_RS_S_z [rad s^-1],"
                "a_RS_S_x [m s^-2],a_RS_S_y [m s^-2],a_RS_S_z [m s^-2]\n")
        for t, g, a in zip(ts, gyro, accel):
            f.write("%d,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g\n" % (int(t), g[0], g[1], g[2], a[0], a[1], a[2]))


def merge_streams(imu_ts, cam_ts):
    """Messages of both streams in ascending timestamp order (loader.cpp:55-57); at equal timestamps the IMU message
    goes first, the order scripts/pyxivo_pcw.py:121-129 documents (the reference's std::sort leaves ties unspecified).
    -> list of (ts, kind, index), kind 0 = IMU, 1 = camera."""
    ev = [(int(t), 0, i) for i, t in enumerate(imu_ts)] + [(int(t), 1, i) for i, t in enumerate(cam_ts)]
    ev.sort(key=lambda e: (e[0], e[1]))
    return ev


def write_trajectory(path, ts_ns, Tsb, Wsb):
    """`ts Tsb Wsb` per line (vio.cpp:101-106)."""
    with open(path, "w") as f:
        for t, T, W in zip(ts_ns, Tsb, Wsb):
            f.write("%d %.9g %.9g %.9g %.9g %.9g %.9g\n" % (int(t), T[0], T[1], T[2], W[0], W[1], W[2]))


def read_trajectory(path):
    ts, rows = [], []
    with open(path) as f:
        for line in f:
            c = line.split()
            if not c or c[0].startswith("#"):
                continue
            ts.append(int(c[0]))                      # ns stamps exceed the 53-bit mantissa of a double
            rows.append([float(v) for v in c[1:7]])
    a = np.array(rows, dtype=float).reshape(-1, 6)
    return np.array(ts, dtype=np.int64), a[:, 0:3], a[:, 3:6]


def ate_rmse(est, gt, align=True):
    """Absolute trajectory error: RMSE of |R est + t - gt| with (R, t) the least-squares rigid alignment
    (Horn / Kabsch, as evaluate_ate.py does), or of |est - gt| when align is False."""
    est = np.asarray(est, dtype=float); gt = np.asarray(gt, dtype=float)
    if align:
        me, mg = est.mean(0), gt.mean(0)
        U, _, Vt = np.linalg.svd((gt - mg).T @ (est - me))
        S = np.eye(3)
        if np.linalg.det(U) * np.linalg.det(Vt) < 0:
            S[2, 2] = -1.0
        R = U @ S @ Vt
        est = (est - me) @ R.T + mg
    return float(np.sqrt(np.mean(np.sum((est - gt) ** 2, axis=1))))
