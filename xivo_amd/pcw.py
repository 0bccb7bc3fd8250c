"""Image-free sequence source: a point-cloud world seen by a moving camera + a simulated IMU.

Restates what the reference's simulation scripts feed `Estimator::InertialMeas` / `VisualMeasPointCloud`:
  * `RandomPCW` follows scripts/point_cloud_world.py:44-131 (uniform points in a box, pinhole visibility test, pixel
    noise, track ids that start at 10000 - `counter0` of src/feature.h - are handed out when a point becomes visible
    and dropped when it leaves the image, so a point that comes back is a NEW track, as for `Tracker::UpdatePointCloud`,
    src/tracker.cpp:632-702);
  * `TrajectorySim` plays the role of scripts/imu_sim.py:IMUSimBase with the curves of scripts/imu_trajectories.py
    (LissajousSim :289-313, TrefoilSim :316-341), but in closed form: position, velocity and acceleration are the
    analytic curve and its derivatives, orientation is exp(hat(w(t))) with the body rate from the right Jacobian, so no
    ODE solve / interpolation table is needed and ground truth is exact at any t.
The accelerometer model is the one the filter integrates (src/estimator.cpp:598-613: Vsb' = Rsb (accel - ba) + Rsg g):
accel = Rsb^T (a_s - g_s) + ba + noise, gyro = w_b + bg + noise.

Host-side numpy only; nothing here touches the GPU or the oracle.
"""
import numpy as np


def _hat(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def so3_exp(w):
    th = float(np.linalg.norm(w))
    W = _hat(w)
    if th < 1e-9:
        return np.eye(3) + W + 0.5 * W @ W
    return np.eye(3) + np.sin(th) / th * W + (1.0 - np.cos(th)) / (th * th) * W @ W


def so3_log(R):
    c = max(-1.0, min(1.0, (np.trace(R) - 1.0) * 0.5))
    th = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-9:
        return 0.5 * v
    return th / (2.0 * np.sin(th)) * v


def _right_jacobian(w):
    th = float(np.linalg.norm(w))
    W = _hat(w)
    if th < 1e-6:
        return np.eye(3) - 0.5 * W + W @ W / 6.0
    return np.eye(3) - (1.0 - np.cos(th)) / th ** 2 * W + (th - np.sin(th)) / th ** 3 * W @ W


class RandomPCW:
    """scripts/point_cloud_world.py RandomPCW: `npts` points uniform in xlim x ylim x zlim."""

    def __init__(self, npts=1000, xlim=(-10, 10), ylim=(-10, 10), zlim=(-5, 5), seed=0):
        self.rng = np.random.default_rng(seed)
        lo = np.array([xlim[0], ylim[0], zlim[0]], dtype=float)
        hi = np.array([xlim[1], ylim[1], zlim[1]], dtype=float)
        self.Xs = self.rng.uniform(lo, hi, size=(npts, 3))
        self.ids = np.full(npts, -1, dtype=np.int64)
        self.next_pt_id = 10000

    def generate_measurements(self, Rsc, Tsc, K, imw, imh, noise_px_std):
        """-> (feature_ids [n], xp_and_depths [n x 3]) of the points inside the image, ascending point order
        (point_cloud_world.py:64-99). Noise is drawn for every point in front of the camera, visible or not, as the
        reference does, so that the stream does not depend on who is visible."""
        Xc = (self.Xs - Tsc) @ Rsc          # rows: Rsc^T (Xs - Tsc)
        front = Xc[:, 2] > 0
        z = np.where(front, Xc[:, 2], 1.0)
        u = K[0, 0] * Xc[:, 0] / z + K[0, 2]
        v = K[1, 1] * Xc[:, 1] / z + K[1, 2]
        vis = front & (u >= 0) & (v >= 0) & (u <= imw) & (v <= imh)
        noise = noise_px_std * self.rng.standard_normal((self.Xs.shape[0], 2))
        new = vis & (self.ids < 0)
        n_new = int(new.sum())
        self.ids[new] = self.next_pt_id + np.arange(n_new)
        self.next_pt_id += n_new
        self.ids[~vis] = -1
        sel = np.nonzero(vis)[0]
        out = np.stack([u[sel] + noise[sel, 0], v[sel] + noise[sel, 1], Xc[sel, 2]], axis=1)
        return self.ids[sel].copy(), out


_CURVES = {
    # p(s), p'(s), p''(s); s = rate * t
    "lissajous": (lambda s: np.array([4 * np.cos(3 * s), 0.1 * np.sin(7 * s), 4 * np.sin(2 * s)]),
                  lambda s: np.array([-12 * np.sin(3 * s), 0.7 * np.cos(7 * s), 8 * np.cos(2 * s)]),
                  lambda s: np.array([-36 * np.cos(3 * s), -4.9 * np.sin(7 * s), -16 * np.sin(2 * s)])),
    "trefoil": (lambda s: np.array([(4 + np.cos(3 * s)) * np.cos(2 * s), (4 + np.cos(3 * s)) * np.sin(2 * s), np.sin(3 * s)]),
                lambda s: np.array([-3 * np.sin(3 * s) * np.cos(2 * s) - 2 * (4 + np.cos(3 * s)) * np.sin(2 * s),
                                    -3 * np.sin(3 * s) * np.sin(2 * s) + 2 * (4 + np.cos(3 * s)) * np.cos(2 * s),
                                    3 * np.cos(3 * s)]),
                lambda s: np.array([12 * np.sin(2 * s) * np.sin(3 * s) - 9 * np.cos(2 * s) * np.cos(3 * s)
                                    - 4 * np.cos(2 * s) * (np.cos(3 * s) + 4),
                                    -4 * np.sin(2 * s) * (np.cos(3 * s) + 4) - 12 * np.cos(2 * s) * np.sin(3 * s)
                                    - 9 * np.cos(3 * s) * np.sin(2 * s),
                                    -9 * np.sin(3 * s)])),
}


class TrajectorySim:
    """Ground truth + IMU samples along an analytic curve (see the module docstring).

    `rate` slows the reference curves down (their accelerations reach 36 m/s^2 at rate 1); the position is shifted
    so that Tsb(0) = 0, and Rsb(0) = I, like the reference sims (imu_sim.py:226-228)."""

    def __init__(self, motion_type="lissajous", rate=0.1, rot_amp=0.2, noise_accel=1e-4, noise_gyro=1e-5,
                 bias_accel=(0, 0, 0), bias_gyro=(0, 0, 0), grav_s=(0, 0, -9.8), seed=1):
        self.p, self.dp, self.ddp = _CURVES[motion_type]
        self.rate = float(rate)
        self.p0 = self.p(0.0)
        self.rot_amp = float(rot_amp)
        self.rot_w = np.array([0.3, 0.4, 0.1]) * 3.0   # rad/s of the three rotation-vector components
        self.noise_accel, self.noise_gyro = noise_accel, noise_gyro
        self.bias_accel = np.asarray(bias_accel, dtype=float)
        self.bias_gyro = np.asarray(bias_gyro, dtype=float)
        self.grav_s = np.asarray(grav_s, dtype=float)
        self.rng = np.random.default_rng(seed)

    def _w(self, t):
        return self.rot_amp * np.sin(self.rot_w * t), self.rot_amp * self.rot_w * np.cos(self.rot_w * t)

    def gsb(self, t):
        w, _ = self._w(t)
        return so3_exp(w), self.p(self.rate * t) - self.p0

    def vel(self, t):
        return self.rate * self.dp(self.rate * t)

    def real_accel_gyro(self, t):
        w, wd = self._w(t)
        Rsb = so3_exp(w)
        a_s = self.rate ** 2 * self.ddp(self.rate * t)
        return Rsb.T @ a_s, _right_jacobian(w) @ wd

    def meas(self, t):
        """-> (accel, gyro) as an IMU at time t would report them."""
        Rsb, _ = self.gsb(t)
        a_b, w_b = self.real_accel_gyro(t)
        accel = a_b - Rsb.T @ self.grav_s + self.bias_accel + self.noise_accel * self.rng.standard_normal(3)
        gyro = w_b + self.bias_gyro + self.noise_gyro * self.rng.standard_normal(3)
        return accel, gyro


# ---- the same two simulators for B sequences at once (numpy over the sequence axis): what feeds thousands of
# ---- filters per frame; formulas identical to RandomPCW / TrajectorySim above (tests compare them)
def so3_exp_batch(w):
    """w: [B, 3] -> [B, 3, 3]"""
    th = np.linalg.norm(w, axis=1)
    W = np.zeros((w.shape[0], 3, 3))
    W[:, 0, 1], W[:, 0, 2], W[:, 1, 0], W[:, 1, 2], W[:, 2, 0], W[:, 2, 1] = -w[:, 2], w[:, 1], w[:, 2], -w[:, 0], -w[:, 1], w[:, 0]
    small = th < 1e-9
    ths = np.where(small, 1.0, th)
    a = np.where(small, 1.0, np.sin(ths) / ths)
    b = np.where(small, 0.5, (1.0 - np.cos(ths)) / (ths * ths))
    return np.eye(3)[None] + a[:, None, None] * W + b[:, None, None] * (W @ W), W, th


class BatchTrajectorySim:
    """B trajectories: motion[b] in {"lissajous", "trefoil"}, rate[b]; one noise stream for all."""

    def __init__(self, motion, rate, rot_amp=0.2, noise_accel=1e-4, noise_gyro=1e-5, grav_s=(0, 0, -9.8), seed=1):
        self.is_tre = np.array([m == "trefoil" for m in motion])
        self.rate = np.asarray(rate, dtype=float)
        self.B = len(self.rate)
        self.rot_amp = float(rot_amp)
        self.rot_w = np.array([0.3, 0.4, 0.1]) * 3.0
        self.noise_accel, self.noise_gyro = noise_accel, noise_gyro
        self.grav_s = np.asarray(grav_s, dtype=float)
        self.rng = np.random.default_rng(seed)
        self.p0 = self._curve(np.zeros(self.B))[0]

    def _curve(self, s):
        c2, s2, c3, s3, c7, s7 = np.cos(2 * s), np.sin(2 * s), np.cos(3 * s), np.sin(3 * s), np.cos(7 * s), np.sin(7 * s)
        pl = np.stack([4 * c3, 0.1 * s7, 4 * s2], 1)
        dl = np.stack([-12 * s3, 0.7 * c7, 8 * c2], 1)
        al = np.stack([-36 * c3, -4.9 * s7, -16 * s2], 1)
        pt = np.stack([(4 + c3) * c2, (4 + c3) * s2, s3], 1)
        dt = np.stack([-3 * s3 * c2 - 2 * (4 + c3) * s2, -3 * s3 * s2 + 2 * (4 + c3) * c2, 3 * c3], 1)
        at = np.stack([12 * s2 * s3 - 9 * c2 * c3 - 4 * c2 * (c3 + 4), -4 * s2 * (c3 + 4) - 12 * c2 * s3 - 9 * c3 * s2, -9 * s3], 1)
        m = self.is_tre[:, None]
        return np.where(m, pt, pl), np.where(m, dt, dl), np.where(m, at, al)

    def gsb(self, t):
        w = np.tile(self.rot_amp * np.sin(self.rot_w * t), (self.B, 1))
        R, _, _ = so3_exp_batch(w)
        return R, self._curve(self.rate * t)[0] - self.p0

    def vel(self, t):
        return self.rate[:, None] * self._curve(self.rate * t)[1]

    def meas(self, t):
        """-> (accel [B, 3], gyro [B, 3])"""
        w = np.tile(self.rot_amp * np.sin(self.rot_w * t), (self.B, 1))
        wd = self.rot_amp * self.rot_w * np.cos(self.rot_w * t)
        R, W, th = so3_exp_batch(w)
        a_s = (self.rate ** 2)[:, None] * self._curve(self.rate * t)[2]
        Jr = _right_jacobian(w[0])                       # the orientation profile is shared by all sequences
        accel = np.einsum("bji,bj->bi", R, a_s - self.grav_s) + self.noise_accel * self.rng.standard_normal((self.B, 3))
        gyro = (Jr @ wd)[None] + self.noise_gyro * self.rng.standard_normal((self.B, 3))
        return accel, gyro


class BatchPCW:
    """B point-cloud worlds with RandomPCW's track-id semantics; generate() returns the concatenated track lists in the
    layout xivo::hip::BatchEstimator::VisualMeasPointCloud takes (offsets, ids, (x, y, depth) rows)."""

    def __init__(self, B, npts=1000, xlim=(-10, 10), ylim=(-10, 10), zlim=(-5, 5), seed=0, Xs=None):
        self.rng = np.random.default_rng(seed)
        lo = np.array([xlim[0], ylim[0], zlim[0]], dtype=float); hi = np.array([xlim[1], ylim[1], zlim[1]], dtype=float)
        self.Xs = self.rng.uniform(lo, hi, size=(B, npts, 3)) if Xs is None else np.asarray(Xs, dtype=float)
        self.ids = np.full(self.Xs.shape[:2], -1, dtype=np.int64)
        self.next_pt_id = np.full(self.Xs.shape[0], 10000, dtype=np.int64)

    def generate(self, Rsc, Tsc, K, imw, imh, noise_px_std):
        Xc = np.einsum("bpj,bji->bpi", self.Xs - Tsc[:, None, :], Rsc)       # Rsc^T (Xs - Tsc)
        front = Xc[..., 2] > 0
        z = np.where(front, Xc[..., 2], 1.0)
        u = K[0, 0] * Xc[..., 0] / z + K[0, 2]
        v = K[1, 1] * Xc[..., 1] / z + K[1, 2]
        vis = front & (u >= 0) & (v >= 0) & (u <= imw) & (v <= imh)
        noise = noise_px_std * self.rng.standard_normal(u.shape + (2,))
        new = vis & (self.ids < 0)
        rank = np.cumsum(new, axis=1) - 1                                     # order of appearance inside a world
        self.ids = np.where(new, self.next_pt_id[:, None] + rank, self.ids)
        self.next_pt_id = self.next_pt_id + new.sum(axis=1)
        self.ids = np.where(vis, self.ids, -1)
        off = np.zeros(self.Xs.shape[0] + 1, dtype=np.int32)
        off[1:] = np.cumsum(vis.sum(axis=1))
        sel = np.nonzero(vis)                                                 # row-major: sequences in order, points ascending
        meas = np.stack([u[sel] + noise[sel][:, 0], v[sel] + noise[sel][:, 1], Xc[..., 2][sel]], axis=1)
        return off, self.ids[sel].copy(), np.ascontiguousarray(meas)
