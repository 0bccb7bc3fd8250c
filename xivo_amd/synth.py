"""Seeded synthetic inputs for tests/ and bench.py.

Plumbing, not product: numpy only, no reference to the oracle or the library.
"""
import math
import numpy as np


def s_level(N, F, batch, seed=0, dense=False, G=8, fix_group_block=False):
    """Boundary-level set: P = A A^T / N + 1e-3 I, A ~ U(-1, 1); H with XIVO's
    row-pair sparsity (2x3 blocks at cols {0,3,15,18}, one group block, one
    feature block; entries ~ N(0, 100^2)) or dense U(-1,1); inn ~ N(0, 1.5^2);
    diagR = 2.25 (visual_meas_std^2, cfg/tumvi_cam0.json)."""
    rng = np.random.default_rng(seed)
    M = 2 * F
    A = rng.uniform(-1, 1, size=(batch, N, N))
    P = A @ np.transpose(A, (0, 2, 1)) / N + 1e-3 * np.eye(N)[None]
    if dense:
        H = rng.uniform(-1, 1, size=(batch, M, N))
    else:
        H = np.zeros((batch, M, N))
        G = max(1, min(G, (N - 23 - 3) // 6))
        fbeg = 23 + 6 * G
        fslots = max(1, (N - fbeg) // 3)
        for i in range(F):
            g = 23 + 6 * (i % G)
            f = fbeg + 3 * (i % fslots)
            cols = [0, 3, 15, 18, g, f] + ([g + 3] if fix_group_block else [])
            for c in cols:
                H[:, 2 * i:2 * i + 2, c:c + 3] = rng.normal(0, 100.0, size=(batch, 2, 3))
    inn = rng.normal(0, 1.5, size=(batch, M))
    diagR = np.full((batch, M), 2.25)
    return P, H, inn, diagR


def _rot(w):
    th = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3)
    return np.eye(3) + math.sin(th) / th * W + (1 - math.cos(th)) / (th * th) * W @ W


PINHOLE = dict(model=0, rows=480, cols=640, fx=580.0, fy=580.0, cx=320.0, cy=240.0, d=[])
# cfg/tumvi_cam0.json camera_cfg (equidistant, 512x512)
EQUI = dict(model=3, rows=512, cols=512, fx=190.97847715128717, fy=190.9733070521226,
            cx=254.93170605935475, cy=256.8974428996504,
            d=[0.0034823894022493434, 0.0007150348452162257, -0.0020532361418706202, 0.00020293673591811182])
RADTAN = dict(model=2, rows=480, cols=640, fx=458.654, fy=457.296, cx=367.215, cy=248.375,
              d=[0.00019359, 1.76187114e-05, -0.28340811, 0.07395907, 0.0])
ATAN = dict(model=1, rows=480, cols=640, fx=562.3, fy=561.9, cx=318.6, cy=243.9, d=[0.926])


def g_level(n_groups, n_features, F, batch, seed=0, cam=PINHOLE, N=None, pix_noise=1.5):
    """Layout-faithful scene: per filter a current pose, `n_groups` group
    anchors, and F in-state features (feature i lives in slot i, anchored to
    group i % n_groups), all visible from both the anchor and the current
    camera. Returns dict of numpy arrays (3x3 matrices row-major [i, j])."""
    rng = np.random.default_rng(seed)
    Rbc = _rot(np.array([0.01, -0.02, 1.55]))
    Tbc = np.array([0.04, -0.05, 0.01])
    out = dict(Rsb=np.empty((batch, 3, 3)), Tsb=np.empty((batch, 3)), Rbc=np.tile(Rbc, (batch, 1, 1)),
               Tbc=np.tile(Tbc, (batch, 1)), gR=np.empty((batch, n_groups, 3, 3)),
               gT=np.empty((batch, n_groups, 3)), x=np.empty((batch, F, 3)), xp=np.empty((batch, F, 2)),
               ref=np.empty((batch, F), dtype=np.int32), sind=np.empty((batch, F), dtype=np.int32),
               Xcn=np.empty((batch, F, 3)))
    fx, fy, cx, cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    for b in range(batch):
        out["Rsb"][b] = _rot(rng.uniform(-0.15, 0.15, 3))
        out["Tsb"][b] = rng.uniform(-0.3, 0.3, 3)
        for g in range(n_groups):
            out["gR"][b, g] = _rot(rng.uniform(-0.15, 0.15, 3))
            out["gT"][b, g] = rng.uniform(-0.3, 0.3, 3)
        for i in range(F):
            g = i % n_groups
            while True:
                z = rng.uniform(1.5, 5.0)
                u = rng.uniform(0.2, 0.8) * cam["cols"]; v = rng.uniform(0.2, 0.8) * cam["rows"]
                xc = np.array([(u - cx) / fx, (v - cy) / fy])  # pinhole-style back-projection (any model: just a ray)
                Xc = np.array([xc[0] * z, xc[1] * z, z])
                Xs = out["gR"][b, g] @ (Rbc @ Xc + Tbc) + out["gT"][b, g]
                Xcn = Rbc.T @ (out["Rsb"][b].T @ (Xs - out["Tsb"][b]) - Tbc)
                if Xcn[2] > 0.5 and abs(Xcn[0] / Xcn[2]) < 0.6 and abs(Xcn[1] / Xcn[2]) < 0.6:
                    break
            out["x"][b, i] = [xc[0], xc[1], math.log(z)]
            out["ref"][b, i] = g
            out["sind"][b, i] = i
            out["Xcn"][b, i] = Xcn
    out["pix_noise"] = rng.normal(0, pix_noise, size=(batch, F, 2))
    out["cam"] = cam
    out["n_groups"], out["n_features"], out["F"] = n_groups, n_features, F
    out["N"] = 23 + 6 * n_groups + 3 * n_features if N is None else N
    return out


def scene_structs(sc):
    """A g_level scene as the C-ABI struct arrays (poses, groups, feats; column-major 3x3), without pixels: the timing
    scripts and bench.py take the predicted pixels from the device itself (jacobians with xp = 0 give inn = -prediction)."""
    from .lib import feat_dtype, group_dtype, pose_dtype
    B, F = sc["x"].shape[:2]
    G = sc["gR"].shape[1]
    poses = np.zeros(B, dtype=pose_dtype); groups = np.zeros((B, G), dtype=group_dtype); feats = np.zeros((B, F), dtype=feat_dtype)
    cm = lambda R: np.asarray(R).T.reshape(-1)
    for b in range(B):
        poses[b]["Rsb"], poses[b]["Tsb"] = cm(sc["Rsb"][b]), sc["Tsb"][b]
        poses[b]["Rbc"], poses[b]["Tbc"] = cm(sc["Rbc"][b]), sc["Tbc"][b]
        poses[b]["Rsg"] = np.eye(3).reshape(-1)
        for g in range(G):
            groups[b, g]["Rsb"], groups[b, g]["Tsb"] = cm(sc["gR"][b, g]), sc["gT"][b, g]
    feats["x"] = sc["x"]; feats["ref_sind"] = sc["ref"]; feats["sind"] = sc["sind"]
    return poses, groups, feats


def spd_covariance(N, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    A = rng.uniform(-1, 1, size=(N, N))
    return (A @ A.T / N + 1e-3 * np.eye(N)) * scale
