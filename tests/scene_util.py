"""Turn a synth.g_level scene into the C-ABI struct arrays (column-major 3x3)."""
import numpy as np

import xivo_oracle as orc
from xivo_amd.lib import feat_dtype, group_dtype, pose_dtype


def cm(R):
    return np.asarray(R).T.reshape(-1)


def scene_arrays(sc, cam, xp=None):
    B, F = sc["x"].shape[:2]
    G = sc["gR"].shape[1]
    poses = np.zeros(B, dtype=pose_dtype)
    groups = np.zeros((B, G), dtype=group_dtype)
    feats = np.zeros((B, F), dtype=feat_dtype)
    if xp is None:
        xp = np.empty((B, F, 2))
        for b in range(B):
            for i in range(F):
                Xcn = sc["Xcn"][b, i]
                xp[b, i] = orc.camera_project(cam, Xcn[:2] / Xcn[2])[0] + sc["pix_noise"][b, i]
    for b in range(B):
        poses[b]["Rsb"] = cm(sc["Rsb"][b]); poses[b]["Tsb"] = sc["Tsb"][b]
        poses[b]["Rbc"] = cm(sc["Rbc"][b]); poses[b]["Tbc"] = sc["Tbc"][b]
        for g in range(G):
            groups[b, g]["Rsb"] = cm(sc["gR"][b, g]); groups[b, g]["Tsb"] = sc["gT"][b, g]
        for i in range(F):
            feats[b, i]["x"] = sc["x"][b, i]; feats[b, i]["xp"] = xp[b, i]
            feats[b, i]["ref_sind"] = sc["ref"][b, i]; feats[b, i]["sind"] = sc["sind"][b, i]
    return poses, groups, feats, xp


def oracle_jacobians(sc, cam, lay, xp, b):
    F = sc["x"].shape[1]
    Js, inns, blocks = [], [], []
    for i in range(F):
        r, s = int(sc["ref"][b, i]), int(sc["sind"][b, i])
        J, inn, blk = orc.compute_jacobian(sc["x"][b, i], xp[b, i], sc["gR"][b, r], sc["gT"][b, r], sc["Rsb"][b],
                                           sc["Tsb"][b], sc["Rbc"][b], sc["Tbc"][b], cam, lay, r, s)
        Js.append(J); inns.append(inn); blocks.append(blk)
    return np.array(Js), np.array(inns), np.array(blocks)


def spd(N, seed):
    rng = np.random.default_rng(seed)
    A = rng.uniform(-1, 1, size=(N, N))
    return A @ A.T / N + 1e-3 * np.eye(N)
