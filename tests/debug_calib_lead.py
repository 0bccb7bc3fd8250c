"""Debugging aid (not collected by pytest; lives under tests/ because it builds its scene with the tests' helpers, which use
the oracle): the intermediate products of one online-calibration update on the sparse pipeline against numpy.
XIVO_HIP_DUMP_DIR=<dir> python tests/debug_calib_lead.py"""
import os, sys, tempfile
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
d = os.environ.get("XIVO_HIP_DUMP_DIR") or tempfile.mkdtemp(); os.makedirs(d, exist_ok=True); os.environ["XIVO_HIP_DUMP_DIR"] = d
import test_calib_gpu as T
from scene_util import spd
cam, lay, sc, poses, groups, feats, xp, calib, cals, ctx = T.setup("pinhole", True, True, True, B=3, ng=8, nf=20, seed=11)
B, F = poses.shape[0], feats.shape[1]
P = np.array([spd(lay.N, 80 + b) * 1e-4 for b in range(B)])
with ctx:
    ctx.upload_P(P); ctx.set_scene(poses, groups, feats); ctx.set_calib_state(calib)
    ctx.filter_update(T.R_VIS, T.MH, T.MULT, 5, use_gating=False)
    print("path", ctx.last_path())
    H, inn, dR = ctx.get_H(0)
np.savez(os.path.join(d, "host.npz"), H=H, P=P[0], dR=dR, inn=inn)
M, N = H.shape; Np = -(-N // 16) * 16; Mp = -(-M // 16) * 16
ld = lambda n: np.fromfile(os.path.join(d, n + ".f64"))
PHT = ld("PHT")[:Np * Mp].reshape(Mp, Np).T[:N, :M]; HP = ld("HP")[:Mp * Np].reshape(Np, Mp).T[:M, :N]
S = ld("S")[:Mp * Mp].reshape(Mp, Mp).T[:M, :M]; L = ld("Hlead")[:Mp * 48].reshape(48, Mp).T[:M]
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
Hl = np.zeros_like(H); cols = [lay.td] + list(range(lay.Cg, lay.Cg + 9)) + [9, 10, 11] + list(range(lay.cam_begin, lay.cam_begin + lay.cam_dim))
Hl[:, cols] = H[:, cols]
print("lead block", rel(L, Hl[:, :48]), "nonzero cols", np.nonzero(np.abs(L).sum(0))[0])
print("PHT", rel(PHT, P[0] @ H.T), "HP", rel(HP, H @ P[0]), "PHT without lead", rel(PHT, P[0] @ (H - Hl).T))
Sref = H @ P[0] @ H.T + np.diag(dR)
print("S lower", rel(np.tril(S), np.tril(Sref)), "S", rel(S, Sref))

E = PHT - P[0] @ H.T
print("PHT err by state row block of 16:", [float(np.round(np.linalg.norm(E[i:i + 16]) / np.linalg.norm((P[0] @ H.T)[i:i + 16]), 3)) for i in range(0, N, 16)])
print("PHT err by measurement column:", np.round(np.linalg.norm(E, axis=0) / np.linalg.norm(P[0] @ H.T, axis=0), 3))
print("HP vs PHT^T", rel(HP, PHT.T))
E0 = PHT - P[0] @ (H - Hl).T
print("vs ell only, by column:", np.round(np.linalg.norm(E0, axis=0) / np.linalg.norm(P[0] @ H.T, axis=0), 3))
G = PHT - P[0] @ (H - Hl).T; Gref = P[0] @ Hl.T
print("lead part: got/ref norms", np.linalg.norm(G), np.linalg.norm(Gref), "rel", rel(G, Gref), "rel vs transposed-P form", rel(G, P[0][:, :48] @ L.T))
