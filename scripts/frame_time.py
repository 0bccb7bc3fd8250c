"""Per-frame stage costs of the resident loop at 4096 filters (N = 251, 60 features): run on the GPU box."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from xivo_amd import synth
from xivo_amd.lib import Context, imu_dtype, FLAG_PROFILE
B, ng, nf = 4096, 8, 60
cam = synth.PINHOLE
sc = synth.g_level(ng, nf, nf, 64, seed=1, cam=cam, N=251)
N = sc["N"]
poses, groups, feats = synth.scene_structs(sc)
rep = B // 64
poses = np.tile(poses, rep); groups = np.tile(groups, (rep, 1)); feats = np.tile(feats, (rep, 1))
ctx = Context(N, 2 * nf, B, flags=0)
ctx.set_layout(N, 23, ng, 23 + 6 * ng, nf, cam)
P = synth.spd_covariance(N, 3, 1e-4)[None]
for b0 in range(0, B, 64): ctx.upload_P(np.repeat(P, 64, axis=0), b0)
ctx.set_scene(poses, groups, feats)
ctx.jacobians_instate()                                   # measured pixel = the device's own prediction + noise
_, inn0 = ctx.get_jacobians()
feats["xp"] = -inn0 + np.random.default_rng(7).normal(0, 1.5, inn0.shape)
ctx.set_scene(poses, groups, feats)
imu = np.zeros(B, dtype=imu_dtype); imu["accel"][:, 2] = 9.8; imu["dt"] = 0.005
Qi = np.eye(12) * 1e-6; Qm = np.eye(23) * 1e-8; g = np.array([0, 0, -9.8])
def t(fn, n=5):
    fn(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    ctx.sync(); return (time.perf_counter() - t0) / n * 1e3
print("propagate RK4 ms/4096 (incl. H2D of IMU structs, sync):", round(t(lambda: ctx.propagate(imu, Qi, Qm, g, "RK4", 0.002)), 3))
imu10 = np.repeat(imu[:, None], 10, axis=1)
print("propagate RK4, 10 IMU samples per call ms/4096:", round(t(lambda: ctx.propagate(imu10, Qi, Qm, g, "RK4", 0.002)), 3))
print("propagate PD  ms/4096:", round(t(lambda: ctx.propagate(imu, Qi, Qm, g, "PrinceDormand", 0.002)), 3))
print("filter_update ms/4096:", round(t(lambda: ctx.filter_update(2.25, 5.991, 1.1, 5, True)), 3))
print("absorb_error  ms/4096:", round(t(lambda: ctx.absorb_error()), 3))
print("set_scene     ms/4096 (H2D 16 MB):", round(t(lambda: ctx.set_scene(poses, groups, feats)), 3))
