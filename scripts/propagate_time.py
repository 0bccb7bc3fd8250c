"""GPU time of xivo_hip_propagate's two kernels (HIP events via the library's profile API) at 4096 filters, N = 251:
RK4 / Dormand-Prince, 1 and 16 IMU samples per call (dt 2.5 ms, stepsize 2 ms: two sub-steps per sample)."""
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
ctx = Context(N, 2 * nf, B, flags=FLAG_PROFILE)
ctx.set_layout(N, 23, ng, 23 + 6 * ng, nf, cam)
P = synth.spd_covariance(N, 3, 1e-4)[None]
for b0 in range(0, B, 64): ctx.upload_P(np.repeat(P, 64, axis=0), b0)
ctx.set_scene(poses, groups, feats)
ctx.jacobians_instate()                                   # measured pixel = the device's own prediction + noise
_, inn0 = ctx.get_jacobians()
feats["xp"] = -inn0 + np.random.default_rng(7).normal(0, 1.5, inn0.shape)
ctx.set_scene(poses, groups, feats)
Qi = np.eye(12) * 1e-6; Qm = np.eye(23) * 1e-8; g = np.array([0, 0, -9.8])
for method in ("RK4", "PrinceDormand"):
    for K in (1, 16):
        imu = np.zeros((B, K), dtype=imu_dtype); imu["accel"][:, :, 2] = 9.8; imu["dt"] = 0.0025
        imu["gyro"] = 0.1; imu["slope_gyro"] = 1.0
        ctx.propagate(imu, Qi, Qm, g, method, 0.002); ctx.profile_reset()
        t0 = time.perf_counter()
        for _ in range(5): ctx.propagate(imu, Qi, Qm, g, method, 0.002)
        wall = (time.perf_counter() - t0) / 5 * 1e3
        pr = {k: v for k, v in ctx.profile_get().items() if v["launches"]}
        print(method, "samples/call", K, "wall ms/call", round(wall, 3),
              {k: round(v["ms"] / v["launches"], 4) for k, v in pr.items()})
