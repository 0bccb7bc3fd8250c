import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from xivo_amd import pcw, sequence, lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = sequence.SequenceConfig()
sims = [pcw.TrajectorySim(seed=b) for b in range(B)]
poses = sequence.initial_poses(cfg, sims)
P0 = np.repeat(cfg.P_init()[None], B, axis=0)
be = sequence.HipBackend(cfg, B, poses, P0, flags=L.FLAG_PROFILE)
rng = np.random.default_rng(0)
ops = []
fx, cx, cy = 275.0, 320.0, 240.0
std = np.array([1 / fx, 1 / fx, 0.1]); P3 = np.diag(std * std).reshape(-1)
for b in range(B):
    ops.append(sequence._op(b, L.EDIT_ADD_GROUP, 0))
    for j in range(30):
        xp = rng.uniform([80, 60], [560, 420])
        x = [(xp[0] - cx) / fx, (xp[1] - cy) / fx, np.log(rng.uniform(1, 6))]
        ops.append(sequence._op(b, L.EDIT_ADD_FEATURE, j, j, 0, v=np.concatenate([x, xp + rng.normal(size=2), P3])))
be.edit(np.array(ops, dtype=L.edit_dtype))
c = be.ctx
def t(fn, n=5):
    fn(); c.sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    c.sync(); return (time.perf_counter() - t0) / n * 1e3
c.profile_reset()
print("B", B, "filter_update wall ms", round(t(lambda: c.filter_update(1.0, 5.991, 1.1, 5, True)), 3))
pr = c.profile_get()
print({k: (round(v["ms"] / max(v["launches"], 1), 4), v["launches"], v["kernel"][:50]) for k, v in pr.items() if v["launches"]})
print("get_gate ms", round(t(lambda: c.get_gate(30)), 3), "absorb ms", round(t(lambda: c.absorb_error()), 3), "path", c.last_path())
