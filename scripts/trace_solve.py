"""Reads the phase stamps of the XIVO_TRACE build (scripts/trace_solve.sh) after a few bench-style steps at the metric point
and prints where the in-solve update kernel's time goes (median over the sampled workgroups, shader-clock cycles and us at
the clock inferred from the kernel's event time)."""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("XIVO_HIP_LIBRARY", os.path.join(ROOT, "xivo_amd", "csrc", "build", "abl", "libxivo_hip_trace.so"))
from xivo_amd import synth
from xivo_amd.lib import Context, load_library
N, F, B = 250, 80, int(sys.argv[1]) if len(sys.argv) > 1 else 16384
M = 2 * F
P, H, inn, dR = synth.s_level(N, F, 64, seed=1000)
ctx = Context(N, M, B)
for b0 in range(0, B, 64):
    ctx.upload_P(P[:min(64, B - b0)], b0=b0)
dH = ctx.device_array(np.transpose(H, (0, 2, 1)), total=B); di = ctx.device_array(inn, total=B); dr = ctx.device_array(dR, total=B)
ctx.snapshot_P()
for _ in range(3):
    ctx.set_measurements_device(dH, di, dr, M, B); ctx.update_dense_gated(F, 2.25, 5.991, 1.1, 5, B)
ctx.sync()
lib = load_library()
n = 512 * 32
buf = (C.c_ulonglong * n)()
assert lib.xivo_hip_debug_read_trace(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(512, 32).astype(np.int64)
t = t[: B // 64]
t = t[t[:, 0] > 0]
names = ["0 start", "1 rhs issued + factor copied", "2 barrier", "3 forward done", "4 stash written", "5 backward done", "6 barrier (factor dead)"]
rows = []
for i in range(1, 7):
    rows.append((names[i], np.median(t[:, i] - t[:, i - 1])))
nph = 4
prev = 6
for p in range(nph):
    rows.append((f"phase {p}: to barrier wait", np.median(t[:, 8 + 3 * p] - t[:, prev])))
    rows.append((f"phase {p}: vmcnt(0) drain", np.median(t[:, 9 + 3 * p] - t[:, 8 + 3 * p])))
    rows.append((f"phase {p}: barrier", np.median(t[:, 10 + 3 * p] - t[:, 9 + 3 * p])))
    prev = 10 + 3 * p
rows.append(("last phase tiles -> end", np.median(t[:, 8 + 3 * nph] - t[:, prev])))
tot = np.median(t[:, 8 + 3 * nph] - t[:, 0])
print(json.dumps({"sampled_workgroups": int(len(t)), "total_cycles_median": float(tot),
                  "segments_cycles": {k: float(v) for k, v in rows}}, indent=1))

# ---- second level: every wave inside the product phases
n2 = 128 * 16 * 4 * 16
buf2 = (C.c_ulonglong * n2)()
if hasattr(lib, "xivo_hip_debug_read_trace2") and lib.xivo_hip_debug_read_trace2(buf2, n2) == 0:
    t2 = np.frombuffer(buf2, dtype=np.uint64).reshape(128, 16, 4, 16).astype(np.int64)[: min(128, B // 64)]
    # per phase: time origin = the earliest "after barrier" stamp of the workgroup
    print("per wave, phase 1 (cycles after the phase barrier, median over workgroups): barrier-exit fixup-done | per tile: mfma-done, stored | ... | loop end")
    ph = 1
    org = t2[:, :, ph, 0].min(axis=1)[:, None, None]
    rel = np.where(t2[:, :, ph, :] > 0, t2[:, :, ph, :] - org, -1)
    med = np.median(rel, axis=0)
    for wv in range(16):
        print("wave %2d:" % wv, " ".join("%6d" % v for v in med[wv][[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 14]]))
