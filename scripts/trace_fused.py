"""Reads the phase stamps of the XIVO_FUSED_TRACE build of fused_update.hip (scripts/build_variant.sh ftrace "-DXIVO_FUSED_TRACE=1"
fused_update.hip) after a few bench-style steps and prints where the one-kernel update's time goes (median over the sampled
workgroups, shader-clock ticks of s_memtime - 100 MHz on gfx950 - converted to us)."""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("XIVO_HIP_LIBRARY", os.path.join(ROOT, "xivo_amd", "csrc", "build", "abl", "libxivo_hip_ftrace.so"))
from xivo_amd import synth
from xivo_amd.lib import Context, load_library
N = int(sys.argv[1]) if len(sys.argv) > 1 else 203
F = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
M = 2 * F
P, H, inn, dR = synth.s_level(N, F, 64, seed=1000)
ctx = Context(N, M, B)
for b0 in range(0, B, 64):
    ctx.upload_P(P[:min(64, B - b0)], b0=b0)
dH = ctx.device_array(np.transpose(H, (0, 2, 1)), total=B); di = ctx.device_array(inn, total=B); dr = ctx.device_array(dR, total=B)
ctx.snapshot_P()
for _ in range(3):
    ctx.restore_P()
    ctx.set_measurements_device(dH, di, dr, M, B); ctx.update_dense_gated(F, float(dR[0, 0]), 5.991, 1.1, 5, B)
ctx.sync()
lib = load_library()
n = 512 * 16
buf = (C.c_ulonglong * n)()
assert lib.xivo_hip_debug_read_fused_trace(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(512, 16).astype(np.int64)[: max(1, B // 64)]
t = t[t[:, 0] > 0]
names = ["0 start", "1 coefficients staged", "2 H P gathered", "3 S walked", "4 gated", "5 factored", "6 forward", "7 backward + dx", "8 product"]
tick_us = 0.01   # s_memtime counts at 100 MHz
out = {"shape": [N, F, B], "sampled_workgroups": int(len(t)), "total_us_median": float(np.median(t[:, 8] - t[:, 0]) * tick_us),
       "segments_us": {names[i]: float(np.median(t[:, i] - t[:, i - 1]) * tick_us) for i in range(1, 9)}}
out["phase2_us"] = {"gather done -> staging barrier": float(np.median(t[:, 9] - t[:, 2]) * tick_us), "slab write + read-back": float(np.median(t[:, 10] - t[:, 9]) * tick_us),
                    "barrier": float(np.median(t[:, 11] - t[:, 10]) * tick_us), "walk (first pass)": float(np.median(t[:, 12] - t[:, 11]) * tick_us), "rest": float(np.median(t[:, 3] - t[:, 12]) * tick_us)}
print(json.dumps(out, indent=1))

n2 = 128 * 16 * 32
buf2 = (C.c_ulonglong * n2)()
if hasattr(lib, "xivo_hip_debug_read_fused_trace2") and lib.xivo_hip_debug_read_fused_trace2(buf2, n2) == 0:
    t2 = np.frombuffer(buf2, dtype=np.uint64).reshape(128, 16, 32).astype(np.int64)[: min(128, max(1, B // 64))]
    nb = (M + 15) // 16
    org = t2[:, 0, 0][:, None, None]
    rel = np.where(t2 > 0, t2 - org, -1)
    med = np.median(rel, axis=0)
    print("per wave: gather start, gather end, backward end, product end (cycles after wave 0 entered column 0 of the factorisation)")
    for wv in range(min(16, (N + 15) // 16)):
        print("wave %2d:" % wv, " ".join("%7d" % med[wv][q] for q in (28, 29, 30, 31)))
    print("factorisation, per wave (cycles after wave 0 entered column 0): per column j: enter | owner: diagonal update done | arrive at barrier | leave barrier")
    for wv in range(min(16, (N + 15) // 16)):
        print("wave %2d:" % wv, "  ".join(" ".join("%6d" % med[wv][4 * j + q] for q in range(4)) for j in range(nb)))
    print("pivot stamps of block column 0 (wave 0), cycles between consecutive pivots:", " ".join("%d" % v for v in np.diff(med[0][16:32])))
