#!/usr/bin/env python
"""How far each way of evaluating the Joseph update lands from an extended-precision evaluation of the reference's
expression (estimator.cpp:1257-1288 in numpy longdouble, 64-bit mantissa), as cond(S) grows:
  oracle     the as-coded fp64 sequence on the CPU (oracle/xivo_oracle.py)
  in_solve   library default: whitened form P - (W - D)^T (W + D) inside the solve kernel (trsm_lds_f64_kernel<.,4>)
  fused      the one-kernel route of round 6 at the same form (fused_update_f64_kernel); in_solve runs with XIVO_HIP_FLAG_MULTI_KERNEL
  reassoc    XIVO_HIP_FLAG_STANDALONE_TAIL: T = K(HP) - P, G = T H^T + K R, P+ = G K^T - T from stand-alone kernels
  symmetric  XIVO_HIP_FLAG_SYMMETRIC_FORM: P - W^T W
  as_coded_device  XIVO_HIP_FLAG_DENSE_H: the as-coded product sequence (A = KH - I, A P A^T + K R K^T) on the device
Prints one JSON object; run on a GPU box:  python tests/joseph_forms_accuracy.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

N, F, B = 150, 40, 2


def cases():
    from xivo_amd import synth
    rng = np.random.default_rng(3)
    out = []
    for decades, r in ((2, 1e-2), (5, 1e-4), (8, 1e-6), (11, 1e-9)):
        _, H, inn, _ = synth.s_level(N, F, B, seed=5)
        H *= 0.05
        P = np.empty((B, N, N))
        for b in range(B):
            Q, _ = np.linalg.qr(rng.normal(size=(N, N)))
            P[b] = (Q * np.logspace(-decades, 0, N)) @ Q.T
            P[b] = 0.5 * (P[b] + P[b].T)
        out.append((P, H, inn, np.full((B, 2 * F), r)))
    return out


def extended(H, P, R):
    ld = np.longdouble
    H, P = H.astype(ld), P.astype(ld)
    S = H @ P @ H.T + np.diag(R.astype(ld))
    # gain by extended-precision Cholesky + substitution (numpy.linalg has no longdouble solve)
    M = S.shape[0]
    L = np.zeros_like(S)
    for j in range(M):
        L[j, j] = np.sqrt(S[j, j] - L[j, :j] @ L[j, :j])
        L[j + 1:, j] = (S[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    Y = np.zeros_like(H @ P)
    HP = H @ P
    for i in range(M):
        Y[i] = (HP[i] - L[i, :i] @ Y[:i]) / L[i, i]
    Kt = np.zeros_like(Y)
    for i in range(M - 1, -1, -1):
        Kt[i] = (Y[i] - L[i + 1:, i] @ Kt[i + 1:]) / L[i, i]
    K = Kt.T
    A = np.eye(P.shape[0], dtype=ld) - K @ H
    return A @ P @ A.T + (K * R.astype(ld)) @ K.T, np.linalg.cond(S.astype(np.float64))


def gpu(flags):
    from xivo_amd.lib import Context
    res = []
    for (P, H, inn, dR) in cases():
        with Context(N, 2 * F, B, flags=flags) as ctx:
            ctx.upload_P(P); ctx.set_measurements(H, inn, dR); ctx.update_joseph()
            res.append((ctx.download_P().tolist(), ctx.get_status().tolist()))
    return res


if __name__ == "__main__":
    if len(sys.argv) > 1:                       # child: one library configuration, P+ as JSON
        print(json.dumps(gpu(int(sys.argv[1]))))
        sys.exit(0)
    import xivo_oracle as orc
    from xivo_amd.lib import FLAG_SYMMETRIC_FORM

    def child(flags, env_extra):
        env = dict(os.environ); env.update(env_extra)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(flags)], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])

    from xivo_amd.lib import FLAG_DENSE_H
    from xivo_amd.lib import FLAG_STANDALONE_TAIL
    from xivo_amd.lib import FLAG_MULTI_KERNEL
    runs = {"fused": child(0, {}), "in_solve": child(FLAG_MULTI_KERNEL, {}), "reassoc": child(FLAG_STANDALONE_TAIL, {}),
            "symmetric": child(FLAG_SYMMETRIC_FORM, {}), "as_coded_device": child(FLAG_DENSE_H, {})}
    rel = lambda a, b: float(np.linalg.norm((a - b).astype(np.float64)) / np.linalg.norm(b.astype(np.float64)))
    table = []
    for ci, (P, H, inn, dR) in enumerate(cases()):
        row = {}
        for b in range(B):
            ref, cond = extended(H[b], P[b], dR[b])
            row.setdefault("cond_S", []).append(cond)
            _, P_or, _ = orc.update_joseph(H[b], P[b], inn[b], dR[b])
            row.setdefault("oracle", []).append(rel(P_or.astype(np.longdouble), ref))
            for name, r in runs.items():
                Pg = np.array(r[ci][0][b])
                row.setdefault(name, []).append(rel(Pg.astype(np.longdouble), ref))
                row.setdefault(name + "_min_eig_over_max", []).append(float(np.linalg.eigvalsh(Pg).min() / np.linalg.eigvalsh(Pg).max()))
                row.setdefault(name + "_status", []).append(r[ci][1][b])
        table.append({k: (max(v) if k != "cond_S" and "min_eig" not in k else (min(v) if "min_eig" in k else max(v))) for k, v in row.items()})
    print(json.dumps(table, indent=1))
