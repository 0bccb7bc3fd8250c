"""Oracle backend of xivo_amd.sequence.SequenceRunner (TEST INFRASTRUCTURE): the same per-frame calls answered by the numpy
restatement in oracle/xivo_oracle.py, one filter at a time. The runner's decisions (who enters / leaves the state)
depend on the inlier masks the backend returns, so device and oracle runs stay in lock step only if gating agrees."""
import numpy as np

import xivo_oracle as orc
from xivo_amd import lib as L


class OracleBackend:
    def __init__(self, cfg, B, poses0, P0):
        self.cfg, self.B, self.F = cfg, B, cfg.n_features
        self.lay = orc.Layout(cfg.n_groups, cfg.n_features)
        assert self.lay.N == cfg.N
        self.Qimu, self.Qmodel = cfg.Qimu_matrix(), cfg.Qmodel_matrix()
        self.st = []
        for b in range(B):
            p = poses0[b]
            self.st.append(dict(
                Rsb=p["Rsb"].reshape(3, 3).T.copy(), Tsb=p["Tsb"].copy(), Vsb=p["Vsb"].copy(), bg=p["bg"].copy(),
                ba=p["ba"].copy(), Rbc=p["Rbc"].reshape(3, 3).T.copy(), Tbc=p["Tbc"].copy(),
                Rsg=p["Rsg"].reshape(3, 3).T.copy(),
                gR=np.repeat(np.eye(3)[None], cfg.n_groups, axis=0), gT=np.zeros((cfg.n_groups, 3)),
                x=np.zeros((self.F, 3)), xp=np.zeros((self.F, 2)), sind=np.full(self.F, -1), ref=np.full(self.F, -1),
                P=np.array(P0[b], dtype=float)))

    def propagate(self, imu):
        c = self.cfg
        method = "RK4" if c.integration_method == "RK4" else "PD"
        for b, s in enumerate(self.st):
            X = orc.MotionState(s["Rsb"], s["Tsb"], s["Vsb"], s["bg"], s["ba"], s["Rsg"])
            P = s["P"]
            for k in range(imu.shape[1]):
                r = imu[b, k]
                X, P = orc.propagate(X, P, r["gyro"], r["accel"], r["slope_gyro"], r["slope_accel"], float(r["dt"]),
                                     self.Qimu, self.Qmodel, c.gravity, method=method, stepsize=c.stepsize)
            s["Rsb"], s["Tsb"], s["Vsb"], s["bg"], s["ba"], s["Rsg"], s["P"] = X.Rsb, X.Tsb, X.Vsb, X.bg, X.ba, X.Rsg, P

    def edit(self, ops):
        """include/xivo_hip.h XIVO_EDIT_*: per filter in array order"""
        lay = self.lay
        for o in ops:
            s = self.st[int(o["b"])]
            k, i0, i1, i2, v = int(o["kind"]), int(o["i0"]), int(o["i1"]), int(o["i2"]), o["v"]
            if k == L.EDIT_P_ZERO_RC:
                s["P"] = orc.p_zero_rc(s["P"], i0, i1)
            elif k == L.EDIT_P_COPY_RC:
                s["P"] = orc.p_copy_rc(s["P"], i0, i1, i2)
            elif k == L.EDIT_P_SET_BLOCK3:
                s["P"] = orc.p_set_block3(s["P"], i0, v[:9].reshape(3, 3).T)
            elif k == L.EDIT_ADD_GROUP:                       # estimator.cpp:801-816
                s["gR"][i0] = s["Rsb"]; s["gT"][i0] = s["Tsb"]
                off = lay.group_begin + 6 * i0
                s["P"] = orc.p_copy_rc(s["P"], off, 0, 3)
                s["P"] = orc.p_copy_rc(s["P"], off + 3, 3, 3)
            elif k == L.EDIT_REMOVE_GROUP:                    # estimator.cpp:745-759
                s["P"] = orc.p_zero_rc(s["P"], lay.group_begin + 6 * i0, 6)
            elif k == L.EDIT_ADD_FEATURE:                     # estimator.cpp:820-846, feature.cpp:753-760
                s["x"][i0] = v[0:3]; s["xp"][i0] = v[3:5]; s["sind"][i0] = i1; s["ref"][i0] = i2
                off = lay.feature_begin + 3 * i1
                s["P"] = orc.p_zero_rc(s["P"], off, 3)
                s["P"] = orc.p_set_block3(s["P"], off, v[5:14].reshape(3, 3).T)
            elif k == L.EDIT_REMOVE_FEATURE:                  # estimator.cpp:762-783
                if s["sind"][i0] >= 0:
                    s["P"] = orc.p_zero_rc(s["P"], lay.feature_begin + 3 * int(s["sind"][i0]), 3)
                    s["sind"][i0] = -1
            elif k == L.EDIT_SET_XP:
                s["xp"][i0] = v[0:2]
            else:
                raise ValueError(k)

    def set_pixels(self, xp):
        """include/xivo_hip.h xivo_hip_set_pixels: NaN pairs leave the entry untouched"""
        for b, s in enumerate(self.st):
            ok = ~np.isnan(xp[b]).any(axis=1)
            s["xp"][ok] = xp[b][ok]

    def update(self):
        c, lay = self.cfg, self.lay
        R = c.visual_meas_std ** 2
        mask = np.zeros((self.B, self.F), dtype=bool)
        for b, s in enumerate(self.st):
            here = np.nonzero(s["sind"] >= 0)[0]
            if len(here) == 0:
                continue
            Js, inns = [], []
            for j in here:
                r = int(s["ref"][j])
                J, inn, _ = orc.compute_jacobian(s["x"][j], s["xp"][j], s["gR"][r], s["gT"][r], s["Rsb"], s["Tsb"],
                                                 s["Rbc"], s["Tbc"], c.cam, lay, r, int(s["sind"][j]))
                Js.append(J); inns.append(inn)
            Js, inns = np.array(Js), np.array(inns)
            if len(here) > c.min_inliers:                     # src/manager.cpp:635
                m, _, _ = orc.mh_gate(orc.mh_distances(Js, s["P"], inns, R), c.MH_thresh, c.MH_adjust_factor,
                                      c.min_inliers)
            else:
                m = np.ones(len(here), dtype=bool)
            mask[b, here] = m
            H, inn, dR = orc.stack_measurements(Js[m], inns[m], s["ref"][here][m], s["sind"][here][m], lay, R,
                                                   fix_group_block=c.fix_group_block)
            dx, s["P"], _ = orc.update_joseph(H, s["P"], inn, dR)
            orc.absorb_error(s, dx, lay, range(c.n_groups), here[m])
        return mask

    def poses(self):
        return np.array([s["Rsb"] for s in self.st]), np.array([s["Tsb"] for s in self.st])

    def covariance(self):
        return np.array([s["P"] for s in self.st])

    def close(self):
        pass
