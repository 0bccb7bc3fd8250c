"""Run B point-cloud-world sequences at once on one MI355X (BASELINE configs 1 / 5 surrogate: TUM-VI sizes, N = 203,
<= 30 features) and report tracking error + where the time goes. Mirrors scripts/pyxivo_pcw.py's options."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xivo_amd import formats, pcw, sequence  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-sequences", type=int, default=256)
    ap.add_argument("-npts", type=int, default=1000)
    ap.add_argument("-total_time", type=float, default=2.0)
    ap.add_argument("-imu_dt", type=float, default=0.0025)
    ap.add_argument("-vision_dt", type=float, default=0.04)
    ap.add_argument("-noise_vision_std", type=float, default=1.0)
    ap.add_argument("-integration_method", default="PrinceDormand")
    ap.add_argument("-as_coded_group_block", action="store_true", help="reproduce src/feature.cpp:675-676")
    ap.add_argument("-host", default="python", choices=["python", "cpp"],
                    help="host side of the frame loop: xivo_amd/sequence.py or xivo::hip::BatchEstimator (C++)")
    ap.add_argument("-vectorized", action="store_true",
                    help="thousands of sequences: vectorised simulators + the C++ host side (xivo_amd.sequence.run_pcw_batch)")
    ap.add_argument("-dump", default="", help="directory for per-sequence `ts Tsb Wsb` trajectories")
    ap.add_argument("-gpus", type=int, default=1,
                    help="BASELINE config 5 without a launcher: spawn this many ranks (one per GPU), sequence s on rank s mod gpus")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from xivo_amd.shard import spawn_ranks
        sys.exit(spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], a.gpus))
    if a.vectorized:
        cfg = sequence.SequenceConfig(integration_method=a.integration_method, fix_group_block=not a.as_coded_group_block)
        tm = {}
        t0 = time.perf_counter()
        out = sequence.run_pcw_batch(cfg, a.sequences, total_time=a.total_time, imu_dt=a.imu_dt, vision_dt=a.vision_dt,
                                     noise_vision_std=a.noise_vision_std, npts=a.npts, timers=tm)
        wall = time.perf_counter() - t0
        st = out["estimator"].stats(); out["estimator"].close()
        frames = len(out["ts"])
        ate = np.sqrt(np.mean(np.sum((out["Tsb"] - out["gt_Tsb"]) ** 2, axis=2), axis=0))
        print(json.dumps({
            "sequences": a.sequences, "frames_per_sequence": frames, "N": cfg.N, "integration": a.integration_method,
            "host": "cpp, vectorised simulators",
            "ate_m": {"median": float(np.median(ate)), "p90": float(np.quantile(ate, 0.9)), "max": float(ate.max())},
            "updates": st["updates"], "mh_rejected": st["mh_rejected"], "wall_s": wall, "simulator_s": tm.get("sim", 0.0),
            "frame_calls_s": tm.get("frame", 0.0), "host_cpp_lifecycle_s": st["host_seconds"],
            "frames_per_s_in_frame_calls": a.sequences * frames / tm["frame"],
            "ms_per_frame_of_all_sequences": 1e3 * tm["frame"] / frames}))
        return
    # BASELINE config 5: under torch.distributed.run, sequence s runs on rank s mod world (one rank per GPU, no data-path
    # collective; the ranks only meet to add up the report)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    from xivo_amd.shard import sequence_to_gpu
    mine = [s_ for s_ in range(a.sequences) if sequence_to_gpu(s_, world) == rank]
    B = len(mine)
    cfg = sequence.SequenceConfig(integration_method=a.integration_method, fix_group_block=not a.as_coded_group_block)
    worlds = [pcw.RandomPCW(npts=a.npts, seed=b) for b in mine]
    sims = [pcw.TrajectorySim("lissajous" if b % 2 == 0 else "trefoil", rate=0.08 + 0.04 * (b % 7) / 7, seed=1000 + b)
            for b in mine]
    timers = {}
    device = int(os.environ.get("LOCAL_RANK", 0))
    t0 = time.perf_counter()
    if a.host == "cpp":
        out = sequence.run_pcw_cpp(cfg, worlds, sims, total_time=a.total_time, imu_dt=a.imu_dt, vision_dt=a.vision_dt,
                                   noise_vision_std=a.noise_vision_std, device=device)
        st = out["estimator"].stats()
        timers = {"host_cpp_lifecycle": st["host_seconds"]}

        class _R:      # the report below reads these two counters
            n_updates, n_rejected = st["updates"], st["mh_rejected"]
        out["runner"] = _R
        out["estimator"].close()
    else:
        out = sequence.run_pcw(lambda c_, B_, p_, P_: sequence.HipBackend(c_, B_, p_, P_, device=device), cfg, worlds, sims,
                               total_time=a.total_time, imu_dt=a.imu_dt, vision_dt=a.vision_dt,
                               noise_vision_std=a.noise_vision_std, timers=timers)
        out["backend"].close()
    wall = time.perf_counter() - t0
    frames = len(out["ts"])
    ate = np.array([formats.ate_rmse(out["Tsb"][:, b], out["gt_Tsb"][:, b], align=False) for b in range(B)])
    dev = sum(timers.get(k, 0.0) for k in ("propagate", "edit", "update"))
    r = out["runner"]
    if a.dump:
        os.makedirs(a.dump, exist_ok=True)
        for k, b in enumerate(mine):
            formats.write_trajectory(os.path.join(a.dump, "seq%04d.txt" % b), out["ts"], out["Tsb"][:, k], out["Wsb"][:, k])
    n_upd, n_rej = r.n_updates, r.n_rejected
    if dist is not None:
        parts = [None] * world
        dist.all_gather_object(parts, (ate.tolist(), wall, dev, n_upd, n_rej))
        ate = np.array(sum((p_[0] for p_ in parts), []))
        wall, dev = max(p_[1] for p_ in parts), max(p_[2] for p_ in parts)      # whole job = slowest rank
        n_upd, n_rej = sum(p_[3] for p_ in parts), sum(p_[4] for p_ in parts)
        B = a.sequences
    if rank != 0:
        return
    print(json.dumps({
        "sequences": B, "n_gpus": world, "frames_per_sequence": frames, "imu_samples_per_frame": int(round(a.vision_dt / a.imu_dt)),
        "N": cfg.N, "max_features": cfg.n_features, "integration": a.integration_method, "host": a.host,
        "ate_m": {"median": float(np.median(ate)), "p90": float(np.quantile(ate, 0.9)), "max": float(ate.max())},
        "updates": n_upd, "mh_rejected": n_rej,
        "wall_s": wall, "device_path_s": dev,
        "phase_s": {k: round(v, 4) for k, v in sorted(timers.items())},
        "device_frames_per_s": B * frames / dev if dev > 0 else None,
        "ms_per_frame_per_batch": {k: round(1e3 * timers.get(k, 0.0) / frames, 3) for k in ("propagate", "edit", "update")},
    }))


if __name__ == "__main__":
    main()
