"""Sharding of independent filters / sequences over ranks (replicas
only, no data-path collective). Used by bench.py; covered by a world_size-2 gloo test."""


def shard_range(n_items, world, rank):
    """Contiguous block partition: rank r gets items [lo, hi). Sizes differ by at most 1."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sequence_to_gpu(seq_index, n_gpus):
    """Config 5: sequence s runs on GPU s mod n_gpus."""
    return seq_index % n_gpus


def max_over_ranks(dist, value, device="cpu"):
    """Whole-job time = the slowest rank's time (bench contract)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value, device="cpu"):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(dist, value):
    """Every rank's value, in rank order (rank 0 reports per-rank rates and their min / max)."""
    if dist is None:
        return [float(value)]
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    out = [torch.zeros(1, dtype=torch.float64) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def spawn_ranks(argv, n, python=None):
    """`python bench.py --gpus N` / `run_pcw.py --gpus N` without a launcher: start the N ranks ourselves with the
    environment torch.distributed.run would give them (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR / PORT on
    127.0.0.1). Rank 0 inherits stdout (it prints the one JSON line); returns the first non-zero exit code."""
    import os
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([python or sys.executable] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc
