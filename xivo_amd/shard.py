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
