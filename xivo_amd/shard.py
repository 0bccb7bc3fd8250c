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


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def plan_affinity(local_rank, local_world, device_nodes, node_cpus, all_cpus):
    """Cores for one rank of a one-rank-per-GPU job. device_nodes[r] = NUMA node of rank r's GPU (-1 unknown),
    node_cpus = {node: [cpu, ...]}. Ranks whose GPUs sit on the same node split that node's cores evenly (in rank order);
    a rank whose node is unknown gets an even slice of all cores. Pure function: the CPU test drives it with made-up
    topologies."""
    node = device_nodes[local_rank] if local_rank < len(device_nodes) else -1
    if node >= 0 and node_cpus.get(node):
        peers = [r for r in range(local_world) if r < len(device_nodes) and device_nodes[r] == node]
        pool = sorted(node_cpus[node])
    else:
        node, peers, pool = -1, list(range(local_world)), sorted(all_cpus)
    k = peers.index(local_rank) if local_rank in peers else 0
    per = max(1, len(pool) // max(1, len(peers)))
    mine = pool[k * per:(k + 1) * per] or pool
    return {"numa_node": node, "cpus": mine, "ranks_on_node": len(peers)}


def bind_rank(local_rank, local_world, device_count=None, numa_node_of=None):
    """Binds this process to the cores next to its GPU (one rank per GPU) and sizes the OpenMP team of the C++ host side
    (xivo::hip::BatchEstimator) to its share. Returns what it did for the bench line; never raises."""
    import os
    info = {"numa_node": -1, "n_cpus": None, "bound": False}
    try:
        all_cpus = sorted(os.sched_getaffinity(0))
        ndev = device_count or local_world
        nodes = [(numa_node_of(r % ndev) if numa_node_of else -1) for r in range(local_world)]
        node_cpus = {}
        for nd in set(x for x in nodes if x >= 0):
            try:
                with open(f"/sys/devices/system/node/node{nd}/cpulist") as f:
                    node_cpus[nd] = [c for c in _parse_cpulist(f.read()) if c in set(all_cpus)]
            except OSError:
                pass
        plan = plan_affinity(local_rank, local_world, nodes, node_cpus, all_cpus)
        info.update(numa_node=plan["numa_node"], n_cpus=len(plan["cpus"]), ranks_on_node=plan["ranks_on_node"])
        if local_world > 1:
            os.sched_setaffinity(0, plan["cpus"])
            info["bound"] = True
        # BatchEstimator caps its team at 8 threads; never more than this rank's share of the cores
        os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(8, len(plan["cpus"])))))
        info["omp_threads"] = int(os.environ["OMP_NUM_THREADS"])
    except Exception as e:   # affinity is an optimisation, not a requirement
        info["error"] = repr(e)
    return info


def gather_objects(dist, obj):
    """Every rank's (small, picklable) object on every rank, in rank order."""
    if dist is None:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def spawn_ranks(argv, n, python=None):
    """`python bench.py --gpus N` / `run_pcw.py --gpus N` without a launcher: start the N ranks ourselves with the
    environment torch.distributed.run would give them (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR / PORT on
    127.0.0.1). Rank 0 inherits stdout (it prints the one JSON line); the other ranks write stdout + stderr to
    <XIVO_RANK_LOG_DIR or ./gpurun_out/rank_logs>/rank<r>.log, and the tail of the log of any rank that fails is copied to
    stderr. Returns the first non-zero exit code."""
    import os
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    log_dir = os.environ.get("XIVO_RANK_LOG_DIR") or os.path.join(os.getcwd(), "gpurun_out", "rank_logs")
    os.makedirs(log_dir, exist_ok=True)
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        log = None if r == 0 else open(os.path.join(log_dir, f"rank{r}.log"), "w")
        logs.append(log)
        procs.append(subprocess.Popen([python or sys.executable] + list(argv), env=env,
                                      stdout=log, stderr=subprocess.STDOUT if log else None))
    rc = 0
    for r, p in enumerate(procs):
        p.wait()
        if logs[r]:
            logs[r].close()
            if p.returncode:
                with open(os.path.join(log_dir, f"rank{r}.log")) as f:
                    sys.stderr.write(f"[rank {r} exited with {p.returncode}] " + f.read()[-2000:] + "\n")
        rc = rc or p.returncode
    return rc
