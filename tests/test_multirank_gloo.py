"""world_size-2 gloo test of the N>1 path of bench.py: filters shard with no data-path
collective; the only communication is the barrier and the max-over-ranks timing."""
import os
import socket

import pytest


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from xivo_amd.shard import shard_range, max_over_ranks, sum_over_ranks
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(11, world, rank)
    dist.barrier()
    tmax = max_over_ranks(dist, 1.0 + rank)           # rank 1 is "slower"
    total = sum_over_ranks(dist, hi - lo)
    q.put((rank, lo, hi, tmax, total))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing():
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps: p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps: p.join(60)
    assert [(r[1], r[2]) for r in res] == [(0, 6), (6, 11)]       # disjoint, covering, sizes differ by <= 1
    assert all(r[3] == 2.0 for r in res)                           # max over ranks
    assert all(r[4] == 11 for r in res)                            # every filter owned exactly once


def test_shard_range_properties():
    from xivo_amd.shard import shard_range, sequence_to_gpu
    for n in (0, 1, 7, 8, 1024, 1027):
        for w in (1, 2, 4, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert [sequence_to_gpu(s, 8) for s in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]


def _seq_worker(rank, world, port, q):
    """BASELINE config 5 shape: sequence s runs on rank s mod world (scripts/run_pcw.py); here with the oracle backend."""
    import numpy as np
    import torch.distributed as dist
    from seq_oracle import OracleBackend
    from xivo_amd import pcw, sequence
    from xivo_amd.shard import sequence_to_gpu
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = [s for s in range(3) if sequence_to_gpu(s, world) == rank]
    cfg = sequence.SequenceConfig(n_groups=4, n_features=8)
    out = sequence.run_pcw(OracleBackend, cfg, [pcw.RandomPCW(npts=300, seed=s) for s in mine],
                           [pcw.TrajectorySim(seed=50 + s) for s in mine], total_time=0.12)
    parts = [None] * world
    dist.all_gather_object(parts, {s: out["Tsb"][:, k].tolist() for k, s in enumerate(mine)})
    q.put((rank, parts))
    dist.barrier()
    dist.destroy_process_group()


def test_sequences_shard_over_ranks_without_data_exchange():
    """Three sequences over two ranks (s mod world) give exactly the trajectories of one process running all three:
    the path shards by sequence, the ranks only meet to collect the report."""
    torch = pytest.importorskip("torch")
    import numpy as np
    import torch.multiprocessing as mp
    from seq_oracle import OracleBackend
    from xivo_amd import pcw, sequence
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_seq_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps: p.start()
    res = sorted(q.get(timeout=300) for _ in ps)
    for p in ps: p.join(60)
    merged = {}
    for part in res[0][1]:
        merged.update(part)
    assert sorted(merged) == [0, 1, 2] and res[0][1] == res[1][1]
    assert sorted(res[0][1][0]) == [0, 2] and sorted(res[0][1][1]) == [1]
    cfg = sequence.SequenceConfig(n_groups=4, n_features=8)
    one = sequence.run_pcw(OracleBackend, cfg, [pcw.RandomPCW(npts=300, seed=s) for s in range(3)],
                           [pcw.TrajectorySim(seed=50 + s) for s in range(3)], total_time=0.12)
    for s in range(3):
        assert np.array_equal(np.array(merged[s]), one["Tsb"][:, s])


def test_bench_gpus_2_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts the two ranks itself (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* on 127.0.0.1), they rendezvous over gloo, rank 0 prints one JSON line with n_gpus = 2 and one
    rate per rank. --dry-run = the launcher / barrier / reduction path without device work (no GPU on this box)."""
    import json
    import subprocess
    import sys
    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                       # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["dry_run"] is True
    assert len(out["per_rank_updates_per_s"]) == 2
    # the job time is the slower rank's (rank 1 sleeps twice as long per step in the dry path)
    assert out["ms_per_step"] >= 2.0
    # self-verifying multi-rank line: one parity record and one affinity record per rank, gathered on rank 0
    assert [p["rank"] for p in out["per_rank_parity"]] == [0, 1]
    aff = out["per_rank_affinity"]
    assert len(aff) == 2 and all(a["bound"] and a["omp_threads"] >= 1 for a in aff)
    assert sum(a["n_cpus"] for a in aff) <= (os.cpu_count() or 1)        # the ranks share the cores, they do not overlap


def test_eight_rank_dry_run_and_affinity_plan(tmp_path):
    """The 8-GPU launch of BASELINE config 5, as far as a CPU box can take it: eight ranks rendezvous, every rank reports,
    ranks > 0 log to files; and the core plan for a two-socket node (GPUs 0-3 on node 0, 4-7 on node 1)."""
    import json
    import subprocess
    import sys
    pytest.importorskip("torch")
    from xivo_amd.shard import plan_affinity
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    node_cpus = {0: list(range(0, 96)), 1: list(range(96, 192))}
    plans = [plan_affinity(r, 8, nodes, node_cpus, list(range(192))) for r in range(8)]
    assert all(len(p["cpus"]) == 24 and p["ranks_on_node"] == 4 for p in plans)
    assert all(set(plans[r]["cpus"]) <= set(node_cpus[nodes[r]]) for r in range(8))
    allc = [c for p in plans for c in p["cpus"]]
    assert len(allc) == len(set(allc)) == 192                             # disjoint and complete
    unknown = plan_affinity(3, 8, [-1] * 8, {}, list(range(64)))
    assert unknown["numa_node"] == -1 and unknown["cpus"] == list(range(24, 32))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["XIVO_RANK_LOG_DIR"] = str(tmp_path)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 8 and len(out["per_rank_parity"]) == 8 and len(out["per_rank_affinity"]) == 8
    assert sorted(f for f in os.listdir(tmp_path)) == [f"rank{i}.log" for i in range(1, 8)]
