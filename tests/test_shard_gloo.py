"""N > 1 path on CPU: two gloo processes shard the (gas, band) task table and issue the single
final all-reduce that bench.py uses (RCCL on the GPU box)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ecckd_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tasks = shard.task_table(["composite", "h2o", "o3", "co2", "ch4", "n2o"], 13)
    mine = shard.shard_tasks(len(tasks), rank, world)
    # stand-in for the per-task work: deterministic "passes" and "cost" per task
    passes = sum(10.0 + (t % 7) for t in mine)
    cost = sum(0.01 * (t + 1) for t in mine)
    elapsed = 1.0 + 0.5 * rank
    out = shard.reduce_scalars(elapsed, passes, cost)
    q.put((rank, mine, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_single_allreduce():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    all_tasks = sorted(res[0][1] + res[1][1])
    assert all_tasks == list(range(78))                     # 6 gases x 13 bands, each exactly once
    assert set(res[0][1]).isdisjoint(res[1][1])
    exp_passes = sum(10.0 + (t % 7) for t in range(78))
    exp_cost = sum(0.01 * (t + 1) for t in range(78))
    for _, _, (tmax, passes, cost) in res:
        assert tmax == 1.5
        assert passes == exp_passes
        assert cost == pytest.approx(exp_cost, rel=1e-15)
    assert res[0][2] == res[1][2]                           # bit-identical on every rank


def test_shard_tasks_properties():
    sys.path.insert(0, ROOT)
    from ecckd_amd import shard
    for n in (0, 1, 7, 78):
        for w in (1, 2, 8):
            got = sorted(sum((shard.shard_tasks(n, r, w) for r in range(w)), []))
            assert got == list(range(n))
            sizes = [len(shard.shard_tasks(n, r, w)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.shard_tasks(5, 2, 2)
    # without an initialised process group the reduction is the identity
    assert shard.reduce_scalars(1.0, 2.0, 3.0) == (1.0, 2.0, 3.0)
