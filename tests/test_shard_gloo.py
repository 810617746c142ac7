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
    # the driver's own dealing: contiguous shares of the table, results gathered on rank 0
    dealt = [tasks[t] for t in shard.deal_tasks(len(tasks), rank, world)]
    assert shard.world() == (rank, world)
    gathered = shard.gather_to_root([(g, b, {"error": [0.5 * b]}) for g, b in dealt])
    q.put((rank, mine, out, dealt, gathered))
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
    # contiguous dealing: 39 tasks each, rank 0 = composite, h2o, o3; rank 1 = co2, ch4, n2o; nothing twice, nothing missing
    assert [len(r[3]) for r in res] == [39, 39]
    assert {g for g, _ in res[0][3]} == {"composite", "h2o", "o3"} and {g for g, _ in res[1][3]} == {"co2", "ch4", "n2o"}
    assert res[1][4] is None and len(res[0][4]) == 2
    got = [(g, b) for part in res[0][4] for g, b, _ in part]
    assert got == [(g, b) for g in ["composite", "h2o", "o3", "co2", "ch4", "n2o"] for b in range(13)]
    for _, _, (tmax, passes, cost), _, _ in res:
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


# ---- optimize_lut: profiles sharded over the ranks, one all-reduce of [gradient, cost] per evaluation ----

def _quadratic(seed, ncol, nx):
    rs = np.random.RandomState(seed)
    return rs.normal(size=(ncol, 3, nx)), rs.normal(size=(ncol, 3))      # per-profile rows: J = sum_c 0.5 |A_c x - b_c|^2


def _cost_grad(A, b, x):
    r = np.einsum("cij,j->ci", A, x) - b
    return 0.5 * float((r * r).sum()), np.einsum("cij,ci->j", A, r)


def _opt_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import ctypes as C
    import torch.distributed as dist
    from ecckd_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ncol, nx = 7, 11
    A, b = _quadratic(3, ncol, nx)
    scene = dict(pressure_hl=np.zeros((ncol, 5)), temperature_hl=np.zeros((ncol, 5)), flux_dn=A, flux_up=b, albedo=np.arange(4.0), tsi=1361.0,
                 gas_present=np.ones(3, dtype=np.int32))
    mine = shard.shard_scene_columns(scene, rank, world)
    fn, keep = shard.make_allreduce_callback()
    x = np.linspace(-1.0, 1.0, nx)
    out = []
    for it in range(3):                                       # a few "evaluations": every rank must see the same sums
        J, g = _cost_grad(mine["flux_dn"], mine["flux_up"], x)
        buf = np.concatenate([g, [J]])                        # the layout ecckd_opt hands to the callback: [gradient, cost]
        rc = fn(buf.ctypes.data_as(C.c_void_p), buf.size, None, C.c_void_p(1))      # user != NULL: host buffer (CPU test)
        assert rc == 0
        out.append(buf.copy())
        x = x - 0.01 * buf[:-1]                               # identical step on every rank
    q.put((rank, mine["pressure_hl"].shape[0], mine["albedo"].tolist(), out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_profile_sharded_cost_and_gradient():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_opt_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [4, 3] and res[0][2] == res[1][2] == [0.0, 1.0, 2.0, 3.0]     # 7 profiles -> 4 + 3; shared fields whole
    A, b = _quadratic(3, 7, 11)
    x = np.linspace(-1.0, 1.0, 11)
    for it in range(3):
        J, g = _cost_grad(A, b, x)
        for r in res:
            assert np.allclose(r[3][it][:-1], g, rtol=1e-13, atol=1e-13) and r[3][it][-1] == pytest.approx(J, rel=1e-13)
        assert np.array_equal(res[0][3][it], res[1][3][it])      # bit-identical on both ranks: same L-BFGS decisions
        x = x - 0.01 * res[0][3][it][:-1]


def test_column_range_properties():
    sys.path.insert(0, ROOT)
    from ecckd_amd import shard
    for n in (1, 2, 7, 50, 150):
        for w in (1, 2, 4, 8):
            if n < w:
                continue
            r = [shard.column_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [e - b for b, e in r]
            assert min(sizes) >= 1 and max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.shard_scene_columns(dict(pressure_hl=np.zeros((1, 3))), 1, 2)


def test_deal_tasks_contiguous_and_complete():
    from ecckd_amd import shard
    for ntasks, world in ((104, 8), (78, 8), (9, 2), (3, 8), (13, 1), (0, 4)):
        parts = [shard.deal_tasks(ntasks, r, world) for r in range(world)]
        assert sum(parts, []) == list(range(ntasks))                              # every task once, table order kept
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    # 8 gases x 13 bands on 8 ranks: one gas per rank; 6 gases: a rank touches at most two gases
    table = shard.task_table(range(8), 13)
    assert all({table[t][0] for t in shard.deal_tasks(104, r, 8)} == {r} for r in range(8))
    table = shard.task_table(range(6), 13)
    assert all(len({table[t][0] for t in shard.deal_tasks(78, r, 8)}) <= 2 for r in range(8))
    with pytest.raises(ValueError):
        shard.deal_tasks(10, 3, 3)
    assert shard.world() == (0, 1) and shard.gather_to_root("x") == ["x"]


def _reorder_worker(rank, world, port, q, nwav):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    from ecckd_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rs = np.random.RandomState(7)
    key_all = rs.normal(size=nwav)
    key_all[rs.uniform(size=nwav) < 0.05] = -0.5            # a tie group, as the zero columns of a spectrum give
    col_all = np.abs(rs.normal(size=nwav))
    seen = []

    def key_of_range(b, e):                                 # stand-in for K1 over this rank's wavenumbers
        seen.append((b, e))
        return key_all[b:e], col_all[b:e]

    def sort_on_root(key):                                  # std::stable_sort of the indices by key, then rank[ordered[i]] = i
        order = np.argsort(key.numpy(), kind="stable")
        rank_of = np.empty(nwav, dtype=np.int32)
        rank_of[order] = np.arange(nwav, dtype=np.int32)
        return rank_of

    key, col, rnk = shard.reorder_single_band(key_of_range, nwav, sort_on_root)
    q.put((rank, seen, None if key is None else (key.numpy().tobytes(), col.numpy().tobytes(), rnk.tobytes())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nwav", [70_001, 4096])
def test_single_band_reorder_with_the_key_sweep_split_by_range(nwav):
    """SURVEY 8e, reorder row, one band (the fsck structure): the key sweep split by wavenumber range over two gloo ranks, the
    pieces gathered on rank 0, one stable sort there - the same key, column optical depth and rank as one process gives, the
    ranges whole tiles, disjoint and complete (ragged last share)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reorder_worker, args=(r, world, port, q, nwav)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (b0, e0), (b1, e1) = res[0][1][0], res[1][1][0]
    assert b0 == 0 and e0 == b1 and e1 == nwav and e0 % 256 == 0 and abs((e0 - b0) - (e1 - b1)) < 512
    assert res[1][2] is None
    rs = np.random.RandomState(7)
    key_all = rs.normal(size=nwav)
    key_all[rs.uniform(size=nwav) < 0.05] = -0.5
    col_all = np.abs(rs.normal(size=nwav))
    order = np.argsort(key_all, kind="stable")
    rank_of = np.empty(nwav, dtype=np.int32)
    rank_of[order] = np.arange(nwav, dtype=np.int32)
    assert res[0][2] == (key_all.tobytes(), col_all.tobytes(), rank_of.tobytes())


def test_wavenumber_range_properties():
    from ecckd_amd import shard
    for nwav in (1, 255, 256, 257, 7_200_000, 3_300_001):
        for world in (1, 2, 3, 8):
            pieces = [shard.wavenumber_range(nwav, r, world) for r in range(world)]
            assert pieces[0][0] == 0 and pieces[-1][1] == nwav
            assert all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
            assert all(b % 256 == 0 for b, _ in pieces if b < nwav)
            sizes = [e - b for b, e in pieces]
            assert max(sizes) - min(sizes) <= 256 + 255
