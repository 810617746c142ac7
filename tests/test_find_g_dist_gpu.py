"""(gas, band) searches of find_g_points dealt to several processes (find_g_points.cpp:655, :1152, :1456).

The GPU box has one GPU, so the two ranks of this test share it and talk through gloo; on a multi-GPU node the same
driver runs one process per GPU over RCCL.  Checked: the g-points file written by rank 0 of the two-process run is
BYTE-identical to the single-process one (an interval's error does not depend on its batch, so a band ends at the same
g points whichever bands are searched next to it), every (gas, band) task is searched exactly once, rank 1 - which
prepares no band of the first gas - rebuilds the first gas's Planck matrix bit for bit (ecckd_planck_hl_sorted_dev),
and both ranks hold the same all-reduced final cost."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from ecckd_amd import synthetic as syn

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NLAY, NWAV = 30, 24000
BANDS = (np.array([0.0, 700.0, 1500.0]), np.array([700.0, 1500.0, 3260.0]))
GASES = {"h2o": (41, 30.0, 5e-3), "co2": (43, 8.0, 4e-4), "o3": (47, 3.0, 1e-6)}
KW = dict(tolerance_tolerance=0.02, max_iterations=40, flux_weight=0.02)
TOL = [0.08, 0.05, 0.1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gas_specs(d):
    names = list(GASES)
    return [dict(name=g, input=os.path.join(d, f"{g}.nc"), reordering_input=os.path.join(d, f"order_{g}.nc"),
                 background=[dict(path=os.path.join(d, f"{o}.nc")) for o in names if o != g], min_g_points=2)
            for g in names]


def _worker(rank, world, port, d, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ecckd_amd import api, pipeline
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with api.Context(0) as ctx:
            res = pipeline.find_g_points(ctx, _gas_specs(d), BANDS[0], BANDS[1], TOL, output_path=os.path.join(d, "gpoints_dist.nc"), **KW)
        q.put((rank, res["tasks"], res["cost_sum"], res["comp_cost_sum"], res.get("ng")))
    except Exception as exc:                                  # report instead of hanging the partner in a collective
        q.put((rank, repr(exc)))
        raise
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_write_the_same_g_points_file(ctx, tmp_path):
    from test_pipeline_gpu import _write_spectrum
    import test_pipeline_gpu
    from ecckd_amd import pipeline
    d = str(tmp_path)
    p = syn.pressure_grid(NLAY)
    t_hl = syn.temperature_profile(p)
    wn, _ = syn.wavenumber_grid(NWAV)
    old = test_pipeline_gpu.NLAY
    test_pipeline_gpu.NLAY = NLAY
    try:
        for g, (seed, scale, vmr) in GASES.items():
            od = syn.optical_depth(np, p, wn, syn.SEED_BASE + seed, nlines=40, column_scale=scale, dtype="float32")
            _write_spectrum(tmp_path / f"{g}.nc", g, p, t_hl, wn, od, vmr)
    finally:
        test_pipeline_gpu.NLAY = old
    for g in GASES:
        pipeline.reorder_spectrum(ctx, tmp_path / f"{g}.nc", tmp_path / f"order_{g}.nc", BANDS[0], BANDS[1])
    one = pipeline.find_g_points(ctx, _gas_specs(d), BANDS[0], BANDS[1], TOL, output_path=tmp_path / "gpoints_one.nc", **KW)
    assert one["ng"] >= 6 and one["n_unassigned"] == 0 and len(one["tasks"]) == 9
    assert one["cost_sum"] == pytest.approx(sum(sum(g["error"]) for g in one["gases"]), rel=1e-14)
    # one band at a time (the reference's order of evaluation) ends at the same g points as the side-by-side searches
    seq = pipeline.find_g_points(ctx, _gas_specs(d), BANDS[0], BANDS[1], TOL, output_path=tmp_path / "gpoints_seq.nc",
                                 sequential_bands=True, **KW)
    assert (tmp_path / "gpoints_seq.nc").read_bytes() == (tmp_path / "gpoints_one.nc").read_bytes()
    assert seq["comp_cost_sum"] == pytest.approx(one["comp_cost_sum"], rel=1e-12)

    world = 2
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_worker, args=(r, world, port, d, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=120)
    assert all(len(r) > 2 for r in res), res
    assert all(pr.exitcode == 0 for pr in procs)
    res.sort(key=lambda r: r[0])
    # 3 gases x 3 bands in contiguous shares: rank 0 = gas 0 and two bands of gas 1, rank 1 = the rest (none of gas 0)
    assert res[0][1] == [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1)] and res[1][1] == [(1, 2), (2, 0), (2, 1), (2, 2)]
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3]                 # the all-reduced scalars, identical on both ranks
    assert res[0][2] == pytest.approx(one["cost_sum"], rel=1e-13) and res[0][3] == pytest.approx(one["comp_cost_sum"], rel=1e-13)
    assert res[0][4] == one["ng"]
    assert (tmp_path / "gpoints_dist.nc").read_bytes() == (tmp_path / "gpoints_one.nc").read_bytes()


@pytest.mark.parametrize("nlay", [54, 30])
def test_planck_matrix_of_an_ordering_is_the_gas_preparations(ctx, nlay):
    """ecckd_planck_hl_sorted_dev == the planck_hl rows of a gas prepared with the same ordering, bit for bit (54 layers:
    the mirror preparation kernel with its scalar-register exp; 30: the run-time kernel), and a gas prepared WITH that
    matrix as planck_hl_reuse equals one that reuses the first gas's own rows."""
    from conftest import make_lw_case
    from ecckd_amd import api
    n = 30_000
    p, wn, dwn, od = make_lw_case(n, nlay=nlay, seed=5)
    _, _, _, od2 = make_lw_case(n, nlay=nlay, seed=6, column_scale=3.0)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=ctx.device)
    t_hl = syn.temperature_profile(p)
    key, _ = api.reorder_key_lw(ctx, p, api.idealised_temperature(p), dev(wn), dev(dwn), dev(od), 0.5)
    rank, _ = api.stable_argsort_bands(ctx, key, [0], [n - 1], want_ordered=False)
    key2, _ = api.reorder_key_lw(ctx, p, api.idealised_temperature(p), dev(wn), dev(dwn), dev(od2), 0.5)
    rank2, _ = api.stable_argsort_bands(ctx, key2, [0], [n - 1], want_ordered=False)
    first = api.GasLW(ctx, p, t_hl, dev(wn), dev(dwn), rank, dev(od), None, "transmission", 0.02)
    planck = api.planck_hl_sorted(ctx, t_hl, dev(wn), dev(dwn), rank)
    assert np.array_equal(planck.cpu().numpy(), first.view("planck_hl"))
    a = api.GasLW(ctx, p, t_hl, dev(wn), dev(dwn), rank2, dev(od2), dev(od), "transmission", 0.02,
                  planck_hl_reuse=first.view_ptr("planck_hl")[0])
    b = api.GasLW(ctx, p, t_hl, dev(wn), dev(dwn), rank2, dev(od2), dev(od), "transmission", 0.02, planck_hl_reuse=planck.data_ptr())
    for name in ("hr", "weighted_metric", "flux_dn_surf", "flux_up_toa"):
        assert np.array_equal(a.view(name), b.view(name)), name
    e_a = a.calc_error_batch(0, n, [0.0, 0.4], [0.4, 1.0])
    assert np.array_equal(e_a, b.calc_error_batch(0, n, [0.0, 0.4], [0.4, 1.0]))
    a.close(); b.close(); first.close()


def _reorder_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ecckd_amd import api, pipeline
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with api.Context(0) as ctx:
            n, nlay = 70_001, 54
            p = syn.pressure_grid(nlay)
            wn_h, dwn_h = syn.wavenumber_grid(n)
            od = torch.as_tensor(syn.optical_depth(np, p, wn_h, syn.SEED_BASE + 5, nlines=40), device=ctx.device)
            wn, dwn = torch.as_tensor(wn_h, device=ctx.device), torch.as_tensor(dwn_h, device=ctx.device)
            key, col, rnk = pipeline.reorder_single_band_sharded(ctx, p, wn, dwn, od, 0.5)
            out = None
            if rank == 0:
                k1, c1 = api.reorder_key_lw(ctx, p, api.idealised_temperature(p), wn, dwn, od, 0.5)
                r1, _ = api.stable_argsort_bands(ctx, k1, [0], [n - 1], want_ordered=False)
                out = (bool(torch.equal(key, k1)), bool(torch.equal(col, c1)), bool(torch.equal(rnk, r1)))
        q.put((rank, out))
    except Exception as exc:
        q.put((rank, repr(exc)))
        raise
    dist.barrier()
    dist.destroy_process_group()


def test_single_band_reorder_split_by_wavenumber_range(ctx):
    """one band (fsck) reordered by two processes - each sweeps its own wavenumber range (K1), rank 0 gathers and sorts:
    key, column optical depth and rank equal the single-process ones bit for bit."""
    world = 2
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_reorder_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0] == (True, True, True) and res[1] is None
