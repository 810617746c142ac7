"""Profile-sharded optimize_lut over several ranks (optimize_lut row): every rank holds its share of the
training profiles and the ranks sum [gradient, cost] with ONE all-reduce per evaluation (ecckd_opt_set_allreduce).

The GPU box has one GPU, so the two ranks of this test share it and reduce through gloo (staged through the host);
on a multi-GPU node the same callback issues an RCCL all-reduce on the device buffer.  Checked: the reduced cost and
gradient equal the single-process values on the full training set (summation order aside), both ranks hold
bit-identical values, and a few L-BFGS iterations end in the same state on both ranks and next to the single-process one."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(flux_weight=0.2, flux_profile_weight=0.05, broadband_weight=0.4, prior_error=4.0, pressure_corr=0.95,
           temperature_corr=0.95, conc_corr=0.9, cap_relative_linear=0.8)
NITER = 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(ctx):
    """Model, and two training scenes whose band fluxes come from a perturbed "truth" model (run on the device)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ckd_synth
    from ecckd_amd import api
    model = ckd_synth.make_model(seed=21)
    truth = dict(model, gases=[dict(g, molar_abs=g["molar_abs"] * np.exp(0.2 * np.random.RandomState(i).normal(size=g["molar_abs"].shape)))
                               for i, g in enumerate(model["gases"])])
    scenes = ckd_synth.make_scenes(model, nscene=2, ncol=5, nlay=16)
    ib, nband = model["iband_per_g"], model["nband"]
    out = []
    for sc in scenes:
        sc = dict(sc, gas_present=None)
        fl = api.run_ckd(ctx, truth, sc, per_gas=False)
        band = lambda a: np.stack([a[..., ib == b].sum(-1) for b in range(nband)], axis=-1)
        sc["flux_dn"], sc["flux_up"] = band(fl["spectral_flux_dn_lw"]), band(fl["spectral_flux_up_lw"])
        out.append(sc)
    return model, out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from ecckd_amd import api, shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with api.Context(0) as ctx:
            model, scenes = _problem(ctx)
            mine = [shard.shard_scene_columns(s, rank, world) for s in scenes]
            opt = api.Optimizer(ctx, model, mine, **CFG)
            opt.set_allreduce()
            x0 = opt.initial_state()
            J0, g0 = opt.cost_grad(x0)
            res = opt.minimize(max_iterations=NITER, convergence_criterion=0.0)
            od, _ = opt.forward(res["x"])                     # the diagnostic pass must not enter the collective
            opt.close()
        q.put((rank, J0, g0, res, [int(np.asarray(s["pressure_hl"]).shape[0]) for s in mine], float(od.sum())))
    except Exception as exc:                                  # report instead of hanging the partner in a collective
        q.put((rank, repr(exc)))
        raise
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_match_single_process(ctx):
    from ecckd_amd import api
    model, scenes = _problem(ctx)
    full = api.Optimizer(ctx, model, scenes, **CFG)
    x0 = full.initial_state()
    J_ref, g_ref = full.cost_grad(x0)
    ref = full.minimize(max_iterations=NITER, convergence_criterion=0.0)
    full.close()

    world = 2
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    assert all(len(r) > 2 for r in res), res
    assert all(p.exitcode == 0 for p in procs)
    res.sort(key=lambda r: r[0])
    assert res[0][4] == [3, 3] and res[1][4] == [2, 2]                       # 5 profiles per scene -> 3 + 2
    for _, J0, g0, _, _, _ in res:
        assert J0 == pytest.approx(J_ref, rel=1e-12)
        assert np.allclose(g0, g_ref, rtol=1e-9, atol=1e-12 * np.abs(g_ref).max())
    # both ranks saw the same reduced numbers, so they took the same decisions and hold the same state
    assert res[0][1] == res[1][1] and np.array_equal(res[0][2], res[1][2])
    a, b = res[0][3], res[1][3]
    assert a["status"] == b["status"] and a["iterations"] == b["iterations"] and a["cost"] == b["cost"]
    assert np.array_equal(a["x"], b["x"])
    # and they followed the single-process trajectory up to summation-order rounding
    assert a["iterations"] == ref["iterations"] and a["cost"] < 0.9 * J_ref
    assert a["cost"] == pytest.approx(ref["cost"], rel=1e-6)
    moved = np.abs(ref["x"] - x0) > 0
    assert np.allclose(a["x"][moved], ref["x"][moved], rtol=1e-5, atol=1e-7)
