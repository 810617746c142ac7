"""GPU parity for the shortwave branches of find_g_points (rows a5, a10-a13 SW):
gas preparation, batched interval errors for every averaging method including
total-transmission, against the CPU oracle.  Tolerances as in test_find_g_gpu.py."""
import math
import sys

import numpy as np
import pytest
import torch

from conftest import make_lw_case

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _every_request_on_the_device(monkeypatch):
    """These tests compare evaluations of the same interval with each other (other batches, other sweep kernels): the memo
    of interval errors would answer the second one without running it."""
    monkeypatch.setenv("ECCKD_NO_ERROR_MEMO", "1")

ERR_RTOL = 1e-9
MU0 = 0.5


def _dev(ctx, a):
    return torch.as_tensor(np.ascontiguousarray(a), device=ctx.device)


def _sw_problem(oracle, nwav, nlay=30, seed=41, method="total-transmission", min_scaling=0.5, max_scaling=2.5,
                with_albedo=True, bg_kind="double"):
    from ecckd_amd import synthetic as syn
    lo, hi = 250.0, 50000.0
    p, wn, dwn, od32 = make_lw_case(nwav, nlay=nlay, seed=seed, lo=lo, hi=hi, column_scale=5.0)
    od = od32.astype(np.float64)
    _, _, _, bg32 = make_lw_case(nwav, nlay=nlay, seed=seed + 100, lo=lo, hi=hi, column_scale=0.5)
    bg = bg32.astype(np.float64) * 0.5 + 1e-5
    if bg_kind == "float":
        bg = bg.astype(np.float32).astype(np.float64)       # a FLOAT background file: the oracle sees the same values
    elif bg_kind == "none":
        bg = np.zeros_like(bg)
    ssi = syn.solar_spectral_irradiance(wn, dwn)
    key, col, st = oracle.reorder_key(p, None, wn, dwn, od, ssi, 0.25)
    assert st == 0
    _, oi, rank = oracle.stable_argsort_bands(wn, key, [lo], [hi])
    ireorder = np.empty(nwav, dtype=np.int64)
    ireorder[rank] = np.arange(nwav)
    od_s, bg_s, ssi_s = od[:, ireorder], bg[:, ireorder], ssi[ireorder]
    albedo = np.where(wn < 10000.0, 0.15, 0.0) if with_albedo else None      # find_g_points.cpp:921-923
    alb_s = albedo[ireorder] if with_albedo else None
    fdn = oracle.radiative_transfer_direct_sw(MU0, ssi_s, bg_s + od_s)
    hr = oracle.heating_rate(p, fdn, None)
    o = dict(p=p, wn=wn, od=od, bg=bg, ssi=ssi, rank=rank, albedo=albedo, od_s=od_s, bg_s=bg_s, ssi_s=ssi_s,
             hr=hr, fds=fdn[-1].copy(), fut=np.zeros(nwav), lw=oracle.layer_weight(p, 0.0),
             metric=oracle.metric(method, od_s), extras=None)
    if method == "total-transmission":
        ex = dict(min_scaling=min_scaling, max_scaling=max_scaling)
        for tag, sc in (("low", min_scaling), ("high", max_scaling)):
            if with_albedo:
                d, u = oracle.radiative_transfer_norayleigh_sw(MU0, ssi_s, bg_s + sc * od_s, alb_s)
                ex[f"flux_up_toa_{tag}"] = u[0].copy()
            else:
                d = oracle.radiative_transfer_direct_sw(MU0, ssi_s, bg_s + sc * od_s)
                ex[f"flux_up_toa_{tag}"] = np.zeros(nwav)
            ex[f"hr_{tag}"] = oracle.heating_rate(p, d, None)
            ex[f"flux_dn_surf_{tag}"] = d[-1].copy()
        o["extras"] = ex
    return o


def _make_gas(ctx, o, method, flux_weight=0.02):
    from ecckd_amd import api
    ex = o["extras"] or {}
    return api.GasSW(ctx, o["p"], _dev(ctx, o["ssi"]), _dev(ctx, o["rank"].astype(np.int32)), _dev(ctx, o["od"]),
                     _dev(ctx, o["bg"]), method, flux_weight, 0.0, MU0,
                     _dev(ctx, o["albedo"]) if o["albedo"] is not None else None,
                     ex.get("min_scaling", 1.0), ex.get("max_scaling", 1.0))


def _oracle_eq(oracle, o, method, flux_weight, albedo):
    return oracle.CkdEquipartitionSW(method, flux_weight, o["lw"], MU0, o["p"], o["ssi_s"], albedo, o["fds"],
                                     o["fut"], o["bg_s"], o["metric"], o["hr"], o["extras"])


def test_gas_prep_sw_matches_oracle(ctx, oracle):
    o = _sw_problem(oracle, 6000, nlay=30)
    gas = _make_gas(ctx, o, "total-transmission")
    assert np.array_equal(gas.view("ssi")[0], o["ssi_s"])
    assert np.array_equal(gas.view("bg_optical_depth"), o["bg_s"])
    # heating rate = conv * (difference of direct fluxes): absolute error ~ a few ulp of the flux
    conv = (9.80665 / 1004.0) / np.diff(o["p"])
    tol = lambda ref: 1e-12 * np.abs(ref).max(axis=0, keepdims=True) + 1e-14 * conv[:, None] * MU0 * o["ssi_s"][None, :]
    assert np.all(np.abs(gas.view("hr") - o["hr"]) <= tol(o["hr"]))
    assert np.allclose(gas.view("flux_dn_surf")[0], o["fds"], rtol=1e-12, atol=1e-300)
    ex = o["extras"]
    assert np.all(np.abs(gas.view("hr_low") - ex["hr_low"]) <= tol(ex["hr_low"]))
    assert np.all(np.abs(gas.view("hr_high") - ex["hr_high"]) <= tol(ex["hr_high"]))
    fx = gas.view("flux_extras")
    for row, name in enumerate(["flux_dn_surf_low", "flux_up_toa_low", "flux_dn_surf_high", "flux_up_toa_high"]):
        assert np.allclose(fx[row], ex[name], rtol=1e-12, atol=1e-300), name
    assert (ex["flux_up_toa_low"] > 0).any()
    gas.close()


@pytest.mark.parametrize("bg_kind", ["none", "float", "double"])
@pytest.mark.parametrize("method,with_albedo", [("total-transmission", True), ("total-transmission", False),
                                                ("transmission", True), ("logarithmic", True)])
def test_gas_prep_sw_fast_path_54_layers(ctx, oracle, bg_kind, method, with_albedo, n=9001):
    """The 54-layer FLOAT-spectrum preparations: all FLOAT (or no background) takes k_scatter_column_halves +
    k_gas_prep_sw_staged (columns staged through LDS in runs of 18 layers), a DOUBLE (merged) background the column-reading
    kernel.  A number of points that fills neither the last wave nor the last block; the prepared arrays against the oracle,
    and interval errors (which go through the per-wave row sums formed inside the preparation)."""
    from ecckd_amd import api
    o = _sw_problem(oracle, n, nlay=54, seed=61, method=method, with_albedo=with_albedo, bg_kind=bg_kind)
    od32 = o["od"].astype(np.float32)
    assert np.array_equal(od32.astype(np.float64), o["od"])
    bg = None if bg_kind == "none" else o["bg"].astype(np.float32) if bg_kind == "float" else o["bg"]
    ex = o["extras"] or {}
    gas = api.GasSW(ctx, o["p"], _dev(ctx, o["ssi"]), _dev(ctx, o["rank"].astype(np.int32)), _dev(ctx, od32),
                    _dev(ctx, bg) if bg is not None else None, method, 0.02, 0.0, MU0,
                    _dev(ctx, o["albedo"]) if o["albedo"] is not None else None,
                    ex.get("min_scaling", 1.0), ex.get("max_scaling", 1.0))
    assert np.array_equal(gas.view("ssi")[0], o["ssi_s"])
    assert np.array_equal(gas.view("bg_optical_depth"), o["bg_s"])
    conv = (9.80665 / 1004.0) / np.diff(o["p"])
    tol = lambda ref: 1e-12 * np.abs(ref).max(axis=0, keepdims=True) + 1e-14 * conv[:, None] * MU0 * o["ssi_s"][None, :]
    assert np.all(np.abs(gas.view("hr") - o["hr"]) <= tol(o["hr"]))
    assert np.allclose(gas.view("flux_dn_surf")[0], o["fds"], rtol=1e-12, atol=1e-300)
    if method == "total-transmission":
        assert np.all(np.abs(gas.view("hr_low") - ex["hr_low"]) <= tol(ex["hr_low"]))
        assert np.all(np.abs(gas.view("hr_high") - ex["hr_high"]) <= tol(ex["hr_high"]))
        fx = gas.view("flux_extras")
        for row, name in enumerate(["flux_dn_surf_low", "flux_up_toa_low", "flux_dn_surf_high", "flux_up_toa_high"]):
            assert np.allclose(fx[row], ex[name], rtol=1e-12, atol=1e-300), name
        assert (ex["flux_up_toa_low"] > 0).any() == with_albedo
    if n < 1000:
        gas.close()
        return
    gas.set_band_albedo(0.15)
    eq = _oracle_eq(oracle, o, method, 0.02, 0.15)
    b1 = np.array([0.0, 0.25, 0.6, 0.0, 0.993])
    b2 = np.array([0.25, 0.6, 1.0, 1.0, 1.0])
    err = gas.calc_error_batch(0, n, b1, b2)
    ref = np.array([eq.calc_error(x, y) for x, y in zip(b1, b2)])
    assert np.array_equal(np.isnan(err), np.isnan(ref))
    ok = np.isfinite(ref)
    assert ok.sum() >= 3 and np.allclose(err[ok], ref[ok], rtol=ERR_RTOL, atol=1e-10)
    gas.close()


@pytest.mark.parametrize("bg_kind", ["none", "float", "double"])
def test_gas_prep_sw_staged_and_gathering_kernels_agree(ctx, oracle, monkeypatch, bg_kind):
    """The same arithmetic per point in both preparation kernels: every prepared array bit for bit; the interval errors (row sums
    formed per wave inside the staged kernel, per 256-point tile by k_tile_sums after the other) to rounding."""
    from ecckd_amd import api
    n = 20011
    o = _sw_problem(oracle, n, nlay=54, seed=67, bg_kind=bg_kind)
    od32 = o["od"].astype(np.float32)
    bg = None if bg_kind == "none" else o["bg"].astype(np.float32) if bg_kind == "float" else o["bg"]
    ex = o["extras"]

    def make():
        g = api.GasSW(ctx, o["p"], _dev(ctx, o["ssi"]), _dev(ctx, o["rank"].astype(np.int32)), _dev(ctx, od32),
                      _dev(ctx, bg) if bg is not None else None, "total-transmission", 0.02, 0.0, MU0, _dev(ctx, o["albedo"]),
                      ex["min_scaling"], ex["max_scaling"])
        g.set_band_albedo(0.15)
        return g
    staged = make()
    monkeypatch.setenv("ECCKD_SW_PREP_GATHER", "1")
    gathered = make()
    for name in ("ssi", "bg_optical_depth", "weighted_metric", "hr", "flux_dn_surf", "flux_up_toa", "hr_low", "hr_high", "flux_extras"):
        assert np.array_equal(staged.view(name), gathered.view(name)), name
    b1 = np.array([0.0, 0.2, 0.55, 0.0])
    b2 = np.array([0.2, 0.55, 1.0, 1.0])
    assert np.allclose(staged.calc_error_batch(0, n, b1, b2), gathered.calc_error_batch(0, n, b1, b2), rtol=1e-9, atol=1e-12)
    staged.close()
    gathered.close()


@pytest.mark.parametrize("bg_kind", ["none", "float"])
def test_gas_prep_sw_staged_less_than_a_wave(ctx, oracle, bg_kind):
    """41 points: one partly filled wave, three idle ones in the only block."""
    test_gas_prep_sw_fast_path_54_layers(ctx, oracle, bg_kind, "total-transmission", True, n=41)


@pytest.mark.parametrize("method", ["linear", "transmission", "transmission-2", "square-root", "logarithmic"])
@pytest.mark.parametrize("albedo", [0.0, 0.15])
def test_interval_errors_sw_match_oracle(ctx, oracle, method, albedo):
    n = 16000
    o = _sw_problem(oracle, n, nlay=30, seed=43, method=method)
    gas = _make_gas(ctx, o, method, flux_weight=0.02)
    gas.set_band_albedo(albedo)
    eq = _oracle_eq(oracle, o, method, 0.02, albedo)
    cuts = np.sort(np.random.RandomState(6).uniform(0, 1, 9))
    b1 = np.concatenate([[0.0], cuts, [0.0, 0.4, 0.9999]])
    b2 = np.concatenate([cuts, [1.0], [1.0, 0.4002, 1.0]])
    err = gas.calc_error_batch(0, n, b1, b2)
    ref = np.array([eq.calc_error(x, y) for x, y in zip(b1, b2)])
    # The SW "transmission" fits clamp BEFORE normalising (find_g_points.cpp:123-124), so where the
    # interval's solar irradiance sums to < the clamped numerator the reference takes log of a
    # negative number: NaN in the reference, NaN here, at the same intervals.
    assert np.array_equal(np.isnan(err), np.isnan(ref))
    if method in ("linear", "square-root", "logarithmic"):
        assert np.all(np.isfinite(err))
    err, ref = err[np.isfinite(ref)], ref[np.isfinite(ref)]
    # absolute floor: errors are K/d differences of sums whose rounding noise is ~1e-12 K/d
    assert np.allclose(err, ref, rtol=ERR_RTOL, atol=1e-10)
    gas.close()


@pytest.mark.parametrize("with_albedo,band_albedo", [(True, 0.15), (True, 0.0), (False, 0.0)])
def test_interval_errors_total_transmission(ctx, oracle, with_albedo, band_albedo):
    """find_g_points.cpp:341-386: error = 0.5*(cost at fit*min_scaling vs low truth + cost at fit*max_scaling
    vs high truth); the fit matches the broadband direct transmission layer by layer (:171-204)."""
    n = 16000
    o = _sw_problem(oracle, n, nlay=30, seed=47, method="total-transmission", with_albedo=with_albedo)
    gas = _make_gas(ctx, o, "total-transmission", flux_weight=0.02)
    gas.set_band_albedo(band_albedo)
    eq = _oracle_eq(oracle, o, "total-transmission", 0.02, band_albedo)
    b1 = np.array([0.0, 0.2, 0.55, 0.9, 0.0, 0.31])
    b2 = np.array([0.2, 0.55, 0.9, 1.0, 1.0, 0.3102])
    # a band offset too: sorted indices [2000, 13999]
    err = gas.calc_error_batch(0, n, b1, b2)
    ref = np.array([eq.calc_error(x, y) for x, y in zip(b1, b2)])
    # absolute floor: errors are K/d differences of sums whose rounding noise is ~1e-12 K/d
    assert np.allclose(err, ref, rtol=ERR_RTOL, atol=1e-10)
    gas.close()


@pytest.mark.parametrize("nlay,method,mu0", [(54, "total-transmission", 0.5), (54, "total-transmission", 0.6),
                                             (30, "total-transmission", 0.6), (20, "total-transmission", 0.5),
                                             (54, "transmission", 0.5), (54, "transmission", 0.6), (30, "linear", 0.6)])
def test_interval_errors_sweep_variants(ctx, oracle, monkeypatch, nlay, method, mu0):
    """Every instantiation of the shortwave sweep: one or both fits per launch (total-transmission evaluates the fit scaled by
    min_scaling and by max_scaling on one fetch of the column), the transmittance kept from the way down at the reference's
    cos_sza = 0.5 or evaluated twice at any other, compile-time or run-time number of layers."""
    monkeypatch.setattr(sys.modules[__name__], "MU0", mu0)
    n = 9000
    o = _sw_problem(oracle, n, nlay=nlay, seed=59, method=method)
    gas = _make_gas(ctx, o, method, flux_weight=0.02)
    gas.set_band_albedo(0.15)
    eq = _oracle_eq(oracle, o, method, 0.02, 0.15)
    b1 = np.array([0.0, 0.25, 0.6, 0.0])
    b2 = np.array([0.25, 0.6, 1.0, 1.0])
    err = gas.calc_error_batch(0, n, b1, b2)
    ref = np.array([eq.calc_error(x, y) for x, y in zip(b1, b2)])
    ok = np.isfinite(ref)
    assert ok.sum() >= 3 and np.allclose(err[ok], ref[ok], rtol=ERR_RTOL, atol=1e-10)
    gas.close()


def test_find_g_band_sw_matches_reference_search(ctx, oracle):
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    n = 24000
    o = _sw_problem(oracle, n, nlay=30, seed=53, method="total-transmission")
    gas = _make_gas(ctx, o, "total-transmission", flux_weight=0.02)
    gas.set_band_albedo(0.15)
    eq = _oracle_eq(oracle, o, "total-transmission", 0.02, 0.15)
    tol = 0.05 * eq.calc_error(0.0, 1.0)
    st, b, e, cc = gas.find_g_band(0, n - 1, tol, tolerance_tolerance=0.02, max_iterations=40)
    ref = oracle.RefEquipartition(eq.calc_error, resolution=1.0 / n, partition_tolerance=0.02,
                                  partition_max_iterations=40)
    rst, rb, re = ref.equipartition_e(tol)
    assert len(b) == len(rb) and len(b) >= 4 and st == rst
    assert [math.ceil(x * (n - 1)) for x in b[:-1]] == [math.ceil(x * (n - 1)) for x in rb[:-1]]
    assert np.allclose(e, re, rtol=1e-8)
    gas.close()


@pytest.mark.parametrize("method", ["transmission", "logarithmic", "total-transmission"])
def test_fit_optical_depth_sw_matches_oracle(ctx, oracle, method):
    """a11 seam: ecckd_fit_optical_depth vs fit_optical_depth_sw / _sw_total_trans, including the
    saturated layers where the reference returns +inf (log(0))."""
    import ctypes as C
    n = 16000
    o = _sw_problem(oracle, n, nlay=30, seed=43, method=method)
    gas = _make_gas(ctx, o, method)
    b1 = np.array([0.0, 0.3, 0.9999])
    b2 = np.array([0.3, 0.99, 1.0])
    fit = gas.fit_optical_depth(0, n, b1, b2)
    L = oracle.lib()
    for k in range(3):
        i1, i2 = math.ceil(b1[k] * (n - 1)), math.floor(b2[k] * (n - 1))
        ref = np.empty(30)
        if method == "total-transmission":
            L.orc_fit_optical_depth_sw_total_trans(C.c_int(30), C.c_size_t(n), C.c_size_t(i1), C.c_size_t(i2),
                                                   oracle._p(o["ssi_s"]), oracle._p(np.ascontiguousarray(o["bg_s"])),
                                                   oracle._p(np.ascontiguousarray(o["od_s"])), oracle._p(ref))
        else:
            L.orc_fit_optical_depth_sw(C.c_int(oracle.AVG[method]), C.c_int(30), C.c_size_t(n), C.c_size_t(i1),
                                       C.c_size_t(i2), oracle._p(o["ssi_s"]),
                                       oracle._p(np.ascontiguousarray(o["metric"])), oracle._p(ref))
        assert np.array_equal(np.isinf(fit[k]), np.isinf(ref))
        m = np.isfinite(ref)
        # log(1-v) amplifies near saturation; the total-transmission fit is a difference of two
        # logs of order 1, so its absolute accuracy is that of the interval sums (~1e-14)
        assert np.allclose(fit[k][m], ref[m], rtol=1e-7, atol=1e-13)
    gas.close()


def test_shortwave_bands_side_by_side_match_one_band_at_a_time(ctx, oracle):
    """ecckd_find_g_bands_ex on a shortwave gas: four bands with their own surface albedos (0.15 below 10 000 cm-1, 0 above, as
    find_g_points.cpp:757-761) searched side by side, the albedo travelling with every interval, against the band loop
    (ecckd_gas_set_band_albedo + ecckd_find_g_band_ex per band): same g points, errors equal to rounding; the gas's own band
    albedo is left alone."""
    n = 32000
    o = _sw_problem(oracle, n, nlay=30, seed=61, method="total-transmission")
    gas = _make_gas(ctx, o, "total-transmission", flux_weight=0.02)
    begin = np.array([0, 6000, 15000, 26000])
    end = np.array([5999, 14999, 25999, n - 1])
    albedo = [0.15, 0.0, 0.15, 0.0]
    tol = [0.05, 0.03, 0.08, 0.05]
    one_by_one = []
    for k in range(4):
        gas.set_band_albedo(albedo[k])
        one_by_one.append(gas.find_g_band_ex(int(begin[k]), int(end[k]), tol[k], 0.02, 30, min_g_points=2))
    gas.set_band_albedo(0.07)
    options = [dict(min_g_points=2, band_albedo=albedo[k]) for k in range(4)]
    together = gas.find_g_bands_ex(begin, end, tol, 0.02, 30, options=options)
    assert sum(len(r["error"]) for r in one_by_one) >= 10
    for k, (a, b) in enumerate(zip(one_by_one, together)):
        assert a["status"] == b["status"] and np.array_equal(a["rank1"], b["rank1"]) and np.array_equal(a["rank2"], b["rank2"]), k
        assert np.allclose(a["error"], b["error"], rtol=1e-9, atol=1e-12), k
    # the albedo matters (else this test would not notice a mix-up) and the merged batch takes it per interval
    ib, npt = np.repeat(begin[[0, 2]], 2), np.repeat((end - begin + 1)[[0, 2]], 2)
    lo, hi = np.tile([0.0, 0.5], 2), np.tile([0.5, 1.0], 2)
    merged = gas.calc_error_multi(ib, npt, lo, hi, band_albedo=[0.15, 0.15, 0.0, 0.0])
    single = []
    for k, a in ((0, 0.15), (2, 0.0)):
        gas.set_band_albedo(a)
        single.append(gas.calc_error_batch(int(begin[k]), int(end[k] - begin[k] + 1), [0.0, 0.5], [0.5, 1.0]))
    assert np.allclose(merged, np.concatenate(single), rtol=1e-9, atol=1e-12)
    gas.set_band_albedo(0.15)
    with_albedo = gas.calc_error_batch(int(begin[2]), int(end[2] - begin[2] + 1), [0.0, 0.5], [0.5, 1.0])
    assert not np.allclose(with_albedo, single[1], rtol=1e-6)
    gas.close()


@pytest.mark.parametrize("nlay,method", [(54, "total-transmission"), (30, "transmission"), (12, "total-transmission")])
def test_interval_error_sw_does_not_depend_on_the_batch(ctx, oracle, nlay, method):
    """Shortwave twin of test_find_g_gpu.py::test_interval_error_does_not_depend_on_the_batch: an interval's error has the
    same bits alone, with its neighbours, in another order and next to another band's intervals (with another albedo)."""
    n = 120_000
    o = _sw_problem(oracle, n, nlay=nlay, seed=47, method=method)
    gas = _make_gas(ctx, o, method)
    gas.set_band_albedo(0.15)
    rs = np.random.RandomState(13)
    cuts = np.concatenate([[0.0], np.sort(rs.uniform(0, 1, 11)), [1.0]])
    b1, b2 = cuts[:-1], cuts[1:]
    together = gas.calc_error_batch(0, n, b1, b2)
    alone = np.array([gas.calc_error_batch(0, n, [x], [y])[0] for x, y in zip(b1, b2)])
    perm = rs.permutation(len(b1))
    assert np.array_equal(together, alone)
    assert np.array_equal(together[perm], gas.calc_error_batch(0, n, b1[perm], b2[perm]))
    ib, nb = 10_007, 70_000
    e_band = gas.calc_error_batch(ib, nb, b1, b2)
    ibegin = np.concatenate([[0], np.full(len(b1), ib), [90_000]])
    npts = np.concatenate([[9_000], np.full(len(b1), nb), [30_000]])
    alb = np.concatenate([[0.0], np.full(len(b1), 0.15), [0.0]])
    e_multi = gas.calc_error_multi(ibegin, npts, np.concatenate([[0.0], b1, [0.0]]), np.concatenate([[1.0], b2, [1.0]]), band_albedo=alb)
    assert np.array_equal(e_band, e_multi[1:-1])
    gas.close()
