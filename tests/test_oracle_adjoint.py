"""The oracle's hand-written reverse mode (oracle/oracle_adjoint.c) against central differences of the oracle's own forward
cost: it is the CPU stand-in for the gradient the reference takes from Adept's tape (solve_adept.cpp:91, :201-203), and
what the device adjoint is compared with element by element in tests/test_optimize_gpu.py."""
import numpy as np
import pytest

import ckd_synth

CFG = dict(flux_weight=0.2, flux_profile_weight=0.05, broadband_weight=0.4, spectral_boundary_weight=0.3,
           negative_od_penalty=1.0e4, pressure_weight_power=0.5, prior_error=4.0, pressure_corr=0.95,
           temperature_corr=0.95, conc_corr=0.9, cap_relative_linear=0.0)


@pytest.mark.parametrize("ch4_low", [False, True])
def test_adjoint_matches_central_differences(oracle, ch4_low):
    model = ckd_synth.make_model(seed=3)
    truth = ckd_synth.make_model(seed=3)
    rs = np.random.RandomState(8)
    for g in truth["gases"]:
        g["molar_abs"] = g["molar_abs"] * np.exp(0.25 * rs.normal(size=g["molar_abs"].shape))
    scenes = ckd_synth.make_scenes(model, ch4_low=ch4_low)
    t = ckd_synth.Oracle(oracle, truth, scenes, CFG)
    for s in scenes:
        bf = t.band_fluxes(t.x0, s)
        s["flux_dn"], s["flux_up"] = np.ascontiguousarray(bf[:, 0]), np.ascontiguousarray(bf[:, 1])
        f = t.fluxes(t.x0, s)
        s["spectral_flux_dn_surf"] = np.ascontiguousarray(f[:, 0, -1, :])
        s["spectral_flux_up_toa"] = np.ascontiguousarray(f[:, 1, 0, :])
    orc = ckd_synth.Oracle(oracle, model, scenes, CFG)
    free = orc.x0 > -1.0e20
    x = orc.x0 + np.where(free, 0.2 * rs.normal(size=orc.x0.size), 0.0)
    if ch4_low:
        sizes = np.cumsum([0] + orc.sizes)
        x[sizes[3]:sizes[4]] += 9.0                         # cells with negative optical depth: penalty + clamp
    J, g = orc.cost_grad_rt(x)
    assert J == pytest.approx(orc.cost_rt(x), rel=1e-13)
    assert np.all(g[~free] == 0.0) and np.all(np.isfinite(g))
    for trial in range(4):
        d = np.where(free, rs.normal(size=x.size), 0.0)
        d /= np.linalg.norm(d)
        h = 1e-5
        fd = (orc.cost_rt(x + h * d) - orc.cost_rt(x - h * d)) / (2 * h)
        assert np.dot(g, d) == pytest.approx(fd, rel=2e-6, abs=1e-9 * abs(J))
    idx = rs.choice(np.nonzero(free & (np.abs(g) > 1e-3 * np.abs(g).max()))[0], 6, replace=False)
    for i in idx:
        h = 1e-5
        e = np.zeros_like(x); e[i] = h
        fd = (orc.cost_rt(x + e) - orc.cost_rt(x - e)) / (2 * h)
        assert g[i] == pytest.approx(fd, rel=2e-5, abs=1e-8 * np.abs(g).max())


@pytest.mark.parametrize("variant", ["base", "spectral_only", "direct_only", "mixed_albedo", "boundary", "negative_od", "profile_flux"])
def test_shortwave_adjoint_matches_central_differences(oracle, variant):
    """The shortwave reverse mode (orc_calc_cost_function_ckd_sw_ad: calc_cost_function_sw.cpp:116-277 differentiated by hand)
    against central differences of the oracle's own shortwave cost, in every branch of the cost function: spectral only
    (no 1/nband scaling, :243), all albedos <= 0 (direct beam only, :145-150), mixed albedos (broadband upwelling terms
    dropped, :252, :264), per-g boundary weights (:271-274), negative optical depths (solve_adept.cpp:107-116)."""
    kw, over = {}, {}
    if variant == "spectral_only":
        over = dict(broadband_weight=0.0)
    if variant == "direct_only":
        kw = dict(albedo=(0.0, 0.0, -1.0))
    if variant == "mixed_albedo":
        kw = dict(albedo=(0.2, 0.0, 0.1))
    if variant == "profile_flux":
        over = dict(flux_profile_weight=0.3)
    boundary = variant == "boundary"
    cfg = dict(CFG, spectral_boundary_weight=0.0, **over)
    model = ckd_synth.make_model_sw(seed=3)
    truth = ckd_synth.make_model_sw(seed=3)
    rs = np.random.RandomState(8)
    for g in truth["gases"]:
        g["molar_abs"] = g["molar_abs"] * np.exp(0.25 * rs.normal(size=g["molar_abs"].shape))
    bw = 0.02 * rs.uniform(size=model["ng"]) if boundary else None
    scenes = ckd_synth.make_scenes_sw(model, kw.get("albedo", (0.15, 0.3, 0.06)), boundary_weights=bw, ch4_low=(variant == "negative_od"))
    t = ckd_synth.OracleSW(oracle, truth, scenes, cfg)
    for s in scenes:
        bf = t.band_fluxes(t.x0, s)
        s["flux_dn"], s["flux_up"] = np.ascontiguousarray(bf[:, 0]), np.ascontiguousarray(bf[:, 1])
        if boundary:
            s["spectral_flux_dn_surf"] = np.ascontiguousarray(t.fluxes(t.x0, s)[:, 0, -1, :])
    orc = ckd_synth.OracleSW(oracle, model, scenes, cfg)
    free = orc.x0 > -1.0e20
    x = orc.x0 + np.where(free, 0.2 * rs.normal(size=orc.x0.size), 0.0)
    if variant == "negative_od":
        sizes = np.cumsum([0] + orc.sizes)
        x[sizes[3]:sizes[4]] += 9.0
    J, g = orc.cost_grad_rt(x)
    assert J == pytest.approx(orc.cost_rt(x), rel=1e-13)
    assert np.all(g[~free] == 0.0) and np.all(np.isfinite(g)) and np.abs(g).max() > 0
    for trial in range(4):
        d = np.where(free, rs.normal(size=x.size), 0.0)
        d /= np.linalg.norm(d)
        h = 1e-5
        fd = (orc.cost_rt(x + h * d) - orc.cost_rt(x - h * d)) / (2 * h)
        assert np.dot(g, d) == pytest.approx(fd, rel=2e-6, abs=1e-9 * abs(J))
    idx = rs.choice(np.nonzero(free & (np.abs(g) > 1e-3 * np.abs(g).max()))[0], 6, replace=False)
    for i in idx:
        h = 1e-5
        e = np.zeros_like(x); e[i] = h
        fd = (orc.cost_rt(x + e) - orc.cost_rt(x - e)) / (2 * h)
        assert g[i] == pytest.approx(fd, rel=2e-5, abs=1e-8 * np.abs(g).max())
