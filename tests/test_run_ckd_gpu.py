"""GPU parity for run_ckd (run_ckd.cpp:27-373): optical depths per gas and in total, Planck
function, spectral and broadband fluxes of a CKD model on a set of profiles, and the headline accuracy
metric built from them - the heating-rate RMS error of plot/calc_hr_error.m:1-23 - between the device
path and the CPU oracle (target of BASELINE.json: within 1e-6 K/day)."""
import numpy as np
import pytest

import ckd_synth

pytestmark = pytest.mark.gpu


def calc_hr_error(pressure_hl, hr, hr_ref, pressure_range=(0.0, np.inf)):
    """plot/calc_hr_error.m:1-23 (pressure in hPa, arrays (level, profile)): cube-root-pressure weighted RMS."""
    weight = pressure_hl[1:] ** (1.0 / 3.0) - pressure_hl[:-1] ** (1.0 / 3.0)
    pfl = 0.5 * (pressure_hl[:-1] + pressure_hl[1:])
    weight = np.where((pfl < pressure_range[0]) | (pfl >= pressure_range[1]), 0.0, weight)
    weight = weight / weight.sum(axis=0, keepdims=True)
    return np.sqrt(np.sum(weight * (hr - hr_ref) ** 2) / pressure_hl.shape[1])


def _hr_K_per_day(oracle, p, dn, up=None):
    return np.stack([oracle.heating_rate(p[c], dn[c][:, None], None if up is None else up[c][:, None])[:, 0]
                     for c in range(p.shape[0])]) * 86400.0


def test_run_ckd_lw(ctx, oracle):
    from ecckd_amd import api
    model = ckd_synth.make_model(seed=5)
    scene = ckd_synth.make_scenes(model, nscene=1, ncol=6, nlay=24)[0]
    orc = ckd_synth.Oracle(oracle, model, [scene], {})
    out = api.run_ckd(ctx, model, scene)
    od_ref = np.maximum(orc.optical_depth(orc.x0, scene), 0.0)
    assert np.allclose(out["optical_depth"], od_ref, rtol=1e-12, atol=1e-300)
    fl_ref = orc.fluxes(orc.x0, scene)
    assert np.allclose(out["spectral_flux_dn_lw"], fl_ref[:, 0], rtol=1e-10, atol=1e-300)
    assert np.allclose(out["spectral_flux_up_lw"], fl_ref[:, 1], rtol=1e-10)
    pl_ref = np.stack([orc.planck(T) for T in scene["temperature_hl"]])
    assert np.allclose(out["planck_hl"], pl_ref, rtol=1e-14) and np.array_equal(out["planck_surf"], out["planck_hl"][:, -1])
    # per-gas optical depths: each gas alone, whatever its concentration dependence
    for i, g in enumerate(model["gases"]):
        alone = dict(model, gases=[gg if j == i else dict(gg, molar_abs=np.zeros_like(gg["molar_abs"]))
                                   for j, gg in enumerate(model["gases"])])
        o1 = ckd_synth.Oracle(oracle, alone, [scene], {})
        ref = np.maximum(o1.optical_depth(o1.x0, scene), 0.0)
        assert np.allclose(out[g["name"] + "_optical_depth"], ref, rtol=1e-12, atol=1e-300), g["name"]
    # heating-rate RMS error between the device path and the CPU path (K/day)
    p = scene["pressure_hl"]
    hr = _hr_K_per_day(oracle, p, out["flux_dn_lw"], out["flux_up_lw"])
    hr_ref = _hr_K_per_day(oracle, p, fl_ref[:, 0].sum(-1), fl_ref[:, 1].sum(-1))
    rmse = calc_hr_error(p.T / 100.0, hr.T, hr_ref.T)
    assert rmse < 1e-6
    assert np.abs(hr_ref).max() > 0.1                               # a non-trivial profile


def test_run_ckd_lw_gas_list_and_scaling(ctx, oracle):
    from ecckd_amd import api
    model = ckd_synth.make_model(seed=6)
    scene = ckd_synth.make_scenes(model, nscene=1, ncol=3, nlay=18)[0]
    out = api.run_ckd(ctx, model, scene, gases=["h2o", "co2"], scalings={2: 4.0}, per_gas=False)
    sc2 = dict(scene, vmr_fl=scene["vmr_fl"].copy(), gas_present=np.array([0, 1, 1, 0, 0], dtype=np.int32))
    sc2["vmr_fl"][:, 2] *= 4.0
    m2 = dict(model, gases=[g if i in (1, 2) else dict(g, molar_abs=np.zeros_like(g["molar_abs"]))
                            for i, g in enumerate(model["gases"])])
    orc = ckd_synth.Oracle(oracle, m2, [sc2], {})
    assert np.allclose(out["optical_depth"], np.maximum(orc.optical_depth(orc.x0, dict(sc2, gas_present=None)), 0.0),
                       rtol=1e-12, atol=1e-300)


def test_run_ckd_sw(ctx, oracle):
    from ecckd_amd import api
    model = ckd_synth.make_model_sw(seed=5)
    scene = ckd_synth.make_scenes_sw(model, (0.0, 0.0, 0.0), mu0=(0.5,), nscene=1, ncol=5, nlay=24)[0]
    orc = ckd_synth.OracleSW(oracle, model, [scene], {})
    out = api.run_ckd(ctx, model, scene, per_gas=False)
    od_gas = np.maximum(ckd_synth.Oracle.optical_depth(orc, orc.x0, scene), 0.0)
    assert np.allclose(out["optical_depth"], od_gas, rtol=1e-12, atol=1e-300)
    ray = orc.optical_depth(orc.x0, scene) - ckd_synth.Oracle.optical_depth(orc, orc.x0, scene)
    assert np.allclose(out["rayleigh_optical_depth"], ray, rtol=1e-9)
    assert np.allclose(out["incoming_sw"], np.tile(orc.ssi(scene), (5, 1)), rtol=1e-15)
    dn_ref = np.stack([oracle.radiative_transfer_direct_sw(0.5, orc.ssi(scene), od_gas[c] + out["rayleigh_optical_depth"][c])
                       for c in range(5)])
    assert np.allclose(out["spectral_flux_dn_direct_sw"], dn_ref, rtol=1e-11, atol=1e-300)
    p = scene["pressure_hl"]
    hr = _hr_K_per_day(oracle, p, out["flux_dn_direct_sw"])
    hr_ref = _hr_K_per_day(oracle, p, dn_ref.sum(-1))
    assert calc_hr_error(p.T / 100.0, hr.T, hr_ref.T) < 1e-6
