"""GPU parity for the line-by-line band-flux evaluator against the oracle's planck_function +
radiative_transfer_lw / _direct_sw / _norayleigh_sw, summed per band on the CPU."""
import numpy as np
import pytest
import torch

from conftest import make_lw_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,nlay,nwav", [("float32", 54, 20011), ("float64", 17, 777)])
def test_lbl_band_fluxes_lw(ctx, oracle, dtype, nlay, nwav):
    from ecckd_amd import api, synthetic as syn
    p, wn, dwn, od = make_lw_case(nwav, nlay=nlay, seed=5, dtype=dtype)
    t_hl = syn.temperature_profile(p)
    b1 = np.array([0.0, 700.0, 700.0, 2000.0])
    b2 = np.array([700.0, 700.0, 2000.0, 3260.0])                 # the second band is empty
    begin = np.array([np.nonzero((wn >= a) & (wn < b))[0][0] if ((wn >= a) & (wn < b)).any() else 1 for a, b in zip(b1, b2)])
    end = np.array([np.nonzero((wn >= a) & (wn < b))[0][-1] if ((wn >= a) & (wn < b)).any() else 0 for a, b in zip(b1, b2)])
    dev = lambda a: torch.as_tensor(a, device=ctx.device)
    dn, up = api.lbl_band_fluxes_lw(ctx, t_hl, dev(wn), dev(dwn), dev(od), begin, end)
    planck = oracle.planck_function(t_hl, wn, dwn)
    fdn, fup = oracle.radiative_transfer_lw(planck, od.astype(np.float64), np.ones(nwav), planck[-1])
    for b in range(4):
        sl = slice(begin[b], end[b] + 1) if end[b] >= begin[b] else slice(0, 0)
        assert np.allclose(dn[b], fdn[:, sl].sum(1), rtol=1e-10, atol=1e-300)
        assert np.allclose(up[b], fup[:, sl].sum(1), rtol=1e-10, atol=1e-300)
    assert np.all(dn[1] == 0) and np.all(up[1] == 0) and dn[0, -1] > 0
    # the spectral fluxes at the boundaries (do_write_spectral_boundary_fluxes): per wavenumber against the oracle, zero outside
    # the bands (here: none outside), and the same band fluxes beside them
    dn2, up2, sdn, tup = api.lbl_band_fluxes_lw(ctx, t_hl, dev(wn), dev(dwn), dev(od), begin, end, boundary=True)
    assert np.array_equal(dn2, dn) and np.array_equal(up2, up)
    covered = np.zeros(nwav, bool)
    for b in range(4):
        if end[b] >= begin[b]:
            covered[begin[b]:end[b] + 1] = True
    assert np.allclose(sdn.cpu().numpy()[covered], fdn[-1][covered], rtol=1e-11, atol=1e-300)
    assert np.allclose(tup.cpu().numpy()[covered], fup[0][covered], rtol=1e-11, atol=1e-300)
    assert np.all(sdn.cpu().numpy()[~covered] == 0) and np.all(tup.cpu().numpy()[~covered] == 0)
    # Stefan-Boltzmann sanity: the opaque limit of the whole spectrum approaches sigma T^4 of the lowest layers
    assert up[:, -1].sum() == pytest.approx(planck[-1].sum(), rel=1e-12)


@pytest.mark.parametrize("with_albedo", [False, True])
def test_lbl_band_fluxes_sw(ctx, oracle, with_albedo):
    from ecckd_amd import api, synthetic as syn
    nwav, nlay = 15001, 30
    p, wn, dwn, od = make_lw_case(nwav, nlay=nlay, seed=6, lo=250.0, hi=50000.0, column_scale=3.0)
    ssi = syn.solar_spectral_irradiance(wn, dwn)
    albedo = np.where(wn < 12000.0, 0.2, 0.05) if with_albedo else None
    begin, end = np.array([0, 5000]), np.array([4999, nwav - 1])
    dev = lambda a: torch.as_tensor(a, device=ctx.device)
    dn, up = api.lbl_band_fluxes_sw(ctx, 0.6, dev(ssi), dev(od), begin, end, albedo=dev(albedo) if with_albedo else None)
    if with_albedo:
        fdn, fup = oracle.radiative_transfer_norayleigh_sw(0.6, ssi, od.astype(np.float64), albedo)
    else:
        fdn, fup = oracle.radiative_transfer_direct_sw(0.6, ssi, od.astype(np.float64)), np.zeros((nlay + 1, nwav))
    for b in range(2):
        sl = slice(begin[b], end[b] + 1)
        assert np.allclose(dn[b], fdn[:, sl].sum(1), rtol=1e-11, atol=1e-300)
        assert np.allclose(up[b], fup[:, sl].sum(1), rtol=1e-11, atol=1e-300)
    assert dn[:, 0].sum() == pytest.approx(0.6 * ssi.sum(), rel=1e-12)
    # boundary fluxes of a partial cover: bands [100, 4999] and [9000, end]
    begin2, end2 = np.array([100, 9000]), np.array([4999, nwav - 1])
    _, _, sdn, tup = api.lbl_band_fluxes_sw(ctx, 0.6, dev(ssi), dev(od), begin2, end2, albedo=dev(albedo) if with_albedo else None,
                                            boundary=True)
    covered = np.zeros(nwav, bool); covered[100:5000] = True; covered[9000:] = True
    assert np.allclose(sdn.cpu().numpy()[covered], fdn[-1][covered], rtol=1e-11, atol=1e-300)
    assert np.allclose(tup.cpu().numpy()[covered], fup[0][covered], rtol=1e-11, atol=1e-300)
    assert np.all(sdn.cpu().numpy()[~covered] == 0) and np.all(tup.cpu().numpy()[~covered] == 0)
