"""GPU parity for create_look_up_table (rows a15, a16): g-point averaging under every averaging
method, min/max, molar-absorption conversion, g-point fractions and the Planck LUT, against the
CPU oracle (oracle_lut.c).  Tolerance rtol 1e-10 (segmented sums are re-associated; the weights
use the device exp)."""
import numpy as np
import pytest
import torch

from conftest import make_lw_case

pytestmark = pytest.mark.gpu

METHODS = ["linear", "transmission", "transmission-2", "transmission-3", "transmission-10", "square-root",
           "logarithmic", "hybrid-logarithmic-transmission-3"]


def _dev(ctx, a):
    return torch.as_tensor(np.ascontiguousarray(a), device=ctx.device)


def _gmap_case(nwav=30000, nlay=24, ng=11, seed=61, empty_g=None):
    p, wn, dwn, od = make_lw_case(nwav, nlay=nlay, seed=seed)
    rs = np.random.RandomState(seed)
    # g points as find_g_points produces them: runs of consecutive wavenumbers share a g point
    run = np.repeat(rs.randint(0, ng, nwav // 40 + 1), 40)[:nwav]
    g_point = run.astype(np.int32)
    g_point[rs.uniform(size=nwav) < 0.01] = -1          # unassigned points
    if empty_g is not None:
        g_point[g_point == empty_g] = (empty_g + 1) % ng
    return p, wn, dwn, od, g_point


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_average_lw_matches_oracle(ctx, oracle, method, dtype):
    from ecckd_amd import api, synthetic as syn
    ng = 11
    p, wn, dwn, od, g_point = _gmap_case(ng=ng)
    od = od.astype(dtype)
    t_hl = syn.temperature_profile(p)
    t_fl = 0.5 * (t_hl[:-1] * p[:-1] + t_hl[1:] * p[1:]) / (0.5 * (p[:-1] + p[1:]))   # create_look_up_table.cpp:316-317
    gm = api.GPointMap(ctx, _dev(ctx, g_point), ng, _dev(ctx, wn), _dev(ctx, dwn))
    ma, mn, mx = gm.average_optical_depth(p, _dev(ctx, od), method, 4.0e-4, temperature_fl=t_fl)
    planck_fl = oracle.planck_function(t_fl, wn, dwn)
    oma, omn, omx, ne = oracle.average_optical_depth_to_g_point(ng, 4.0e-4, p, g_point, od.astype(np.float64),
                                                                planck_fl, method)
    assert ne == 0
    assert np.array_equal(mn, omn) or np.allclose(mn, omn, rtol=1e-14, atol=0)
    assert np.allclose(mx, omx, rtol=1e-14, atol=0)
    assert np.allclose(ma, oma, rtol=1e-10, atol=1e-300)
    assert np.all(ma >= mn * (1 - 1e-15)) and np.all(ma <= mx * (1 + 1e-15))
    gm.close()


@pytest.mark.parametrize("method", ["transmission-3", "logarithmic", "linear"])
def test_average_sw_and_plain_optical_depth(ctx, oracle, method):
    """Shortwave weighting by ssi (create_look_up_table.cpp:335) and reference_surface_vmr <= 0."""
    from ecckd_amd import api, synthetic as syn
    ng = 7
    p, wn, dwn, od, g_point = _gmap_case(nwav=20000, nlay=16, ng=ng, seed=67)
    ssi = syn.solar_spectral_irradiance(wn + 250.0, dwn)
    gm = api.GPointMap(ctx, _dev(ctx, g_point), ng, _dev(ctx, wn), _dev(ctx, dwn))
    ma, mn, mx = gm.average_optical_depth(p, _dev(ctx, od), method, -1.0, ssi=_dev(ctx, ssi))
    w = np.tile(ssi, (16, 1))
    oma, omn, omx, _ = oracle.average_optical_depth_to_g_point(ng, -1.0, p, g_point, od.astype(np.float64), w, method)
    assert np.allclose(ma, oma, rtol=1e-10, atol=1e-300)
    assert np.allclose(mn, omn, rtol=1e-14, atol=0) and np.allclose(mx, omx, rtol=1e-14, atol=0)
    gm.close()


def test_empty_g_point_and_errors(ctx, oracle):
    from ecckd_amd import api, EcckdError, synthetic as syn
    ng = 6
    p, wn, dwn, od, g_point = _gmap_case(nwav=8000, nlay=10, ng=ng, seed=71, empty_g=3)
    gm = api.GPointMap(ctx, _dev(ctx, g_point), ng, _dev(ctx, wn), _dev(ctx, dwn))
    counts = gm.counts()
    assert counts[3] == 0 and counts.sum() == (g_point >= 0).sum()
    t_fl = np.full(10, 250.0)
    ma, mn, mx = gm.average_optical_depth(p, _dev(ctx, od), "transmission", 1.0, temperature_fl=t_fl)
    # "No wavenumbers with g_point == ig: skipping" -> zeros (average_optical_depth.cpp:135-141)
    assert np.all(ma[:, 3] == 0.0) and np.all(mn[:, 3] == 0.0) and np.all(mx[:, 3] == 0.0)
    oma, *_ = oracle.average_optical_depth_to_g_point(ng, 1.0, p, g_point, od.astype(np.float64),
                                                      oracle.planck_function(t_fl, wn, dwn), "transmission")
    assert np.allclose(ma, oma, rtol=1e-10, atol=1e-300)
    with pytest.raises(EcckdError) as e:
        gm.average_optical_depth(p, _dev(ctx, od), "median", 1.0, temperature_fl=t_fl)
    assert e.value.code == 147
    gm.close()
    with pytest.raises(EcckdError) as e:
        api.GPointMap(ctx, _dev(ctx, np.full(100, 9, dtype=np.int32)), 4, _dev(ctx, wn[:100]), _dev(ctx, dwn[:100]))
    assert e.value.code == 147


def test_gpoint_fraction_and_planck_lut(ctx, oracle):
    from ecckd_amd import api
    ng = 9
    p, wn, dwn, od, g_point = _gmap_case(nwav=40000, nlay=4, ng=ng, seed=73)
    gm = api.GPointMap(ctx, _dev(ctx, g_point), ng, _dev(ctx, wn), _dev(ctx, dwn))
    # create_look_up_table.cpp:516-535: 10 cm-1 grid adapted to the bands
    w1 = 10.0 * np.arange(0, 326)
    w2 = w1 + 10.0
    frac = gm.gpoint_fraction(w1, w2)
    ofrac = oracle.gpoint_fraction(ng, g_point, wn, dwn, w1, w2)
    assert np.allclose(frac, ofrac, rtol=1e-12, atol=1e-300)
    assert np.allclose(frac.sum(1), 1.0, rtol=1e-12)
    t_lut = np.arange(120.0, 351.0)                       # :581
    lut = gm.planck_lut(t_lut)
    olut = oracle.planck_lut(ng, t_lut, g_point, wn, dwn)
    assert np.allclose(lut, olut, rtol=1e-11, atol=1e-300)
    gm.close()
