"""CPU checks of the oracle restatement (rows a2-a8) against closed-form cases.

The reference ships no golden vectors for this arithmetic, so the
restatement is pinned by analytic identities of the same equations.
"""
import numpy as np
import pytest

from conftest import make_lw_case

D = 1.66


def test_planck_matches_closed_form(oracle):
    # planck_function.cpp:29-50 written independently with numpy
    h, c, k = 6.62606896e-34, 2.99792458e8, 1.3806504e-23
    T = np.array([180.0, 250.0, 300.0])
    wn = np.array([10.0, 667.0, 1500.0, 3000.0])
    dwn = np.array([0.01, 0.02, 0.5, 1.0])
    B = oracle.planck_function(T, wn, dwn)
    f = wn * 100.0 * c
    ref = (dwn * 2.0 * h * 100.0 * c * np.pi / c**2) * f**3 / np.expm1((h / k) * f[None, :] / T[:, None])
    assert np.allclose(B, ref, rtol=1e-12, atol=0)
    # Stefan-Boltzmann: a fine grid integrates to sigma T^4 (within the 1e-3 of the truncated grid)
    wn2 = np.linspace(0.5, 9999.5, 10000)
    tot = oracle.planck_function([288.0], wn2, np.ones_like(wn2)).sum()
    assert abs(tot / (5.670374419e-8 * 288.0**4) - 1.0) < 2e-3


def test_rt_lw_isothermal(oracle):
    # isothermal column, black surface: up = B at every level, dn(surface) = B (1 - prod(1-eps))
    rs = np.random.RandomState(0)
    nlay, nwav = 7, 5
    od = 10.0 ** rs.uniform(-3, 1, (nlay, nwav))
    B = rs.uniform(0.1, 2.0, nwav)
    planck = np.tile(B, (nlay + 1, 1))
    fdn, fup = oracle.radiative_transfer_lw(planck, od, np.ones(nwav), B)
    assert np.allclose(fup, planck, rtol=1e-13)
    trans = np.prod(np.exp(-D * od), axis=0)
    assert np.allclose(fdn[-1], B * (1.0 - trans), rtol=1e-12)
    assert np.all(fdn[0] == 0.0)


def test_rt_lw_small_od_branches(oracle):
    # a3: eps <= 1e-5 -> factor = 0.5 eps; zero optical depth gives zero emissivity, no NaN
    nlay, nwav = 3, 4
    od = np.array([[0.0, 1e-9, 5e-6, 1e-3]] * nlay)
    planck = np.linspace(1.0, 2.0, nlay + 1)[:, None] * np.ones(nwav)
    fdn, fup = oracle.radiative_transfer_lw(planck, od, np.ones(nwav), planck[-1])
    assert np.all(np.isfinite(fdn)) and np.all(np.isfinite(fup))
    assert np.all(fdn[:, 0] == 0.0)
    eps = 1.0 - np.exp(-D * od[0, 1])
    # one thin layer: dn1 = B0 (eps - eps/2) + B1 eps/2
    assert np.isclose(fdn[1, 1], planck[0, 1] * 0.5 * eps + planck[1, 1] * 0.5 * eps, rtol=1e-12)


def test_rt_lw_bb_equals_spectral_sum_when_formulas_coincide(oracle):
    # a4's max-form factor equals a3's branch form when eps > 1e-5 and od > 1e-5/D
    rs = np.random.RandomState(1)
    nlay, nwav = 9, 33
    od = 10.0 ** rs.uniform(-3, 1.5, (nlay, nwav))
    grey = 10.0 ** rs.uniform(-3, 0, nlay)
    wn = np.linspace(100, 2000, nwav)
    T = np.linspace(200, 290, nlay + 1)
    planck = oracle.planck_function(T, wn, np.ones(nwav))
    fdn, fup = oracle.radiative_transfer_lw(planck, od + grey[:, None], np.ones(nwav), planck[-1])
    bdn, bup = oracle.radiative_transfer_lw_bb(planck, od, grey, np.ones(nwav), planck[-1])
    assert np.allclose(bdn, fdn.sum(1), rtol=1e-13)
    assert np.allclose(bup, fup.sum(1), rtol=1e-13)


def test_rt_lw_bb_small_od_formula_differs_from_spectral(oracle):
    # a4 (:117-119) floors the factor at 0.5e-5: a (nearly) transparent layer still mixes 0.5e-5 of
    # planck(l+1) in, unlike a3.  Pin the documented difference so nobody "fixes" it.
    planck = np.array([[1.0], [3.0]])
    bdn, bup = oracle.radiative_transfer_lw_bb(planck, np.zeros((1, 1)), np.zeros(1), np.ones(1), np.array([3.0]))
    assert np.isclose(bdn[1], 1.0 * (0.0 - 0.5e-5) + 3.0 * 0.5e-5, rtol=1e-12)
    fdn, _ = oracle.radiative_transfer_lw(planck, np.zeros((1, 1)), np.ones(1), np.array([3.0]))
    assert fdn[1, 0] == 0.0


def test_rt_sw_direct_and_norayleigh(oracle):
    rs = np.random.RandomState(2)
    nlay, nwav = 6, 11
    od = 10.0 ** rs.uniform(-3, 0.5, (nlay, nwav))
    ssi = rs.uniform(0.1, 1.0, nwav)
    mu0 = 0.5
    fdn = oracle.radiative_transfer_direct_sw(mu0, ssi, od)
    cum = np.vstack([np.zeros(nwav), np.cumsum(od, 0)])
    assert np.allclose(fdn, mu0 * ssi * np.exp(-cum / mu0), rtol=1e-12)
    alb = np.full(nwav, 0.15)
    fdn2, fup2 = oracle.radiative_transfer_norayleigh_sw(mu0, ssi, od, alb)
    assert np.array_equal(fdn, fdn2)
    up_expect = fdn[-1] * alb * np.exp(-2.0 * (cum[-1] - cum))
    assert np.allclose(fup2, up_expect, rtol=1e-12)


def test_heating_rate(oracle):
    p = np.array([100.0, 300.0, 1000.0])
    fdn = np.array([[1.0, 2.0], [3.0, 5.0], [4.0, 4.0]])
    fup = np.array([[6.0, 1.0], [5.0, 2.0], [4.5, 2.5]])
    hr = oracle.heating_rate(p, fdn, fup)
    conv = -(9.80665 / 1004.0) / np.diff(p)
    expect = conv[:, None] * (np.diff(fdn, axis=0) - np.diff(fup, axis=0))
    assert np.allclose(hr, expect, rtol=1e-14)
    hr_sw = oracle.heating_rate(p, fdn, None)
    assert np.allclose(hr_sw, conv[:, None] * np.diff(fdn, axis=0), rtol=1e-14)


def test_reorder_key_lw_structure(oracle):
    p, wn, dwn, od = make_lw_case(2000, nlay=20, seed=3)
    t = oracle.idealised_temperature(p)
    key, col, st = oracle.reorder_key(p, t, wn, dwn, od.astype(np.float64), None, 0.5)
    assert st == 0
    assert np.allclose(col, od.astype(np.float64).sum(0), rtol=1e-13)
    thin = col < 0.5
    assert thin.any() and (~thin).any()
    assert np.array_equal(key[thin], -0.5 + col[thin])
    # thick columns: key is a heating-weighted mean pseudo-height, inside the column
    hmax = np.log(p[-1]) - np.log(p[0])
    assert np.all(key[~thin] > 0.0) and np.all(key[~thin] < hmax)
    # zero columns tie exactly at -threshold
    assert np.all(key[col == 0.0] == -0.5) and (col == 0.0).sum() > 10


def test_reorder_key_sw_threshold_height(oracle):
    p = np.array([1.0, 10.0, 100.0, 1000.0])
    od = np.array([[0.1, 0.0], [0.2, 0.01], [0.3, 0.02]])
    key, col, st = oracle.reorder_key(p, None, np.array([1.0, 2.0]), np.ones(2), od, np.ones(2), 0.25)
    assert st == 0
    h = np.log(p[-1]) - np.log(p)
    # column 0 crosses 0.25 inside layer 1 (cum 0.1 -> 0.3)
    expect = ((0.25 - 0.1) * h[2] + (0.3 - 0.25) * h[1]) / 0.2
    assert np.isclose(key[0], expect, rtol=1e-14)
    assert np.isclose(key[1], 0.03 - 0.25, rtol=1e-14)


def test_stable_argsort_bands_matches_numpy(oracle):
    rs = np.random.RandomState(4)
    n = 5000
    wn = np.linspace(0.0, 100.0, n)
    key = np.round(rs.normal(size=n), 1)  # many ties
    b1 = np.array([10.0, 40.0, 70.0])
    b2 = np.array([40.0, 70.0, 100.0])
    iband, oi, rank = oracle.stable_argsort_bands(wn, key, b1, b2)
    expect_oi = np.arange(n)
    for b in range(3):
        m = (wn >= b1[b]) & ((wn < b2[b]) if b < 2 else (wn <= b2[b]))
        idx = np.nonzero(m)[0]
        assert np.all(iband[idx] == b)
        expect_oi[idx[0]:idx[-1] + 1] = idx[0] + np.argsort(key[idx[0]:idx[-1] + 1], kind="stable")
    assert np.array_equal(oi, expect_oi)
    assert np.array_equal(rank[oi], np.arange(n))
    assert np.all(iband[wn < 10.0] == -1)
