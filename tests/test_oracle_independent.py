"""Independent checks of the CPU oracle (oracle/oracle_*.c) for the floating-point rows of SURVEY 8a: every formula is written
out here a second time, straight from the reference's source lines, in 40-digit arithmetic (mpmath) - no code is shared with
the oracle, whose double-precision results must agree with the many-digit ones to rounding.  The reference ships no golden
vectors and does not build without Adept, so this is what stands behind "the oracle follows the reference":
two restatements by different routes (C in double, here in multi-precision Python) of the lines cited at each function.

Rows: a2 planck_function, a3 radiative_transfer_lw, a4 radiative_transfer_lw_bb, a5 the shortwave pair, a6 heating_rate,
a7 the longwave sorting key, a10-a12 fit_optical_depth_lw (all five averaging methods, with the reference's level shift
in the logarithmic one) + calc_cost_function_lw through CkdEquipartition::calc_error's index mapping, a15 the g-point
averages of create_look_up_table, a17 / a21 the CKD model's optical depths and Planck look-up, a18 the optimiser's cost function
(longwave and shortwave)."""
import math

import mpmath as mp
import numpy as np
import pytest

mp.mp.dps = 40
D = mp.mpf("1.66")                       # LW_DIFFUSIVITY, constants.h:24
G = mp.mpf("9.80665")                    # ACCEL_GRAVITY, constants.h:22
CP = mp.mpf("1004.0")                    # SPECIFIC_HEAT_AIR, constants.h:23


def M(a):
    """numpy array of doubles -> nested lists of mpf (exact)"""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 0:
        return mp.mpf(float(a))
    return [M(x) for x in a]


def F(x):
    """mpf (nested lists) -> float64 array"""
    return np.array(x, dtype=object).astype(np.float64) if isinstance(x, list) else float(x)


def planck_mp(temperature, wn, dwn):
    """planck_function.cpp:29-53"""
    h, c, k = mp.mpf("6.62606896e-34"), mp.mpf("2.99792458e8"), mp.mpf("1.3806504e-23")
    inv = 100 * c
    pi = mp.mpf("3.14159265358979323846")
    out = []
    for t in temperature:
        row = []
        for w, d in zip(wn, dwn):
            freq = w * inv
            pre = (d * 2 * h * inv * pi / (c * c)) * freq ** 3
            row.append(pre / (mp.exp((h / k) * (freq / t)) - 1))
        out.append(row)
    return out


def rt_lw_mp(planck, od, surf_emissivity, surf_planck):
    """radiative_transfer_lw.cpp:27-60: branch form of the factor"""
    nlay, nwav = len(od), len(od[0])
    eps = [[1 - mp.exp(-D * od[l][i]) for i in range(nwav)] for l in range(nlay)]
    fac = [[(1 - eps[l][i] * (1 / D) / od[l][i]) if eps[l][i] > mp.mpf("1.0e-5") else mp.mpf("0.5") * eps[l][i] for i in range(nwav)]
           for l in range(nlay)]
    dn = [[mp.mpf(0)] * nwav for _ in range(nlay + 1)]
    up = [[mp.mpf(0)] * nwav for _ in range(nlay + 1)]
    for l in range(nlay):
        for i in range(nwav):
            dn[l + 1][i] = dn[l][i] * (1 - eps[l][i]) + planck[l][i] * (eps[l][i] - fac[l][i]) + planck[l + 1][i] * fac[l][i]
    for i in range(nwav):
        up[nlay][i] = surf_planck[i] * surf_emissivity[i] + (1 - surf_emissivity[i]) * dn[nlay][i]
    for l in range(nlay - 1, -1, -1):
        for i in range(nwav):
            up[l][i] = up[l + 1][i] * (1 - eps[l][i]) + planck[l + 1][i] * (eps[l][i] - fac[l][i]) + planck[l][i] * fac[l][i]
    return dn, up


def rt_lw_bb_mp(planck, spectral_od, grey_od, surf_emissivity, surf_planck):
    """radiative_transfer_lw.cpp:87-142: max form of the factor, broadband sums per half level"""
    nlay, nwav = len(spectral_od), len(spectral_od[0])
    te = mp.mpf("1.0e-5")

    def eps_fac(l, i):
        od = spectral_od[l][i] + grey_od[l]
        e = 1 - mp.exp(-D * od)
        f = max(1 - (1 / D) * max(e, te) / max(od, te / D), mp.mpf("0.5") * te)
        return e, f
    flux = [mp.mpf(0)] * nwav
    dn, up = [mp.mpf(0)] * (nlay + 1), [mp.mpf(0)] * (nlay + 1)
    for l in range(nlay):
        for i in range(nwav):
            e, f = eps_fac(l, i)
            flux[i] = flux[i] * (1 - e) + planck[l][i] * (e - f) + planck[l + 1][i] * f
        dn[l + 1] = mp.fsum(flux)
    for i in range(nwav):
        flux[i] = surf_planck[i] * surf_emissivity[i] + (1 - surf_emissivity[i]) * flux[i]
    up[nlay] = mp.fsum(flux)
    for l in range(nlay - 1, -1, -1):
        for i in range(nwav):
            e, f = eps_fac(l, i)
            flux[i] = flux[i] * (1 - e) + planck[l + 1][i] * (e - f) + planck[l][i] * f
        up[l] = mp.fsum(flux)
    return dn, up


_exp = np.vectorize(math.exp, otypes=[np.float64])     # the C library's exp, the one the oracle calls: near the branch a last-bit
                                                       # difference in exp is amplified by 1 / eps^2


def rt_lw_np(planck, od, surf_emissivity, surf_planck):
    """The same lines (radiative_transfer_lw.cpp:43-59) once more, in double with the reference's order of operations: where a
    layer's emissivity is small, 1 - exp(-D tau) and 1 - eps/(D tau) lose digits IN DOUBLE (relative error ~1e-16 / eps^2
    in the factor) - in the reference, in the oracle and here alike - so that regime is compared double against double."""
    Dd = 1.66
    eps = 1.0 - _exp(-Dd * od)
    with np.errstate(divide="ignore", invalid="ignore"):
        fac = np.where(eps > 1.0e-5, 1.0 - eps * (1.0 / Dd) / od, 0.5 * eps)
    nlay = od.shape[0]
    dn = np.zeros((nlay + 1, od.shape[1]))
    up = np.zeros_like(dn)
    for l in range(nlay):
        dn[l + 1] = dn[l] * (1.0 - eps[l]) + planck[l] * (eps[l] - fac[l]) + planck[l + 1] * fac[l]
    up[nlay] = surf_planck * surf_emissivity + (1.0 - surf_emissivity) * dn[nlay]
    for l in range(nlay - 1, -1, -1):
        up[l] = up[l + 1] * (1.0 - eps[l]) + planck[l + 1] * (eps[l] - fac[l]) + planck[l] * fac[l]
    return dn, up


def rt_lw_bb_np(planck, spectral_od, grey_od, surf_emissivity, surf_planck):
    """radiative_transfer_lw.cpp:109-141 in double, the reference's order of operations"""
    Dd, te = 1.66, 1.0e-5
    nlay = spectral_od.shape[0]
    flux = np.zeros(spectral_od.shape[1])
    dn, up = np.zeros(nlay + 1), np.zeros(nlay + 1)

    def eps_fac(l):
        od = spectral_od[l] + grey_od[l]
        e = 1.0 - _exp(-Dd * od)
        return e, np.maximum(1.0 - (1.0 / Dd) * np.maximum(e, te) / np.maximum(od, te / Dd), 0.5 * te)
    for l in range(nlay):
        e, f = eps_fac(l)
        flux = flux * (1.0 - e) + planck[l] * (e - f) + planck[l + 1] * f
        dn[l + 1] = flux.sum()
    flux = surf_planck * surf_emissivity + (1.0 - surf_emissivity) * flux
    up[nlay] = flux.sum()
    for l in range(nlay - 1, -1, -1):
        e, f = eps_fac(l)
        flux = flux * (1.0 - e) + planck[l + 1] * (e - f) + planck[l] * f
        up[l] = flux.sum()
    return dn, up


def heating_rate_mp(p, dn, up=None):
    """heating_rate.h:30-50 (a matrix of fluxes) / :55-72 (one profile)"""
    nlay = len(p) - 1
    conv = [-(G / CP) / (p[l + 1] - p[l]) for l in range(nlay)]
    if not isinstance(dn[0], list):
        return [conv[l] * (dn[l + 1] - dn[l] - (up[l + 1] if up else 0) + (up[l] if up else 0)) for l in range(nlay)]
    nwav = len(dn[0])
    return [[conv[l] * (dn[l + 1][i] - dn[l][i] - (up[l + 1][i] if up else 0) + (up[l][i] if up else 0)) for i in range(nwav)]
            for l in range(nlay)]


def _case(nlay, nwav, seed, thin_fraction=0.2):
    rs = np.random.RandomState(seed)
    p = np.concatenate([[1.0], np.cumsum(rs.uniform(50.0, 400.0, nlay))]) * 100000.0 / (1.0 + 400.0 * nlay)
    p = np.sort(p)
    t = np.linspace(190.0, 295.0, nlay + 1) + rs.uniform(-5, 5, nlay + 1)
    wn = np.sort(rs.uniform(20.0, 2500.0, nwav))
    dwn = rs.uniform(0.0005, 0.002, nwav)
    od = 10.0 ** rs.uniform(-2.5, 0.7, (nlay, nwav))
    thin = rs.uniform(size=(nlay, nwav)) < thin_fraction
    # thin layers on both sides of the emissivity threshold 1e-5 of the factor's branch (eps = 2.6e-6 .. 2.6e-5): there the two
    # forms of the factor differ by ~eps/3 relative, far above the tolerances below, while 1 - exp(-D tau) - which the
    # reference itself evaluates in double - still carries eleven digits
    od[thin] = 10.0 ** rs.uniform(-5.8, -4.8, int(thin.sum()))
    return p, t, wn, dwn, od


def test_planck_a2(oracle):
    p, t, wn, dwn, _ = _case(6, 9, 0)
    ref = F(planck_mp(M(t), M(wn), M(dwn)))
    assert np.allclose(oracle.planck_function(t, wn, dwn), ref, rtol=2e-14, atol=0)


def test_radiative_transfer_lw_a3(oracle):
    p, t, wn, dwn, od = _case(7, 11, 1)
    planck = oracle.planck_function(t, wn, dwn)
    emis = np.linspace(0.9, 1.0, wn.size)
    # thin layers on both sides of the branch (and zero optical depth): double against double
    od[0, 0] = 0.0
    dn, up = rt_lw_np(planck, od, emis, planck[-1])
    fdn, fup = oracle.radiative_transfer_lw(planck, od, emis, planck[-1])
    assert np.allclose(fdn, dn, rtol=1e-13, atol=0) and np.allclose(fup, up, rtol=1e-13, atol=0)
    assert np.any((1.0 - np.exp(-1.66 * od) > 1e-5) & (od < 1e-4)) and np.any((1.0 - np.exp(-1.66 * od) <= 1e-5) & (od > 0))
    # well-conditioned layers (eps >= 0.05: the factor keeps thirteen digits): forty digits against the oracle's double
    thick = np.maximum(od, 3e-2)
    dn, up = rt_lw_mp(M(planck), M(thick), M(emis), M(planck[-1]))
    fdn, fup = oracle.radiative_transfer_lw(planck, thick, emis, planck[-1])
    assert np.allclose(fdn, F(dn), rtol=1e-12, atol=1e-300) and np.allclose(fup, F(up), rtol=1e-12, atol=0)
    # and in between the two agree as far as double arithmetic allows: relative error of the factor ~ 1e-16 / eps^2
    mid = np.maximum(od, 3e-3)
    dn, up = rt_lw_mp(M(planck), M(mid), M(emis), M(planck[-1]))
    fdn, fup = oracle.radiative_transfer_lw(planck, mid, emis, planck[-1])
    assert np.allclose(fdn, F(dn), rtol=1e-10, atol=1e-300) and np.allclose(fup, F(up), rtol=1e-10, atol=0)


def test_radiative_transfer_lw_bb_a4(oracle):
    p, t, wn, dwn, od = _case(8, 37, 2)
    planck = oracle.planck_function(t, wn, dwn)
    grey = 10.0 ** np.random.RandomState(3).uniform(-4, 0, od.shape[0])
    emis = np.ones(wn.size)
    dn, up = rt_lw_bb_np(planck, od, grey, emis, planck[-1])
    fdn, fup = oracle.radiative_transfer_lw_bb(planck, od, grey, emis, planck[-1])
    assert np.allclose(fdn, dn, rtol=1e-13, atol=0) and np.allclose(fup, up, rtol=1e-13, atol=0)       # thin layers included
    thick = np.maximum(od, 3e-2)
    dn, up = rt_lw_bb_mp(M(planck), M(thick), M(grey), M(emis), M(planck[-1]))
    fdn, fup = oracle.radiative_transfer_lw_bb(planck, thick, grey, emis, planck[-1])
    assert np.allclose(fdn, F(dn), rtol=1e-12, atol=0) and np.allclose(fup, F(up), rtol=1e-12, atol=0)
    # the max form (a4) and the branch form (a3) of the factor differ where the emissivity is below 1e-5: a column of thin
    # layers alone tells them apart (a3: factor = eps / 2; a4: factor = 5e-6 whatever the emissivity)
    thin = np.full((3, 2), 1.0e-7)
    pl3 = planck[:4, :2]
    a4, _ = oracle.radiative_transfer_lw_bb(pl3, thin, np.zeros(3), np.ones(2), pl3[-1])
    a3, _ = rt_lw_np(pl3, thin, np.ones(2), pl3[-1])
    assert not np.allclose(a3.sum(1)[1:], a4[1:], rtol=1e-3, atol=0)
    b4, _ = rt_lw_bb_np(pl3, thin, np.zeros(3), np.ones(2), pl3[-1])
    assert np.allclose(a4, b4, rtol=1e-14, atol=0)


def test_shortwave_pair_a5_and_heating_rate_a6(oracle):
    rs = np.random.RandomState(4)
    nlay, nwav, mu0 = 6, 8, 0.5
    p = np.sort(rs.uniform(10.0, 1.0e5, nlay + 1))
    od = 10.0 ** rs.uniform(-3, 0.5, (nlay, nwav))
    ssi = rs.uniform(0.1, 2.0, nwav)
    alb = rs.uniform(0.0, 0.5, nwav)
    odm, ssim, albm = M(od), M(ssi), M(alb)
    dn = [[mp.mpf(mu0) * s for s in ssim]]
    for l in range(nlay):                                       # radiative_transfer_sw.cpp:38-42
        dn.append([dn[l][i] * mp.exp(-odm[l][i] / mp.mpf(mu0)) for i in range(nwav)])
    up = [None] * (nlay + 1)
    up[nlay] = [dn[nlay][i] * albm[i] for i in range(nwav)]      # :71
    for l in range(nlay - 1, -1, -1):                           # :72-74, the two-stream zenith angle of 60 degrees
        up[l] = [up[l + 1][i] * mp.exp(-2 * odm[l][i]) for i in range(nwav)]
    assert np.allclose(oracle.radiative_transfer_direct_sw(mu0, ssi, od), F(dn), rtol=1e-14)
    fdn, fup = oracle.radiative_transfer_norayleigh_sw(mu0, ssi, od, alb)
    assert np.allclose(fdn, F(dn), rtol=1e-14) and np.allclose(fup, F(up), rtol=1e-14)
    hr = heating_rate_mp(M(p), dn, up)
    assert np.allclose(oracle.heating_rate(p, fdn, fup), F(hr), rtol=1e-11, atol=1e-22)
    assert np.allclose(oracle.heating_rate(p, fdn), F(heating_rate_mp(M(p), dn)), rtol=1e-11, atol=1e-22)


def test_longwave_sorting_key_a7(oracle):
    """reorder_spectrum.cpp:111-190: idealised temperature, Planck, radiative transfer, cooling only, peak-cooling pseudo
    height; columns thinner than the threshold are keyed by their optical depth (:187-190)."""
    rs = np.random.RandomState(5)
    nlay, nwav, thr = 9, 12, 0.5
    p = np.geomspace(2.0, 1.0e5, nlay + 1)
    wn = np.sort(rs.uniform(50.0, 2200.0, nwav))
    dwn = np.full(nwav, 0.001)
    od = 10.0 ** rs.uniform(-2.0, 0.6, (nlay, nwav))
    od[:, :3] *= 1e-3                                           # three thin columns
    pm = M(p)
    lp = [mp.log(x) for x in pm]
    x0, x1 = mp.log(1), mp.log(100000)                           # :121-124 linear in ln p, extrapolated
    t = [(273.15 - 100.0) + (l - x0) / (x1 - x0) * ((273.15 + 15.0) - (273.15 - 100.0)) for l in lp]
    assert np.allclose(oracle.idealised_temperature(p), F(t), rtol=1e-15)
    planck = planck_mp(t, M(wn), M(dwn))
    one = [mp.mpf(1)] * nwav
    dn, up = rt_lw_mp(planck, M(od), one, planck[-1])
    hr = heating_rate_mp(pm, dn, up)
    hr = [[min(v, mp.mpf(0)) for v in row] for row in hr]        # :175
    ph = [lp[-1] - mp.mpf("0.5") * (lp[l] + lp[l + 1]) for l in range(nlay)]
    dh = [lp[l + 1] - lp[l] for l in range(nlay)]
    key = []
    for i in range(nwav):
        col = mp.fsum(M(od[:, i]))
        k = mp.fsum(hr[l][i] * dh[l] * ph[l] for l in range(nlay)) / mp.fsum(hr[l][i] * dh[l] for l in range(nlay))
        key.append(-thr + col if col < thr else k)
    okey, ocol, st = oracle.reorder_key(p, oracle.idealised_temperature(p), wn, dwn, od, None, thr)
    assert st == 0 and np.allclose(ocol, od.sum(0), rtol=1e-15)
    # the key is a ratio of sums of heating rates, themselves differences of fluxes: conditioning costs a few digits
    assert np.allclose(okey, F(key), rtol=1e-10, atol=0)
    assert np.all(okey[:3] < 0) and np.all(okey[3:] > 0)


def _fit_mp(method, metric, planck, i1, i2):
    """fit_optical_depth_lw, find_g_points.cpp:54-106"""
    nlay = len(metric)
    idx = range(i1, i2 + 1)
    out = []
    for z in range(nlay):
        if method == "logarithmic":
            nz = [i for i in idx if metric[z][i] > 0]
            if not nz:
                out.append(mp.mpf(0))
                continue
            # :86-87: the numerator is weighted by the Planck function at the layer's BASE (iz+1), the denominator sums it at
            # the layer's TOP (iz) - as written in the reference
            v = mp.exp(mp.fsum(mp.log(metric[z][i]) * planck[z + 1][i] for i in nz) / mp.fsum(planck[z][i] for i in nz))
            out.append(v if len(nz) == len(idx) else v * mp.mpf(len(nz)) / mp.mpf(len(idx)))
            continue
        avg = mp.fsum(metric[z][i] * planck[z + 1][i] for i in idx) / mp.fsum(planck[z + 1][i] for i in idx)
        if method == "linear":
            out.append(avg)
        elif method == "transmission":
            out.append(abs(-mp.log(1 - min(mp.mpf("0.9999999999999999"), avg)) / D))
        elif method == "transmission-2":
            out.append(abs(-mp.log(1 - min(mp.mpf("0.9999999999999999"), avg)) / (D * 2)))
        elif method == "square-root":
            out.append(avg * avg)
    return out


def _metric_np(method, od):
    """find_g_points.cpp:1119-1146"""
    if method == "transmission":
        return 1.0 - _exp(-od * 1.66)
    if method == "transmission-2":
        return 1.0 - _exp(-od * 1.66 * 2.0)
    if method == "square-root":
        return np.sqrt(od)
    return od


@pytest.mark.parametrize("method", ["linear", "transmission", "transmission-2", "square-root", "logarithmic"])
def test_interval_error_a10_a11_a12(oracle, method):
    """CkdEquipartition::calc_error for the longwave (find_g_points.cpp:282-337): index mapping (ceil / floor of the bounds),
    fit_optical_depth_lw, calc_cost_function_lw.cpp:24-110 with radiative_transfer_lw_bb and heating_rate_single."""
    p, t, wn, dwn, od = _case(6, 41, 6, thin_fraction=0.1)
    rs = np.random.RandomState(7)
    n, nlay = wn.size, od.shape[0]
    if method == "logarithmic":
        od[:, ::7] = 0.0                                        # some zeros: the third branch of :79-99
        od[2, :] = 0.0                                          # a layer without absorber: the second
    bg = 10.0 ** rs.uniform(-3, 0, od.shape)
    planck = oracle.planck_function(t, wn, dwn)
    emis = np.ones(n)
    fdn, fup = oracle.radiative_transfer_lw(planck, bg + od, emis, planck[-1])
    hr = oracle.heating_rate(p, fdn, fup)
    flux_weight = 0.02
    lw = np.sqrt(p[1:]) - np.sqrt(p[:-1])                        # :1093-1099
    lw /= lw.sum()
    assert np.allclose(oracle.layer_weight(p, 0.0), lw, rtol=1e-15)
    metric = oracle.metric(method, od)
    assert np.allclose(metric, _metric_np(method, od), rtol=1e-15, atol=0)
    eq = oracle.CkdEquipartitionLW(method, flux_weight, lw, p, emis, planck[-1], fdn[-1], fup[0], planck, bg, metric, hr)
    pm, plm, bgm, mem, hrm, lwm = M(p), M(planck), M(bg), M(metric), M(hr), M(lw)
    for b1, b2 in ((0.0, 1.0), (0.1, 0.55), (0.5, 0.5001), (0.33, 0.9)):
        i1, i2 = int(np.ceil(b1 * (n - 1))), int(np.floor(b2 * (n - 1)))
        i2 = max(i2, i1)
        fit = _fit_mp(method, mem, plm, i1, i2)
        sub = lambda a: [row[i1:i2 + 1] for row in a]
        dn, up = rt_lw_bb_mp(sub(plm), sub(bgm), fit, [mp.mpf(1)] * (i2 - i1 + 1), plm[-1][i1:i2 + 1])
        hr_fit = heating_rate_mp(pm, dn, up)
        hr_true = [mp.fsum(hrm[l][i1:i2 + 1]) for l in range(nlay)]
        w = mp.mpf(3600 * 24)
        cost = mp.sqrt(w * w * mp.fsum(lwm[l] * (hr_fit[l] - hr_true[l]) ** 2 for l in range(nlay))
                       + mp.mpf(flux_weight) * ((dn[-1] - mp.fsum(M(fdn[-1][i1:i2 + 1]))) ** 2 + (up[0] - mp.fsum(M(fup[0][i1:i2 + 1]))) ** 2))
        got = eq.calc_error(b1, b2)
        # the error is the norm of DIFFERENCES of sums that agree to many digits for a good fit: absolute floor
        assert got == pytest.approx(float(cost), rel=1e-9, abs=1e-11), (method, b1, b2)


@pytest.mark.parametrize("method", ["linear", "transmission", "transmission-2", "transmission-3", "transmission-10", "square-root",
                                    "logarithmic", "hybrid-logarithmic-transmission-3"])
def test_g_point_average_a15(oracle, method):
    """average_optical_depth_to_g_point (average_optical_depth.cpp:22-190): Planck-weighted average of the metric per g point
    and layer, turned back into an optical depth, clamped into [min, max] of the g point's optical depths, then the molar
    absorption (g * 0.001 * M_air / vmr) * od / dp."""
    rs = np.random.RandomState(8)
    nlay, nwav, ng, vmr = 6, 60, 4, 4.0e-4
    p = np.sort(np.concatenate([rs.uniform(100.0, 9000.0, 3), rs.uniform(12000.0, 1.0e5, nlay - 2)]))     # layers on either side of 100 hPa
    od = 10.0 ** rs.uniform(-3, 0.3, (nlay, nwav))
    if "logarithmic" in method:
        od[:, ::9] = 0.0
    weight = rs.uniform(0.1, 2.0, (nlay, nwav))
    gp = rs.randint(0, ng, nwav)
    gp[:ng] = np.arange(ng)                                     # no empty g point
    got, gmin, gmax, n_empty = oracle.average_optical_depth_to_g_point(ng, vmr, p, gp, od, weight, method)
    assert n_empty == 0
    odm, wm = M(od), M(weight)
    cap = mp.mpf("0.9999999999999999")

    def transmission(l, idx, k):
        s = D * k
        avg = mp.fsum((1 - mp.exp(-odm[l][i] * s)) * wm[l][i] for i in idx) / mp.fsum(wm[l][i] for i in idx)
        return abs(-mp.log(1 - min(cap, avg)) / s)

    def logarithmic(l, idx):
        nz = [i for i in idx if odm[l][i] > 0]
        if not nz:
            return mp.mpf(0)
        if len(nz) == len(idx):
            return mp.exp(mp.fsum(mp.log(odm[l][i]) * wm[l][i] for i in idx) / mp.fsum(wm[l][i] for i in idx))
        return mp.exp(mp.fsum(mp.log(odm[l][i]) * wm[l][i] for i in nz) / mp.fsum(wm[l][i] for i in nz)) * mp.mpf(len(nz)) / mp.mpf(len(idx))

    for l in range(nlay):
        to_molar = (G * mp.mpf("0.001") * mp.mpf("28.970")) / mp.mpf(vmr) / (mp.mpf(float(p[l + 1])) - mp.mpf(float(p[l])))
        pfl = 0.5 * (p[l] + p[l + 1])
        for g in range(ng):
            idx = [i for i in range(nwav) if gp[i] == g]
            ws = mp.fsum(wm[l][i] for i in idx)
            if method == "linear":
                v = mp.fsum(wm[l][i] * odm[l][i] for i in idx) / ws
            elif method.startswith("transmission"):
                v = transmission(l, idx, {"transmission": 1, "transmission-2": 2, "transmission-3": 3, "transmission-10": 10}[method])
            elif method == "square-root":
                v = (mp.fsum(wm[l][i] * mp.sqrt(odm[l][i]) for i in idx) / ws) ** 2
            elif method == "logarithmic":
                v = logarithmic(l, idx)
            else:                                              # :109-133: logarithmic above 100 hPa of pressure, transmission-3 below
                v = logarithmic(l, idx) if pfl > 100.0e2 else transmission(l, idx, 3)
            lo, hi = min(odm[l][i] for i in idx), max(odm[l][i] for i in idx)
            v = max(lo, min(v, hi))                            # :151
            if lo > 0 and lo >= hi:                            # :158-165 (a g point of one wavenumber)
                lo, hi = lo * mp.mpf("0.99"), hi * mp.mpf("1.01")
            assert got[l, g] == pytest.approx(float(v * to_molar), rel=1e-11, abs=1e-300), (method, l, g)
            assert gmin[l, g] == pytest.approx(float(lo * to_molar), rel=1e-13, abs=1e-300)
            assert gmax[l, g] == pytest.approx(float(hi * to_molar), rel=1e-13)


def _fit_sw_mp(method, metric, ssi, i1, i2):
    """fit_optical_depth_sw, find_g_points.cpp:111-164.  Note where the cap sits in the transmission methods: on the SUM, before
    the normalisation (:123-124) - unlike the longwave - as written in the reference."""
    idx = range(i1, i2 + 1)
    norm = 1 / mp.fsum(ssi[i] for i in idx)
    cap = mp.mpf("0.9999999999999999")
    out = []
    for z in range(len(metric)):
        if method == "logarithmic":
            nz = [i for i in idx if metric[z][i] > 0]
            if not nz:
                out.append(mp.mpf(0))
                continue
            v = mp.exp(mp.fsum(mp.log(metric[z][i]) * ssi[i] for i in nz) / mp.fsum(ssi[i] for i in nz))
            out.append(v if len(nz) == len(idx) else v * mp.mpf(len(nz)) / mp.mpf(len(idx)))
            continue
        tot = mp.fsum(metric[z][i] * ssi[i] for i in idx)
        if method == "linear":
            out.append(tot * norm)
        elif method == "transmission":
            out.append(abs(-mp.log(1 - min(cap, tot) * norm) / D))
        elif method == "transmission-2":
            out.append(abs(-mp.log(1 - min(cap, tot) * norm) / (D * 2)))
        elif method == "square-root":
            out.append((tot * norm) ** 2)
    return out


def _fit_sw_total_trans_mp(ssi, bg, od, i1, i2):
    """fit_optical_depth_sw_total_trans, find_g_points.cpp:171-204: the grey optical depth that reproduces the broadband direct
    transmission of every layer at a zenith angle of 60 degrees, less that of the background alone"""
    idx = list(range(i1, i2 + 1))
    nz = len(od)
    flux = [ssi[i] for i in idx]
    bgflux = list(flux)
    top = bgtop = mp.fsum(flux)
    fit = [mp.mpf(0)] * nz
    for z in range(nz):
        bgflux = [bgflux[k] * mp.exp(-2 * bg[z][i]) for k, i in enumerate(idx)]
        flux = [flux[k] * mp.exp(-2 * (bg[z][i] + od[z][i])) for k, i in enumerate(idx)]
        bgbase, base = mp.fsum(bgflux), mp.fsum(flux)
        assert bgbase > 0 and base > 0          # the fall-back of :196-198 (everything absorbed) is not reached by this case
        fit[z] = -mp.mpf("0.5") * mp.log(base / top) - (-mp.mpf("0.5") * mp.log(bgbase / bgtop))
        top, bgtop = base, bgbase
    return fit


def _cost_sw_mp(mu0, p, ssi, albedo, bg, fit, fds_true, fut_true, hr_true, flux_weight, lw):
    """calc_cost_function_sw.cpp:24-108 over radiative_transfer_direct_sw_bb / _norayleigh_sw_bb (radiative_transfer_sw.cpp:118-184);
    the heating rate of the fit comes from the DIRECT beam alone (:90)"""
    nlay, n = len(bg), len(ssi)
    flux = [mu0 * s for s in ssi]
    dn = [mu0 * mp.fsum(ssi)]
    for l in range(nlay):
        flux = [flux[i] * mp.exp((-1 / mu0) * (bg[l][i] + fit[l])) for i in range(n)]
        dn.append(mp.fsum(flux))
    up0 = mp.mpf(0)
    if albedo > 0:
        flux = [f * albedo for f in flux]
        for l in range(nlay - 1, -1, -1):
            flux = [flux[i] * mp.exp(-2 * (bg[l][i] + fit[l])) for i in range(n)]
        up0 = mp.fsum(flux)
    hr_fit = heating_rate_mp(p, dn)
    w = mp.mpf(3600 * 24)
    return mp.sqrt(w * w * mp.fsum(lw[l] * (hr_fit[l] - hr_true[l]) ** 2 for l in range(nlay))
                   + flux_weight * ((dn[-1] - fds_true) ** 2 + (up0 - fut_true) ** 2))


@pytest.mark.parametrize("albedo", [0.0, 0.15])
@pytest.mark.parametrize("method", ["linear", "transmission", "transmission-2", "square-root", "logarithmic", "total-transmission"])
def test_interval_error_shortwave_a5_a11_a12(oracle, method, albedo):
    """CkdEquipartition::calc_error, shortwave branches (find_g_points.cpp:338-402): fit_optical_depth_sw or, for
    total-transmission, fit_optical_depth_sw_total_trans with the two scaled evaluations against their own truths, averaged."""
    rs = np.random.RandomState(9)
    nlay, n, mu0 = 6, 37, 0.5
    p = np.sort(rs.uniform(50.0, 1.0e5, nlay + 1))
    od = 10.0 ** rs.uniform(-3, 0.2, (nlay, n))
    if method == "logarithmic":
        od[:, ::6] = 0.0
        od[3, :] = 0.0
    bg = 10.0 ** rs.uniform(-3, -0.5, (nlay, n))
    ssi = rs.uniform(0.2, 2.0, n)
    lw = np.sqrt(p[1:]) - np.sqrt(p[:-1])
    lw /= lw.sum()
    flux_weight = 0.02

    def truth(scale):                                            # find_g_points.cpp:1003-1006, :1011-1034: direct-beam heating rate
        if albedo > 0:
            d, u = oracle.radiative_transfer_norayleigh_sw(mu0, ssi, bg + scale * od, np.full(n, albedo))
            fut = u[0].copy()
        else:
            d = oracle.radiative_transfer_direct_sw(mu0, ssi, bg + scale * od)
            fut = np.zeros(n)
        return oracle.heating_rate(p, d, None), d[-1].copy(), fut
    hr, fds, fut = truth(1.0)
    if method != "total-transmission":
        fut = np.zeros(n)                                        # flux_up = 0 for the truth of the other methods (:1003-1006)
    metric = od if method == "total-transmission" else oracle.metric(method, od)
    extras = None
    if method == "total-transmission":
        extras = dict(min_scaling=0.5, max_scaling=2.5)
        for tag, sc in (("low", 0.5), ("high", 2.5)):
            h, f, u = truth(sc)
            extras[f"hr_{tag}"], extras[f"flux_dn_surf_{tag}"], extras[f"flux_up_toa_{tag}"] = h, f, u
    eq = oracle.CkdEquipartitionSW(method, flux_weight, lw, mu0, p, ssi, albedo, fds, fut, bg, metric, hr, extras)
    pm, ssim, bgm, odm, mem, lwm = M(p), M(ssi), M(bg), M(od), M(metric), M(lw)
    for b1, b2 in ((0.0, 1.0), (0.2, 0.7), (0.6, 0.95)):
        i1, i2 = int(np.ceil(b1 * (n - 1))), max(int(np.floor(b2 * (n - 1))), int(np.ceil(b1 * (n - 1))))
        sl = slice(i1, i2 + 1)
        sub = lambda a: [row[sl] for row in a]

        def cost(fit, h, f, u):
            return _cost_sw_mp(mp.mpf(mu0), pm, ssim[sl], mp.mpf(albedo), sub(bgm), fit, mp.fsum(M(f[sl])), mp.fsum(M(u[sl])),
                               [mp.fsum(row[sl]) for row in M(h)], mp.mpf(flux_weight), lwm)
        if method == "total-transmission":
            fit = _fit_sw_total_trans_mp(ssim, bgm, odm, i1, i2)
            want = mp.mpf("0.5") * (cost([v * mp.mpf("0.5") for v in fit], extras["hr_low"], extras["flux_dn_surf_low"], extras["flux_up_toa_low"])
                                    + cost([v * mp.mpf("2.5") for v in fit], extras["hr_high"], extras["flux_dn_surf_high"], extras["flux_up_toa_high"]))
        else:
            want = cost(_fit_sw_mp(method, mem, ssim, i1, i2), hr, fds, fut)
        got = eq.calc_error(b1, b2)
        if mp.isnan(want) or not mp.im(want) == 0:
            assert np.isnan(got)
        else:
            assert got == pytest.approx(float(want), rel=1e-9, abs=1e-11), (method, albedo, b1, b2)


# ---- a17 / a18 / a21: the CKD model's optical depths, its Planck look-up and the optimiser's cost function ----------------

def _ckd_od_mp(model, gas, p_hl, t_fl, vmr):
    """CkdModel::calc_optical_depth, ckd_model.cpp:925-1085 (linear interpolation of the coefficients): pressure and
    temperature indices clamped to [0, n - 1.0001], the temperature grid of the look-up table itself interpolated in
    pressure, a third interpolation in log concentration for a gas with a look-up table in concentration."""
    logp = M(model["log_pressure"])
    temp = M(model["temperature"])                   # (nt, np)
    nt, npr = len(temp), len(temp[0])
    dlp = logp[1] - logp[0]
    dt = temp[1][0] - temp[0][0]
    gw = 1 / (G * mp.mpf("0.001") * mp.mpf("28.970"))
    k = M(gas["molar_abs"])
    ng = len(k[0][0]) if gas["conc"] != "lut" else len(k[0][0][0])
    out = []
    for c in range(len(p_hl)):
        col = []
        for l in range(len(p_hl[c]) - 1):
            pi = (mp.log(mp.mpf("0.5") * (p_hl[c][l + 1] + p_hl[c][l])) - logp[0]) / dlp
            pi = max(mp.mpf(0), min(pi, npr - mp.mpf("1.0001")))
            ip0 = int(mp.floor(pi))
            pw1 = pi - ip0
            pw0 = 1 - pw1
            t0 = pw0 * temp[0][ip0] + pw1 * temp[0][ip0 + 1]
            ti = (t_fl[c][l] - t0) / dt
            ti = max(mp.mpf(0), min(ti, nt - mp.mpf("1.0001")))
            it0 = int(mp.floor(ti))
            tw1 = ti - it0
            tw0 = 1 - tw1
            simple = gw * (p_hl[c][l + 1] - p_hl[c][l])

            def bilinear(tab):
                return [tw0 * (pw0 * tab[it0][ip0][g] + pw1 * tab[it0][ip0 + 1][g]) + tw1 * (pw0 * tab[it0 + 1][ip0][g] + pw1 * tab[it0 + 1][ip0 + 1][g])
                        for g in range(ng)]
            if gas["conc"] == "none":
                col.append([simple * v for v in bilinear(k)])
            elif gas["conc"] == "lut":
                vm = M(gas["vmr"])
                ci = (mp.log(vmr[c][l]) - mp.log(vm[0])) / mp.log(vm[1] / vm[0])
                ci = max(mp.mpf(0), min(ci, len(vm) - mp.mpf("1.0001")))
                ic0 = int(mp.floor(ci))
                cw1 = ci - ic0
                a, b = bilinear(k[ic0]), bilinear(k[ic0 + 1])
                col.append([simple * vmr[c][l] * ((1 - cw1) * a[g] + cw1 * b[g]) for g in range(ng)])
            else:
                w = simple * (vmr[c][l] - mp.mpf(gas.get("reference_vmr", 0.0)) if gas["conc"] == "relative-linear" else vmr[c][l])
                col.append([w * v for v in bilinear(k)])
        out.append(col)
    return out


def _ckd_planck_mp(model, t):
    """CkdModel::calc_planck_function, ckd_model.cpp:1111-1137: linear in temperature, extrapolated above the table, towards zero
    below it"""
    tp, pf = M(model["temperature_planck"]), M(model["planck_function"])
    dt = tp[1] - tp[0]
    out = []
    for x in t:
        ti = (x - tp[0]) / dt
        if ti >= 0:
            i0 = min(int(mp.floor(ti)), len(tp) - 2)
            w1 = ti - i0
            out.append([(1 - w1) * pf[i0][g] + w1 * pf[i0 + 1][g] for g in range(len(pf[0]))])
        else:
            out.append([(x / tp[0]) * v for v in pf[0]])
    return out


def _cost_ckd_lw_mp(p, planck, od, flux_dn, flux_up, hr, sfd, sfu, cfg, lw, rel_dn, rel_up, band_of_g, nband):
    """calc_cost_function_ckd_lw, calc_cost_function_lw.cpp:114-241"""
    nlay, ng = len(od), len(od[0])
    one = [mp.mpf(1)] * ng
    dn, up = rt_lw_mp(planck, od, one, planck[-1])
    if rel_dn is not None:
        dn = [[dn[i][g] - rel_dn[i][g] for g in range(ng)] for i in range(nlay + 1)]
        up = [[up[i][g] - rel_up[i][g] for g in range(ng)] for i in range(nlay + 1)]
    bdn = [[mp.fsum(dn[i][g] for g in range(ng) if band_of_g[g] == b) for b in range(nband)] for i in range(nlay + 1)]
    bup = [[mp.fsum(up[i][g] for g in range(ng) if band_of_g[g] == b) for b in range(nband)] for i in range(nlay + 1)]
    hrf = heating_rate_mp(p, bdn, bup)
    w = mp.mpf(3600 * 24)
    fw, fpw, bw, sbw = (mp.mpf(cfg[k]) for k in ("flux_weight", "flux_profile_weight", "broadband_weight", "spectral_boundary_weight"))
    iw = [fpw * mp.mpf("0.5") * (lw[l] + lw[l + 1]) for l in range(nlay - 1)]
    cost = mp.mpf(0)
    for b in range(nband):
        cost += w * w * mp.fsum(lw[l] * (hrf[l][b] - hr[l][b]) ** 2 for l in range(nlay)) \
            + fw * ((bdn[nlay][b] - flux_dn[nlay][b]) ** 2 + (bup[0][b] - flux_up[0][b]) ** 2)
        if fpw > 0:
            cost += mp.fsum(iw[i - 1] * ((bdn[i][b] - flux_dn[i][b]) ** 2 + (bup[i][b] - flux_up[i][b]) ** 2) for i in range(1, nlay))
    cost = cost * (1 - bw) / nband \
        + bw * w * w * mp.fsum(lw[l] * mp.fsum(hrf[l][b] - hr[l][b] for b in range(nband)) ** 2 for l in range(nlay)) \
        + bw * fw * (mp.fsum(bdn[nlay][b] - flux_dn[nlay][b] for b in range(nband)) ** 2 + mp.fsum(bup[0][b] - flux_up[0][b] for b in range(nband)) ** 2)
    if fpw > 0:
        cost += bw * mp.fsum(iw[i - 1] * (mp.fsum(bdn[i][b] - flux_dn[i][b] for b in range(nband)) ** 2
                                          + mp.fsum(bup[i][b] - flux_up[i][b] for b in range(nband)) ** 2) for i in range(1, nlay))
    if sbw > 0 and sfd is not None:
        cost += sbw * mp.fsum((dn[nlay][g] - sfd[g]) ** 2 + (up[0][g] - sfu[g]) ** 2 for g in range(ng))
    return cost


def test_ckd_optical_depth_planck_and_cost_a17_a18(oracle):
    """The optimiser's forward model on a small CKD model with one gas of each concentration dependence (none, linear,
    relative-linear, look-up table): optical depths and Planck look-up against the restatements above, then the whole
    longwave cost of one column - band sums, heating rates, boundary / profile / broadband / spectral-boundary terms,
    relative-to fluxes - in forty digits against oracle_ckd.c's double."""
    import ckd_synth as cs
    model = cs.make_model(ng=6, nt=4, np_=7, nband=2, seed=21, nconc=3)
    scenes = cs.make_scenes(model, nscene=1, ncol=2, nlay=5, seed=22)
    sc = scenes[0]
    cfg = dict(flux_weight=0.2, flux_profile_weight=0.3, broadband_weight=0.4, spectral_boundary_weight=0.05, negative_od_penalty=1e4,
               pressure_weight_power=0.5)
    orc = cs.Oracle(oracle, model, scenes, cfg)
    x = orc.x0 + 0.05 * np.random.RandomState(23).standard_normal(orc.x0.size)
    ks = orc.coeffs(x)
    p, T, vmr = sc["pressure_hl"], sc["temperature_hl"], sc["vmr_fl"]
    t_fl = (T[:, :-1] * p[:, :-1] + T[:, 1:] * p[:, 1:]) / (p[:, :-1] + p[:, 1:])             # solve_adept.cpp:37-39
    total = orc.optical_depth(x, sc)
    want = None
    kinds = set()
    for i, g in enumerate(model["gases"]):
        kinds.add(g["conc"])
        od = _ckd_od_mp(model, dict(g, molar_abs=ks[i]), M(p), M(t_fl), M(vmr[:, i, :]))
        want = od if want is None else [[[want[c][l][q] + od[c][l][q] for q in range(len(od[c][l]))] for l in range(len(od[c]))] for c in range(len(od))]
    assert {"none", "lut"} <= kinds and len(kinds) >= 3, kinds
    assert np.allclose(total, F(want), rtol=1e-12, atol=0)
    # Planck look-up, inside and on both sides of the table
    tp = np.asarray(model["temperature_planck"])
    tq = np.array([tp[0] - 25.0, tp[0], 0.5 * (tp[2] + tp[3]), tp[-1], tp[-1] + 30.0])
    assert np.allclose(orc.planck(tq), F(_ckd_planck_mp(model, M(tq))), rtol=1e-14, atol=0)
    # the cost of column 0: truth fluxes perturbed so that every term is non-zero, relative-to fluxes, spectral boundary fluxes
    rs = np.random.RandomState(24)
    nlay, ng, nband = p.shape[1] - 1, total.shape[2], model["nband"]
    ib = np.asarray(model["iband_per_g"])
    c = 0
    pl = orc.planck(T[c])
    fl = orc.fluxes(x, sc)[c]                                      # (2, nhl, ng)
    band = lambda a: np.stack([a[:, ib == b].sum(-1) for b in range(nband)], axis=-1)
    fd_true, fu_true = band(fl[0]) * (1 + 0.03 * rs.standard_normal((nlay + 1, nband))), band(fl[1]) * (1 + 0.03 * rs.standard_normal((nlay + 1, nband)))
    hr_true = oracle.heating_rate(p[c], fd_true, fu_true)
    rel_dn, rel_up = 0.1 * fl[0] * rs.uniform(0.5, 1.5, fl[0].shape), 0.1 * fl[1] * rs.uniform(0.5, 1.5, fl[1].shape)
    sfd, sfu = fl[0][-1] * rs.uniform(0.9, 1.1, ng), fl[1][0] * rs.uniform(0.9, 1.1, ng)
    lw = np.sqrt(p[c, 1:]) - np.sqrt(p[c, :-1])
    lw /= lw.sum()
    import ctypes as C
    P = oracle._p
    cc = np.ascontiguousarray
    L = oracle.lib()
    L.orc_calc_cost_function_ckd_lw.restype = C.c_double
    od_c = np.maximum(total[c], 0.0)
    got = L.orc_calc_cost_function_ckd_lw(
        C.c_int(nlay), C.c_int(ng), C.c_int(nband), P(cc(p[c])), P(cc(pl)), P(np.ones(nband)), P(cc(pl[-1])), P(cc(od_c)), P(cc(fd_true)),
        P(cc(fu_true)), P(cc(hr_true)), P(cc(sfd)), P(cc(sfu)), C.c_double(cfg["flux_weight"]), C.c_double(cfg["flux_profile_weight"]),
        C.c_double(cfg["broadband_weight"]), C.c_double(cfg["spectral_boundary_weight"]), P(cc(lw)), P(cc(rel_dn)), P(cc(rel_up)),
        cc(ib, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int)))
    want = _cost_ckd_lw_mp(M(p[c]), M(pl), M(od_c), M(fd_true), M(fu_true), M(hr_true), M(sfd), M(sfu), cfg, M(lw), M(rel_dn), M(rel_up),
                           list(ib), nband)
    assert got == pytest.approx(float(want), rel=1e-10)


def _cost_ckd_sw_mp(mu0, p, ssi, albedo_band, od, flux_dn, flux_up, hr, sfd, sbw, cfg, lw, rel_dn, rel_up, band_of_g, nband):
    """calc_cost_function_ckd_sw, calc_cost_function_sw.cpp:116-277: direct beam (and, unless every band's albedo is <= 0, the
    surface-reflected flux with the band's albedo), band sums, heating rate from the DOWNWELLING flux alone (:198), twenty-fold
    weight on the upwelling flux at the top (:215), broadband upwelling terms only if every band has a positive albedo
    (:249, :261), per-g boundary weights on the surface flux (:269-272)."""
    nlay, ng = len(od), len(od[0])
    dn = [[mu0 * s for s in ssi]]
    for l in range(nlay):
        dn.append([dn[l][g] * mp.exp(-od[l][g] / mu0) for g in range(ng)])
    up = [[mp.mpf(0)] * ng for _ in range(nlay + 1)]
    if not all(a <= 0 for a in albedo_band):
        up[nlay] = [dn[nlay][g] * albedo_band[band_of_g[g]] for g in range(ng)]
        for l in range(nlay - 1, -1, -1):
            up[l] = [up[l + 1][g] * mp.exp(-2 * od[l][g]) for g in range(ng)]
    if rel_dn is not None:
        dn = [[dn[i][g] - rel_dn[i][g] for g in range(ng)] for i in range(nlay + 1)]
        up = [[up[i][g] - rel_up[i][g] for g in range(ng)] for i in range(nlay + 1)]
    bdn = [[mp.fsum(dn[i][g] for g in range(ng) if band_of_g[g] == b) for b in range(nband)] for i in range(nlay + 1)]
    bup = [[mp.fsum(up[i][g] for g in range(ng) if band_of_g[g] == b) for b in range(nband)] for i in range(nlay + 1)]
    hrf = heating_rate_mp(p, bdn)
    w = mp.mpf(3600 * 24)
    fw, fpw, bw = (mp.mpf(cfg[k]) for k in ("flux_weight", "flux_profile_weight", "broadband_weight"))
    iw = [fpw * mp.mpf("0.5") * (lw[l] + lw[l + 1]) for l in range(nlay - 1)]
    cost = mp.mpf(0)
    for b in range(nband):
        cost += w * w * mp.fsum(lw[l] * (hrf[l][b] - hr[l][b]) ** 2 for l in range(nlay)) \
            + fw * ((bdn[nlay][b] - flux_dn[nlay][b]) ** 2 + 20 * (bup[0][b] - flux_up[0][b]) ** 2)
        if fpw > 0:
            cost += mp.fsum(iw[i - 1] * ((bdn[i][b] - flux_dn[i][b]) ** 2 + (bup[i][b] - flux_up[i][b]) ** 2) for i in range(1, nlay))
    all_pos = all(a > 0 for a in albedo_band)
    if bw > 0:
        cost = cost * (1 - bw) / nband + bw * w * w * mp.fsum(lw[l] * mp.fsum(hrf[l][b] - hr[l][b] for b in range(nband)) ** 2 for l in range(nlay))
        cost += bw * fw * mp.fsum(bdn[nlay][b] - flux_dn[nlay][b] for b in range(nband)) ** 2
        if all_pos:
            cost += bw * fw * mp.fsum(bup[0][b] - flux_up[0][b] for b in range(nband)) ** 2
        if fpw > 0:
            cost += bw * mp.fsum(iw[i - 1] * mp.fsum(bdn[i][b] - flux_dn[i][b] for b in range(nband)) ** 2 for i in range(1, nlay))
            if all_pos:
                cost += bw * mp.fsum(iw[i - 1] * mp.fsum(bup[i][b] - flux_up[i][b] for b in range(nband)) ** 2 for i in range(1, nlay))
    if sbw is not None and sfd is not None:
        cost += mp.fsum(sbw[g] * (dn[nlay][g] - sfd[g]) ** 2 for g in range(ng))
    return cost


@pytest.mark.parametrize("albedo_case", ["all positive", "one band without", "none"])
def test_ckd_cost_shortwave_a18(oracle, albedo_case):
    import ctypes as C
    rs = np.random.RandomState(31)
    nlay, ng, nband, mu0 = 6, 7, 3, 0.6
    p = np.sort(rs.uniform(100.0, 1.0e5, nlay + 1))
    od = 10.0 ** rs.uniform(-2.5, 0.0, (nlay, ng))
    ssi = rs.uniform(20.0, 300.0, ng)
    ib = np.array([0, 0, 1, 1, 1, 2, 2])
    albedo = {"all positive": [0.15, 0.1, 0.2], "one band without": [0.15, 0.0, 0.2], "none": [0.0, 0.0, 0.0]}[albedo_case]
    albedo = np.array(albedo)
    band = lambda a: np.stack([a[:, ib == b].sum(-1) for b in range(nband)], axis=-1)
    if np.all(albedo <= 0):
        d, u = oracle.radiative_transfer_direct_sw(mu0, ssi, od), np.zeros((nlay + 1, ng))
    else:
        d, u = oracle.radiative_transfer_norayleigh_sw(mu0, ssi, od, albedo[ib])
    fd_true = band(d) * (1 + 0.02 * rs.standard_normal((nlay + 1, nband)))
    fu_true = band(u) * (1 + 0.02 * rs.standard_normal((nlay + 1, nband))) + 0.01
    hr_true = oracle.heating_rate(p, fd_true, None)
    rel_dn, rel_up = 0.05 * d * rs.uniform(0.5, 1.5, d.shape), 0.05 * u * rs.uniform(0.5, 1.5, u.shape)
    sfd, sbw = d[-1] * rs.uniform(0.9, 1.1, ng), rs.uniform(0.0, 0.05, ng)
    lw = np.sqrt(p[1:]) - np.sqrt(p[:-1])
    lw /= lw.sum()
    cfg = dict(flux_weight=0.2, flux_profile_weight=0.3, broadband_weight=0.4)
    P, cc = oracle._p, np.ascontiguousarray
    L = oracle.lib()
    L.orc_calc_cost_function_ckd_sw.restype = C.c_double
    got = L.orc_calc_cost_function_ckd_sw(
        C.c_int(nlay), C.c_int(ng), C.c_int(nband), C.c_double(mu0), P(cc(p)), P(cc(ssi)), P(cc(albedo)), P(cc(od)), P(cc(fd_true)),
        P(cc(fu_true)), P(cc(hr_true)), P(cc(sfd)), C.c_double(cfg["flux_weight"]), C.c_double(cfg["flux_profile_weight"]),
        C.c_double(cfg["broadband_weight"]), P(cc(sbw)), P(cc(lw)), P(cc(rel_dn)), P(cc(rel_up)),
        cc(ib, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int)))
    want = _cost_ckd_sw_mp(mp.mpf(mu0), M(p), M(ssi), M(albedo), M(od), M(fd_true), M(fu_true), M(hr_true), M(sfd), M(sbw), cfg, M(lw),
                           M(rel_dn), M(rel_up), list(ib), nband)
    assert got == pytest.approx(float(want), rel=1e-11)
