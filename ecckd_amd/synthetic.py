"""Deterministic synthetic CKDMIP-like spectra.

The ~700 GB CKDMIP line-by-line dataset is not available, so tests and bench.py run
the hot path on synthetic columns with the same shapes and statistics: pressure grid
log-spaced to 1013.25 hPa with a thin top layer, Lorentz-line optical depths stored as
FLOAT like the CKDMIP files, column optical depths spanning ~1e-6..1e4 and ~5 % exactly
zero columns (which produce the large tie groups of reorder_spectrum.cpp:187-190).

`xp` is numpy (tests, CPU baseline) or torch (device-resident generation in bench.py);
the line parameters always come from numpy's RandomState(seed) on the host.
"""
import numpy as np

SEED_BASE = 20260501


def pressure_grid(nlay=54):
    """Half-level pressures (Pa): 0.01 Pa lid, then log-spaced 1 Pa .. 101325 Pa."""
    p = np.empty(nlay + 1)
    p[0] = 0.01
    p[1:] = np.exp(np.linspace(np.log(1.0), np.log(101325.0), nlay))
    return p


def temperature_profile(pressure_hl):
    """Smooth 190-300 K profile: tropospheric lapse, isothermal tropopause, warm stratopause."""
    p = np.asarray(pressure_hl, dtype=np.float64)
    x = np.log(p / 101325.0)  # 0 at surface, negative aloft
    trop = 300.0 + 22.0 * np.maximum(x, -5.0)          # ~6.5 K/km analogue down to 190
    strat = 190.0 + 60.0 * np.exp(-0.5 * ((x + 7.0) / 1.6) ** 2)
    return np.maximum(trop, strat)


def wavenumber_grid(nwav, lo=0.0, hi=3260.0):
    """Uniform mid-point grid of nwav intervals on [lo, hi] (cm-1) and its spacing."""
    dw = (hi - lo) / nwav
    wn = lo + (np.arange(nwav, dtype=np.float64) + 0.5) * dw
    return wn, np.full(nwav, dw)


def line_parameters(seed, nlines, lo, hi, log10_strength_sigma=1.5):
    rs = np.random.RandomState(seed)
    centre = rs.uniform(lo, hi, nlines)
    strength = 10.0 ** rs.normal(0.0, log10_strength_sigma, nlines)
    gamma0 = rs.uniform(0.04, 0.12, nlines) * (hi - lo) / 3260.0 * 8.0
    return centre, strength, gamma0


def optical_depth(xp, pressure_hl, wavenumber, seed, nlines=96, column_scale=30.0, zero_fraction=0.05,
                  continuum=1.0e-7, dtype="float32", device=None, chunk=1 << 19, lo=None, hi=None):
    """(nlay, nwav) layer optical depths of one synthetic gas.

    od[l, j] = dp_l/p_s * column_scale * (continuum + sum_k S_k * Lorentz(nu_j - nu_k; gamma_k * p_l/p_s)),
    with a random ~zero_fraction of the columns set exactly to zero.
    """
    p = np.asarray(pressure_hl, dtype=np.float64)
    nlay = p.size - 1
    is_torch = xp.__name__ == "torch"
    if is_torch:
        wn_host = wavenumber.detach().cpu().numpy() if hasattr(wavenumber, "detach") else np.asarray(wavenumber)
    else:
        wn_host = np.asarray(wavenumber)
    nwav = wn_host.size
    lo = float(wn_host[0]) if lo is None else lo
    hi = float(wn_host[-1]) if hi is None else hi
    centre, strength, gamma0 = line_parameters(seed, nlines, lo, hi)
    ps = p[-1]
    dp = (p[1:] - p[:-1]) / ps
    pfl = 0.5 * (p[1:] + p[:-1]) / ps
    rs = np.random.RandomState(seed + 7919)
    zero_cols = rs.uniform(size=nwav) < zero_fraction

    if is_torch:
        kw = dict(dtype=xp.float64, device=device)
        out = xp.empty((nlay, nwav), dtype=getattr(xp, dtype), device=device)
        c_t = xp.as_tensor(centre, **kw)[:, None]
        s_t = xp.as_tensor(strength, **kw)[:, None]
        g_t = xp.as_tensor(gamma0, **kw)[:, None]
        wn_t = xp.as_tensor(wn_host, **kw)
        z_t = xp.as_tensor(zero_cols, device=device)
        for j0 in range(0, nwav, chunk):
            j1 = min(nwav, j0 + chunk)
            d2 = (wn_t[None, j0:j1] - c_t) ** 2
            for l in range(nlay):
                g = g_t * max(pfl[l], 1.0e-4) + 2.0e-4
                line = (s_t * g / (np.pi * (d2 + g * g))).sum(0)
                row = dp[l] * column_scale * (continuum + line)
                row = xp.where(z_t[j0:j1], xp.zeros_like(row), row)
                out[l, j0:j1] = row.to(out.dtype)
        return out

    out = np.empty((nlay, nwav), dtype=dtype)
    for j0 in range(0, nwav, chunk):
        j1 = min(nwav, j0 + chunk)
        d2 = (wn_host[None, j0:j1] - centre[:, None]) ** 2
        for l in range(nlay):
            g = gamma0[:, None] * max(pfl[l], 1.0e-4) + 2.0e-4
            line = (strength[:, None] * g / (np.pi * (d2 + g * g))).sum(0)
            row = dp[l] * column_scale * (continuum + line)
            row[zero_cols[j0:j1]] = 0.0
            out[l, j0:j1] = row.astype(dtype)
    return out


def band_line_parameters(seed, nlines, lo, hi, nclusters=7):
    """Line list with the structure of a molecular spectrum rather than uniform noise: `nclusters` vibration-rotation
    bands (random centres, widths 25-120 cm-1, strengths spread over five decades), each a comb of lines whose strengths
    fall off from the band centre with a Boltzmann-like envelope times a log-normal factor (hot bands, isotopologues).
    -> (centre, strength, gamma0), sorted by centre."""
    rs = np.random.RandomState(seed)
    span = hi - lo
    band_centre = rs.uniform(lo + 0.03 * span, hi - 0.03 * span, nclusters)
    band_width = rs.uniform(25.0, 120.0, nclusters) * span / 3260.0
    band_strength = 10.0 ** rs.uniform(-4.0, 1.0, nclusters)
    band_strength[rs.randint(nclusters)] = 30.0                     # one fundamental that saturates the column
    which = rs.randint(0, nclusters, nlines)
    # two-sided exponential envelope: most lines near the centre, a long tail of weak ones far out in the wings
    offset = rs.laplace(0.0, 1.0, nlines) * band_width[which]
    centre = np.clip(band_centre[which] + offset, lo, hi)
    envelope = np.exp(-np.abs(offset) / band_width[which])
    strength = band_strength[which] * envelope * 10.0 ** rs.normal(0.0, 0.8, nlines) * (nclusters * 40.0 / nlines)
    gamma0 = rs.uniform(0.04, 0.12, nlines) * span / 3260.0
    order = np.argsort(centre, kind="stable")
    return centre[order], strength[order], gamma0[order]


def optical_depth_lines(xp, pressure_hl, wavenumber, seed, nlines=12000, column_scale=30.0, zero_fraction=0.05,
                        continuum=1.0e-7, cutoff=10.0, dtype="float32", device=None, chunk=1 << 13, lo=None, hi=None,
                        nclusters=7):
    """(nlay, nwav) layer optical depths of one synthetic gas with a CKDMIP-like line list (>= 1e4 lines,
    band_line_parameters): Lorentz lines, pressure-broadened half-width gamma0 * p/p_s, cut off `cutoff` cm-1 from the
    line centre (CKDMIP cuts at 25 cm-1) with the value at the cut subtracted so that lines end continuously, over a
    weak continuum; ~zero_fraction of the columns exactly zero (the tie groups of reorder_spectrum.cpp:187-190).
    Work per chunk of wavenumbers is limited to the lines within reach: cost ~ nlay * nwav * nlines * 2 cutoff / (hi - lo)
    evaluations (1e4 lines over 7.2e6 points: seconds on the device; keep nwav <= 1e6 on the host)."""
    p = np.asarray(pressure_hl, dtype=np.float64)
    nlay = p.size - 1
    is_torch = xp.__name__ == "torch"
    if is_torch:
        wn_host = wavenumber.detach().cpu().numpy() if hasattr(wavenumber, "detach") else np.asarray(wavenumber)
    else:
        wn_host = np.asarray(wavenumber)
    nwav = wn_host.size
    lo = float(wn_host[0]) if lo is None else lo
    hi = float(wn_host[-1]) if hi is None else hi
    cutoff = cutoff * (hi - lo) / 3260.0
    centre, strength, gamma0 = band_line_parameters(seed, nlines, lo, hi, nclusters)
    ps = p[-1]
    dp = (p[1:] - p[:-1]) / ps
    pfl = np.maximum(0.5 * (p[1:] + p[:-1]) / ps, 1.0e-4)
    rs = np.random.RandomState(seed + 7919)
    zero_cols = rs.uniform(size=nwav) < zero_fraction
    # a chunk about as wide as the cut-off keeps the lines per chunk near the minimum (those within reach of its points);
    # on the device the (nlay, lines, points) temporaries are held under ~1 GB
    per_cm = nwav / max(hi - lo, 1e-30)
    chunk = max(256, min(chunk, int(2.0 * cutoff * per_cm)))
    if is_torch:
        reach = max(1.0, nlines * (chunk / per_cm + 2.0 * cutoff) / max(hi - lo, 1e-30))
        while chunk > 256 and nlay * reach * chunk * 8 > 1.0e9:
            chunk //= 2
            reach = max(1.0, nlines * (chunk / per_cm + 2.0 * cutoff) / max(hi - lo, 1e-30))
        kw = dict(dtype=xp.float64, device=device)
        out = xp.empty((nlay, nwav), dtype=getattr(xp, dtype), device=device)
        pfl_t = xp.as_tensor(pfl, **kw)[:, None, None]
        scale_t = xp.as_tensor(dp * column_scale, **kw)[:, None]
    else:
        out = np.empty((nlay, nwav), dtype=dtype)
    for j0 in range(0, nwav, chunk):
        j1 = min(nwav, j0 + chunk)
        k0 = np.searchsorted(centre, wn_host[j0] - cutoff, "left")
        k1 = np.searchsorted(centre, wn_host[j1 - 1] + cutoff, "right")
        c, st, g0 = centre[k0:k1], strength[k0:k1], gamma0[k0:k1]
        if is_torch:
            wn_c = xp.as_tensor(wn_host[j0:j1], **kw)
            if k1 > k0:
                d2 = (wn_c[None, :] - xp.as_tensor(c, **kw)[:, None]) ** 2              # (lines, points)
                inside = d2 <= cutoff * cutoff
                g = xp.as_tensor(g0, **kw)[None, :, None] * pfl_t + 2.0e-4               # (nlay, lines, 1)
                s_t = xp.as_tensor(st, **kw)[None, :, None]
                shape = g / (np.pi * (d2[None] + g * g)) - g / (np.pi * (cutoff * cutoff + g * g))
                line = (s_t * xp.where(inside[None], shape, xp.zeros_like(shape))).sum(1)   # (nlay, points)
            else:
                line = xp.zeros((nlay, j1 - j0), **kw)
            rows = scale_t * (continuum + line)
            rows = xp.where(xp.as_tensor(zero_cols[j0:j1], device=device)[None, :], xp.zeros_like(rows), rows)
            out[:, j0:j1] = rows.to(out.dtype)
        else:
            d2 = (wn_host[None, j0:j1] - c[:, None]) ** 2
            inside = d2 <= cutoff * cutoff
            for l in range(nlay):
                g = (g0 * pfl[l] + 2.0e-4)[:, None]
                shape = g / (np.pi * (d2 + g * g)) - g / (np.pi * (cutoff * cutoff + g * g))
                line = (st[:, None] * np.where(inside, shape, 0.0)).sum(0)          # zeros where no line is within reach
                row = dp[l] * column_scale * (continuum + line)
                row[zero_cols[j0:j1]] = 0.0
                out[l, j0:j1] = row.astype(dtype)
    return out


def solar_spectral_irradiance(wavenumber, d_wavenumber, tsi=1361.0, t_sun=5777.0):
    """5777 K Planck spectrum per interval, scaled to total `tsi` W m-2."""
    wn = np.asarray(wavenumber, dtype=np.float64)
    x = 1.438776877 * wn / t_sun
    b = wn ** 3 / np.expm1(np.maximum(x, 1e-12)) * np.asarray(d_wavenumber)
    return b * (tsi / b.sum())


# reference test/config.h:141-150 band definitions (cm-1) used by the parity cases
LW_NARROW_BANDS = (
    np.array([0, 350, 500, 630, 700, 820, 980, 1080, 1180, 1390, 1480, 1800, 2080], dtype=np.float64),
    np.array([350, 500, 630, 700, 820, 980, 1080, 1180, 1390, 1480, 1800, 2080, 3260], dtype=np.float64),
)
LW_FSCK_BAND = (np.array([0.0]), np.array([3260.0]))


# ---------------------------------------------------------------------------------------
# Synthetic CKD model + LBL-like training scenes for optimize_lut (SURVEY.md 8d, config 5 shapes:
# ng = 64, nt = 6, np = 53, H2O look-up table with 12 mole fractions, 50 columns per scenario).

def ckd_model(ng=12, nt=5, np_=16, nband=3, seed=0, nconc=4):
    rs = np.random.RandomState(seed)
    log_p = np.linspace(np.log(2.0), np.log(110000.0), np_)
    t_mid = 200.0 + 12.0 * (log_p - log_p[0])               # p-dependent temperature grid
    temperature = t_mid[None, :] + 20.0 * (np.arange(nt)[:, None] - (nt - 1) / 2)
    tpl = np.arange(120.0, 351.0, 1.0)
    # Planck function per g point: a few "spectral" bins of sigma T^4 shape
    centre = np.linspace(300.0, 1500.0, ng)
    x = 1.4388 * centre[None, :] / tpl[:, None]
    planck = 8.0 * centre[None, :] ** 3 / np.expm1(x) * 1e-7
    iband = np.sort(rs.randint(0, nband, ng)).astype(np.int32)
    iband[0], iband[-1] = 0, nband - 1
    for b in range(nband):                                  # every band populated
        if not (iband == b).any():
            iband[b] = b
    iband = np.sort(iband)
    g_strength = 10.0 ** np.linspace(-3.0, 1.5, ng)         # weak ... strong g points

    def lut(nconc=None, scale=1.0):
        shape = (nt, np_, ng) if nconc is None else (nconc, nt, np_, ng)
        base = scale * g_strength * (1.0 + 0.3 * rs.uniform(size=shape))
        pfac = np.exp(0.4 * (log_p - log_p[-1]))[:, None]   # pressure broadening-ish
        base = base * (0.2 + pfac) if nconc is None else base * (0.2 + pfac)[None, None]
        return base

    vmr_h2o = np.exp(np.linspace(np.log(1e-6), np.log(4e-2), nconc))
    gases = [
        dict(name="composite", conc="none", active=True, molar_abs=lut(scale=2e-4)),
        dict(name="h2o", conc="lut", active=True, molar_abs=lut(nconc, scale=3.0), vmr=vmr_h2o),
        dict(name="co2", conc="linear", active=True, molar_abs=lut(scale=30.0)),
        dict(name="ch4", conc="relative-linear", active=True, molar_abs=lut(scale=50.0), reference_vmr=1.8e-6),
        dict(name="o3", conc="linear", active=False, molar_abs=lut(scale=500.0)),
    ]
    gases[1]["molar_abs"][:, :, :, 0] = 0.0                 # a g point where h2o does not absorb: x pinned at MIN_X
    for g in gases:
        g["min_molar_abs"] = g["molar_abs"] * 0.5
        g["max_molar_abs"] = g["molar_abs"] * 2.0
    gases[2]["min_molar_abs"] = gases[2]["min_molar_abs"].copy()
    gases[2]["min_molar_abs"][:, :, 3] = 0.0                # exercises the k_min == 0 bound rule
    return dict(log_pressure=log_p, temperature=temperature, temperature_planck=tpl, planck_function=planck,
                iband_per_g=iband, gases=gases, nband=nband)


def ckd_scenes(model, nscene=2, ncol=4, nlay=18, seed=1, ch4_low=False):
    rs = np.random.RandomState(seed)
    ngas = len(model["gases"])
    scenes = []
    for s in range(nscene):
        p = np.empty((ncol, nlay + 1))
        T = np.empty((ncol, nlay + 1))
        vmr = np.empty((ncol, ngas, nlay))
        for c in range(ncol):
            p[c] = np.concatenate([[1.0], np.exp(np.linspace(np.log(5.0), np.log(101325.0 - 2000 * c), nlay))])
            T[c] = 210.0 + 80.0 * (p[c] / p[c, -1]) ** 0.25 + rs.uniform(-3, 3, nlay + 1) + 4.0 * s
            pf = 0.5 * (p[c, 1:] + p[c, :-1]) / p[c, -1]
            vmr[c, 0] = 1.0
            vmr[c, 1] = np.clip(2e-2 * pf ** 3 * (1 + 0.5 * s) * rs.uniform(0.5, 1.5), 2e-6, 5e-2)
            vmr[c, 2] = 4e-4 * (1.0 + s)
            vmr[c, 3] = (0.9e-6 if ch4_low else 1.8e-6 * (1.0 + 0.5 * s)) * np.ones(nlay)
            vmr[c, 4] = 5e-6 * np.exp(-((np.log(pf) + 5.0) / 1.5) ** 2) + 2e-8
        present = np.ones(ngas, dtype=np.int32)
        if s == 1:
            present[4] = 0                                  # o3 missing from the second training file
        scenes.append(dict(pressure_hl=p, temperature_hl=T, vmr_fl=vmr, gas_present=present))
    return scenes


