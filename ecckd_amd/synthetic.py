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


