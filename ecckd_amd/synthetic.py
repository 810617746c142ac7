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
