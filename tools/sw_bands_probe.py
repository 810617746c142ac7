"""Timing probe: the shortwave band searches of one gas over 32 bands (BASELINE configs[2]'s shape: nwav = 3.3e6 over
250-50000 cm-1, total-transmission, reference albedo 0.15 below 10 000 cm-1), one band after the other as the reference's
band loop does, against the same searches side by side (ecckd_find_g_bands_ex)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecckd_amd import api, synthetic as syn

nwav, nlay, nband = int(sys.argv[1]) if len(sys.argv) > 1 else 3300000, 54, 32
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
ctx = api.Context(0)
dev = ctx.device
p = syn.pressure_grid(nlay)
wn_h, dwn_h = syn.wavenumber_grid(nwav, 250.0, 50000.0)
kw = dict(device=dev, lo=250.0, hi=50000.0)
od = syn.optical_depth(torch, p, wn_h, syn.SEED_BASE + 3, nlines=96, column_scale=5.0, **kw)
bg = syn.optical_depth(torch, p, wn_h, syn.SEED_BASE + 1003, nlines=24, column_scale=0.5, zero_fraction=0.0, **kw)
ssi = torch.as_tensor(syn.solar_spectral_irradiance(wn_h, dwn_h), device=dev)
edges = np.geomspace(250.0, 50000.0, nband + 1)                      # 32 bands of equal width in log wavenumber
b1, b2 = edges[:-1], edges[1:].copy()
b2[-1] = 50001.0
iband, begin, end = api.band_ranges(wn_h, b1, b2)
band_albedo = np.where(b2 <= 10000.0, 0.15, 0.0)
alb = torch.as_tensor(np.where(wn_h < b2[band_albedo > 0].max(), 0.15, 0.0), device=dev)
key, col = api.reorder_key_sw(ctx, p, od, 0.25)
rank, _ = api.stable_argsort_bands(ctx, key, begin, end, want_ordered=False)
gas = api.GasSW(ctx, p, ssi, rank, od, bg, "total-transmission", flux_weight=0.02, albedo=alb)
for rep in range(2):
    ctx.synchronize(); t0 = time.perf_counter()
    one = []
    for k in range(nband):
        gas.set_band_albedo(float(band_albedo[k]))
        one.append(gas.find_g_band_ex(int(begin[k]), int(end[k]), tol, 0.02, 60))
    ctx.synchronize(); dt1 = time.perf_counter() - t0
for rep in range(2):
    ctx.synchronize(); t0 = time.perf_counter()
    res = gas.find_g_bands_ex(begin, end, tol, 0.02, 60, options=[dict(band_albedo=float(a)) for a in band_albedo])
    ctx.synchronize(); dt2 = time.perf_counter() - t0
gas.close()
same = all(np.array_equal(a["rank1"], b["rank1"]) and a["status"] == b["status"] for a, b in zip(one, res))
passes = lambda rs: sum(r["comp_cost"] * (int(e) - int(b) + 1) / nwav for r, b, e in zip(rs, begin, end))
print(f"{nband} bands one after the other: {1e3 * dt1:.1f} ms, ng={sum(len(r['error']) for r in one)}, passes={passes(one):.1f}, {nwav * passes(one) / dt1:.3e} points/s")
print(f"{nband} bands side by side:        {1e3 * dt2:.1f} ms, ng={sum(len(r['error']) for r in res)}, passes={passes(res):.1f}, {nwav * passes(res) / dt2:.3e} points/s; same g points: {same}")
