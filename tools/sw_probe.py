"""Timing probe for the shortwave find_g_points path (BASELINE configs[2] shapes): nwav = 3.3e6, nlay = 54,
one band of the 32-band structure cannot be timed alone, so the whole 250-50000 cm-1 range is one band here."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from ecckd_amd import api, synthetic as syn

nwav, nlay = int(sys.argv[1]) if len(sys.argv) > 1 else 3300000, 54
method = sys.argv[2] if len(sys.argv) > 2 else "total-transmission"
min_scaling, max_scaling = (float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (1.0, 1.0)
ctx = api.Context(0)
dev = ctx.device
p = syn.pressure_grid(nlay)
wn_h, dwn_h = syn.wavenumber_grid(nwav, 250.0, 50000.0)
kw = dict(device=dev, lo=250.0, hi=50000.0)
od = syn.optical_depth(torch, p, wn_h, syn.SEED_BASE + 3, nlines=96, column_scale=5.0, **kw)
bg = syn.optical_depth(torch, p, wn_h, syn.SEED_BASE + 1003, nlines=24, column_scale=0.5, zero_fraction=0.0, **kw)
band_albedo = float(sys.argv[5]) if len(sys.argv) > 5 else 0.15        # find_g_points.cpp:757-761: 0.15 below 10000 cm-1, else 0
alb = torch.full((nwav,), band_albedo, dtype=torch.float64, device=dev)
ssi = torch.as_tensor(syn.solar_spectral_irradiance(wn_h, dwn_h), device=dev)
key, col = api.reorder_key_sw(ctx, p, od, 0.25)
rnk = torch.empty(nwav, dtype=torch.int32, device=dev)
api.stable_argsort_bands(ctx, key, [0], [nwav - 1], rank=rnk, want_ordered=False, sync=True)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    key, col = api.reorder_key_sw(ctx, p, od, 0.25, key=key, col_od=col)
    api.stable_argsort_bands(ctx, key, [0], [nwav - 1], rank=rnk, want_ordered=False, sync=False)
    gas = api.GasSW(ctx, p, ssi, rnk, od, bg, method, flux_weight=0.02, min_scaling=min_scaling, max_scaling=max_scaling, albedo=alb)
    gas.set_band_albedo(band_albedo)
    ctx.synchronize(); t1 = time.perf_counter()
    st, b, e, cc = gas.find_g_band(0, nwav - 1, 0.02, 0.02, 60)
    ctx.synchronize(); t2 = time.perf_counter()
    gas.close()
    print(f"iter {it}: prep {1e3*(t1-t0):.1f} ms, search {1e3*(t2-t1):.1f} ms, ng={len(e)}, N_pass={cc:.1f}, "
          f"{nwav*(1+cc)/(t2-t0):.3e} points/s")
