import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
os.environ["ECCKD_NO_ERROR_MEMO"] = "1"
from ecckd_amd import api, synthetic as syn
ctx = api.Context(0); dev = ctx.device
nlay, nwav = 54, 3_300_000
p = syn.pressure_grid(nlay)
wn_h, dwn_h = syn.wavenumber_grid(nwav, 250.0, 50000.0)
kw = dict(device=dev, lo=250.0, hi=50000.0)
od = syn.optical_depth(torch, p, wn_h, syn.SEED_BASE + 3, nlines=96, column_scale=5.0, **kw)
bg = syn.optical_depth(torch, p, wn_h, syn.SEED_BASE + 1003, nlines=24, column_scale=0.5, zero_fraction=0.0, **kw)
ssi = torch.as_tensor(syn.solar_spectral_irradiance(wn_h, dwn_h), device=dev)
alb = torch.full((nwav,), 0.15, dtype=torch.float64, device=dev)
key, col = api.reorder_key_sw(ctx, p, od, 0.25)
rank, _ = api.stable_argsort_bands(ctx, key, [0], [nwav - 1], want_ordered=False)
gas = api.GasSW(ctx, p, ssi, rank, od, bg, "total-transmission", flux_weight=0.02, albedo=alb)
gas.set_band_albedo(0.15)
b1 = np.array([0.0, 0.0, 0.37, 0.62, 0.9, 0.999, 0.5])
b2 = np.array([1.0, 0.37, 0.62, 0.9, 0.999, 1.0, 0.5000004])
np.set_printoptions(precision=6, linewidth=200)
print("fast    ", gas.calc_error_batch(0, nwav, b1, b2))
os.environ["ECCKD_RT_GENERIC"] = "1"
print("generic ", gas.calc_error_batch(0, nwav, b1, b2))
print("generic one by one", np.array([gas.calc_error_batch(0, nwav, b1[k:k+1], b2[k:k+1])[0] for k in range(7)]))
del os.environ["ECCKD_RT_GENERIC"]
print("fast one by one", np.array([gas.calc_error_batch(0, nwav, b1[k:k+1], b2[k:k+1])[0] for k in range(7)]))
os.environ["ECCKD_SW_NO_SAME"] = "1"
