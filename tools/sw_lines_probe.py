import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from ecckd_amd import api, synthetic as syn
ctx = api.Context(0); dev = ctx.device
nwav, nlay = 3_300_000, 54
p = syn.pressure_grid(nlay)
wn_h, dwn_h = syn.wavenumber_grid(nwav, 250.0, 50000.0)
wn = torch.as_tensor(wn_h, device=dev)
od = syn.optical_depth_lines(torch, p, wn, syn.SEED_BASE + 501, column_scale=5.0, device=dev, lo=250.0, hi=50000.0)
bg = syn.optical_depth_lines(torch, p, wn, syn.SEED_BASE + 1501, nlines=4000, column_scale=0.5, zero_fraction=0.0, nclusters=5, device=dev, lo=250.0, hi=50000.0)
ssi = torch.as_tensor(syn.solar_spectral_irradiance(wn_h, dwn_h), device=dev)
alb = torch.full((nwav,), 0.15, dtype=torch.float64, device=dev)
for tol in (0.2, 0.1, 0.047, 0.03, 0.019):
    for rep in range(2):
        ctx.synchronize(); t0 = time.perf_counter()
        key, col = api.reorder_key_sw(ctx, p, od, 0.25)
        rnk, _ = api.stable_argsort_bands(ctx, key, [0], [nwav - 1], want_ordered=False)
        gas = api.GasSW(ctx, p, ssi, rnk, od, bg, "total-transmission", 0.02, 0.0, 0.5, alb, 0.5, 2.5)
        gas.set_band_albedo(0.15)
        st, b, e, cc = gas.find_g_band(0, nwav - 1, tol, 0.01, 60)
        ctx.synchronize(); dt = time.perf_counter() - t0
        es = gas.eval_stats(); gas.close()
    print(f"tol {tol}: status {st} ng {len(e)} passes(ref counter) {cc:.1f} swept {es['points_evaluated']/nwav:.1f} ms {dt*1e3:.1f} max err {max(e):.4f}", flush=True)
