"""Timing probe: the longwave band searches of one gas over the 13 narrow bands (BASELINE configs[3] shapes), one band
after the other as find_g_points does, against the single-band (FSCK) search of the bench."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from ecckd_amd import api, synthetic as syn

nwav, nlay = int(sys.argv[1]) if len(sys.argv) > 1 else 7200000, 54
ctx = api.Context(0)
dev = ctx.device
p = syn.pressure_grid(nlay)
wn_h, dwn_h = syn.wavenumber_grid(nwav)
wn, dwn = torch.as_tensor(wn_h, device=dev), torch.as_tensor(dwn_h, device=dev)
od = syn.optical_depth(torch, p, wn, syn.SEED_BASE + 1, nlines=32, device=dev, chunk=1 << 20)
bg = syn.optical_depth(torch, p, wn, syn.SEED_BASE + 1001, nlines=24, column_scale=3.0, zero_fraction=0.0, device=dev, chunk=1 << 20)
b1, b2 = syn.LW_NARROW_BANDS
iband, begin, end = api.band_ranges(wn_h, b1, b2)
key, col = api.reorder_key_lw(ctx, p, api.idealised_temperature(p), wn, dwn, od, 0.5)
for label, (bb, ee) in (("1 band", ([0], [nwav - 1])), ("%d bands" % len(begin), (begin, end))):
    rank, _ = api.stable_argsort_bands(ctx, key, bb, ee, want_ordered=False)
    gas = api.GasLW(ctx, p, syn.temperature_profile(p), wn, dwn, rank, od, bg, "transmission", flux_weight=0.0)
    for rep in range(2):
        ctx.synchronize(); t0 = time.perf_counter()
        cc_tot, ng_tot = 0.0, 0
        for i0, i1 in zip(bb, ee):
            st, b, e, cc = gas.find_g_band(int(i0), int(i1), 0.0161 * (1 if len(bb) == 1 else float(sys.argv[2]) if len(sys.argv) > 2 else 1), 0.01, 60)
            cc_tot += cc * (int(i1) - int(i0) + 1) / nwav
            ng_tot += len(e)
        ctx.synchronize(); dt = time.perf_counter() - t0
    print(f"{label}: search {1e3 * dt:.1f} ms, ng={ng_tot}, passes over the spectrum={cc_tot:.1f}, {nwav * cc_tot / dt:.3e} points/s")
    if len(bb) > 1:
        tol = 0.0161 * (float(sys.argv[2]) if len(sys.argv) > 2 else 1)
        for rep in range(2):
            ctx.synchronize(); t0 = time.perf_counter()
            res = gas.find_g_bands_ex(bb, ee, tol, 0.01, 60)
            ctx.synchronize(); dt = time.perf_counter() - t0
        cc_tot = sum(r["comp_cost"] * (int(i1) - int(i0) + 1) / nwav for r, i0, i1 in zip(res, bb, ee))
        print(f"{label} side by side: search {1e3 * dt:.1f} ms, ng={sum(len(r['error']) for r in res)}, passes={cc_tot:.1f}, "
              f"{nwav * cc_tot / dt:.3e} points/s")
    gas.close()
