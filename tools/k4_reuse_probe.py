import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from ecckd_amd import api, synthetic as syn
ctx = api.Context(0)
nwav, nlay = 7_200_000, 54
p = syn.pressure_grid(nlay); t = syn.temperature_profile(p)
wn, dwn = syn.wavenumber_grid(nwav)
dev = lambda a: torch.as_tensor(a, device=ctx.device)
od = syn.optical_depth_lines(torch, p, wn, syn.SEED_BASE + 1, nlines=3000, column_scale=30.0, device=ctx.device)
bg = syn.optical_depth_lines(torch, p, wn, syn.SEED_BASE + 2, nlines=3000, column_scale=8.0, device=ctx.device)
wnd, dwnd = dev(wn), dev(dwn)
key = torch.empty(nwav, dtype=torch.float64, device=ctx.device); col = torch.empty_like(key)
api.reorder_key_lw(ctx, p, t, wnd, dwnd, od, 0.5, key=key, col_od=col)
rnk = torch.empty(nwav, dtype=torch.int32, device=ctx.device)
api.stable_argsort_bands(ctx, key, [0], [nwav - 1], rank=rnk, want_ordered=False, sync=True)
for label, reuse_from in (("own Planck matrix", None), ("reused Planck matrix", "first")):
    first = api.GasLW(ctx, p, t, wnd, dwnd, rnk, od, bg, "transmission", flux_weight=0.0)
    ts = []
    for rep in range(3):
        ctx.synchronize(); t0 = time.perf_counter()
        g = api.GasLW(ctx, p, t, wnd, dwnd, rnk, od, bg, "transmission", flux_weight=0.0,
                      planck_hl_reuse=first.view_ptr("planck_hl")[0] if reuse_from else None)
        ctx.synchronize(); ts.append(time.perf_counter() - t0)
        g.close()
    first.close()
    print(f"{label}: whole preparation (2 scatters + K4 + sums) {min(ts) * 1e3:.2f} ms")
