"""Timing probe for the longwave sweep kernel alone: a fixed whole-partition batch (38 intervals over the 7.2e6 points) and a
few single intervals, timed by the library's HIP events around every launch (ECCKD_BG64=1: with DOUBLE background rows;
ECCKD_RT_PERSISTENT=1: one resident round of blocks walking the chunks).
usage: python tools/sweep_probe.py [nwav]"""
import os, sys, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecckd_amd import api, synthetic as syn

os.environ["ECCKD_NO_ERROR_MEMO"] = "1"
nwav, nlay = int(sys.argv[1]) if len(sys.argv) > 1 else 7200000, 54
ctx = api.Context(0)
dev = ctx.device
p = syn.pressure_grid(nlay)
wn_h, dwn_h = syn.wavenumber_grid(nwav)
wn, dwn = torch.as_tensor(wn_h, device=dev), torch.as_tensor(dwn_h, device=dev)
od = syn.optical_depth_lines(torch, p, wn, syn.SEED_BASE + 1, device=dev)
bg = syn.optical_depth_lines(torch, p, wn, syn.SEED_BASE + 1001, column_scale=3.0, device=dev)
key, col = api.reorder_key_lw(ctx, p, api.idealised_temperature(p), wn, dwn, od, 0.5)
rank, _ = api.stable_argsort_bands(ctx, key, [0], [nwav - 1], want_ordered=False)
gas = api.GasLW(ctx, p, syn.temperature_profile(p), wn, dwn, rank, od, bg, "transmission", flux_weight=0.0)
bpp = gas.sweep_bytes_per_point()
cuts = np.linspace(0.0, 1.0, 39) ** 1.5
out = {"bytes_per_point": bpp}
even = np.linspace(0.0, 1.0, 39)
for label, (b1, b2) in (("whole partition, 38 intervals", (cuts[:-1], cuts[1:])), ("whole partition, 38 equal intervals", (even[:-1], even[1:])),
                        ("whole partition, 1 interval", ([0.0], [1.0])), ("whole partition, 4 equal intervals", ([0.0, 0.25, 0.5, 0.75], [0.25, 0.5, 0.75, 1.0])),
                        ("one interval of 190 000 points", ([0.4], [0.4 + 190000.0 / nwav])),
                        ("one interval of 40 000 points", ([0.4], [0.4 + 40000.0 / nwav])),
                        ("one interval of 1 000 000 points", ([0.4], [0.4 + 1000000.0 / nwav]))):
    for _ in range(3):
        gas.calc_error_batch(0, nwav, b1, b2)
    ctx.profile_enable(1)
    for _ in range(20):
        gas.calc_error_batch(0, nwav, b1, b2)
    calls, ms, pts = ctx.profile_get("k_rt_lw_bb")
    ctx.profile_enable(0)
    out[label] = {"launches": calls, "us_per_launch": 1e3 * ms / calls, "points_per_launch": pts / calls, "TBs": pts * bpp / (ms * 1e-3) / 1e12}
print(json.dumps(out))
