"""Timing of the kernels DESIGN.md section 4 had no measurement for: K2 (shortwave sorting key), K6 (g-point averaging of one
spectrum column), K7 (Planck look-up table, g-point fractions), at BASELINE sizes, HIP events on the library's stream.
Prints one JSON object.  Usage: python tools/k267_probe.py"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecckd_amd import api, synthetic as syn

out = {}
with api.Context(0) as ctx:
    dev = ctx.device
    nlay = 54
    p = syn.pressure_grid(nlay)
    # ---- K2: nwav = 3.3e6 over 250-50000 cm-1 (configs[2]) ----
    nsw = 3_300_000
    wn_s, dwn_s = syn.wavenumber_grid(nsw, 250.0, 50000.0)
    od_s = syn.optical_depth_lines(torch, p, wn_s, syn.SEED_BASE + 3, column_scale=5.0, device=dev, lo=250.0, hi=50000.0)
    key = torch.empty(nsw, dtype=torch.float64, device=dev); col = torch.empty_like(key)
    api.reorder_key_sw(ctx, p, od_s, 0.25, key=key, col_od=col)
    ts = []
    for _ in range(5):
        ctx.timer_begin(); api.reorder_key_sw(ctx, p, od_s, 0.25, key=key, col_od=col); ts.append(ctx.timer_end())
    b = nlay * 4 + 16
    out["K2 k_reorder_key_sw"] = {"nwav": nsw, "ms": min(ts), "algorithmic_bytes_per_point": b, "GBs": nsw * b / (min(ts) * 1e-3) / 1e9,
                                  "frac_of_8TBs": nsw * b / (min(ts) * 1e-3) / 8e12}
    del od_s
    # ---- K6 / K7: nwav = 7.2e6, ng = 38 g points from a sorted-key quantile map ----
    nwav, ng = 7_200_000, 38
    wn_h, dwn_h = syn.wavenumber_grid(nwav)
    wn, dwn = torch.as_tensor(wn_h, device=dev), torch.as_tensor(dwn_h, device=dev)
    od = syn.optical_depth_lines(torch, p, wn, syn.SEED_BASE + 1, device=dev)
    k, _ = api.reorder_key_lw(ctx, p, api.idealised_temperature(p), wn, dwn, od, 0.5)
    rank, _ = api.stable_argsort_bands(ctx, k, [0], [nwav - 1], want_ordered=False)
    edges = (nwav * (np.linspace(0.0, 1.0, ng + 1) ** 0.35)).astype(np.int64)      # many points in the weak g points, few in the strong
    edges[-1] = nwav
    g_point = torch.bucketize(rank.long(), torch.as_tensor(edges[1:-1], device=dev), right=True).to(torch.int32)
    t_hl = syn.temperature_profile(p)
    t_fl = (t_hl[:-1] * p[:-1] + t_hl[1:] * p[1:]) / (p[:-1] + p[1:])
    ctx.timer_begin(); gm = api.GPointMap(ctx, g_point, ng, wn, dwn); t_create = ctx.timer_end()
    gm.average_optical_depth(p, od, "transmission", reference_surface_vmr=1e-3, temperature_fl=t_fl)
    ts = []
    for _ in range(5):
        ctx.timer_begin(); gm.average_optical_depth(p, od, "transmission", reference_surface_vmr=1e-3, temperature_fl=t_fl); ts.append(ctx.timer_end())
    b6 = nlay * 4 + 16 + 4
    out["K6 average_to_gpoints (one column)"] = {"nwav": nwav, "ng": ng, "ms": min(ts), "gmap_create_ms": t_create, "algorithmic_bytes_per_point": b6,
                                                 "GBs": nwav * b6 / (min(ts) * 1e-3) / 1e9, "frac_of_8TBs": nwav * b6 / (min(ts) * 1e-3) / 8e12}
    tl = np.arange(120.0, 351.0)
    gm.planck_lut(tl)
    ts = []
    for _ in range(3):
        ctx.timer_begin(); gm.planck_lut(tl); ts.append(ctx.timer_end())
    out["K7 planck_lut (231 temperatures)"] = {"nwav": nwav, "ms": min(ts), "planck_evaluations": 231 * nwav,
                                               "evaluations_per_s": 231 * nwav / (min(ts) * 1e-3)}
    w1 = 10.0 * np.arange(0, 326, dtype=np.float64); w2 = w1 + 10.0
    gm.gpoint_fraction(w1, w2)
    ts = []
    for _ in range(3):
        ctx.timer_begin(); gm.gpoint_fraction(w1, w2); ts.append(ctx.timer_end())
    out["K7 gpoint_fraction (326 intervals)"] = {"nwav": nwav, "ms": min(ts), "algorithmic_bytes_per_point": 20, "GBs": nwav * 20 / (min(ts) * 1e-3) / 1e9}
    gm.close()
print(json.dumps(out, indent=1))
