"""What K6a could reach with coalesced loads: the same kernel on a g-point map whose g points are CONTIGUOUS runs of wavenumbers
(the g-sorted permutation is then the identity: every 4-byte gather of a wave falls into 256 consecutive bytes), beside the
headline-like map (g points interleaved along the wavenumber axis).  Upper bound for a natural-order K6."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ecckd_amd import api, synthetic as syn

out = {}
with api.Context(0) as ctx:
    dev = ctx.device
    nlay, nwav, ng = 54, 7_200_000, 38
    p = syn.pressure_grid(nlay)
    wn_h, dwn_h = syn.wavenumber_grid(nwav)
    wn, dwn = torch.as_tensor(wn_h, device=dev), torch.as_tensor(dwn_h, device=dev)
    od = syn.optical_depth_lines(torch, p, wn, syn.SEED_BASE + 1, device=dev)
    t_hl = syn.temperature_profile(p)
    t_fl = (t_hl[:-1] * p[:-1] + t_hl[1:] * p[1:]) / (p[:-1] + p[1:])
    edges = (nwav * (np.linspace(0.0, 1.0, ng + 1) ** 0.35)).astype(np.int64)
    edges[-1] = nwav
    k, _ = api.reorder_key_lw(ctx, p, api.idealised_temperature(p), wn, dwn, od, 0.5)
    rank, _ = api.stable_argsort_bands(ctx, k, [0], [nwav - 1], want_ordered=False)
    maps = {"interleaved (by sorting key)": torch.bucketize(rank.long(), torch.as_tensor(edges[1:-1], device=dev), right=True).to(torch.int32),
            "contiguous runs": torch.bucketize(torch.arange(nwav, device=dev), torch.as_tensor(edges[1:-1], device=dev), right=True).to(torch.int32)}
    for name, g_point in maps.items():
        gm = api.GPointMap(ctx, g_point, ng, wn, dwn)
        gm.average_optical_depth(p, od, "transmission", reference_surface_vmr=1e-3, temperature_fl=t_fl)
        ts = []
        for _ in range(5):
            ctx.timer_begin(); gm.average_optical_depth(p, od, "transmission", reference_surface_vmr=1e-3, temperature_fl=t_fl); ts.append(ctx.timer_end())
        out[name] = {"ms_per_column": min(ts)}
        gm.close()
print(json.dumps(out))
