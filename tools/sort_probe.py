#!/usr/bin/env python3
"""K3 (radix sort) timing: nwav keys of the headline generator's sorting variable, one band and the 13 narrow bands; the two
scatter kernels (ECCKD_SORT_DIRECT=1: every lane stores to its final position) must give the same ranks.
    python tools/sort_probe.py [--nwav 7200000] [--reps 20]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nwav", type=int, default=7_200_000)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch
    from ecckd_amd import api, synthetic as syn
    ctx = api.Context(0)
    dev = ctx.device
    n = args.nwav
    p = syn.pressure_grid(54)
    wn_h, dwn_h = syn.wavenumber_grid(n)
    wn, dwn = torch.as_tensor(wn_h, device=dev), torch.as_tensor(dwn_h, device=dev)
    od = syn.optical_depth_lines(torch, p, wn, syn.SEED_BASE + 1, nlines=12000, device=dev)
    t_ideal = api.idealised_temperature(p)
    key, col = api.reorder_key_lw(ctx, p, t_ideal, wn, dwn, od, 0.5)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        api.reorder_key_lw(ctx, p, t_ideal, wn, dwn, od, 0.5, key=key, col_od=col)
    ctx.synchronize()
    k1_ms = (time.perf_counter() - t0) / args.reps * 1e3
    key_checksum = float(key.double().sum().item())
    del od
    b1, b2 = syn.LW_NARROW_BANDS
    _, begin, end = api.band_ranges(wn_h, b1, b2)
    out = {"nwav": n, "k1_key_lw_ms_per_call": k1_ms, "k1_key_sum": key_checksum}
    rank = torch.empty(n, dtype=torch.int32, device=dev)
    for name, (bb, be) in (("one_band", ([0], [n - 1])), ("thirteen_bands", (begin, end))):
        api.stable_argsort_bands(ctx, key, bb, be, rank=rank, want_ordered=False, sync=True)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            api.stable_argsort_bands(ctx, key, bb, be, rank=rank, want_ordered=False, sync=False)
        ctx.synchronize()
        out[name + "_ms"] = (time.perf_counter() - t0) / args.reps * 1e3
        out[name + "_rank_checksum"] = int((rank.long() * torch.arange(n, device=dev) % 1000003).sum().item())
    out["mode"] = "direct" if os.environ.get("ECCKD_SORT_DIRECT") else "lds"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
