#!/usr/bin/env python3
"""bench.py - LW reorder + find_g_points hot path on synthetic CKDMIP-like spectra.

One "step" = one pass of the hot path over one device-resident column
(BASELINE.json configs[1]: LW FSCK, one band, nwav = 7.2e6, nlay = 54, FLOAT optical
depths): K1 sorting key + K3 stable sort (reorder_spectrum.cpp:111-300) and, once the
gas is prepared, the g-point partition search (find_g_points.cpp:1152-1266) with every
interval-error evaluation on the device.  Metric = wavenumber-points/s =
nwav * (1 + N_pass) / t, N_pass = sum of (bound2-bound1) over all calc_error calls
(find_g_points.cpp:320), summed over ranks.  Ranks process independent (gas, band)
shards (weak scaling, no data-path collective).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 measured)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nwav", type=int, default=7_200_000)
    ap.add_argument("--nlay", type=int, default=54)
    ap.add_argument("--tolerance", type=float, default=0.0161)  # fsck, test/do_all_lw.sh:59-60
    ap.add_argument("--cpu-sample", type=int, default=1 << 16)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-find-g", action="store_true")
    return ap.parse_args()


def cpu_baseline_reorder(nwav_sample, nlay, seed):
    """Oracle ("port") LW reorder on the host cores: K1 restatement + stable sort."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    from ecckd_amd import synthetic as syn
    p = syn.pressure_grid(nlay)
    wn, dwn = syn.wavenumber_grid(nwav_sample)
    od = syn.optical_depth(np, p, wn, seed, nlines=32).astype(np.float64)
    t = pyoracle.idealised_temperature(p)
    t0 = time.perf_counter()
    key, col, st = pyoracle.reorder_key(p, t, wn, dwn, od, None, 0.5)
    pyoracle.stable_argsort_bands(wn, key, np.array([0.0]), np.array([3260.0]))
    dt = time.perf_counter() - t0
    return nwav_sample, dt


def main():
    args = parse()
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from ecckd_amd import api, synthetic as syn

    ctx = api.Context(local_rank)
    dev = ctx.device
    nwav, nlay = args.nwav, args.nlay
    p = syn.pressure_grid(nlay)
    wn_h, dwn_h = syn.wavenumber_grid(nwav)
    wn = torch.as_tensor(wn_h, device=dev)
    dwn = torch.as_tensor(dwn_h, device=dev)
    # each rank owns a different synthetic gas (independent shard)
    od = syn.optical_depth(torch, p, wn, syn.SEED_BASE + 1 + rank, nlines=32, device=dev, chunk=1 << 20)
    t_hl = api.idealised_temperature(p)
    key = torch.empty(nwav, dtype=torch.float64, device=dev)
    col = torch.empty(nwav, dtype=torch.float64, device=dev)
    rnk = torch.empty(nwav, dtype=torch.int32, device=dev)
    oi = torch.empty(nwav, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    k1_ms = []

    def step(timed):
        ctx.timer_begin()
        api.reorder_key_lw(ctx, p, t_hl, wn, dwn, od, 0.5, key=key, col_od=col)
        ms = ctx.timer_end()
        if timed:
            k1_ms.append(ms)
        api.stable_argsort_bands(ctx, key, [0], [nwav - 1], rank=rnk, ordered_index=oi, sync=False)
        return 1.0  # passes over the spectrum in this step (1 for reorder)

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    passes = 0.0
    for _ in range(args.steps):
        passes += step(True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        pp = torch.tensor([passes], dtype=torch.float64, device=dev)
        dist.all_reduce(pp, op=dist.ReduceOp.SUM)
        passes = float(pp.item())

    if rank == 0:
        points = nwav * passes
        k1 = float(np.mean(k1_ms)) * 1e-3
        k1_bytes = nwav * (nlay * od.element_size() + 32)  # nlay*s + 32 B per point
        out = {
            "metric": "wavenumber-points/s (LW reorder+find_g)",
            "value": points / dt,
            "unit": "wavenumber-points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "LW FSCK 1 band, 1 synthetic gas/rank, nwav=%d, nlay=%d, od f32; reorder only"
                       % (nwav, nlay), "passes_per_step": passes / args.steps / world},
            "roofline": {"bound": "hbm", "kernel": "k_reorder_key_lw<float>", "achieved": k1_bytes / k1 / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k1_bytes / k1 / 1e9 / HBM_PEAK_GBS,
                         "traffic": None, "avg_ms": k1 * 1e3},
        }
        if world == 1 and not args.no_cpu:
            n_s, cdt = cpu_baseline_reorder(args.cpu_sample, nlay, syn.SEED_BASE + 1)
            out["cpu_baseline"] = {"value": n_s / cdt, "unit": "wavenumber-points/s", "cores": os.cpu_count(),
                                   "kind": "port", "sample": "oracle LW reorder of nwav=%d (same generator)" % n_s}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
