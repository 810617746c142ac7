#!/usr/bin/env python3
"""bench.py - LW reorder + find_g_points hot path on synthetic CKDMIP-like spectra.

One "step" = one find_g_points job of BASELINE.json configs[1], "LW FSCK all well-mixed gases" (ecckd_amd/fsck_job.py):
one band over 0-3260 cm-1, nwav = 7.2e6, nlay = 54, the gases composite, h2o, o3, co2, ch4, n2o, FLOAT spectra as
in the CKDMIP files, every gas with a background MERGED from 2-5 spectra in double (read_merged_spectrum.cpp:135-166, as
test/find_g_points_lw.sh:176-285 configures it), tolerance 0.0161 K/d (test/do_all_lw.sh:59-60).  Per gas:
   merged background                       (read_merged_spectrum.cpp:117-166)
   K1 sorting key + K3 stable sort        (reorder_spectrum.cpp:111-300)
   K4 gas preparation                     (find_g_points.cpp:872-1150)
then the g-point partition searches of ALL gases side by side, one HIP stream per gas (K5, find_g_points.cpp:1152-1266), the
overlap of the gases' g points and the merged g-point map (:1452-1483).
Metric = wavenumber-points/s = nwav * (ngas + N_pass) / t, N_pass = the passes over the spectrum the searches swept on the
device (the reference's counter total_comp_cost, find_g_points.cpp:320, also counts the requests the memo of interval
errors answered; reported beside it).  Ranks run independent jobs (weak scaling, no data-path collective); rank 0 prints ONE
JSON line.  Beside the headline, on one GPU: the same job gas after gas, and round 3's single-gas step (`single_gas`).

--config 1 (default)  BASELINE configs[1], the configuration the metric is quoted on: as above.
--config 3            BASELINE configs[3]: ONE find_g_points job over the 13 narrow longwave bands (test/config.h:141-142)
                      and 8 gases incl. the CFCs at nwav = 7.2e6, its 104 (gas, band) searches dealt to the ranks
                      (pipeline.find_g_points_resident: contiguous shares of the task table, results gathered on rank 0
                      for the overlap of the gases' g points, one all-reduce of the final cost).  Strong scaling.
--config 4            BASELINE configs[4]: optimize_lut, longwave and shortwave, 8 scenarios x 50 profiles (x 3 solar
                      zenith angles in the shortwave), nx ~ 3e5, training profiles sharded over the ranks with one
                      all-reduce of [gradient, cost] per evaluation.  Metric: L-BFGS iterations/s.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K1_VALU_PER_POINT = 574.259e6 * 64 / 7.2e6   # SQ_INSTS_VALU (wave instructions) per launch of 7.2e6 points, profiles/r02_pmc_find_g_kernels.md
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nwav", type=int, default=None)           # 7.2e6 (longwave); 3.3e6 for --config 2 (shortwave)
    ap.add_argument("--nlay", type=int, default=54)
    ap.add_argument("--tolerance", type=float, default=0.0161)   # fsck, test/do_all_lw.sh:59-60
    ap.add_argument("--tolerance-tolerance", type=float, default=0.01)  # test/find_g_points_lw.sh
    ap.add_argument("--max-iterations", type=int, default=60)
    ap.add_argument("--cpu-sample", type=int, default=1 << 15)      # points per gas of the CPU baseline's sample of the six-gas job
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-lut-opt", action="store_true")
    ap.add_argument("--no-sw", action="store_true")
    # scripts: 2000-3000 iterations per pass (test/optimize_lut_lw.sh); the first iterations of a run carry its set-up (the
    # history's allocation, the longest line searches): 40 iterations give 3 500-3 900 / s, 300 and more 5 800-6 000 / s
    ap.add_argument("--lut-opt-iterations", type=int, default=300)
    # opt-in: the profile-sharded optimiser over ALL ranks (each rank trains on its own 8 x 50 profiles, one RCCL
    # all-reduce of [gradient, cost] per evaluation).  Off by default so that the driver's scaling runs time the
    # headline metric only.
    ap.add_argument("--lut-dist", action="store_true")
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3, 4])
    # synthetic spectra: "lines" = CKDMIP-like line list (>= 1e4 lines in vibration-rotation bands, synthetic.optical_depth_lines);
    # "legacy" = the 32 isolated Lorentz lines of round 1
    ap.add_argument("--spectra", default="lines", choices=["lines", "legacy"])
    ap.add_argument("--nlines", type=int, default=12000)
    ap.add_argument("--ngas", type=int, default=8)                       # config 3
    ap.add_argument("--narrow-tolerance", type=float, default=0.013)     # narrow bands, test/do_all_lw.sh:46-60
    # HIP events around every N-th launch of the sweep kernel inside the timed region (1 = every launch): the two event records
    # and the event query cost several microseconds per batch of the search
    ap.add_argument("--profile-stride", type=int, default=4)
    ap.add_argument("--sw-tolerance", type=float, default=0.05)          # config 2 (test/do_all_sw.sh: heating_rate_tolerance per model)
    # the strong-scaling leg beside the headline: ONE configs[3] job dealt to all ranks (default: on when N > 1)
    ap.add_argument("--strong-leg", dest="strong_leg", action="store_true", default=None)
    ap.add_argument("--no-strong-leg", dest="strong_leg", action="store_false")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--fsck-gases", type=int, default=6)               # config 1: gases of the job (composite h2o o3 co2 ch4 n2o)
    # config 1: gases searched side by side on the device (0 = all the host has cores for, 1 = gas after gas as the reference)
    ap.add_argument("--gases-side-by-side", type=int, default=0)
    ap.add_argument("--no-gas-after-gas", action="store_true")         # skip the same job with the reference's gas loop beside the headline
    ap.add_argument("--no-single-gas", action="store_true")            # skip round 3's single-gas step beside the headline
    # gases of the job the CPU baseline runs on its sample (the first n of composite h2o o3 co2 ch4 n2o): three of them at 2^15
    # points are ~35 s on 16 cores, all six ~75 s
    ap.add_argument("--cpu-gases", type=int, default=3)
    # no device work at all: rendezvous (gloo), the one all-reduce, the JSON line.  For the CPU test of the launcher.
    ap.add_argument("--dry-run", action="store_true")
    args = ap.parse_args()
    if args.nwav is None:
        args.nwav = 3_300_000 if args.config == 2 else 7_200_000
    return args


def make_inputs(xp, nwav, nlay, seed, device=None, spectra="lines", nlines=12000, column_scale=30.0):
    """Synthetic target gas + background (both FLOAT like the CKDMIP spectra files)."""
    from ecckd_amd import synthetic as syn
    p = syn.pressure_grid(nlay)
    wn_h, dwn_h = syn.wavenumber_grid(nwav)
    wn = xp.as_tensor(wn_h, device=device) if device is not None else wn_h
    if spectra == "legacy":
        kw = dict(device=device, chunk=1 << 20) if device is not None else {}
        od = syn.optical_depth(xp, p, wn, seed, nlines=32, column_scale=column_scale, **kw)
        bg = syn.optical_depth(xp, p, wn, seed + 1000, nlines=24, column_scale=3.0, zero_fraction=0.0, **kw)
    else:
        kw = dict(device=device) if device is not None else {}
        od = syn.optical_depth_lines(xp, p, wn, seed, nlines=nlines, column_scale=column_scale, **kw)
        bg = syn.optical_depth_lines(xp, p, wn, seed + 1000, nlines=max(nlines // 3, 1), column_scale=3.0, zero_fraction=0.0,
                                     nclusters=5, **kw)
    return p, wn_h, dwn_h, od, bg


def SPECTRA_OF_JOB(args):
    from ecckd_amd import fsck_job
    gases = fsck_job.GASES[:args.fsck_gases]
    return sorted({g[1] for g in gases} | {b for g in gases for b, _ in g[2]})


def cpu_baseline(args, nwav_s, nlay, seed, tol, tol_tol, max_it, ctx, ngas=None):
    """The oracle ("port") on the host cores: the SAME find_g_points job (ecckd_amd/fsck_job.py: the gases with their merged
    backgrounds) at `nwav_s` points per gas - reorder_spectrum -> find_g_points gas after gas, in C end to end
    (oracle/oracle_chain.c: restated pieces composed as the reference's main() functions do, the partition search by the
    REFERENCE's own Equipartition built into oracle/_ref, a C callback for calc_error - no Python inside; the first gas's Planck
    matrix kept for the later ones as find_g_points.cpp:529, :970-984), OpenMP at the reference's sites
    (planck_function.cpp:50, equipartition.h:101 as find_g_points.cpp:231 enables it).  The sample spectra are generated on the
    device and handed to the host; a background is the double sum of its scaled spectra (read_merged_spectrum.cpp:135-166)."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as o
    from ecckd_amd import fsck_job
    job = fsck_job.FsckJob(ctx, nwav_s, nlay, ngas=ngas or args.cpu_gases, nlines=args.nlines, seed=seed, spectra=args.spectra)
    host = {name: np.ascontiguousarray(od.cpu().numpy(), dtype=np.float32) for name, od in job.od.items()}
    p, wn, dwn, t_hl = np.ascontiguousarray(job.p), job.wn_h, job.dwn_h, np.ascontiguousarray(job.t_file)
    gases = job.gases
    job.close()
    L = o.lib()
    ref = os.path.join(ROOT, "oracle", "_ref", "libequipartition_ref.so")
    if not os.path.exists(ref):
        raise RuntimeError("oracle/_ref not built")
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    I64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    b1, b2, tolv = np.array([0.0]), np.array([3260.0]), np.array([float(tol)])
    planck = np.empty((nlay + 1) * nwav_s)
    cap = 4096
    L.orc_find_g_lw_chain_ex.restype = C.c_int
    per_gas, failed = [], []
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(1)
    os.dup2(devnull, 1)  # the reference search prints progress to stdout
    wall0 = time.perf_counter()
    try:
        for gi, (name, target, bgs) in enumerate(gases):
            bg64 = None
            for bname, conc in bgs:
                refv = fsck_job.SPECTRA[bname][2]
                term = host[bname].astype(np.float64) * (1.0 if conc < 0.0 else conc / refv)
                bg64 = term if bg64 is None else bg64 + term
            bg64 = np.ascontiguousarray(bg64)
            ng, st = np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int32)
            cc, secs = np.zeros(1), np.zeros(3)
            rank = np.zeros(nwav_s, dtype=np.int32)
            r1, r2 = np.zeros(cap, dtype=np.int64), np.zeros(cap, dtype=np.int64)
            err, med, key = np.zeros(cap), np.zeros(cap), np.zeros(nwav_s)
            L.orc_chain_set_background64(P(bg64))
            rc = L.orc_find_g_lw_chain_ex(ref.encode(), C.c_int(nlay), C.c_size_t(nwav_s), P(p), P(t_hl), P(wn), P(dwn),
                                          host[target].ctypes.data_as(C.POINTER(C.c_float)), None, C.c_double(0.5), C.c_int(1),
                                          P(b1), P(b2), C.c_int(1), C.c_double(0.0), C.c_double(0.0), P(tolv), C.c_double(tol_tol),
                                          C.c_int(max_it), C.c_int(1), ng.ctypes.data_as(C.POINTER(C.c_int)), P(cc),
                                          st.ctypes.data_as(C.POINTER(C.c_int)), P(secs), rank.ctypes.data_as(C.POINTER(C.c_int32)),
                                          C.c_int(cap), I64(r1), I64(r2), P(err), P(med), P(key), P(planck), C.c_int(1 if gi == 0 else 2))
            if rc >= 20:
                # the search of this gas ended in one of calc_error's THROWs (find_g_points.cpp:296-312: bounds one rounding out
                # of order on a sample this small) - the reference run would have stopped here; the gas is left out of both sums
                failed.append(name)
                continue
            if rc:
                raise RuntimeError("orc_find_g_lw_chain_ex failed with code %d for %s" % (rc, name))
            per_gas.append(dict(name=name, ng=int(ng[0]), n_pass=float(cc[0]), status=int(st[0]), seconds=secs.copy()))
    finally:
        os.dup2(saved, 1)
        os.close(devnull)
    wall = time.perf_counter() - wall0
    secs = np.sum([g["seconds"] for g in per_gas], axis=0)
    n_pass = sum(g["n_pass"] for g in per_gas)
    return dict(points=nwav_s * (len(per_gas) + n_pass), seconds=float(secs.sum()), wall=wall, ng=[g["ng"] for g in per_gas],
                n_pass=n_pass, n_pass_per_gas=[g["n_pass"] for g in per_gas], status=[g["status"] for g in per_gas],
                gases=[g["name"] for g in per_gas], failed_gases=failed, stage_seconds=dict(reorder=secs[0], preparation=secs[1], search=secs[2]))


def sw_find_g_bench(ctx, nwav=3_300_000, nlay=54, tol=0.047, nlines=12000):
    """The shortwave twin of the step, reported beside the headline (N = 1 only): reorder key + sort + gas preparation +
    search of ONE band of BASELINE configs[2]'s shape (nwav = 3.3e6 over 250-50000 cm-1, CKDMIP-like line spectra,
    total-transmission as test/find_g_points_sw.sh with its tolerance_tolerance 0.01, the fsck tolerance of
    test/do_all_sw.sh:54, scalings as the tool clamps them (find_g_points.cpp:666-667), reference albedo 0.15, cos_sza 0.5).
    Throughput = wavenumber-points x (1 + passes swept by the search) / time, as for the longwave metric."""
    import torch
    from ecckd_amd import api, synthetic as syn
    dev = ctx.device
    lo, hi = 250.0, 50000.0
    p = syn.pressure_grid(nlay)
    wn_h, dwn_h = syn.wavenumber_grid(nwav, lo, hi)
    wn = torch.as_tensor(wn_h, device=dev)
    od = syn.optical_depth_lines(torch, p, wn, syn.SEED_BASE + 501, nlines=nlines, column_scale=5.0, device=dev, lo=lo, hi=hi)
    bg = syn.optical_depth_lines(torch, p, wn, syn.SEED_BASE + 1501, nlines=max(nlines // 3, 1), column_scale=0.5, zero_fraction=0.0,
                                 nclusters=5, device=dev, lo=lo, hi=hi)
    ssi = torch.as_tensor(syn.solar_spectral_irradiance(wn_h, dwn_h), device=dev)
    alb = torch.full((nwav,), 0.15, dtype=torch.float64, device=dev)
    key = torch.empty(nwav, dtype=torch.float64, device=dev)
    col = torch.empty(nwav, dtype=torch.float64, device=dev)
    rnk = torch.empty(nwav, dtype=torch.int32, device=dev)
    out = None
    for it in range(2):                                     # first pass warms the allocator
        ctx.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        api.reorder_key_sw(ctx, p, od, 0.25, key=key, col_od=col)
        api.stable_argsort_bands(ctx, key, [0], [nwav - 1], rank=rnk, want_ordered=False, sync=False)
        gas = api.GasSW(ctx, p, ssi, rnk, od, bg, "total-transmission", 0.02, 0.0, 0.5, alb, 0.5, 2.5)
        gas.set_band_albedo(0.15)
        st, b, e, cc = gas.find_g_band(0, nwav - 1, tol, 0.01, 60)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        swept = gas.eval_stats()["points_evaluated"] / nwav
        gas.close()
        out = {"value": nwav * (1.0 + swept) / dt, "unit": "wavenumber-points/s", "ms": dt * 1e3, "nwav": nwav, "nlay": nlay,
               "ng": len(e), "n_pass": swept, "n_pass_reference_counter": cc, "search_status": int(st), "tolerance": tol,
               "workload": "SW, one band 250-50000 cm-1 of configs[2]'s shape, line spectra, total-transmission, albedo 0.15"}
    del od, bg
    return out


GAS_NAMES = ["composite", "h2o", "o3", "co2", "ch4", "n2o", "cfc11", "cfc12"]      # test/create_lut_lw.sh:143
GAS_COLUMN_SCALE = [30.0, 100.0, 10.0, 50.0, 5.0, 5.0, 0.5, 0.5]


def config3_bench(args, ctx, dist, rank, world):
    """BASELINE configs[3]: one find_g_points job, 13 narrow longwave bands x `ngas` gases, nwav = 7.2e6, its (gas, band)
    searches dealt to the ranks.  One step = for every gas this rank searches a band of: reorder (K1 key + K3 per-band sort),
    gas preparation (K4), the side-by-side searches of its bands (K5); then the gather of the results on rank 0, the overlap
    of the gases' g points and the merged g-point map there.  Strong scaling: the job is the same for every N."""
    import torch
    from ecckd_amd import api, pipeline, shard, synthetic as syn
    dev = ctx.device
    nwav, nlay, ngas = args.nwav, args.nlay, args.ngas
    names = GAS_NAMES[:ngas]
    b1, b2 = syn.LW_NARROW_BANDS
    nband = len(b1)
    p = syn.pressure_grid(nlay)
    wn_h, dwn_h = syn.wavenumber_grid(nwav)
    wn, dwn = torch.as_tensor(wn_h, device=dev), torch.as_tensor(dwn_h, device=dev)
    _, begin, end = api.band_ranges(wn_h, b1, b2)
    t_ideal, t_file = api.idealised_temperature(p), syn.temperature_profile(p)
    tasks = shard.task_table(range(ngas), nband)
    mine = [tasks[t] for t in shard.deal_tasks(len(tasks), rank, world)]
    my_gases = sorted({g for g, _ in mine})
    spectra = {}
    for gi in my_gases:                                    # resident before the timed region (the tools read them from files)
        _, _, _, od, bg = make_inputs(torch, nwav, nlay, syn.SEED_BASE + 301 + 17 * gi, device=dev, spectra=args.spectra,
                                      nlines=args.nlines, column_scale=GAS_COLUMN_SCALE[gi % len(GAS_COLUMN_SCALE)])
        spectra[gi] = (od, bg)

    def reorder(od):
        key, _ = api.reorder_key_lw(ctx, p, t_ideal, wn, dwn, od, 0.5)
        rnk, _ = api.stable_argsort_bands(ctx, key, begin, end, want_ordered=False)
        return key, rnk

    first_order = None
    if 0 not in my_gases and my_gases:
        # stands for the first gas's ordering FILE, which a process that searches none of its bands reads for the shared
        # Planck matrix (find_g_points.cpp:529, :970-984): made once, outside the timed region
        _, _, _, od0, _ = make_inputs(torch, nwav, nlay, syn.SEED_BASE + 301, device=dev, spectra=args.spectra, nlines=args.nlines,
                                      column_scale=GAS_COLUMN_SCALE[0])
        first_order = dict(temperature_hl=t_file, wn=wn, dwn=dwn, rank=reorder(od0)[1])
        del od0

    def load_gas(gi):
        od, bg = spectra[gi]
        key, rnk = reorder(od)                            # the gas's reorder_spectrum step: part of the timed work
        return dict(pressure_hl=p, temperature_hl=t_file, wn=wn, dwn=dwn, rank=rnk, od=od, bg=bg, sorting_variable=key,
                    band_begin=begin, band_end=end, min_g_points=np.ones(nband, dtype=int), max_g_points=np.full(nband, 256))

    out = {}

    def step():
        res = pipeline.find_g_points_resident(ctx, names, load_gas, nband, args.narrow_tolerance, (lambda: first_order),
                                              "transmission", 0.0, 0.0, args.tolerance_tolerance, args.max_iterations,
                                              rank=rank, world_size=world, gases_side_by_side=args.gases_side_by_side)
        out.update(res)
        return res["points"]

    return step, out, dict(ngas=ngas, nband=nband, tasks=len(tasks), tasks_this_rank=len(mine), gases_this_rank=len(my_gases))


SW_GAS_NAMES = ["h2o", "o3", "co2"]
SW_COLUMN_SCALE = [5.0, 1.5, 3.0]


def config2_bench(args, ctx, dist, rank, world):
    """BASELINE configs[2]: one SHORTWAVE find_g_points job, 32 bands (equal width in log wavenumber over 250-50000 cm-1) x
    H2O + O3 + CO2 at nwav = 3.3e6, total-transmission averaging with the tool's scaling clamps (find_g_points.cpp:666-667),
    reference albedo 0.15 in the bands below 10 000 cm-1 (:756-760, :921-923), cos_sza 0.5; the 96 (gas, band) searches dealt
    to the ranks as in configs[3].  One step = reorder (K2 key + K3 per-band sort) + gas preparation + the side-by-side
    searches of every gas this rank has a band of, then gather / overlap / merged map on rank 0."""
    import torch
    from ecckd_amd import api, pipeline, shard, synthetic as syn
    dev = ctx.device
    nwav, nlay, nband, lo, hi = args.nwav, args.nlay, 32, 250.0, 50000.0
    names = SW_GAS_NAMES
    ngas = len(names)
    p = syn.pressure_grid(nlay)
    wn_h, dwn_h = syn.wavenumber_grid(nwav, lo, hi)
    wn, dwn = torch.as_tensor(wn_h, device=dev), torch.as_tensor(dwn_h, device=dev)
    edges = np.geomspace(lo, hi, nband + 1)
    b1, b2 = edges[:-1], edges[1:].copy()
    b2[-1] = hi + 1.0
    _, begin, end = api.band_ranges(wn_h, b1, b2)
    band_albedo = np.where(b2 <= 10000.0, 0.15, 0.0)
    sw = dict(ssi=torch.as_tensor(syn.solar_spectral_irradiance(wn_h, dwn_h), device=dev), cos_sza=0.5, band_albedo=band_albedo,
              albedo=torch.as_tensor(np.where(wn_h < b2[band_albedo > 0].max(), 0.15, 0.0), device=dev))
    tasks = shard.task_table(range(ngas), nband)
    mine = [tasks[t] for t in shard.deal_tasks(len(tasks), rank, world)]
    my_gases = sorted({g for g, _ in mine})
    spectra = {}
    for gi in my_gases:
        od = syn.optical_depth_lines(torch, p, wn, syn.SEED_BASE + 501 + 17 * gi, nlines=args.nlines, column_scale=SW_COLUMN_SCALE[gi],
                                     device=dev, lo=lo, hi=hi)
        bg = syn.optical_depth_lines(torch, p, wn, syn.SEED_BASE + 1501 + 17 * gi, nlines=max(args.nlines // 3, 1), column_scale=0.5,
                                     zero_fraction=0.0, nclusters=5, device=dev, lo=lo, hi=hi)
        spectra[gi] = (od, bg)
    rank0 = torch.zeros(nwav, dtype=torch.int32, device=dev)       # a process without tasks only asks for the number of points

    def load_gas(gi):
        od, bg = spectra[gi]
        key, _ = api.reorder_key_sw(ctx, p, od, 0.25)            # the gas's reorder_spectrum step: part of the timed work
        rnk, _ = api.stable_argsort_bands(ctx, key, begin, end, want_ordered=False)
        return dict(pressure_hl=p, temperature_hl=None, wn=wn, dwn=dwn, rank=rnk, od=od, bg=bg, sorting_variable=key,
                    band_begin=begin, band_end=end, min_g_points=np.ones(nband, dtype=int), max_g_points=np.full(nband, 256),
                    min_scaling=0.5, max_scaling=2.5)

    out = {}

    def step():
        res = pipeline.find_g_points_resident(ctx, names, load_gas, nband, args.sw_tolerance, (lambda: dict(rank=rank0)),
                                              "total-transmission", 0.02, 0.0, 0.02, args.max_iterations,
                                              rank=rank, world_size=world, sw=sw, gases_side_by_side=args.gases_side_by_side)
        out.update(res)
        return res["points"]

    return step, out, dict(ngas=ngas, nband=nband, tasks=len(tasks), tasks_this_rank=len(mine), gases_this_rank=len(my_gases))


def ckd_model_sw(model, seed=0):
    """Shortwave variant of synthetic.ckd_model: solar irradiance and Rayleigh scattering per g point, no Planck table."""
    m = dict(model, gases=[dict(g) for g in model["gases"]])
    ng = m["planck_function"].shape[1]
    rs = np.random.RandomState(seed + 99)
    ssi = rs.uniform(0.5, 1.5, ng)
    m["solar_irradiance"] = 1340.0 * ssi / ssi.sum()
    m["rayleigh_molar_scattering"] = 10.0 ** rs.uniform(-7.0, -5.5, ng)
    m["planck_function"] = None
    m["temperature_planck"] = None
    m["ng"] = ng
    for g in m["gases"]:
        for k in ("molar_abs", "min_molar_abs", "max_molar_abs"):
            g[k] = g[k] * 0.002
    return m


L2_GATHER_PEAK_GBS = 17600.0   # MI355X_MICROARCH.md, L2 section: rows shared by every workgroup gathered from the XCDs' L2s, 16.8-18.8 TB/s


def _lut_opt_traffic(sw):
    if sw:
        return None
    for r in (4,):
        q = os.path.join(ROOT, "profiles", "r%02d_traffic_lut_opt.json" % r)
        if os.path.exists(q):
            with open(q) as f:
                return json.load(f)["corrected_bytes_per_iteration"]
    return None


def lut_opt_cpu_baseline(model, scenes, cfg, evaluations=10):
    """The serial CPU side of one optimize_lut iteration (solve_adept.cpp:72-211 + calc_background_cost_function,
    ckd_model.cpp:840-877), ONE thread as the reference runs it (optimize_lut.cpp:159): the oracle's forward model
    (oracle_ckd.c), its hand-written reverse mode (oracle_adjoint.c) and the prior through the dense LAPACK inverse the
    reference stores (the inverse itself is built once, outside the timed evaluations).  Adept's tape record / reverse is
    not reproduced, so this is a LOWER bound on the reference's time per iteration."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pyoracle
    import ckd_synth
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        threadpool_limits = None
    full = dict(cfg, spectral_boundary_weight=0.0, negative_od_penalty=1.0e4, pressure_weight_power=0.5, cap_relative_linear=0.0)
    orc = ckd_synth.Oracle(pyoracle, model, scenes, full)
    x = orc.x0.copy()
    ctxmgr = threadpool_limits(limits=1) if threadpool_limits else None
    if ctxmgr:
        ctxmgr.__enter__()
    try:
        orc.prior_matrices()                                    # dense inverse: once per run in the reference too
        t0 = time.perf_counter()
        for _ in range(evaluations):
            J_rt, g_rt = orc.cost_grad_rt(x)
            J_b, g_b = orc.cost_prior(x, full["prior_error"])
        dt = (time.perf_counter() - t0) / evaluations
    finally:
        if ctxmgr:
            ctxmgr.__exit__(None, None, None)
    return {"value": 1.0 / dt, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": "%d cost + gradient evaluations of the SAME longwave problem (8 scenarios x 50 profiles, nx = %d) by the oracle "
                      "(oracle_ckd.c forward, oracle_adjoint.c reverse, dense-inverse prior through one BLAS thread): %.2f s per "
                      "evaluation, counted as one evaluation per iteration; J = %.6g" % (evaluations, x.size, dt, J_rt + J_b),
            "seconds_per_evaluation": dt, "cost_at_start": J_rt + J_b}


def lut_opt_bench(ctx, iterations, sharded=False, rank=0, world=1, sw=False, cpu=False):
    """Second half of the headline metric: LUT-optimisation iterations/s (solve_adept.cpp:295-299 logs one
    line per L-BFGS iteration).  Synthetic CKD model with the shapes of configs[4]: ng = 64, 6 x 53 (T, p)
    grid, H2O look-up table with 12 mole fractions (nx ~ 3e5), 8 scenarios x 50 columns x 54 layers."""
    from ecckd_amd import api, synthetic as syn
    model = syn.ckd_model(ng=64, nt=6, np_=53, nband=13, seed=11, nconc=12)
    truth = syn.ckd_model(ng=64, nt=6, np_=53, nband=13, seed=11, nconc=12)
    if sw:
        model, truth = ckd_model_sw(model, 11), ckd_model_sw(truth, 11)
    rs = np.random.RandomState(12)
    for g in truth["gases"]:
        g["molar_abs"] = g["molar_abs"] * np.exp(0.25 * rs.normal(size=g["molar_abs"].shape))
    scenes = syn.ckd_scenes(model, nscene=8, ncol=50, nlay=54, seed=13 + 100 * rank)   # sharded: every rank its own profiles
    if not sw:
        cfg = dict(flux_weight=0.2, flux_profile_weight=0.0, broadband_weight=0.5, prior_error=4.0, pressure_corr=0.95,
                   temperature_corr=0.95, conc_corr=0.95)       # test/optimize_lut_lw.sh:55 final pass
    else:
        cfg = dict(flux_weight=0.2, flux_profile_weight=0.0, broadband_weight=0.5, prior_error=2.0, pressure_corr=0.8,
                   temperature_corr=0.8, conc_corr=0.8)         # test/optimize_lut_sw.sh:75
        mu0 = (1.0, 0.5, 0.1)                                   # three zenith angles per profile (lbl_fluxes.cpp:82)
        rep = lambda a: np.ascontiguousarray(np.repeat(a, len(mu0), axis=0))
        scenes = [dict(pressure_hl=rep(s["pressure_hl"]), temperature_hl=rep(s["temperature_hl"]), vmr_fl=rep(s["vmr_fl"]),
                       gas_present=s["gas_present"], mu0=np.tile(np.asarray(mu0), s["pressure_hl"].shape[0]), tsi=1361.0,
                       albedo=np.full(13, 0.15)) for s in scenes]
    ib = model["iband_per_g"]
    nhl = 55
    ncs = scenes[0]["pressure_hl"].shape[0]                   # columns per scene: 50, or 150 with the zenith angles
    for s in scenes:                                         # placeholders, replaced by the truth model's fluxes
        s["flux_dn"] = np.zeros((ncs, nhl, 13))
        s["flux_up"] = np.zeros((ncs, nhl, 13))
    t_opt = api.Optimizer(ctx, truth, scenes, **cfg)
    _, fl = t_opt.forward(t_opt.initial_state())
    t_opt.close()
    band = np.stack([fl[..., ib == b].sum(-1) for b in range(13)], axis=-1)   # (ncol, 2, nhl, nband)
    c0 = 0
    for s in scenes:
        s["flux_dn"] = np.ascontiguousarray(band[c0:c0 + ncs, 0])
        s["flux_up"] = np.ascontiguousarray(band[c0:c0 + ncs, 1])
        c0 += ncs
    opt = api.Optimizer(ctx, model, scenes, **cfg)
    if sharded:
        opt.set_allreduce()
    x0 = opt.initial_state()
    J0, g0 = opt.cost_grad(x0)                               # warm-up
    t0 = time.perf_counter()
    res = opt.minimize(max_iterations=iterations, convergence_criterion=0.0, bounded=True)
    dt = time.perf_counter() - t0
    n_eval = 0
    t1 = time.perf_counter()
    for _ in range(20):
        opt.cost_grad(x0)
    dt_eval = (time.perf_counter() - t1) / 20
    out = {"iters_per_s": res["iterations"] / dt, "iterations": res["iterations"], "nx": opt.nx,
           "cells": world * 8 * ncs * 54 * 64, "cost_grad_ms": dt_eval * 1e3, "J0": J0, "J_final": res["cost"],
           "status": res["status"], "region": "shortwave" if sw else "longwave", "profiles_per_rank": 8 * ncs}
    # What bounds an iteration: the look-up gather of K8a (every cell reads one ng-long coefficient row per table entry) and
    # its transpose in K8b (every node reads the dJ/dtau row of each cell that references it) are served by the L2s - the
    # coefficient vector (2.4 MB) and dJ/dtau live there -, then the three vector passes of the L-BFGS update.  Algorithmic
    # bytes per iteration (one evaluation) over the L2 gather rate of the microarchitecture guide:
    nent = sum(8 if g["conc"] == "lut" else 4 for g in model["gases"]) + (1 if sw else 0)
    nent_active = sum((8 if g["conc"] == "lut" else 4) for g in model["gases"] if g["active"])
    ncell = 8 * ncs * 54
    gather = ncell * nent * 64 * 8.0
    scatter = ncell * nent_active * 64 * 8.0
    vectors = 42 * opt.nx * 8.0            # update 21, direction 15, step 6 vectors of nx doubles
    bytes_it = gather + scatter + vectors
    ach = bytes_it * out["iters_per_s"] / 1e9
    out["roofline"] = {"bound": "l2 gather", "achieved": ach, "peak": L2_GATHER_PEAK_GBS, "unit": "GB/s", "frac": ach / L2_GATHER_PEAK_GBS,
                       "algorithmic_bytes_per_iteration": bytes_it,
                       "bytes": {"k8a_gather": gather, "k8b_transpose_gather": scatter, "lbfgs_vector_passes": vectors},
                       "floor_us_per_iteration": bytes_it / (L2_GATHER_PEAK_GBS * 1e9) * 1e6,
                       "measured_us_per_iteration": 1e6 / out["iters_per_s"],
                       "launches_per_iteration": 8, "host_waits_per_iteration": 2,
                       # HBM bytes per iteration from the committed counter passes (2 x FETCH_SIZE + WRITE_SIZE of the optimiser's
                       # kernels, tools/opt_traffic.py): the gathers themselves are served by the L2s
                       "traffic": _lut_opt_traffic(sw),
                       "note": "kernel averages of the same command: profiles/r03_opt_kernel_stats.csv"}
    if cpu and not sw and not sharded:
        out["cpu_baseline"] = lut_opt_cpu_baseline(model, scenes, cfg)
    if sharded:
        out["ranks"] = world
        out["allreduce_bytes_per_evaluation"] = (opt.nx + 1) * 8
    opt.close()
    return out


def find_g_main(args, ctx, dist, rank, world, barrier, use_dist):
    """configs[1] (default; weak scaling: every rank its own single-gas, single-band job) and configs[3] (one 13-band,
    8-gas job dealt to the ranks; strong scaling): warm-up, K timed steps between barriers, ONE all-reduce of
    [elapsed, points, final cost], the JSON line on rank 0."""
    import torch
    from ecckd_amd import api, synthetic as syn
    dev = ctx.device
    nwav, nlay = args.nwav, args.nlay
    info = {}
    extra = {}
    if args.config in (2, 3):
        step3, info, extra = (config3_bench if args.config == 3 else config2_bench)(args, ctx, dist, rank, world)
        od = None
        def step():
            return step3() / nwav             # passes over the nwav-point spectrum done by this rank
    else:
        # every rank runs its own six-gas job on its own synthetic spectra: independent jobs, SURVEY.md 8e
        from ecckd_amd import fsck_job
        job = fsck_job.FsckJob(ctx, nwav, nlay, ngas=args.fsck_gases, nlines=args.nlines, seed=syn.SEED_BASE + 1 + 170 * rank,
                               spectra=args.spectra)
        od = None
        torch.cuda.synchronize()

        def job_step(side_by_side):
            res = job.run(args.tolerance, args.tolerance_tolerance, args.max_iterations, gases_side_by_side=side_by_side)
            info.update(ng=res["ng"], ng_per_gas=[int(sum(g["n_g_points"])) for g in res["gases"]],
                        status=[int(g["status"][0]) for g in res["gases"]], comp_cost=res["comp_cost_sum"], cost_sum=res["cost_sum"],
                        n_unassigned=res["n_unassigned"], phase_seconds=res["phase_seconds"],
                        sweep_bytes_per_point=(2 * nlay + 1) * 8, points=res["points"])
            # passes over the spectrum in this step: one reorder / preparation pass per gas + what the searches actually swept
            # on the device.  (The reference's counter total_comp_cost counts every interval the searches ask for; the library
            # answers an interval it has evaluated before from its memo, so fewer points are swept than the counter says.)
            return res["points"] / nwav

        def step():
            return job_step(args.gases_side_by_side)
    for _ in range(args.warmup):
        step()
    ctx.profile_enable(max(1, args.profile_stride))
    barrier()
    t0 = time.perf_counter()
    passes = 0.0
    for _ in range(args.steps):
        passes += step()
    barrier()
    dt = time.perf_counter() - t0
    rt_calls, rt_ms, rt_pts = ctx.profile_get("k_rt_lw_bb")            # the launches that were timed
    all_calls, _, all_pts = ctx.profile_get("k_rt_lw_bb.all")          # every launch
    k1_calls, k1_ms, k1_pts = ctx.profile_get("k_reorder_key_lw")
    # the single collective of the path: max elapsed, total passes, total final cost (RCCL over xGMI)
    from ecckd_amd import shard
    dt, passes, total_cost, ranks_seen = shard.reduce_scalars(dt, passes, info.get("cost_sum", 0.0), device=dev, count=True)

    gw_calls, gw_ms, gw_pts = ctx.profile_get("find_g_gases")          # the windows in which the gases' searches ran side by side

    # Beside the headline, one GPU only, never as `value`:
    #  * what a caller that hands over HOST buffers pays on top (the tools do: spectra come from NetCDF files): the job's FLOAT
    #    spectra over PCIe from pinned memory, one after the other (the tools' reader overlaps this with the previous gas);
    #  * the same job gas after gas - the reference's loop - with the sweep launches timed while they have the device to
    #    themselves;
    #  * round 3's step: ONE gas with a single-file FLOAT background (kept as FLOAT pairs: 656 B per point).
    h2d_ms = None
    gas_after_gas = None
    single_gas = None
    if rank == 0 and world == 1 and args.config == 1:
        try:
            any_od = next(iter(job.od.values()))
            host_od = torch.empty((nlay, nwav), dtype=torch.float32).pin_memory()
            host_od.copy_(any_od.cpu())
            dst = torch.empty_like(any_od)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(len(job.od)):
                dst.copy_(host_od, non_blocking=True)
            torch.cuda.synchronize()
            h2d_ms = (time.perf_counter() - t1) * 1e3
            del host_od, dst
        except RuntimeError:
            h2d_ms = None
        head_info = dict(info)
    if rank == 0 and world == 1 and args.config == 1 and not args.no_gas_after_gas:
        ctx.profile_enable(max(1, args.profile_stride))
        barrier()
        t1 = time.perf_counter()
        p_seq = job_step(1)
        barrier()
        dt_seq = time.perf_counter() - t1
        seq_info = dict(info)
        info.clear()
        info.update(head_info)
        same = all(seq_info.get(k) == head_info.get(k) for k in ("ng", "ng_per_gas", "status", "cost_sum", "comp_cost", "points", "n_unassigned"))
        sq_calls, sq_ms, sq_pts = ctx.profile_get("k_rt_lw_bb")
        sq_all, _, sq_all_pts = ctx.profile_get("k_rt_lw_bb.all")
        k1_calls, k1_ms, k1_pts = ctx.profile_get("k_reorder_key_lw")       # with the device to itself (side by side it shares it)
        bpp = (2 * nlay + 1) * 8
        gas_after_gas = {"ms_per_step": dt_seq * 1e3, "value": nwav * p_seq / dt_seq, "unit": "wavenumber-points/s",
                         "phase_ms": {k: round(v * 1e3, 2) for k, v in seq_info.get("phase_seconds", {}).items()},
                         # g points per gas, search statuses, the sum of the g points' errors (bit for bit), points swept
                         "identical_results": bool(same),
                         "sweep_launches": sq_all, "sweep_launches_timed": sq_calls, "avg_launch_ms": sq_ms / max(sq_calls, 1),
                         "points_per_launch": sq_pts / max(sq_calls, 1),
                         "achieved_GBs_per_launch": sq_pts * bpp / max(sq_ms * 1e-3, 1e-12) / 1e9,
                         "frac_per_launch": sq_pts * bpp / max(sq_ms * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS,
                         "note": "the same job with the reference's gas loop (find_g_points.cpp:655): merge, reorder, prepare, search, "
                                 "release, one gas after the other on one stream; every sweep launch has the device to itself"}
    if rank == 0 and world == 1 and args.config == 1 and not args.no_single_gas:
        job.close()
        torch.cuda.empty_cache()
        p1, wn_h, dwn_h, od1, bg1 = make_inputs(torch, nwav, nlay, syn.SEED_BASE + 1, device=dev, spectra=args.spectra, nlines=args.nlines)
        wn1, dwn1 = torch.as_tensor(wn_h, device=dev), torch.as_tensor(dwn_h, device=dev)
        t_ideal, t_file = api.idealised_temperature(p1), syn.temperature_profile(p1)
        sg = {}

        def single_step():
            key1, _ = api.reorder_key_lw(ctx, p1, t_ideal, wn1, dwn1, od1, 0.5)
            rnk1, _ = api.stable_argsort_bands(ctx, key1, [0], [nwav - 1], want_ordered=False)
            gas = api.GasLW(ctx, p1, t_file, wn1, dwn1, rnk1, od1, bg1, "transmission", flux_weight=0.0)
            st, b, e, cc = gas.find_g_band(0, nwav - 1, args.tolerance, args.tolerance_tolerance, args.max_iterations)
            sg.update(ng=len(e), status=int(st), comp_cost=cc, cost_sum=float(np.sum(e)), eval_stats=gas.eval_stats(),
                      bytes=gas.sweep_bytes_per_point())
            gas.close()
            return 1.0 + sg["eval_stats"]["points_evaluated"] / nwav

        single_step()
        ctx.profile_enable(max(1, args.profile_stride))
        barrier()
        t1 = time.perf_counter()
        nst = max(2, min(args.steps, 5))
        ps = sum(single_step() for _ in range(nst))
        barrier()
        dt1 = time.perf_counter() - t1
        s_calls, s_ms, s_pts = ctx.profile_get("k_rt_lw_bb")
        s_gbs = s_pts * sg["bytes"] / max(s_ms * 1e-3, 1e-12) / 1e9
        single_gas = {"workload": "round 3's step: ONE gas, single-file FLOAT background (FLOAT pairs in the sweep), nwav=%d" % nwav,
                      "value": nwav * ps / dt1, "unit": "wavenumber-points/s", "ms_per_step": dt1 / nst * 1e3, "steps": nst,
                      "ng": sg["ng"], "search_status": sg["status"], "n_pass_per_step": ps / nst - 1.0,
                      "n_pass_reference_counter": sg["comp_cost"], "final_cost_sum_K_per_day": sg["cost_sum"],
                      "roofline": {"kernel": "k_rt_lw_bb_mirror<54,true>", "algorithmic_bytes_per_point": sg["bytes"],
                                   "achieved": s_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": s_gbs / HBM_PEAK_GBS,
                                   "avg_launch_ms": s_ms / max(s_calls, 1), "points_per_launch": s_pts / max(s_calls, 1)}}
        del od1, bg1

    lut_sharded = None
    if args.lut_dist and use_dist and not args.no_lut_opt:
        lut_sharded = lut_opt_bench(ctx, args.lut_opt_iterations, sharded=True, rank=rank, world=world)

    if rank == 0:
        points = nwav * passes
        # dominant kernel: K5c k_rt_lw_bb.  Algorithmic bytes per point and pass: planck (nlay+1) rows of f64 +
        # background optical depth (nlay) rows = (2*nlay+1)*8 B with DOUBLE background rows; a FLOAT background (the bench's, as
        # in the CKDMIP files) is kept as FLOAT pairs and widened in the sweep - the same bits - so (nlay+1)*8 + nlay*4 B
        # are what the algorithm has to read (ecckd_gas_sweep_bytes_per_point tells which layout the gas holds).
        rt_bytes_per_pt = info.get("sweep_bytes_per_point", (2 * nlay + 1) * 8) if isinstance(info, dict) else (2 * nlay + 1) * 8
        rt_gbs = rt_pts * rt_bytes_per_pt / (rt_ms * 1e-3) / 1e9 if rt_ms > 0 else 0.0
        k1_bytes_per_pt = nlay * 4 + 32                  # FLOAT optical depths
        # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and
        # WRITE_SIZE collected separately; 2*FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md section HBM),
        # scaled to this run's points per launch
        traffic = None
        tpath = next((q for q in (os.path.join(ROOT, "profiles", "r%02d_traffic_k_rt_lw_bb.json" % r) for r in (4, 3, 2, 1)) if os.path.exists(q)), "")
        if os.path.exists(tpath) and rt_calls:
            with open(tpath) as f:
                traffic = json.load(f)["corrected_bytes_per_point"] * rt_pts / rt_calls
        out = {
            "metric": "wavenumber-points/s (LW reorder+find_g)" if args.config != 2 else "wavenumber-points/s (SW reorder+find_g)",
            "value": points / dt,
            "unit": "wavenumber-points/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,          # ones summed by the path's single all-reduce (RCCL): the ranks that took part
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if args.config == 1 else "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": ({"workload": "configs[1]: ONE find_g_points job per rank, LW FSCK (1 band 0-3260 cm-1) x %d gases (%s), every gas "
                                    "with a background merged in double from %s spectra (DOUBLE rows in the sweep: %d B per point), "
                                    "nwav=%d, nlay=%d, spectra FLOAT, tolerance %g K/d, averaging transmission; per gas merge + reorder "
                                    "(key, sort) + preparation, then %s, overlap of the gases' g points, merged g-point map"
                                    % (args.fsck_gases, " ".join(job.names), "/".join(str(len(v)) for v in job.background_names.values()),
                                       (2 * nlay + 1) * 8, nwav, nlay, args.tolerance,
                                       "the searches gas after gas" if args.gases_side_by_side == 1 else
                                       "the searches of all gases side by side (one HIP stream per gas)"),
                        "gases": job.names, "background_spectra": job.background_names,
                        "gases_side_by_side": args.gases_side_by_side if args.gases_side_by_side else len(job.names),
                        "n_pass_per_step": (passes / args.steps / world) - len(job.names),
                        "n_pass_reference_counter": info.get("comp_cost"), "ng_merged": info.get("ng"), "ng_per_gas": info.get("ng_per_gas"),
                        "search_status_per_gas": info.get("status"), "n_unassigned": info.get("n_unassigned"),
                        "final_cost_sum_K_per_day": total_cost,
                        "phase_ms_last_step": {k: round(v * 1e3, 2) for k, v in info.get("phase_seconds", {}).items()}}
                       if args.config == 1 else
                       {"workload": ("configs[3]: ONE find_g_points job, 13 narrow LW bands (test/config.h:141-142) x %d gases "
                                     "(%s), nwav=%d, nlay=%d, od FLOAT, tolerance %g K/d, averaging transmission; the %d "
                                     "(gas, band) searches dealt to the ranks in contiguous shares, results gathered on "
                                     "rank 0 (overlap of the gases' g points, merged g-point map)"
                                     % (extra["ngas"], " ".join(GAS_NAMES[:extra["ngas"]]), nwav, nlay, args.narrow_tolerance,
                                        extra["tasks"])) if args.config == 3 else
                                    ("configs[2]: ONE shortwave find_g_points job, 32 bands (log-spaced, 250-50000 cm-1) x %d gases "
                                     "(%s), nwav=%d, nlay=%d, od FLOAT, tolerance %g K/d, averaging total-transmission, albedo 0.15 "
                                     "below 10000 cm-1; the %d (gas, band) searches dealt to the ranks in contiguous shares, "
                                     "results gathered on rank 0" % (extra["ngas"], " ".join(SW_GAS_NAMES), nwav, nlay,
                                                                     args.sw_tolerance, extra["tasks"])),
                        "passes_over_the_spectrum_per_step": passes / args.steps, "ng_merged": info.get("ng"),
                        "ng_per_gas": [int(sum(g["n_g_points"])) for g in info.get("gases", [])],
                        "searches_not_converged": int(sum(st != 0 for g in info.get("gases", []) for st in g["status"])),
                        "n_unassigned": info.get("n_unassigned"), "final_cost_sum_K_per_day": total_cost,
                        "phase_ms_last_step_rank_0": {k: round(v * 1e3, 2) for k, v in info.get("phase_seconds", {}).items()},
                        "tasks_on_rank_0": extra["tasks_this_rank"]}),
            "spectra": {"generator": args.spectra, "lines_per_gas": args.nlines if args.spectra == "lines" else 32},
            # who does what at this world size, and what crosses the links: the (gas, band) searches of every rank and the path's
            # single collective (ecckd_amd/shard.py: one all-reduce of [elapsed, points, final cost, 1])
            "sharding": ({"scaling": "weak", "jobs_per_rank": 1, "searches_per_rank": [args.fsck_gases] * world,
                          "collectives_per_run": 1, "allreduce_bytes": 4 * 8, "data_path_collectives": 0}
                         if args.config == 1 else
                         {"scaling": "strong", "tasks": extra["tasks"],
                          "searches_per_rank": [len(shard.deal_tasks(extra["tasks"], r, world)) for r in range(world)],
                          "collectives_per_step": "1 gather of the per-band results to rank 0 (a few numbers per g point), 1 all-reduce of "
                                                  "[work counter, final cost] (16 B), 1 max-reduce of every gas's g-point map to rank 0 "
                                                  "(nwav int32 per gas)",
                          "data_path_collectives": 0}),
            "search": {"error_batches_per_step": all_calls / max(args.steps, 1),
                       "points_per_batch": all_pts / max(all_calls, 1)},
            "roofline": {"bound": "hbm", "kernel": "k_rt_lw_bb", "achieved": rt_gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": rt_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": rt_bytes_per_pt * rt_pts / max(rt_calls, 1),
                         "launches": all_calls, "launches_timed": rt_calls,
                         "timing": "HIP events around every %d-th launch inside the timed region; averages over the timed launches" % max(1, args.profile_stride),
                         "avg_launch_ms": rt_ms / max(rt_calls, 1),
                         "algorithmic_bytes_per_point": rt_bytes_per_pt,
                         "points_per_launch": rt_pts / max(rt_calls, 1),
                         "share_of_step_time": rt_ms / max(rt_calls, 1) * all_calls * 1e-3 / dt,
                         # K1 is bound by fp64 vector issue, not by HBM: ~94 VALU instructions per layer and point (2 exp, 2
                         # divisions; rocprofv3 SQ_INSTS_VALU, profiles/) at 4 cycles per wave64 fp64 instruction
                         "k_reorder_key_lw": {"bound": "fp64 issue", "avg_launch_ms": k1_ms / max(k1_calls, 1),
                                              "hbm_achieved_GBs": k1_pts * k1_bytes_per_pt / max(k1_ms * 1e-3, 1e-12) / 1e9,
                                              "algorithmic_bytes_per_point": k1_bytes_per_pt,
                                              "valu_instructions_per_point": K1_VALU_PER_POINT,
                                              "issue_floor_ms": K1_VALU_PER_POINT * (k1_pts / max(k1_calls, 1)) / 64.0 * 4.0
                                                                / (256 * 4 * 2.4e9) * 1e3,
                                              "frac_of_fp64_issue_roof": (K1_VALU_PER_POINT * (k1_pts / max(k1_calls, 1)) / 64.0 * 4.0
                                                                          / (256 * 4 * 2.4e9) * 1e3) / max(k1_ms / max(k1_calls, 1), 1e-12)}},
        }
        if args.config == 1 and gw_calls and gw_ms > 0:
            # The gases' searches ran side by side, every gas on its own stream: sweep launches of several streams overlap, so
            # a launch's own duration (its HIP events) includes the time it shared the device and says nothing about the memory
            # system.  What does: the algorithmic bytes of ALL sweep launches over the time in which the searches ran - the
            # window between two HIP events that ecckd_find_g_gases records on the context's stream round the searches (every
            # lane synchronised at both ends).  Average launch duration = window / launches: the device time a launch cost.
            # The window also holds the interval-sum and cost kernels and the host's turnarounds, so the figure is a LOWER
            # bound of the sweep's own rate.  Per-launch figures with the device to itself: `gas_after_gas`.
            agg = gw_pts * rt_bytes_per_pt / (gw_ms * 1e-3) / 1e9
            rl = out["roofline"]
            rl.update({"kernel": "k_rt_lw_bb_mirror<54,false>", "achieved": agg, "frac": agg / HBM_PEAK_GBS,
                       "algorithmic_bytes_per_launch": rt_bytes_per_pt * all_pts / max(all_calls, 1),
                       "points_per_launch": all_pts / max(all_calls, 1), "avg_launch_ms": gw_ms / max(all_calls, 1),
                       "traffic": (traffic / (rt_pts / max(rt_calls, 1)) * (all_pts / max(all_calls, 1))) if traffic else None,
                       "search_windows": gw_calls, "search_window_ms_per_step": gw_ms / max(args.steps, 1),
                       "points_swept_per_step": gw_pts / max(args.steps, 1),
                       "share_of_step_time": gw_ms * 1e-3 / dt,
                       "timing": "the gases' sweeps overlap on %d streams: achieved = algorithmic bytes of all sweep launches / the "
                                 "HIP-event window round the side-by-side searches (ecckd_find_g_gases), avg_launch_ms = that window / "
                                 "launches; a lower bound (the window holds the small kernels and host turnarounds too)"
                                 % len(job.names),
                       "per_launch_under_overlap": {"launches_timed": rt_calls, "avg_launch_ms": rt_ms / max(rt_calls, 1),
                                                    "points_per_launch": rt_pts / max(rt_calls, 1), "achieved": rt_gbs,
                                                    "note": "HIP events round every %d-th launch on its own stream: durations include "
                                                            "the time shared with the other gases' kernels" % max(1, args.profile_stride)}})
        if gas_after_gas is not None:
            out["gas_after_gas"] = gas_after_gas
            out["gas_after_gas"]["side_by_side_speedup"] = gas_after_gas["ms_per_step"] / (dt / args.steps * 1e3)
        if single_gas is not None:
            out["single_gas"] = single_gas
        if args.config == 2:
            # the shortwave sweep: one launch evaluates an interval with both scaled fits from ONE fetch of the column of
            # optical depths, (nlay + 1) * 8 B per point (background optical depths + solar irradiance); it is bound by fp64
            # issue (2 x 2 nlay exp per point), the HBM figure is reported against the same peak for comparison
            sw_calls, sw_ms, sw_pts = ctx.profile_get("k_rt_sw_bb")
            sw_all, _, sw_all_pts = ctx.profile_get("k_rt_sw_bb.all")
            sw_bytes = (nlay + 1) * 8
            sw_gbs = sw_pts * sw_bytes / (sw_ms * 1e-3) / 1e9 if sw_ms > 0 else 0.0
            exp_per_pt = 2 * 2 * nlay                      # two fits x (direct + reflected beam) x nlay
            out["roofline"] = {"bound": "hbm", "kernel": "k_rt_sw_bb_fast", "achieved": sw_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": sw_gbs / HBM_PEAK_GBS, "traffic": None,
                               "algorithmic_bytes_per_point": sw_bytes, "launches": sw_all, "launches_timed": sw_calls,
                               "avg_launch_ms": sw_ms / max(sw_calls, 1), "points_per_launch": sw_pts / max(sw_calls, 1),
                               "binding_roof": {"bound": "fp64 issue", "exp_per_point": exp_per_pt,
                                                "exp_per_second": sw_pts * exp_per_pt / max(sw_ms * 1e-3, 1e-12),
                                                "note": "~21 fp64 VALU instructions per exp (fastmath.hpp): see profiles/r01_pmc_k_rt_sw_bb_fast.md"},
                               "timing": "HIP events around every %d-th batch's sweep launches inside the timed region" % max(1, args.profile_stride)}
        if h2d_ms is not None:
            out["pcie_inclusive"] = {"h2d_ms_per_step": h2d_ms, "bytes_per_step": len(SPECTRA_OF_JOB(args)) * nlay * nwav * 4,
                                     "value_not_overlapped": points / (dt + args.steps * h2d_ms * 1e-3), "unit": "wavenumber-points/s",
                                     "note": "the job's FLOAT spectra (each read once: a gas's spectrum is the target once and part of "
                                             "other gases' backgrounds) uploaded from pinned host memory in front of every step, not "
                                             "overlapped; the tools' reader streams a gas's files while the previous gas is prepared"}
        if lut_sharded is not None:
            out["lut_opt"] = lut_sharded
        elif world == 1 and not args.no_lut_opt and args.config == 1:
            out["lut_opt"] = lut_opt_bench(ctx, args.lut_opt_iterations, cpu=not args.no_cpu)
            out["lut_opt"]["shortwave"] = lut_opt_bench(ctx, args.lut_opt_iterations, sw=True)
        if world == 1 and not args.no_sw and args.config == 1:
            out["sw_find_g"] = sw_find_g_bench(ctx)
        if world == 1 and not args.no_e2e and args.config == 1:
            # the end-to-end number north_star asks for: the do_all_lw chain with the tools as fresh child processes at the
            # headline size against the CPU oracle chain (tools/e2e_bench.py); a failure there must not cost the headline line
            try:
                if args.config == 1:
                    job.close()
                torch.cuda.empty_cache()
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import e2e_bench
                out["e2e"] = e2e_bench.run(ctx, nwav=nwav, nlay=nlay)
            except Exception as exc:                                      # noqa: BLE001
                import traceback
                out["e2e"] = {"error": repr(exc), "traceback": traceback.format_exc()[-1500:]}
            # the shipped path of the headline job itself: the six-gas FSCK job by bin/reorder_spectrum + bin/find_g_points on files
            try:
                import fsck_tools_bench
                out["tools_fsck_job"] = fsck_tools_bench.run(ctx, nwav=nwav, nlay=nlay, tolerance=args.tolerance, nlines=args.nlines)
            except Exception as exc:                                      # noqa: BLE001
                import traceback
                out["tools_fsck_job"] = {"error": repr(exc), "traceback": traceback.format_exc()[-1500:]}
        if world == 1 and not args.no_cpu and args.config == 1:
            cb = cpu_baseline(args, args.cpu_sample, nlay, syn.SEED_BASE + 1, args.tolerance, args.tolerance_tolerance,
                              args.max_iterations, ctx)
            out["cpu_baseline"] = {"value": cb["points"] / cb["seconds"], "unit": "wavenumber-points/s",
                                   "cores": int(os.environ["OMP_NUM_THREADS"]), "kind": "port",
                                   "nwav": args.cpu_sample, "gases": cb["gases"], "gases_whose_search_threw": cb["failed_gases"], "ng": cb["ng"], "n_pass": cb["n_pass"],
                                   "n_pass_per_gas": cb["n_pass_per_gas"], "search_status": cb["status"],
                                   "seconds": cb["seconds"], "headline_over_sample_points": nwav / args.cpu_sample,
                                   "sample": "oracle/oracle_chain.c (C end to end: reorder + gas preparation + the reference's "
                                             "equipartition.cpp from oracle/_ref over the oracle's calc_error, OpenMP at the "
                                             "reference's sites), the same %d-gas job (merged double backgrounds, the first gas's "
                                             "Planck matrix kept) from the same generator at nwav=%d per gas: N_pass=%.1f in all "
                                             "(every request swept, as the reference does), %.1f s (reorder %.2f, preparation "
                                             "%.2f, search %.2f)"
                                             % (len(cb["gases"]), args.cpu_sample, cb["n_pass"], cb["seconds"],
                                                cb["stage_seconds"]["reorder"], cb["stage_seconds"]["preparation"],
                                                cb["stage_seconds"]["search"])}
    return out


def config4_main(args, ctx, dist, rank, world, barrier):
    """BASELINE configs[4]: optimize_lut on the shapes of the LW + SW training (8 scenarios x 50 profiles, x 3 zenith
    angles in the shortwave; ng = 64, 6 x 53 (T, p) grid, 12 H2O mole fractions: nx ~ 3e5), the training profiles sharded
    over the ranks: every rank holds its own 8 x 50 profiles (weak scaling in the profiles) and ONE all-reduce of
    [gradient, cost] per evaluation keeps the ranks' L-BFGS states identical.  A step = `--lut-opt-iterations` L-BFGS
    iterations of the longwave problem followed by as many of the shortwave one."""
    sharded = dist is not None
    res = {}
    t = {}
    for name, sw in (("longwave", False), ("shortwave", True)):
        barrier()
        t0 = time.perf_counter()
        res[name] = lut_opt_bench(ctx, args.lut_opt_iterations, sharded=sharded, rank=rank, world=world, sw=sw,
                                  cpu=(rank == 0 and not args.no_cpu))
        barrier()
        t[name] = time.perf_counter() - t0
    if rank != 0:
        return None
    its = sum(r["iterations"] for r in res.values())
    secs = sum(r["iterations"] / r["iters_per_s"] for r in res.values())
    return {"metric": "LUT-opt iterations/s", "value": its / secs, "unit": "iterations/s", "n_gpus": world, "steps": 1, "warmup": 0,
            "ms_per_step": secs * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "configs[4]: optimize_lut LW then SW, 8 scenarios x 50 profiles per rank (x 3 solar zenith "
                                   "angles in the SW), ng=64, nx=%d, profile-sharded with one all-reduce of [gradient, cost] "
                                   "per evaluation" % res["longwave"]["nx"],
                       "iterations_per_region": args.lut_opt_iterations},
            "roofline": res["longwave"].get("roofline"), "cpu_baseline": res["longwave"].get("cpu_baseline"),
            "longwave": res["longwave"], "shortwave": res["shortwave"],
            "setup_inclusive_seconds": t}


def launch_ranks(args):
    """`bench.py --gpus N` typed on its own (no launcher around it: WORLD_SIZE unset): start the N ranks as CHILD processes
    through torch.distributed.run, BEFORE this process makes any GPU call (it never does: it only relays), and hand rank 0's
    JSON line on.  Never an exec: a process that has touched the GPU must not be replaced."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    import uuid
    # ECCKD_RUN_ID: the identity rank-aware tools stamp their part files with (bin/find_g_points): unique per launch, so that a
    # part left by an earlier launch with the same configuration and port is never taken for this one's
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               ECCKD_RUN_ID=os.environ.get("ECCKD_RUN_ID", uuid.uuid4().hex))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + "\n")          # banners and warnings of the ranks: not on the JSON channel
    if proc.returncode != 0 or line is None:
        sys.stderr.write("bench.py: the %d-rank job failed (exit code %d)\n" % (args.gpus, proc.returncode))
        sys.exit(proc.returncode or 1)
    sys.stdout.write(line + "\n")
    sys.stdout.flush()
    sys.exit(0)


def dry_main(args):
    """--dry-run: the rendezvous, the path's one all-reduce (gloo on the host) and the JSON line - no device work.  What the
    CPU test of `--gpus N` runs (tests/test_bench_launcher.py); never a measurement: value is null."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    from ecckd_amd import shard
    dt, passes, cost, seen = shard.reduce_scalars(1.0 + rank, 1.0, 0.0, count=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "wavenumber-points/s (LW reorder+find_g)", "value": None, "unit": "wavenumber-points/s",
                          "n_gpus": world, "ranks_seen": seen, "gpus_requested": args.gpus, "dry_run": True,
                          "steps": args.steps, "warmup": args.warmup, "passes_all_ranks": passes, "max_elapsed": dt}))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)                         # does not return
    if args.dry_run:
        return dry_main(args)
    # host threads for the CPU baseline: the GPU box gives 16 cores per GPU; more OpenMP threads than that only spin
    ncores = min(16, len(os.sched_getaffinity(0)))
    os.environ.setdefault("OMP_NUM_THREADS", str(ncores))
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d ranks; reporting n_gpus = %d\n" % (args.gpus, world, world))
    dist = None
    use_dist = world > 1 or os.environ.get("ECCKD_BENCH_FORCE_DIST") == "1"   # the override exercises the RCCL path on one GPU
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from ecckd_amd import api, synthetic as syn

    ctx = api.Context(local_rank)
    dev = ctx.device
    nwav, nlay = args.nwav, args.nlay
    out = None

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    if args.config == 4:
        out = config4_main(args, ctx, dist, rank, world, barrier)
    else:
        out = find_g_main(args, ctx, dist, rank, world, barrier, use_dist)
        strong = args.strong_leg if args.strong_leg is not None else world > 1
        if args.config == 1 and strong:
            # the strong-scaling leg north_star describes: ONE configs[3] job (13 bands x 8 gases) dealt to all the ranks
            import copy
            a3 = copy.copy(args)
            a3.config = 3
            torch.cuda.empty_cache()
            leg = find_g_main(a3, ctx, dist, rank, world, barrier, use_dist)
            if rank == 0:
                out["strong_leg"] = {k: leg[k] for k in ("metric", "value", "unit", "n_gpus", "ranks_seen", "steps", "warmup",
                                                         "ms_per_step", "scaling", "config", "search")}
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    # RCCL leaves a version banner in the C stdio buffer of stdout; push it out first so that the JSON line
    # is the LAST line of stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
