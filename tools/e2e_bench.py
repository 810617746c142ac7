#!/usr/bin/env python3
"""The end-to-end number of BASELINE.json:north_star: the chain of test/do_all_lw.sh (:83-96) at the size the metric is quoted on.

  GPU : the command-line tools as FRESH CHILD PROCESSES on synthetic CKDMIP-format files - reorder_spectrum per gas,
        find_g_points, create_look_up_table, optimize_lut, run_ckd (raw and optimised model) - at nwav = 7.2e6, nlay = 54,
        four gases (water vapour as a look-up table in two mole fractions, three linear gases), the 13 narrow longwave bands
        of test/config.h:141-142, three idealised temperature columns, 50 training / evaluation profiles.  Every hand-over is
        a NetCDF file, every option a configuration key.  Stage times are process wall times (start-up, HIP initialisation
        and file I/O included).
  CPU : the oracle chain (oracle_chain.c with the reference's own equipartition.cpp, oracle_lut.c, oracle_ckd.c +
        oracle_adjoint.c under the library's L-BFGS, oracle_rt.c) on the host cores at TWO reduced sizes of the same generator;
        the stages whose work is proportional to the number of wavenumbers (reorder, find_g_points, create_look_up_table) are
        scaled to the full size with the exponent measured between the two sizes (reported; ~1), the others taken as they are.
  Agreement: at the larger reduced size the tools run too, and the two chains' g-point maps, tables and the heating rates of
        the evaluation profiles (RMS difference weighted as plot/calc_hr_error.m:1-23, K/day) are compared.

TEST / BENCH INFRASTRUCTURE: uses oracle/ as the checker and the CPU baseline only.  bench.py calls run(); standalone:
  python tools/e2e_bench.py [--nwav 7200000] [--cpu-nwav 131072] [--workdir DIR]
"""
import argparse
import ctypes as C
import json
import os
import shutil
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# name: (seed offset, column scale, surface mole fraction, concentration dependence)
GASES = {"h2o": (61, 40.0, 5e-3, "lut"), "co2": (63, 10.0, 4e-4, "linear"), "o3": (65, 4.0, 3e-6, "linear"), "ch4": (67, 2.0, 1.8e-6, "linear")}
FIND_G = dict(heating_rate_tolerance=0.05, max_iterations=60, tolerance_tolerance=0.01, flux_weight=0.0)      # find_g_points_lw.sh
OPT = dict(max_iterations=500, flux_weight=0.2, flux_profile_weight=0.0, broadband_weight=0.5, prior_error=4.0, convergence_criterion=0.0)
OPT_DEFAULTS = dict(spectral_boundary_weight=0.0, negative_od_penalty=1.0e4, pressure_weight_power=0.5, pressure_corr=0.5,
                    temperature_corr=0.5, conc_corr=0.5)                # optimize_lut.cpp:97-113, :185
NCOL_TRAIN = 50
X4 = 4.0                                                                # second water-vapour concentration


def bands():
    from ecckd_amd import synthetic as syn
    b1, b2 = syn.LW_NARROW_BANDS
    return np.asarray(b1, dtype=np.float64), np.asarray(b2, dtype=np.float64)


def _big_netcdf_file():
    """scipy's classic-file writer with the one thing it lacks for a 4.7 GB variable: the redundant 32-bit `vsize` header field
    of a variable of 2^31 bytes or more is written as 2^32 - 1, as the format prescribes for 64-bit-offset files (the data
    offsets are 64-bit already); such a variable has to be the last one of the file."""
    from scipy.io import netcdf_file

    class BigNc(netcdf_file):
        def _pack_int(self, value):
            if value >= 2 ** 31:
                self.fp.write(np.array(0xFFFFFFFF, ">u4").tobytes())
            else:
                super()._pack_int(value)

    return BigNc


def _write_spectrum(path, gas, p1, temps, wn, od32, vmr):
    """CKDMIP-layout spectrum file (classic 64-bit offset): the same optical depths in every temperature column."""
    netcdf_file = _big_netcdf_file()
    nlay, nwav = od32.shape
    ncol = len(temps)
    w = netcdf_file(str(path), "w", version=2)
    for dim, n in (("column", ncol), ("half_level", nlay + 1), ("level", nlay), ("wavenumber", nwav)):
        w.createDimension(dim, n)
    w.createVariable("pressure_hl", "d", ("column", "half_level"))[:] = np.tile(p1, (ncol, 1))
    w.createVariable("temperature_hl", "d", ("column", "half_level"))[:] = np.stack(temps)
    w.createVariable("wavenumber", "d", ("wavenumber",))[:] = wn
    w.createVariable("mole_fraction_fl", "d", ("column", "level"))[:] = np.full((ncol, nlay), vmr)
    w.createVariable("reference_surface_mole_fraction", "d", ())[...] = vmr
    v = w.createVariable("optical_depth", "f", ("column", "level", "wavenumber"))      # the big one: last
    big = od32.astype(">f4")                                     # one conversion, ncol copies
    for c in range(ncol):
        v[c] = big
    w.constituent_id = gas
    w.close()


def make_inputs(ctx, d, nwav, nlay, nlines=3000):
    """"present" spectra (1 column per gas), "idealised" spectra (3 temperature columns; water vapour also at X4 times its mole
    fraction), the line-by-line training fluxes of NCOL_TRAIN profiles (lbl.nc, by the device's LBL evaluator from scaled sums
    of the gases' spectra) and the same profiles for run_ckd (eval.nc)."""
    import torch
    from scipy.io import netcdf_file
    from ecckd_amd import api, synthetic as syn
    b1, b2 = bands()
    p1 = syn.pressure_grid(nlay)
    wn, dwn = syn.wavenumber_grid(nwav)
    wn_d, dwn_d = torch.as_tensor(wn, device=ctx.device), torch.as_tensor(dwn, device=ctx.device)
    t0 = syn.temperature_profile(p1)
    ideal_t = [t0 - 20.0, t0, t0 + 20.0]
    names = list(GASES)
    base_d = {}
    base = {}
    for g, (s, sc, vmr, _) in GASES.items():
        t = syn.optical_depth_lines(torch, p1, wn_d, syn.SEED_BASE + s, nlines=nlines, column_scale=sc, device=ctx.device)
        base_d[g] = t
        base[g] = (t.cpu().numpy(), vmr)
    t_files = time.perf_counter()
    for g in names:
        od, vmr = base[g]
        _write_spectrum(os.path.join(d, f"present_{g}.nc"), g, p1, [t0], wn, od, vmr)
        _write_spectrum(os.path.join(d, f"ideal_{g}.nc"), g, p1, ideal_t, wn, od, vmr)
    _write_spectrum(os.path.join(d, "ideal_h2o_x4.nc"), "h2o", p1, ideal_t, wn, base["h2o"][0] * np.float32(X4), base["h2o"][1] * X4)
    t_files = time.perf_counter() - t_files
    # training / evaluation profiles: temperatures round the standard profile, each gas scaled by its own factor
    rs = np.random.RandomState(20260501)
    ncol = NCOL_TRAIN
    T = np.stack([t0 + dt for dt in np.linspace(-14.0, 14.0, ncol)])
    amount = {g: (rs.uniform(0.4, 3.2, ncol) if g == "h2o" else rs.uniform(0.5, 2.0, ncol)) for g in names}
    _, begin, end = api.band_ranges(wn, b1, b2)
    bdn, bup = [], []
    for c in range(ncol):
        od = None
        for g in names:
            term = base_d[g].double() * float(amount[g][c])
            od = term if od is None else od + term
        dn, up = api.lbl_band_fluxes_lw(ctx, T[c], wn_d, dwn_d, od, begin, end)
        bdn.append(dn.T); bup.append(up.T)
        del od
    bdn, bup = np.stack(bdn), np.stack(bup)
    del base_d
    torch.cuda.empty_cache()
    vmr = np.stack([np.stack([np.full(nlay, base[g][1] * amount[g][c]) for g in names]) for c in range(ncol)])
    nband = len(b1)
    w = netcdf_file(os.path.join(d, "lbl.nc"), "w", version=2)
    for dim, n in (("column", ncol), ("half_level", nlay + 1), ("level", nlay), ("gas", len(names)), ("band", nband)):
        w.createDimension(dim, n)
    for name, dims, a in (("pressure_hl", ("column", "half_level"), np.tile(p1, (ncol, 1))), ("temperature_hl", ("column", "half_level"), T),
                          ("mole_fraction_fl", ("column", "gas", "level"), vmr), ("flux_dn_lw", ("column", "half_level"), bdn.sum(-1)),
                          ("flux_up_lw", ("column", "half_level"), bup.sum(-1)), ("band_flux_dn_lw", ("column", "half_level", "band"), bdn),
                          ("band_flux_up_lw", ("column", "half_level", "band"), bup), ("band_wavenumber1_lw", ("band",), b1),
                          ("band_wavenumber2_lw", ("band",), b2)):
        w.createVariable(name, "d", dims)[:] = a
    w.constituent_id = " ".join(names)
    w.close()
    w = netcdf_file(os.path.join(d, "eval.nc"), "w", version=2)
    for dim, n in (("column", ncol), ("half_level", nlay + 1), ("level", nlay)):
        w.createDimension(dim, n)
    w.createVariable("pressure_hl", "d", ("column", "half_level"))[:] = np.tile(p1, (ncol, 1))
    w.createVariable("temperature_hl", "d", ("column", "half_level"))[:] = T
    for i, g in enumerate(names):
        w.createVariable(g + "_mole_fraction_fl", "d", ("column", "level"))[:] = vmr[:, i, :]
    w.close()
    return dict(p1=p1, wn=wn, dwn=dwn, base=base, ncol=ncol, T=T, vmr=vmr, bdn=bdn, bup=bup, nlay=nlay, nwav=nwav, ideal_t=ideal_t,
                t0=t0, names=names, seconds_writing_spectra=t_files)


def hr_k_per_day(p1, flux_dn, flux_up):
    conv = -(9.80665 / 1004.0) / np.diff(p1) * 86400.0
    return conv[None, :] * (np.diff(flux_dn, axis=1) - np.diff(flux_up, axis=1))


def hr_rms_difference(p1, hr_a, hr_b):
    """plot/calc_hr_error.m:1-23: weights d(p^(1/3)) normalised per profile, mean over profiles."""
    w = np.diff((p1 / 100.0) ** (1.0 / 3.0))
    w = w / w.sum()
    return float(np.sqrt(np.sum(w[None, :] * (hr_a - hr_b) ** 2) / hr_a.shape[0]))


def gpu_chain(d, inp):
    from scipy.io import netcdf_file
    bindir = os.path.join(ROOT, "bin")
    b1, b2 = bands()
    names = inp["names"]
    secs, fixed = {}, {}
    per_process = []

    def run(stage, name, *args):
        t0 = time.perf_counter()
        r = subprocess.run([os.path.join(bindir, name), *[str(a) for a in args]], cwd=d, capture_output=True, text=True, timeout=3600,
                           env=dict(os.environ, ECCKD_LOG_TIMES="1"))
        dt = time.perf_counter() - t0
        secs[stage] = secs.get(stage, 0.0) + dt
        # the tool's own clock (ECCKD_LOG_TIMES: seconds since its start in front of every log line): last stamp = its run time
        stamps = re.findall(r"^\[\s*([0-9.]+)\] (.{0,48})", r.stdout, flags=re.M)
        rec = {"tool": name, "seconds": round(dt, 3), "last_log_stamp": float(stamps[-1][0]) if stamps else None}
        if stage == "reorder_spectrum":
            rec["log"] = ["%s %s" % st for st in stamps if "Band " not in st[1]]
        per_process.append(rec)
        if os.environ.get("ECCKD_E2E_LOGDIR"):                 # every tool's timed log, for looking at a stage from inside
            ld = os.path.join(os.environ["ECCKD_E2E_LOGDIR"], os.path.basename(os.path.normpath(d)))
            os.makedirs(ld, exist_ok=True)
            with open(os.path.join(ld, "%02d_%s.log" % (len(per_process), name)), "w") as lf:
                lf.write(r.stdout + r.stderr)
        if r.returncode != 0:
            raise RuntimeError(f"{name} failed ({r.returncode}): {r.stderr[-2000:]}")
        return r

    w1, w2 = " ".join("%g" % v for v in b1), " ".join("%g" % v for v in b2)
    for g in names:
        run("reorder_spectrum", "reorder_spectrum", f"input=present_{g}.nc", f"output=order_{g}.nc", f"wavenumber1={w1}", f"wavenumber2={w2}")
    with open(os.path.join(d, "find_g.cfg"), "w") as f:
        f.write("heating_rate_tolerance %g\nmax_iterations %d\ntolerance_tolerance %g\nflux_weight %g\naveraging_method transmission\n"
                "gases %s\n" % (FIND_G["heating_rate_tolerance"], FIND_G["max_iterations"], FIND_G["tolerance_tolerance"], FIND_G["flux_weight"],
                                " ".join(names)))
        for g in names:
            others = " ".join(f"present_{o}.nc" for o in names if o != g)
            f.write("\\begin %s\n input present_%s.nc\n reordering_input order_%s.nc\n background_input \"%s\"\n\\end %s\n" % (g, g, g, others, g))
    run("find_g_points", "find_g_points", "find_g.cfg", "output=gpoints.nc")
    with open(os.path.join(d, "lut.cfg"), "w") as f:
        f.write("input gpoints.nc\noutput raw_ckd.nc\ngases %s\n" % " ".join(names))
        for g in names:
            if GASES[g][3] == "lut":
                f.write("\\begin %s\n conc_dependence lut\n input \"ideal_%s.nc ideal_%s_x4.nc\"\n\\end %s\n" % (g, g, g, g))
            else:
                f.write("\\begin %s\n conc_dependence linear\n input ideal_%s.nc\n\\end %s\n" % (g, g, g))
    run("create_look_up_table", "create_look_up_table", "lut.cfg")
    r = run("optimize_lut", "optimize_lut", "input=raw_ckd.nc", "output=ckd.nc", "training_input=lbl.nc", *[f"{k}={v}" for k, v in OPT.items()])
    its = [l for l in r.stdout.splitlines() if "Iteration " in l]
    out = {"iterations": max(len(its) - 1, 0)}
    for tag, ckd in (("raw", "raw_ckd.nc"), ("optimised", "ckd.nc")):
        run("run_ckd", "run_ckd", f"ckd_model={ckd}", "input=eval.nc", f"output=fluxes_{tag}.nc")
        f = netcdf_file(os.path.join(d, f"fluxes_{tag}.nc"), "r", mmap=False)
        out[tag] = (f.variables["flux_dn_lw"][...].astype(np.float64), f.variables["flux_up_lw"][...].astype(np.float64))
        f.close()
    out["per_process"] = per_process
    return secs, out


def cpu_chain(ctx, d, inp):
    """The oracle chain; needs the GPU only for the L-BFGS vector arithmetic of the library's minimizer (the cost function and
    its gradient come from the oracle through ecckd_opt_set_evaluator)."""
    import pyoracle as o
    import ckd_synth
    from ecckd_amd import api, ncio, pipeline
    o.build()
    L = o.lib()
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    p1, wn, nlay, nwav, names = inp["p1"], inp["wn"], inp["nlay"], inp["nwav"], inp["names"]
    # the spectra files carry no d_wavenumber: every tool derives it from the grid as read_spectrum.cpp:55-65 does - half the
    # distance between a point's neighbours, and HALF of its neighbour's value at the two ends - and so does the oracle chain
    dwn = np.empty(nwav)
    dwn[1:-1] = 0.5 * (wn[2:] - wn[:-2])
    dwn[0], dwn[-1] = 0.5 * dwn[1], 0.5 * dwn[-2]
    b1, b2 = bands()
    nband = len(b1)
    t_hl = np.ascontiguousarray(inp["t0"])
    secs = {"reorder_spectrum": 0.0, "find_g_points": 0.0, "create_look_up_table": 0.0, "optimize_lut": 0.0, "run_ckd": 0.0}
    ref = os.path.join(ROOT, "oracle", "_ref", "libequipartition_ref.so").encode()
    L.orc_find_g_lw_chain_ex.restype = C.c_int
    cap = 2048
    planck_first = np.zeros((nlay + 1, nwav))
    per_gas, ranks = [], {}
    devnull, saved = os.open(os.devnull, os.O_WRONLY), os.dup(1)
    os.dup2(devnull, 1)                       # the reference search prints its progress to stdout
    try:
        for k, g in enumerate(names):
            od32 = np.ascontiguousarray(inp["base"][g][0], dtype=np.float32)
            # the merged background (read_merged_spectrum.cpp:135-166): the other gases' spectra summed in double
            bg64 = np.zeros((nlay, nwav))
            for o_ in names:
                if o_ != g:
                    bg64 += inp["base"][o_][0]
            bg64 = np.ascontiguousarray(bg64)
            L.orc_chain_set_background64(P(bg64))
            ng, st = np.zeros(nband, dtype=np.int32), np.zeros(nband, dtype=np.int32)
            cc, s3 = np.zeros(nband), np.zeros(3)
            rank = np.zeros(nwav, dtype=np.int32)
            r1, r2 = np.zeros(cap, dtype=np.int64), np.zeros(cap, dtype=np.int64)
            err, med, key = np.zeros(cap), np.zeros(cap), np.zeros(nwav)
            tol = np.full(nband, FIND_G["heating_rate_tolerance"])
            rc = L.orc_find_g_lw_chain_ex(ref, C.c_int(nlay), C.c_size_t(nwav), P(p1), P(t_hl), P(wn), P(dwn),
                                          od32.ctypes.data_as(C.POINTER(C.c_float)), None,
                                          C.c_double(0.5), C.c_int(nband), P(b1), P(b2), C.c_int(1),
                                          C.c_double(FIND_G["flux_weight"]), C.c_double(0.0), P(tol), C.c_double(FIND_G["tolerance_tolerance"]),
                                          C.c_int(FIND_G["max_iterations"]), C.c_int(1), ng.ctypes.data_as(C.POINTER(C.c_int)), P(cc),
                                          st.ctypes.data_as(C.POINTER(C.c_int)), P(s3), rank.ctypes.data_as(C.POINTER(C.c_int32)),
                                          C.c_int(cap), r1.ctypes.data_as(C.POINTER(C.c_int64)), r2.ctypes.data_as(C.POINTER(C.c_int64)),
                                          P(err), P(med), P(key), P(planck_first), C.c_int(1 if k == 0 else 2))
            if rc:
                raise RuntimeError("orc_find_g_lw_chain_ex failed: %d" % rc)
            secs["reorder_spectrum"] += s3[0]
            secs["find_g_points"] += s3[1] + s3[2]
            n = int(ng.sum())
            per_gas.append(dict(name=g, n_g_points=[int(v) for v in ng], rank1=r1[:n].copy(), rank2=r2[:n].copy(), error=err[:n].copy(),
                                sorting_variable=med[:n].astype(np.float32).astype(np.float64), comp_cost=cc.copy(), status=st.copy()))
            ranks[g] = rank
    finally:
        os.dup2(saved, 1)
        os.close(devnull)
    t0 = time.perf_counter()
    ng_all, band_number, g_min, g_max = o.overlap_g_points([g["n_g_points"] for g in per_gas], [g["sorting_variable"] for g in per_gas])
    gas_gp = []
    for g in per_gas:                                        # SingleGasData::store_g_points (single_gas_data.h:56-62)
        gp = np.full(nwav, -1, dtype=np.int64)
        order = np.argsort(ranks[g["name"]], kind="stable")
        for i, (a, b) in enumerate(zip(g["rank1"], g["rank2"])):
            gp[order[a:b + 1]] = i
        gas_gp.append(gp)
    g_point = np.full(nwav, -1, dtype=np.int32)              # find_g_points.cpp:1459-1475
    for ig in range(ng_all):
        found = np.ones(nwav, dtype=bool)
        for k in range(len(per_gas)):
            found &= (gas_gp[k] >= g_min[k][ig]) & (gas_gp[k] <= g_max[k][ig])
        g_point[found] = ig
    secs["find_g_points"] += time.perf_counter() - t0

    # ---- create_look_up_table (create_look_up_table.cpp:225-606) ----
    t0 = time.perf_counter()
    ng = ng_all
    model = dict(gases=[], iband_per_g=np.asarray(band_number, dtype=np.int32), log_pressure=np.log(0.5 * (p1[1:] + p1[:-1])),
                 nband=nband, ng=ng)
    tfl_rows = None
    for name in names:
        conc = GASES[name][3]
        files = [(name, 1.0), (name, X4)] if conc == "lut" else [(name, 1.0)]
        tabs = [np.zeros((len(files), 3, nlay, ng)) for _ in range(3)]
        vmrs = []
        tfl_rows = np.zeros((3, nlay))
        for ic, (gname, factor) in enumerate(files):
            od = inp["base"][gname][0] * np.float32(factor)
            ref_vmr = inp["base"][gname][1] * factor
            for it, t in enumerate(inp["ideal_t"]):
                t_fl = (t[:-1] * p1[:-1] + t[1:] * p1[1:]) / (p1[:-1] + p1[1:])          # :310-311
                weight = o.planck_function(t_fl, wn, dwn)                               # :323
                k, kmin, kmax, _ = o.average_optical_depth_to_g_point(ng, ref_vmr, p1, g_point, od.astype(np.float64), weight, "transmission")
                for tab, v in zip(tabs, (k, kmin, kmax)):
                    tab[ic, it] = v
                tfl_rows[it] = t_fl
            vmrs.append(ref_vmr)
        sq = (lambda a: a) if conc == "lut" else (lambda a: a[0])
        gd = dict(name=name, conc=conc, active=True, molar_abs=sq(tabs[0]), min_molar_abs=sq(tabs[1]), max_molar_abs=sq(tabs[2]))
        if conc == "lut":
            gd["vmr"] = np.array(vmrs)
        model["gases"].append(gd)
    model["temperature"] = tfl_rows
    model["temperature_planck"] = np.arange(120.0, 351.0)
    model["planck_function"] = o.planck_lut(ng, model["temperature_planck"], g_point, wn, dwn)
    model["wavenumber1"] = 10.0 * np.arange(0, 326, dtype=np.float64)
    model["wavenumber2"] = 10.0 * np.arange(1, 327, dtype=np.float64)
    model["gpoint_fraction"] = o.gpoint_fraction(ng, g_point, wn, dwn, model["wavenumber1"], model["wavenumber2"])
    secs["create_look_up_table"] = time.perf_counter() - t0
    # The raw model reaches optimize_lut and run_ckd through the CKD-definition FILE, whose tables, temperatures, pressures and
    # mole fractions are FLOAT variables (CkdModel::write, ckd_model.cpp:318-326, :352, :418-445; read back by CkdModel::read,
    # :32-286): the oracle chain rounds where the file rounds
    f32 = lambda a: np.asarray(a).astype(np.float32).astype(np.float64)
    raw = dict(model, gases=[dict(g, **{k: f32(g[k]) for k in ("molar_abs", "min_molar_abs", "max_molar_abs", "vmr") if k in g})
                             for g in model["gases"]])
    raw["planck_function"] = f32(model["planck_function"])
    raw["temperature"] = f32(model["temperature"])
    raw["log_pressure"] = np.log(f32(np.exp(model["log_pressure"])))
    raw["temperature_planck"] = f32(model["temperature_planck"])

    # ---- optimize_lut (optimize_lut.cpp:60-330): the library's L-BFGS over the oracle's cost function and gradient ----
    t0 = time.perf_counter()
    s = ncio.read_lbl_fluxes(os.path.join(d, "lbl.nc"), names, ctx=ctx)
    raw["iband_per_g"] = pipeline.iband_per_g(raw, s["band_wavenumber1"], s["band_wavenumber2"])
    scene = pipeline._scene_for_optimizer(s, False)
    cfg = dict(OPT_DEFAULTS, **{k: OPT[k] for k in ("flux_weight", "flux_profile_weight", "broadband_weight", "prior_error")})
    cfg["cap_relative_linear"] = 0.8
    orc = ckd_synth.Oracle(o, raw, [scene], cfg)

    def cost_grad(x):
        J, g = orc.cost_grad_rt(x)
        Jb, gb = orc.cost_prior(x, cfg["prior_error"])
        g = g + gb
        g[np.abs(g) < 1.0e-80] = 0.0
        return J + Jb, np.where(x > -1.0e20, g, 0.0)

    opt = api.Optimizer(ctx, raw, [scene], **cfg)
    opt.set_evaluator(cost_grad)
    res = opt.minimize(max_iterations=OPT["max_iterations"], convergence_criterion=OPT["convergence_criterion"], bounded=True)
    optimised = dict(raw, gases=[dict(g) for g in raw["gases"]])
    for i, g in enumerate(optimised["gases"]):
        g["molar_abs"] = opt.coefficients(res["x"], i, np.asarray(g["molar_abs"]).shape).astype(np.float32).astype(np.float64)
    opt.close()
    secs["optimize_lut"] = time.perf_counter() - t0

    # ---- run_ckd (run_ckd.cpp:27-373) on the evaluation profiles ----
    t0 = time.perf_counter()
    ev = dict(pressure_hl=np.tile(p1, (inp["ncol"], 1)), temperature_hl=inp["T"], vmr_fl=inp["vmr"], gas_present=None)
    out = {"iterations": res["iterations"], "status": res["status"], "ng": ng, "g_point": g_point, "per_gas": per_gas,
           "models": {"raw": raw, "optimised": optimised}, "cfg": cfg}
    for tag, m in (("raw", raw), ("optimised", optimised)):
        oo = ckd_synth.Oracle(o, m, [ev], cfg)
        f = oo.fluxes(oo.x0, ev)                              # (ncol, 2, nhl, ng)
        out[tag] = (f32(f[:, 0].sum(-1)), f32(f[:, 1].sum(-1)))   # FLOAT in run_ckd's file (run_ckd.cpp:221-230)
    secs["run_ckd"] = time.perf_counter() - t0
    return secs, out


def oracle_fluxes_of_model(model, inp, cfg):
    """Broadband fluxes (DOUBLE, in memory) of the evaluation profiles from a CKD model by the oracle's evaluator (oracle_ckd.c):
    what run_ckd computes before it writes its FLOAT file (run_ckd.cpp:221-230, :355-356)."""
    import pyoracle as o
    import ckd_synth
    ev = dict(pressure_hl=np.tile(inp["p1"], (inp["ncol"], 1)), temperature_hl=inp["T"], vmr_fl=inp["vmr"], gas_present=None)
    oo = ckd_synth.Oracle(o, model, [ev], cfg)
    f = oo.fluxes(oo.x0, ev)
    return f[:, 0].sum(-1), f[:, 1].sum(-1)


def compare_models(a, b):
    """Table by table: how many coefficients of two CKD models differ and by how much (relative), and which one differs most."""
    out = {}
    for ga, gb in zip(a["gases"], b["gases"]):
        ka, kb = np.asarray(ga["molar_abs"], dtype=np.float64), np.asarray(gb["molar_abs"], dtype=np.float64)
        if ka.shape != kb.shape:
            out[ga["name"]] = {"shapes": [list(ka.shape), list(kb.shape)]}
            continue
        rel = np.abs(ka - kb) / np.maximum(np.maximum(np.abs(ka), np.abs(kb)), 1e-300)
        rel[(ka == 0) & (kb == 0)] = 0.0
        worst = np.unravel_index(int(np.argmax(rel)), rel.shape)
        out[ga["name"]] = {"coefficients": int(ka.size), "different": int((ka != kb).sum()), "max_relative_difference": float(rel.max()),
                           "worst_index_(conc,)temperature,pressure,g": [int(v) for v in worst],
                           "different_by_more_than_1e-6": int((rel > 1e-6).sum())}
    return out


SIZE_STAGES = ("reorder_spectrum", "find_g_points", "create_look_up_table")      # work proportional to nwav


def host_memory():
    """MemAvailable / Cached / Dirty / Writeback of /proc/meminfo in GB and the cgroup's limit: the tools read their inputs
    from the page cache, so the state of the host's memory is part of the measurement."""
    out = {}
    try:
        for line in open("/proc/meminfo"):
            k, v = line.split(":")
            if k in ("MemTotal", "MemAvailable", "Cached", "Dirty", "Writeback"):
                out[k + "_GB"] = round(int(v.split()[0]) / 1048576.0, 2)
        for f in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
            if os.path.exists(f):
                v = open(f).read().strip()
                out["cgroup_limit_GB"] = v if v == "max" else round(int(v) / 2.0**30, 2)
                break
        for f in ("/sys/fs/cgroup/memory.current", "/sys/fs/cgroup/memory/memory.usage_in_bytes"):
            if os.path.exists(f):
                out["cgroup_current_GB"] = round(int(open(f).read().strip()) / 2.0**30, 2)
                break
    except Exception as exc:       # diagnostics only
        out["error"] = repr(exc)
    return out


def run(ctx, nwav=7_200_000, nlay=54, cpu_nwav=(1 << 17, 1 << 18), workdir=None, keep=False):
    from ecckd_amd import ncio
    top = workdir or tempfile.mkdtemp(prefix="ecckd_e2e_")
    os.makedirs(top, exist_ok=True)
    ncores = int(os.environ.get("OMP_NUM_THREADS", "1"))
    out = {}
    try:
        # ---- the tools at full size ----
        d = os.path.join(top, "full")
        os.makedirs(d, exist_ok=True)
        # the inputs are 19 spectra of nlay x nwav FLOATs (4 present-day, 4 x 3 idealised temperatures, 3 of h2o at 4 x the
        # mole fraction): say so at once if the work directory has no room for them
        need = 19 * nlay * nwav * 4 * 1.1
        free = shutil.disk_usage(top).free
        if free < need:
            raise RuntimeError("e2e: %s has %.1f GB free, the input spectra need %.1f GB" % (top, free / 1e9, need / 1e9))
        t0 = time.perf_counter()
        inp = make_inputs(ctx, d, nwav, nlay)
        setup_s = time.perf_counter() - t0
        p1 = inp["p1"]
        hr_lbl_full = hr_k_per_day(p1, inp["bdn"].sum(-1), inp["bup"].sum(-1))
        write_s = inp["seconds_writing_spectra"]
        del inp["base"]
        # the chain starts from files at rest: the write-back of the 30 GB just written is part of the setup, not of stage 1
        mem_before_sync = host_memory()
        t0 = time.perf_counter()
        os.sync()
        sync_s = time.perf_counter() - t0
        mem_at_start = host_memory()
        g_secs, g_out = gpu_chain(d, inp)
        gpf = ncio.read_g_points(os.path.join(d, "gpoints.nc"))
        ng_full = int(gpf["g_point"].max()) + 1
        del gpf
        hr_full = {k: hr_rms_difference(p1, hr_k_per_day(p1, *g_out[k]), hr_lbl_full) for k in ("raw", "optimised")}
        input_bytes = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d) if f.startswith(("present_", "ideal_")))
        if not keep:
            shutil.rmtree(d, ignore_errors=True)
        # ---- the oracle chain (and the tools again) at the reduced sizes ----
        cpu = {}
        agreement = None
        for n in cpu_nwav:
            dn = os.path.join(top, "cpu_%d" % n)
            os.makedirs(dn, exist_ok=True)
            inp_n = make_inputs(ctx, dn, n, nlay)
            c_secs, c_out = cpu_chain(ctx, dn, inp_n)
            cpu[n] = c_secs
            if n == max(cpu_nwav):
                s_secs, s_out = gpu_chain(dn, inp_n)
                gp_tools = ncio.read_g_points(os.path.join(dn, "gpoints.nc"))["g_point"]
                hr_t = {k: hr_k_per_day(p1, *s_out[k]) for k in ("raw", "optimised")}
                hr_c = {k: hr_k_per_day(p1, *c_out[k]) for k in ("raw", "optimised")}
                hr_l = hr_k_per_day(p1, inp_n["bdn"].sum(-1), inp_n["bup"].sum(-1))
                # ... and in DOUBLE: run_ckd's file holds FLOAT fluxes (run_ckd.cpp:221-230) - 3e-5 W m-2 of rounding on a 400 W m-2
                # flux is 0.03 K/day in a top layer of 1 Pa - so the two chains' MODELS (the tools' from their CKD files) are
                # evaluated by one evaluator in double
                in_double, tables = {}, {}
                for tag, fname in (("raw", "raw_ckd.nc"), ("optimised", "ckd.nc")):
                    mt = ncio.read_ckd_model(os.path.join(dn, fname))
                    ft = oracle_fluxes_of_model(mt, inp_n, c_out["cfg"])
                    fc = oracle_fluxes_of_model(c_out["models"][tag], inp_n, c_out["cfg"])
                    in_double[tag] = hr_rms_difference(p1, hr_k_per_day(p1, *ft), hr_k_per_day(p1, *fc))
                    tables[tag] = compare_models(mt, c_out["models"][tag])
                agreement = {"nwav": n, "ng_tools": int(gp_tools.max()) + 1, "ng_oracle": int(c_out["ng"]),
                             "g_point_maps_identical": bool(np.array_equal(gp_tools, c_out["g_point"])),
                             "wavenumbers_in_another_g_point": int((gp_tools != c_out["g_point"]).sum()) if gp_tools.shape == c_out["g_point"].shape else None,
                             "iterations_tools": s_out["iterations"], "iterations_oracle": c_out["iterations"],
                             "hr_rms_difference_K_per_day_tools_vs_oracle": {k: hr_rms_difference(p1, hr_t[k], hr_c[k]) for k in hr_t},
                             "hr_rms_difference_K_per_day_of_the_two_chains_models_in_double": in_double,
                             "coefficient_tables_tools_vs_oracle": tables,
                             "hr_rms_error_against_lbl_K_per_day": {"tools_" + k: hr_rms_difference(p1, hr_t[k], hr_l) for k in hr_t}
                                                                   | {"oracle_" + k: hr_rms_difference(p1, hr_c[k], hr_l) for k in hr_c},
                             "tools_seconds_at_this_size": {k: round(v, 3) for k, v in s_secs.items()}}
            if not keep:
                shutil.rmtree(dn, ignore_errors=True)
        n_lo, n_hi = min(cpu_nwav), max(cpu_nwav)
        expo = {}
        scaled = {}
        for k in cpu[n_hi]:
            if k in SIZE_STAGES and n_lo != n_hi and cpu[n_lo][k] > 0 and cpu[n_hi][k] > 0:
                expo[k] = float(np.log(cpu[n_hi][k] / cpu[n_lo][k]) / np.log(n_hi / n_lo))
            # scaled LINEARLY (the work per wavenumber does not grow with the grid; the measured exponent is reported beside it)
            scaled[k] = cpu[n_hi][k] * (nwav / n_hi if k in SIZE_STAGES else 1.0)
        g_tot, c_tot = sum(g_secs.values()), sum(scaled.values())
        nproc = 2 + len(inp["names"]) + 3
        out = {
            "workload": "do_all_lw chain (test/do_all_lw.sh:83-96), the tools as fresh child processes on synthetic CKDMIP-format files: "
                        "%d gases (%s; h2o as a look-up table in 2 mole fractions), 13 narrow bands, nwav=%d, nlay=%d, 3 idealised "
                        "temperature columns, %d training / evaluation profiles; find_g_points tolerance %g K/d, optimize_lut %d iterations"
                        % (len(inp["names"]), " ".join(inp["names"]), nwav, nlay, NCOL_TRAIN, FIND_G["heating_rate_tolerance"], OPT["max_iterations"]),
            "gpu_tools_seconds": {k: round(v, 3) for k, v in g_secs.items()}, "gpu_tools_total_seconds": round(g_tot, 3),
            "tool_processes": nproc, "input_spectra_bytes": input_bytes, "ng": ng_full,
            "hr_rms_error_against_lbl_K_per_day": hr_full,
            "cpu_oracle_seconds": {str(n): {k: round(v, 3) for k, v in cpu[n].items()} for n in cpu},
            "cpu_scaling_exponent_between_the_two_sizes": expo,
            "cpu_oracle_seconds_scaled_to_full_size": {k: round(v, 2) for k, v in scaled.items()},
            "cpu_oracle_total_seconds_scaled": round(c_tot, 2), "cpu_cores": ncores,
            # the CPU side is the oracle chain at the larger reduced size with its size-proportional stages scaled LINEARLY to the full size (the
            # exponent measured between the two reduced sizes is beside it: > 1 for find_g_points, so this understates the CPU)
            "speedup_vs_linearly_scaled_cpu": c_tot / g_tot,
            "speedup_per_stage_vs_linearly_scaled_cpu": {k: scaled[k] / g_secs[k] for k in scaled if g_secs.get(k)},
            "agreement_at_reduced_size": agreement,
            "setup_seconds_not_timed": {"generating_and_writing_inputs": round(setup_s, 1), "of_which_writing_spectra_files": round(write_s, 1),
                                        "sync_after_writing": round(sync_s, 1)},
            "tool_processes_seconds": g_out.get("per_process"),
            "host_memory": {"after_writing": mem_before_sync, "at_chain_start": mem_at_start},
            "note": "each tool is a separate process: its wall time includes process start (~0.25 s of HIP initialisation), reading its "
                    "NetCDF inputs from the page cache and writing its outputs; CPU: oracle chain (restated reference + the reference's "
                    "equipartition.cpp) on %d cores, size-proportional stages scaled linearly from nwav=%d" % (ncores, n_hi),
        }
    finally:
        if not keep and workdir is None:
            shutil.rmtree(top, ignore_errors=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nwav", type=int, default=7_200_000)
    ap.add_argument("--nlay", type=int, default=54)
    ap.add_argument("--cpu-nwav", type=int, nargs="+", default=[1 << 17, 1 << 18])
    ap.add_argument("--workdir", default=None)
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()
    os.environ.setdefault("OMP_NUM_THREADS", str(min(16, len(os.sched_getaffinity(0)))))
    from ecckd_amd import api
    with api.Context(0) as ctx:
        res = run(ctx, args.nwav, args.nlay, tuple(args.cpu_nwav), args.workdir, args.keep)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
