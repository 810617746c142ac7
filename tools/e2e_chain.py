#!/usr/bin/env python3
"""The chain of test/do_all_lw.sh (:83-96) on synthetic CKDMIP-format files, run twice and compared:

  GPU : the six command-line tools (bin/reorder_spectrum, find_g_points, create_look_up_table, optimize_lut, run_ckd x 2),
        every hand-over a NetCDF file, every option a configuration key - what a user of the reference's scripts runs;
  CPU : the oracle chain - the restated reference code of oracle/ composed as the reference's main() functions compose it
        (reorder + gas preparation + the reference's own equipartition.cpp in C end to end: oracle_chain.c; g-point
        averaging: oracle_lut.c; cost function: oracle_ckd.c with the hand-written reverse mode of oracle_adjoint.c under
        the library's L-BFGS; evaluation: oracle_ckd.c + oracle_rt.c), OpenMP at the reference's sites.

Reported: wall time per stage for both, and how far the two chains' results are apart - the g points (identical or not), the
raw and optimised look-up tables, and the heating rates of the evaluation profiles (RMS difference weighted as
plot/calc_hr_error.m:1-23, K/day).  TEST / BENCH INFRASTRUCTURE: uses oracle/ as the checker and the CPU baseline only.

  python tools/e2e_chain.py [--nwav N] [--nlay L] [--workdir DIR] [--json]
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

BANDS = (np.array([0.0, 1300.0]), np.array([1300.0, 3260.0]))
GASES = {"h2o": (61, 40.0, 5e-3), "co2": (63, 10.0, 4e-4)}          # seed offset, column scale, mole fraction
FIND_G = dict(heating_rate_tolerance=0.1, max_iterations=30, tolerance_tolerance=0.02, flux_weight=0.02)
OPT = dict(max_iterations=60, flux_weight=0.2, flux_profile_weight=0.05, broadband_weight=0.5, prior_error=8.0,
           convergence_criterion=0.0)
OPT_DEFAULTS = dict(spectral_boundary_weight=0.0, negative_od_penalty=1.0e4, pressure_weight_power=0.5, pressure_corr=0.5,
                    temperature_corr=0.5, conc_corr=0.5)                # optimize_lut.cpp:97-113, :185


def make_inputs(ctx, d, nwav, nlay):
    """"present" spectra of two gases (1 column), "idealised" spectra (3 temperature columns, water vapour also at 4x),
    line-by-line training fluxes of three columns (lbl.nc, from the LBL stand-in) and their profiles for run_ckd (eval.nc)."""
    import torch
    from scipy.io import netcdf_file
    from ecckd_amd import api, ncio, synthetic as syn
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=ctx.device)
    p1 = syn.pressure_grid(nlay)
    wn, _ = syn.wavenumber_grid(nwav)
    base = {g: (syn.optical_depth_lines(torch, p1, wn, syn.SEED_BASE + s, nlines=3000, column_scale=sc, device=ctx.device).cpu().numpy(), v)
            for g, (s, sc, v) in GASES.items()}

    def write(path, gas, temps, factor=1.0):
        ncol = len(temps)
        w = netcdf_file(str(path), "w", version=2)
        for dim, n in (("column", ncol), ("half_level", nlay + 1), ("level", nlay), ("wavenumber", nwav)):
            w.createDimension(dim, n)
        od, vmr = base[gas]
        w.createVariable("pressure_hl", "d", ("column", "half_level"))[:] = np.tile(p1, (ncol, 1))
        w.createVariable("temperature_hl", "d", ("column", "half_level"))[:] = np.stack(temps)
        w.createVariable("wavenumber", "d", ("wavenumber",))[:] = wn
        w.createVariable("mole_fraction_fl", "d", ("column", "level"))[:] = np.full((ncol, nlay), vmr * factor)
        w.createVariable("optical_depth", "f", ("column", "level", "wavenumber"))[:] = np.tile((od * np.float32(factor))[None], (ncol, 1, 1))
        w.createVariable("reference_surface_mole_fraction", "d", ())[...] = vmr * factor
        w.constituent_id = gas
        w.close()

    t0 = syn.temperature_profile(p1)
    ideal_t = [t0 - 20.0, t0, t0 + 20.0]
    for g in base:
        write(os.path.join(d, f"present_{g}.nc"), g, [t0])
        write(os.path.join(d, f"ideal_{g}.nc"), g, ideal_t)
    write(os.path.join(d, "ideal_h2o_x4.nc"), "h2o", ideal_t, factor=4.0)
    ncol = 3
    T = np.stack([t0 - 8.0, t0 + 3.0, t0 + 11.0])
    amount = {"h2o": np.array([0.7, 1.5, 3.0]), "co2": np.array([1.0, 2.0, 0.5])}
    sel = [np.nonzero((wn >= a) & (wn < b + (b == 3260.0)))[0] for a, b in zip(*BANDS)]
    begin, end = [int(s[0]) for s in sel], [int(s[-1]) for s in sel]
    dwn = ncio.read_spectrum(os.path.join(d, "present_h2o.nc"))["d_wavenumber_cm_1"]
    bdn, bup = [], []
    for c in range(ncol):
        od = sum(base[g][0].astype(np.float64) * amount[g][c] for g in base)
        dn, up = api.lbl_band_fluxes_lw(ctx, T[c], dev(wn), dev(dwn), dev(od), begin, end)
        bdn.append(dn.T); bup.append(up.T)
    bdn, bup = np.stack(bdn), np.stack(bup)
    vmr = np.stack([np.stack([np.full(nlay, base[g][1] * amount[g][c]) for g in ("h2o", "co2")]) for c in range(ncol)])
    w = netcdf_file(os.path.join(d, "lbl.nc"), "w", version=2)
    for dim, n in (("column", ncol), ("half_level", nlay + 1), ("level", nlay), ("gas", 2), ("band", 2)):
        w.createDimension(dim, n)
    for name, dims, a in (("pressure_hl", ("column", "half_level"), np.tile(p1, (ncol, 1))), ("temperature_hl", ("column", "half_level"), T),
                          ("mole_fraction_fl", ("column", "gas", "level"), vmr), ("flux_dn_lw", ("column", "half_level"), bdn.sum(-1)),
                          ("flux_up_lw", ("column", "half_level"), bup.sum(-1)), ("band_flux_dn_lw", ("column", "half_level", "band"), bdn),
                          ("band_flux_up_lw", ("column", "half_level", "band"), bup), ("band_wavenumber1_lw", ("band",), BANDS[0]),
                          ("band_wavenumber2_lw", ("band",), BANDS[1])):
        w.createVariable(name, "d", dims)[:] = a
    w.constituent_id = "h2o co2"
    w.close()
    w = netcdf_file(os.path.join(d, "eval.nc"), "w", version=2)
    for dim, n in (("column", ncol), ("half_level", nlay + 1), ("level", nlay)):
        w.createDimension(dim, n)
    w.createVariable("pressure_hl", "d", ("column", "half_level"))[:] = np.tile(p1, (ncol, 1))
    w.createVariable("temperature_hl", "d", ("column", "half_level"))[:] = T
    for i, g in enumerate(("h2o", "co2")):
        w.createVariable(g + "_mole_fraction_fl", "d", ("column", "level"))[:] = vmr[:, i, :]
    w.close()
    return dict(p1=p1, wn=wn, dwn=dwn, base=base, ncol=ncol, T=T, vmr=vmr, bdn=bdn, bup=bup, nlay=nlay, nwav=nwav, ideal_t=ideal_t, t0=t0)


def hr_k_per_day(p1, flux_dn, flux_up):
    """Heating rates (ncol, nlay) in K/day of broadband fluxes (ncol, nhl) (heating_rate.h:30-50)."""
    conv = -(9.80665 / 1004.0) / np.diff(p1) * 86400.0
    return conv[None, :] * (np.diff(flux_dn, axis=1) - np.diff(flux_up, axis=1))


def hr_rms_difference(p1, hr_a, hr_b):
    """plot/calc_hr_error.m:1-23: weights d(p^(1/3)) normalised per profile, mean over profiles."""
    w = np.diff((p1 / 100.0) ** (1.0 / 3.0))
    w = w / w.sum()
    return float(np.sqrt(np.sum(w[None, :] * (hr_a - hr_b) ** 2) / hr_a.shape[0]))


# ---------------------------------------------------------------------------------------------------------------------
def write_netcdf4_spectra(d):
    """The five spectra once more as NetCDF-4 (*.h5: chunks of 1 x 1 x 262 144 FLOATs, shuffle + deflate 2), what the CKDMIP files
    and the reference's scripts use; written through the HDF5 library (tests/h5_fixture.py), not timed."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import h5_fixture as h5
    from scipy.io import netcdf_file
    if not h5.available():
        return False
    for name in [f"present_{g}" for g in GASES] + [f"ideal_{g}" for g in GASES] + ["ideal_h2o_x4"]:
        f = netcdf_file(os.path.join(d, name + ".nc"), "r", mmap=False)
        v = {k: f.variables[k][...] for k in ("pressure_hl", "temperature_hl", "wavenumber", "mole_fraction_fl", "optical_depth",
                                               "reference_surface_mole_fraction")}
        gas = f.constituent_id.decode() if isinstance(f.constituent_id, bytes) else f.constituent_id
        f.close()
        nwav = v["wavenumber"].size
        h5.write(os.path.join(d, name + ".h5"), {
            "pressure_hl": (v["pressure_hl"], "f8", None, None), "temperature_hl": (v["temperature_hl"], "f8", None, None),
            "wavenumber": (v["wavenumber"], "f8", (min(nwav, 262144),), None), "mole_fraction_fl": (v["mole_fraction_fl"], "f8", None, None),
            "reference_surface_mole_fraction": (float(v["reference_surface_mole_fraction"]), "f8", None, None),
            "optical_depth": (v["optical_depth"], "f4", (1, 1, min(nwav, 262144)), None)}, {"constituent_id": gas})
    return True


def gpu_chain(d, inp, ext="nc"):
    """ext: the spectra the tools read are the classic files (nc) or their NetCDF-4 twins (h5); everything the tools write is
    classic either way."""
    from scipy.io import netcdf_file
    bindir = os.path.join(ROOT, "bin")
    secs = {}

    def run(stage, name, *args):
        t0 = time.perf_counter()
        r = subprocess.run([os.path.join(bindir, name), *[str(a) for a in args]], cwd=d, capture_output=True, text=True, timeout=3600)
        secs[stage] = secs.get(stage, 0.0) + time.perf_counter() - t0
        if r.returncode != 0:
            raise RuntimeError(f"{name} failed ({r.returncode}): {r.stderr[-2000:]}")
        return r

    for g in GASES:
        run("reorder_spectrum", "reorder_spectrum", f"input=present_{g}.{ext}", f"output=order_{g}.nc", "wavenumber1=0 1300", "wavenumber2=1300 3260")
    with open(os.path.join(d, "find_g.cfg"), "w") as f:
        f.write("heating_rate_tolerance %g\nmax_iterations %d\ntolerance_tolerance %g\nflux_weight %g\naveraging_method transmission\n"
                "gases h2o co2\n"
                "\\begin h2o\n input present_h2o.%s\n reordering_input order_h2o.nc\n background_input present_co2.%s\n\\end h2o\n"
                "\\begin co2\n input present_co2.%s\n reordering_input order_co2.nc\n background_input present_h2o.%s\n\\end co2\n"
                % (FIND_G["heating_rate_tolerance"], FIND_G["max_iterations"], FIND_G["tolerance_tolerance"], FIND_G["flux_weight"], ext, ext, ext, ext))
    run("find_g_points", "find_g_points", "find_g.cfg", "output=gpoints.nc")
    with open(os.path.join(d, "lut.cfg"), "w") as f:
        f.write("input gpoints.nc\noutput raw_ckd.nc\ngases h2o co2\n"
                "\\begin h2o\n conc_dependence lut\n input \"ideal_h2o.%s ideal_h2o_x4.%s\"\n\\end h2o\n"
                "\\begin co2\n conc_dependence linear\n input ideal_co2.%s\n\\end co2\n" % (ext, ext, ext))
    run("create_look_up_table", "create_look_up_table", "lut.cfg")
    r = run("optimize_lut", "optimize_lut", "input=raw_ckd.nc", "output=ckd.nc", "training_input=lbl.nc", *[f"{k}={v}" for k, v in OPT.items()])
    its = [l for l in r.stdout.splitlines() if l.startswith("Iteration ")]
    out = {"iterations": len(its) - 1, "optimize_log_tail": r.stdout.strip().splitlines()[-1]}
    for tag, ckd in (("raw", "raw_ckd.nc"), ("optimised", "ckd.nc")):
        run("run_ckd", "run_ckd", f"ckd_model={ckd}", "input=eval.nc", f"output=fluxes_{tag}.nc")
        f = netcdf_file(os.path.join(d, f"fluxes_{tag}.nc"), "r", mmap=False)
        out[tag] = (f.variables["flux_dn_lw"][...].astype(np.float64), f.variables["flux_up_lw"][...].astype(np.float64))
        f.close()
    return secs, out


# ---------------------------------------------------------------------------------------------------------------------
def cpu_chain(ctx, d, inp, threads):
    """The oracle chain; needs the GPU only for the L-BFGS vector arithmetic of the library's minimizer (the cost function and
    its gradient come from the oracle through ecckd_opt_set_evaluator)."""
    import pyoracle as o
    import ckd_synth
    from ecckd_amd import api, ncio, pipeline
    o.build()
    L = o.lib()
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    p1, wn, dwn, nlay, nwav = inp["p1"], inp["wn"], inp["dwn"], inp["nlay"], inp["nwav"]
    t_hl = np.ascontiguousarray(inp["t0"])
    secs = {"reorder_spectrum": 0.0, "find_g_points": 0.0, "create_look_up_table": 0.0, "optimize_lut": 0.0, "run_ckd": 0.0}
    ref = os.path.join(ROOT, "oracle", "_ref", "libequipartition_ref.so").encode()
    L.orc_find_g_lw_chain_ex.restype = C.c_int
    nband, cap = 2, 512
    planck_first = np.zeros((nlay + 1, nwav))
    per_gas, ranks = [], {}
    names = list(GASES)
    devnull, saved = os.open(os.devnull, os.O_WRONLY), os.dup(1)
    os.dup2(devnull, 1)                       # the reference search prints its progress to stdout
    try:
        for k, g in enumerate(names):
            od32 = np.ascontiguousarray(inp["base"][g][0], dtype=np.float32)
            bg32 = np.ascontiguousarray(inp["base"][names[1 - k]][0], dtype=np.float32)
            ng, st = np.zeros(nband, dtype=np.int32), np.zeros(nband, dtype=np.int32)
            cc, s3 = np.zeros(nband), np.zeros(3)
            rank = np.zeros(nwav, dtype=np.int32)
            r1, r2 = np.zeros(cap, dtype=np.int64), np.zeros(cap, dtype=np.int64)
            err, med, key = np.zeros(cap), np.zeros(cap), np.zeros(nwav)
            tol = np.full(nband, FIND_G["heating_rate_tolerance"])
            rc = L.orc_find_g_lw_chain_ex(ref, C.c_int(nlay), C.c_size_t(nwav), P(p1), P(t_hl), P(wn), P(dwn),
                                          od32.ctypes.data_as(C.POINTER(C.c_float)), bg32.ctypes.data_as(C.POINTER(C.c_float)),
                                          C.c_double(0.5), C.c_int(nband), P(BANDS[0]), P(BANDS[1]), C.c_int(1),
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
            # write_order stores the sorting variable as FLOAT (write_order.cpp:45-139); find_g_points reads it back
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
        rk = ranks[g["name"]]
        order = np.argsort(rk)
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
                 nband=2, ng=ng)
    tfl_rows = None
    for name, conc, files in (("h2o", "lut", [("h2o", 1.0), ("h2o", 4.0)]), ("co2", "linear", [("co2", 1.0)])):
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
    # the definition file stores the coefficients as FLOAT (ckd_model.cpp:418, :454-455): optimize_lut starts from those
    raw = dict(model, gases=[dict(g, **{k: np.asarray(g[k]).astype(np.float32).astype(np.float64) for k in ("molar_abs", "min_molar_abs", "max_molar_abs")})
                             for g in model["gases"]])
    raw["planck_function"] = model["planck_function"].astype(np.float32).astype(np.float64)

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
    out = {"iterations": res["iterations"], "status": res["status"], "ng": ng, "g_point": g_point, "per_gas": per_gas}
    for tag, m in (("raw", raw), ("optimised", optimised)):
        oo = ckd_synth.Oracle(o, m, [ev], cfg)
        f = oo.fluxes(oo.x0, ev)                              # (ncol, 2, nhl, ng)
        out[tag] = (f[:, 0].sum(-1), f[:, 1].sum(-1))
    secs["run_ckd"] = time.perf_counter() - t0
    out["model_raw"], out["model_optimised"] = raw, optimised
    return secs, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nwav", type=int, default=1 << 15)
    ap.add_argument("--nlay", type=int, default=30)
    ap.add_argument("--workdir", default=None)
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--netcdf4", action="store_true", help="run the tools a second time on NetCDF-4 twins of the spectra")
    args = ap.parse_args()
    ncores = min(16, len(os.sched_getaffinity(0)))
    os.environ.setdefault("OMP_NUM_THREADS", str(ncores))
    from ecckd_amd import api, ncio
    res = run(args.nwav, args.nlay, args.workdir, netcdf4=args.netcdf4)
    print(json.dumps(res) if args.json else json.dumps(res, indent=1))


def run(nwav=1 << 15, nlay=30, workdir=None, netcdf4=False):
    from ecckd_amd import api, ncio
    d = workdir or tempfile.mkdtemp(prefix="ecckd_e2e_")
    os.makedirs(d, exist_ok=True)
    ncores = int(os.environ.get("OMP_NUM_THREADS", "1"))
    with api.Context(0) as ctx:
        inp = make_inputs(ctx, d, nwav, nlay)
        g_secs, g_out = gpu_chain(d, inp)
        gpoints_classic = ncio.read_g_points(os.path.join(d, "gpoints.nc"))["g_point"]
        h5_secs = h5_same = None
        if netcdf4 and write_netcdf4_spectra(d):
            h5_secs, _ = gpu_chain(d, inp, ext="h5")
            h5_same = bool(np.array_equal(ncio.read_g_points(os.path.join(d, "gpoints.nc"))["g_point"], gpoints_classic))
            g_secs2, g_out = gpu_chain(d, inp)               # the files the comparison below reads are the classic run's again
        c_secs, c_out = cpu_chain(ctx, d, inp, ncores)
    gpf = ncio.read_g_points(os.path.join(d, "gpoints.nc"))
    same_g = bool(np.array_equal(gpf["g_point"], c_out["g_point"]))
    raw_file = ncio.read_ckd_model(os.path.join(d, "raw_ckd.nc"))
    opt_file = ncio.read_ckd_model(os.path.join(d, "ckd.nc"))
    rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if a.shape == b.shape else None

    def table_report(a, b):
        """How the two chains' tables differ: the transmission average of a saturated g point is 1 - (a few 1e-16), and the
        fitted optical depth -ln(1 - mean)/D (find_g_points.cpp:64-68, average_optical_depth.cpp:43-133) then moves by per
        cent with the order in which the mean was summed - in the reference as much as here.  Entries are therefore
        compared where the layer is NOT opaque at that g point as well."""
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        if a.shape != b.shape:
            return None
        r = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
        big = r > 1e-5
        return {"max_rel_diff": float(r.max()), "entries": int(r.size), "entries_differing_more_than_1e-5": int(big.sum()),
                "smallest_value_among_those_relative_to_table_max": float((np.abs(b[big]).min() / np.abs(b).max())) if big.any() else None,
                "median_rel_diff": float(np.median(r))}

    # stage-local parity of run_ckd: the CPU evaluation of the GPU chain's OWN definition files against bin/run_ckd's fluxes
    import pyoracle as o
    import ckd_synth
    ev = dict(pressure_hl=np.tile(inp["p1"], (inp["ncol"], 1)), temperature_hl=inp["T"], vmr_fl=inp["vmr"], gas_present=None)
    run_ckd_stage = {}
    for tag, m in (("raw", raw_file), ("optimised", opt_file)):
        oo = ckd_synth.Oracle(o, dict(m, nband=2), [ev], dict(OPT_DEFAULTS, flux_weight=0.2, flux_profile_weight=0.0, broadband_weight=0.5, prior_error=8.0))
        f = oo.fluxes(oo.x0, ev)
        run_ckd_stage[tag] = hr_rms_difference(inp["p1"], hr_k_per_day(inp["p1"], f[:, 0].sum(-1), f[:, 1].sum(-1)),
                                               hr_k_per_day(inp["p1"], *g_out[tag]))
    p1 = inp["p1"]
    hr = {k: hr_k_per_day(p1, *v[k]) for k in ("raw", "optimised") for v in (g_out,)}
    hr_c = {k: hr_k_per_day(p1, *c_out[k]) for k in ("raw", "optimised")}
    hr_lbl = hr_k_per_day(p1, inp["bdn"].sum(-1), inp["bup"].sum(-1))
    g_tot, c_tot = sum(g_secs.values()), sum(c_secs.values())
    return {
        "workload": "do_all_lw chain (test/do_all_lw.sh:83-96) on synthetic files: 2 gases (h2o look-up table with 2 mole "
                    "fractions, co2 linear), 2 bands, nwav=%d, nlay=%d, 3 idealised temperature columns, 3 training / "
                    "evaluation profiles; find_g_points tolerance %g K/d, optimize_lut %d iterations"
                    % (nwav, nlay, FIND_G["heating_rate_tolerance"], OPT["max_iterations"]),
        "gpu_tools_seconds": {k: round(v, 4) for k, v in g_secs.items()}, "gpu_tools_total_seconds": round(g_tot, 4),
        "gpu_tools_seconds_netcdf4_spectra": {k: round(v, 4) for k, v in h5_secs.items()} if h5_secs else None,
        "gpu_tools_total_seconds_netcdf4_spectra": round(sum(h5_secs.values()), 4) if h5_secs else None,
        "g_point_maps_identical_classic_vs_netcdf4_spectra": h5_same,
        "cpu_oracle_seconds": {k: round(v, 4) for k, v in c_secs.items()}, "cpu_oracle_total_seconds": round(c_tot, 4),
        "cpu_cores": ncores, "speedup_total": c_tot / g_tot,
        "speedup_per_stage": {k: c_secs[k] / g_secs[k] for k in c_secs if g_secs.get(k)},
        "note": "each GPU stage is a separate process: its wall time includes process start, HIP initialisation and file I/O",
        "agreement": {
            "ng_gpu": int(gpf["g_point"].max()) + 1, "ng_cpu": int(c_out["ng"]), "g_point_maps_identical": same_g,
            "raw_table_max_rel_diff": [rel(np.asarray(a["molar_abs"]), np.asarray(b["molar_abs"]))
                                       for a, b in zip(raw_file["gases"], c_out["model_raw"]["gases"])] if same_g else None,
            "raw_tables": [table_report(a["molar_abs"], b["molar_abs"]) for a, b in zip(raw_file["gases"], c_out["model_raw"]["gases"])]
                          if same_g else None,
            "run_ckd_stage_hr_rms_difference_K_per_day_cpu_vs_tool_on_the_same_file": run_ckd_stage,
            "optimised_table_max_rel_diff": [rel(np.asarray(a["molar_abs"]), np.asarray(b["molar_abs"]))
                                             for a, b in zip(opt_file["gases"], c_out["model_optimised"]["gases"])] if same_g else None,
            "iterations_gpu": g_out["iterations"], "iterations_cpu": c_out["iterations"],
            "hr_rms_difference_K_per_day": {k: hr_rms_difference(p1, hr[k], hr_c[k]) for k in hr},
            "hr_rms_error_against_lbl_K_per_day": {"gpu_" + k: hr_rms_difference(p1, hr[k], hr_lbl) for k in hr}
                                                  | {"cpu_" + k: hr_rms_difference(p1, hr_c[k], hr_lbl) for k in hr_c}},
    }


if __name__ == "__main__":
    main()
