"""Host-side mirrors of the reference executables' driver logic on top of the C ABI (api.py) and the classic
NetCDF layer (ncio.py).  Only orchestration lives here; every nwav-sized operation runs on the device.

create_look_up_table : src/ecckd/create_look_up_table.cpp:60-606 (without the base_wavenumber_boundary split)
"""
import numpy as np

from . import api, ncio
from ._lib import EcckdError, PARAMETER_ERROR, PROCESSING_ERROR


def _to_device(od, dev):
    """FLOAT on the device when the values are FLOAT (as stored in the CKDMIP files), else DOUBLE: nothing is rounded."""
    import torch
    od32 = od.astype(np.float32)
    return torch.as_tensor(od32 if np.array_equal(od32.astype(np.float64), od) else od, device=dev)


def remove_empty_g_points(g_point, band_number, solar_irradiance=None):
    """create_look_up_table.cpp:111-168: g points that occupy none of the spectrum are dropped and the rest renumbered.
    As in the reference, the new "band_number" of a kept g point is its OLD g-point index (:142), not its band."""
    g_point = np.asarray(g_point)
    ng = int(g_point.max()) + 1
    present = np.zeros(ng, dtype=bool)
    present[g_point[g_point >= 0]] = True
    if present.all():
        return g_point.astype(np.int32), np.asarray(band_number), solar_irradiance
    g_point_map = np.nonzero(present)[0]
    new_g = np.full(g_point.size, -1, dtype=np.int32)
    lookup = np.full(ng, -1, dtype=np.int32)
    lookup[g_point_map] = np.arange(g_point_map.size, dtype=np.int32)
    ok = g_point >= 0
    new_g[ok] = lookup[g_point[ok]]
    if (new_g < 0).any():
        raise EcckdError(1, "Some unassigned spectral points after mapping")          # THROW(1), :146-149
    new_band = g_point_map.copy()
    new_ssi = None if solar_irradiance is None else np.asarray(solar_irradiance)[g_point_map]
    return new_g, new_band, new_ssi


def split_base_g_points(g_point, band_number, band_wn1, band_wn2, base_wavenumber_boundary, wavenumber, ssi, solar_irradiance):
    """create_look_up_table.cpp:160-223: the base (first) g point of every band that contains one of the boundaries is split by
    wavenumber; the g points above move up.  Returns (g_point, band_number, solar_irradiance)."""
    g_point = np.array(g_point, dtype=np.int32)
    band_number = np.array(band_number, dtype=np.int64)
    solar = np.array(solar_irradiance, dtype=np.float64)
    bwb = np.asarray(base_wavenumber_boundary, dtype=np.float64)
    for iband in range(len(band_wn1)):
        inner = bwb[(bwb > band_wn1[iband]) & (bwb < band_wn2[iband])]
        if inner.size == 0:
            continue
        m = inner.size
        ig = int(np.nonzero(band_number == iband)[0].min())
        new_band = np.concatenate([band_number[:ig + 1], np.full(m, iband), band_number[ig + 1:]])
        new_solar = np.concatenate([solar[:ig], np.zeros(m + 1), solar[ig + 1:]])
        bounds = np.concatenate([[band_wn1[iband]], inner, [band_wn2[iband]]])
        old = g_point.copy()
        g_point = np.where(old > ig, old + m, old)
        for k in range(m + 1):
            g_point[(old == ig) & (wavenumber >= bounds[k]) & (wavenumber < bounds[k + 1])] = ig + k
        for k in range(m + 1):
            new_solar[ig + k] = ssi[g_point == ig + k].sum()
        band_number, solar = new_band, new_solar
    return g_point, band_number, solar


def create_look_up_table(ctx, g_point, band_number, band_wn1, band_wn2, gases, averaging_method="transmission",
                         temperature_stride=1, ssi=None, solar_irradiance=None, base_wavenumber_boundary=None, ssi_wavenumber=None):
    """The look-up-table assembly of create_look_up_table.cpp:225-606 from classic NetCDF spectra.

    g_point[nwav], band_number[ng]: from the g-points file (find_g_points' output).
    gases: list of dict(name=..., conc="none"|"linear"|"lut"|"relative-linear", and
             conc == "none":  inputs=[dict(path=..., scaling=-1, conc=-1), ...]  (read_merged_spectrum of the well-mixed files)
             otherwise:       inputs=[path, ...] (one file; one per mole fraction for "lut"), reference_conc for "relative-linear")
    ssi[nwav] selects the shortwave (weights = solar spectral irradiance, :333-340), otherwise Planck weights at each
    layer's temperature (:316-327).  Returns the model dict of api.Optimizer / ncio.write_ckd_model."""
    import torch
    is_sw = ssi is not None
    ng_file = int(np.asarray(g_point).max()) + 1
    g_point, band_number, solar_irradiance = remove_empty_g_points(g_point, band_number, solar_irradiance)
    save_g_points = int(g_point.max()) + 1 != ng_file                     # the numbering left the g-points file's (:575-577)
    if base_wavenumber_boundary is not None and len(base_wavenumber_boundary) > 0:
        if not is_sw or ssi_wavenumber is None:
            raise EcckdError(PARAMETER_ERROR, "base_wavenumber_boundary needs the ssi file (its wavenumbers and irradiances)")
        g_point, band_number, solar_irradiance = split_base_g_points(g_point, band_number, band_wn1, band_wn2, base_wavenumber_boundary,
                                                                      np.asarray(ssi_wavenumber), np.asarray(ssi), solar_irradiance)
        save_g_points = True
    ng = int(g_point.max()) + 1
    dev = ctx.device
    d_ssi = torch.as_tensor(np.asarray(ssi, dtype=np.float64), device=dev) if is_sw else None
    gmap = hr_wavenumber = None
    model = dict(gases=[], iband_per_g=np.asarray(band_number, dtype=np.int32))
    temperature_fl = None

    def column(s, od_dev, ref_vmr):
        nonlocal gmap, temperature_fl, hr_wavenumber
        p, t = s["pressure_hl"], s["temperature_hl"]
        if gmap is None:
            wn = torch.as_tensor(s["wavenumber_cm_1"], device=dev)
            dwn = torch.as_tensor(s["d_wavenumber_cm_1"], device=dev)
            gmap = api.GPointMap(ctx, torch.as_tensor(g_point, device=dev), ng, wn, dwn)
            hr_wavenumber = np.asarray(s["wavenumber_cm_1"], dtype=np.float64)
            model["log_pressure"] = np.log(0.5 * (p[1:] + p[:-1]))
        t_fl = (t[:-1] * p[:-1] + t[1:] * p[1:]) / (p[:-1] + p[1:])                  # :310-311
        out = gmap.average_optical_depth(p, od_dev, averaging_method, reference_surface_vmr=ref_vmr,
                                         temperature_fl=None if is_sw else t_fl, ssi=d_ssi)
        return t_fl, out

    for spec in gases:
        name, conc = spec["name"], spec["conc"]
        if conc not in ("none", "linear", "lut", "relative-linear"):
            raise EcckdError(PARAMETER_ERROR, f'conc_dependence "{conc}" not understood')
        g = dict(name=name, conc=conc, active=True)
        files = spec["inputs"]
        nconc = len(files) if conc == "lut" else 1
        tables, vmrs, tfl_rows = None, [], None
        for iconc in range(nconc):
            ncol, icol = 1, 0
            while icol < ncol:
                if conc == "none":
                    # read_merged_spectrum (:293-296): sum of the scaled well-mixed spectra, accumulated on the device
                    merged, first = None, None
                    ci = spec.get("conc_input")                          # dict(path, iprofile): read_merged_spectrum.cpp:47-61
                    rows_vmr, mols = [], []
                    for item in files:
                        s = ncio.read_spectrum(item["path"], icol * temperature_stride)
                        first = first or s
                        pc = cr = None
                        if ci is not None:
                            with ncio.NcFile(ci["path"]) as cf:
                                pc = cf.read("pressure_fl", ci["iprofile"])
                                cr = cf.read(s["molecule"].split(" ")[0] + "_mole_fraction_fl", ci["iprofile"])
                        sp, vrow = api.merge_scaling(s["pressure_hl"], item.get("scaling", -1.0), item.get("conc", -1.0),
                                                     s["reference_surface_vmr"], s["vmr_fl"], pressure_conc=pc, conc_req=cr)
                        rows_vmr.append(vrow); mols.append(s["molecule"])
                        merged = api.merge_spectrum(ctx, _to_device(s["optical_depth"], dev), sp, merged)
                    if icol == 0:                                         # create_look_up_table.cpp:311-313
                        g["composite_vmr"], g["composite_molecules"] = np.stack(rows_vmr), " ".join(mols)
                    s, od_dev, ref_vmr = first, merged, 1.0                            # reference_surface_vmr = 1 (:283)
                else:
                    s = ncio.read_spectrum(files[iconc], icol * temperature_stride)
                    od_dev = _to_device(s["optical_depth"], dev)
                    ref_vmr = s["reference_surface_vmr"]
                    if conc == "lut" and ref_vmr < 0.0:
                        raise EcckdError(PARAMETER_ERROR, "Invalid reference_surface_vmr for constructing VMR-dependent look-up table")
                ncol = (s["ncol"] + temperature_stride - 1) // temperature_stride
                t_fl, (k, kmin, kmax) = column(s, od_dev, ref_vmr)
                if tables is None:
                    shape = (nconc, ncol) + k.shape
                    tables = [np.zeros(shape) for _ in range(3)]
                    tfl_rows = np.zeros((ncol, k.shape[0]))
                for tab, v in zip(tables, (k, kmin, kmax)):
                    tab[iconc, icol] = v
                tfl_rows[icol] = t_fl
                icol += 1
            if conc == "lut":
                vmrs.append(ref_vmr)
        temperature_fl = tfl_rows                                                     # the last gas's, as the reference keeps it
        squeeze = (lambda a: a) if conc == "lut" else (lambda a: a[0])
        g["molar_abs"], g["min_molar_abs"], g["max_molar_abs"] = (squeeze(t) for t in tables)
        if conc == "lut":
            g["vmr"] = np.array(vmrs)
        if conc == "relative-linear":
            if "reference_conc" not in spec:
                raise EcckdError(PARAMETER_ERROR, f"{name}.reference_conc must be provided if conc_dependence is relative-linear")
            g["reference_vmr"] = float(spec["reference_conc"])
        model["gases"].append(g)

    model["temperature"] = temperature_fl
    # fraction of the spectrum contributing to each g point on a 10 (LW) / 50 (SW) cm-1 grid (:507-548)
    dwav = 50 if is_sw else 10
    startwav = int(np.floor(np.min(band_wn1) / dwav) * dwav)
    endwav = int(np.ceil(np.max(band_wn2) / dwav) * dwav)
    model["wavenumber1"] = dwav * np.arange(startwav // dwav, endwav // dwav, dtype=np.float64)
    model["wavenumber2"] = dwav * np.arange(startwav // dwav + 1, endwav // dwav + 1, dtype=np.float64)
    model["gpoint_fraction"] = gmap.gpoint_fraction(model["wavenumber1"], model["wavenumber2"])
    model["wavenumber1_band"], model["wavenumber2_band"] = np.asarray(band_wn1, float), np.asarray(band_wn2, float)
    model["nband"] = len(model["wavenumber1_band"])
    if is_sw:
        model["solar_irradiance"] = np.asarray(solar_irradiance, dtype=np.float64)
        model["planck_function"] = model["temperature_planck"] = None
        # solar irradiance of each interval (:556-561) and the Rayleigh coefficient of each g point
        # (CkdModel::calc_rayleigh_molar_scat, ckd_model.h:368-385; Bucholtz 1995, rayleigh_scattering.h:25-43)
        w1, w2, ssi64 = model["wavenumber1"], model["wavenumber2"], np.asarray(ssi, dtype=np.float64)
        k = np.searchsorted(w2, hr_wavenumber, side="left")                 # wavenumber1 < w <= wavenumber2
        ok = (k < w2.size) & (hr_wavenumber > w1[np.minimum(k, w1.size - 1)])
        model["solar_spectral_irradiance"] = np.bincount(k[ok], weights=ssi64[ok], minlength=w1.size)
        um = 10000.0 / (0.5 * (w1 + w2))
        xs = np.where(um < 0.5, 3.01577e-32 * um ** -(3.55212 + 1.35579 * um + 0.11563 / um),
                      4.01061e-32 * um ** -(3.99668 + 0.00110298 * um + 0.0271393 / um))
        molar_column = 1.0e5 / (9.80665 * 0.001 * 28.970)
        trans_hr = np.exp(-molar_column * xs * 6.02214076e23 / 0.5)
        gf, si = model["gpoint_fraction"], model["solar_spectral_irradiance"]
        model["rayleigh_molar_scattering"] = -np.log(np.maximum(1.0e-14, (gf @ (si * trans_hr)) / (gf @ si))) * 0.5 / molar_column
    else:
        model["temperature_planck"] = np.arange(120.0, 351.0)                         # :581
        model["planck_function"] = gmap.planck_lut(model["temperature_planck"])        # :585-591
    model["ng"] = ng
    if save_g_points:
        model["save_g_points"], model["wavenumber_hr"], model["g_point_hr"] = True, hr_wavenumber, g_point
    gmap.close()
    return model


def iband_per_g(model, wavenumber1, wavenumber2):
    """CkdModel::iband_per_g (ckd_model.h:287-306): the LBL band each g point lies in, from gpoint_fraction."""
    gf, w1, w2 = np.asarray(model["gpoint_fraction"]), np.asarray(model["wavenumber1"]), np.asarray(model["wavenumber2"])
    iband = np.full(gf.shape[0], -1, dtype=np.int32)
    for ib in range(len(wavenumber1)):
        weight = gf[:, (w1 >= wavenumber1[ib]) & (w2 <= wavenumber2[ib])].sum(axis=1)
        if np.any((weight > 0.05) & ((weight < 0.95) | (weight > 1.05))):
            raise EcckdError(1, "G-points do not lie entirely within requested bands")
        iband[weight > 0.5] = ib
    if np.any(iband < 0):
        raise EcckdError(1, "Some g-points not inside a band")
    return iband


def optimize_lut(ctx, model, training_files, relative_to=None, band_mapping=None, gmap=None, max_iterations=3000,
                 convergence_criterion=0.0, bounded=True, max_no_rayleigh_wavenumber=None, erythemal_weight=0.0,
                 remove_min_max=False, **cfg):
    """The driver of optimize_lut.cpp:60-330 on top of api.Optimizer: training scenes from LBL flux files
    (ncio.read_lbl_fluxes), optional "relative_to" file, bounded L-BFGS, optimised coefficients back into the model.

    model: dict from ncio.read_ckd_model (its `active` flags select the gases being optimised, optimize_lut.cpp:161).
    cfg:   flux_weight, flux_profile_weight, broadband_weight, spectral_boundary_weight, prior_error, ... (api.Optimizer).
    remove_min_max: drop the min / max tables from the model that is returned (`ckd_model.save_min_max(false)`,
           optimize_lut.cpp:243-244, :308-310), so that ncio.write_ckd_model leaves the _min / _max variables out.
    Returns (model with optimised molar_abs, result dict of Optimizer.minimize)."""
    names = [g["name"] for g in model["gases"]]
    is_sw = model.get("solar_irradiance") is not None

    def load(path):
        s = ncio.read_lbl_fluxes(path, names, band_mapping=band_mapping, gmap=gmap, ctx=ctx)
        if s["have_band_fluxes"]:
            s["iband_per_g"] = iband_per_g(model, s["band_wavenumber1"], s["band_wavenumber2"])     # :273-276
        if is_sw and max_no_rayleigh_wavenumber is not None:
            ncio.mask_rayleigh_up(s, max_no_rayleigh_wavenumber)                                     # :278-283
        if is_sw and erythemal_weight > 0.0 and "erythemal_spectrum" in s:
            s["spectral_boundary_weights"] = erythemal_weight * s["erythemal_spectrum"]              # solve_adept.cpp:182
        return s

    rel = None
    if relative_to is not None:                                                                      # :204-236
        rel = load(relative_to)
        m0 = dict(model, iband_per_g=rel.get("iband_per_g", model["iband_per_g"]))
        ref_opt = api.Optimizer(ctx, m0, [_scene_for_optimizer(rel, is_sw)], **cfg)
        _, fl = ref_opt.forward(ref_opt.initial_state(), unclamped=True)     # od = value(aod), no clamp (:231-234)
        ref_opt.close()
    scenes = []
    iband = None
    for path in training_files:
        s = load(path)
        if rel is not None:
            ncio.subtract_lbl_fluxes(s, rel)                                                         # :251-254
        sc = _scene_for_optimizer(s, is_sw)
        if rel is not None:
            sc["relative_flux_dn"], sc["relative_flux_up"] = np.ascontiguousarray(fl[:, 0]), np.ascontiguousarray(fl[:, 1])
        scenes.append(sc)
        iband = s.get("iband_per_g", iband)
    if not scenes:
        raise EcckdError(PARAMETER_ERROR, '"training_input" not specified')
    m = dict(model)
    if iband is not None:
        m["iband_per_g"] = iband
    opt = api.Optimizer(ctx, m, scenes, **dict(dict(cap_relative_linear=0.8), **cfg))                # :185
    res = opt.minimize(max_iterations=max_iterations, convergence_criterion=convergence_criterion, bounded=bounded)
    out = dict(model, gases=[dict(g) for g in model["gases"]])
    for i, g in enumerate(out["gases"]):
        g["molar_abs"] = opt.coefficients(res["x"], i, np.asarray(g["molar_abs"]).shape)
        if remove_min_max:
            g.pop("min_molar_abs", None)
            g.pop("max_molar_abs", None)
    opt.close()
    return out, res


def _scene_for_optimizer(s, is_sw):
    keys = ["pressure_hl", "temperature_hl", "vmr_fl", "gas_present", "flux_dn", "flux_up", "spectral_flux_dn_surf",
            "spectral_flux_up_toa", "spectral_boundary_weights"]
    if is_sw:
        keys += ["mu0", "tsi", "albedo"]
    return {k: s[k] for k in keys if s.get(k) is not None}


def reorder_spectrum(ctx, input_path, output_path, band_bound1, band_bound2, iprofile=0, threshold_optical_depth=None,
                     ssi=None, config_str="", history=None):
    """reorder_spectrum.cpp:45-310 for one gas: spectrum file in, reordering file out.  Returns the order dict."""
    s = ncio.read_spectrum(input_path, iprofile)
    thr = threshold_optical_depth if threshold_optical_depth is not None else (0.5 if ssi is None else 0.25)
    od = s["optical_depth"]
    od32 = od.astype(np.float32)
    od_in = od32 if np.array_equal(od32.astype(np.float64), od) else od
    key, col, iband, rank = api.reorder_spectrum(ctx, s["pressure_hl"], s["wavenumber_cm_1"], s["d_wavenumber_cm_1"], od_in, ssi, thr,
                                                 band_bound1, band_bound2)
    # the file stores the bounds clamped to the range of the data (reorder_spectrum.cpp:268-273)
    wn = s["wavenumber_cm_1"]
    band_bound1, band_bound2 = np.array(band_bound1, dtype=np.float64), np.array(band_bound2, dtype=np.float64)
    band_bound1[0], band_bound2[-1] = max(wn[0], band_bound1[0]), min(wn[-1], band_bound2[-1])
    ncio.write_order(output_path, band_bound1, band_bound2, s["wavenumber_cm_1"], s["d_wavenumber_cm_1"], iband, rank, key, col,
                     molecule=s["molecule"] or "", config_str=config_str, history=history)
    return dict(spectrum=s, key=key, column_optical_depth=col, band_number=iband, rank=rank)


def reorder_single_band_sharded(ctx, pressure_hl, wn, dwn, od, threshold_optical_depth=0.5, group=None):
    """reorder_spectrum of ONE longwave band (the fsck structure) with the key sweep split by wavenumber range over the processes
    of `group`: every process runs K1 on its own whole-tile range of the resident spectrum, the keys and column
    optical depths travel to rank 0 once (16 B per wavenumber), rank 0 runs the one stable sort (K3).  A wavenumber's key is a
    function of its column alone, so key, column optical depth and rank have the bits one process gives.
    -> (key, col_od, rank) device tensors on rank 0, (None, None, None) elsewhere."""
    from . import shard
    t_ideal = api.idealised_temperature(pressure_hl)
    nwav = od.shape[1]

    def key_of_range(b, e):
        if e <= b:
            import torch
            z = torch.empty(0, dtype=torch.float64, device=ctx.device)
            return z, z
        key, col = api.reorder_key_lw(ctx, pressure_hl, t_ideal, wn[b:e], dwn[b:e], od[:, b:e], threshold_optical_depth)
        ctx.synchronize()
        return key, col

    def sort_on_root(key):
        rnk, _ = api.stable_argsort_bands(ctx, key, [0], [nwav - 1], want_ordered=False)
        return rnk

    return shard.reorder_single_band(key_of_range, nwav, sort_on_root, group=group, device=ctx.device)


def _prepare_gas(ctx, g, averaging_method, flux_weight, min_pressure, planck_reuse=None, sw=None):
    """Gas preparation (find_g_points.cpp:872-1150) of one gas whose spectra are on the device, and its sorting variable in
    sorted order (:781).  g, sw, planck_reuse: see _search_gas.  -> (gas handle, sorting variable sorted, band albedo or None)"""
    import time
    timing = g.setdefault("timing", {})
    t0 = time.perf_counter()
    if sw is None:
        gas = api.GasLW(ctx, g["pressure_hl"], g["temperature_hl"], g["wn"], g["dwn"], g["rank"], g["od"], g.get("bg"),
                        averaging_method, flux_weight, min_pressure, planck_hl_reuse=planck_reuse)
        band_albedo = None
    else:
        gas = api.GasSW(ctx, g["pressure_hl"], sw["ssi"], g["rank"], g["od"], g.get("bg"), averaging_method, flux_weight,
                        min_pressure, sw["cos_sza"], sw["albedo"], g.get("min_scaling", 1.0), g.get("max_scaling", 1.0))
        band_albedo = sw["band_albedo"]
    sv_sorted = api.gather_f64(ctx, g["sorting_variable"], api.invert_permutation(ctx, g["rank"]))
    ctx.synchronize()
    timing["preparation"] = timing.get("preparation", 0.0) + time.perf_counter() - t0
    return gas, sv_sorted, band_albedo


def _band_requests(g, bands, tol, band_albedo):
    """The arguments of ecckd_find_g_bands_ex for `bands` of gas g: first / last sorted index, tolerance and options per band;
    a shortwave band brings its albedo (init_sw(..., band_albedo(jband), ...)) with it."""
    begin, end = g["band_begin"], g["band_end"]
    opts = [dict(min_g_points=int(g["min_g_points"][b]), max_g_points=int(g["max_g_points"][b])) for b in bands]
    if band_albedo is not None:
        opts = [dict(o, band_albedo=float(band_albedo[b])) for b, o in zip(bands, opts)]
    return dict(ibegin=[int(begin[b]) for b in bands], iend=[int(end[b]) for b in bands],
                heating_rate_tolerance=np.ascontiguousarray(np.asarray(tol)[bands]), options=opts)


def _finish_gas(gas, g, bands, band_res, sv_sorted):
    """The median sorting variable of every g point of these bands in one call (:1404-1409) and the per-band result dicts."""
    import time
    t2 = time.perf_counter()
    r1_all = np.concatenate([np.asarray(res["rank1"], dtype=np.int64) for res in band_res])
    r2_all = np.concatenate([np.asarray(res["rank2"], dtype=np.int64) for res in band_res])
    med_all = gas.median_sorting_variable(sv_sorted, r1_all, r2_all)
    out = []
    k0 = 0
    for b, res in zip(bands, band_res):
        n = len(res["rank1"])
        med = med_all[k0:k0 + n]
        k0 += n
        out.append((b, dict(rank1=[int(v) for v in res["rank1"]], rank2=[int(v) for v in res["rank2"]],
                            error=[float(v) for v in res["error"]], status=int(res["status"]),
                            comp_cost=float(res["comp_cost"]), sorting_variable=[float(v) for v in med])))
    timing = g.setdefault("timing", {})
    timing["medians"] = timing.get("medians", 0.0) + time.perf_counter() - t2
    return out


def _search_gas(ctx, g, bands, tol, tolerance_tolerance, max_iterations, averaging_method, flux_weight, min_pressure,
                sequential_bands, planck_reuse=None, sw=None):
    """One gas of the loop find_g_points.cpp:655-1450 with everything already on the device: gas preparation (:872-1150),
    the searches of `bands` (:1152-1414, side by side unless sequential_bands) and the median sorting variable of every
    g point (:1404-1409).

    g:  dict(pressure_hl, temperature_hl (host), wn, dwn, rank (int32), od, bg (or None), sorting_variable: device tensors in
        ORIGINAL wavenumber order; band_begin[nband], band_end[nband]: first / last sorted index of every band;
        min_g_points[nband], max_g_points[nband]; shortwave: min_scaling, max_scaling)
    sw: None (longwave; planck_reuse = device pointer of the first gas's Planck matrix or None) or
        dict(ssi, albedo: device tensors, band_albedo[nband], cos_sza).
    Returns (gas handle - the caller closes it -, [(band, dict(rank1, rank2, error, status, comp_cost, sorting_variable))])."""
    import time
    gas, sv_sorted, band_albedo = _prepare_gas(ctx, g, averaging_method, flux_weight, min_pressure, planck_reuse, sw)
    timing = g["timing"]
    t1 = time.perf_counter()
    req = _band_requests(g, bands, tol, band_albedo)
    if len(bands) > 1 and not sequential_bands:
        # bands side by side, sharing their error batches (ecckd_find_g_bands_ex): same decisions per band
        band_res = gas.find_g_bands_ex(req["ibegin"], req["iend"], req["heating_rate_tolerance"], tolerance_tolerance, max_iterations,
                                       req["options"])
    else:
        band_res = []
        for k, b in enumerate(bands):
            o = dict(req["options"][k])
            o.pop("band_albedo", None)
            if band_albedo is not None:
                gas.set_band_albedo(band_albedo[b])                                                   # init_sw(..., band_albedo(jband), ...)
            band_res.append(gas.find_g_band_ex(req["ibegin"][k], req["iend"][k], float(tol[b]), tolerance_tolerance, max_iterations, **o))
    timing["search"] = timing.get("search", 0.0) + time.perf_counter() - t1
    return gas, _finish_gas(gas, g, bands, band_res, sv_sorted)


def _deal(ngas, nband, rank, world_size, group):
    from . import shard
    if rank is None or world_size is None:
        rank, world_size = shard.world(group)
    tasks = shard.task_table(range(ngas), nband)
    mine = [tasks[t] for t in shard.deal_tasks(len(tasks), rank, world_size)]
    my_bands = {}
    for gi, b in mine:
        my_bands.setdefault(gi, []).append(b)
    return rank, world_size, tasks, mine, my_bands


def _collect(local, ntasks, rank, world_size, group, dev):
    """The only communication of a sharded find_g_points: the per-band results to rank 0, ONE all-reduce of the final cost
    (sum of the g points' errors) and of the work counter."""
    from . import shard
    cost_local = float(sum(sum(r["error"]) for _, _, r in local))
    comp_local = float(sum(r["comp_cost"] for _, _, r in local))
    gathered = shard.gather_to_root(local, group)
    if world_size > 1:
        _, comp_sum, cost_sum = shard.reduce_scalars(0.0, comp_local, cost_local, device=dev if _is_nccl(group) else None, group=group)
    else:
        comp_sum, cost_sum = comp_local, cost_local
    by_task = None
    if rank == 0:
        by_task = {(gi, b): r for part in gathered for gi, b, r in part}
        if len(by_task) != ntasks:
            raise EcckdError(PROCESSING_ERROR, "find_g_points: %d of %d (gas, band) searches came back" % (len(by_task), ntasks))
    return by_task, cost_sum, comp_sum


def _per_gas_tables(names, nband, by_task):
    """SingleGasData of every gas (single_gas_data.h:24-80) from the per-band results, bands in order."""
    per_gas = []
    for gi, name in enumerate(names):
        out = dict(name=name, n_g_points=[], band_number=[], rank1=[], rank2=[], error=[], sorting_variable=[], status=[], comp_cost=[])
        for b in range(nband):
            r = by_task[(gi, b)]
            n = len(r["error"])
            out["n_g_points"].append(n)
            out["band_number"] += [b] * n
            out["rank1"] += r["rank1"]; out["rank2"] += r["rank2"]; out["error"] += r["error"]
            out["sorting_variable"] += r["sorting_variable"]
            out["status"].append(r["status"]); out["comp_cost"].append(r["comp_cost"])
        per_gas.append(out)
    return per_gas


def find_g_points(ctx, gases, band_bound1, band_bound2, heating_rate_tolerance, output_path=None, averaging_method="transmission",
                  flux_weight=0.02, min_pressure=0.0, tolerance_tolerance=0.02, max_iterations=60, iprofile=0, ssi=None,
                  max_no_rayleigh_wavenumber=10000.0, reference_albedo=0.15, cos_sza=0.5, sequential_bands=False,
                  rank=None, world_size=None, group=None):
    """The main loop of find_g_points.cpp:655-1660 over classic files (shortwave when `ssi[nwav]` is given: solar weights,
    reference albedo 0.15 below max_no_rayleigh_wavenumber (:469, :522, :757-761, :921-923), REFERENCE_COS_SZA = 0.5,
    per-gas min_scaling / max_scaling (:661-667)): per gas the merged background, the gas
    preparation, the band searches and the median sorting variables; then the overlap of the gases' g points, the merged
    g-point map and the g-points file.

    gases: list of dict(name, input=spectrum file, reordering_input=order file, background=[dict(path, scaling, conc), ...],
                        min_g_points=1, max_g_points=256).

    Several processes (one per GPU, torch.distributed initialised - RCCL on the GPU box - or explicit rank / world_size):
    the (gas, band) searches are independent problems (:655, :1152) and are dealt to the processes as contiguous shares of
    the task table (shard.deal_tasks); a process reads and prepares only the gases of which it searches a band.  Nothing is
    exchanged while searching.  The per-band results (a few numbers per g point) are gathered on rank 0, which does what
    follows the gas loop in the reference (:1452-1660: overlap, merged map, file); ONE all-reduce combines the final cost
    (sum of the g points' errors) and the work counters.  Returns the result dict on rank 0, a summary elsewhere; both hold
    `cost_sum` and `comp_cost_sum`, identical on every rank.  The g points do not depend on the number of processes."""
    import torch
    dev = ctx.device
    nband = len(band_bound1)
    ngas = len(gases)
    rank, world_size, tasks, mine, my_bands = _deal(ngas, nband, rank, world_size, group)
    tol = np.broadcast_to(np.asarray(heating_rate_tolerance, dtype=np.float64), (nband,))           # :762-771
    first_lw_gas = None
    planck_first = None            # the first gas's Planck matrix where this process does not prepare the first gas itself
    local = []                     # (gas index, band, result of the search)
    for gi in sorted(my_bands):
        spec = gases[gi]
        s = ncio.read_spectrum(spec["input"], iprofile)
        order = ncio.read_order(spec["reordering_input"])
        wn = s["wavenumber_cm_1"]
        bg = None
        for item in spec.get("background", []):                                                       # read_merged_spectrum (:891)
            b = ncio.read_spectrum(item["path"], iprofile)
            sp, _ = api.merge_scaling(b["pressure_hl"], item.get("scaling", -1.0), item.get("conc", -1.0),
                                      b["reference_surface_vmr"], b["vmr_fl"])
            bg = api.merge_spectrum(ctx, _to_device(b["optical_depth"], dev), sp, bg)
        iband = order["band_number"]
        idx = [np.nonzero(iband == b)[0] for b in range(nband)]
        g = dict(pressure_hl=s["pressure_hl"], temperature_hl=s["temperature_hl"], wn=torch.as_tensor(wn, device=dev),
                 dwn=torch.as_tensor(s["d_wavenumber_cm_1"], device=dev), rank=torch.as_tensor(order["rank"], device=dev),
                 od=_to_device(s["optical_depth"], dev), bg=bg, sorting_variable=torch.as_tensor(order["sorting_variable"], device=dev),
                 band_begin=[int(i[0]) if i.size else -1 for i in idx], band_end=[int(i[-1]) if i.size else -1 for i in idx],
                 min_g_points=np.broadcast_to(np.asarray(spec.get("min_g_points", 1)), (nband,)),       # per band, :733-754
                 max_g_points=np.broadcast_to(np.asarray(spec.get("max_g_points", 256)), (nband,)),
                 min_scaling=spec.get("min_scaling", 1.0), max_scaling=spec.get("max_scaling", 1.0))
        sw = reuse = None
        if ssi is None:
            # the reference evaluates the Planck function once, on the FIRST gas's reordered grid, and keeps using that
            # matrix for the later gases (find_g_points.cpp:529, :970-984): reproduced.  The process that prepares the first
            # gas keeps it alive; any other builds the same matrix from the first gas's ordering file and profile
            if gi > 0 and first_lw_gas is not None:
                reuse = first_lw_gas.view_ptr("planck_hl")[0]
            elif gi > 0:
                if planck_first is None:
                    s0 = ncio.read_spectrum(gases[0]["input"], iprofile, optical_depth=False)
                    o0 = ncio.read_order(gases[0]["reordering_input"])
                    planck_first = api.planck_hl_sorted(ctx, s0["temperature_hl"], torch.as_tensor(s0["wavenumber_cm_1"], device=dev),
                                                        torch.as_tensor(s0["d_wavenumber_cm_1"], device=dev),
                                                        torch.as_tensor(o0["rank"], device=dev))
                reuse = planck_first.data_ptr()
        else:
            b2 = np.asarray(band_bound2, dtype=np.float64)
            no_ray = b2 <= max_no_rayleigh_wavenumber
            wn_limit = b2[no_ray].max() if no_ray.any() else 0.0                                      # :761
            sw = dict(ssi=torch.as_tensor(np.asarray(ssi, dtype=np.float64), device=dev), cos_sza=cos_sza,
                      band_albedo=np.where(no_ray, reference_albedo, 0.0),                            # :756-760
                      albedo=torch.as_tensor(np.where(wn < wn_limit, reference_albedo, 0.0), device=dev))   # :921-923
        gas, res = _search_gas(ctx, g, my_bands[gi], tol, tolerance_tolerance, max_iterations, averaging_method, flux_weight,
                               min_pressure, sequential_bands, reuse, sw)
        local += [(gi, b, r) for b, r in res]
        if ssi is None and gi == 0:
            first_lw_gas = gas
        else:
            gas.close()
    if first_lw_gas is not None:
        first_lw_gas.close()
    planck_first = None
    by_task, cost_sum, comp_sum = _collect(local, len(tasks), rank, world_size, group, dev)
    if rank != 0:
        return dict(rank=rank, tasks=mine, cost_sum=cost_sum, comp_cost_sum=comp_sum)
    # ---- what follows the gas loop (:1452-1660) ----
    per_gas = _per_gas_tables([spec["name"] for spec in gases], nband, by_task)
    gas_gp = []
    wn = None
    for spec, out in zip(gases, per_gas):
        order = ncio.read_order(spec["reordering_input"])
        wn = order["wavenumber"]
        gp = api.gas_g_point(ctx, torch.as_tensor(order["rank"], device=dev), out["rank1"], out["rank2"])
        out["g_point"] = gp.cpu().numpy()
        gas_gp.append(gp)
    ng, band_number, g_min, g_max = api.overlap_g_points([g["n_g_points"] for g in per_gas],
                                                         [np.asarray(g["sorting_variable"]) for g in per_gas])
    g_point, n_unassigned = api.merge_g_points(ctx, gas_gp, g_min, g_max)
    for k, g in enumerate(per_gas):
        g["g_min"], g["g_max"] = g_min[k], g_max[k]
    result = dict(ng=ng, band_number=band_number, g_point=g_point.cpu().numpy(), n_unassigned=n_unassigned, gases=per_gas,
                  wavenumber=wn, cost_sum=cost_sum, comp_cost_sum=comp_sum, rank=0, tasks=mine)
    if output_path is not None:
        ncio.write_g_points(output_path, band_bound1, band_bound2, band_number, per_gas, wn, result["g_point"])
    return result


def find_g_points_resident(ctx, names, load_gas, nband, heating_rate_tolerance, first_gas_order=None, averaging_method="transmission",
                           flux_weight=0.02, min_pressure=0.0, tolerance_tolerance=0.02, max_iterations=60,
                           sequential_bands=False, rank=None, world_size=None, group=None, merged_map=True, sw=None,
                           gases_side_by_side=1):
    """find_g_points on spectra that are already resident in HBM (bench.py, the full-size tests): the same dealing of the
    (gas, band) tasks, the same per-gas work (_search_gas) and the same collection as find_g_points, with `load_gas(gi)`
    handing over the device tensors of gas gi (the dict _search_gas takes; the call may do the gas's reorder_spectrum step,
    K1 + K3, itself) instead of reading files.  Longwave unless `sw` is given: dict(ssi, albedo: device tensors,
    band_albedo[nband], cos_sza) as _search_gas takes it (no shared Planck matrix then; first_gas_order is only asked for the
    number of wavenumbers by a process without tasks).

    first_gas_order: callable -> dict(temperature_hl, wn, dwn, rank) of the FIRST gas for a process that searches none of
    its bands (what the file driver reads from the first gas's ordering file), for the shared Planck matrix.
    merged_map: every gas's per-wavenumber g points travel to rank 0 (one reduce per gas; each wavenumber is set by exactly
    one process), which forms the merged g-point map (:1459-1475).
    gases_side_by_side: 1 = gas after gas as the reference's loop (:655): load, prepare, search, release; otherwise every gas
    of this process is loaded and prepared first (all resident: ~9.5 GB per gas at 7.2e6 points), then ALL their band searches
    run side by side (ecckd_find_g_gases: one host thread and HIP stream per gas; n > 1: at most n gases at a time, 0: what the
    host has cores for) - same g points, same errors, the gases' latency-bound batches in each other's shadow.
    -> rank 0: dict(ng, band_number, gases, g_point (device), n_unassigned, cost_sum, comp_cost_sum, points);
       others: dict(cost_sum, comp_cost_sum, points).  `points`: wavenumber points this process worked through: one pass over its bands (reorder, preparation) + the points
    its searches swept on the device (an interval asked for twice is swept once: the memo of interval errors)."""
    import torch
    dev = ctx.device
    ngas = len(names)
    rank, world_size, tasks, mine, my_bands = _deal(ngas, nband, rank, world_size, group)
    tol = np.broadcast_to(np.asarray(heating_rate_tolerance, dtype=np.float64), (nband,))
    first_lw_gas = None
    planck_first = None
    import time
    local, maps = [], {}
    points = 0.0
    nwav = None
    phase = {"reorder": 0.0, "preparation": 0.0, "search": 0.0, "medians": 0.0, "maps": 0.0, "collect": 0.0}
    side_by_side = gases_side_by_side != 1 and len(my_bands) > 1
    prepared = {}
    if side_by_side:
        # every gas of this process is loaded and prepared; its searches start as soon as it is (GasSearchJob.add returns at once)
        # and run while the next gas is merged, reordered and prepared on the context's stream
        job = api.GasSearchJob(tolerance_tolerance, max_iterations, max_concurrent=gases_side_by_side)
        ts = None
        order = sorted(my_bands)
        for gi in order:
            tl = time.perf_counter()
            g = load_gas(gi)
            ctx.synchronize()
            phase["reorder"] += time.perf_counter() - tl
            reuse = None
            if sw is not None:
                pass
            elif gi > 0 and first_lw_gas is not None:
                reuse = first_lw_gas.view_ptr("planck_hl")[0]
            elif gi > 0:
                if planck_first is None:
                    o0 = first_gas_order()
                    planck_first = api.planck_hl_sorted(ctx, o0["temperature_hl"], o0["wn"], o0["dwn"], o0["rank"])
                reuse = planck_first.data_ptr()
            gas, sv_sorted, band_albedo = _prepare_gas(ctx, g, averaging_method, flux_weight, min_pressure, reuse, sw)
            for k in ("od", "bg"):                # the spectra are in the gas's rows now
                g.pop(k, None)
            req = _band_requests(g, my_bands[gi], tol, band_albedo)
            prepared[gi] = (g, gas, sv_sorted)
            if gi == 0 and sw is None:
                first_lw_gas = gas
            ts = ts or time.perf_counter()
            job.add(gas, req["ibegin"], req["iend"], req["heating_rate_tolerance"], req["options"])
        tw = time.perf_counter()
        all_res = job.wait()
        phase["search"] += time.perf_counter() - tw           # what is left of the searches when the last gas is prepared
        phase["search_window"] = phase.get("search_window", 0.0) + time.perf_counter() - ts
        for gi, band_res in zip(order, all_res):
            g, gas, sv_sorted = prepared[gi]
            prepared[gi] = (g, gas, _finish_gas(gas, g, my_bands[gi], band_res, sv_sorted))
    for gi in sorted(my_bands):
        if side_by_side:
            g, gas, res = prepared.pop(gi)
            nwav = g["rank"].numel()
        else:
            tl = time.perf_counter()
            g = load_gas(gi)
            ctx.synchronize()
            phase["reorder"] += time.perf_counter() - tl
            nwav = g["rank"].numel()
            reuse = None
            if sw is not None:
                pass
            elif gi > 0 and first_lw_gas is not None:
                reuse = first_lw_gas.view_ptr("planck_hl")[0]
            elif gi > 0:
                if planck_first is None:
                    o0 = first_gas_order()
                    planck_first = api.planck_hl_sorted(ctx, o0["temperature_hl"], o0["wn"], o0["dwn"], o0["rank"])
                reuse = planck_first.data_ptr()
            gas, res = _search_gas(ctx, g, my_bands[gi], tol, tolerance_tolerance, max_iterations, averaging_method, flux_weight,
                                   min_pressure, sequential_bands, reuse, sw)
        for b, r in res:
            r["index_range"] = (int(g["band_begin"][b]), int(g["band_end"][b]))
            points += g["band_end"][b] - g["band_begin"][b] + 1            # the reorder / preparation pass over the band
        points += gas.eval_stats()["points_evaluated"]                       # what the searches swept on the device
        local += [(gi, b, r) for b, r in res]
        for k in ("preparation", "search", "medians"):
            phase[k] += g["timing"].get(k, 0.0)
        tm = time.perf_counter()
        if merged_map:
            # this process's bands of the gas: (band << 16) + g point of every wavenumber counted from the band's first (-1
            # elsewhere).  One launch over the spectrum with the bands' rank ranges one after the other (a band's ranks lie in its
            # own index range, so no other band's wavenumbers fall into them), then from that running count to band and count
            # within the band - what the root can renumber once it knows every band's number of g points.
            if res:
                r1 = np.concatenate([np.asarray(r["rank1"], dtype=np.int64) for _, r in res])
                r2 = np.concatenate([np.asarray(r["rank2"], dtype=np.int64) for _, r in res])
                assert nband < 32768 and all(len(r["rank1"]) < 65536 for _, r in res)
                within = np.concatenate([(b << 16) + np.arange(len(r["rank1"]), dtype=np.int32) for b, r in res]).astype(np.int32)
                gp = api.gas_g_point(ctx, g["rank"], r1, r2)
                tbl = torch.as_tensor(within, device=dev)
                gp = torch.where(gp >= 0, tbl[gp.clamp(min=0).long()], gp)
            else:
                gp = torch.full((nwav,), -1, dtype=torch.int32, device=dev)
            maps[gi] = gp
        if gi == 0 and sw is None:
            first_lw_gas = gas
        else:
            gas.close()
        phase["maps"] += time.perf_counter() - tm
    if first_lw_gas is not None:
        first_lw_gas.close()
    planck_first = None
    tc = time.perf_counter()
    by_task, cost_sum, comp_sum = _collect(local, len(tasks), rank, world_size, group, dev)
    gas_gp = []
    if merged_map:
        if nwav is None:
            nwav = first_gas_order()["rank"].numel()
        for gi in range(ngas):
            gp = maps.get(gi)
            if gp is None:
                gp = torch.full((nwav,), -1, dtype=torch.int32, device=dev)
            gas_gp.append(_max_to_root(gp, world_size, group))
    if rank != 0:
        phase["collect"] = time.perf_counter() - tc
        return dict(rank=rank, tasks=mine, cost_sum=cost_sum, comp_cost_sum=comp_sum, points=points, phase_seconds=phase)
    per_gas = _per_gas_tables(names, nband, by_task)
    ng, band_number, g_min, g_max = api.overlap_g_points([g["n_g_points"] for g in per_gas],
                                                         [np.asarray(g["sorting_variable"]) for g in per_gas])
    result = dict(ng=ng, band_number=band_number, gases=per_gas, cost_sum=cost_sum, comp_cost_sum=comp_sum, rank=0, tasks=mine,
                  points=points, phase_seconds=phase)
    for k, g in enumerate(per_gas):
        g["g_min"], g["g_max"] = g_min[k], g_max[k]
    if merged_map:
        # the gathered maps hold (band << 16) + the g point counted from the band's first: add the number of g points of the gas's
        # earlier bands (SingleGasData::store_g_points numbers them through the bands, single_gas_data.h:56-62)
        for gi, (gp, out) in enumerate(zip(gas_gp, per_gas)):
            first = torch.as_tensor(np.concatenate([[0], np.cumsum(out["n_g_points"])[:-1]]).astype(np.int32), device=gp.device)
            renumbered = first[(gp >> 16).clamp(min=0).long()] + (gp & 0xffff)
            gas_gp[gi] = torch.where(gp >= 0, renumbered, gp)
        result["g_point"], result["n_unassigned"] = api.merge_g_points(ctx, gas_gp, g_min, g_max)
        result["gas_g_point"] = gas_gp
    phase["collect"] = time.perf_counter() - tc
    return result


def _max_to_root(t, world_size, group=None):
    """Element-wise maximum over the processes on rank 0 (one reduce; gloo works on a host copy)."""
    if world_size == 1:
        return t
    import torch.distributed as dist
    if _is_nccl(group):
        dist.reduce(t, dst=0, op=dist.ReduceOp.MAX, group=group)
        return t
    h = t.cpu()
    dist.reduce(h, dst=0, op=dist.ReduceOp.MAX, group=group)
    return h.to(t.device)


def _is_nccl(group=None):
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl"
