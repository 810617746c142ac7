"""End-to-end slice of do_all_lw.sh on a small synthetic case, every hand-over going through the ABI:

  spectra files (NetCDF classic) -> read_spectrum -> reorder_spectrum -> write_order / read_order
  -> per gas: merged background, gas preparation, band search (2 bands), median sorting variable
  -> overlap_g_points -> merged g-point map -> g-point averaging of each gas + Planck look-up table.

The same chain is run with the CPU oracle (and the reference-built partition search); the index results
(ranks, g-point boundaries, merged g-point map) must be identical, the averaged coefficients agree to 1e-9.
"""
import numpy as np
import pytest
import torch
from scipy.io import netcdf_file

from ecckd_amd import synthetic as syn

pytestmark = pytest.mark.gpu

NLAY, NWAV = 30, 12000
BANDS = (np.array([0.0, 1300.0]), np.array([1300.0, 3260.0]))
TOL, TOLTOL, MAXIT = 0.08, 0.02, 40


def _write_spectrum(path, gas, p, t, wn, od, vmr):
    w = netcdf_file(str(path), "w", version=2)
    for d, n in (("column", 1), ("half_level", NLAY + 1), ("level", NLAY), ("wavenumber", wn.size)):
        w.createDimension(d, n)
    w.createVariable("pressure_hl", "d", ("column", "half_level"))[:] = p[None]
    w.createVariable("temperature_hl", "d", ("column", "half_level"))[:] = t[None]
    w.createVariable("wavenumber", "d", ("wavenumber",))[:] = wn
    w.createVariable("mole_fraction_fl", "d", ("column", "level"))[:] = np.full((1, NLAY), vmr)
    w.createVariable("optical_depth", "f", ("column", "level", "wavenumber"))[:] = od[None]
    w.createVariable("reference_surface_mole_fraction", "d", ())[...] = vmr
    w.constituent_id = gas
    w.close()


def test_files_to_gpoints(ctx, oracle, tmp_path):
    from ecckd_amd import api, ncio
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=ctx.device)
    p = syn.pressure_grid(NLAY)
    t_hl = syn.temperature_profile(p)
    wn, _ = syn.wavenumber_grid(NWAV)
    gases = {"h2o": (41, 30.0, 5e-3), "co2": (43, 8.0, 4e-4)}
    for g, (seed, scale, vmr) in gases.items():
        od = syn.optical_depth(np, p, wn, syn.SEED_BASE + seed, nlines=40, column_scale=scale, dtype="float32")
        _write_spectrum(tmp_path / f"{g}.nc", g, p, t_hl, wn, od, vmr)

    # ---- reorder_spectrum for each gas, through the order file ----
    spec, order = {}, {}
    for g in gases:
        s = ncio.read_spectrum(tmp_path / f"{g}.nc", 0)
        assert s["molecule"] == g
        dwn = s["d_wavenumber_cm_1"]                      # derived: the files carry no d_wavenumber
        od32 = s["optical_depth"].astype(np.float32)
        key, col, iband, rank = api.reorder_spectrum(ctx, s["pressure_hl"], wn, dwn, od32, None, 0.5, BANDS[0], BANDS[1])
        ncio.write_order(tmp_path / f"order_{g}.nc", BANDS[0], BANDS[1], wn, dwn, iband, rank, key, col, molecule=g)
        order[g] = ncio.read_order(tmp_path / f"order_{g}.nc")
        assert np.array_equal(order[g]["rank"], rank)
        # oracle side of the same step
        okey, ocol, _ = oracle.reorder_key(p, oracle.idealised_temperature(p), wn, dwn, s["optical_depth"], None, 0.5)
        _, _, orank = oracle.stable_argsort_bands(wn, key, BANDS[0], BANDS[1])    # same keys -> same stable order
        assert np.allclose(key, okey, rtol=1e-9, atol=1e-12) and np.array_equal(rank, orank)
        spec[g] = dict(s, od32=od32, dwn=dwn, key=key, rank=rank.astype(np.int64), iband=iband)

    nband = 2
    ib = spec["h2o"]["iband"]
    band_rng = [(int(np.nonzero(ib == b)[0][0]), int(np.nonzero(ib == b)[0][-1])) for b in range(nband)]

    # ---- find_g_points per gas (background = the other gas) ----
    n_g_points, medians, gas_gp, ref_gas_gp = [], [], [], []
    first_gas = first_planck = None
    for g, other in (("h2o", "co2"), ("co2", "h2o")):
        s, bg32 = spec[g], spec[other]["od32"]
        d_rank = dev(s["rank"].astype(np.int32))
        # The Planck matrix is evaluated once, on the first gas's reordered grid, and reused as it is by the later gases
        # (find_g_points.cpp:529, :970-984): the first gas handle stays alive and lends its matrix.
        gas = api.GasLW(ctx, p, t_hl, dev(wn), dev(s["dwn"]), d_rank, dev(s["od32"]), dev(bg32), "transmission", 0.02, 0.0,
                        planck_hl_reuse=first_gas.view_ptr("planck_hl")[0] if first_gas is not None else None)
        # oracle: the same preparation on the CPU (find_g_points.cpp:891-1150)
        ireorder = np.empty(NWAV, dtype=np.int64)
        ireorder[s["rank"]] = np.arange(NWAV)
        od_s, bg_s = s["od32"].astype(np.float64)[:, ireorder], bg32.astype(np.float64)[:, ireorder]
        if first_gas is None:
            first_gas, first_planck = gas, oracle.planck_function(t_hl, wn[ireorder], s["dwn"][ireorder])
        planck = first_planck
        fdn, fup = oracle.radiative_transfer_lw(planck, bg_s + od_s, np.ones(NWAV), planck[-1])
        hr = oracle.heating_rate(p, fdn, fup)
        key_s = s["key"][ireorder]
        sv_sorted = api.gather_f64(ctx, dev(s["key"]), api.invert_permutation(ctx, d_rank))
        r1_all, r2_all, med_all, ngp = [], [], [], []
        o_r1, o_r2, o_med = [], [], []
        for b, (i0, i1) in enumerate(band_rng):
            res = gas.find_g_band_ex(i0, i1, TOL, TOLTOL, MAXIT)
            r1_all += list(res["rank1"]); r2_all += list(res["rank2"]); ngp.append(len(res["error"]))
            med_all += list(gas.median_sorting_variable(sv_sorted, res["rank1"], res["rank2"]))
            sl = slice(i0, i1 + 1)
            eq = oracle.CkdEquipartitionLW("transmission", 0.02, oracle.layer_weight(p, 0.0), p, np.ones(i1 - i0 + 1),
                                           planck[-1][sl], fdn[-1][sl].copy(), fup[0][sl].copy(), planck[:, sl], bg_s[:, sl],
                                           oracle.metric("transmission", od_s[:, sl]), hr[:, sl])
            ref = oracle.RefEquipartition(eq.calc_error, resolution=1.0 / (i1 - i0 + 1), partition_tolerance=TOLTOL,
                                          partition_max_iterations=MAXIT)
            st, bnd, err = ref.equipartition_e(TOL)
            npts = i1 - i0 + 1
            for k in range(len(err)):
                a, c = int(np.ceil(bnd[k] * (npts - 1))) + i0, int(np.floor(bnd[k + 1] * (npts - 1))) + i0
                o_r1.append(a); o_r2.append(c)
                o_med.append(oracle.median_sorting_variable(key_s, planck[-1], a, c))
        assert r1_all == o_r1 and r2_all == o_r2, g
        assert np.array_equal(med_all, o_med), g
        n_g_points.append(ngp)
        medians.append(np.array(med_all))
        gas_gp.append(api.gas_g_point(ctx, d_rank, r1_all, r2_all))
        ref_gp = np.full(NWAV, -1, dtype=np.int64)          # SingleGasData::store_g_points (single_gas_data.h:56-62)
        for k, (a, c) in enumerate(zip(o_r1, o_r2)):
            ref_gp[(s["rank"] >= a) & (s["rank"] <= c)] = k
        ref_gas_gp.append(ref_gp)
        assert np.array_equal(gas_gp[-1].cpu().numpy(), ref_gp)
        if gas is not first_gas:
            gas.close()
    first_gas.close()

    # ---- overlap of the gases' g points and the merged map (find_g_points.cpp:1443-1475) ----
    ng, band_number, g_min, g_max = api.overlap_g_points(n_g_points, medians)
    ong, oband, omin, omax = oracle.overlap_g_points(n_g_points, medians)
    assert ng == ong and np.array_equal(band_number, oband) and np.array_equal(g_min, omin) and np.array_equal(g_max, omax)
    g_point, n_unassigned = api.merge_g_points(ctx, gas_gp, g_min, g_max)
    ref_map = np.full(NWAV, -1, dtype=np.int64)
    for ig in range(ng):
        found = np.ones(NWAV, dtype=bool)
        for k in range(2):
            found &= (ref_gas_gp[k] >= g_min[k, ig]) & (ref_gas_gp[k] <= g_max[k, ig])
        ref_map[found] = ig
    assert np.array_equal(g_point.cpu().numpy(), ref_map) and n_unassigned == int((ref_map < 0).sum())
    assert ng >= 4 and n_unassigned == 0

    # ---- the same through the driver mirrors: reorder_spectrum -> order files -> find_g_points -> g-points file ----
    from ecckd_amd import pipeline
    for g in gases:
        pipeline.reorder_spectrum(ctx, tmp_path / f"{g}.nc", tmp_path / f"order2_{g}.nc", BANDS[0], BANDS[1])
        assert np.array_equal(ncio.read_order(tmp_path / f"order2_{g}.nc")["rank"], spec[g]["rank"])
    res = pipeline.find_g_points(ctx, [dict(name="h2o", input=tmp_path / "h2o.nc", reordering_input=tmp_path / "order2_h2o.nc",
                                            background=[dict(path=tmp_path / "co2.nc")]),
                                       dict(name="co2", input=tmp_path / "co2.nc", reordering_input=tmp_path / "order2_co2.nc",
                                            background=[dict(path=tmp_path / "h2o.nc")])],
                                 BANDS[0], BANDS[1], TOL, output_path=tmp_path / "gpoints.nc", tolerance_tolerance=TOLTOL,
                                 max_iterations=MAXIT)
    assert res["ng"] == ng and np.array_equal(res["g_point"], ref_map) and res["n_unassigned"] == 0
    gpf = ncio.read_g_points(tmp_path / "gpoints.nc")
    assert np.array_equal(gpf["g_point"], ref_map) and np.array_equal(gpf["band_number"], band_number)
    with ncio.NcFile(tmp_path / "gpoints.nc") as f:
        assert f.att_text("constituent_id") == "h2o co2" and int(f.read("n_gases")) == 2
        assert np.array_equal(f.read("h2o_rank1"), res["gases"][0]["rank1"]) and f.var_info("co2_g_point")[0] == 3

    # ---- create_look_up_table pieces on the merged map: averaging of each gas + Planck look-up table ----
    dwn = spec["h2o"]["dwn"]
    gmap = api.GPointMap(ctx, g_point, ng, dev(wn), dev(dwn))
    t_fl = 0.5 * (t_hl[1:] + t_hl[:-1])
    for g, (_, _, vmr) in gases.items():
        k_abs, k_min, k_max = gmap.average_optical_depth(p, dev(spec[g]["od32"]), "transmission", temperature_fl=t_fl,
                                                         reference_surface_vmr=vmr)
        planck_fl = oracle.planck_function(t_fl, wn, dwn)
        ok, omn, omx, n_empty = oracle.average_optical_depth_to_g_point(ng, vmr, p, ref_map, spec[g]["od32"].astype(np.float64),
                                                                       planck_fl, "transmission")
        assert n_empty == 0
        assert np.allclose(k_abs, ok, rtol=1e-9, atol=1e-300) and np.allclose(k_min, omn, rtol=1e-12) and np.allclose(k_max, omx, rtol=1e-12)
    t_lut = np.arange(120.0, 351.0, 1.0)
    got = gmap.planck_lut(t_lut)
    want = oracle.planck_lut(ng, t_lut, ref_map, wn, dwn)
    assert np.allclose(got, want, rtol=1e-11)
    gmap.close()


def test_create_look_up_table_from_files(ctx, oracle, tmp_path):
    """create_look_up_table.cpp:225-606 through ecckd_amd.pipeline: every gas type, merged well-mixed background,
    empty g point removal, gpoint_fraction, Planck look-up table; then the CKD file round trip and run_ckd."""
    from ecckd_amd import api, ncio, pipeline
    rs = np.random.RandomState(7)
    nlay, nwav, ncol = 12, 6000, 3
    p1 = syn.pressure_grid(nlay)
    wn, _ = syn.wavenumber_grid(nwav)
    p = np.tile(p1, (ncol, 1))
    t = np.stack([syn.temperature_profile(p1) + 15.0 * (c - 1) for c in range(ncol)])

    def write(path, gas, seed, scale, vmr):
        w = netcdf_file(str(path), "w", version=2)
        for d, n in (("column", ncol), ("half_level", nlay + 1), ("level", nlay), ("wavenumber", nwav)):
            w.createDimension(d, n)
        od = np.stack([syn.optical_depth(np, p1, wn, syn.SEED_BASE + seed, nlines=30, column_scale=scale * (1 + 0.1 * c),
                                         dtype="float32") for c in range(ncol)])
        w.createVariable("pressure_hl", "d", ("column", "half_level"))[:] = p
        w.createVariable("temperature_hl", "d", ("column", "half_level"))[:] = t
        w.createVariable("wavenumber", "d", ("wavenumber",))[:] = wn
        w.createVariable("mole_fraction_fl", "d", ("column", "level"))[:] = np.full((ncol, nlay), vmr)
        w.createVariable("optical_depth", "f", ("column", "level", "wavenumber"))[:] = od
        w.createVariable("reference_surface_mole_fraction", "d", ())[...] = vmr
        w.constituent_id = gas
        w.close()
        return od.astype(np.float64)

    files = {}
    for name, seed, scale, vmr in (("o2", 1, 0.5, 0.209), ("n2", 2, 0.2, 0.781), ("co2", 3, 8.0, 4e-4), ("ch4", 4, 2.0, 1.8e-6),
                                   ("h2o_a", 5, 3.0, 1e-3), ("h2o_b", 6, 30.0, 1e-2)):
        files[name] = (tmp_path / f"{name}.nc", write(tmp_path / f"{name}.nc", name.split("_")[0], seed, scale, vmr), vmr)
    # a g-point map with one empty g point (5) to be removed
    g_point = rs.randint(0, 7, nwav).astype(np.int32)
    g_point[g_point == 5] = 6
    band_number = np.array([0, 0, 0, 1, 1, 1, 1])
    gases = [dict(name="composite", conc="none", inputs=[dict(path=files["o2"][0]), dict(path=files["n2"][0], scaling=0.5)]),
             dict(name="co2", conc="linear", inputs=[files["co2"][0]]),
             dict(name="ch4", conc="relative-linear", inputs=[files["ch4"][0]], reference_conc=1.8e-6),
             dict(name="h2o", conc="lut", inputs=[files["h2o_a"][0], files["h2o_b"][0]])]
    model = pipeline.create_look_up_table(ctx, g_point, band_number, [0.0, 1300.0], [1300.0, 3260.0], gases)
    # ---- oracle ----
    gp = g_point.copy()
    gp[gp == 6] = 5                                               # renumbered
    assert model["ng"] == 6 and np.array_equal(model["iband_per_g"], [0, 1, 2, 3, 4, 6])    # :142 keeps the OLD g index
    dwn = np.empty(nwav); dwn[1:-1] = 0.5 * (wn[2:] - wn[:-2]); dwn[0] = 0.5 * dwn[1]; dwn[-1] = 0.5 * dwn[-2]
    t_fl = (t[:, :-1] * p[:, :-1] + t[:, 1:] * p[:, 1:]) / (p[:, :-1] + p[:, 1:])
    assert np.allclose(model["temperature"], t_fl, rtol=1e-15) and np.allclose(model["log_pressure"], np.log(0.5 * (p1[1:] + p1[:-1])))

    def avg(od, vmr, c):
        planck = oracle.planck_function(t_fl[c], wn, dwn)
        return oracle.average_optical_depth_to_g_point(6, vmr, p1, gp, od, planck, "transmission")[:3]

    by_name = {g["name"]: g for g in model["gases"]}
    for c in range(ncol):
        merged = files["o2"][1][c] + files["n2"][1][c] * 0.5
        for got, want in zip((by_name["composite"][k][c] for k in ("molar_abs", "min_molar_abs", "max_molar_abs")), avg(merged, 1.0, c)):
            assert np.allclose(got, want, rtol=1e-9, atol=1e-300)
        assert np.allclose(by_name["co2"]["molar_abs"][c], avg(files["co2"][1][c], 4e-4, c)[0], rtol=1e-9, atol=1e-300)
        assert np.allclose(by_name["ch4"]["molar_abs"][c], avg(files["ch4"][1][c], 1.8e-6, c)[0], rtol=1e-9, atol=1e-300)
        for ic, key in enumerate(("h2o_a", "h2o_b")):
            assert np.allclose(by_name["h2o"]["molar_abs"][ic, c], avg(files[key][1][c], files[key][2], c)[0], rtol=1e-9, atol=1e-300)
    assert np.array_equal(by_name["h2o"]["vmr"], [1e-3, 1e-2]) and by_name["ch4"]["reference_vmr"] == 1.8e-6
    assert model["wavenumber1"][0] == 0.0 and model["wavenumber2"][-1] == 3260.0 and model["wavenumber1"].size == 326
    assert np.allclose(model["gpoint_fraction"], oracle.gpoint_fraction(6, gp, wn, dwn, model["wavenumber1"], model["wavenumber2"]), rtol=1e-12)
    assert np.allclose(model["planck_function"], oracle.planck_lut(6, np.arange(120.0, 351.0), gp, wn, dwn), rtol=1e-11)
    # ---- CKD file round trip, then the model runs ----
    model["iband_per_g"] = np.array([0, 0, 0, 1, 1, 1], dtype=np.int32)      # a consistent band map for the evaluation below
    path = tmp_path / "ckd.nc"
    ncio.write_ckd_model(str(path), model, model_id="test")
    back = ncio.read_ckd_model(str(path))
    scene = dict(pressure_hl=p, temperature_hl=t, vmr_fl=np.stack([np.ones((ncol, nlay)), np.full((ncol, nlay), 4e-4),
                                                                  np.full((ncol, nlay), 2.0e-6), np.full((ncol, nlay), 3e-3)], axis=1))
    out = api.run_ckd(ctx, back, scene, per_gas=False)
    assert np.all(np.isfinite(out["flux_dn_lw"])) and out["flux_up_lw"][:, -1].min() > 0 and out["optical_depth"].min() >= 0


def make_optimize_files(ctx, tmp_path, boundary=False):
    """A raw CKD definition (raw.nc) and two LBL band-flux training files made with run_ckd from a perturbed "truth".
    boundary=True adds the high-resolution surface / TOA fluxes (5 wavenumbers per g point) and gpoints.nc."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    import ckd_synth
    from ecckd_amd import api, ncio, pipeline
    model = ckd_synth.make_model(seed=12)
    ng = len(model["iband_per_g"])
    nband = model["nband"]
    # spectral description of the model: one 10 cm-1 interval per g point, bands as contiguous g ranges
    gf = np.eye(ng)
    ib = model["iband_per_g"]
    model.update(wavenumber1=np.arange(ng) * 10.0, wavenumber2=np.arange(1, ng + 1) * 10.0, gpoint_fraction=gf,
                 wavenumber1_band=np.array([10.0 * np.nonzero(ib == b)[0][0] for b in range(nband)]),
                 wavenumber2_band=np.array([10.0 * (np.nonzero(ib == b)[0][-1] + 1) for b in range(nband)]))
    ncio.write_ckd_model(str(tmp_path / "raw.nc"), model)
    truth = dict(model, gases=[dict(g, molar_abs=g["molar_abs"] * np.exp(0.2 * np.random.RandomState(i).normal(size=g["molar_abs"].shape)))
                               for i, g in enumerate(model["gases"])])
    scenes = ckd_synth.make_scenes(model, nscene=2, ncol=4, nlay=16)
    names = [g["name"] for g in model["gases"]]
    paths = []
    for k, sc in enumerate(scenes):
        sc = dict(sc, gas_present=None)
        out = api.run_ckd(ctx, truth, sc, per_gas=False)
        band = lambda a: np.stack([a[..., ib == b].sum(-1) for b in range(nband)], axis=-1)
        w = netcdf_file(str(tmp_path / f"lbl{k}.nc"), "w", version=2)
        for d, n in (("column", 4), ("half_level", 17), ("level", 16), ("gas", len(names)), ("band", nband)):
            w.createDimension(d, n)
        for name, dims, a in (("pressure_hl", ("column", "half_level"), sc["pressure_hl"]),
                              ("temperature_hl", ("column", "half_level"), sc["temperature_hl"]),
                              ("mole_fraction_fl", ("column", "gas", "level"), sc["vmr_fl"]),
                              ("flux_dn_lw", ("column", "half_level"), out["flux_dn_lw"]),
                              ("flux_up_lw", ("column", "half_level"), out["flux_up_lw"]),
                              ("band_flux_dn_lw", ("column", "half_level", "band"), band(out["spectral_flux_dn_lw"])),
                              ("band_flux_up_lw", ("column", "half_level", "band"), band(out["spectral_flux_up_lw"])),
                              ("band_wavenumber1_lw", ("band",), model["wavenumber1_band"]),
                              ("band_wavenumber2_lw", ("band",), model["wavenumber2_band"])):
            w.createVariable(name, "d", dims)[:] = a
        if boundary:
            K = 5
            share = np.array([0.1, 0.3, 0.2, 0.25, 0.15])                        # how a g point's flux spreads over its wavenumbers
            w.createDimension("wavenumber", ng * K)
            hi = lambda a: (a[:, :, None] * share[None, None, :]).reshape(a.shape[0], ng * K)
            w.createVariable("spectral_flux_dn_surf_lw", "d", ("column", "wavenumber"))[:] = hi(out["spectral_flux_dn_lw"][:, -1, :])
            w.createVariable("spectral_flux_up_toa_lw", "d", ("column", "wavenumber"))[:] = hi(out["spectral_flux_up_lw"][:, 0, :])
        w.constituent_id = " ".join(names)
        w.close()
        paths.append(str(tmp_path / f"lbl{k}.nc"))
    if boundary:
        wn_hi = (np.arange(ng * 5) + 0.5) * 2.0
        ncio.write_g_points(str(tmp_path / "gpoints.nc"), model["wavenumber1_band"], model["wavenumber2_band"], ib, [], wn_hi,
                            np.repeat(np.arange(ng), 5))
    return model, truth, scenes, paths, ib, names


def test_optimize_lut_from_files(ctx, oracle, tmp_path):
    """optimize_lut.cpp driver through ecckd_amd.pipeline: CKD file + LBL band-flux files (made with run_ckd from a
    "truth" model) -> bounded L-BFGS -> CKD file; the cost falls and the written file reproduces the optimised fluxes."""
    from ecckd_amd import api, ncio, pipeline
    model, truth, scenes, paths, ib, names = make_optimize_files(ctx, tmp_path)
    raw = ncio.read_ckd_model(str(tmp_path / "raw.nc"), active_gases=["composite", "h2o", "co2", "ch4"])
    assert np.array_equal(pipeline.iband_per_g(raw, raw["wavenumber1_band"], raw["wavenumber2_band"]), ib)
    cfg = dict(flux_weight=0.2, flux_profile_weight=0.05, broadband_weight=0.4, prior_error=4.0, pressure_corr=0.95,
               temperature_corr=0.95, conc_corr=0.9)
    first = api.Optimizer(ctx, dict(raw), [pipeline._scene_for_optimizer(ncio.read_lbl_fluxes(q, names), False) for q in paths],
                          **dict(cfg, cap_relative_linear=0.8))
    J0 = first.cost_grad(first.initial_state(), False)
    first.close()
    opt_model, res = pipeline.optimize_lut(ctx, raw, paths, max_iterations=60, **cfg)
    assert res["status"] in (0, 2) and res["cost"] < 0.5 * J0
    ncio.write_ckd_model(str(tmp_path / "opt.nc"), opt_model, model_id="optimised")
    back = ncio.read_ckd_model(str(tmp_path / "opt.nc"))
    sc0 = dict(scenes[0], gas_present=None)
    a = api.run_ckd(ctx, opt_model, sc0, per_gas=False)["flux_dn_lw"]
    b = api.run_ckd(ctx, back, sc0, per_gas=False)["flux_dn_lw"]
    assert np.allclose(a, b, rtol=1e-5)                               # the file stores FLOAT coefficients
    # closer to the truth than the raw model
    tr = api.run_ckd(ctx, truth, sc0, per_gas=False)["flux_dn_lw"]
    r0 = api.run_ckd(ctx, raw, sc0, per_gas=False)["flux_dn_lw"]
    assert np.abs(a - tr).max() < np.abs(r0 - tr).max()


def make_do_all_inputs(ctx, tmp_path):
    """Inputs of the do_all_lw chain on a small synthetic problem: "present" spectra of two gases (1 column), "idealised"
    spectra (3 temperature columns, water vapour also at 4x), and a line-by-line training file (lbl.nc) with the band fluxes
    of three evaluation columns computed by the LBL stand-in (ecckd_lbl_band_fluxes_lw)."""
    from ecckd_amd import api, ncio
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=ctx.device)
    nlay, nwav = 16, 8000
    p1 = syn.pressure_grid(nlay)
    wn, _ = syn.wavenumber_grid(nwav)
    bands = (np.array([0.0, 1300.0]), np.array([1300.0, 3260.0]))
    base = {"h2o": (syn.optical_depth(np, p1, wn, syn.SEED_BASE + 61, nlines=40, column_scale=40.0, dtype="float32"), 5e-3),
            "co2": (syn.optical_depth(np, p1, wn, syn.SEED_BASE + 63, nlines=30, column_scale=10.0, dtype="float32"), 4e-4)}

    def write(path, gas, temps, factor=1.0):
        ncol = len(temps)
        w = netcdf_file(str(path), "w", version=2)
        for d, n in (("column", ncol), ("half_level", nlay + 1), ("level", nlay), ("wavenumber", nwav)):
            w.createDimension(d, n)
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
        write(tmp_path / f"present_{g}.nc", g, [t0])
        write(tmp_path / f"ideal_{g}.nc", g, ideal_t)
    write(tmp_path / "ideal_h2o_x4.nc", "h2o", ideal_t, factor=4.0)
    # line-by-line training fluxes of three evaluation columns (other temperatures, other gas amounts)
    ncol = 3
    T = np.stack([t0 - 8.0, t0 + 3.0, t0 + 11.0])
    amount = {"h2o": np.array([0.7, 1.5, 3.0]), "co2": np.array([1.0, 2.0, 0.5])}
    begin = [int(np.nonzero((wn >= a) & (wn < b + (b == 3260.0)))[0][0]) for a, b in zip(*bands)]
    end = [int(np.nonzero((wn >= a) & (wn < b + (b == 3260.0)))[0][-1]) for a, b in zip(*bands)]
    dwn = ncio.read_spectrum(tmp_path / "present_h2o.nc")["d_wavenumber_cm_1"]
    bdn, bup = [], []
    for c in range(ncol):
        od = sum(base[g][0].astype(np.float64) * amount[g][c] for g in base)
        dn, up = api.lbl_band_fluxes_lw(ctx, T[c], dev(wn), dev(dwn), dev(od), begin, end)
        bdn.append(dn.T); bup.append(up.T)
    bdn, bup = np.stack(bdn), np.stack(bup)                       # (ncol, nhl, nband)
    w = netcdf_file(str(tmp_path / "lbl.nc"), "w", version=2)
    for d, n in (("column", ncol), ("half_level", nlay + 1), ("level", nlay), ("gas", 2), ("band", 2)):
        w.createDimension(d, n)
    vmr = np.stack([np.stack([np.full(nlay, base[g][1] * amount[g][c]) for g in ("h2o", "co2")]) for c in range(ncol)])
    for name, dims, a in (("pressure_hl", ("column", "half_level"), np.tile(p1, (ncol, 1))), ("temperature_hl", ("column", "half_level"), T),
                          ("mole_fraction_fl", ("column", "gas", "level"), vmr), ("flux_dn_lw", ("column", "half_level"), bdn.sum(-1)),
                          ("flux_up_lw", ("column", "half_level"), bup.sum(-1)), ("band_flux_dn_lw", ("column", "half_level", "band"), bdn),
                          ("band_flux_up_lw", ("column", "half_level", "band"), bup), ("band_wavenumber1_lw", ("band",), bands[0]),
                          ("band_wavenumber2_lw", ("band",), bands[1])):
        w.createVariable(name, "d", dims)[:] = a
    w.constituent_id = "h2o co2"
    w.close()
    return dict(p1=p1, wn=wn, bands=bands, base=base, ncol=ncol, T=T, vmr=vmr, bdn=bdn, bup=bup, nlay=nlay)


def hr_error_against_lbl(oracle, inp, flux_dn, flux_up):
    """Heating-rate RMS error (plot/calc_hr_error.m) of broadband fluxes (ncol, nhl) against the line-by-line ones."""
    from test_run_ckd_gpu import calc_hr_error
    p1, ncol = inp["p1"], inp["ncol"]
    hr_of = lambda dn, up: np.stack([oracle.heating_rate(p1, dn[c][:, None], up[c][:, None])[:, 0] for c in range(ncol)]) * 86400.0
    P = np.tile(p1, (ncol, 1))
    return calc_hr_error(P.T / 100.0, hr_of(flux_dn, flux_up).T, hr_of(inp["bdn"].sum(-1), inp["bup"].sum(-1)).T)


def test_do_all_lw_synthetic(ctx, oracle, tmp_path):
    """The whole chain of test/do_all_lw.sh on a small synthetic problem, every step through the driver mirrors:
    reorder_spectrum -> find_g_points -> create_look_up_table -> LBL training fluxes -> optimize_lut -> run_ckd,
    judged by the heating-rate RMS error of plot/calc_hr_error.m against the line-by-line fluxes."""
    from ecckd_amd import api, ncio, pipeline
    inp = make_do_all_inputs(ctx, tmp_path)
    bands, p1, ncol = inp["bands"], inp["p1"], inp["ncol"]
    # 1-2. reorder each gas, partition
    for g in inp["base"]:
        pipeline.reorder_spectrum(ctx, tmp_path / f"present_{g}.nc", tmp_path / f"order_{g}.nc", bands[0], bands[1])
    gp = pipeline.find_g_points(ctx, [dict(name="h2o", input=tmp_path / "present_h2o.nc", reordering_input=tmp_path / "order_h2o.nc",
                                           background=[dict(path=tmp_path / "present_co2.nc")]),
                                      dict(name="co2", input=tmp_path / "present_co2.nc", reordering_input=tmp_path / "order_co2.nc",
                                           background=[dict(path=tmp_path / "present_h2o.nc")])],
                                bands[0], bands[1], 0.3, output_path=tmp_path / "gpoints.nc", max_iterations=30)
    assert 4 <= gp["ng"] <= 40 and gp["n_unassigned"] == 0
    gpf = ncio.read_g_points(tmp_path / "gpoints.nc")
    # 3. raw look-up table
    raw = pipeline.create_look_up_table(ctx, gpf["g_point"], gpf["band_number"], bands[0], bands[1],
                                        [dict(name="h2o", conc="lut", inputs=[tmp_path / "ideal_h2o.nc", tmp_path / "ideal_h2o_x4.nc"]),
                                         dict(name="co2", conc="linear", inputs=[tmp_path / "ideal_co2.nc"])])
    ncio.write_ckd_model(str(tmp_path / "raw_ckd.nc"), raw)
    # 4-5. optimise against the line-by-line training fluxes
    model = ncio.read_ckd_model(str(tmp_path / "raw_ckd.nc"))
    opt_model, res = pipeline.optimize_lut(ctx, model, [str(tmp_path / "lbl.nc")], max_iterations=80, flux_weight=0.2,
                                           flux_profile_weight=0.05, broadband_weight=0.5, prior_error=8.0)
    print("optimize_lut:", {k: v for k, v in res.items() if k != "x"})
    assert res["status"] in (0, 2, 3)                            # optimize_lut.cpp:315-324 fails only for status >= 6
    # 6. evaluate both models against the line-by-line heating rates
    scene = dict(pressure_hl=np.tile(p1, (ncol, 1)), temperature_hl=inp["T"], vmr_fl=inp["vmr"])
    errs = {}
    for tag, m in (("raw", model), ("optimised", opt_model)):
        out = api.run_ckd(ctx, m, scene, per_gas=False)
        errs[tag] = hr_error_against_lbl(oracle, inp, out["flux_dn_lw"], out["flux_up_lw"])
    print("heating-rate RMS error (K/day):", errs)
    assert np.isfinite(errs["raw"]) and errs["optimised"] < 0.9 * errs["raw"]
    assert errs["optimised"] < 1.0                               # K/day on this toy problem (tolerance 0.3 K/day per gas and band)


def test_find_g_points_sw_from_files(ctx, oracle, tmp_path):
    """Shortwave branch of the find_g_points driver (find_g_points.cpp:655-1660 with `ssi`): two bands either side of
    max_no_rayleigh_wavenumber (band albedo 0.15 / 0), total-transmission averaging, solar-weighted medians; the band
    searches are replayed by the CPU oracle + the reference-built partition search and must agree index for index."""
    from ecckd_amd import api, ncio, pipeline
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    nwav, lo, hi, mu0 = 16000, 250.0, 50000.0, 0.5
    b1, b2 = np.array([lo, 10000.0]), np.array([10000.0, hi])
    p = syn.pressure_grid(NLAY)
    t_hl = syn.temperature_profile(p)
    wn, dwn = syn.wavenumber_grid(nwav, lo, hi)
    ssi = syn.solar_spectral_irradiance(wn, dwn)
    gases = {"h2o": (61, 5.0, 5e-3), "o3": (67, 1.5, 1e-6)}
    for g, (seed, scale, vmr) in gases.items():
        od = syn.optical_depth(np, p, wn, syn.SEED_BASE + seed, nlines=40, column_scale=scale, dtype="float32", lo=lo, hi=hi)
        _write_spectrum(tmp_path / f"{g}.nc", g, p, t_hl, wn, od, vmr)
        pipeline.reorder_spectrum(ctx, tmp_path / f"{g}.nc", tmp_path / f"order_{g}.nc", b1, b2, ssi=ssi)
    specs = [dict(name="h2o", input=tmp_path / "h2o.nc", reordering_input=tmp_path / "order_h2o.nc",
                  background=[dict(path=tmp_path / "o3.nc")]),
             dict(name="o3", input=tmp_path / "o3.nc", reordering_input=tmp_path / "order_o3.nc",
                  background=[dict(path=tmp_path / "h2o.nc")])]
    tol = 0.03
    res = pipeline.find_g_points(ctx, specs, b1, b2, tol, output_path=tmp_path / "gpoints_sw.nc", averaging_method="total-transmission",
                                 tolerance_tolerance=TOLTOL, max_iterations=MAXIT, ssi=ssi)
    assert res["n_unassigned"] == 0 and res["ng"] >= 4

    # ---- oracle replay of every band search (find_g_points.cpp:891-1410) ----
    albedo = np.where(wn < 10000.0, 0.15, 0.0)
    lw = oracle.layer_weight(p, 0.0)
    ods = {g: ncio.read_spectrum(tmp_path / f"{g}.nc", 0)["optical_depth"] for g in gases}
    for k, (g, other) in enumerate((("h2o", "o3"), ("o3", "h2o"))):
        order = ncio.read_order(tmp_path / f"order_{g}.nc")
        rank, iband = order["rank"].astype(np.int64), order["band_number"]
        ireorder = np.empty(nwav, dtype=np.int64)
        ireorder[rank] = np.arange(nwav)
        od_s, bg_s, ssi_s, alb_s = ods[g][:, ireorder], ods[other][:, ireorder], ssi[ireorder], albedo[ireorder]
        key_s = order["sorting_variable"][ireorder]
        fdn = oracle.radiative_transfer_direct_sw(mu0, ssi_s, bg_s + od_s)
        hr = oracle.heating_rate(p, fdn, None)
        ex = dict(min_scaling=0.5, max_scaling=2.5)
        for tag, sc in (("low", 0.5), ("high", 2.5)):
            d, u = oracle.radiative_transfer_norayleigh_sw(mu0, ssi_s, bg_s + sc * od_s, alb_s)
            ex[f"flux_up_toa_{tag}"], ex[f"flux_dn_surf_{tag}"] = u[0].copy(), d[-1].copy()
            ex[f"hr_{tag}"] = oracle.heating_rate(p, d, None)
        metric = oracle.metric("total-transmission", od_s)
        o_r1, o_r2, o_med = [], [], []
        for b in range(2):
            idx = np.nonzero(iband == b)[0]
            i0, i1 = int(idx[0]), int(idx[-1])
            sl, npts = slice(i0, i1 + 1), i1 - i0 + 1
            exb = {kk: (v[..., sl] if isinstance(v, np.ndarray) else v) for kk, v in ex.items()}
            eq = oracle.CkdEquipartitionSW("total-transmission", 0.02, lw, mu0, p, ssi_s[sl], 0.15 if b == 0 else 0.0,
                                           fdn[-1][sl].copy(), np.zeros(npts), bg_s[:, sl], metric[:, sl], hr[:, sl], exb)
            ref = oracle.RefEquipartition(eq.calc_error, resolution=1.0 / npts, partition_tolerance=TOLTOL,
                                          partition_max_iterations=MAXIT)
            st, bnd, err = ref.equipartition_e(tol)
            for j in range(len(err)):
                a, c = int(np.ceil(bnd[j] * (npts - 1))) + i0, int(np.floor(bnd[j + 1] * (npts - 1))) + i0
                o_r1.append(a); o_r2.append(c)
                o_med.append(oracle.median_sorting_variable(key_s, ssi_s, a, c))
        got = res["gases"][k]
        assert list(got["rank1"]) == o_r1 and list(got["rank2"]) == o_r2, g
        assert np.array_equal(got["sorting_variable"], o_med), g
    back = ncio.read_g_points(tmp_path / "gpoints_sw.nc")
    assert np.array_equal(back["g_point"], res["g_point"])
