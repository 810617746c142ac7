"""The NetCDF classic reader / writer (file parts of rows a1, a9, a21) against an independent implementation
(scipy.io.netcdf_file) in both directions, and on the reference's own classic data file
(data/mie_droplet_scattering.nc, read in place when the reference tree is present)."""
import os

import numpy as np
import pytest
from scipy.io import netcdf_file

from ecckd_amd import ncio, EcckdError

# the reference's own NetCDF-classic data file, read where it lies (not copied into the repo); absent on the GPU box
GOLDEN = "/root/reference/data/mie_droplet_scattering.nc"
needs_reference = pytest.mark.skipif(not os.path.exists(GOLDEN), reason="reference tree not present")


@needs_reference
def test_reads_the_reference_data_file():
    ref = netcdf_file(GOLDEN, "r", mmap=False)
    with ncio.NcFile(GOLDEN) as f:
        assert f.dim("effective_radius") == 50 and f.dim("wavenumber") == 396
        for name, v in ref.variables.items():
            t, shape = f.var_info(name)
            assert shape == v.shape and t == 5                     # all FLOAT
            assert np.array_equal(f.read(name), np.asarray(v.data, dtype=np.float64))
            assert f.att_text("long_name", name) == v.long_name.decode()
        assert np.array_equal(f.read("asymmetry_factor", 7), ref.variables["asymmetry_factor"].data[7].astype(np.float64))
        assert f.att_text("title") == ref.title.decode()
        assert f.att_text("units", "wavenumber") == "cm-1"
        assert f.att_text("no_such_attribute") is None and not f.exist("no_such_variable")
    ref.close()


@pytest.mark.parametrize("version", [1, 2])
def test_reads_what_scipy_writes_including_records(tmp_path, version):
    rs = np.random.RandomState(version)
    p = str(tmp_path / "s.nc")
    w = netcdf_file(p, "w", version=version)
    w.createDimension("time", None)
    w.createDimension("x", 5)
    w.createDimension("y", 3)
    w.history = "made by scipy"
    w.scale = np.array([1.5, -2.5])
    data = {}
    for name, tc, dims in (("a", "d", ("x", "y")), ("b", "f", ("x",)), ("c", "i", ("y",)), ("d", "h", ("x",)),
                           ("e", "b", ("y",)), ("r1", "d", ("time", "x")), ("r2", "h", ("time",)), ("s", "f", ())):
        v = w.createVariable(name, tc, dims)
        shape = tuple(4 if d == "time" else {"x": 5, "y": 3}[d] for d in dims)
        arr = (rs.uniform(-100, 100, shape)).astype(tc)
        if shape == ():
            v[...] = arr
        else:
            v[:] = arr
        v.units = "unit_" + name
        data[name] = np.asarray(arr, dtype=np.float64)
    w.close()
    # the independent READER is the oracle (scipy's writer pads short record variables inconsistently)
    r = netcdf_file(p, "r", mmap=False)
    data = {k: np.asarray(v.data if v.shape else v.getValue(), dtype=np.float64) for k, v in r.variables.items()}
    r.close()
    with ncio.NcFile(p) as f:
        assert f.dim("time") == 4
        for name, arr in data.items():
            assert np.array_equal(f.read(name), arr), name
            assert f.att_text("units", name) == "unit_" + name
        assert np.array_equal(f.read("r1", 2), data["r1"][2]) and np.array_equal(f.read("a", 4), data["a"][4])
        assert f.att_text("history") == "made by scipy"
        assert np.array_equal(f.att_values("scale"), [1.5, -2.5])


def test_scipy_reads_what_the_writer_writes(tmp_path):
    rs = np.random.RandomState(3)
    p = str(tmp_path / "w.nc")
    w = ncio.NcWriter(p)
    w.define_dimension("level", 4)
    w.define_dimension("g_point", 6)
    want = {}
    for name, t, dims in (("od", "float", ("level", "g_point")), ("k", "double", ("level", "g_point")), ("n", "int", ("g_point",)),
                          ("band", "short", ("g_point",)), ("flag", "byte", ("level",)), ("scalar", "double", ())):
        w.define_variable(name, t, *dims)
        w.write_attribute("long_name", "the " + name, var=name)
        shape = tuple({"level": 4, "g_point": 6}[d] for d in dims)
        want[name] = np.round(rs.uniform(-100, 100, shape), 0 if t in ("int", "short", "byte") else 6)
    w.write_attribute("title", "writer test")
    w.write_attribute("valid_range", [0.0, 1.0], var="od")
    w.end_define_mode()
    for name, arr in want.items():
        w.write(name, arr)
    w.close()
    r = netcdf_file(p, "r", mmap=False)
    assert r.version_byte == 1 and r.title == b"writer test"
    assert r.variables["od"].typecode() == "f" and r.variables["band"].typecode() == "h" and r.variables["flag"].typecode() == "b"
    for name, arr in want.items():
        got = np.asarray(r.variables[name].data if arr.shape else r.variables[name].getValue(), dtype=np.float64)
        expect = arr.astype(np.float32).astype(np.float64) if name == "od" else arr
        assert np.array_equal(got, expect), name
        assert r.variables[name].long_name == ("the " + name).encode()
    assert np.array_equal(r.variables["od"].valid_range, [0.0, 1.0])
    r.close()


def test_write_order_file(tmp_path):
    """write_order.cpp:24-143: names, external types, attributes; read back by scipy and by read_order."""
    rs = np.random.RandomState(4)
    n = 1000
    wn = np.linspace(0.0, 3260.0, n)
    rank = rs.permutation(n).astype(np.int32)
    iband = np.where(wn < 1500.0, 0, 1).astype(np.int16)
    iband[:3] = -1
    key = rs.uniform(-0.5, 12.0, n)
    col = rs.lognormal(0, 3, n)
    p = str(tmp_path / "order.nc")
    ncio.write_order(p, [0.0, 1500.0], [1500.0, 3260.0], wn, np.full(n, wn[1] - wn[0]), iband, rank, key, col,
                     molecule="h2o", config_str="iprofile 0\n", history="today: reorder_spectrum x.cfg")
    r = netcdf_file(p, "r", mmap=False)
    assert r.dimensions == {"band": 2, "wavenumber": n}
    types = {k: v.typecode() for k, v in r.variables.items()}
    assert types == dict(wavenumber1_band="f", wavenumber2_band="f", wavenumber="d", d_wavenumber="f", band_number="h",
                         rank="i", column_optical_depth="f", sorting_variable="f")
    assert r.title == b"Optimal reordering of the absorption spectrum of H2O" and r.molecule == b"h2o"
    assert r.config == b"iprofile 0\n" and r.history.startswith(b"today")
    assert r.variables["rank"].long_name == b"Rank when reordered" and r.variables["wavenumber"].units == b"cm-1"
    assert b"rank(i) provides the rank of wavenumber i." in r.variables["rank"].comment
    assert np.array_equal(r.variables["rank"].data, rank) and np.array_equal(r.variables["band_number"].data, iband)
    assert np.array_equal(r.variables["wavenumber"].data, wn)
    assert np.array_equal(r.variables["sorting_variable"].data, key.astype(np.float32))
    r.close()
    o = ncio.read_order(p)
    assert np.array_equal(o["rank"], rank) and np.array_equal(o["band_number"], iband) and o["molecule"] == "h2o"
    assert np.array_equal(o["sorting_variable"], key.astype(np.float32).astype(np.float64))
    # without column optical depth and molecule (:88, :115-118)
    ncio.write_order(p, [0.0], [3260.0], wn, np.full(n, 1.0), iband, rank, key)
    r = netcdf_file(p, "r", mmap=False)
    assert "column_optical_depth" not in r.variables and r.title == b"Optimal reordering of the absorption spectrum of a gas"
    r.close()


def test_read_spectrum(tmp_path):
    """read_spectrum.cpp:20-87 on a CKDMIP-shaped classic file written by scipy."""
    rs = np.random.RandomState(5)
    ncol, nlev, nwav = 3, 6, 400
    p = str(tmp_path / "spectra.nc")
    w = netcdf_file(p, "w", version=2)
    for d, n in (("column", ncol), ("half_level", nlev + 1), ("level", nlev), ("wavenumber", nwav)):
        w.createDimension(d, n)
    arrs = dict(pressure_hl=("f", ("column", "half_level"), np.sort(rs.uniform(1, 1e5, (ncol, nlev + 1)), axis=1)),
                temperature_hl=("f", ("column", "half_level"), rs.uniform(190, 300, (ncol, nlev + 1))),
                wavenumber=("d", ("wavenumber",), np.cumsum(rs.uniform(1e-3, 2e-3, nwav))),
                mole_fraction_fl=("f", ("column", "level"), rs.uniform(1e-6, 1e-2, (ncol, nlev))),
                optical_depth=("f", ("column", "level", "wavenumber"), rs.lognormal(-2, 2, (ncol, nlev, nwav))))
    for name, (tc, dims, a) in arrs.items():
        w.createVariable(name, tc, dims)[:] = a.astype(tc)
    w.createVariable("reference_surface_mole_fraction", "f", ())[...] = np.float32(4.15e-4)
    w.constituent_id = "co2"
    w.close()
    s = ncio.read_spectrum(p, iprofile=1)
    assert s["ncol"] == ncol and s["molecule"] == "co2"
    assert s["reference_surface_vmr"] == float(np.float32(4.15e-4))
    for name, key in (("pressure_hl", "pressure_hl"), ("temperature_hl", "temperature_hl"), ("mole_fraction_fl", "vmr_fl"),
                      ("optical_depth", "optical_depth")):
        assert np.array_equal(s[key], arrs[name][2].astype(arrs[name][0])[1].astype(np.float64)), name
    wn = arrs["wavenumber"][2]
    assert np.array_equal(s["wavenumber_cm_1"], wn)
    assert np.array_equal(s["d_wavenumber_cm_1"][1:-1], 0.5 * (wn[2:] - wn[:-2]))          # :58-65
    assert s["d_wavenumber_cm_1"][0] == 0.5 * s["d_wavenumber_cm_1"][1]


def test_rejects_hdf5_and_bad_calls(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    with pytest.raises(EcckdError) as e:
        ncio.NcFile(str(p))
    assert e.value.code == 147
    with pytest.raises(EcckdError):
        ncio.NcFile(str(tmp_path / "missing.nc"))
    q = str(tmp_path / "small.nc")
    w = ncio.NcWriter(q)
    w.define_dimension("wavenumber", 7)
    w.define_variable("wavenumber", "double", "wavenumber")
    w.end_define_mode()
    w.write("wavenumber", np.arange(7.0))
    w.close()
    with ncio.NcFile(q) as f:
        with pytest.raises(EcckdError):
            f.read("wavenumber", 7)                              # slice outside the slowest dimension


@pytest.mark.parametrize("sw", [False, True])
def test_ckd_model_file_round_trip(tmp_path, sw):
    """CkdModel::write / ::read (ckd_model.cpp:290-641, :32-286): names, types and the model dict round trip."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    import ckd_synth
    model = ckd_synth.make_model_sw(seed=2) if sw else ckd_synth.make_model(seed=2)
    ng = len(model["iband_per_g"])
    nwav = 33
    rs = np.random.RandomState(1)
    gf = rs.uniform(size=(ng, nwav)); gf /= gf.sum(0, keepdims=True)
    model.update(wavenumber1=np.arange(nwav) * 10.0, wavenumber2=np.arange(1, nwav + 1) * 10.0, gpoint_fraction=gf,
                 wavenumber1_band=np.array([0.0, 100.0, 200.0]), wavenumber2_band=np.array([100.0, 200.0, 330.0]))
    p = str(tmp_path / "ckd.nc")
    ncio.write_ckd_model(p, model, model_id="synthetic", config="x 1\n", summary="test")
    r = netcdf_file(p, "r", mmap=False)
    assert r.constituent_id == b"composite h2o co2 ch4 o3" and r.model_id == b"synthetic"
    assert r.variables["h2o_conc_dependence_code"].getValue() == 2 and r.variables["ch4_conc_dependence_code"].getValue() == 3
    assert r.variables["h2o_molar_absorption_coeff"].dimensions == ("h2o_mole_fraction", "temperature", "pressure", "g_point")
    assert r.variables["composite_molar_absorption_coeff"].typecode() == "f" and r.variables["band_number"].typecode() == "h"
    assert b"2: Look-up table" in r.variables["co2_conc_dependence_code"].definition
    assert ("solar_irradiance" in r.variables) == sw and ("planck_function" in r.variables) == (not sw)
    r.close()
    back = ncio.read_ckd_model(p, active_gases=["h2o", "co2"])
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    assert np.array_equal(back["temperature"], f32(model["temperature"]))
    assert np.allclose(back["log_pressure"], model["log_pressure"], rtol=1e-7)
    assert np.array_equal(back["iband_per_g"], model["iband_per_g"]) and back["nband"] == 3
    for g0, g1 in zip(model["gases"], back["gases"]):
        assert g1["name"] == g0["name"] and g1["conc"] == g0["conc"] and g1["active"] == (g0["name"] in ("h2o", "co2"))
        assert np.array_equal(g1["molar_abs"], f32(g0["molar_abs"])) and np.array_equal(g1["max_molar_abs"], f32(g0["max_molar_abs"]))
        if g0["conc"] == "lut":
            assert np.array_equal(g1["vmr"], f32(g0["vmr"]))
        if g0["conc"] == "relative-linear":
            assert g1["reference_vmr"] == float(np.float32(g0["reference_vmr"]))
    if sw:
        assert np.array_equal(back["solar_irradiance"], f32(model["solar_irradiance"])) and back["planck_function"] is None
    else:
        assert np.array_equal(back["planck_function"], f32(model["planck_function"]))


def _write_lbl(path, sw, rs, ncol=4, nlev=5, nband=6):
    w = netcdf_file(str(path), "w", version=2)
    dims = dict(column=ncol, half_level=nlev + 1, level=nlev, gas=3, band=nband)
    if sw:
        dims["mu0"] = 5
    for d, n in dims.items():
        w.createDimension(d, n)
    a = {}
    a["pressure_hl"] = (("column", "half_level"), np.sort(rs.uniform(1, 1e5, (ncol, nlev + 1)), axis=1))
    a["temperature_hl"] = (("column", "half_level"), rs.uniform(200, 300, (ncol, nlev + 1)))
    a["mole_fraction_fl"] = (("column", "gas", "level"), rs.uniform(1e-6, 1e-2, (ncol, 3, nlev)))
    if sw:
        a["mu0"] = (("mu0",), np.array([0.1, 0.3, 0.5, 0.7, 0.9]))
        a["flux_dn_direct_sw"] = (("column", "mu0", "half_level"), rs.uniform(1, 1000, (ncol, 5, nlev + 1)))
        a["flux_up_sw"] = (("column", "mu0", "half_level"), rs.uniform(1, 100, (ncol, 5, nlev + 1)))
        a["band_flux_dn_direct_sw"] = (("column", "mu0", "half_level", "band"), rs.uniform(1, 100, (ncol, 5, nlev + 1, nband)))
        a["band_flux_up_sw"] = (("column", "mu0", "half_level", "band"), rs.uniform(1, 10, (ncol, 5, nlev + 1, nband)))
        a["band_wavenumber1_sw"] = (("band",), np.arange(nband) * 1000.0 + 250.0)
        a["band_wavenumber2_sw"] = (("band",), np.arange(1, nband + 1) * 1000.0 + 250.0)
    else:
        a["flux_dn_lw"] = (("column", "half_level"), rs.uniform(1, 400, (ncol, nlev + 1)))
        a["flux_up_lw"] = (("column", "half_level"), rs.uniform(1, 400, (ncol, nlev + 1)))
        a["band_flux_dn_lw"] = (("column", "half_level", "band"), rs.uniform(1, 40, (ncol, nlev + 1, nband)))
        a["band_flux_up_lw"] = (("column", "half_level", "band"), rs.uniform(1, 40, (ncol, nlev + 1, nband)))
        a["band_wavenumber1_lw"] = (("band",), np.arange(nband) * 500.0)
        a["band_wavenumber2_lw"] = (("band",), np.arange(1, nband + 1) * 500.0)
    for name, (d, v) in a.items():
        w.createVariable(name, "d", d)[:] = v
    w.constituent_id = "h2o-no-continuum o3 co2"
    w.close()
    return {k: v[1] for k, v in a.items()}


@pytest.mark.parametrize("sw", [False, True])
def test_read_lbl_fluxes(tmp_path, sw):
    """LblFluxes::read (lbl_fluxes.cpp:52-397): SZA replication, band mapping, effective albedo, gas mapping."""
    rs = np.random.RandomState(11)
    p = tmp_path / "lbl.nc"
    a = _write_lbl(p, sw, rs)
    bm = np.array([0, 0, 1, 1, 1, 2])
    s = ncio.read_lbl_fluxes(p, ["composite", "h2o", "co2", "ch4", "o3"], band_mapping=bm)
    assert s["is_sw"] == sw and s["molecules"] == ["h2o", "o3", "co2"]
    assert np.array_equal(s["gas_present"], [0, 1, 1, 0, 1])
    ncol = 4 * (3 if sw else 1)
    rep = (lambda x: np.repeat(x, 3, axis=0)) if sw else (lambda x: x)
    assert np.array_equal(s["pressure_hl"], rep(a["pressure_hl"])) and s["vmr_fl"].shape == (ncol, 5, 5)
    assert np.array_equal(s["vmr_fl"][:, 4], rep(a["mole_fraction_fl"])[:, 1]) and np.all(s["vmr_fl"][:, 3] == 0)
    agg = lambda x: np.stack([x[..., bm == j].sum(-1) for j in range(3)], axis=-1)
    if sw:
        sel = lambda x: x[:, [0, 2, 4]].reshape((ncol,) + x.shape[2:])
        assert np.array_equal(s["mu0"], np.tile([0.1, 0.5, 0.9], 4))
        assert s["tsi"] == a["flux_dn_direct_sw"][0, 0, 0] / 0.1
        dn, up = agg(sel(a["band_flux_dn_direct_sw"])), agg(sel(a["band_flux_up_sw"]))
        assert np.allclose(s["flux_dn"], dn, rtol=1e-15) and np.allclose(s["flux_up"], up, rtol=1e-15)
        assert np.allclose(s["albedo"], up[:, -1].sum(0) / dn[:, -1].sum(0), rtol=1e-15)
        assert np.array_equal(s["band_wavenumber2"], [2250.0, 5250.0, 6250.0])
        ncio.mask_rayleigh_up(s, 5000.0)
        assert np.all(s["albedo"][1:] == 0) and s["albedo"][0] > 0 and np.all(s["flux_up"][..., 1:] == 0)
        assert np.all(s["broadband_flux_up"] == 0)
    else:
        assert np.allclose(s["flux_dn"], agg(a["band_flux_dn_lw"]), rtol=1e-15)
        assert np.array_equal(s["broadband_flux_up"], a["flux_up_lw"])
        assert np.array_equal(s["band_wavenumber1"], [0.0, 1000.0, 2500.0])
    other = dict(s, flux_dn=s["flux_dn"] * 0.25, flux_up=s["flux_up"] * 0.5, broadband_flux_dn=s["broadband_flux_dn"] * 0,
                 broadband_flux_up=s["broadband_flux_up"] * 0)
    before = s["flux_dn"].copy()
    ncio.subtract_lbl_fluxes(s, other)
    assert np.allclose(s["flux_dn"], 0.75 * before)


# ---- NetCDF-4 (HDF5) read path: csrc/nc_hdf5.cpp over the system's HDF5 library ----

import sys  # noqa: E402


def _h5_spectrum(path, vlen=False):
    import h5_fixture as h5
    rs = np.random.RandomState(5)
    ncol, nlay, nwav = 3, 6, 1000
    od = (rs.uniform(0, 2, (ncol, nlay, nwav)) ** 4).astype(np.float32)
    od[:, :, ::7] = 0.0
    p = np.cumsum(rs.uniform(1.0, 100.0, (ncol, nlay + 1)), axis=1)
    t = rs.uniform(200.0, 300.0, (ncol, nlay + 1))
    wn = np.linspace(0.0, 3260.0, nwav)
    vmr = rs.uniform(1e-4, 1e-3, (ncol, nlay))
    h5.write(path, {
        "pressure_hl": (p, "f4", None, {"units": "Pa"}),
        "temperature_hl": (t, "f4", None, {"units": "K", "valid_range": [100.0, 400.0]}),
        "wavenumber": (wn, "f8", (250,), {"long_name": "Wavenumber"}),
        "mole_fraction_fl": (vmr, "f4", None, None),
        "reference_surface_mole_fraction": (np.float32(4e-4), "f4", None, None),
        "optical_depth": (od, "f4", (1, nlay, 256), {"long_name": "Layer optical depth"}),         # chunked + shuffle + deflate
        "band_index": (np.arange(nwav) % 3, "i2", None, None),
    }, {"constituent_id": ("co2",) if vlen else "co2", "title": "synthetic NetCDF-4 spectrum", "profile_scale": 2.5})
    return od, p.astype(np.float32), t.astype(np.float32), wn, vmr.astype(np.float32)


@pytest.mark.parametrize("vlen", [False, True])
def test_netcdf4_spectrum_is_read_through_hdf5(tmp_path, vlen):
    """What read_spectrum needs from a CKDMIP *.h5 file: shapes, one-column slices of the deflated optical depth, FLOAT ->
    double conversion, derived d_wavenumber, text attributes stored as fixed-length (NC_CHAR) or variable-length (NC_STRING)."""
    sys.path.insert(0, os.path.dirname(__file__))
    import h5_fixture as h5
    from ecckd_amd import ncio
    if not h5.available():
        pytest.skip("no HDF5 shared library with the deflate filter in this environment")
    path = tmp_path / "spectrum.h5"
    od, p, t, wn, vmr = _h5_spectrum(path, vlen)
    with ncio.NcFile(path) as f:
        assert f.exist("optical_depth") and not f.exist("d_wavenumber")
        name, shape = f.var_info("optical_depth")[0], f.var_info("optical_depth")[1]
        assert tuple(shape) == od.shape
        assert f.att_text("constituent_id") == "co2" and f.att_text("title") == "synthetic NetCDF-4 spectrum"
        assert f.att_text("units", "pressure_hl") == "Pa" and f.att_text("nothing") is None
        assert np.array_equal(f.read("optical_depth"), od.astype(np.float64))
        assert np.array_equal(f.read("optical_depth", 2), od[2].astype(np.float64))
        assert np.array_equal(f.read("band_index"), np.arange(1000) % 3)
        assert float(f.read("reference_surface_mole_fraction").reshape(-1)[0]) == np.float32(4e-4)
    for icol in (0, 2):
        s = ncio.read_spectrum(path, icol)
        assert s["ncol"] == 3 and s["molecule"] == "co2"
        assert np.array_equal(s["optical_depth"], od[icol].astype(np.float64))
        assert np.array_equal(s["pressure_hl"], p[icol].astype(np.float64)) and np.array_equal(s["temperature_hl"], t[icol].astype(np.float64))
        assert np.array_equal(s["wavenumber_cm_1"], wn) and np.array_equal(s["vmr_fl"], vmr[icol].astype(np.float64))
        dwn = s["d_wavenumber_cm_1"]
        assert np.allclose(dwn[1:-1], 0.5 * (wn[2:] - wn[:-2])) and dwn[0] == 0.5 * dwn[1]
    with pytest.raises(Exception):
        with ncio.NcFile(path) as f:
            f.read("no_such_variable")


def test_netcdf4_chunks_are_inflated_in_parallel(tmp_path, monkeypatch):
    """Chunked + shuffled + deflated FLOAT / DOUBLE variables are taken apart outside the HDF5 library (raw chunks pulled by the
    caller, inflated / unshuffled / placed by worker threads, csrc/nc_hdf5.cpp::h5_read_real_parallel): chunk shapes that do not
    divide the variable, chunks spanning several slices, DOUBLE storage, a 1-D variable - same values as the library's own
    H5Dread path (ECCKD_NO_PARALLEL_INFLATE)."""
    sys.path.insert(0, os.path.dirname(__file__))
    import h5_fixture as h5
    from ecckd_amd import ncio
    if not h5.available():
        pytest.skip("no HDF5 shared library with the deflate filter in this environment")
    rs = np.random.RandomState(0)
    od = (rs.standard_normal((3, 17, 50_003)) * 10).astype(np.float32)
    od[1, 3, :7] = [0.0, -0.0, np.float32(1e-42), np.inf, -np.inf, 1.0, -1.0]
    x64 = rs.standard_normal((4, 5, 3001))
    path = tmp_path / "chunky.h5"
    h5.write(path, {"optical_depth": (od, "f4", (2, 5, 4096), None), "x64": (x64, "f8", (1, 2, 777), None),
                    "flat": (od[0, 0], "f4", (1000,), None)})
    for no_parallel in (False, True):
        if no_parallel:
            monkeypatch.setenv("ECCKD_NO_PARALLEL_INFLATE", "1")
        with ncio.NcFile(path) as f:
            for k in range(3):
                assert np.array_equal(f.read("optical_depth", k).view(np.uint64), od[k].astype(np.float64).view(np.uint64)), k
            assert np.array_equal(f.read("optical_depth"), od.astype(np.float64))
            assert np.array_equal(f.read("x64", 3), x64[3]) and np.array_equal(f.read("x64"), x64)
            assert np.array_equal(f.read("flat"), od[0, 0].astype(np.float64))


def test_damaged_classic_files_are_refused_or_read_never_crashed(tmp_path):
    """The classic reader on files with flipped header bytes, cut-off tails and inflated counts: an EcckdError or a result,
    never a crash, a hang or a read outside the file (a child process walks 600 damaged copies; it must end normally)."""
    import subprocess, textwrap
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    good = tmp_path / "good.nc"
    w = netcdf_file(str(good), "w", version=2)
    w.createDimension("column", 2); w.createDimension("level", 5); w.createDimension("wavenumber", 301)
    w.createVariable("wavenumber", "d", ("wavenumber",))[:] = np.linspace(1.0, 3000.0, 301)
    w.createVariable("pressure_hl", "d", ("column", "level"))[:] = np.arange(10.0).reshape(2, 5)
    w.createVariable("optical_depth", "f", ("column", "level", "wavenumber"))[:] = np.random.RandomState(0).rand(2, 5, 301).astype("f4")
    w.constituent_id = "h2o"
    w.close()
    child = textwrap.dedent(f"""
        import numpy as np, sys
        sys.path.insert(0, {repr(ROOT)})
        from ecckd_amd import ncio, EcckdError
        raw = bytearray(open({repr(str(good))}, 'rb').read())
        header = 400                                 # dimensions, attributes and variable records live here
        rs = np.random.RandomState(1)
        ok = refused = 0
        for k in range(600):
            b = bytearray(raw)
            what = k % 4
            if what == 0:
                for _ in range(1 + k % 3): b[rs.randint(header)] = rs.randint(256)
            elif what == 1:
                del b[rs.randint(8, len(b)):]
            elif what == 2:
                p = rs.randint(header - 4); b[p:p + 4] = (0x7fffffff).to_bytes(4, 'big')      # a huge count somewhere
            else:
                p = rs.randint(header - 8); b[p:p + 8] = rs.bytes(8)
            path = {repr(str(tmp_path))} + '/bad.nc'
            open(path, 'wb').write(b)
            try:
                f = ncio.NcFile(path)
                for name in ('wavenumber', 'pressure_hl', 'optical_depth'):
                    try:
                        f.var_info(name); f.read(name); f.read(name, 1)
                    except (EcckdError, KeyError):            # KeyError: the name itself was hit
                        pass
                try:
                    f.att_text('constituent_id')
                except (EcckdError, KeyError):
                    pass
                f.close()
                ok += 1
            except EcckdError:
                refused += 1
        print(ok, refused)
    """)
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    ok, refused = map(int, r.stdout.split())
    assert ok + refused == 600 and refused > 50


def test_streaming_record_count_and_unwritten_tail(tmp_path):
    """(i) numrecs = 0xFFFFFFFF (a streaming producer's "indeterminate") is derived from the file size as the NetCDF
    library does; (ii) a file whose LAST fixed variable was never completely written (NC_NOFILL, or cut off) keeps its other
    variables readable - only the variable that extends past the end is refused, when it is asked for."""
    from ecckd_amd import ncio, EcckdError
    path = tmp_path / "rec.nc"
    w = netcdf_file(str(path), "w", version=1)
    w.createDimension("time", None); w.createDimension("x", 3)
    w.createVariable("fixed", "d", ("x",))[:] = [1.0, 2.0, 3.0]
    v = w.createVariable("series", "d", ("time", "x"))
    for k in range(4):
        v[k] = np.arange(3.0) + 10 * k
    w.close()
    raw = bytearray(path.read_bytes())
    assert int.from_bytes(raw[4:8], "big") == 4
    raw[4:8] = (0xFFFFFFFF).to_bytes(4, "big")
    path.write_bytes(raw)
    with ncio.NcFile(str(path)) as f:
        assert f.var_info("series")[1] == (4, 3)
        assert np.array_equal(f.read("series"), np.arange(3.0)[None, :] + 10 * np.arange(4.0)[:, None])
        assert np.array_equal(f.read("fixed"), [1.0, 2.0, 3.0])
    path2 = tmp_path / "tail.nc"
    w = netcdf_file(str(path2), "w", version=2)
    w.createDimension("x", 50)
    w.createVariable("a", "d", ("x",))[:] = np.arange(50.0)
    w.createVariable("b", "d", ("x",))[:] = np.arange(50.0) * 2
    w.close()
    raw = path2.read_bytes()
    path2.write_bytes(raw[:-100])                       # the tail of the last variable is missing
    with ncio.NcFile(str(path2)) as f:
        assert np.array_equal(f.read("a"), np.arange(50.0))
        with pytest.raises(EcckdError, match="extends past the end"):
            f.read("b")
        with pytest.raises(EcckdError, match="extends past the end"):
            f.var_info("b")


def test_long_variables_are_written_by_several_threads(tmp_path):
    """A 1-D variable of millions of values (the ordering / g-points files of a 7.2e6-point spectrum) is cut into ranges that
    are encoded and written by a thread each: every value must land where the single-threaded writer puts it - checked
    through an independent reader (scipy) for every external type of the ordering file, at a length that is not a multiple
    of the piece size."""
    from scipy.io import netcdf_file
    n = 5 * (1 << 20) + 12345
    rs = np.random.RandomState(3)
    wn = np.cumsum(rs.uniform(1e-4, 2e-4, n))
    iband = (np.arange(n) // (n // 3 + 1)).astype(np.int16)
    rank = rs.permutation(n).astype(np.int32)
    key = rs.normal(size=n)
    col = rs.lognormal(size=n)
    p = str(tmp_path / "order_big.nc")
    ncio.write_order(p, [0.0], [3260.0], wn, np.full(n, 1e-4), iband, rank, key, col, molecule="h2o")
    f = netcdf_file(p, "r", mmap=False)
    assert np.array_equal(f.variables["wavenumber"][:], wn)
    assert np.array_equal(f.variables["rank"][:], rank)
    assert np.array_equal(f.variables["band_number"][:], iband)
    assert np.array_equal(f.variables["sorting_variable"][:], key.astype(np.float32))
    assert np.array_equal(f.variables["column_optical_depth"][:], col.astype(np.float32))
    f.close()
    o = ncio.read_order(p)
    assert np.array_equal(o["rank"], rank) and np.array_equal(o["wavenumber"], wn)


def test_large_record_variable_slices_next_to_another_record_variable(tmp_path):
    """the threaded read of a long run took the offset of slice k of a RECORD variable as k * (values per
    slice) although the records of a file lie `recsize` apart (all record variables of one record side by side).  Two record
    variables, one of them past the 4 * 2^20 values per record at which the threaded path starts: every slice must come back
    as an independent reader (scipy) sees it, and the whole variable too."""
    nbig = 4 * (1 << 20) + 4097
    rs = np.random.RandomState(11)
    p = str(tmp_path / "two_records.nc")
    w = netcdf_file(p, "w", version=2)
    w.createDimension("column", None); w.createDimension("three", 3); w.createDimension("wavenumber", nbig)
    small = w.createVariable("small", "d", ("column", "three"))
    big = w.createVariable("big", "f", ("column", "wavenumber"))
    rows = rs.normal(size=(3, nbig)).astype(np.float32)
    for k in range(3):
        small[k] = np.arange(3.0) + 10 * k
        big[k] = rows[k]
    w.close()
    r = netcdf_file(p, "r", mmap=False)
    seen = np.array(r.variables["big"][:], dtype=np.float64)
    r.close()
    assert np.array_equal(seen, rows.astype(np.float64))
    with ncio.NcFile(p) as f:
        for k in range(3):
            assert np.array_equal(f.read("big", k), seen[k]), k
            assert np.array_equal(f.read("small", k), np.arange(3.0) + 10 * k)
        assert np.array_equal(f.read("big"), seen)


def test_netcdf4_output_by_file_name(tmp_path, monkeypatch):
    """OutputDataFile.cpp:84-157: a file named *.h5 / *.hdf is written as NetCDF-4 (HDF5), *.nc as classic; deflate_variable
    (:345-359) = shuffle + deflate level 2 in the former, nothing in the latter.  The NetCDF-4 layout (dimension scales,
    `_nc4_non_coord_` for a variable that shares its name with a dimension it does not run over, `_Netcdf4Dimid`) is checked
    through the HDF5 library itself and read back by this repository's reader; there is no NetCDF library here to read it with
    (DESIGN: unpinned)."""
    import ctypes as C
    import h5_fixture
    if not h5_fixture.available():
        pytest.skip("no HDF5 library")
    rs = np.random.RandomState(5)
    nwav, ng = 3 * (1 << 20) + 17, 7
    wn = np.cumsum(rs.uniform(1e-4, 2e-4, nwav))
    gp = rs.randint(0, ng, nwav).astype(np.int16)
    err = rs.uniform(0, 1, ng).astype(np.float32)

    def write(path):
        w = ncio.NcWriter(path)
        w.define_dimension("band", 2)
        w.define_dimension("g_point", ng)
        w.define_dimension("wavenumber", nwav)
        w.define_variable("n_gases", "int")
        w.define_variable("wavenumber1_band", "float", "band")
        w.define_variable("h2o_error", "float", "g_point")
        w.define_variable("wavenumber", "double", "wavenumber")          # a coordinate variable
        w.define_variable("g_point", "short", "wavenumber")              # shares its name with the dimension g_point
        w.deflate_variable("g_point")
        w.deflate_variable("wavenumber")
        w.write_attribute("units", "cm-1", "wavenumber")
        w.write_attribute("constituent_id", "h2o")
        w.write_attribute("scale", [1.5, -2.0])
        is4 = w.is_netcdf4
        w.end_define_mode()
        w.write("n_gases", [1]); w.write("wavenumber1_band", [0.0, 1300.0]); w.write("h2o_error", err)
        w.write("wavenumber", wn); w.write("g_point", gp)
        w.close()
        return is4

    p4, p3 = str(tmp_path / "gpoints.h5"), str(tmp_path / "gpoints.nc")
    assert write(p4) is True and write(p3) is False
    assert open(p4, "rb").read(8) == b"\x89HDF\r\n\x1a\n" and open(p3, "rb").read(3) == b"CDF"
    assert os.path.getsize(p4) < 0.8 * os.path.getsize(p3)               # the deflated variables
    for p in (p4, p3):
        with ncio.NcFile(p) as f:
            assert f.dim("g_point") == ng and f.dim("wavenumber") == nwav and f.dim("band") == 2
            assert np.array_equal(f.read("g_point"), gp) and np.array_equal(f.read("wavenumber"), wn)
            assert np.array_equal(f.read("h2o_error"), err.astype(np.float64)) and int(np.asarray(f.read("n_gases")).reshape(-1)[0]) == 1
            assert f.att_text("units", "wavenumber") == "cm-1" and f.att_text("constituent_id") == "h2o"
            assert np.array_equal(f.att_values("scale"), [1.5, -2.0])
            assert f.var_info("g_point")[1] == (nwav,) and not f.exist("band")        # a dimension without a variable is no variable
    # the HDF5 view of the NetCDF-4 file
    h = h5_fixture.lib()
    hid = C.c_int64
    for name, res, args in (("H5Fopen", hid, [C.c_char_p, C.c_uint, hid]), ("H5Lexists", C.c_int, [hid, C.c_char_p, hid]),
                            ("H5Dopen2", hid, [hid, C.c_char_p, hid]), ("H5Dget_create_plist", hid, [hid]),
                            ("H5Pget_layout", C.c_int, [hid]), ("H5Pget_nfilters", C.c_int, [hid]),
                            ("H5Aexists_by_name", C.c_int, [hid, C.c_char_p, C.c_char_p, hid])):
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args
    f = h.H5Fopen(p4.encode(), 0, 0)
    assert f >= 0
    assert h.H5Lexists(f, b"_nc4_non_coord_g_point", 0) > 0 and h.H5Lexists(f, b"g_point", 0) > 0 and h.H5Lexists(f, b"band", 0) > 0
    for ds, chunked, nfilters in ((b"_nc4_non_coord_g_point", True, 2), (b"wavenumber", True, 2), (b"h2o_error", False, 0)):
        d = h.H5Dopen2(f, ds, 0)
        pl = h.H5Dget_create_plist(d)
        assert (h.H5Pget_layout(pl) == 2) == chunked and h.H5Pget_nfilters(pl) == nfilters, ds     # H5D_CHUNKED = 2
        h.H5Pclose(pl); h.H5Dclose(d)
    for ds in (b"g_point", b"band", b"wavenumber"):                       # every dimension is a dimension scale with its id
        assert h.H5Aexists_by_name(f, ds, b"CLASS", 0) > 0 and h.H5Aexists_by_name(f, ds, b"_Netcdf4Dimid", 0) > 0, ds
    for ds in (b"_nc4_non_coord_g_point", b"h2o_error", b"wavenumber1_band"):     # every variable is attached to its dimensions' scales
        assert h.H5Aexists_by_name(f, ds, b"DIMENSION_LIST", 0) > 0, ds
    assert h.H5Aexists_by_name(f, b"wavenumber", b"DIMENSION_LIST", 0) <= 0      # a coordinate variable is not attached to itself
    h.H5Fclose(f)
    # ECCKD_CLASSIC_OUTPUT: classic under the name asked for
    monkeypatch.setenv("ECCKD_CLASSIC_OUTPUT", "1")
    p5 = str(tmp_path / "classic.h5")
    assert write(p5) is False and open(p5, "rb").read(3) == b"CDF"


def test_netcdf4_deflated_variables_by_threads_and_by_the_library(tmp_path, monkeypatch):
    """A deflated variable's chunks are built by worker threads and handed to HDF5 as finished chunks (H5Dwrite_chunk);
    ECCKD_H5_SERIAL_WRITE=1 sends the same values through the library's own filter pipeline.  Both files read back to the
    values written - whole variables, slices of a 3-D variable whose rows span several chunks with a ragged last one, every
    file type - and have the same size to within the chunk index."""
    import h5_fixture
    if not h5_fixture.available():
        pytest.skip("no HDF5 library")
    rs = np.random.RandomState(11)
    n = (1 << 18) * 2 + 4321
    cube = rs.standard_normal((2, 3, n))
    as_int = rs.randint(-2**31, 2**31 - 1, n).astype(np.float64)
    as_short = rs.randint(-2**15, 2**15 - 1, n).astype(np.float64)
    as_float = rs.standard_normal(n)

    def write(path):
        w = ncio.NcWriter(path)
        w.define_dimension("column", 2)
        w.define_dimension("level", 3)
        w.define_dimension("wavenumber", n)
        w.define_variable("optical_depth", "double", "column", "level", "wavenumber")
        w.define_variable("rank", "int", "wavenumber")
        w.define_variable("band_number", "short", "wavenumber")
        w.define_variable("sorting_variable", "float", "wavenumber")
        for v in ("optical_depth", "rank", "band_number", "sorting_variable"):
            w.deflate_variable(v)
        assert w.is_netcdf4
        w.end_define_mode()
        w.write_slice("optical_depth", 1, cube[1])
        w.write_slice("optical_depth", 0, cube[0])
        w.write("rank", as_int); w.write("band_number", as_short); w.write("sorting_variable", as_float)
        w.close()

    direct, serial = str(tmp_path / "direct.h5"), str(tmp_path / "serial.h5")
    write(direct)
    monkeypatch.setenv("ECCKD_H5_SERIAL_WRITE", "1")
    write(serial)
    monkeypatch.delenv("ECCKD_H5_SERIAL_WRITE")
    for p in (direct, serial):
        with ncio.NcFile(p) as f:
            assert np.array_equal(f.read("optical_depth").reshape(cube.shape), cube)
            assert np.array_equal(f.read("optical_depth", 1).reshape(3, n), cube[1])
            assert np.array_equal(f.read("rank"), as_int) and np.array_equal(f.read("band_number"), as_short)
            assert np.array_equal(f.read("sorting_variable"), as_float.astype(np.float32).astype(np.float64))
    assert abs(os.path.getsize(direct) - os.path.getsize(serial)) < 4096
