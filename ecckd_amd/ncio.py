"""NetCDF classic file access through the C ABI (ecckd_nc_*), and the host-side mirrors of the reference's
file routines on top of it: read_spectrum (read_spectrum.cpp:20-87), write_order (write_order.cpp:24-143) and
the reading of a reordering file as find_g_points does it (find_g_points.cpp:555-565).

The reference picks the on-disk format by file extension (src/tools/DataFile.cpp:88-96): *.nc / *.cdf are
NetCDF classic, which is what this module reads and writes; *.h5 (NetCDF-4 / HDF5) needs a library that is
not in the image and is rejected with PARAMETER_ERROR."""
import ctypes as C

import numpy as np

from . import _lib
from .api import check

NC_TYPES = {"byte": 1, "char": 2, "short": 3, "int": 4, "float": 5, "double": 6,
            "ubyte": 7, "ushort": 8, "uint": 9, "int64": 10, "uint64": 11}


def _b(s):
    return None if s is None else s.encode()


class NcFile:
    """Read access to one NetCDF classic file."""

    def __init__(self, path):
        self.lib = _lib.load_library()
        h = C.c_void_p()
        check(self.lib.ecckd_nc_open(_b(str(path)), C.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ecckd_nc_close(self.handle)
            self.handle = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def dim(self, name):
        n = C.c_size_t()
        check(self.lib.ecckd_nc_inq_dim(self.handle, _b(name), C.byref(n)))
        return n.value

    def var_info(self, name):
        """-> (nc_type, shape) or None if the variable does not exist (DataFile::exist)."""
        ex, t, nd = C.c_int(), C.c_int(), C.c_int()
        shape = (C.c_size_t * 16)()
        check(self.lib.ecckd_nc_inq_var(self.handle, _b(name), C.byref(ex), C.byref(t), C.byref(nd), shape, 16))
        if not ex.value:
            return None
        return t.value, tuple(shape[k] for k in range(nd.value))

    def exist(self, name):
        return self.var_info(name) is not None

    def read_dev(self, ctx, name, index=None, dtype=None):
        """The variable (or index `index` of its slowest dimension) as a tensor in device memory (ecckd_nc_read_dev: the file's
        bytes go through pinned buffers and are decoded on the device).  dtype: torch.float32 / float64; default: FLOAT stays
        float32, everything else float64."""
        import torch
        info = self.var_info(name)
        if info is None:
            raise _lib.EcckdError(_lib.PARAMETER_ERROR, f'variable "{name}" not found')
        t, shape = info
        shape = shape[1:] if index is not None else shape
        if dtype is None:
            dtype = torch.float32 if t == 5 else torch.float64
        out = torch.empty(shape, dtype=dtype, device=ctx.device)
        ctx.fence_from_torch()
        check(self.lib.ecckd_nc_read_dev(ctx.handle, self.handle, _b(name), -1 if index is None else int(index),
                                         4 if dtype == torch.float32 else 8, C.c_void_p(out.data_ptr()), out.numel()))
        return out

    def read(self, name, index=None):
        """The variable as float64 (every external type is converted, DataFileEngineNetcdf.cpp:593-599);
        `index` selects one entry of the slowest dimension (:582-590)."""
        info = self.var_info(name)
        if info is None:
            raise KeyError(name)
        shape = info[1] if index is None else info[1][1:]
        out = np.empty(shape, dtype=np.float64)
        check(self.lib.ecckd_nc_read_double(self.handle, _b(name), -1 if index is None else int(index),
                                            out.ctypes.data_as(C.POINTER(C.c_double)), out.size))
        return out

    def att_text(self, name, var=None):
        ex = C.c_int()
        buf = C.create_string_buffer(1 << 16)
        check(self.lib.ecckd_nc_read_att_text(self.handle, _b(var), _b(name), C.byref(ex), buf, len(buf)))
        return buf.value.decode() if ex.value else None

    def att_values(self, name, var=None):
        n = C.c_int()
        out = np.empty(4096)
        check(self.lib.ecckd_nc_read_att_double(self.handle, _b(var), _b(name), C.byref(n),
                                                out.ctypes.data_as(C.POINTER(C.c_double)), out.size))
        return None if n.value < 0 else out[:n.value].copy()


class NcWriter:
    """Define-then-write access (OutputDataFile): dimensions, variables, attributes, end_define, data."""

    def __init__(self, path):
        self.lib = _lib.load_library()
        h = C.c_void_p()
        check(self.lib.ecckd_nc_create(_b(str(path)), C.byref(h)))
        self.handle = h
        self.dims = {}

    def define_dimension(self, name, length):
        i = C.c_int()
        check(self.lib.ecckd_nc_def_dim(self.handle, _b(name), int(length), C.byref(i)))
        self.dims[name] = i.value

    def define_variable(self, name, nc_type, *dims):
        ids = (C.c_int * max(len(dims), 1))(*[self.dims[d] for d in dims])
        check(self.lib.ecckd_nc_def_var(self.handle, _b(name), NC_TYPES[nc_type], len(dims), ids, None))

    def write_attribute(self, name, value, var=None):
        if isinstance(value, str):
            check(self.lib.ecckd_nc_put_att_text(self.handle, _b(var), _b(name), _b(value)))
        else:
            v = np.atleast_1d(np.asarray(value, dtype=np.float64))
            check(self.lib.ecckd_nc_put_att_double(self.handle, _b(var), _b(name), NC_TYPES["double"], v.size,
                                                   v.ctypes.data_as(C.POINTER(C.c_double))))

    def deflate_variable(self, name):
        """OutputDataFile::deflate_variable (:345-359): shuffle + deflate level 2 where the file is written as NetCDF-4 (a name
        ending in .h5 / .hdf), nothing in a classic file."""
        check(self.lib.ecckd_nc_deflate_var(self.handle, _b(name)))

    @property
    def is_netcdf4(self):
        v = C.c_int()
        check(self.lib.ecckd_nc_is_netcdf4(self.handle, C.byref(v)))
        return bool(v.value)

    def end_define_mode(self):
        check(self.lib.ecckd_nc_enddef(self.handle))

    def write(self, name, data):
        d = np.ascontiguousarray(data, dtype=np.float64)
        check(self.lib.ecckd_nc_write_double(self.handle, _b(name), d.ctypes.data_as(C.POINTER(C.c_double)), d.size))

    def write_slice(self, name, index, data):
        """One index of the variable's slowest dimension (ecckd_nc_write_slice_double)."""
        d = np.ascontiguousarray(data, dtype=np.float64)
        check(self.lib.ecckd_nc_write_slice_double(self.handle, _b(name), int(index), d.ctypes.data_as(C.POINTER(C.c_double)), d.size))

    def close(self):
        if getattr(self, "handle", None):
            h, self.handle = self.handle, None
            check(self.lib.ecckd_nc_close(h))


def read_spectrum(path, iprofile=0, optical_depth=True):
    """read_spectrum.cpp:20-87 for one column of a CKDMIP spectral file -> dict with the reference's names.
    optical_depth=False: the grids and profiles only (a process that needs the first gas's Planck matrix, not its spectrum)."""
    with NcFile(path) as f:
        out = {"ncol": f.var_info("pressure_hl")[1][0]}
        out["pressure_hl"] = f.read("pressure_hl", iprofile)
        out["temperature_hl"] = f.read("temperature_hl", iprofile) if f.exist("temperature_hl") else None   # :46-52
        wn = f.read("wavenumber")
        out["wavenumber_cm_1"] = wn
        if f.exist("d_wavenumber"):
            out["d_wavenumber_cm_1"] = f.read("d_wavenumber")
        else:                                                   # :58-65 (ecckd_derive_d_wavenumber_dev on the device)
            dwn = np.empty_like(wn)
            dwn[1:-1] = 0.5 * (wn[2:] - wn[:-2])
            dwn[0] = 0.5 * dwn[1]
            dwn[-1] = 0.5 * dwn[-2]
            out["d_wavenumber_cm_1"] = dwn
        out["molecule"] = f.att_text("constituent_id")
        out["reference_surface_vmr"] = (float(f.read("reference_surface_mole_fraction").reshape(-1)[0])
                                        if f.exist("reference_surface_mole_fraction") else -1.0)     # :68-73
        info = f.var_info("mole_fraction_fl")
        if info is not None and len(info[1]) == 2:                                                  # :76-82
            out["vmr_fl"] = f.read("mole_fraction_fl", iprofile)
        else:
            out["vmr_fl"] = np.full(out["pressure_hl"].size - 1, -1.0)
        if optical_depth:
            out["optical_depth"] = f.read("optical_depth", iprofile)
    return out


def write_order(path, band_bound1, band_bound2, wavenumber, d_wavenumber, iband, rank, sorting_variable,
                column_optical_depth=None, molecule="", config_str="", history=None):
    """write_order.cpp:24-143."""
    lib = _lib.load_library()
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    b1, b2, wn, dwn, sv = f64(band_bound1), f64(band_bound2), f64(wavenumber), f64(d_wavenumber), f64(sorting_variable)
    ib = np.ascontiguousarray(iband, dtype=np.int16)
    rk = np.ascontiguousarray(rank, dtype=np.int32)
    col = f64(column_optical_depth) if column_optical_depth is not None else None
    check(lib.ecckd_write_order_file(_b(str(path)), _b(molecule), _b(config_str), _b(history), b1.size, P(b1), P(b2), wn.size,
                                     P(wn), P(dwn), ib.ctypes.data_as(C.POINTER(C.c_int16)),
                                     rk.ctypes.data_as(C.POINTER(C.c_int32)), P(col) if col is not None else None, P(sv)))


def read_order(path):
    """The variables find_g_points reads from a reordering file (find_g_points.cpp:555-565, :676-681)."""
    with NcFile(path) as f:
        return dict(rank=f.read("rank").astype(np.int32), band_number=f.read("band_number").astype(np.int16),
                    sorting_variable=f.read("sorting_variable"), wavenumber=f.read("wavenumber"),
                    wavenumber1_band=f.read("wavenumber1_band"), wavenumber2_band=f.read("wavenumber2_band"),
                    molecule=f.att_text("molecule"))


# ---- CKD definition files (CkdModel::read ckd_model.cpp:32-286, CkdModel::write :290-641) ----------------------

K_NAME = "molar_absorption_coeff"          # constants.h:21
CONC_CODE = {"none": 0, "linear": 1, "lut": 2, "relative-linear": 3}
CONC_NAME = {v: k for k, v in CONC_CODE.items()}


def write_ckd_model(path, model, model_id="", history="", config="", summary=""):
    """CkdModel::write: the CKD definition file create_look_up_table and optimize_lut produce (classic format).

    model: the dict api.Optimizer / api.run_ckd take (gases need a "name"), plus wavenumber1, wavenumber2,
    gpoint_fraction[ng, nwav], wavenumber1_band, wavenumber2_band."""
    w = NcWriter(path)
    temp = np.asarray(model["temperature"], dtype=np.float64)
    nt, npres = temp.shape
    gf = np.asarray(model["gpoint_fraction"], dtype=np.float64)
    ng, nwav = gf.shape
    is_sw = model.get("solar_irradiance") is not None
    w.define_dimension("temperature", nt)
    w.define_dimension("pressure", npres)
    w.define_dimension("g_point", ng)
    if not is_sw:
        w.define_dimension("temperature_planck", len(model["temperature_planck"]))
    w.define_dimension("wavenumber", nwav)
    w.define_dimension("band", len(model["wavenumber1_band"]))
    save_gp = bool(model.get("save_g_points")) and model.get("g_point_hr") is not None       # CkdModel::save_g_points, ckd_model.h:314-318
    if save_gp:
        w.define_dimension("wavenumber_hr", len(model["wavenumber_hr"]))

    def var(name, t, dims, long_name, units=None, comment=None):
        w.define_variable(name, t, *dims)
        w.write_attribute("long_name", long_name, var=name)
        if units:
            w.write_attribute("units", units, var=name)
        if comment:
            w.write_attribute("comment", comment, var=name)

    var("n_gases", "int", (), "Number of gases treated", comment='The gases are listed in the global attribute "constituent_id".')
    var("temperature", "float", ("temperature", "pressure"), "Temperature", "K")
    var("pressure", "float", ("pressure",), "Pressure", "Pa")
    tsi = model.get("reference_total_solar_irradiance")
    if is_sw:
        if tsi is not None and tsi > 0.0:                                                       # ckd_model.cpp:331-335
            var("reference_total_solar_irradiance", "float", (), "Reference total solar irradiance", "W m-2")
        var("solar_irradiance", "float", ("g_point",), "Solar irradiance across each g point", "W m-2")
        if model.get("solar_spectral_irradiance") is not None:                                  # :341-345
            var("solar_spectral_irradiance", "float", ("wavenumber",), "Solar irradiance in each spectral interval", "W m-2")
    else:
        var("temperature_planck", "float", ("temperature_planck",), "Temperature for Planck function look-up table", "K")
        var("planck_function", "float", ("temperature_planck", "g_point"), "Planck function look-up table", "W m-2")
    var("wavenumber1", "float", ("wavenumber",), "Lower wavenumber bound of spectral interval", "cm-1")
    var("wavenumber2", "float", ("wavenumber",), "Upper wavenumber bound of spectral interval", "cm-1")
    var("gpoint_fraction", "float", ("g_point", "wavenumber"), "Fraction of spectrum contributing to each g-point")
    var("wavenumber1_band", "float", ("band",), "Lower wavenumber bound of band", "cm-1")
    var("wavenumber2_band", "float", ("band",), "Upper wavenumber bound of band", "cm-1")
    var("band_number", "short", ("g_point",), "Band number of each g point")
    if save_gp:
        var("wavenumber_hr", "double", ("wavenumber_hr",), "High-resolution wavenumber", "cm-1")
        var("g_point", "short", ("wavenumber_hr",), "G point")
    if is_sw and model.get("rayleigh_molar_scattering") is not None:
        var("rayleigh_molar_scattering_coeff", "float", ("g_point",), "Rayleigh molar scattering coefficient in each g-point",
            "m2 mol-1")
    if model_id:
        w.write_attribute("model_id", model_id)
    names = [g["name"] for g in model["gases"]]
    w.write_attribute("constituent_id", " ".join(names))
    for g in model["gases"]:
        mol, Mol = g["name"], g["name"].upper()
        code = mol + "_conc_dependence_code"
        var(code, "short", (), Mol + " concentration dependence code")
        w.write_attribute("definition", "0: No dependence of absorption on concentration (background gases)\n"
                          "1: Absorption varies linearly with concentration\n"
                          "2: Look-up table for concentration-dependence of absorption\n"
                          "3: Linear dependence on concentration minus a reference value", var=code)
        k = mol + "_" + K_NAME
        dims = ("temperature", "pressure", "g_point")
        if g["conc"] == "lut":
            w.define_dimension(mol + "_mole_fraction", len(g["vmr"]))
            var(mol + "_mole_fraction", "float", (mol + "_mole_fraction",), Mol + " mole fraction for look-up table", "1")
            dims = (mol + "_mole_fraction",) + dims
        if g["conc"] == "relative-linear":
            var(mol + "_reference_mole_fraction", "float", (), "Reference mole fraction of " + Mol, "1")
        what = "background gases" if g["conc"] == "none" else Mol
        var(k, "float", dims, "Molar absorption coefficient of " + what, "m2 mol-1")
        if g.get("min_molar_abs") is not None and g.get("max_molar_abs") is not None:
            var(k + "_min", "float", dims, "Minimum molar absorption coefficient of " + what, "m2 mol-1")
            var(k + "_max", "float", dims, "Maximum molar absorption coefficient of " + what, "m2 mol-1")
        if g["conc"] == "none" and g.get("composite_vmr") is not None:                         # ckd_model.cpp:431-438
            cv = np.atleast_2d(np.asarray(g["composite_vmr"], dtype=np.float64))
            w.define_dimension(mol + "_gas", cv.shape[0])
            var(mol + "_mole_fraction", "float", (mol + "_gas", "pressure"), "Mole fractions of the gases that make up " + Mol, "1",
                comment='The gases that make up ' + Mol + ' are listed in the global attribute "' + mol + '_constituent_id".')
            w.write_attribute(mol + "_constituent_id", g.get("composite_molecules", ""))
    if history:
        w.write_attribute("history", history)
    w.write_attribute("config", config)
    w.write_attribute("summary", summary)
    w.end_define_mode()
    w.write("n_gases", [len(names)])
    w.write("pressure", np.exp(np.asarray(model["log_pressure"], dtype=np.float64)))
    w.write("temperature", temp)
    if is_sw:
        if tsi is not None and tsi > 0.0:
            w.write("reference_total_solar_irradiance", [tsi])
        w.write("solar_irradiance", model["solar_irradiance"])
        if model.get("solar_spectral_irradiance") is not None:
            w.write("solar_spectral_irradiance", model["solar_spectral_irradiance"])
        if model.get("rayleigh_molar_scattering") is not None:
            w.write("rayleigh_molar_scattering_coeff", model["rayleigh_molar_scattering"])
    else:
        w.write("temperature_planck", model["temperature_planck"])
        w.write("planck_function", model["planck_function"])
    w.write("wavenumber1", model["wavenumber1"])
    w.write("wavenumber2", model["wavenumber2"])
    w.write("gpoint_fraction", gf)
    w.write("wavenumber1_band", model["wavenumber1_band"])
    w.write("wavenumber2_band", model["wavenumber2_band"])
    w.write("band_number", model["iband_per_g"])
    if save_gp:
        w.write("wavenumber_hr", model["wavenumber_hr"])
        w.write("g_point", model["g_point_hr"])
    for g in model["gases"]:
        mol = g["name"]
        w.write(mol + "_conc_dependence_code", [CONC_CODE[g["conc"]]])
        if g["conc"] == "lut":
            w.write(mol + "_mole_fraction", g["vmr"])
        if g["conc"] == "relative-linear":
            w.write(mol + "_reference_mole_fraction", [g["reference_vmr"]])
        if g["conc"] == "none" and g.get("composite_vmr") is not None:
            w.write(mol + "_mole_fraction", np.atleast_2d(g["composite_vmr"]))
        w.write(mol + "_" + K_NAME, g["molar_abs"])
        if g.get("min_molar_abs") is not None and g.get("max_molar_abs") is not None:
            w.write(mol + "_" + K_NAME + "_min", g["min_molar_abs"])
            w.write(mol + "_" + K_NAME + "_max", g["max_molar_abs"])
    w.close()


def read_ckd_model(path, active_gases=None):
    """CkdModel::read -> the model dict of api.Optimizer / api.run_ckd (gases active as in `active_gases`, all if None)."""
    with NcFile(path) as f:
        m = {}
        if f.exist("solar_irradiance"):
            m["solar_irradiance"] = f.read("solar_irradiance")
            m["planck_function"] = m["temperature_planck"] = None
            if f.exist("rayleigh_molar_scattering_coeff"):
                m["rayleigh_molar_scattering"] = f.read("rayleigh_molar_scattering_coeff")
            if f.exist("solar_spectral_irradiance"):
                m["solar_spectral_irradiance"] = f.read("solar_spectral_irradiance")
            if f.exist("reference_total_solar_irradiance"):
                m["reference_total_solar_irradiance"] = float(f.read("reference_total_solar_irradiance").reshape(-1)[0])
        else:
            m["temperature_planck"] = f.read("temperature_planck")
            m["planck_function"] = f.read("planck_function")
        m["temperature"] = f.read("temperature")
        m["log_pressure"] = np.log(f.read("pressure"))
        for k in ("wavenumber1", "wavenumber2", "gpoint_fraction", "wavenumber1_band", "wavenumber2_band"):
            m[k] = f.read(k)
        m["iband_per_g"] = f.read("band_number").astype(np.int32)
        if f.exist("g_point"):                                                            # ckd_model.cpp:72-75; not written back (:471)
            m["wavenumber_hr"], m["g_point_hr"] = f.read("wavenumber_hr"), f.read("g_point").astype(np.int32)
        m["nband"] = m["wavenumber1_band"].size
        m["ng"] = m["gpoint_fraction"].shape[0]
        m["model_id"] = f.att_text("model_id")
        names = f.att_text("constituent_id").split(" ")
        assert int(f.read("n_gases")) == len(names)
        gases = []
        for mol in names:
            code = int(f.read(mol + "_conc_dependence_code"))
            g = dict(name=mol, conc=CONC_NAME[code], molar_abs=f.read(mol + "_" + K_NAME),
                     active=active_gases is None or mol in active_gases)
            if code == 2:
                g["vmr"] = f.read(mol + "_mole_fraction")
            if code == 3:
                g["reference_vmr"] = float(f.read(mol + "_reference_mole_fraction").reshape(-1)[0])
            if code == 0 and f.exist(mol + "_mole_fraction"):                             # :193-195
                g["composite_vmr"] = f.read(mol + "_mole_fraction")
                g["composite_molecules"] = f.att_text(mol + "_constituent_id") or ""
            if f.exist(mol + "_" + K_NAME + "_min"):
                g["min_molar_abs"] = f.read(mol + "_" + K_NAME + "_min")
                g["max_molar_abs"] = f.read(mol + "_" + K_NAME + "_max")
            gases.append(g)
        m["gases"] = gases
    return m


# ---- LBL training fluxes (LblFluxes::read lbl_fluxes.cpp:52-397, make_gas_mapping, mask_rayleigh_up, subtract) -----

def read_lbl_fluxes(path, model_molecules, band_mapping=None, gmap=None, ctx=None):
    """LblFluxes::read + make_gas_mapping -> a training-scene dict for api.Optimizer.

    model_molecules: the CKD model's gas names in model order (CkdModel::molecules).
    band_mapping:    optional narrow-band -> wide-band index per file band (:150-176, :272-296).
    gmap, ctx:       an api.GPointMap of the g-points file and its context; with them the high-resolution boundary
                     fluxes are summed per g point on the device (:180-246, :300-325) and the shortwave erythemal
                     weights are formed (:198-230)."""
    bm = None if band_mapping is None else np.asarray(band_mapping, dtype=np.int64)
    with NcFile(path) as f:
        p = f.read("pressure_hl")
        t = f.read("temperature_hl")
        vmr_file = f.read("mole_fraction_fl")                       # (column, gas, level)
        ncol = p.shape[0]
        is_sw = f.exist("mu0")
        out = {"is_sw": is_sw}
        dom = "sw" if is_sw else "lw"

        def map_bands(a, wn1, wn2):                                  # sum narrow bands into wide ones
            nb = int(bm.max()) + 1
            new = np.stack([a[..., bm == j].sum(-1) for j in range(nb)], axis=-1)
            return new, np.array([wn1[bm == j].min() for j in range(nb)]), np.array([wn2[bm == j].max() for j in range(nb)])

        if is_sw:
            mu0_all = f.read("mu0")
            index_sza = [0, 2, 4]                                    # :82
            nsza = len(index_sza)
            rep = lambda a: np.repeat(a, nsza, axis=0)               # repeat_matrix / repeat_array3D
            p, t, vmr_file = rep(p), rep(t), rep(vmr_file)
            pick = lambda a: np.ascontiguousarray(a[:, index_sza].reshape((ncol * nsza,) + a.shape[2:]))
            flux_dn, flux_up = pick(f.read("flux_dn_direct_sw")), pick(f.read("flux_up_sw"))
            out["mu0"] = np.tile(mu0_all[index_sza], ncol)
            out["tsi"] = float(flux_dn[0, 0] / out["mu0"][0])        # :116
            have_band = False
            if f.exist("spectral_flux_dn_direct_sw"):
                sdn, sup = pick(f.read("spectral_flux_dn_direct_sw")), pick(f.read("spectral_flux_up_sw"))
            elif f.exist("band_flux_dn_direct_sw"):
                sdn, sup = pick(f.read("band_flux_dn_direct_sw")), pick(f.read("band_flux_up_sw"))
                have_band = True
                wn1, wn2 = f.read("band_wavenumber1_sw"), f.read("band_wavenumber2_sw")
            else:
                sdn = sup = None
            if sdn is not None:
                if have_band and bm is not None:
                    sdn, wn1n, wn2n = map_bands(sdn, wn1, wn2)
                    sup, _, _ = map_bands(sup, wn1, wn2)
                    wn1, wn2 = wn1n, wn2n
                out["albedo"] = sup[:, -1, :].sum(0) / sdn[:, -1, :].sum(0)      # effective_spectral_albedo_ (:147-148, :166-167)
            ncol *= nsza
            hi_dn, hi_up = "spectral_flux_dn_direct_surf_sw", "spectral_flux_up_toa_sw"
        else:
            flux_dn, flux_up = f.read("flux_dn_lw"), f.read("flux_up_lw")
            have_band = False
            if f.exist("spectral_flux_up_lw"):
                sup, sdn = f.read("spectral_flux_up_lw"), f.read("spectral_flux_dn_lw")
            elif f.exist("band_flux_up_lw"):
                sup, sdn = f.read("band_flux_up_lw"), f.read("band_flux_dn_lw")
                wn1, wn2 = f.read("band_wavenumber1_lw"), f.read("band_wavenumber2_lw")
                have_band = True
                if bm is not None:
                    sup, wn1n, wn2n = map_bands(sup, wn1, wn2)
                    sdn, _, _ = map_bands(sdn, wn1, wn2)
                    wn1, wn2 = wn1n, wn2n
            else:
                sup = sdn = None
            hi_dn, hi_up = "spectral_flux_dn_surf_lw", "spectral_flux_up_toa_lw"
        out.update(pressure_hl=p, temperature_hl=t, broadband_flux_dn=flux_dn, broadband_flux_up=flux_up,
                   flux_dn=sdn, flux_up=sup, have_band_fluxes=have_band)
        if have_band:
            out["band_wavenumber1"], out["band_wavenumber2"] = wn1, wn2

        # high-resolution boundary fluxes summed per g point (device: ecckd_gmap_sum_rows)
        if f.exist(hi_dn) and f.exist(hi_up) and gmap is not None and ctx is not None:
            import torch
            dn_rows, up_rows = [], []
            nfile_col = f.var_info(hi_dn)[1][0]
            for icol in range(nfile_col):
                d, u = f.read(hi_dn, icol), f.read(hi_up, icol)      # LW: (wavenumber,), SW: (sza, wavenumber)
                if is_sw:
                    d, u = d[index_sza], u[index_sza]
                dn_rows.append(np.atleast_2d(d)); up_rows.append(np.atleast_2d(u))
            rows = torch.as_tensor(np.concatenate(dn_rows + up_rows), device=ctx.device)
            sums = gmap.sum_rows(rows)
            out["spectral_flux_dn_surf"], out["spectral_flux_up_toa"] = sums[:ncol], sums[ncol:]
            if is_sw:
                out["erythemal_spectrum"] = gmap.erythemal_spectrum()

        # gases: the file's constituent list mapped onto the model's (make_gas_mapping); "h2o-no-continuum" -> "h2o"
        file_gases = [m.split("-")[0] for m in f.att_text("constituent_id").split(" ")]
        out["molecules"] = file_gases
        nlay = p.shape[1] - 1
        vmr = np.zeros((ncol, len(model_molecules), nlay))
        present = np.zeros(len(model_molecules), dtype=np.int32)
        for i, mol in enumerate(model_molecules):
            if mol in file_gases:
                vmr[:, i, :] = vmr_file[:, file_gases.index(mol), :]
                present[i] = 1
        out["vmr_fl"], out["gas_present"] = vmr, present
    return out


def mask_rayleigh_up(scene, max_no_rayleigh_wavenumber):
    """LblFluxes::mask_rayleigh_up (lbl_fluxes.cpp:415-429): bands above the limit lose their upwelling."""
    idx = scene["band_wavenumber2"] > max_no_rayleigh_wavenumber
    if idx.any():
        scene["albedo"] = np.where(idx, 0.0, scene["albedo"])
        scene["flux_up"] = np.where(idx[None, None, :], 0.0, scene["flux_up"])
        scene["broadband_flux_up"] = np.zeros_like(scene["broadband_flux_up"])
    return scene


def subtract_lbl_fluxes(scene, source):
    """LblFluxes::subtract (:431-440): train on the difference to a relative-to scene."""
    for k in ("flux_dn", "flux_up", "broadband_flux_dn", "broadband_flux_up"):
        scene[k] = scene[k] - source[k]
    return scene


# ---- g-points file (find_g_points.cpp:1487-1660) ------------------------------------------------------------

def write_g_points(path, band_bound1, band_bound2, band_number, gases, wavenumber, g_point, solar_irradiance=None,
                   config_str="", history=None):
    """gases: list of dict(name, n_g_points[nband], band_number, rank1, rank2, error, sorting_variable, g_min, g_max,
    g_point[nwav]) as SingleGasData holds them."""
    w = NcWriter(path)
    ng = len(band_number)
    w.define_dimension("band", len(band_bound1))
    w.define_dimension("g_point", ng)
    for g in gases:
        w.define_dimension(g["name"] + "_g_point", len(g["rank1"]))
    w.define_dimension("wavenumber", len(wavenumber))
    w.define_variable("n_gases", "int")
    w.define_variable("wavenumber1_band", "float", "band")
    w.define_variable("wavenumber2_band", "float", "band")
    w.define_variable("band_number", "short", "g_point")
    if solar_irradiance is not None:
        w.define_variable("solar_irradiance", "float", "g_point")
    for g in gases:
        m, d = g["name"], g["name"] + "_g_point"
        w.define_variable(m + "_n_g_points", "int", "band")
        w.define_variable(m + "_band_number", "short", d)
        w.define_variable(m + "_rank1", "int", d)
        w.define_variable(m + "_rank2", "int", d)
        w.define_variable(m + "_error", "float", d)
        w.define_variable(m + "_sorting_variable", "float", d)
        w.define_variable(m + "_g_min", "int", "g_point")
        w.define_variable(m + "_g_max", "int", "g_point")
    w.define_variable("wavenumber", "double", "wavenumber")
    w.define_variable("g_point", "short", "wavenumber")
    w.deflate_variable("g_point")                                          # find_g_points.cpp:1580
    for g in gases:
        w.define_variable(g["name"] + "_g_point", "short", "wavenumber")
        w.deflate_variable(g["name"] + "_g_point")                         # :1587
    w.write_attribute("constituent_id", " ".join(g["name"] for g in gases))
    if history:
        w.write_attribute("history", history)
    w.write_attribute("config", config_str)
    w.end_define_mode()
    w.write("n_gases", [len(gases)])
    w.write("wavenumber1_band", band_bound1)
    w.write("wavenumber2_band", band_bound2)
    w.write("band_number", band_number)
    if solar_irradiance is not None:
        w.write("solar_irradiance", solar_irradiance)
    for g in gases:
        m = g["name"]
        for k in ("n_g_points", "band_number", "rank1", "rank2", "error", "sorting_variable", "g_min", "g_max", "g_point"):
            w.write(m + "_" + k, g[k])
    w.write("wavenumber", wavenumber)
    w.write("g_point", g_point)
    w.close()


def read_g_points(path):
    """What create_look_up_table and optimize_lut read from a g-points file (create_look_up_table.cpp:86-106)."""
    with NcFile(path) as f:
        return dict(g_point=f.read("g_point").astype(np.int32), band_number=f.read("band_number").astype(np.int32),
                    wavenumber=f.read("wavenumber"), wavenumber1_band=f.read("wavenumber1_band"),
                    wavenumber2_band=f.read("wavenumber2_band"),
                    solar_irradiance=f.read("solar_irradiance") if f.exist("solar_irradiance") else None)
