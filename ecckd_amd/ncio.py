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

    def end_define_mode(self):
        check(self.lib.ecckd_nc_enddef(self.handle))

    def write(self, name, data):
        d = np.ascontiguousarray(data, dtype=np.float64)
        check(self.lib.ecckd_nc_write_double(self.handle, _b(name), d.ctypes.data_as(C.POINTER(C.c_double)), d.size))

    def close(self):
        if getattr(self, "handle", None):
            h, self.handle = self.handle, None
            check(self.lib.ecckd_nc_close(h))


def read_spectrum(path, iprofile=0):
    """read_spectrum.cpp:20-87 for one column of a CKDMIP spectral file -> dict with the reference's names."""
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
        out["reference_surface_vmr"] = (float(f.read("reference_surface_mole_fraction"))
                                        if f.exist("reference_surface_mole_fraction") else -1.0)     # :68-73
        info = f.var_info("mole_fraction_fl")
        if info is not None and len(info[1]) == 2:                                                  # :76-82
            out["vmr_fl"] = f.read("mole_fraction_fl", iprofile)
        else:
            out["vmr_fl"] = np.full(out["pressure_hl"].size - 1, -1.0)
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
