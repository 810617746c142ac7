"""Test helper: writes small HDF5 files the way the NetCDF-4 library lays out CKDMIP spectra (root-level datasets, chunked +
shuffle + deflate, scalar string attributes), by calling the system's HDF5 C library through ctypes.  Only used to make
inputs for the NetCDF-4 read path (ecckd_amd/csrc/nc_hdf5.cpp); there is no h5py / netCDF4 module in the image."""
import ctypes as C
import ctypes.util
import os

import numpy as np

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        names = [os.environ.get("ECCKD_HDF5_LIB"), "libhdf5.so", "libhdf5_serial.so", "libhdf5.so.103", "/opt/conda/lib/libhdf5.so",
                 ctypes.util.find_library("hdf5")]
        for n in names:
            if not n:
                continue
            try:
                _LIB = C.CDLL(n)
                break
            except OSError:
                continue
        if _LIB is None:
            return None
        h = _LIB
        hid = C.c_int64
        h.H5open()
        for f, res, args in (("H5Fcreate", hid, [C.c_char_p, C.c_uint, hid, hid]), ("H5Fclose", C.c_int, [hid]),
                             ("H5Screate_simple", hid, [C.c_int, C.POINTER(C.c_ulonglong), C.c_void_p]), ("H5Screate", hid, [C.c_int]),
                             ("H5Sclose", C.c_int, [hid]), ("H5Pcreate", hid, [hid]), ("H5Pclose", C.c_int, [hid]),
                             ("H5Pset_chunk", C.c_int, [hid, C.c_int, C.POINTER(C.c_ulonglong)]), ("H5Pset_shuffle", C.c_int, [hid]),
                             ("H5Pset_deflate", C.c_int, [hid, C.c_uint]),
                             ("H5Dcreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]),
                             ("H5Dwrite", C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]), ("H5Dclose", C.c_int, [hid]),
                             ("H5Tcopy", hid, [hid]), ("H5Tset_size", C.c_int, [hid, C.c_size_t]), ("H5Tclose", C.c_int, [hid]),
                             ("H5Acreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid]), ("H5Awrite", C.c_int, [hid, hid, C.c_void_p]),
                             ("H5Aclose", C.c_int, [hid]), ("H5Zfilter_avail", C.c_int, [C.c_int])):
            fn = getattr(h, f)
            fn.restype, fn.argtypes = res, args
    return _LIB


def _g(name):
    return C.c_int64.in_dll(lib(), name).value


def available():
    return lib() is not None and lib().H5Zfilter_avail(1) > 0          # 1 = H5Z_FILTER_DEFLATE


_TYPES = {"f4": "H5T_NATIVE_FLOAT_g", "f8": "H5T_NATIVE_DOUBLE_g", "i4": "H5T_NATIVE_INT_g", "i2": "H5T_NATIVE_SHORT_g"}


def _put_att(loc, name, value):
    h = lib()
    if isinstance(value, str) or isinstance(value, tuple):
        vlen = isinstance(value, tuple)                                  # ("text",) -> variable-length string (NC_STRING)
        text = (value[0] if vlen else value).encode()
        t = h.H5Tcopy(_g("H5T_C_S1_g"))
        h.H5Tset_size(t, C.c_size_t(-1).value if vlen else max(len(text), 1))
        sp = h.H5Screate(0)                                              # H5S_SCALAR
        a = h.H5Acreate2(loc, name.encode(), t, sp, 0, 0)
        if vlen:
            buf = C.c_char_p(text)
            assert h.H5Awrite(a, t, C.byref(buf)) >= 0
        else:
            assert h.H5Awrite(a, t, C.create_string_buffer(text, max(len(text), 1))) >= 0
        h.H5Aclose(a); h.H5Sclose(sp); h.H5Tclose(t)
    else:
        v = np.atleast_1d(np.asarray(value, dtype=np.float64))
        dims = (C.c_ulonglong * 1)(v.size)
        sp = h.H5Screate_simple(1, dims, None)
        a = h.H5Acreate2(loc, name.encode(), _g("H5T_NATIVE_DOUBLE_g"), sp, 0, 0)
        assert h.H5Awrite(a, _g("H5T_NATIVE_DOUBLE_g"), v.ctypes.data_as(C.c_void_p)) >= 0
        h.H5Aclose(a); h.H5Sclose(sp)


def write(path, variables, attributes=None):
    """variables: {name: (array, storage dtype "f4"|"f8"|"i4"|"i2", chunks or None, {attr: value})}; chunked variables are
    shuffled and deflated (level 2, as OutputDataFile::deflate_variable)."""
    h = lib()
    f = h.H5Fcreate(str(path).encode(), 2, 0, 0)                         # H5F_ACC_TRUNC
    assert f >= 0
    for name, (arr, store, chunks, atts) in variables.items():
        a = np.ascontiguousarray(arr, dtype=np.dtype(store))
        nd = max(a.ndim, 0)
        sp = h.H5Screate_simple(nd, (C.c_ulonglong * max(nd, 1))(*a.shape), None) if nd else h.H5Screate(0)
        dcpl = 0
        if chunks is not None:
            dcpl = h.H5Pcreate(_g("H5P_CLS_DATASET_CREATE_ID_g"))
            h.H5Pset_chunk(dcpl, nd, (C.c_ulonglong * nd)(*chunks))
            h.H5Pset_shuffle(dcpl)
            h.H5Pset_deflate(dcpl, 2)
        d = h.H5Dcreate2(f, name.encode(), _g(_TYPES[store]), sp, 0, dcpl, 0)
        assert d >= 0, name
        assert h.H5Dwrite(d, _g(_TYPES[store]), 0, 0, 0, a.ctypes.data_as(C.c_void_p)) >= 0
        for k, v in (atts or {}).items():
            _put_att(d, k, v)
        h.H5Dclose(d); h.H5Sclose(sp)
        if dcpl:
            h.H5Pclose(dcpl)
    for k, v in (attributes or {}).items():
        _put_att(f, k, v)
    h.H5Fclose(f)
