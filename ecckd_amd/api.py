"""Host-side mirror of the reference's hot-path interface over the C ABI.

Function names and argument meaning follow the reference's C++ functions (cited per
function); arrays are torch tensors on the context's device (PyTorch is only the owner
of device memory here) or numpy arrays where the reference keeps small host vectors.
Every call goes through libecckd_hip.so; nothing here computes on the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import EcckdError, check  # noqa: F401


def _torch():
    import torch
    return torch


def _dptr(t):
    return C.c_void_p(t.data_ptr())


def _hptr(a, ctype=C.c_double):
    return a.ctypes.data_as(C.POINTER(ctype))


def _od_type(t):
    torch = _torch()
    if t.dtype == torch.float32:
        return _lib.F32
    if t.dtype == torch.float64:
        return _lib.F64
    raise TypeError(f"optical depth must be float32 or float64, got {t.dtype}")


class Context:
    """One device + one HIP stream (ecckd_init / ecckd_destroy)."""

    def __init__(self, device=0):
        self.lib = _lib.load_library()
        torch = _torch()
        if not torch.cuda.is_available():
            raise EcckdError(_lib.UNEXPECTED_EXCEPTION,
                             "no HIP device visible to PyTorch; ecckd_amd has no CPU fallback")
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        h = C.c_void_p()
        check(self.lib.ecckd_init(self.device_index, C.byref(h)))
        self.handle = h
        self._stream = None

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ecckd_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def synchronize(self):
        check(self.lib.ecckd_synchronize(self.handle))

    @property
    def torch_stream(self):
        """The context's HIP stream wrapped for torch (events must be recorded on it)."""
        if self._stream is None:
            torch = _torch()
            self._stream = torch.cuda.ExternalStream(self.lib.ecckd_stream(self.handle), device=self.device)
        return self._stream

    def fence_from_torch(self):
        """Make work queued on torch's current stream visible to the context's stream."""
        torch = _torch()
        torch.cuda.current_stream(self.device).synchronize()

    def timer_begin(self):
        check(self.lib.ecckd_timer_begin(self.handle))

    def timer_end(self):
        ms = C.c_float()
        check(self.lib.ecckd_timer_end(self.handle, C.byref(ms)))
        return float(ms.value)


def idealised_temperature(pressure_hl):
    """reorder_spectrum.cpp:121-124."""
    lib = _lib.load_library()
    p = np.ascontiguousarray(pressure_hl, dtype=np.float64)
    t = np.empty_like(p)
    check(lib.ecckd_idealised_temperature(p.size, _hptr(p), _hptr(t)))
    return t


def band_ranges(wavenumber, band_bound1, band_bound2):
    """reorder_spectrum.cpp:277-289 -> (iband[int16], begin[int64], end[int64])."""
    lib = _lib.load_library()
    wn = np.ascontiguousarray(wavenumber, dtype=np.float64)
    b1 = np.ascontiguousarray(band_bound1, dtype=np.float64)
    b2 = np.ascontiguousarray(band_bound2, dtype=np.float64)
    nband = b1.size
    iband = np.empty(wn.size, dtype=np.int16)
    bb = np.empty(nband, dtype=np.int64)
    be = np.empty(nband, dtype=np.int64)
    check(lib.ecckd_band_ranges(wn.size, _hptr(wn), nband, _hptr(b1), _hptr(b2),
                                _hptr(iband, C.c_int16), _hptr(bb, C.c_int64), _hptr(be, C.c_int64)))
    return iband, bb, be


def reorder_key_lw(ctx, pressure_hl, temperature_hl, wavenumber, d_wavenumber, optical_depth,
                   threshold_optical_depth=0.5, key=None, col_od=None):
    """K1 (synchronous: returns after the kernel's error flag has been read back).
    K1: reorder_spectrum.cpp:111-228, longwave.  Device tensors in, device tensors out."""
    torch = _torch()
    p = np.ascontiguousarray(pressure_hl, dtype=np.float64)
    t = np.ascontiguousarray(temperature_hl, dtype=np.float64)
    nlay = p.size - 1
    if optical_depth.dim() != 2 or optical_depth.shape[0] != nlay:
        raise EcckdError(_lib.PARAMETER_ERROR, "optical_depth must be (nlay, nwav)")
    nwav = optical_depth.shape[1]
    if optical_depth.stride(1) != 1 and nwav > 1:
        raise EcckdError(_lib.PARAMETER_ERROR, "optical_depth rows must be contiguous")
    stride = optical_depth.stride(0) if nlay > 1 else max(nwav, 1)
    if key is None:
        key = torch.empty(nwav, dtype=torch.float64, device=ctx.device)
    if col_od is None:
        col_od = torch.empty(nwav, dtype=torch.float64, device=ctx.device)
    ctx.fence_from_torch()
    check(ctx.lib.ecckd_reorder_key_lw_dev(ctx.handle, nlay, nwav, _hptr(p), _hptr(t), _dptr(wavenumber),
                                           _dptr(d_wavenumber), _dptr(optical_depth), _od_type(optical_depth),
                                           stride, float(threshold_optical_depth), _dptr(key), _dptr(col_od)))
    return key, col_od


def reorder_key_sw(ctx, pressure_hl, optical_depth, threshold_optical_depth=0.25, key=None, col_od=None):
    """K2: reorder_spectrum.cpp:150-158, :197-228, shortwave."""
    torch = _torch()
    p = np.ascontiguousarray(pressure_hl, dtype=np.float64)
    nlay = p.size - 1
    if optical_depth.dim() != 2 or optical_depth.shape[0] != nlay:
        raise EcckdError(_lib.PARAMETER_ERROR, "optical_depth must be (nlay, nwav)")
    nwav = optical_depth.shape[1]
    stride = optical_depth.stride(0) if nlay > 1 else max(nwav, 1)
    if key is None:
        key = torch.empty(nwav, dtype=torch.float64, device=ctx.device)
    if col_od is None:
        col_od = torch.empty(nwav, dtype=torch.float64, device=ctx.device)
    ctx.fence_from_torch()
    check(ctx.lib.ecckd_reorder_key_sw_dev(ctx.handle, nlay, nwav, _hptr(p), _dptr(optical_depth),
                                           _od_type(optical_depth), stride, float(threshold_optical_depth),
                                           _dptr(key), _dptr(col_od)))
    return key, col_od


def stable_argsort_bands(ctx, key, band_begin, band_end, rank=None, ordered_index=None, want_ordered=True,
                         sync=True):
    """K3: reorder_spectrum.cpp:262-300.  Returns (rank, ordered_index) int32 device tensors."""
    torch = _torch()
    nwav = key.numel()
    bb = np.ascontiguousarray(band_begin, dtype=np.int64)
    be = np.ascontiguousarray(band_end, dtype=np.int64)
    if rank is None:
        rank = torch.empty(nwav, dtype=torch.int32, device=ctx.device)
    if ordered_index is None and want_ordered:
        ordered_index = torch.empty(nwav, dtype=torch.int32, device=ctx.device)
    ctx.fence_from_torch()
    check(ctx.lib.ecckd_stable_argsort_bands_dev(ctx.handle, nwav, _dptr(key), bb.size, _hptr(bb, C.c_int64),
                                                 _hptr(be, C.c_int64), _dptr(rank),
                                                 _dptr(ordered_index) if ordered_index is not None else None))
    if sync:
        ctx.synchronize()
    return rank, ordered_index


def reorder_spectrum(ctx, pressure_hl, wavenumber, d_wavenumber, optical_depth, ssi=None,
                     threshold_optical_depth=0.5, band_bound1=None, band_bound2=None):
    """Host-array wrapper of the whole reorder path (ecckd_reorder_spectrum).

    Returns (sorting_variable, column_optical_depth, band_number, rank) as numpy arrays,
    the variables write_order.cpp:45-139 stores.
    """
    p = np.ascontiguousarray(pressure_hl, dtype=np.float64)
    wn = np.ascontiguousarray(wavenumber, dtype=np.float64)
    dwn = np.ascontiguousarray(d_wavenumber, dtype=np.float64)
    od = np.ascontiguousarray(optical_depth)
    if od.dtype == np.float32:
        od_type = _lib.F32
    elif od.dtype == np.float64:
        od_type = _lib.F64
    else:
        raise TypeError("optical_depth must be float32 or float64")
    nlay, nwav = (od.shape + (0,))[:2] if od.ndim == 2 else (0, 0)
    if band_bound1 is None:
        # reorder_spectrum.cpp:237-242
        band_bound1 = np.array([max(0.0, wn[0] - dwn[0])]) if nwav else np.array([0.0])
        band_bound2 = np.array([wn[-1] + dwn[-1]]) if nwav else np.array([0.0])
    b1 = np.ascontiguousarray(band_bound1, dtype=np.float64)
    b2 = np.ascontiguousarray(band_bound2, dtype=np.float64)
    key = np.empty(nwav, dtype=np.float64)
    col = np.empty(nwav, dtype=np.float64)
    iband = np.empty(nwav, dtype=np.int16)
    rank = np.empty(nwav, dtype=np.int32)
    s = np.ascontiguousarray(ssi, dtype=np.float64) if ssi is not None else None
    check(ctx.lib.ecckd_reorder_spectrum(ctx.handle, nlay, nwav, _hptr(p), _hptr(wn), _hptr(dwn),
                                         od.ctypes.data_as(C.c_void_p), od_type,
                                         _hptr(s) if s is not None else None,
                                         float(threshold_optical_depth), b1.size, _hptr(b1), _hptr(b2),
                                         _hptr(key), _hptr(col), _hptr(iband, C.c_int16),
                                         _hptr(rank, C.c_int32)))
    return key, col, iband, rank
