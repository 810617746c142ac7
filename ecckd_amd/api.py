"""Host-side mirror of the reference's hot-path interface over the C ABI.

Function names and argument meaning follow the reference's C++ functions (cited per
function); arrays are torch tensors on the context's device (PyTorch is only the owner
of device memory here) or numpy arrays where the reference keeps small host vectors.
Every call goes through libecckd_hip.so; nothing here computes on the CPU.
"""
import ctypes as C
import weakref

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
        self._children = weakref.WeakSet()  # handles that must be destroyed before the context

    def close(self):
        if getattr(self, "handle", None):
            for child in list(self._children):
                child.close()
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

    def trim_cache(self):
        """Hand the device blocks that released handles have parked in the context back to the driver (ecckd_trim_cache)."""
        check(self.lib.ecckd_trim_cache(self.handle))

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

    def profile_enable(self, on=True):
        """on = True / 1: HIP events around every launch of the dominant kernels; on = N > 1: around every N-th launch of the
        sweep kernel (profile_get("k_rt_lw_bb.all") then counts all launches)."""
        check(self.lib.ecckd_profile_enable(self.handle, int(on)))

    def profile_get(self, kernel):
        """(calls, total_ms, units) of a dominant kernel since profile_enable()."""
        calls, ms, units = C.c_longlong(), C.c_double(), C.c_double()
        check(self.lib.ecckd_profile_get(self.handle, kernel.encode(), C.byref(calls), C.byref(ms), C.byref(units)))
        return calls.value, ms.value, units.value

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


# ---------------------------------------------------------------------------------------
# find_g_points

class PartitionSearch:
    """Equal-error partition search over a batched Python error callback
    (ecckd_partition_*; replaces class Equipartition, equipartition.h:63-208)."""

    def __init__(self, error_fn, resolution=0.0, partition_tolerance=0.05, partition_max_iterations=20,
                 line_search_max_iterations=10, cubic=False, minimize_frac_range=True, trace=False):
        self.lib = _lib.load_library()
        self.calls = []
        # trace=True: self.events lists, in order, ("req", bound1[], bound2[], error[]) for every evaluation and
        # ("dec", site, lhs, rhs, taken) for every comparison that steers the search (ecckd_partition_set_trace)
        self.events = [] if trace else None

        def cb(n, b1, b2, err, _user):
            try:
                bb1 = [b1[i] for i in range(n)]
                bb2 = [b2[i] for i in range(n)]
                e = error_fn(bb1, bb2)
                for i in range(n):
                    err[i] = e[i]
                self.calls.append((bb1, bb2, list(e)))
                if self.events is not None:
                    self.events.append(("req", bb1, bb2, list(e)))
                return 0
            except Exception:  # surfaced as PROCESSING_ERROR by the search
                import traceback
                traceback.print_exc()
                return _lib.PROCESSING_ERROR

        self._cb = _lib.ERROR_FN(cb)
        h = C.c_void_p()
        check(self.lib.ecckd_partition_create(C.cast(self._cb, C.c_void_p), None, C.byref(h)))
        self.handle = h
        check(self.lib.ecckd_partition_configure(h, resolution, partition_tolerance, partition_max_iterations,
                                                 line_search_max_iterations, int(cubic), int(minimize_frac_range)))
        if trace:
            ev = self.events
            self._trace_cb = _lib.TRACE_FN(lambda site, lhs, rhs, taken, _u: ev.append(("dec", site, lhs, rhs, taken)))
            check(self.lib.ecckd_partition_set_trace(h, C.cast(self._trace_cb, C.c_void_p), None))

    def __del__(self):
        try:
            self.lib.ecckd_partition_destroy(self.handle)
        except Exception:
            pass

    def equipartition_n(self, bounds):
        b = np.ascontiguousarray(bounds, dtype=np.float64).copy()
        ni = b.size - 1
        err = np.zeros(ni)
        st = C.c_int()
        check(self.lib.ecckd_partition_n(self.handle, ni, _hptr(b), _hptr(err), C.byref(st)))
        return st.value, b, err

    def equipartition_e(self, target_error, bound0=0.0, boundn=1.0, capacity=4096):
        b = np.zeros(capacity + 1)
        err = np.zeros(capacity)
        ni, st = C.c_int(), C.c_int()
        check(self.lib.ecckd_partition_e(self.handle, target_error, bound0, boundn, C.byref(ni), _hptr(b),
                                         _hptr(err), capacity, C.byref(st)))
        n = ni.value
        return st.value, b[:n + 1].copy(), err[:n].copy()


def planck_hl_sorted(ctx, temperature_hl, wavenumber, d_wavenumber, rank, out=None):
    """The Planck matrix (nlay+1, nwav) of the ordering `rank`, bit-identical to GasLW(...).view("planck_hl") for that
    ordering and temperature profile (ecckd_planck_hl_sorted_dev): what a process that does not hold the FIRST gas passes
    as planck_hl_reuse (find_g_points.cpp:529, :970-984).  Device tensors in, device tensor out."""
    torch = _torch()
    t = np.ascontiguousarray(temperature_hl, dtype=np.float64)
    nwav = rank.numel()
    if out is None:
        out = torch.empty((t.size, nwav), dtype=torch.float64, device=ctx.device)
    ctx.fence_from_torch()
    check(ctx.lib.ecckd_planck_hl_sorted_dev(ctx.handle, t.size - 1, nwav, _hptr(t), _dptr(wavenumber), _dptr(d_wavenumber),
                                             _dptr(rank), _dptr(out)))
    return out


class GasLW:
    """A prepared longwave gas (ecckd_gas_create_lw): find_g_points.cpp:872-1150 done once on
    the device, then batched interval errors (CkdEquipartition::calc_error, :291-405)."""

    def __init__(self, ctx, pressure_hl, temperature_hl, wavenumber, d_wavenumber, rank, optical_depth,
                 bg_optical_depth=None, averaging_method="transmission", flux_weight=0.02, min_pressure=0.0,
                 planck_hl_reuse=None):
        self.ctx = ctx
        self.lib = ctx.lib
        p = np.ascontiguousarray(pressure_hl, dtype=np.float64)
        t = np.ascontiguousarray(temperature_hl, dtype=np.float64)
        self.nlay = p.size - 1
        if optical_depth.dim() != 2 or optical_depth.shape[0] != self.nlay:
            raise EcckdError(_lib.PARAMETER_ERROR, "optical_depth must be (nlay, nwav)")
        self.nwav = optical_depth.shape[1]
        stride = optical_depth.stride(0) if self.nlay > 1 else self.nwav
        if bg_optical_depth is not None and (bg_optical_depth.shape != optical_depth.shape or
                                             bg_optical_depth.stride(0) != optical_depth.stride(0)):
            raise EcckdError(_lib.PARAMETER_ERROR, "background optical depth must match the target's layout")
        if averaging_method not in _lib.AVG:
            raise EcckdError(_lib.PARAMETER_ERROR, f'Averaging method "{averaging_method}" not understood')
        h = C.c_void_p()
        ctx.fence_from_torch()
        check(self.lib.ecckd_gas_create_lw(
            ctx.handle, self.nlay, self.nwav, _hptr(p), _hptr(t), _dptr(wavenumber), _dptr(d_wavenumber),
            _dptr(rank), _dptr(bg_optical_depth) if bg_optical_depth is not None else None,
            _od_type(bg_optical_depth) if bg_optical_depth is not None else 0,
            _dptr(optical_depth), _od_type(optical_depth), stride, _lib.AVG[averaging_method],
            float(flux_weight), float(min_pressure),
            C.c_void_p(planck_hl_reuse) if planck_hl_reuse else None, C.byref(h)))
        self.handle = h
        ctx._children.add(self)

    def close(self):
        if getattr(self, "handle", None):
            if self.ctx.handle:  # the context must outlive its gases
                self.lib.ecckd_gas_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def view_ptr(self, name):
        ptr, r, c = C.c_void_p(), C.c_size_t(), C.c_size_t()
        check(self.lib.ecckd_gas_view(self.handle, name.encode(), C.byref(ptr), C.byref(r), C.byref(c)))
        return ptr.value, r.value, c.value

    def view(self, name):
        """Copy of a resident sorted array as a numpy array (rows, cols)."""
        ptr, r, c = self.view_ptr(name)
        out = np.empty((r, c), dtype=np.float64)
        check(self.lib.ecckd_d2h(self.ctx.handle, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes))
        return out

    def layer_weight(self):
        w = np.empty(self.nlay)
        check(self.lib.ecckd_gas_layer_weight(self.handle, _hptr(w)))
        return w

    def comp_cost(self, reset=False):
        return float(self.lib.ecckd_gas_comp_cost(self.handle, int(reset)))

    def eval_stats(self):
        """What the memo of interval errors saved (ecckd_gas_eval_stats): dict(requests, memo_hits, points_requested,
        points_evaluated)."""
        rq, hit, pr, pe = C.c_longlong(), C.c_longlong(), C.c_double(), C.c_double()
        check(self.lib.ecckd_gas_eval_stats(self.handle, C.byref(rq), C.byref(hit), C.byref(pr), C.byref(pe)))
        return dict(requests=rq.value, memo_hits=hit.value, points_requested=pr.value, points_evaluated=pe.value)

    def reset_memo(self):
        """Forget the interval errors answered so far and zero the counters (ecckd_gas_reset_memo)."""
        check(self.lib.ecckd_gas_reset_memo(self.handle))

    def sweep_bytes_per_point(self):
        """Bytes the error sweep reads per point (ecckd_gas_sweep_bytes_per_point): 656 at 54 layers with a FLOAT background
        (kept as FLOAT pairs), 872 with a DOUBLE one."""
        b = C.c_double()
        check(self.lib.ecckd_gas_sweep_bytes_per_point(self.handle, C.byref(b)))
        return b.value

    def calc_error_batch(self, ibegin, npoints, bound1, bound2):
        b1 = np.ascontiguousarray(bound1, dtype=np.float64)
        b2 = np.ascontiguousarray(bound2, dtype=np.float64)
        err = np.empty(b1.size)
        check(self.lib.ecckd_calc_error_batch(self.handle, int(ibegin), int(npoints), b1.size, _hptr(b1),
                                              _hptr(b2), _hptr(err)))
        return err

    def fit_optical_depth(self, ibegin, npoints, bound1, bound2):
        """fit_optical_depth_{lw,sw,sw_total_trans} for a batch of intervals -> (n, nlay)."""
        b1 = np.ascontiguousarray(bound1, dtype=np.float64)
        b2 = np.ascontiguousarray(bound2, dtype=np.float64)
        out = np.empty((b1.size, self.nlay))
        check(self.lib.ecckd_fit_optical_depth(self.handle, int(ibegin), int(npoints), b1.size, _hptr(b1),
                                               _hptr(b2), _hptr(out)))
        return out

    def find_g_band(self, ibegin, iend, heating_rate_tolerance, tolerance_tolerance=0.02, max_iterations=60,
                    min_g_points=1, max_g_points=256, capacity=1024):
        """find_g_points.cpp:1152-1266 for one band -> (status, bounds, error, comp_cost)."""
        b = np.zeros(capacity + 1)
        e = np.zeros(capacity)
        ng, st, cc = C.c_int(), C.c_int(), C.c_double()
        check(self.lib.ecckd_find_g_band(self.handle, int(ibegin), int(iend), float(heating_rate_tolerance),
                                         float(tolerance_tolerance), int(max_iterations), int(min_g_points),
                                         int(max_g_points), C.byref(ng), _hptr(b), _hptr(e), capacity,
                                         C.byref(st), C.byref(cc)))
        n = ng.value
        return st.value, b[:n + 1].copy(), e[:n].copy(), cc.value


    def find_g_band_ex(self, ibegin, iend, heating_rate_tolerance, tolerance_tolerance=0.02, max_iterations=60,
                       min_g_points=1, max_g_points=256, subbands=None, g_split=0.0, base_split=1.0,
                       base_wn_bound=None, wavenumber=None, rank=None, capacity=1024):
        """Everything find_g_points.cpp:1152-1414 does for one band: sub-band searches (`subbands` =
        (isubband1, isubband2, iupperindex) from subband_setup), min/max restarts, the base split (`rank` is
        re-ranked in place when `base_wn_bound` has interior boundaries) and the rank range of each g point.
        -> dict(status, bounds, error, rank1, rank2, comp_cost)."""
        o = _lib.BandOptions()
        o.min_g_points, o.max_g_points = int(min_g_points), int(max_g_points)
        keep = []
        if subbands is not None:
            i1 = np.ascontiguousarray(subbands[0], dtype=np.int64)
            i2 = np.ascontiguousarray(subbands[1], dtype=np.int64)
            keep += [i1, i2]
            o.nsubband = i1.size
            o.isubband1 = i1.ctypes.data_as(C.POINTER(C.c_int64))
            o.isubband2 = i2.ctypes.data_as(C.POINTER(C.c_int64))
            o.iupperindex = int(subbands[2])
        o.g_split, o.base_split = float(g_split), float(base_split)
        if base_wn_bound is not None:
            bw = np.ascontiguousarray(base_wn_bound, dtype=np.float64)
            keep.append(bw)
            o.nbase_wn_bound = bw.size
            o.base_wn_bound = bw.ctypes.data_as(C.POINTER(C.c_double))
            if bw.size > 2:
                o.d_wavenumber, o.d_rank, o.nwav = wavenumber.data_ptr(), rank.data_ptr(), rank.numel()
        b = np.zeros(capacity + 1)
        e = np.zeros(capacity)
        r1 = np.zeros(capacity, dtype=np.int64)
        r2 = np.zeros(capacity, dtype=np.int64)
        ng, st, cc = C.c_int(), C.c_int(), C.c_double()
        self.ctx.fence_from_torch()
        check(self.lib.ecckd_find_g_band_ex(self.handle, int(ibegin), int(iend), float(heating_rate_tolerance),
                                            float(tolerance_tolerance), int(max_iterations), C.byref(o), C.byref(ng),
                                            _hptr(b), _hptr(e), r1.ctypes.data_as(C.POINTER(C.c_int64)),
                                            r2.ctypes.data_as(C.POINTER(C.c_int64)), capacity, C.byref(st), C.byref(cc)))
        n = ng.value
        return dict(status=st.value, bounds=b[:n + 1].copy(), error=e[:n].copy(), rank1=r1[:n].copy(),
                    rank2=r2[:n].copy(), comp_cost=cc.value)

    def _band_options(self, keep, min_g_points=1, max_g_points=256, subbands=None, g_split=0.0, base_split=1.0, base_wn_bound=None,
                      wavenumber=None, rank=None, band_albedo=0.0):
        o = _lib.BandOptions()
        o.min_g_points, o.max_g_points = int(min_g_points), int(max_g_points)
        o.band_albedo = float(band_albedo)            # find_g_bands_ex on a shortwave gas: this band's surface albedo
        if subbands is not None:
            i1 = np.ascontiguousarray(subbands[0], dtype=np.int64)
            i2 = np.ascontiguousarray(subbands[1], dtype=np.int64)
            keep += [i1, i2]
            o.nsubband = i1.size
            o.isubband1 = i1.ctypes.data_as(C.POINTER(C.c_int64))
            o.isubband2 = i2.ctypes.data_as(C.POINTER(C.c_int64))
            o.iupperindex = int(subbands[2])
        o.g_split, o.base_split = float(g_split), float(base_split)
        if base_wn_bound is not None:
            bw = np.ascontiguousarray(base_wn_bound, dtype=np.float64)
            keep.append(bw)
            o.nbase_wn_bound = bw.size
            o.base_wn_bound = bw.ctypes.data_as(C.POINTER(C.c_double))
            if bw.size > 2:
                o.d_wavenumber, o.d_rank, o.nwav = wavenumber.data_ptr(), rank.data_ptr(), rank.numel()
        return o

    def calc_error_multi(self, ibegin, npoints, bound1, bound2, band_albedo=None):
        """Interval errors of several bands in one batch (ecckd_calc_error_multi): interval k is the fraction
        [bound1[k], bound2[k]] of the band starting at sorted index ibegin[k] with npoints[k] points; shortwave:
        band_albedo[k] is the surface albedo of its band (None: the gas's band albedo)."""
        alb = None if band_albedo is None else np.ascontiguousarray(band_albedo, dtype=np.float64)
        ib = np.ascontiguousarray(ibegin, dtype=np.uint64)
        npt = np.ascontiguousarray(npoints, dtype=np.uint64)
        b1 = np.ascontiguousarray(bound1, dtype=np.float64)
        b2 = np.ascontiguousarray(bound2, dtype=np.float64)
        err = np.empty(b1.size)
        check(self.lib.ecckd_calc_error_multi(self.handle, b1.size, ib.ctypes.data_as(C.POINTER(C.c_size_t)),
                                              npt.ctypes.data_as(C.POINTER(C.c_size_t)), _hptr(alb) if alb is not None else None,
                                              _hptr(b1), _hptr(b2), _hptr(err)))
        return err

    def find_g_bands_ex(self, ibegin, iend, heating_rate_tolerance, tolerance_tolerance=0.02, max_iterations=60, options=None,
                        capacity=1024):
        """Every band of the gas at once (ecckd_find_g_bands_ex): the band searches run side by side and share their error
        batches.  `options`: one dict of find_g_band_ex keyword options per band (or None).  -> list of result dicts."""
        nband = len(ibegin)
        ib = np.ascontiguousarray(ibegin, dtype=np.uint64)
        ie = np.ascontiguousarray(iend, dtype=np.uint64)
        tol = np.ascontiguousarray(np.broadcast_to(np.asarray(heating_rate_tolerance, dtype=np.float64), (nband,)))
        keep = []
        opts = (_lib.BandOptions * nband)()
        for k in range(nband):
            opts[k] = self._band_options(keep, **((options[k] if options else None) or {}))
        b = np.zeros((nband, capacity + 1))
        e = np.zeros((nband, capacity))
        r1 = np.zeros((nband, capacity), dtype=np.int64)
        r2 = np.zeros((nband, capacity), dtype=np.int64)
        ng = np.zeros(nband, dtype=np.int32)
        st = np.zeros(nband, dtype=np.int32)
        cc = np.zeros(nband)
        self.ctx.fence_from_torch()
        check(self.lib.ecckd_find_g_bands_ex(self.handle, nband, ib.ctypes.data_as(C.POINTER(C.c_size_t)),
                                             ie.ctypes.data_as(C.POINTER(C.c_size_t)), _hptr(tol), float(tolerance_tolerance),
                                             int(max_iterations), C.cast(opts, C.c_void_p), ng.ctypes.data_as(C.POINTER(C.c_int)),
                                             _hptr(b), _hptr(e), r1.ctypes.data_as(C.POINTER(C.c_int64)),
                                             r2.ctypes.data_as(C.POINTER(C.c_int64)), capacity, st.ctypes.data_as(C.POINTER(C.c_int)),
                                             _hptr(cc)))
        return [dict(status=int(st[k]), bounds=b[k, :ng[k] + 1].copy(), error=e[k, :ng[k]].copy(), rank1=r1[k, :ng[k]].copy(),
                     rank2=r2[k, :ng[k]].copy(), comp_cost=float(cc[k])) for k in range(nband)]

    def median_sorting_variable(self, sorting_variable_sorted, ind1, ind2):
        """calc_median_sorting_variable (find_g_points.cpp:35-49) per g point; `sorting_variable_sorted` is a
        device tensor in this gas's sorted order."""
        i1 = np.ascontiguousarray(ind1, dtype=np.int64)
        i2 = np.ascontiguousarray(ind2, dtype=np.int64)
        out = np.zeros(i1.size)
        self.ctx.fence_from_torch()
        check(self.lib.ecckd_gas_median_sorting_variable(self.handle, _dptr(sorting_variable_sorted), i1.size,
                                                         i1.ctypes.data_as(C.POINTER(C.c_int64)),
                                                         i2.ctypes.data_as(C.POINTER(C.c_int64)), _hptr(out)))
        return out


class GasSearchJob:
    """The band searches of several prepared gases side by side on one device (ecckd_find_g_gases_begin / _add / _wait: one host
    thread and one HIP stream per gas, every gas the launch trains it runs alone).  add() starts a gas's search at once and
    returns: the caller goes on to load and prepare the next gas while the gases added so far are being searched; wait() joins
    the searches -> per gas (in the order added) the list of per-band result dicts of Gas.find_g_bands_ex.
    max_concurrent: gases at a time (1 = each gas searched inside add(), the reference's gas-after-gas order; 0 = what the host
    has cores for)."""

    def __init__(self, tolerance_tolerance=0.02, max_iterations=60, max_concurrent=0, capacity=1024):
        self.lib = _lib.load_library()
        self.capacity = capacity
        self.handle = C.c_void_p()
        check(self.lib.ecckd_find_g_gases_begin(float(tolerance_tolerance), int(max_iterations), int(max_concurrent), C.byref(self.handle)))
        self._keep, self._out = [], []

    def add(self, gas, ibegin, iend, heating_rate_tolerance, options=None):
        """Start the search of the bands [ibegin[k], iend[k]] of `gas` (arguments as Gas.find_g_bands_ex)."""
        capacity = self.capacity
        nband = len(ibegin)
        ib = np.ascontiguousarray(ibegin, dtype=np.uint64)
        ie = np.ascontiguousarray(iend, dtype=np.uint64)
        tol = np.ascontiguousarray(np.broadcast_to(np.asarray(heating_rate_tolerance, dtype=np.float64), (nband,)))
        opts = (_lib.BandOptions * nband)()
        for j in range(nband):
            opts[j] = gas._band_options(self._keep, **((options[j] if options else None) or {}))
        b = np.zeros((nband, capacity + 1)); e = np.zeros((nband, capacity))
        r1 = np.zeros((nband, capacity), dtype=np.int64); r2 = np.zeros((nband, capacity), dtype=np.int64)
        ng = np.zeros(nband, dtype=np.int32); st = np.zeros(nband, dtype=np.int32); cc = np.zeros(nband)
        q = _lib.GasSearch()
        q.gas, q.nband = gas.handle, nband
        q.ibegin, q.iend = ib.ctypes.data_as(C.POINTER(C.c_size_t)), ie.ctypes.data_as(C.POINTER(C.c_size_t))
        q.heating_rate_tolerance, q.opt = _hptr(tol), C.cast(opts, C.c_void_p)
        q.ng, q.bounds, q.error = ng.ctypes.data_as(C.POINTER(C.c_int)), _hptr(b), _hptr(e)
        q.rank1, q.rank2 = r1.ctypes.data_as(C.POINTER(C.c_int64)), r2.ctypes.data_as(C.POINTER(C.c_int64))
        q.capacity, q.status, q.comp_cost = capacity, st.ctypes.data_as(C.POINTER(C.c_int)), _hptr(cc)
        self._keep += [ib, ie, tol, opts, q, gas]
        self._out.append((nband, b, e, r1, r2, ng, st, cc))
        gas.ctx.fence_from_torch()
        check(self.lib.ecckd_find_g_gases_add(self.handle, C.byref(q)))

    def wait(self):
        h, self.handle = self.handle, None
        check(self.lib.ecckd_find_g_gases_wait(h))
        res = [[dict(status=int(st[j]), bounds=b[j, :ng[j] + 1].copy(), error=e[j, :ng[j]].copy(), rank1=r1[j, :ng[j]].copy(),
                     rank2=r2[j, :ng[j]].copy(), comp_cost=float(cc[j])) for j in range(nband)]
               for nband, b, e, r1, r2, ng, st, cc in self._out]
        self._keep, self._out = [], []
        return res

    def __del__(self):
        if getattr(self, "handle", None):
            try:
                self.lib.ecckd_find_g_gases_wait(self.handle)     # never leave searches running on buffers that go away
            except Exception:                                     # noqa: BLE001
                pass
            self.handle = None


def find_g_gases(gases, requests, tolerance_tolerance=0.02, max_iterations=60, max_concurrent=0, capacity=1024):
    """All at once: requests[k] = dict(ibegin, iend, heating_rate_tolerance, options=None) as Gas.find_g_bands_ex takes them for
    gases[k] -> per gas the list of per-band result dicts.  See GasSearchJob."""
    job = GasSearchJob(tolerance_tolerance, max_iterations, 1 if len(gases) == 1 else max_concurrent, capacity)
    first_error = None
    for gas, r in zip(gases, requests):
        try:
            job.add(gas, r["ibegin"], r["iend"], r["heating_rate_tolerance"], r.get("options"))
        except EcckdError as exc:
            first_error = exc
            break
    if first_error is not None:
        try:
            job.wait()
        except EcckdError:
            pass
        raise first_error
    return job.wait()


def regroup_rank_by_wavenumber(ctx, wavenumber, rank, rank_lo, rank_hi, wn_bound):
    """Stable re-ranking of the ranks [rank_lo, rank_hi] by wavenumber group (find_g_points.cpp:832-866,
    :1311-1346); `rank` (int32 device tensor, original order) is updated in place -> group sizes."""
    wb = np.ascontiguousarray(wn_bound, dtype=np.float64)
    cnt = np.zeros(wb.size - 1, dtype=np.int64)
    ctx.fence_from_torch()
    check(ctx.lib.ecckd_regroup_rank_by_wavenumber_dev(ctx.handle, rank.numel(), _dptr(wavenumber), _dptr(rank),
                                                       int(rank_lo), int(rank_hi), wb.size - 1, _hptr(wb),
                                                       cnt.ctypes.data_as(C.POINTER(C.c_int64))))
    return cnt


def subband_setup(ctx, wavenumber, rank, ibegin, iend, g_split, band_bound1, band_bound2, boundaries):
    """find_g_points.cpp:799-868 for one band -> (isubband1, isubband2, iupperindex) or None if not split."""
    bd = np.ascontiguousarray(boundaries, dtype=np.float64)
    i1 = np.zeros(bd.size + 1, dtype=np.int64)
    i2 = np.zeros(bd.size + 1, dtype=np.int64)
    nsub, iup = C.c_int(), C.c_int64()
    ctx.fence_from_torch()
    check(ctx.lib.ecckd_subband_setup_dev(ctx.handle, rank.numel(), _dptr(wavenumber), _dptr(rank), int(ibegin),
                                          int(iend), float(g_split), float(band_bound1), float(band_bound2), bd.size,
                                          _hptr(bd), C.byref(nsub), i1.ctypes.data_as(C.POINTER(C.c_int64)),
                                          i2.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(iup)))
    if nsub.value == 0:
        return None
    return i1[:nsub.value].copy(), i2[:nsub.value].copy(), int(iup.value)


def gather_f64(ctx, src, index):
    """dst[i] = src[index[i]] on the device (the reordering gathers of find_g_points.cpp:781,:865)."""
    import torch
    dst = torch.empty(index.numel(), dtype=torch.float64, device=src.device)
    ctx.fence_from_torch()
    check(ctx.lib.ecckd_gather_f64_dev(ctx.handle, index.numel(), _dptr(src), _dptr(index), _dptr(dst)))
    ctx.synchronize()          # the kernel ran on the context's stream, torch reads on its own
    return dst


def invert_permutation(ctx, perm):
    """inverse[perm[i]] = i (ireorder from irank, find_g_points.cpp:778-779)."""
    import torch
    inv = torch.empty_like(perm)
    ctx.fence_from_torch()
    check(ctx.lib.ecckd_invert_permutation_dev(ctx.handle, perm.numel(), _dptr(perm), _dptr(inv)))
    return inv


class GasSW(GasLW):
    """A prepared shortwave gas (ecckd_gas_create_sw); same batched-error interface as GasLW."""

    def __init__(self, ctx, pressure_hl, ssi, rank, optical_depth, bg_optical_depth=None,
                 averaging_method="total-transmission", flux_weight=0.02, min_pressure=0.0, cos_sza=0.5,
                 albedo=None, min_scaling=1.0, max_scaling=1.0):
        self.ctx = ctx
        self.lib = ctx.lib
        p = np.ascontiguousarray(pressure_hl, dtype=np.float64)
        self.nlay = p.size - 1
        if optical_depth.dim() != 2 or optical_depth.shape[0] != self.nlay:
            raise EcckdError(_lib.PARAMETER_ERROR, "optical_depth must be (nlay, nwav)")
        self.nwav = optical_depth.shape[1]
        stride = optical_depth.stride(0) if self.nlay > 1 else self.nwav
        if averaging_method not in _lib.AVG:
            raise EcckdError(_lib.PARAMETER_ERROR, f'Averaging method "{averaging_method}" not understood')
        # find_g_points.cpp:666-667
        min_scaling = min(0.5, min_scaling)
        max_scaling = max(2.5, max_scaling)
        h = C.c_void_p()
        ctx.fence_from_torch()
        check(self.lib.ecckd_gas_create_sw(
            ctx.handle, self.nlay, self.nwav, _hptr(p), _dptr(ssi), _dptr(albedo) if albedo is not None else None,
            _dptr(rank), _dptr(bg_optical_depth) if bg_optical_depth is not None else None,
            _od_type(bg_optical_depth) if bg_optical_depth is not None else 0,
            _dptr(optical_depth), _od_type(optical_depth), stride, _lib.AVG[averaging_method],
            float(flux_weight), float(min_pressure), float(cos_sza), float(min_scaling), float(max_scaling),
            C.byref(h)))
        self.handle = h
        self.min_scaling, self.max_scaling = min_scaling, max_scaling
        ctx._children.add(self)

    def set_band_albedo(self, albedo):
        check(self.lib.ecckd_gas_set_band_albedo(self.handle, float(albedo)))


# ---------------------------------------------------------------------------------------
# optimize_lut

CONC = {"none": 0, "linear": 1, "lut": 2, "relative-linear": 3}


def _f64c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _keep_ptr(keep, a, ctype=C.c_double):
    if a is None:
        return None
    keep.append(a)
    return a.ctypes.data_as(C.POINTER(ctype))


def _opt_model(model, keep):
    """dict -> ecckd_opt_model (arrays appended to `keep` stay alive with the caller)."""
    f64, ptr = _f64c, lambda a, ct=C.c_double: _keep_ptr(keep, a, ct)
    gases = (_lib.OptGas * len(model["gases"]))()
    for i, g in enumerate(model["gases"]):
        ma = f64(g["molar_abs"])
        gases[i].conc_dependence = CONC[g["conc"]]
        gases[i].is_active = int(g.get("active", True))
        gases[i].nconc = ma.shape[0] if g["conc"] == "lut" else 1
        gases[i].vmr = ptr(f64(g["vmr"])) if g.get("vmr") is not None else None
        gases[i].reference_vmr = float(g.get("reference_vmr", 0.0))
        gases[i].molar_abs = ptr(ma)
        gases[i].min_molar_abs = ptr(f64(g["min_molar_abs"])) if g.get("min_molar_abs") is not None else None
        gases[i].max_molar_abs = ptr(f64(g["max_molar_abs"])) if g.get("max_molar_abs") is not None else None
    keep.append(gases)
    temp = f64(model["temperature"])
    m = _lib.OptModel()
    if model.get("planck_function") is not None:
        pf = f64(model["planck_function"])
        m.ng, m.ntp = pf.shape[1], pf.shape[0]
        m.temperature_planck = ptr(f64(model["temperature_planck"]))
        m.planck_function = ptr(pf)
    else:                                      # shortwave model: solar irradiance instead of a Planck LUT
        m.ng, m.ntp = len(model["solar_irradiance"]), 0
    m.nt, m.np = temp.shape[0], temp.shape[1]
    m.log_pressure = ptr(f64(model["log_pressure"]))
    m.temperature = ptr(temp)
    m.iband_per_g = ptr(np.ascontiguousarray(model["iband_per_g"], dtype=np.int32), C.c_int)
    m.ngas = len(model["gases"])
    m.gases = gases
    m.logarithmic_interpolation = int(model.get("logarithmic_interpolation", False))
    m.solar_irradiance = ptr(f64(model["solar_irradiance"])) if model.get("solar_irradiance") is not None else None
    m.rayleigh_molar_scattering = (ptr(f64(model["rayleigh_molar_scattering"]))
                                   if model.get("rayleigh_molar_scattering") is not None else None)
    return m


def _opt_scene(sc, s, keep):
    """Fill one ecckd_opt_scene from a dict."""
    f64, ptr = _f64c, lambda a, ct=C.c_double: _keep_ptr(keep, a, ct)
    p = f64(s["pressure_hl"])
    sc.ncol, sc.nlay = p.shape[0], p.shape[1] - 1
    sc.pressure_hl = ptr(p)
    sc.temperature_hl = ptr(f64(s["temperature_hl"]))
    sc.vmr_fl = ptr(f64(s["vmr_fl"])) if s.get("vmr_fl") is not None else None
    sc.gas_present = (ptr(np.ascontiguousarray(s["gas_present"], dtype=np.int32), C.c_int)
                      if s.get("gas_present") is not None else None)
    sc.surf_emissivity = ptr(f64(s["surf_emissivity"])) if s.get("surf_emissivity") is not None else None
    if s.get("flux_dn") is not None:
        fd = f64(s["flux_dn"])
        sc.nband = fd.shape[2]
        sc.flux_dn = ptr(fd)
        sc.flux_up = ptr(f64(s["flux_up"]))
    for k in ("spectral_flux_dn_surf", "spectral_flux_up_toa", "mu0", "albedo", "spectral_boundary_weights",
              "relative_flux_dn", "relative_flux_up"):
        setattr(sc, k, ptr(f64(s[k])) if s.get(k) is not None else None)
    sc.tsi = float(s.get("tsi", 0.0))


class Optimizer:
    """ecckd_opt_*: the cost function / gradient of solve_adept.cpp:240-292 and the L-BFGS driver
    of :310-417 on the device.

    model: dict(log_pressure[np], temperature[nt,np], temperature_planck[ntp], planck_function[ntp,ng],
                iband_per_g[ng], gases=[dict(conc, active, molar_abs, vmr=None, reference_vmr=0,
                                             min_molar_abs=None, max_molar_abs=None), ...])
    scenes: list of dict(pressure_hl[ncol,nhl], temperature_hl, vmr_fl[ncol,ngas,nlay], flux_dn[ncol,nhl,nband],
                         flux_up, gas_present=None, surf_emissivity=None, spectral_flux_dn_surf=None,
                         spectral_flux_up_toa=None)
    """

    def __init__(self, ctx, model, scenes, **cfg):
        self.ctx = ctx
        self.lib = ctx.lib
        keep = []
        m = _opt_model(model, keep)
        sc = (_lib.OptScene * len(scenes))()
        for i, s in enumerate(scenes):
            _opt_scene(sc[i], s, keep)
        c = _lib.OptConfig()
        defaults = dict(flux_weight=0.2, flux_profile_weight=0.0, broadband_weight=0.5, spectral_boundary_weight=0.0,
                        negative_od_penalty=1.0e4, pressure_weight_power=0.5, prior_error=1.0, min_prior_error=0.0,
                        max_prior_error=0.0, prior_error_scaling=1.0, pressure_corr=0.8, temperature_corr=0.8,
                        conc_corr=0.8, cap_relative_linear=0.8)
        defaults.update(cfg)
        for k, v in defaults.items():
            setattr(c, k, float(v))
        h = C.c_void_p()
        check(self.lib.ecckd_opt_create(ctx.handle, C.byref(m), len(scenes), C.cast(sc, C.c_void_p), C.byref(c),
                                        C.byref(h)))
        self.handle = h
        self.nx = int(self.lib.ecckd_opt_nx(h))
        self.ng = m.ng
        self.ncol = sum(int(s.ncol) for s in sc)
        self.nlay = int(sc[0].nlay)
        ctx._children.add(self)

    def close(self):
        if getattr(self, "handle", None):
            if self.ctx.handle:
                self.lib.ecckd_opt_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_allreduce(self, group=None, add_prior=None):
        """Profile-sharded optimisation (ecckd_opt_set_allreduce): this handle holds one rank's share of the training
        profiles (shard.shard_scene_columns); [gradient, cost] is summed over `group` after every evaluation with one
        all-reduce.  The prior is added by rank 0 unless `add_prior` says otherwise."""
        import torch.distributed as dist
        from . import shard
        fn, keep = shard.make_allreduce_callback(group)
        self._allreduce = (fn, keep)                      # the library keeps the pointer: keep the object alive
        if add_prior is None:
            add_prior = dist.get_rank(group) == 0
        check(self.lib.ecckd_opt_set_allreduce(self.handle, C.cast(fn, C.c_void_p), None, int(bool(add_prior))))

    def set_evaluator(self, fn):
        """Run minimize() over fn(x) -> (J, gradient) (numpy, on the host) instead of the device's cost function
        (ecckd_opt_set_evaluator); None restores the device evaluation."""
        if fn is None:
            self._evaluator = None
            check(self.lib.ecckd_opt_set_evaluator(self.handle, None, None))
            return

        def cb(nx, px, pj, pg, _user):
            try:
                x = np.ctypeslib.as_array(px, shape=(nx,)).copy()
                J, g = fn(x)
                pj[0] = float(J)
                np.ctypeslib.as_array(pg, shape=(nx,))[:] = g
                return 0
            except Exception as exc:                        # never let an exception cross the C boundary
                import sys
                print(f"ecckd evaluator callback failed: {exc!r}", file=sys.stderr)
                return 1

        self._evaluator = _lib.EVALUATOR_FN(cb)
        check(self.lib.ecckd_opt_set_evaluator(self.handle, C.cast(self._evaluator, C.c_void_p), None))

    def set_progress(self, fn):
        """fn(iteration, cost, gradient_norm) once per L-BFGS iteration (report_progress, solve_adept.cpp:295-299); also
        starts the activity timers read by timings()."""
        self._progress = _lib.PROGRESS_FN(lambda it, cost, gnorm, _user: fn(it, cost, gnorm)) if fn is not None else None
        check(self.lib.ecckd_opt_set_progress(self.handle, C.cast(self._progress, C.c_void_p) if fn is not None else None, None))

    def timings(self):
        """Seconds in the reference's three activities (solve_adept.cpp:216-218): dict(minimizer, a_priori, radiative_transfer)."""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        check(self.lib.ecckd_opt_timings(self.handle, C.byref(a), C.byref(b), C.byref(c)))
        return {"minimizer": a.value, "a-priori": b.value, "radiative transfer": c.value}

    def initial_state(self, bounds=False):
        x = np.empty(self.nx)
        if not bounds:
            check(self.lib.ecckd_opt_initial_state(self.handle, _hptr(x), None, None))
            return x
        lo, hi = np.empty(self.nx), np.empty(self.nx)
        check(self.lib.ecckd_opt_initial_state(self.handle, _hptr(x), _hptr(lo), _hptr(hi)))
        return x, lo, hi

    def cost_grad(self, x, want_grad=True):
        x = np.ascontiguousarray(x, dtype=np.float64)
        J = C.c_double()
        g = np.empty(self.nx) if want_grad else None
        check(self.lib.ecckd_opt_cost_grad(self.handle, _hptr(x), C.byref(J), _hptr(g) if want_grad else None))
        return (J.value, g) if want_grad else J.value

    def forward(self, x, unclamped=False):
        """Total optical depths and CKD fluxes at state x; unclamped: without the clamp of negative optical depths (the
        reference's "relative_to" evaluation, optimize_lut.cpp:229-234)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        od = np.empty((self.ncol, self.nlay, self.ng))
        fl = np.empty((self.ncol, 2, self.nlay + 1, self.ng))
        check(self.lib.ecckd_opt_forward_ex(self.handle, _hptr(x), 1 if unclamped else 0, _hptr(od), _hptr(fl)))
        return od, fl

    def coefficients(self, x, gas, shape):
        out = np.empty(shape)
        xx = np.ascontiguousarray(x, dtype=np.float64)
        check(self.lib.ecckd_opt_coefficients(self.handle, _hptr(xx), int(gas), _hptr(out)))
        return out

    def minimize(self, max_iterations=100, convergence_criterion=0.02, bounded=True):
        x = np.empty(self.nx)
        st, it = C.c_int(), C.c_int()
        J, gn = C.c_double(), C.c_double()
        check(self.lib.ecckd_opt_minimize(self.handle, int(max_iterations), float(convergence_criterion),
                                          int(bounded), _hptr(x), C.byref(st), C.byref(it), C.byref(J), C.byref(gn)))
        return dict(x=x, status=st.value, iterations=it.value, cost=J.value, gradient_norm=gn.value)


# ---------------------------------------------------------------------------------------
# create_look_up_table

def run_ckd(ctx, model, scene, gases=None, scalings=None, per_gas=True):
    """run_ckd.cpp:27-373 for the profiles of `scene` (pressure_hl, temperature_hl, vmr_fl[ncol,ngas,nlay],
    gas_present; shortwave: mu0[ncol], tsi).  `gases` restricts the gas list like the "gases" key (:68-74),
    `scalings` = {gas index: factor} mirrors co2_scaling & co (:76-85, :288-307).  Returns the variables
    run_ckd writes, keyed by their NetCDF names."""
    ngas = len(model["gases"])
    names = [g.get("name", str(i)) for i, g in enumerate(model["gases"])]
    is_sw = model.get("solar_irradiance") is not None
    sc_d = dict(scene)
    if scalings:
        vmr = np.array(scene["vmr_fl"], dtype=np.float64, copy=True)
        for i, f in scalings.items():
            vmr[:, i, :] *= f
        sc_d["vmr_fl"] = vmr
    present = np.ones(ngas, dtype=np.int32) if scene.get("gas_present") is None else np.array(scene["gas_present"], dtype=np.int32)
    if gases is not None:
        present = present * np.array([1 if (i in gases or names[i] in gases) else 0 for i in range(ngas)], dtype=np.int32)
    p = _f64c(scene["pressure_hl"])
    ncol, nhl = p.shape
    nlay = nhl - 1
    ng = len(model["iband_per_g"])

    def one(mask, want_all):
        keep = []
        m = _opt_model(model, keep)
        sc = _lib.OptScene()
        _opt_scene(sc, dict(sc_d, gas_present=mask, flux_dn=None), keep)
        od = np.empty((ncol, nlay, ng))
        ray = np.empty((ncol, nlay, ng)) if (want_all and is_sw) else None
        pl = np.empty((ncol, nhl, ng)) if want_all else None
        fl = np.empty((ncol, 2, nhl, ng)) if want_all else None
        check(ctx.lib.ecckd_run_ckd(ctx.handle, C.byref(m), C.byref(sc), _hptr(od), _hptr(ray) if ray is not None else None,
                                    _hptr(pl) if pl is not None else None, _hptr(fl) if fl is not None else None))
        return od, ray, pl, fl

    od, ray, pl, fl = one(present, True)
    dom = "sw" if is_sw else "lw"
    out = {"pressure_hl": p, "optical_depth": od}
    if per_gas:
        for i in range(ngas):
            if present[i]:
                mask = np.zeros(ngas, dtype=np.int32)
                mask[i] = 1
                out[names[i] + "_optical_depth"] = one(mask, False)[0]
    if not is_sw:
        out["planck_hl"], out["planck_surf"] = pl, pl[:, -1, :].copy()
        out["spectral_flux_dn_lw"], out["spectral_flux_up_lw"] = fl[:, 0], fl[:, 1]
        out["flux_dn_lw"], out["flux_up_lw"] = fl[:, 0].sum(-1), fl[:, 1].sum(-1)
    else:
        out["rayleigh_optical_depth"] = ray
        out["incoming_sw"] = pl[:, 0, :].copy()
        out["spectral_flux_dn_direct_sw"] = fl[:, 0]
        out["flux_dn_direct_sw"] = fl[:, 0].sum(-1)
    return out


def derive_d_wavenumber(ctx, wavenumber):
    """read_spectrum.cpp:55-65 for a grid stored without d_wavenumber (device tensor in, device tensor out)."""
    import torch
    out = torch.empty_like(wavenumber)
    ctx.fence_from_torch()
    check(ctx.lib.ecckd_derive_d_wavenumber_dev(ctx.handle, wavenumber.numel(), _dptr(wavenumber), _dptr(out)))
    ctx.synchronize()
    return out


def merge_scaling(pressure_hl, scaling=-1.0, conc=-1.0, reference_surface_vmr=-1.0, vmr_fl=None, pressure_conc=None,
                  conc_req=None):
    """read_merged_spectrum.cpp:117-147 -> (scaling_profile[nlay], vmr_fl row as stored at :153-165)."""
    p = _f64c(pressure_hl)
    nlay = p.size - 1
    sp, vo = np.empty(nlay), np.empty(nlay)
    v = _f64c(vmr_fl) if vmr_fl is not None else None
    pc = _f64c(pressure_conc) if pressure_conc is not None else None
    cr = _f64c(conc_req) if conc_req is not None else None
    check(_lib.load_library().ecckd_merge_scaling(nlay, _hptr(p), float(scaling), float(conc), float(reference_surface_vmr),
                                                  _hptr(v) if v is not None else None, 0 if pc is None else pc.size,
                                                  _hptr(pc) if pc is not None else None,
                                                  _hptr(cr) if cr is not None else None, _hptr(sp), _hptr(vo)))
    return sp, vo


def merge_spectrum(ctx, optical_depth, scaling_profile, merged=None):
    """merged (+)= optical_depth * scaling(level) (read_merged_spectrum.cpp:152-166); FLOAT or DOUBLE device
    tensor (nlay, nwav) in, DOUBLE device tensor out (allocated when `merged` is None)."""
    import torch
    first = merged is None
    if first:
        merged = torch.empty(optical_depth.shape, dtype=torch.float64, device=optical_depth.device)
    sp = _f64c(scaling_profile)
    ctx.fence_from_torch()
    check(ctx.lib.ecckd_merge_spectrum_dev(ctx.handle, optical_depth.shape[0], optical_depth.shape[1], _dptr(optical_depth),
                                           _od_type(optical_depth), optical_depth.stride(0), _hptr(sp), int(first),
                                           _dptr(merged), merged.stride(0)))
    return merged


def lbl_band_fluxes_lw(ctx, temperature_hl, wavenumber, d_wavenumber, optical_depth, band_begin, band_end, boundary=False, nangle=0):
    """Line-by-line longwave fluxes of one column summed per band (planck_function + radiative_transfer_lw):
    device tensors wavenumber, d_wavenumber, optical_depth (nlay, nwav) -> (flux_dn, flux_up), each (nband, nlay+1);
    boundary=True: also the spectral fluxes at the boundaries, device tensors (nwav,): (..., surface down, TOA up)."""
    t = _f64c(temperature_hl)
    nlay = t.size - 1
    b0 = np.ascontiguousarray(band_begin, dtype=np.int64)
    b1 = np.ascontiguousarray(band_end, dtype=np.int64)
    dn, up = np.empty((b0.size, nlay + 1)), np.empty((b0.size, nlay + 1))
    torch = _torch()
    nwav = optical_depth.shape[1]
    sdn = torch.empty(nwav, dtype=torch.float64, device=ctx.device) if boundary else None
    tup = torch.empty(nwav, dtype=torch.float64, device=ctx.device) if boundary else None
    ctx.fence_from_torch()
    # nangle = 0: the reference's two-stream form; N > 0: N Gauss-Legendre zenith angles per hemisphere (the CKDMIP tool's nangle)
    check(ctx.lib.ecckd_lbl_band_fluxes_lw_angles(ctx.handle, int(nangle), nlay, nwav, _hptr(t), _dptr(wavenumber),
                                                  _dptr(d_wavenumber), _dptr(optical_depth), _od_type(optical_depth),
                                                  optical_depth.stride(0), b0.size, _hptr(b0, C.c_int64), _hptr(b1, C.c_int64),
                                                  _hptr(dn), _hptr(up), _dptr(sdn) if boundary else None, _dptr(tup) if boundary else None))
    if boundary:
        ctx.synchronize()
        return dn, up, sdn, tup
    return dn, up


def gauss_legendre_01(n):
    """Nodes (ascending) and weights of the n-point Gauss-Legendre rule on (0, 1) (host only)."""
    from . import _lib
    mu, w = np.empty(n), np.empty(n)
    check(_lib.load_library().ecckd_gauss_legendre_01(int(n), _hptr(mu), _hptr(w)))
    return mu, w


def lbl_band_fluxes_sw(ctx, cos_sza, ssi, optical_depth, band_begin, band_end, albedo=None, boundary=False):
    """Line-by-line shortwave direct (and, with a per-wavenumber albedo, reflected) fluxes summed per band; boundary=True: also
    the spectral direct flux at the surface and the upwelling one at the top, device tensors (nwav,)."""
    nlay = optical_depth.shape[0]
    b0 = np.ascontiguousarray(band_begin, dtype=np.int64)
    b1 = np.ascontiguousarray(band_end, dtype=np.int64)
    dn, up = np.empty((b0.size, nlay + 1)), np.empty((b0.size, nlay + 1))
    torch = _torch()
    nwav = optical_depth.shape[1]
    sdn = torch.empty(nwav, dtype=torch.float64, device=ctx.device) if boundary else None
    tup = torch.empty(nwav, dtype=torch.float64, device=ctx.device) if boundary else None
    ctx.fence_from_torch()
    check(ctx.lib.ecckd_lbl_band_fluxes_sw_ex(ctx.handle, nlay, nwav, float(cos_sza), _dptr(ssi),
                                              _dptr(albedo) if albedo is not None else None, _dptr(optical_depth),
                                              _od_type(optical_depth), optical_depth.stride(0), b0.size, _hptr(b0, C.c_int64),
                                              _hptr(b1, C.c_int64), _hptr(dn), _hptr(up), _dptr(sdn) if boundary else None,
                                              _dptr(tup) if boundary else None))
    if boundary:
        ctx.synchronize()
        return dn, up, sdn, tup
    return dn, up


def scale_lut(ctx, model, flux_sums, pressure_hl, temperature_hl, vmr_fl, gas_present, mu0):
    """scale_lut.cpp:117-189 + CkdModel::scale_optical_depth for one reference profile.  `flux_sums` (nz+1, ng)
    from GPointMap.sum_rows of the LBL direct spectral flux.  -> (list of scaled molar_abs arrays, scaling[nz, ng])."""
    keep = []
    m = _opt_model(model, keep)
    p = _f64c(pressure_hl)
    nz = p.size - 1
    ng = len(model["iband_per_g"])
    outs = [np.empty_like(_f64c(g["molar_abs"])) for g in model["gases"]]
    arr = (C.POINTER(C.c_double) * len(outs))(*[_hptr(o) for o in outs])
    scaling = np.empty((nz, ng))
    gp = np.ascontiguousarray(gas_present, dtype=np.int32)
    vm = _f64c(vmr_fl)
    check(ctx.lib.ecckd_scale_lut(ctx.handle, C.byref(m), nz, _hptr(p), _hptr(_f64c(temperature_hl)), _hptr(vm),
                                  gp.ctypes.data_as(C.POINTER(C.c_int)), float(mu0), _hptr(_f64c(flux_sums)),
                                  _hptr(scaling), arr))
    return outs, scaling


class GPointMap:
    """ecckd_gmap_*: wavenumbers sorted by g point; the segmented reductions of
    create_look_up_table.cpp (average_optical_depth_to_g_point, gpoint_fraction, Planck LUT)."""

    def __init__(self, ctx, g_point, ng, wavenumber, d_wavenumber):
        self.ctx, self.lib, self.ng = ctx, ctx.lib, int(ng)
        self.nwav = g_point.numel()
        h = C.c_void_p()
        ctx.fence_from_torch()
        check(self.lib.ecckd_gmap_create(ctx.handle, self.nwav, _dptr(g_point), self.ng, _dptr(wavenumber),
                                         _dptr(d_wavenumber), C.byref(h)))
        self.handle = h
        ctx._children.add(self)

    def close(self):
        if getattr(self, "handle", None):
            if self.ctx.handle:
                self.lib.ecckd_gmap_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def counts(self):
        c = np.empty(self.ng, dtype=np.int64)
        check(self.lib.ecckd_gmap_counts(self.handle, _hptr(c, C.c_int64)))
        return c

    def average_optical_depth(self, pressure_hl, optical_depth, averaging_method="transmission",
                              reference_surface_vmr=1.0, temperature_fl=None, ssi=None):
        """average_optical_depth_to_g_point -> (molar_abs, min_molar_abs, max_molar_abs), each (nlay, ng)."""
        p = np.ascontiguousarray(pressure_hl, dtype=np.float64)
        nlay = p.size - 1
        if averaging_method not in _lib.AVG:
            raise EcckdError(_lib.PARAMETER_ERROR, f'averaging_method "{averaging_method}" not understood')
        t = np.ascontiguousarray(temperature_fl, dtype=np.float64) if temperature_fl is not None else None
        out = [np.empty((nlay, self.ng)) for _ in range(3)]
        stride = optical_depth.stride(0) if nlay > 1 else self.nwav
        self.ctx.fence_from_torch()
        check(self.lib.ecckd_average_to_gpoints(self.handle, nlay, _hptr(p), _hptr(t) if t is not None else None,
                                                _dptr(ssi) if ssi is not None else None, _dptr(optical_depth),
                                                _od_type(optical_depth), stride, _lib.AVG[averaging_method],
                                                float(reference_surface_vmr), _hptr(out[0]), _hptr(out[1]),
                                                _hptr(out[2])))
        return tuple(out)

    def sum_rows(self, rows):
        """Per-g-point sums of every row of a (nrows, nwav) device tensor (scale_lut.cpp:119-124) -> (nrows, ng)."""
        out = np.empty((rows.shape[0], self.ng))
        self.ctx.fence_from_torch()
        check(self.lib.ecckd_gmap_sum_rows(self.handle, rows.shape[0], _dptr(rows), _od_type(rows), rows.stride(0),
                                           _hptr(out)))
        return out

    def erythemal_spectrum(self):
        """sqrt(erythemal action spectrum) per g point, 5777 K Planck weighted (lbl_fluxes.cpp:198-230)."""
        out = np.empty(self.ng)
        check(self.lib.ecckd_gmap_erythemal_spectrum(self.handle, _hptr(out)))
        return out

    def gpoint_fraction(self, wavenumber1, wavenumber2):
        w1 = np.ascontiguousarray(wavenumber1, dtype=np.float64)
        w2 = np.ascontiguousarray(wavenumber2, dtype=np.float64)
        out = np.empty((self.ng, w1.size))
        check(self.lib.ecckd_gpoint_fraction(self.handle, w1.size, _hptr(w1), _hptr(w2), _hptr(out)))
        return out

    def planck_lut(self, temperature_lut):
        t = np.ascontiguousarray(temperature_lut, dtype=np.float64)
        out = np.empty((t.size, self.ng))
        check(self.lib.ecckd_planck_lut(self.handle, t.size, _hptr(t), _hptr(out)))
        return out


# ---------------------------------------------------------------------------------------
# find_g_points: overlap of the gases' g points (a14)

def overlap_g_points(n_g_points, sorting_variables):
    """overlap_g_points (single_gas_data.cpp:24-124).  n_g_points: (ngas, nband) ints;
    sorting_variables: list of per-gas arrays (one value per single-gas g point).
    Returns (ng, band_number[ng], g_min[ngas, ng], g_max[ngas, ng])."""
    lib = _lib.load_library()
    ngp = np.ascontiguousarray(n_g_points, dtype=np.int32)
    ngas, nband = ngp.shape
    offs = np.zeros(ngas, dtype=np.int32)
    offs[1:] = np.cumsum([len(sv) for sv in sorting_variables])[:-1]
    sv = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.float64) for x in sorting_variables]))
    cap = int(ngp.sum()) + 1
    ng = C.c_int()
    band = np.empty(cap, dtype=np.int32)
    gmin = np.empty((ngas, cap), dtype=np.int32)
    gmax = np.empty((ngas, cap), dtype=np.int32)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    check(lib.ecckd_overlap_g_points(ngas, nband, ip(ngp), ip(offs), _hptr(sv), cap, C.byref(ng), ip(band), ip(gmin),
                                     ip(gmax)))
    n = ng.value
    return n, band[:n].copy(), gmin[:, :n].copy(), gmax[:, :n].copy()


def gas_g_point(ctx, rank, rank1, rank2):
    """SingleGasData::store_g_points (single_gas_data.h:56-62) -> int32 device tensor."""
    torch = _torch()
    r1 = np.ascontiguousarray(rank1, dtype=np.int32)
    r2 = np.ascontiguousarray(rank2, dtype=np.int32)
    out = torch.empty(rank.numel(), dtype=torch.int32, device=ctx.device)
    ctx.fence_from_torch()
    check(ctx.lib.ecckd_gas_g_point_dev(ctx.handle, rank.numel(), _dptr(rank), r1.size, _hptr(r1, C.c_int32),
                                        _hptr(r2, C.c_int32), _dptr(out)))
    return out


def merge_g_points(ctx, gas_g_points, g_min, g_max):
    """find_g_points.cpp:1459-1475 -> (g_point int32 device tensor, number unassigned)."""
    torch = _torch()
    gmin = np.ascontiguousarray(g_min, dtype=np.int32)
    gmax = np.ascontiguousarray(g_max, dtype=np.int32)
    ngas, ng = gmin.shape
    n = gas_g_points[0].numel()
    ptrs = (C.c_void_p * ngas)(*[t.data_ptr() for t in gas_g_points])
    out = torch.empty(n, dtype=torch.int32, device=ctx.device)
    cnt = C.c_int64()
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    ctx.fence_from_torch()
    check(ctx.lib.ecckd_merge_g_points_dev(ctx.handle, n, ngas, ptrs, ng, ng, ip(gmin), ip(gmax), _dptr(out),
                                           C.byref(cnt)))
    return out, cnt.value


def inflate(ctx, streams, out_bytes):
    """zlib streams (bytes objects) inflated on the device (ecckd_inflate) -> (list of bytes, status array); out_bytes[s] =
    the number of bytes stream s must inflate to."""
    n = len(streams)
    raw = b"".join(streams)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(s) for s in streams])
    want = np.ascontiguousarray(out_bytes, dtype=np.uint64)
    buf = np.frombuffer(raw, dtype=np.uint8) if raw else np.zeros(1, dtype=np.uint8)
    out = np.zeros(max(int(want.sum()), 1), dtype=np.uint8)
    status = np.zeros(max(n, 1), dtype=np.int32)
    check(ctx.lib.ecckd_inflate(ctx.handle, n, buf.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.POINTER(C.c_ulonglong)),
                                want.ctypes.data_as(C.POINTER(C.c_ulonglong)), out.ctypes.data_as(C.c_void_p),
                                status.ctypes.data_as(C.POINTER(C.c_int))))
    ends = np.concatenate([[0], np.cumsum(want)]).astype(np.int64)
    return [out[ends[k]:ends[k + 1]].tobytes() for k in range(n)], status[:n]
