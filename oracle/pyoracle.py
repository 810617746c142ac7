"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module;
the product package ecckd_amd never does.  The library is oracle/libecckd_oracle.so
(restatement, built by oracle/Makefile) and, when present, oracle/_ref/
libequipartition_ref.so (the reference's own equipartition.cpp compiled where it lies).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORC = None
_REF = None

dp = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int32)

AVG = {"linear": 0, "transmission": 1, "transmission-2": 2, "square-root": 3,
       "logarithmic": 4, "total-transmission": 5, "transmission-3": 6, "transmission-10": 7,
       "hybrid-logarithmic-transmission-3": 8}


def build(force=False):
    """(Re)build the oracle with its Makefile (gcc; g++ for oracle/_ref)."""
    so = os.path.join(_HERE, "libecckd_oracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return so


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def lib():
    global _ORC
    if _ORC is None:
        so = build()
        _ORC = C.CDLL(so)
        _ORC.orc_calc_cost_function_lw.restype = C.c_double
        _ORC.orc_calc_cost_function_sw.restype = C.c_double
        _ORC.orc_ckd_calc_error.restype = C.c_double
        _ORC.orc_median_sorting_variable.restype = C.c_double
    return _ORC


def ref_lib():
    """The reference-built partition search, or None if oracle/_ref was never built."""
    global _REF
    if _REF is None:
        so = os.path.join(_HERE, "_ref", "libequipartition_ref.so")
        if not os.path.exists(so):
            return None
        _REF = C.CDLL(so)
        _REF.refep_create.restype = C.c_void_p
        _REF.refep_status_string.restype = C.c_char_p
    return _REF


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def planck_function(temperature, wavenumber, d_wavenumber):
    t, wn, dwn = _f64(np.atleast_1d(temperature)), _f64(wavenumber), _f64(d_wavenumber)
    out = np.empty((t.size, wn.size))
    lib().orc_planck_function(C.c_int(t.size), _p(t), C.c_size_t(wn.size), _p(wn), _p(dwn), _p(out))
    return out


def radiative_transfer_lw(planck, od, surf_emissivity, surf_planck):
    planck, od = _f64(planck), _f64(od)
    nlay, nwav = od.shape
    fdn = np.empty((nlay + 1, nwav))
    fup = np.empty((nlay + 1, nwav))
    lib().orc_radiative_transfer_lw(C.c_int(nlay), C.c_size_t(nwav), _p(planck), _p(od),
                                    _p(_f64(surf_emissivity)), _p(_f64(surf_planck)), _p(fdn), _p(fup))
    return fdn, fup


def radiative_transfer_lw_bb(planck, spectral_od, grey_od, surf_emissivity, surf_planck):
    planck, spectral_od, grey_od = _f64(planck), _f64(spectral_od), _f64(grey_od)
    nlay, nwav = spectral_od.shape
    fdn = np.empty(nlay + 1)
    fup = np.empty(nlay + 1)
    lib().orc_radiative_transfer_lw_bb(C.c_int(nlay), C.c_size_t(nwav), C.c_size_t(nwav), _p(planck),
                                       _p(spectral_od), _p(grey_od), _p(_f64(surf_emissivity)),
                                       _p(_f64(surf_planck)), _p(fdn), _p(fup))
    return fdn, fup


def radiative_transfer_direct_sw(cos_sza, ssi, od):
    od = _f64(od)
    nlay, nwav = od.shape
    fdn = np.empty((nlay + 1, nwav))
    lib().orc_radiative_transfer_direct_sw(C.c_int(nlay), C.c_size_t(nwav), C.c_double(cos_sza),
                                           _p(_f64(ssi)), _p(od), _p(fdn))
    return fdn


def radiative_transfer_norayleigh_sw(cos_sza, ssi, od, albedo):
    od = _f64(od)
    nlay, nwav = od.shape
    fdn = np.empty((nlay + 1, nwav))
    fup = np.empty((nlay + 1, nwav))
    lib().orc_radiative_transfer_norayleigh_sw(C.c_int(nlay), C.c_size_t(nwav), C.c_double(cos_sza),
                                               _p(_f64(ssi)), _p(od), _p(_f64(albedo)), _p(fdn), _p(fup))
    return fdn, fup


def heating_rate(pressure_hl, flux_dn, flux_up=None):
    flux_dn = _f64(flux_dn)
    nhl, nwav = flux_dn.shape
    hr = np.empty((nhl - 1, nwav))
    fu = _f64(flux_up) if flux_up is not None else None
    lib().orc_heating_rate(C.c_int(nhl - 1), C.c_size_t(nwav), _p(_f64(pressure_hl)), _p(flux_dn), _p(fu), _p(hr))
    return hr


def idealised_temperature(pressure_hl):
    p = _f64(pressure_hl)
    t = np.empty_like(p)
    lib().orc_idealised_temperature(C.c_int(p.size), _p(p), _p(t))
    return t


def reorder_key(pressure_hl, temperature_hl, wavenumber, d_wavenumber, od, ssi=None, thr=0.5):
    """a7; returns (key, col_od, status)."""
    od = _f64(od)
    nlay, nwav = od.shape
    key = np.empty(nwav)
    col = np.empty(nwav)
    t = _f64(temperature_hl) if temperature_hl is not None else None
    s = _f64(ssi) if ssi is not None else None
    lib().orc_reorder_key.restype = C.c_int
    st = lib().orc_reorder_key(C.c_int(nlay), C.c_size_t(nwav), _p(_f64(pressure_hl)), _p(t),
                               _p(_f64(wavenumber)), _p(_f64(d_wavenumber)), _p(od), _p(s),
                               C.c_double(thr), _p(key), _p(col))
    return key, col, st


def stable_argsort_bands(wavenumber, key, band_bound1, band_bound2):
    """a8; returns (iband, ordered_index, rank) int32."""
    wn, key = _f64(wavenumber), _f64(key)
    b1, b2 = _f64(band_bound1), _f64(band_bound2)
    n = wn.size
    iband = np.empty(n, dtype=np.int32)
    oi = np.empty(n, dtype=np.int32)
    rank = np.empty(n, dtype=np.int32)
    lib().orc_stable_argsort_bands(C.c_size_t(n), _p(wn), _p(key), C.c_int(b1.size), _p(b1), _p(b2),
                                   _p(iband, C.c_int32), _p(oi, C.c_int32), _p(rank, C.c_int32))
    return iband, oi, rank


def layer_weight(pressure_hl, min_pressure=0.0):
    p = _f64(pressure_hl)
    w = np.empty(p.size - 1)
    lib().orc_layer_weight(C.c_int(p.size - 1), _p(p), C.c_double(min_pressure), _p(w))
    return w


def metric(method, od):
    od = _f64(od)
    out = np.empty_like(od)
    lib().orc_metric(C.c_int(AVG[method]), C.c_size_t(od.size), _p(od), _p(out))
    return out


class OrcCkdEquipartition(C.Structure):
    _fields_ = [("do_sw", C.c_int), ("method", C.c_int), ("nlay", C.c_int), ("npoints", C.c_size_t),
                ("stride", C.c_size_t), ("flux_weight", C.c_double), ("cos_sza", C.c_double),
                ("surf_albedo", C.c_double), ("layer_weight", dp), ("pressure_hl", dp), ("ssi", dp),
                ("surf_emissivity", dp), ("surf_planck", dp), ("flux_dn_surf", dp), ("flux_up_toa", dp),
                ("planck_hl", dp), ("bg_od", dp), ("metric", dp), ("hr", dp),
                ("flux_dn_surf_low", dp), ("flux_up_toa_low", dp), ("flux_dn_surf_high", dp),
                ("flux_up_toa_high", dp), ("hr_low", dp), ("hr_high", dp),
                ("min_scaling", C.c_double), ("max_scaling", C.c_double), ("total_comp_cost", C.c_double)]


class CkdEquipartitionLW:
    """CkdEquipartition::init_lw + calc_error (find_g_points.cpp:208-405) on band-local arrays."""

    def __init__(self, method, flux_weight, layer_weight_, pressure_hl, surf_emissivity, surf_planck,
                 flux_dn_surf, flux_up_toa, planck_hl, bg_od, metric_, hr):
        self.keep = [_f64(a) for a in (layer_weight_, pressure_hl, surf_emissivity, surf_planck,
                                       flux_dn_surf, flux_up_toa, planck_hl, bg_od, metric_, hr)]
        lw, p, se, sp, fds, fut, pl, bg, me, h = self.keep
        nlay, n = bg.shape
        s = OrcCkdEquipartition()
        s.do_sw, s.method, s.nlay, s.npoints, s.stride = 0, AVG[method], nlay, n, n
        s.flux_weight = flux_weight
        s.layer_weight, s.pressure_hl = _p(lw), _p(p)
        s.surf_emissivity, s.surf_planck = _p(se), _p(sp)
        s.flux_dn_surf, s.flux_up_toa = _p(fds), _p(fut)
        s.planck_hl, s.bg_od, s.metric, s.hr = _p(pl), _p(bg), _p(me), _p(h)
        s.total_comp_cost = 0.0
        self.s = s

    def calc_error(self, b1, b2):
        st = C.c_int(0)
        e = lib().orc_ckd_calc_error(C.byref(self.s), C.c_double(b1), C.c_double(b2), C.byref(st))
        if st.value:
            raise RuntimeError(f"calc_error PROCESSING_ERROR path {st.value} for bounds {b1} {b2}")
        return e

    @property
    def total_comp_cost(self):
        return self.s.total_comp_cost


class CkdEquipartitionSW:
    """CkdEquipartition::init_sw(+extras) + calc_error on band-local arrays."""

    def __init__(self, method, flux_weight, layer_weight_, cos_sza, pressure_hl, ssi, surf_albedo,
                 flux_dn_surf, flux_up_toa, bg_od, metric_, hr, extras=None):
        self.keep = [_f64(a) for a in (layer_weight_, pressure_hl, ssi, flux_dn_surf, flux_up_toa, bg_od,
                                       metric_, hr)]
        lw, p, si, fds, fut, bg, me, h = self.keep
        nlay, n = bg.shape
        s = OrcCkdEquipartition()
        s.do_sw, s.method, s.nlay, s.npoints, s.stride = 1, AVG[method], nlay, n, n
        s.flux_weight, s.cos_sza, s.surf_albedo = flux_weight, cos_sza, surf_albedo
        s.layer_weight, s.pressure_hl, s.ssi = _p(lw), _p(p), _p(si)
        s.flux_dn_surf, s.flux_up_toa = _p(fds), _p(fut)
        s.bg_od, s.metric, s.hr = _p(bg), _p(me), _p(h)
        if extras is not None:
            self.keep2 = [_f64(extras[k]) for k in ("flux_dn_surf_low", "flux_up_toa_low", "flux_dn_surf_high",
                                                    "flux_up_toa_high", "hr_low", "hr_high")]
            (s.flux_dn_surf_low, s.flux_up_toa_low, s.flux_dn_surf_high, s.flux_up_toa_high,
             s.hr_low, s.hr_high) = [_p(a) for a in self.keep2]
            s.min_scaling, s.max_scaling = extras["min_scaling"], extras["max_scaling"]
        s.total_comp_cost = 0.0
        self.s = s

    calc_error = CkdEquipartitionLW.calc_error
    total_comp_cost = CkdEquipartitionLW.total_comp_cost


_CB = C.CFUNCTYPE(C.c_double, C.c_double, C.c_double, C.c_void_p)

EP_STATUS = ["EP_SUCCESS", "EP_MAX_ITERATIONS_REACHED", "EP_FAILED_TO_CONVERGE", "EP_RESOLUTION_LIMIT_REACHED",
             "EP_NO_PROGRESS", "EP_FAILURE", "EP_INPUT_ERROR"]


class RefEquipartition:
    """The REFERENCE's Equipartition (oracle/_ref) driven by a Python calc_error callable."""

    def __init__(self, calc_error, resolution=0.0, partition_tolerance=0.05, partition_max_iterations=20,
                 line_search_max_iterations=10, cubic=False, minimize_frac_range=True, verbose=0, parallel=False):
        r = ref_lib()
        if r is None:
            raise RuntimeError("oracle/_ref/libequipartition_ref.so not built")
        self.r = r
        self.calls = []

        def cb(b1, b2, _user):
            e = float(calc_error(b1, b2))
            self.calls.append((b1, b2, e))
            return e

        self._cb = _CB(cb)
        self.h = C.c_void_p(r.refep_create(self._cb, None))
        r.refep_set_verbose(self.h, C.c_int(verbose))
        r.refep_set_resolution(self.h, C.c_double(resolution))
        r.refep_set_partition_tolerance(self.h, C.c_double(partition_tolerance))
        r.refep_set_partition_max_iterations(self.h, C.c_int(partition_max_iterations))
        r.refep_set_line_search_max_iterations(self.h, C.c_int(line_search_max_iterations))
        r.refep_set_cubic_interpolation(self.h, C.c_int(1 if cubic else 0))
        r.refep_set_minimize_frac_range(self.h, C.c_int(1 if minimize_frac_range else 0))
        # find_g_points.cpp:231 turns the OpenMP loop of calc_error_all on (equipartition.h:100-104); the tests keep
        # it off so that the recorded call sequence is deterministic
        r.refep_set_parallel(self.h, C.c_int(1 if parallel else 0))

    def __del__(self):
        try:
            self.r.refep_destroy(self.h)
        except Exception:
            pass

    def equipartition_n(self, bounds):
        b = _f64(bounds).copy()
        ni = b.size - 1
        err = np.zeros(ni)
        st = self.r.refep_equipartition_n(self.h, C.c_int(ni), _p(b), _p(err))
        return st, b, err

    def equipartition_e(self, target_error, bound0=0.0, boundn=1.0, cap=4096):
        b = np.zeros(cap + 1)
        err = np.zeros(cap)
        ni = C.c_int(0)
        st = self.r.refep_equipartition_e(self.h, C.c_double(target_error), C.c_double(bound0),
                                          C.c_double(boundn), C.byref(ni), _p(b), _p(err), C.c_int(cap))
        n = ni.value
        return st, b[:n + 1].copy(), err[:n].copy()


def average_optical_depth_to_g_point(ng, reference_surface_vmr, pressure_hl, g_point, od, weight, method):
    """a15; returns (molar_abs, min, max, n_empty), each (nlay, ng)."""
    od, weight = _f64(od), _f64(weight)
    nlay, nwav = od.shape
    p = _f64(pressure_hl)
    pfl = _f64(0.5 * (p[1:] + p[:-1]))
    gp = np.ascontiguousarray(g_point, dtype=np.int32)
    out = [np.empty((nlay, ng)) for _ in range(3)]
    L = lib()
    L.orc_average_optical_depth_to_g_point.restype = C.c_int
    ne = L.orc_average_optical_depth_to_g_point(C.c_int(ng), C.c_double(reference_surface_vmr), C.c_int(nlay),
                                                C.c_size_t(nwav), _p(pfl), _p(p), _p(gp, C.c_int32), _p(od),
                                                _p(weight), C.c_int(AVG[method]), _p(out[0]), _p(out[1]), _p(out[2]))
    return out[0], out[1], out[2], ne


def gpoint_fraction(ng, g_point, wavenumber, d_wavenumber, wavenumber1, wavenumber2):
    gp = np.ascontiguousarray(g_point, dtype=np.int32)
    w1, w2 = _f64(wavenumber1), _f64(wavenumber2)
    out = np.empty((ng, w1.size))
    lib().orc_gpoint_fraction(C.c_int(ng), C.c_int(w1.size), C.c_size_t(gp.size), _p(gp, C.c_int32), _p(_f64(wavenumber)),
                              _p(_f64(d_wavenumber)), _p(w1), _p(w2), _p(out))
    return out


def planck_lut(ng, temperature_lut, g_point, wavenumber, d_wavenumber):
    gp = np.ascontiguousarray(g_point, dtype=np.int32)
    t = _f64(temperature_lut)
    out = np.empty((t.size, ng))
    lib().orc_planck_lut(C.c_int(ng), C.c_int(t.size), _p(t), C.c_size_t(gp.size), _p(gp, C.c_int32),
                         _p(_f64(wavenumber)), _p(_f64(d_wavenumber)), _p(out))
    return out


def overlap_g_points(n_g_points, sorting_variables):
    """a14 -- pure-Python restatement of overlap_g_points, reference src/ecckd/single_gas_data.cpp:24-124
    (small integer logic).  Returns (ng, band_number, g_min[ngas][ng], g_max[ngas][ng])."""
    ngp = np.asarray(n_g_points)
    ngas, nband = ngp.shape
    ng_band = [1 - ngas + int(ngp[:, b].sum()) for b in range(nband)]           # :30-38
    ng = sum(ng_band)
    band_number = np.repeat(np.arange(nband), ng_band)
    g_min = np.zeros((ngas, ng), dtype=np.int64)
    g_max = np.zeros((ngas, ng), dtype=np.int64)
    ig = 0
    ig_gas = [0] * ngas
    for b in range(nband):
        start = list(ig_gas)
        for i in range(ngas):                                                    # :63-70
            g_min[i, ig] = g_max[i, ig] = start[i]
        for _ in range(1, ng_band[b]):
            best, found = 1.0e30, -1
            for i in range(ngas):                                                # :73-90
                mine = 1.0e30
                if ig_gas[i] < start[i] + ngp[i, b] - 1:
                    mine = sorting_variables[i][ig_gas[i] + 1]
                if mine < best:
                    best, found = mine, i
            assert found >= 0
            ig_gas[found] += 1
            ig += 1
            for i in range(ngas):                                                # :103-112
                if i == found:
                    g_min[i, ig] = g_max[i, ig] = ig_gas[i]
                else:
                    g_min[i, ig], g_max[i, ig] = start[i], ig_gas[i]
        ig += 1                                                                  # :118-121
        ig_gas = [v + 1 for v in ig_gas]
    return ng, band_number, g_min, g_max


def median_sorting_variable(sorting_variable, weight, i1, i2):
    """calc_median_sorting_variable, find_g_points.cpp:35-49 (arrays in sorted order, inclusive indices)."""
    return float(lib().orc_median_sorting_variable(_p(_f64(sorting_variable)), _p(_f64(weight)),
                                                   C.c_size_t(int(i1)), C.c_size_t(int(i2))))
