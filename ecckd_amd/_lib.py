"""ctypes loader for libecckd_hip.so (the C ABI declared in include/ecckd_hip.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OK = 0
OUT_OF_MEMORY = 130
UNEXPECTED_EXCEPTION = 131
PARAMETER_ERROR = 147
PROCESSING_ERROR = 148

F32 = 4
F64 = 8

AVG = {
    "linear": 0,
    "transmission": 1,
    "transmission-2": 2,
    "square-root": 3,
    "logarithmic": 4,
    "total-transmission": 5,
    "transmission-3": 6,
    "transmission-10": 7,
    "hybrid-logarithmic-transmission-3": 8,
}


class EcckdError(RuntimeError):
    """Raised when an ABI call returns a non-zero reference exit code."""

    def __init__(self, code, message):
        super().__init__(f"ecckd error {code}: {message}")
        self.code = code
        self.message = message


def library_path():
    return os.path.join(_HERE, "libecckd_hip.so")


_c_double_p = C.POINTER(C.c_double)
_c_int64_p = C.POINTER(C.c_int64)
_c_int32_p = C.POINTER(C.c_int32)
_c_int16_p = C.POINTER(C.c_int16)

# name -> (restype, argtypes); kept in one table so the symbol-export test can
# check it against include/ecckd_hip.h.
SIGNATURES = {
    "ecckd_version": (C.c_int, []),
    "ecckd_last_error": (C.c_char_p, []),
    "ecckd_init": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "ecckd_destroy": (C.c_int, [C.c_void_p]),
    "ecckd_synchronize": (C.c_int, [C.c_void_p]),
    "ecckd_trim_cache": (C.c_int, [C.c_void_p]),
    "ecckd_stream": (C.c_void_p, [C.c_void_p]),
    "ecckd_dev_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "ecckd_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ecckd_mem_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "ecckd_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ecckd_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ecckd_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "ecckd_profile_get": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_longlong), _c_double_p, _c_double_p]),
    "ecckd_timer_begin": (C.c_int, [C.c_void_p]),
    "ecckd_timer_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "ecckd_idealised_temperature": (C.c_int, [C.c_int, _c_double_p, _c_double_p]),
    "ecckd_reorder_key_lw_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, _c_double_p, _c_double_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t,
                                           C.c_double, C.c_void_p, C.c_void_p]),
    "ecckd_reorder_key_sw_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, _c_double_p,
                                           C.c_void_p, C.c_int, C.c_size_t, C.c_double,
                                           C.c_void_p, C.c_void_p]),
    "ecckd_stable_argsort_bands_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int,
                                                 _c_int64_p, _c_int64_p, C.c_void_p, C.c_void_p]),
    "ecckd_band_ranges": (C.c_int, [C.c_size_t, _c_double_p, C.c_int, _c_double_p, _c_double_p,
                                    _c_int16_p, _c_int64_p, _c_int64_p]),
    "ecckd_reorder_spectrum": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, _c_double_p, _c_double_p,
                                         _c_double_p, C.c_void_p, C.c_int, _c_double_p, C.c_double,
                                         C.c_int, _c_double_p, _c_double_p, _c_double_p, _c_double_p,
                                         _c_int16_p, _c_int32_p]),
    "ecckd_reorder_spectrum_od_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, _c_double_p, _c_double_p,
                                                _c_double_p, C.c_void_p, C.c_int, _c_double_p, C.c_double,
                                                C.c_int, _c_double_p, _c_double_p, _c_double_p, _c_double_p,
                                                _c_int16_p, _c_int32_p]),
    "ecckd_gas_create_lw": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, _c_double_p, _c_double_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_double, C.c_double,
                                      C.c_void_p, C.POINTER(C.c_void_p)]),
    "ecckd_planck_hl_sorted_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, _c_double_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p]),
    "ecckd_gas_create_sw": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, _c_double_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_size_t,
                                      C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                      C.POINTER(C.c_void_p)]),
    "ecckd_gas_set_band_albedo": (C.c_int, [C.c_void_p, C.c_double]),
    "ecckd_gas_eval_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), _c_double_p, _c_double_p]),
    "ecckd_gas_sweep_bytes_per_point": (C.c_int, [C.c_void_p, _c_double_p]),
    "ecckd_gas_destroy": (C.c_int, [C.c_void_p]),
    "ecckd_gas_view": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                 C.POINTER(C.c_size_t)]),
    "ecckd_gas_layer_weight": (C.c_int, [C.c_void_p, _c_double_p]),
    "ecckd_gas_comp_cost": (C.c_double, [C.c_void_p, C.c_int]),
    "ecckd_calc_error_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, _c_double_p,
                                         _c_double_p, _c_double_p]),
    "ecckd_fit_optical_depth": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, _c_double_p,
                                          _c_double_p, _c_double_p]),
    "ecckd_partition_create": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "ecckd_partition_destroy": (C.c_int, [C.c_void_p]),
    "ecckd_partition_configure": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int,
                                            C.c_int]),
    "ecckd_partition_n": (C.c_int, [C.c_void_p, C.c_int, _c_double_p, _c_double_p, C.POINTER(C.c_int)]),
    "ecckd_partition_e": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_int),
                                    _c_double_p, _c_double_p, C.c_int, C.POINTER(C.c_int)]),
    "ecckd_partition_set_trace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ecckd_partition_status_string": (C.c_char_p, [C.c_int]),
    "ecckd_opt_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "ecckd_opt_destroy": (C.c_int, [C.c_void_p]),
    "ecckd_opt_nx": (C.c_size_t, [C.c_void_p]),
    "ecckd_opt_initial_state": (C.c_int, [C.c_void_p, _c_double_p, _c_double_p, _c_double_p]),
    "ecckd_opt_cost_grad": (C.c_int, [C.c_void_p, _c_double_p, _c_double_p, _c_double_p]),
    "ecckd_opt_forward": (C.c_int, [C.c_void_p, _c_double_p, _c_double_p, _c_double_p]),
    "ecckd_opt_set_evaluator": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ecckd_opt_set_progress": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ecckd_opt_timings": (C.c_int, [C.c_void_p, _c_double_p, _c_double_p, _c_double_p]),
    "ecckd_opt_forward_ex": (C.c_int, [C.c_void_p, _c_double_p, C.c_int, _c_double_p, _c_double_p]),
    "ecckd_opt_coefficients": (C.c_int, [C.c_void_p, _c_double_p, C.c_int, _c_double_p]),
    "ecckd_opt_minimize": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, _c_double_p, C.POINTER(C.c_int),
                                     C.POINTER(C.c_int), _c_double_p, _c_double_p]),
    "ecckd_gmap_create": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                    C.POINTER(C.c_void_p)]),
    "ecckd_gmap_destroy": (C.c_int, [C.c_void_p]),
    "ecckd_gmap_counts": (C.c_int, [C.c_void_p, _c_int64_p]),
    "ecckd_average_to_gpoints": (C.c_int, [C.c_void_p, C.c_int, _c_double_p, _c_double_p, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_size_t, C.c_int, C.c_double, _c_double_p, _c_double_p,
                                           _c_double_p]),
    "ecckd_gpoint_fraction": (C.c_int, [C.c_void_p, C.c_int, _c_double_p, _c_double_p, _c_double_p]),
    "ecckd_planck_lut": (C.c_int, [C.c_void_p, C.c_int, _c_double_p, _c_double_p]),
    "ecckd_overlap_g_points": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), _c_double_p, C.c_int,
                                         C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                         C.POINTER(C.c_int)]),
    "ecckd_gas_g_point_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, _c_int32_p, _c_int32_p,
                                        C.c_void_p]),
    "ecckd_merge_g_points_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int,
                                           C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, _c_int64_p]),
    "ecckd_find_g_band": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_double, C.c_double, C.c_int,
                                    C.c_int, C.c_int, C.POINTER(C.c_int), _c_double_p, _c_double_p, C.c_int,
                                    C.POINTER(C.c_int), _c_double_p]),
    "ecckd_regroup_rank_by_wavenumber_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                                       C.c_size_t, C.c_int, _c_double_p, _c_int64_p]),
    "ecckd_subband_setup_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                          C.c_double, C.c_double, C.c_double, C.c_int, _c_double_p,
                                          C.POINTER(C.c_int), _c_int64_p, _c_int64_p, _c_int64_p]),
    "ecckd_find_g_band_ex": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_double, C.c_double, C.c_int,
                                       C.c_void_p, C.POINTER(C.c_int), _c_double_p, _c_double_p, _c_int64_p,
                                       _c_int64_p, C.c_int, C.POINTER(C.c_int), _c_double_p]),
    "ecckd_gas_median_sorting_variable": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _c_int64_p, _c_int64_p,
                                                    _c_double_p]),
    "ecckd_run_ckd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    "ecckd_scale_lut": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _c_double_p, _c_double_p, _c_double_p,
                                  C.POINTER(C.c_int), C.c_double, _c_double_p, _c_double_p, C.POINTER(_c_double_p)]),
    "ecckd_gmap_sum_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_size_t, _c_double_p]),
    "ecckd_derive_d_wavenumber_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ecckd_merge_scaling": (C.c_int, [C.c_int, _c_double_p, C.c_double, C.c_double, C.c_double, _c_double_p, C.c_int,
                                      _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    "ecckd_merge_spectrum_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t, _c_double_p,
                                           C.c_int, C.c_void_p, C.c_size_t]),
    "ecckd_gmap_erythemal_spectrum": (C.c_int, [C.c_void_p, _c_double_p]),
    "ecckd_nc_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "ecckd_nc_close": (C.c_int, [C.c_void_p]),
    "ecckd_nc_inq_dim": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_size_t)]),
    "ecckd_nc_inq_var": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_size_t), C.c_int]),
    "ecckd_nc_read_double": (C.c_int, [C.c_void_p, C.c_char_p, C.c_longlong, _c_double_p, C.c_size_t]),
    "ecckd_nc_read_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_longlong, C.c_int, C.c_void_p, C.c_size_t]),
    "ecckd_inflate": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_void_p,
                                C.POINTER(C.c_int)]),
    "ecckd_inflate_host": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]),
    "ecckd_nc_read_att_text": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "ecckd_nc_read_att_double": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), _c_double_p, C.c_size_t]),
    "ecckd_nc_create": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "ecckd_nc_def_dim": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    "ecckd_nc_def_var": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ecckd_nc_put_att_text": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p]),
    "ecckd_nc_put_att_double": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, _c_double_p]),
    "ecckd_nc_deflate_var": (C.c_int, [C.c_void_p, C.c_char_p]),
    "ecckd_nc_is_netcdf4": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "ecckd_nc_enddef": (C.c_int, [C.c_void_p]),
    "ecckd_nc_write_double": (C.c_int, [C.c_void_p, C.c_char_p, _c_double_p, C.c_size_t]),
    "ecckd_nc_write_slice_double": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, _c_double_p, C.c_size_t]),
    "ecckd_write_order_file": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, _c_double_p, _c_double_p,
                                         C.c_size_t, _c_double_p, _c_double_p, _c_int16_p, _c_int32_p, _c_double_p,
                                         _c_double_p]),
    "ecckd_lbl_band_fluxes_lw": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, _c_double_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_size_t, C.c_int, _c_int64_p, _c_int64_p, _c_double_p, _c_double_p]),
    "ecckd_lbl_band_fluxes_sw": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_size_t, C.c_int, _c_int64_p, _c_int64_p, _c_double_p, _c_double_p]),
    "ecckd_lbl_band_fluxes_lw_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, _c_double_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_int, C.c_size_t, C.c_int, _c_int64_p, _c_int64_p, _c_double_p, _c_double_p,
                                              C.c_void_p, C.c_void_p]),
    "ecckd_lbl_band_fluxes_lw_angles": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_size_t, _c_double_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_int, C.c_size_t, C.c_int, _c_int64_p, _c_int64_p, _c_double_p, _c_double_p,
                                              C.c_void_p, C.c_void_p]),
    "ecckd_gauss_legendre_01": (C.c_int, [C.c_int, _c_double_p, _c_double_p]),
    "ecckd_rt_lw_gpoints": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    "ecckd_rt_sw_gpoints": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _c_double_p, _c_double_p, _c_double_p,
                                      _c_double_p]),
    "ecckd_lbl_band_fluxes_sw_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_int, C.c_size_t, C.c_int, _c_int64_p, _c_int64_p, _c_double_p, _c_double_p,
                                              C.c_void_p, C.c_void_p]),
    "ecckd_gather_f64_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ecckd_invert_permutation_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ecckd_calc_error_multi": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), _c_double_p, _c_double_p,
                                         _c_double_p, _c_double_p]),
    "ecckd_find_g_bands_ex": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), _c_double_p, C.c_double,
                                        C.c_int, C.c_void_p, C.POINTER(C.c_int), _c_double_p, _c_double_p, _c_int64_p, _c_int64_p,
                                        C.c_int, C.POINTER(C.c_int), _c_double_p]),
    "ecckd_gas_reset_memo": (C.c_int, [C.c_void_p]),
    "ecckd_find_g_gases": (C.c_int, [C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_int]),
    "ecckd_find_g_gases_begin": (C.c_int, [C.c_double, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "ecckd_find_g_gases_add": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ecckd_find_g_gases_wait": (C.c_int, [C.c_void_p]),
    "ecckd_opt_set_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "ecckd_cfg_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "ecckd_cfg_from_args": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p)]),
    "ecckd_cfg_append_file": (C.c_int, [C.c_void_p, C.c_char_p]),
    "ecckd_cfg_append_text": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "ecckd_cfg_register": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "ecckd_cfg_destroy": (C.c_int, [C.c_void_p]),
    "ecckd_cfg_file_name": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "ecckd_cfg_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "ecckd_cfg_entry": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t),
                                  C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ecckd_cfg_exists": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]),
    "ecckd_cfg_get_boolean": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]),
    "ecckd_cfg_get_int": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ecckd_cfg_get_real": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, _c_double_p, C.POINTER(C.c_int)]),
    "ecckd_cfg_get_string": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t,
                                       C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "ecckd_cfg_size": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                 C.POINTER(C.c_int)]),
    "ecckd_cfg_get_real_vector": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, _c_double_p, C.c_int, C.POINTER(C.c_int)]),
    "ecckd_cfg_get_int_vector": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_int,
                                           C.POINTER(C.c_int)]),
    "ecckd_cfg_sprint": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
}


class BandOptions(C.Structure):
    """ecckd_band_options (include/ecckd_hip.h)."""
    _fields_ = [("min_g_points", C.c_int), ("max_g_points", C.c_int), ("nsubband", C.c_int),
                ("isubband1", _c_int64_p), ("isubband2", _c_int64_p), ("iupperindex", C.c_int64),
                ("g_split", C.c_double), ("base_split", C.c_double), ("nbase_wn_bound", C.c_int),
                ("base_wn_bound", _c_double_p), ("d_wavenumber", C.c_void_p), ("d_rank", C.c_void_p),
                ("nwav", C.c_size_t), ("band_albedo", C.c_double)]


class GasSearch(C.Structure):
    """ecckd_gas_search (include/ecckd_hip.h): one gas's request of ecckd_find_g_gases."""
    _fields_ = [("gas", C.c_void_p), ("nband", C.c_int), ("ibegin", C.POINTER(C.c_size_t)), ("iend", C.POINTER(C.c_size_t)),
                ("heating_rate_tolerance", _c_double_p), ("opt", C.c_void_p), ("ng", C.POINTER(C.c_int)), ("bounds", _c_double_p),
                ("error", _c_double_p), ("rank1", _c_int64_p), ("rank2", _c_int64_p), ("capacity", C.c_int),
                ("status", C.POINTER(C.c_int)), ("comp_cost", _c_double_p), ("rc", C.c_int)]


class OptGas(C.Structure):
    _fields_ = [("conc_dependence", C.c_int), ("is_active", C.c_int), ("nconc", C.c_int), ("vmr", _c_double_p),
                ("reference_vmr", C.c_double), ("molar_abs", _c_double_p), ("min_molar_abs", _c_double_p),
                ("max_molar_abs", _c_double_p)]


class OptModel(C.Structure):
    _fields_ = [("ng", C.c_int), ("nt", C.c_int), ("np", C.c_int), ("log_pressure", _c_double_p),
                ("temperature", _c_double_p), ("ntp", C.c_int), ("temperature_planck", _c_double_p),
                ("planck_function", _c_double_p), ("iband_per_g", C.POINTER(C.c_int)), ("ngas", C.c_int),
                ("gases", C.POINTER(OptGas)), ("logarithmic_interpolation", C.c_int),
                ("solar_irradiance", _c_double_p), ("rayleigh_molar_scattering", _c_double_p)]


class OptScene(C.Structure):
    _fields_ = [("ncol", C.c_int), ("nlay", C.c_int), ("nband", C.c_int), ("pressure_hl", _c_double_p),
                ("temperature_hl", _c_double_p), ("vmr_fl", _c_double_p), ("gas_present", C.POINTER(C.c_int)),
                ("surf_emissivity", _c_double_p), ("flux_dn", _c_double_p), ("flux_up", _c_double_p),
                ("spectral_flux_dn_surf", _c_double_p), ("spectral_flux_up_toa", _c_double_p),
                ("mu0", _c_double_p), ("tsi", C.c_double), ("albedo", _c_double_p),
                ("spectral_boundary_weights", _c_double_p), ("temperature_fl", _c_double_p),
                ("relative_flux_dn", _c_double_p), ("relative_flux_up", _c_double_p)]


class OptConfig(C.Structure):
    _fields_ = [("flux_weight", C.c_double), ("flux_profile_weight", C.c_double), ("broadband_weight", C.c_double),
                ("spectral_boundary_weight", C.c_double), ("negative_od_penalty", C.c_double),
                ("pressure_weight_power", C.c_double), ("prior_error", C.c_double), ("min_prior_error", C.c_double),
                ("max_prior_error", C.c_double), ("prior_error_scaling", C.c_double), ("pressure_corr", C.c_double),
                ("temperature_corr", C.c_double), ("conc_corr", C.c_double), ("cap_relative_linear", C.c_double)]


TRACE_FN = C.CFUNCTYPE(None, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p)
ERROR_FN = C.CFUNCTYPE(C.c_int, C.c_int, _c_double_p, _c_double_p, _c_double_p, C.c_void_p)
EVALUATOR_FN = C.CFUNCTYPE(C.c_int, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)   # ecckd_evaluator_fn
PROGRESS_FN = C.CFUNCTYPE(None, C.c_int, C.c_double, C.c_double, C.c_void_p)                 # ecckd_progress_fn
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p)   # ecckd_allreduce_fn


def load_library():
    """Load the in-tree libecckd_hip.so; fails loudly if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # PyTorch-ROCm ships its own libamdhip64; it must be in the process BEFORE this library is
    # loaded so that both resolve to ONE HIP runtime (two runtimes in one process: the second
    # sees no device).  Importing torch does not initialise the GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    _LIB = lib
    return lib


def check(rc):
    if rc != OK:
        msg = load_library().ecckd_last_error()
        raise EcckdError(rc, msg.decode("utf-8", "replace") if msg else "")
