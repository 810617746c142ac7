/* ecckd_hip.h - C ABI of libecckd_hip.so, the MI355X (gfx950) implementation of
 * ecCKD's spectral-integration hot path.
 *
 * The reference (ecmwf-ifs/ecckd 1.6) has no plugin/FFI layer: its hot path is a
 * set of C++ free functions and one virtual callback called from the four
 * executables' main().  Each entry point below replaces one of those in-process
 * seams (cited as reference file:line); INTEGRATION.md shows the call a
 * maintainer would substitute at each site.
 *
 * Conventions
 *  - Plain C types only.  All 2-D arrays are row-major (level, wavenumber) with
 *    wavenumber fastest, exactly as the reference's adept::Matrix holds them.
 *  - Pointers named d_* are DEVICE pointers (from ecckd_dev_alloc, hipMalloc or
 *    any allocator on the same device); h_* are host pointers.  Entry points
 *    without a _dev suffix take host pointers and stage through the context's
 *    stream.
 *  - Every function returns 0 on success or one of the reference's exit codes
 *    (src/include/EsaExitCodes.h:16-51): ECCKD_PARAMETER_ERROR (147),
 *    ECCKD_PROCESSING_ERROR (148), ECCKD_OUT_OF_MEMORY (130); HIP failures map
 *    to ECCKD_UNEXPECTED_EXCEPTION (131).  Nothing throws across the boundary.
 *    ecckd_last_error() returns a message for the calling thread.
 *  - A context owns one HIP stream; calls on one context are single-caller
 *    (the reference calls calc_error from OpenMP threads, equipartition.h:100;
 *    here the batch IS the parallelism).  Work is asynchronous on the stream
 *    unless the function returns host-visible results.
 *  - There is no CPU fallback: if no gfx950 device is usable the call fails.
 */
#ifndef ECCKD_HIP_H
#define ECCKD_HIP_H 1

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ECCKD_OK 0
#define ECCKD_OUT_OF_MEMORY 130
#define ECCKD_UNEXPECTED_EXCEPTION 131
#define ECCKD_PARAMETER_ERROR 147
#define ECCKD_PROCESSING_ERROR 148

/* element type of an optical-depth matrix on the device */
#define ECCKD_F32 4 /* FLOAT as stored in the CKDMIP files */
#define ECCKD_F64 8 /* the reference's in-memory Real */

/* averaging methods, find_g_points.cpp:1119-1150 / :54-204 */
#define ECCKD_AVG_LINEAR 0
#define ECCKD_AVG_TRANSMISSION 1
#define ECCKD_AVG_TRANSMISSION_2 2
#define ECCKD_AVG_SQUARE_ROOT 3
#define ECCKD_AVG_LOGARITHMIC 4
#define ECCKD_AVG_TOTAL_TRANSMISSION 5
/* create_look_up_table only, average_optical_depth.cpp:60-126 */
#define ECCKD_AVG_TRANSMISSION_3 6
#define ECCKD_AVG_TRANSMISSION_10 7
#define ECCKD_AVG_HYBRID_LOG_TRANSMISSION_3 8

typedef struct ecckd_ctx ecckd_ctx;

/* ---- context, memory, stream ------------------------------------------------ */
int ecckd_version(void);
const char* ecckd_last_error(void);
int ecckd_init(int device, ecckd_ctx** ctx);
int ecckd_destroy(ecckd_ctx* ctx);
int ecckd_synchronize(ecckd_ctx* ctx);
/* Device buffers released by gas / g-point-map handles are parked in the context and reused for later
 * handles of the same shapes (up to ECCKD_CACHE_GB, default 96); this returns them to the driver. */
int ecckd_trim_cache(ecckd_ctx* ctx);
/* the context's hipStream_t, for callers that record their own events */
void* ecckd_stream(ecckd_ctx* ctx);
int ecckd_dev_alloc(ecckd_ctx* ctx, size_t bytes, void** d_ptr);
int ecckd_dev_free(ecckd_ctx* ctx, void* d_ptr);
/* free and total device memory in bytes as the driver reports them (what another process on the same GPU holds counts as used) */
int ecckd_mem_info(ecckd_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
int ecckd_h2d(ecckd_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int ecckd_d2h(ecckd_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
/* per-kernel timing of the dominant kernels with HIP events on the context's stream:
 * kernel = "k_rt_lw_bb" | "k_reorder_key_lw"; units = wavenumber points processed.
 * Enabling resets the counters.  on = 1: every launch is timed; on = N > 1: every N-th launch of the sweep kernel (the
 * two event records and the event query cost several microseconds per batch of a search: a sample keeps them out of the
 * way); "k_rt_lw_bb.all" then gives the number of launches and points of ALL launches (ms = 0). */
int ecckd_profile_enable(ecckd_ctx* ctx, int on);
int ecckd_profile_get(ecckd_ctx* ctx, const char* kernel, long long* calls, double* ms, double* units);
/* timing on the context's stream (hipEvent pair): begin .. end -> milliseconds */
int ecckd_timer_begin(ecckd_ctx* ctx);
int ecckd_timer_end(ecckd_ctx* ctx, float* ms);

/* ---- reorder_spectrum: sorting key ------------------------------------------
 * Replaces reorder_spectrum.cpp:111-228 (planck_function, radiative_transfer_lw
 * / radiative_transfer_direct_sw, heating_rate, peak-cooling height, thin-column
 * override, threshold height) for one column.
 *   h_pressure_hl[nlay+1], h_temperature_hl[nlay+1] (LW: the idealised profile of
 *   :121-124, see ecckd_idealised_temperature), d_wavenumber/d_d_wavenumber[nwav],
 *   d_od[nlay][od_stride] of type od_type, outputs d_key[nwav], d_col_od[nwav]. */
int ecckd_idealised_temperature(int nhl, const double* h_pressure_hl, double* h_temperature_hl);

int ecckd_reorder_key_lw_dev(ecckd_ctx* ctx, int nlay, size_t nwav,
                             const double* h_pressure_hl, const double* h_temperature_hl,
                             const double* d_wavenumber, const double* d_d_wavenumber,
                             const void* d_od, int od_type, size_t od_stride,
                             double threshold_optical_depth,
                             double* d_key, double* d_col_od);

/* SW (reorder_spectrum.cpp:150-158, :197-228): key = pseudo-height where the
 * optical depth from TOA reaches the threshold.  Returns ECCKD_PROCESSING_ERROR
 * where the reference would `throw;` (key > 30, :214-216). */
int ecckd_reorder_key_sw_dev(ecckd_ctx* ctx, int nlay, size_t nwav,
                             const double* h_pressure_hl,
                             const void* d_od, int od_type, size_t od_stride,
                             double threshold_optical_depth,
                             double* d_key, double* d_col_od);

/* ---- reorder_spectrum: per-band stable sort ----------------------------------
 * Replaces reorder_spectrum.cpp:262-300 (std::stable_sort per band + rank).
 * Bands are index ranges [h_band_begin[b], h_band_end[b]] (inclusive, as
 * index(0)..index(end) at :290-293).  Points outside every band keep
 * ordered_index[j] = rank[j] = j.  Stable; -0.0 == +0.0; NaN keys sort last
 * (undefined behaviour in the reference).  d_ordered_index may be NULL. */
int ecckd_stable_argsort_bands_dev(ecckd_ctx* ctx, size_t nwav, const double* d_key,
                                   int nband, const int64_t* h_band_begin,
                                   const int64_t* h_band_end,
                                   int32_t* d_rank, int32_t* d_ordered_index);

/* Band membership, reorder_spectrum.cpp:277-289: h_iband[nwav] (-1 outside) and
 * the inclusive index range of each band (begin > end if empty).  Host-side. */
int ecckd_band_ranges(size_t nwav, const double* h_wavenumber, int nband,
                      const double* h_band_bound1, const double* h_band_bound2,
                      int16_t* h_iband, int64_t* h_band_begin, int64_t* h_band_end);

/* Host-pointer convenience wrapper of the whole reorder hot path
 * (key + sort), reorder_spectrum.cpp:111-300.  h_ssi == NULL selects LW. */
int ecckd_reorder_spectrum(ecckd_ctx* ctx, int nlay, size_t nwav,
                           const double* h_pressure_hl, const double* h_wavenumber,
                           const double* h_d_wavenumber, const void* h_od, int od_type,
                           const double* h_ssi, double threshold_optical_depth,
                           int nband, const double* h_band_bound1,
                           const double* h_band_bound2,
                           double* h_key, double* h_col_od, int16_t* h_iband,
                           int32_t* h_rank);
/* the same with the optical depths already in device memory (ecckd_nc_read_dev) */
int ecckd_reorder_spectrum_od_dev(ecckd_ctx* ctx, int nlay, size_t nwav,
                           const double* h_pressure_hl, const double* h_wavenumber,
                           const double* h_d_wavenumber, const void* d_od, int od_type,
                           const double* h_ssi, double threshold_optical_depth,
                           int nband, const double* h_band_bound1,
                           const double* h_band_bound2,
                           double* h_key, double* h_col_od, int16_t* h_iband,
                           int32_t* h_rank);

/* ---- find_g_points: gas preparation (K4) -------------------------------------
 * Replaces find_g_points.cpp:872-1150 for one gas: invert the rank from the order
 * file (:779-780), gather-reorder background and target columns (:899, :927-929),
 * Planck function on the reordered wavenumbers (:970-979), LW radiative transfer of
 * background+target (:993-995), heating rate (:1041), surface/TOA flux rows
 * (:1044-1053), layer weights (:1093-1099) and the averaging metric (:1119-1150).
 * All inputs are in ORIGINAL wavenumber order; the handle keeps the results resident
 * on the device in SORTED order.  d_bg_od may be NULL (no background_input:
 * zeros, :902-905).  d_planck_hl_reuse: the planck_hl view of a previously created
 * gas; the reference computes planck_hl only for the first gas and reuses it
 * (:966-983) - pass NULL to compute it for this gas's ordering. */
typedef struct ecckd_gas ecckd_gas;

int ecckd_gas_create_lw(ecckd_ctx* ctx, int nlay, size_t nwav,
                        const double* h_pressure_hl, const double* h_temperature_hl,
                        const double* d_wavenumber, const double* d_d_wavenumber,
                        const int32_t* d_rank,
                        const void* d_bg_od, int bg_type,
                        const void* d_od, int od_type, size_t src_stride,
                        int averaging_method, double flux_weight, double min_pressure,
                        const double* d_planck_hl_reuse, ecckd_gas** gas);

/* The Planck matrix alone, d_planck_hl[nlay+1][nwav] in the order given by d_rank: bit-identical to the "planck_hl" view
 * of a gas created with the same ordering and temperature profile.  The reference evaluates the matrix for the FIRST gas
 * and reuses it for every later one (find_g_points.cpp:529, :970-984); a process that searches only later gases (the
 * (gas, band) tasks are dealt to one process per GPU) builds it from the first gas's ordering file with this call and
 * passes it as d_planck_hl_reuse. */
int ecckd_planck_hl_sorted_dev(ecckd_ctx* ctx, int nlay, size_t nwav, const double* h_temperature_hl,
                               const double* d_wavenumber, const double* d_d_wavenumber, const int32_t* d_rank,
                               double* d_planck_hl);
/* Shortwave twin (find_g_points.cpp do_sw branches: radiative_transfer_direct_sw :1003-1006,
 * the two scaled truth fields of the total-transmission method :1008-1034, :1060-1090).
 * d_ssi and d_albedo (NULL = direct beam only) are per-wavenumber arrays in ORIGINAL order;
 * min_scaling/max_scaling are the values after the clamps of :666-667.  The band albedo used
 * by the fitted side (band_albedo(jband), :1169) is set with ecckd_gas_set_band_albedo before
 * evaluating a band. */
int ecckd_gas_create_sw(ecckd_ctx* ctx, int nlay, size_t nwav, const double* h_pressure_hl,
                        const double* d_ssi, const double* d_albedo, const int32_t* d_rank,
                        const void* d_bg_od, int bg_type, const void* d_od, int od_type,
                        size_t src_stride, int averaging_method, double flux_weight,
                        double min_pressure, double cos_sza, double min_scaling, double max_scaling,
                        ecckd_gas** gas);
int ecckd_gas_set_band_albedo(ecckd_gas* gas, double surf_albedo);
int ecckd_gas_destroy(ecckd_gas* gas);
/* device view of a resident array: "planck_hl", "bg_optical_depth", "weighted_metric",
 * "hr", "flux_dn_surf", "flux_up_toa", "wavenumber", "d_wavenumber" (LW), "ssi", "hr_low",
 * "hr_high", "flux_extras" (SW) */
int ecckd_gas_view(ecckd_gas* gas, const char* name, const double** d_ptr, size_t* rows, size_t* cols);
int ecckd_gas_layer_weight(ecckd_gas* gas, double* h_layer_weight);
/* total_comp_cost of find_g_points.cpp:320 accumulated by ecckd_calc_error_batch */
double ecckd_gas_comp_cost(ecckd_gas* gas, int reset);

/* ---- find_g_points: batched interval error (K5) ------------------------------
 * Replaces the virtual Equipartition::calc_error (equipartition.h:95) as implemented
 * by CkdEquipartition::calc_error (find_g_points.cpp:291-405) and the loop over it,
 * Equipartition::calc_error_all (equipartition.h:98-116): error[k] for the n
 * intervals [bound1[k], bound2[k]] (fractions of the band) of the band that starts
 * at sorted index ibegin and has npoints points (CkdEquipartition::init_lw, :208-235).
 * Returns ECCKD_PROCESSING_ERROR on the reference's throw(PROCESSING_ERROR) paths
 * (:298-313).  Synchronous. */
int ecckd_calc_error_batch(ecckd_gas* gas, size_t ibegin, size_t npoints, int n,
                           const double* h_bound1, const double* h_bound2, double* h_error);
/* The same with a band per interval: interval k is the fraction [h_bound1[k], h_bound2[k]] of the band that starts at
 * sorted index h_ibegin[k] and has h_npoints[k] points.  One launch train for intervals of several bands.
 * h_band_albedo[k] (shortwave; NULL = the gas's band albedo for all): the surface albedo of interval k's band. */
int ecckd_calc_error_multi(ecckd_gas* gas, int n, const size_t* h_ibegin, const size_t* h_npoints,
                           const double* h_band_albedo, const double* h_bound1, const double* h_bound2,
                           double* h_error);

/* An interval's error depends on the interval alone (its bits are the same in any batch), so the library keeps a memo per
 * gas: an interval the search has asked for before - calc_error_all re-evaluates every interval of a partition of which
 * one bound has moved (equipartition.h:98-116) - is answered from it.  The reference's work counter (ecckd_gas_comp_cost,
 * find_g_points.cpp:320) counts every request all the same.  This call reports what the memo saved: intervals asked for /
 * answered from the memo, wavenumber points asked for / actually swept on the device.  Environment ECCKD_NO_ERROR_MEMO
 * (read per call) switches the memo off. */
int ecckd_gas_eval_stats(ecckd_gas* gas, long long* requests, long long* memo_hits, double* points_requested,
                         double* points_evaluated);
/* Forget the interval errors the gas has answered so far (its memo) and zero the counters above: the next search sweeps
 * every interval again - for timing the same prepared gas twice. */
int ecckd_gas_reset_memo(ecckd_gas* gas);
/* Bytes the error sweep reads per spectral point of an interval it evaluates.  Longwave: (nlay+1) Planck values in DOUBLE
 * plus nlay background optical depths - 4 bytes each when every background value of the gas is exactly a float (a FLOAT
 * spectrum as the CKDMIP files hold it: the gas then keeps them as FLOAT pairs and the sweep widens them, the same bits as
 * reading the DOUBLE rows), 8 bytes otherwise.  Shortwave: nlay background optical depths + the solar irradiance. */
int ecckd_gas_sweep_bytes_per_point(ecckd_gas* gas, double* bytes);

/* The fitted grey optical depth alone: replaces fit_optical_depth_lw / fit_optical_depth_sw /
 * fit_optical_depth_sw_total_trans (find_g_points.cpp:54-106, :112-165, :171-204) for n
 * intervals; h_od_fit[n][nlay]. */
int ecckd_fit_optical_depth(ecckd_gas* gas, size_t ibegin, size_t npoints, int n,
                            const double* h_bound1, const double* h_bound2, double* h_od_fit);

/* ---- equal-error partition search over a batched error callback ---------------
 * Replaces class Equipartition (equipartition.h:63-208, equipartition.cpp).  The
 * callback evaluates n intervals at once and returns non-zero to abort.  status
 * receives the EpStatus value (equipartition.h:32-40). */
typedef int (*ecckd_error_fn)(int n, const double* bound1, const double* bound2, double* error, void* user);
typedef struct ecckd_partition ecckd_partition;
int ecckd_partition_create(ecckd_error_fn fn, void* user, ecckd_partition** p);
int ecckd_partition_destroy(ecckd_partition* p);
int ecckd_partition_configure(ecckd_partition* p, double resolution, double partition_tolerance,
                              int partition_max_iterations, int line_search_max_iterations,
                              int cubic_interpolation, int minimize_frac_range);
int ecckd_partition_n(ecckd_partition* p, int ni, double* bounds, double* error, int* status);
int ecckd_partition_e(ecckd_partition* p, double target_error, double bound0, double boundn,
                      int* ni, double* bounds, double* error, int capacity, int* status);
const char* ecckd_partition_status_string(int status);
/* Audit hook: every floating-point comparison that steers the search is reported, in the order it is taken, as (site, lhs,
 * rhs, outcome); the sites are numbered in csrc/partition_search.cpp.  fn = NULL switches it off.  The search itself is
 * unchanged (tests/test_partition_search.py runs with and without). */
typedef void (*ecckd_trace_fn)(int site, double lhs, double rhs, int taken, void* user);
int ecckd_partition_set_trace(ecckd_partition* p, ecckd_trace_fn fn, void* user);

/* One band of the main loop, find_g_points.cpp:1152-1266 (no sub-bands / base split):
 * equipartition_e to the heating-rate tolerance, restart with equipartition_n from
 * bounds sqrt(i/ng) if ng is outside [min_g_points, max_g_points] (:1232-1257). */
int ecckd_find_g_band(ecckd_gas* gas, size_t ibegin, size_t iend, double heating_rate_tolerance,
                      double tolerance_tolerance, int max_iterations, int min_g_points,
                      int max_g_points, int* ng, double* bounds, double* error, int capacity,
                      int* status, double* comp_cost);

/* ---- read_spectrum / read_merged_spectrum arithmetic (a1) ------------------------
 * File access stays with the caller; these are the nwav-sized operations behind it.
 * ecckd_derive_d_wavenumber_dev: read_spectrum.cpp:55-65, the spacing of a grid stored without
 *   "d_wavenumber" (interior: half the distance between the neighbours; ends: half their neighbour's).
 * ecckd_merge_scaling: the scaling of one constituent, read_merged_spectrum.cpp:117-147 - a requested
 *   concentration profile (nconc > 0: interp in pressure, ends clamped, divided by the file's own mole
 *   fraction) or the scalar rules (conc == 0 -> 0; conc > 0 -> conc / reference_surface_vmr, PARAMETER_ERROR
 *   if the file has none; scaling < 0 -> 1).  h_vmr_fl_out[nlay] is the row of vmr_fl the reference
 *   stores (:153-165); may be NULL.
 * ecckd_merge_spectrum_dev: merged (+)= od * scaling(level), :152-166; `first` overwrites instead of
 *   accumulating.  od is FLOAT (as in the CKDMIP files) or DOUBLE, merged is DOUBLE [nlay][merged_stride]. */
int ecckd_derive_d_wavenumber_dev(ecckd_ctx* ctx, size_t nwav, const double* d_wavenumber, double* d_d_wavenumber);
int ecckd_merge_scaling(int nlay, const double* h_pressure_hl, double scaling, double conc, double reference_surface_vmr,
                        const double* h_vmr_fl_one_gas, int nconc, const double* h_pressure_conc,
                        const double* h_conc_req, double* h_scaling_profile, double* h_vmr_fl_out);
int ecckd_merge_spectrum_dev(ecckd_ctx* ctx, int nlay, size_t nwav, const void* d_od, int od_type, size_t od_stride,
                             const double* h_scaling_profile, int first, double* d_merged, size_t merged_stride);

/* ---- sub-bands and base split of find_g_points --------------------------------
 * find_g_points.cpp:786-870 (sub-bands of the optically thin part of a band) and :1311-1346
 * (wavenumber split of the base g point) both re-rank a contiguous range of ranks
 * [rank_lo, rank_hi] so that the points are grouped by wavenumber interval
 * [h_wn_bound[s], h_wn_bound[s+1]) while keeping their rank order inside each group (a stable
 * partition).  d_rank[nwav] (rank of every wavenumber, original order) is updated in place;
 * h_count[nsub] receives the group sizes.  Returns ECCKD_PARAMETER_ERROR ("Failed to account
 * for all wavenumbers in split", :855-858, :1335-1338) if a point of the range lies in no group. */
int ecckd_regroup_rank_by_wavenumber_dev(ecckd_ctx* ctx, size_t nwav, const double* d_wavenumber,
                                         int32_t* d_rank, size_t rank_lo, size_t rank_hi, int nsub,
                                         const double* h_wn_bound, int64_t* h_count);

/* Sub-band set-up of one band, find_g_points.cpp:799-868.  [ibegin, iend] is the rank range of
 * the band (ibandloc(0), ibandloc(end)); h_boundary[nboundary] the configured
 * subband_wavenumber_boundary.  Outputs: *nsubband (0 if the band is not split), the first /
 * last rank of every sub-band (capacity nboundary + 1 each) and *iupperindex (:812).
 * d_rank is re-ranked in place; create the gas from it afterwards. */
int ecckd_subband_setup_dev(ecckd_ctx* ctx, size_t nwav, const double* d_wavenumber, int32_t* d_rank,
                            size_t ibegin, size_t iend, double g_split, double band_bound1,
                            double band_bound2, int nboundary, const double* h_boundary,
                            int* nsubband, int64_t* h_isubband1, int64_t* h_isubband2,
                            int64_t* iupperindex);

/* Everything find_g_points does for one band after the gas is prepared (:1152-1414): the
 * partition search (plain, :1231-1258, or per sub-band, :1186-1229), the base split
 * (:1265-1383, incl. the re-ranking of the base g point by wavenumber) and the rank range of
 * every g point (:1396-1401).  Zero-initialise the struct for the plain case. */
typedef struct {
  int min_g_points, max_g_points;
  /* sub-bands, from ecckd_subband_setup_dev; nsubband <= 1: none */
  int nsubband;
  const int64_t* isubband1;
  const int64_t* isubband2;
  int64_t iupperindex;
  double g_split;
  /* base split: base_split == 1 and nbase_wn_bound < 3: none (:1268-1271) */
  double base_split;
  int nbase_wn_bound;             /* nwavsplit + 1 entries, or 0 */
  const double* base_wn_bound;    /* band_bound1, interior boundaries, band_bound2 + 1 (:1294-1301) */
  const double* d_wavenumber;     /* [nwav] original order; needed if nwavsplit > 1 */
  int32_t* d_rank;                /* [nwav] original order, re-ranked in place if nwavsplit > 1 */
  size_t nwav;
  double band_albedo;             /* ecckd_find_g_bands_ex on a shortwave gas: this band's surface albedo (init_sw's band_albedo(jband),
                                   * find_g_points.cpp:1177); the one-band call uses ecckd_gas_set_band_albedo instead */
} ecckd_band_options;
int ecckd_find_g_band_ex(ecckd_gas* gas, size_t ibegin, size_t iend, double heating_rate_tolerance,
                         double tolerance_tolerance, int max_iterations,
                         const ecckd_band_options* opt, int* ng, double* bounds, double* error,
                         int64_t* rank1, int64_t* rank2, int capacity, int* status,
                         double* comp_cost);

/* All bands of a gas at once.  The band loop of find_g_points.cpp:1152 runs its searches one after the other, but they
 * are independent: here every band's search (exactly ecckd_find_g_band_ex, same decisions) runs in its own host thread
 * and the error evaluations the searches ask for at the same time are merged into ONE batch (ecckd_calc_error_multi),
 * so that narrow bands, too small to fill the GPU on their own, share it: the 13 longwave bands of the ecRad structure
 * go from 3.0e9 to 3.65e9 wavenumber-points/s (one band of the same size on its own: 5.4e9; the tail of the searches,
 * when only the widest bands are still refining, stays latency-bound).  Shortwave gases too: every band brings its surface
 * albedo in opt[b].band_albedo and the sweep takes it per interval (the gas's own band albedo, ecckd_gas_set_band_albedo,
 * is neither used nor changed).  Results per band as ecckd_find_g_band_ex, arrays [nband] or [nband][capacity(+1)].
 * An interval's error has the same bits alone and in any batch (every interval is summed in chunks of ITS OWN size), so a
 * search side by side ends where it ends on its own, converged or not.
 * Measured: 13 longwave bands 3.0e9 -> 3.65e9 points/s, 32 shortwave bands 1.7e9 -> 3.2e9. */
int ecckd_find_g_bands_ex(ecckd_gas* gas, int nband, const size_t* ibegin, const size_t* iend,
                          const double* heating_rate_tolerance, double tolerance_tolerance, int max_iterations,
                          const ecckd_band_options* opt /* [nband] */, int* ng, double* bounds, double* error,
                          int64_t* rank1, int64_t* rank2, int capacity, int* status, double* comp_cost);

/* The gas loop of find_g_points.cpp:655-1266 with the searches of SEVERAL prepared gases side by side on one device.  The
 * reference searches gas after gas; the searches are independent (each gas has its own prepared rows, the shared Planck
 * matrix is only read) and one search cannot fill the chip - most of its error batches are one or two intervals, bound by
 * launch and memory latency, each depending on the one before.  Every gas gets a host thread and a HIP stream of its own
 * (with its own pinned result slots and timing events) and runs exactly the launch trains it runs alone, so every search
 * takes the decisions it takes alone (same g points, same errors to the last bit) while one gas's short batches run in the
 * shadow of another gas's whole-partition passes.  One request per gas: the arguments of ecckd_find_g_bands_ex.
 * max_concurrent: gases searched at a time; <= 0: as many as the host has cores for (a gas with several bands runs a
 * thread per band).  The gases may belong to one context or to several.  req[k].rc = the gas's own return code. */
typedef struct {
  ecckd_gas* gas;
  int nband;
  const size_t* ibegin;                    /* [nband] */
  const size_t* iend;                      /* [nband] */
  const double* heating_rate_tolerance;    /* [nband] */
  const ecckd_band_options* opt;           /* [nband] */
  int* ng;                                 /* [nband] */
  double* bounds;                          /* [nband][capacity + 1] */
  double* error;                           /* [nband][capacity] */
  int64_t* rank1;                          /* [nband][capacity] or NULL */
  int64_t* rank2;                          /* [nband][capacity] or NULL */
  int capacity;
  int* status;                             /* [nband] */
  double* comp_cost;                       /* [nband] or NULL */
  int rc;                                  /* out */
} ecckd_gas_search;
int ecckd_find_g_gases(int ngas, ecckd_gas_search* req, double tolerance_tolerance, int max_iterations, int max_concurrent);
/* The same in pieces, for a caller that reads and prepares the NEXT gas while the gases it has prepared are being searched (the
 * tool: a gas's files stream in and its preparation runs on the context's stream while the lanes search): _begin opens a job,
 * _add starts the search of one prepared gas at once (its request and everything it points to must stay where it is until
 * _wait), _wait joins the searches, returns the first failure and closes the job.  A gas whose search re-ranks its base g
 * point by wavenumber (opt.nbase_wn_bound > 2) needs the context itself and is searched inside _add. */
typedef struct ecckd_gas_search_job ecckd_gas_search_job;
int ecckd_find_g_gases_begin(double tolerance_tolerance, int max_iterations, int max_concurrent, ecckd_gas_search_job** job);
int ecckd_find_g_gases_add(ecckd_gas_search_job* job, ecckd_gas_search* req);
int ecckd_find_g_gases_wait(ecckd_gas_search_job* job);

/* calc_median_sorting_variable (find_g_points.cpp:35-49) for n g points: the sorting variable
 * at the point where the cumulative weight (LW: surface Planck function, SW: solar irradiance,
 * :1404-1409) first reaches half of the interval's total.  d_sorting_variable_sorted[n points]
 * is in the gas's sorted order (gather it with ecckd_gather_f64_dev).  The cumulative sum is a
 * blocked scan, so the index can differ from the reference's sequential sum only where the
 * cumulative weight is within ~1e-13 (relative) of one half. */
int ecckd_gas_median_sorting_variable(ecckd_gas* gas, const double* d_sorting_variable_sorted, int n,
                                      const int64_t* h_ind1, const int64_t* h_ind2, double* h_median);

/* d_dst[i] = d_src[d_index[i]], i < n (the reordering gathers of find_g_points.cpp:781,:865). */
int ecckd_gather_f64_dev(ecckd_ctx* ctx, size_t n, const double* d_src, const int32_t* d_index,
                         double* d_dst);
/* d_inverse[d_perm[i]] = i (ireorder(irank) = range(0,n-1), find_g_points.cpp:778-779). */
int ecckd_invert_permutation_dev(ecckd_ctx* ctx, size_t n, const int32_t* d_perm, int32_t* d_inverse);

/* ---- line-by-line band fluxes --------------------------------------
 * A stand-in for the external CKDMIP tool that makes the training fluxes (test/run_lw_lbl_evaluation.sh),
 * restricted to the no-scattering radiative transfer the reference itself contains.  One column; the
 * spectral fluxes are summed over the inclusive wavenumber index ranges [h_band_begin[b], h_band_end[b]]
 * (ecckd_band_ranges) -> h_flux_*[nband][nlay+1], i.e. "band_flux_dn_lw" / "band_flux_up_lw" of a flux file.
 *   longwave:  planck_function (planck_function.cpp:22-54) + radiative_transfer_lw (radiative_transfer_lw.cpp:27-60),
 *              unit surface emissivity, surface Planck function at temperature_hl(end);
 *   shortwave: radiative_transfer_direct_sw, and with d_albedo[nwav] != NULL radiative_transfer_norayleigh_sw
 *              (radiative_transfer_sw.cpp:26-77); h_flux_up may be NULL.  No Rayleigh scattering. */
int ecckd_lbl_band_fluxes_lw(ecckd_ctx* ctx, int nlay, size_t nwav, const double* h_temperature_hl,
                             const double* d_wavenumber, const double* d_d_wavenumber, const void* d_od, int od_type,
                             size_t od_stride, int nband, const int64_t* h_band_begin, const int64_t* h_band_end,
                             double* h_flux_dn, double* h_flux_up);
int ecckd_lbl_band_fluxes_sw(ecckd_ctx* ctx, int nlay, size_t nwav, double cos_sza, const double* d_ssi,
                             const double* d_albedo, const void* d_od, int od_type, size_t od_stride, int nband,
                             const int64_t* h_band_begin, const int64_t* h_band_end, double* h_flux_dn_direct,
                             double* h_flux_up);
/* The same with the spectral fluxes at the boundaries as well (the CKDMIP tool's do_write_spectral_boundary_fluxes: what
 * LblFluxes::read maps to g points, lbl_fluxes.cpp:183-246, :301-325): d_surf_dn[nwav] = downwelling (shortwave: direct) flux at
 * the surface, d_toa_up[nwav] = upwelling flux at the top of the atmosphere, per wavenumber, device arrays, either may be NULL;
 * zero outside the bands. */
int ecckd_lbl_band_fluxes_lw_ex(ecckd_ctx* ctx, int nlay, size_t nwav, const double* h_temperature_hl,
                                const double* d_wavenumber, const double* d_d_wavenumber, const void* d_od, int od_type,
                                size_t od_stride, int nband, const int64_t* h_band_begin, const int64_t* h_band_end,
                                double* h_flux_dn, double* h_flux_up, double* d_surf_dn, double* d_toa_up);
/* The same with the zenith-angle quadrature of the external CKDMIP tool (namelist key nangle, test/run_ckd_lw.sh:28,
 * test/copy_to_ckdmip_lw.sh:32): nangle = 0 is the classic two-stream form above (diffusivity 1.66); nangle = N > 0 integrates
 * N Gauss-Legendre angles per hemisphere, flux = sum_k 2 w_k mu_k L(mu_k), each L(mu_k) the no-scattering recurrence of
 * radiative_transfer_lw.cpp:27-60 along the slant path tau / mu_k.  N <= 16.  The CKDMIP tool is not part of the reference's
 * sources: its node set is unpinned. */
int ecckd_lbl_band_fluxes_lw_angles(ecckd_ctx* ctx, int nangle, int nlay, size_t nwav, const double* h_temperature_hl,
                                    const double* d_wavenumber, const double* d_d_wavenumber, const void* d_od, int od_type,
                                    size_t od_stride, int nband, const int64_t* h_band_begin, const int64_t* h_band_end,
                                    double* h_flux_dn, double* h_flux_up, double* d_surf_dn, double* d_toa_up);
/* Gauss-Legendre nodes mu_k (ascending) and weights w_k of n points on (0, 1): sum_k w_k f(mu_k) ~ int_0^1 f(mu) dmu. */
int ecckd_gauss_legendre_01(int n, double* h_mu, double* h_weight);
/* Fluxes per g point from what run_ckd wrote (optical depth [ncol][nlay][ng], Planck function [ncol][nlay+1][ng] or incoming
 * solar flux [ncol][ng]): the `--ckd` evaluation of the CKDMIP tools as the scripts use it (test/run_ckd_lw.sh:133-137,
 * test/run_ckd_sw.sh:125-128), restricted to the reference's own no-scattering transfer - radiative_transfer_lw.cpp:27-60 with
 * unit emissivity (nangle as ecckd_lbl_band_fluxes_lw_angles) and radiative_transfer_sw.cpp:45-77.  Host arrays in and out
 * ([ncol][nlay+1][ng] fluxes), the arithmetic on the device. */
int ecckd_rt_lw_gpoints(ecckd_ctx* ctx, int nangle, int ncol, int nlay, int ng, const double* h_planck_hl, const double* h_od,
                        double* h_flux_dn, double* h_flux_up);
int ecckd_rt_sw_gpoints(ecckd_ctx* ctx, int ncol, int nlay, int ng, double cos_sza, double albedo, const double* h_incoming,
                        const double* h_od, double* h_flux_dn, double* h_flux_up);

int ecckd_lbl_band_fluxes_sw_ex(ecckd_ctx* ctx, int nlay, size_t nwav, double cos_sza, const double* d_ssi,
                                const double* d_albedo, const void* d_od, int od_type, size_t od_stride, int nband,
                                const int64_t* h_band_begin, const int64_t* h_band_end, double* h_flux_dn_direct,
                                double* h_flux_up, double* d_surf_dn_direct, double* d_toa_up);

/* ---- NetCDF classic files (file parts of a1, a9, a21) ----------------------------
 * A self-contained reader / writer for the classic on-disk formats CDF-1, CDF-2 (64-bit offset)
 * and CDF-5 (64-bit data): what the reference reads / writes for *.nc, *.cdf names
 * (src/tools/DataFile.cpp:88-96, OutputDataFile.cpp:84-157).  NetCDF-4 files (HDF5 containers, the *.h5
 * names of the scripts) are recognised by their signature and READ through the system's HDF5 library,
 * loaded at run time (csrc/nc_hdf5.cpp; ECCKD_HDF5_LIB overrides the search); files are always WRITTEN in
 * the classic format, which the NetCDF library reads back whatever the file is called.  Reads convert every external type to double like
 * nc_get_vara_double (DataFileEngineNetcdf.cpp:593-599); slice >= 0 selects one index of the
 * slowest dimension like DataFile::read(M, "v", j) (:582-590), slice < 0 the whole variable.
 * var == NULL or "" addresses the global attributes.  nc_type: 1 byte, 2 char, 3 short, 4 int,
 * 5 float, 6 double (7-11: CDF-5 unsigned / 64-bit integers). */
typedef struct ecckd_nc ecckd_nc;
int ecckd_nc_open(const char* path, ecckd_nc** file);
int ecckd_nc_close(ecckd_nc* file);
int ecckd_nc_inq_dim(ecckd_nc* file, const char* name, size_t* length);
int ecckd_nc_inq_var(ecckd_nc* file, const char* name, int* exists, int* nc_type, int* ndims, size_t* shape,
                     int shape_capacity);
int ecckd_nc_read_double(ecckd_nc* file, const char* name, long long slice, double* out, size_t capacity);
/* One index of the slowest dimension (slice >= 0) or the whole variable straight into device memory as FLOAT (out_type 4)
 * or DOUBLE (8): replaces DataFile::read -> nc_get_vara_double -> element-wise copy (DataFileEngineNetcdf.cpp:593-608) for
 * the spectra.  A contiguous FLOAT / DOUBLE variable of a classic file is streamed - reader threads fill pinned buffers,
 * a copy stream ships them, a kernel decodes the big-endian values on the device, reading overlaps shipping; anything
 * else (NetCDF-4, record variables, integer types) is read on the host and uploaded once.  capacity in elements. */
int ecckd_nc_read_dev(ecckd_ctx* ctx, ecckd_nc* file, const char* name, long long slice, int out_type, void* d_out,
                      size_t capacity);
/* zlib streams (RFC 1950 / 1951: what HDF5's deflate filter leaves in a chunk) inflated on the device, one wavefront per
 * stream - the NetCDF-4 spectra are read this way by ecckd_nc_read_dev (the reference leaves it to the HDF5 library on the
 * reading thread, DataFileEngineNetcdf.cpp:593-608).  Host-pointer form: stream s is h_in[h_in_off[s] .. h_in_off[s + 1]) and
 * must inflate to exactly h_out_bytes[s] bytes, written one stream after the other to h_out; h_status[s] = 0, or 1 bad
 * header, 2 bad block, 3 bad code, 4 distance too far back, 5 more output than expected, 6 less, 7 input exhausted.
 * The Adler-32 trailer is not verified. */
int ecckd_inflate(ecckd_ctx* ctx, int nstreams, const void* h_in, const unsigned long long* h_in_off,
                  const unsigned long long* h_out_bytes, void* h_out, int* h_status);
/* One zlib stream inflated on the calling thread by the in-tree decoder the worker threads of ecckd_nc_read_dev use before
 * they fall back to zlib (csrc/fast_inflate.cpp; no device involved).  *ok = 1: the stream inflates to exactly out_bytes
 * bytes with the right Adler-32 and out holds them; 0: not taken or a check failed. */
int ecckd_inflate_host(const void* in, size_t in_bytes, void* out, size_t out_bytes, int* ok);
int ecckd_nc_read_att_text(ecckd_nc* file, const char* var, const char* att, int* exists, char* out, size_t capacity);
int ecckd_nc_read_att_double(ecckd_nc* file, const char* var, const char* att, int* nelems, double* out, size_t capacity);
/* writing: define, ecckd_nc_enddef (picks CDF-1 / CDF-2 / CDF-5 from the sizes), then whole variables */
int ecckd_nc_create(const char* path, ecckd_nc** file);
int ecckd_nc_def_dim(ecckd_nc* file, const char* name, size_t length, int* dimid);
int ecckd_nc_def_var(ecckd_nc* file, const char* name, int nc_type, int ndims, const int* dimids, int* varid);
int ecckd_nc_put_att_text(ecckd_nc* file, const char* var, const char* att, const char* text);
int ecckd_nc_put_att_double(ecckd_nc* file, const char* var, const char* att, int nc_type, int n, const double* values);
/* deflate_variable (OutputDataFile.cpp:345-359): shuffle + deflate level 2 for this variable when the file is written as
 * NetCDF-4 (a name ending in .h5 / .hdf where the HDF5 libraries can be loaded; ECCKD_CLASSIC_OUTPUT forces classic), nothing
 * otherwise.  ecckd_nc_is_netcdf4: which of the two an open handle is. */
int ecckd_nc_deflate_var(ecckd_nc* file, const char* name);
int ecckd_nc_is_netcdf4(ecckd_nc* file, int* is_netcdf4);
int ecckd_nc_enddef(ecckd_nc* file);
int ecckd_nc_write_double(ecckd_nc* file, const char* name, const double* data, size_t count);
/* one index of the slowest dimension of a fixed-size variable (count = the elements below that dimension) */
int ecckd_nc_write_slice_double(ecckd_nc* file, const char* name, size_t slice, const double* data, size_t count);
/* write_order (write_order.cpp:24-143): same variables, external types and attributes; `history` is the
 * line OutputDataFile::append_history would add (may be NULL); column_optical_depth may be NULL (:88). */
int ecckd_write_order_file(const char* path, const char* molecule, const char* config_str, const char* history, int nband,
                           const double* band_bound1, const double* band_bound2, size_t nwav, const double* wavenumber,
                           const double* d_wavenumber, const int16_t* iband, const int32_t* rank,
                           const double* column_optical_depth, const double* sorting_variable);

/* ---- configuration: `exe [key=value ...] [file.cfg]` (drop-in surface) ------
 * Host-only.  Replaces DataFile(argc, argv) -> DataFileEngineCfg (src/tools/DataFileEngineCfg.cpp:61-80) and
 * the rc_* functions of src/tools/readconfig.c it calls, one entry point per function used:
 *   ecckd_cfg_from_args     the constructor: rc_read(NULL) + rc_register_files + rc_get_file (first argument
 *                           without '=' whose name contains ".cfg") + rc_append + rc_register_args
 *   ecckd_cfg_append_file   rc_append (readconfig.c:557-880; grammar in csrc/config.cpp)
 *   ecckd_cfg_register      rc_register (:885-896); value NULL -> "1"
 *   ecckd_cfg_exists        rc_exists;   ecckd_cfg_get_boolean  rc_get_boolean (:1262-1287)
 *   ecckd_cfg_get_int/_real rc_assign_int / rc_assign_real (:1293-1407): *value untouched and *found = 0 when absent
 *   ecckd_cfg_get_string    rc_get_string (isub < 0) / rc_get_substring (isub >= 0) (:1481-1500, :1622-1633)
 *   ecckd_cfg_size          rc_size (:1653-1666): number of items and the declared [m][n]
 *   ecckd_cfg_get_real_vector / _int_vector   rc_get_real_vector / rc_get_int_vector (:1720-1790)
 *   ecckd_cfg_sprint        rc_sprint (:1113-1253): the string stored in the `config` attribute of every output file
 * `scope` (may be NULL) is the section of DataFile::read(x, scope, name): rc_set_section(scope) around the call.
 * Strings are copied into buf[cap] (always terminated) and *len receives the full length.
 * A file that cannot be opened or parsed returns ECCKD_CANNOT_OPEN_MANDATORY_FILE like open_absolute (:32-52). */
#define ECCKD_CANNOT_OPEN_MANDATORY_FILE 139
typedef struct ecckd_cfg ecckd_cfg;
int ecckd_cfg_create(ecckd_cfg** cfg);
int ecckd_cfg_from_args(int argc, const char* const* argv, ecckd_cfg** cfg);
int ecckd_cfg_append_file(ecckd_cfg* cfg, const char* path);
int ecckd_cfg_append_text(ecckd_cfg* cfg, const char* text, const char* name);
int ecckd_cfg_register(ecckd_cfg* cfg, const char* param, const char* value);
int ecckd_cfg_destroy(ecckd_cfg* cfg);
int ecckd_cfg_file_name(const ecckd_cfg* cfg, char* buf, size_t cap, size_t* len);
int ecckd_cfg_count(const ecckd_cfg* cfg, int* n);
int ecckd_cfg_entry(const ecckd_cfg* cfg, int i, char* param, size_t param_cap, char* value, size_t value_cap,
                    size_t* value_len, int* has_value, int* m, int* n);
int ecckd_cfg_exists(const ecckd_cfg* cfg, const char* scope, const char* param, int* exists);
int ecckd_cfg_get_boolean(const ecckd_cfg* cfg, const char* scope, const char* param, int* value);
int ecckd_cfg_get_int(const ecckd_cfg* cfg, const char* scope, const char* param, int* value, int* found);
int ecckd_cfg_get_real(const ecckd_cfg* cfg, const char* scope, const char* param, double* value, int* found);
int ecckd_cfg_get_string(const ecckd_cfg* cfg, const char* scope, const char* param, int isub, char* buf, size_t cap,
                         size_t* len, int* found);
int ecckd_cfg_size(const ecckd_cfg* cfg, const char* scope, const char* param, int* count, int* m, int* n);
int ecckd_cfg_get_real_vector(const ecckd_cfg* cfg, const char* scope, const char* param, double* buf, int cap, int* len);
int ecckd_cfg_get_int_vector(const ecckd_cfg* cfg, const char* scope, const char* param, int* buf, int cap, int* len);
int ecckd_cfg_sprint(const ecckd_cfg* cfg, char* buf, size_t cap, size_t* len);

/* ---- optimize_lut: cost function, gradient and minimisation (K8/K9) ------------
 * Replaces CkdOptimizable::calc_cost_function_gradient (solve_adept.cpp:240-292), i.e.
 * calc_cost_function_and_gradient (:72-211: CkdModel::calc_optical_depth ckd_model.cpp:925-1102,
 * negative-OD penalty, calc_cost_function_ckd_lw calc_cost_function_lw.cpp:116-232, Adept
 * reverse pass) plus CkdModel::calc_background_cost_function (ckd_model.cpp:840-877), and
 * solve_adept itself (:310-417).  Longwave (calc_cost_function_ckd_lw) and shortwave
 * (calc_cost_function_ckd_sw, calc_cost_function_sw.cpp:116-277), linear LUT interpolation.
 * All arrays here are HOST arrays; the handle uploads them once. */
typedef struct {
  int conc_dependence;          /* 0 none, 1 linear, 2 look-up table, 3 relative-linear (ckd_model.cpp:418-482) */
  int is_active;                /* in the gas list being optimised (optimize_lut.cpp:161) */
  int nconc;                    /* conc_dependence 2: number of mole fractions */
  const double* vmr;            /* [nconc], uniform in log (ckd_model.cpp:1006-1009) */
  double reference_vmr;         /* conc_dependence 3 */
  const double* molar_abs;      /* [(nconc)][nt][np][ng], g fastest */
  const double* min_molar_abs;  /* same shape or NULL */
  const double* max_molar_abs;
} ecckd_opt_gas;

typedef struct {
  int ng, nt, np;
  const double* log_pressure;        /* [np], evenly spaced (ckd_model.cpp:946-948) */
  const double* temperature;         /* [nt][np] */
  int ntp;
  const double* temperature_planck;  /* [ntp] */
  const double* planck_function;     /* [ntp][ng] */
  const int* iband_per_g;            /* [ng] g point -> band of the training fluxes (ckd_model.h:287-306) */
  int ngas;
  const ecckd_opt_gas* gases;
  int logarithmic_interpolation;     /* must be 0 (ckd_model.h:359) */
  /* shortwave (NULL / NULL = longwave model): solar irradiance per g point and the Rayleigh molar
   * scattering coefficient per g point (ckd_model.h:242-252; kept fixed, not optimised) */
  const double* solar_irradiance;           /* [ng] */
  const double* rayleigh_molar_scattering;  /* [ng] or NULL */
} ecckd_opt_model;

typedef struct {
  int ncol, nlay, nband;
  const double* pressure_hl;     /* [ncol][nlay+1] */
  const double* temperature_hl;  /* [ncol][nlay+1] */
  const double* vmr_fl;          /* [ncol][ngas][nlay], gases in MODEL order (LblFluxes::gas_mapping applied) */
  const int* gas_present;        /* [ngas] 0 where the training file lacks the gas (gas_mapping < 0); NULL = all */
  const double* surf_emissivity; /* [ncol][nband] or NULL = 1 (lbl_fluxes.cpp:395-396) */
  const double* flux_dn;         /* [ncol][nlay+1][nband] LBL band fluxes (spectral_flux_dn_) */
  const double* flux_up;
  const double* spectral_flux_dn_surf;  /* [ncol][ng] or NULL (lbl_fluxes.cpp:301-328) */
  const double* spectral_flux_up_toa;
  /* shortwave scenes (lbl_fluxes.cpp:68-148): columns already replicated per solar zenith angle */
  const double* mu0;                        /* [ncol] */
  double tsi;                               /* total solar irradiance of the training file (:127) */
  const double* albedo;                     /* [nband] effective spectral albedo (:147-148, mask_rayleigh_up) */
  const double* spectral_boundary_weights;  /* [ng] erythemal_weight * erythemal_spectrum_ (solve_adept.cpp:182) or NULL */
  /* optional [ncol][nlay] full-level temperature; NULL = pressure-weighted mean of temperature_hl
   * (solve_adept.cpp:38-41, run_ckd.cpp:120-122).  scale_lut uses the plain mean (scale_lut.cpp:108). */
  const double* temperature_fl;
  /* optional [ncol][nlay+1][ng]: CKD fluxes of the "relative_to" scene at the initial coefficients
   * (optimize_lut.cpp:204-236, obtained with ecckd_opt_forward on that scene), subtracted from the forward
   * model per g point (solve_adept.cpp:118-148, calc_cost_function_lw.cpp:162-165); the caller subtracts the
   * relative-to LBL fluxes from flux_dn / flux_up itself (LblFluxes::subtract, optimize_lut.cpp:251-254) */
  const double* relative_flux_dn;
  const double* relative_flux_up;
} ecckd_opt_scene;

typedef struct {
  double flux_weight, flux_profile_weight, broadband_weight, spectral_boundary_weight;
  double negative_od_penalty;    /* optimize_lut.cpp:149 default 1e4 */
  double pressure_weight_power;  /* solve_adept.cpp:131-143 default 0.5 */
  double prior_error, min_prior_error, max_prior_error, prior_error_scaling;
  double pressure_corr, temperature_corr, conc_corr;
  double cap_relative_linear;    /* optimize_lut.cpp:185 passes 0.8; 0 disables */
} ecckd_opt_config;

typedef struct ecckd_opt ecckd_opt;
int ecckd_opt_create(ecckd_ctx* ctx, const ecckd_opt_model* model, int nscene,
                     const ecckd_opt_scene* scenes, const ecckd_opt_config* config, ecckd_opt** opt);
int ecckd_opt_destroy(ecckd_opt* opt);
/* length of the state vector x = ln(molar_abs) of the active gases, in model gas order */
size_t ecckd_opt_nx(ecckd_opt* opt);
/* initial state (MIN_X = -1e20 where the coefficient is 0) and log-space bounds, solve_adept.cpp:335-353 */
int ecckd_opt_initial_state(ecckd_opt* opt, double* h_x, double* h_x_min, double* h_x_max);
int ecckd_opt_cost_grad(ecckd_opt* opt, const double* h_x, double* J, double* h_grad);
/* total optical depth [ncol][nlay][ng] and CKD fluxes [ncol][2][nlay+1][ng] at state h_x
 * (calc_total_optical_depth, LblFluxes::calc_ckd_fluxes lbl_fluxes.cpp:443-471) */
int ecckd_opt_forward(ecckd_opt* opt, const double* h_x, double* h_od, double* h_flux);
/* the same with unclamped != 0: negative total optical depths are NOT set to zero before the radiative transfer - how the
 * reference evaluates its "relative_to" scene (optimize_lut.cpp:229-234: od = value(aod) -> LblFluxes::calc_ckd_fluxes;
 * the clamp belongs to the cost function alone, solve_adept.cpp:107-116) */
int ecckd_opt_forward_ex(ecckd_opt* opt, const double* h_x, int unclamped, double* h_od, double* h_flux);
int ecckd_opt_coefficients(ecckd_opt* opt, const double* h_x, int gas, double* h_molar_abs);
/* status follows adept::MinimizerStatus: 0 success, 2 max iterations, 3 failed to converge,
 * >= 6 anomalous (optimize_lut.cpp:315-319 exits 1 for those) */
int ecckd_opt_minimize(ecckd_opt* opt, int max_iterations, double convergence_criterion, int is_bounded,
                       double* h_x, int* status, int* n_iterations, double* J_final, double* gnorm_final);

/* Profile-sharded optimisation over several GPUs (optimize_lut row, "profiles sharded"): the cost
 * function is a sum over training profiles (calc_cost_function_and_gradient accumulates it profile by profile,
 * solve_adept.cpp:157), so every rank creates its handle from ITS share of the columns of every scene and the
 * ranks only have to agree on  sum_ranks [dJ/dx (nx doubles), J]  after each evaluation.  `fn` is called once per
 * evaluation with that device buffer (count = nx + 1 doubles; the context's stream is idle when it is called)
 * and must return 0 after summing it in place over the ranks - one RCCL all-reduce (ecckd_amd/shard.py makes
 * the callback from torch.distributed).  The prior term (CkdModel::calc_background_cost_function) is not a sum over
 * profiles: exactly one rank passes add_prior = 1.  With identical reduced values on every rank the L-BFGS
 * iterations of ecckd_opt_minimize take the same decisions everywhere, so every rank returns the same state.
 * fn == NULL restores single-GPU behaviour.  ecckd_opt_forward never calls fn. */
typedef int (*ecckd_allreduce_fn)(void* d_buf, size_t count, void* stream, void* user);
int ecckd_opt_set_allreduce(ecckd_opt* opt, ecckd_allreduce_fn fn, void* user, int add_prior);

/* ecckd_opt_minimize over a cost function and gradient supplied by the caller (host arrays) instead of the library's
 * kernels: the L-BFGS iteration, its bounds and its line search stay exactly the library's.  With the reference's own
 * CkdOptimizable::calc_cost_function_gradient (solve_adept.cpp:240-292) behind fn this runs the library's minimizer over
 * the reference's cost function; the parity tests put the CPU oracle there and compare the two trajectories.
 * fn returns 0, or non-zero to abort (PROCESSING_ERROR).  fn == NULL restores the device evaluation. */
typedef int (*ecckd_evaluator_fn)(size_t nx, const double* h_x, double* J, double* h_grad, void* user);
int ecckd_opt_set_evaluator(ecckd_opt* opt, ecckd_evaluator_fn fn, void* user);

/* The minimizer's progress line and the three timed activities of the reference (solve_adept.cpp:216-218 "minimizer",
 * "a-priori", "radiative transfer"; report_progress :295-299 "Iteration n: cost function = ..., gradient norm = ...").
 * fn is called once per L-BFGS iteration of ecckd_opt_minimize, on the calling thread.  Setting it also starts the
 * activity timers (HIP events around the kernels: "radiative transfer" = look-up + sweeps + cost + their adjoint,
 * "a-priori" = the gradient kernel that carries the prior term, "minimizer" = the rest of ecckd_opt_minimize). */
typedef void (*ecckd_progress_fn)(int iteration, double cost, double gradient_norm, void* user);
int ecckd_opt_set_progress(ecckd_opt* opt, ecckd_progress_fn fn, void* user);
int ecckd_opt_timings(ecckd_opt* opt, double* minimizer_s, double* a_priori_s, double* radiative_transfer_s);

/* ---- run_ckd -------------------------------------------------------
 * Replaces the compute part of run_ckd.cpp:27-373 for the profiles of one scene (flux arrays
 * of the scene are ignored; mu0 and tsi are used for a shortwave model, run_ckd.cpp:92,:358
 * passes REFERENCE_COS_SZA).  Outputs, any of which may be NULL:
 *   h_od[ncol][nlay][ng]           sum of the gases' optical depths, clamped at 0 ("optical_depth", :318)
 *   h_rayleigh_od[ncol][nlay][ng]  shortwave only ("rayleigh_optical_depth")
 *   h_planck_hl[ncol][nlay+1][ng]  longwave "planck_hl" (row nlay = "planck_surf"); shortwave: row 0 of
 *                                  every column = "incoming_sw" = tsi / sum(ssi) * ssi
 *   h_flux[ncol][2][nlay+1][ng]    longwave: [0] = spectral_flux_dn_lw, [1] = spectral_flux_up_lw;
 *                                  shortwave: [0] = spectral_flux_dn_direct_sw, [1] = 0
 * Gases absent from the scene (gas_present[i] == 0) are skipped like run_ckd's "gases" list (:270-276);
 * call once per gas with a one-hot gas_present for the "<gas>_optical_depth" variables. */
int ecckd_run_ckd(ecckd_ctx* ctx, const ecckd_opt_model* model, const ecckd_opt_scene* scene, double* h_od,
                  double* h_rayleigh_od, double* h_planck_hl, double* h_flux);

/* ---- scale_lut -----------------------------------------------------
 * scale_lut.cpp:117-133 needs, per g point, the sum over its wavenumbers of every row of a
 * (nrows, nwav) matrix (the LBL direct-beam spectral flux at each half level): h_sums[nrows][ng].
 * Declared below, after ecckd_gmap: ecckd_gmap_sum_rows.
 *
 * ecckd_scale_lut does the rest of scale_lut.cpp:117-189 and CkdModel::scale_optical_depth
 * (ckd_model.cpp:1151-1176) for one reference profile: od_best = -mu0 log(flux_base / flux_top) per layer
 * and g point (-1 where the flux has vanished), od_total from the CKD model (NOT clamped at zero,
 * plain-mean full-level temperature), scaling = od_best / od_total (1 where od_best <= 0), interpolated in
 * log pressure to the model's grid (adept::interp restated as linear interpolation with linear
 * extrapolation) and applied to every gas, clamped to [min, max] where those exist.
 *   h_vmr_fl[ngas][nz] in model gas order; gas_present[ngas] = gas is in the LBL file's constituent list
 *   (or is "composite"); h_scaling[nz][ng] (optional output); h_molar_abs_out[ngas] = caller buffers shaped
 *   like each gas's molar_abs. */
int ecckd_scale_lut(ecckd_ctx* ctx, const ecckd_opt_model* model, int nz, const double* h_pressure_hl,
                    const double* h_temperature_hl, const double* h_vmr_fl, const int* gas_present, double mu0,
                    const double* h_flux_sums, double* h_scaling, double* const* h_molar_abs_out);

/* ---- create_look_up_table (K6/K7) ----------------------------------------------
 * A g-point map: the wavenumbers sorted by g point once (stable), so that every g point is a
 * contiguous segment.  d_g_point[nwav] as read from the g-points file (-1 = unassigned,
 * create_look_up_table.cpp:92-106); wavenumbers must ascend. */
typedef struct ecckd_gmap ecckd_gmap;
int ecckd_gmap_create(ecckd_ctx* ctx, size_t nwav, const int32_t* d_g_point, int ng,
                      const double* d_wavenumber, const double* d_d_wavenumber, ecckd_gmap** gmap);
int ecckd_gmap_destroy(ecckd_gmap* gmap);
/* wavenumbers per g point: zero = "occupies none of the spectrum" (create_look_up_table.cpp:111-118) */
int ecckd_gmap_counts(ecckd_gmap* gmap, int64_t* h_counts);

/* Replaces average_optical_depth_to_g_point (average_optical_depth.cpp:22-197) for one column:
 * weights are the Planck function at h_temperature_fl[nlay] (longwave,
 * create_look_up_table.cpp:316-327) or d_ssi[nwav] (shortwave, :335); exactly one of the two
 * must be given.  Outputs (nlay, ng) row-major; min/max may be NULL.  reference_surface_vmr <= 0
 * returns optical depths instead of molar absorption (:186-193). */
int ecckd_average_to_gpoints(ecckd_gmap* gmap, int nlay, const double* h_pressure_hl,
                             const double* h_temperature_fl, const double* d_ssi,
                             const void* d_od, int od_type, size_t od_stride, int averaging_method,
                             double reference_surface_vmr, double* h_molar_abs,
                             double* h_min_molar_abs, double* h_max_molar_abs);
/* h_sums[nrows][ng] = sum over the wavenumbers of each g point of d_rows[r][.] (scale_lut.cpp:119-124) */
int ecckd_gmap_sum_rows(ecckd_gmap* gmap, int nrows, const void* d_rows, int rows_type, size_t row_stride,
                        double* h_sums);
/* LblFluxes::read, lbl_fluxes.cpp:198-230: the square root of the erythemal action spectrum (Webb et al.
 * 2011) averaged over each g point with a 5777 K Planck weight, h_erythemal[ng]; NaN for an empty g point
 * (0/0 as in the reference). */
int ecckd_gmap_erythemal_spectrum(ecckd_gmap* gmap, double* h_erythemal);
/* create_look_up_table.cpp:537-548: h_gpoint_fraction[ng][nint] over the coarse intervals
 * (wavenumber1, wavenumber2] */
int ecckd_gpoint_fraction(ecckd_gmap* gmap, int nint, const double* h_wavenumber1,
                          const double* h_wavenumber2, double* h_gpoint_fraction);
/* create_look_up_table.cpp:581-591: h_planck_lut[nlut][ng] */
int ecckd_planck_lut(ecckd_gmap* gmap, int nlut, const double* h_temperature_lut, double* h_planck_lut);

/* ---- find_g_points: spectral overlap of the gases (a14) ----------------------
 * ecckd_overlap_g_points replaces overlap_g_points (single_gas_data.cpp:24-124), host-side:
 *   h_n_g_points[ngas][nband]; h_sorting_variable = the gases' per-g-point median sorting
 *   variables concatenated, gas i starting at h_gas_offset[i]; outputs *h_ng,
 *   h_band_number[capacity], h_g_min/h_g_max[ngas][capacity].
 * ecckd_gas_g_point_dev replaces SingleGasData::store_g_points (single_gas_data.h:56-62);
 * ecckd_merge_g_points_dev replaces find_g_points.cpp:1459-1475 (h_d_gas_g_point: host array
 * of ngas device pointers; g_min/g_max rows have `stride` elements). */
int ecckd_overlap_g_points(int ngas, int nband, const int* h_n_g_points, const int* h_gas_offset,
                           const double* h_sorting_variable, int capacity, int* h_ng,
                           int* h_band_number, int* h_g_min, int* h_g_max);
int ecckd_gas_g_point_dev(ecckd_ctx* ctx, size_t nwav, const int32_t* d_rank, int ng,
                          const int32_t* h_rank1, const int32_t* h_rank2, int32_t* d_g_point);
int ecckd_merge_g_points_dev(ecckd_ctx* ctx, size_t nwav, int ngas,
                             const int32_t* const* h_d_gas_g_point, int ng, int stride,
                             const int* h_g_min, const int* h_g_max, int32_t* d_g_point,
                             int64_t* h_n_unassigned);

#ifdef __cplusplus
}
#endif
#endif
