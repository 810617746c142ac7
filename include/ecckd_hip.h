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

typedef struct ecckd_ctx ecckd_ctx;

/* ---- context, memory, stream ------------------------------------------------ */
int ecckd_version(void);
const char* ecckd_last_error(void);
int ecckd_init(int device, ecckd_ctx** ctx);
int ecckd_destroy(ecckd_ctx* ctx);
int ecckd_synchronize(ecckd_ctx* ctx);
/* the context's hipStream_t, for callers that record their own events */
void* ecckd_stream(ecckd_ctx* ctx);
int ecckd_dev_alloc(ecckd_ctx* ctx, size_t bytes, void** d_ptr);
int ecckd_dev_free(ecckd_ctx* ctx, void* d_ptr);
int ecckd_h2d(ecckd_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int ecckd_d2h(ecckd_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
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

#ifdef __cplusplus
}
#endif
#endif
