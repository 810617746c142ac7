/* ecckd_oracle.h - CPU restatement of the ecCKD spectral-integration hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check
 * in __graft_entry__.py and the cpu_baseline leg of bench.py may load it; the
 * product library (libecckd_hip.so) never links, loads or calls anything here.
 *
 * Every function restates one function of the reference (ecmwf-ifs/ecckd 1.6)
 * in plain C with the same operation order (layers outer, wavenumber inner,
 * sums in index order); the reference file:line it follows is cited at each
 * definition.  OpenMP is used at exactly the reference's sites.
 *
 * Parity pin status: the reference ships no golden vectors or known-answer
 * tests for the radiative-transfer / reorder / cost-function arithmetic, and
 * the reference cannot be built here (needs Adept >= 2.1 and NetCDF, absent).
 * => "parity unpinned" for those rows: the restatement is checked against
 * closed-form cases (tests/test_oracle_*.py).  The partition search (a13) IS
 * pinned: oracle/_ref compiles the reference's own equipartition.cpp and the
 * known-answer generator of test_equipartition.cpp is committed under
 * tests/golden/.
 *
 * All 2-D arrays are row-major (level, wavenumber), wavenumber fastest,
 * exactly as the reference holds them.
 */
#ifndef ECCKD_ORACLE_H
#define ECCKD_ORACLE_H 1

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* constants.h:22-26 */
#define ORC_ACCEL_GRAVITY 9.80665
#define ORC_SPECIFIC_HEAT_AIR 1004.0
#define ORC_LW_DIFFUSIVITY 1.66
#define ORC_MOLAR_MASS_DRY_AIR 28.970
/* calc_cost_function_sw.h:20 */
#define ORC_REFERENCE_COS_SZA 0.5

/* ---- a2: planck_function.cpp:22-54 ---- */
void orc_planck_function(int nt, const double* temperature, size_t nwav,
                         const double* wavenumber_cm_1,
                         const double* d_wavenumber_cm_1,
                         double* planck /* [nt][nwav] */);

/* ---- a3: radiative_transfer_lw.cpp:27-60 ---- */
void orc_radiative_transfer_lw(int nlay, size_t nwav, const double* planck,
                               const double* od, const double* surf_emissivity,
                               const double* surf_planck, double* flux_dn,
                               double* flux_up);

/* ---- a4: radiative_transfer_lw.cpp:87-142 ---- */
void orc_radiative_transfer_lw_bb(int nlay, size_t nwav, size_t stride,
                                  const double* planck, const double* spectral_od,
                                  const double* grey_od,
                                  const double* surf_emissivity,
                                  const double* surf_planck, double* flux_dn,
                                  double* flux_up);

/* ---- a5: radiative_transfer_sw.cpp:26-43, 49-77, 118-141, 147-184 ---- */
void orc_radiative_transfer_direct_sw(int nlay, size_t nwav, double cos_sza,
                                      const double* ssi, const double* od,
                                      double* flux_dn);
void orc_radiative_transfer_norayleigh_sw(int nlay, size_t nwav, double cos_sza,
                                          const double* ssi, const double* od,
                                          const double* albedo, double* flux_dn,
                                          double* flux_up);
void orc_radiative_transfer_direct_sw_bb(int nlay, size_t nwav, size_t stride,
                                         double cos_sza, const double* ssi,
                                         const double* spectral_od,
                                         const double* grey_od, double* flux_dn);
void orc_radiative_transfer_norayleigh_sw_bb(int nlay, size_t nwav, size_t stride,
                                             double cos_sza, const double* ssi,
                                             const double* spectral_od,
                                             const double* grey_od, double albedo,
                                             double* flux_dn, double* flux_up);

/* ---- a6: heating_rate.h:30-50, 55-72 (flux_up may be NULL = "empty") ---- */
void orc_heating_rate(int nlay, size_t nwav, const double* pressure_hl,
                      const double* flux_dn, const double* flux_up, double* hr);
void orc_heating_rate_single(int nlay, const double* pressure_hl,
                             const double* flux_dn, const double* flux_up,
                             double* hr);

/* ---- a7: reorder_spectrum.cpp:111-228 (do_sw = ssi != NULL) ----
 * temperature_hl: the idealised profile of :121-124 computed by the caller
 * (LW only).  Returns 0, or 1 if the SW key exceeds 30 (bare throw, :214). */
int orc_reorder_key(int nlay, size_t nwav, const double* pressure_hl,
                    const double* temperature_hl, const double* wavenumber_cm_1,
                    const double* d_wavenumber_cm_1, const double* od,
                    const double* ssi, double threshold_optical_depth,
                    double* sorting_variable, double* column_optical_depth);

/* Idealised temperature profile, reorder_spectrum.cpp:121-124 (adept::interp
 * restated as linear interpolation/extrapolation in ln p). */
void orc_idealised_temperature(int nhl, const double* pressure_hl, double* t_hl);

/* ---- a8: reorder_spectrum.cpp:262-300 ----
 * band membership from unclamped bounds (:281-288); iband = -1 outside;
 * std::stable_sort with '<' on the key restated as a bottom-up merge sort. */
void orc_stable_argsort_bands(size_t nwav, const double* wavenumber_cm_1,
                              const double* key, int nband,
                              const double* band_bound1, const double* band_bound2,
                              int32_t* iband, int32_t* ordered_index,
                              int32_t* rank);

/* ---- a10: gas prep of find_g_points.cpp:872-1150 (inputs already reordered) */
#define ORC_AVG_LINEAR 0
#define ORC_AVG_TRANSMISSION 1
#define ORC_AVG_TRANSMISSION_2 2
#define ORC_AVG_SQUARE_ROOT 3
#define ORC_AVG_LOGARITHMIC 4
#define ORC_AVG_TOTAL_TRANSMISSION 5

void orc_layer_weight(int nlay, const double* pressure_hl, double min_pressure,
                      double* layer_weight); /* find_g_points.cpp:1093-1099 */
void orc_metric(int method, size_t n, const double* od, double* metric); /* :1119-1150 */

/* ---- a11: find_g_points.cpp:54-106, 112-165, 171-204 ---- */
void orc_fit_optical_depth_lw(int method, int nlay, size_t stride, size_t i1,
                              size_t i2, const double* planck_hl,
                              const double* metric, double* od_fit);
void orc_fit_optical_depth_sw(int method, int nlay, size_t stride, size_t i1,
                              size_t i2, const double* ssi, const double* metric,
                              double* od_fit);
void orc_fit_optical_depth_sw_total_trans(int nlay, size_t stride, size_t i1,
                                          size_t i2, const double* ssi,
                                          const double* bg_od, const double* od,
                                          double* od_fit);

/* ---- a12: calc_cost_function_lw.cpp:24-110, calc_cost_function_sw.cpp:21-110
 * All spectral arrays point at the first wavenumber of the range [0,n). */
double orc_calc_cost_function_lw(int nlay, size_t n, size_t stride,
                                 const double* pressure_hl, const double* planck_hl,
                                 const double* surf_emissivity,
                                 const double* surf_planck, const double* bg_od,
                                 const double* od_fit, const double* flux_dn_surf,
                                 const double* flux_up_toa, const double* hr,
                                 double flux_weight, const double* layer_weight);
double orc_calc_cost_function_sw(int nlay, size_t n, size_t stride, double cos_sza,
                                 const double* pressure_hl, const double* ssi,
                                 double albedo, const double* bg_od,
                                 const double* od_fit, const double* flux_dn_surf,
                                 const double* flux_up_toa, const double* hr,
                                 double flux_weight, const double* layer_weight);

/* ---- a13 (error part): CkdEquipartition, find_g_points.cpp:206-426 ---- */
typedef struct {
  int do_sw;
  int method;
  int nlay;
  size_t npoints;
  size_t stride; /* row stride of the 2-D arrays */
  double flux_weight;
  double cos_sza;
  double surf_albedo;
  const double* layer_weight;
  const double* pressure_hl;
  const double* ssi;             /* SW */
  const double* surf_emissivity; /* LW */
  const double* surf_planck;     /* LW */
  const double* flux_dn_surf;
  const double* flux_up_toa;
  const double* planck_hl; /* LW (nlay+1, stride) */
  const double* bg_od;     /* (nlay, stride) */
  const double* metric;    /* (nlay, stride) */
  const double* hr;        /* (nlay, stride) */
  /* total-transmission extras, find_g_points.cpp:263-278 */
  const double* flux_dn_surf_low;
  const double* flux_up_toa_low;
  const double* flux_dn_surf_high;
  const double* flux_up_toa_high;
  const double* hr_low;
  const double* hr_high;
  double min_scaling, max_scaling;
  double total_comp_cost;
} orc_ckd_equipartition;

/* Returns the interval error, or NaN after setting *status != 0 on the
 * reference's throw(PROCESSING_ERROR) paths (find_g_points.cpp:298-313). */
double orc_ckd_calc_error(orc_ckd_equipartition* eq, double bound1, double bound2,
                          int* status);

/* Planck-weighted median, find_g_points.cpp:34-49 */
double orc_median_sorting_variable(const double* sorting_variable,
                                   const double* weight, size_t i1, size_t i2);

/* ---- reverse mode by hand of the longwave optimize_lut forward model (oracle_adjoint.c): the gradient the reference gets
 * from Adept's tape (solve_adept.cpp:91, :201-203).  d_od / d_molar_abs are ACCUMULATED into. */
double orc_calc_cost_function_ckd_lw_ad(int nlay, int ng, int nband, const double* pressure_hl, const double* planck_hl,
                                        const double* surf_emiss_orig, const double* surf_planck,
                                        const double* optical_depth, const double* flux_dn, const double* flux_up,
                                        const double* hr, const double* spectral_flux_dn_surf,
                                        const double* spectral_flux_up_toa, double flux_weight,
                                        double flux_profile_weight, double broadband_weight,
                                        double spectral_boundary_weight, const double* layer_weight,
                                        const double* relative_ckd_flux_dn, const double* relative_ckd_flux_up,
                                        const int* band_mapping, double* d_od);
double orc_calc_cost_function_ckd_sw_ad(int nlay, int ng, int nband, double cos_sza, const double* pressure_hl,
                                        const double* ssi, const double* albedo, const double* optical_depth,
                                        const double* flux_dn, const double* flux_up, const double* hr,
                                        const double* spectral_flux_dn_surf, double flux_weight,
                                        double flux_profile_weight, double broadband_weight,
                                        const double* spectral_boundary_weights, const double* layer_weight,
                                        const double* relative_ckd_flux_dn, const double* relative_ckd_flux_up,
                                        const int* band_mapping, double* d_od);
int orc_ckd_optical_depth_ad(int ng, int nt, int np, const double* log_pressure, const double* temperature,
                             int conc_dependence, int nconc, const double* vmr_lut, double reference_vmr, int ncol, int nlay,
                             const double* pressure_hl, const double* temperature_fl, const double* vmr_fl,
                             const double* d_od, double* d_molar_abs);

/* ---- the chain reorder_spectrum -> find_g_points for one longwave gas, in C end to end (oracle_chain.c): what bench.py
 * times as `cpu_baseline`.  refep_path: oracle/_ref/libequipartition_ref.so (the reference's own Equipartition). */
int orc_find_g_lw_chain(const char* refep_path, int nlay, size_t nwav, const double* pressure_hl,
                        const double* temperature_hl, const double* wn, const double* dwn, const float* od32,
                        const float* bg32, double threshold_optical_depth, int nband, const double* band_bound1,
                        const double* band_bound2, int method, double flux_weight, double min_pressure,
                        const double* tolerance, double tolerance_tolerance, int max_iterations, int parallel,
                        int* ng, double* comp_cost, int* status, double* seconds, int32_t* rank);

int orc_find_g_lw_chain_ex(const char* refep_path, int nlay, size_t nwav, const double* pressure_hl,
                           const double* temperature_hl, const double* wn, const double* dwn, const float* od32,
                           const float* bg32, double threshold_optical_depth, int nband, const double* band_bound1,
                           const double* band_bound2, int method, double flux_weight, double min_pressure,
                           const double* tolerance, double tolerance_tolerance, int max_iterations, int parallel,
                           int* ng, double* comp_cost, int* status, double* seconds, int32_t* rank, int capacity,
                           int64_t* rank1, int64_t* rank2, double* error_out, double* median_out, double* key_out,
                           double* planck_io, int planck_mode);

#ifdef __cplusplus
}
#endif
void orc_chain_set_background64(const double* bg64);   /* oracle_chain.c */
#endif
