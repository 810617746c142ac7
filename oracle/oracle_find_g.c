/* oracle_find_g.c - CPU restatement (TEST INFRASTRUCTURE, see ecckd_oracle.h) of
 * the inner loop of reference src/ecckd/find_g_points.cpp: fitted grey optical
 * depth of an interval, its heating-rate/flux cost, and the interval-error
 * callback of the partition search. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ecckd_oracle.h"

/* reference src/ecckd/find_g_points.cpp:1093-1099 */
void orc_layer_weight(int nlay, const double* pressure_hl, double min_pressure,
                      double* layer_weight) {
  double s = 0.0;
  for (int l = 0; l < nlay; ++l) {
    layer_weight[l] = sqrt(pressure_hl[l + 1]) - sqrt(pressure_hl[l]);
    double pressure_fl = 0.5 * (pressure_hl[l + 1] + pressure_hl[l]);
    if (pressure_fl < min_pressure) layer_weight[l] = 0.0;
  }
  for (int l = 0; l < nlay; ++l) s += layer_weight[l];
  for (int l = 0; l < nlay; ++l) layer_weight[l] /= s;
}

/* reference src/ecckd/find_g_points.cpp:1119-1150 */
void orc_metric(int method, size_t n, const double* od, double* metric) {
  switch (method) {
    case ORC_AVG_LINEAR:
    case ORC_AVG_LOGARITHMIC:
    case ORC_AVG_TOTAL_TRANSMISSION:
      for (size_t i = 0; i < n; ++i) metric[i] = od[i];
      break;
    case ORC_AVG_TRANSMISSION:
      for (size_t i = 0; i < n; ++i) metric[i] = 1.0 - exp(-od[i] * ORC_LW_DIFFUSIVITY);
      break;
    case ORC_AVG_TRANSMISSION_2:
      for (size_t i = 0; i < n; ++i) metric[i] = 1.0 - exp(-od[i] * ORC_LW_DIFFUSIVITY * 2.0);
      break;
    case ORC_AVG_SQUARE_ROOT:
      for (size_t i = 0; i < n; ++i) metric[i] = sqrt(od[i]);
      break;
  }
}

/* reference src/ecckd/find_g_points.cpp:34-49 */
double orc_median_sorting_variable(const double* sorting_variable,
                                   const double* weight, size_t i1, size_t i2) {
  double s = 0.0;
  for (size_t i = i1; i <= i2; ++i) s += weight[i];
  double half = 0.5 * s;
  double cum = 0.0;
  size_t iind = i1;
  for (; iind < i2; ++iind) {
    cum += weight[iind];
    if (cum >= half) break;
  }
  return sorting_variable[iind];
}

/* a11 -- reference src/ecckd/find_g_points.cpp:54-106.  Weights are
 * planck_hl(range(1,end), .) i.e. the half-level BELOW each layer; the
 * "logarithmic" branch uses planck_hl(iz+1,.) in the numerator and
 * planck_hl(iz,.) in the denominator (:86-87, :95-96). */
void orc_fit_optical_depth_lw(int method, int nlay, size_t stride, size_t i1,
                              size_t i2, const double* planck_hl,
                              const double* metric, double* od_fit) {
  if (method == ORC_AVG_LOGARITHMIC) {
    for (int iz = 0; iz < nlay; ++iz) {
      const double* m = metric + (size_t)iz * stride;
      const double* pn = planck_hl + (size_t)(iz + 1) * stride;
      const double* pd = planck_hl + (size_t)iz * stride;
      double num = 0.0, den = 0.0;
      size_t nnz = 0;
      for (size_t j = i1; j <= i2; ++j) {
        if (m[j] > 0.0) {
          num += log(m[j]) * pn[j];
          den += pd[j];
          ++nnz;
        }
      }
      if (nnz == i2 - i1 + 1) od_fit[iz] = exp(num / den);
      else if (nnz == 0) od_fit[iz] = 0.0;
      else od_fit[iz] = exp(num / den) * ((double)nnz / (double)(i2 - i1 + 1));
    }
    return;
  }
  for (int iz = 0; iz < nlay; ++iz) {
    const double* m = metric + (size_t)iz * stride;
    const double* p = planck_hl + (size_t)(iz + 1) * stride;
    double num = 0.0, den = 0.0;
    for (size_t j = i1; j <= i2; ++j) {
      num += m[j] * p[j];
      den += p[j];
    }
    double v = num / den;
    switch (method) {
      case ORC_AVG_LINEAR:
        od_fit[iz] = v;
        break;
      case ORC_AVG_TRANSMISSION:
        v = fmin(0.9999999999999999, v);
        od_fit[iz] = fabs(-log(1.0 - v) / ORC_LW_DIFFUSIVITY);
        break;
      case ORC_AVG_TRANSMISSION_2:
        v = fmin(0.9999999999999999, v);
        od_fit[iz] = fabs(-log(1.0 - v) / (ORC_LW_DIFFUSIVITY * 2.0));
        break;
      case ORC_AVG_SQUARE_ROOT:
        od_fit[iz] = v * v;
        break;
      default:
        od_fit[iz] = NAN; /* reference throws PARAMETER_ERROR (:101-104) */
    }
  }
}

/* a11 -- reference src/ecckd/find_g_points.cpp:112-165.  In the transmission
 * branches min(0.9999999999999999, .) is applied BEFORE norm_factor (:123-124). */
void orc_fit_optical_depth_sw(int method, int nlay, size_t stride, size_t i1,
                              size_t i2, const double* ssi, const double* metric,
                              double* od_fit) {
  double ssum = 0.0;
  for (size_t j = i1; j <= i2; ++j) ssum += ssi[j];
  const double norm_factor = 1.0 / ssum;
  if (method == ORC_AVG_LOGARITHMIC || method == ORC_AVG_TOTAL_TRANSMISSION) {
    for (int iz = 0; iz < nlay; ++iz) {
      const double* m = metric + (size_t)iz * stride;
      double num = 0.0, den = 0.0;
      size_t nnz = 0;
      for (size_t j = i1; j <= i2; ++j) {
        if (m[j] > 0.0) {
          num += log(m[j]) * ssi[j];
          den += ssi[j];
          ++nnz;
        }
      }
      if (nnz == i2 - i1 + 1) od_fit[iz] = exp(num / den);
      else if (nnz == 0) od_fit[iz] = 0.0;
      else od_fit[iz] = exp(num / den) * ((double)nnz / (double)(i2 - i1 + 1));
    }
    return;
  }
  for (int iz = 0; iz < nlay; ++iz) {
    const double* m = metric + (size_t)iz * stride;
    double num = 0.0;
    for (size_t j = i1; j <= i2; ++j) num += m[j] * ssi[j];
    switch (method) {
      case ORC_AVG_LINEAR:
        od_fit[iz] = num * norm_factor;
        break;
      case ORC_AVG_TRANSMISSION: {
        double v = fmin(0.9999999999999999, num) * norm_factor;
        od_fit[iz] = fabs(-log(1.0 - v) / ORC_LW_DIFFUSIVITY);
        break;
      }
      case ORC_AVG_TRANSMISSION_2: {
        double v = fmin(0.9999999999999999, num) * norm_factor;
        od_fit[iz] = fabs(-log(1.0 - v) / (ORC_LW_DIFFUSIVITY * 2.0));
        break;
      }
      case ORC_AVG_SQUARE_ROOT: {
        double v = num * norm_factor;
        od_fit[iz] = v * v;
        break;
      }
      default:
        od_fit[iz] = NAN;
    }
  }
}

/* a11 -- reference src/ecckd/find_g_points.cpp:171-204.  On a non-positive
 * broadband flux the WHOLE vector is overwritten with the linear average and
 * the loop continues (:197-199). */
void orc_fit_optical_depth_sw_total_trans(int nlay, size_t stride, size_t i1,
                                          size_t i2, const double* ssi,
                                          const double* bg_od, const double* od,
                                          double* od_fit) {
  const size_t n = i2 - i1 + 1;
  double* flux_dn = (double*)malloc(n * sizeof(double));
  double* bg_flux_dn = (double*)malloc(n * sizeof(double));
  double ssum = 0.0;
  for (size_t j = 0; j < n; ++j) {
    flux_dn[j] = ssi[i1 + j];
    bg_flux_dn[j] = ssi[i1 + j];
    ssum += ssi[i1 + j];
  }
  double bb_flux_dn_top = ssum, bb_bg_flux_dn_top = ssum;
  const double norm_factor = 1.0 / ssum;
  for (int iz = 0; iz < nlay; ++iz) {
    const double* b = bg_od + (size_t)iz * stride + i1;
    const double* o = od + (size_t)iz * stride + i1;
    double bb_bg_flux_dn_base = 0.0, bb_flux_dn_base = 0.0;
    for (size_t j = 0; j < n; ++j) {
      bg_flux_dn[j] *= exp(-2.0 * b[j]);
      flux_dn[j] *= exp(-2.0 * (b[j] + o[j]));
    }
    for (size_t j = 0; j < n; ++j) bb_bg_flux_dn_base += bg_flux_dn[j];
    for (size_t j = 0; j < n; ++j) bb_flux_dn_base += flux_dn[j];
    if (bb_bg_flux_dn_base > 0.0 && bb_flux_dn_base > 0.0) {
      double bg_od_fit = -0.5 * log(bb_bg_flux_dn_base / bb_bg_flux_dn_top);
      od_fit[iz] = -0.5 * log(bb_flux_dn_base / bb_flux_dn_top) - bg_od_fit;
    } else {
      for (int kz = 0; kz < nlay; ++kz) {
        const double* ok = od + (size_t)kz * stride + i1;
        double num = 0.0;
        for (size_t j = 0; j < n; ++j) num += ok[j] * ssi[i1 + j];
        od_fit[kz] = num * norm_factor;
      }
    }
    bb_flux_dn_top = bb_flux_dn_base;
    bb_bg_flux_dn_top = bb_bg_flux_dn_base;
  }
  free(flux_dn);
  free(bg_flux_dn);
}

/* a12 -- reference src/ecckd/calc_cost_function_lw.cpp:24-110 (index empty).
 * true sums :50-52; broadband RT with the grey fit :80-86; heating rate :102;
 * result :107-109. */
double orc_calc_cost_function_lw(int nlay, size_t n, size_t stride,
                                 const double* pressure_hl, const double* planck_hl,
                                 const double* surf_emissivity,
                                 const double* surf_planck, const double* bg_od,
                                 const double* od_fit, const double* flux_dn_surf,
                                 const double* flux_up_toa, const double* hr,
                                 double flux_weight, const double* layer_weight) {
  static const double hr_weight = 3600.0 * 24.0;
  double* hr_true = (double*)malloc((size_t)nlay * sizeof(double));
  double* hr_fit = (double*)malloc((size_t)nlay * sizeof(double));
  double* flux_dn_fit = (double*)malloc((size_t)(nlay + 1) * sizeof(double));
  double* flux_up_fit = (double*)malloc((size_t)(nlay + 1) * sizeof(double));
  double flux_dn_surf_true = 0.0, flux_up_toa_true = 0.0;
  for (int l = 0; l < nlay; ++l) {
    const double* h = hr + (size_t)l * stride;
    double s = 0.0;
    for (size_t j = 0; j < n; ++j) s += h[j];
    hr_true[l] = s;
  }
  for (size_t j = 0; j < n; ++j) flux_dn_surf_true += flux_dn_surf[j];
  for (size_t j = 0; j < n; ++j) flux_up_toa_true += flux_up_toa[j];

  orc_radiative_transfer_lw_bb(nlay, n, stride, planck_hl, bg_od, od_fit,
                               surf_emissivity, surf_planck, flux_dn_fit, flux_up_fit);
  orc_heating_rate_single(nlay, pressure_hl, flux_dn_fit, flux_up_fit, hr_fit);

  double s = 0.0;
  for (int l = 0; l < nlay; ++l)
    s += layer_weight[l] * ((hr_fit[l] - hr_true[l]) * (hr_fit[l] - hr_true[l]));
  double dsurf = flux_dn_fit[nlay] - flux_dn_surf_true;
  double dtoa = flux_up_fit[0] - flux_up_toa_true;
  double ans = sqrt(hr_weight * hr_weight * s + flux_weight * (dsurf * dsurf + dtoa * dtoa));
  free(hr_true);
  free(hr_fit);
  free(flux_dn_fit);
  free(flux_up_fit);
  return ans;
}

/* a12 -- reference src/ecckd/calc_cost_function_sw.cpp:21-110 (index empty).
 * albedo <= 0 -> direct only (:60-66), else no-Rayleigh with upwelling
 * (:67-71); heating rate from the direct beam only (:92). */
double orc_calc_cost_function_sw(int nlay, size_t n, size_t stride, double cos_sza,
                                 const double* pressure_hl, const double* ssi,
                                 double albedo, const double* bg_od,
                                 const double* od_fit, const double* flux_dn_surf,
                                 const double* flux_up_toa, const double* hr,
                                 double flux_weight, const double* layer_weight) {
  static const double hr_weight = 3600.0 * 24.0;
  double* hr_true = (double*)malloc((size_t)nlay * sizeof(double));
  double* hr_fit = (double*)malloc((size_t)nlay * sizeof(double));
  double* flux_dn_fit = (double*)malloc((size_t)(nlay + 1) * sizeof(double));
  double* flux_up_fit = (double*)calloc((size_t)(nlay + 1), sizeof(double));
  double flux_dn_surf_true = 0.0, flux_up_toa_true = 0.0;
  for (int l = 0; l < nlay; ++l) {
    const double* h = hr + (size_t)l * stride;
    double s = 0.0;
    for (size_t j = 0; j < n; ++j) s += h[j];
    hr_true[l] = s;
  }
  for (size_t j = 0; j < n; ++j) flux_dn_surf_true += flux_dn_surf[j];
  for (size_t j = 0; j < n; ++j) flux_up_toa_true += flux_up_toa[j];

  if (albedo <= 0.0) {
    orc_radiative_transfer_direct_sw_bb(nlay, n, stride, cos_sza, ssi, bg_od, od_fit,
                                        flux_dn_fit);
  } else {
    orc_radiative_transfer_norayleigh_sw_bb(nlay, n, stride, cos_sza, ssi, bg_od, od_fit,
                                            albedo, flux_dn_fit, flux_up_fit);
  }
  orc_heating_rate_single(nlay, pressure_hl, flux_dn_fit, NULL, hr_fit);

  double s = 0.0;
  for (int l = 0; l < nlay; ++l)
    s += layer_weight[l] * ((hr_fit[l] - hr_true[l]) * (hr_fit[l] - hr_true[l]));
  double dsurf = flux_dn_fit[nlay] - flux_dn_surf_true;
  double dtoa = flux_up_fit[0] - flux_up_toa_true;
  double ans = sqrt(hr_weight * hr_weight * s + flux_weight * (dsurf * dsurf + dtoa * dtoa));
  free(hr_true);
  free(hr_fit);
  free(flux_dn_fit);
  free(flux_up_fit);
  return ans;
}

/* a13 (error callback) -- reference src/ecckd/find_g_points.cpp:282-405.
 * lower = ceil(b1*(n-1)), upper = floor(b2*(n-1)) (:282-287); error paths
 * :298-313; upper<lower by one is corrected (:314-318); total_comp_cost
 * accumulates bound2-bound1 (:320). */
double orc_ckd_calc_error(orc_ckd_equipartition* eq, double bound1, double bound2,
                          int* status) {
  const long npoints = (long)eq->npoints;
  long ibound1 = (long)ceil(bound1 * (double)(npoints - 1));
  long ibound2 = (long)floor(bound2 * (double)(npoints - 1));
  *status = 0;
  if (ibound1 < 0 || ibound2 >= npoints) { *status = 1; return NAN; }
  else if (bound2 < bound1) { *status = 2; return NAN; }
  else if (ibound2 + 1 < ibound1) { *status = 3; return NAN; }
  else if (ibound2 < ibound1) ibound2 = ibound1;

  eq->total_comp_cost += bound2 - bound1;

  const size_t i1 = (size_t)ibound1, i2 = (size_t)ibound2;
  const size_t n = i2 - i1 + 1;
  const int nlay = eq->nlay;
  double* od_fit = (double*)malloc((size_t)nlay * sizeof(double));
  double ans;
  if (!eq->do_sw) {
    orc_fit_optical_depth_lw(eq->method, nlay, eq->stride, i1, i2, eq->planck_hl,
                             eq->metric, od_fit);
    ans = orc_calc_cost_function_lw(nlay, n, eq->stride, eq->pressure_hl,
                                    eq->planck_hl + i1, eq->surf_emissivity + i1,
                                    eq->surf_planck + i1, eq->bg_od + i1, od_fit,
                                    eq->flux_dn_surf + i1, eq->flux_up_toa + i1,
                                    eq->hr + i1, eq->flux_weight, eq->layer_weight);
  } else if (eq->method == ORC_AVG_TOTAL_TRANSMISSION) {
    double* od_scaled = (double*)malloc((size_t)nlay * sizeof(double));
    orc_fit_optical_depth_sw_total_trans(nlay, eq->stride, i1, i2, eq->ssi, eq->bg_od,
                                         eq->metric, od_fit);
    for (int l = 0; l < nlay; ++l) od_scaled[l] = od_fit[l] * eq->min_scaling;
    double cf_low = orc_calc_cost_function_sw(
        nlay, n, eq->stride, eq->cos_sza, eq->pressure_hl, eq->ssi + i1, eq->surf_albedo,
        eq->bg_od + i1, od_scaled, eq->flux_dn_surf_low + i1, eq->flux_up_toa_low + i1,
        eq->hr_low + i1, eq->flux_weight, eq->layer_weight);
    for (int l = 0; l < nlay; ++l) od_scaled[l] = od_fit[l] * eq->max_scaling;
    double cf_high = orc_calc_cost_function_sw(
        nlay, n, eq->stride, eq->cos_sza, eq->pressure_hl, eq->ssi + i1, eq->surf_albedo,
        eq->bg_od + i1, od_scaled, eq->flux_dn_surf_high + i1, eq->flux_up_toa_high + i1,
        eq->hr_high + i1, eq->flux_weight, eq->layer_weight);
    ans = 0.5 * (cf_low + cf_high);
    free(od_scaled);
  } else {
    orc_fit_optical_depth_sw(eq->method, nlay, eq->stride, i1, i2, eq->ssi, eq->metric,
                             od_fit);
    ans = orc_calc_cost_function_sw(nlay, n, eq->stride, eq->cos_sza, eq->pressure_hl,
                                    eq->ssi + i1, eq->surf_albedo, eq->bg_od + i1, od_fit,
                                    eq->flux_dn_surf + i1, eq->flux_up_toa + i1,
                                    eq->hr + i1, eq->flux_weight, eq->layer_weight);
  }
  free(od_fit);
  return ans;
}
