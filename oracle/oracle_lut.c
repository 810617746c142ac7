/* oracle_lut.c - CPU restatement (TEST INFRASTRUCTURE, see ecckd_oracle.h) of the
 * create_look_up_table hot path: averaging of line-by-line optical depth to g points,
 * g-point fractions and the Planck look-up table. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ecckd_oracle.h"

#define M_LINEAR 0
#define M_TRANSMISSION 1
#define M_TRANSMISSION_2 2
#define M_SQUARE_ROOT 3
#define M_LOGARITHMIC 4
#define M_TRANSMISSION_3 6
#define M_TRANSMISSION_10 7
#define M_HYBRID_LOG_TRANS3 8

static double trans_fit(double num, double den, double k) {
  double v = fmin(0.9999999999999999, num / den);
  return fabs(-log(1.0 - v) / (ORC_LW_DIFFUSIVITY * 1.0 * k));
}

/* a15 -- reference src/ecckd/average_optical_depth.cpp:22-197.
 * g_point[nwav] (-1 = unassigned); od, weight: (nlay, nwav); outputs (nlay, ng).
 * Returns the number of empty g points (the reference warns and writes zeros, :135-141). */
int orc_average_optical_depth_to_g_point(int ng, double reference_surface_vmr, int nlay, size_t nwav,
                                         const double* pressure_fl, const double* pressure_hl,
                                         const int32_t* g_point, const double* od, const double* weight,
                                         int method, double* molar_abs, double* min_molar_abs,
                                         double* max_molar_abs) {
  const double OD_SCALING = 1.0;
  int nempty = 0;
  size_t* index = (size_t*)malloc(nwav * sizeof(size_t));
  for (int ig = 0; ig < ng; ++ig) {
    size_t n = 0;
    for (size_t j = 0; j < nwav; ++j)
      if (g_point[j] == ig) index[n++] = j;
    for (int iz = 0; iz < nlay; ++iz) {
      const double* o = od + (size_t)iz * nwav;
      const double* w = weight + (size_t)iz * nwav;
      double fit = 0.0, mn = 0.0, mx = 0.0;
      if (n > 0) {
        int lm = method;
        if (method == M_HYBRID_LOG_TRANS3) lm = (pressure_fl[iz] > 100.0e2) ? M_LOGARITHMIC : M_TRANSMISSION_3;
        double num = 0.0, den = 0.0;
        if (lm == M_LINEAR) {
          for (size_t q = 0; q < n; ++q) { num += o[index[q]] * w[index[q]]; den += w[index[q]]; }
          fit = num / den;
        } else if (lm == M_TRANSMISSION || lm == M_TRANSMISSION_2 || lm == M_TRANSMISSION_3 || lm == M_TRANSMISSION_10) {
          const double k = lm == M_TRANSMISSION ? 1.0 : lm == M_TRANSMISSION_2 ? 2.0 : lm == M_TRANSMISSION_3 ? 3.0 : 10.0;
          for (size_t q = 0; q < n; ++q) {
            num += (1.0 - exp(-o[index[q]] * (ORC_LW_DIFFUSIVITY * OD_SCALING * k))) * w[index[q]];
            den += w[index[q]];
          }
          fit = trans_fit(num, den, k);
        } else if (lm == M_SQUARE_ROOT) {
          for (size_t q = 0; q < n; ++q) { num += sqrt(o[index[q]]) * w[index[q]]; den += w[index[q]]; }
          fit = num / den;
          fit *= fit;
        } else { /* logarithmic, :79-99 */
          size_t nnz = 0;
          double den_all = 0.0;
          for (size_t q = 0; q < n; ++q) {
            den_all += w[index[q]];
            if (o[index[q]] > 0.0) { num += log(o[index[q]]) * w[index[q]]; den += w[index[q]]; ++nnz; }
          }
          if (nnz == n) fit = exp(num / den_all);
          else if (nnz == 0) fit = 0.0;
          else fit = exp(num / den) * ((double)nnz / (double)n);
        }
        mn = mx = o[index[0]];
        for (size_t q = 1; q < n; ++q) { if (o[index[q]] < mn) mn = o[index[q]]; if (o[index[q]] > mx) mx = o[index[q]]; }
        /* :151-165 */
        fit = fmax(mn, fmin(fit, mx));
        if (mn > fit) mn = fit;
        if (mn > 0.0 && mn >= mx) { mn *= 0.99; mx *= 1.01; }
      }
      const double dp = pressure_hl[iz + 1] - pressure_hl[iz];
      double scale = 1.0;
      if (reference_surface_vmr > 0.0)
        scale = ((ORC_ACCEL_GRAVITY * 0.001 * ORC_MOLAR_MASS_DRY_AIR) / reference_surface_vmr);
      if (reference_surface_vmr > 0.0) {
        molar_abs[(size_t)iz * ng + ig] = scale * fit / dp;
        if (min_molar_abs) { min_molar_abs[(size_t)iz * ng + ig] = scale * mn / dp; max_molar_abs[(size_t)iz * ng + ig] = scale * mx / dp; }
      } else {
        molar_abs[(size_t)iz * ng + ig] = fit;
        if (min_molar_abs) { min_molar_abs[(size_t)iz * ng + ig] = mn; max_molar_abs[(size_t)iz * ng + ig] = mx; }
      }
    }
    if (n == 0) ++nempty;
  }
  free(index);
  return nempty;
}

/* a16 -- reference src/ecckd/create_look_up_table.cpp:537-548: fraction of each g point's
 * spectral width in each coarse interval (wavenumber1, wavenumber2]. */
void orc_gpoint_fraction(int ng, int nint, size_t nwav, const int32_t* g_point, const double* wavenumber_cm_1,
                         const double* d_wavenumber_cm_1, const double* wavenumber1, const double* wavenumber2,
                         double* gpoint_fraction /* [ng][nint] */) {
  for (int ig = 0; ig < ng; ++ig) {
    double wav_per_gpoint = 0.0;
    for (size_t j = 0; j < nwav; ++j)
      if (g_point[j] == ig) wav_per_gpoint += d_wavenumber_cm_1[j];
    for (int iw = 0; iw < nint; ++iw) {
      double s = 0.0;
      for (size_t j = 0; j < nwav; ++j)
        if (g_point[j] == ig && wavenumber_cm_1[j] > wavenumber1[iw] && wavenumber_cm_1[j] <= wavenumber2[iw])
          s += d_wavenumber_cm_1[j];
      gpoint_fraction[(size_t)ig * nint + iw] = s / wav_per_gpoint;
    }
  }
}

/* a16 -- reference src/ecckd/create_look_up_table.cpp:581-591: Planck function summed over the
 * wavenumbers of each g point for the LUT temperatures. planck_lut[nlut][ng]. */
void orc_planck_lut(int ng, int nlut, const double* temperature_lut, size_t nwav, const int32_t* g_point,
                    const double* wavenumber_cm_1, const double* d_wavenumber_cm_1, double* planck_lut) {
  double* wn = (double*)malloc(nwav * sizeof(double));
  double* dwn = (double*)malloc(nwav * sizeof(double));
  for (int ig = 0; ig < ng; ++ig) {
    size_t n = 0;
    for (size_t j = 0; j < nwav; ++j)
      if (g_point[j] == ig) { wn[n] = wavenumber_cm_1[j]; dwn[n] = d_wavenumber_cm_1[j]; ++n; }
    double* tmp = (double*)malloc((size_t)nlut * (n ? n : 1) * sizeof(double));
    orc_planck_function(nlut, temperature_lut, n, wn, dwn, tmp);
    for (int it = 0; it < nlut; ++it) {
      double s = 0.0;
      for (size_t q = 0; q < n; ++q) s += tmp[(size_t)it * n + q];
      planck_lut[(size_t)it * ng + ig] = s;
    }
    free(tmp);
  }
  free(wn);
  free(dwn);
}
