/* oracle_reorder.c - CPU restatement (TEST INFRASTRUCTURE, see ecckd_oracle.h)
 * of the sorting-key calculation and per-band stable sort of
 * reference src/ecckd/reorder_spectrum.cpp. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ecckd_oracle.h"

/* reference src/ecckd/reorder_spectrum.cpp:121-124: idealised temperature,
 * linear in ln p between (1 Pa, 173.15 K) and (1e5 Pa, 288.15 K).  Adept's
 * interp() is restated as linear interpolation that extrapolates linearly
 * outside the two knots ("hard parts": flagged). */
void orc_idealised_temperature(int nhl, const double* pressure_hl, double* t_hl) {
  const double x0 = log(1.0), x1 = log(100000.0);
  const double y0 = 273.15 - 100.0, y1 = 273.15 + 15.0;
  for (int i = 0; i < nhl; ++i) {
    double w = (log(pressure_hl[i]) - x0) / (x1 - x0);
    t_hl[i] = (1.0 - w) * y0 + w * y1;
  }
}

/* a7 -- reference src/ecckd/reorder_spectrum.cpp:111-228.
 * LW: planck :127, surface planck :130-131, RT :141; SW: direct RT :157.
 * column od :162; heating rate :170; cooling only :175; peak height :178-183;
 * thin columns :187-190; threshold height :197-222; SW uses it as key :224-228. */
int orc_reorder_key(int nlay, size_t nwav, const double* pressure_hl,
                    const double* temperature_hl, const double* wavenumber_cm_1,
                    const double* d_wavenumber_cm_1, const double* od,
                    const double* ssi, double thr, double* key, double* col_od) {
  const int nhl = nlay + 1;
  const int do_sw = (ssi != NULL);
  double* flux_dn = (double*)malloc((size_t)nhl * nwav * sizeof(double));
  double* flux_up = (double*)malloc((size_t)nhl * nwav * sizeof(double));
  double* hr = (double*)malloc((size_t)nlay * nwav * sizeof(double));
  int status = 0;

  if (!do_sw) {
    double* planck_hl = (double*)malloc((size_t)nhl * nwav * sizeof(double));
    double* surf_planck = (double*)malloc(nwav * sizeof(double));
    double* surf_emissivity = (double*)malloc(nwav * sizeof(double));
    orc_planck_function(nhl, temperature_hl, nwav, wavenumber_cm_1, d_wavenumber_cm_1,
                        planck_hl);
    orc_planck_function(1, temperature_hl + nlay, nwav, wavenumber_cm_1,
                        d_wavenumber_cm_1, surf_planck);
    for (size_t j = 0; j < nwav; ++j) surf_emissivity[j] = 1.0;
    orc_radiative_transfer_lw(nlay, nwav, planck_hl, od, surf_emissivity, surf_planck,
                              flux_dn, flux_up);
    free(planck_hl);
    free(surf_planck);
    free(surf_emissivity);
  } else {
    orc_radiative_transfer_direct_sw(nlay, nwav, ORC_REFERENCE_COS_SZA, ssi, od, flux_dn);
    memset(flux_up, 0, (size_t)nhl * nwav * sizeof(double));
  }

  /* :162 column_optical_depth = sum(optical_depth, 0) */
  for (size_t j = 0; j < nwav; ++j) col_od[j] = 0.0;
  for (int l = 0; l < nlay; ++l) {
    const double* o = od + (size_t)l * nwav;
    for (size_t j = 0; j < nwav; ++j) col_od[j] += o[j];
  }

  orc_heating_rate(nlay, nwav, pressure_hl, flux_dn, flux_up, hr);
  if (!do_sw) {
    for (size_t idx = 0; idx < (size_t)nlay * nwav; ++idx)
      if (hr[idx] > 0.0) hr[idx] = 0.0;
  }

  /* :178-183 */
  {
    double* num = (double*)calloc(nwav, sizeof(double));
    double* den = (double*)calloc(nwav, sizeof(double));
    const double log_ps = log(pressure_hl[nlay]);
    for (int l = 0; l < nlay; ++l) {
      const double pseudo_height = log_ps - 0.5 * (log(pressure_hl[l]) + log(pressure_hl[l + 1]));
      const double d_height = log(pressure_hl[l + 1]) - log(pressure_hl[l]);
      const double w = d_height * pseudo_height;
      const double* h = hr + (size_t)l * nwav;
      for (size_t j = 0; j < nwav; ++j) {
        num[j] += h[j] * w;
        den[j] += h[j] * d_height;
      }
    }
    for (size_t j = 0; j < nwav; ++j) key[j] = num[j] / den[j];
    free(num);
    free(den);
  }
  /* :187-190 */
  if (thr > 0.0) {
    for (size_t j = 0; j < nwav; ++j)
      if (col_od[j] < thr) key[j] = -thr + col_od[j];
  }

  /* :197-222 (computed for LW too; only a throw can be observed there) */
  {
    double* pseudo_height_hl = (double*)malloc((size_t)nhl * sizeof(double));
    const double log_ps = log(pressure_hl[nlay]);
    for (int i = 0; i < nhl; ++i) pseudo_height_hl[i] = log_ps - log(pressure_hl[i]);
    for (size_t j = 0; j < nwav; ++j) {
      double oth;
      if (col_od[j] <= thr) {
        oth = col_od[j] - thr;
      } else {
        double cum_od = 0.0;
        oth = 0.0;
        for (int l = 0; l < nlay; ++l) {
          double o = od[(size_t)l * nwav + j];
          double next_cum_od = cum_od + o;
          if (next_cum_od >= thr) {
            oth = ((thr - cum_od) * pseudo_height_hl[l + 1] +
                   (next_cum_od - thr) * pseudo_height_hl[l]) /
                  fmax(1.0e-12, o);
            if (oth > 30.0) status = 1;
            break;
          }
          cum_od = next_cum_od;
        }
      }
      if (do_sw) key[j] = oth;
    }
    free(pseudo_height_hl);
  }

  free(flux_dn);
  free(flux_up);
  free(hr);
  return status;
}

/* Bottom-up stable merge sort of idx[0..n) by key[idx] with strict '<' --
 * the observable behaviour of std::stable_sort(..., MyCompare) at
 * reference src/ecckd/reorder_spectrum.cpp:29-34, :294. */
static void stable_sort_indices(int32_t* idx, size_t n, const double* key) {
  int32_t* tmp = (int32_t*)malloc(n * sizeof(int32_t));
  int32_t* src = idx;
  int32_t* dst = tmp;
  for (size_t width = 1; width < n; width *= 2) {
    for (size_t lo = 0; lo < n; lo += 2 * width) {
      size_t mid = lo + width < n ? lo + width : n;
      size_t hi = lo + 2 * width < n ? lo + 2 * width : n;
      size_t a = lo, b = mid, o = lo;
      while (a < mid && b < hi) {
        /* take from the right run only if strictly less: keeps ties stable */
        if (key[src[b]] < key[src[a]]) dst[o++] = src[b++];
        else dst[o++] = src[a++];
      }
      while (a < mid) dst[o++] = src[a++];
      while (b < hi) dst[o++] = src[b++];
    }
    int32_t* t = src; src = dst; dst = t;
  }
  if (src != idx) memcpy(idx, src, n * sizeof(int32_t));
  free(tmp);
}

/* a8 -- reference src/ecckd/reorder_spectrum.cpp:262-300.
 * Band membership with the unclamped bounds, last band closed on the right
 * (:281-288); the sorted range is index(0)..index(end) (:290-294); rank :299-300. */
void orc_stable_argsort_bands(size_t nwav, const double* wavenumber_cm_1,
                              const double* key, int nband,
                              const double* band_bound1, const double* band_bound2,
                              int32_t* iband, int32_t* ordered_index,
                              int32_t* rank) {
  for (size_t j = 0; j < nwav; ++j) {
    ordered_index[j] = (int32_t)j;
    iband[j] = -1;
  }
  for (int jband = 0; jband < nband; ++jband) {
    size_t first = nwav, last = 0;
    int found = 0;
    for (size_t j = 0; j < nwav; ++j) {
      int in = (jband < nband - 1)
                   ? (wavenumber_cm_1[j] >= band_bound1[jband] && wavenumber_cm_1[j] < band_bound2[jband])
                   : (wavenumber_cm_1[j] >= band_bound1[jband] && wavenumber_cm_1[j] <= band_bound2[jband]);
      if (in) {
        iband[j] = jband;
        if (!found) { first = j; found = 1; }
        last = j;
      }
    }
    if (found) stable_sort_indices(ordered_index + first, last - first + 1, key);
  }
  for (size_t i = 0; i < nwav; ++i) rank[ordered_index[i]] = (int32_t)i;
}
