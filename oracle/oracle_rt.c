/* oracle_rt.c - CPU restatement (TEST INFRASTRUCTURE, see ecckd_oracle.h) of the
 * reference's Planck function, no-scattering LW / direct SW radiative transfer
 * and heating-rate kernels.  Same loop order as the reference: layers outer,
 * wavenumber inner, sums in index order. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ecckd_oracle.h"

/* a2 -- reference src/ecckd/planck_function.cpp:22-54.
 * constants :29-33; freq :48; prefactor :49-50; OpenMP over temperature :50. */
void orc_planck_function(int nt, const double* temperature, size_t nwav,
                         const double* wavenumber_cm_1,
                         const double* d_wavenumber_cm_1, double* planck) {
  static const double h = 6.62606896e-34;
  static const double c = 2.99792458e8;
  static const double k = 1.3806504e-23;
  const double inv_cm_2_Hz = 100.0 * c;
  static const double pi = 3.14159265358979323846;
  double* freq = (double*)malloc(nwav * sizeof(double));
  double* prefactor = (double*)malloc(nwav * sizeof(double));
  for (size_t j = 0; j < nwav; ++j) {
    freq[j] = wavenumber_cm_1[j] * inv_cm_2_Hz;
    /* (d_wavenumber*2.0*h*inv_cm_2_Hz*pi/(c*c)) * (freq*freq*freq), left to right */
    prefactor[j] = (d_wavenumber_cm_1[j] * 2.0 * h * inv_cm_2_Hz * pi / (c * c)) *
                   (freq[j] * freq[j] * freq[j]);
  }
#pragma omp parallel for
  for (int i = 0; i < nt; ++i) {
    double* row = planck + (size_t)i * nwav;
    const double ti = temperature[i];
    for (size_t j = 0; j < nwav; ++j) {
      row[j] = prefactor[j] / (exp((h / k) * (freq[j] / ti)) - 1.0);
    }
  }
  free(freq);
  free(prefactor);
}

/* a3 -- reference src/ecckd/radiative_transfer_lw.cpp:27-60.
 * emissivity :41; factor :42-43 (emissivity > 1e-5 ? 1 - eps/(D*od) : 0.5*eps);
 * down sweep :45-50; surface :52-53; up sweep :55-59. */
void orc_radiative_transfer_lw(int nlay, size_t nwav, const double* planck,
                               const double* od, const double* surf_emissivity,
                               const double* surf_planck, double* flux_dn,
                               double* flux_up) {
  double* emissivity = (double*)malloc((size_t)nlay * nwav * sizeof(double));
  double* factor = (double*)malloc((size_t)nlay * nwav * sizeof(double));
  for (size_t idx = 0; idx < (size_t)nlay * nwav; ++idx) {
    double e = 1.0 - exp(-ORC_LW_DIFFUSIVITY * od[idx]);
    emissivity[idx] = e;
    factor[idx] = (e > 1.0e-5) ? 1.0 - e * (1.0 / ORC_LW_DIFFUSIVITY) / od[idx] : 0.5 * e;
  }
  for (size_t j = 0; j < nwav; ++j) flux_dn[j] = 0.0;
  for (int l = 0; l < nlay; ++l) {
    const double* e = emissivity + (size_t)l * nwav;
    const double* f = factor + (size_t)l * nwav;
    const double* p0 = planck + (size_t)l * nwav;
    const double* p1 = planck + (size_t)(l + 1) * nwav;
    const double* d0 = flux_dn + (size_t)l * nwav;
    double* d1 = flux_dn + (size_t)(l + 1) * nwav;
    for (size_t j = 0; j < nwav; ++j) {
      d1[j] = d0[j] * (1.0 - e[j]) + p0[j] * (e[j] - f[j]) + p1[j] * f[j];
    }
  }
  {
    const double* dn = flux_dn + (size_t)nlay * nwav;
    double* up = flux_up + (size_t)nlay * nwav;
    for (size_t j = 0; j < nwav; ++j) {
      up[j] = surf_planck[j] * surf_emissivity[j] + (1.0 - surf_emissivity[j]) * dn[j];
    }
  }
  for (int l = nlay - 1; l >= 0; --l) {
    const double* e = emissivity + (size_t)l * nwav;
    const double* f = factor + (size_t)l * nwav;
    const double* p0 = planck + (size_t)l * nwav;
    const double* p1 = planck + (size_t)(l + 1) * nwav;
    const double* u1 = flux_up + (size_t)(l + 1) * nwav;
    double* u0 = flux_up + (size_t)l * nwav;
    for (size_t j = 0; j < nwav; ++j) {
      u0[j] = u1[j] * (1.0 - e[j]) + p1[j] * (e[j] - f[j]) + p0[j] * f[j];
    }
  }
  free(emissivity);
  free(factor);
}

/* a4 -- reference src/ecckd/radiative_transfer_lw.cpp:87-142.
 * Note the small-optical-depth formula differs from a3 (:117-119): factor =
 * max(1 - (1/D)*max(eps,1e-5)/max(od,1e-5/D), 0.5e-5); exp evaluated in both
 * sweeps (:114, :131).  Arrays have row stride `stride`, n points used. */
void orc_radiative_transfer_lw_bb(int nlay, size_t nwav, size_t stride,
                                  const double* planck, const double* spectral_od,
                                  const double* grey_od,
                                  const double* surf_emissivity,
                                  const double* surf_planck, double* flux_dn,
                                  double* flux_up) {
  static const double THRESHOLD_EMISSIVITY = 1.0e-5;
  double* flux = (double*)malloc(nwav * sizeof(double));
  for (size_t j = 0; j < nwav; ++j) flux[j] = 0.0;
  flux_dn[0] = 0.0;
  for (int l = 0; l < nlay; ++l) {
    const double* sod = spectral_od + (size_t)l * stride;
    const double* p0 = planck + (size_t)l * stride;
    const double* p1 = planck + (size_t)(l + 1) * stride;
    double s = 0.0;
    for (size_t j = 0; j < nwav; ++j) {
      double od = sod[j] + grey_od[l];
      double e = 1.0 - exp(-ORC_LW_DIFFUSIVITY * od);
      double f = fmax(1.0 - (1.0 / ORC_LW_DIFFUSIVITY) * fmax(e, THRESHOLD_EMISSIVITY) /
                                fmax(od, THRESHOLD_EMISSIVITY / ORC_LW_DIFFUSIVITY),
                      0.5 * THRESHOLD_EMISSIVITY);
      flux[j] = flux[j] * (1.0 - e) + p0[j] * (e - f) + p1[j] * f;
      s += flux[j];
    }
    flux_dn[l + 1] = s;
  }
  {
    double s = 0.0;
    for (size_t j = 0; j < nwav; ++j) {
      flux[j] = surf_planck[j] * surf_emissivity[j] + (1.0 - surf_emissivity[j]) * flux[j];
      s += flux[j];
    }
    flux_up[nlay] = s;
  }
  for (int l = nlay - 1; l >= 0; --l) {
    const double* sod = spectral_od + (size_t)l * stride;
    const double* p0 = planck + (size_t)l * stride;
    const double* p1 = planck + (size_t)(l + 1) * stride;
    double s = 0.0;
    for (size_t j = 0; j < nwav; ++j) {
      double od = sod[j] + grey_od[l];
      double e = 1.0 - exp(-ORC_LW_DIFFUSIVITY * od);
      double f = fmax(1.0 - (1.0 / ORC_LW_DIFFUSIVITY) * fmax(e, THRESHOLD_EMISSIVITY) /
                                fmax(od, THRESHOLD_EMISSIVITY / ORC_LW_DIFFUSIVITY),
                      0.5 * THRESHOLD_EMISSIVITY);
      flux[j] = flux[j] * (1.0 - e) + p1[j] * (e - f) + p0[j] * f;
      s += flux[j];
    }
    flux_up[l] = s;
  }
  free(flux);
}

/* a5 -- reference src/ecckd/radiative_transfer_sw.cpp:26-43 */
void orc_radiative_transfer_direct_sw(int nlay, size_t nwav, double cos_sza,
                                      const double* ssi, const double* od,
                                      double* flux_dn) {
  const double minus_sec_sza = -1.0 / cos_sza;
  for (size_t j = 0; j < nwav; ++j) flux_dn[j] = cos_sza * ssi[j];
  for (int l = 0; l < nlay; ++l) {
    const double* d0 = flux_dn + (size_t)l * nwav;
    double* d1 = flux_dn + (size_t)(l + 1) * nwav;
    const double* o = od + (size_t)l * nwav;
    for (size_t j = 0; j < nwav; ++j) d1[j] = d0[j] * exp(minus_sec_sza * o[j]);
  }
}

/* a5 -- reference src/ecckd/radiative_transfer_sw.cpp:49-77 (two-stream sec = 2, :66) */
void orc_radiative_transfer_norayleigh_sw(int nlay, size_t nwav, double cos_sza,
                                          const double* ssi, const double* od,
                                          const double* albedo, double* flux_dn,
                                          double* flux_up) {
  static const double minus_sec_tsza = -2.0;
  orc_radiative_transfer_direct_sw(nlay, nwav, cos_sza, ssi, od, flux_dn);
  {
    const double* dn = flux_dn + (size_t)nlay * nwav;
    double* up = flux_up + (size_t)nlay * nwav;
    for (size_t j = 0; j < nwav; ++j) up[j] = dn[j] * albedo[j];
  }
  for (int l = nlay - 1; l >= 0; --l) {
    const double* u1 = flux_up + (size_t)(l + 1) * nwav;
    double* u0 = flux_up + (size_t)l * nwav;
    const double* o = od + (size_t)l * nwav;
    for (size_t j = 0; j < nwav; ++j) u0[j] = u1[j] * exp(minus_sec_tsza * o[j]);
  }
}

/* a5 -- reference src/ecckd/radiative_transfer_sw.cpp:118-141 */
void orc_radiative_transfer_direct_sw_bb(int nlay, size_t nwav, size_t stride,
                                         double cos_sza, const double* ssi,
                                         const double* spectral_od,
                                         const double* grey_od, double* flux_dn) {
  const double minus_sec_sza = -1.0 / cos_sza;
  double* flux = (double*)malloc(nwav * sizeof(double));
  double s = 0.0;
  for (size_t j = 0; j < nwav; ++j) s += ssi[j];
  flux_dn[0] = cos_sza * s;
  for (size_t j = 0; j < nwav; ++j) flux[j] = cos_sza * ssi[j];
  for (int l = 0; l < nlay; ++l) {
    const double* sod = spectral_od + (size_t)l * stride;
    s = 0.0;
    for (size_t j = 0; j < nwav; ++j) {
      flux[j] = flux[j] * exp(minus_sec_sza * (sod[j] + grey_od[l]));
      s += flux[j];
    }
    flux_dn[l + 1] = s;
  }
  free(flux);
}

/* a5 -- reference src/ecckd/radiative_transfer_sw.cpp:147-184 */
void orc_radiative_transfer_norayleigh_sw_bb(int nlay, size_t nwav, size_t stride,
                                             double cos_sza, const double* ssi,
                                             const double* spectral_od,
                                             const double* grey_od, double albedo,
                                             double* flux_dn, double* flux_up) {
  const double minus_sec_sza = -1.0 / cos_sza;
  static const double minus_sec_tsza = -2.0;
  double* flux = (double*)malloc(nwav * sizeof(double));
  double s = 0.0;
  for (size_t j = 0; j < nwav; ++j) s += ssi[j];
  flux_dn[0] = cos_sza * s;
  for (size_t j = 0; j < nwav; ++j) flux[j] = cos_sza * ssi[j];
  for (int l = 0; l < nlay; ++l) {
    const double* sod = spectral_od + (size_t)l * stride;
    s = 0.0;
    for (size_t j = 0; j < nwav; ++j) {
      flux[j] = flux[j] * exp(minus_sec_sza * (sod[j] + grey_od[l]));
      s += flux[j];
    }
    flux_dn[l + 1] = s;
  }
  s = 0.0;
  for (size_t j = 0; j < nwav; ++j) {
    flux[j] *= albedo;
    s += flux[j];
  }
  flux_up[nlay] = s;
  for (int l = nlay - 1; l >= 0; --l) {
    const double* sod = spectral_od + (size_t)l * stride;
    s = 0.0;
    for (size_t j = 0; j < nwav; ++j) {
      flux[j] = flux[j] * exp(minus_sec_tsza * (sod[j] + grey_od[l]));
      s += flux[j];
    }
    flux_up[l] = s;
  }
  free(flux);
}

/* a6 -- reference src/ecckd/heating_rate.h:30-50; flux_up == NULL is the
 * "empty" direct-only shortwave case (:41-45). */
void orc_heating_rate(int nlay, size_t nwav, const double* pressure_hl,
                      const double* flux_dn, const double* flux_up, double* hr) {
  for (int l = 0; l < nlay; ++l) {
    const double conversion =
        -(ORC_ACCEL_GRAVITY / ORC_SPECIFIC_HEAT_AIR) / (pressure_hl[l + 1] - pressure_hl[l]);
    const double* d0 = flux_dn + (size_t)l * nwav;
    const double* d1 = flux_dn + (size_t)(l + 1) * nwav;
    double* h = hr + (size_t)l * nwav;
    if (!flux_up) {
      for (size_t j = 0; j < nwav; ++j) h[j] = conversion * (d1[j] - d0[j]);
    } else {
      const double* u0 = flux_up + (size_t)l * nwav;
      const double* u1 = flux_up + (size_t)(l + 1) * nwav;
      for (size_t j = 0; j < nwav; ++j) h[j] = conversion * (d1[j] - d0[j] - u1[j] + u0[j]);
    }
  }
}

/* a6 -- reference src/ecckd/heating_rate.h:55-72 */
void orc_heating_rate_single(int nlay, const double* pressure_hl,
                             const double* flux_dn, const double* flux_up,
                             double* hr) {
  for (int l = 0; l < nlay; ++l) {
    const double conv = -((ORC_ACCEL_GRAVITY / ORC_SPECIFIC_HEAT_AIR) /
                          (pressure_hl[l + 1] - pressure_hl[l]));
    if (!flux_up) {
      hr[l] = conv * (flux_dn[l + 1] - flux_dn[l]);
    } else {
      hr[l] = conv * (flux_dn[l + 1] - flux_dn[l] - flux_up[l + 1] + flux_up[l]);
    }
  }
}
