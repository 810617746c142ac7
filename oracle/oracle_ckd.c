/* oracle_ckd.c - CPU restatement (TEST INFRASTRUCTURE, see ecckd_oracle.h) of the
 * forward model of optimize_lut: CKD look-up-table interpolation, Planck LUT
 * interpolation and the per-profile longwave cost function.  Forward only: the
 * gradient is checked by finite differences of this function (the reference
 * obtains it from Adept's reverse-mode tape, solve_adept.cpp:91,201-203). */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ecckd_oracle.h"

#define ORC_CONC_NONE 0
#define ORC_CONC_LINEAR 1
#define ORC_CONC_LUT 2
#define ORC_CONC_RELATIVE_LINEAR 3

/* a17 -- reference src/ecckd/ckd_model.cpp:925-1102 (linear interpolation branch,
 * logarithmic_interpolation = false, ckd_model.h:359).  od[ncol][nlay][ng] is
 * OVERWRITTEN with this gas's optical depth.  vmr_fl may be NULL ("empty").
 * molar_abs is (nt,np,ng) or, for LUT gases, (nconc,nt,np,ng). */
int orc_ckd_optical_depth(int ng, int nt, int np, const double* log_pressure,
                          const double* temperature /* [nt][np] */, int conc_dependence,
                          int nconc, const double* vmr_lut, double reference_vmr,
                          const double* molar_abs, int ncol, int nlay,
                          const double* pressure_hl /* [ncol][nlay+1] */,
                          const double* temperature_fl /* [ncol][nlay] */,
                          const double* vmr_fl /* [ncol][nlay] or NULL */, double* od) {
  const double log_p_0 = log_pressure[0];
  const double d_log_p = log_pressure[1] - log_pressure[0];
  const double d_t = temperature[1 * np + 0] - temperature[0];
  const double global_weight = 1.0 / (ORC_ACCEL_GRAVITY * 0.001 * ORC_MOLAR_MASS_DRY_AIR);
  for (int icol = 0; icol < ncol; ++icol) {
    for (int ip = 0; ip < nlay; ++ip) {
      const double p1 = pressure_hl[icol * (nlay + 1) + ip + 1], p0 = pressure_hl[icol * (nlay + 1) + ip];
      double log_pressure_fl = log(0.5 * (p1 + p0));
      double pindex0 = (log_pressure_fl - log_p_0) / d_log_p;
      pindex0 = fmax(0.0, fmin(pindex0, np - 1.0001));
      int ip0 = (int)pindex0;
      double pweight1 = pindex0 - ip0, pweight0 = 1.0 - pweight1;
      double t_0 = pweight0 * temperature[ip0] + pweight1 * temperature[ip0 + 1];
      double tindex0 = (temperature_fl[icol * nlay + ip] - t_0) / d_t;
      tindex0 = fmax(0.0, fmin(tindex0, nt - 1.0001));
      int it0 = (int)tindex0;
      double tweight1 = tindex0 - it0, tweight0 = 1.0 - tweight1;
      double simple_weight = global_weight * (p1 - p0);
      double weight = 0.0;
      int no_vmr_provided = 1;
      if (vmr_fl) {
        if (conc_dependence == ORC_CONC_RELATIVE_LINEAR) weight = simple_weight * (vmr_fl[icol * nlay + ip] - reference_vmr);
        else weight = simple_weight * vmr_fl[icol * nlay + ip];
        no_vmr_provided = 0;
      }
      double* out = od + ((size_t)icol * nlay + ip) * ng;
#define K3(it, ipp) (molar_abs + ((size_t)(it) * np + (ipp)) * ng)
#define K4(ic, it, ipp) (molar_abs + (((size_t)(ic) * nt + (it)) * np + (ipp)) * ng)
      if (conc_dependence == ORC_CONC_LUT) {
        if (no_vmr_provided) return 1;
        double log_conc = log(vmr_fl[icol * nlay + ip]);
        double d_log_c = log(vmr_lut[1] / vmr_lut[0]);
        double cindex0 = (log_conc - log(vmr_lut[0])) / d_log_c;
        cindex0 = fmax(0.0, fmin(cindex0, nconc - 1.0001));
        int ic0 = (int)cindex0;
        double cweight1 = cindex0 - ic0, cweight0 = 1.0 - cweight1;
        for (int g = 0; g < ng; ++g) {
          out[g] = weight *
                   (cweight0 * (tweight0 * (pweight0 * K4(ic0, it0, ip0)[g] + pweight1 * K4(ic0, it0, ip0 + 1)[g]) +
                                tweight1 * (pweight0 * K4(ic0, it0 + 1, ip0)[g] + pweight1 * K4(ic0, it0 + 1, ip0 + 1)[g])) +
                    cweight1 * (tweight0 * (pweight0 * K4(ic0 + 1, it0, ip0)[g] + pweight1 * K4(ic0 + 1, it0, ip0 + 1)[g]) +
                                tweight1 * (pweight0 * K4(ic0 + 1, it0 + 1, ip0)[g] + pweight1 * K4(ic0 + 1, it0 + 1, ip0 + 1)[g])));
        }
      } else {
        double wgt;
        if (conc_dependence == ORC_CONC_NONE) wgt = simple_weight;
        else {
          if (no_vmr_provided) return 1;
          wgt = weight;
        }
        for (int g = 0; g < ng; ++g) {
          out[g] = wgt * (tweight0 * (pweight0 * K3(it0, ip0)[g] + pweight1 * K3(it0, ip0 + 1)[g]) +
                          tweight1 * (pweight0 * K3(it0 + 1, ip0)[g] + pweight1 * K3(it0 + 1, ip0 + 1)[g]));
        }
      }
#undef K3
#undef K4
    }
  }
  return 0;
}

/* a21 -- reference src/ecckd/ckd_model.cpp:1107-1145: Planck LUT interpolation,
 * linear to zero below the first LUT temperature (:1139-1142). */
void orc_ckd_planck(int ntp, const double* temperature_planck, const double* planck_function /* [ntp][ng] */,
                    int ng, int n, const double* temperature, double* planck /* [n][ng] */) {
  const double d_t = temperature_planck[1] - temperature_planck[0];
  const double t0 = temperature_planck[0];
  for (int it = 0; it < n; ++it) {
    double tindex0 = (temperature[it] - t0) / d_t;
    if (tindex0 >= 0) {
      int it0 = (int)tindex0;
      if (it0 > ntp - 2) it0 = ntp - 2;
      double tweight1 = tindex0 - it0, tweight0 = 1.0 - tweight1;
      for (int g = 0; g < ng; ++g)
        planck[(size_t)it * ng + g] = tweight0 * planck_function[(size_t)it0 * ng + g] +
                                      tweight1 * planck_function[(size_t)(it0 + 1) * ng + g];
    } else {
      for (int g = 0; g < ng; ++g) planck[(size_t)it * ng + g] = (temperature[it] / t0) * planck_function[g];
    }
  }
}

/* a18 -- reference src/ecckd/calc_cost_function_lw.cpp:116-232, forward value.
 * Arrays are (level, g) / (level, band) row-major.  band_mapping[ng] = g -> band
 * (the LW call passes lbl1.iband_per_g, solve_adept.cpp:170).  relative_ckd_flux
 * arrays (:162-165) may be NULL. */
double orc_calc_cost_function_ckd_lw(int nlay, int ng, int nband, const double* pressure_hl,
                                     const double* planck_hl, const double* surf_emiss_orig,
                                     const double* surf_planck, const double* optical_depth,
                                     const double* flux_dn, const double* flux_up, const double* hr,
                                     const double* spectral_flux_dn_surf,
                                     const double* spectral_flux_up_toa, double flux_weight,
                                     double flux_profile_weight, double broadband_weight,
                                     double spectral_boundary_weight, const double* layer_weight,
                                     const double* relative_ckd_flux_dn,
                                     const double* relative_ckd_flux_up, const int* band_mapping) {
  static const double hr_weight = 3600.0 * 24.0;
  const int nhl = nlay + 1;
  double* fdn_orig = (double*)malloc((size_t)nhl * ng * sizeof(double));
  double* fup_orig = (double*)malloc((size_t)nhl * ng * sizeof(double));
  double* semis = (double*)malloc((size_t)ng * sizeof(double));
  for (int g = 0; g < ng; ++g) semis[g] = surf_emiss_orig[band_mapping[g]];
  orc_radiative_transfer_lw(nlay, (size_t)ng, planck_hl, optical_depth, semis, surf_planck, fdn_orig, fup_orig);
  if (relative_ckd_flux_dn) {
    for (int i = 0; i < nhl * ng; ++i) {
      fdn_orig[i] -= relative_ckd_flux_dn[i];
      fup_orig[i] -= relative_ckd_flux_up[i];
    }
  }
  double* fdn = (double*)calloc((size_t)nhl * nband, sizeof(double));
  double* fup = (double*)calloc((size_t)nhl * nband, sizeof(double));
  for (int b = 0; b < nband; ++b)
    for (int i = 0; i < nhl; ++i) {
      double sd = 0.0, su = 0.0;
      for (int g = 0; g < ng; ++g)
        if (band_mapping[g] == b) {
          sd += fdn_orig[i * ng + g];
          su += fup_orig[i * ng + g];
        }
      fdn[i * nband + b] = sd;
      fup[i * nband + b] = su;
    }
  double* hrf = (double*)malloc((size_t)nlay * nband * sizeof(double));
  orc_heating_rate(nlay, (size_t)nband, pressure_hl, fdn, fup, hrf);

  double cost_fn = 0.0;
  for (int b = 0; b < nband; ++b) {
    double s = 0.0;
    for (int l = 0; l < nlay; ++l) {
      double d = hrf[l * nband + b] - hr[l * nband + b];
      s += layer_weight[l] * d * d;
    }
    double ds = fdn[nlay * nband + b] - flux_dn[nlay * nband + b];
    double dt = fup[b] - flux_up[b];
    cost_fn += hr_weight * hr_weight * s + flux_weight * (ds * ds + dt * dt);
    if (flux_profile_weight > 0.0) {
      double sp = 0.0;
      for (int i = 1; i < nlay; ++i) {
        double iw = flux_profile_weight * 0.5 * (layer_weight[i - 1] + layer_weight[i]);
        double dd = fdn[i * nband + b] - flux_dn[i * nband + b];
        double du = fup[i * nband + b] - flux_up[i * nband + b];
        sp += iw * (dd * dd + du * du);
      }
      cost_fn += sp;
    }
  }
  {
    double sbb = 0.0;
    for (int l = 0; l < nlay; ++l) {
      double r = 0.0;
      for (int b = 0; b < nband; ++b) r += hrf[l * nband + b] - hr[l * nband + b];
      sbb += layer_weight[l] * (r * r);
    }
    double rs = 0.0, rt = 0.0;
    for (int b = 0; b < nband; ++b) {
      rs += fdn[nlay * nband + b] - flux_dn[nlay * nband + b];
      rt += fup[b] - flux_up[b];
    }
    cost_fn = (cost_fn * (1.0 - broadband_weight)) / nband + broadband_weight * hr_weight * hr_weight * sbb +
              broadband_weight * flux_weight * (rs * rs + rt * rt);
  }
  if (flux_profile_weight > 0.0) {
    double sp = 0.0;
    for (int i = 1; i < nlay; ++i) {
      double iw = flux_profile_weight * 0.5 * (layer_weight[i - 1] + layer_weight[i]);
      double ed = 0.0, eu = 0.0;
      for (int b = 0; b < nband; ++b) {
        ed += fdn[i * nband + b] - flux_dn[i * nband + b];
        eu += fup[i * nband + b] - flux_up[i * nband + b];
      }
      sp += iw * (ed * ed + eu * eu);
    }
    cost_fn += broadband_weight * sp;
  }
  if (spectral_boundary_weight > 0.0 && spectral_flux_dn_surf && spectral_flux_up_toa) {
    double s = 0.0;
    for (int g = 0; g < ng; ++g) {
      double a = fdn_orig[nlay * ng + g] - spectral_flux_dn_surf[g];
      double b = fup_orig[g] - spectral_flux_up_toa[g];
      s += a * a + b * b;
    }
    cost_fn += spectral_boundary_weight * s;
  }
  free(fdn_orig); free(fup_orig); free(semis); free(fdn); free(fup); free(hrf);
  return cost_fn;
}

/* a18 (SW) -- reference src/ecckd/calc_cost_function_sw.cpp:116-277, forward value.
 * albedo[nband] is the effective spectral albedo per band; heating rate from the direct beam
 * only (:197); TOA upwelling error weighted by 20 (:214); the broadband terms are applied only
 * if broadband_weight > 0 (:243) and the upwelling ones only if all(albedo > 0) (:252, :264);
 * spectral_boundary_weights[ng] multiplies the squared surface-down error per g (:271-274). */
double orc_calc_cost_function_ckd_sw(int nlay, int ng, int nband, double cos_sza, const double* pressure_hl,
                                     const double* ssi, const double* albedo, const double* optical_depth,
                                     const double* flux_dn, const double* flux_up, const double* hr,
                                     const double* spectral_flux_dn_surf, double flux_weight,
                                     double flux_profile_weight, double broadband_weight,
                                     const double* spectral_boundary_weights, const double* layer_weight,
                                     const double* relative_ckd_flux_dn, const double* relative_ckd_flux_up,
                                     const int* band_mapping) {
  static const double hr_weight = 3600.0 * 24.0;
  const int nhl = nlay + 1;
  double* fdn_orig = (double*)malloc((size_t)nhl * ng * sizeof(double));
  double* fup_orig = (double*)calloc((size_t)nhl * ng, sizeof(double));
  int all_nonpos = 1, all_pos = 1;
  for (int b = 0; b < nband; ++b) {
    if (albedo[b] > 0.0) all_nonpos = 0;
    if (!(albedo[b] > 0.0)) all_pos = 0;
  }
  if (all_nonpos) {
    orc_radiative_transfer_direct_sw(nlay, (size_t)ng, cos_sza, ssi, optical_depth, fdn_orig);
  } else {
    double* alb_g = (double*)malloc((size_t)ng * sizeof(double));
    for (int g = 0; g < ng; ++g) alb_g[g] = albedo[band_mapping[g]];
    orc_radiative_transfer_norayleigh_sw(nlay, (size_t)ng, cos_sza, ssi, optical_depth, alb_g, fdn_orig, fup_orig);
    free(alb_g);
  }
  if (relative_ckd_flux_dn) { /* calc_cost_function_sw.cpp:160-163 */
    for (size_t q = 0; q < (size_t)nhl * ng; ++q) {
      fdn_orig[q] -= relative_ckd_flux_dn[q];
      fup_orig[q] -= relative_ckd_flux_up[q];
    }
  }
  double* fdn = (double*)calloc((size_t)nhl * nband, sizeof(double));
  double* fup = (double*)calloc((size_t)nhl * nband, sizeof(double));
  for (int b = 0; b < nband; ++b)
    for (int i = 0; i < nhl; ++i) {
      double sd = 0.0, su = 0.0;
      for (int g = 0; g < ng; ++g)
        if (band_mapping[g] == b) { sd += fdn_orig[i * ng + g]; su += fup_orig[i * ng + g]; }
      fdn[i * nband + b] = sd;
      fup[i * nband + b] = su;
    }
  double* hrf = (double*)malloc((size_t)nlay * nband * sizeof(double));
  orc_heating_rate(nlay, (size_t)nband, pressure_hl, fdn, NULL, hrf);
  double cost_fn = 0.0;
  for (int b = 0; b < nband; ++b) {
    double s = 0.0;
    for (int l = 0; l < nlay; ++l) {
      double d = hrf[l * nband + b] - hr[l * nband + b];
      s += layer_weight[l] * d * d;
    }
    double ds = fdn[nlay * nband + b] - flux_dn[nlay * nband + b];
    double dt = fup[b] - flux_up[b];
    double local = hr_weight * hr_weight * s + flux_weight * (ds * ds + 20.0 * dt * dt);
    if (flux_profile_weight > 0.0) {
      for (int i = 1; i < nlay; ++i) {
        double iw = flux_profile_weight * 0.5 * (layer_weight[i - 1] + layer_weight[i]);
        double dd = fdn[i * nband + b] - flux_dn[i * nband + b];
        double du = fup[i * nband + b] - flux_up[i * nband + b];
        local += iw * (dd * dd + du * du);
      }
    }
    cost_fn += local;
  }
  if (broadband_weight > 0.0) {
    double sbb = 0.0;
    for (int l = 0; l < nlay; ++l) {
      double r = 0.0;
      for (int b = 0; b < nband; ++b) r += hrf[l * nband + b] - hr[l * nband + b];
      sbb += layer_weight[l] * (r * r);
    }
    cost_fn = (cost_fn * (1.0 - broadband_weight)) / nband + broadband_weight * hr_weight * hr_weight * sbb;
    double rs = 0.0, rt = 0.0;
    for (int b = 0; b < nband; ++b) {
      rs += fdn[nlay * nband + b] - flux_dn[nlay * nband + b];
      rt += fup[b] - flux_up[b];
    }
    cost_fn += broadband_weight * flux_weight * (rs * rs);
    if (all_pos) cost_fn += broadband_weight * flux_weight * (rt * rt);
    if (flux_profile_weight > 0.0) {
      double spd = 0.0, spu = 0.0;
      for (int i = 1; i < nlay; ++i) {
        double iw = flux_profile_weight * 0.5 * (layer_weight[i - 1] + layer_weight[i]);
        double ed = 0.0, eu = 0.0;
        for (int b = 0; b < nband; ++b) {
          ed += fdn[i * nband + b] - flux_dn[i * nband + b];
          eu += fup[i * nband + b] - flux_up[i * nband + b];
        }
        spd += iw * (ed * ed);
        spu += iw * (eu * eu);
      }
      cost_fn += broadband_weight * spd;
      if (all_pos) cost_fn += broadband_weight * spu;
    }
  }
  if (spectral_boundary_weights && spectral_flux_dn_surf) {
    double s = 0.0;
    for (int g = 0; g < ng; ++g) {
      double a = fdn_orig[nlay * ng + g] - spectral_flux_dn_surf[g];
      s += spectral_boundary_weights[g] * a * a;
    }
    cost_fn += s;
  }
  free(fdn_orig); free(fup_orig); free(fdn); free(fup); free(hrf);
  return cost_fn;
}
