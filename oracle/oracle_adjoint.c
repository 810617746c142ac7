/* oracle_adjoint.c - reverse mode BY HAND of the longwave optimize_lut forward model (TEST INFRASTRUCTURE, see
 * ecckd_oracle.h).  The reference obtains dJ/dx from Adept's tape (solve_adept.cpp:91, :201-203); Adept is not available
 * here, so the oracle differentiates its own restatement of calc_cost_function_ckd_lw (calc_cost_function_lw.cpp:116-232)
 * and of CkdModel::calc_optical_depth (ckd_model.cpp:925-1102) statement by statement, in the reverse order of the
 * forward code in oracle_ckd.c / oracle_rt.c.  It shares no code with the device adjoint (k_opt_forward_adjoint /
 * k_opt_gradient); tests compare the two gradients element by element and run the SAME minimizer over either. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ecckd_oracle.h"

/* Cost of one profile as orc_calc_cost_function_ckd_lw, and d_od[nlay][ng] += d cost / d optical_depth. */
double orc_calc_cost_function_ckd_lw_ad(int nlay, int ng, int nband, const double* pressure_hl, const double* planck_hl,
                                        const double* surf_emiss_orig, const double* surf_planck,
                                        const double* optical_depth, const double* flux_dn, const double* flux_up,
                                        const double* hr, const double* spectral_flux_dn_surf,
                                        const double* spectral_flux_up_toa, double flux_weight,
                                        double flux_profile_weight, double broadband_weight,
                                        double spectral_boundary_weight, const double* layer_weight,
                                        const double* relative_ckd_flux_dn, const double* relative_ckd_flux_up,
                                        const int* band_mapping, double* d_od) {
  static const double hr_weight = 3600.0 * 24.0;
  const double W2 = hr_weight * hr_weight;
  const double D = ORC_LW_DIFFUSIVITY;
  const int nhl = nlay + 1;
  const size_t nlg = (size_t)nlay * ng, nhg = (size_t)nhl * ng, nhb = (size_t)nhl * nband;
  double* eps = (double*)malloc(nlg * sizeof(double));
  double* fac = (double*)malloc(nlg * sizeof(double));
  double* dn = (double*)malloc(nhg * sizeof(double));   /* per-g fluxes BEFORE the relative-to subtraction */
  double* up = (double*)malloc(nhg * sizeof(double));
  double* semis = (double*)malloc((size_t)ng * sizeof(double));
  double* fdn = (double*)calloc(nhb, sizeof(double));
  double* fup = (double*)calloc(nhb, sizeof(double));
  double* hrf = (double*)malloc((size_t)nlay * nband * sizeof(double));
  double* a_fdn = (double*)calloc(nhb, sizeof(double));  /* adjoints of the band fluxes */
  double* a_fup = (double*)calloc(nhb, sizeof(double));
  double* a_dn = (double*)calloc(nhg, sizeof(double));   /* adjoints of the per-g fluxes */
  double* a_up = (double*)calloc(nhg, sizeof(double));
  double* conv = (double*)malloc((size_t)nlay * sizeof(double));

  /* ---------------- forward, keeping what the reverse pass needs ---------------- */
  for (int g = 0; g < ng; ++g) semis[g] = surf_emiss_orig[band_mapping[g]];
  for (size_t i = 0; i < nlg; ++i) {                       /* radiative_transfer_lw.cpp:41-43 */
    const double e = 1.0 - exp(-D * optical_depth[i]);
    eps[i] = e;
    fac[i] = (e > 1.0e-5) ? 1.0 - e * (1.0 / D) / optical_depth[i] : 0.5 * e;
  }
  for (int g = 0; g < ng; ++g) dn[g] = 0.0;
  for (int l = 0; l < nlay; ++l)
    for (int g = 0; g < ng; ++g) {
      const size_t i = (size_t)l * ng + g;
      dn[i + ng] = dn[i] * (1.0 - eps[i]) + planck_hl[i] * (eps[i] - fac[i]) + planck_hl[i + ng] * fac[i];
    }
  for (int g = 0; g < ng; ++g)
    up[(size_t)nlay * ng + g] = surf_planck[g] * semis[g] + (1.0 - semis[g]) * dn[(size_t)nlay * ng + g];
  for (int l = nlay - 1; l >= 0; --l)
    for (int g = 0; g < ng; ++g) {
      const size_t i = (size_t)l * ng + g;
      up[i] = up[i + ng] * (1.0 - eps[i]) + planck_hl[i + ng] * (eps[i] - fac[i]) + planck_hl[i] * fac[i];
    }
  for (int i = 0; i < nhl; ++i)
    for (int g = 0; g < ng; ++g) {
      const int b = band_mapping[g];
      double d = dn[(size_t)i * ng + g], u = up[(size_t)i * ng + g];
      if (relative_ckd_flux_dn) { d -= relative_ckd_flux_dn[(size_t)i * ng + g]; u -= relative_ckd_flux_up[(size_t)i * ng + g]; }
      fdn[(size_t)i * nband + b] += d;
      fup[(size_t)i * nband + b] += u;
    }
  for (int l = 0; l < nlay; ++l) {
    conv[l] = -(ORC_ACCEL_GRAVITY / ORC_SPECIFIC_HEAT_AIR) / (pressure_hl[l + 1] - pressure_hl[l]);
    for (int b = 0; b < nband; ++b)
      hrf[(size_t)l * nband + b] = conv[l] * (fdn[(size_t)(l + 1) * nband + b] - fdn[(size_t)l * nband + b] -
                                              fup[(size_t)(l + 1) * nband + b] + fup[(size_t)l * nband + b]);
  }

  /* ---------------- cost (calc_cost_function_lw.cpp:186-229) and the adjoints of the band fluxes ---------------- */
  const double alpha = (1.0 - broadband_weight) / nband;   /* weight of the per-band terms in the final mix (:209-221) */
  double cost_bands = 0.0;
  double rs = 0.0, rt = 0.0, sbb = 0.0;
  for (int b = 0; b < nband; ++b) {
    rs += fdn[(size_t)nlay * nband + b] - flux_dn[(size_t)nlay * nband + b];
    rt += fup[b] - flux_up[b];
  }
  for (int l = 0; l < nlay; ++l) {
    double r = 0.0;
    for (int b = 0; b < nband; ++b) r += hrf[(size_t)l * nband + b] - hr[(size_t)l * nband + b];
    sbb += layer_weight[l] * (r * r);
    for (int b = 0; b < nband; ++b) {
      const double d = hrf[(size_t)l * nband + b] - hr[(size_t)l * nband + b];
      /* d cost / d hrf(l,b): per-band term and broadband term */
      const double a_h = alpha * W2 * 2.0 * layer_weight[l] * d + broadband_weight * W2 * 2.0 * layer_weight[l] * r;
      a_fdn[(size_t)(l + 1) * nband + b] += a_h * conv[l];
      a_fdn[(size_t)l * nband + b] -= a_h * conv[l];
      a_fup[(size_t)(l + 1) * nband + b] -= a_h * conv[l];
      a_fup[(size_t)l * nband + b] += a_h * conv[l];
    }
  }
  for (int b = 0; b < nband; ++b) {
    double s = 0.0;
    for (int l = 0; l < nlay; ++l) {
      const double d = hrf[(size_t)l * nband + b] - hr[(size_t)l * nband + b];
      s += layer_weight[l] * d * d;
    }
    const double ds = fdn[(size_t)nlay * nband + b] - flux_dn[(size_t)nlay * nband + b];
    const double dt = fup[b] - flux_up[b];
    cost_bands += W2 * s + flux_weight * (ds * ds + dt * dt);
    a_fdn[(size_t)nlay * nband + b] += alpha * flux_weight * 2.0 * ds + broadband_weight * flux_weight * 2.0 * rs;
    a_fup[b] += alpha * flux_weight * 2.0 * dt + broadband_weight * flux_weight * 2.0 * rt;
    if (flux_profile_weight > 0.0) {
      double sp = 0.0;
      for (int i = 1; i < nlay; ++i) {
        const double iw = flux_profile_weight * 0.5 * (layer_weight[i - 1] + layer_weight[i]);
        const double dd = fdn[(size_t)i * nband + b] - flux_dn[(size_t)i * nband + b];
        const double du = fup[(size_t)i * nband + b] - flux_up[(size_t)i * nband + b];
        sp += iw * (dd * dd + du * du);
        a_fdn[(size_t)i * nband + b] += alpha * iw * 2.0 * dd;
        a_fup[(size_t)i * nband + b] += alpha * iw * 2.0 * du;
      }
      cost_bands += sp;
    }
  }
  double cost = cost_bands * alpha + broadband_weight * W2 * sbb + broadband_weight * flux_weight * (rs * rs + rt * rt);
  if (flux_profile_weight > 0.0) {
    double sp = 0.0;
    for (int i = 1; i < nlay; ++i) {
      const double iw = flux_profile_weight * 0.5 * (layer_weight[i - 1] + layer_weight[i]);
      double ed = 0.0, eu = 0.0;
      for (int b = 0; b < nband; ++b) {
        ed += fdn[(size_t)i * nband + b] - flux_dn[(size_t)i * nband + b];
        eu += fup[(size_t)i * nband + b] - flux_up[(size_t)i * nband + b];
      }
      sp += iw * (ed * ed + eu * eu);
      for (int b = 0; b < nband; ++b) {
        a_fdn[(size_t)i * nband + b] += broadband_weight * iw * 2.0 * ed;
        a_fup[(size_t)i * nband + b] += broadband_weight * iw * 2.0 * eu;
      }
    }
    cost += broadband_weight * sp;
  }
  /* band sums -> every g of the band receives the band's adjoint (:171-184 reversed) */
  for (int i = 0; i < nhl; ++i)
    for (int g = 0; g < ng; ++g) {
      a_dn[(size_t)i * ng + g] = a_fdn[(size_t)i * nband + band_mapping[g]];
      a_up[(size_t)i * ng + g] = a_fup[(size_t)i * nband + band_mapping[g]];
    }
  if (spectral_boundary_weight > 0.0 && spectral_flux_dn_surf && spectral_flux_up_toa) {   /* :223-229 */
    double s = 0.0;
    for (int g = 0; g < ng; ++g) {
      double a = dn[(size_t)nlay * ng + g] - spectral_flux_dn_surf[g];
      double b = up[g] - spectral_flux_up_toa[g];
      if (relative_ckd_flux_dn) { a -= relative_ckd_flux_dn[(size_t)nlay * ng + g]; b -= relative_ckd_flux_up[g]; }
      s += a * a + b * b;
      a_dn[(size_t)nlay * ng + g] += spectral_boundary_weight * 2.0 * a;
      a_up[g] += spectral_boundary_weight * 2.0 * b;
    }
    cost += spectral_boundary_weight * s;
  }

  /* ---------------- reverse of radiative_transfer_lw (radiative_transfer_lw.cpp:45-59 backwards) ---------------- */
  double* a_eps = (double*)calloc(nlg, sizeof(double));
  double* a_fac = (double*)calloc(nlg, sizeof(double));
  for (int g = 0; g < ng; ++g) {
    for (int l = 0; l < nlay; ++l) {                     /* reverse of the up sweep */
      const size_t i = (size_t)l * ng + g;
      const double au = a_up[i];
      a_up[i + ng] += au * (1.0 - eps[i]);
      a_eps[i] += au * (-up[i + ng] + planck_hl[i + ng]);
      a_fac[i] += au * (-planck_hl[i + ng] + planck_hl[i]);
    }
    /* surface: up(nlay) = B_s emis + (1 - emis) dn(nlay) */
    a_dn[(size_t)nlay * ng + g] += a_up[(size_t)nlay * ng + g] * (1.0 - semis[g]);
    for (int l = nlay - 1; l >= 0; --l) {                /* reverse of the down sweep */
      const size_t i = (size_t)l * ng + g;
      const double ad = a_dn[i + ng];
      a_dn[i] += ad * (1.0 - eps[i]);
      a_eps[i] += ad * (-dn[i] + planck_hl[i]);
      a_fac[i] += ad * (-planck_hl[i] + planck_hl[i + ng]);
    }
  }
  for (size_t i = 0; i < nlg; ++i) {                     /* emissivity and factor (:41-43) */
    const double tau = optical_depth[i];
    double a_e = a_eps[i];
    double a_tau = 0.0;
    if (eps[i] > 1.0e-5) {
      a_e += a_fac[i] * (-(1.0 / D) / tau);
      a_tau += a_fac[i] * (eps[i] * (1.0 / D) / (tau * tau));
    } else {
      a_e += 0.5 * a_fac[i];
    }
    a_tau += a_e * D * (1.0 - eps[i]);                   /* d eps / d tau = D exp(-D tau) */
    d_od[i] += a_tau;
  }
  free(eps); free(fac); free(dn); free(up); free(semis); free(fdn); free(fup); free(hrf); free(a_fdn); free(a_fup);
  free(a_dn); free(a_up); free(conv); free(a_eps); free(a_fac);
  return cost;
}

#define ORC_CONC_NONE 0
#define ORC_CONC_LINEAR 1
#define ORC_CONC_LUT 2
#define ORC_CONC_RELATIVE_LINEAR 3

/* Transpose of orc_ckd_optical_depth (CkdModel::calc_optical_depth, ckd_model.cpp:925-1102): the optical depth is linear
 * in the gas's table, od = sum_nodes w * molar_abs(node); d_molar_abs(node) += w * d_od.  Same index arithmetic as the
 * forward routine. */
int orc_ckd_optical_depth_ad(int ng, int nt, int np, const double* log_pressure, const double* temperature,
                             int conc_dependence, int nconc, const double* vmr_lut, double reference_vmr, int ncol, int nlay,
                             const double* pressure_hl, const double* temperature_fl, const double* vmr_fl,
                             const double* d_od /* [ncol][nlay][ng] */, double* d_molar_abs) {
  const double log_p_0 = log_pressure[0];
  const double d_log_p = log_pressure[1] - log_pressure[0];
  const double d_t = temperature[1 * np + 0] - temperature[0];
  const double global_weight = 1.0 / (ORC_ACCEL_GRAVITY * 0.001 * ORC_MOLAR_MASS_DRY_AIR);
  for (int icol = 0; icol < ncol; ++icol)
    for (int ip = 0; ip < nlay; ++ip) {
      const double p1 = pressure_hl[icol * (nlay + 1) + ip + 1], p0 = pressure_hl[icol * (nlay + 1) + ip];
      double pindex0 = (log(0.5 * (p1 + p0)) - log_p_0) / d_log_p;
      pindex0 = fmax(0.0, fmin(pindex0, np - 1.0001));
      const int ip0 = (int)pindex0;
      const double pw1 = pindex0 - ip0, pw0 = 1.0 - pw1;
      const double t_0 = pw0 * temperature[ip0] + pw1 * temperature[ip0 + 1];
      double tindex0 = (temperature_fl[icol * nlay + ip] - t_0) / d_t;
      tindex0 = fmax(0.0, fmin(tindex0, nt - 1.0001));
      const int it0 = (int)tindex0;
      const double tw1 = tindex0 - it0, tw0 = 1.0 - tw1;
      const double simple_weight = global_weight * (p1 - p0);
      double wgt;
      if (conc_dependence == ORC_CONC_NONE) wgt = simple_weight;
      else {
        if (!vmr_fl) return 1;
        wgt = simple_weight * (conc_dependence == ORC_CONC_RELATIVE_LINEAR ? vmr_fl[icol * nlay + ip] - reference_vmr
                                                                          : vmr_fl[icol * nlay + ip]);
      }
      const double* a = d_od + ((size_t)icol * nlay + ip) * ng;
      int ic0 = 0;
      double cw[2] = {1.0, 0.0};
      int ncz = 1;
      if (conc_dependence == ORC_CONC_LUT) {
        const double d_log_c = log(vmr_lut[1] / vmr_lut[0]);
        double cindex0 = (log(vmr_fl[icol * nlay + ip]) - log(vmr_lut[0])) / d_log_c;
        cindex0 = fmax(0.0, fmin(cindex0, nconc - 1.0001));
        ic0 = (int)cindex0;
        cw[1] = cindex0 - ic0;
        cw[0] = 1.0 - cw[1];
        ncz = 2;
      }
      for (int c = 0; c < ncz; ++c)
        for (int dt = 0; dt < 2; ++dt)
          for (int dp = 0; dp < 2; ++dp) {
            const double w = wgt * cw[c] * (dt ? tw1 : tw0) * (dp ? pw1 : pw0);
            double* k = d_molar_abs + ((((size_t)(ic0 + c) * nt) + (it0 + dt)) * np + (ip0 + dp)) * ng;
            for (int g = 0; g < ng; ++g) k[g] += w * a[g];
          }
    }
  return 0;
}

/* Shortwave twin: cost of one profile as orc_calc_cost_function_ckd_sw (calc_cost_function_sw.cpp:116-277) and
 * d_od[nlay][ng] += d cost / d optical_depth, reverse mode by hand of the oracle's OWN forward statements
 * (radiative_transfer_direct_sw / _norayleigh_sw, radiative_transfer_sw.cpp:26-77; the all-albedo <= 0 direct-only branch
 * :145-150; band sums; heating rate from the direct beam :197; per-band terms with the twenty-fold top-of-atmosphere weight
 * :214; the broadband mix only if broadband_weight > 0 :243 and its upwelling parts only if all albedos > 0 :252, :264; the
 * per-g boundary term :271-274).  Shares no code with the device adjoint. */
double orc_calc_cost_function_ckd_sw_ad(int nlay, int ng, int nband, double cos_sza, const double* pressure_hl,
                                        const double* ssi, const double* albedo, const double* optical_depth,
                                        const double* flux_dn, const double* flux_up, const double* hr,
                                        const double* spectral_flux_dn_surf, double flux_weight,
                                        double flux_profile_weight, double broadband_weight,
                                        const double* spectral_boundary_weights, const double* layer_weight,
                                        const double* relative_ckd_flux_dn, const double* relative_ckd_flux_up,
                                        const int* band_mapping, double* d_od) {
  static const double hr_weight = 3600.0 * 24.0;
  const double W2 = hr_weight * hr_weight;
  const int nhl = nlay + 1;
  const size_t nlg = (size_t)nlay * ng, nhg = (size_t)nhl * ng, nhb = (size_t)nhl * nband;
  const double minus_sec_sza = -1.0 / cos_sza;
  double* td = (double*)malloc(nlg * sizeof(double));    /* exp(-tau / mu0) */
  double* tu = (double*)malloc(nlg * sizeof(double));    /* exp(-2 tau) */
  double* dn = (double*)malloc(nhg * sizeof(double));    /* per-g fluxes before the relative-to subtraction */
  double* up = (double*)calloc(nhg, sizeof(double));
  double* alb_g = (double*)malloc((size_t)ng * sizeof(double));
  double* fdn = (double*)calloc(nhb, sizeof(double));
  double* fup = (double*)calloc(nhb, sizeof(double));
  double* hrf = (double*)malloc((size_t)nlay * nband * sizeof(double));
  double* a_fdn = (double*)calloc(nhb, sizeof(double));
  double* a_fup = (double*)calloc(nhb, sizeof(double));
  double* a_dn = (double*)calloc(nhg, sizeof(double));
  double* a_up = (double*)calloc(nhg, sizeof(double));
  double* conv = (double*)malloc((size_t)nlay * sizeof(double));
  int all_nonpos = 1, all_pos = 1;
  for (int b = 0; b < nband; ++b) {
    if (albedo[b] > 0.0) all_nonpos = 0;
    if (!(albedo[b] > 0.0)) all_pos = 0;
  }
  /* ---------------- forward ---------------- */
  for (int g = 0; g < ng; ++g) alb_g[g] = all_nonpos ? 0.0 : albedo[band_mapping[g]];
  for (size_t i = 0; i < nlg; ++i) {
    td[i] = exp(minus_sec_sza * optical_depth[i]);
    tu[i] = exp(-2.0 * optical_depth[i]);
  }
  for (int g = 0; g < ng; ++g) dn[g] = cos_sza * ssi[g];
  for (int l = 0; l < nlay; ++l)
    for (int g = 0; g < ng; ++g) dn[(size_t)(l + 1) * ng + g] = dn[(size_t)l * ng + g] * td[(size_t)l * ng + g];
  if (!all_nonpos) {
    for (int g = 0; g < ng; ++g) up[(size_t)nlay * ng + g] = dn[(size_t)nlay * ng + g] * alb_g[g];
    for (int l = nlay - 1; l >= 0; --l)
      for (int g = 0; g < ng; ++g) up[(size_t)l * ng + g] = up[(size_t)(l + 1) * ng + g] * tu[(size_t)l * ng + g];
  }
  for (int i = 0; i < nhl; ++i)
    for (int g = 0; g < ng; ++g) {
      const int b = band_mapping[g];
      double d = dn[(size_t)i * ng + g], u = up[(size_t)i * ng + g];
      if (relative_ckd_flux_dn) { d -= relative_ckd_flux_dn[(size_t)i * ng + g]; u -= relative_ckd_flux_up[(size_t)i * ng + g]; }
      fdn[(size_t)i * nband + b] += d;
      fup[(size_t)i * nband + b] += u;
    }
  for (int l = 0; l < nlay; ++l) {
    conv[l] = -(ORC_ACCEL_GRAVITY / ORC_SPECIFIC_HEAT_AIR) / (pressure_hl[l + 1] - pressure_hl[l]);
    for (int b = 0; b < nband; ++b)      /* heating rate from the direct beam only */
      hrf[(size_t)l * nband + b] = conv[l] * (fdn[(size_t)(l + 1) * nband + b] - fdn[(size_t)l * nband + b]);
  }
  /* ---------------- cost and the adjoints of the band fluxes ---------------- */
  const int mix = broadband_weight > 0.0;
  const double alpha = mix ? (1.0 - broadband_weight) / nband : 1.0;     /* weight of the per-band terms */
  double cost_bands = 0.0;
  for (int b = 0; b < nband; ++b) {
    double s = 0.0;
    for (int l = 0; l < nlay; ++l) {
      const double d = hrf[(size_t)l * nband + b] - hr[(size_t)l * nband + b];
      s += layer_weight[l] * d * d;
      const double a_h = alpha * W2 * 2.0 * layer_weight[l] * d;
      a_fdn[(size_t)(l + 1) * nband + b] += a_h * conv[l];
      a_fdn[(size_t)l * nband + b] -= a_h * conv[l];
    }
    const double ds = fdn[(size_t)nlay * nband + b] - flux_dn[(size_t)nlay * nband + b];
    const double dt = fup[b] - flux_up[b];
    cost_bands += W2 * s + flux_weight * (ds * ds + 20.0 * dt * dt);
    a_fdn[(size_t)nlay * nband + b] += alpha * flux_weight * 2.0 * ds;
    a_fup[b] += alpha * flux_weight * 40.0 * dt;
    if (flux_profile_weight > 0.0)
      for (int i = 1; i < nlay; ++i) {
        const double iw = flux_profile_weight * 0.5 * (layer_weight[i - 1] + layer_weight[i]);
        const double dd = fdn[(size_t)i * nband + b] - flux_dn[(size_t)i * nband + b];
        const double du = fup[(size_t)i * nband + b] - flux_up[(size_t)i * nband + b];
        cost_bands += iw * (dd * dd + du * du);
        a_fdn[(size_t)i * nband + b] += alpha * iw * 2.0 * dd;
        a_fup[(size_t)i * nband + b] += alpha * iw * 2.0 * du;
      }
  }
  double cost = cost_bands * alpha;
  if (mix) {
    double sbb = 0.0, rs = 0.0, rt = 0.0;
    for (int l = 0; l < nlay; ++l) {
      double r = 0.0;
      for (int b = 0; b < nband; ++b) r += hrf[(size_t)l * nband + b] - hr[(size_t)l * nband + b];
      sbb += layer_weight[l] * (r * r);
      const double a_h = broadband_weight * W2 * 2.0 * layer_weight[l] * r;
      for (int b = 0; b < nband; ++b) {
        a_fdn[(size_t)(l + 1) * nband + b] += a_h * conv[l];
        a_fdn[(size_t)l * nband + b] -= a_h * conv[l];
      }
    }
    for (int b = 0; b < nband; ++b) {
      rs += fdn[(size_t)nlay * nband + b] - flux_dn[(size_t)nlay * nband + b];
      rt += fup[b] - flux_up[b];
    }
    cost += broadband_weight * W2 * sbb + broadband_weight * flux_weight * (rs * rs);
    for (int b = 0; b < nband; ++b) a_fdn[(size_t)nlay * nband + b] += broadband_weight * flux_weight * 2.0 * rs;
    if (all_pos) {
      cost += broadband_weight * flux_weight * (rt * rt);
      for (int b = 0; b < nband; ++b) a_fup[b] += broadband_weight * flux_weight * 2.0 * rt;
    }
    if (flux_profile_weight > 0.0)
      for (int i = 1; i < nlay; ++i) {
        const double iw = flux_profile_weight * 0.5 * (layer_weight[i - 1] + layer_weight[i]);
        double ed = 0.0, eu = 0.0;
        for (int b = 0; b < nband; ++b) {
          ed += fdn[(size_t)i * nband + b] - flux_dn[(size_t)i * nband + b];
          eu += fup[(size_t)i * nband + b] - flux_up[(size_t)i * nband + b];
        }
        cost += broadband_weight * iw * (ed * ed);
        for (int b = 0; b < nband; ++b) a_fdn[(size_t)i * nband + b] += broadband_weight * iw * 2.0 * ed;
        if (all_pos) {
          cost += broadband_weight * iw * (eu * eu);
          for (int b = 0; b < nband; ++b) a_fup[(size_t)i * nband + b] += broadband_weight * iw * 2.0 * eu;
        }
      }
  }
  for (int i = 0; i < nhl; ++i)
    for (int g = 0; g < ng; ++g) {
      a_dn[(size_t)i * ng + g] = a_fdn[(size_t)i * nband + band_mapping[g]];
      a_up[(size_t)i * ng + g] = a_fup[(size_t)i * nband + band_mapping[g]];
    }
  if (spectral_boundary_weights && spectral_flux_dn_surf) {               /* calc_cost_function_sw.cpp:271-274 */
    double s = 0.0;
    for (int g = 0; g < ng; ++g) {
      double a = dn[(size_t)nlay * ng + g] - spectral_flux_dn_surf[g];
      if (relative_ckd_flux_dn) a -= relative_ckd_flux_dn[(size_t)nlay * ng + g];
      s += spectral_boundary_weights[g] * a * a;
      a_dn[(size_t)nlay * ng + g] += spectral_boundary_weights[g] * 2.0 * a;
    }
    cost += s;
  }
  /* ---------------- reverse of the two sweeps ---------------- */
  for (int g = 0; g < ng; ++g) {
    if (!all_nonpos) {
      for (int l = 0; l < nlay; ++l) {                     /* up(l) = up(l+1) tu(l) */
        const size_t i = (size_t)l * ng + g;
        const double au = a_up[i];
        a_up[i + ng] += au * tu[i];
        d_od[i] += au * up[i + ng] * tu[i] * (-2.0);
      }
      a_dn[(size_t)nlay * ng + g] += a_up[(size_t)nlay * ng + g] * alb_g[g];   /* up(nlay) = dn(nlay) albedo */
    }
    for (int l = nlay - 1; l >= 0; --l) {                  /* dn(l+1) = dn(l) td(l) */
      const size_t i = (size_t)l * ng + g;
      const double ad = a_dn[i + ng];
      a_dn[i] += ad * td[i];
      d_od[i] += ad * dn[i] * td[i] * minus_sec_sza;
    }
  }
  free(td); free(tu); free(dn); free(up); free(alb_g); free(fdn); free(fup); free(hrf); free(a_fdn); free(a_fup);
  free(a_dn); free(a_up); free(conv);
  return cost;
}
