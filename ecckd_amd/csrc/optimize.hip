// optimize.hip - K8/K9: cost function and gradient of optimize_lut on gfx950, and the
// L-BFGS driver that replaces solve_adept.
//
// Replaces reference src/ecckd/solve_adept.cpp:
//   CkdOptimizable::calc_cost_function_gradient (:240-292)  -> ecckd_opt_cost_grad
//   calc_cost_function_and_gradient (:72-211)               -> K8a forward + hand-written adjoint
//   calc_total_optical_depth / CkdModel::calc_optical_depth
//       (:24-69, ckd_model.cpp:925-1102)                    -> sparse LUT interpolation, host-built
//   calc_cost_function_ckd_lw (calc_cost_function_lw.cpp:116-232)
//   CkdModel::calc_background_cost_function (ckd_model.cpp:840-877) -> K9 Kronecker stencil
//   solve_adept (:310-417, adept::Minimizer L-BFGS)         -> ecckd_opt_minimize
//
// Design.  The reference records ~1e6 (profile, layer, g) cells on an Adept tape and
// reverses it, serially, every iteration.  Here:
//  * The LUT interpolation weights depend only on (profile, layer, gas), not on g or on the
//    state, so they are computed ONCE on the host into a fixed-width sparse table
//    (cell -> up to 8 nodes per gas, coefficient = weight * interpolation weight).  The
//    optical depth of a cell is then a gather of contiguous ng-long rows of the state
//    (g is the fastest index of the LUT, ckd_model.cpp:153,216) - fully coalesced.
//  * K8a: one block per profile, one thread per g point: gather, negative-OD penalty, LW
//    two-stream forward sweep (fluxes kept in LDS), band sums, cost terms, then the
//    HAND-DERIVED adjoint of the two sweeps -> dJ/d(optical depth) per cell.
//  * K8b: the transpose of the gather as a second, host-built sparse table (node -> cells),
//    so the gradient is accumulated in a fixed order with NO atomics (bitwise reproducible),
//    fused with the chain rule x = ln k (:276-283).
//  * K9: the prior.  The reference forms B = rho_t^|dt| * rho_p^|dp| (* rho_c^|dc|), inverts it
//    densely with LAPACK and multiplies (ckd_model.cpp:706-713, :857).  B is a Kronecker
//    product of AR(1) matrices, so B^-1 is the Kronecker product of tridiagonals: a 9- or
//    27-point stencil per element.  Entries below 1e-6 are dropped as the reference does.
// The dominant costs are launch latency and ~50 MB of traffic per iteration;
// the metric is iterations per second.
#include "common.hpp"
#include <chrono>
#include "fastmath.hpp"

#include <algorithm>
#include <atomic>
#include <immintrin.h>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr double kD = ECCKD_LW_DIFFUSIVITY;
constexpr double MIN_X = -1.0e20;  // solve_adept.cpp:21

struct GasInfo {
  int conc = 0;       // 0 none, 1 linear, 2 lut, 3 relative-linear (ckd_model.cpp:1020-1086)
  int active = 0;
  int nconc = 1;
  size_t nnode = 0;   // nconc*nt*np
  size_t ix = 0;      // offset of this gas in the coefficient vector k
  double reference_vmr = 0.0;
  std::vector<double> vmr;
  std::vector<double> sigma;  // background_error per g (ckd_model.cpp:672-683)
};

// per-dimension tridiagonal of an AR(1) inverse
struct Tri { std::vector<double> lo, di, up; };

Tri ar1_inverse(int n, double rho) {
  Tri t;
  t.lo.assign(n, 0.0); t.di.assign(n, 1.0); t.up.assign(n, 0.0);
  if (n == 1) return t;
  const double den = 1.0 - rho * rho;
  for (int i = 0; i < n; ++i) {
    t.di[i] = ((i == 0 || i == n - 1) ? 1.0 : 1.0 + rho * rho) / den;
    if (i > 0) t.lo[i] = -rho / den;
    if (i < n - 1) t.up[i] = -rho / den;
  }
  return t;
}

__device__ __forceinline__ double block_reduce_sum(double v, double* s_red) {
  // fixed-order reduction over a block of up to 1024 threads; result valid in every thread
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwave = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < nwave; ++w) t += s_red[w];
  return t;
}

// K8a.  grid = ncolumns (all scenes), block = ng rounded up to 64.
// LDS: tau[nlay][ng] | fdn[nhl][ng] | fup[nhl][ng] | Bdn[nhl][nband] | Bup[nhl][nband] |
//      Gdn[nhl][nband] | Gup[nhl][nband] | red[16]
__global__ void k_opt_forward_adjoint(
    int do_sw, const double* __restrict__ mu0 /*[ncol], SW*/,
    int ray_ent /* run_ckd mode: entry of the Rayleigh term, added AFTER the clamp; else -1 */,
    int keep_negative /* scale_lut mode: the optical depth is reported without the clamp at zero */,
    const double* __restrict__ rel_flux /* [ncol][2][nhl][ng] CKD fluxes of the relative-to scene, or NULL (solve_adept.cpp:118-125) */,
    int nlay, int ng, int ngpad, int nband, int nent,
    const double* __restrict__ k,            // [nk] coefficients of every gas
    const int* __restrict__ ent_idx,         // [ncell][nent] start of an ng-long row of k, or -1
    const double* __restrict__ ent_coef,     // [ncell][nent]
    const int* __restrict__ band_of_g,       // [ng]
    const double* __restrict__ planck_hl,    // [ncol][nhl][ng]
    const double* __restrict__ surf_emis,    // [ncol][nband]
    const double* __restrict__ conv,         // [ncol][nlay] heating-rate conversion
    const double* __restrict__ layer_weight, // [ncol][nlay] normalised
    const double* __restrict__ hr_true,      // [ncol][nlay][nband]
    const double* __restrict__ fdn_true,     // [ncol][nhl][nband]
    const double* __restrict__ fup_true,     // [ncol][nhl][nband]
    const double* __restrict__ sfds,         // [ncol][ng] or NULL
    const double* __restrict__ sfut,         // [ncol][ng] or NULL
    double flux_weight, double flux_profile_weight, double broadband_weight,
    double spectral_boundary_weight, double negative_od_penalty,
    double* __restrict__ dtau,               // [ncell][ng] dJ/d(optical depth)
    double* __restrict__ jcol,               // [ncol] cost of each profile (incl. penalty)
    double* __restrict__ od_out,             // optional [ncell][ng] total optical depth (after clamp)
    double* __restrict__ flux_out) {         // optional [ncol][2][nhl][ng]
  extern __shared__ double smem[];
  const int nhl = nlay + 1;
  double* s_tau = smem;
  double* s_fdn = s_tau + (size_t)nlay * ng;
  double* s_fup = s_fdn + (size_t)nhl * ng;
  double* s_bdn = s_fup + (size_t)nhl * ng;
  double* s_bup = s_bdn + (size_t)nhl * nband;
  double* s_gdn = s_bup + (size_t)nhl * nband;
  double* s_gup = s_gdn + (size_t)nhl * nband;
  double* s_red = s_gup + (size_t)nhl * nband;
  unsigned char* s_clamp = (unsigned char*)(s_red + 16);  // [nlay][ng] 1 where a negative optical depth was clamped
  const int col = blockIdx.x;
  // blockDim = ngpad * (layer groups): the LUT gather is parallel over (layer group, g); the
  // sweeps and their adjoint are sequential in the layers and run on the first group only
  const int g = threadIdx.x % ngpad;
  const int lgrp = threadIdx.x / ngpad;
  const int nlgrp = blockDim.x / ngpad;
  const bool in_range = g < ng;
  const bool live = in_range && lgrp == 0;
  const double hr_weight = 3600.0 * 24.0;
  const double* pl = planck_hl + (size_t)col * nhl * ng;
  const size_t cell0 = (size_t)col * nlay;

  // ---- optical depth: sparse gather (calc_total_optical_depth, solve_adept.cpp:24-69) and
  //      the negative-OD penalty with clamp (:107-116)
  double penalty = 0.0;
  if (in_range) {
    for (int l = lgrp; l < nlay; l += nlgrp) {
      const int* ei = ent_idx + (cell0 + l) * nent;
      const double* ec = ent_coef + (cell0 + l) * nent;
      double tau = 0.0, tau_ray = 0.0;
      // eight (then four) entries at a time: the index -> coefficient loads of a batch are independent, so their (L2) latencies
      // overlap; the products are still added in entry order, absent entries (idx < 0) add an exact zero
      int e = 0;
      for (; e + 8 <= nent; e += 8) {
        int ix[8];
        double cv[8], kv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { ix[q] = ei[e + q]; cv[q] = ec[e + q]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) kv[q] = ix[q] >= 0 ? k[(size_t)ix[q] + g] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const double t = ix[q] >= 0 ? cv[q] * kv[q] : 0.0;
          if (e + q == ray_ent) tau_ray = t; else tau += t;
        }
      }
      for (; e + 4 <= nent; e += 4) {
        int ix[4];
        double cv[4], kv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { ix[q] = ei[e + q]; cv[q] = ec[e + q]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) kv[q] = ix[q] >= 0 ? k[(size_t)ix[q] + g] : 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double t = ix[q] >= 0 ? cv[q] * kv[q] : 0.0;
          if (e + q == ray_ent) tau_ray = t; else tau += t;
        }
      }
      for (; e < nent; ++e) {
        const int idx = ei[e];
        if (idx >= 0) {
          if (e == ray_ent) tau_ray = ec[e] * k[(size_t)idx + g];
          else tau += ec[e] * k[(size_t)idx + g];
        }
      }
      const double tau_raw = tau;
      if (tau < 0.0) {
        // penalty = negative_od_penalty * tau^2, then the optical depth is SET to 0 (:110-113), so
        // the only derivative that survives for this cell is the penalty's
        penalty += tau * tau;
        dtau[(cell0 + l) * ng + g] = 2.0 * negative_od_penalty * tau;
        tau = 0.0;
        s_clamp[l * ng + g] = 1;
      } else {
        s_clamp[l * ng + g] = 0;
      }
      if (od_out) od_out[(cell0 + l) * ng + g] = (keep_negative && s_clamp[l * ng + g]) ? tau_raw : tau;   // run_ckd mode: molecular absorption only (run_ckd.cpp:318-326)
      // keep_negative == 2: the sweeps see the optical depth as it comes out of the look-up tables, negative cells included:
      // the "relative_to" fluxes, od = value(aod) -> LblFluxes::calc_ckd_fluxes (optimize_lut.cpp:231-234), which has no clamp
      if (keep_negative == 2) tau = tau_raw;
      tau += tau_ray;
      s_tau[l * ng + g] = tau;
    }
  }
  __syncthreads();
  // ---- forward sweeps ----
  // shortwave: are all band albedos > 0 (broadband/profile upwelling terms, calc_cost_function_sw.cpp:252,:264)?
  // all <= 0 selects the direct-only model with zero upwelling (:145-150)
  bool all_pos = true, all_nonpos = true;
  double cos_sza = 1.0;
  if (do_sw) {
    for (int b = 0; b < nband; ++b) {
      const bool pos = surf_emis[(size_t)col * nband + b] > 0.0;
      all_pos = all_pos && pos;
      all_nonpos = all_nonpos && !pos;
    }
    cos_sza = mu0[col];
  }
  if (live && !do_sw) {
    // radiative_transfer_lw.cpp:41-59
    double dn = 0.0;
    s_fdn[g] = 0.0;
    for (int l = 0; l < nlay; ++l) {
      const double tau = s_tau[l * ng + g];
      const double eps = 1.0 - ecckd::exp_fast(-kD * tau);
      const double fac = (eps > 1.0e-5) ? 1.0 - ecckd::div_fast(eps * (1.0 / kD), tau) : 0.5 * eps;
      dn = dn * (1.0 - eps) + pl[l * ng + g] * (eps - fac) + pl[(l + 1) * ng + g] * fac;
      s_fdn[(l + 1) * ng + g] = dn;
    }
    const double es = surf_emis[(size_t)col * nband + band_of_g[g]];
    double up = pl[nlay * ng + g] * es + (1.0 - es) * dn;  // surf_planck = planck_hl(end), optimize_lut.cpp:267
    s_fup[nlay * ng + g] = up;
    for (int l = nlay - 1; l >= 0; --l) {
      const double tau = s_tau[l * ng + g];
      const double eps = 1.0 - ecckd::exp_fast(-kD * tau);
      const double fac = (eps > 1.0e-5) ? 1.0 - ecckd::div_fast(eps * (1.0 / kD), tau) : 0.5 * eps;
      up = up * (1.0 - eps) + pl[(l + 1) * ng + g] * (eps - fac) + pl[l * ng + g] * fac;
      s_fup[l * ng + g] = up;
    }
  }
  if (live && do_sw) {
    // radiative_transfer_direct_sw / _norayleigh_sw (radiative_transfer_sw.cpp:26-77); row 0 of the
    // per-profile "planck" block carries the scaled solar irradiance per g (solve_adept.cpp:176-186),
    // surf_emis the effective band albedo.  Albedo 0 gives up = 0, which is the direct-only branch.
    const double minus_sec_sza = -1.0 / cos_sza;
    double dn = cos_sza * pl[g];
    s_fdn[g] = dn;
    for (int l = 0; l < nlay; ++l) {
      dn = dn * ecckd::exp_fast(minus_sec_sza * s_tau[l * ng + g]);
      s_fdn[(l + 1) * ng + g] = dn;
    }
    const double alb = all_nonpos ? 0.0 : surf_emis[(size_t)col * nband + band_of_g[g]];
    double up = dn * alb;
    s_fup[nlay * ng + g] = up;
    for (int l = nlay - 1; l >= 0; --l) {
      up = up * ecckd::exp_fast(-2.0 * s_tau[l * ng + g]);
      s_fup[l * ng + g] = up;
    }
  }
  if (live) {
    if (flux_out) {
      double* fo = flux_out + (size_t)col * 2 * nhl * ng;
      for (int i = 0; i < nhl; ++i) {
        fo[i * ng + g] = s_fdn[i * ng + g];
        fo[(nhl + i) * ng + g] = s_fup[i * ng + g];
      }
    }
  }
  __syncthreads();
  // ---- band sums (calc_cost_function_lw.cpp:171-184) ----
  for (int t = threadIdx.x; t < nhl * nband; t += blockDim.x) {
    const int i = t / nband, b = t % nband;
    double sd = 0.0, su = 0.0;
    const double* rel = rel_flux ? rel_flux + (size_t)col * 2 * nhl * ng : nullptr;
    for (int gg = 0; gg < ng; ++gg) {
      if (band_of_g[gg] == b) {
        // the relative-to fluxes are subtracted per g point before the band sums (calc_cost_function_lw.cpp:162-165)
        sd += rel ? s_fdn[i * ng + gg] - rel[i * ng + gg] : s_fdn[i * ng + gg];
        su += rel ? s_fup[i * ng + gg] - rel[(nhl + i) * ng + gg] : s_fup[i * ng + gg];
      }
    }
    s_bdn[t] = sd;
    s_bup[t] = su;
  }
  __syncthreads();
  // ---- cost (:186-229) and its derivative w.r.t. the band fluxes ----
  const double* cv = conv + (size_t)col * nlay;
  const double* lw = layer_weight + (size_t)col * nlay;
  const double* hrt = hr_true + (size_t)col * nlay * nband;
  const double* fdt = fdn_true + (size_t)col * nhl * nband;
  const double* fut = fup_true + (size_t)col * nhl * nband;
  // LW always mixes spectral and broadband (calc_cost_function_lw.cpp:209); SW only if broadband_weight > 0
  // (calc_cost_function_sw.cpp:243)
  const double spec_scale = (!do_sw || broadband_weight > 0.0) ? (1.0 - broadband_weight) / nband : 1.0;
  const double up_in_hr = do_sw ? 0.0 : 1.0;  // SW heating rate from the direct beam only (:197)
  for (int t = threadIdx.x; t < nhl * nband; t += blockDim.x) { s_gdn[t] = 0.0; s_gup[t] = 0.0; }
  __syncthreads();
  double jpart = 0.0;
  // heating-rate terms: thread per layer (the broadband residual needs all bands of a layer).  A half level receives the
  // contributions of the layer above and of the layer below it: the even layers add theirs first, then the odd ones - each
  // slot has ONE writer per pass, a fixed order without atomics
  for (int pass = 0; pass < 2; ++pass) {
    for (int l = threadIdx.x; l < nlay; l += blockDim.x) {
      if ((l & 1) != pass) continue;
      double rsum = 0.0;
      for (int b = 0; b < nband; ++b) {
        const double hrf = cv[l] * (s_bdn[(l + 1) * nband + b] - s_bdn[l * nband + b] -
                                    up_in_hr * (s_bup[(l + 1) * nband + b] - s_bup[l * nband + b]));
        rsum += hrf - hrt[l * nband + b];
      }
      for (int b = 0; b < nband; ++b) {
        const double hrf = cv[l] * (s_bdn[(l + 1) * nband + b] - s_bdn[l * nband + b] -
                                    up_in_hr * (s_bup[(l + 1) * nband + b] - s_bup[l * nband + b]));
        const double r = hrf - hrt[l * nband + b];
        jpart += spec_scale * hr_weight * hr_weight * lw[l] * r * r;
        const double dhr = 2.0 * hr_weight * hr_weight * lw[l] * (spec_scale * r + broadband_weight * rsum);
        // hr_l = conv_l * (dn[l+1] - dn[l] - up[l+1] + up[l])
        s_gdn[(l + 1) * nband + b] += dhr * cv[l];
        s_gdn[l * nband + b] += -dhr * cv[l];
        if (!do_sw) {
          s_gup[(l + 1) * nband + b] += -dhr * cv[l];
          s_gup[l * nband + b] += dhr * cv[l];
        }
      }
      jpart += broadband_weight * hr_weight * hr_weight * lw[l] * rsum * rsum;
    }
    __syncthreads();
  }
  // boundary-flux and flux-profile terms: thread per half level
  for (int i = threadIdx.x; i < nhl; i += blockDim.x) {
    const bool is_surf = (i == nlay), is_toa = (i == 0);
    const bool interior = (i >= 1 && i <= nlay - 1);
    // weights of the squared band residuals of dn / up at this level: spectral (s) and broadband (b)
    double wd_s = 0.0, wu_s = 0.0, wd_b = 0.0, wu_b = 0.0;
    if (is_surf) { wd_s = flux_weight; wd_b = flux_weight; }
    if (is_toa) {
      // SW: 20 x on the spectral TOA upwelling (:214); its broadband term only if all albedos > 0 (:252)
      wu_s = do_sw ? 20.0 * flux_weight : flux_weight;
      wu_b = (!do_sw || all_pos) ? flux_weight : 0.0;
    }
    if (interior && flux_profile_weight > 0.0) {
      const double iw = flux_profile_weight * 0.5 * (lw[i - 1] + lw[i]);
      wd_s = iw; wu_s = iw; wd_b = iw;
      wu_b = (!do_sw || all_pos) ? iw : 0.0;   // :264
    }
    if (wd_s != 0.0 || wu_s != 0.0) {
      double sd = 0.0, su = 0.0;
      for (int b = 0; b < nband; ++b) {
        sd += s_bdn[i * nband + b] - fdt[i * nband + b];
        su += s_bup[i * nband + b] - fut[i * nband + b];
      }
      for (int b = 0; b < nband; ++b) {
        const double dd = s_bdn[i * nband + b] - fdt[i * nband + b];
        const double du = s_bup[i * nband + b] - fut[i * nband + b];
        jpart += spec_scale * (wd_s * dd * dd + wu_s * du * du);
        s_gdn[i * nband + b] += 2.0 * (spec_scale * wd_s * dd + broadband_weight * wd_b * sd);
        s_gup[i * nband + b] += 2.0 * (spec_scale * wu_s * du + broadband_weight * wu_b * su);
      }
      jpart += broadband_weight * (wd_b * sd * sd + wu_b * su * su);
    }
  }
  __syncthreads();
  // ---- adjoint of the two sweeps, per g ----
  if (live && do_sw) {
    const int b = band_of_g[g];
    const double alb = all_nonpos ? 0.0 : surf_emis[(size_t)col * nband + b];
    double g_dn_surf_extra = 0.0;
    if (sfds && sfut) {
      // calc_cost_function_sw.cpp:271-274: per-g weights (erythemal) on the surface direct flux
      const double wgt = sfut[(size_t)col * ng + g];
      const double reld = rel_flux ? rel_flux[((size_t)col * 2 * nhl + nlay) * ng + g] : 0.0;
      const double a = (s_fdn[nlay * ng + g] - reld) - sfds[(size_t)col * ng + g];
      jpart += wgt * a * a;
      g_dn_surf_extra = 2.0 * wgt * a;
    }
    // up_l = up_{l+1} * exp(-2 tau_l): d up_l / d tau_l = -2 up_l
    double up_bar = s_gup[b];
    for (int l = 0; l < nlay; ++l) {
      const double tau = s_tau[l * ng + g];
      if (s_clamp[l * ng + g] == 0) dtau[(cell0 + l) * ng + g] = up_bar * s_fup[l * ng + g] * (-2.0);
      up_bar = up_bar * ecckd::exp_fast(-2.0 * tau) + s_gup[(l + 1) * nband + b];
    }
    // dn_{l+1} = dn_l * exp(-tau_l / mu0): d dn_{l+1} / d tau_l = -dn_{l+1} / mu0
    const double minus_sec_sza = -1.0 / cos_sza;
    double dn_bar = s_gdn[nlay * nband + b] + g_dn_surf_extra + up_bar * alb;
    for (int l = nlay - 1; l >= 0; --l) {
      const double tau = s_tau[l * ng + g];
      if (s_clamp[l * ng + g] == 0) dtau[(cell0 + l) * ng + g] += dn_bar * s_fdn[(l + 1) * ng + g] * minus_sec_sza;
      dn_bar = dn_bar * ecckd::exp_fast(minus_sec_sza * tau) + s_gdn[l * nband + b];
    }
  }
  if (live && !do_sw) {
    const int b = band_of_g[g];
    const double es = surf_emis[(size_t)col * nband + b];
    double g_dn_surf_extra = 0.0, g_up_toa_extra = 0.0;
    if (spectral_boundary_weight > 0.0 && sfds && sfut) {
      // :223-229 on the un-banded fluxes
      const double reld = rel_flux ? rel_flux[((size_t)col * 2 * nhl + nlay) * ng + g] : 0.0;
      const double relu = rel_flux ? rel_flux[((size_t)col * 2 * nhl + nhl) * ng + g] : 0.0;
      const double a = (s_fdn[nlay * ng + g] - reld) - sfds[(size_t)col * ng + g];
      const double c = (s_fup[g] - relu) - sfut[(size_t)col * ng + g];
      jpart += spectral_boundary_weight * (a * a + c * c);
      g_dn_surf_extra = 2.0 * spectral_boundary_weight * a;
      g_up_toa_extra = 2.0 * spectral_boundary_weight * c;
    }
    // Adjoint of the up sweep.  It ran l = nlay-1 .. 0 (up_l from up_{l+1}), so the adjoint of
    // up[l] is complete once levels 0..l-1 have been visited: u_bar[0] = seed[0],
    // u_bar[l+1] = seed[l+1] + u_bar[l]*(1-eps_l).  u_bar[l] overwrites the forward up[l] in LDS
    // (the forward values are re-generated from the surface in the combine loop below).
    double up_bar = s_gup[b] + g_up_toa_extra;  // level 0
    for (int l = 0; l < nlay; ++l) {
      const double eps = 1.0 - ecckd::exp_fast(-kD * s_tau[l * ng + g]);
      s_fup[l * ng + g] = up_bar;
      up_bar = up_bar * (1.0 - eps) + s_gup[(l + 1) * nband + b];
    }
    // up_bar is now the adjoint of up[nlay]; through the surface condition it feeds dn[nlay].
    double dn_bar = s_gdn[nlay * nband + b] + g_dn_surf_extra + up_bar * (1.0 - es);
    // forward up[l+1] values: rebuild from the surface
    double up_fwd_next = pl[nlay * ng + g] * es + (1.0 - es) * s_fdn[nlay * ng + g];  // up[nlay]
    for (int l = nlay - 1; l >= 0; --l) {
      const bool clamped = s_clamp[l * ng + g] != 0;
      const double tau = s_tau[l * ng + g];
      const double ex = ecckd::exp_fast(-kD * tau);
      const double eps = 1.0 - ex;
      const bool thick = eps > 1.0e-5;
      const double rtau = thick ? ecckd::div_fast(1.0 / kD, tau) : 0.0;      // 1 / (D tau)
      const double fac = thick ? 1.0 - eps * rtau : 0.5 * eps;
      const double b0 = pl[l * ng + g], b1 = pl[(l + 1) * ng + g];
      const double ubar_l = s_fup[l * ng + g];  // adjoint of up[l]
      const double dn_l = s_fdn[l * ng + g];
      // partials of both sweeps for layer l
      //   dn_{l+1} = dn_l*(1-eps) + B_l*(eps-fac) + B_{l+1}*fac
      //   up_l     = up_{l+1}*(1-eps) + B_{l+1}*(eps-fac) + B_l*fac
      const double t_bar = dn_bar * dn_l + ubar_l * up_fwd_next;
      const double a_bar = dn_bar * b0 + ubar_l * b1;
      const double f_bar = dn_bar * b1 + ubar_l * b0;
      const double eps_bar = -t_bar + a_bar;
      const double fac_bar = f_bar - a_bar;
      // fac(eps, tau): thick: 1 - eps/(D tau); thin: eps/2   (radiative_transfer_lw.cpp:42-43)
      const double dfac_deps = thick ? -rtau : 0.5;
      const double dfac_dtau = thick ? eps * rtau * (kD * rtau) : 0.0;        // eps / (D tau^2)
      const double tau_bar = (eps_bar + fac_bar * dfac_deps) * (kD * ex) + fac_bar * dfac_dtau;
      if (!clamped) dtau[(cell0 + l) * ng + g] = tau_bar;
      // propagate
      const double up_l = up_fwd_next * (1.0 - eps) + b1 * (eps - fac) + b0 * fac;
      up_fwd_next = up_l;
      dn_bar = dn_bar * (1.0 - eps) + s_gdn[l * nband + b];
    }
  }
  jpart += negative_od_penalty * penalty;
  const double jtot = block_reduce_sum(jpart, s_red);
  if (threadIdx.x == 0) jcol[col] = jtot;
}

// ---- K8a, the form that runs (the kernel above stays as the general fall-back and as the cross-check of this one) ----
// Same arithmetic per cell and the same sequential recurrences, but only what IS sequential runs on one wave:
//   A  every thread (layer group, g) owns the cells (l = lgrp + j * nlgrp, g), j < NC: LUT gather, clamp + penalty, then the
//      layer's transmittance and its two source terms (longwave: 1 - eps, B_l (eps - fac) + B_l+1 fac, B_l+1 (eps - fac) +
//      B_l fac; shortwave: the two transmittances) - the `exp` and the division of every cell, 16 layers side by side;
//   B  ONE wave runs the two recurrences dn_l+1 = dn_l t_l + s_l, up_l = up_l+1 t_l + s'_l: one FMA per layer from LDS;
//   C, D  band sums and cost terms as in the kernel above (same code, same summation order);
//   E  the adjoint recurrences (again one FMA per layer on one wave), then every thread forms dJ/dtau of ITS cells from the
//      adjoint and forward fluxes at its layer and what it kept in registers (tau, exp(-D tau)).
// The sequential part of a profile shrinks from 4 x nlay (exp + division + recurrence) to 4 x nlay FMAs.  LDS per profile:
// longwave t | dn | up, shortwave dn | up only (the transmittances sit in the rows the fluxes then overwrite, and come back
// from the owners' registers for the adjoint recurrences) -> 107 KB / 79 KB at nlay = 54, ng = 64: two shortwave blocks per CU.
__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// Optical depth of one cell for the 64 g points of a wave (calc_total_optical_depth, solve_adept.cpp:24-69).  The cell is the
// same for the whole wave, so its table entries (node row, coefficient) are fetched ONCE, one entry per lane (two coalesced
// loads instead of 2 x nent same-address loads per lane; the caller fetches them for all its cells before the first is used),
// and handed round with v_readlane: the row offsets arrive in scalar registers and the loads of the coefficient rows are
// independent of each other - sixteen in flight per batch, the products still added in entry order; absent entries (idx < 0)
// add nothing.  `active` = this lane has a g point.
constexpr int K8A_GB = 8;      // coefficient rows per batch; two batches are in flight (the next one is issued before the
                               // current one is used)
struct GatherBatch { int ix[K8A_GB]; double cv[K8A_GB], kv[K8A_GB]; };

// (row_stride = 1: my_idx is the offset of the row in k; else the row's number, of row_stride doubles each)
__device__ __forceinline__ void gather_issue(GatherBatch& b, int my_idx, double my_c, int q0, int ne, const double* __restrict__ k, int gl,
                                             int row_stride = 1) {
#pragma unroll
  for (int q = 0; q < K8A_GB; ++q) {
    const int e = (q0 + q) & 63;                          // (lanes at and beyond `ne` hold idx = -1: no test needed; a batch
    b.ix[q] = __builtin_amdgcn_readlane(my_idx, e);       //  never wraps round because ne <= 64 and q0 is a multiple of 8)
    b.cv[q] = readlane_f64(my_c, e);
  }
  // the loads are UNCONDITIONAL (an absent entry reads row 0 and is dropped by the select in gather_use): a test round a
  // load makes the compiler wait for everything in flight where the paths join, one memory round trip per entry
#pragma unroll
  for (int q = 0; q < K8A_GB; ++q) b.kv[q] = k[(size_t)max(b.ix[q], 0) * row_stride + gl];
}

__device__ __forceinline__ void gather_use(const GatherBatch& b, int e_first, int ray_ent, double& tau, double& tau_ray) {
#pragma unroll
  for (int q = 0; q < K8A_GB; ++q) {
    const double t = b.ix[q] >= 0 ? b.cv[q] * b.kv[q] : 0.0;
    if (e_first + q == ray_ent) tau_ray = t; else tau += t;
  }
}

// NB > 0: the table has at most NB * 8 entries - all NB batches are issued before the first is used, in straight-line code
// (no loop, no test: the compiler counts the loads in flight exactly).  NB = 0: any length, two batches in flight.
template <int NB>
__device__ __forceinline__ void gather_batches(int my_idx, double my_c, int e0, int ne, const double* __restrict__ k, int gl,
                                               int ray_ent, double& tau, double& tau_ray) {
  if constexpr (NB > 0) {
    GatherBatch b[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) gather_issue(b[i], my_idx, my_c, i * K8A_GB, ne, k, gl);
#pragma unroll
    for (int i = 0; i < NB; ++i) gather_use(b[i], e0 + i * K8A_GB, ray_ent, tau, tau_ray);
  } else {
    GatherBatch A, B;
    gather_issue(A, my_idx, my_c, 0, ne, k, gl);
    for (int q0 = 0; q0 < ne; q0 += 2 * K8A_GB) {
      // (the issues are unconditional - past the end they fetch row 0 and drop it -: a conditional issue makes the compiler
      // assume the fewest loads in flight and wait for all of them)
      gather_issue(B, my_idx, my_c, (q0 + K8A_GB) & 63, ne, k, gl);
      gather_use(A, e0 + q0, ray_ent, tau, tau_ray);
      gather_issue(A, my_idx, my_c, (q0 + 2 * K8A_GB) & 63, ne, k, gl);
      if (q0 + K8A_GB < ne) gather_use(B, e0 + q0 + K8A_GB, ray_ent, tau, tau_ray);
    }
  }
}

constexpr int K8A_CH = 8;      // layers per prefetch chunk of the sequential recurrences
constexpr int K8A_MAXR = 4;    // (half level, band) slots per thread in the cost phase: (nlay + 1) * nband <= K8A_MAXR * blockDim

template <int NC, bool SW, int NB>
__global__ void __launch_bounds__(1024)
k_opt_forward_adjoint_cells(
    const double* __restrict__ mu0, int ray_ent, int keep_negative, const double* __restrict__ rel_flux,
    int nlay, int ng, int ngpad, int nband, int nent,
    const double* __restrict__ k, const int* __restrict__ ent_idx, const double* __restrict__ ent_coef,
    const int* __restrict__ band_of_g, const int* __restrict__ band_ptr /*[nband+1]*/, const int* __restrict__ band_g /*[ng]: g points by band*/,
    const double* __restrict__ planck_hl, const double* __restrict__ surf_emis,
    const double* __restrict__ conv, const double* __restrict__ layer_weight, const double* __restrict__ hr_true,
    const double* __restrict__ fdn_true, const double* __restrict__ fup_true, const double* __restrict__ sfds,
    const double* __restrict__ sfut, double flux_weight, double flux_profile_weight, double broadband_weight,
    double spectral_boundary_weight, double negative_od_penalty, double* __restrict__ dtau, double* __restrict__ jcol,
    double* __restrict__ od_out, double* __restrict__ flux_out) {
  extern __shared__ double smem[];
  const int nhl = nlay + 1;
  double* s_T = smem;                                            // longwave only: [nlay][ng] 1 - eps
  double* s_D = s_T + (SW ? (size_t)0 : (size_t)nlay * ng);      // [nhl][ng] down: sources / transmittances, fluxes, adjoints
  double* s_U = s_D + (size_t)nhl * ng;                          // [nhl][ng] up
  double* s_bdn = s_U + (size_t)nhl * ng;                        // [nhl][nband] band fluxes, then dJ/d(band flux)
  double* s_bup = s_bdn + (size_t)nhl * nband;
  double* s_r = s_bup + (size_t)nhl * nband;                     // [nlay][nband] heating-rate residuals
  double* s_rsum = s_r + (size_t)nlay * nband;                   // [nlay] their broadband sums
  double* s_sd = s_rsum + nlay;                                  // [nhl] broadband flux residuals
  double* s_su = s_sd + nhl;
  double* s_red = s_su + nhl;                                    // [16]
  int* s_bp = (int*)(s_red + 16);                                // [nband + 1]
  int* s_bg = s_bp + nband + 1;                                  // [ng]
  const int col = blockIdx.x;
  const int g = threadIdx.x % ngpad;
  const int lgrp = __builtin_amdgcn_readfirstlane(threadIdx.x / ngpad);     // ngpad is a multiple of 64: uniform in a wave
  const int nlgrp = blockDim.x / ngpad;
  const bool in_range = g < ng;
  const bool live = in_range && lgrp == 0;
  const double hr_weight = 3600.0 * 24.0;
  const double* pl = planck_hl + (size_t)col * nhl * ng;
  const size_t cell0 = (size_t)col * nlay;
  bool all_pos = true, all_nonpos = true;
  double cos_sza = 1.0;
  if (SW) {
    for (int b = 0; b < nband; ++b) {
      const bool pos = surf_emis[(size_t)col * nband + b] > 0.0;
      all_pos = all_pos && pos;
      all_nonpos = all_nonpos && !pos;
    }
    cos_sza = mu0[col];
  }
  const double minus_sec_sza = -1.0 / cos_sza;
  for (int t = threadIdx.x; t <= nband; t += blockDim.x) s_bp[t] = band_ptr[t];
  for (int t = threadIdx.x; t < ng; t += blockDim.x) s_bg[t] = band_g[t];

  // ---- A: the cells of this thread ----
  double c_tau[NC], c_t[NC];       // optical depth as the sweeps see it; exp(-D tau) (longwave) / exp(-tau / mu0) (shortwave)
  double c_tu[SW ? NC : 1];        // shortwave: exp(-2 tau)
  unsigned clamp_mask = 0;
  double penalty = 0.0;
  // the table entries of all the cells of this thread first (entries 0..63; tables with more come in further rounds below)
  const int lane = threadIdx.x & 63;
  const int gl = in_range ? g : 0;        // lanes beyond the last g point read (and drop) the first one's values
  int t_idx[NC];
  double t_c[NC];
  const int ne0 = min(64, nent);
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int l = lgrp + j * nlgrp;
    const bool have = l < nlay && lane < ne0;
    t_idx[j] = have ? ent_idx[(cell0 + l) * nent + lane] : -1;
    t_c[j] = have ? ent_coef[(cell0 + l) * nent + lane] : 0.0;
  }
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int l = lgrp + j * nlgrp;
    c_tau[j] = 0.0; c_t[j] = 1.0;
    if constexpr (SW) c_tu[j] = 1.0;
    if (l < nlay) {
      double tau = 0.0, tau_ray = 0.0;
      gather_batches<NB>(t_idx[j], t_c[j], 0, ne0, k, gl, ray_ent, tau, tau_ray);
      for (int e0 = 64; e0 < nent; e0 += 64) {
        const int ne = min(64, nent - e0);
        const int my_idx = lane < ne ? ent_idx[(cell0 + l) * nent + e0 + lane] : -1;
        const double my_c = lane < ne ? ent_coef[(cell0 + l) * nent + e0 + lane] : 0.0;
        gather_batches<0>(my_idx, my_c, e0, ne, k, gl, ray_ent, tau, tau_ray);
      }
      if (in_range) {
        const double tau_raw = tau;
        bool clamped = false;
        if (tau < 0.0) {
          penalty += tau * tau;                                       // solve_adept.cpp:110-113
          dtau[(cell0 + l) * ng + g] = 2.0 * negative_od_penalty * tau;
          tau = 0.0;
          clamped = true;
          clamp_mask |= 1u << j;
        }
        if (od_out) od_out[(cell0 + l) * ng + g] = (keep_negative && clamped) ? tau_raw : tau;
        if (keep_negative == 2) tau = tau_raw;
        tau += tau_ray;
        c_tau[j] = tau;
        if constexpr (!SW) {
          const double ex = ecckd::exp_fast(-kD * tau);
          const double eps = 1.0 - ex;
          const double fac = (eps > 1.0e-5) ? 1.0 - ecckd::div_fast(eps * (1.0 / kD), tau) : 0.5 * eps;   // radiative_transfer_lw.cpp:41-43
          const double b0 = pl[l * ng + g], b1 = pl[(l + 1) * ng + g];
          c_t[j] = ex;
          s_T[l * ng + g] = 1.0 - eps;
          s_D[(l + 1) * ng + g] = b0 * (eps - fac) + b1 * fac;        // :47-49, the part of flux_dn(l+1) that is not flux_dn(l)
          s_U[l * ng + g] = b1 * (eps - fac) + b0 * fac;              // :56-58
        } else {
          const double td = ecckd::exp_fast(minus_sec_sza * tau), tu = ecckd::exp_fast(-2.0 * tau);
          c_t[j] = td;
          c_tu[j] = tu;
          s_D[(l + 1) * ng + g] = td;
          s_U[l * ng + g] = tu;
        }
      }
    }
  }
  __syncthreads();
  // ---- B: the two recurrences, one wave; the rows of a chunk of layers are fetched before its chain of FMAs starts ----
  if (live) {
    // full chunks without a test per layer (a scalar branch per step costs more than the step), then the remainder
    double dn = SW ? cos_sza * pl[g] : 0.0;
    double* pD = s_D + g;
    double* pU = s_U + g;
    const double* pT = s_T + g;
    pD[0] = dn;
    int l = 0;
    for (; l + K8A_CH <= nlay; l += K8A_CH) {
      double tt[K8A_CH], ss[K8A_CH];
#pragma unroll
      for (int q = 0; q < K8A_CH; ++q) {
        ss[q] = pD[(l + q + 1) * ng];
        if constexpr (!SW) tt[q] = pT[(l + q) * ng];
      }
#pragma unroll
      for (int q = 0; q < K8A_CH; ++q) {
        if constexpr (SW) dn = dn * ss[q]; else dn = dn * tt[q] + ss[q];
        pD[(l + q + 1) * ng] = dn;
      }
    }
    for (; l < nlay; ++l) {
      if constexpr (SW) dn = dn * pD[(l + 1) * ng]; else dn = dn * pT[l * ng] + pD[(l + 1) * ng];
      pD[(l + 1) * ng] = dn;
    }
    double up;
    if constexpr (SW) {
      const double alb = all_nonpos ? 0.0 : surf_emis[(size_t)col * nband + band_of_g[g]];
      up = dn * alb;
    } else {
      const double es = surf_emis[(size_t)col * nband + band_of_g[g]];
      up = pl[nlay * ng + g] * es + (1.0 - es) * dn;              // surf_planck = planck_hl(end), optimize_lut.cpp:267
    }
    pU[nlay * ng] = up;
    l = nlay - 1;
    for (; l - (K8A_CH - 1) >= 0; l -= K8A_CH) {
      double tt[K8A_CH], ss[K8A_CH];
#pragma unroll
      for (int q = 0; q < K8A_CH; ++q) {
        ss[q] = pU[(l - q) * ng];
        if constexpr (!SW) tt[q] = pT[(l - q) * ng];
      }
#pragma unroll
      for (int q = 0; q < K8A_CH; ++q) {
        if constexpr (SW) up = up * ss[q]; else up = up * tt[q] + ss[q];
        pU[(l - q) * ng] = up;
      }
    }
    for (; l >= 0; --l) {
      if constexpr (SW) up = up * pU[l * ng]; else up = up * pT[l * ng] + pU[l * ng];
      pU[l * ng] = up;
    }
  }
  __syncthreads();
  if (flux_out && in_range) {
    double* fo = flux_out + (size_t)col * 2 * nhl * ng;
    for (int i = lgrp; i < nhl; i += nlgrp) {
      fo[i * ng + g] = s_D[i * ng + g];
      fo[(nhl + i) * ng + g] = s_U[i * ng + g];
    }
  }
  // ---- C: band sums (calc_cost_function_lw.cpp:171-184), the g points of a band in increasing order ----
  const double* cv = conv + (size_t)col * nlay;
  const double* lw = layer_weight + (size_t)col * nlay;
  const double* hrt = hr_true + (size_t)col * nlay * nband;
  const double* fdt = fdn_true + (size_t)col * nhl * nband;
  const double* fut = fup_true + (size_t)col * nhl * nband;
  for (int t = threadIdx.x; t < nhl * nband; t += blockDim.x) {
    const int i = t / nband, b = t % nband;
    double sd = 0.0, su = 0.0;
    const double* rel = rel_flux ? rel_flux + (size_t)col * 2 * nhl * ng : nullptr;
    for (int q = s_bp[b]; q < s_bp[b + 1]; ++q) {
      const int gg = s_bg[q];
      // the relative-to fluxes are subtracted per g point before the band sums (calc_cost_function_lw.cpp:162-165)
      sd += rel ? s_D[i * ng + gg] - rel[i * ng + gg] : s_D[i * ng + gg];
      su += rel ? s_U[i * ng + gg] - rel[(nhl + i) * ng + gg] : s_U[i * ng + gg];
    }
    s_bdn[t] = sd;
    s_bup[t] = su;
  }
  __syncthreads();
  // ---- D: cost (:186-229) and its derivative with respect to the band fluxes ----
  // D1 one thread per (layer, band): heating-rate residual; D2 per layer / half level: the broadband sums, bands in order;
  // D3 per (half level, band): dJ/d(band flux) gathered from the two layers that touch the level and the boundary / profile
  // terms - every slot has one writer, no atomics.
  const double spec_scale = (!SW || broadband_weight > 0.0) ? (1.0 - broadband_weight) / nband : 1.0;   // :209, sw :243
  const double up_in_hr = SW ? 0.0 : 1.0;                         // shortwave heating rate from the direct beam only (:197)
  const double hrw2 = hr_weight * hr_weight;
  double jpart = 0.0;
  for (int t = threadIdx.x; t < nlay * nband; t += blockDim.x) {
    const int l = t / nband, b = t % nband;
    const double hrf = cv[l] * (s_bdn[(l + 1) * nband + b] - s_bdn[l * nband + b] -
                                up_in_hr * (s_bup[(l + 1) * nband + b] - s_bup[l * nband + b]));
    const double r = hrf - hrt[t];
    s_r[t] = r;
    jpart += spec_scale * hrw2 * lw[l] * r * r;
  }
  __syncthreads();
  // weights of the squared residuals of dn / up at half level i: spectral (s) and broadband (b)
  auto level_weights = [&](int i, double& wd_s, double& wu_s, double& wd_b, double& wu_b) {
    wd_s = 0.0; wu_s = 0.0; wd_b = 0.0; wu_b = 0.0;
    if (i == nlay) { wd_s = flux_weight; wd_b = flux_weight; }
    if (i == 0) {
      wu_s = SW ? 20.0 * flux_weight : flux_weight;              // calc_cost_function_sw.cpp:214
      wu_b = (!SW || all_pos) ? flux_weight : 0.0;               // :252
    }
    if (i >= 1 && i <= nlay - 1 && flux_profile_weight > 0.0) {
      const double iw = flux_profile_weight * 0.5 * (lw[i - 1] + lw[i]);
      wd_s = iw; wu_s = iw; wd_b = iw;
      wu_b = (!SW || all_pos) ? iw : 0.0;                        // :264
    }
  };
  for (int t = threadIdx.x; t < nlay + nhl; t += blockDim.x) {
    if (t < nlay) {
      double rsum = 0.0;
      for (int b = 0; b < nband; ++b) rsum += s_r[t * nband + b];
      s_rsum[t] = rsum;
      jpart += broadband_weight * hrw2 * lw[t] * rsum * rsum;
    } else {
      const int i = t - nlay;
      double wd_s, wu_s, wd_b, wu_b, sd = 0.0, su = 0.0;
      level_weights(i, wd_s, wu_s, wd_b, wu_b);
      if (wd_s != 0.0 || wu_s != 0.0) {
        for (int b = 0; b < nband; ++b) {
          sd += s_bdn[i * nband + b] - fdt[i * nband + b];
          su += s_bup[i * nband + b] - fut[i * nband + b];
        }
        jpart += broadband_weight * (wd_b * sd * sd + wu_b * su * su);
      }
      s_sd[i] = sd;
      s_su[i] = su;
    }
  }
  __syncthreads();
  {
    // (at most two rounds: nhl * nband <= 2 * blockDim is not required, the values wait in registers for the barrier)
    constexpr int MAXR = K8A_MAXR;
    double gd_r[MAXR], gu_r[MAXR];
    int nr = 0;
    for (int t = threadIdx.x; t < nhl * nband && nr < MAXR; t += blockDim.x, ++nr) {
      const int i = t / nband, b = t % nband;
      double gd = 0.0, gu = 0.0;
      {
      if (i >= 1) {                                               // layer i-1 ends at this level
        const int l = i - 1;
        const double dhr = 2.0 * hrw2 * lw[l] * (spec_scale * s_r[l * nband + b] + broadband_weight * s_rsum[l]);
        gd = dhr * cv[l];
        if (!SW) gu = -dhr * cv[l];
      }
      if (i <= nlay - 1) {                                        // layer i starts at it
        const double dhr = 2.0 * hrw2 * lw[i] * (spec_scale * s_r[i * nband + b] + broadband_weight * s_rsum[i]);
        gd += -dhr * cv[i];
        if (!SW) gu += dhr * cv[i];
      }
      double wd_s, wu_s, wd_b, wu_b;
      level_weights(i, wd_s, wu_s, wd_b, wu_b);
      if (wd_s != 0.0 || wu_s != 0.0) {
        const double dd = s_bdn[t] - fdt[t];
        const double du = s_bup[t] - fut[t];
        jpart += spec_scale * (wd_s * dd * dd + wu_s * du * du);
        gd += 2.0 * (spec_scale * wd_s * dd + broadband_weight * wd_b * s_sd[i]);
        gu += 2.0 * (spec_scale * wu_s * du + broadband_weight * wu_b * s_su[i]);
      }
      }
      gd_r[nr] = gd;
      gu_r[nr] = gu;
    }
    __syncthreads();
    nr = 0;
    for (int t = threadIdx.x; t < nhl * nband && nr < MAXR; t += blockDim.x, ++nr) {
      s_bdn[t] = gd_r[nr];                                        // from here on: dJ/d(band flux dn), dJ/d(band flux up)
      s_bup[t] = gu_r[nr];
    }
  }
  double* s_gdn = s_bdn;
  double* s_gup = s_bup;
  // ---- E: adjoint.  The owners take the forward fluxes at their cells before the rows are reused ----
  double c_f0[NC], c_f1[NC];      // longwave: dn_l, up_l+1; shortwave: dn_l+1, up_l
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int l = lgrp + j * nlgrp;
    c_f0[j] = 0.0; c_f1[j] = 0.0;
    if (in_range && l < nlay) {
      c_f0[j] = SW ? s_D[(l + 1) * ng + g] : s_D[l * ng + g];
      c_f1[j] = SW ? s_U[l * ng + g] : s_U[(l + 1) * ng + g];
    }
  }
  double g_dn_surf_extra = 0.0, g_up_toa_extra = 0.0;
  if (live && sfds && sfut && (SW || spectral_boundary_weight > 0.0)) {
    // per-g boundary terms: calc_cost_function_lw.cpp:223-229 on the un-banded fluxes; calc_cost_function_sw.cpp:271-274
    // with per-g (erythemal) weights on the surface direct flux
    const double reld = rel_flux ? rel_flux[((size_t)col * 2 * nhl + nlay) * ng + g] : 0.0;
    const double a = (s_D[nlay * ng + g] - reld) - sfds[(size_t)col * ng + g];
    if (SW) {
      const double wgt = sfut[(size_t)col * ng + g];
      jpart += wgt * a * a;
      g_dn_surf_extra = 2.0 * wgt * a;
    } else {
      const double relu = rel_flux ? rel_flux[((size_t)col * 2 * nhl + nhl) * ng + g] : 0.0;
      const double c = (s_U[g] - relu) - sfut[(size_t)col * ng + g];
      jpart += spectral_boundary_weight * (a * a + c * c);
      g_dn_surf_extra = 2.0 * spectral_boundary_weight * a;
      g_up_toa_extra = 2.0 * spectral_boundary_weight * c;
    }
  }
  __syncthreads();
  if constexpr (SW) {
    // the transmittances back into the rows the fluxes occupied
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int l = lgrp + j * nlgrp;
      if (in_range && l < nlay) {
        s_D[(l + 1) * ng + g] = c_t[j];
        s_U[l * ng + g] = c_tu[j];
      }
    }
    __syncthreads();
  }
  if (live) {
    const int b = band_of_g[g];
    // u_bar[0] = seed[0], u_bar[l+1] = seed[l+1] + u_bar[l] t_l: row l of s_U becomes the adjoint of up[l]
    double up_bar = s_gup[b] + g_up_toa_extra;
    double* pD = s_D + g;
    double* pU = s_U + g;
    const double* pT = s_T + g;
    const double* gU = s_gup + b;
    const double* gD = s_gdn + b;
    int l = 0;
    for (; l + K8A_CH <= nlay; l += K8A_CH) {
      double tt[K8A_CH], ss[K8A_CH];
#pragma unroll
      for (int q = 0; q < K8A_CH; ++q) {
        tt[q] = SW ? pU[(l + q) * ng] : pT[(l + q) * ng];
        ss[q] = gU[(l + q + 1) * nband];
      }
#pragma unroll
      for (int q = 0; q < K8A_CH; ++q) {
        pU[(l + q) * ng] = up_bar;
        up_bar = up_bar * tt[q] + ss[q];
      }
    }
    for (; l < nlay; ++l) {
      const double t = SW ? pU[l * ng] : pT[l * ng];
      pU[l * ng] = up_bar;
      up_bar = up_bar * t + gU[(l + 1) * nband];
    }
    // through the surface condition into dn[nlay]; row l+1 of s_D becomes the adjoint of dn[l+1]
    const double refl = SW ? (all_nonpos ? 0.0 : surf_emis[(size_t)col * nband + b]) : 1.0 - surf_emis[(size_t)col * nband + b];
    double dn_bar = gD[nlay * nband] + g_dn_surf_extra + up_bar * refl;
    l = nlay - 1;
    for (; l - (K8A_CH - 1) >= 0; l -= K8A_CH) {
      double tt[K8A_CH], ss[K8A_CH];
#pragma unroll
      for (int q = 0; q < K8A_CH; ++q) {
        tt[q] = SW ? pD[(l - q + 1) * ng] : pT[(l - q) * ng];
        ss[q] = gD[(l - q) * nband];
      }
#pragma unroll
      for (int q = 0; q < K8A_CH; ++q) {
        pD[(l - q + 1) * ng] = dn_bar;
        dn_bar = dn_bar * tt[q] + ss[q];
      }
    }
    for (; l >= 0; --l) {
      const double t = SW ? pD[(l + 1) * ng] : pT[l * ng];
      pD[(l + 1) * ng] = dn_bar;
      dn_bar = dn_bar * t + gD[l * nband];
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int l = lgrp + j * nlgrp;
    if (in_range && l < nlay && !((clamp_mask >> j) & 1u)) {
      const double ubar_l = s_U[l * ng + g], dnbar = s_D[(l + 1) * ng + g];
      double tau_bar;
      if constexpr (SW) {
        // up_l = up_l+1 exp(-2 tau_l), dn_l+1 = dn_l exp(-tau_l / mu0)
        tau_bar = ubar_l * c_f1[j] * (-2.0) + dnbar * c_f0[j] * minus_sec_sza;
      } else {
        const double tau = c_tau[j], ex = c_t[j];
        const double eps = 1.0 - ex;
        const bool thick = eps > 1.0e-5;
        const double rtau = thick ? ecckd::div_fast(1.0 / kD, tau) : 0.0;      // 1 / (D tau)
        const double b0 = pl[l * ng + g], b1 = pl[(l + 1) * ng + g];
        //   dn_{l+1} = dn_l*(1-eps) + B_l*(eps-fac) + B_{l+1}*fac
        //   up_l     = up_{l+1}*(1-eps) + B_{l+1}*(eps-fac) + B_l*fac
        const double t_bar = dnbar * c_f0[j] + ubar_l * c_f1[j];
        const double a_bar = dnbar * b0 + ubar_l * b1;
        const double f_bar = dnbar * b1 + ubar_l * b0;
        const double eps_bar = -t_bar + a_bar;
        const double fac_bar = f_bar - a_bar;
        // fac(eps, tau): thick: 1 - eps/(D tau); thin: eps/2   (radiative_transfer_lw.cpp:42-43)
        const double dfac_deps = thick ? -rtau : 0.5;
        const double dfac_dtau = thick ? eps * rtau * (kD * rtau) : 0.0;        // eps / (D tau^2)
        tau_bar = (eps_bar + fac_bar * dfac_deps) * (kD * ex) + fac_bar * dfac_dtau;
      }
      dtau[(cell0 + l) * ng + g] = tau_bar;
    }
  }
  jpart += negative_od_penalty * penalty;
  const double jtot = block_reduce_sum(jpart, s_red);
  if (threadIdx.x == 0) jcol[col] = jtot;
}

// K8b + K9: gradient of the state, two launches of one WAVE per task, four waves per block, no LDS, no barrier.
//   grad_k = sum over the cells that reference this node of coef * dtau[cell][g]   (reference order)
//   grad_x = grad_k * k + B^-1 (x - x_prior) / sigma_g^2                            (:273-283)
// The nodes are referenced very unevenly (a few hundred references on average, thousands where the training profiles crowd
// one corner of the table): with a block or a wave per NODE the kernel lasted as long as its busiest node.  So the references
// of a node are cut into runs of at most K8B_RUN (host: ecckd_opt_create) and
//   k_opt_gradient_gather: a wave per (run, 64 g points) adds its run in reference order -> run_sum[run][g];
//   k_opt_gradient_finish: a wave per (node, 64 g points) adds the node's runs in order, then chain rule and prior.
// Every sum has a fixed order: bitwise reproducible.  The references come 64 at a time, one (cell, coefficient) pair per lane,
// v_readlane hands them round as scalars, and the dJ/dtau rows are loaded eight at a time, two batches in flight.
// Which wave takes which task is a pure speed choice (the order tables built by ecckd_opt_create): the blocks b, b + 8, ...
// share an XCD and walk ONE eighth of the pressure axis in order, so the dJ/dtau rows they gather (cells whose pressure lies
// between two neighbouring nodes) tend to stay in that XCD's L2.
constexpr int K8B_WAVES = 4;
constexpr int K8B_RUN = 64;
__global__ void __launch_bounds__(64 * K8B_WAVES)
k_opt_gradient_gather(int nslot, int ng, int nchunk, const int* __restrict__ run_order /*[nslot]: run * nchunk + chunk, -1 = none*/,
                      const int* __restrict__ run_r0 /*[nrun + 1]: first reference of each run*/, const int* __restrict__ run_last /*[nrun]: 1 + last reference*/,
                      const int* __restrict__ ref_cell, const double* __restrict__ ref_coef, const double* __restrict__ dtau,
                      double* __restrict__ run_sum /*[nrun][nchunk][64]*/) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int vb = blockIdx.x; vb * K8B_WAVES < nslot; vb += gridDim.x) {        // gridDim is a multiple of 8: vb stays on its XCD
    const int slot = vb * K8B_WAVES + wave;
    const int ordered = slot < nslot ? run_order[slot] : -1;
    if (ordered < 0) continue;
    const int run = ordered / nchunk;
    const int g = (ordered % nchunk) * 64 + lane;
    const int gl = g < ng ? g : 0;
    const int r0 = run_r0[run], r1 = run_last[run];
    const int myr = r0 + lane;
    const int my_cell = myr < r1 ? ref_cell[myr] : -1;
    const double my_w = myr < r1 ? ref_coef[myr] : 0.0;
    const int cnt = r1 - r0;                                   // <= K8B_RUN = 64
    double gk = 0.0;
    // the issues are unconditional (lanes past the end hold cell -1 and weight 0: row 0 is fetched and dropped), so that the
    // compiler counts the loads in flight exactly
    GatherBatch A, B;
    gather_issue(A, my_cell, my_w, 0, cnt, dtau, gl, ng);
    for (int q0 = 0; q0 < cnt; q0 += 2 * K8A_GB) {
      gather_issue(B, my_cell, my_w, (q0 + K8A_GB) & 63, cnt, dtau, gl, ng);
#pragma unroll
      for (int q = 0; q < K8A_GB; ++q) gk += A.ix[q] >= 0 ? A.cv[q] * A.kv[q] : 0.0;
      gather_issue(A, my_cell, my_w, (q0 + 2 * K8A_GB) & 63, cnt, dtau, gl, ng);
      const bool use_b = q0 + K8A_GB < cnt;                    // (at q0 + 8 = 64 the issue above wrapped round to batch 0)
#pragma unroll
      for (int q = 0; q < K8A_GB; ++q) gk += (use_b && B.ix[q] >= 0) ? B.cv[q] * B.kv[q] : 0.0;
    }
    run_sum[(size_t)ordered * 64 + lane] = gk;
  }
}

__global__ void __launch_bounds__(64 * K8B_WAVES)
k_opt_gradient_finish(int nslot /*entries of node_order*/, int ng, int nchunk /*64-g chunks per node*/, const double* __restrict__ x,
                      const double* __restrict__ x_prior, const double* __restrict__ k,
                      const int* __restrict__ node_run0 /*[nnode + 1]: first run of each node*/, const double* __restrict__ run_sum,
                      // prior: per node the (at most 27) neighbours of its Kronecker stencil and their weights, -1 = none
                      const int* __restrict__ node_gas, const int* __restrict__ st_nb /*[nnode][27]*/,
                      const double* __restrict__ st_w /*[nnode][27]*/, const double* __restrict__ inv_sigma2 /*[ngas][ng]*/,
                      int have_prior, double* __restrict__ grad, double* __restrict__ jb_part /*[nnode][nchunk]*/,
                      const int* __restrict__ node_order /*[nslot]: node * nchunk + chunk of each wave, -1 = none*/) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int vb = blockIdx.x; vb * K8B_WAVES < nslot; vb += gridDim.x) {
    const int slot = vb * K8B_WAVES + wave;
    const int ordered = slot < nslot ? node_order[slot] : -1;
    if (ordered < 0) continue;
    const size_t node = (size_t)(ordered / nchunk);
    const int chunk = ordered % nchunk;
    const int g = chunk * 64 + lane;
    const bool active = g < ng;
    const int gl = active ? g : 0;
    // the node's runs, in order (eight loads in flight)
    double gk = 0.0;
    const int t1 = node_run0[node + 1];
    for (int t = node_run0[node]; t < t1; t += 8) {
      double v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = run_sum[((size_t)min(t + q, t1 - 1) * nchunk + chunk) * 64 + lane];
#pragma unroll
      for (int q = 0; q < 8; ++q) gk += (t + q < t1) ? v[q] : 0.0;
    }
    // B^-1 (x - x_prior) / sigma_g^2 at this node (calc_background_cost_function, ckd_model.cpp:852-865): the stencil's
    // neighbours come one per lane and go round as scalars, their 2 x 27 loads are in flight together, the terms are added in
    // the order of the host's loops (concentration, temperature, pressure offsets -1, 0, 1)
    double gb = 0.0;
    if (have_prior) {
      const int my_nb = lane < 27 ? st_nb[node * 27 + lane] : -1;
      const double my_w = lane < 27 ? st_w[node * 27 + lane] : 0.0;
      double acc = 0.0;
#pragma unroll
      for (int q0 = 0; q0 < 27; q0 += 9) {
        int nb[9];
        double w[9], xa[9], xb[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) { nb[q] = __builtin_amdgcn_readlane(my_nb, q0 + q); w[q] = readlane_f64(my_w, q0 + q); }
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          xa[q] = x[(size_t)max(nb[q], 0) * ng + gl];
          xb[q] = x_prior[(size_t)max(nb[q], 0) * ng + gl];
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) acc += nb[q] >= 0 ? w[q] * (xa[q] - xb[q]) : 0.0;
      }
      if (active) gb = acc * inv_sigma2[(size_t)node_gas[node] * ng + g];
    }
    double jb = 0.0;
    if (active) {
      const size_t e = node * ng + g;
      const double xv = x[e];
      double gx = 0.0;
      if (xv > MIN_X) {
        gx = gk * k[e];
        if (have_prior) {
          gx += gb;
          jb = 0.5 * (xv - x_prior[e]) * gb;
        }
        if (fabs(gx) < 1.0e-80) gx = 0.0;  // :286
      }
      grad[e] = gx;
    }
    // the node's share of the background cost: lanes in shuffle order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) jb += __shfl_down(jb, off, 64);
    if (lane == 0) jb_part[ordered] = jb;
  }
}

// k = exp(x) for the active part (:242-249)
__global__ void __launch_bounds__(256)
k_opt_exp(size_t nx, const double* __restrict__ x, double* __restrict__ k) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < nx) k[e] = (x[e] > MIN_X) ? exp(x[e]) : 0.0;
}


// ---------------------------------------------------------------------------------------
// Device-resident L-BFGS vector kernels.  All reductions are two-stage with a fixed order:
// stage 1 writes one partial per block (lanes by shuffle, then the block's waves in order), stage 2 (inside the consumer
// kernel, or k_lb_finish) sums the partials in index order.
constexpr int VEC_BLOCKS = 512;    // room for partials per reduction; the launches use min(512, ceil(n / 1024)) blocks
constexpr int VEC_THREADS = 1024;  // ONE element per thread at nx ~ 3e5: every load of a pass is in flight at once
constexpr int VEC_WAVES = VEC_THREADS / 64;
constexpr int FIN_THREADS = 256;   // k_lb_finish

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// the nblk (<= 512) partials of one reduction, summed by every caller in the same order (blockDim >= 256)
__device__ __forceinline__ double sum_partials(const double* __restrict__ part, int nblk, double* s_tmp /*[4]*/) {
  const int t = threadIdx.x;
  double v = 0.0;
  if (t < 256) {
    if (t < nblk) v = part[t];
    if (t + 256 < nblk) v += part[t + 256];
  }
  v = wave_sum(v);
  __syncthreads();
  if ((t & 63) == 0 && t < 256) s_tmp[t >> 6] = v;
  __syncthreads();
  return ((s_tmp[0] + s_tmp[1]) + s_tmp[2]) + s_tmp[3];
}

// NACC per-thread accumulators -> one partial per (accumulator, block).  Within a wave the NACC sums are taken TOGETHER: at the
// exchange over lane distance 32 a lane passes on one half of its values and adds what it receives to the half it keeps, at
// distance 16 a half of those, ... - NACC + NACC/2 + ... exchanges in all instead of 6 per value (every exchange is an LDS
// crossbar operation that all the waves of a CU queue for: the plain form cost the update kernel more than its memory traffic).
// Lane l ends up with the wave's sum of accumulator bitreverse6(l).  Same tree for every value, fixed: bitwise reproducible.
template <int USED>
__device__ __forceinline__ void fold_level(double* v, bool upper, int mask) {
  constexpr int NEXT = (USED + 1) / 2;
#pragma unroll
  for (int i = 0; i < NEXT; ++i) {
    const double a = v[2 * i];
    const double b = (2 * i + 1 < USED) ? v[2 * i + 1] : 0.0;
    const double send = upper ? a : b;
    double keep = upper ? b : a;
    keep += __shfl_xor(send, mask, 64);
    v[i] = keep;
  }
}

template <int NACC>
__device__ __forceinline__ void write_partials(double (&acc)[NACC], int nused, double* __restrict__ part, double* s_all /*[NACC][VEC_WAVES]*/) {
  static_assert(NACC <= 64, "one value per lane at the end");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int U1 = (NACC + 1) / 2, U2 = (U1 + 1) / 2, U3 = (U2 + 1) / 2, U4 = (U3 + 1) / 2, U5 = (U4 + 1) / 2;
  fold_level<NACC>(acc, (lane & 32) != 0, 32);
  fold_level<U1>(acc, (lane & 16) != 0, 16);
  fold_level<U2>(acc, (lane & 8) != 0, 8);
  fold_level<U3>(acc, (lane & 4) != 0, 4);
  fold_level<U4>(acc, (lane & 2) != 0, 2);
  fold_level<U5>(acc, (lane & 1) != 0, 1);
  const int kk = (int)(__brev((unsigned)lane) >> 26);           // the accumulator this lane now holds
  if (kk < NACC) s_all[kk * VEC_WAVES + wave] = acc[0];
  __syncthreads();
  if ((int)threadIdx.x < nused) {
    double t = 0.0;
    for (int w = 0; w < VEC_WAVES; ++w) t += s_all[threadIdx.x * VEC_WAVES + w];
    part[(size_t)threadIdx.x * VEC_BLOCKS + blockIdx.x] = t;
  }
}

// ---- L-BFGS on the device, compact representation (Byrd, Nocedal & Schnabel 1994) -----------------------------
// With S = [s_0 .. s_{m-1}], Y = [y_0 .. y_{m-1}] (oldest first), R = upper triangle of S^T Y, D = diag(s_i.y_i):
//   H g = gamma g + S top - gamma Y u,   u = R^-1 (S^T g),   top = R^-T ((D + gamma Y^T Y) u - gamma Y^T g)
// which is the two-loop recursion in matrix form.  Per iteration: ONE pass that stores the new curvature pair and takes
// every dot product the update needs (k_lb_update), the m x m solves on the host, the direction in one more pass
// (k_lb_direction), the trial point and its coefficients exp(x) (k_lb_step): three vector launches and two waits of the host
// on pinned result slots (the dots; the cost of the trial point).
constexpr int LB_M = 6;
struct LbSlots { int n; int slot[LB_M]; };                       // ring slots of the pairs, oldest first
struct LbCoef { double gamma; double cs[LB_M]; double cy[LB_M]; };
constexpr int LB_NPART = 3 + 5 * LB_M;

// The accepted step x -> xn, g -> gn (new_slot >= 0): curvature pair s = xn - x, y = gn - g into its ring slot, and its dots
// with the pairs that stay (keep): partials [1+2M] s.y, [2+2M] y.y, [3+2M+a] s.Y_a, [3+3M+a] S_a.y, [3+4M+a] y.Y_a.
// Then, at the new point: q = projected gradient (0 at active bounds and at pinned elements) and partials [0] |q|^2,
// [1+a] S_a.q, [1+M+a] Y_a.q for the pairs that stay and, as the newest (a = keep.n), the pair just stored.
// new_slot < 0 (first iteration): only q and |q|^2 at (xn, gn).
__global__ void __launch_bounds__(VEC_THREADS)
k_lb_update(size_t n, LbSlots keep, int new_slot, const double* __restrict__ x, const double* __restrict__ xn,
            const double* __restrict__ g, const double* __restrict__ gn, const double* __restrict__ xmin,
            const double* __restrict__ xmax, double* __restrict__ q, double* __restrict__ S, double* __restrict__ Y,
            double* __restrict__ part) {
  __shared__ double s_all[LB_NPART * VEC_WAVES];
  double acc[LB_NPART];
#pragma unroll
  for (int k = 0; k < LB_NPART; ++k) acc[k] = 0.0;
  const bool pair = new_slot >= 0;
  for (size_t i = (size_t)blockIdx.x * VEC_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * VEC_THREADS) {
    // every load of the element first, none behind a test of the number of pairs (a branch per pair put a full memory round
    // trip behind each): the slots of unused pairs hold zeros or the finite values of dropped pairs (S and Y are cleared when
    // they are allocated), their partials are never read
    double sa[LB_M], ya[LB_M];
#pragma unroll
    for (int a = 0; a < LB_M; ++a) {
      sa[a] = S[(size_t)keep.slot[a] * n + i];
      ya[a] = Y[(size_t)keep.slot[a] * n + i];
    }
    const double xi = xn[i];
    double gi = gn[i];
    const double xo = x[i], go = g[i];
    const double lo = xmin ? xmin[i] : 0.0, hi = xmin ? xmax[i] : 0.0;
    const double si = pair ? xi - xo : 0.0, yi = pair ? gi - go : 0.0;
    if (xmin && ((xi <= lo && gi > 0.0) || (xi >= hi && gi < 0.0))) gi = 0.0;
    if (!(xi > MIN_X)) gi = 0.0;
    q[i] = gi;
    acc[0] += gi * gi;
    acc[1 + 2 * LB_M] += si * yi;
    acc[2 + 2 * LB_M] += yi * yi;
#pragma unroll
    for (int a = 0; a < LB_M; ++a) {
      // position keep.n is the pair being stored: the newest of this iteration's direction
      const double sq = (a == keep.n && pair) ? si : sa[a];
      const double yq = (a == keep.n && pair) ? yi : ya[a];
      // (positions behind the pairs in use alias ring slot 0: dropped by a select, never multiplied)
      const bool used = a < keep.n || (a == keep.n && pair);
      acc[1 + a] += used ? sq * gi : 0.0;
      acc[1 + LB_M + a] += used ? yq * gi : 0.0;
      acc[3 + 2 * LB_M + a] += a < keep.n ? si * ya[a] : 0.0;
      acc[3 + 3 * LB_M + a] += a < keep.n ? sa[a] * yi : 0.0;
      acc[3 + 4 * LB_M + a] += a < keep.n ? yi * ya[a] : 0.0;
    }
    if (pair) {
      S[(size_t)new_slot * n + i] = si;     // the slot being written is never among the pairs that stay
      Y[(size_t)new_slot * n + i] = yi;
    }
  }
  write_partials<LB_NPART>(acc, LB_NPART, part, s_all);
}

// d = -gamma q - sum cs_a S_a + sum cy_a Y_a; partials of d.g and d.d
__global__ void __launch_bounds__(VEC_THREADS)
k_lb_direction(size_t n, LbSlots sl, LbCoef cf, const double* __restrict__ q, const double* __restrict__ S,
               const double* __restrict__ Y, const double* __restrict__ g, double* __restrict__ d,
               double* __restrict__ part_dg /* part_dd follows: [2][VEC_BLOCKS] */) {
  __shared__ double s_all[2 * VEC_WAVES];
  double acc[2] = {0.0, 0.0};
  for (size_t i = (size_t)blockIdx.x * VEC_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * VEC_THREADS) {
    // (no test of the number of pairs round the LOADS - a branch per pair put a memory round trip behind each -; an unused
    // position, which aliases ring slot 0, is dropped by a select: 0 x a stale non-finite value would be NaN)
    double sv[LB_M], yv[LB_M];
#pragma unroll
    for (int k = 0; k < LB_M; ++k) {
      sv[k] = S[(size_t)sl.slot[k] * n + i];
      yv[k] = Y[(size_t)sl.slot[k] * n + i];
    }
    double di = -cf.gamma * q[i];
#pragma unroll
    for (int k = 0; k < LB_M; ++k) di += (k < sl.n) ? cf.cy[k] * yv[k] - cf.cs[k] * sv[k] : 0.0;
    // a variable held by an active bound (or pinned) does not move: without this the slope d.g of the Armijo test would
    // count a decrease that the clamped trial point cannot deliver
    if (q[i] == 0.0 && g[i] != 0.0) di = 0.0;
    d[i] = di;
    acc[0] += di * g[i];
    acc[1] += di * di;
  }
  write_partials<2>(acc, 2, part_dg, s_all);
}

// xn = clamp(x + step*d), pinned elements stay; kn = exp(xn), the coefficients the cost function starts from (0 where pinned,
// solve_adept.cpp:242-249).  step < 0: chosen here from |d| (every block reduces the same partials in the same order):
// first iteration min(1, 1/|d|), later 1, never longer than max_step (solve_adept.cpp:331), times `hint`; out[0] = step, out[1] = d.d,
// out[2] = d.g (pinned host slots)
__global__ void __launch_bounds__(VEC_THREADS)
k_lb_step(size_t n, double step, int first, double max_step, double hint, const double* __restrict__ part_dg,
          const double* __restrict__ part_dd, const double* __restrict__ x, const double* __restrict__ d,
          const double* __restrict__ xmin, const double* __restrict__ xmax, double* __restrict__ xn,
          double* __restrict__ kn, double* __restrict__ out) {
  __shared__ double s_tmp[4];
  if (step < 0.0) {
    const double dd = sum_partials(part_dd, gridDim.x, s_tmp);
    const double dn = sqrt(dd);
    step = first ? fmin(1.0, 1.0 / fmax(dn, 1e-300)) : 1.0;
    if (step * dn > max_step) step = max_step / dn;
    step *= hint;             // the fraction of that step the line search starts from (ecckd_opt_minimize)
    if (blockIdx.x == 0) {
      const double dg = sum_partials(part_dg, gridDim.x, s_tmp);
      if (threadIdx.x == 0) { out[0] = step; out[1] = dd; out[2] = dg; }
    }
  }
  for (size_t i = (size_t)blockIdx.x * VEC_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * VEC_THREADS) {
    double v = x[i] + step * d[i];
    if (xmin) v = fmin(fmax(v, xmin[i]), xmax[i]);
    const bool free_el = x[i] > MIN_X;
    if (!free_el) v = x[i];
    xn[i] = v;
    kn[i] = free_el ? exp(v) : 0.0;
  }
}

// out[k] = sum of partial array k: one block per array.  `out` is pinned, host-coherent memory: the host watches the slots
// (no copy, no stream synchronisation; see wait_slots)
__global__ void __launch_bounds__(FIN_THREADS)
k_lb_finish(const double* __restrict__ part, int nblk, double* __restrict__ out) {
  __shared__ double s_tmp[4];
  const int k = blockIdx.x;
  const double v = sum_partials(part + (size_t)k * VEC_BLOCKS, nblk, s_tmp);
  if (threadIdx.x == 0) out[k] = v;
}

// The cost of an evaluation: the profiles' costs, then the prior's node terms, each summed in a fixed order (thread t takes
// the elements t, t + 256, ...; the block combines in thread order), delivered to the host's slot and, for the profile-sharded
// all-reduce, into the slot behind the gradient.
__global__ void __launch_bounds__(256)
k_opt_sum_cost(const double* __restrict__ jcol, size_t ncol, const double* __restrict__ jb, size_t nnode, int prior,
               double* __restrict__ h_out /*pinned*/, double* __restrict__ d_out /*or NULL*/) {
  __shared__ double s_tmp[4];
  double a = 0.0, b = 0.0;
  // a thread's elements are loaded together (one memory round trip instead of one per element: the kernel is a single block
  // whose time is its chain of loads) and added in the order they always were
  constexpr int U = 8;
  for (size_t i0 = threadIdx.x; i0 < ncol; i0 += (size_t)256 * U) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t i = i0 + (size_t)256 * u; v[u] = jcol[i < ncol ? i : ncol - 1]; }
#pragma unroll
    for (int u = 0; u < U; ++u) if (i0 + (size_t)256 * u < ncol) a += v[u];
  }
  if (prior && nnode > 0)
    for (size_t i0 = threadIdx.x; i0 < nnode; i0 += (size_t)256 * U) {
      double v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { const size_t i = i0 + (size_t)256 * u; v[u] = jb[i < nnode ? i : nnode - 1]; }
#pragma unroll
      for (int u = 0; u < U; ++u) if (i0 + (size_t)256 * u < nnode) b += v[u];
    }
  double r[2] = {a, b};
  double tot[2];
  for (int k = 0; k < 2; ++k) {
    double v = r[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = v;
    __syncthreads();
    tot[k] = ((s_tmp[0] + s_tmp[1]) + s_tmp[2]) + s_tmp[3];
  }
  if (threadIdx.x == 0) {
    const double J = tot[0] + tot[1];
    if (d_out) *d_out = J;
    *h_out = J;
  }
}

}  // namespace

struct ecckd_opt {
  ecckd_ctx* ctx = nullptr;
  int ng = 0, nt = 0, np = 0, nband = 0, ngas = 0, nlay = 0, nent = 0;
  size_t nx = 0, nk = 0, ncol = 0, ncell = 0, nnode_active = 0;
  std::vector<GasInfo> gases;       // in k order: active gases first
  std::vector<int> user_to_k;       // user gas index -> position in `gases`
  ecckd_opt_config cfg{};
  bool have_prior = false;
  bool have_boundary = false;
  bool do_sw = false;
  int eval_ray_ent = -1;   // >= 0 only inside ecckd_run_ckd
  int eval_keep_negative = 0;
  double* d_rel = nullptr;   // relative-to CKD fluxes [ncol][2][nhl][ng]
  int ray_ent = -1;        // entry index of the Rayleigh pseudo gas
  double* d_mu0 = nullptr;
  // host copies
  std::vector<double> h_k0, h_kmin, h_kmax;
  // device
  double *d_k = nullptr, *d_x = nullptr, *d_xprior = nullptr, *d_grad = nullptr, *d_dtau = nullptr;
  double *d_jcol = nullptr, *d_jb = nullptr;
  int* d_ent_idx = nullptr; double* d_ent_coef = nullptr; int* d_band = nullptr;
  int *d_band_ptr = nullptr, *d_band_g = nullptr;   // the g points of each band in increasing order (CSR)
  int* d_node_order = nullptr;                      // K8b finish: wave slot -> node * nchunk + chunk (XCD-aware), -1 = idle
  int *d_run_order = nullptr, *d_run_r0 = nullptr, *d_run_last = nullptr, *d_node_run0 = nullptr;   // K8b gather: the runs
  double* d_run_sum = nullptr;                      // [nrun][nchunk][64]
  unsigned gradient_grid = 0, gather_grid = 0;
  int grad_nchunk = 1, grad_nslot = 0, gather_nslot = 0;
  size_t grad_nrun = 0;
  double *d_planck = nullptr, *d_semis = nullptr, *d_conv = nullptr, *d_lw = nullptr;
  double *d_hr = nullptr, *d_fdn = nullptr, *d_fup = nullptr, *d_sfds = nullptr, *d_sfut = nullptr;
  int *d_ref_ptr = nullptr, *d_ref_cell = nullptr; double* d_ref_coef = nullptr;
  int *d_node_gas = nullptr, *d_st_nb = nullptr;    // K9: gas of each active node, its stencil's neighbours [nnode][27]
  double *d_st_w = nullptr, *d_inv_sigma2 = nullptr;
  double* d_od_out = nullptr; double* d_flux_out = nullptr;
  unsigned grad_blocks = 0;
  // pinned host staging: per-profile costs, per-node prior terms, L-BFGS scalars (pageable read-backs cost
  // ~30 us each through the runtime's staging path)
  // pinned, host-coherent result slots that the kernels write and the host watches: [0, 64) L-BFGS dot products, [64, 67)
  // step / d.d / d.g, [70] the cost of an evaluation
  double* h_pin = nullptr;
  double* d_pin = nullptr;    // the same memory as the device sees it
  double* h_rb = nullptr;     // = h_pin
  // device L-BFGS workspace (allocated by ecckd_opt_minimize)
  double *d_xmin = nullptr, *d_xmax = nullptr, *d_xn = nullptr, *d_gn = nullptr, *d_dir = nullptr, *d_q = nullptr;
  double *d_S = nullptr, *d_Y = nullptr, *d_part = nullptr, *d_sc = nullptr;
  // profile-sharded optimisation: sum of [gradient, J] over the ranks (ecckd_opt_set_allreduce)
  ecckd_allreduce_fn reduce_fn = nullptr;
  void* reduce_user = nullptr;
  bool add_prior = true;     // exactly one rank contributes the prior term
  // timing
  long long n_eval = 0;
  // progress line and activity timers of solve_adept.cpp (:216-218 "minimizer", "a-priori", "radiative transfer"; :295-299)
  ecckd_evaluator_fn eval_fn = nullptr;    // cost / gradient supplied by the caller instead of the device kernels
  void* eval_user = nullptr;
  std::vector<double> eval_x, eval_g;
  ecckd_progress_fn progress_fn = nullptr;
  void* progress_user = nullptr;
  hipEvent_t tev[3] = {nullptr, nullptr, nullptr};
  double t_rt = 0.0, t_prior = 0.0, t_minimizer = 0.0;
};

namespace {

template <typename T>
int upload(ecckd_ctx* ctx, T** d, const std::vector<T>& h) {
  ECCKD_HIP_CHECK(hipMalloc((void**)d, std::max<size_t>(h.size(), 1) * sizeof(T)));
  if (!h.empty())
    ECCKD_HIP_CHECK(hipMemcpyAsync(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return ECCKD_OK;
}

void opt_free(ecckd_opt* o) {
  if (!o) return;
  if (o->ctx) (void)hipStreamSynchronize(o->ctx->stream);
  void* ptrs[] = {o->d_k, o->d_x, o->d_xprior, o->d_grad, o->d_dtau, o->d_jcol, o->d_jb, o->d_ent_idx, o->d_ent_coef,
                  o->d_band, o->d_band_ptr, o->d_band_g, o->d_node_order, o->d_run_order, o->d_run_r0, o->d_run_last, o->d_node_run0,
                  o->d_run_sum, o->d_planck, o->d_semis, o->d_conv, o->d_lw, o->d_hr, o->d_fdn, o->d_fup, o->d_sfds,
                  o->d_sfut, o->d_ref_ptr, o->d_ref_cell, o->d_ref_coef, o->d_node_gas, o->d_st_nb, o->d_st_w, o->d_inv_sigma2, o->d_od_out, o->d_flux_out, o->d_xmin, o->d_xmax,
                  o->d_xn, o->d_gn, o->d_dir, o->d_q, o->d_S, o->d_Y, o->d_part, o->d_sc, o->d_mu0, o->d_rel};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (o->h_pin) (void)hipHostFree(o->h_pin);
  for (hipEvent_t e : o->tev)
    if (e) (void)hipEventDestroy(e);
  delete o;
}

// CkdModel::calc_planck_function, ckd_model.cpp:1121-1145
void planck_lut(int ntp, const double* tpl, const double* pf, int ng, double t, double* out) {
  const double d_t = tpl[1] - tpl[0], t0 = tpl[0];
  const double tindex0 = (t - t0) / d_t;
  if (tindex0 >= 0) {
    int it0 = std::min((int)tindex0, ntp - 2);
    const double w1 = tindex0 - it0, w0 = 1.0 - w1;
    for (int g = 0; g < ng; ++g) out[g] = w0 * pf[(size_t)it0 * ng + g] + w1 * pf[(size_t)(it0 + 1) * ng + g];
  } else {
    for (int g = 0; g < ng; ++g) out[g] = (t / t0) * pf[g];
  }
}

}  // namespace

extern "C" {

int ecckd_opt_create(ecckd_ctx* ctx, const ecckd_opt_model* m, int nscene, const ecckd_opt_scene* scenes,
                     const ecckd_opt_config* cfg, ecckd_opt** out) {
  ECCKD_REQUIRE(ctx && m && scenes && cfg && out && nscene > 0, "ecckd_opt_create: NULL/empty argument");
  *out = nullptr;
  ECCKD_REQUIRE(m->ng > 0 && m->ng <= 1024 && m->nt >= 2 && m->np >= 2 && m->ngas > 0 && m->gases &&
                    m->log_pressure && m->temperature && m->iband_per_g,
                "ecckd_opt_create: bad model dimensions (ng=%d nt=%d np=%d ngas=%d)", m->ng, m->nt, m->np, m->ngas);
  const bool do_sw = m->solar_irradiance != nullptr;
  ECCKD_REQUIRE(do_sw || (m->ntp >= 2 && m->temperature_planck && m->planck_function),
                "ecckd_opt_create: Planck look-up table missing");
  ECCKD_REQUIRE(!m->logarithmic_interpolation,
                "ecckd_opt_create: logarithmic LUT interpolation is not supported (ckd_model.h:359 default is linear)");
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  const int ng = m->ng, nt = m->nt, np = m->np;
  ecckd_opt* o = new ecckd_opt();
  o->ctx = ctx;
  o->ng = ng; o->nt = nt; o->np = np; o->ngas = m->ngas;
  o->cfg = *cfg;
  o->do_sw = do_sw;
  int nband = 0;
  for (int g = 0; g < ng; ++g) nband = std::max(nband, m->iband_per_g[g] + 1);
  o->nband = nband;

  // ---- coefficient vector: active gases first (they are the state x), then fixed gases ----
  std::vector<int> order;
  for (int a = 1; a >= 0; --a)
    for (int i = 0; i < m->ngas; ++i)
      if ((m->gases[i].is_active != 0) == (a == 1)) order.push_back(i);
  o->user_to_k.assign(m->ngas, -1);
  size_t off = 0;
  for (size_t pos = 0; pos < order.size(); ++pos) {
    const ecckd_opt_gas& ug = m->gases[order[pos]];
    GasInfo gi;
    gi.conc = ug.conc_dependence;
    gi.active = ug.is_active != 0;
    gi.nconc = (ug.conc_dependence == 2) ? ug.nconc : 1;
    if (!(ug.conc_dependence >= 0 && ug.conc_dependence <= 3) || !ug.molar_abs ||
        (ug.conc_dependence == 2 && (ug.nconc < 2 || !ug.vmr))) {
      opt_free(o);
      return ecckd::fail(ECCKD_PARAMETER_ERROR, "ecckd_opt_create: gas %d is ill-defined", order[pos]);
    }
    gi.nnode = (size_t)gi.nconc * nt * np;
    gi.ix = off;
    gi.reference_vmr = ug.reference_vmr;
    if (ug.conc_dependence == 2) gi.vmr.assign(ug.vmr, ug.vmr + ug.nconc);
    off += gi.nnode * ng;
    if (gi.active) { o->nx = off; o->nnode_active += gi.nnode; }
    o->user_to_k[order[pos]] = (int)pos;
    o->gases.push_back(gi);
  }
  // Rayleigh scattering (calc_total_optical_depth, solve_adept.cpp:33-36; ckd_model.h:242-252): a fixed
  // one-node pseudo gas at the end of the coefficient vector
  const bool have_rayleigh = do_sw && m->rayleigh_molar_scattering != nullptr;
  const size_t rayleigh_ix = off;
  if (have_rayleigh) off += ng;
  o->nk = off;
  ECCKD_REQUIRE(o->nx > 0, "ecckd_opt_create: no active gas to optimise");
  ECCKD_REQUIRE(o->nk < (size_t)0x7fffffff, "ecckd_opt_create: coefficient vector too long for int32 indexing");
  o->h_k0.assign(o->nk, 0.0);
  o->h_kmin.assign(o->nx, 0.0);
  o->h_kmax.assign(o->nx, 0.0);
  bool have_minmax = true;
  for (size_t pos = 0; pos < order.size(); ++pos) {
    const ecckd_opt_gas& ug = m->gases[order[pos]];
    const GasInfo& gi = o->gases[pos];
    std::memcpy(&o->h_k0[gi.ix], ug.molar_abs, gi.nnode * ng * sizeof(double));
    if (gi.active) {
      if (ug.min_molar_abs && ug.max_molar_abs) {
        std::memcpy(&o->h_kmin[gi.ix], ug.min_molar_abs, gi.nnode * ng * sizeof(double));
        std::memcpy(&o->h_kmax[gi.ix], ug.max_molar_abs, gi.nnode * ng * sizeof(double));
      } else {
        have_minmax = false;
      }
    }
  }
  if (!have_minmax) { o->h_kmin.clear(); o->h_kmax.clear(); }
  if (have_rayleigh) std::memcpy(&o->h_k0[rayleigh_ix], m->rayleigh_molar_scattering, ng * sizeof(double));

  // cap_relative_linear_coeffts(0.8), optimize_lut.cpp:185, ckd_model.cpp:883-917
  {
    int ibg = -1;
    for (size_t pos = 0; pos < o->gases.size(); ++pos)
      if (o->gases[pos].conc == 0) ibg = (int)pos;  // the LAST gas with NONE, as the reference loop leaves it
    if (ibg >= 0 && cfg->cap_relative_linear > 0.0) {
      const GasInfo& bg = o->gases[ibg];
      for (GasInfo& gi : o->gases) {
        if (gi.active && gi.conc == 3 && gi.nnode == bg.nnode) {
          for (size_t e = 0; e < gi.nnode * ng; ++e) {
            const double cap = o->h_k0[bg.ix + e] / (gi.reference_vmr * cfg->cap_relative_linear);
            if (o->h_k0[gi.ix + e] * (gi.reference_vmr * cfg->cap_relative_linear) > o->h_k0[bg.ix + e])
              o->h_k0[gi.ix + e] = std::min(o->h_k0[gi.ix + e], cap);
          }
        }
      }
    }
  }

  // ---- prior (create_error_covariances, ckd_model.cpp:646-832) ----
  o->have_prior = true;  // the reference always adds calc_background_cost_function (:273)
  std::vector<double> tri;
  std::vector<int> tri_off(o->gases.size(), 0), gas_dims(o->gases.size() * 4, 0);
  std::vector<double> inv_sigma2(o->gases.size() * ng, 0.0);
  std::vector<int> node_gas, node_ic, node_it, node_ip;
  {
    size_t node0 = 0;
    for (size_t pos = 0; pos < o->gases.size(); ++pos) {
      GasInfo& gi = o->gases[pos];
      gas_dims[pos * 4 + 0] = gi.nconc; gas_dims[pos * 4 + 1] = nt; gas_dims[pos * 4 + 2] = np;
      gas_dims[pos * 4 + 3] = (int)node0;
      if (!gi.active) continue;
      tri_off[pos] = (int)tri.size();
      const Tri tc = ar1_inverse(gi.nconc, cfg->conc_corr), tt = ar1_inverse(nt, cfg->temperature_corr),
                tp = ar1_inverse(np, cfg->pressure_corr);
      for (const Tri* t : {&tc, &tt, &tp}) {
        tri.insert(tri.end(), t->lo.begin(), t->lo.end());
        tri.insert(tri.end(), t->di.begin(), t->di.end());
        tri.insert(tri.end(), t->up.begin(), t->up.end());
      }
      // background_error per g: fixed prior_error, or estimated from min/max (:679-745)
      gi.sigma.assign(ng, cfg->prior_error > 0.0 ? cfg->prior_error : 1.0);
      if (cfg->prior_error <= 0.0 && have_minmax) {
        for (int g = 0; g < ng; ++g) {
          double local_sum = 0.0;
          int local_count = 0;
          for (size_t nd = 0; nd < gi.nnode; ++nd) {
            const size_t e = gi.ix + nd * ng + g;
            if (o->h_k0[e] > 0.0) {
              if (o->h_kmin[e] > 0.0) local_sum += 0.25 * std::log(o->h_kmax[e] / o->h_kmin[e]);
              else local_sum += 0.5 * std::log(o->h_kmax[e] / o->h_k0[e]);
              ++local_count;
            }
          }
          if (local_count > 0) gi.sigma[g] = cfg->prior_error_scaling * local_sum / local_count;
        }
        if (cfg->min_prior_error > 0.0) for (double& sg : gi.sigma) sg = std::max(cfg->min_prior_error, sg);
        if (cfg->max_prior_error > 0.0) for (double& sg : gi.sigma) sg = std::min(sg, cfg->max_prior_error);
      }
      for (int g = 0; g < ng; ++g) inv_sigma2[pos * ng + g] = 1.0 / (gi.sigma[g] * gi.sigma[g]);
      for (int ic = 0; ic < gi.nconc; ++ic)
        for (int it = 0; it < nt; ++it)
          for (int ip = 0; ip < np; ++ip) {
            node_gas.push_back((int)pos); node_ic.push_back(ic); node_it.push_back(it); node_ip.push_back(ip);
          }
      node0 += gi.nnode;
    }
  }

  // ---- training scenes -> flat per-profile arrays and the sparse interpolation table ----
  const int nlay = scenes[0].nlay;
  if (nlay < 1 || nlay > 128) {
    opt_free(o);
    return ecckd::fail(ECCKD_PARAMETER_ERROR, "ecckd_opt_create: nlay = %d outside the supported range 1..128", nlay);
  }
  o->nlay = nlay;
  const int nhl = nlay + 1;
  size_t ncol = 0;
  for (int s = 0; s < nscene; ++s) {
    if (scenes[s].nlay != nlay || scenes[s].ncol <= 0 || !scenes[s].pressure_hl || !scenes[s].temperature_hl ||
        !scenes[s].flux_dn || !scenes[s].flux_up || scenes[s].nband != nband ||
        (do_sw && (!scenes[s].mu0 || !scenes[s].albedo || !(scenes[s].tsi > 0.0)))) {
      opt_free(o);
      return ecckd::fail(ECCKD_PARAMETER_ERROR, "ecckd_opt_create: scene %d is inconsistent (nlay/nband/arrays)", s);
    }
    ncol += scenes[s].ncol;
  }
  o->ncol = ncol;
  o->ncell = ncol * nlay;
  int nent = have_rayleigh ? 1 : 0;
  for (const GasInfo& gi : o->gases) nent += (gi.conc == 2) ? 8 : 4;
  o->ray_ent = have_rayleigh ? nent - 1 : -1;
  o->nent = nent;
  std::vector<int> ent_idx(o->ncell * nent, -1);
  std::vector<double> ent_coef(o->ncell * nent, 0.0);
  std::vector<double> planck(ncol * nhl * ng), semis(ncol * nband, 1.0), conv(ncol * nlay), lwv(ncol * nlay);
  std::vector<double> hr(ncol * nlay * nband), fdn(ncol * nhl * nband), fup(ncol * nhl * nband);
  std::vector<double> sfds, sfut;
  bool any_boundary = false;
  for (int s = 0; s < nscene; ++s)
    any_boundary = any_boundary || (scenes[s].spectral_flux_dn_surf &&
                                    (do_sw ? scenes[s].spectral_boundary_weights != nullptr : scenes[s].spectral_flux_up_toa != nullptr));
  std::vector<double> mu0v(do_sw ? ncol : 0);
  double solar_sum = 0.0;
  if (do_sw)
    for (int g = 0; g < ng; ++g) solar_sum += m->solar_irradiance[g];
  o->have_boundary = any_boundary;
  if (any_boundary) { sfds.assign(ncol * ng, 0.0); sfut.assign(ncol * ng, 0.0); }

  const double log_p_0 = m->log_pressure[0];
  const double d_log_p = m->log_pressure[1] - m->log_pressure[0];
  const double d_t = m->temperature[1 * np + 0] - m->temperature[0];
  const double global_weight = 1.0 / (ECCKD_ACCEL_GRAVITY * 0.001 * 28.970);
  size_t col0 = 0;
  for (int s = 0; s < nscene; ++s) {
    const ecckd_opt_scene& sc = scenes[s];
    for (int c = 0; c < sc.ncol; ++c) {
      const size_t col = col0 + c;
      const double* p = sc.pressure_hl + (size_t)c * nhl;
      const double* T = sc.temperature_hl + (size_t)c * nhl;
      if (!do_sw) {
        for (int i = 0; i < nhl; ++i) planck_lut(m->ntp, m->temperature_planck, m->planck_function, ng, T[i], &planck[(col * nhl + i) * ng]);
        if (sc.surf_emissivity) std::memcpy(&semis[col * nband], sc.surf_emissivity + (size_t)c * nband, nband * sizeof(double));
      } else {
        // tsi_scaling * ckd_model.solar_irradiance() (solve_adept.cpp:176,186) in row 0; band albedo
        const double tsi_scaling = sc.tsi / solar_sum;
        for (int g = 0; g < ng; ++g) planck[(col * nhl) * ng + g] = tsi_scaling * m->solar_irradiance[g];
        std::memcpy(&semis[col * nband], sc.albedo, nband * sizeof(double));
        mu0v[col] = sc.mu0[c];
      }
      // layer weights, solve_adept.cpp:131-143
      double wsum = 0.0;
      for (int l = 0; l < nlay; ++l) {
        double w;
        if (cfg->pressure_weight_power == 0.5) w = std::sqrt(p[l + 1]) - std::sqrt(p[l]);
        else if (cfg->pressure_weight_power == 1.0) w = p[l + 1] - p[l];
        else w = std::pow(p[l + 1], cfg->pressure_weight_power) - std::pow(p[l], cfg->pressure_weight_power);
        lwv[col * nlay + l] = w;
        wsum += w;
        conv[col * nlay + l] = -(ECCKD_ACCEL_GRAVITY / ECCKD_SPECIFIC_HEAT_AIR) / (p[l + 1] - p[l]);
      }
      for (int l = 0; l < nlay; ++l) lwv[col * nlay + l] /= wsum;
      // truth: band fluxes and their heating rate (lbl_fluxes.cpp:380-388, heating_rate.h:30-50)
      std::memcpy(&fdn[col * nhl * nband], sc.flux_dn + (size_t)c * nhl * nband, (size_t)nhl * nband * sizeof(double));
      std::memcpy(&fup[col * nhl * nband], sc.flux_up + (size_t)c * nhl * nband, (size_t)nhl * nband * sizeof(double));
      for (int l = 0; l < nlay; ++l)
        for (int b = 0; b < nband; ++b) {
          const double* d = &fdn[col * nhl * nband];
          const double* u = &fup[col * nhl * nband];
          // LW: net flux divergence; SW: direct beam only (lbl_fluxes.cpp:365-369 passes an empty flux_up)
          hr[(col * nlay + l) * nband + b] =
              do_sw ? conv[col * nlay + l] * (d[(l + 1) * nband + b] - d[l * nband + b])
                    : conv[col * nlay + l] * (d[(l + 1) * nband + b] - d[l * nband + b] - u[(l + 1) * nband + b] + u[l * nband + b]);
        }
      if (any_boundary && sc.spectral_flux_dn_surf && (do_sw ? sc.spectral_boundary_weights != nullptr : sc.spectral_flux_up_toa != nullptr)) {
        std::memcpy(&sfds[col * ng], sc.spectral_flux_dn_surf + (size_t)c * ng, ng * sizeof(double));
        // SW: the second array carries the per-g weights (calc_cost_function_sw.cpp:271-274)
        if (do_sw) std::memcpy(&sfut[col * ng], sc.spectral_boundary_weights, ng * sizeof(double));
        else std::memcpy(&sfut[col * ng], sc.spectral_flux_up_toa + (size_t)c * ng, ng * sizeof(double));
      }
      // interpolation entries, ckd_model.cpp:960-1086
      for (int l = 0; l < nlay; ++l) {
        const size_t cell = col * nlay + l;
        // full-level temperature, solve_adept.cpp:38-41
        const double t_fl = sc.temperature_fl ? sc.temperature_fl[(size_t)c * nlay + l]
                                              : (T[l] * p[l] + T[l + 1] * p[l + 1]) / (p[l] + p[l + 1]);
        const double log_pressure_fl = std::log(0.5 * (p[l + 1] + p[l]));
        double pindex0 = (log_pressure_fl - log_p_0) / d_log_p;
        pindex0 = std::fmax(0.0, std::fmin(pindex0, np - 1.0001));
        const int ip0 = (int)pindex0;
        const double pw1 = pindex0 - ip0, pw0 = 1.0 - pw1;
        const double t_0 = pw0 * m->temperature[ip0] + pw1 * m->temperature[ip0 + 1];
        double tindex0 = (t_fl - t_0) / d_t;
        tindex0 = std::fmax(0.0, std::fmin(tindex0, nt - 1.0001));
        const int it0 = (int)tindex0;
        const double tw1 = tindex0 - it0, tw0 = 1.0 - tw1;
        const double simple_weight = global_weight * (p[l + 1] - p[l]);
        int e = 0;
        for (size_t pos = 0; pos < o->gases.size(); ++pos) {
          const GasInfo& gi = o->gases[pos];
          const int ugas = order[pos];
          const int width = (gi.conc == 2) ? 8 : 4;
          const bool present = sc.gas_present ? sc.gas_present[ugas] != 0 : true;
          // ckd_model.cpp:995-1004, :1066-1073 and solve_adept.cpp:47-67: a gas without a
          // concentration dependence always contributes with the dry-air weight; the others need
          // their mole fraction from the training file and are skipped when it is absent
          double weight = 0.0, vm = 0.0;
          bool use = true;
          if (gi.conc == 0) {
            weight = simple_weight;
          } else if (present && sc.vmr_fl) {
            vm = sc.vmr_fl[((size_t)c * m->ngas + ugas) * nlay + l];
            weight = (gi.conc == 3) ? simple_weight * (vm - gi.reference_vmr) : simple_weight * vm;
          } else {
            use = false;
          }
          if (use) {
            if (gi.conc == 2) {
              const double log_conc = std::log(vm);
              const double d_log_c = std::log(gi.vmr[1] / gi.vmr[0]);
              double cindex0 = (log_conc - std::log(gi.vmr[0])) / d_log_c;
              cindex0 = std::fmax(0.0, std::fmin(cindex0, gi.nconc - 1.0001));
              const int ic0 = (int)cindex0;
              const double cw1 = cindex0 - ic0, cw0 = 1.0 - cw1;
              int q = 0;
              for (int dc = 0; dc < 2; ++dc)
                for (int dt = 0; dt < 2; ++dt)
                  for (int dp = 0; dp < 2; ++dp) {
                    const size_t node = ((size_t)(ic0 + dc) * nt + (it0 + dt)) * np + (ip0 + dp);
                    ent_idx[cell * nent + e + q] = (int)(gi.ix + node * ng);
                    ent_coef[cell * nent + e + q] = weight * (dc ? cw1 : cw0) * (dt ? tw1 : tw0) * (dp ? pw1 : pw0);
                    ++q;
                  }
            } else {
              int q = 0;
              for (int dt = 0; dt < 2; ++dt)
                for (int dp = 0; dp < 2; ++dp) {
                  const size_t node = (size_t)(it0 + dt) * np + (ip0 + dp);
                  ent_idx[cell * nent + e + q] = (int)(gi.ix + node * ng);
                  ent_coef[cell * nent + e + q] = weight * (dt ? tw1 : tw0) * (dp ? pw1 : pw0);
                  ++q;
                }
            }
          }
          e += width;
        }
        if (have_rayleigh) {
          // moles of air per unit area in the layer (ckd_model.h:246-247) times the molar scattering
          ent_idx[cell * nent + e] = (int)rayleigh_ix;
          ent_coef[cell * nent + e] = (p[l + 1] - p[l]) * (1.0 / (ECCKD_ACCEL_GRAVITY * 0.001 * 28.970));
        }
      }
    }
    col0 += sc.ncol;
  }

  // ---- transpose of the gather for the active nodes: node -> (cell, coef), in cell order ----
  std::vector<int> ref_ptr(o->nnode_active + 1, 0);
  for (size_t cell = 0; cell < o->ncell; ++cell)
    for (int e = 0; e < nent; ++e) {
      const int idx = ent_idx[cell * nent + e];
      if (idx >= 0 && (size_t)idx < o->nx && ent_coef[cell * nent + e] != 0.0) ++ref_ptr[(size_t)idx / ng + 1];
    }
  for (size_t i = 0; i < o->nnode_active; ++i) ref_ptr[i + 1] += ref_ptr[i];
  std::vector<int> ref_cell(ref_ptr.back());
  std::vector<double> ref_coef(ref_ptr.back());
  {
    std::vector<int> fill(ref_ptr.begin(), ref_ptr.end() - 1);
    for (size_t cell = 0; cell < o->ncell; ++cell)
      for (int e = 0; e < nent; ++e) {
        const int idx = ent_idx[cell * nent + e];
        if (idx >= 0 && (size_t)idx < o->nx && ent_coef[cell * nent + e] != 0.0) {
          const int slot = fill[(size_t)idx / ng]++;
          ref_cell[slot] = (int)cell;
          ref_coef[slot] = ent_coef[cell * nent + e];
        }
      }
  }

  // ---- K8b: which block takes which node.  The pressure axis is cut into 8 contiguous pieces of about equal numbers of
  // references; piece x is walked, pressure index ascending, by the blocks x, x + 8, x + 16, ... (one XCD, in dispatch order)
  std::vector<int> node_order, run_order, run_r0_v, run_last_v, node_run0_v;
  {
    constexpr int NX = 8;
    std::vector<long long> refs_of_ip(np, 0);
    for (size_t nd = 0; nd < o->nnode_active; ++nd) refs_of_ip[node_ip[nd]] += ref_ptr[nd + 1] - ref_ptr[nd] + 1;
    long long total = 0;
    for (long long v : refs_of_ip) total += v;
    std::vector<int> piece_of_ip(np, 0);
    long long run = 0;
    for (int ip = 0; ip < np; ++ip) {
      piece_of_ip[ip] = (int)std::min<long long>(NX - 1, run * NX / std::max<long long>(total, 1));
      run += refs_of_ip[ip];
    }
    std::vector<std::vector<int>> list(NX);
    for (int ip = 0; ip < np; ++ip)
      for (size_t nd = 0; nd < o->nnode_active; ++nd)
        if (node_ip[nd] == ip) list[piece_of_ip[ip]].push_back((int)nd);
    // a wave per task, K8B_WAVES waves per block: virtual block j of piece x (= j * NX + x) takes the next K8B_WAVES entries of
    // the piece's list.  Tasks of the first launch: the runs (at most K8B_RUN references of one node) x 64-g chunks; of the
    // second: the nodes x 64-g chunks.
    const int nchunk = (ng + 63) / 64;
    o->grad_nchunk = nchunk;
    const bool plain = [] { const char* e = std::getenv("ECCKD_K8B_PLAIN_ORDER"); return e && e[0] == '1'; }();
    std::vector<int> node_run0(o->nnode_active + 1, 0), run_r0, run_last;
    for (size_t nd = 0; nd < o->nnode_active; ++nd) {
      node_run0[nd] = (int)run_r0.size();
      for (int r = ref_ptr[nd]; r < ref_ptr[nd + 1]; r += K8B_RUN) {
        run_r0.push_back(r);
        run_last.push_back(std::min(r + K8B_RUN, ref_ptr[nd + 1]));
      }
    }
    node_run0[o->nnode_active] = (int)run_r0.size();
    o->grad_nrun = run_r0.size();
    auto lay_out = [&](const std::vector<std::vector<int>>& entries) {
      size_t longest = 0;
      for (const auto& l : entries) longest = std::max(longest, (l.size() + K8B_WAVES - 1) / K8B_WAVES);
      std::vector<int> order(longest * NX * K8B_WAVES, -1);
      for (int x = 0; x < NX; ++x)
        for (size_t q = 0; q < entries[x].size(); ++q) order[((q / K8B_WAVES) * NX + x) * K8B_WAVES + q % K8B_WAVES] = entries[x][q];
      return order;
    };
    std::vector<std::vector<int>> node_entries(NX), run_entries(NX);
    for (int x = 0; x < NX; ++x)
      for (int nd : list[x]) {
        for (int c = 0; c < nchunk; ++c) node_entries[x].push_back(nd * nchunk + c);
        for (int t = node_run0[nd]; t < node_run0[nd + 1]; ++t)
          for (int c = 0; c < nchunk; ++c) run_entries[x].push_back(t * nchunk + c);
      }
    if (plain) {
      for (auto& v : node_entries) v.clear();
      for (auto& v : run_entries) v.clear();
      for (size_t q = 0; q < o->nnode_active * nchunk; ++q) node_entries[(q / K8B_WAVES) % NX].push_back((int)q);
      for (size_t q = 0; q < run_r0.size() * nchunk; ++q) run_entries[(q / K8B_WAVES) % NX].push_back((int)q);
    }
    node_order = lay_out(node_entries);
    run_order = lay_out(run_entries);
    run_r0_v = run_r0; run_last_v = run_last; node_run0_v = node_run0;
    // every virtual block resident at once where that fits (8 blocks of 4 waves per CU), else a grid-stride loop
    auto grid_of = [&](size_t nslot) {
      unsigned gsz = (unsigned)std::min<size_t>((nslot + K8B_WAVES - 1) / K8B_WAVES, (size_t)std::max(ctx->num_cu, 1) * 8);
      return std::max(8u, (gsz + 7) / 8 * 8);
    };
    o->gradient_grid = grid_of(node_order.size());
    o->gather_grid = grid_of(run_order.size());
  }

  // ---- K9: the prior's stencil per node.  B^-1 is the Kronecker product of the three AR(1) inverses (tridiagonal each):
  // up to 27 neighbours, weight = wc * wt * wp, entries below MIN_ERROR_COVARIANCE dropped as the reference zeroes them in its
  // dense inverse (ckd_model.cpp:650, :709-713, :776-780); neighbours in the order (dc, dt, dp) = (-1,-1,-1) ... (1,1,1)
  std::vector<int> st_nb(o->nnode_active * 27, -1);
  std::vector<double> st_w(o->nnode_active * 27, 0.0);
  for (size_t nd = 0; nd < o->nnode_active; ++nd) {
    const int gas = node_gas[nd];
    const int nconc = gas_dims[gas * 4 + 0];
    const size_t node0 = (size_t)gas_dims[gas * 4 + 3];
    const double* tc = tri.data() + tri_off[gas];
    const double* tt = tc + 3 * nconc;
    const double* tp = tt + 3 * nt;
    const int ic = node_ic[nd], it = node_it[nd], ip = node_ip[nd];
    int slot = 0;
    for (int dc = -1; dc <= 1; ++dc) {
      const int jc = ic + dc;
      if (jc < 0 || jc >= nconc) continue;
      const double wc = tc[(dc + 1) * nconc + ic];
      if (wc == 0.0) continue;
      for (int dt = -1; dt <= 1; ++dt) {
        const int jt = it + dt;
        if (jt < 0 || jt >= nt) continue;
        const double wt = tt[(dt + 1) * nt + it];
        if (wt == 0.0) continue;
        for (int dp = -1; dp <= 1; ++dp) {
          const int jp = ip + dp;
          if (jp < 0 || jp >= np) continue;
          const double w = wc * wt * tp[(dp + 1) * np + ip];
          if (std::fabs(w) < 1.0e-6) continue;             // MIN_ERROR_COVARIANCE
          st_nb[nd * 27 + slot] = (int)(node0 + ((size_t)jc * nt + jt) * np + jp);
          st_w[nd * 27 + slot] = w;
          ++slot;
        }
      }
    }
  }

  std::vector<int> band(m->iband_per_g, m->iband_per_g + ng);
  int rc = ECCKD_OK;
#define UP(dst, vec) do { rc = upload(ctx, &o->dst, vec); if (rc) { opt_free(o); return rc; } } while (0)
  UP(d_k, o->h_k0);
  std::vector<int> band_ptr(nband + 1, 0), band_g;
  for (int b = 0; b < nband; ++b) {
    for (int g = 0; g < ng; ++g) if (band[g] == b) band_g.push_back(g);
    band_ptr[b + 1] = (int)band_g.size();
  }
  band_g.resize(ng, 0);                     // (a g point with a negative band number belongs to no band)
  UP(d_ent_idx, ent_idx); UP(d_ent_coef, ent_coef); UP(d_band, band); UP(d_band_ptr, band_ptr); UP(d_band_g, band_g);
  UP(d_planck, planck); UP(d_semis, semis); UP(d_conv, conv); UP(d_lw, lwv);
  UP(d_hr, hr); UP(d_fdn, fdn); UP(d_fup, fup);
  if (any_boundary) { UP(d_sfds, sfds); UP(d_sfut, sfut); }
  if (do_sw) UP(d_mu0, mu0v);
  {
    bool any_rel = false;
    for (int s2 = 0; s2 < nscene; ++s2) any_rel = any_rel || (scenes[s2].relative_flux_dn && scenes[s2].relative_flux_up);
    if (any_rel) {
      std::vector<double> rel(ncol * 2 * nhl * ng, 0.0);
      size_t c0 = 0;
      for (int s2 = 0; s2 < nscene; ++s2) {
        const ecckd_opt_scene& sc = scenes[s2];
        if (sc.relative_flux_dn && sc.relative_flux_up)
          for (size_t c = 0; c < (size_t)sc.ncol; ++c) {
            std::memcpy(&rel[((c0 + c) * 2) * nhl * ng], sc.relative_flux_dn + c * nhl * ng, nhl * ng * sizeof(double));
            std::memcpy(&rel[((c0 + c) * 2 + 1) * nhl * ng], sc.relative_flux_up + c * nhl * ng, nhl * ng * sizeof(double));
          }
        c0 += sc.ncol;
      }
      UP(d_rel, rel);
    }
  }
  UP(d_ref_ptr, ref_ptr); UP(d_ref_cell, ref_cell); UP(d_ref_coef, ref_coef); UP(d_node_order, node_order);
  UP(d_run_order, run_order); UP(d_run_r0, run_r0_v); UP(d_run_last, run_last_v); UP(d_node_run0, node_run0_v);
  o->grad_nslot = (int)node_order.size();
  o->gather_nslot = (int)run_order.size();
  UP(d_node_gas, node_gas); UP(d_st_nb, st_nb); UP(d_st_w, st_w); UP(d_inv_sigma2, inv_sigma2);
#undef UP
  o->grad_blocks = (unsigned)((o->nx + 255) / 256);
  auto dalloc = [&](double** p, size_t n) { return hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(double)); };
  if (dalloc(&o->d_x, o->nx) != hipSuccess || dalloc(&o->d_xprior, o->nx) != hipSuccess ||
      dalloc(&o->d_grad, o->nx + 1) != hipSuccess || dalloc(&o->d_dtau, o->ncell * ng) != hipSuccess ||
      dalloc(&o->d_run_sum, std::max<size_t>(o->grad_nrun, 1) * o->grad_nchunk * 64) != hipSuccess ||
      dalloc(&o->d_jcol, ncol) != hipSuccess || dalloc(&o->d_jb, o->nnode_active * (size_t)((ng + 63) / 64)) != hipSuccess) {
    opt_free(o);
    return ecckd::fail(ECCKD_OUT_OF_MEMORY, "ecckd_opt_create: device allocation failed");
  }
  ECCKD_HIP_CHECK(hipHostMalloc((void**)&o->h_pin, 80 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
  ECCKD_HIP_CHECK(hipHostGetDevicePointer((void**)&o->d_pin, o->h_pin, 0));
  o->h_rb = o->h_pin;
  // x_prior = ln k0 (MIN_X where k0 <= 0), solve_adept.cpp:335-341
  std::vector<double> xp(o->nx);
  for (size_t e = 0; e < o->nx; ++e) xp[e] = o->h_k0[e] > 0.0 ? std::log(o->h_k0[e]) : MIN_X;
  ECCKD_HIP_CHECK(hipMemcpyAsync(o->d_xprior, xp.data(), o->nx * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  *out = o;
  return ECCKD_OK;
}

int ecckd_opt_destroy(ecckd_opt* o) {
  opt_free(o);
  return ECCKD_OK;
}

size_t ecckd_opt_nx(ecckd_opt* o) { return o ? o->nx : 0; }

// Initial state and log-space bounds, solve_adept.cpp:335-353.  h_x_min/h_x_max may be NULL;
// unbounded elements get -/+ infinity.  Returns the gas order of the state in h_gas_order
// (user gas indices of the active gases, in state order) if not NULL.
int ecckd_opt_set_evaluator(ecckd_opt* o, ecckd_evaluator_fn fn, void* user) {
  ECCKD_REQUIRE(o, "ecckd_opt_set_evaluator: NULL handle");
  o->eval_fn = fn;
  o->eval_user = user;
  return ECCKD_OK;
}

int ecckd_opt_set_progress(ecckd_opt* o, ecckd_progress_fn fn, void* user) {
  ECCKD_REQUIRE(o, "ecckd_opt_set_progress: NULL handle");
  o->progress_fn = fn;
  o->progress_user = user;
  if (!o->tev[0]) {
    ECCKD_HIP_CHECK(hipSetDevice(o->ctx->device));
    for (int k = 0; k < 3; ++k) ECCKD_HIP_CHECK(hipEventCreate(&o->tev[k]));
  }
  return ECCKD_OK;
}

int ecckd_opt_timings(ecckd_opt* o, double* minimizer_s, double* a_priori_s, double* radiative_transfer_s) {
  ECCKD_REQUIRE(o, "ecckd_opt_timings: NULL handle");
  if (minimizer_s) *minimizer_s = o->t_minimizer;
  if (a_priori_s) *a_priori_s = o->t_prior;
  if (radiative_transfer_s) *radiative_transfer_s = o->t_rt;
  return ECCKD_OK;
}

int ecckd_opt_set_allreduce(ecckd_opt* o, ecckd_allreduce_fn fn, void* user, int add_prior) {
  ECCKD_REQUIRE(o, "ecckd_opt_set_allreduce: NULL argument");
  o->reduce_fn = fn;
  o->reduce_user = user;
  o->add_prior = fn == nullptr || add_prior != 0;
  return ECCKD_OK;
}

int ecckd_opt_initial_state(ecckd_opt* o, double* h_x, double* h_x_min, double* h_x_max) {
  ECCKD_REQUIRE(o && h_x, "ecckd_opt_initial_state: NULL argument");
  for (size_t e = 0; e < o->nx; ++e) h_x[e] = o->h_k0[e] > 0.0 ? std::log(o->h_k0[e]) : MIN_X;
  if (h_x_min && h_x_max) {
    for (size_t e = 0; e < o->nx; ++e) {
      h_x_min[e] = -INFINITY;
      h_x_max[e] = INFINITY;
      if (!o->h_kmin.empty()) {
        if (o->h_kmin[e] > 0.0) h_x_min[e] = std::log(o->h_kmin[e]);
        if (o->h_kmax[e] > 0.0) h_x_max[e] = std::log(o->h_kmax[e]);
        if (o->h_kmin[e] == 0.0 && o->h_k0[e] > 0.0 && o->h_kmax[e] > 0.0)
          h_x_min[e] = std::min(3.0 * h_x[e] - 2.0 * h_x_max[e], h_x_max[e] - 1.0);
      }
    }
  }
  return ECCKD_OK;
}

// CkdOptimizable::calc_cost_function_gradient (solve_adept.cpp:240-292).
// cost and gradient at the DEVICE state d_x -> d_grad; J on the host (one stream sync).
// Results the host needs at once (the cost of a trial point, the dot products of the L-BFGS update) arrive in pinned,
// host-coherent slots written by the last kernel that produces them; the host marks the slots as pending and watches them
// instead of queueing a copy and synchronising the stream - two waits per iteration, each a PCIe write away from the kernel's
// end (the same scheme as the interval errors of find_g.hip).  kOptPending: a NaN payload no arithmetic produces.
constexpr unsigned long long kOptPending = 0x7ff4dead0b5e55edULL;

static void opt_mark_pending(double* h_slots, int count) {
  volatile unsigned long long* s = reinterpret_cast<volatile unsigned long long*>(h_slots);
  for (int k = 0; k < count; ++k) s[k] = kOptPending;
  std::atomic_thread_fence(std::memory_order_release);
}

static int opt_wait_slots(ecckd_ctx* ctx, const double* h_slots, int count) {
  static const bool no_poll = std::getenv("ECCKD_NO_POLL") != nullptr;   // A/B knob: wait through the runtime
  const volatile unsigned long long* s = reinterpret_cast<const volatile unsigned long long*>(h_slots);
  auto all_there = [&] {
    for (int k = 0; k < count; ++k)
      if (s[k] == kOptPending) return false;
    return true;
  };
  if (no_poll) {
    ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return all_there() ? ECCKD_OK : ecckd::fail(ECCKD_UNEXPECTED_EXCEPTION, "ecckd_opt: results were not delivered by the device");
  }
  for (unsigned spins = 1;; ++spins) {
    if (all_there()) break;
    if ((spins & 0x3fff) == 0) {
      const hipError_t q = hipStreamQuery(ctx->stream);       // drained (or dead) without delivering?
      if (q == hipSuccess) {
        if (all_there()) break;
        return ecckd::fail(ECCKD_UNEXPECTED_EXCEPTION, "ecckd_opt: results were not delivered by the device");
      }
      if (q != hipErrorNotReady)
        return ecckd::fail(ECCKD_UNEXPECTED_EXCEPTION, "ecckd_opt: device failure while waiting for results: %s", hipGetErrorString(q));
    }
    _mm_pause();
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return ECCKD_OK;
}

static int opt_launch_forward(ecckd_opt* o) {
  ecckd_ctx* ctx = o->ctx;
  const int nlay = o->nlay, nhl = nlay + 1, ng = o->ng, nband = o->nband;
  const int ngpad = (ng + 63) / 64 * 64;
  static const int env_threads = [] { const char* e = std::getenv("ECCKD_K8A_THREADS"); const int v = e ? std::atoi(e) : 0; return v >= 64 && v <= 1024 ? v : 0; }();
  const char* env_generic = std::getenv("ECCKD_K8A_GENERIC");       // read at every call: the tests switch between the two kernels
  const bool force_generic = env_generic && env_generic[0] == '1';
  // the cell-parallel kernel: longwave 1 024 threads (16 layer groups at ng = 64, one block per CU), shortwave 512 (8 groups,
  // 79 KB of LDS: two blocks per CU); up to 8 cells per thread, else the general kernel
  {
    const int threads_c = env_threads ? env_threads : (o->do_sw ? 512 : 1024);
    const int lgroups = std::max(1, threads_c / ngpad);
    const int nc = (nlay + lgroups - 1) / lgroups;
    const size_t lds = ((o->do_sw ? (size_t)0 : (size_t)nlay * ng) + 2 * (size_t)nhl * ng + 2 * (size_t)nhl * nband + (size_t)nlay * nband +
                        nlay + 2 * (size_t)nhl + 16) * sizeof(double) + ((size_t)nband + 1 + ng) * sizeof(int);
    if (!force_generic && nc <= 8 && lds <= 160 * 1024 && ngpad * lgroups <= 1024 && nhl * nband <= K8A_MAXR * ngpad * lgroups) {
      const int threads = ngpad * lgroups;
#define ECCKD_K8A_CELLS(NC_, SW_, NB_)                                                                                              \
      do {                                                                                                                            \
        ECCKD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_opt_forward_adjoint_cells<NC_, SW_, NB_>),             \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                              \
        hipLaunchKernelGGL((k_opt_forward_adjoint_cells<NC_, SW_, NB_>), dim3((unsigned)o->ncol), dim3(threads), lds, ctx->stream,  \
                           o->d_mu0, o->eval_ray_ent, o->eval_keep_negative, o->d_rel, nlay, ng, ngpad, nband, o->nent, o->d_k,     \
                           o->d_ent_idx, o->d_ent_coef, o->d_band, o->d_band_ptr, o->d_band_g, o->d_planck, o->d_semis, o->d_conv, \
                           o->d_lw, o->d_hr, o->d_fdn,                                                                               \
                           o->d_fup, o->d_sfds, o->d_sfut, o->cfg.flux_weight, o->cfg.flux_profile_weight,                          \
                           o->cfg.broadband_weight, o->cfg.spectral_boundary_weight, o->cfg.negative_od_penalty, o->d_dtau,         \
                           o->d_jcol, o->d_od_out, o->d_flux_out);                                                                  \
      } while (0)
#define ECCKD_K8A_NB(NC_, SW_)                                                                                                      \
      do {                                                                                                                            \
        switch (nb) {                                                                                                                 \
          case 1: ECCKD_K8A_CELLS(NC_, SW_, 1); break;                                                                                \
          case 2: ECCKD_K8A_CELLS(NC_, SW_, 2); break;                                                                                \
          case 3: ECCKD_K8A_CELLS(NC_, SW_, 3); break;                                                                                \
          case 4: ECCKD_K8A_CELLS(NC_, SW_, 4); break;                                                                                \
          default: ECCKD_K8A_CELLS(NC_, SW_, 0); break;                                                                               \
        }                                                                                                                             \
      } while (0)
      const int nb = (std::min(o->nent, 64) + K8A_GB - 1) / K8A_GB;       // 8-entry batches of a cell's table (first 64 entries)
      if (o->do_sw) { if (nc <= 4) ECCKD_K8A_NB(4, true); else ECCKD_K8A_NB(8, true); }
      else { if (nc <= 4) ECCKD_K8A_NB(4, false); else ECCKD_K8A_NB(8, false); }
#undef ECCKD_K8A_NB
#undef ECCKD_K8A_CELLS
      ECCKD_HIP_CHECK(hipGetLastError());
      return ECCKD_OK;
    }
  }
  const int k8a_threads = env_threads ? env_threads : 1024;
  const int lgroups = std::max(1, k8a_threads / ngpad);
  const int threads = ngpad * lgroups;
  const size_t lds = ((size_t)nlay * ng + 2 * (size_t)nhl * ng + 4 * (size_t)nhl * nband + 16) * sizeof(double) +
                     ecckd_align_up((size_t)nlay * ng, 16);
  ECCKD_REQUIRE(lds <= 160 * 1024, "ecckd_opt_cost_grad: nlay*ng too large for the per-profile LDS tile (%zu B)", lds);
  ECCKD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_opt_forward_adjoint),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL(k_opt_forward_adjoint, dim3((unsigned)o->ncol), dim3(threads), lds, ctx->stream, o->do_sw ? 1 : 0,
                     o->d_mu0, o->eval_ray_ent, o->eval_keep_negative, o->d_rel, nlay, ng, ngpad, nband, o->nent, o->d_k, o->d_ent_idx, o->d_ent_coef, o->d_band, o->d_planck, o->d_semis, o->d_conv,
                     o->d_lw, o->d_hr, o->d_fdn, o->d_fup, o->d_sfds, o->d_sfut, o->cfg.flux_weight,
                     o->cfg.flux_profile_weight, o->cfg.broadband_weight, o->cfg.spectral_boundary_weight,
                     o->cfg.negative_od_penalty, o->d_dtau, o->d_jcol, o->d_od_out, o->d_flux_out);
  ECCKD_HIP_CHECK(hipGetLastError());
  return ECCKD_OK;
}

static int opt_cost_grad_dev(ecckd_opt* o, const double* d_x, double* d_grad, double* J, bool k_ready = false) {
  ecckd_ctx* ctx = o->ctx;
  if (o->eval_fn && o->d_od_out == nullptr) {
    // the caller's cost function and gradient in place of the device's: the minimizer itself is unchanged
    o->eval_x.resize(o->nx);
    o->eval_g.assign(o->nx, 0.0);
    ECCKD_HIP_CHECK(hipMemcpyAsync(o->eval_x.data(), d_x, o->nx * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const int rc = o->eval_fn(o->nx, o->eval_x.data(), J, o->eval_g.data(), o->eval_user);
    if (rc != 0) return ecckd::fail(ECCKD_PROCESSING_ERROR, "ecckd_opt: the cost-function callback failed (%d)", rc);
    ECCKD_HIP_CHECK(hipMemcpyAsync(d_grad, o->eval_g.data(), o->nx * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    o->n_eval++;
    return ECCKD_OK;
  }
  const bool reduce = o->reduce_fn != nullptr && o->d_od_out == nullptr;   // not for the diagnostic forward pass
  const bool prior = o->have_prior && (!reduce || o->add_prior);
  if (!k_ready)     // (the minimizer's step kernel leaves exp(x) of its trial point in d_k)
    hipLaunchKernelGGL(k_opt_exp, dim3(o->grad_blocks), dim3(256), 0, ctx->stream, o->nx, d_x, o->d_k);
  const int ng = o->ng;
  const bool timed = o->tev[0] != nullptr;
  if (timed) ECCKD_HIP_CHECK(hipEventRecord(o->tev[0], ctx->stream));
  ECCKD_CHECK(opt_launch_forward(o));
  if (timed) ECCKD_HIP_CHECK(hipEventRecord(o->tev[1], ctx->stream));
  hipLaunchKernelGGL(k_opt_gradient_gather, dim3(o->gather_grid), dim3(64 * K8B_WAVES), 0, ctx->stream, o->gather_nslot, ng, o->grad_nchunk,
                     o->d_run_order, o->d_run_r0, o->d_run_last, o->d_ref_cell, o->d_ref_coef, o->d_dtau, o->d_run_sum);
  hipLaunchKernelGGL(k_opt_gradient_finish, dim3(o->gradient_grid), dim3(64 * K8B_WAVES), 0, ctx->stream, o->grad_nslot, ng, o->grad_nchunk,
                     d_x, o->d_xprior, o->d_k, o->d_node_run0, o->d_run_sum, o->d_node_gas, o->d_st_nb, o->d_st_w, o->d_inv_sigma2,
                     prior ? 1 : 0, d_grad, o->d_jb, o->d_node_order);
  ECCKD_HIP_CHECK(hipGetLastError());
  if (timed) ECCKD_HIP_CHECK(hipEventRecord(o->tev[2], ctx->stream));
  // the cost: profiles, then the prior's node terms, summed on the device in a fixed order and delivered to the host's slot
  // (and, for the profile-sharded all-reduce, into the slot behind the gradient: ONE collective of nx + 1 doubles)
  double* h_cost = o->h_pin + 70;
  opt_mark_pending(h_cost, 1);
  hipLaunchKernelGGL(k_opt_sum_cost, dim3(1), dim3(256), 0, ctx->stream, o->d_jcol, o->ncol, o->d_jb, o->nnode_active * (size_t)o->grad_nchunk, prior ? 1 : 0,
                     o->d_pin + 70, reduce ? d_grad + o->nx : nullptr);
  ECCKD_HIP_CHECK(hipGetLastError());
  ECCKD_CHECK(opt_wait_slots(ctx, h_cost, 1));
  if (timed) {
    // "radiative transfer": look-up, penalty, two sweeps, cost and their adjoint; "a-priori": the gradient kernel, which
    // gathers the adjoint back onto the coefficients and adds the prior term (the reference times the prior alone there)
    float ms = 0.f;
    ECCKD_HIP_CHECK(hipEventSynchronize(o->tev[2]));
    ECCKD_HIP_CHECK(hipEventElapsedTime(&ms, o->tev[0], o->tev[1]));
    o->t_rt += 1e-3 * ms;
    ECCKD_HIP_CHECK(hipEventElapsedTime(&ms, o->tev[1], o->tev[2]));
    o->t_prior += 1e-3 * ms;
  }
  *J = *h_cost;
  if (reduce) {
    double* slot = o->h_pin + 71;
    ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const int rc = o->reduce_fn(d_grad, o->nx + 1, (void*)ctx->stream, o->reduce_user);
    if (rc != 0) return ecckd::fail(ECCKD_PROCESSING_ERROR, "ecckd_opt: the all-reduce callback failed (%d)", rc);
    ECCKD_HIP_CHECK(hipMemcpyAsync(slot, d_grad + o->nx, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *J = *slot;
  }
  o->n_eval++;
  return ECCKD_OK;
}

int ecckd_opt_cost_grad(ecckd_opt* o, const double* h_x, double* J, double* h_grad) {
  ECCKD_REQUIRE(o && h_x && J, "ecckd_opt_cost_grad: NULL argument");
  ecckd_ctx* ctx = o->ctx;
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  ECCKD_HIP_CHECK(hipMemcpyAsync(o->d_x, h_x, o->nx * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ECCKD_CHECK(opt_cost_grad_dev(o, o->d_x, o->d_grad, J));
  if (h_grad) {
    ECCKD_HIP_CHECK(hipMemcpyAsync(h_grad, o->d_grad, o->nx * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  return ECCKD_OK;
}

// Forward products for checking/diagnostics: total optical depth [ncol][nlay][ng] (after the
// negative clamp) and the CKD fluxes [ncol][2][nlay+1][ng] (LblFluxes::calc_ckd_fluxes,
// lbl_fluxes.cpp:443-471) at state h_x.
int ecckd_opt_forward(ecckd_opt* o, const double* h_x, double* h_od, double* h_flux) {
  return ecckd_opt_forward_ex(o, h_x, 0, h_od, h_flux);
}

// unclamped != 0: optical depths and fluxes WITHOUT the clamp of negative total optical depths at zero, as the reference
// evaluates the "relative_to" scene (optimize_lut.cpp:229-234: calc_total_optical_depth -> value() -> calc_ckd_fluxes;
// the clamp lives in calc_cost_function_and_gradient only, solve_adept.cpp:107-116).
int ecckd_opt_forward_ex(ecckd_opt* o, const double* h_x, int unclamped, double* h_od, double* h_flux) {
  ECCKD_REQUIRE(o && h_x, "ecckd_opt_forward: NULL argument");
  ecckd_ctx* ctx = o->ctx;
  const size_t nod = o->ncell * o->ng, nfl = o->ncol * 2 * (o->nlay + 1) * o->ng;
  if (!o->d_od_out) ECCKD_HIP_CHECK(hipMalloc((void**)&o->d_od_out, nod * sizeof(double)));
  if (!o->d_flux_out) ECCKD_HIP_CHECK(hipMalloc((void**)&o->d_flux_out, nfl * sizeof(double)));
  double J;
  const int saved_mode = o->eval_keep_negative;
  if (unclamped) o->eval_keep_negative = 2;
  int rc = ecckd_opt_cost_grad(o, h_x, &J, nullptr);
  o->eval_keep_negative = saved_mode;
  if (rc == ECCKD_OK) {
    if (h_od) rc = ecckd_d2h(ctx, h_od, o->d_od_out, nod * sizeof(double));
    if (rc == ECCKD_OK && h_flux) rc = ecckd_d2h(ctx, h_flux, o->d_flux_out, nfl * sizeof(double));
  }
  (void)hipFree(o->d_od_out); (void)hipFree(o->d_flux_out);
  o->d_od_out = nullptr; o->d_flux_out = nullptr;
  return rc;
}

// The optimised molar absorption coefficients of one (user-indexed) gas at state h_x:
// k = exp(x), 0 where x is pinned at MIN_X (solve_adept.cpp:242-249).
int ecckd_opt_coefficients(ecckd_opt* o, const double* h_x, int gas, double* h_molar_abs) {
  ECCKD_REQUIRE(o && h_molar_abs && gas >= 0 && gas < o->ngas, "ecckd_opt_coefficients: bad argument");
  const GasInfo& gi = o->gases[o->user_to_k[gas]];
  for (size_t e = 0; e < gi.nnode * o->ng; ++e) {
    const size_t ge = gi.ix + e;
    if (gi.active && h_x) h_molar_abs[e] = h_x[ge] > MIN_X ? std::exp(h_x[ge]) : 0.0;
    else h_molar_abs[e] = o->h_k0[ge];
  }
  return ECCKD_OK;
}

// ---------------------------------------------------------------------------------------
// solve_adept (solve_adept.cpp:310-417).  adept::Minimizer is a third-party L-BFGS that is not
// available here (trajectory parity unpinned); this is a standard L-BFGS (m = 6, compact
// representation, see the k_lb_* kernels) with Armijo backtracking + curvature-guarded updates, the step
// capped at max_step_size = 2 in the 2-norm (:331) and simple projection onto [x_min, x_max] when
// bounded (:344-353).  Status values follow adept::MinimizerStatus (0 success, 2 max
// iterations, 3 failed to converge, 6 invalid cost function, 7 invalid gradient).
int ecckd_opt_minimize(ecckd_opt* o, int max_iterations, double convergence_criterion, int is_bounded,
                       double* h_x, int* status, int* n_iterations, double* J_final, double* gnorm_final) {
  ECCKD_REQUIRE(o && h_x && status, "ecckd_opt_minimize: NULL argument");
  ecckd_ctx* ctx = o->ctx;
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  const size_t n = o->nx;
  constexpr int M = LB_M;
  // state, bounds and history live on the device; the host only sees scalars
  std::vector<double> x0(n), xmin, xmax;
  const bool bounded = is_bounded && !o->h_kmin.empty();
  if (bounded) {
    xmin.resize(n); xmax.resize(n);
    ECCKD_CHECK(ecckd_opt_initial_state(o, x0.data(), xmin.data(), xmax.data()));
  } else {
    ECCKD_CHECK(ecckd_opt_initial_state(o, x0.data(), nullptr, nullptr));
  }
  auto dalloc = [&](double** p, size_t cnt) -> int {
    if (*p) return ECCKD_OK;
    ECCKD_HIP_CHECK(hipMalloc((void**)p, cnt * sizeof(double)));
    return ECCKD_OK;
  };
  constexpr int NPART = LB_NPART;   // update: |q|^2, S.q, Y.q (1 + 2M) | pair dots (2 + 3M)
  ECCKD_CHECK(dalloc(&o->d_xn, n)); ECCKD_CHECK(dalloc(&o->d_gn, n + 1)); ECCKD_CHECK(dalloc(&o->d_dir, n));
  ECCKD_CHECK(dalloc(&o->d_q, n)); ECCKD_CHECK(dalloc(&o->d_S, (size_t)M * n)); ECCKD_CHECK(dalloc(&o->d_Y, (size_t)M * n));
  ECCKD_HIP_CHECK(hipMemsetAsync(o->d_S, 0, (size_t)M * n * sizeof(double), ctx->stream));     // unused pairs are read (with zero weight)
  ECCKD_HIP_CHECK(hipMemsetAsync(o->d_Y, 0, (size_t)M * n * sizeof(double), ctx->stream));
  ECCKD_CHECK(dalloc(&o->d_part, (size_t)(NPART + 2) * VEC_BLOCKS));
  if (bounded) {
    ECCKD_CHECK(dalloc(&o->d_xmin, n)); ECCKD_CHECK(dalloc(&o->d_xmax, n));
    ECCKD_HIP_CHECK(hipMemcpyAsync(o->d_xmin, xmin.data(), n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ECCKD_HIP_CHECK(hipMemcpyAsync(o->d_xmax, xmax.data(), n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  }
  ECCKD_HIP_CHECK(hipMemcpyAsync(o->d_x, x0.data(), n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  const double* bmin = bounded ? o->d_xmin : nullptr;
  const double* bmax = bounded ? o->d_xmax : nullptr;
  double* x = o->d_x; double* g = o->d_grad; double* xn = o->d_xn; double* gn = o->d_gn;
  double* part_a = o->d_part;                                  // the update's NPART partial arrays
  double* part_dg = o->d_part + (size_t)NPART * VEC_BLOCKS;    // direction: d.g, then d.d
  double* part_dd = part_dg + VEC_BLOCKS;
  // scalars the kernels deliver into the pinned slots: [0 .. 1+2M) direction dots, [1+2M .. 3+5M) pair dots, [64..67) step, d.d, d.g
  double* sc = o->d_pin;
  double* sc_step = o->d_pin + 64;
  double* h_rb = o->h_rb;
  double* h_step = o->h_rb + 64;
  // one block of 16 waves per CU is what the update kernel's registers allow: no more blocks than CUs (a second, nearly empty
  // round would cost a whole pass again)
  const int nblk = (int)std::min<size_t>(std::min<size_t>(VEC_BLOCKS, (size_t)std::max(ctx->num_cu, 1)), std::max<size_t>(1, (n + VEC_THREADS - 1) / VEC_THREADS));
  const dim3 vb(nblk), vt(VEC_THREADS);
  const double max_step = 2.0;  // minimizer.set_max_step_size(2.0), solve_adept.cpp:331

  // host side of the compact representation: S^T Y and Y^T Y of the stored pairs, by ring slot
  double SY[M][M] = {}, YY[M][M] = {};
  int ord[M];        // ring slots, oldest first
  int hist = 0;
  double gamma = 1.0;
  bool pending = false;   // a pair has been written to `pend_slot`; its dots arrive with the next direction's
  int pend_slot = 0;

  const auto wall0 = std::chrono::steady_clock::now();
  const double dev0 = o->t_rt + o->t_prior;
  double J = 0.0;
  ECCKD_CHECK(opt_cost_grad_dev(o, x, g, &J));
  int st = 2, it = 0;
  double gnorm = 0.0;
  if (!(J == J)) { *status = 6; if (J_final) *J_final = J; return ECCKD_OK; }
  {
    // projected gradient and its norm at the starting point (no pair yet)
    LbSlots none;
    none.n = 0;
    for (int a = 0; a < M; ++a) none.slot[a] = 0;
    opt_mark_pending(h_rb, NPART);
    hipLaunchKernelGGL(k_lb_update, vb, vt, 0, ctx->stream, n, none, -1, x, x, g, g, bmin, bmax, o->d_q, o->d_S, o->d_Y, part_a);
    hipLaunchKernelGGL(k_lb_finish, dim3(NPART), dim3(FIN_THREADS), 0, ctx->stream, part_a, nblk, sc);
    ECCKD_HIP_CHECK(hipGetLastError());
  }
  for (it = 0; it <= max_iterations; ++it) {
    // |q|^2, S^T q, Y^T q at the current point (the pending pair included, as the newest) and the pending pair's own dots:
    // launched at the end of the previous iteration (k_lb_update), awaited here
    ECCKD_CHECK(opt_wait_slots(ctx, h_rb, NPART));
    int npairs = hist;           // pairs that enter this direction
    if (pending) {
      const double* rp = h_rb + 1 + 2 * M;   // s.y, y.y, s.Y_a, S_a.y, y.Y_a for a < hist
      const double sy = rp[0], yy = rp[1];
      if (sy > 1.0e-12) {
        for (int a = 0; a < hist; ++a) {
          SY[pend_slot][ord[a]] = rp[2 + a];
          SY[ord[a]][pend_slot] = rp[2 + M + a];
          YY[pend_slot][ord[a]] = YY[ord[a]][pend_slot] = rp[2 + 2 * M + a];
        }
        SY[pend_slot][pend_slot] = sy;
        YY[pend_slot][pend_slot] = yy;
        ord[hist++] = pend_slot;
        gamma = sy / yy;
        npairs = hist;
      }
      pending = false;
    }
    gnorm = std::sqrt(h_rb[0]);
    if (o->progress_fn) o->progress_fn(it, J, gnorm, o->progress_user);   // report_progress, solve_adept.cpp:295-299
    if (!(gnorm == gnorm)) { st = 7; break; }
    if (gnorm <= convergence_criterion) { st = 0; break; }
    if (it == max_iterations) { st = 2; break; }

    bool restarted = false;
    double Jn = J, step = 0.0, dg = 0.0;
    bool ok = false;
    for (;;) {
      // coefficients of the direction: u = R^-1 p, top = R^-T ((D + gamma YY) u - gamma qy)
      LbCoef cf;
      cf.gamma = npairs > 0 ? gamma : 1.0;
      for (int a = 0; a < M; ++a) { cf.cs[a] = 0.0; cf.cy[a] = 0.0; }
      LbSlots sd;
      sd.n = npairs;
      for (int a = 0; a < M; ++a) sd.slot[a] = a < npairs ? ord[a] : 0;
      if (npairs > 0) {
        double pv[M], qv[M], u[M], w[M], top[M];
        for (int a = 0; a < npairs; ++a) {
          // position of pair a in the dot arrays of this iteration: the stored pairs first, the pending one last
          pv[a] = h_rb[1 + a];
          qv[a] = h_rb[1 + M + a];
        }
        for (int a = npairs - 1; a >= 0; --a) {
          double v = pv[a];
          for (int c = a + 1; c < npairs; ++c) v -= SY[ord[a]][ord[c]] * u[c];
          u[a] = v / SY[ord[a]][ord[a]];
        }
        for (int a = 0; a < npairs; ++a) {
          double v = SY[ord[a]][ord[a]] * u[a] - cf.gamma * qv[a];
          for (int c = 0; c < npairs; ++c) v += cf.gamma * YY[ord[a]][ord[c]] * u[c];
          w[a] = v;
        }
        for (int a = 0; a < npairs; ++a) {
          double v = w[a];
          for (int c = 0; c < a; ++c) v -= SY[ord[c]][ord[a]] * top[c];
          top[a] = v / SY[ord[a]][ord[a]];
        }
        for (int a = 0; a < npairs; ++a) { cf.cs[a] = top[a]; cf.cy[a] = cf.gamma * u[a]; }
      }
      hipLaunchKernelGGL(k_lb_direction, vb, vt, 0, ctx->stream, n, sd, cf, o->d_q, o->d_S, o->d_Y, g, o->d_dir, part_dg);
      // the trial point (and its coefficients) with the step chosen on the device, then the cost there; one wait for both
      opt_mark_pending(h_step, 3);
      hipLaunchKernelGGL(k_lb_step, vb, vt, 0, ctx->stream, n, -1.0, npairs == 0 ? 1 : 0, max_step, 1.0, part_dg,
                         part_dd, x, o->d_dir, bmin, bmax, xn, o->d_k, sc_step);
      ECCKD_HIP_CHECK(hipGetLastError());
      ECCKD_CHECK(opt_cost_grad_dev(o, xn, gn, &Jn, true));
      ECCKD_CHECK(opt_wait_slots(ctx, h_step, 3));
      step = h_step[0];
      dg = h_step[2];
      if (!(dg < 0.0)) {
        // not a descent direction: drop the history, steepest descent on the projected gradient
        // steepest descent on a non-zero projected gradient that is not a descent direction either: the gradient is not
        // a number (MINIMIZER_STATUS_INVALID_GRADIENT) or no direction can be found (..._DIRECTION_FAILURE) - not "converged"
        if (restarted || npairs == 0) { ok = false; st = (dg != dg) ? 7 : 4; break; }
        restarted = true;
        hist = 0;
        npairs = 0;
        continue;
      }
      ok = (Jn == Jn && Jn <= J + 1.0e-4 * step * dg);
      // (Tried in round 4 and dropped: a line search that remembers - the next iteration starting from the fraction of the capped
      // step that was accepted last, doubled after a first-trial acceptance.  Over 300 iterations at nx = 3.05e5: 5 800 against
      // 6 000 iterations/s longwave, 3 830 against 3 730 shortwave, final costs 66.62 against 66.39 and 53.62 against 53.56 -
      // fewer evaluations per iteration, smaller steps, nothing gained.)
      for (int ls = 1; !ok && ls < 30; ++ls) {
        step *= 0.5;
        hipLaunchKernelGGL(k_lb_step, vb, vt, 0, ctx->stream, n, step, 0, max_step, 1.0, part_dg, part_dd, x, o->d_dir, bmin,
                           bmax, xn, o->d_k, sc_step);
        ECCKD_CHECK(opt_cost_grad_dev(o, xn, gn, &Jn, true));
        ok = (Jn == Jn && Jn <= J + 1.0e-4 * step * dg);
      }
      if (!ok && npairs > 0 && !restarted) {
        // the quasi-Newton direction did not give a sufficient decrease (typically many variables at their bounds):
        // forget the history and try the projected steepest descent before giving up
        restarted = true;
        hist = 0;
        npairs = 0;
        continue;
      }
      if (!ok) st = 3;
      break;
    }
    if (!ok) break;
    // curvature pair into the next ring slot (replacing the oldest when the ring is full); its dots with the pairs that
    // stay, and the projected gradient at the new point with every pair's dots, in ONE pass; read at the top of the loop
    if (hist == M) {
      pend_slot = ord[0];
      for (int a = 1; a < M; ++a) ord[a - 1] = ord[a];
      --hist;
    } else {
      bool used[M] = {};
      for (int a = 0; a < hist; ++a) used[ord[a]] = true;
      pend_slot = 0;
      while (used[pend_slot]) ++pend_slot;
    }
    LbSlots sp;
    sp.n = hist;
    for (int a = 0; a < M; ++a) sp.slot[a] = a < hist ? ord[a] : 0;
    opt_mark_pending(h_rb, NPART);
    hipLaunchKernelGGL(k_lb_update, vb, vt, 0, ctx->stream, n, sp, pend_slot, x, xn, g, gn, bmin, bmax, o->d_q, o->d_S, o->d_Y, part_a);
    hipLaunchKernelGGL(k_lb_finish, dim3(NPART), dim3(FIN_THREADS), 0, ctx->stream, part_a, nblk, sc);
    ECCKD_HIP_CHECK(hipGetLastError());
    pending = true;
    std::swap(x, xn);
    std::swap(g, gn);
    J = Jn;
  }
  // (a break may leave the slots of an update in flight: they belong to this handle's pinned block, nobody else reads them)
  ECCKD_HIP_CHECK(hipMemcpyAsync(h_x, x, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  o->t_minimizer += std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count() - (o->t_rt + o->t_prior - dev0);
  *status = st;
  if (n_iterations) *n_iterations = it;
  if (J_final) *J_final = J;
  if (gnorm_final) *gnorm_final = gnorm;
  return ECCKD_OK;
}

// ---------------------------------------------------------------------------------------
// run_ckd (run_ckd.cpp:27-373): evaluate a CKD model on a set of profiles.  The gas optical depths
// are summed and clamped at zero (:318), Rayleigh scattering is added after the clamp (:361), the
// longwave fluxes come from radiative_transfer_lw with unit emissivity and the surface Planck
// function at temperature_hl(end) (:343-347), the shortwave direct beam from
// radiative_transfer_direct_sw at the scene's mu0 with tsi / sum(ssi) scaling (:358-363).
// The coefficients are used as they are (no exp(log k) round trip).
static int run_ckd_impl(ecckd_ctx* ctx, const ecckd_opt_model* m, const ecckd_opt_scene* scene, double* h_od,
                        double* h_rayleigh_od, double* h_planck_hl, double* h_flux, int keep_negative) {
  ECCKD_REQUIRE(ctx && m && scene && m->gases && m->iband_per_g, "ecckd_run_ckd: NULL argument");
  ECCKD_REQUIRE(scene->ncol > 0 && scene->nlay > 0 && scene->pressure_hl && scene->temperature_hl,
                "ecckd_run_ckd: scene needs pressure_hl and temperature_hl");
  const bool do_sw = m->solar_irradiance != nullptr;
  std::vector<ecckd_opt_gas> gases(m->gases, m->gases + m->ngas);
  // gases outside the scene's list are skipped whatever their concentration dependence (run_ckd.cpp:270-276;
  // the optimiser keeps concentration-independent gases, solve_adept.cpp:55-67): give them zero coefficients
  size_t zsize = 0;
  for (int i = 0; i < m->ngas; ++i)
    zsize = std::max(zsize, (size_t)(m->gases[i].conc_dependence == 2 ? m->gases[i].nconc : 1) * m->nt * m->np * m->ng);
  std::vector<double> zero_k(zsize, 0.0);
  for (int i = 0; i < m->ngas; ++i) {
    ecckd_opt_gas& g = gases[i];
    g.is_active = 1;
    g.min_molar_abs = nullptr;
    g.max_molar_abs = nullptr;
    if (scene->gas_present && !scene->gas_present[i]) g.molar_abs = zero_k.data();
  }
  ecckd_opt_model mm = *m;
  mm.gases = gases.data();
  int nband = 0;
  for (int g = 0; g < m->ng; ++g) nband = std::max(nband, m->iband_per_g[g] + 1);
  const size_t ncol = (size_t)scene->ncol, nhl = (size_t)scene->nlay + 1, nlay = (size_t)scene->nlay, ng = (size_t)m->ng;
  std::vector<double> zeros(ncol * nhl * (size_t)nband, 0.0), albedo0((size_t)nband, 0.0);
  ecckd_opt_scene sc = *scene;
  sc.nband = nband;
  sc.flux_dn = zeros.data();
  sc.flux_up = zeros.data();
  sc.surf_emissivity = nullptr;          // unit emissivity (:339-340)
  sc.spectral_flux_dn_surf = nullptr;
  sc.spectral_flux_up_toa = nullptr;
  sc.spectral_boundary_weights = nullptr;
  sc.relative_flux_dn = nullptr;
  sc.relative_flux_up = nullptr;
  if (do_sw) sc.albedo = albedo0.data();  // direct beam only
  ecckd_opt_config cfg;
  std::memset(&cfg, 0, sizeof(cfg));
  cfg.pressure_weight_power = 0.5;
  cfg.prior_error = 1.0;
  cfg.pressure_corr = cfg.temperature_corr = cfg.conc_corr = 0.5;
  ecckd_opt* o = nullptr;
  ECCKD_CHECK(ecckd_opt_create(ctx, &mm, 1, &sc, &cfg, &o));
  int rc = ECCKD_OK;
  do {
    hipError_t e = hipMemcpyAsync(o->d_k, o->h_k0.data(), o->nk * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMalloc((void**)&o->d_od_out, o->ncell * ng * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&o->d_flux_out, ncol * 2 * nhl * ng * sizeof(double));
    if (e != hipSuccess) { rc = ecckd::fail(ECCKD_UNEXPECTED_EXCEPTION, "ecckd_run_ckd: %s", hipGetErrorString(e)); break; }
    o->eval_ray_ent = o->ray_ent;
    o->eval_keep_negative = keep_negative;
    rc = opt_launch_forward(o);
    if (rc != ECCKD_OK) break;
    if (h_od) rc = ecckd_d2h(ctx, h_od, o->d_od_out, o->ncell * ng * sizeof(double));
    if (rc == ECCKD_OK && h_flux) rc = ecckd_d2h(ctx, h_flux, o->d_flux_out, ncol * 2 * nhl * ng * sizeof(double));
    if (rc == ECCKD_OK && h_planck_hl) rc = ecckd_d2h(ctx, h_planck_hl, o->d_planck, ncol * nhl * ng * sizeof(double));
  } while (0);
  if (rc == ECCKD_OK && h_rayleigh_od) {
    // CkdModel::calc_rayleigh_optical_depth (ckd_model.h:242-252)
    for (size_t c = 0; c < ncol; ++c)
      for (size_t l = 0; l < nlay; ++l) {
        const double* p = scene->pressure_hl + c * nhl;
        const double moles = (p[l + 1] - p[l]) * (1.0 / (ECCKD_ACCEL_GRAVITY * 0.001 * 28.970));
        for (size_t g = 0; g < ng; ++g)
          h_rayleigh_od[(c * nlay + l) * ng + g] = m->rayleigh_molar_scattering ? moles * m->rayleigh_molar_scattering[g] : 0.0;
      }
  }
  (void)hipFree(o->d_od_out); (void)hipFree(o->d_flux_out);
  o->d_od_out = nullptr; o->d_flux_out = nullptr;
  ecckd_opt_destroy(o);
  return rc;
}


int ecckd_run_ckd(ecckd_ctx* ctx, const ecckd_opt_model* m, const ecckd_opt_scene* scene, double* h_od,
                  double* h_rayleigh_od, double* h_planck_hl, double* h_flux) {
  return run_ckd_impl(ctx, m, scene, h_od, h_rayleigh_od, h_planck_hl, h_flux, 0);
}

// ---------------------------------------------------------------------------------------
// scale_lut (scale_lut.cpp:117-189) + CkdModel::scale_optical_depth (ckd_model.cpp:1151-1176)
int ecckd_scale_lut(ecckd_ctx* ctx, const ecckd_opt_model* m, int nz, const double* h_pressure_hl,
                    const double* h_temperature_hl, const double* h_vmr_fl, const int* gas_present, double mu0,
                    const double* h_flux_sums, double* h_scaling, double* const* h_molar_abs_out) {
  ECCKD_REQUIRE(ctx && m && nz > 0 && h_pressure_hl && h_temperature_hl && h_flux_sums && h_molar_abs_out,
                "ecckd_scale_lut: NULL argument");
  const int ng = m->ng, np = m->np, nt = m->nt;
  // od_best (:117-133)
  std::vector<double> od_best((size_t)nz * ng), od_total((size_t)nz * ng), scaling((size_t)nz * ng);
  for (int g = 0; g < ng; ++g) {
    double flux_top = h_flux_sums[g];
    for (int iz = 0; iz < nz; ++iz) {
      const double flux_base = h_flux_sums[(size_t)(iz + 1) * ng + g];
      od_best[(size_t)iz * ng + g] = (flux_base <= 0.0) ? -1.0 : -mu0 * std::log(flux_base / flux_top);
      flux_top = flux_base;
    }
  }
  // od_total (:137-182): the CKD model on the reference profile, plain-mean temperature (:108), no clamp
  std::vector<double> t_fl(nz);
  for (int l = 0; l < nz; ++l) t_fl[l] = 0.5 * (h_temperature_hl[l] + h_temperature_hl[l + 1]);
  ecckd_opt_scene sc;
  std::memset(&sc, 0, sizeof(sc));
  sc.ncol = 1;
  sc.nlay = nz;
  sc.pressure_hl = h_pressure_hl;
  sc.temperature_hl = h_temperature_hl;
  sc.temperature_fl = t_fl.data();
  sc.vmr_fl = h_vmr_fl;
  sc.gas_present = gas_present;
  const double mu0v[1] = {mu0};
  sc.mu0 = mu0v;
  sc.tsi = 1.0;
  ECCKD_CHECK(run_ckd_impl(ctx, m, &sc, od_total.data(), nullptr, nullptr, nullptr, 1));
  for (size_t i = 0; i < scaling.size(); ++i) scaling[i] = (od_best[i] <= 0.0) ? 1.0 : od_best[i] / od_total[i];  // :186-187
  if (h_scaling) std::memcpy(h_scaling, scaling.data(), scaling.size() * sizeof(double));
  // interp(log(pressure_fl), scaling, log_pressure_) (ckd_model.cpp:1152): per g point, linear in log p with
  // linear extrapolation outside the profile
  std::vector<double> x(nz);
  for (int l = 0; l < nz; ++l) x[l] = std::log(0.5 * (h_pressure_hl[l] + h_pressure_hl[l + 1]));
  std::vector<double> local((size_t)np * ng);
  for (int ip = 0; ip < np; ++ip) {
    const double xi = m->log_pressure[ip];
    int j = 0;
    if (nz >= 2) {
      while (j < nz - 2 && xi > x[j + 1]) ++j;
    }
    for (int g = 0; g < ng; ++g) {
      if (nz == 1) { local[(size_t)ip * ng + g] = scaling[g]; continue; }
      const double w = (xi - x[j]) / (x[j + 1] - x[j]);
      local[(size_t)ip * ng + g] = (1.0 - w) * scaling[(size_t)j * ng + g] + w * scaling[(size_t)(j + 1) * ng + g];
    }
  }
  // every gas scaled equally (:1157-1175), then held inside [min, max]
  for (int i = 0; i < m->ngas; ++i) {
    const ecckd_opt_gas& ug = m->gases[i];
    const int nconc = ug.conc_dependence == 2 ? ug.nconc : 1;
    double* out = h_molar_abs_out[i];
    ECCKD_REQUIRE(out && ug.molar_abs, "ecckd_scale_lut: gas %d has no coefficient buffer", i);
    for (int ic = 0; ic < nconc; ++ic)
      for (int it = 0; it < nt; ++it)
        for (int ip = 0; ip < np; ++ip)
          for (int g = 0; g < ng; ++g) {
            const size_t e = (((size_t)ic * nt + it) * np + ip) * ng + g;
            double v = ug.molar_abs[e] * local[(size_t)ip * ng + g];
            if (ug.min_molar_abs && ug.max_molar_abs) v = std::fmax(ug.min_molar_abs[e], std::fmin(v, ug.max_molar_abs[e]));
            out[e] = v;
          }
  }
  return ECCKD_OK;
}

}  // extern "C"
