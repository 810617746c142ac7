// lbl_fluxes.hip - line-by-line band fluxes of one column (a stand-in for the external
// CKDMIP tool that test/run_lw_lbl_evaluation.sh uses to make the training fluxes, restricted to the
// no-scattering radiative transfer the reference itself contains):
//   longwave:  planck_function (planck_function.cpp:22-54) + radiative_transfer_lw (radiative_transfer_lw.cpp:27-60,
//              unit surface emissivity, surface Planck function at temperature_hl(end)), fluxes summed per band;
//   shortwave: radiative_transfer_direct_sw / _norayleigh_sw (radiative_transfer_sw.cpp:26-77).
// One thread per wavenumber; per half level the fluxes of a block are reduced wave -> block in a fixed order and
// written as chunk partials, which the host adds in chunk order (bitwise reproducible).
#include "common.hpp"
#include "fastmath.hpp"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr int LBL_THREADS = 256;
__device__ constexpr double kPlanckH = 6.62606896e-34;
__device__ constexpr double kLightC = 2.99792458e8;
__device__ constexpr double kPi = 3.14159265358979323846;

struct BandChunk { long long i1, i2; int band; int pad; };

// The zenith angles of the longwave fluxes.  n = 1 with sec = 1.66, weight = 1: the classic two-stream form the reference
// itself uses (radiative_transfer_lw.cpp:27-60, LW_DIFFUSIVITY).  n = nangle > 0: Gauss-Legendre quadrature in mu = cos(zenith
// angle) over one hemisphere, flux = sum_k 2 w_k mu_k L(mu_k), each L(mu_k) the same no-scattering recurrence with the slant
// path tau / mu_k in place of 1.66 tau (CKDMIP's `nangle`, Hogan & Matricardi 2020, GMD 13, 6501-6521, section 3.2: "N angles
// per hemisphere").  The CKDMIP tool is not among the reference's sources: its node set is unpinned, see DESIGN.md.
constexpr int LBL_MAX_ANGLES = 16;
struct Angles { int n; double sec[LBL_MAX_ANGLES]; double weight[LBL_MAX_ANGLES]; };

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}

// LDS: acc[4][2*nhl]
template <typename OdT>
__global__ void __launch_bounds__(LBL_THREADS)
k_lbl_fluxes_lw(int nang, const double* __restrict__ ang /*[nang] secants, [nang] weights*/, int nlay, size_t od_stride, const BandChunk* __restrict__ chunks, const double* __restrict__ hk,
                const double* __restrict__ wn, const double* __restrict__ dwn, const OdT* __restrict__ od,
                double* __restrict__ partial, double* __restrict__ surf_dn /* [nwav] or NULL */,
                double* __restrict__ toa_up /* [nwav] or NULL */) {
  extern __shared__ double s_acc[];
  const int nhl = nlay + 1;
  const BandChunk c = chunks[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int t = tid; t < 4 * 2 * nhl; t += LBL_THREADS) s_acc[t] = 0.0;
  __syncthreads();
  double* acc_dn = s_acc + wave * 2 * nhl;
  double* acc_up = acc_dn + nhl;
  const long long i = c.i1 + tid;
  const bool live = i <= c.i2;
  const size_t j = live ? (size_t)i : (size_t)c.i2;
  const double inv_cm_2_Hz = 100.0 * kLightC;
  const double freq = wn[j] * inv_cm_2_Hz;
  const double pref = live ? (dwn[j] * 2.0 * kPlanckH * inv_cm_2_Hz * kPi / (kLightC * kLightC)) * (freq * freq * freq) : 0.0;
  auto planck = [&](int level) { return ecckd::div_fast(pref, ecckd::exp_fast(freq * hk[level]) - 1.0); };
  double surf_acc = 0.0, toa_acc = 0.0;
  for (int a = 0; a < nang; ++a) {
    const double sec = ang[a], wgt = ang[nang + a];          // (uniform: scalar loads)
    const double rsec = 1.0 / sec;
    auto layer = [&](int l, double& eps, double& fac) {
      const double tau = (double)od[(size_t)l * od_stride + j];
      eps = 1.0 - ecckd::exp_fast(-sec * tau);
      fac = (eps > 1.0e-5) ? 1.0 - ecckd::div_fast(eps * rsec, tau) : 0.5 * eps;   // :41-43
    };
    // down sweep from zero at the top of the atmosphere (:45-50); dead lanes carry pref = 0 -> all fluxes 0
    double flux = 0.0;
    double b_prev = planck(0);
    for (int l = 0; l < nlay; ++l) {
      double eps, fac;
      layer(l, eps, fac);
      const double b_next = planck(l + 1);
      flux = flux * (1.0 - eps) + b_prev * (eps - fac) + b_next * fac;
      const double s = wave_sum(wgt * flux);
      if (lane == 0) acc_dn[l + 1] += s;
      b_prev = b_next;
    }
    surf_acc += wgt * flux;                         // the spectral flux at the boundary (do_write_spectral_boundary_fluxes)
    // surface: emissivity 1, Planck function at temperature_hl(end) (:52-53)
    flux = b_prev * 1.0 + (1.0 - 1.0) * flux;
    {
      const double s = wave_sum(wgt * flux);
      if (lane == 0) acc_up[nlay] += s;
    }
    for (int l = nlay - 1; l >= 0; --l) {                                    // :55-59
      double eps, fac;
      layer(l, eps, fac);
      const double b_l = planck(l);
      flux = flux * (1.0 - eps) + b_prev * (eps - fac) + b_l * fac;
      const double s = wave_sum(wgt * flux);
      if (lane == 0) acc_up[l] += s;
      b_prev = b_l;
    }
    toa_acc += wgt * flux;
  }
  if (surf_dn && live) surf_dn[j] = surf_acc;
  if (toa_up && live) toa_up[j] = toa_acc;
  __syncthreads();
  for (int t = tid; t < 2 * nhl; t += LBL_THREADS)
    partial[(size_t)blockIdx.x * 2 * nhl + t] = ((s_acc[t] + s_acc[2 * nhl + t]) + s_acc[4 * nhl + t]) + s_acc[6 * nhl + t];
}

template <typename OdT>
__global__ void __launch_bounds__(LBL_THREADS)
k_lbl_fluxes_sw(int nlay, size_t od_stride, const BandChunk* __restrict__ chunks, double cos_sza,
                const double* __restrict__ ssi, const double* __restrict__ albedo /* per wavenumber or NULL */,
                const OdT* __restrict__ od, double* __restrict__ partial, double* __restrict__ surf_dn /* [nwav] or NULL */,
                double* __restrict__ toa_up /* [nwav] or NULL */) {
  extern __shared__ double s_acc[];
  const int nhl = nlay + 1;
  const BandChunk c = chunks[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int t = tid; t < 4 * 2 * nhl; t += LBL_THREADS) s_acc[t] = 0.0;
  __syncthreads();
  double* acc_dn = s_acc + wave * 2 * nhl;
  double* acc_up = acc_dn + nhl;
  const long long i = c.i1 + tid;
  const bool live = i <= c.i2;
  const size_t j = live ? (size_t)i : (size_t)c.i2;
  const double minus_sec_sza = -1.0 / cos_sza;
  double flux = live ? cos_sza * ssi[j] : 0.0;                              // radiative_transfer_sw.cpp:39
  {
    const double s = wave_sum(flux);
    if (lane == 0) acc_dn[0] += s;
  }
  for (int l = 0; l < nlay; ++l) {
    flux = flux * exp(minus_sec_sza * (double)od[(size_t)l * od_stride + j]);
    const double s = wave_sum(flux);
    if (lane == 0) acc_dn[l + 1] += s;
  }
  if (surf_dn && live) surf_dn[j] = flux;
  if (toa_up && live && !albedo) toa_up[j] = 0.0;
  if (albedo) {                                                             // :70-76
    flux = flux * albedo[j];
    {
      const double s = wave_sum(flux);
      if (lane == 0) acc_up[nlay] += s;
    }
    for (int l = nlay - 1; l >= 0; --l) {
      flux = flux * exp(-2.0 * (double)od[(size_t)l * od_stride + j]);
      const double s = wave_sum(flux);
      if (lane == 0) acc_up[l] += s;
    }
    if (toa_up && live) toa_up[j] = flux;
  }
  __syncthreads();
  for (int t = tid; t < 2 * nhl; t += LBL_THREADS)
    partial[(size_t)blockIdx.x * 2 * nhl + t] = ((s_acc[t] + s_acc[2 * nhl + t]) + s_acc[4 * nhl + t]) + s_acc[6 * nhl + t];
}

// Fluxes of (level, g point) matrices - what run_ckd leaves for a flux evaluation (test/run_ckd_lw.sh:133-137: optical depth and
// Planck function per g point) - one thread per (column, g point): radiative_transfer_lw.cpp:27-60 with unit emissivity along the
// slant path sec * tau, nangle = 0 the two-stream form (sec = 1.66), otherwise the sum over the Gauss-Legendre angles with the
// weights 2 w mu (as k_lbl_lw does per wavenumber).
__global__ void __launch_bounds__(256)
k_rt_lw_gpoints(int ncol, int nlay, int ng, int nsec, const double* __restrict__ sec_wgt /*[2][nsec]*/, const double* __restrict__ planck,
                const double* __restrict__ od, double* __restrict__ dn, double* __restrict__ up) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)ncol * ng) return;
  const size_t c = t / ng, g = t % ng;
  const double* pl = planck + c * (size_t)(nlay + 1) * ng + g;
  const double* tau = od + c * (size_t)nlay * ng + g;
  double* fd = dn + c * (size_t)(nlay + 1) * ng + g;
  double* fu = up + c * (size_t)(nlay + 1) * ng + g;
  for (int l = 0; l <= nlay; ++l) { fd[(size_t)l * ng] = 0.0; fu[(size_t)l * ng] = 0.0; }
  for (int a = 0; a < nsec; ++a) {
    const double sec = sec_wgt[a], w = sec_wgt[nsec + a];
    double f = 0.0;
    for (int l = 0; l < nlay; ++l) {
      const double x = tau[(size_t)l * ng];
      const double e = 1.0 - exp(-sec * x);
      const double fac = e > 1.0e-5 ? 1.0 - e * (1.0 / sec) / x : 0.5 * e;
      f = f * (1.0 - e) + pl[(size_t)l * ng] * (e - fac) + pl[(size_t)(l + 1) * ng] * fac;
      fd[(size_t)(l + 1) * ng] += w * f;
    }
    f = pl[(size_t)nlay * ng];
    fu[(size_t)nlay * ng] += w * f;
    for (int l = nlay - 1; l >= 0; --l) {
      const double x = tau[(size_t)l * ng];
      const double e = 1.0 - exp(-sec * x);
      const double fac = e > 1.0e-5 ? 1.0 - e * (1.0 / sec) / x : 0.5 * e;
      f = f * (1.0 - e) + pl[(size_t)(l + 1) * ng] * (e - fac) + pl[(size_t)l * ng] * fac;
      fu[(size_t)l * ng] += w * f;
    }
  }
}

// direct beam and surface-reflected upwelling flux per g point: radiative_transfer_sw.cpp:45-77 (norayleigh)
__global__ void __launch_bounds__(256)
k_rt_sw_gpoints(int ncol, int nlay, int ng, double mu0, double albedo, const double* __restrict__ incoming /*[ncol][ng]*/,
                const double* __restrict__ od, double* __restrict__ dn, double* __restrict__ up) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)ncol * ng) return;
  const size_t c = t / ng, g = t % ng;
  const double* tau = od + c * (size_t)nlay * ng + g;
  double* fd = dn + c * (size_t)(nlay + 1) * ng + g;
  double* fu = up + c * (size_t)(nlay + 1) * ng + g;
  double f = mu0 * incoming[c * ng + g];
  fd[0] = f;
  for (int l = 0; l < nlay; ++l) { f = f * exp(-tau[(size_t)l * ng] / mu0); fd[(size_t)(l + 1) * ng] = f; }
  f = f * albedo;
  fu[(size_t)nlay * ng] = f;
  for (int l = nlay - 1; l >= 0; --l) { f = f * exp(-2.0 * tau[(size_t)l * ng]); fu[(size_t)l * ng] = f; }
}

struct Buf {
  void* p = nullptr;
  ~Buf() { if (p) (void)hipFree(p); }
};

int make_chunks(size_t nwav, int nband, const int64_t* b0, const int64_t* b1, std::vector<BandChunk>& chunks) {
  for (int b = 0; b < nband; ++b) {
    if (b1[b] < b0[b]) continue;                       // empty band
    ECCKD_REQUIRE(b0[b] >= 0 && (size_t)b1[b] < nwav, "band %d range [%lld,%lld] outside the spectrum", b, (long long)b0[b],
                  (long long)b1[b]);
    for (long long i = b0[b]; i <= b1[b]; i += LBL_THREADS)
      chunks.push_back(BandChunk{i, std::min<long long>(i + LBL_THREADS - 1, b1[b]), b, 0});
  }
  return ECCKD_OK;
}

int combine(ecckd_ctx* ctx, int nlay, int nband, const std::vector<BandChunk>& chunks, const double* d_partial,
            double* h_flux_dn, double* h_flux_up) {
  const int nhl = nlay + 1;
  std::vector<double> part(chunks.size() * 2 * nhl);
  if (!chunks.empty()) ECCKD_CHECK(ecckd_d2h(ctx, part.data(), d_partial, part.size() * sizeof(double)));
  std::fill(h_flux_dn, h_flux_dn + (size_t)nband * nhl, 0.0);
  if (h_flux_up) std::fill(h_flux_up, h_flux_up + (size_t)nband * nhl, 0.0);
  for (size_t c = 0; c < chunks.size(); ++c)          // chunk order = wavenumber order within each band
    for (int i = 0; i < nhl; ++i) {
      h_flux_dn[(size_t)chunks[c].band * nhl + i] += part[c * 2 * nhl + i];
      if (h_flux_up) h_flux_up[(size_t)chunks[c].band * nhl + i] += part[c * 2 * nhl + nhl + i];
    }
  return ECCKD_OK;
}

}  // namespace

extern "C" {

int ecckd_lbl_band_fluxes_lw(ecckd_ctx* ctx, int nlay, size_t nwav, const double* h_temperature_hl,
                             const double* d_wavenumber, const double* d_d_wavenumber, const void* d_od, int od_type,
                             size_t od_stride, int nband, const int64_t* h_band_begin, const int64_t* h_band_end,
                             double* h_flux_dn, double* h_flux_up) {
  return ecckd_lbl_band_fluxes_lw_ex(ctx, nlay, nwav, h_temperature_hl, d_wavenumber, d_d_wavenumber, d_od, od_type, od_stride, nband,
                                     h_band_begin, h_band_end, h_flux_dn, h_flux_up, nullptr, nullptr);
}

int ecckd_lbl_band_fluxes_lw_ex(ecckd_ctx* ctx, int nlay, size_t nwav, const double* h_temperature_hl,
                                const double* d_wavenumber, const double* d_d_wavenumber, const void* d_od, int od_type,
                                size_t od_stride, int nband, const int64_t* h_band_begin, const int64_t* h_band_end,
                                double* h_flux_dn, double* h_flux_up, double* d_surf_dn, double* d_toa_up) {
  return ecckd_lbl_band_fluxes_lw_angles(ctx, 0, nlay, nwav, h_temperature_hl, d_wavenumber, d_d_wavenumber, d_od, od_type, od_stride,
                                         nband, h_band_begin, h_band_end, h_flux_dn, h_flux_up, d_surf_dn, d_toa_up);
}

int ecckd_gauss_legendre_01(int n, double* h_mu, double* h_weight) {
  ECCKD_REQUIRE(n >= 1 && n <= LBL_MAX_ANGLES && h_mu && h_weight, "ecckd_gauss_legendre_01: 1 <= n <= %d", LBL_MAX_ANGLES);
  // nodes of P_n on (-1, 1) by Newton's iteration from the Chebyshev guess, mapped to (0, 1); ascending mu
  for (int i = 0; i < n; ++i) {
    double t = std::cos(3.14159265358979323846 * (i + 0.75) / (n + 0.5));
    double dp = 1.0;
    for (int it = 0; it < 100; ++it) {
      double p0 = 1.0, p1 = t;
      for (int k = 2; k <= n; ++k) { const double pk = ((2.0 * k - 1.0) * t * p1 - (k - 1.0) * p0) / k; p0 = p1; p1 = pk; }
      if (n == 1) { p0 = 1.0; p1 = t; }
      dp = n * (t * p1 - p0) / (t * t - 1.0);
      const double dt = p1 / dp;
      t -= dt;
      if (std::fabs(dt) < 1e-16) break;
    }
    {
      double p0 = 1.0, p1 = t;
      for (int k = 2; k <= n; ++k) { const double pk = ((2.0 * k - 1.0) * t * p1 - (k - 1.0) * p0) / k; p0 = p1; p1 = pk; }
      dp = n * (t * p1 - p0) / (t * t - 1.0);
    }
    const double w = 2.0 / ((1.0 - t * t) * dp * dp);
    h_mu[n - 1 - i] = 0.5 * (1.0 + t);
    h_weight[n - 1 - i] = 0.5 * w;
  }
  return ECCKD_OK;
}

int ecckd_lbl_band_fluxes_lw_angles(ecckd_ctx* ctx, int nangle, int nlay, size_t nwav, const double* h_temperature_hl,
                                    const double* d_wavenumber, const double* d_d_wavenumber, const void* d_od, int od_type,
                                    size_t od_stride, int nband, const int64_t* h_band_begin, const int64_t* h_band_end,
                                    double* h_flux_dn, double* h_flux_up, double* d_surf_dn, double* d_toa_up) {
  ECCKD_REQUIRE(nangle >= 0 && nangle <= LBL_MAX_ANGLES, "ecckd_lbl_band_fluxes_lw: nangle = %d outside 0..%d", nangle, LBL_MAX_ANGLES);
  Angles ang{};
  if (nangle == 0) {
    ang.n = 1; ang.sec[0] = ECCKD_LW_DIFFUSIVITY; ang.weight[0] = 1.0;
  } else {
    double mu[LBL_MAX_ANGLES], w[LBL_MAX_ANGLES];
    ECCKD_CHECK(ecckd_gauss_legendre_01(nangle, mu, w));
    ang.n = nangle;
    for (int a = 0; a < nangle; ++a) { ang.sec[a] = 1.0 / mu[a]; ang.weight[a] = 2.0 * w[a] * mu[a]; }
  }
  ECCKD_REQUIRE(ctx && nlay > 0 && h_temperature_hl && d_wavenumber && d_d_wavenumber && d_od && nband > 0 && h_band_begin &&
                h_band_end && h_flux_dn && h_flux_up, "ecckd_lbl_band_fluxes_lw: bad argument");
  ECCKD_REQUIRE(od_type == ECCKD_F32 || od_type == ECCKD_F64, "ecckd_lbl_band_fluxes_lw: od_type must be 4 or 8");
  ECCKD_REQUIRE(od_stride >= nwav, "ecckd_lbl_band_fluxes_lw: od_stride (%zu) < nwav (%zu)", od_stride, nwav);
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  const int nhl = nlay + 1;
  std::vector<BandChunk> chunks;
  ECCKD_CHECK(make_chunks(nwav, nband, h_band_begin, h_band_end, chunks));
  // wavenumbers outside every band carry no flux
  if (d_surf_dn) ECCKD_HIP_CHECK(hipMemsetAsync(d_surf_dn, 0, nwav * sizeof(double), ctx->stream));
  if (d_toa_up) ECCKD_HIP_CHECK(hipMemsetAsync(d_toa_up, 0, nwav * sizeof(double), ctx->stream));
  std::vector<double> hk(nhl);
  for (int i = 0; i < nhl; ++i) {
    ECCKD_REQUIRE(h_temperature_hl[i] > 0.0, "ecckd_lbl_band_fluxes_lw: temperature_hl must be positive");
    hk[i] = (6.62606896e-34 / 1.3806504e-23) / h_temperature_hl[i];
  }
  Buf d_chunks, d_hk, d_part;
  if (!chunks.empty()) {
    ECCKD_HIP_CHECK(hipMalloc(&d_chunks.p, chunks.size() * sizeof(BandChunk)));
    ECCKD_HIP_CHECK(hipMalloc(&d_hk.p, (nhl + 2 * LBL_MAX_ANGLES) * sizeof(double)));
    ECCKD_HIP_CHECK(hipMalloc(&d_part.p, chunks.size() * 2 * nhl * sizeof(double)));
    ECCKD_CHECK(ecckd_h2d(ctx, d_chunks.p, chunks.data(), chunks.size() * sizeof(BandChunk)));
    for (int a = 0; a < ang.n; ++a) hk.push_back(ang.sec[a]);
    for (int a = 0; a < ang.n; ++a) hk.push_back(ang.weight[a]);
    ECCKD_CHECK(ecckd_h2d(ctx, d_hk.p, hk.data(), hk.size() * sizeof(double)));
    const double* d_ang = (const double*)d_hk.p + nhl;
    const size_t lds = (size_t)4 * 2 * nhl * sizeof(double);
    if (od_type == ECCKD_F32)
      hipLaunchKernelGGL(k_lbl_fluxes_lw<float>, dim3((unsigned)chunks.size()), dim3(LBL_THREADS), lds, ctx->stream, ang.n, d_ang, nlay,
                         od_stride, (const BandChunk*)d_chunks.p, (const double*)d_hk.p, d_wavenumber, d_d_wavenumber,
                         (const float*)d_od, (double*)d_part.p, d_surf_dn, d_toa_up);
    else
      hipLaunchKernelGGL(k_lbl_fluxes_lw<double>, dim3((unsigned)chunks.size()), dim3(LBL_THREADS), lds, ctx->stream, ang.n, d_ang, nlay,
                         od_stride, (const BandChunk*)d_chunks.p, (const double*)d_hk.p, d_wavenumber, d_d_wavenumber,
                         (const double*)d_od, (double*)d_part.p, d_surf_dn, d_toa_up);
    ECCKD_HIP_CHECK(hipGetLastError());
  }
  return combine(ctx, nlay, nband, chunks, (const double*)d_part.p, h_flux_dn, h_flux_up);
}

int ecckd_lbl_band_fluxes_sw(ecckd_ctx* ctx, int nlay, size_t nwav, double cos_sza, const double* d_ssi,
                             const double* d_albedo, const void* d_od, int od_type, size_t od_stride, int nband,
                             const int64_t* h_band_begin, const int64_t* h_band_end, double* h_flux_dn_direct,
                             double* h_flux_up) {
  return ecckd_lbl_band_fluxes_sw_ex(ctx, nlay, nwav, cos_sza, d_ssi, d_albedo, d_od, od_type, od_stride, nband, h_band_begin,
                                     h_band_end, h_flux_dn_direct, h_flux_up, nullptr, nullptr);
}

int ecckd_lbl_band_fluxes_sw_ex(ecckd_ctx* ctx, int nlay, size_t nwav, double cos_sza, const double* d_ssi,
                                const double* d_albedo, const void* d_od, int od_type, size_t od_stride, int nband,
                                const int64_t* h_band_begin, const int64_t* h_band_end, double* h_flux_dn_direct,
                                double* h_flux_up, double* d_surf_dn_direct, double* d_toa_up) {
  double* const d_surf_dn = d_surf_dn_direct;
  ECCKD_REQUIRE(ctx && nlay > 0 && d_ssi && d_od && nband > 0 && h_band_begin && h_band_end && h_flux_dn_direct,
                "ecckd_lbl_band_fluxes_sw: bad argument");
  ECCKD_REQUIRE(cos_sza > 0.0, "ecckd_lbl_band_fluxes_sw: cos_sza must be positive");
  ECCKD_REQUIRE(od_type == ECCKD_F32 || od_type == ECCKD_F64, "ecckd_lbl_band_fluxes_sw: od_type must be 4 or 8");
  ECCKD_REQUIRE(od_stride >= nwav, "ecckd_lbl_band_fluxes_sw: od_stride (%zu) < nwav (%zu)", od_stride, nwav);
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  const int nhl = nlay + 1;
  std::vector<BandChunk> chunks;
  ECCKD_CHECK(make_chunks(nwav, nband, h_band_begin, h_band_end, chunks));
  if (d_surf_dn) ECCKD_HIP_CHECK(hipMemsetAsync(d_surf_dn, 0, nwav * sizeof(double), ctx->stream));
  if (d_toa_up) ECCKD_HIP_CHECK(hipMemsetAsync(d_toa_up, 0, nwav * sizeof(double), ctx->stream));
  Buf d_chunks, d_part;
  if (!chunks.empty()) {
    ECCKD_HIP_CHECK(hipMalloc(&d_chunks.p, chunks.size() * sizeof(BandChunk)));
    ECCKD_HIP_CHECK(hipMalloc(&d_part.p, chunks.size() * 2 * nhl * sizeof(double)));
    ECCKD_CHECK(ecckd_h2d(ctx, d_chunks.p, chunks.data(), chunks.size() * sizeof(BandChunk)));
    const size_t lds = (size_t)4 * 2 * nhl * sizeof(double);
    if (od_type == ECCKD_F32)
      hipLaunchKernelGGL(k_lbl_fluxes_sw<float>, dim3((unsigned)chunks.size()), dim3(LBL_THREADS), lds, ctx->stream, nlay,
                         od_stride, (const BandChunk*)d_chunks.p, cos_sza, d_ssi, d_albedo, (const float*)d_od, (double*)d_part.p,
                         d_surf_dn, d_toa_up);
    else
      hipLaunchKernelGGL(k_lbl_fluxes_sw<double>, dim3((unsigned)chunks.size()), dim3(LBL_THREADS), lds, ctx->stream, nlay,
                         od_stride, (const BandChunk*)d_chunks.p, cos_sza, d_ssi, d_albedo, (const double*)d_od, (double*)d_part.p,
                         d_surf_dn, d_toa_up);
    ECCKD_HIP_CHECK(hipGetLastError());
  }
  return combine(ctx, nlay, nband, chunks, (const double*)d_part.p, h_flux_dn_direct, h_flux_up);
}

// The fluxes of a CKD model's g points from the optical depths and Planck functions run_ckd wrote (the `--ckd` mode of the
// CKDMIP tools as the scripts use them, test/run_ckd_lw.sh:133-137, test/run_ckd_sw.sh:125-128).  Host arrays in and out (a few
// thousand values per column), the radiative transfer on the device.
int ecckd_rt_lw_gpoints(ecckd_ctx* ctx, int nangle, int ncol, int nlay, int ng, const double* h_planck_hl, const double* h_od,
                        double* h_flux_dn, double* h_flux_up) {
  ECCKD_REQUIRE(ctx && ncol > 0 && nlay > 0 && ng > 0 && h_planck_hl && h_od && h_flux_dn && h_flux_up, "ecckd_rt_lw_gpoints: bad argument");
  ECCKD_REQUIRE(nangle >= 0 && nangle <= LBL_MAX_ANGLES, "ecckd_rt_lw_gpoints: nangle %d outside 0..%d", nangle, LBL_MAX_ANGLES);
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  const int nsec = nangle == 0 ? 1 : nangle;
  std::vector<double> sw(2 * (size_t)nsec);
  if (nangle == 0) { sw[0] = ECCKD_LW_DIFFUSIVITY; sw[1] = 1.0; }
  else {
    std::vector<double> mu(nangle), w(nangle);
    ECCKD_CHECK(ecckd_gauss_legendre_01(nangle, mu.data(), w.data()));
    for (int a = 0; a < nangle; ++a) { sw[a] = 1.0 / mu[a]; sw[nsec + a] = 2.0 * w[a] * mu[a]; }
  }
  const size_t nl = (size_t)ncol * nlay * ng, nh = (size_t)ncol * (nlay + 1) * ng;
  const size_t b_sw = ecckd_align_up(sw.size() * sizeof(double), 256), b_l = ecckd_align_up(nl * sizeof(double), 256),
               b_h = ecckd_align_up(nh * sizeof(double), 256);
  ECCKD_CHECK(ecckd::ensure_scratch(ctx, b_sw + b_l + 3 * b_h));
  char* q = (char*)ctx->scratch;
  double* d_sw = (double*)q; q += b_sw;
  double* d_od = (double*)q; q += b_l;
  double* d_pl = (double*)q; q += b_h;
  double* d_dn = (double*)q; q += b_h;
  double* d_up = (double*)q;
  ECCKD_HIP_CHECK(hipMemcpyAsync(d_sw, sw.data(), sw.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ECCKD_HIP_CHECK(hipMemcpyAsync(d_od, h_od, nl * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ECCKD_HIP_CHECK(hipMemcpyAsync(d_pl, h_planck_hl, nh * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_rt_lw_gpoints, dim3((unsigned)(((size_t)ncol * ng + 255) / 256)), dim3(256), 0, ctx->stream, ncol, nlay, ng, nsec, d_sw,
                     d_pl, d_od, d_dn, d_up);
  ECCKD_HIP_CHECK(hipGetLastError());
  ECCKD_HIP_CHECK(hipMemcpyAsync(h_flux_dn, d_dn, nh * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ECCKD_HIP_CHECK(hipMemcpyAsync(h_flux_up, d_up, nh * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return ECCKD_OK;
}

int ecckd_rt_sw_gpoints(ecckd_ctx* ctx, int ncol, int nlay, int ng, double cos_sza, double albedo, const double* h_incoming,
                        const double* h_od, double* h_flux_dn, double* h_flux_up) {
  ECCKD_REQUIRE(ctx && ncol > 0 && nlay > 0 && ng > 0 && h_incoming && h_od && h_flux_dn && h_flux_up, "ecckd_rt_sw_gpoints: bad argument");
  ECCKD_REQUIRE(cos_sza > 0.0, "ecckd_rt_sw_gpoints: cos_sza must be positive");
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  const size_t nl = (size_t)ncol * nlay * ng, nh = (size_t)ncol * (nlay + 1) * ng, ni = (size_t)ncol * ng;
  const size_t b_i = ecckd_align_up(ni * sizeof(double), 256), b_l = ecckd_align_up(nl * sizeof(double), 256),
               b_h = ecckd_align_up(nh * sizeof(double), 256);
  ECCKD_CHECK(ecckd::ensure_scratch(ctx, b_i + b_l + 2 * b_h));
  char* q = (char*)ctx->scratch;
  double* d_in = (double*)q; q += b_i;
  double* d_od = (double*)q; q += b_l;
  double* d_dn = (double*)q; q += b_h;
  double* d_up = (double*)q;
  ECCKD_HIP_CHECK(hipMemcpyAsync(d_in, h_incoming, ni * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ECCKD_HIP_CHECK(hipMemcpyAsync(d_od, h_od, nl * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_rt_sw_gpoints, dim3((unsigned)((ni + 255) / 256)), dim3(256), 0, ctx->stream, ncol, nlay, ng, cos_sza, albedo, d_in,
                     d_od, d_dn, d_up);
  ECCKD_HIP_CHECK(hipGetLastError());
  ECCKD_HIP_CHECK(hipMemcpyAsync(h_flux_dn, d_dn, nh * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ECCKD_HIP_CHECK(hipMemcpyAsync(h_flux_up, d_up, nh * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return ECCKD_OK;
}

}  // extern "C"
