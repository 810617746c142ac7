// reorder.hip - K1/K2: per-wavenumber sorting key of reorder_spectrum on gfx950.
//
// Replaces, for one column, reference src/ecckd/reorder_spectrum.cpp:111-228:
//   planck_function            (planck_function.cpp:22-54)
//   radiative_transfer_lw      (radiative_transfer_lw.cpp:27-60)
//   heating_rate               (heating_rate.h:30-50)
//   peak-cooling pseudo height (reorder_spectrum.cpp:175-190)
//   threshold height           (reorder_spectrum.cpp:197-228, the SW key)
//
// Data layout: optical depth is (level, wavenumber) row-major with row stride
// od_stride, f32 (as in the CKDMIP files) or f64.  One thread owns one
// wavenumber, so every load of a layer row is a fully coalesced 256/512-byte
// wave access; nothing but the two 8-byte results per point is written.
//
// The reference materialises seven (nlay+1, nwav) f64 matrices in RAM; here
// the Planck function, emissivity and both flux sweeps live in registers and
// the per-layer net-flux increments of the down sweep in LDS, so HBM traffic
// is the algorithmic minimum: nlay*sizeof(od) + 16 B read, 16 B written.
#include "common.hpp"
#include "fastmath.hpp"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr int KEY_THREADS = 256;

// Per-level constants, packed in one device array:
//   hk_over_t[nhl] | conv[nlay] | dh[nlay] | dhph[nlay] | ph_hl[nhl]
struct LevelLayout {
  int nlay;
  __host__ __device__ int hk() const { return 0; }
  __host__ __device__ int conv() const { return nlay + 1; }
  __host__ __device__ int dh() const { return 2 * nlay + 1; }
  __host__ __device__ int dhph() const { return 3 * nlay + 1; }
  __host__ __device__ int phhl() const { return 4 * nlay + 1; }
  __host__ __device__ int total() const { return 5 * nlay + 2; }
};

// planck_function.cpp:29-33
__device__ constexpr double kPlanckH = 6.62606896e-34;
__device__ constexpr double kLightC = 2.99792458e8;
__device__ constexpr double kPi = 3.14159265358979323846;

// K1.  LW sorting key.  Dynamic LDS: double[nlay][blockDim.x] holding the
// down-sweep flux increments dn[l+1]-dn[l], then the clamped heating rates.
template <typename OdT>
__global__ void __launch_bounds__(KEY_THREADS)
k_reorder_key_lw(int nlay, size_t nwav, size_t od_stride, const double* __restrict__ lev,
                 const double* __restrict__ wn, const double* __restrict__ dwn,
                 const OdT* __restrict__ od, double thr, double* __restrict__ key,
                 double* __restrict__ col_od_out, int* __restrict__ err_flag) {
  extern __shared__ double s_col[];  // [nlay][blockDim.x]
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nwav) return;  // no block-level barrier below
  const int tid = threadIdx.x;
  const int bs = blockDim.x;
  const LevelLayout L{nlay};
  const double* hk = lev + L.hk();
  const double* conv = lev + L.conv();
  const double* dh = lev + L.dh();
  const double* dhph = lev + L.dhph();
  const double* phhl = lev + L.phhl();

  // planck_function.cpp:48-50, same operation order
  const double inv_cm_2_Hz = 100.0 * kLightC;
  const double freq = wn[j] * inv_cm_2_Hz;
  const double pref = (dwn[j] * 2.0 * kPlanckH * inv_cm_2_Hz * kPi / (kLightC * kLightC)) *
                      (freq * freq * freq);

  // ---- down sweep (radiative_transfer_lw.cpp:45-50) ----
  double b_prev = ecckd::div_fast(pref, ecckd::exp_fast(freq * hk[0]) - 1.0);
  double dn = 0.0;
  double col = 0.0;
  double thr_height = 0.0;
  bool crossed = false;
  const OdT* odp = od + j;
  for (int l = 0; l < nlay; ++l) {
    const double tau = (double)odp[(size_t)l * od_stride];
    const double eps = 1.0 - ecckd::exp_fast(-ECCKD_LW_DIFFUSIVITY * tau);
    // :42-43  factor = eps > 1e-5 ? 1 - eps*(1/D)/tau : 0.5*eps
    const double fac = (eps > 1.0e-5) ? 1.0 - ecckd::div_fast(eps * (1.0 / ECCKD_LW_DIFFUSIVITY), tau) : 0.5 * eps;
    const double b_next = ecckd::div_fast(pref, ecckd::exp_fast(freq * hk[l + 1]) - 1.0);
    const double dn_next = dn * (1.0 - eps) + b_prev * (eps - fac) + b_next * fac;
    s_col[l * bs + tid] = dn_next - dn;
    // reorder_spectrum.cpp:199-222 (threshold height; in LW only its throw is observable)
    const double next_col = col + tau;
    if (!crossed && next_col >= thr) {
      thr_height = ((thr - col) * phhl[l + 1] + (next_col - thr) * phhl[l]) / fmax(1.0e-12, tau);
      crossed = true;
    }
    col = next_col;
    dn = dn_next;
    b_prev = b_next;
  }
  if (col > thr && thr_height > 30.0) atomicOr(err_flag, 1);

  // ---- surface (:52-53, emissivity 1) and up sweep (:55-59) ----
  // surf_planck is planck at temperature_hl(end) (reorder_spectrum.cpp:130-131)
  // = b_prev; with surf_emissivity = 1 the reflected term is 0*dn.
  double up = b_prev * 1.0 + (1.0 - 1.0) * dn;
  for (int l = nlay - 1; l >= 0; --l) {
    const double tau = (double)odp[(size_t)l * od_stride];
    const double eps = 1.0 - ecckd::exp_fast(-ECCKD_LW_DIFFUSIVITY * tau);
    const double fac = (eps > 1.0e-5) ? 1.0 - ecckd::div_fast(eps * (1.0 / ECCKD_LW_DIFFUSIVITY), tau) : 0.5 * eps;
    const double b_l = ecckd::div_fast(pref, ecckd::exp_fast(freq * hk[l]) - 1.0);
    const double up_l = up * (1.0 - eps) + b_prev * (eps - fac) + b_l * fac;
    // heating_rate.h:47-48: conv * (dn[l+1]-dn[l]-up[l+1]+up[l]), left to right
    double hr = conv[l] * (s_col[l * bs + tid] - up + up_l);
    // reorder_spectrum.cpp:175: only cooling
    if (hr > 0.0) hr = 0.0;
    s_col[l * bs + tid] = hr;
    up = up_l;
    b_prev = b_l;
  }

  // ---- peak-cooling pseudo height (:178-183), sums over layers ascending ----
  double num = 0.0, den = 0.0;
  for (int l = 0; l < nlay; ++l) {
    const double hr = s_col[l * bs + tid];
    num += hr * dhph[l];
    den += hr * dh[l];
  }
  double k = num / den;
  // :187-190
  if (thr > 0.0 && col < thr) k = -thr + col;
  key[j] = k;
  col_od_out[j] = col;
}


// K1 fast path: NLAY known at compile time, one wave per SIMD (512 registers per lane: the
// per-layer flux increment, emissivity and upward source of the whole column stay in
// VGPR/AGPRs), ONE Planck exp and ONE emissivity exp per layer instead of two each, no LDS.
template <int NLAY, typename OdT>
__global__ void __launch_bounds__(KEY_THREADS, 1)
k_reorder_key_lw_fast(size_t nwav, size_t od_stride, const double* __restrict__ lev,
                      const double* __restrict__ wn, const double* __restrict__ dwn,
                      const OdT* __restrict__ od, double thr, double* __restrict__ key,
                      double* __restrict__ col_od_out, int* __restrict__ err_flag) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nwav) return;
  const LevelLayout L{NLAY};
  const double* hk = lev + L.hk();
  const double* conv = lev + L.conv();
  const double* dh = lev + L.dh();
  const double* dhph = lev + L.dhph();
  const double* phhl = lev + L.phhl();
  const double inv_cm_2_Hz = 100.0 * kLightC;
  const double freq = wn[j] * inv_cm_2_Hz;
  const double pref = (dwn[j] * 2.0 * kPlanckH * inv_cm_2_Hz * kPi / (kLightC * kLightC)) *
                      (freq * freq * freq);
  const ecckd::ExpConsts ek = ecckd::exp_consts();
  const double neg_d = ecckd::sgpr_pin(-ECCKD_LW_DIFFUSIVITY), inv_d = ecckd::sgpr_pin(1.0 / ECCKD_LW_DIFFUSIVITY);
  const double thin = ecckd::sgpr_pin(1.0e-5);
  const OdT* odp = od + j;
  // the optical depths are fetched a quarter of the column ahead of their use
  constexpr int AHEAD = 14;
  OdT tau_in[NLAY];
#pragma unroll
  for (int l = 0; l < AHEAD; ++l) tau_in[l] = odp[(size_t)l * od_stride];

  double dd[NLAY];  // dn[l+1] - dn[l], later the clamped heating rate
  double ee[NLAY];  // emissivity
  double ss[NLAY];  // upward source B_{l+1}(eps - fac) + B_l fac
  double b_prev = ecckd::div_fast(pref, ecckd::exp_fast_s(freq * hk[0], ek) - 1.0);
  double dn = 0.0, col = 0.0, thr_height = 0.0;
  bool crossed = false;
#pragma unroll
  for (int l = 0; l < NLAY; ++l) {
    if (l + AHEAD < NLAY) tau_in[l + AHEAD] = odp[(size_t)(l + AHEAD) * od_stride];
    const double tau = (double)tau_in[l];
    const double eps = 1.0 - ecckd::exp_fast_s(neg_d * tau, ek);
    const double fac = (eps > thin) ? 1.0 - ecckd::div_fast(eps * inv_d, tau) : 0.5 * eps;
    const double b_next = ecckd::div_fast(pref, ecckd::exp_fast_s(freq * hk[l + 1], ek) - 1.0);
    const double emf = eps - fac;
    const double dn_next = dn * (1.0 - eps) + b_prev * emf + b_next * fac;
    dd[l] = dn_next - dn;
    ee[l] = eps;
    ss[l] = b_next * emf + b_prev * fac;
    const double next_col = col + tau;
    if (!crossed && next_col >= thr) {
      thr_height = ((thr - col) * phhl[l + 1] + (next_col - thr) * phhl[l]) / fmax(1.0e-12, tau);
      crossed = true;
    }
    col = next_col;
    dn = dn_next;
    b_prev = b_next;
  }
  if (col > thr && thr_height > 30.0) atomicOr(err_flag, 1);
  double up = b_prev * 1.0 + (1.0 - 1.0) * dn;
#pragma unroll
  for (int l = NLAY - 1; l >= 0; --l) {
    const double up_l = up * (1.0 - ee[l]) + ss[l];
    double hr = conv[l] * (dd[l] - up + up_l);
    if (hr > 0.0) hr = 0.0;
    dd[l] = hr;
    up = up_l;
  }
  double num = 0.0, den = 0.0;
#pragma unroll
  for (int l = 0; l < NLAY; ++l) {
    num += dd[l] * dhph[l];
    den += dd[l] * dh[l];
  }
  double k = num / den;
  if (thr > 0.0 && col < thr) k = -thr + col;
  key[j] = k;
  col_od_out[j] = col;
}

// K2.  SW sorting key (reorder_spectrum.cpp:197-228): pseudo height at which
// the optical depth accumulated from TOA reaches the threshold.  The direct
// beam RT and heating rate of :150-183 do not influence the SW key (it is
// overwritten at :227), so this kernel is a pure HBM stream.
template <typename OdT>
__global__ void __launch_bounds__(KEY_THREADS)
k_reorder_key_sw(int nlay, size_t nwav, size_t od_stride, const double* __restrict__ lev,
                 const OdT* __restrict__ od, double thr, double* __restrict__ key,
                 double* __restrict__ col_od_out, int* __restrict__ err_flag) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nwav) return;
  const LevelLayout L{nlay};
  const double* phhl = lev + L.phhl();
  const OdT* odp = od + j;
  double col = 0.0, thr_height = 0.0;
  bool crossed = false;
  for (int l = 0; l < nlay; ++l) {
    const double tau = (double)odp[(size_t)l * od_stride];
    const double next_col = col + tau;
    if (!crossed && next_col >= thr) {
      thr_height = ((thr - col) * phhl[l + 1] + (next_col - thr) * phhl[l]) / fmax(1.0e-12, tau);
      crossed = true;
    }
    col = next_col;
  }
  double k;
  if (col <= thr) {
    k = col - thr;
  } else {
    k = thr_height;
    if (k > 30.0) atomicOr(err_flag, 1);
  }
  key[j] = k;
  col_od_out[j] = col;
}

int upload_level_consts(ecckd_ctx* ctx, int nlay, const double* p_hl, const double* t_hl,
                        double** d_lev, int** d_flag) {
  const LevelLayout L{nlay};
  const int n = L.total();
  std::vector<double> h(n, 0.0);
  const double hk = 6.62606896e-34 / 1.3806504e-23;  // h/k, planck_function.cpp:29-31
  for (int i = 0; i <= nlay; ++i) {
    h[L.hk() + i] = t_hl ? hk / t_hl[i] : 0.0;
    h[L.phhl() + i] = std::log(p_hl[nlay]) - std::log(p_hl[i]);  // reorder_spectrum.cpp:196
  }
  for (int l = 0; l < nlay; ++l) {
    // heating_rate.h:38
    h[L.conv() + l] = -(ECCKD_ACCEL_GRAVITY / ECCKD_SPECIFIC_HEAT_AIR) / (p_hl[l + 1] - p_hl[l]);
    // reorder_spectrum.cpp:178-180
    const double pseudo_height =
        std::log(p_hl[nlay]) - 0.5 * (std::log(p_hl[l]) + std::log(p_hl[l + 1]));
    const double d_height = std::log(p_hl[l + 1]) - std::log(p_hl[l]);
    h[L.dh() + l] = d_height;
    h[L.dhph() + l] = d_height * pseudo_height;
  }
  const size_t bytes = ecckd_align_up((size_t)n * sizeof(double), 256) + 256;
  ECCKD_CHECK(ecckd::ensure_scratch(ctx, bytes));
  ECCKD_CHECK(ecckd::ensure_pinned(ctx, bytes));
  std::memcpy(ctx->pinned, h.data(), (size_t)n * sizeof(double));
  std::memset((char*)ctx->pinned + bytes - 256, 0, 256);
  ECCKD_HIP_CHECK(hipMemcpyAsync(ctx->scratch, ctx->pinned, bytes, hipMemcpyHostToDevice, ctx->stream));
  *d_lev = (double*)ctx->scratch;
  *d_flag = (int*)((char*)ctx->scratch + bytes - 256);
  return ECCKD_OK;
}

int check_flag(ecckd_ctx* ctx, const int* d_flag, const char* what) {
  int flag = 0;
  ECCKD_HIP_CHECK(hipMemcpyAsync(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (flag) {
    // the reference executes a bare `throw;` here (reorder_spectrum.cpp:214-216)
    return ecckd::fail(ECCKD_PROCESSING_ERROR, "%s: threshold pseudo-height exceeds 30", what);
  }
  return ECCKD_OK;
}

bool pressures_ok(int nlay, const double* p) {
  for (int i = 0; i <= nlay; ++i)
    if (!(p[i] > 0.0)) return false;
  for (int l = 0; l < nlay; ++l)
    if (!(p[l + 1] > p[l])) return false;
  return true;
}

}  // namespace

extern "C" {

int ecckd_reorder_key_lw_dev(ecckd_ctx* ctx, int nlay, size_t nwav, const double* h_pressure_hl,
                             const double* h_temperature_hl, const double* d_wavenumber,
                             const double* d_d_wavenumber, const void* d_od, int od_type,
                             size_t od_stride, double thr, double* d_key, double* d_col_od) {
  ECCKD_REQUIRE(ctx, "ecckd_reorder_key_lw_dev: ctx is NULL");
  ECCKD_REQUIRE(nlay > 0, "ecckd_reorder_key_lw_dev: nlay must be positive (got %d)", nlay);
  ECCKD_REQUIRE(h_pressure_hl && h_temperature_hl && d_wavenumber && d_d_wavenumber && d_od && d_key && d_col_od,
                "ecckd_reorder_key_lw_dev: NULL array argument");
  ECCKD_REQUIRE(od_type == ECCKD_F32 || od_type == ECCKD_F64, "ecckd_reorder_key_lw_dev: od_type must be 4 or 8");
  ECCKD_REQUIRE(od_stride >= nwav, "ecckd_reorder_key_lw_dev: od_stride (%zu) < nwav (%zu)", od_stride, nwav);
  ECCKD_REQUIRE(pressures_ok(nlay, h_pressure_hl), "ecckd_reorder_key_lw_dev: pressure_hl must be positive and increasing");
  for (int i = 0; i <= nlay; ++i)
    ECCKD_REQUIRE(h_temperature_hl[i] > 0.0, "ecckd_reorder_key_lw_dev: temperature_hl must be positive");
  if (nwav == 0) return ECCKD_OK;
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));

  // block size such that the [nlay][threads] f64 column store fits in LDS
  int threads = KEY_THREADS;
  while ((size_t)nlay * threads * sizeof(double) > 160 * 1024 && threads > 64) threads /= 2;
  const size_t lds = (size_t)nlay * threads * sizeof(double);
  ECCKD_REQUIRE(lds <= 160 * 1024, "ecckd_reorder_key_lw_dev: nlay = %d exceeds the supported maximum (320)", nlay);

  double* d_lev = nullptr;
  int* d_flag = nullptr;
  ECCKD_CHECK(upload_level_consts(ctx, nlay, h_pressure_hl, h_temperature_hl, &d_lev, &d_flag));
  const unsigned blocks = (unsigned)((nwav + threads - 1) / threads);
  if (ctx->profile) ECCKD_HIP_CHECK(hipEventRecord(ctx->pev0, ctx->stream));
  if (nlay == 54) {
    const unsigned fblocks = (unsigned)((nwav + KEY_THREADS - 1) / KEY_THREADS);
    if (od_type == ECCKD_F32)
      hipLaunchKernelGGL((k_reorder_key_lw_fast<54, float>), dim3(fblocks), dim3(KEY_THREADS), 0, ctx->stream, nwav,
                         od_stride, d_lev, d_wavenumber, d_d_wavenumber, (const float*)d_od, thr, d_key, d_col_od, d_flag);
    else
      hipLaunchKernelGGL((k_reorder_key_lw_fast<54, double>), dim3(fblocks), dim3(KEY_THREADS), 0, ctx->stream, nwav,
                         od_stride, d_lev, d_wavenumber, d_d_wavenumber, (const double*)d_od, thr, d_key, d_col_od, d_flag);
  } else if (od_type == ECCKD_F32) {
    ECCKD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_reorder_key_lw<float>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k_reorder_key_lw<float>, dim3(blocks), dim3(threads), lds, ctx->stream, nlay, nwav,
                       od_stride, d_lev, d_wavenumber, d_d_wavenumber, (const float*)d_od, thr, d_key,
                       d_col_od, d_flag);
  } else {
    ECCKD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_reorder_key_lw<double>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k_reorder_key_lw<double>, dim3(blocks), dim3(threads), lds, ctx->stream, nlay, nwav,
                       od_stride, d_lev, d_wavenumber, d_d_wavenumber, (const double*)d_od, thr, d_key,
                       d_col_od, d_flag);
  }
  ECCKD_HIP_CHECK(hipGetLastError());
  if (ctx->profile) ECCKD_HIP_CHECK(hipEventRecord(ctx->pev1, ctx->stream));
  ECCKD_CHECK(check_flag(ctx, d_flag, "ecckd_reorder_key_lw_dev"));
  if (ctx->profile) {
    float ms = 0.f;
    ECCKD_HIP_CHECK(hipEventElapsedTime(&ms, ctx->pev0, ctx->pev1));
    ctx->stat_key_lw.ms += ms;
    ctx->stat_key_lw.units += (double)nwav;
    ctx->stat_key_lw.calls += 1;
  }
  return ECCKD_OK;
}

int ecckd_reorder_key_sw_dev(ecckd_ctx* ctx, int nlay, size_t nwav, const double* h_pressure_hl,
                             const void* d_od, int od_type, size_t od_stride, double thr,
                             double* d_key, double* d_col_od) {
  ECCKD_REQUIRE(ctx, "ecckd_reorder_key_sw_dev: ctx is NULL");
  ECCKD_REQUIRE(nlay > 0, "ecckd_reorder_key_sw_dev: nlay must be positive (got %d)", nlay);
  ECCKD_REQUIRE(h_pressure_hl && d_od && d_key && d_col_od, "ecckd_reorder_key_sw_dev: NULL array argument");
  ECCKD_REQUIRE(od_type == ECCKD_F32 || od_type == ECCKD_F64, "ecckd_reorder_key_sw_dev: od_type must be 4 or 8");
  ECCKD_REQUIRE(od_stride >= nwav, "ecckd_reorder_key_sw_dev: od_stride (%zu) < nwav (%zu)", od_stride, nwav);
  ECCKD_REQUIRE(pressures_ok(nlay, h_pressure_hl), "ecckd_reorder_key_sw_dev: pressure_hl must be positive and increasing");
  if (nwav == 0) return ECCKD_OK;
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  double* d_lev = nullptr;
  int* d_flag = nullptr;
  ECCKD_CHECK(upload_level_consts(ctx, nlay, h_pressure_hl, nullptr, &d_lev, &d_flag));
  const unsigned blocks = (unsigned)((nwav + KEY_THREADS - 1) / KEY_THREADS);
  if (od_type == ECCKD_F32) {
    hipLaunchKernelGGL(k_reorder_key_sw<float>, dim3(blocks), dim3(KEY_THREADS), 0, ctx->stream, nlay, nwav,
                       od_stride, d_lev, (const float*)d_od, thr, d_key, d_col_od, d_flag);
  } else {
    hipLaunchKernelGGL(k_reorder_key_sw<double>, dim3(blocks), dim3(KEY_THREADS), 0, ctx->stream, nlay, nwav,
                       od_stride, d_lev, (const double*)d_od, thr, d_key, d_col_od, d_flag);
  }
  ECCKD_HIP_CHECK(hipGetLastError());
  return check_flag(ctx, d_flag, "ecckd_reorder_key_sw_dev");
}

}  // extern "C"
