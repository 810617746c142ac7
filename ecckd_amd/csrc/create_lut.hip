// create_lut.hip - K6/K7: the create_look_up_table hot path on gfx950.
//
// K6  ecckd_average_to_gpoints  average_optical_depth_to_g_point (reference
//       src/ecckd/average_optical_depth.cpp:22-197): for every g point and layer, the
//       Planck- or solar-weighted average of the line-by-line optical depth under one of nine
//       averaging methods, its min and max, and the conversion to molar absorption.
// K7  ecckd_gpoint_fraction     create_look_up_table.cpp:537-548
//     ecckd_planck_lut          create_look_up_table.cpp:581-591
//
// Design.  The reference does `find(g_point == ig)` and a gather per g point (and an
// O(ng*nint*nwav) triple loop for the fractions).  Here the wavenumbers are sorted by g point
// ONCE per g-point map with the stable radix sort of K3 (a stable sort keeps the wavenumbers
// ascending inside every g point), so each g point is a contiguous segment of a permutation
// and every quantity above is a segmented reduction: fixed-size chunks of a segment are reduced
// by one block each (fixed order), a second tiny kernel combines the chunk partials in order.
// The Planck weights are recomputed from (T, wavenumber) on the fly instead of being read from
// an (nlay, nwav) matrix: 8 B/point/layer of HBM traffic traded for one exp.
#include "common.hpp"
#include "fastmath.hpp"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr double kD = ECCKD_LW_DIFFUSIVITY;
constexpr int GA_THREADS = 256;
constexpr int GA_PPT = 8;                       // points per thread (halves the cross-lane reductions per point)
constexpr int GA_CHUNK = GA_THREADS * GA_PPT;   // sorted positions per block

__device__ constexpr double kPlanckH = 6.62606896e-34;
__device__ constexpr double kLightC = 2.99792458e8;
__device__ constexpr double kPi = 3.14159265358979323846;

// averaging-method codes of the ABI (ECCKD_AVG_*)
constexpr int M_LINEAR = 0, M_TRANS = 1, M_TRANS2 = 2, M_SQRT = 3, M_LOG = 4, M_TRANS3 = 6, M_TRANS10 = 7, M_HYBRID = 8;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}

struct Chunk { long long p0, p1; int g; int pad; };

__global__ void __launch_bounds__(256)
k_gmap_keys(size_t n, const int32_t* __restrict__ g_point, int ng, double* __restrict__ key, int* __restrict__ counts,
            int* __restrict__ flag) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int g = g_point[j];
  if (g >= ng) { atomicOr(flag, 1); key[j] = 1.0e300; return; }
  key[j] = (g >= 0) ? (double)g : 1.0e300;  // unassigned points sort last
  if (g >= 0) atomicAdd(&counts[g], 1);
}

__global__ void __launch_bounds__(256)
k_gmap_gather(size_t n, const int32_t* __restrict__ order, const double* __restrict__ wn, const double* __restrict__ dwn,
              double* __restrict__ wn_s, double* __restrict__ dwn_s) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t j = (size_t)order[i];
  wn_s[i] = wn[j];
  dwn_s[i] = dwn[j];
}

// flag |= 2 if the wavenumbers are not ascending inside some g-point segment
__global__ void __launch_bounds__(256)
k_gmap_check(size_t nassigned, const int32_t* __restrict__ order, const int32_t* __restrict__ g_point,
             const double* __restrict__ wn_s, int* __restrict__ flag) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i + 1 >= nassigned) return;
  if (g_point[order[i]] == g_point[order[i + 1]] && !(wn_s[i + 1] >= wn_s[i])) atomicOr(flag, 2);
}

// Several per-lane values reduced over the wave TOGETHER: at lane distance 32 a lane hands over one half of its values and
// combines what it receives with the half it keeps, at distance 16 a half of those, ... - NV + NV/2 + ... exchanges instead of
// six per value (every exchange is an LDS crossbar operation that all the waves of a CU queue for).  Lane l ends up with the
// wave's result of value bitreverse6(l).  A fixed tree: bitwise reproducible.
template <int USED, typename Op>
__device__ __forceinline__ void fold_level(double* v, bool upper, int mask, Op op) {
  constexpr int NEXT = (USED + 1) / 2;
#pragma unroll
  for (int i = 0; i < NEXT; ++i) {
    const double a = v[2 * i];
    const double b = (2 * i + 1 < USED) ? v[2 * i + 1] : a;      // (odd tail: the same value from both halves)
    const double send = upper ? a : b;
    const double keep = upper ? b : a;
    v[i] = op(keep, __shfl_xor(send, mask, 64));
  }
}
template <int NV, typename Op>
__device__ __forceinline__ void fold_wave(double (&v)[NV], int lane, Op op) {
  constexpr int U1 = (NV + 1) / 2, U2 = (U1 + 1) / 2, U3 = (U2 + 1) / 2, U4 = (U3 + 1) / 2, U5 = (U4 + 1) / 2;
  fold_level<NV>(v, (lane & 32) != 0, 32, op);
  fold_level<U1>(v, (lane & 16) != 0, 16, op);
  fold_level<U2>(v, (lane & 8) != 0, 8, op);
  fold_level<U3>(v, (lane & 4) != 0, 4, op);
  fold_level<U4>(v, (lane & 2) != 0, 2, op);
  fold_level<U5>(v, (lane & 1) != 0, 1, op);
}

// K6a.  partial[chunk][layer][6] = { num, den, den_nz, cnt_nz, min, max }.  Per layer and point: the Planck weight at the
// layer's temperature (planck_function.cpp:48-50) and the averaged quantity of the layer's method
// (average_optical_depth.cpp:43-133) - the same two exp per layer and point as K1, with the same < 1 ulp exp / division
// (fastmath.hpp) instead of the library's -; the six per-layer sums of a wave are folded together (above), the waves' results
// wait in LDS and are combined in wave order once, after the last layer: no barrier inside the layer loop.
// dynamic LDS: [nlay][6][GA_THREADS / 64]
template <typename OdT>
__global__ void __launch_bounds__(GA_THREADS)
k_gavg_partial(int nlay, size_t od_stride, const Chunk* __restrict__ chunks, const int32_t* __restrict__ order,
               const double* __restrict__ wn_s, const double* __restrict__ dwn_s, const OdT* __restrict__ od,
               const double* __restrict__ hk /*[nlay] (h/k)/T_fl, LW*/, const double* __restrict__ ssi /*SW, original order*/,
               const int* __restrict__ layer_method, double* __restrict__ partial) {
  extern __shared__ double s_part[];
  constexpr int NW = GA_THREADS / 64;
  const Chunk c = chunks[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  size_t jj[GA_PPT];
  double freq[GA_PPT], pref[GA_PPT];
  bool live[GA_PPT];
#pragma unroll
  for (int p = 0; p < GA_PPT; ++p) {
    const long long i = c.p0 + (long long)p * GA_THREADS + tid;
    live[p] = i <= c.p1;
    const size_t ii = live[p] ? (size_t)i : (size_t)c.p1;
    jj[p] = (size_t)order[ii];
    if (hk) {
      // planck_function.cpp:48-50
      const double inv_cm_2_Hz = 100.0 * kLightC;
      freq[p] = wn_s[ii] * inv_cm_2_Hz;
      pref[p] = (dwn_s[ii] * 2.0 * kPlanckH * inv_cm_2_Hz * kPi / (kLightC * kLightC)) * (freq[p] * freq[p] * freq[p]);
    } else {
      freq[p] = 0.0;
      pref[p] = ssi[jj[p]];
    }
  }
  const int kk = (int)(__brev((unsigned)lane) >> 26);           // which folded value this lane ends up with
  for (int l = 0; l < nlay; ++l) {
    const int lm = layer_method[l];
    // every optical depth of the layer first: the gathers are independent
    double o[GA_PPT];
#pragma unroll
    for (int p = 0; p < GA_PPT; ++p) o[p] = (double)od[(size_t)l * od_stride + jj[p]];
    double sums[4] = {0.0, 0.0, 0.0, 0.0};      // num, den, den_nz, cnt_nz
    double mm[2] = {INFINITY, INFINITY};        // min, -max
    const double hkl = hk ? hk[l] : 0.0;
    const double dscale = lm == M_TRANS ? kD : lm == M_TRANS2 ? 2.0 * kD : lm == M_TRANS3 ? 3.0 * kD : lm == M_TRANS10 ? 10.0 * kD : 0.0;
#pragma unroll
    for (int p = 0; p < GA_PPT; ++p) {
      if (!live[p]) continue;
      const double w = hk ? ecckd::div_fast(pref[p], ecckd::exp_fast(freq[p] * hkl) - 1.0) : pref[p];
      sums[1] += w;
      mm[0] = fmin(mm[0], o[p]);
      mm[1] = fmin(mm[1], -o[p]);
      if (dscale > 0.0) sums[0] += (1.0 - ecckd::exp_fast(-o[p] * dscale)) * w;
      else if (lm == M_LINEAR) sums[0] += o[p] * w;
      else if (lm == M_SQRT) sums[0] += sqrt(o[p]) * w;
      else if (o[p] > 0.0) { sums[0] += log(o[p]) * w; sums[2] += w; sums[3] += 1.0; }      // logarithmic
    }
    fold_wave<4>(sums, lane, [](double a, double b) { return a + b; });
    fold_wave<2>(mm, lane, [](double a, double b) { return fmin(a, b); });
    if (kk < 4) s_part[((size_t)l * 6 + kk) * NW + wave] = sums[0];
    if (kk < 2) s_part[((size_t)l * 6 + 4 + kk) * NW + wave] = kk == 0 ? mm[0] : -mm[0];
  }
  __syncthreads();
  for (int t = tid; t < nlay * 6; t += GA_THREADS) {
    const double* q = s_part + (size_t)t * NW;
    const int k = t % 6;
    double v = q[0];
    for (int w = 1; w < NW; ++w) v = k < 4 ? v + q[w] : (k == 4 ? fmin(v, q[w]) : fmax(v, q[w]));
    partial[(size_t)blockIdx.x * nlay * 6 + t] = v;
  }
}

// K6b.  grid (ng, nlay), block 64: the chunk partials of one g point and layer - lane j adds the chunks j, j + 64, ... of the g
// point in chunk order, the lanes' sums are folded in a fixed tree (a function of the chunk count alone) -, then fit, clamp,
// min/max repair and conversion to molar absorption (average_optical_depth.cpp:135-193).  (One block per g point walking its
// ~90 chunks layer by layer took 0.46 ms beside K6a's 1.8.)
__global__ void __launch_bounds__(64)
k_gavg_final(int nlay, int ng, const int* __restrict__ seg_chunk0 /*[ng+1]*/, const long long* __restrict__ seg_count /*[ng]*/,
             const int* __restrict__ layer_method, const double* __restrict__ partial, const double* __restrict__ dp /*[nlay]*/,
             double scale /* (g*0.001*M/vmr) or <= 0: plain optical depth */, double* __restrict__ out /*[3][nlay][ng]*/) {
  const int g = blockIdx.x, l = blockIdx.y, lane = threadIdx.x;
  const int c0 = seg_chunk0[g], c1 = seg_chunk0[g + 1];
  const double ntot = (double)seg_count[g];
  {
    double num = 0.0, den = 0.0, den_nz = 0.0, cnt = 0.0, mn = INFINITY, mx = -INFINITY;
    for (int c = c0 + lane; c < c1; c += 64) {
      const double* p = partial + ((size_t)c * nlay + l) * 6;
      num += p[0]; den += p[1]; den_nz += p[2]; cnt += p[3];
      mn = fmin(mn, p[4]); mx = fmax(mx, p[5]);
    }
    num = wave_sum(num); den = wave_sum(den); den_nz = wave_sum(den_nz); cnt = wave_sum(cnt);
    mn = wave_min(mn); mx = wave_max(mx);
    if (lane != 0) return;
    double fit = 0.0;
    if (c1 > c0) {
      const int lm = layer_method[l];
      switch (lm) {
        case M_LINEAR: fit = num / den; break;
        case M_TRANS: fit = fabs(-log(1.0 - fmin(0.9999999999999999, num / den)) / (kD * 1.0 * 1.0)); break;
        case M_TRANS2: fit = fabs(-log(1.0 - fmin(0.9999999999999999, num / den)) / (kD * 1.0 * 2.0)); break;
        case M_TRANS3: fit = fabs(-log(1.0 - fmin(0.9999999999999999, num / den)) / (kD * 1.0 * 3.0)); break;
        case M_TRANS10: fit = fabs(-log(1.0 - fmin(0.9999999999999999, num / den)) / (kD * 1.0 * 10.0)); break;
        case M_SQRT: { const double v = num / den; fit = v * v; break; }
        default:
          if (cnt == ntot) fit = exp(num / den);
          else if (cnt == 0.0) fit = 0.0;
          else fit = exp(num / den_nz) * (cnt / ntot);
      }
      // :151-165
      fit = fmax(mn, fmin(fit, mx));
      if (mn > fit) mn = fit;
      if (mn > 0.0 && mn >= mx) { mn *= 0.99; mx *= 1.01; }
    } else {
      mn = 0.0; mx = 0.0;  // empty g point: zeros (:136-141)
    }
    const size_t o = (size_t)l * ng + g;
    if (scale > 0.0) {
      // :170-184, same operation order: (const / vmr) * tau / dp
      out[o] = scale * fit / dp[l];
      out[(size_t)nlay * ng + o] = scale * mn / dp[l];
      out[2 * (size_t)nlay * ng + o] = scale * mx / dp[l];
    } else {
      out[o] = fit;
      out[(size_t)nlay * ng + o] = mn;
      out[2 * (size_t)nlay * ng + o] = mx;
    }
  }
}

// K7a.  Planck LUT partials: partial[chunk][nlut]
__global__ void __launch_bounds__(GA_THREADS)
k_planck_lut_partial(int nlut, const Chunk* __restrict__ chunks, const double* __restrict__ wn_s,
                     const double* __restrict__ dwn_s, const double* __restrict__ hk_lut, double* __restrict__ partial) {
  const Chunk c = chunks[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double freq[GA_PPT], pref[GA_PPT];
#pragma unroll
  for (int p = 0; p < GA_PPT; ++p) {
    const long long i = c.p0 + (long long)p * GA_THREADS + tid;
    const bool live = i <= c.p1;
    const size_t ii = live ? (size_t)i : (size_t)c.p1;
    const double inv_cm_2_Hz = 100.0 * kLightC;
    freq[p] = wn_s[ii] * inv_cm_2_Hz;
    pref[p] = live ? (dwn_s[ii] * 2.0 * kPlanckH * inv_cm_2_Hz * kPi / (kLightC * kLightC)) * (freq[p] * freq[p] * freq[p]) : 0.0;
  }
  // (1.66e9 Planck evaluations for 231 temperatures x 7.2e6 points: 2.2 ms, ~60 % of the fp64 issue rate; the < 1 ulp exp and
  // division of fastmath.hpp with the waves' sums parked in LDS - no barrier in the loop - measured 2.4 ms: left as it was)
  __shared__ double s_red[4];
  for (int it = 0; it < nlut; ++it) {
    const double h = hk_lut[it];
    double v = 0.0;
#pragma unroll
    for (int p = 0; p < GA_PPT; ++p) v += pref[p] / (exp(freq[p] * h) - 1.0);
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    if (tid == 0) partial[(size_t)blockIdx.x * nlut + it] = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
  }
}

// grid (ng, ceil(nlut / 4)), block 256: a wave per (g point, temperature) - lane j adds the chunks j, j + 64, ... in chunk order, the
// lanes' sums are folded in a fixed tree
__global__ void __launch_bounds__(256)
k_planck_lut_final(int nlut, int ng, const int* __restrict__ seg_chunk0, const double* __restrict__ partial,
                   double* __restrict__ out /*[nlut][ng]*/) {
  const int g = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int it = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (it >= nlut) return;
  double s = 0.0;
  for (int c = seg_chunk0[g] + lane; c < seg_chunk0[g + 1]; c += 64) s += partial[(size_t)c * nlut + it];
  s = wave_sum(s);
  if (lane == 0) out[(size_t)it * ng + g] = s;
}

// Row sums per g point (scale_lut.cpp:119-124): partial[chunk][nrows], gathered through the g-sorted order.
template <typename T>
__global__ void __launch_bounds__(GA_THREADS)
k_rowsum_partial(int nrows, size_t stride, const Chunk* __restrict__ chunks, const int32_t* __restrict__ order,
                 const T* __restrict__ rows, double* __restrict__ partial) {
  __shared__ double s_red[4];
  const Chunk c = chunks[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  size_t src[GA_PPT];
  bool live[GA_PPT];
#pragma unroll
  for (int p = 0; p < GA_PPT; ++p) {
    const long long i = c.p0 + (long long)p * GA_THREADS + tid;
    live[p] = i <= c.p1;
    src[p] = (size_t)order[live[p] ? i : c.p1];
  }
  for (int r = 0; r < nrows; ++r) {
    double v = 0.0;
#pragma unroll
    for (int p = 0; p < GA_PPT; ++p) v += live[p] ? (double)rows[(size_t)r * stride + src[p]] : 0.0;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    if (tid == 0) partial[(size_t)blockIdx.x * nrows + r] = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
  }
}

// Erythemal weight per g point (lbl_fluxes.cpp:198-230): partial[chunk][2] = { sum ery * planck, sum planck }
__global__ void __launch_bounds__(GA_THREADS)
k_erythemal_partial(const Chunk* __restrict__ chunks, const double* __restrict__ wn_s, const double* __restrict__ dwn_s,
                    double* __restrict__ partial) {
  __shared__ double s_red[2][4];
  const Chunk c = chunks[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double num = 0.0, den = 0.0;
#pragma unroll
  for (int p = 0; p < GA_PPT; ++p) {
    const long long i = c.p0 + (long long)p * GA_THREADS + tid;
    if (i > c.p1) continue;
    const double wn = wn_s[i];
    const double wavelength_nm = 1.0e7 / wn;
    double ery = 0.0;
    if (wavelength_nm > 250.0 && wavelength_nm <= 298.0) ery = 1.0;
    if (wavelength_nm > 298.0 && wavelength_nm <= 328.0) ery = pow(10.0, 0.094 * (298.0 - wavelength_nm));
    if (wavelength_nm > 328.0 && wavelength_nm <= 400.0) ery = pow(10.0, 0.015 * (140.0 - wavelength_nm));
    ery = sqrt(ery);
    // planck_function(5777 K), planck_function.cpp:22-54
    const double inv_cm_2_Hz = 100.0 * kLightC;
    const double freq = wn * inv_cm_2_Hz;
    const double pref = (dwn_s[i] * 2.0 * kPlanckH * inv_cm_2_Hz * kPi / (kLightC * kLightC)) * (freq * freq * freq);
    const double pl = pref / (exp((6.62606896e-34 / 1.3806504e-23) * (freq / 5777.0)) - 1.0);
    num += ery * pl;
    den += pl;
  }
  num = wave_sum(num);
  den = wave_sum(den);
  if (lane == 0) { s_red[0][wave] = num; s_red[1][wave] = den; }
  __syncthreads();
  if (tid == 0) {
    partial[(size_t)blockIdx.x * 2 + 0] = ((s_red[0][0] + s_red[0][1]) + s_red[0][2]) + s_red[0][3];
    partial[(size_t)blockIdx.x * 2 + 1] = ((s_red[1][0] + s_red[1][1]) + s_red[1][2]) + s_red[1][3];
  }
}

// K7b.  gpoint_fraction: the spectral width of every g point inside every interval (wavenumber1, wavenumber2], and its total
// width.  Inside its segment of the g-sorted order a g point's wavenumbers ascend, so an interval is a run of positions found by
// two binary searches; grid (nint, ng), block 64 (an interval holds n / (nint ng) points of a g point on average: a few hundred).
__global__ void __launch_bounds__(64)
k_gpoint_width(int nint, const long long* __restrict__ seg_begin /*[ng+1]*/, const double* __restrict__ wn_s,
               const double* __restrict__ dwn_s, const double* __restrict__ w1, const double* __restrict__ w2,
               double* __restrict__ width /*[ng][nint+1]*/) {
  const int iw = blockIdx.x, g = blockIdx.y;
  const long long b = seg_begin[g], e = seg_begin[g + 1];
  // wavenumbers ascend inside the segment: first position with wn > w1, first with wn > w2
  const double a1 = w1[iw], a2 = w2[iw];
  long long l = b, r = e;
  while (l < r) { long long m = (l + r) >> 1; if (wn_s[m] > a1) r = m; else l = m + 1; }
  const long long lo = l;
  r = e;
  while (l < r) { long long m = (l + r) >> 1; if (wn_s[m] > a2) r = m; else l = m + 1; }
  const long long hi = l;
  // four loads in flight per lane
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  long long i = lo + threadIdx.x;
  for (; i + 192 < hi; i += 256) { s0 += dwn_s[i]; s1 += dwn_s[i + 64]; s2 += dwn_s[i + 128]; s3 += dwn_s[i + 192]; }
  for (; i < hi; i += 64) s0 += dwn_s[i];
  const double s = wave_sum((s0 + s1) + (s2 + s3));
  if (threadIdx.x == 0) width[(size_t)g * (nint + 1) + iw] = s;
}

// The TOTAL width of a g point is a sum over its whole segment (n / ng points: 2e5 at 7.2e6 points and 38 g points), which one
// 64-thread block per g point used to walk with one load in flight per lane - 7.3 ms, 20 GB/s.  Now: the chunks of the segment
// (the same 2048-position chunks K6 works on) are summed by a block each, the chunk sums of a g point added in chunk order.
__global__ void __launch_bounds__(GA_THREADS)
k_width_chunk_sums(const Chunk* __restrict__ chunks, const double* __restrict__ dwn_s, double* __restrict__ partial) {
  __shared__ double s_red[4];
  const Chunk c = chunks[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double v[GA_PPT];
#pragma unroll
  for (int p = 0; p < GA_PPT; ++p) {
    const long long i = c.p0 + (long long)p * GA_THREADS + tid;
    v[p] = i <= c.p1 ? dwn_s[i] : 0.0;
  }
  double s = 0.0;
#pragma unroll
  for (int p = 0; p < GA_PPT; ++p) s += v[p];
  s = wave_sum(s);
  if (lane == 0) s_red[wave] = s;
  __syncthreads();
  if (tid == 0) partial[blockIdx.x] = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
}

// grid ng, block 64: lane j adds the chunk sums j, j + 64, ... of the g point in chunk order, the lanes' sums are folded in a fixed tree
__global__ void __launch_bounds__(64)
k_width_totals(int ng, int nint, const int* __restrict__ seg_chunk0, const double* __restrict__ partial,
               double* __restrict__ width /*[ng][nint+1]*/) {
  const int g = blockIdx.x, lane = threadIdx.x;
  double s = 0.0;
  for (int c = seg_chunk0[g] + lane; c < seg_chunk0[g + 1]; c += 64) s += partial[c];
  s = wave_sum(s);
  if (lane == 0) width[(size_t)g * (nint + 1) + nint] = s;
}

}  // namespace

struct ecckd_gmap {
  ecckd_ctx* ctx = nullptr;
  size_t n = 0;          // wavenumbers
  size_t nassigned = 0;  // with g >= 0
  int ng = 0;
  int32_t* order = nullptr;   // [n] original index of each sorted position
  double* wn_s = nullptr;     // [n]
  double* dwn_s = nullptr;    // [n]
  std::vector<long long> seg_begin;  // [ng+1] in sorted positions
  std::vector<Chunk> chunks;
  std::vector<int> seg_chunk0;       // [ng+1]
  Chunk* d_chunks = nullptr;
  int* d_seg_chunk0 = nullptr;
  long long* d_seg_count = nullptr;
  long long* d_seg_begin = nullptr;
  void* work = nullptr;
  size_t work_bytes = 0;
};

namespace {
void gmap_free(ecckd_gmap* m) {
  if (!m) return;
  if (m->ctx) (void)hipStreamSynchronize(m->ctx->stream);
  void* ptrs[] = {m->order, m->wn_s, m->dwn_s, m->d_chunks, m->d_seg_chunk0, m->d_seg_count, m->d_seg_begin, m->work};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete m;
}
int gmap_work(ecckd_gmap* m, size_t bytes) {
  if (bytes <= m->work_bytes) return ECCKD_OK;
  if (m->work) { ECCKD_HIP_CHECK(hipStreamSynchronize(m->ctx->stream)); ECCKD_HIP_CHECK(hipFree(m->work)); m->work = nullptr; }
  ECCKD_HIP_CHECK(hipMalloc(&m->work, bytes));
  m->work_bytes = bytes;
  return ECCKD_OK;
}
}  // namespace

extern "C" {

int ecckd_gmap_create(ecckd_ctx* ctx, size_t nwav, const int32_t* d_g_point, int ng, const double* d_wavenumber,
                      const double* d_d_wavenumber, ecckd_gmap** out) {
  ECCKD_REQUIRE(ctx && out && d_g_point && d_wavenumber && d_d_wavenumber, "ecckd_gmap_create: NULL argument");
  *out = nullptr;
  ECCKD_REQUIRE(nwav > 0 && nwav < (size_t)0x7fffffff && ng > 0, "ecckd_gmap_create: bad sizes (nwav=%zu, ng=%d)", nwav, ng);
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  ecckd_gmap* m = new ecckd_gmap();
  m->ctx = ctx; m->n = nwav; m->ng = ng;
  double* d_key = nullptr; int* d_counts = nullptr; int32_t* d_rank = nullptr;
  auto fail_free = [&](int rc) { if (d_key) (void)hipFree(d_key); if (d_counts) (void)hipFree(d_counts); if (d_rank) (void)hipFree(d_rank); gmap_free(m); return rc; };
#define TRYH(e) do { hipError_t _e = (e); if (_e != hipSuccess) return fail_free(ecckd::fail(ECCKD_UNEXPECTED_EXCEPTION, "%s: %s", #e, hipGetErrorString(_e))); } while (0)
  TRYH(hipMalloc((void**)&d_key, nwav * sizeof(double)));
  TRYH(hipMalloc((void**)&d_counts, (size_t)(ng + 1) * sizeof(int)));
  TRYH(hipMalloc((void**)&d_rank, nwav * sizeof(int32_t)));
  TRYH(hipMalloc((void**)&m->order, nwav * sizeof(int32_t)));
  TRYH(hipMalloc((void**)&m->wn_s, nwav * sizeof(double)));
  TRYH(hipMalloc((void**)&m->dwn_s, nwav * sizeof(double)));
  TRYH(hipMemsetAsync(d_counts, 0, (size_t)(ng + 1) * sizeof(int), ctx->stream));
  const unsigned eb = (unsigned)((nwav + 255) / 256);
  hipLaunchKernelGGL(k_gmap_keys, dim3(eb), dim3(256), 0, ctx->stream, nwav, d_g_point, ng, d_key, d_counts, d_counts + ng);
  const int64_t b0 = 0, b1 = (int64_t)nwav - 1;
  int rc = ecckd_stable_argsort_bands_dev(ctx, nwav, d_key, 1, &b0, &b1, d_rank, m->order);
  if (rc) return fail_free(rc);
  hipLaunchKernelGGL(k_gmap_gather, dim3(eb), dim3(256), 0, ctx->stream, nwav, m->order, d_wavenumber, d_d_wavenumber,
                     m->wn_s, m->dwn_s);
  std::vector<int> counts(ng + 1);
  TRYH(hipMemcpyAsync(counts.data(), d_counts, (size_t)(ng + 1) * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  TRYH(hipStreamSynchronize(ctx->stream));
  if (counts[ng] & 1) return fail_free(ecckd::fail(ECCKD_PARAMETER_ERROR, "ecckd_gmap_create: g_point contains values >= ng (%d)", ng));
  m->seg_begin.assign(ng + 1, 0);
  for (int g = 0; g < ng; ++g) m->seg_begin[g + 1] = m->seg_begin[g] + counts[g];
  m->nassigned = (size_t)m->seg_begin[ng];
  // wavenumbers must ascend inside each g point (they do when the input grid ascends: the sort is stable)
  TRYH(hipMemsetAsync(d_counts + ng, 0, sizeof(int), ctx->stream));
  if (m->nassigned > 1)
    hipLaunchKernelGGL(k_gmap_check, dim3((unsigned)((m->nassigned + 255) / 256)), dim3(256), 0, ctx->stream, m->nassigned,
                       m->order, d_g_point, m->wn_s, d_counts + ng);
  int flag = 0;
  TRYH(hipMemcpyAsync(&flag, d_counts + ng, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  TRYH(hipStreamSynchronize(ctx->stream));
  if (flag) return fail_free(ecckd::fail(ECCKD_PARAMETER_ERROR, "ecckd_gmap_create: wavenumber must be ascending"));
  // chunks
  m->seg_chunk0.assign(ng + 1, 0);
  std::vector<long long> seg_count(ng);
  for (int g = 0; g < ng; ++g) {
    m->seg_chunk0[g] = (int)m->chunks.size();
    seg_count[g] = counts[g];
    for (long long p = m->seg_begin[g]; p < m->seg_begin[g + 1]; p += GA_CHUNK) {
      Chunk c;
      c.p0 = p;
      c.p1 = std::min<long long>(p + GA_CHUNK, m->seg_begin[g + 1]) - 1;
      c.g = g; c.pad = 0;
      m->chunks.push_back(c);
    }
  }
  m->seg_chunk0[ng] = (int)m->chunks.size();
  TRYH(hipMalloc((void**)&m->d_chunks, std::max<size_t>(m->chunks.size(), 1) * sizeof(Chunk)));
  TRYH(hipMalloc((void**)&m->d_seg_chunk0, (size_t)(ng + 1) * sizeof(int)));
  TRYH(hipMalloc((void**)&m->d_seg_count, (size_t)ng * sizeof(long long)));
  TRYH(hipMalloc((void**)&m->d_seg_begin, (size_t)(ng + 1) * sizeof(long long)));
  TRYH(hipMemcpyAsync(m->d_chunks, m->chunks.data(), m->chunks.size() * sizeof(Chunk), hipMemcpyHostToDevice, ctx->stream));
  TRYH(hipMemcpyAsync(m->d_seg_chunk0, m->seg_chunk0.data(), (size_t)(ng + 1) * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  TRYH(hipMemcpyAsync(m->d_seg_count, seg_count.data(), (size_t)ng * sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
  TRYH(hipMemcpyAsync(m->d_seg_begin, m->seg_begin.data(), (size_t)(ng + 1) * sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
  TRYH(hipStreamSynchronize(ctx->stream));
#undef TRYH
  (void)hipFree(d_key); (void)hipFree(d_counts); (void)hipFree(d_rank);
  *out = m;
  return ECCKD_OK;
}

int ecckd_gmap_destroy(ecckd_gmap* m) {
  gmap_free(m);
  return ECCKD_OK;
}

// number of wavenumbers in each g point (create_look_up_table.cpp:111-118 detects empty ones)
int ecckd_gmap_counts(ecckd_gmap* m, int64_t* h_counts) {
  ECCKD_REQUIRE(m && h_counts, "ecckd_gmap_counts: NULL argument");
  for (int g = 0; g < m->ng; ++g) h_counts[g] = m->seg_begin[g + 1] - m->seg_begin[g];
  return ECCKD_OK;
}

int ecckd_average_to_gpoints(ecckd_gmap* m, int nlay, const double* h_pressure_hl, const double* h_temperature_fl,
                             const double* d_ssi, const void* d_od, int od_type, size_t od_stride,
                             int averaging_method, double reference_surface_vmr, double* h_molar_abs,
                             double* h_min_molar_abs, double* h_max_molar_abs) {
  ECCKD_REQUIRE(m && h_pressure_hl && d_od && h_molar_abs && nlay > 0, "ecckd_average_to_gpoints: bad argument");
  ECCKD_REQUIRE((h_temperature_fl != nullptr) != (d_ssi != nullptr),
                "ecckd_average_to_gpoints: give either temperature_fl (longwave) or ssi (shortwave)");
  ECCKD_REQUIRE(od_type == ECCKD_F32 || od_type == ECCKD_F64, "ecckd_average_to_gpoints: od_type must be 4 or 8");
  ECCKD_REQUIRE(od_stride >= m->n, "ecckd_average_to_gpoints: od_stride < nwav");
  const bool known = averaging_method == M_LINEAR || averaging_method == M_TRANS || averaging_method == M_TRANS2 ||
                     averaging_method == M_SQRT || averaging_method == M_LOG || averaging_method == M_TRANS3 ||
                     averaging_method == M_TRANS10 || averaging_method == M_HYBRID;
  // average_optical_depth.cpp:128-131
  ECCKD_REQUIRE(known, "averaging_method %d not understood", averaging_method);
  ecckd_ctx* ctx = m->ctx;
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  const int ng = m->ng;
  const size_t nchunk = m->chunks.size();
  // host-side per-layer tables
  std::vector<double> hk(nlay, 0.0), dp(nlay);
  std::vector<int> lm(nlay, averaging_method);
  for (int l = 0; l < nlay; ++l) {
    dp[l] = h_pressure_hl[l + 1] - h_pressure_hl[l];
    if (h_temperature_fl) hk[l] = (6.62606896e-34 / 1.3806504e-23) / h_temperature_fl[l];
    if (averaging_method == M_HYBRID) {
      // :101-126: logarithmic where pressure_fl > 100 hPa, transmission-3 above
      const double pfl = 0.5 * (h_pressure_hl[l] + h_pressure_hl[l + 1]);
      lm[l] = pfl > 100.0e2 ? M_LOG : M_TRANS3;
    }
  }
  const size_t part_bytes = ecckd_align_up(std::max<size_t>(nchunk, 1) * nlay * 6 * sizeof(double), 256);
  const size_t out_bytes = ecckd_align_up((size_t)3 * nlay * ng * sizeof(double), 256);
  const size_t tab_bytes = ecckd_align_up((size_t)nlay * (2 * sizeof(double) + sizeof(int)), 256);
  ECCKD_CHECK(gmap_work(m, part_bytes + out_bytes + tab_bytes));
  char* w = (char*)m->work;
  double* d_part = (double*)w; w += part_bytes;
  double* d_out = (double*)w; w += out_bytes;
  double* d_hk = (double*)w; double* d_dp = d_hk + nlay; int* d_lm = (int*)(d_dp + nlay);
  ECCKD_HIP_CHECK(hipMemcpyAsync(d_hk, hk.data(), nlay * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ECCKD_HIP_CHECK(hipMemcpyAsync(d_dp, dp.data(), nlay * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ECCKD_HIP_CHECK(hipMemcpyAsync(d_lm, lm.data(), nlay * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // hk/dp/lm are stack-owned
  if (nchunk > 0) {
    if (od_type == ECCKD_F32)
      hipLaunchKernelGGL(k_gavg_partial<float>, dim3((unsigned)nchunk), dim3(GA_THREADS), (size_t)nlay * 6 * (GA_THREADS / 64) * sizeof(double), ctx->stream, nlay, od_stride,
                         m->d_chunks, m->order, m->wn_s, m->dwn_s, (const float*)d_od, h_temperature_fl ? d_hk : nullptr,
                         d_ssi, d_lm, d_part);
    else
      hipLaunchKernelGGL(k_gavg_partial<double>, dim3((unsigned)nchunk), dim3(GA_THREADS), (size_t)nlay * 6 * (GA_THREADS / 64) * sizeof(double), ctx->stream, nlay, od_stride,
                         m->d_chunks, m->order, m->wn_s, m->dwn_s, (const double*)d_od, h_temperature_fl ? d_hk : nullptr,
                         d_ssi, d_lm, d_part);
  }
  // :170-173 (ACCEL_GRAVITY * 0.001 * MOLAR_MASS_DRY_AIR) / reference_surface_vmr
  const double scale = reference_surface_vmr > 0.0 ? (ECCKD_ACCEL_GRAVITY * 0.001 * 28.970) / reference_surface_vmr : -1.0;
  hipLaunchKernelGGL(k_gavg_final, dim3(ng, nlay), dim3(64), 0, ctx->stream, nlay, ng, m->d_seg_chunk0, m->d_seg_count, d_lm,
                     d_part, d_dp, scale, d_out);
  ECCKD_HIP_CHECK(hipGetLastError());
  const size_t one = (size_t)nlay * ng * sizeof(double);
  ECCKD_HIP_CHECK(hipMemcpyAsync(h_molar_abs, d_out, one, hipMemcpyDeviceToHost, ctx->stream));
  if (h_min_molar_abs) ECCKD_HIP_CHECK(hipMemcpyAsync(h_min_molar_abs, d_out + (size_t)nlay * ng, one, hipMemcpyDeviceToHost, ctx->stream));
  if (h_max_molar_abs) ECCKD_HIP_CHECK(hipMemcpyAsync(h_max_molar_abs, d_out + 2 * (size_t)nlay * ng, one, hipMemcpyDeviceToHost, ctx->stream));
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return ECCKD_OK;
}

int ecckd_gpoint_fraction(ecckd_gmap* m, int nint, const double* h_wavenumber1, const double* h_wavenumber2,
                          double* h_gpoint_fraction) {
  ECCKD_REQUIRE(m && nint > 0 && h_wavenumber1 && h_wavenumber2 && h_gpoint_fraction, "ecckd_gpoint_fraction: bad argument");
  ecckd_ctx* ctx = m->ctx;
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  const int ng = m->ng;
  const size_t nchunk = m->chunks.size();
  const size_t wb = ecckd_align_up((size_t)ng * (nint + 1) * sizeof(double), 256);
  const size_t gb = ecckd_align_up((size_t)nint * sizeof(double), 256);
  const size_t cb = ecckd_align_up(std::max<size_t>(nchunk, 1) * sizeof(double), 256);
  ECCKD_CHECK(gmap_work(m, wb + 2 * gb + cb));
  double* d_width = (double*)m->work;
  double* d_w1 = (double*)((char*)m->work + wb);
  double* d_w2 = (double*)((char*)m->work + wb + gb);
  double* d_csum = (double*)((char*)m->work + wb + 2 * gb);
  ECCKD_CHECK(ecckd_h2d(ctx, d_w1, h_wavenumber1, (size_t)nint * sizeof(double)));
  ECCKD_CHECK(ecckd_h2d(ctx, d_w2, h_wavenumber2, (size_t)nint * sizeof(double)));
  hipLaunchKernelGGL(k_gpoint_width, dim3(nint, ng), dim3(64), 0, ctx->stream, nint, m->d_seg_begin, m->wn_s, m->dwn_s,
                     d_w1, d_w2, d_width);
  if (nchunk > 0)
    hipLaunchKernelGGL(k_width_chunk_sums, dim3((unsigned)nchunk), dim3(GA_THREADS), 0, ctx->stream, m->d_chunks, m->dwn_s, d_csum);
  hipLaunchKernelGGL(k_width_totals, dim3(ng), dim3(64), 0, ctx->stream, ng, nint, m->d_seg_chunk0, d_csum, d_width);
  ECCKD_HIP_CHECK(hipGetLastError());
  std::vector<double> width((size_t)ng * (nint + 1));
  ECCKD_CHECK(ecckd_d2h(ctx, width.data(), d_width, width.size() * sizeof(double)));
  for (int g = 0; g < ng; ++g)
    for (int iw = 0; iw < nint; ++iw)  // :542-546
      h_gpoint_fraction[(size_t)g * nint + iw] = width[(size_t)g * (nint + 1) + iw] / width[(size_t)g * (nint + 1) + nint];
  return ECCKD_OK;
}

int ecckd_planck_lut(ecckd_gmap* m, int nlut, const double* h_temperature_lut, double* h_planck_lut) {
  ECCKD_REQUIRE(m && nlut > 0 && h_temperature_lut && h_planck_lut, "ecckd_planck_lut: bad argument");
  ecckd_ctx* ctx = m->ctx;
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  const int ng = m->ng;
  const size_t nchunk = m->chunks.size();
  std::vector<double> hk(nlut);
  for (int i = 0; i < nlut; ++i) {
    ECCKD_REQUIRE(h_temperature_lut[i] > 0.0, "ecckd_planck_lut: temperatures must be positive");
    hk[i] = (6.62606896e-34 / 1.3806504e-23) / h_temperature_lut[i];
  }
  const size_t part_bytes = ecckd_align_up(std::max<size_t>(nchunk, 1) * nlut * sizeof(double), 256);
  const size_t out_bytes = ecckd_align_up((size_t)nlut * ng * sizeof(double), 256);
  const size_t hk_bytes = ecckd_align_up((size_t)nlut * sizeof(double), 256);
  ECCKD_CHECK(gmap_work(m, part_bytes + out_bytes + hk_bytes));
  double* d_part = (double*)m->work;
  double* d_out = (double*)((char*)m->work + part_bytes);
  double* d_hk = (double*)((char*)m->work + part_bytes + out_bytes);
  ECCKD_CHECK(ecckd_h2d(ctx, d_hk, hk.data(), (size_t)nlut * sizeof(double)));
  if (nchunk > 0)
    hipLaunchKernelGGL(k_planck_lut_partial, dim3((unsigned)nchunk), dim3(GA_THREADS), 0, ctx->stream, nlut, m->d_chunks,
                       m->wn_s, m->dwn_s, d_hk, d_part);
  hipLaunchKernelGGL(k_planck_lut_final, dim3(ng, (nlut + 3) / 4), dim3(256), 0, ctx->stream, nlut, ng, m->d_seg_chunk0, d_part, d_out);
  ECCKD_HIP_CHECK(hipGetLastError());
  return ecckd_d2h(ctx, h_planck_lut, d_out, (size_t)nlut * ng * sizeof(double));
}


int ecckd_gmap_sum_rows(ecckd_gmap* m, int nrows, const void* d_rows, int rows_type, size_t row_stride, double* h_sums) {
  ECCKD_REQUIRE(m && nrows > 0 && d_rows && h_sums, "ecckd_gmap_sum_rows: bad argument");
  ECCKD_REQUIRE(rows_type == ECCKD_F32 || rows_type == ECCKD_F64, "ecckd_gmap_sum_rows: rows_type must be ECCKD_F32 or ECCKD_F64");
  ECCKD_REQUIRE(row_stride >= m->n, "ecckd_gmap_sum_rows: row stride %zu shorter than the spectrum (%zu)", row_stride, m->n);
  ecckd_ctx* ctx = m->ctx;
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  const int ng = m->ng;
  const size_t nchunk = m->chunks.size();
  const size_t part_bytes = ecckd_align_up(std::max<size_t>(nchunk, 1) * nrows * sizeof(double), 256);
  const size_t out_bytes = ecckd_align_up((size_t)nrows * ng * sizeof(double), 256);
  ECCKD_CHECK(gmap_work(m, part_bytes + out_bytes));
  double* d_part = (double*)m->work;
  double* d_out = (double*)((char*)m->work + part_bytes);
  if (nchunk > 0) {
    if (rows_type == ECCKD_F32)
      hipLaunchKernelGGL(k_rowsum_partial<float>, dim3((unsigned)nchunk), dim3(GA_THREADS), 0, ctx->stream, nrows, row_stride,
                         m->d_chunks, m->order, (const float*)d_rows, d_part);
    else
      hipLaunchKernelGGL(k_rowsum_partial<double>, dim3((unsigned)nchunk), dim3(GA_THREADS), 0, ctx->stream, nrows, row_stride,
                         m->d_chunks, m->order, (const double*)d_rows, d_part);
  }
  hipLaunchKernelGGL(k_planck_lut_final, dim3(ng, (nrows + 3) / 4), dim3(256), 0, ctx->stream, nrows, ng, m->d_seg_chunk0, d_part, d_out);
  ECCKD_HIP_CHECK(hipGetLastError());
  return ecckd_d2h(ctx, h_sums, d_out, (size_t)nrows * ng * sizeof(double));
}


int ecckd_gmap_erythemal_spectrum(ecckd_gmap* m, double* h_erythemal) {
  ECCKD_REQUIRE(m && h_erythemal, "ecckd_gmap_erythemal_spectrum: NULL argument");
  ecckd_ctx* ctx = m->ctx;
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  const int ng = m->ng;
  const size_t nchunk = m->chunks.size();
  const size_t part_bytes = ecckd_align_up(std::max<size_t>(nchunk, 1) * 2 * sizeof(double), 256);
  const size_t out_bytes = ecckd_align_up((size_t)2 * ng * sizeof(double), 256);
  ECCKD_CHECK(gmap_work(m, part_bytes + out_bytes));
  double* d_part = (double*)m->work;
  double* d_out = (double*)((char*)m->work + part_bytes);
  if (nchunk > 0)
    hipLaunchKernelGGL(k_erythemal_partial, dim3((unsigned)nchunk), dim3(GA_THREADS), 0, ctx->stream, m->d_chunks, m->wn_s,
                       m->dwn_s, d_part);
  hipLaunchKernelGGL(k_planck_lut_final, dim3(ng, 1), dim3(256), 0, ctx->stream, 2, ng, m->d_seg_chunk0, d_part, d_out);
  ECCKD_HIP_CHECK(hipGetLastError());
  std::vector<double> nd(2 * (size_t)ng);
  ECCKD_CHECK(ecckd_d2h(ctx, nd.data(), d_out, nd.size() * sizeof(double)));
  for (int g = 0; g < ng; ++g) h_erythemal[g] = nd[g] / nd[(size_t)ng + g];
  return ECCKD_OK;
}

}  // extern "C"
