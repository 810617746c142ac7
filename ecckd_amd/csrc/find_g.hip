// find_g.hip - K4/K5: the g-point search of find_g_points on gfx950.
//
// K4  ecckd_gas_create_lw    gas preparation, reference src/ecckd/find_g_points.cpp:872-1150:
//       gather-reorder the columns by rank, Planck function, LW radiative
//       transfer of background+target, heating rate, surface/TOA flux rows,
//       averaging metric.  One sweep; results stay resident in HBM in SORTED
//       order, (level, wavenumber) row-major.
// K5  ecckd_calc_error_batch the interval error of CkdEquipartition::calc_error
//       (find_g_points.cpp:291-405) = fit_optical_depth_lw (:54-106) +
//       calc_cost_function_lw (calc_cost_function_lw.cpp:24-110) +
//       radiative_transfer_lw_bb (radiative_transfer_lw.cpp:87-142), for a whole
//       BATCH of intervals per call (the reference evaluates them one by one
//       from OpenMP threads, equipartition.h:98-116).
//
// Design (MI355X): everything the fit and the "true" side of the cost need is a
// plain SUM over the interval of per-point rows that do not depend on the
// interval (metric*planck, planck, hr, flux rows).  K4 therefore also reduces
// those rows over fixed 256-point tiles once; K5a sums tiles + the two ragged
// ends, so the only pass that streams the band per evaluation is the grey-
// optical-depth radiative transfer K5c: it reads planck (nlay+1) and background
// optical depth (nlay) rows = (2*nlay+1)*8 B per point, coalesced 512-B wave
// loads, with per-half-level flux sums reduced wave -> block -> interval in a
// fixed order (bitwise reproducible; no float atomics).
#include "common.hpp"
#include "partition_search.hpp"
#include "fastmath.hpp"

#include <atomic>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <immintrin.h>

namespace {

constexpr int TILE = 256;          // points per tile sum
// wave priorities of the K5c phases (s_setprio): measured choices, see DESIGN.md
#ifndef ECCKD_PRIO_LOAD
#define ECCKD_PRIO_LOAD 3
#endif
#ifndef ECCKD_PRIO_SWEEP1
#define ECCKD_PRIO_SWEEP1 1
#endif
#ifndef ECCKD_PRIO_SWEEP2
#define ECCKD_PRIO_SWEEP2 0
#endif
typedef float float_x2 __attribute__((ext_vector_type(2)));
typedef float_x2 float_x2_store;
constexpr int RT_THREADS = 256;    // K5c block
constexpr int PREP_THREADS = 256;  // K4 block

__device__ constexpr double kPlanckH = 6.62606896e-34;
__device__ constexpr double kLightC = 2.99792458e8;
__device__ constexpr double kPi = 3.14159265358979323846;
__device__ constexpr double kD = ECCKD_LW_DIFFUSIVITY;

// ---------------------------------------------------------------------------
// deterministic reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}

// sum over a 256-thread block; result valid in thread 0.  s4 = 4 doubles of LDS.
__device__ __forceinline__ double block_sum_256(double v, double* s4) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s4[wave] = v;
  __syncthreads();
  return ((s4[0] + s4[1]) + s4[2]) + s4[3];
}

// two (or three) block sums behind ONE pair of barriers; every sum is formed exactly as block_sum_256 forms it
__device__ __forceinline__ void block_sum_256_x3(double& a, double& b, double& c, double (*s4)[4]) {
  a = wave_sum(a);
  b = wave_sum(b);
  c = wave_sum(c);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { s4[0][wave] = a; s4[1][wave] = b; s4[2][wave] = c; }
  __syncthreads();
  a = ((s4[0][0] + s4[0][1]) + s4[0][2]) + s4[0][3];
  b = ((s4[1][0] + s4[1][1]) + s4[1][2]) + s4[1][3];
  c = ((s4[2][0] + s4[2][1]) + s4[2][2]) + s4[2][3];
}

}  // namespace

#include "gas.hpp"

namespace {

// ---------------------------------------------------------------------------
// K4 helpers
__global__ void __launch_bounds__(256)
k_invert_rank(size_t n, const int32_t* __restrict__ rank, int32_t* __restrict__ ireorder, int* __restrict__ err) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  int32_t r = rank[j];
  if (r < 0 || (size_t)r >= n) { atomicOr(err, 2); return; }
  ireorder[r] = (int32_t)j;  // find_g_points.cpp:779-780
}

__global__ void __launch_bounds__(256)
k_check_perm(size_t n, const int32_t* __restrict__ ireorder, int* __restrict__ err) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && ireorder[i] < 0) atomicOr(err, 2);
}

// The Planck matrix of a gas's ordering alone: planck_hl[lev][i] for the wavenumber whose rank is i, evaluated exactly as
// the preparation kernels do (same expression, same exp / division), so that a process which does not search the FIRST
// gas can still hold the matrix the reference keeps from it for all later gases (find_g_points.cpp:529, :970-984).
__global__ void __launch_bounds__(256)
k_planck_sorted(int nhl, size_t n, const int32_t* __restrict__ ireorder, const double* __restrict__ hk,
                const double* __restrict__ wn, const double* __restrict__ dwn, double* __restrict__ planck_hl) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t j = (size_t)ireorder[i];
  const double w = wn[j], dw = dwn[j];
  const double inv_cm_2_Hz = 100.0 * kLightC;
  const double freq = w * inv_cm_2_Hz;
  const double pref = (dw * 2.0 * kPlanckH * inv_cm_2_Hz * kPi / (kLightC * kLightC)) * (freq * freq * freq);
  for (int lev = 0; lev < nhl; ++lev)
    planck_hl[(size_t)lev * n + i] = ecckd::div_fast(pref, ecckd::exp_fast(freq * hk[lev]) - 1.0);
}

__device__ __forceinline__ double metric_of(int method, double od) {
  // find_g_points.cpp:1119-1150
  switch (method) {
    case ECCKD_AVG_TRANSMISSION: return 1.0 - exp(-od * kD);
    case ECCKD_AVG_TRANSMISSION_2: return 1.0 - exp(-od * kD * 2.0);
    case ECCKD_AVG_SQUARE_ROOT: return sqrt(od);
    default: return od;  // linear, logarithmic, total-transmission
  }
}

// K4 (LW).  One thread per SORTED wavenumber i; source column j = ireorder[i].
// Dynamic LDS: double[nlay][blockDim.x] for the down-sweep flux increments.
template <typename BgT, typename OdT>
__global__ void __launch_bounds__(PREP_THREADS)
k_gas_prep_lw(int nlay, size_t n, size_t src_stride, int method,
              const int32_t* __restrict__ ireorder, const double* __restrict__ hk /*[nhl] (h/k)/T*/,
              const double* __restrict__ conv /*[nlay]*/,
              const double* __restrict__ wn, const double* __restrict__ dwn,
              const BgT* __restrict__ bg_src, const OdT* __restrict__ od_src,
              const double* __restrict__ planck_reuse,
              double* __restrict__ wn_sorted, double* __restrict__ dwn_sorted,
              double* __restrict__ planck_hl, double* __restrict__ bg_od, double* __restrict__ w1,
              double* __restrict__ w2, double* __restrict__ cnt,
              double* __restrict__ hr, double* __restrict__ fds, double* __restrict__ fut) {
  extern __shared__ double s_col[];
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int tid = threadIdx.x, bs = blockDim.x;
  const size_t j = (size_t)ireorder[i];
  const bool is_log = (method == ECCKD_AVG_LOGARITHMIC);

  // planck_function.cpp:48-50 on the reordered wavenumbers (find_g_points.cpp:970-979)
  const double w = wn[j], dw = dwn[j];
  wn_sorted[i] = w;
  dwn_sorted[i] = dw;
  const double inv_cm_2_Hz = 100.0 * kLightC;
  const double freq = w * inv_cm_2_Hz;
  const double pref = (dw * 2.0 * kPlanckH * inv_cm_2_Hz * kPi / (kLightC * kLightC)) * (freq * freq * freq);
  auto planck_at = [&](int lev) -> double {
    if (planck_reuse) return planck_reuse[(size_t)lev * n + i];
    return ecckd::div_fast(pref, ecckd::exp_fast(freq * hk[lev]) - 1.0);
  };

  double b_prev = planck_at(0);
  if (!planck_reuse) planck_hl[i] = b_prev;
  double dn = 0.0;
  // inputs are fetched eight layers at a time, ahead of that chunk's stores: a load issued after a store waits for the
  // store to complete on this hardware, so one load per layer would cost one store latency per layer
  constexpr int CH = 8;
  for (int l0 = 0; l0 < nlay; l0 += CH) {
  double bgv[CH], odv[CH], plv[CH];
#pragma unroll
  for (int q = 0; q < CH; ++q) {
    const int lq = l0 + q < nlay ? l0 + q : nlay - 1;
    bgv[q] = bg_src ? (double)bg_src[(size_t)lq * src_stride + j] : 0.0;
    odv[q] = (double)od_src[(size_t)lq * src_stride + j];
    plv[q] = planck_reuse ? planck_reuse[(size_t)(lq + 1) * n + i] : 0.0;
  }
#pragma unroll
  for (int q = 0; q < CH; ++q) {
    const int l = l0 + q;
    if (l >= nlay) break;
    const double bg = bgv[q];
    const double od = odv[q];
    const double tau = bg + od;  // find_g_points.cpp:993
    // radiative_transfer_lw.cpp:41-43
    const double eps = 1.0 - ecckd::exp_fast(-kD * tau);
    const double fac = (eps > 1.0e-5) ? 1.0 - ecckd::div_fast(eps * (1.0 / kD), tau) : 0.5 * eps;
    const double b_next = planck_reuse ? plv[q] : planck_at(l + 1);
    const double dn_next = dn * (1.0 - eps) + b_prev * (eps - fac) + b_next * fac;
    s_col[l * bs + tid] = dn_next - dn;
    const size_t o = (size_t)l * n + i;
    bg_od[o] = bg;
    if (!planck_reuse) planck_hl[(size_t)(l + 1) * n + i] = b_next;
    const double m = metric_of(method, od);
    if (!is_log) {
      // numerator row of fit_optical_depth_lw (find_g_points.cpp:61-62): metric * planck_hl(l+1)
      w1[o] = m * b_next;
    } else {
      // find_g_points.cpp:81-98: log(metric) weighted by planck_hl(l+1) over planck_hl(l), metric > 0 only
      const bool pos = m > 0.0;
      w1[o] = pos ? log(m) * b_next : 0.0;
      w2[o] = pos ? b_prev : 0.0;
      cnt[o] = pos ? 1.0 : 0.0;
    }
    dn = dn_next;
    b_prev = b_next;
  }
  }
  fds[i] = dn;  // flux_dn(end,__), find_g_points.cpp:1045
  // surface: emissivity 1, surf_planck = planck at temperature_hl(end) (:976-978, :987-988)
  double up = b_prev * 1.0 + (1.0 - 1.0) * dn;
  for (int l0 = nlay - 1; l0 >= 0; l0 -= CH) {
  double bgv[CH], odv[CH], plv[CH];
#pragma unroll
  for (int q = 0; q < CH; ++q) {
    const int lq = l0 - q >= 0 ? l0 - q : 0;
    bgv[q] = bg_src ? (double)bg_src[(size_t)lq * src_stride + j] : 0.0;
    odv[q] = (double)od_src[(size_t)lq * src_stride + j];
    plv[q] = planck_reuse ? planck_reuse[(size_t)lq * n + i] : 0.0;
  }
#pragma unroll
  for (int q = 0; q < CH; ++q) {
    const int l = l0 - q;
    if (l < 0) break;
    const double tau = bgv[q] + odv[q];
    const double eps = 1.0 - ecckd::exp_fast(-kD * tau);
    const double fac = (eps > 1.0e-5) ? 1.0 - ecckd::div_fast(eps * (1.0 / kD), tau) : 0.5 * eps;
    const double b_l = planck_reuse ? plv[q] : planck_at(l);
    const double up_l = up * (1.0 - eps) + b_prev * (eps - fac) + b_l * fac;
    // heating_rate.h:47-48
    hr[(size_t)l * n + i] = conv[l] * (s_col[l * bs + tid] - up + up_l);
    up = up_l;
    b_prev = b_l;
  }
  }
  fut[i] = up;  // flux_up(0,__), find_g_points.cpp:1052
}


// K4 fast path, step 1: transposing scatter.  src is (layer, wavenumber) in ORIGINAL order; the column of wavenumber j goes to
// its SORTED position rank[j], as PARTS runs of NLAY / PARTS layers: dst[PARTS][npad][NLAY / PARTS] (npad: n rounded up to whole
// waves).  Reads are coalesced rows through an LDS tile, each point's values leave as one wave store of PARTS contiguous
// segments.  The block of one run for 64 consecutive ranks is contiguous: that is what the preparation kernels copy into LDS
// (instead of 2*NLAY random 4-byte gathers per point, or NLAY-element columns read at a stride of NLAY*4 bytes).
template <int NLAY, typename SrcT, int PARTS>
__global__ void __launch_bounds__(256)
k_scatter_column_halves(size_t n, size_t npad, size_t src_stride, const int32_t* __restrict__ rank, const SrcT* __restrict__ src,
                        SrcT* __restrict__ dst) {
  constexpr int HP = NLAY / PARTS;
  __shared__ SrcT s_tile[NLAY][65];
  const size_t j0 = (size_t)blockIdx.x * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int l = wave; l < NLAY; l += 4) {
    const size_t j = j0 + lane;
    s_tile[l][lane] = (j < n) ? src[(size_t)l * src_stride + j] : (SrcT)0;
  }
  // the ranks of the tile's 64 points: one coalesced load per wave, handed out lane by lane below (a load per point inside the
  // loop made every store wait for its own memory latency)
  const int my_rank = j0 + lane < n ? rank[j0 + lane] : -1;
  __syncthreads();
  const size_t part = (size_t)(lane / HP) * npad;
  const int within = lane % HP;
  for (int p = wave; p < 64; p += 4) {
    const int r = __shfl(my_rank, p, 64);
    if (r < 0) break;
    if (lane < NLAY) dst[(part + (size_t)r) * HP + within] = s_tile[lane][p];
  }
}

// K4 mirror path (same idea as K5c's, see k_rt_lw_bb_mirror): a pair of waves shares 64 points, the even
// wave prepares the upper NLAY/2 layers top-down (first sweep = downwelling from the top of the
// atmosphere), the odd wave the lower NLAY/2 layers bottom-up (first sweep = upwelling from the surface,
// which emits its Planck function); the fluxes at the interface are exchanged through LDS and each wave
// sweeps back through its half in the other direction, forming the heating rate on the way
// (heating_rate_single, heating_rate.h:55-72, with the reference's order of operations).  Half the
// per-lane state of the one-wave version -> two waves per SIMD with half as long dependency chains.
// REUSE: the Planck matrix of an earlier gas is read instead of evaluated (find_g_points.cpp:970-984 keeps the first
// gas's matrix for all later gases); its rows are fetched with the other inputs and planck_hl is not written.
template <int NLAY, typename BgT, typename OdT, bool REUSE>
__global__ void __launch_bounds__(PREP_THREADS, 2)
k_gas_prep_lw_mirror(size_t n, int method, const int32_t* __restrict__ ireorder, const double* __restrict__ hk,
                     const double* __restrict__ conv, const double* __restrict__ wn, const double* __restrict__ dwn,
                     const BgT* __restrict__ bg_col /* [n][NLAY] sorted, or NULL */,
                     const OdT* __restrict__ od_col /* [n][NLAY] sorted */, const double* __restrict__ planck_reuse,
                     double* __restrict__ wn_sorted, double* __restrict__ dwn_sorted, double* __restrict__ planck_hl,
                     double* __restrict__ bg_od, double* __restrict__ w1, double* __restrict__ hr,
                     double* __restrict__ fds, double* __restrict__ fut,
                     double* __restrict__ wave_part /* [3*NLAY+2][nw]: sums over this wave's 64 points of every summable row */,
                     size_t nw,
                     float_x2_store* __restrict__ bg_pair /* FLOAT background only, or NULL: the rows as FLOAT pairs (layers 2p, 2p+1
                                                             of a point side by side) INSTEAD of the DOUBLE rows bg_od */) {
  static_assert(NLAY % 2 == 0 && PREP_THREADS == 256, "two wave pairs per block, equal halves");
  constexpr int H = NLAY / 2;
  constexpr bool STAGED = sizeof(OdT) == 4;   // the columns arrive in two runs of H layers, see below (always, on this path)
  __shared__ double s_x[4][64];
  // Row sums over the wave's 64 points, formed while the values are still in registers (they used to be re-read from
  // HBM by k_tile_sums: 1 312 B per point).  Sixteen rows at a time go through a wave-private transposed LDS tile: lane
  // (rr, qq) adds the 16 points qq*16.. of row rr in index order, the four quarters are combined by two xor-shuffles.
  constexpr int ROWW = 65;
  constexpr int NPUSH = 3 * H + 1;                 // w1 and Planck row per layer, heating rate per layer, boundary flux
  __shared__ double s_sum[4][8 * ROWW];
  // the first sweep's fluxes at the wave's H + 1 levels, [level][lane]: 56 registers that the kernel does not have (it spilled 120
  // bytes per lane with them); before the sweep the block holds the staged inputs
  __shared__ __attribute__((aligned(16))) double s_f1[4][(H + 1) * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = wave & 1, pair = wave >> 1;
  const size_t i = (size_t)blockIdx.x * 128 + (size_t)pair * 64 + lane;
  const bool live = i < n;
  const size_t ii = live ? i : n - 1;
  const size_t j = (size_t)ireorder[ii];
  const double w = wn[j], dw = dwn[j];
  if (live && half == 0) { wn_sorted[i] = w; dwn_sorted[i] = dw; }
  const double inv_cm_2_Hz = 100.0 * kLightC;
  const double freq = w * inv_cm_2_Hz;
  const double pref = (dw * 2.0 * kPlanckH * inv_cm_2_Hz * kPi / (kLightC * kLightC)) * (freq * freq * freq);
  const BgT* bgc = bg_col && !STAGED ? bg_col + ii * NLAY : nullptr;
  const OdT* odc = od_col + (STAGED ? 0 : ii * NLAY);
  const size_t wid = (size_t)blockIdx.x * 2 + pair;          // index of this pair's 64 points among the 64-point groups
  double* sum_tile = s_sum[wave];
  const int rr = lane & 7, qq = lane >> 3;
  int slot = 0;
  // rows of the table ecckd_gas_create_lw builds: A = w1 (0..NLAY), B = planck_hl(l+1) (NLAY..), H = hr (2 NLAY..), then the two boundary rows
  auto row_of_slot = [&](int sl) -> int {
    if (sl < 2 * H) { const int l = sl >> 1; const int L = half ? NLAY - 1 - l : l; return ((sl & 1) ? NLAY : 0) + L; }
    if (sl < 3 * H) { const int l = H - 1 - (sl - 2 * H); const int L = half ? NLAY - 1 - l : l; return 2 * NLAY + L; }
    return 3 * NLAY + (half ? 0 : 1);
  };
  auto push = [&](double v) {
    sum_tile[(slot & 7) * ROWW + lane] = live ? v : 0.0;
    if ((slot & 7) == 7 || slot == NPUSH - 1) {
      const int base = slot & ~7, count = slot - base + 1;
      __builtin_amdgcn_wave_barrier();
      double sum = 0.0;
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += sum_tile[rr * ROWW + qq * 8 + j];
      sum += __shfl_xor(sum, 8, 64);
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      if (wave_part && qq == 0 && rr < count && wid < nw) wave_part[(size_t)row_of_slot(base + rr) * nw + wid] = sum;
      __builtin_amdgcn_wave_barrier();
    }
    ++slot;
  };
  // polynomial coefficients and the other literals pinned in SGPRs (fastmath.hpp): without this a third of
  // the kernel's instructions were s_mov / v_mov pairs rebuilding 64-bit constants next to their uses
  const ecckd::ExpConsts ek = ecckd::exp_consts();
  const double neg_d = ecckd::sgpr_pin(-kD), inv_d = ecckd::sgpr_pin(1.0 / kD), thin = ecckd::sgpr_pin(1.0e-5);
  double pl_in[REUSE ? H + 1 : 1];
  if (REUSE) {
#pragma unroll
    for (int l = 0; l <= H; ++l) pl_in[REUSE ? l : 0] = planck_reuse[(size_t)(half ? NLAY - l : l) * n + ii];
  }
  // local level l is level l (even wave) or NLAY - l (odd wave)
  auto planck = [&](int level) {
    if (REUSE) return pl_in[REUSE ? (half ? NLAY - level : level) : 0];
    return ecckd::div_fast(pref, ecckd::exp_fast_s(freq * hk[level], ek) - 1.0);
  };
  // local layer l is layer l (even wave) or NLAY-1-l (odd wave); its near level is where the first sweep enters
  double ee[H], s2[H];
  double* const f1 = s_f1[wave] + lane;     // f1[level * 64]
  // All inputs of this wave's half column are fetched BEFORE the first store: on this hardware the counter a load
  // waits on also counts the stores issued before it, so a load inside the layer loop would wait for the previous
  // layer's four row stores to complete and the kernel would run at one store latency per layer.
  OdT od_in[H];
  BgT bg_in[H];
  __builtin_amdgcn_s_setprio(3);
  if (STAGED) {
    // Both inputs come as [2][npad][H] (k_scatter_column_halves), this wave's 64 x H block of each is one contiguous
    // piece.  It is copied into LDS (the row-sum tile, idle until the first push) with full-width loads and each lane takes
    // its column from there: read straight from memory a column costs 64 cache lines per load instruction.
    float* st = reinterpret_cast<float*>(s_f1[wave]);
    constexpr int NV = 16 * H;                                // float4s in the block
    static_assert(NV * 16 * 2 <= (H + 1) * 64 * 8, "a DOUBLE block fits too");
    const size_t widc = wid < nw ? wid : nw - 1;              // a pair past the end repeats the last point, as ii does
    const int lc = live ? lane : (int)((n - 1) - widc * 64);
    const size_t blk = ((size_t)half * (nw * 64) + widc * 64) * H;
    auto fill = [&](const float* col) {
#pragma unroll
      for (int t = 0; t < (NV + 63) / 64; ++t) {
        const int at = t * 64 + lane;
        if (at < NV) reinterpret_cast<float4*>(st)[at] = reinterpret_cast<const float4*>(col + blk)[at];
      }
      __builtin_amdgcn_wave_barrier();
    };
    if (sizeof(BgT) == 4) {
      // FLOAT (or no) background: both blocks side by side in the buffer, their loads out together - one memory round trip.
      // No background: the loads still happen, from the target's block (a branch round them would put a full wait behind each).
      static_assert(2 * NV * 16 <= (H + 1) * 64 * 8, "both FLOAT blocks fit side by side");
      const float4* so = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(od_col) + blk);
      const float4* sb = bg_col ? reinterpret_cast<const float4*>(reinterpret_cast<const float*>(bg_col) + blk) : so;
#pragma unroll
      for (int t = 0; t < (NV + 63) / 64; ++t) {
        const int at = t * 64 + lane;
        if (at < NV) {
          const float4 vo = so[at], vb = sb[at];
          reinterpret_cast<float4*>(st)[at] = vo;
          reinterpret_cast<float4*>(st)[NV + at] = vb;
        }
      }
      __builtin_amdgcn_wave_barrier();
      const bool has_bg = bg_col != nullptr;
#pragma unroll
      for (int l = 0; l < H; ++l) {
        const int kk = lc * H + (half ? H - 1 - l : l);
        od_in[l] = (OdT)st[kk];
        bg_in[l] = has_bg ? (BgT)st[64 * H + kk] : (BgT)0;
      }
      __builtin_amdgcn_wave_barrier();
    } else {
    fill(reinterpret_cast<const float*>(od_col));
#pragma unroll
    for (int l = 0; l < H; ++l) od_in[l] = (OdT)st[lc * H + (half ? H - 1 - l : l)];
    __builtin_amdgcn_wave_barrier();
    if (bg_col) {
      // DOUBLE (merged) background: the same with 16-byte pairs of doubles
      const double* colb = reinterpret_cast<const double*>(bg_col) + blk;
      double* std_ = s_f1[wave];
#pragma unroll
      for (int t = 0; t < (2 * NV + 63) / 64; ++t) {
        const int at = t * 64 + lane;
        if (at < 2 * NV) reinterpret_cast<double2*>(std_)[at] = reinterpret_cast<const double2*>(colb)[at];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int l = 0; l < H; ++l) bg_in[l] = (BgT)std_[lc * H + (half ? H - 1 - l : l)];
      __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
      for (int l = 0; l < H; ++l) bg_in[l] = (BgT)0;
    }
    }
  } else {
#pragma unroll
    for (int l = 0; l < H; ++l) od_in[l] = odc[half ? NLAY - 1 - l : l];
    if (bg_col) {   // uniform: one block of loads
#pragma unroll
      for (int l = 0; l < H; ++l) bg_in[l] = bgc[half ? NLAY - 1 - l : l];
    } else {
#pragma unroll
      for (int l = 0; l < H; ++l) bg_in[l] = (BgT)0;
    }
  }
  __builtin_amdgcn_s_setprio(0);
  int lev_near = half ? NLAY : 0;
  double b_near = planck(lev_near);
  if (!REUSE && live) planck_hl[(size_t)lev_near * n + i] = b_near;
  double flux = half ? b_near : 0.0;           // surface: emissivity 1 (radiative_transfer_lw.cpp:52-53)
  f1[0] = flux;
#pragma unroll
  for (int l = 0; l < H; ++l) {
    const int L = half ? NLAY - 1 - l : l;
    const int lev_far = half ? L : L + 1;
    const double bg = (double)bg_in[l];
    const double od = (double)od_in[l];
    const double tau = bg + od;
    const double eps = 1.0 - ecckd::exp_fast_s(neg_d * tau, ek);
    const double fac = (eps > thin) ? 1.0 - ecckd::div_fast(eps * inv_d, tau) : 0.5 * eps;
    const double b_far = planck(lev_far);
    const double emf = eps - fac;
    const double next = flux * (1.0 - eps) + b_near * emf + b_far * fac;
    ee[l] = eps;
    s2[l] = b_far * emf + b_near * fac;
    f1[(l + 1) * 64] = next;
    double m;
    switch (method) {
      case ECCKD_AVG_TRANSMISSION: m = 1.0 - ecckd::exp_fast_s(-od * kD, ek); break;
      case ECCKD_AVG_TRANSMISSION_2: m = 1.0 - ecckd::exp_fast_s(-od * kD * 2.0, ek); break;
      case ECCKD_AVG_SQUARE_ROOT: m = sqrt(od); break;
      default: m = od;
    }
    if (live) {
      const size_t o = (size_t)L * n + i;
      // written once, read by later kernels: streaming stores
      if (sizeof(BgT) == 4 && bg_pair) {
        // the background as the sweep reads it (k_rt_lw_bb_mirror<.., true>): a pair is complete at every second layer of the
        // half column - the even wave has met its layers in the pair's order, the odd wave the other way round; the middle
        // pair of an odd half belongs to both waves, each stores its own float
        if (l & 1) {
          float_x2_store v;
          v.x = half ? (float)bg_in[l] : (float)bg_in[l - 1];
          v.y = half ? (float)bg_in[l - 1] : (float)bg_in[l];
          __builtin_nontemporal_store(v, &bg_pair[(size_t)(L / 2) * n + i]);
        } else if (l == H - 1) {
          ((float*)&bg_pair[(size_t)(L / 2) * n + i])[L & 1] = (float)bg_in[l];
        }
      } else {
        __builtin_nontemporal_store(bg, &bg_od[o]);
      }
      if (!REUSE) __builtin_nontemporal_store(b_far, &planck_hl[(size_t)lev_far * n + i]);   // level NLAY/2 is written by both waves with the same bits
      __builtin_nontemporal_store(m * (half ? b_near : b_far), &w1[o]);          // weight = Planck function at the base of the layer
    }
    push(m * (half ? b_near : b_far));
    push(half ? b_near : b_far);                    // planck_hl(L + 1), the denominator row of the fit
    flux = next;
    b_near = b_far;
  }
  s_x[wave][lane] = flux;
  __syncthreads();
  flux = s_x[wave ^ 1][lane];
#pragma unroll
  for (int l = H - 1; l >= 0; --l) {
    const int L = half ? NLAY - 1 - l : l;
    const double next = flux * (1.0 - ee[l]) + s2[l];
    // conv * (dn(L+1) - dn(L) - up(L+1) + up(L)), left to right
    const double f1a = f1[l * 64], f1b = f1[(l + 1) * 64];
    const double net = half ? (next - flux) - f1a + f1b        // second sweep is downwelling: flux = dn(L), next = dn(L+1)
                            : (f1b - f1a) - flux + next;       // second sweep is upwelling:   flux = up(L+1), next = up(L)
    const double hrv = conv[L] * net;
    if (live) __builtin_nontemporal_store(hrv, &hr[(size_t)L * n + i]);
    push(hrv);
    flux = next;
  }
  if (live) {
    if (half) fds[i] = flux; else fut[i] = flux;
  }
  push(flux);
}

// tile sums from the per-wave sums K4 leaves: TS[r][t] = ((W[4t] + W[4t+1]) + W[4t+2]) + W[4t+3]
__global__ void __launch_bounds__(256)
k_combine_wave_sums(int nrows, size_t nw, size_t ntiles, const double* __restrict__ wp, double* __restrict__ ts) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (t >= ntiles || r >= nrows) return;
  const double* w = wp + (size_t)r * nw + 4 * t;
  const size_t left = nw - 4 * t;
  double s = w[0];
  if (left > 1) s += w[1];
  if (left > 2) s += w[2];
  if (left > 3) s += w[3];
  ts[(size_t)r * ntiles + t] = s;
}

// tile sums of every row: TS[r][t] = sum_{i in tile t} rows[r][i]
__global__ void __launch_bounds__(TILE)
k_tile_sums(int nrows, size_t n, size_t ntiles, const double* const* __restrict__ rows,
            double* __restrict__ ts) {
  // four rows per round: four independent loads in flight, one barrier pair for the four block sums; each sum is the
  // same tree as block_sum_256 (wave shuffle tree, then ((w0 + w1) + w2) + w3)
  __shared__ double s4[4][4];
  const size_t t = blockIdx.x;
  const size_t i = t * TILE + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r0 = 0; r0 < nrows; r0 += 4) {
    double v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (r0 + q < nrows && i < n) ? rows[r0 + q][i] : 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = wave_sum(v[q]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) s4[q][wave] = v[q];
    }
    __syncthreads();
    if (threadIdx.x < 4 && r0 + (int)threadIdx.x < nrows) {
      const int q = threadIdx.x;
      ts[(size_t)(r0 + q) * ntiles + t] = ((s4[q][0] + s4[q][1]) + s4[q][2]) + s4[q][3];
    }
  }
}

// ---------------------------------------------------------------------------
// K5a: interval sums.  grid (nrows, nint); block 256.
// sums[k][r] = sum_{i=i1_k..i2_k} rows[r][i] = ragged head + whole tiles + ragged tail
struct Interval {
  long long i1, i2;      // inclusive global sorted indices
  long long chunk0;      // first RT chunk of this interval
  long long chunk_pts;   // points per RT chunk of THIS interval: a function of its length alone (interval_chunk_pts), so that the
                         // order in which its points are summed - and with it every bit of its error - does not depend on what else
                         // is evaluated in the same batch
  long long npoints;     // band length (for the logarithmic fit)
  double albedo;         // shortwave: surface albedo of the interval's band
};

// Batches of up to KARG_MAX intervals (nearly all of a search: next_bound_below / _above evaluate ONE interval per call,
// equipartition.cpp:638-805; calc_error_all a partition of a few dozen) hand their interval table to the first kernel of the
// train in its ARGUMENTS instead of a host-to-device copy in front of it (a blit kernel of ~18 us): one stream operation and
// one dependency gap less per batch.  The first kernel leaves the table in device memory for
// the kernels behind it.
constexpr int KARG_MAX = 64;     // 3 KB of the 4 KB argument segment
struct IntervalArgs { Interval iv[KARG_MAX]; };

// The argument table is read where it lies, in the kernel-argument segment, with scalar loads at a block-uniform offset:
// IntervalArgs MUST be the kernel's FIRST parameter (offset 0 of the segment).  Indexing the by-value struct itself makes the
// compiler copy it to scratch (392 bytes per lane: every wave launch then waits for a scratch slot and the kernel's time
// grows by ~2.3 us per interval of the batch).
__device__ __forceinline__ Interval interval_of(int use_ka, const Interval* __restrict__ iv, int k) {
  if (!use_ka) return iv[k];
  typedef const __attribute__((address_space(4))) long long* karg_ptr;
  karg_ptr q = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + (size_t)k * (sizeof(Interval) / sizeof(long long));
  static_assert(sizeof(Interval) == 6 * sizeof(long long), "Interval is six 8-byte fields");
  Interval me;
  me.i1 = q[0]; me.i2 = q[1]; me.chunk0 = q[2]; me.chunk_pts = q[3]; me.npoints = q[4];
  me.albedo = __longlong_as_double(q[5]);
  return me;
}

// this thread's share of sum_{i=i1..i2} row[i]: ragged head points, whole 256-point tiles up to the next 65 536-point
// super tile, whole super tiles, the tiles behind them, ragged tail points - at most one value of each kind per thread
// (the sums of tiles and super tiles were formed once per gas), so the sums of an interval cost a handful of loads
// however long it is.  trow / srow: the row's tile and super-tile sums.
constexpr int SUPER = 256;                     // tiles per super tile
__device__ __forceinline__ double interval_row_acc(const double* __restrict__ row, const double* __restrict__ trow,
                                                   const double* __restrict__ srow, long long i1, long long i2, int tid) {
  const long long t1 = (i1 + TILE - 1) / TILE;   // first whole tile
  const long long t2 = (i2 + 1) / TILE;          // one past the last whole tile
  double acc = 0.0;
  if (t1 >= t2) {
    // no whole tile inside: at most 2*TILE-2 raw points
    for (long long i = i1 + tid; i <= i2; i += 256) acc += row[i];
    return acc;
  }
  const long long head_end = t1 * TILE;  // exclusive
  if (i1 + tid < head_end) acc += row[i1 + tid];
  const long long s1 = (t1 + SUPER - 1) / SUPER;  // first whole super tile
  const long long s2 = t2 / SUPER;                // one past the last whole super tile
  if (s1 >= s2) {
    // no whole super tile inside: at most 2*SUPER-2 tiles
    for (long long t = t1 + tid; t < t2; t += 256) acc += trow[t];
  } else {
    if (t1 + tid < s1 * SUPER) acc += trow[t1 + tid];
    for (long long q = s1 + tid; q < s2; q += 256) acc += srow[q];
    if (s2 * SUPER + tid < t2) acc += trow[s2 * SUPER + tid];
  }
  const long long tail = t2 * TILE + tid;
  if (tail <= i2) acc += row[tail];
  return acc;
}

// super-tile sums of every row: SS[r][q] = sum of the row's tile sums q*SUPER .. q*SUPER+SUPER-1, grid (nsuper, nrows)
__global__ void __launch_bounds__(256)
k_super_sums(size_t ntiles, size_t nsuper, const double* __restrict__ ts, double* __restrict__ ss) {
  __shared__ double s4[4];
  const size_t q = blockIdx.x, r = blockIdx.y;
  const size_t t = q * SUPER + threadIdx.x;
  const double v = t < ntiles ? ts[r * ntiles + t] : 0.0;
  const double sum = block_sum_256(v, s4);
  if (threadIdx.x == 0) ss[r * nsuper + q] = sum;
}

__global__ void __launch_bounds__(256)
k_interval_sums(IntervalArgs ka, int nrows, size_t ntiles, size_t nsuper, const Interval* __restrict__ iv, int use_ka,
                Interval* __restrict__ iv_out, const double* const* __restrict__ rows, const double* __restrict__ ts,
                const double* __restrict__ ss, double* __restrict__ sums) {
  __shared__ double s4[4];
  const int r = blockIdx.x, k = blockIdx.y;
  const int tid = threadIdx.x;
  (void)ka;
  const Interval me = interval_of(use_ka, iv, k);
  if (use_ka && r == 0 && tid == 0) iv_out[k] = me;
  const double acc = interval_row_acc(rows[r], ts + (size_t)r * ntiles, ss + (size_t)r * nsuper, me.i1, me.i2, tid);
  const double s = block_sum_256(acc, s4);
  if (tid == 0) sums[(size_t)k * nrows + r] = s;
}

// fit_optical_depth_lw (find_g_points.cpp:54-106) for one layer from the interval's sums: a = sum of the weighted metric,
// b = sum of the weights, nnz = number of points with a positive metric (logarithmic method), ntot = points of the interval
__device__ __forceinline__ double fit_lw_layer(int method, double a, double b, double nnz, double ntot) {
#pragma clang fp contract(off)
  switch (method) {
    case ECCKD_AVG_LINEAR: return a / b;
    case ECCKD_AVG_TRANSMISSION: return fabs(-log(1.0 - fmin(0.9999999999999999, a / b)) / kD);             // :64-68
    case ECCKD_AVG_TRANSMISSION_2: return fabs(-log(1.0 - fmin(0.9999999999999999, a / b)) / (kD * 2.0));
    case ECCKD_AVG_SQUARE_ROOT: { const double v = a / b; return v * v; }
    case ECCKD_AVG_LOGARITHMIC:                                                                              // :79-99
      if (nnz == ntot) return exp(a / b);
      if (nnz == 0.0) return 0.0;
      return exp(a / b) * (nnz / ntot);
    default: return nan("");
  }
}

// The rows of a longwave gas by number, from their base pointers in the kernel arguments (the device table of row pointers
// costs a dependent load in front of every row's own loads).
struct LwRowBases {
  const double *w1, *w2, *cnt, *planck_hl, *hr, *fds, *fut;
  size_t n;
  int is_log;
  __device__ __forceinline__ const double* row(const RowMap& R, int nlay, int r) const {
    if (r < R.B) return w1 + (size_t)(r - R.A) * n;
    if (r < R.B + nlay) return is_log ? w2 + (size_t)(r - R.B) * n : planck_hl + (size_t)(r - R.B + 1) * n;
    if (is_log && r < R.H) return cnt + (size_t)(r - R.N) * n;
    if (r < R.FDS) return hr + (size_t)(r - R.H) * n;
    return r == R.FDS ? fds : fut;
  }
};

// K5a + K5b in one launch for the longwave: the fit of layer l needs only the sums of its own rows (A+l, B+l and, for the
// logarithmic method, N+l), so the block that owns layer l adds up those rows and finishes the fit itself; the remaining
// rows (heating rate, boundary fluxes) get one block each as before.  grid (nlay + rows from R.H on, nint), block 256.
__global__ void __launch_bounds__(256)
k_interval_sums_fit_lw(IntervalArgs ka, int nlay, int method, RowMap R, size_t ntiles, size_t nsuper, const Interval* __restrict__ iv,
                       int use_ka, Interval* __restrict__ iv_out, LwRowBases rows,
                       const double* __restrict__ ts, const double* __restrict__ ss, double* __restrict__ sums,
                       double* __restrict__ od_fit) {
  __shared__ double s4[4];
  const int bx = blockIdx.x, k = blockIdx.y, tid = threadIdx.x;
  (void)ka;
  const Interval me = interval_of(use_ka, iv, k);
  if (use_ka && bx == 0 && tid == 0) iv_out[k] = me;
  const long long i1 = me.i1, i2 = me.i2;
  double* out = sums + (size_t)k * R.total;
  if (bx >= nlay) {
    const int r = R.H + (bx - nlay);
    const double s = block_sum_256(interval_row_acc(rows.row(R, nlay, r), ts + (size_t)r * ntiles, ss + (size_t)r * nsuper, i1, i2, tid), s4);
    if (tid == 0) out[r] = s;
    return;
  }
  const int l = bx;
  const bool is_log = method == ECCKD_AVG_LOGARITHMIC;
  // the loads of the two (three) rows go out together: one memory round trip and one pair of barriers instead of two (three)
  __shared__ double s43[3][4];
  double a = interval_row_acc(rows.row(R, nlay, R.A + l), ts + (size_t)(R.A + l) * ntiles, ss + (size_t)(R.A + l) * nsuper, i1, i2, tid);
  double b = interval_row_acc(rows.row(R, nlay, R.B + l), ts + (size_t)(R.B + l) * ntiles, ss + (size_t)(R.B + l) * nsuper, i1, i2, tid);
  double nnz = 0.0;
  if (is_log) nnz = interval_row_acc(rows.row(R, nlay, R.N + l), ts + (size_t)(R.N + l) * ntiles, ss + (size_t)(R.N + l) * nsuper, i1, i2, tid);
  block_sum_256_x3(a, b, nnz, s43);
  if (tid == 0) {
    out[R.A + l] = a;
    out[R.B + l] = b;
    if (is_log) out[R.N + l] = nnz;
    const double fit = fit_lw_layer(method, a, b, nnz, (double)(i2 - i1 + 1));
    od_fit[(size_t)k * nlay + l] = fit;
    // the same values bottom-up behind the table: the sweep's odd waves read their layers at ascending addresses too
    od_fit[((size_t)gridDim.y + k) * nlay + (nlay - 1 - l)] = fit;
  }
}

// K5b: fitted grey optical depth per (interval, layer).  grid nint, block 128.
__global__ void __launch_bounds__(128)
k_fit_lw(int nlay, int method, RowMap R, const Interval* __restrict__ iv,
         const double* __restrict__ sums, double* __restrict__ od_fit) {
#pragma clang fp contract(off)
  const int k = blockIdx.x;
  const double* s = sums + (size_t)k * R.total;
  for (int l = threadIdx.x; l < nlay; l += blockDim.x) {
    const double a = s[R.A + l], b = s[R.B + l];
    double fit;
    switch (method) {
      case ECCKD_AVG_LINEAR:
        fit = a / b;
        break;
      case ECCKD_AVG_TRANSMISSION:  // find_g_points.cpp:64-68
        fit = fabs(-log(1.0 - fmin(0.9999999999999999, a / b)) / kD);
        break;
      case ECCKD_AVG_TRANSMISSION_2:
        fit = fabs(-log(1.0 - fmin(0.9999999999999999, a / b)) / (kD * 2.0));
        break;
      case ECCKD_AVG_SQUARE_ROOT: {
        const double v = a / b;
        fit = v * v;
        break;
      }
      case ECCKD_AVG_LOGARITHMIC: {  // find_g_points.cpp:79-99
        const double nnz = s[R.N + l];
        const double ntot = (double)(iv[k].i2 - iv[k].i1 + 1);
        if (nnz == ntot) fit = exp(a / b);
        else if (nnz == 0.0) fit = 0.0;
        else fit = exp(a / b) * (nnz / ntot);
        break;
      }
      default:
        fit = nan("");
    }
    od_fit[(size_t)k * nlay + l] = fit;
  }
}

// K5c: grey-optical-depth LW radiative transfer with broadband sums,
// radiative_transfer_lw_bb (radiative_transfer_lw.cpp:87-142) for every interval.
// One block = one chunk of `chunk_pts` consecutive points of ONE interval,
// processed as sub-tiles of 256 points (one point per thread).  Per half-level
// the 64 lanes' fluxes are reduced in the wave and lane 0 accumulates into
// s_acc[wave][...]; the four waves are combined in order at the end and the
// block writes its 2*(nlay+1) partial sums to `partial[chunk]`.
// The up sweep re-evaluates emissivity and factor exactly as the reference does
// (:128-137) from re-loaded rows (second touch is served by L2/Infinity Cache).
__global__ void __launch_bounds__(RT_THREADS)
k_rt_lw_bb(int nlay, size_t n, int nint, const Interval* __restrict__ iv,
           const double* __restrict__ planck_hl, const double* __restrict__ bg_od,
           const double* __restrict__ od_fit, double* __restrict__ partial) {
  extern __shared__ double s_mem[];  // [4][2*nhl] accumulators | [nlay] grey od
  const int nhl = nlay + 1;
  double* s_acc = s_mem;
  double* s_grey = s_mem + 4 * 2 * nhl;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // locate this block's interval: chunk0 is ascending
  const long long chunk = blockIdx.x;
  int lo = 0, hi = nint - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (iv[mid].chunk0 <= chunk) lo = mid; else hi = mid - 1;
  }
  const int k = lo;
  const long long chunk_pts = iv[k].chunk_pts;
  const long long c = chunk - iv[k].chunk0;
  const long long p0 = iv[k].i1 + c * chunk_pts;
  long long p1 = p0 + chunk_pts - 1;
  if (p1 > iv[k].i2) p1 = iv[k].i2;

  for (int t = tid; t < 4 * 2 * nhl; t += RT_THREADS) s_acc[t] = 0.0;
  for (int l = tid; l < nlay; l += RT_THREADS) s_grey[l] = od_fit[(size_t)k * nlay + l];
  __syncthreads();
  double* acc_dn = s_acc + wave * 2 * nhl;
  double* acc_up = acc_dn + nhl;

  constexpr double THRESHOLD_EMISSIVITY = 1.0e-5;
  for (long long base = p0; base <= p1; base += RT_THREADS) {
    const long long i = base + tid;
    const bool live = i <= p1;
    const size_t ii = live ? (size_t)i : (size_t)p1;
    // ---- down sweep (:109-124) ----
    double flux = 0.0;
    double b_prev = planck_hl[ii];
    for (int l = 0; l < nlay; ++l) {
      const double od = bg_od[(size_t)l * n + ii] + s_grey[l];
      const double b_next = planck_hl[(size_t)(l + 1) * n + ii];
      const double eps = 1.0 - exp(-kD * od);
      const double fac = fmax(1.0 - (1.0 / kD) * fmax(eps, THRESHOLD_EMISSIVITY) /
                                         fmax(od, THRESHOLD_EMISSIVITY / kD),
                              0.5 * THRESHOLD_EMISSIVITY);
      flux = flux * (1.0 - eps) + b_prev * (eps - fac) + b_next * fac;
      const double s = wave_sum(live ? flux : 0.0);
      if (lane == 0) acc_dn[l + 1] += s;
      b_prev = b_next;
    }
    // ---- surface (:126-128), emissivity 1, surf_planck = planck_hl(nlay) ----
    flux = b_prev * 1.0 + (1.0 - 1.0) * flux;
    {
      const double s = wave_sum(live ? flux : 0.0);
      if (lane == 0) acc_up[nlay] += s;
    }
    // ---- up sweep (:130-141) ----
    for (int l = nlay - 1; l >= 0; --l) {
      const double od = bg_od[(size_t)l * n + ii] + s_grey[l];
      const double b_l = planck_hl[(size_t)l * n + ii];
      const double eps = 1.0 - exp(-kD * od);
      const double fac = fmax(1.0 - (1.0 / kD) * fmax(eps, THRESHOLD_EMISSIVITY) /
                                         fmax(od, THRESHOLD_EMISSIVITY / kD),
                              0.5 * THRESHOLD_EMISSIVITY);
      flux = flux * (1.0 - eps) + b_prev * (eps - fac) + b_l * fac;
      const double s = wave_sum(live ? flux : 0.0);
      if (lane == 0) acc_up[l] += s;
      b_prev = b_l;
    }
  }
  __syncthreads();
  for (int t = tid; t < 2 * nhl; t += RT_THREADS) {
    partial[(size_t)chunk * 2 * nhl + t] =
        ((s_acc[t] + s_acc[2 * nhl + t]) + s_acc[4 * nhl + t]) + s_acc[6 * nhl + t];
  }
}


// ---------------------------------------------------------------------------
// fp64 helpers for the hot loops.
//
// exp(y) for y <= 0 (emissivity: eps = 1 - exp(-D*od)).  Cody-Waite reduction
// y = k ln2 + r, |r| <= ln2/2, degree-12 Taylor/Horner in FMA, scaled with
// ldexp.  Max error < 1 ulp on [-745, 0]; arguments below -745 flush to 0.
__device__ __forceinline__ double exp_nonpos(double y) {
  const double kf = __builtin_rint(y * 1.4426950408889634074);
  double r = __builtin_fma(kf, -6.93147180369123816490e-01, y);
  r = __builtin_fma(kf, -1.90821492927058770002e-10, r);
  double p = 2.08767569878680989792e-09;            // 1/12!
  p = __builtin_fma(p, r, 2.50521083854417187751e-08);  // 1/11!
  p = __builtin_fma(p, r, 2.75573192239858906526e-07);  // 1/10!
  p = __builtin_fma(p, r, 2.75573192239858906526e-06);  // 1/9!
  p = __builtin_fma(p, r, 2.48015873015873015873e-05);  // 1/8!
  p = __builtin_fma(p, r, 1.98412698412698412698e-04);  // 1/7!
  p = __builtin_fma(p, r, 1.38888888888888888889e-03);  // 1/6!
  p = __builtin_fma(p, r, 8.33333333333333333333e-03);  // 1/5!
  p = __builtin_fma(p, r, 4.16666666666666666667e-02);  // 1/4!
  p = __builtin_fma(p, r, 1.66666666666666666667e-01);  // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  const int k = (int)fmax(kf, -1100.0);
  return __builtin_amdgcn_ldexp(p, k);
}

// a / b for b well inside the normal range: v_rcp_f64 seed + 2 Newton steps + 1 residual
// correction (relative error < 1 ulp).
__device__ __forceinline__ double fast_div(double a, double b) {
  double y = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  double q = a * y;
  const double res = __builtin_fma(-b, q, a);
  return __builtin_fma(res, y, q);
}


// Emissivity and factor of TWO layers at once with the two dependency chains interleaved in
// the source: at 2 waves/SIMD the fp64 FMA latency is not covered by other waves, so the
// instruction-level parallelism has to be in the stream itself (radiative_transfer_lw.cpp:114-119).
__device__ __forceinline__ void eps_fac_pair(double od0, double od1, double& eps0, double& fac0,
                                             double& eps1, double& fac1) {
  constexpr double TE = 1.0e-5;
  const double y0 = -kD * od0, y1 = -kD * od1;
  const double k0 = __builtin_rint(y0 * 1.4426950408889634074), k1 = __builtin_rint(y1 * 1.4426950408889634074);
  double r0 = __builtin_fma(k0, -6.93147180369123816490e-01, y0), r1 = __builtin_fma(k1, -6.93147180369123816490e-01, y1);
  r0 = __builtin_fma(k0, -1.90821492927058770002e-10, r0); r1 = __builtin_fma(k1, -1.90821492927058770002e-10, r1);
  // reciprocal seeds issued early: independent of the exp chains
  const double d0 = fmax(od0, TE / kD), d1 = fmax(od1, TE / kD);
  double q0 = __builtin_amdgcn_rcp(d0), q1 = __builtin_amdgcn_rcp(d1);
  double p0 = 2.08767569878680989792e-09, p1 = 2.08767569878680989792e-09;
#define ECCKD_STEP(c) p0 = __builtin_fma(p0, r0, c); p1 = __builtin_fma(p1, r1, c);
  ECCKD_STEP(2.50521083854417187751e-08)
  double e0 = __builtin_fma(-d0, q0, 1.0), e1 = __builtin_fma(-d1, q1, 1.0);
  ECCKD_STEP(2.75573192239858906526e-07)
  q0 = __builtin_fma(q0, e0, q0); q1 = __builtin_fma(q1, e1, q1);
  ECCKD_STEP(2.75573192239858906526e-06)
  e0 = __builtin_fma(-d0, q0, 1.0); e1 = __builtin_fma(-d1, q1, 1.0);
  ECCKD_STEP(2.48015873015873015873e-05)
  q0 = __builtin_fma(q0, e0, q0); q1 = __builtin_fma(q1, e1, q1);
  ECCKD_STEP(1.98412698412698412698e-04)
  ECCKD_STEP(1.38888888888888888889e-03)
  ECCKD_STEP(8.33333333333333333333e-03)
  ECCKD_STEP(4.16666666666666666667e-02)
  ECCKD_STEP(1.66666666666666666667e-01)
  ECCKD_STEP(0.5)
  ECCKD_STEP(1.0)
  ECCKD_STEP(1.0)
#undef ECCKD_STEP
  const int i0 = (int)fmax(k0, -1100.0), i1 = (int)fmax(k1, -1100.0);
  eps0 = 1.0 - __builtin_amdgcn_ldexp(p0, i0);
  eps1 = 1.0 - __builtin_amdgcn_ldexp(p1, i1);
  // max(eps, TE) / max(od, TE/D) with one residual correction
  const double n0 = fmax(eps0, TE), n1 = fmax(eps1, TE);
  double t0 = n0 * q0, t1 = n1 * q1;
  const double s0 = __builtin_fma(-d0, t0, n0), s1 = __builtin_fma(-d1, t1, n1);
  t0 = __builtin_fma(s0, q0, t0); t1 = __builtin_fma(s1, q1, t1);
  fac0 = fmax(1.0 - (1.0 / kD) * t0, 0.5 * TE);
  fac1 = fmax(1.0 - (1.0 / kD) * t1, 0.5 * TE);
}

// Emissivity and factor of one layer (the tail of an odd-length half column).
__device__ __forceinline__ void eps_fac_one(double od, double& eps, double& fac) {
  constexpr double TE = 1.0e-5;
  eps = 1.0 - exp_nonpos(-kD * od);
  fac = fmax(1.0 - (1.0 / kD) * fast_div(fmax(eps, TE), fmax(od, TE / kD)), 0.5 * TE);
}

// The background optical depths of a longwave gas as FLOAT pairs: out[p][i] = (bg_od[2p][i], bg_od[2p+1][i]).  A value that
// is not a float (or is a subnormal float: the conversion back may flush it) raises `inexact`, and the gas keeps to its
// DOUBLE rows.
__global__ void __launch_bounds__(256)
k_pack_bg32(int npair, size_t n, const double* __restrict__ bg_od, float_x2_store* __restrict__ out, int* __restrict__ inexact) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  bool bad = false;
#pragma unroll 3
  for (int p = 0; p < npair; ++p) {
    const double x = __builtin_nontemporal_load(&bg_od[(size_t)(2 * p) * n + i]);
    const double y = __builtin_nontemporal_load(&bg_od[(size_t)(2 * p + 1) * n + i]);
    const float fx = (float)x, fy = (float)y;
    bad |= !((double)fx == x) || !((double)fy == y);
    bad |= (fx != 0.f && fabsf(fx) < 1.17549435e-38f) || (fy != 0.f && fabsf(fy) < 1.17549435e-38f);
    float_x2_store v;
    v.x = fx; v.y = fy;
    __builtin_nontemporal_store(v, &out[(size_t)p * n + i]);
  }
  if (bad) atomicOr(inexact, 1);
}

// ... and back: the DOUBLE rows of a gas that holds only the pairs, when something asks for them (ecckd_gas_view, the
// run-time-nlay sweep).
__global__ void __launch_bounds__(256)
k_unpack_bg32(int npair, size_t n, const float_x2_store* __restrict__ pairs, double* __restrict__ bg_od) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  for (int p = 0; p < npair; ++p) {
    const float_x2_store v = pairs[(size_t)p * n + i];
    bg_od[(size_t)(2 * p) * n + i] = (double)v.x;
    bg_od[(size_t)(2 * p + 1) * n + i] = (double)v.y;
  }
}

// K5c mirror path.  The two-stream equations are symmetric under turning the column upside down:
// the up sweep through layer l, up_l = up_{l+1} t + B_{l+1}(eps-fac) + B_l fac, is the down sweep
// dn_{l+1} = dn_l t + B_l(eps-fac) + B_{l+1} fac with the two Planck values exchanged
// (radiative_transfer_lw.cpp:121-123,:138-139).  So a PAIR of waves shares 64 spectral points: the even
// wave owns the upper NLAY/2 layers and loads them top-down, the odd wave owns the lower NLAY/2 layers
// and loads them bottom-up, and both run the SAME instruction stream:
//   first sweep  - even: downward from the top of the atmosphere (input 0),
//                  odd:  upward from the surface (input = surface Planck function, emissivity 1),
//                  keeping the layer transmittance and the source of the opposite direction;
//   exchange     - through LDS, one barrier per tile: each wave receives the true flux that enters its
//                  half from the other side;
//   second sweep - even: upward through its half, odd: downward through its half.
// Every flux is the exact sequential recurrence of the reference; nothing is recomputed.  Per-lane
// state is NLAY+1 doubles instead of 2*NLAY+1, which fits 3 waves per SIMD - the dependent
// v_fma_f64 chains need that (tools/fp64_latency.hip).  Level sums: as in the fast path, through a
// wave-private transposed LDS tile every 16 slots.
//
// BG32: the background optical depths come from the packed FLOAT rows (k_pack_bg32: row p holds the layers 2p and 2p+1 of
// a point side by side, 8 bytes per point like a DOUBLE row) instead of the DOUBLE rows - 656 instead of 872 bytes per point.
// The rows exist only when every value IS a float (a FLOAT background spectrum, as the CKDMIP files store it), so the
// converted value is the DOUBLE row's value bit for bit and so is everything computed from it.  The even wave takes the
// pairs top-down, the odd wave bottom-up with the two halves of a pair exchanged; with an odd number of layers per half the
// middle pair is shared: .x is the even wave's last layer, .y the odd wave's.
template <int NLAY, bool BG32>
__global__ void __launch_bounds__(RT_THREADS, 3)
k_rt_lw_bb_mirror(size_t n, int nint, long long nchunks, const Interval* __restrict__ iv,
                  const double* __restrict__ planck_hl, const double* __restrict__ bg_od,
                  const float_x2* __restrict__ bg_pair,
                  const double* __restrict__ od_fit, double* __restrict__ partial) {
  static_assert(NLAY % 2 == 0, "the column is split into two equal halves");
  constexpr int NHL = NLAY + 1;
  constexpr int H = NLAY / 2;
  constexpr int NSLOT = 2 * H + 1;               // slot 0: input of the first sweep, 1..H first sweep, H+1..2H second
  constexpr int NCH = (NSLOT + 15) / 16;
  constexpr int ROW = 65;
  constexpr int PTS = RT_THREADS / 2;            // points per block iteration: two wave pairs
  __shared__ double s_tile[4][16 * ROW];
  __shared__ double s_out[4][NCH * 16];
  __shared__ double s_x[2][4][64];               // [tile parity][wave][lane] flux handed to the partner wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave & 1, pair = wave >> 1;

  // rows of this wave: layers / levels 0,1,2,.. (even wave) or NLAY-1,NLAY-2,.. / NLAY,NLAY-1,.. (odd wave)
  const long long row_step = half ? -(long long)n : (long long)n;
  const double* __restrict__ od0 = bg_od + (half ? (size_t)(NLAY - 1) * n : 0);
  const float_x2* __restrict__ pr0 = bg_pair + (half ? (size_t)(NLAY / 2 - 1) * n : 0);
  const float* __restrict__ mid = (const float*)(bg_pair + (size_t)(H / 2) * n) + half;
  const double* __restrict__ pl0 = planck_hl + (half ? (size_t)NLAY * n : 0);
  double* tile = s_tile[wave];
  const int rr = lane & 15, qq = lane >> 4;
  int parity = 0;
  int klo = 0;

  // A block works through the chunks blockIdx.x, blockIdx.x + gridDim.x, ...; the launch gives every chunk its own block
  // (gridDim.x = nchunks) unless ECCKD_RT_PERSISTENT asks for one resident round of blocks (see the launch site).  A chunk is
  // what is summed by itself, in the same order, into its own row of `partial`, either way.
  for (long long chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
  int lo = klo, hi = nint - 1;
  while (lo < hi) {
    int mid_k = (lo + hi + 1) >> 1;
    if (iv[mid_k].chunk0 <= chunk) lo = mid_k; else hi = mid_k - 1;
  }
  const int k = lo;
  klo = k;
  const long long chunk_pts = iv[k].chunk_pts;
  const long long c = chunk - iv[k].chunk0;
  const long long p0 = iv[k].i1 + c * chunk_pts;
  long long p1 = p0 + chunk_pts - 1;
  if (p1 > iv[k].i2) p1 = iv[k].i2;
  // wave-uniform -> scalar loads; the odd waves read the bottom-up copy (k_interval_sums_fit_lw), so that both halves take
  // their layers at ascending constant offsets
  const double* __restrict__ grey = od_fit + ((size_t)(half ? nint : 0) + k) * NLAY;
  double acc[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) acc[j] = 0.0;

  for (long long base = p0; base <= p1; base += PTS, parity ^= 1) {
    const long long i = base + pair * 64 + lane;
    const bool live = i <= p1;
    const size_t ii = live ? (size_t)i : (size_t)p1;
    double a[H];       // optical depth -> transmittance
    double b[H + 1];   // Planck function -> source of the second sweep
    __builtin_amdgcn_s_setprio(ECCKD_PRIO_LOAD);     // a wave that is about to issue its 55 loads goes ahead of the waves that are computing
    // Addresses: a wave-uniform row pointer (scalar registers, advanced row by row with scalar adds) + the point's 32-bit byte
    // offset in a row (8 bytes per point in the DOUBLE rows and in the FLOAT-pair rows alike).  The row step is made opaque
    // once per tile, otherwise the compiler keeps all 42-55 row pointers of the unrolled loads in scalar registers across the
    // tile loop and spills them.
    const unsigned voff = (unsigned)ii * 8u;
    long long step_b = row_step * 8;
    asm volatile("" : "+s"(step_b));
    if constexpr (BG32) {
      float_x2 v[H / 2];
      float vm = 0.f;
      {
        const char* rp = (const char*)pr0;
#pragma unroll
        for (int q = 0; q < H / 2; ++q, rp += step_b) v[q] = __builtin_nontemporal_load((const float_x2*)(rp + voff));
        if (H & 1) vm = __builtin_nontemporal_load((const float*)((const char*)mid + voff));
      }
      {
        const char* rp = (const char*)pl0;
#pragma unroll
        for (int l = 0; l <= H; ++l, rp += step_b) b[l] = __builtin_nontemporal_load((const double*)(rp + voff));
      }
#pragma unroll
      for (int q = 0; q < H / 2; ++q) {
        a[2 * q] = (double)(half ? v[q].y : v[q].x);
        a[2 * q + 1] = (double)(half ? v[q].x : v[q].y);
      }
      if (H & 1) a[H - 1] = (double)vm;
    } else {
      const char* rp = (const char*)od0;
#pragma unroll
      for (int l = 0; l < H; ++l, rp += step_b) a[l] = __builtin_nontemporal_load((const double*)(rp + voff));
      rp = (const char*)pl0;
#pragma unroll
      for (int l = 0; l <= H; ++l, rp += step_b) b[l] = __builtin_nontemporal_load((const double*)(rp + voff));
    }
    __builtin_amdgcn_s_setprio(ECCKD_PRIO_SWEEP1);   // first sweep: the partner wave waits for its result

    // A lane past the end of the chunk (only the chunk's last tile has any) sweeps zero Planck functions: every flux of its
    // column is then an exact zero and adds nothing to the level sums - no select per level.
    if (base + PTS - 1 > p1) {
      const double keep = live ? 1.0 : 0.0;
#pragma unroll
      for (int l = 0; l <= H; ++l) b[l] *= keep;
    }
    int slot = 0;
    auto push = [&](double flux) {
      tile[(slot & 15) * ROW + lane] = flux;
      if ((slot & 15) == 15 || slot == NSLOT - 1) {
        const int ch = slot >> 4;
        __builtin_amdgcn_wave_barrier();
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += tile[rr * ROW + qq * 16 + j];
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        acc[ch] += sum;
        __builtin_amdgcn_wave_barrier();
      }
      ++slot;
    };

    // input of the first sweep: nothing comes down at the top; the surface emits its Planck function (:126-128)
    double flux = half ? b[0] : 0.0;
    push(flux);
    auto layer = [&](int l, double eps, double fac) {
      const double emf = eps - fac;
      const double t = 1.0 - eps;
      const double near = b[l], far = b[l + 1];
      flux = flux * t + near * emf + far * fac;
      a[l] = t;
      b[l] = far * emf + near * fac;          // source of the opposite direction, for the second sweep
      push(flux);
    };
#pragma unroll
    for (int l = 0; l + 1 < H; l += 2) {
      double eps0, fac0, eps1, fac1;
      eps_fac_pair(a[l] + grey[l], a[l + 1] + grey[l + 1], eps0, fac0, eps1, fac1);
      layer(l, eps0, fac0);
      layer(l + 1, eps1, fac1);
    }
    if (H & 1) {
      double eps0, fac0;
      eps_fac_one(a[H - 1] + grey[H - 1], eps0, fac0);
      layer(H - 1, eps0, fac0);
    }
    // the flux that enters this half from the other side is the partner wave's result
    s_x[parity][wave][lane] = flux;
    __builtin_amdgcn_s_setprio(ECCKD_PRIO_SWEEP2);
    __syncthreads();
    flux = s_x[parity][wave ^ 1][lane];
#pragma unroll
    for (int l = H - 1; l >= 0; --l) {
      flux = flux * a[l] + b[l];
      push(flux);
    }
    // (the rows NSLOT % 16 .. 15 of the last group keep the values of the group before: they reach only the sums of slots
    // >= NSLOT, which nothing reads)
  }

  if (lane < 16) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) s_out[wave][j * 16 + lane] = acc[j];
  }
  __syncthreads();      // the next chunk's s_out is written behind at least one more barrier (its tile loop)
  for (int t = tid; t < 2 * NHL; t += RT_THREADS) {
    // even waves (0, 2): slot 1+i = dn[i+1], slot H+1+k = up[H-1-k];
    // odd waves (1, 3):  slot 0 = up[NLAY], slot 1+i = up[NLAY-1-i], slot H+1+k = dn[H+1+k]
    int par = -1, sl = -1;
    if (t < NHL) {                       // flux_dn at level t
      if (t >= 1 && t <= H) { par = 0; sl = t; }
      else if (t > H) { par = 1; sl = t; }
    } else {                             // flux_up at level u
      const int u = t - NHL;
      if (u == NLAY) { par = 1; sl = 0; }
      else if (u >= H) { par = 1; sl = NLAY - u; }
      else { par = 0; sl = 2 * H - u; }
    }
    double v = 0.0;
    if (sl >= 0) v = s_out[par][sl] + s_out[par + 2][sl];
    partial[(size_t)chunk * 2 * NHL + t] = v;
  }
  }
}

constexpr int COST_GROUPS = 64;    // K5d: groups of 32 chunks per interval it has room for (2 048 chunks)

// K5d: combine chunk partials of each interval in order, heating rate, cost
// (calc_cost_function_lw.cpp:100-109).  grid nint, block 1024 = 8 groups x 128.
__global__ void __launch_bounds__(1024)
k_cost_lw(int nlay, RowMap R, const Interval* __restrict__ iv, long long nchunks_total,
          const double* __restrict__ partial, const double* __restrict__ sums,
          const double* __restrict__ conv, const double* __restrict__ layer_weight,
          double flux_weight, double* __restrict__ err) {
  extern __shared__ double s_mem[];  // [COST_GROUPS][2*nhl] | [2*nhl] | [nlay]
  const int nhl = nlay + 1;
  const int nv = 2 * nhl;
  double* s_grp = s_mem;
  double* s_flux = s_mem + COST_GROUPS * nv;
  double* s_term = s_flux + nv;
  const int k = blockIdx.x;
  const long long c0 = iv[k].chunk0;
  const long long c1 = (k + 1 < (int)gridDim.x) ? iv[k + 1].chunk0 : nchunks_total;
  const int tid = threadIdx.x;
  // The interval's chunks are added up in groups of 32 consecutive chunks (each group from 0.0 in chunk order), the groups
  // then in ascending order: a function of the interval's chunk count alone.  Thread group g (128 threads: one per flux
  // value) takes the chunk groups g, g + 8, ...; sixteen loads are in flight together.
  const int g = tid >> 7, t = tid & 127;
  const long long ncg = (c1 - c0 + 31) / 32;
  for (long long cg = g; cg < ncg; cg += 8) {
    const long long cb = c0 + cg * 32;
    const int cnt = (int)((c1 - cb) < 32 ? (c1 - cb) : 32);
    if (t < nv) {
      // sixteen loads in flight, unconditionally (behind the end of a short group: its last chunk again, the value replaced
      // by 0.0), added in chunk order.  (All 32 of a group in flight: 11.7 instead of 10.1 us per launch - the one CU that sums
      // an interval is bound by its own load path, not by the round trips.)
      double a = 0.0;
#pragma unroll 1
      for (int h = 0; h < 32; h += 16) {
        double q[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) q[u] = partial[(size_t)(cb + (h + u < cnt ? h + u : cnt - 1)) * nv + t];
#pragma unroll
        for (int u = 0; u < 16; ++u) a += h + u < cnt ? q[u] : 0.0;      // + 0.0 changes nothing
      }
      s_grp[(size_t)cg * nv + t] = a;
    }
  }
  __syncthreads();
  for (int v = tid; v < nv; v += 1024) {
    double a = 0.0;
    for (long long cg = 0; cg < ncg; ++cg) a += s_grp[(size_t)cg * nv + v];
    s_flux[v] = a;
  }
  __syncthreads();
  const double* s = sums + (size_t)k * R.total;
  const double* dn = s_flux;
  const double* up = s_flux + nhl;
  for (int l = tid; l < nlay; l += 1024) {
    // heating_rate_single (heating_rate.h:55-72) and the weighted squared difference
    const double hr_fit = conv[l] * (dn[l + 1] - dn[l] - up[l + 1] + up[l]);
    const double d = hr_fit - s[R.H + l];
    s_term[l] = layer_weight[l] * (d * d);
  }
  __syncthreads();
  if (tid == 0) {
    double ss = 0.0;
    for (int l = 0; l < nlay; ++l) ss += s_term[l];
    const double hr_weight = 3600.0 * 24.0;
    const double dsurf = dn[nlay] - s[R.FDS];
    const double dtoa = up[0] - s[R.FUT];
    err[k] = sqrt(hr_weight * hr_weight * ss + flux_weight * (dsurf * dsurf + dtoa * dtoa));
  }
}


// ===========================================================================
// Shortwave twins (find_g_points.cpp do_sw branches).
//
// K4-SW.  Gas preparation: direct-beam RT of background+target
// (radiative_transfer_direct_sw, radiative_transfer_sw.cpp:26-43), heating rate from
// the direct beam only (find_g_points.cpp:1003-1006, :1040-1041), metric * ssi rows, and
// for the total-transmission method the per-point direct fluxes that
// fit_optical_depth_sw_total_trans sums (:171-204) plus the two scaled "truth" fields
// (:1011-1034, :1060-1090).  One thread per sorted wavenumber.
// This is the general form (any number of layers, FLOAT or DOUBLE spectra): the (level, wavenumber) matrices are gathered
// through ireorder, the tile sums of the rows are made afterwards by k_tile_sums.  54 layers of FLOAT optical depths take
// k_gas_prep_sw_staged below.
template <typename BgT, typename OdT>
__global__ void __launch_bounds__(PREP_THREADS)
k_gas_prep_sw(int nlay, size_t n, size_t src_stride, int method, double cos_sza, double min_scaling,
              double max_scaling, const int32_t* __restrict__ ireorder, const double* __restrict__ conv,
              const double* __restrict__ ssi_src, const double* __restrict__ albedo_src,
              const BgT* __restrict__ bg_src, const OdT* __restrict__ od_src,
              double* __restrict__ ssi_s, double* __restrict__ bg_od, double* __restrict__ w1,
              double* __restrict__ w2, double* __restrict__ cnt, double* __restrict__ hr,
              double* __restrict__ fds, double* __restrict__ fut, double* __restrict__ tf,
              double* __restrict__ tg, double* __restrict__ hr_low, double* __restrict__ hr_high,
              double* __restrict__ fx) {
  // no contraction of a * b + c beyond the explicit fma of exp_fast: this kernel and the staged one then give the same bits
#pragma clang fp contract(off)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t j = (size_t)ireorder[i];
  const bool is_log = method == ECCKD_AVG_LOGARITHMIC;
  const bool is_tt = method == ECCKD_AVG_TOTAL_TRANSMISSION;
  const double s = ssi_src[j];
  ssi_s[i] = s;
  const double minus_sec_sza = -1.0 / cos_sza;
  double flux = cos_sza * s;
  double fl_low = flux, fl_high = flux;
  double tfv = s, tgv = s;  // :178-179 start from ssi, not cos_sza*ssi
  // inputs are fetched nine layers at a time and one chunk ahead: a load issued after a store waits for that store on this
  // hardware (one counter, in order), so one load per layer would cost one store latency per layer; requested before the
  // chunk's own stores, the wait for them overlaps the chunk's arithmetic
  constexpr int CH = 9;
  BgT bgn[CH];
  OdT odn[CH];
  // no background: the loads still happen (from the target's own array, same shape, when its elements are at least as wide)
  // and their values are dropped - a branch round each load would put a full wait behind it
  const bool has_bg = bg_src != nullptr;
  constexpr bool BG_ALIAS = sizeof(BgT) <= sizeof(OdT);
  const BgT* __restrict__ bsrc = (has_bg || !BG_ALIAS) ? bg_src : reinterpret_cast<const BgT*>(od_src);
  auto load_bg = [&](size_t at) -> BgT {
    if (BG_ALIAS) { const BgT v = bsrc[at]; return has_bg ? v : (BgT)0; }
    return has_bg ? bg_src[at] : (BgT)0;
  };
  auto fetch = [&](int l0) {
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const int lq = l0 + q < nlay ? l0 + q : nlay - 1;
      bgn[q] = load_bg((size_t)lq * src_stride + j);
      odn[q] = od_src[(size_t)lq * src_stride + j];
    }
  };
  fetch(0);
  for (int l0 = 0; l0 < nlay; l0 += CH) {
  BgT bgv[CH];
  OdT odv[CH];
#pragma unroll
  for (int q = 0; q < CH; ++q) { bgv[q] = bgn[q]; odv[q] = odn[q]; }
  if (l0 + CH < nlay) fetch(l0 + CH);
#pragma unroll
  for (int q = 0; q < CH; ++q) {
    const int l = l0 + q;
    if (l >= nlay) continue;
    const double bg = (double)bgv[q];
    const double od = (double)odv[q];
    const size_t o = (size_t)l * n + i;
    __builtin_nontemporal_store(bg, &bg_od[o]);      // written once, read by later kernels: streaming stores
    const double flux_next = flux * ecckd::exp_fast(minus_sec_sza * (bg + od));
    const double hrv = conv[l] * (flux_next - flux);
    __builtin_nontemporal_store(hrv, &hr[o]);
    flux = flux_next;
    double m = od;                       // find_g_points.cpp:1119-1150: linear, logarithmic, total-transmission
    if (method == ECCKD_AVG_TRANSMISSION) m = 1.0 - ecckd::exp_fast(-od * kD);
    else if (method == ECCKD_AVG_TRANSMISSION_2) m = 1.0 - ecckd::exp_fast(-od * kD * 2.0);
    else if (method == ECCKD_AVG_SQUARE_ROOT) m = sqrt(od);
    if (!is_log) {
      const double a = m * s;
      __builtin_nontemporal_store(a, &w1[o]);
    } else {
      const bool pos = m > 0.0;
      const double a = pos ? log(m) * s : 0.0, b = pos ? s : 0.0, c = pos ? 1.0 : 0.0;
      __builtin_nontemporal_store(a, &w1[o]); __builtin_nontemporal_store(b, &w2[o]); __builtin_nontemporal_store(c, &cnt[o]);
    }
    if (is_tt) {
      // :191-192
      tgv *= ecckd::exp_fast(-2.0 * bg);
      tfv *= ecckd::exp_fast(-2.0 * (bg + od));
      const double lo_next = fl_low * ecckd::exp_fast(minus_sec_sza * (bg + min_scaling * od));
      const double hl = conv[l] * (lo_next - fl_low);
      fl_low = lo_next;
      const double hi_next = fl_high * ecckd::exp_fast(minus_sec_sza * (bg + max_scaling * od));
      const double hh = conv[l] * (hi_next - fl_high);
      fl_high = hi_next;
      __builtin_nontemporal_store(tgv, &tg[o]); __builtin_nontemporal_store(tfv, &tf[o]);
      __builtin_nontemporal_store(hl, &hr_low[o]); __builtin_nontemporal_store(hh, &hr_high[o]);
    }
  }
  }
  fds[i] = flux;
  fut[i] = 0.0;  // find_g_points.cpp:1047-1050: no upwelling in the base truth
  if (is_tt) {
    double up_low = 0.0, up_high = 0.0;
    if (albedo_src) {
      // radiative_transfer_norayleigh_sw (radiative_transfer_sw.cpp:72-76), two-stream secant 2
      const double alb = albedo_src[j];
      up_low = fl_low * alb;
      up_high = fl_high * alb;
      for (int l = nlay - 1; l >= 0; --l) {
        const double bg = (double)load_bg((size_t)l * src_stride + j);
        const double od = (double)od_src[(size_t)l * src_stride + j];
        up_low = up_low * ecckd::exp_fast(-2.0 * (bg + min_scaling * od));
        up_high = up_high * ecckd::exp_fast(-2.0 * (bg + max_scaling * od));
      }
    }
    fx[i] = fl_low;
    fx[2 * n + i] = fl_high;
    fx[n + i] = up_low;
    fx[3 * n + i] = up_high;
  }
}

// K4-SW, staged form (54 layers, FLOAT optical depths: the CKDMIP spectra).  The same arithmetic as k_gas_prep_sw, with the
// inputs brought in differently: k_scatter_column_halves leaves each point's column at its rank as PARTS runs of H = 54 / PARTS
// layers, [PARTS][npad][H], so that the H-layer block of a wave's 64 points is one contiguous, 256-byte aligned piece.  The wave
// copies it into LDS with full-width loads and every lane reads its own column back from there.  Reading the columns
// straight from memory, one 4-byte element per lane and layer at a stride of 216 bytes, took 64 cache lines per load
// instruction and more time than the arithmetic and the row stores together (ablation).  The upward
// sweep of the total-transmission truths starts on the part that is still staged and fetches the others once more.
// Row sums: the (at most seven) values a layer adds to the table go through a wave-private [7][65] tile, summed per layer.
// Three parts of 18 layers: 51 KB of LDS per block, three blocks (three waves per SIMD) per compute unit.
constexpr int SWS_TR = 7, SWS_TW = 65, SW_STAGE_PARTS = 3;
template <int PARTS, typename BgT>
__global__ void __launch_bounds__(PREP_THREADS, sizeof(BgT) == 4 ? PARTS : 2)
k_gas_prep_sw_staged(size_t n, size_t npad, int method, double cos_sza, double min_scaling, double max_scaling,
                     const int32_t* __restrict__ ireorder, const double* __restrict__ conv,
                     const double* __restrict__ ssi_src, const double* __restrict__ albedo_src,
                     const BgT* __restrict__ bg_half, const float* __restrict__ od_half,
                     double* __restrict__ ssi_s, double* __restrict__ bg_od, double* __restrict__ w1,
                     double* __restrict__ w2, double* __restrict__ cnt, double* __restrict__ hr,
                     double* __restrict__ fds, double* __restrict__ fut, double* __restrict__ tf,
                     double* __restrict__ tg, double* __restrict__ hr_low, double* __restrict__ hr_high,
                     double* __restrict__ fx, RowMap R, double* __restrict__ wave_part, size_t nw) {
#pragma clang fp contract(off)
  constexpr int H = 54 / PARTS;
  static_assert(H * PARTS == 54 && (64 * H) % 4 == 0, "whole parts, whole float4s");
  __shared__ __attribute__((aligned(16))) float s_od[4][64 * H];
  __shared__ __attribute__((aligned(16))) BgT s_bg[4][64 * H];     // a DOUBLE (merged) background: 70 KB per block, two blocks per CU
  __shared__ double s_sum[4][SWS_TR * SWS_TW];
  __shared__ int s_row[4][SWS_TR];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t wid = (size_t)blockIdx.x * 4 + wave;
  if (wid * 64 >= n) return;   // a whole wave past the end (no block-wide barrier below)
  const size_t i0 = wid * 64 + lane;
  const bool live = i0 < n;
  const size_t i = live ? i0 : n - 1;
  const bool has_bg = bg_half != nullptr;
  float* const my_od = s_od[wave];
  BgT* const my_bg = s_bg[wave];
  double* const my_sum = s_sum[wave];
  int* const my_row = s_row[wave];

  // this wave's 64 x H block of part h: contiguous and 256-byte aligned (npad is a multiple of 64)
  auto stage = [&](int h) {
    constexpr int NV = 64 * H / 4, NIT = (NV + 63) / 64;                    // float4s of the target's block
    constexpr int NVB = NV * (int)(sizeof(BgT) / 4);                        // 16-byte pieces of the background's block
    const float4* so = reinterpret_cast<const float4*>(od_half + ((size_t)h * npad + wid * 64) * H);
    // no background: the loads still happen, from the target's block (a branch round them would put a full wait behind each)
    const float4* sb = has_bg ? reinterpret_cast<const float4*>(bg_half + ((size_t)h * npad + wid * 64) * H) : so;
    __builtin_amdgcn_wave_barrier();   // the lanes are done with the part staged before
    // both arrays' loads go out together, then the LDS writes: one memory round trip per staging
#pragma unroll
    for (int t = 0; t < NIT; ++t) {
      const int at = t * 64 + lane;
      if (at < NV) {
        const float4 vo = so[at], vb = sb[at];
        reinterpret_cast<float4*>(my_od)[at] = vo;
        reinterpret_cast<float4*>(my_bg)[at] = vb;
      }
    }
    if (NVB > NV) {                    // a DOUBLE background: the second half of its block
#pragma unroll
      for (int t = 0; t < NIT; ++t) {
        const int at = NV + t * 64 + lane;
        if (at < NVB) reinterpret_cast<float4*>(my_bg)[at] = sb[at];
      }
    }
    __builtin_amdgcn_wave_barrier();
  };
  int slot = 0;
  auto push = [&](int row, double v) {
    my_sum[slot * SWS_TW + lane] = live ? v : 0.0;
    my_row[slot] = row;
    ++slot;
  };
  // the sums of the pushed rows over the wave's 64 points: lane (r, q) adds eight points of row r, three exchanges add the eighths
  auto flush = [&]() {
    const int rr = lane & 7, qq = lane >> 3;
    __builtin_amdgcn_wave_barrier();
    double sum = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) sum += rr < slot ? my_sum[rr * SWS_TW + qq * 8 + q] : 0.0;
    sum += __shfl_xor(sum, 8, 64);
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    if (qq == 0 && rr < slot) wave_part[(size_t)my_row[rr] * nw + wid] = sum;
    __builtin_amdgcn_wave_barrier();
    slot = 0;
  };

  const size_t j = (size_t)ireorder[i];
  const bool is_log = method == ECCKD_AVG_LOGARITHMIC;
  const bool is_tt = method == ECCKD_AVG_TOTAL_TRANSMISSION;
  const double s = ssi_src[j];
  if (live) ssi_s[i] = s;
  const double minus_sec_sza = -1.0 / cos_sza;
  double flux = cos_sza * s;
  double fl_low = flux, fl_high = flux;
  double tfv = s, tgv = s;  // :178-179 start from ssi, not cos_sza*ssi
  for (int h = 0; h < PARTS; ++h) {
    stage(h);
#pragma unroll 2
    for (int k = 0; k < H; ++k) {
      const int l = h * H + k;
      const double bg = has_bg ? (double)my_bg[lane * H + k] : 0.0;
      const double od = (double)my_od[lane * H + k];
      const size_t o = (size_t)l * n + i;
      if (live) __builtin_nontemporal_store(bg, &bg_od[o]);      // written once, read by later kernels: streaming stores
      const double flux_next = flux * ecckd::exp_fast(minus_sec_sza * (bg + od));
      const double hrv = conv[l] * (flux_next - flux);
      if (live) __builtin_nontemporal_store(hrv, &hr[o]);
      push(R.H + l, hrv);
      flux = flux_next;
      double m = od;                       // find_g_points.cpp:1119-1150: linear, logarithmic, total-transmission
      if (method == ECCKD_AVG_TRANSMISSION) m = 1.0 - ecckd::exp_fast(-od * kD);
      else if (method == ECCKD_AVG_TRANSMISSION_2) m = 1.0 - ecckd::exp_fast(-od * kD * 2.0);
      else if (method == ECCKD_AVG_SQUARE_ROOT) m = sqrt(od);
      if (!is_log) {
        const double a = m * s;
        if (live) __builtin_nontemporal_store(a, &w1[o]);
        push(R.A + l, a);
        push(R.B + l, s);
      } else {
        const bool pos = m > 0.0;
        const double a = pos ? log(m) * s : 0.0, b = pos ? s : 0.0, c = pos ? 1.0 : 0.0;
        if (live) { __builtin_nontemporal_store(a, &w1[o]); __builtin_nontemporal_store(b, &w2[o]); __builtin_nontemporal_store(c, &cnt[o]); }
        push(R.A + l, a);
        push(R.B + l, b);
        push(R.N + l, c);
      }
      if (is_tt) {
        // :191-192
        tgv *= ecckd::exp_fast(-2.0 * bg);
        tfv *= ecckd::exp_fast(-2.0 * (bg + od));
        const double lo_next = fl_low * ecckd::exp_fast(minus_sec_sza * (bg + min_scaling * od));
        const double hl = conv[l] * (lo_next - fl_low);
        fl_low = lo_next;
        const double hi_next = fl_high * ecckd::exp_fast(minus_sec_sza * (bg + max_scaling * od));
        const double hh = conv[l] * (hi_next - fl_high);
        fl_high = hi_next;
        if (live) {
          __builtin_nontemporal_store(tgv, &tg[o]); __builtin_nontemporal_store(tfv, &tf[o]);
          __builtin_nontemporal_store(hl, &hr_low[o]); __builtin_nontemporal_store(hh, &hr_high[o]);
        }
        push(R.TG + l, tgv);
        push(R.TF + l, tfv);
        push(R.HL + l, hl);
        push(R.HH + l, hh);
      }
      flush();
    }
  }
  if (live) {
    fds[i] = flux;
    fut[i] = 0.0;  // find_g_points.cpp:1047-1050: no upwelling in the base truth
  }
  push(R.FDS, flux);
  push(R.FUT, 0.0);
  if (is_tt) {
    double up_low = 0.0, up_high = 0.0;
    if (albedo_src) {
      // radiative_transfer_norayleigh_sw (radiative_transfer_sw.cpp:72-76), two-stream secant 2; the lowest part is still staged
      const double alb = albedo_src[j];
      up_low = fl_low * alb;
      up_high = fl_high * alb;
      for (int h = PARTS - 1; h >= 0; --h) {
        if (h != PARTS - 1) stage(h);
#pragma unroll 3
        for (int k = H - 1; k >= 0; --k) {
          const double bg = has_bg ? (double)my_bg[lane * H + k] : 0.0;
          const double od = (double)my_od[lane * H + k];
          up_low = up_low * ecckd::exp_fast(-2.0 * (bg + min_scaling * od));
          up_high = up_high * ecckd::exp_fast(-2.0 * (bg + max_scaling * od));
        }
      }
    }
    if (live) {
      fx[i] = fl_low;
      fx[2 * n + i] = fl_high;
      fx[n + i] = up_low;
      fx[3 * n + i] = up_high;
    }
    push(R.FDSL, fl_low);
    push(R.FUTL, up_low);
    push(R.FDSH, fl_high);
    push(R.FUTH, up_high);
  }
  flush();
}

// K5b-SW: fit_optical_depth_sw (find_g_points.cpp:112-165) and
// fit_optical_depth_sw_total_trans (:171-204) from the interval sums.
// Writes npass fits per interval: pass 0 (and pass 1 for total-transmission:
// fit*min_scaling, fit*max_scaling, :357 and :371).
__global__ void __launch_bounds__(128)
k_fit_sw(int nlay, int method, RowMap R, int nint, double min_scaling, double max_scaling,
         const Interval* __restrict__ iv, const double* __restrict__ sums, double* __restrict__ od_fit) {
#pragma clang fp contract(off)
  extern __shared__ double s_fit[];  // [nlay]
  const int k = blockIdx.x;
  const double* s = sums + (size_t)k * R.total;
  if (method != ECCKD_AVG_TOTAL_TRANSMISSION) {
    for (int l = threadIdx.x; l < nlay; l += blockDim.x) {
      const double a = s[R.A + l];
      double fit;
      if (method == ECCKD_AVG_LOGARITHMIC) {
        const double b = s[R.B + l];
        const double nnz = s[R.N + l];
        const double ntot = (double)(iv[k].i2 - iv[k].i1 + 1);
        if (nnz == ntot) fit = exp(a / b);
        else if (nnz == 0.0) fit = 0.0;
        else fit = exp(a / b) * (nnz / ntot);
      } else {
        const double norm_factor = 1.0 / s[R.B];  // sum(ssi(range(i1,i2)))
        switch (method) {
          case ECCKD_AVG_LINEAR: fit = a * norm_factor; break;
          // :123-124: the clamp is applied BEFORE the normalisation
          // contraction is OFF in this kernel: the product must be ROUNDED before the subtraction.
          // A fused 1 - a*(1/b) keeps the rounding error of 1/b and goes negative when a == b
          // (saturated interval), turning the reference's log(0) = -inf into log(<0) = NaN.
          case ECCKD_AVG_TRANSMISSION:
            fit = fabs(-log(1.0 - fmin(0.9999999999999999, a) * norm_factor) / kD); break;
          case ECCKD_AVG_TRANSMISSION_2:
            fit = fabs(-log(1.0 - fmin(0.9999999999999999, a) * norm_factor) / (kD * 2.0)); break;
          case ECCKD_AVG_SQUARE_ROOT: { const double v = a * norm_factor; fit = v * v; break; }
          default: fit = nan("");
        }
      }
      od_fit[(size_t)k * nlay + l] = fit;
    }
    return;
  }
  // total-transmission (:189-203).  The reference walks the layers in order; a layer whose direct beam has
  // vanished replaces the WHOLE vector by the linear average and the walk continues, so the outcome is:
  // layers above and at the LAST such layer carry the linear average, layers below it their own fit.
  // One thread per layer; the last invalid layer is found with an integer LDS max.
  __shared__ int s_last_invalid;
  if (threadIdx.x == 0) s_last_invalid = -1;
  __syncthreads();
  const double ssum = s[R.B];
  const double norm_factor = 1.0 / ssum;
  for (int iz = threadIdx.x; iz < nlay; iz += blockDim.x) {
    const double base_bg = s[R.TG + iz], base = s[R.TF + iz];
    if (!(base_bg > 0.0 && base > 0.0)) atomicMax(&s_last_invalid, iz);
  }
  __syncthreads();
  const int last_invalid = s_last_invalid;
  for (int iz = threadIdx.x; iz < nlay; iz += blockDim.x) {
    if (iz > last_invalid) {
      const double top = iz == 0 ? ssum : s[R.TF + iz - 1];
      const double top_bg = iz == 0 ? ssum : s[R.TG + iz - 1];
      const double bg_od_fit = -0.5 * log(s[R.TG + iz] / top_bg);
      s_fit[iz] = -0.5 * log(s[R.TF + iz] / top) - bg_od_fit;
    } else {
      s_fit[iz] = s[R.A + iz] * norm_factor;
    }
  }
  __syncthreads();
  for (int l = threadIdx.x; l < nlay; l += blockDim.x) {
    od_fit[(size_t)k * nlay + l] = s_fit[l] * min_scaling;
    od_fit[((size_t)nint + k) * nlay + l] = s_fit[l] * max_scaling;
  }
}

// K5c-SW: radiative_transfer_direct_sw_bb / _norayleigh_sw_bb
// (radiative_transfer_sw.cpp:118-141, :147-184) for every interval: reads ssi and the
// background optical depth rows, (nlay+1)*8 B per point.  partial[chunk] =
// { sum ssi, dn[1..nlay], up[0..nlay] } (dn[0] = cos_sza * sum ssi is formed in K5d as the
// reference does, :128).  Same chunking and reduction order as the LW kernel.
__global__ void __launch_bounds__(RT_THREADS)
k_rt_sw_bb(int nlay, size_t n, int nint, const Interval* __restrict__ iv,
           double cos_sza, const double* __restrict__ ssi, const double* __restrict__ bg_od,
           const double* __restrict__ od_fit, double* __restrict__ partial) {
  extern __shared__ double s_mem[];  // [4][2*nhl] | [nlay]
  const int nhl = nlay + 1;
  double* s_acc = s_mem;
  double* s_grey = s_mem + 4 * 2 * nhl;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long chunk = blockIdx.x;
  int lo = 0, hi = nint - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (iv[mid].chunk0 <= chunk) lo = mid; else hi = mid - 1;
  }
  const int k = lo;
  const double albedo = iv[k].albedo;
  const long long chunk_pts = iv[k].chunk_pts;
  const long long c = chunk - iv[k].chunk0;
  const long long p0 = iv[k].i1 + c * chunk_pts;
  long long p1 = p0 + chunk_pts - 1;
  if (p1 > iv[k].i2) p1 = iv[k].i2;
  for (int t = tid; t < 4 * 2 * nhl; t += RT_THREADS) s_acc[t] = 0.0;
  for (int l = tid; l < nlay; l += RT_THREADS) s_grey[l] = od_fit[(size_t)k * nlay + l];
  __syncthreads();
  double* acc_dn = s_acc + wave * 2 * nhl;
  double* acc_up = acc_dn + nhl;
  const double minus_sec_sza = -1.0 / cos_sza;
  for (long long base = p0; base <= p1; base += RT_THREADS) {
    const long long i = base + tid;
    const bool live = i <= p1;
    const size_t ii = live ? (size_t)i : (size_t)p1;
    const double sv = live ? ssi[ii] : 0.0;
    {
      const double s0 = wave_sum(sv);
      if (lane == 0) acc_dn[0] += s0;
    }
    double flux = cos_sza * sv;
    for (int l = 0; l < nlay; ++l) {
      flux = flux * exp(minus_sec_sza * (bg_od[(size_t)l * n + ii] + s_grey[l]));
      const double sl = wave_sum(flux);
      if (lane == 0) acc_dn[l + 1] += sl;
    }
    if (albedo > 0.0) {
      flux *= albedo;
      {
        const double sl = wave_sum(flux);
        if (lane == 0) acc_up[nlay] += sl;
      }
      for (int l = nlay - 1; l >= 0; --l) {
        flux = flux * exp(-2.0 * (bg_od[(size_t)l * n + ii] + s_grey[l]));
        const double sl = wave_sum(flux);
        if (lane == 0) acc_up[l] += sl;
      }
    }
  }
  __syncthreads();
  for (int t = tid; t < 2 * nhl; t += RT_THREADS) {
    partial[(size_t)chunk * 2 * nhl + t] =
        ((s_acc[t] + s_acc[2 * nhl + t]) + s_acc[4 * nhl + t]) + s_acc[6 * nhl + t];
  }
}



// K5c-SW fast path: NLAY known at compile time.  The column of total optical depths (background + grey) stays in
// registers for both sweeps (direct beam down, radiative_transfer_sw.cpp:134-139, reflected beam up, :176-183:
// two different exponents per layer, no sources), level sums through the wave-private transposed LDS tile of the
// longwave path.  Slots: 0 = sum of the solar irradiance, 1..NLAY = flux_dn below each layer, NLAY+1 = flux_up at the
// surface, NLAY+2.. = flux_up above each layer going up.
// NFIT = 2 (total-transmission with min_scaling != max_scaling, find_g_points.cpp:374-386): the column is fetched once and
// swept with both fitted optical depths, partial sums of the second fit nchunks * 2 NHL further on.
// SAME: cos_sza == 0.5, so -tau / cos_sza == -2 tau exactly and the transmittance of the way down is the one of the way up
// (radiative_transfer_sw.cpp:134-139 and :176-183): the last fit overwrites the column with it instead of evaluating it twice.
// (three blocks per CU only where the column's transmittance is shared by the two beams: without SAME the second exponential's
// temporaries do not fit 168 registers - 56 B per lane went to scratch - and the cosine is the reference's 0.5 everywhere but in tests)
template <int NLAY, int NFIT, bool SAME, int OCC = (NFIT == 1 && SAME ? 3 : 2)>
__global__ void __launch_bounds__(RT_THREADS, OCC)
k_rt_sw_bb_fast(size_t n, int nint, const Interval* __restrict__ iv, double cos_sza,
                const double* __restrict__ ssi, const double* __restrict__ bg_od, const double* __restrict__ od_fit,
                double* __restrict__ partial) {
  constexpr int NHL = NLAY + 1;
  constexpr int NSLOT = 2 * NLAY + 2;
  constexpr int NCH = (NSLOT + 15) / 16;
  constexpr int ROW = 65;
  __shared__ double s_tile[4][16 * ROW];
  __shared__ double s_out[NFIT][4][NCH * 16];
  __shared__ double s_grey[NFIT][NLAY];   // the fitted optical depths: LDS broadcasts (as scalars they take 2 NLAY SGPRs per fit and spill)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long chunk = blockIdx.x;
  int lo = 0, hi = nint - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (iv[mid].chunk0 <= chunk) lo = mid; else hi = mid - 1;
  }
  const int k = lo;
  const long long chunk_pts = iv[k].chunk_pts;
  const long long c = chunk - iv[k].chunk0;
  const long long p0 = iv[k].i1 + c * chunk_pts;
  long long p1 = p0 + chunk_pts - 1;
  if (p1 > iv[k].i2) p1 = iv[k].i2;
  for (int t = tid; t < NFIT * NLAY; t += RT_THREADS) {
    const int f = t / NLAY, l = t - f * NLAY;
    s_grey[f][l] = od_fit[((size_t)f * nint + k) * NLAY + l];
  }
  __syncthreads();
  double* tile = s_tile[wave];
  const int rr = lane & 15, qq = lane >> 4;
  double acc[NFIT][NCH];
#pragma unroll
  for (int f = 0; f < NFIT; ++f) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) acc[f][j] = 0.0;
  }
  const double minus_sec_sza = -1.0 / cos_sza;
  // exp with its coefficients in scalar registers (fastmath.hpp).  A saturated interval fits an infinite optical depth
  // (log(0), find_g_points.cpp:123-124): the argument is held at -800, where the result has already underflowed to 0,
  // because exp_fast_s(-inf) would be inf - inf.
  const ecckd::ExpConsts ek = ecckd::exp_consts();
  auto trans = [&](double x) { return ecckd::exp_fast_s(fmax(x, -800.0), ek); };
  const double albedo = iv[k].albedo;
  const bool reflect = albedo > 0.0;                 // :366-373: no upwelling without a reflecting surface
  for (long long base = p0; base <= p1; base += RT_THREADS) {
    const long long i = base + tid;
    const bool live = i <= p1;
    const size_t ii = live ? (size_t)i : (size_t)p1;
    double tau[NLAY];
    __builtin_amdgcn_s_setprio(3);     // as in the longwave kernel: the wave that is about to fetch its column goes first
    {
      // one wave-uniform row pointer walked with scalar adds + the point's 32-bit byte offset (as in the longwave kernel: the
      // 54 row pointers of the unrolled loads otherwise live in scalar registers across the tile loop and spill)
      const unsigned voff = (unsigned)ii * 8u;
      long long step_b = (long long)n * 8;
      asm volatile("" : "+s"(step_b));
      const char* rp = (const char*)bg_od;
#pragma unroll
      for (int l = 0; l < NLAY; ++l, rp += step_b) tau[l] = __builtin_nontemporal_load((const double*)(rp + voff));
    }
    const double sv = live ? ssi[ii] : 0.0;   // a lane past the end carries zero flux through both sweeps: nothing else to mask
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int f = 0; f < NFIT; ++f) {
      const double* grey = s_grey[f];
      int slot = 0;
      auto push = [&](double v) {
        tile[(slot & 15) * ROW + lane] = v;
        if ((slot & 15) == 15 || slot == NSLOT - 1) {
          const int ch = slot >> 4;
          __builtin_amdgcn_wave_barrier();
          double sum = 0.0;
#pragma unroll
          for (int j = 0; j < 16; ++j) sum += tile[rr * ROW + qq * 16 + j];
          sum += __shfl_xor(sum, 16, 64);
          sum += __shfl_xor(sum, 32, 64);
          acc[f][ch] += sum;
          __builtin_amdgcn_wave_barrier();
        }
        ++slot;
      };
      push(sv);
      double flux = cos_sza * sv;
      const bool last = f == NFIT - 1;          // compile time after unrolling: the column is free to be overwritten
      if (last && SAME) {
#pragma unroll
        for (int l = 0; l < NLAY; ++l) {
          const double xd = -2.0 * (tau[l] + grey[l]);
          tau[l] = trans(xd);
          flux = flux * tau[l];
          push(flux);
        }
      } else if (last) {
#pragma unroll
        for (int l = 0; l < NLAY; ++l) {
          tau[l] += grey[l];
          const double xd = minus_sec_sza * tau[l];
          flux = flux * trans(xd);
          push(flux);
        }
      } else {
#pragma unroll
        for (int l = 0; l < NLAY; ++l) {
          const double xd = minus_sec_sza * (tau[l] + grey[l]);
          flux = flux * trans(xd);
          push(flux);
        }
      }
      flux = reflect ? flux * albedo : 0.0;
      push(flux);
#pragma unroll
      for (int l = NLAY - 1; l >= 0; --l) {
        if (last && SAME) {
          if (reflect) flux = flux * tau[l];
        } else {
          const double xu = -2.0 * (last ? tau[l] : tau[l] + grey[l]);
          if (reflect) flux = flux * trans(xu);
        }
        push(flux);
      }
      if ((NSLOT & 15) != 0) {
#pragma unroll
        for (int j = (NSLOT & 15); j < 16; ++j) tile[j * ROW + lane] = 0.0;
      }
    }
  }
  if (lane < 16) {
#pragma unroll
    for (int f = 0; f < NFIT; ++f) {
#pragma unroll
      for (int j = 0; j < NCH; ++j) s_out[f][wave][j * 16 + lane] = acc[f][j];
    }
  }
  __syncthreads();
  for (int t = tid; t < NFIT * 2 * NHL; t += RT_THREADS) {
    // flux_dn level t = slot t; flux_up level u = slot NLAY + 1 + (NLAY - u)
    const int f = t / (2 * NHL), tt = t - f * 2 * NHL;
    const int sl = tt < NHL ? tt : NLAY + 1 + (NLAY - (tt - NHL));
    partial[((size_t)f * gridDim.x + chunk) * 2 * NHL + tt] = ((s_out[f][0][sl] + s_out[f][1][sl]) + s_out[f][2][sl]) + s_out[f][3][sl];
  }
}
struct SwTruthRows { int rH[2], rFDS[2], rFUT[2]; };

// K5d-SW: calc_cost_function_sw (calc_cost_function_sw.cpp:86-109): heating rate from the
// direct beam only (:92), truth rows selected by (rH, rFDS, rFUT).
__global__ void __launch_bounds__(1024)
k_cost_sw(int nlay, int ntotal, SwTruthRows rows, const Interval* __restrict__ iv,
          long long nchunks_total, const double* __restrict__ partial_all, size_t partial_pass_stride,
          const double* __restrict__ sums, const double* __restrict__ conv, const double* __restrict__ layer_weight,
          double flux_weight, double cos_sza, double* __restrict__ err_all) {
  // blockIdx.y: which of the (one or two) evaluations of the interval - its truth rows, its sweep's partial sums, its error slot
  const int pass = blockIdx.y;
  const int rH = rows.rH[pass], rFDS = rows.rFDS[pass], rFUT = rows.rFUT[pass];
  const double* __restrict__ partial = partial_all + (size_t)pass * partial_pass_stride;
  double* __restrict__ err = err_all + (size_t)pass * gridDim.x;
  extern __shared__ double s_mem[];
  const int nhl = nlay + 1;
  const int nv = 2 * nhl;
  double* s_grp = s_mem;
  double* s_flux = s_mem + COST_GROUPS * nv;
  double* s_term = s_flux + nv;
  const int k = blockIdx.x;
  const long long c0 = iv[k].chunk0;
  const long long c1 = (k + 1 < (int)gridDim.x) ? iv[k + 1].chunk0 : nchunks_total;
  const int tid = threadIdx.x;
  // The interval's chunks are added up in groups of 32 consecutive chunks (each group from 0.0 in chunk order), the groups
  // then in ascending order: a function of the interval's chunk count alone.  Thread group g (128 threads: one per flux
  // value) takes the chunk groups g, g + 8, ...; sixteen loads are in flight together.
  const int g = tid >> 7, t = tid & 127;
  const long long ncg = (c1 - c0 + 31) / 32;
  for (long long cg = g; cg < ncg; cg += 8) {
    const long long cb = c0 + cg * 32;
    const int cnt = (int)((c1 - cb) < 32 ? (c1 - cb) : 32);
    if (t < nv) {
      // sixteen loads in flight, unconditionally (behind the end of a short group: its last chunk again, the value replaced
      // by 0.0), added in chunk order.  (All 32 of a group in flight: 11.7 instead of 10.1 us per launch - the one CU that sums
      // an interval is bound by its own load path, not by the round trips.)
      double a = 0.0;
#pragma unroll 1
      for (int h = 0; h < 32; h += 16) {
        double q[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) q[u] = partial[(size_t)(cb + (h + u < cnt ? h + u : cnt - 1)) * nv + t];
#pragma unroll
        for (int u = 0; u < 16; ++u) a += h + u < cnt ? q[u] : 0.0;      // + 0.0 changes nothing
      }
      s_grp[(size_t)cg * nv + t] = a;
    }
  }
  __syncthreads();
  for (int v = tid; v < nv; v += 1024) {
    double a = 0.0;
    for (long long cg = 0; cg < ncg; ++cg) a += s_grp[(size_t)cg * nv + v];
    s_flux[v] = a;
  }
  __syncthreads();
  if (tid == 0) s_flux[0] = cos_sza * s_flux[0];  // flux_dn(0) = cos_sza*sum(ssi)
  __syncthreads();
  const double* s = sums + (size_t)k * ntotal;
  const double* dn = s_flux;
  const double* up = s_flux + nhl;
  for (int l = tid; l < nlay; l += 1024) {
    const double hr_fit = conv[l] * (dn[l + 1] - dn[l]);
    const double d = hr_fit - s[rH + l];
    s_term[l] = layer_weight[l] * (d * d);
  }
  __syncthreads();
  if (tid == 0) {
    double ss = 0.0;
    for (int l = 0; l < nlay; ++l) ss += s_term[l];
    const double hr_weight = 3600.0 * 24.0;
    const double dsurf = dn[nlay] - s[rFDS];
    const double dtoa = up[0] - s[rFUT];
    err[k] = sqrt(hr_weight * hr_weight * ss + flux_weight * (dsurf * dsurf + dtoa * dtoa));
  }
}

// The errors of a batch arrive in pinned, host-coherent memory, written by the last kernel of the train.  Waiting for them
// with hipStreamSynchronize costs a wake-up through the runtime per batch, and a search is hundreds of dependent batches
// (equipartition.cpp:638-805: one interval per call); the host instead marks the slots as pending and watches them.
// A pattern no result can have: the kernels end in sqrt(), whose only NaN is the canonical quiet one.
constexpr unsigned long long kPendingBits = 0x7ff4dead5eed0001ULL;

void mark_pending(double* h_slots, int count) {
  volatile unsigned long long* s = reinterpret_cast<volatile unsigned long long*>(h_slots);
  for (int k = 0; k < count; ++k) s[k] = kPendingBits;
  std::atomic_thread_fence(std::memory_order_release);
}

int wait_for_slots(hipStream_t stream, const double* h_slots, int count) {
  static const bool no_poll = std::getenv("ECCKD_NO_POLL") != nullptr;   // A/B knob: wait through the runtime
  if (no_poll) {
    ECCKD_HIP_CHECK(hipStreamSynchronize(stream));
    return ECCKD_OK;
  }
  const volatile unsigned long long* s = reinterpret_cast<const volatile unsigned long long*>(h_slots);
  auto all_there = [&] {
    for (int k = 0; k < count; ++k)
      if (s[k] == kPendingBits) return false;
    return true;
  };
  for (unsigned spins = 1;; ++spins) {
    if (all_there()) break;
    if ((spins & 0x3fff) == 0) {
      // now and then: has the stream drained (or died) without delivering?  An idle stream has made all its writes visible.
      const hipError_t q = hipStreamQuery(stream);
      if (q == hipSuccess) {
        if (all_there()) break;
        return ecckd::fail(ECCKD_UNEXPECTED_EXCEPTION, "interval errors were not delivered by the device");
      }
      if (q != hipErrorNotReady)
        return ecckd::fail(ECCKD_UNEXPECTED_EXCEPTION, "device failure while waiting for interval errors: %s", hipGetErrorString(q));
    }
    if ((spins & 0x3f) == 0 && ecckd::host_oversubscribed()) std::this_thread::yield(); else _mm_pause();
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return ECCKD_OK;
}

// Chunk size of an interval of `len` points: the smallest multiple of `gran` that covers it with at most `blocks` chunks.
long long interval_chunk_pts(long long len, long long blocks, long long gran) {
  long long c = (len + blocks - 1) / blocks;
  c = (c + gran - 1) / gran * gran;
  return c < gran ? gran : c;
}

int gas_ensure_work(ecckd_gas* g, size_t dev_bytes, size_t pinned_bytes) {
  if (dev_bytes > g->work_bytes) {
    if (g->work) {
      ECCKD_HIP_CHECK(hipStreamSynchronize(g->eval_stream()));
      ecckd::dev_release(g->ctx, g->work);
      g->work = nullptr;
      g->work_bytes = 0;
    }
    size_t want = ecckd_align_up(dev_bytes * 2, 1 << 20);
    ECCKD_HIP_CHECK(ecckd::dev_malloc(g->ctx, &g->work, want));
    g->work_bytes = want;
  }
  // the pinned staging area belongs to the context (one caller per context at a time) or to the lane the gas was lent
  if (g->lane) {
    ECCKD_CHECK(ecckd::lane_ensure_pinned(g->lane, pinned_bytes));
    g->pinned = g->lane->pinned;
    g->pinned_bytes = g->lane->pinned_bytes;
    return ECCKD_OK;
  }
  ECCKD_CHECK(ecckd::ensure_pinned(g->ctx, pinned_bytes));
  g->pinned = g->ctx->pinned;
  g->pinned_bytes = g->ctx->pinned_bytes;
  return ECCKD_OK;
}

void gas_free(ecckd_gas* g) {
  if (!g) return;
  ecckd_ctx* ctx = g->ctx;
  if (ctx) (void)hipStreamSynchronize(ctx->stream);
  auto fr = [ctx](void* p) { if (p) ecckd::dev_release(ctx, p); };
  if (g->owns_planck) fr(g->planck_hl);
  fr(g->ssi); fr(g->tf); fr(g->tg); fr(g->hr_low); fr(g->hr_high); fr(g->fx);
  fr(g->bg_od); fr(g->bg_pair); fr(g->w1); fr(g->w2); fr(g->cnt); fr(g->hr); fr(g->fds); fr(g->fut);
  fr(g->wn_sorted); fr(g->dwn_sorted); fr(g->ireorder); fr((void*)g->rows); fr(g->tile_sums); fr(g->super_sums);
  fr(g->lev); fr(g->work);
  delete g;
}

// The DOUBLE rows of the background optical depths.  A longwave gas whose background is FLOAT holds the FLOAT pairs only
// (ecckd_gas_create_lw); the rows are made from them when something asks: ecckd_gas_view, the run-time-nlay sweep.
static int gas_bg_rows(ecckd_gas* g) {
  if (g->bg_od || !g->bg_pair) return ECCKD_OK;
  ecckd_ctx* ctx = g->ctx;
  const hipError_t e = ecckd::dev_malloc(ctx, (void**)&g->bg_od, (size_t)g->nlay * g->n * sizeof(double));
  if (e != hipSuccess) return ecckd::fail(e == hipErrorOutOfMemory ? ECCKD_OUT_OF_MEMORY : ECCKD_UNEXPECTED_EXCEPTION,
                                          "background rows: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(k_unpack_bg32, dim3((unsigned)((g->n + 255) / 256)), dim3(256), 0, ctx->stream, g->nlay / 2, g->n,
                     (const float_x2_store*)g->bg_pair, g->bg_od);
  ECCKD_HIP_CHECK(hipGetLastError());
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return ECCKD_OK;
}

}  // namespace

extern "C" {

int ecckd_gas_create_lw(ecckd_ctx* ctx, int nlay, size_t nwav, const double* h_pressure_hl,
                        const double* h_temperature_hl, const double* d_wavenumber,
                        const double* d_d_wavenumber, const int32_t* d_rank, const void* d_bg_od,
                        int bg_type, const void* d_od, int od_type, size_t src_stride,
                        int averaging_method, double flux_weight, double min_pressure,
                        const double* d_planck_hl_reuse, ecckd_gas** out) {
  ECCKD_REQUIRE(ctx && out, "ecckd_gas_create_lw: NULL ctx/out");
  *out = nullptr;
  ECCKD_REQUIRE(nlay > 0 && nwav > 0, "ecckd_gas_create_lw: empty problem (nlay=%d, nwav=%zu)", nlay, nwav);
  ECCKD_REQUIRE(nwav < (size_t)0x7fffffff, "ecckd_gas_create_lw: nwav exceeds int32 rank range");
  ECCKD_REQUIRE(h_pressure_hl && h_temperature_hl && d_wavenumber && d_d_wavenumber && d_rank && d_od,
                "ecckd_gas_create_lw: NULL array argument");
  ECCKD_REQUIRE(od_type == ECCKD_F32 || od_type == ECCKD_F64, "ecckd_gas_create_lw: od_type must be 4 or 8");
  ECCKD_REQUIRE(!d_bg_od || bg_type == ECCKD_F32 || bg_type == ECCKD_F64, "ecckd_gas_create_lw: bg_type must be 4 or 8");
  ECCKD_REQUIRE(src_stride >= nwav, "ecckd_gas_create_lw: src_stride < nwav");
  // find_g_points.cpp:1146-1149: unknown averaging method is a PARAMETER_ERROR;
  // total-transmission is shortwave-only (fit_optical_depth_lw has no such branch, :101-104)
  ECCKD_REQUIRE(averaging_method >= ECCKD_AVG_LINEAR && averaging_method <= ECCKD_AVG_LOGARITHMIC,
                "Averaging method %d not understood", averaging_method);
  for (int i = 0; i <= nlay; ++i)
    ECCKD_REQUIRE(h_pressure_hl[i] > 0.0 && h_temperature_hl[i] > 0.0 && (i == 0 || h_pressure_hl[i] > h_pressure_hl[i - 1]),
                  "ecckd_gas_create_lw: pressure_hl must be positive and increasing, temperature_hl positive");
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  if ((size_t)nlay * 64 * sizeof(double) > 160 * 1024)
    return ecckd::fail(ECCKD_PARAMETER_ERROR, "ecckd_gas_create_lw: nlay = %d exceeds the supported maximum (320)", nlay);

  ecckd_gas* g = new ecckd_gas();
  g->ctx = ctx;
  g->do_sw = 0;
  g->method = averaging_method;
  g->nlay = nlay;
  g->n = nwav;
  g->flux_weight = flux_weight;
  g->h_pressure_hl.assign(h_pressure_hl, h_pressure_hl + nlay + 1);
  const bool is_log = averaging_method == ECCKD_AVG_LOGARITHMIC;
  const size_t nhl = nlay + 1;
  int rc = ECCKD_OK;
#define GTRY(expr)                                                                       \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      rc = ecckd::fail(_e == hipErrorOutOfMemory ? ECCKD_OUT_OF_MEMORY : ECCKD_UNEXPECTED_EXCEPTION, \
                       "%s failed: %s", #expr, hipGetErrorString(_e));                   \
      gas_free(g);                                                                       \
      return rc;                                                                         \
    }                                                                                    \
  } while (0)
  const size_t mat = (size_t)nlay * nwav * sizeof(double);
  if (d_planck_hl_reuse) {
    g->planck_hl = const_cast<double*>(d_planck_hl_reuse);
    g->owns_planck = false;
  } else {
    GTRY(ecckd::dev_malloc(ctx, (void**)&g->planck_hl, nhl * nwav * sizeof(double)));
  }
  // The background rows.  FLOAT pairs for the sweep (k_rt_lw_bb_mirror<.., true>) whenever every value is a float: with a
  // FLOAT background (or none) on the 54-layer path K4 writes them itself INSTEAD of the DOUBLE rows (which are then made on
  // demand only, gas_bg_rows: 3.1 GB less to write and to hold per gas at 7.2e6 points); otherwise the DOUBLE rows are
  // written and k_pack_bg32 tries to pack them below.  ECCKD_BG64: DOUBLE rows only.
  const bool want_pairs = (nlay == 54 || nlay == 30) && std::getenv("ECCKD_BG64") == nullptr;
  const bool pairs_in_k4 = want_pairs && nlay == 54 && od_type == ECCKD_F32 && !is_log && (!d_bg_od || bg_type == ECCKD_F32);
  if (want_pairs) GTRY(ecckd::dev_malloc(ctx, (void**)&g->bg_pair, (size_t)(nlay / 2) * nwav * 2 * sizeof(float)));
  if (!pairs_in_k4) GTRY(ecckd::dev_malloc(ctx, (void**)&g->bg_od, mat));
  GTRY(ecckd::dev_malloc(ctx, (void**)&g->w1, mat));
  if (is_log) {
    GTRY(ecckd::dev_malloc(ctx, (void**)&g->w2, mat));
    GTRY(ecckd::dev_malloc(ctx, (void**)&g->cnt, mat));
  }
  GTRY(ecckd::dev_malloc(ctx, (void**)&g->hr, mat));
  GTRY(ecckd::dev_malloc(ctx, (void**)&g->fds, nwav * sizeof(double)));
  GTRY(ecckd::dev_malloc(ctx, (void**)&g->fut, nwav * sizeof(double)));
  GTRY(ecckd::dev_malloc(ctx, (void**)&g->wn_sorted, nwav * sizeof(double)));
  GTRY(ecckd::dev_malloc(ctx, (void**)&g->dwn_sorted, nwav * sizeof(double)));
  GTRY(ecckd::dev_malloc(ctx, (void**)&g->ireorder, nwav * sizeof(int32_t)));

  // per-level constants: hk[nhl] | conv[nlay] | layer_weight[nlay] | flag
  std::vector<double> lev(nhl + 2 * nlay + 1, 0.0);
  const double hk = 6.62606896e-34 / 1.3806504e-23;
  for (size_t i = 0; i < nhl; ++i) lev[i] = hk / h_temperature_hl[i];
  g->h_layer_weight.resize(nlay);
  {
    // find_g_points.cpp:1093-1099
    double s = 0.0;
    for (int l = 0; l < nlay; ++l) {
      lev[nhl + l] = -(ECCKD_ACCEL_GRAVITY / ECCKD_SPECIFIC_HEAT_AIR) / (h_pressure_hl[l + 1] - h_pressure_hl[l]);
      double lw = std::sqrt(h_pressure_hl[l + 1]) - std::sqrt(h_pressure_hl[l]);
      double pfl = 0.5 * (h_pressure_hl[l + 1] + h_pressure_hl[l]);
      if (pfl < min_pressure) lw = 0.0;
      g->h_layer_weight[l] = lw;
    }
    for (int l = 0; l < nlay; ++l) s += g->h_layer_weight[l];
    for (int l = 0; l < nlay; ++l) {
      g->h_layer_weight[l] /= s;
      lev[nhl + nlay + l] = g->h_layer_weight[l];
    }
  }
  GTRY(ecckd::dev_malloc(ctx, (void**)&g->lev, lev.size() * sizeof(double)));
  GTRY(hipMemcpyAsync(g->lev, lev.data(), lev.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  GTRY(hipStreamSynchronize(ctx->stream));
  int* d_flag = (int*)(g->lev + nhl + 2 * nlay);

  const unsigned eblocks = (unsigned)((nwav + 255) / 256);
  // rank must be a permutation: every slot of ireorder written exactly once.  Checked on the
  // device BEFORE the gather kernel dereferences ireorder.
  GTRY(hipMemsetAsync(g->ireorder, 0xFF, nwav * sizeof(int32_t), ctx->stream));
  hipLaunchKernelGGL(k_invert_rank, dim3(eblocks), dim3(256), 0, ctx->stream, nwav, d_rank, g->ireorder, d_flag);
  hipLaunchKernelGGL(k_check_perm, dim3(eblocks), dim3(256), 0, ctx->stream, nwav, g->ireorder, d_flag);
  {
    int flag0 = 0;
    GTRY(hipMemcpyAsync(&flag0, d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    GTRY(hipStreamSynchronize(ctx->stream));
    if (flag0) {
      gas_free(g);
      return ecckd::fail(ECCKD_PARAMETER_ERROR, "ecckd_gas_create_lw: rank is not a permutation of 0..nwav-1");
    }
  }

  int threads = PREP_THREADS;
  while ((size_t)nlay * threads * sizeof(double) > 160 * 1024 && threads > 64) threads /= 2;
  const size_t lds = (size_t)nlay * threads * sizeof(double);
  const unsigned pblocks = (unsigned)((nwav + threads - 1) / threads);
  const double* hkd = g->lev;
  const double* convd = g->lev + nhl;
#define LAUNCH_PREP(BG, OD)                                                                                   \
  do {                                                                                                        \
    GTRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gas_prep_lw<BG, OD>),                            \
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                        \
    hipLaunchKernelGGL((k_gas_prep_lw<BG, OD>), dim3(pblocks), dim3(threads), lds, ctx->stream, nlay, nwav,   \
                       src_stride, averaging_method, g->ireorder, hkd, convd, d_wavenumber, d_d_wavenumber,   \
                       (const BG*)d_bg_od, (const OD*)d_od, d_planck_hl_reuse, g->wn_sorted, g->dwn_sorted,   \
                       g->planck_hl, g->bg_od, g->w1, g->w2, g->cnt, g->hr, g->fds, g->fut);                  \
  } while (0)
  const bool bg32 = d_bg_od && bg_type == ECCKD_F32;
  // fast path: 54 layers, FLOAT target spectrum (as stored in the CKDMIP files), FLOAT or DOUBLE (merged) background,
  // with or without the Planck matrix of an earlier gas, no log metric
  const bool fast = nlay == 54 && od_type == ECCKD_F32 && !is_log;
  // temporaries of the fast path; whatever is still held when the function returns (error paths included) goes back to
  // the context's cache once the stream has drained
  struct Temps {
    ecckd_ctx* ctx;
    void* p[3] = {nullptr, nullptr, nullptr};
    void drop(void*& q) {
      for (void*& r : p)
        if (r && r == q) { ecckd::dev_release(ctx, r); r = nullptr; }
      q = nullptr;
    }
    ~Temps() {
      bool any = false;
      for (void* r : p) any = any || r;
      if (!any) return;
      (void)hipStreamSynchronize(ctx->stream);
      for (void* r : p) if (r) ecckd::dev_release(ctx, r);
    }
  } temps{ctx};
  void*& od_col_v = temps.p[0];
  void*& bg_col = temps.p[1];
  void*& wave_part_v = temps.p[2];   // per-wave row sums left by K4, combined into the tile sums below
  size_t nw64 = 0;
  if (fast) {
    const size_t bg_elem = bg32 ? sizeof(float) : sizeof(double);
    nw64 = (nwav + 63) / 64;
    const size_t ncol = nw64 * 64;             // the columns in two runs of 27 layers, [2][ncol][27], staged through LDS by K4
    GTRY(ecckd::dev_malloc(ctx, &od_col_v, ncol * 54 * sizeof(float)));
    if (d_bg_od) GTRY(ecckd::dev_malloc(ctx, &bg_col, ncol * 54 * bg_elem));
    float* od_col = (float*)od_col_v;
    const unsigned tblocks = (unsigned)((nwav + 63) / 64);
    hipLaunchKernelGGL((k_scatter_column_halves<54, float, 2>), dim3(tblocks), dim3(256), 0, ctx->stream, nwav, ncol, src_stride,
                       d_rank, (const float*)d_od, od_col);
    if (d_bg_od && bg32)
      hipLaunchKernelGGL((k_scatter_column_halves<54, float, 2>), dim3(tblocks), dim3(256), 0, ctx->stream, nwav, ncol, src_stride,
                         d_rank, (const float*)d_bg_od, (float*)bg_col);
    else if (d_bg_od)
      hipLaunchKernelGGL((k_scatter_column_halves<54, double, 2>), dim3(tblocks), dim3(256), 0, ctx->stream, nwav, ncol, src_stride,
                         d_rank, (const double*)d_bg_od, (double*)bg_col);
    const unsigned fblocks = (unsigned)((nwav + 127) / 128);
    GTRY(ecckd::dev_malloc(ctx, &wave_part_v, (size_t)(3 * 54 + 2) * nw64 * sizeof(double)));
    double* wave_part = (double*)wave_part_v;
#define LAUNCH_MIRROR(BG, REUSE)                                                                                              \
  hipLaunchKernelGGL((k_gas_prep_lw_mirror<54, BG, float, REUSE>), dim3(fblocks), dim3(PREP_THREADS), 0, ctx->stream, nwav,    \
                     averaging_method, g->ireorder, hkd, convd, d_wavenumber, d_d_wavenumber, (const BG*)bg_col,               \
                     (const float*)od_col, d_planck_hl_reuse, g->wn_sorted, g->dwn_sorted, g->planck_hl, g->bg_od, g->w1,      \
                     g->hr, g->fds, g->fut, wave_part, nw64, pairs_in_k4 ? (float_x2_store*)g->bg_pair : nullptr)
    if (bg32 || !d_bg_od) { if (d_planck_hl_reuse) LAUNCH_MIRROR(float, true); else LAUNCH_MIRROR(float, false); }
    else { if (d_planck_hl_reuse) LAUNCH_MIRROR(double, true); else LAUNCH_MIRROR(double, false); }
#undef LAUNCH_MIRROR
    GTRY(hipGetLastError());
    GTRY(hipStreamSynchronize(ctx->stream));
    temps.drop(od_col_v);
    temps.drop(bg_col);
  } else if (bg32 && od_type == ECCKD_F32) LAUNCH_PREP(float, float);
  else if (bg32) LAUNCH_PREP(float, double);
  else if (od_type == ECCKD_F32) LAUNCH_PREP(double, float);
  else LAUNCH_PREP(double, double);
#undef LAUNCH_PREP
  GTRY(hipGetLastError());

  // row table + tile sums
  RowMap R;
  R.A = 0;
  R.B = nlay;
  if (is_log) R.N = 2 * nlay;
  R.H = (is_log ? 3 : 2) * nlay;
  R.FDS = R.H + nlay;
  R.FUT = R.FDS + 1;
  R.total = R.FUT + 1;
  g->rm = R;
  g->nrows = R.total;
  std::vector<const double*> rows(g->nrows);
  for (int l = 0; l < nlay; ++l) {
    rows[R.A + l] = g->w1 + (size_t)l * nwav;
    // denominator of the fit: planck_hl(l+1) (find_g_points.cpp:62), or for the
    // logarithmic method planck_hl(l) masked by metric > 0 (:87)
    rows[R.B + l] = is_log ? g->w2 + (size_t)l * nwav : g->planck_hl + (size_t)(l + 1) * nwav;
    if (is_log) rows[R.N + l] = g->cnt + (size_t)l * nwav;
    rows[R.H + l] = g->hr + (size_t)l * nwav;
  }
  rows[R.FDS] = g->fds;
  rows[R.FUT] = g->fut;
  GTRY(ecckd::dev_malloc(ctx, (void**)&g->rows, rows.size() * sizeof(double*)));
  GTRY(hipMemcpyAsync((void*)g->rows, rows.data(), rows.size() * sizeof(double*), hipMemcpyHostToDevice, ctx->stream));
  GTRY(hipStreamSynchronize(ctx->stream));
  g->ntiles = (nwav + TILE - 1) / TILE;
  GTRY(ecckd::dev_malloc(ctx, (void**)&g->tile_sums, (size_t)g->nrows * g->ntiles * sizeof(double)));
  if (wave_part_v) {
    static_assert(TILE == 256, "a tile is four 64-point groups");
    hipLaunchKernelGGL(k_combine_wave_sums, dim3((unsigned)((g->ntiles + 255) / 256), (unsigned)g->nrows), dim3(256), 0, ctx->stream,
                       g->nrows, nw64, g->ntiles, (const double*)wave_part_v, g->tile_sums);
    GTRY(hipGetLastError());
    GTRY(hipStreamSynchronize(ctx->stream));
    temps.drop(wave_part_v);
  } else {
    hipLaunchKernelGGL(k_tile_sums, dim3((unsigned)g->ntiles), dim3(TILE), 0, ctx->stream, g->nrows, nwav, g->ntiles,
                       (const double* const*)g->rows, g->tile_sums);
  }
  GTRY(hipGetLastError());
  g->nsuper = (g->ntiles + SUPER - 1) / SUPER;
  GTRY(ecckd::dev_malloc(ctx, (void**)&g->super_sums, (size_t)g->nrows * g->nsuper * sizeof(double)));
  hipLaunchKernelGGL(k_super_sums, dim3((unsigned)g->nsuper, (unsigned)g->nrows), dim3(256), 0, ctx->stream, g->ntiles, g->nsuper,
                     (const double*)g->tile_sums, g->super_sums);
  GTRY(hipGetLastError());
  // FLOAT pairs from the DOUBLE rows where K4 has not written them itself: kept if every value is a float
  if (want_pairs && !pairs_in_k4) {
    hipLaunchKernelGGL(k_pack_bg32, dim3(eblocks), dim3(256), 0, ctx->stream, nlay / 2, nwav, (const double*)g->bg_od,
                       (float_x2_store*)g->bg_pair, d_flag + 1);
    GTRY(hipGetLastError());
  }
  int flag[2] = {0, 0};
  GTRY(hipMemcpyAsync(flag, d_flag, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  GTRY(hipStreamSynchronize(ctx->stream));
#undef GTRY
  if (flag[0]) {
    gas_free(g);
    return ecckd::fail(ECCKD_PARAMETER_ERROR, "ecckd_gas_create_lw: rank is not a permutation of 0..nwav-1");
  }
  if (want_pairs && !pairs_in_k4 && flag[1]) {          // a DOUBLE background with values between the floats: the DOUBLE rows serve
    ecckd::dev_release(ctx, g->bg_pair);
    g->bg_pair = nullptr;
  }
  *out = g;
  return ECCKD_OK;
}


// planck_hl[nlay+1][nwav] in the order given by d_rank, bit-identical to the matrix ecckd_gas_create_lw builds for a gas
// with that ordering and temperature profile (planck_function.cpp:22-54 on the reordered grid, find_g_points.cpp:970-979).
int ecckd_planck_hl_sorted_dev(ecckd_ctx* ctx, int nlay, size_t nwav, const double* h_temperature_hl,
                               const double* d_wavenumber, const double* d_d_wavenumber, const int32_t* d_rank,
                               double* d_planck_hl) {
  ECCKD_REQUIRE(ctx && h_temperature_hl && d_wavenumber && d_d_wavenumber && d_rank && d_planck_hl,
                "ecckd_planck_hl_sorted_dev: NULL argument");
  ECCKD_REQUIRE(nlay > 0 && nwav > 0 && nwav < (size_t)0x7fffffff, "ecckd_planck_hl_sorted_dev: bad size (nlay=%d, nwav=%zu)", nlay, nwav);
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  const int nhl = nlay + 1;
  std::vector<double> hk(nhl);
  for (int i = 0; i < nhl; ++i) {
    ECCKD_REQUIRE(h_temperature_hl[i] > 0.0, "ecckd_planck_hl_sorted_dev: temperature_hl must be positive");
    hk[i] = (6.62606896e-34 / 1.3806504e-23) / h_temperature_hl[i];
  }
  const size_t hk_bytes = ecckd_align_up(nhl * sizeof(double), 256), flag_bytes = 256;
  ECCKD_CHECK(ecckd::ensure_scratch(ctx, hk_bytes + flag_bytes + nwav * sizeof(int32_t)));
  double* d_hk = (double*)ctx->scratch;
  int* d_flag = (int*)((char*)ctx->scratch + hk_bytes);
  int32_t* d_ireorder = (int32_t*)((char*)ctx->scratch + hk_bytes + flag_bytes);
  ECCKD_HIP_CHECK(hipMemcpyAsync(d_hk, hk.data(), nhl * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ECCKD_HIP_CHECK(hipMemsetAsync(d_flag, 0, sizeof(int), ctx->stream));
  ECCKD_HIP_CHECK(hipMemsetAsync(d_ireorder, 0xFF, nwav * sizeof(int32_t), ctx->stream));
  const unsigned eblocks = (unsigned)((nwav + 255) / 256);
  hipLaunchKernelGGL(k_invert_rank, dim3(eblocks), dim3(256), 0, ctx->stream, nwav, d_rank, d_ireorder, d_flag);
  hipLaunchKernelGGL(k_check_perm, dim3(eblocks), dim3(256), 0, ctx->stream, nwav, d_ireorder, d_flag);
  int flag = 0;
  ECCKD_HIP_CHECK(hipMemcpyAsync(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));   // also: hk is a local
  ECCKD_REQUIRE(flag == 0, "ecckd_planck_hl_sorted_dev: rank is not a permutation of 0..nwav-1");
  hipLaunchKernelGGL(k_planck_sorted, dim3(eblocks), dim3(256), 0, ctx->stream, nhl, nwav, d_ireorder, d_hk, d_wavenumber,
                     d_d_wavenumber, d_planck_hl);
  ECCKD_HIP_CHECK(hipGetLastError());
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return ECCKD_OK;
}

// Shortwave gas (find_g_points.cpp do_sw branches of :872-1150).  d_albedo: per-wavenumber
// surface albedo in ORIGINAL order (:921-923), used only for the up-welling of the two scaled
// truth fields of the total-transmission method; NULL = direct beam only (:1027-1034).
// min_scaling / max_scaling are the values AFTER the clamps of :666-667.
int ecckd_gas_create_sw(ecckd_ctx* ctx, int nlay, size_t nwav, const double* h_pressure_hl,
                        const double* d_ssi, const double* d_albedo, const int32_t* d_rank,
                        const void* d_bg_od, int bg_type, const void* d_od, int od_type,
                        size_t src_stride, int averaging_method, double flux_weight,
                        double min_pressure, double cos_sza, double min_scaling, double max_scaling,
                        ecckd_gas** out) {
  ECCKD_REQUIRE(ctx && out, "ecckd_gas_create_sw: NULL ctx/out");
  *out = nullptr;
  ECCKD_REQUIRE(nlay > 0 && nwav > 0, "ecckd_gas_create_sw: empty problem (nlay=%d, nwav=%zu)", nlay, nwav);
  ECCKD_REQUIRE(nwav < (size_t)0x7fffffff, "ecckd_gas_create_sw: nwav exceeds int32 rank range");
  ECCKD_REQUIRE(h_pressure_hl && d_ssi && d_rank && d_od, "ecckd_gas_create_sw: NULL array argument");
  ECCKD_REQUIRE(od_type == ECCKD_F32 || od_type == ECCKD_F64, "ecckd_gas_create_sw: od_type must be 4 or 8");
  ECCKD_REQUIRE(!d_bg_od || bg_type == ECCKD_F32 || bg_type == ECCKD_F64, "ecckd_gas_create_sw: bg_type must be 4 or 8");
  ECCKD_REQUIRE(src_stride >= nwav, "ecckd_gas_create_sw: src_stride < nwav");
  ECCKD_REQUIRE(averaging_method >= ECCKD_AVG_LINEAR && averaging_method <= ECCKD_AVG_TOTAL_TRANSMISSION,
                "Averaging method %d not understood", averaging_method);
  ECCKD_REQUIRE(cos_sza > 0.0, "ecckd_gas_create_sw: cos_sza must be positive");
  for (int i = 0; i <= nlay; ++i)
    ECCKD_REQUIRE(h_pressure_hl[i] > 0.0 && (i == 0 || h_pressure_hl[i] > h_pressure_hl[i - 1]),
                  "ecckd_gas_create_sw: pressure_hl must be positive and increasing");
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));

  ecckd_gas* g = new ecckd_gas();
  g->ctx = ctx;
  g->do_sw = 1;
  g->method = averaging_method;
  g->nlay = nlay;
  g->n = nwav;
  g->flux_weight = flux_weight;
  g->cos_sza = cos_sza;
  g->min_scaling = min_scaling;
  g->max_scaling = max_scaling;
  g->owns_planck = true;
  g->h_pressure_hl.assign(h_pressure_hl, h_pressure_hl + nlay + 1);
  const bool is_log = averaging_method == ECCKD_AVG_LOGARITHMIC;
  const bool is_tt = averaging_method == ECCKD_AVG_TOTAL_TRANSMISSION;
  const size_t nhl = nlay + 1;
  int rc = ECCKD_OK;
#define GTRY(expr)                                                                       \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      rc = ecckd::fail(_e == hipErrorOutOfMemory ? ECCKD_OUT_OF_MEMORY : ECCKD_UNEXPECTED_EXCEPTION, \
                       "%s failed: %s", #expr, hipGetErrorString(_e));                   \
      gas_free(g);                                                                       \
      return rc;                                                                         \
    }                                                                                    \
  } while (0)
  const size_t mat = (size_t)nlay * nwav * sizeof(double);
  GTRY(ecckd::dev_malloc(ctx, (void**)&g->ssi, nwav * sizeof(double)));
  GTRY(ecckd::dev_malloc(ctx, (void**)&g->bg_od, mat));
  GTRY(ecckd::dev_malloc(ctx, (void**)&g->w1, mat));
  if (is_log) {
    GTRY(ecckd::dev_malloc(ctx, (void**)&g->w2, mat));
    GTRY(ecckd::dev_malloc(ctx, (void**)&g->cnt, mat));
  }
  GTRY(ecckd::dev_malloc(ctx, (void**)&g->hr, mat));
  GTRY(ecckd::dev_malloc(ctx, (void**)&g->fds, nwav * sizeof(double)));
  GTRY(ecckd::dev_malloc(ctx, (void**)&g->fut, nwav * sizeof(double)));
  if (is_tt) {
    GTRY(ecckd::dev_malloc(ctx, (void**)&g->tf, mat));
    GTRY(ecckd::dev_malloc(ctx, (void**)&g->tg, mat));
    GTRY(ecckd::dev_malloc(ctx, (void**)&g->hr_low, mat));
    GTRY(ecckd::dev_malloc(ctx, (void**)&g->hr_high, mat));
    GTRY(ecckd::dev_malloc(ctx, (void**)&g->fx, 4 * nwav * sizeof(double)));
  }
  GTRY(ecckd::dev_malloc(ctx, (void**)&g->ireorder, nwav * sizeof(int32_t)));

  // per-level constants: (unused hk)[nhl] | conv[nlay] | layer_weight[nlay] | flag
  std::vector<double> lev(nhl + 2 * nlay + 1, 0.0);
  g->h_layer_weight.resize(nlay);
  {
    double sw = 0.0;
    for (int l = 0; l < nlay; ++l) {
      lev[nhl + l] = -(ECCKD_ACCEL_GRAVITY / ECCKD_SPECIFIC_HEAT_AIR) / (h_pressure_hl[l + 1] - h_pressure_hl[l]);
      double lw = std::sqrt(h_pressure_hl[l + 1]) - std::sqrt(h_pressure_hl[l]);
      double pfl = 0.5 * (h_pressure_hl[l + 1] + h_pressure_hl[l]);
      if (pfl < min_pressure) lw = 0.0;
      g->h_layer_weight[l] = lw;
    }
    for (int l = 0; l < nlay; ++l) sw += g->h_layer_weight[l];
    for (int l = 0; l < nlay; ++l) {
      g->h_layer_weight[l] /= sw;
      lev[nhl + nlay + l] = g->h_layer_weight[l];
    }
  }
  GTRY(ecckd::dev_malloc(ctx, (void**)&g->lev, lev.size() * sizeof(double)));
  GTRY(hipMemcpyAsync(g->lev, lev.data(), lev.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  GTRY(hipStreamSynchronize(ctx->stream));
  int* d_flag = (int*)(g->lev + nhl + 2 * nlay);

  const unsigned eblocks = (unsigned)((nwav + 255) / 256);
  GTRY(hipMemsetAsync(g->ireorder, 0xFF, nwav * sizeof(int32_t), ctx->stream));
  hipLaunchKernelGGL(k_invert_rank, dim3(eblocks), dim3(256), 0, ctx->stream, nwav, d_rank, g->ireorder, d_flag);
  hipLaunchKernelGGL(k_check_perm, dim3(eblocks), dim3(256), 0, ctx->stream, nwav, g->ireorder, d_flag);
  {
    int flag0 = 0;
    GTRY(hipMemcpyAsync(&flag0, d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    GTRY(hipStreamSynchronize(ctx->stream));
    if (flag0) {
      gas_free(g);
      return ecckd::fail(ECCKD_PARAMETER_ERROR, "ecckd_gas_create_sw: rank is not a permutation of 0..nwav-1");
    }
  }
  RowMap R;
  int next = 0;
  R.A = next; next += nlay;
  R.B = next; next += nlay;   // rows of ssi (masked by metric > 0 for the logarithmic method)
  if (is_log) { R.N = next; next += nlay; }
  R.H = next; next += nlay;
  R.FDS = next++;
  R.FUT = next++;
  if (is_tt) {
    R.TF = next; next += nlay;
    R.TG = next; next += nlay;
    R.HL = next; next += nlay;
    R.HH = next; next += nlay;
    R.FDSL = next++; R.FUTL = next++; R.FDSH = next++; R.FUTH = next++;
  }
  R.total = next;
  g->rm = R;
  g->nrows = R.total;

  const unsigned pblocks = (unsigned)((nwav + PREP_THREADS - 1) / PREP_THREADS);
  const bool bg32 = d_bg_od && bg_type == ECCKD_F32;
  // 54 layers of FLOAT optical depths (the CKDMIP spectra): the columns are scattered into rank order first, staged through
  // LDS by the preparation, and the row sums are taken from registers (ECCKD_SW_PREP_GATHER: the general, gathering kernel)
  const bool cols = nlay == 54 && od_type == ECCKD_F32 && std::getenv("ECCKD_SW_PREP_GATHER") == nullptr;
  void *od_col = nullptr, *bg_col = nullptr, *wave_part = nullptr;
  auto drop_temps = [&]() {
    if (od_col) ecckd::dev_release(ctx, od_col);
    if (bg_col) ecckd::dev_release(ctx, bg_col);
    if (wave_part) ecckd::dev_release(ctx, wave_part);
    od_col = bg_col = wave_part = nullptr;
  };
  const size_t nw64 = (nwav + 63) / 64;
#define LAUNCH_PREP_SW(BG, OD, BGP, ODP)                                                                          \
  hipLaunchKernelGGL((k_gas_prep_sw<BG, OD>), dim3(pblocks), dim3(PREP_THREADS), 0, ctx->stream, nlay, nwav,       \
                     src_stride, averaging_method, cos_sza, min_scaling, max_scaling, g->ireorder,                 \
                     g->lev + nhl, d_ssi, d_albedo, (const BG*)(BGP), (const OD*)(ODP), g->ssi, g->bg_od,          \
                     g->w1, g->w2, g->cnt, g->hr, g->fds, g->fut, g->tf, g->tg, g->hr_low, g->hr_high, g->fx)
  if (cols) {
    const unsigned tblocks = (unsigned)((nwav + 63) / 64);
    const size_t npad = nw64 * 64;          // the columns in runs of 18 layers, [3][npad][18], staged through LDS by the kernel
    int rc2 = ecckd::dev_malloc(ctx, &od_col, npad * 54 * sizeof(float));
    if (rc2 == ECCKD_OK && d_bg_od) rc2 = ecckd::dev_malloc(ctx, &bg_col, npad * 54 * (bg32 ? sizeof(float) : sizeof(double)));
    if (rc2 == ECCKD_OK) rc2 = ecckd::dev_malloc(ctx, &wave_part, (size_t)g->nrows * nw64 * sizeof(double));
    if (rc2 != ECCKD_OK) { drop_temps(); gas_free(g); return rc2; }
    hipLaunchKernelGGL((k_scatter_column_halves<54, float, SW_STAGE_PARTS>), dim3(tblocks), dim3(256), 0, ctx->stream, nwav, npad, src_stride, d_rank,
                       (const float*)d_od, (float*)od_col);
    if (d_bg_od && bg32)
      hipLaunchKernelGGL((k_scatter_column_halves<54, float, SW_STAGE_PARTS>), dim3(tblocks), dim3(256), 0, ctx->stream, nwav, npad, src_stride,
                         d_rank, (const float*)d_bg_od, (float*)bg_col);
    else if (d_bg_od)
      hipLaunchKernelGGL((k_scatter_column_halves<54, double, SW_STAGE_PARTS>), dim3(tblocks), dim3(256), 0, ctx->stream, nwav, npad, src_stride,
                         d_rank, (const double*)d_bg_od, (double*)bg_col);
#define LAUNCH_PREP_SW_STAGED(BG)                                                                                                     \
  hipLaunchKernelGGL((k_gas_prep_sw_staged<SW_STAGE_PARTS, BG>), dim3(pblocks), dim3(PREP_THREADS), 0, ctx->stream, nwav, npad,        \
                     averaging_method, cos_sza, min_scaling, max_scaling, g->ireorder, g->lev + nhl, d_ssi, d_albedo, (const BG*)bg_col, \
                     (const float*)od_col, g->ssi, g->bg_od, g->w1, g->w2, g->cnt, g->hr, g->fds, g->fut, g->tf, g->tg, g->hr_low,      \
                     g->hr_high, g->fx, R, (double*)wave_part, nw64)
    if (bg32 || !d_bg_od) LAUNCH_PREP_SW_STAGED(float); else LAUNCH_PREP_SW_STAGED(double);
#undef LAUNCH_PREP_SW_STAGED
  }
  else if (bg32 && od_type == ECCKD_F32) LAUNCH_PREP_SW(float, float, d_bg_od, d_od);
  else if (bg32) LAUNCH_PREP_SW(float, double, d_bg_od, d_od);
  else if (od_type == ECCKD_F32) LAUNCH_PREP_SW(double, float, d_bg_od, d_od);
  else LAUNCH_PREP_SW(double, double, d_bg_od, d_od);
#undef LAUNCH_PREP_SW
  {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { (void)hipStreamSynchronize(ctx->stream); drop_temps(); GTRY(e); }
  }

  std::vector<const double*> rows(g->nrows);
  for (int l = 0; l < nlay; ++l) {
    rows[R.A + l] = g->w1 + (size_t)l * nwav;
    rows[R.B + l] = is_log ? g->w2 + (size_t)l * nwav : g->ssi;
    if (is_log) rows[R.N + l] = g->cnt + (size_t)l * nwav;
    rows[R.H + l] = g->hr + (size_t)l * nwav;
    if (is_tt) {
      rows[R.TF + l] = g->tf + (size_t)l * nwav;
      rows[R.TG + l] = g->tg + (size_t)l * nwav;
      rows[R.HL + l] = g->hr_low + (size_t)l * nwav;
      rows[R.HH + l] = g->hr_high + (size_t)l * nwav;
    }
  }
  rows[R.FDS] = g->fds;
  rows[R.FUT] = g->fut;
  if (is_tt) {
    rows[R.FDSL] = g->fx;
    rows[R.FUTL] = g->fx + nwav;
    rows[R.FDSH] = g->fx + 2 * nwav;
    rows[R.FUTH] = g->fx + 3 * nwav;
  }
  GTRY(ecckd::dev_malloc(ctx, (void**)&g->rows, rows.size() * sizeof(double*)));
  GTRY(hipMemcpyAsync((void*)g->rows, rows.data(), rows.size() * sizeof(double*), hipMemcpyHostToDevice, ctx->stream));
  GTRY(hipStreamSynchronize(ctx->stream));
  g->ntiles = (nwav + TILE - 1) / TILE;
  {
    const int rc3 = ecckd::dev_malloc(ctx, (void**)&g->tile_sums, (size_t)g->nrows * g->ntiles * sizeof(double));
    if (rc3 != ECCKD_OK) { (void)hipStreamSynchronize(ctx->stream); drop_temps(); gas_free(g); return rc3; }
  }
  if (wave_part) {
    hipLaunchKernelGGL(k_combine_wave_sums, dim3((unsigned)((g->ntiles + 255) / 256), (unsigned)g->nrows), dim3(256), 0, ctx->stream,
                       g->nrows, nw64, g->ntiles, (const double*)wave_part, g->tile_sums);
  } else {
    hipLaunchKernelGGL(k_tile_sums, dim3((unsigned)g->ntiles), dim3(TILE), 0, ctx->stream, g->nrows, nwav, g->ntiles,
                       (const double* const*)g->rows, g->tile_sums);
  }
  {
    const hipError_t e = hipGetLastError();
    (void)hipStreamSynchronize(ctx->stream);
    drop_temps();
    GTRY(e);
  }
  g->nsuper = (g->ntiles + SUPER - 1) / SUPER;
  GTRY(ecckd::dev_malloc(ctx, (void**)&g->super_sums, (size_t)g->nrows * g->nsuper * sizeof(double)));
  hipLaunchKernelGGL(k_super_sums, dim3((unsigned)g->nsuper, (unsigned)g->nrows), dim3(256), 0, ctx->stream, g->ntiles, g->nsuper,
                     (const double*)g->tile_sums, g->super_sums);
  GTRY(hipGetLastError());
  GTRY(hipStreamSynchronize(ctx->stream));
#undef GTRY
  *out = g;
  return ECCKD_OK;
}

// band albedo of CkdEquipartition::init_sw (find_g_points.cpp:237-261, band_albedo(jband) :1169)
int ecckd_gas_set_band_albedo(ecckd_gas* gas, double surf_albedo) {
  ECCKD_REQUIRE(gas, "ecckd_gas_set_band_albedo: NULL handle");
  gas->surf_albedo = surf_albedo;
  return ECCKD_OK;
}

int ecckd_gas_destroy(ecckd_gas* gas) {
  gas_free(gas);
  return ECCKD_OK;
}

// device views of the resident, sorted arrays (for tests and for reuse of planck_hl)
int ecckd_gas_view(ecckd_gas* gas, const char* name, const double** d_ptr, size_t* rows, size_t* cols) {
  ECCKD_REQUIRE(gas && name && d_ptr, "ecckd_gas_view: NULL argument");
  const size_t nlay = gas->nlay, n = gas->n;
  size_t r = 0;
  const double* p = nullptr;
  if (!strcmp(name, "planck_hl")) { p = gas->planck_hl; r = nlay + 1; }
  else if (!strcmp(name, "bg_optical_depth")) { ECCKD_CHECK(gas_bg_rows(gas)); p = gas->bg_od; r = nlay; }
  else if (!strcmp(name, "weighted_metric")) { p = gas->w1; r = nlay; }
  else if (!strcmp(name, "hr")) { p = gas->hr; r = nlay; }
  else if (!strcmp(name, "flux_dn_surf")) { p = gas->fds; r = 1; }
  else if (!strcmp(name, "flux_up_toa")) { p = gas->fut; r = 1; }
  else if (!strcmp(name, "wavenumber")) { p = gas->wn_sorted; r = 1; }
  else if (!strcmp(name, "d_wavenumber")) { p = gas->dwn_sorted; r = 1; }
  else if (!strcmp(name, "ssi")) { p = gas->ssi; r = 1; }
  else if (!strcmp(name, "hr_low")) { p = gas->hr_low; r = nlay; }
  else if (!strcmp(name, "hr_high")) { p = gas->hr_high; r = nlay; }
  else if (!strcmp(name, "flux_extras")) { p = gas->fx; r = 4; }
  if (!p) return ecckd::fail(ECCKD_PARAMETER_ERROR, "ecckd_gas_view: no array named \"%s\"", name);
  *d_ptr = p;
  if (rows) *rows = r;
  if (cols) *cols = n;
  return ECCKD_OK;
}

int ecckd_gas_layer_weight(ecckd_gas* gas, double* h_layer_weight) {
  ECCKD_REQUIRE(gas && h_layer_weight, "ecckd_gas_layer_weight: NULL argument");
  std::memcpy(h_layer_weight, gas->h_layer_weight.data(), gas->h_layer_weight.size() * sizeof(double));
  return ECCKD_OK;
}

double ecckd_gas_comp_cost(ecckd_gas* gas, int reset) {
  if (!gas) return 0.0;
  double c = gas->total_comp_cost;
  if (reset) gas->total_comp_cost = 0.0;
  return c;
}


// The fitted grey optical depth alone: fit_optical_depth_lw / _sw / _sw_total_trans
// (find_g_points.cpp:54-106, :112-165, :171-204) for n intervals -> h_od_fit[n][nlay]
// (total-transmission: the unscaled fit).  Runs K5a + K5b only.
int ecckd_fit_optical_depth(ecckd_gas* g, size_t ibegin, size_t npoints, int n, const double* bound1,
                            const double* bound2, double* h_od_fit) {
  ECCKD_REQUIRE(g && n > 0 && bound1 && bound2 && h_od_fit, "ecckd_fit_optical_depth: bad argument");
  ECCKD_REQUIRE(npoints > 0 && ibegin + npoints <= g->n, "ecckd_fit_optical_depth: band outside the spectrum");
  ecckd_ctx* ctx = g->ctx;
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  const int nlay = g->nlay;
  std::vector<Interval> iv(n);
  for (int k = 0; k < n; ++k) {
    long long i1 = (long long)std::ceil(bound1[k] * (double)(npoints - 1));
    long long i2 = (long long)std::floor(bound2[k] * (double)(npoints - 1));
    if (i1 < 0 || i2 >= (long long)npoints || i2 + 1 < i1 || bound2[k] < bound1[k])
      return ecckd::fail(ECCKD_PROCESSING_ERROR, "ecckd_fit_optical_depth: bad bounds %.17g-%.17g", bound1[k], bound2[k]);
    if (i2 < i1) i2 = i1;
    iv[k].i1 = (long long)ibegin + i1;
    iv[k].i2 = (long long)ibegin + i2;
    iv[k].chunk0 = k;
    iv[k].chunk_pts = 0;   // no sweep in this call
    iv[k].npoints = (long long)npoints;
    iv[k].albedo = 0.0;
  }
  const size_t iv_bytes = ecckd_align_up((size_t)n * sizeof(Interval), 256);
  const size_t sums_bytes = ecckd_align_up((size_t)n * g->nrows * sizeof(double), 256);
  const size_t fit_bytes = ecckd_align_up((size_t)2 * n * nlay * sizeof(double), 256);
  ECCKD_CHECK(gas_ensure_work(g, iv_bytes + sums_bytes + fit_bytes, iv_bytes + fit_bytes));
  char* w = (char*)g->work;
  Interval* d_iv = (Interval*)w; w += iv_bytes;
  double* d_sums = (double*)w; w += sums_bytes;
  double* d_fit = (double*)w;
  std::memcpy(g->pinned, iv.data(), (size_t)n * sizeof(Interval));
  ECCKD_HIP_CHECK(hipMemcpyAsync(d_iv, g->pinned, (size_t)n * sizeof(Interval), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_interval_sums, dim3(g->nrows, n), dim3(256), 0, ctx->stream, IntervalArgs(), g->nrows, g->ntiles, g->nsuper, d_iv, 0,
                     d_iv, (const double* const*)g->rows, g->tile_sums, g->super_sums, d_sums);
  if (g->do_sw) {
    // unscaled fit: scaling factors of 1
    hipLaunchKernelGGL(k_fit_sw, dim3(n), dim3(128), nlay * sizeof(double), ctx->stream, nlay, g->method, g->rm, n,
                       1.0, 1.0, d_iv, d_sums, d_fit);
  } else {
    hipLaunchKernelGGL(k_fit_lw, dim3(n), dim3(128), 0, ctx->stream, nlay, g->method, g->rm, d_iv, d_sums, d_fit);
  }
  ECCKD_HIP_CHECK(hipGetLastError());
  double* h = (double*)((char*)g->pinned + iv_bytes);
  ECCKD_HIP_CHECK(hipMemcpyAsync(h, d_fit, (size_t)n * nlay * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  std::memcpy(h_od_fit, h, (size_t)n * nlay * sizeof(double));
  return ECCKD_OK;
}

int ecckd_calc_error_batch(ecckd_gas* g, size_t ibegin, size_t npoints, int n, const double* bound1,
                           const double* bound2, double* error) {
  ECCKD_REQUIRE(g && (n == 0 || (bound1 && bound2 && error)), "ecckd_calc_error_batch: NULL argument");
  if (n <= 0) return ECCKD_OK;
  const std::vector<size_t> ib((size_t)n, ibegin), np((size_t)n, npoints);
  return ecckd_calc_error_multi(g, n, ib.data(), np.data(), nullptr, bound1, bound2, error);
}

// ECCKD_TURNAROUND_LOG=1: where the host's time between two batches of a search goes (printed when the process ends)
struct TurnaroundLog {
  bool on = std::getenv("ECCKD_TURNAROUND_LOG") != nullptr;
  std::chrono::steady_clock::time_point seen;
  bool have_seen = false;
  double to_entry = 0, to_first = 0, to_last = 0, wait = 0;
  long long n = 0;
  ~TurnaroundLog() {
    if (on && n)
      std::fprintf(stderr, "search batches %lld: errors seen -> next evaluation entered %.2f us, -> first launch issued %.2f us, -> last launch "
                           "issued %.2f us, waiting for the errors %.2f us\n", n, 1e6 * to_entry / n, 1e6 * to_first / n, 1e6 * to_last / n, 1e6 * wait / n);
  }
};
static TurnaroundLog g_turn;

// Errors of the intervals iv[0..n) (first / last sorted index and albedo filled in) on the device -> error[0..n).
static int eval_intervals(ecckd_gas* g, std::vector<Interval>& iv, double* error) {
  const auto t_entry = g_turn.on ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
  std::chrono::steady_clock::time_point t_first, t_last;
  const int n = (int)iv.size();
  ecckd_ctx* ctx = g->ctx;
  // the lane of this evaluation: the context's own stream, events and counters, or the ones this gas was lent (ecckd_find_g_gases)
  ecckd_lane* const lane = g->lane;
  const hipStream_t stream = g->eval_stream();
  hipEvent_t const pev0 = lane ? lane->pev0 : ctx->pev0, pev1 = lane ? lane->pev1 : ctx->pev1;
  long long& profile_seq = lane ? lane->profile_seq : ctx->profile_seq;
  ecckd_lane_stat& stat_rt_lw = lane ? lane->stat_rt_lw : ctx->stat_rt_lw;
  ecckd_lane_stat& stat_rt_sw = lane ? lane->stat_rt_sw : ctx->stat_rt_sw;
  const int nlay = g->nlay, nhl = nlay + 1;
  long long total_pts = 0;
  for (int k = 0; k < n; ++k) total_pts += iv[k].i2 - iv[k].i1 + 1;

  // Chunking.  Every interval is cut into chunks of ITS OWN size: the smallest multiple of what one block iteration covers
  // (the longwave mirror kernel: two wave pairs = 128 points; the other sweeps: 256) with which the interval fills at most
  // one round of resident blocks.  The chunk size - and with it the order in which the interval's points are added up, chunk
  // by chunk in K5c and partial by partial in K5d - is a function of the interval's length alone, so an interval's error has
  // the same bits whether it is evaluated alone, with its neighbours (calc_error_all, equipartition.h:98-116) or next to
  // other bands' intervals (ecckd_calc_error_multi).  A batch of n intervals launches up to n rounds of small blocks; the
  // hardware's block dispatcher balances them.
  // ECCKD_RT_GENERIC (read per call): the run-time-nlay sweeps instead of the compile-time ones, for cross-checks at full size
  const bool fast_path = (nlay == 54 || nlay == 30) && std::getenv("ECCKD_RT_GENERIC") == nullptr;
  // mirror path: 3 resident blocks per CU (3 waves/SIMD); the two-fit shortwave sweep holds 2
  static const int rt_bpc = std::getenv("ECCKD_RT_BPC") ? std::max(1, std::atoi(std::getenv("ECCKD_RT_BPC"))) : 3;   // tuning knob
  static const int sw2_bpc = std::getenv("ECCKD_SW2_BPC") ? std::max(1, std::atoi(std::getenv("ECCKD_SW2_BPC"))) : 2;   // tuning knob
  const bool sw_two_fits = g->do_sw && g->method == ECCKD_AVG_TOTAL_TRANSMISSION && fast_path;
  const long long target_blocks = (long long)ctx->num_cu * (sw_two_fits ? sw2_bpc : fast_path ? rt_bpc : 8);
  const long long gran = (!g->do_sw && fast_path) ? RT_THREADS / 2 : RT_THREADS;
  long long nchunks = 0;
  for (int k = 0; k < n; ++k) {
    const long long len = iv[k].i2 - iv[k].i1 + 1;
    iv[k].chunk_pts = interval_chunk_pts(len, target_blocks, gran);
    iv[k].chunk0 = nchunks;
    nchunks += (len + iv[k].chunk_pts - 1) / iv[k].chunk_pts;
  }
  ECCKD_REQUIRE(nchunks < 0x7fffffffLL, "ecckd_calc_error_batch: %lld chunks in one batch", nchunks);

  // device work layout: intervals | sums[n][nrows] | od_fit[2][n][nlay] | partial[nchunks][2nhl] | err[2][n]
  const size_t iv_bytes = ecckd_align_up((size_t)n * sizeof(Interval), 256);
  const size_t sums_bytes = ecckd_align_up((size_t)n * g->nrows * sizeof(double), 256);
  const size_t fit_bytes = ecckd_align_up((size_t)2 * n * nlay * sizeof(double), 256);
  const size_t part_bytes = ecckd_align_up((size_t)2 * nchunks * 2 * nhl * sizeof(double), 256);   // x2: both fits of the dual shortwave sweep
  const size_t err_bytes = ecckd_align_up((size_t)2 * n * sizeof(double), 256);
  ECCKD_CHECK(gas_ensure_work(g, iv_bytes + sums_bytes + fit_bytes + part_bytes + err_bytes, iv_bytes + err_bytes));
  char* w = (char*)g->work;
  Interval* d_iv = (Interval*)w; w += iv_bytes;
  double* d_sums = (double*)w; w += sums_bytes;
  double* d_fit = (double*)w; w += fit_bytes;
  double* d_part = (double*)w; w += part_bytes;
  (void)err_bytes;   // the errors are written into the pinned host buffer
  Interval* h_iv = (Interval*)g->pinned;
  double* h_err = (double*)((char*)g->pinned + iv_bytes);
  // the errors are written straight into the pinned host buffer by the last kernel (a few bytes over PCIe): no
  // device-to-host copy in the stream
  if (g->pinned_dev_of != g->pinned) {        // the device alias of the pinned buffer, looked up once per (re)allocation
    ECCKD_HIP_CHECK(hipHostGetDevicePointer((void**)&g->pinned_dev, g->pinned, 0));
    g->pinned_dev_of = g->pinned;
  }
  double* h_err_dev = (double*)((char*)g->pinned_dev + iv_bytes);
  const int nslots = (g->do_sw && g->method == ECCKD_AVG_TOTAL_TRANSMISSION) ? 2 * n : n;
  mark_pending(h_err, nslots);
  static const bool no_karg = std::getenv("ECCKD_NO_KARG") != nullptr;   // A/B knob: always copy the interval table
  const int use_ka = (n <= KARG_MAX && !no_karg) ? 1 : 0;
  IntervalArgs ka;
  if (use_ka) {
    for (int k = 0; k < KARG_MAX; ++k) ka.iv[k] = iv[k < n ? k : n - 1];
  } else {
    std::memset(&ka, 0, sizeof ka);
    std::memcpy(h_iv, iv.data(), (size_t)n * sizeof(Interval));
    ECCKD_HIP_CHECK(hipMemcpyAsync(d_iv, h_iv, (size_t)n * sizeof(Interval), hipMemcpyHostToDevice, stream));
  }

  LwRowBases lw_rows;
  lw_rows.w1 = g->w1; lw_rows.w2 = g->w2; lw_rows.cnt = g->cnt; lw_rows.planck_hl = g->planck_hl; lw_rows.hr = g->hr;
  lw_rows.fds = g->fds; lw_rows.fut = g->fut; lw_rows.n = g->n; lw_rows.is_log = g->method == ECCKD_AVG_LOGARITHMIC;
  if (!g->do_sw)
    hipLaunchKernelGGL(k_interval_sums_fit_lw, dim3(nlay + (g->rm.total - g->rm.H), n), dim3(256), 0, stream, ka, nlay, g->method,
                       g->rm, g->ntiles, g->nsuper, d_iv, use_ka, d_iv, lw_rows, g->tile_sums, g->super_sums,
                       d_sums, d_fit);
  else
    hipLaunchKernelGGL(k_interval_sums, dim3(g->nrows, n), dim3(256), 0, stream, ka, g->nrows, g->ntiles, g->nsuper, d_iv, use_ka,
                       d_iv, (const double* const*)g->rows, g->tile_sums, g->super_sums, d_sums);
  if (g->do_sw) {
    // CkdEquipartition::calc_error, shortwave branches (find_g_points.cpp:341-402)
    const bool is_tt = g->method == ECCKD_AVG_TOTAL_TRANSMISSION;
    const RowMap& R = g->rm;
    hipLaunchKernelGGL(k_fit_sw, dim3(n), dim3(128), nlay * sizeof(double), stream, nlay, g->method, R, n,
                       g->min_scaling, g->max_scaling, d_iv, d_sums, d_fit);
    const size_t rt_lds_sw = (size_t)(4 * 2 * nhl + nlay) * sizeof(double);
    const size_t cost_lds_sw = (size_t)(COST_GROUPS * 2 * nhl + 2 * nhl + nlay) * sizeof(double);
    ECCKD_REQUIRE(target_blocks <= 32 * COST_GROUPS && cost_lds_sw <= 64 * 1024, "ecckd_calc_error_batch: %lld chunks per interval / %d layers exceed the cost kernel's room",
                  target_blocks, nlay);
    const int npass = is_tt ? 2 : 1;
    // total-transmission evaluates the interval with the fit scaled by min_scaling and by max_scaling (:374-386; never equal
    // in the reference: min <= 0.5, max >= 2.5, :666-667): one launch that fetches the column once and sweeps it with both
    // fits (the compile-time-nlay kernels).  At the reference's cos_sza = 0.5 the direct and the reflected beam see the same
    // transmittance exp(-2 tau), bit for bit: the kernels then keep it from the way down (SAME).
    const bool dual = is_tt && fast_path;
    const size_t part_stride = (size_t)nchunks * 2 * nhl;
    static const bool no_same = std::getenv("ECCKD_SW_NO_SAME") != nullptr;   // A/B knob
    const bool same_exp = g->cos_sza == 0.5 && !no_same;
    // HIP events round the sweep launches of every profile_stride-th batch (ecckd_profile_enable), as for the longwave sweep
    const bool timed_sw = ctx->profile && (profile_seq++ % ctx->profile_stride) == 0;
    if (ctx->profile) { stat_rt_sw.all_calls += 1; stat_rt_sw.all_units += (double)total_pts; }
    if (timed_sw) ECCKD_HIP_CHECK(hipEventRecord(pev0, stream));
    for (int pass = 0; pass < (dual ? 1 : npass); ++pass) {
      double* part = d_part + (size_t)pass * part_stride;
#define ECCKD_SW_SWEEP(NL, NF, SM, FIT)                                                                                         \
  hipLaunchKernelGGL((k_rt_sw_bb_fast<NL, NF, SM>), dim3((unsigned)nchunks), dim3(RT_THREADS), 0, stream, g->n, n, \
                     d_iv, g->cos_sza, g->ssi, g->bg_od, FIT, part)
      const double* fit1 = d_fit + (size_t)pass * n * nlay;
      if (!fast_path)
        hipLaunchKernelGGL(k_rt_sw_bb, dim3((unsigned)nchunks), dim3(RT_THREADS), rt_lds_sw, stream, nlay, g->n,
                           n, d_iv, g->cos_sza, g->ssi, g->bg_od, fit1, part);
      else if (nlay == 54 && dual) { if (same_exp) ECCKD_SW_SWEEP(54, 2, true, d_fit); else ECCKD_SW_SWEEP(54, 2, false, d_fit); }
      else if (nlay == 30 && dual) { if (same_exp) ECCKD_SW_SWEEP(30, 2, true, d_fit); else ECCKD_SW_SWEEP(30, 2, false, d_fit); }
      else if (nlay == 54) { if (same_exp) ECCKD_SW_SWEEP(54, 1, true, fit1); else ECCKD_SW_SWEEP(54, 1, false, fit1); }
      else if (nlay == 30) { if (same_exp) ECCKD_SW_SWEEP(30, 1, true, fit1); else ECCKD_SW_SWEEP(30, 1, false, fit1); }
#undef ECCKD_SW_SWEEP
    }
    if (timed_sw) ECCKD_HIP_CHECK(hipEventRecord(pev1, stream));
    // both evaluations in one launch; the errors go straight into the pinned host buffer (a few bytes over PCIe)
    SwTruthRows rows;
    rows.rH[0] = is_tt ? R.HL : R.H;      rows.rH[1] = R.HH;
    rows.rFDS[0] = is_tt ? R.FDSL : R.FDS; rows.rFDS[1] = R.FDSH;
    rows.rFUT[0] = is_tt ? R.FUTL : R.FUT; rows.rFUT[1] = R.FUTH;
    hipLaunchKernelGGL(k_cost_sw, dim3(n, npass), dim3(1024), cost_lds_sw, stream, nlay, R.total, rows, d_iv, nchunks,
                       d_part, part_stride, d_sums, g->lev + nhl, g->lev + nhl + nlay, g->flux_weight, g->cos_sza, h_err_dev);
    ECCKD_HIP_CHECK(hipGetLastError());
    ECCKD_CHECK(wait_for_slots(stream, h_err, nslots));
    for (int k = 0; k < n; ++k) error[k] = is_tt ? 0.5 * (h_err[k] + h_err[n + k]) : h_err[k];  // :386
    if (timed_sw) {
      float ms = 0.f;
      ECCKD_HIP_CHECK(hipEventSynchronize(pev1));
      ECCKD_HIP_CHECK(hipEventElapsedTime(&ms, pev0, pev1));
      stat_rt_sw.ms += ms;
      stat_rt_sw.units += (double)total_pts;
      stat_rt_sw.calls += 1;
    }
    return ECCKD_OK;
  }
  if (g_turn.on) t_first = std::chrono::steady_clock::now();
  const size_t rt_lds = (size_t)(4 * 2 * nhl + nlay) * sizeof(double);
  const bool timed = ctx->profile && (profile_seq++ % ctx->profile_stride) == 0;
  if (ctx->profile) { stat_rt_lw.all_calls += 1; stat_rt_lw.all_units += (double)total_pts; }
  if (timed) ECCKD_HIP_CHECK(hipEventRecord(pev0, stream));
#define ECCKD_LW_MIRROR(NL, P32)                                                                                          \
  hipLaunchKernelGGL((k_rt_lw_bb_mirror<NL, P32>), dim3(mirror_grid), dim3(RT_THREADS), 0, stream, g->n, n, nchunks, d_iv,     \
                     g->planck_hl, g->bg_od, (const float_x2*)g->bg_pair, d_fit, d_part)
  // a block per chunk.  (ECCKD_RT_PERSISTENT: one resident round of blocks, each taking every mirror_grid-th chunk - measured
  // SLOWER on a partition of unequal intervals, 1 069 against 1 015 us per pass of 38 intervals over 7.2e6 points, equal on
  // equal intervals: the hardware's block dispatcher balances chunks of one to three tiles better than a fixed deal.)
  static const bool persistent = std::getenv("ECCKD_RT_PERSISTENT") != nullptr;
  const unsigned mirror_grid = (unsigned)(persistent ? std::min<long long>(nchunks, target_blocks) : nchunks);
  if (fast_path && nlay == 54) {
    if (g->bg_pair) ECCKD_LW_MIRROR(54, true); else ECCKD_LW_MIRROR(54, false);
  } else if (fast_path && nlay == 30) {
    if (g->bg_pair) ECCKD_LW_MIRROR(30, true); else ECCKD_LW_MIRROR(30, false);
#undef ECCKD_LW_MIRROR
  } else {
    ECCKD_CHECK(gas_bg_rows(g));
    hipLaunchKernelGGL(k_rt_lw_bb, dim3((unsigned)nchunks), dim3(RT_THREADS), rt_lds, stream, nlay, g->n,
                       n, d_iv, g->planck_hl, g->bg_od, d_fit, d_part);
  }
  if (timed) ECCKD_HIP_CHECK(hipEventRecord(pev1, stream));
  const size_t cost_lds = (size_t)(COST_GROUPS * 2 * nhl + 2 * nhl + nlay) * sizeof(double);
  ECCKD_REQUIRE(target_blocks <= 32 * COST_GROUPS && cost_lds <= 64 * 1024, "ecckd_calc_error_batch: %lld chunks per interval / %d layers exceed the cost kernel's room",
                target_blocks, nlay);
  hipLaunchKernelGGL(k_cost_lw, dim3(n), dim3(1024), cost_lds, stream, nlay, g->rm, d_iv,
                     nchunks, d_part, d_sums, g->lev + nhl, g->lev + nhl + nlay, g->flux_weight, h_err_dev);
  ECCKD_HIP_CHECK(hipGetLastError());
  if (g_turn.on) t_last = std::chrono::steady_clock::now();
  ECCKD_CHECK(wait_for_slots(stream, h_err, nslots));
  if (g_turn.on) {
    const auto now = std::chrono::steady_clock::now();
    if (g_turn.have_seen && std::chrono::duration<double>(t_entry - g_turn.seen).count() < 500e-6) {   // not the pause between two searches
      g_turn.to_entry += std::chrono::duration<double>(t_entry - g_turn.seen).count();
      g_turn.to_first += std::chrono::duration<double>(t_first - g_turn.seen).count();
      g_turn.to_last += std::chrono::duration<double>(t_last - g_turn.seen).count();
      g_turn.wait += std::chrono::duration<double>(now - t_last).count();
      g_turn.n += 1;
    }
    g_turn.seen = now;
    g_turn.have_seen = true;
  }
  std::memcpy(error, h_err, (size_t)n * sizeof(double));
  if (timed) {
    float ms = 0.f;
    ECCKD_HIP_CHECK(hipEventSynchronize(pev1));    // long past: the cost kernel behind it has delivered
    ECCKD_HIP_CHECK(hipEventElapsedTime(&ms, pev0, pev1));
    stat_rt_lw.ms += ms;
    stat_rt_lw.units += (double)total_pts;
    stat_rt_lw.calls += 1;
    // ECCKD_SWEEP_LOG=<file>: one line per sweep launch (intervals, points, chunks, ms) for tools/sweep_sizes.py
    static FILE* sweep_log = std::getenv("ECCKD_SWEEP_LOG") ? std::fopen(std::getenv("ECCKD_SWEEP_LOG"), "a") : nullptr;
    if (sweep_log) std::fprintf(sweep_log, "%d %lld %lld %.6f\n", n, total_pts, nchunks, (double)ms);
  }
  return ECCKD_OK;
}

// The same for intervals of DIFFERENT bands in one batch: interval k is the fraction [bound1[k], bound2[k]] of the band
// that starts at sorted index ibegin[k] and has npoints[k] points.  One launch train for all of them - the band searches
// of a gas are independent (find_g_points.cpp:1152), so their error evaluations can share the GPU (ecckd_find_g_bands_ex).
int ecckd_calc_error_multi(ecckd_gas* g, int n, const size_t* ibegin_k, const size_t* npoints_k, const double* albedo_k,
                           const double* bound1, const double* bound2, double* error) {
  ECCKD_REQUIRE(g && (n == 0 || (ibegin_k && npoints_k && bound1 && bound2 && error)), "ecckd_calc_error_multi: NULL argument");
  if (n <= 0) return ECCKD_OK;
  ecckd_ctx* ctx = g->ctx;
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));

  // index mapping and error paths of CkdEquipartition::calc_error (find_g_points.cpp:282-320)
  std::vector<Interval> iv(n);
  for (int k = 0; k < n; ++k) {
    const size_t ibegin = ibegin_k[k], npoints = npoints_k[k];
    ECCKD_REQUIRE(npoints > 0 && ibegin + npoints <= g->n,
                  "ecckd_calc_error_batch: band [%zu,%zu) outside the spectrum (%zu points)", ibegin, ibegin + npoints, g->n);
    const double b1 = bound1[k], b2 = bound2[k];
    long long i1 = (long long)std::ceil(b1 * (double)(npoints - 1));
    long long i2 = (long long)std::floor(b2 * (double)(npoints - 1));
    if (i1 < 0 || i2 >= (long long)npoints || !(b1 == b1) || !(b2 == b2))
      return ecckd::fail(ECCKD_PROCESSING_ERROR,
                         "requested bounds %.17g-%.17g corresponding to indices %lld-%lld outside valid range 0-%zu",
                         b1, b2, i1, i2, npoints - 1);
    if (b2 < b1) return ecckd::fail(ECCKD_PROCESSING_ERROR, "requested bounds out of order: %.17g-%.17g", b1, b2);
    if (i2 + 1 < i1) return ecckd::fail(ECCKD_PROCESSING_ERROR, "requested indices out of order: %lld-%lld", i1, i2);
    if (i2 < i1) i2 = i1;
    iv[k].i1 = (long long)ibegin + i1;
    iv[k].i2 = (long long)ibegin + i2;
    iv[k].chunk0 = 0;
    iv[k].chunk_pts = 0;
    iv[k].npoints = (long long)npoints;
    iv[k].albedo = albedo_k ? albedo_k[k] : g->surf_albedo;
  }
  for (int k = 0; k < n; ++k) g->total_comp_cost += bound2[k] - bound1[k];  // :320: the reference's counter counts every request

  // The memo: only intervals not seen before (and once each) go to the device.  ECCKD_NO_ERROR_MEMO (read per call): every
  // request is evaluated, as the reference does - for cross-checks.
  const bool use_memo = std::getenv("ECCKD_NO_ERROR_MEMO") == nullptr;
  if (g->error_memo.size() > (size_t)4000000) g->error_memo.clear();
  std::vector<Interval> todo;
  std::vector<int> slot(n, -1);            // index into todo, or -1: served from the memo
  std::unordered_map<IntervalKey, int, IntervalKeyHash> in_batch;
  for (int k = 0; k < n; ++k) {
    IntervalKey key;
    key.i1 = iv[k].i1; key.i2 = iv[k].i2;
    std::memcpy(&key.albedo_bits, &iv[k].albedo, sizeof(double));
    if (!g->do_sw) key.albedo_bits = 0;
    const double len = (double)(iv[k].i2 - iv[k].i1 + 1);
    g->memo_requests += 1;
    g->points_requested += len;
    if (use_memo) {
      auto hit = g->error_memo.find(key);
      if (hit != g->error_memo.end()) { error[k] = hit->second; g->memo_hits += 1; continue; }
      auto dup = in_batch.find(key);
      if (dup != in_batch.end()) { slot[k] = dup->second; g->memo_hits += 1; continue; }
      in_batch.emplace(key, (int)todo.size());
    }
    slot[k] = (int)todo.size();
    todo.push_back(iv[k]);
    g->points_evaluated += len;
  }
  if (todo.empty()) return ECCKD_OK;
  std::vector<double> fresh(todo.size());
  ECCKD_CHECK(eval_intervals(g, todo, fresh.data()));
  for (int k = 0; k < n; ++k)
    if (slot[k] >= 0) error[k] = fresh[slot[k]];
  if (use_memo)
    for (size_t t = 0; t < todo.size(); ++t) {
      IntervalKey key;
      key.i1 = todo[t].i1; key.i2 = todo[t].i2;
      std::memcpy(&key.albedo_bits, &todo[t].albedo, sizeof(double));
      if (!g->do_sw) key.albedo_bits = 0;
      g->error_memo.emplace(key, fresh[t]);
    }
  return ECCKD_OK;
}

// What the memo of interval errors saved: intervals asked for / found in the memo, wavenumber points asked for (what the
// reference would have swept) / actually swept on the device.
int ecckd_gas_eval_stats(ecckd_gas* gas, long long* requests, long long* memo_hits, double* points_requested, double* points_evaluated) {
  ECCKD_REQUIRE(gas, "ecckd_gas_eval_stats: NULL handle");
  if (requests) *requests = gas->memo_requests;
  if (memo_hits) *memo_hits = gas->memo_hits;
  if (points_requested) *points_requested = gas->points_requested;
  if (points_evaluated) *points_evaluated = gas->points_evaluated;
  return ECCKD_OK;
}

// Forget every interval error this gas has answered and zero the counters of ecckd_gas_eval_stats: the next search sweeps
// every interval again (timing one prepared gas twice; the reference has no memo at all).
int ecckd_gas_reset_memo(ecckd_gas* gas) {
  ECCKD_REQUIRE(gas, "ecckd_gas_reset_memo: NULL handle");
  gas->error_memo.clear();
  gas->memo_requests = gas->memo_hits = 0;
  gas->points_requested = gas->points_evaluated = 0.0;
  gas->total_comp_cost = 0.0;
  return ECCKD_OK;
}

// Bytes the error sweep (K5c) reads per spectral point of an interval: the Planck rows and the background rows as this gas
// holds them (FLOAT pairs if every background value is a float, DOUBLE rows otherwise); shortwave: background rows + ssi.
int ecckd_gas_sweep_bytes_per_point(ecckd_gas* gas, double* bytes) {
  ECCKD_REQUIRE(gas && bytes, "ecckd_gas_sweep_bytes_per_point: NULL argument");
  const int nlay = gas->nlay;
  if (gas->do_sw) *bytes = (double)(nlay + 1) * 8.0;
  else *bytes = (double)(nlay + 1) * 8.0 + (double)nlay * (gas->bg_pair ? 4.0 : 8.0);
  return ECCKD_OK;
}

// ---------------------------------------------------------------------------
// partition search over a host callback (replaces Equipartition, equipartition.h:63-208)
struct ecckd_partition {
  ecckd_error_fn fn = nullptr;
  void* user = nullptr;
  ecckd::PartitionSearch* ps = nullptr;
};

int ecckd_partition_create(ecckd_error_fn fn, void* user, ecckd_partition** out) {
  ECCKD_REQUIRE(fn && out, "ecckd_partition_create: NULL argument");
  ecckd_partition* p = new ecckd_partition();
  p->fn = fn;
  p->user = user;
  p->ps = new ecckd::PartitionSearch(
      [p](int n, const double* b1, const double* b2, double* e) { return p->fn(n, b1, b2, e, p->user); });
  *out = p;
  return ECCKD_OK;
}

int ecckd_partition_destroy(ecckd_partition* p) {
  if (p) {
    delete p->ps;
    delete p;
  }
  return ECCKD_OK;
}

int ecckd_partition_configure(ecckd_partition* p, double resolution, double partition_tolerance,
                              int partition_max_iterations, int line_search_max_iterations,
                              int cubic_interpolation, int minimize_frac_range) {
  ECCKD_REQUIRE(p, "ecckd_partition_configure: NULL handle");
  p->ps->set_resolution(resolution);
  p->ps->set_partition_tolerance(partition_tolerance);
  p->ps->set_partition_max_iterations(partition_max_iterations);
  p->ps->set_line_search_max_iterations(line_search_max_iterations);
  p->ps->set_cubic_interpolation(cubic_interpolation != 0);
  p->ps->set_minimize_frac_range(minimize_frac_range != 0);
  return ECCKD_OK;
}

int ecckd_partition_n(ecckd_partition* p, int ni, double* bounds, double* error, int* status) {
  ECCKD_REQUIRE(p && ni > 0 && bounds && error && status, "ecckd_partition_n: bad argument");
  *status = p->ps->equipartition_n(ni, bounds, error);
  if (p->ps->evaluator_status()) return p->ps->evaluator_status();
  return ECCKD_OK;
}

int ecckd_partition_e(ecckd_partition* p, double target_error, double bound0, double boundn, int* ni,
                      double* bounds, double* error, int capacity, int* status) {
  ECCKD_REQUIRE(p && ni && bounds && error && status, "ecckd_partition_e: NULL argument");
  std::vector<double> b, e;
  int n = 0;
  *status = p->ps->equipartition_e(target_error, bound0, boundn, n, b, e);
  if (p->ps->evaluator_status()) return p->ps->evaluator_status();
  *ni = n;
  if (*status == ecckd::PS_INPUT_ERROR) return ECCKD_OK;
  ECCKD_REQUIRE(n <= capacity, "ecckd_partition_e: %d intervals exceed the caller's capacity %d", n, capacity);
  std::memcpy(bounds, b.data(), (size_t)(n + 1) * sizeof(double));
  std::memcpy(error, e.data(), (size_t)n * sizeof(double));
  return ECCKD_OK;
}

int ecckd_partition_set_trace(ecckd_partition* p, ecckd_trace_fn fn, void* user) {
  ECCKD_REQUIRE(p, "ecckd_partition_set_trace: NULL handle");
  if (fn) p->ps->set_trace([fn, user](int site, double lhs, double rhs, int taken) { fn(site, lhs, rhs, taken, user); });
  else p->ps->set_trace(nullptr);
  return ECCKD_OK;
}

const char* ecckd_partition_status_string(int status) { return ecckd::partition_status_string(status); }

}  // extern "C"
