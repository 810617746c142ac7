// overlap.hip - a14: the spectral overlap of the gases' g points.
//
// Replaces reference src/ecckd/single_gas_data.cpp:24-124 (overlap_g_points, the hypercube
// partition of Hogan 2010), SingleGasData::store_g_points (single_gas_data.h:56-62) and the
// per-wavenumber g-point assignment of find_g_points.cpp:1459-1475.  The overlap itself is
// O(ng*ngas) integer logic and stays on the host; the two per-wavenumber maps (rank ->
// single-gas g point, single-gas g points -> merged g point) are O(ng*ngas*nwav) `where`
// passes in the reference and one kernel each here.
#include "common.hpp"

#include <cstring>
#include <vector>

namespace {

// single_gas_data.h:56-62: g_point = ig for rank in [rank1(ig), rank2(ig)]; later ig win
__global__ void __launch_bounds__(256)
k_gas_g_point(size_t n, const int32_t* __restrict__ rank, int ng, const int32_t* __restrict__ rank1,
              const int32_t* __restrict__ rank2, int32_t* __restrict__ g_point) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int32_t r = rank[j];
  int32_t g = -1;
  for (int ig = 0; ig < ng; ++ig)
    if (r >= rank1[ig] && r <= rank2[ig]) g = ig;
  g_point[j] = g;
}

// find_g_points.cpp:1459-1475: merged g point ig contains wavenumber j if every gas's g point
// lies in [g_min(ig), g_max(ig)]; ig ascends, later matches overwrite
__global__ void __launch_bounds__(256)
k_merge_g_points(size_t n, int ngas, int ng, const int32_t* const* __restrict__ gas_g_point,
                 const int32_t* __restrict__ g_min /*[ngas][ng]*/, const int32_t* __restrict__ g_max,
                 int32_t* __restrict__ g_point, unsigned long long* __restrict__ n_unassigned) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  int32_t g = -1;
  for (int ig = 0; ig < ng; ++ig) {
    bool found = true;
    for (int igas = 0; igas < ngas; ++igas) {
      const int32_t gg = gas_g_point[igas][j];
      if (gg < g_min[igas * ng + ig] || gg > g_max[igas * ng + ig]) { found = false; break; }
    }
    if (found) g = ig;
  }
  g_point[j] = g;
  if (g < 0) atomicAdd(n_unassigned, 1ull);  // integer count: order-independent
}

}  // namespace

extern "C" {

int ecckd_overlap_g_points(int ngas, int nband, const int* h_n_g_points, const int* h_gas_offset,
                           const double* h_sorting_variable, int capacity, int* h_ng, int* h_band_number,
                           int* h_g_min, int* h_g_max) {
  ECCKD_REQUIRE(ngas > 0 && nband > 0 && h_n_g_points && h_gas_offset && h_sorting_variable && h_ng &&
                    h_band_number && h_g_min && h_g_max, "ecckd_overlap_g_points: bad argument");
  // Eq. 7 of Hogan (2010): ng_band = 1 - ngas + sum_i ng_i  (single_gas_data.cpp:30-38)
  std::vector<int> ng_band(nband);
  int ng = 0;
  for (int b = 0; b < nband; ++b) {
    ng_band[b] = 1 - ngas;
    for (int i = 0; i < ngas; ++i) ng_band[b] += h_n_g_points[i * nband + b];
    ng += ng_band[b];
  }
  *h_ng = ng;
  ECCKD_REQUIRE(ng <= capacity, "ecckd_overlap_g_points: %d g points exceed the caller's capacity %d", ng, capacity);
  {
    int ig = 0;
    for (int b = 0; b < nband; ++b)
      for (int q = 0; q < ng_band[b]; ++q) h_band_number[ig++] = b;
  }
  int ig = 0;
  std::vector<int> ig_gas(ngas, 0), start(ngas);
  for (int b = 0; b < nband; ++b) {
    start = ig_gas;
    // first merged g point of a band: intersection of the weakest interval of every gas (:63-70)
    for (int i = 0; i < ngas; ++i) {
      h_g_min[i * capacity + ig] = start[i];
      h_g_max[i * capacity + ig] = start[i];
    }
    for (int q = 1; q < ng_band[b]; ++q) {
      // advance the gas whose next interval has the smallest sorting variable (:73-95)
      double best = 1.0e30;
      int found = -1;
      for (int i = 0; i < ngas; ++i) {
        double mine = 1.0e30;
        if (ig_gas[i] < start[i] + h_n_g_points[i * nband + b] - 1) mine = h_sorting_variable[h_gas_offset[i] + ig_gas[i] + 1];
        if (mine < best) { best = mine; found = i; }
      }
      // the reference executes a bare `throw;` here (:92-95)
      if (found < 0) return ecckd::fail(ECCKD_PROCESSING_ERROR, "Could not locate next gas to advance");
      ++ig_gas[found];
      ++ig;
      for (int i = 0; i < ngas; ++i) {
        if (i == found) {
          h_g_min[i * capacity + ig] = ig_gas[i];
          h_g_max[i * capacity + ig] = ig_gas[i];
        } else {
          h_g_min[i * capacity + ig] = start[i];
          h_g_max[i * capacity + ig] = ig_gas[i];
        }
      }
    }
    ++ig;
    for (int i = 0; i < ngas; ++i) ++ig_gas[i];
  }
  return ECCKD_OK;
}

int ecckd_gas_g_point_dev(ecckd_ctx* ctx, size_t nwav, const int32_t* d_rank, int ng, const int32_t* h_rank1,
                          const int32_t* h_rank2, int32_t* d_g_point) {
  ECCKD_REQUIRE(ctx && d_rank && h_rank1 && h_rank2 && d_g_point && ng > 0, "ecckd_gas_g_point_dev: bad argument");
  if (nwav == 0) return ECCKD_OK;
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  const size_t bytes = ecckd_align_up((size_t)2 * ng * sizeof(int32_t), 256);
  ECCKD_CHECK(ecckd::ensure_scratch(ctx, bytes));
  int32_t* d_r1 = (int32_t*)ctx->scratch;
  int32_t* d_r2 = d_r1 + ng;
  ECCKD_CHECK(ecckd_h2d(ctx, d_r1, h_rank1, (size_t)ng * sizeof(int32_t)));
  ECCKD_CHECK(ecckd_h2d(ctx, d_r2, h_rank2, (size_t)ng * sizeof(int32_t)));
  hipLaunchKernelGGL(k_gas_g_point, dim3((unsigned)((nwav + 255) / 256)), dim3(256), 0, ctx->stream, nwav, d_rank, ng, d_r1,
                     d_r2, d_g_point);
  ECCKD_HIP_CHECK(hipGetLastError());
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return ECCKD_OK;
}

int ecckd_merge_g_points_dev(ecckd_ctx* ctx, size_t nwav, int ngas, const int32_t* const* h_d_gas_g_point, int ng,
                             int stride, const int* h_g_min, const int* h_g_max, int32_t* d_g_point,
                             int64_t* h_n_unassigned) {
  ECCKD_REQUIRE(ctx && h_d_gas_g_point && h_g_min && h_g_max && d_g_point && ngas > 0 && ng > 0 && stride >= ng,
                "ecckd_merge_g_points_dev: bad argument");
  if (nwav == 0) return ECCKD_OK;
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  const size_t tb = ecckd_align_up((size_t)ngas * ng * sizeof(int32_t), 256);
  const size_t pb = ecckd_align_up((size_t)ngas * sizeof(void*), 256);
  ECCKD_CHECK(ecckd::ensure_scratch(ctx, 2 * tb + pb + 256));
  char* w = (char*)ctx->scratch;
  int32_t* d_min = (int32_t*)w; w += tb;
  int32_t* d_max = (int32_t*)w; w += tb;
  const int32_t** d_ptrs = (const int32_t**)w; w += pb;
  unsigned long long* d_cnt = (unsigned long long*)w;
  std::vector<int32_t> mn((size_t)ngas * ng), mx((size_t)ngas * ng);
  for (int i = 0; i < ngas; ++i)
    for (int g = 0; g < ng; ++g) {
      mn[(size_t)i * ng + g] = h_g_min[(size_t)i * stride + g];
      mx[(size_t)i * ng + g] = h_g_max[(size_t)i * stride + g];
    }
  ECCKD_CHECK(ecckd_h2d(ctx, d_min, mn.data(), mn.size() * sizeof(int32_t)));
  ECCKD_CHECK(ecckd_h2d(ctx, d_max, mx.data(), mx.size() * sizeof(int32_t)));
  ECCKD_CHECK(ecckd_h2d(ctx, (void*)d_ptrs, h_d_gas_g_point, (size_t)ngas * sizeof(void*)));
  ECCKD_HIP_CHECK(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(k_merge_g_points, dim3((unsigned)((nwav + 255) / 256)), dim3(256), 0, ctx->stream, nwav, ngas, ng,
                     (const int32_t* const*)d_ptrs, d_min, d_max, d_g_point, d_cnt);
  ECCKD_HIP_CHECK(hipGetLastError());
  unsigned long long cnt = 0;
  ECCKD_CHECK(ecckd_d2h(ctx, &cnt, d_cnt, sizeof(cnt)));
  if (h_n_unassigned) *h_n_unassigned = (int64_t)cnt;
  return ECCKD_OK;
}

}  // extern "C"
