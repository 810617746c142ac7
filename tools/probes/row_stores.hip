// Probe: store bandwidth of a kernel that writes R row streams at once (the shape of the gas preparation: every thread
// owns one spectral point and writes one value into each of R rows), row-major rows[r][i] against a blocked layout
// rows[i / B][r][i % B]; then the same stores with arithmetic between them (WORK fused multiply-adds in four chains per store) at a
// limited number of waves per SIMD (dynamic LDS takes the room), which is how the preparation kernels run.  build: hipcc -O3 --offload-arch=gfx950 tools/probes/row_stores.hip -o tools/probes/row_stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../ecckd_amd/csrc/fastmath.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool NT>
__global__ void __launch_bounds__(256) k_rows(size_t n, int R, size_t blk, double* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double v = (double)i;
  for (int r = 0; r < R; ++r) {
    v = v * 1.0000001 + 1.0;
    const size_t a = blk ? (i / blk) * (size_t)R * blk + (size_t)r * blk + (i % blk) : (size_t)r * n + i;
    if (NT) __builtin_nontemporal_store(v, &out[a]); else out[a] = v;
  }
}

template <int WORK>
__global__ void __launch_bounds__(256) k_rows_work(size_t n, int R, double* __restrict__ out) {
  extern __shared__ double pad[];
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double a = (double)i, b = a + 1.0, c = a + 2.0, d = a + 3.0;
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int w = 0; w < WORK / 4; ++w) {
      a = a * 1.0000001 + 1.0; b = b * 1.0000002 + 1.0; c = c * 1.0000003 + 1.0; d = d * 1.0000004 + 1.0;
    }
    __builtin_nontemporal_store(a + b + c + d, &out[(size_t)r * n + i]);
  }
  if (a == -1.0) pad[threadIdx.x] = a;
}

// one exp (ecckd::exp_fast, the preparation kernels' own) per store, as in the shortwave preparation: seven of each per layer
template <int NEXP>
__global__ void __launch_bounds__(256) k_rows_exp(size_t n, int R, double* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double a = 1.0 + 1e-9 * (double)i, b = a * 0.5, c = a * 0.25, d = a * 0.125;
  for (int r = 0; r < R; r += 4) {
#pragma unroll
    for (int w = 0; w < NEXP; ++w) {
      a = ecckd::exp_fast(-a); b = ecckd::exp_fast(-b); c = ecckd::exp_fast(-c); d = ecckd::exp_fast(-d);
    }
    __builtin_nontemporal_store(a, &out[(size_t)r * n + i]);
    if (r + 1 < R) __builtin_nontemporal_store(b, &out[(size_t)(r + 1) * n + i]);
    if (r + 2 < R) __builtin_nontemporal_store(c, &out[(size_t)(r + 2) * n + i]);
    if (r + 3 < R) __builtin_nontemporal_store(d, &out[(size_t)(r + 3) * n + i]);
  }
}

template <int NEXP>
static int run_exp(size_t n, double* d, hipEvent_t e0, hipEvent_t e1) {
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rows_exp<NEXP>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, n, 378, d);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("R 378 row-major nontemporal, %d exp per store: %.2f ms = %.2f TB/s, %.1f G exp/s\n", NEXP, best, (double)n * 378 * 8 / best / 1e9,
         (double)n * 378 * NEXP / best / 1e6);
  return 0;
}

template <int WORK>
static int run_work(size_t n, double* d, hipEvent_t e0, hipEvent_t e1) {
  for (int waves : {8, 3, 2, 1}) {
    // waves per SIMD = 4 * blocks per CU / 4: one 256-thread block is one wave on every SIMD
    const size_t lds = waves >= 8 ? 0 : (size_t)(160 * 1024 / waves) - 1024;
    CK(hipFuncSetAttribute((const void*)k_rows_work<WORK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_rows_work<WORK>, dim3((unsigned)(n / 256)), dim3(256), lds, 0, n, 378, d);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("R 378 row-major nontemporal, %3d FMAs per store, %d waves/SIMD: %.2f ms = %.2f TB/s, %.1f TFLOP/s\n", WORK, waves, best,
           (double)n * 378 * 8 / best / 1e9, (double)n * 378 * WORK * 2 / best / 1e9);
  }
  return 0;
}

int main() {
  const size_t n = 7200000 / 4096 * 4096;
  double* d = nullptr;
  CK(hipMalloc(&d, n * 384 * sizeof(double)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int R : {8, 64, 219, 378})
    for (size_t blk : {(size_t)0, (size_t)256, (size_t)4096})
      for (int nt = 0; nt < 2; ++nt) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
          CK(hipEventRecord(e0));
          if (nt) hipLaunchKernelGGL(k_rows<true>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, n, R, blk, d);
          else hipLaunchKernelGGL(k_rows<false>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, n, R, blk, d);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
        }
        printf("R %3d  layout %-12s %s: %.2f ms = %.2f TB/s\n", R, blk == 0 ? "row-major" : blk == 256 ? "blocked 256" : "blocked 4096",
               nt ? "nontemporal" : "plain      ", best, (double)n * R * 8 / best / 1e9);
      }
  if (run_exp<1>(n, d, e0, e1) || run_exp<2>(n, d, e0, e1) || run_exp<4>(n, d, e0, e1)) return 1;
  if (run_work<0>(n, d, e0, e1) || run_work<16>(n, d, e0, e1) || run_work<48>(n, d, e0, e1) || run_work<96>(n, d, e0, e1)) return 1;
  return 0;
}
