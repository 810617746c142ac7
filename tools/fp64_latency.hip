// fp64 VALU issue/latency microbenchmark for gfx950: K independent v_fma_f64 chains per lane,
// W waves per block (one block per CU when grid = #CU).  Prints shader cycles per FMA per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int K>
__global__ void chain(double* out, long long* cyc, int n, double a, double b) {
  double x[K];
#pragma unroll
  for (int k = 0; k < K; ++k) x[k] = threadIdx.x * 1e-3 + k;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < K; ++k) x[k] = __builtin_fma(x[k], a, b);
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int K>
void run(int waves, int n) {
  const int blocks = 256;
  double* out; long long* cyc;
  hipMalloc(&out, blocks * waves * 64 * sizeof(double));
  hipMalloc(&cyc, blocks * waves * sizeof(long long));
  hipLaunchKernelGGL(chain<K>, dim3(blocks), dim3(waves * 64), 0, 0, out, cyc, n, 0.999, 1e-3);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(chain<K>, dim3(blocks), dim3(waves * 64), 0, 0, out, cyc, n, 0.999, 1e-3);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks * waves);
  hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= h.size();
  // s_memtime counts at 100 MHz: report wall-time based numbers as well
  const double fma_per_simd = (double)n * K * waves / 4.0;   // waves spread over 4 SIMDs
  printf("K=%d waves/CU=%2d  counter ticks/wave=%.0f  kernel %.3f ms -> %.2f ns per wave-FMA per SIMD (%.2f cycles at 2.3 GHz)\n",
         K, waves, avg, ms, ms * 1e6 / fma_per_simd, ms * 1e6 / fma_per_simd * 2.3);
  hipFree(out); hipFree(cyc);
}

int main() {
  const int n = 200000;
  for (int waves : {4, 8, 16}) {
    run<1>(waves, n); run<2>(waves, n); run<4>(waves, n); run<8>(waves, n);
  }
  return 0;
}
