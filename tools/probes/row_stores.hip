// Probe: store bandwidth of a kernel that writes R row streams at once (the shape of the gas preparation: every thread
// owns one spectral point and writes one value into each of R rows), row-major rows[r][i] against a blocked layout
// rows[i / B][r][i % B].  build: hipcc -O3 --offload-arch=gfx950 tools/probes/row_stores.hip -o tools/probes/row_stores
#include <hip/hip_runtime.h>
#include <cstdio>
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
  return 0;
}
