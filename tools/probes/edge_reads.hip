// Probe: why do the interval sums (K5a) take ~1.6 us more per interval?  Emulates their reads - for each of `nint` intervals
// and each of 111 rows two ragged edges of 256 doubles - on (a) the row-major layout rows[r][i] (row stride nwav) and (b) a
// blocked layout rows[i / BLK][r][i % BLK], in which the edges of all rows of an interval fall into one or two pages.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/edge_reads.hip -o tools/probes/edge_reads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_edges(const double* __restrict__ base, size_t row_stride, size_t blk, size_t blk_stride,
                                               const long long* __restrict__ edges, int rows_per_block, double* __restrict__ out) {
  __shared__ double s4[4];
  const int k = blockIdx.y, tid = threadIdx.x;
  double acc = 0.0;
  for (int q = 0; q < rows_per_block; ++q) {
    const size_t r = (size_t)blockIdx.x * rows_per_block + q;
    for (int e = 0; e < 2; ++e) {
      const size_t i = (size_t)edges[2 * k + e] + tid;
      const size_t addr = blk ? (i / blk) * blk_stride + r * blk + (i % blk) : r * row_stride + i;
      acc += base[addr];
    }
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((tid & 63) == 0) s4[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) out[(size_t)k * gridDim.x + blockIdx.x] = s4[0] + s4[1] + s4[2] + s4[3];
}

int main() {
  const size_t nwav = 7200000, nrows = 112, blk = 4096;
  const size_t nblk = (nwav + blk - 1) / blk;
  double* d = nullptr; double* out = nullptr; long long* d_edges = nullptr;
  CK(hipMalloc(&d, nblk * blk * nrows * sizeof(double)));
  CK(hipMemset(d, 0, nblk * blk * nrows * sizeof(double)));
  CK(hipMalloc(&out, 1 << 20));
  CK(hipMalloc(&d_edges, 4096 * sizeof(long long)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int nint : {1, 2, 8, 16, 45}) {
    for (int layout = 0; layout < 2; ++layout) {
      float best = 1e9f, sum = 0.f;
      const int reps = 20;
      for (int rep = 0; rep < reps + 2; ++rep) {
        std::vector<long long> edges(2 * nint);
        for (int k = 0; k < 2 * nint; ++k) edges[k] = (long long)((double)rand() / RAND_MAX * (nwav - 600));
        CK(hipMemcpy(d_edges, edges.data(), edges.size() * sizeof(long long), hipMemcpyHostToDevice));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_edges, dim3(56, nint), dim3(256), 0, 0, d, layout ? 0 : nwav, layout ? blk : 0, blk * nrows, d_edges, 2, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
      }
      printf("nint %2d  %s: mean %.1f us  best %.1f us\n", nint, layout ? "blocked 4096 " : "row-major    ", 1e3f * sum / reps, 1e3f * best);
    }
  }
  return 0;
}
