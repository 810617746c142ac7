// nc_stream.hip - a variable (or one slice of it) of a NetCDF file straight into HBM.
//
// The reference reads every spectrum through nc_get_vara_double into a host matrix of doubles, element by element
// (DataFileEngineNetcdf.cpp:593-608), and its documentation names that reading as where much of the wall-clock time goes
// (doc/ecckd_documentation.tex:226-229, :526-528).  Here the bytes of a contiguous variable go from the file into pinned
// host buffers (several reader threads, pread), from there over PCIe on a copy stream, and a small kernel turns the
// big-endian external values into FLOAT or DOUBLE in place on the device - the read of chunk k+1 overlaps the upload and
// the decoding of chunk k, and no host core touches the values.  A 54 x 7.2e6 FLOAT spectrum is 1.5 GB: page-cache /
// disk speed decides, not a conversion loop.  Files the streamer cannot take apart (NetCDF-4 / HDF5 with its filters,
// record variables, integer types) go through the element-wise host path and one upload.
#include "common.hpp"
#include "nc_classic.hpp"
#include "nc_hdf5.hpp"

#include <atomic>
#include <cstring>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {

constexpr size_t CHUNK_BYTES = (size_t)16 << 20;
constexpr int NBUF = 6;            // pinned buffers in flight
constexpr int NREADERS = 4;        // pread threads

__device__ __forceinline__ unsigned bswap32(unsigned v) { return __builtin_bswap32(v); }
__device__ __forceinline__ unsigned long long bswap64(unsigned long long v) { return __builtin_bswap64(v); }

// raw big-endian FLOAT / DOUBLE -> native float / double
template <typename FileT, typename OutT>
__global__ void __launch_bounds__(256)
k_decode_be(size_t n, const unsigned char* __restrict__ raw, OutT* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (sizeof(FileT) == 4) {
    const unsigned u = bswap32(reinterpret_cast<const unsigned*>(raw)[i]);
    out[i] = (OutT)__uint_as_float(u);
  } else {
    const unsigned long long u = bswap64(reinterpret_cast<const unsigned long long*>(raw)[i]);
    out[i] = (OutT)__longlong_as_double((long long)u);
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
k_widen_from_double(size_t n, const double* __restrict__ in, T* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (T)in[i];
}

struct Streamer {
  hipStream_t copy = nullptr;
  void* pinned[NBUF] = {};
  void* d_raw[NBUF] = {};
  hipEvent_t done[NBUF] = {};
  bool ready = false;
};

Streamer* streamer_of(ecckd_ctx* ctx) { return (Streamer*)ctx->stream_impl; }

int ensure_streamer(ecckd_ctx* ctx) {
  if (ctx->stream_impl) return ECCKD_OK;
  Streamer* s = new Streamer;
  ctx->stream_impl = s;
  ECCKD_HIP_CHECK(hipStreamCreateWithFlags(&s->copy, hipStreamNonBlocking));
  for (int b = 0; b < NBUF; ++b) {
    ECCKD_HIP_CHECK(hipHostMalloc(&s->pinned[b], CHUNK_BYTES, hipHostMallocDefault));
    ECCKD_HIP_CHECK(hipMalloc(&s->d_raw[b], CHUNK_BYTES));
    ECCKD_HIP_CHECK(hipEventCreateWithFlags(&s->done[b], hipEventDisableTiming));
  }
  s->ready = true;
  return ECCKD_OK;
}

}  // namespace

namespace ecckd {
void streamer_delete(ecckd_ctx* ctx) {
  Streamer* s = (Streamer*)ctx->stream_impl;
  if (!s) return;
  if (s->copy) (void)hipStreamSynchronize(s->copy);
  for (int b = 0; b < NBUF; ++b) {
    if (s->pinned[b]) (void)hipHostFree(s->pinned[b]);
    if (s->d_raw[b]) (void)hipFree(s->d_raw[b]);
    if (s->done[b]) (void)hipEventDestroy(s->done[b]);
  }
  if (s->copy) (void)hipStreamDestroy(s->copy);
  delete s;
  ctx->stream_impl = nullptr;
}
}  // namespace ecckd

extern "C" {

int ecckd_nc_read_dev(ecckd_ctx* ctx, ecckd_nc* file, const char* name, long long slice, int out_type, void* d_out,
                      size_t capacity) {
  ECCKD_REQUIRE(ctx && file && name && d_out, "ecckd_nc_read_dev: NULL argument");
  ECCKD_REQUIRE(out_type == ECCKD_F32 || out_type == ECCKD_F64, "ecckd_nc_read_dev: out_type must be 4 or 8");
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  ecckd::NcSlice sl;
  ECCKD_CHECK(ecckd::nc_locate_slice(file, name, slice, &sl));
  const bool stream_it = sl.contiguous && (sl.nc_type == 5 || sl.nc_type == 6) && std::getenv("ECCKD_NO_STREAMED_READ") == nullptr;
  if (!stream_it) {
    // element-wise host path (NetCDF-4 through the HDF5 library, record variables, integer types), then one upload
    int exists = 0, type = 0, nd = 0;
    size_t sh[8] = {};
    ECCKD_CHECK(ecckd_nc_inq_var(file, name, &exists, &type, &nd, sh, 8));
    ECCKD_REQUIRE(exists, "ecckd_nc_read_dev: no variable \"%s\"", name);
    size_t n = 1;
    for (int k = (slice >= 0 ? 1 : 0); k < nd; ++k) n *= sh[k];
    ECCKD_REQUIRE(n <= capacity, "ecckd_nc_read_dev: \"%s\" needs %zu values, buffer holds %zu", name, n, capacity);
    if (ecckd::H5File* h5 = ecckd::nc_h5_handle(file)) {
      // NetCDF-4: raw chunks pulled by this thread, inflated / unshuffled by worker threads straight into the output type
      bool handled = false;
      std::vector<unsigned char> typed(n * (size_t)out_type);
      ECCKD_CHECK(ecckd::h5_read_real_parallel(h5, name, slice, out_type, typed.data(), n, &handled));
      if (handled) return ecckd_h2d(ctx, d_out, typed.data(), typed.size());
    }
    std::vector<double> host(n);
    ECCKD_CHECK(ecckd_nc_read_double(file, name, slice, host.data(), n));
    if (out_type == ECCKD_F64) return ecckd_h2d(ctx, d_out, host.data(), n * sizeof(double));
    std::vector<float> f32(host.begin(), host.end());
    return ecckd_h2d(ctx, d_out, f32.data(), n * sizeof(float));
  }
  const size_t n = (size_t)sl.count;
  ECCKD_REQUIRE(n <= capacity, "ecckd_nc_read_dev: \"%s\" needs %zu values, buffer holds %zu", name, n, capacity);
  if (n == 0) return ECCKD_OK;
  ECCKD_CHECK(ensure_streamer(ctx));
  Streamer* st = streamer_of(ctx);
  const size_t ts = sl.nc_type == 5 ? 4 : 8;
  const size_t per_chunk = CHUNK_BYTES / ts;
  const size_t nchunks = (n + per_chunk - 1) / per_chunk;

  // Reader threads fill the pinned buffers; this thread (the only one that talks to the device) ships them in order.
  // state of chunk c's buffer (c % NBUF): `filled[c]` set by its reader, the buffer freed again when its copy has run.
  std::vector<std::atomic<int>> filled(nchunks);
  for (auto& f : filled) f.store(0, std::memory_order_relaxed);
  std::atomic<long long> shipped{-1};          // highest chunk whose buffer is free again
  std::atomic<int> failed{0};
  auto reader = [&](int r) {
    for (size_t c = (size_t)r; c < nchunks; c += NREADERS) {
      // buffer c % NBUF is free once chunk c - NBUF has been copied
      while ((long long)c - NBUF > shipped.load(std::memory_order_acquire) && !failed.load()) std::this_thread::yield();
      if (failed.load()) return;
      const size_t first = c * per_chunk, cnt = std::min(per_chunk, n - first);
      size_t got = 0;
      const size_t want = cnt * ts;
      unsigned char* dst = (unsigned char*)st->pinned[c % NBUF];
      while (got < want) {
        const ssize_t k = pread(sl.fd, dst + got, want - got, (off_t)(sl.offset + first * ts + got));
        if (k <= 0) { failed.store(1); return; }
        got += (size_t)k;
      }
      filled[c].store(1, std::memory_order_release);
    }
  };
  std::vector<std::thread> threads;
  const int nreaders = (int)std::min<size_t>(NREADERS, nchunks);
  for (int r = 0; r < nreaders; ++r) threads.emplace_back(reader, r);
  int rc = ECCKD_OK;
  for (size_t c = 0; c < nchunks && rc == ECCKD_OK; ++c) {
    while (!filled[c].load(std::memory_order_acquire) && !failed.load()) std::this_thread::yield();
    if (failed.load()) { rc = ecckd::fail(ECCKD_PROCESSING_ERROR, "ecckd_nc_read_dev: short read of \"%s\"", name); break; }
    const int b = (int)(c % NBUF);
    const size_t first = c * per_chunk, cnt = std::min(per_chunk, n - first);
    hipError_t e = hipMemcpyAsync(st->d_raw[b], st->pinned[b], cnt * ts, hipMemcpyHostToDevice, st->copy);
    if (e == hipSuccess) e = hipEventRecord(st->done[b], st->copy);
    const unsigned blocks = (unsigned)((cnt + 255) / 256);
    if (e == hipSuccess) {
      const unsigned char* raw = (const unsigned char*)st->d_raw[b];
      if (ts == 4 && out_type == ECCKD_F32) hipLaunchKernelGGL((k_decode_be<float, float>), dim3(blocks), dim3(256), 0, st->copy, cnt, raw, (float*)d_out + first);
      else if (ts == 4) hipLaunchKernelGGL((k_decode_be<float, double>), dim3(blocks), dim3(256), 0, st->copy, cnt, raw, (double*)d_out + first);
      else if (out_type == ECCKD_F32) hipLaunchKernelGGL((k_decode_be<double, float>), dim3(blocks), dim3(256), 0, st->copy, cnt, raw, (float*)d_out + first);
      else hipLaunchKernelGGL((k_decode_be<double, double>), dim3(blocks), dim3(256), 0, st->copy, cnt, raw, (double*)d_out + first);
      e = hipGetLastError();
    }
    if (e != hipSuccess) { failed.store(1); rc = ecckd::fail(ECCKD_UNEXPECTED_EXCEPTION, "ecckd_nc_read_dev: %s", hipGetErrorString(e)); break; }
    // the pinned buffer of chunk c is free once its copy has run; the device staging buffer once its decode has: the next
    // user of both is chunk c + NBUF, which is issued on the same stream (ordered behind) - only the HOST buffer needs a wait
    if (c + 1 >= (size_t)NBUF) {
      const size_t old = c + 1 - NBUF;            // make buffer (old % NBUF) available to the readers
      (void)hipEventSynchronize(st->done[old % NBUF]);
      shipped.store((long long)old, std::memory_order_release);
    }
  }
  if (rc != ECCKD_OK) failed.store(1);
  shipped.store((long long)nchunks, std::memory_order_release);
  for (std::thread& t : threads) t.join();
  const hipError_t e = hipStreamSynchronize(st->copy);
  if (rc == ECCKD_OK && e != hipSuccess) rc = ecckd::fail(ECCKD_UNEXPECTED_EXCEPTION, "ecckd_nc_read_dev: %s", hipGetErrorString(e));
  return rc;
}

}  // extern "C"
