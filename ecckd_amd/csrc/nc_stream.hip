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

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <fcntl.h>
#include <unistd.h>
#include <vector>

namespace {

constexpr size_t CHUNK_BYTES = (size_t)16 << 20;
constexpr int NBUF = 12;           // pinned buffers in flight
// pread threads (ECCKD_READ_THREADS): one per buffer by default - 503 MB from the page cache in 34-36 ms the first time and 14-18 ms
// after with six buffers and threads, against 51-53 / 30-38 ms with four threads striding over six buffers
// (tools/classic_read_probe.py); twelve of each: create_look_up_table over 23 GB of spectra 2.23 -> 1.84 s (tools/e2e_bench.py)
constexpr int NREADERS_DEFAULT = NBUF, NREADERS_MAX = NBUF;
inline int nreaders_wanted() {
  static const int n = [] {
    const char* e = std::getenv("ECCKD_READ_THREADS");
    const int v = e ? std::atoi(e) : NREADERS_DEFAULT;
    return v < 1 ? 1 : v > NREADERS_MAX ? NREADERS_MAX : v;
  }();
  return n;
}

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

struct H5Dev {                 // buffers of the device-inflate path (read_h5_dev), kept with the streamer
  void* pinned[2] = {};        // raw chunks of a batch, as read from the file
  void* d_comp[2] = {};
  hipEvent_t done[2] = {};
  size_t comp_cap = 0;
  void* d_plain = nullptr;     // inflated bytes of a batch
  size_t plain_cap = 0;
  void* d_meta = nullptr;      // per batch: descriptors | origins | shuffled flags | status
  void* h_meta[2] = {};
  size_t meta_cap = 0;
};

struct Streamer {
  hipStream_t copy = nullptr;
  void* pinned[NBUF] = {};
  void* d_raw[NBUF] = {};
  hipEvent_t done[NBUF] = {};
  hipEvent_t entry = nullptr;    // recorded on the context's stream when a read begins: the copy stream starts behind it
  bool ready = false;
  void* h5dev = nullptr;         // H5Dev: buffers of the device-inflate path (created on first use)
};

Streamer* streamer_of(ecckd_ctx* ctx) { return (Streamer*)ctx->stream_impl; }

int ensure_streamer(ecckd_ctx* ctx) {
  if (ctx->stream_impl) return ECCKD_OK;
  Streamer* s = new Streamer;
  ctx->stream_impl = s;
  ECCKD_HIP_CHECK(hipStreamCreateWithFlags(&s->copy, hipStreamNonBlocking));
  // one pinned and one device allocation cut into NBUF pieces: six calls of each cost three times as much of a tool's start
  ECCKD_HIP_CHECK(hipHostMalloc(&s->pinned[0], NBUF * CHUNK_BYTES, hipHostMallocDefault));
  ECCKD_HIP_CHECK(hipMalloc(&s->d_raw[0], NBUF * CHUNK_BYTES));
  for (int b = 0; b < NBUF; ++b) {
    s->pinned[b] = (char*)s->pinned[0] + (size_t)b * CHUNK_BYTES;
    s->d_raw[b] = (char*)s->d_raw[0] + (size_t)b * CHUNK_BYTES;
    ECCKD_HIP_CHECK(hipEventCreateWithFlags(&s->done[b], hipEventDisableTiming));
  }
  ECCKD_HIP_CHECK(hipEventCreateWithFlags(&s->entry, hipEventDisableTiming));
  s->ready = true;
  return ECCKD_OK;
}

}  // namespace

namespace ecckd {
void streamer_delete(ecckd_ctx* ctx) {
  Streamer* s = (Streamer*)ctx->stream_impl;
  if (!s) return;
  if (s->copy) (void)hipStreamSynchronize(s->copy);
  if (s->pinned[0]) (void)hipHostFree(s->pinned[0]);
  if (s->d_raw[0]) (void)hipFree(s->d_raw[0]);
  for (int b = 0; b < NBUF; ++b)
    if (s->done[b]) (void)hipEventDestroy(s->done[b]);
  if (s->copy) (void)hipStreamDestroy(s->copy);
  if (H5Dev* d = (H5Dev*)s->h5dev) {
    for (int k = 0; k < 2; ++k) {
      if (d->pinned[k]) (void)hipHostFree(d->pinned[k]);
      if (d->d_comp[k]) (void)hipFree(d->d_comp[k]);
      if (d->h_meta[k]) (void)hipHostFree(d->h_meta[k]);
      if (d->done[k]) (void)hipEventDestroy(d->done[k]);
    }
    if (d->d_plain) (void)hipFree(d->d_plain);
    if (d->d_meta) (void)hipFree(d->d_meta);
    delete d;
  }
  delete s;
  ctx->stream_impl = nullptr;
}
}  // namespace ecckd

namespace ecckd {
size_t inflate_in_slack();
int inflate_launch(ecckd_ctx* ctx, hipStream_t stream, int nstreams, const void* d_in, const void* d_desc, void* d_out, int* d_status);
}  // namespace ecckd

namespace {

// what the inflate kernel reads per stream (inflate.hip: StreamDesc)
struct InfDesc { unsigned long long in_off, in_bytes, out_off, out_bytes; };

struct PlaceGeom {
  int nd, shuffle, ts, out_ts;
  unsigned long long cdims[8], lo[8], hi[8];
  unsigned long long chunk_elems;
};

// The values of the chunks of one batch (inflated bytes at plain + c * chunk_elems * ts, byte-shuffled or not per chunk) to their
// places in the requested box: undo the shuffle filter (byte b of value e sits at b * chunk_elems + e), convert to the output
// type.  grid (value tiles, chunks); consecutive threads take consecutive values of the chunk's innermost dimension.
__global__ void __launch_bounds__(256)
k_place_chunks(PlaceGeom G, const unsigned char* __restrict__ plain, const unsigned long long* __restrict__ origin /* [nchunks][8] */,
               const int* __restrict__ shuffled, void* __restrict__ out) {
  const unsigned long long e = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= G.chunk_elems) return;
  const int c = blockIdx.y;
  const unsigned long long* org = origin + (size_t)c * 8;
  // position inside the chunk -> dataset coordinates -> position in the box
  unsigned long long rem = e, dst = 0, stride = 1;
  bool inside = true;
  for (int k = G.nd - 1; k >= 0; --k) {
    const unsigned long long i = rem % G.cdims[k];
    rem /= G.cdims[k];
    const unsigned long long x = org[k] + i;
    inside = inside && x >= G.lo[k] && x < G.hi[k];
    dst += (x - G.lo[k]) * stride;
    stride *= G.hi[k] - G.lo[k];
  }
  if (!inside) return;
  const unsigned char* p = plain + (size_t)c * G.chunk_elems * G.ts;
  unsigned long long bits = 0;
  if (shuffled[c]) {
    for (int b = 0; b < G.ts; ++b) bits |= (unsigned long long)p[(size_t)b * G.chunk_elems + e] << (8 * b);
  } else {
    for (int b = 0; b < G.ts; ++b) bits |= (unsigned long long)p[e * G.ts + b] << (8 * b);
  }
  double v;
  if (G.ts == 4) v = (double)__uint_as_float((unsigned)bits);
  else v = __longlong_as_double((long long)bits);
  if (G.out_ts == 4) {
    if (G.ts == 4) reinterpret_cast<unsigned*>(out)[dst] = (unsigned)bits;     // FLOAT stays the same bits
    else reinterpret_cast<float*>(out)[dst] = (float)v;
  } else {
    reinterpret_cast<double*>(out)[dst] = v;
  }
}

}  // namespace

namespace ecckd {

// One slice (or all) of a chunked, deflated FLOAT / DOUBLE variable of a NetCDF-4 file, straight to the device.  This thread
// pulls the RAW chunks out of the file (H5Dread_chunk: the HDF5 library is not thread-safe); the inflated - still
// byte-shuffled - chunks are collected in device memory and ONE launch of k_place_chunks undoes the shuffle filter, converts
// and places them.  Who inflates:
//   host (default): worker threads (zlib, ~0.65 GB/s each) inflate into a ring of pinned slots which this thread ships as they
//     fill - the reading, the inflating and the upload overlap, and no host core unshuffles, converts or copies values;
//   device (ECCKD_GPU_INFLATE=1): the raw chunks are shipped as they are (about half the bytes) and every chunk is inflated by one
//     wavefront (inflate.hip, ~6 MB/s per wavefront, 1 024 at a time: ~6 GB/s on a whole chip).  With the 16 host cores a GPU
//     box has per GPU the worker threads are faster (~10 GB/s); the device path is for hosts whose cores are busy or few.
// *handled = false: layout not supported, or an unwritten chunk (the caller falls back to the host-only paths).
static int read_h5_dev(ecckd_ctx* ctx, H5File* h5, const char* name, long long slice, int out_type, void* d_out, size_t capacity,
                       bool on_device, bool* handled) {
  *handled = false;
  if (!on_device && !h5_has_zlib(h5)) return ECCKD_OK;
  H5ChunkReader* rd = nullptr;
  H5ChunkPlan P;
  ECCKD_CHECK(h5_chunks_open(h5, name, slice, capacity, &rd, &P));
  if (!rd) return ECCKD_OK;
  const size_t chunk_bytes = P.chunk_elems * P.ts;
  const size_t slack = inflate_in_slack();
  if (chunk_bytes >= ((size_t)1 << 31) || P.nchunks >= ((size_t)1 << 30)) { h5_chunks_close(rd); return ECCKD_OK; }
  struct Closer { H5ChunkReader* r; ~Closer() { h5_chunks_close(r); } } closer{rd};
  ECCKD_CHECK(ensure_streamer(ctx));
  Streamer* st = streamer_of(ctx);
  if (!st->h5dev) st->h5dev = new H5Dev;
  H5Dev* D = (H5Dev*)st->h5dev;
  auto grow = [&](void** p, size_t* cap, size_t want, bool host) -> int {
    if (*cap >= want) return ECCKD_OK;
    (void)hipStreamSynchronize(ctx->stream);
    if (*p) { if (host) (void)hipHostFree(*p); else (void)hipFree(*p); *p = nullptr; *cap = 0; }
    if (host) ECCKD_HIP_CHECK(hipHostMalloc(p, want, hipHostMallocDefault));
    else ECCKD_HIP_CHECK(hipMalloc(p, want));
    *cap = want;
    return ECCKD_OK;
  };
  const size_t n = P.nchunks;
  // device: inflated chunks | origins | shuffled flags | (device inflate) descriptors, status, raw chunks
  const size_t meta_bytes = ecckd_align_up(n * (8 * sizeof(unsigned long long) + sizeof(int) + sizeof(InfDesc) + sizeof(int)), 256);
  const size_t raw_room = on_device ? n * ecckd_align_up(chunk_bytes + chunk_bytes / 512 + 64 + slack, 16) : 0;   // deflate never grows a chunk by more
  const size_t plain_bytes = ecckd_align_up(n * chunk_bytes, 256);
  ECCKD_CHECK(grow(&D->d_plain, &D->plain_cap, plain_bytes + meta_bytes + raw_room, false));
  unsigned char* d_plain = (unsigned char*)D->d_plain;
  unsigned long long* d_org = (unsigned long long*)(d_plain + plain_bytes);
  int* d_shuf = (int*)(d_org + n * 8);
  InfDesc* d_desc = (InfDesc*)(d_plain + plain_bytes + ecckd_align_up(n * (8 * sizeof(unsigned long long) + sizeof(int)), 16));
  int* d_status = (int*)(d_desc + n);
  unsigned char* d_raw = d_plain + plain_bytes + meta_bytes;
  // pinned staging: two buffers (device inflate: raw chunks as they come; host inflate: a ring of slots of one chunk each)
  const size_t stage_bytes = std::max((size_t)64 << 20, 4 * (chunk_bytes + slack + 64));
  for (int k = 0; k < 2; ++k) {
    size_t cap = D->comp_cap;
    ECCKD_CHECK(grow(&D->pinned[k], &cap, stage_bytes, true));
    if (!D->done[k]) ECCKD_HIP_CHECK(hipEventCreateWithFlags(&D->done[k], hipEventDisableTiming));
  }
  D->comp_cap = std::max(D->comp_cap, stage_bytes);
  std::vector<unsigned long long> org(n * 8, 0);
  std::vector<int> shuf(n, 0);
  std::vector<InfDesc> desc(on_device ? n : 0);
  bool fallback = false;
  int rc = ECCKD_OK;

  if (on_device) {
    int buf = 0;
    bool used[2] = {false, false};
    size_t filled = 0, raw_total = 0;          // bytes in the open staging buffer; bytes shipped before it
    auto ship = [&]() -> int {
      if (filled == 0) return ECCKD_OK;
      ECCKD_HIP_CHECK(hipMemcpyAsync(d_raw + raw_total, D->pinned[buf], filled, hipMemcpyHostToDevice, ctx->stream));
      ECCKD_HIP_CHECK(hipEventRecord(D->done[buf], ctx->stream));
      used[buf] = true;
      raw_total += filled;
      filled = 0;
      buf ^= 1;
      if (used[buf]) ECCKD_HIP_CHECK(hipEventSynchronize(D->done[buf]));
      return ECCKD_OK;
    };
    for (size_t j = 0; j < n; ++j) {
      size_t bytes = 0;
      int unwritten = 0;
      ECCKD_CHECK(h5_chunks_next(rd, &org[j * 8], &bytes, &unwritten));
      const size_t need = ecckd_align_up(bytes + slack, 16);
      if (unwritten || bytes == 0 || need > stage_bytes || raw_total + filled + need > raw_room) { fallback = true; break; }
      if (filled + need > stage_bytes) ECCKD_CHECK(ship());
      int deflated = 0, shuffled = 0;
      ECCKD_CHECK(h5_chunks_read(rd, (char*)D->pinned[buf] + filled, &deflated, &shuffled));
      if (!deflated) { fallback = true; break; }          // a chunk stored without the deflate filter: left to the host path
      std::memset((char*)D->pinned[buf] + filled + bytes, 0, need - bytes);
      desc[j].in_off = raw_total + filled; desc[j].in_bytes = bytes;
      desc[j].out_off = j * chunk_bytes; desc[j].out_bytes = chunk_bytes;
      shuf[j] = shuffled;
      filled += need;
    }
    if (!fallback) {
      ECCKD_CHECK(ship());
      ECCKD_CHECK(ecckd_h2d(ctx, d_desc, desc.data(), n * sizeof(InfDesc)));
      ECCKD_CHECK(inflate_launch(ctx, ctx->stream, (int)n, d_raw, d_desc, d_plain, d_status));
    }
  } else {
    // ---- worker threads inflate into pinned slots, this thread reads the raw chunks and ships the filled slots ----
    const size_t slot_bytes = ecckd_align_up(chunk_bytes, 256);
    const size_t per_buf = stage_bytes / slot_bytes, nslots = 2 * per_buf;
    auto slot_ptr = [&](size_t s) { return (unsigned char*)D->pinned[s / per_buf] + (s % per_buf) * slot_bytes; };
    // the raw bytes of a chunk come from the calling thread (H5Dread_chunk: the library is not thread-safe, ~7 GB/s) or, where
    // the library can say where the chunk lies in the file (1.10.5 on), from the worker itself with pread: the calling thread
    // then only walks the chunk index
    struct Job { std::vector<unsigned char> raw; unsigned long long addr = ~0ull; size_t bytes = 0; };
    std::vector<Job> jobs(n);
    const bool no_locate = std::getenv("ECCKD_H5_SERIAL_READ") != nullptr;   // (per call: the probe and the tests switch it)
    int fd = -1;
    if (h5_can_locate(h5) && !no_locate) fd = open(h5_path(h5), O_RDONLY);
    struct FdCloser { int& f; ~FdCloser() { if (f >= 0) close(f); } } fd_closer{fd};
    std::vector<std::atomic<int>> state(n);               // 0 not read yet, 1 raw bytes there, 2 inflated into its slot, -1 failed
    for (auto& s_ : state) s_.store(0, std::memory_order_relaxed);
    std::atomic<size_t> next{0};
    std::atomic<long long> freed{(long long)nslots - 1};   // chunk j may use slot j % nslots once chunk j - nslots has been shipped: j <= freed
    std::atomic<int> stop{0};
    const unsigned nworkers = (unsigned)std::max(1, std::min(16, ecckd::host_cores()));
    auto worker = [&]() {
      for (;;) {
        const size_t j = next.fetch_add(1);
        if (j >= n) return;
        while ((state[j].load(std::memory_order_acquire) == 0 || (long long)j > freed.load(std::memory_order_acquire)) && !stop.load())
          std::this_thread::yield();
        if (stop.load()) return;
        if (jobs[j].addr != ~0ull) {
          jobs[j].raw.resize(jobs[j].bytes);
          size_t got = 0;
          while (got < jobs[j].bytes) {
            const ssize_t k = pread(fd, jobs[j].raw.data() + got, jobs[j].bytes - got, (off_t)(jobs[j].addr + got));
            if (k <= 0) break;
            got += (size_t)k;
          }
          if (got != jobs[j].bytes) { state[j].store(-1, std::memory_order_release); stop.store(1); return; }
        }
        const bool ok = h5_inflate_host(h5, slot_ptr(j % nslots), chunk_bytes, jobs[j].raw.data(), jobs[j].raw.size());
        std::vector<unsigned char>().swap(jobs[j].raw);
        state[j].store(ok ? 2 : -1, std::memory_order_release);
        if (!ok) { stop.store(1); return; }
      }
    };
    std::vector<std::thread> pool;
    for (unsigned w = 0; w < nworkers; ++w) pool.emplace_back(worker);
    size_t shipped = 0;                        // chunks whose slot has been handed to the copy engine
    std::vector<hipEvent_t> ev;                // one event per batch of shipped slots, to free them
    std::vector<size_t> ev_upto;
    size_t ev_done = 0;
    auto ship_ready = [&](bool wait_all) -> int {
      for (;;) {
        size_t k = shipped;
        while (k < n && state[k].load(std::memory_order_acquire) == 2) ++k;
        if (k > shipped) {
          // consecutive slots in the ring are consecutive in memory inside one staging buffer: one copy per run
          size_t a0 = shipped;
          while (a0 < k) {
            const size_t s0 = a0 % nslots, in_buf = per_buf - (s0 % per_buf);
            const size_t cnt = std::min(k - a0, in_buf);
            if (slot_bytes == chunk_bytes)
              ECCKD_HIP_CHECK(hipMemcpyAsync(d_plain + a0 * chunk_bytes, slot_ptr(s0), cnt * chunk_bytes, hipMemcpyHostToDevice, ctx->stream));
            else
              for (size_t q = 0; q < cnt; ++q)
                ECCKD_HIP_CHECK(hipMemcpyAsync(d_plain + (a0 + q) * chunk_bytes, slot_ptr(s0 + q), chunk_bytes, hipMemcpyHostToDevice, ctx->stream));
            a0 += cnt;
          }
          hipEvent_t e;
          ECCKD_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
          ECCKD_HIP_CHECK(hipEventRecord(e, ctx->stream));
          ev.push_back(e);
          ev_upto.push_back(k);
          shipped = k;
        }
        // slots whose copy has run are free again
        while (ev_done < ev.size() && hipEventQuery(ev[ev_done]) == hipSuccess) {
          freed.store((long long)ev_upto[ev_done] - 1 + (long long)nslots, std::memory_order_release);
          ++ev_done;
        }
        if (!wait_all || shipped == n || stop.load()) return ECCKD_OK;
        for (size_t q = shipped; q < n; ++q) if (state[q].load() < 0) return ECCKD_OK;
        std::this_thread::yield();
      }
    };
    const auto t_begin = std::chrono::steady_clock::now();
    for (size_t j = 0; j < n && rc == ECCKD_OK && !stop.load(); ++j) {
      size_t bytes = 0;
      int unwritten = 0;
      rc = h5_chunks_next(rd, &org[j * 8], &bytes, &unwritten);
      if (rc != ECCKD_OK) break;
      if (unwritten || bytes == 0) { fallback = true; break; }
      int deflated = 0, shuffled = 0;
      if (fd >= 0 && j > 0) {
        rc = h5_chunks_locate(rd, &jobs[j].addr, &deflated, &shuffled);
        if (rc != ECCKD_OK) break;
        if (jobs[j].addr == ~0ull) { fallback = true; break; }
        jobs[j].bytes = bytes;
      } else {
        // the first chunk through the library, and - where chunks can be located - once more with pread: the same bytes, or
        // the addresses are not file offsets (a user block) and every chunk goes through the library
        unsigned long long a0 = ~0ull;
        if (fd >= 0) {
          int d0 = 0, s0 = 0;
          rc = h5_chunks_locate(rd, &a0, &d0, &s0, false);
          if (rc != ECCKD_OK) break;
        }
        jobs[j].raw.resize(bytes);
        rc = h5_chunks_read(rd, jobs[j].raw.data(), &deflated, &shuffled);
        if (rc != ECCKD_OK) break;
        if (fd >= 0) {
          std::vector<unsigned char> again(bytes);
          size_t got = 0;
          while (a0 != ~0ull && got < bytes) {
            const ssize_t k = pread(fd, again.data() + got, bytes - got, (off_t)(a0 + got));
            if (k <= 0) break;
            got += (size_t)k;
          }
          if (got != bytes || std::memcmp(again.data(), jobs[j].raw.data(), bytes) != 0) { close(fd); fd = -1; }
        }
      }
      if (!deflated) { fallback = true; break; }
      shuf[j] = shuffled;
      state[j].store(1, std::memory_order_release);
      rc = ship_ready(false);
    }
    const auto t_walked = std::chrono::steady_clock::now();
    if (rc != ECCKD_OK || fallback) stop.store(1);
    if (!stop.load()) rc = ship_ready(true);
    const bool inflate_failed = stop.load() && rc == ECCKD_OK && !fallback;
    stop.store(1);
    for (std::thread& th : pool) th.join();
    (void)hipStreamSynchronize(ctx->stream);
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    if (std::getenv("ECCKD_H5_TIMES")) {
      const auto t_end = std::chrono::steady_clock::now();
      std::fprintf(stderr, "%s \"%s\": %zu chunks, %s by the calling thread in %.1f ms, all inflated and shipped %.1f ms later (%u threads)\n",
                   h5_path(h5), name, n, fd >= 0 ? "located" : "read", std::chrono::duration<double, std::milli>(t_walked - t_begin).count(),
                   std::chrono::duration<double, std::milli>(t_end - t_walked).count(), nworkers);
    }
    if (rc != ECCKD_OK) return rc;
    if (inflate_failed) return fail(ECCKD_PROCESSING_ERROR, "a chunk of \"%s\" could not be inflated", name);
  }
  if (fallback) { (void)hipStreamSynchronize(ctx->stream); return ECCKD_OK; }

  PlaceGeom G;
  G.nd = P.nd; G.shuffle = P.shuffle; G.ts = (int)P.ts; G.out_ts = out_type; G.chunk_elems = P.chunk_elems;
  for (int k = 0; k < 8; ++k) { G.cdims[k] = k < P.nd ? P.cdims[k] : 1; G.lo[k] = k < P.nd ? P.lo[k] : 0; G.hi[k] = k < P.nd ? P.hi[k] : 1; }
  ECCKD_CHECK(ecckd_h2d(ctx, d_org, org.data(), n * 8 * sizeof(unsigned long long)));
  ECCKD_CHECK(ecckd_h2d(ctx, d_shuf, shuf.data(), n * sizeof(int)));
  for (size_t c0 = 0; c0 < n; c0 += 32768) {
    const size_t cnt = std::min<size_t>(32768, n - c0);
    hipLaunchKernelGGL(k_place_chunks, dim3((unsigned)((P.chunk_elems + 255) / 256), (unsigned)cnt), dim3(256), 0, ctx->stream, G,
                       (const unsigned char*)d_plain + c0 * chunk_bytes, d_org + c0 * 8, d_shuf + c0, d_out);
  }
  ECCKD_HIP_CHECK(hipGetLastError());
  if (on_device) {
    std::vector<int> status(n);
    ECCKD_CHECK(ecckd_d2h(ctx, status.data(), d_status, n * sizeof(int)));
    for (size_t k = 0; k < n; ++k)
      if (status[k] != 0)
        return fail(ECCKD_PROCESSING_ERROR, "chunk %zu of \"%s\" is not a deflate stream of the chunk's size (code %d)", k, name, status[k]);
  }
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  *handled = true;
  return ECCKD_OK;
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
      bool handled = false;
      // NetCDF-4: raw chunks straight to the device, inflated by worker threads (default) or on the device (ECCKD_GPU_INFLATE=1),
      // unshuffled / converted / placed by a kernel; ECCKD_NO_DEVICE_PLACE=1: the host-only path below
      const char* gi = std::getenv("ECCKD_GPU_INFLATE");
      if (!std::getenv("ECCKD_NO_DEVICE_PLACE") && !std::getenv("ECCKD_NO_PARALLEL_INFLATE")) {
        ECCKD_CHECK(ecckd::read_h5_dev(ctx, h5, name, slice, out_type, d_out, n, gi && gi[0] == '1', &handled));
        if (handled) return ECCKD_OK;
      }
      // raw chunks pulled by this thread, inflated / unshuffled / placed by worker threads, one upload
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
  // like every other entry point this one is ordered behind what the caller has queued on the context's stream: work there
  // that still reads or writes d_out finishes before the private copy stream touches it (the read itself ends synchronously)
  ECCKD_HIP_CHECK(hipEventRecord(st->entry, ctx->stream));
  ECCKD_HIP_CHECK(hipStreamWaitEvent(st->copy, st->entry, 0));
  const size_t ts = sl.nc_type == 5 ? 4 : 8;
  const size_t per_chunk = CHUNK_BYTES / ts;
  const size_t nchunks = (n + per_chunk - 1) / per_chunk;

  // Reader threads fill the pinned buffers; this thread (the only one that talks to the device) ships them in order.
  // state of chunk c's buffer (c % NBUF): `filled[c]` set by its reader, the buffer freed again when its copy has run.
  std::vector<std::atomic<int>> filled(nchunks);
  for (auto& f : filled) f.store(0, std::memory_order_relaxed);
  std::atomic<long long> shipped{-1};          // highest chunk whose buffer is free again
  std::atomic<int> failed{0};
  const int nreaders = (int)std::min<size_t>((size_t)nreaders_wanted(), nchunks);
  auto reader = [&](int r) {
    for (size_t c = (size_t)r; c < nchunks; c += (size_t)nreaders) {
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
