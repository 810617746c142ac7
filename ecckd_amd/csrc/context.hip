// context.hip - context lifetime, device memory helpers, stream timing and the
// error channel of the C ABI (include/ecckd_hip.h).
#include "common.hpp"
#include <map>
#include <unordered_map>
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <sched.h>
#include <thread>

namespace ecckd {

int host_cores() {
  static const int cores = [] {
    if (const char* e = std::getenv("ECCKD_HOST_CORES")) {
      const int v = std::atoi(e);
      if (v > 0) return v;
    }
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) n = CPU_COUNT(&set);
    // cgroup v2 "quota period" (a container's CPU share; "max" = unlimited), then the v1 pair
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
      long long quota = 0, period = 0;
      if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
        n = std::min<long long>(n, (quota + period - 1) / period);
      std::fclose(f);
    } else if (FILE* q = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
      long long quota = -1, period = 0;
      if (std::fscanf(q, "%lld", &quota) != 1) quota = -1;
      std::fclose(q);
      if (FILE* pf = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
        if (std::fscanf(pf, "%lld", &period) != 1) period = 0;
        std::fclose(pf);
      }
      if (quota > 0 && period > 0) n = std::min<long long>(n, (quota + period - 1) / period);
    }
    // one process per GPU: the launcher's ranks on this node share the cores
    if (const char* e = std::getenv("LOCAL_WORLD_SIZE")) {
      const int lws = std::atoi(e);
      if (lws > 1) n /= lws;
    }
    return std::max(1, n);
  }();
  return cores;
}

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

namespace {
struct DevCache {
  std::unordered_map<void*, size_t> live;       // blocks handed out
  std::multimap<size_t, void*> free_blocks;     // cached, by size
  size_t cached_bytes = 0;
  size_t limit_bytes = (size_t)96 << 30;        // 288 GB of HBM per GPU: keep up to 96 GB parked (ECCKD_CACHE_GB)
};
DevCache* cache_of(ecckd_ctx* ctx) {
  if (!ctx->cache_impl) {
    DevCache* c = new DevCache;
    if (const char* e = std::getenv("ECCKD_CACHE_GB")) c->limit_bytes = (size_t)std::atof(e) * ((size_t)1 << 30);
    ctx->cache_impl = c;
  }
  return (DevCache*)ctx->cache_impl;
}
}  // namespace

namespace {
void cache_trim_locked(ecckd_ctx* ctx, DevCache* c) {
  if (!c->free_blocks.empty()) (void)hipStreamSynchronize(ctx->stream);
  for (auto& kv : c->free_blocks) (void)hipFree(kv.second);
  c->free_blocks.clear();
  c->cached_bytes = 0;
}
}  // namespace

hipError_t dev_malloc(ecckd_ctx* ctx, void** p, size_t bytes) {
  std::lock_guard<std::mutex> lock(ctx->cache_mutex);
  DevCache* c = cache_of(ctx);
  if (bytes == 0) bytes = 1;
  auto it = c->free_blocks.find(bytes);
  if (it != c->free_blocks.end()) {
    *p = it->second;
    c->cached_bytes -= bytes;
    c->free_blocks.erase(it);
    c->live[*p] = bytes;
    return hipSuccess;
  }
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipErrorOutOfMemory && !c->free_blocks.empty()) {
    (void)hipGetLastError();
    cache_trim_locked(ctx, c);
    e = hipMalloc(p, bytes);
  }
  if (e == hipSuccess) c->live[*p] = bytes;
  return e;
}

void dev_release(ecckd_ctx* ctx, void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> lock(ctx->cache_mutex);
  DevCache* c = cache_of(ctx);
  auto it = c->live.find(p);
  if (it == c->live.end()) { (void)hipFree(p); return; }   // not ours (allocated before the cache existed)
  const size_t bytes = it->second;
  c->live.erase(it);
  if (c->cached_bytes + bytes > c->limit_bytes) { (void)hipFree(p); return; }
  c->free_blocks.emplace(bytes, p);
  c->cached_bytes += bytes;
}

void dev_cache_delete(ecckd_ctx* ctx) {
  delete (DevCache*)ctx->cache_impl;
  ctx->cache_impl = nullptr;
}

void dev_cache_trim(ecckd_ctx* ctx) {
  std::lock_guard<std::mutex> lock(ctx->cache_mutex);
  if (!ctx->cache_impl) return;
  cache_trim_locked(ctx, (DevCache*)ctx->cache_impl);
}

// The lanes of a context: a stream, pinned result slots and timing events each (common.hpp).
ecckd_lane* lane_acquire(ecckd_ctx* ctx) {
  std::lock_guard<std::mutex> lock(ctx->lane_mutex);
  for (ecckd_lane* l : ctx->lanes)
    if (!l->busy) {
      l->busy = true;
      l->profile_seq = 0;
      l->stat_rt_lw = ecckd_lane_stat();
      l->stat_rt_sw = ecckd_lane_stat();
      return l;
    }
  ecckd_lane* l = new ecckd_lane();
  // (Lanes at the lowest stream priority, so that the caller's preparation of the next gas does not queue behind the searches,
  // were measured and changed nothing: configs[3] 188 against 166 ms, configs[1] 3.08 against 3.03 s.)
  if (hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&l->pev0) != hipSuccess ||
      hipEventCreate(&l->pev1) != hipSuccess) {
    if (l->stream) (void)hipStreamDestroy(l->stream);
    if (l->pev0) (void)hipEventDestroy(l->pev0);
    delete l;
    (void)fail(ECCKD_UNEXPECTED_EXCEPTION, "a stream for a side-by-side search could not be created");
    return nullptr;
  }
  l->busy = true;
  ctx->lanes.push_back(l);
  return l;
}

void lane_release(ecckd_ctx* ctx, ecckd_lane* lane) {
  if (!lane) return;
  std::lock_guard<std::mutex> lock(ctx->lane_mutex);
  auto fold = [](ecckd_lane_stat& into, const ecckd_lane_stat& from) {
    into.ms += from.ms; into.units += from.units; into.calls += from.calls;
    into.all_units += from.all_units; into.all_calls += from.all_calls;
  };
  fold(ctx->stat_rt_lw, lane->stat_rt_lw);
  fold(ctx->stat_rt_sw, lane->stat_rt_sw);
  lane->busy = false;
}

int lane_ensure_pinned(ecckd_lane* lane, size_t bytes) {
  if (bytes <= lane->pinned_bytes) return ECCKD_OK;
  if (lane->pinned) {
    ECCKD_HIP_CHECK(hipStreamSynchronize(lane->stream));
    ECCKD_HIP_CHECK(hipHostFree(lane->pinned));
    lane->pinned = nullptr;
    lane->pinned_bytes = 0;
  }
  const size_t want = ecckd_align_up(bytes * 2, 4096);
  ECCKD_HIP_CHECK(hipHostMalloc(&lane->pinned, want, hipHostMallocMapped | hipHostMallocCoherent));
  lane->pinned_bytes = want;
  return ECCKD_OK;
}

int ensure_scratch(ecckd_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->scratch_bytes) return ECCKD_OK;
  if (ctx->scratch) {
    ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ECCKD_HIP_CHECK(hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
  }
  size_t want = ecckd_align_up(bytes + bytes / 8, 1 << 20);
  ECCKD_HIP_CHECK(hipMalloc(&ctx->scratch, want));
  ctx->scratch_bytes = want;
  return ECCKD_OK;
}

int ensure_pinned(ecckd_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->pinned_bytes) return ECCKD_OK;
  if (ctx->pinned) {
    ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ECCKD_HIP_CHECK(hipHostFree(ctx->pinned));
    ctx->pinned = nullptr;
    ctx->pinned_bytes = 0;
  }
  size_t want = ecckd_align_up(bytes * 2, 4096);
  // host-coherent: results that a kernel writes here are visible to the host while the stream is still running (wait_for_slots)
  ECCKD_HIP_CHECK(hipHostMalloc(&ctx->pinned, want, hipHostMallocMapped | hipHostMallocCoherent));
  ctx->pinned_bytes = want;
  return ECCKD_OK;
}

}  // namespace ecckd

extern "C" {

int ecckd_version(void) { return 100; }

const char* ecckd_last_error(void) { return ecckd::g_err; }

int ecckd_init(int device, ecckd_ctx** out) {
  if (!out) return ecckd::fail(ECCKD_PARAMETER_ERROR, "ecckd_init: ctx is NULL");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    return ecckd::fail(ECCKD_UNEXPECTED_EXCEPTION,
                       "ecckd_init: no HIP device available (%s); there is no CPU fallback",
                       e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  }
  if (device < 0 || device >= ndev)
    return ecckd::fail(ECCKD_PARAMETER_ERROR, "ecckd_init: device %d out of range [0,%d)", device, ndev);
  ECCKD_HIP_CHECK(hipSetDevice(device));
  hipDeviceProp_t prop;
  ECCKD_HIP_CHECK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    return ecckd::fail(ECCKD_UNEXPECTED_EXCEPTION,
                       "ecckd_init: device %d is %s; this library is built for gfx950 only",
                       device, prop.gcnArchName);
  }
  ecckd_ctx* ctx = new ecckd_ctx();
  ctx->device = device;
  ctx->num_cu = prop.multiProcessorCount;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
    delete ctx;
    return ecckd::fail(ECCKD_UNEXPECTED_EXCEPTION, "ecckd_init: stream/event creation failed");
  }
  *out = ctx;
  return ECCKD_OK;
}

int ecckd_destroy(ecckd_ctx* ctx) {
  if (!ctx) return ECCKD_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ecckd::streamer_delete(ctx);
  for (ecckd_lane* l : ctx->lanes) {
    (void)hipStreamSynchronize(l->stream);
    if (l->pinned) (void)hipHostFree(l->pinned);
    (void)hipEventDestroy(l->pev0);
    (void)hipEventDestroy(l->pev1);
    (void)hipStreamDestroy(l->stream);
    delete l;
  }
  ctx->lanes.clear();
  ecckd::dev_cache_trim(ctx);
  ecckd::dev_cache_delete(ctx);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  (void)hipEventDestroy(ctx->ev0);
  (void)hipEventDestroy(ctx->ev1);
  if (ctx->pev0) (void)hipEventDestroy(ctx->pev0);
  if (ctx->pev1) (void)hipEventDestroy(ctx->pev1);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return ECCKD_OK;
}

int ecckd_trim_cache(ecckd_ctx* ctx) {
  ECCKD_REQUIRE(ctx, "ecckd_trim_cache: ctx is NULL");
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  ecckd::dev_cache_trim(ctx);
  return ECCKD_OK;
}

int ecckd_synchronize(ecckd_ctx* ctx) {
  ECCKD_REQUIRE(ctx, "ecckd_synchronize: ctx is NULL");
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return ECCKD_OK;
}

void* ecckd_stream(ecckd_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int ecckd_dev_alloc(ecckd_ctx* ctx, size_t bytes, void** d_ptr) {
  ECCKD_REQUIRE(ctx && d_ptr, "ecckd_dev_alloc: NULL argument");
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  // through the context's caching allocator: a tool allocates the same nwav-sized arrays for gas after gas, and a hipFree
  // waits for every stream of the device - with other gases' searches in flight (ecckd_find_g_gases_add) for milliseconds
  ECCKD_HIP_CHECK(ecckd::dev_malloc(ctx, d_ptr, bytes ? bytes : 1));
  return ECCKD_OK;
}

int ecckd_mem_info(ecckd_ctx* ctx, size_t* free_bytes, size_t* total_bytes) {
  ECCKD_REQUIRE(ctx && free_bytes && total_bytes, "ecckd_mem_info: NULL argument");
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  ECCKD_HIP_CHECK(hipMemGetInfo(free_bytes, total_bytes));
  return ECCKD_OK;
}

int ecckd_dev_free(ecckd_ctx* ctx, void* d_ptr) {
  ECCKD_REQUIRE(ctx, "ecckd_dev_free: ctx is NULL");
  if (!d_ptr) return ECCKD_OK;
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));     // whatever still reads it on the context's stream
  ecckd::dev_release(ctx, d_ptr);                         // parked for the next request of this size (ecckd_trim_cache frees)
  return ECCKD_OK;
}

int ecckd_h2d(ecckd_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
  ECCKD_REQUIRE(ctx && (bytes == 0 || (d_dst && h_src)), "ecckd_h2d: NULL argument");
  if (!bytes) return ECCKD_OK;
  ECCKD_HIP_CHECK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return ECCKD_OK;
}

int ecckd_d2h(ecckd_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
  ECCKD_REQUIRE(ctx && (bytes == 0 || (h_dst && d_src)), "ecckd_d2h: NULL argument");
  if (!bytes) return ECCKD_OK;
  ECCKD_HIP_CHECK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return ECCKD_OK;
}

int ecckd_profile_enable(ecckd_ctx* ctx, int on) {
  ECCKD_REQUIRE(ctx, "ecckd_profile_enable: ctx is NULL");
  if (on && !ctx->pev0) {
    ECCKD_HIP_CHECK(hipEventCreate(&ctx->pev0));
    ECCKD_HIP_CHECK(hipEventCreate(&ctx->pev1));
  }
  ctx->profile = on != 0;
  ctx->profile_stride = on > 1 ? on : 1;
  ctx->profile_seq = 0;
  ctx->stat_rt_lw = ecckd_ctx::KernelStat();
  ctx->stat_key_lw = ecckd_ctx::KernelStat();
  ctx->stat_rt_sw = ecckd_ctx::KernelStat();
  ctx->stat_sort = ecckd_ctx::KernelStat();
  ctx->stat_gases = ecckd_ctx::KernelStat();
  return ECCKD_OK;
}

int ecckd_profile_get(ecckd_ctx* ctx, const char* kernel, long long* calls, double* ms, double* units) {
  ECCKD_REQUIRE(ctx && kernel, "ecckd_profile_get: NULL argument");
  const ecckd_ctx::KernelStat* st = nullptr;
  if (!strcmp(kernel, "k_rt_lw_bb")) st = &ctx->stat_rt_lw;
  else if (!strcmp(kernel, "k_reorder_key_lw")) st = &ctx->stat_key_lw;
  else if (!strcmp(kernel, "k_rt_sw_bb")) st = &ctx->stat_rt_sw;
  else if (!strcmp(kernel, "radix_sort")) st = &ctx->stat_sort;
  else if (!strcmp(kernel, "find_g_gases")) st = &ctx->stat_gases;
  if (!strcmp(kernel, "k_rt_sw_bb.all")) {
    if (calls) *calls = ctx->stat_rt_sw.all_calls;
    if (ms) *ms = 0.0;
    if (units) *units = ctx->stat_rt_sw.all_units;
    return ECCKD_OK;
  }
  if (!strcmp(kernel, "k_rt_lw_bb.all")) {     // every launch since profile_enable, timed or not (ms = 0)
    if (calls) *calls = ctx->stat_rt_lw.all_calls;
    if (ms) *ms = 0.0;
    if (units) *units = ctx->stat_rt_lw.all_units;
    return ECCKD_OK;
  }
  if (!st) return ecckd::fail(ECCKD_PARAMETER_ERROR, "ecckd_profile_get: unknown kernel \"%s\"", kernel);
  if (calls) *calls = st->calls;
  if (ms) *ms = st->ms;
  if (units) *units = st->units;
  return ECCKD_OK;
}

int ecckd_timer_begin(ecckd_ctx* ctx) {
  ECCKD_REQUIRE(ctx, "ecckd_timer_begin: ctx is NULL");
  ECCKD_HIP_CHECK(hipEventRecord(ctx->ev0, ctx->stream));
  return ECCKD_OK;
}

int ecckd_timer_end(ecckd_ctx* ctx, float* ms) {
  ECCKD_REQUIRE(ctx && ms, "ecckd_timer_end: NULL argument");
  ECCKD_HIP_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
  ECCKD_HIP_CHECK(hipEventSynchronize(ctx->ev1));
  ECCKD_HIP_CHECK(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  return ECCKD_OK;
}

// reference src/ecckd/reorder_spectrum.cpp:121-124: T linear in ln p between
// (1 Pa, 173.15 K) and (1e5 Pa, 288.15 K); adept::interp restated as linear
// interpolation with linear extrapolation outside the knots.
int ecckd_idealised_temperature(int nhl, const double* h_pressure_hl, double* h_temperature_hl) {
  ECCKD_REQUIRE(nhl > 0 && h_pressure_hl && h_temperature_hl, "ecckd_idealised_temperature: bad argument");
  const double x0 = std::log(1.0), x1 = std::log(100000.0);
  const double y0 = 273.15 - 100.0, y1 = 273.15 + 15.0;
  for (int i = 0; i < nhl; ++i) {
    double w = (std::log(h_pressure_hl[i]) - x0) / (x1 - x0);
    h_temperature_hl[i] = (1.0 - w) * y0 + w * y1;
  }
  return ECCKD_OK;
}

// reference src/ecckd/reorder_spectrum.cpp:277-289: membership uses the
// unclamped bounds, last band closed on the right.
int ecckd_band_ranges(size_t nwav, const double* h_wavenumber, int nband,
                      const double* h_band_bound1, const double* h_band_bound2,
                      int16_t* h_iband, int64_t* h_band_begin, int64_t* h_band_end) {
  ECCKD_REQUIRE(h_wavenumber && nband > 0 && h_band_bound1 && h_band_bound2 && h_band_begin && h_band_end,
                "ecckd_band_ranges: bad argument");
  if (h_iband)
    for (size_t j = 0; j < nwav; ++j) h_iband[j] = -1;
  for (int b = 0; b < nband; ++b) {
    int64_t first = 0, last = -1;
    bool found = false;
    for (size_t j = 0; j < nwav; ++j) {
      double w = h_wavenumber[j];
      bool in = (b < nband - 1) ? (w >= h_band_bound1[b] && w < h_band_bound2[b])
                                : (w >= h_band_bound1[b] && w <= h_band_bound2[b]);
      if (in) {
        if (h_iband) h_iband[j] = (int16_t)b;
        if (!found) { first = (int64_t)j; found = true; }
        last = (int64_t)j;
      }
    }
    h_band_begin[b] = first;
    h_band_end[b] = last;
  }
  return ECCKD_OK;
}

}  // extern "C"
