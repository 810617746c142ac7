// common.hpp - context, error plumbing and launch helpers shared by the HIP
// translation units of libecckd_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstddef>
#include <mutex>
#include <vector>

#include "../../include/ecckd_hip.h"

// reference src/ecckd/constants.h:22-26
#define ECCKD_ACCEL_GRAVITY 9.80665
#define ECCKD_SPECIFIC_HEAT_AIR 1004.0
#define ECCKD_LW_DIFFUSIVITY 1.66

// What a train of error evaluations needs of its own: the stream it is issued on, pinned result slots the host watches, the
// events and counters of the optional kernel timing.  A context has its own set (the members of ecckd_ctx); ecckd_find_g_gases
// lends one lane to every gas it searches side by side with others (one host thread and one HIP stream per gas), so that a
// gas's latency-bound batches of one or two intervals run while another gas's whole-partition pass streams.
struct ecckd_lane_stat { double ms = 0.0; double units = 0.0; long long calls = 0; double all_units = 0.0; long long all_calls = 0; };
struct ecckd_lane {
  hipStream_t stream = nullptr;
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  hipEvent_t pev0 = nullptr, pev1 = nullptr;
  long long profile_seq = 0;
  ecckd_lane_stat stat_rt_lw, stat_rt_sw;
  bool busy = false;
};

struct ecckd_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int num_cu = 256;
  // growable device scratch (never shrinks; freed with the context)
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  // pinned host staging for small results
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  // optional per-kernel timing of the dominant kernels (HIP events on `stream`)
  bool profile = false;
  int profile_stride = 1;        // the sweep kernel: every profile_stride-th launch is timed (ecckd_profile_enable(ctx, stride))
  long long profile_seq = 0;
  hipEvent_t pev0 = nullptr, pev1 = nullptr;
  typedef ecckd_lane_stat KernelStat;
  KernelStat stat_rt_lw;     // k_rt_lw_bb: units = wavenumber points processed
  KernelStat stat_key_lw;    // k_reorder_key_lw: units = wavenumber points
  KernelStat stat_rt_sw;     // k_rt_sw_bb: units = wavenumber points (both fits of a total-transmission evaluation: one launch)
  KernelStat stat_sort;      // whole K3 pass sequence: units = keys sorted
  KernelStat stat_gases;     // ecckd_find_g_gases: ms = the window in which the gases' searches ran (HIP events on `stream`, which is idle
                             // meanwhile: the lanes are synchronised at both ends), units = wavenumber points their sweeps processed
  // Caching device allocator (ecckd::dev_malloc / dev_release): blocks released by a handle are kept
  // and handed out again for a request of the same size, so that preparing gas after gas (13 GB of
  // resident rows each) does not pay hipMalloc / hipFree - page-table set-up that costs hundreds of
  // milliseconds on some hosts - inside the sweep.  Everything runs on `stream`, so reuse is ordered.
  void* cache_impl = nullptr;
  std::mutex cache_mutex;          // dev_malloc / dev_release are called by the gas threads of ecckd_find_g_gases
  // lanes lent to gases searched side by side (created on demand, kept for the next call, destroyed with the context)
  std::vector<ecckd_lane*> lanes;
  std::mutex lane_mutex;
  // streaming file reader (nc_stream.hip): copy stream, pinned and device staging buffers, created on first use
  void* stream_impl = nullptr;
};

namespace ecckd {

void set_error(const char* fmt, ...);

// Returns a reference exit code (see ecckd_hip.h) after recording the message.
int fail(int code, const char* fmt, ...);

int ensure_scratch(ecckd_ctx* ctx, size_t bytes);
// hipMalloc-compatible: returns hipSuccess / hipErrorOutOfMemory (after trimming the cache and retrying)
hipError_t dev_malloc(ecckd_ctx* ctx, void** p, size_t bytes);
void dev_release(ecckd_ctx* ctx, void* p);          // back to the cache (or hipFree above the cache limit)
void dev_cache_trim(ecckd_ctx* ctx);                // hipFree every cached block
void dev_cache_delete(ecckd_ctx* ctx);
int ensure_pinned(ecckd_ctx* ctx, size_t bytes);
ecckd_lane* lane_acquire(ecckd_ctx* ctx);            // nullptr on failure (error recorded)
void lane_release(ecckd_ctx* ctx, ecckd_lane* lane); // folds the lane's timing counters into the context's
int lane_ensure_pinned(ecckd_lane* lane, size_t bytes);
void streamer_delete(ecckd_ctx* ctx);               // nc_stream.hip
// Cores THIS process may count on (context.hip): the affinity mask, capped by the cgroup's CPU quota, divided by the processes
// the launcher started on this node (LOCAL_WORLD_SIZE: one process per GPU); ECCKD_HOST_CORES overrides.  At least 1.
int host_cores();

}  // namespace ecckd

#define ECCKD_HIP_CHECK(expr)                                                          \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      return ecckd::fail(_e == hipErrorOutOfMemory ? ECCKD_OUT_OF_MEMORY               \
                                                   : ECCKD_UNEXPECTED_EXCEPTION,       \
                         "%s failed at %s:%d: %s", #expr, __FILE__, __LINE__,          \
                         hipGetErrorString(_e));                                       \
    }                                                                                  \
  } while (0)

#define ECCKD_CHECK(expr)             \
  do {                                \
    int _rc = (expr);                 \
    if (_rc != ECCKD_OK) return _rc;  \
  } while (0)

#define ECCKD_REQUIRE(cond, ...)                                        \
  do {                                                                  \
    if (!(cond)) return ecckd::fail(ECCKD_PARAMETER_ERROR, __VA_ARGS__); \
  } while (0)

static inline size_t ecckd_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
