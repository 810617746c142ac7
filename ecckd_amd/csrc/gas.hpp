// gas.hpp - the prepared-gas handle shared by find_g.hip and find_g_band.hip.
#pragma once
#include "common.hpp"
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace ecckd {
bool host_oversubscribed();     // find_g_band.hip: more spinning search / batcher threads than cores (they then yield while waiting)
}

// Row table of a gas: which summable per-point rows exist and where (indices into
// ecckd_gas::rows / the per-interval sums).  Blocks of nlay consecutive rows start at the
// given offsets; -1 = absent.
struct RowMap {
  int A = -1;     // numerator of the fit: metric * weight      (log: log(metric) * weight, metric > 0)
  int B = -1;     // denominator of the fit: weight             (log: weight where metric > 0)
  int N = -1;     // log only: count of metric > 0
  int H = -1;     // true heating rate
  int FDS = -1, FUT = -1;  // true surface-down / TOA-up flux (single rows)
  // shortwave total-transmission extras (find_g_points.cpp:171-204, :263-278)
  int TF = -1, TG = -1;    // direct-beam flux below each layer with / without the target gas
  int HL = -1, HH = -1;    // true heating rate for min_scaling / max_scaling
  int FDSL = -1, FUTL = -1, FDSH = -1, FUTH = -1;
  int total = 0;
};

// An interval's error is a function of the interval alone (its first and last sorted index and, in the shortwave, the
// surface albedo of its band): the partition search asks for the same interval again and again - calc_error_all
// re-evaluates every interval of a partition of which one bound has moved (equipartition.h:98-116), line_search and the
// pairwise shuffles come back to bounds they have seen (equipartition.cpp:162-196, :499-531) - and gets the same bits
// from the memo as from the device.
struct IntervalKey {
  long long i1, i2;
  unsigned long long albedo_bits;
  bool operator==(const IntervalKey& o) const { return i1 == o.i1 && i2 == o.i2 && albedo_bits == o.albedo_bits; }
};
struct IntervalKeyHash {
  size_t operator()(const IntervalKey& k) const {
    unsigned long long h = (unsigned long long)k.i1 * 0x9E3779B97F4A7C15ULL;
    h ^= (unsigned long long)k.i2 + 0x9E3779B97F4A7C15ULL + (h << 6) + (h >> 2);
    h ^= k.albedo_bits + 0x9E3779B97F4A7C15ULL + (h << 6) + (h >> 2);
    return (size_t)h;
  }
};

// ---------------------------------------------------------------------------
// opaque handle
struct ecckd_gas {
  ecckd_ctx* ctx = nullptr;
  int do_sw = 0;
  int method = 0;
  int nlay = 0;
  size_t n = 0;  // wavenumbers (sorted order)
  double flux_weight = 0.0;
  double total_comp_cost = 0.0;
  // memo of interval errors and its counters: intervals asked for / found in the memo, points asked for / swept on the device
  std::unordered_map<IntervalKey, double, IntervalKeyHash> error_memo;
  long long memo_requests = 0, memo_hits = 0;
  double points_requested = 0.0, points_evaluated = 0.0;
  // device arrays, all in sorted order
  double* planck_hl = nullptr;  // [nlay+1][n]
  bool owns_planck = true;
  double* bg_od = nullptr;      // [nlay][n]
  float* bg_pair = nullptr;     // [nlay/2][n][2]  longwave, only if every background value is a float: layers 2p, 2p+1 of a point side by side
  double* w1 = nullptr;         // [nlay][n]  metric * weight          (log: log(metric)*weight)
  double* w2 = nullptr;         // [nlay][n]  log only: weight of the denominator where metric > 0
  double* cnt = nullptr;        // [nlay][n]  log only: 1 where metric > 0
  double* hr = nullptr;         // [nlay][n]
  double* fds = nullptr;        // [n] flux_dn_surf
  double* fut = nullptr;        // [n] flux_up_toa
  double* wn_sorted = nullptr;  // [n]
  double* dwn_sorted = nullptr; // [n]
  int32_t* ireorder = nullptr;  // [n]
  // shortwave
  double cos_sza = 0.5;
  double surf_albedo = 0.0;  // band albedo of CkdEquipartition::init_sw (find_g_points.cpp:237-261)
  double min_scaling = 1.0, max_scaling = 1.0;
  double* ssi = nullptr;        // [n] sorted
  double* tf = nullptr;         // [nlay][n] total-transmission: direct flux below layer l, bg+target
  double* tg = nullptr;         // [nlay][n]                      ... background only
  double* hr_low = nullptr;     // [nlay][n]
  double* hr_high = nullptr;    // [nlay][n]
  double* fx = nullptr;         // [4][n] fds_low, fut_low, fds_high, fut_high
  // row table for interval sums
  RowMap rm;
  int nrows = 0;
  const double** rows = nullptr;  // device array of nrows row pointers
  double* tile_sums = nullptr;    // [nrows][ntiles]   sums over 256 points
  size_t ntiles = 0;
  double* super_sums = nullptr;   // [nrows][nsuper]   sums over 256 tiles
  size_t nsuper = 0;
  // per-level constants on device: conv[nlay] | layer_weight[nlay]
  double* lev = nullptr;
  std::vector<double> h_pressure_hl;
  std::vector<double> h_layer_weight;
  // per-call work buffers (grown on demand)
  void* work = nullptr;
  size_t work_bytes = 0;
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  void* pinned_dev = nullptr;      // device alias of `pinned` (hipHostGetDevicePointer), valid while pinned_dev_of == pinned
  void* pinned_dev_of = nullptr;
  // the lane the error evaluations of this gas are issued on: nullptr = the context's own stream, pinned slots and counters;
  // ecckd_find_g_gases lends one per gas for the length of the call
  ecckd_lane* lane = nullptr;
  hipStream_t eval_stream() const { return lane ? lane->stream : ctx->stream; }
};

