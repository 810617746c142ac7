// find_g_band.hip - the per-band driver logic of find_g_points on top of the batched interval
// errors: sub-bands of the optically thin part of a band (find_g_points.cpp:786-870, :1186-1229),
// the plain search with its min/max g-point restarts (:1231-1258), the base split (:1265-1383),
// the rank range of every g point (:1396-1401) and the median sorting variable (:35-49).
// The index work that touches every wavenumber (re-ranking a rank range by wavenumber group,
// gathers, the cumulative-weight search) runs on the device; the search itself is the host
// PartitionSearch that drives ecckd_calc_error_batch.
#include "gas.hpp"
#include "partition_search.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <immintrin.h>

namespace {

constexpr int MAX_GROUPS = 32;
struct GroupBounds {
  int n;
  double b[MAX_GROUPS + 1];
};

// key[r - lo] = wavenumber group of the point whose rank r lies in [lo, hi]; group sizes by integer
// atomics (order-independent).  Group n = "in no group".
__global__ void __launch_bounds__(256)
k_regroup_key(size_t nwav, const double* __restrict__ wn, const int32_t* __restrict__ rank, long long lo,
              long long hi, GroupBounds gb, double* __restrict__ key, unsigned long long* __restrict__ count) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nwav) return;
  const long long r = rank[j];
  if (r < lo || r > hi) return;
  const double w = wn[j];
  int s = gb.n;
  for (int q = 0; q < gb.n; ++q)
    if (w >= gb.b[q] && w < gb.b[q + 1]) { s = q; break; }
  key[r - lo] = (double)s;
  atomicAdd(&count[s], 1ULL);
}

__global__ void __launch_bounds__(256)
k_regroup_apply(size_t nwav, int32_t* __restrict__ rank, long long lo, long long hi,
                const int32_t* __restrict__ newpos) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nwav) return;
  const long long r = rank[j];
  if (r < lo || r > hi) return;
  rank[j] = (int32_t)(lo + newpos[r - lo]);
}

__global__ void __launch_bounds__(256)
k_gather_f64(size_t n, const double* __restrict__ src, const int32_t* __restrict__ index, double* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[index[i]];
}

__global__ void __launch_bounds__(256)
k_invert_perm(size_t n, const int32_t* __restrict__ perm, int32_t* __restrict__ inv, int* __restrict__ err) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t r = perm[i];
  if (r < 0 || (size_t)r >= n) { atomicOr(err, 1); return; }
  inv[r] = (int32_t)i;
}

// calc_median_sorting_variable (find_g_points.cpp:35-49): the sorting variable at the first index of [i1, i2] where the
// running sum of the weights reaches half their total (the reference's loop ends at i2 - 1 and falls through to i2).
// Two kernels: 256-point tile sums of the weight row, once per call; then one block per g point adds the ragged head, the
// whole tiles and the ragged tail for the total, and finds the crossing top-down - head, tile (a block-wide scan over the
// tile sums, then a walk through the one segment that holds the crossing), point inside that tile, tail.  The reference
// adds the weights one by one; here sums are formed by tile and by scan, so where the running sum comes within rounding
// of half the total the crossing can fall on the neighbouring point (whose sorting variable differs from its
// neighbour's by the local spacing of the sorted keys).
constexpr int MED_TILE = 256;

__global__ void __launch_bounds__(256)
k_median_tile_sums(size_t n, const double* __restrict__ weight, double* __restrict__ ts) {
  __shared__ double s[256];
  const size_t i = (size_t)blockIdx.x * MED_TILE + threadIdx.x;
  s[threadIdx.x] = i < n ? weight[i] : 0.0;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) ts[blockIdx.x] = s[0];
}

// inclusive scan of one value per thread over the 256 threads of the block (Hillis-Steele, fixed order); the scanned
// values stay in s[] until the next call: s[t-1] is thread t's exclusive value, s[255] the total
__device__ __forceinline__ void block_scan_256(double v, double* s) {
  const int t = threadIdx.x;
  __syncthreads();
  s[t] = v;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const double add = t >= off ? s[t - off] : 0.0;
    __syncthreads();
    s[t] += add;
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256)
k_median_sorting(const long long* __restrict__ ind1, const long long* __restrict__ ind2, const double* __restrict__ weight,
                 const double* __restrict__ ts, const double* __restrict__ sv, double* __restrict__ out) {
  __shared__ double s_scan[256];
  __shared__ long long s_found;     // smallest index at which the running sum has reached half the total, or LLONG_MAX
  __shared__ long long s_tile;
  __shared__ double s_before;
  const long long NONE = 0x7fffffffffffffffLL;
  const int tid = threadIdx.x;
  const int k = blockIdx.x;
  const long long i1 = ind1[k], i2 = ind2[k];
  const long long last = i2 - 1;                         // the loop of :42-47 looks at i1 .. i2-1
  const long long t1 = (i1 + MED_TILE - 1) / MED_TILE;   // first whole tile
  const long long t2 = (i2 + 1) / MED_TILE;              // one past the last whole tile
  const bool tiles = t1 < t2;
  const long long head_end = tiles ? t1 * MED_TILE : i2 + 1;   // exclusive; without a whole tile everything is "head"
  const long long seg = tiles ? (t2 - t1 + 255) / 256 : 0;     // whole tiles per thread, contiguous
  const long long ta = t1 + (long long)tid * seg, tb = tiles ? (ta + seg < t2 ? ta + seg : t2) : ta;
  // ---- total weight of [i1, i2]: head points, whole tiles, tail points ----
  double part = 0.0;
  for (long long i = i1 + tid; i < head_end; i += 256) part += weight[i];
  block_scan_256(part, s_scan);
  double total = s_scan[255];
  double seg_sum = 0.0;
  for (long long t = ta; t < tb; ++t) seg_sum += ts[t];
  if (tiles) {
    block_scan_256(seg_sum, s_scan);
    total += s_scan[255];
    const long long tail0 = t2 * MED_TILE;
    block_scan_256((tail0 + tid <= i2) ? weight[tail0 + tid] : 0.0, s_scan);
    total += s_scan[255];
  }
  const double half = 0.5 * total;
  if (tid == 0) { s_found = NONE; s_tile = -1; }
  __syncthreads();
  // ---- the crossing, top-down.  `carry` = running sum in front of what is being looked at; a thread's exclusive value is
  //      its neighbour's inclusive one, so exactly one element is the first to reach half ----
  double carry = 0.0;
  for (long long base = i1; base < head_end; base += 256) {       // head: rounds of 256 points
    const long long i = base + tid;
    const bool in = i < head_end && i <= last;
    block_scan_256(in ? weight[i] : 0.0, s_scan);
    const double before = carry + (tid ? s_scan[tid - 1] : 0.0), inc = carry + s_scan[tid];
    // the very first point may reach half with nothing before it (an interval without weight: half == 0, the reference
    // stops at i1)
    if (in && inc >= half && (before < half || i == i1)) atomicMin(&s_found, i);
    carry += s_scan[255];
    __syncthreads();
    if (s_found != NONE) break;
  }
  if (tiles && s_found == NONE) {
    block_scan_256(seg_sum, s_scan);                               // which thread's run of tiles
    const double before = carry + (tid ? s_scan[tid - 1] : 0.0), inc = carry + s_scan[tid];
    const double all_tiles = s_scan[255];
    if (ta < tb && inc >= half && before < half) {
      double cum = before;                                         // which tile of the run: walk it
      for (long long t = ta; t < tb; ++t) {
        const double next = cum + ts[t];
        if (next >= half || t == tb - 1) { s_tile = t; s_before = cum; break; }
        cum = next;
      }
    }
    __syncthreads();
    if (s_tile >= 0) {                                             // which point of that tile
      const long long i = s_tile * MED_TILE + tid;
      const bool in = i <= last;
      block_scan_256(in ? weight[i] : 0.0, s_scan);
      const double b0 = s_before + (tid ? s_scan[tid - 1] : 0.0), inc2 = s_before + s_scan[tid];
      if (in && inc2 >= half && b0 < half) atomicMin(&s_found, i);
      __syncthreads();
      // (the tile's own sum said "reached", the scan of its points rounds differently: the tile's last point then)
      if (tid == 0 && s_found == NONE) s_found = (s_tile + 1) * MED_TILE - 1 <= last ? (s_tile + 1) * MED_TILE - 1 : NONE;
    } else {                                                       // tail
      const long long i = t2 * MED_TILE + tid;
      const bool in = i <= last;
      block_scan_256(in ? weight[i] : 0.0, s_scan);
      const double base3 = carry + all_tiles;
      const double b0 = base3 + (tid ? s_scan[tid - 1] : 0.0), inc3 = base3 + s_scan[tid];
      if (in && inc3 >= half && b0 < half) atomicMin(&s_found, i);
    }
  }
  __syncthreads();
  if (tid == 0) out[k] = sv[s_found != NONE ? s_found : i2];
}

// a temporary from the context's caching allocator (a hipFree would wait for every stream of the device, the other gases' searches
// included)
struct DevBuf {
  ecckd_ctx* ctx = nullptr;
  void* p = nullptr;
  explicit DevBuf(ecckd_ctx* c) : ctx(c) {}
  hipError_t alloc(size_t bytes) { return ecckd::dev_malloc(ctx, &p, bytes); }
  ~DevBuf() { if (p) { (void)hipStreamSynchronize(ctx->stream); ecckd::dev_release(ctx, p); } }
};

}  // namespace

extern "C" {

int ecckd_gather_f64_dev(ecckd_ctx* ctx, size_t n, const double* d_src, const int32_t* d_index, double* d_dst) {
  ECCKD_REQUIRE(ctx && (n == 0 || (d_src && d_index && d_dst)), "ecckd_gather_f64_dev: NULL argument");
  if (n == 0) return ECCKD_OK;
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_gather_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, d_src, d_index, d_dst);
  ECCKD_HIP_CHECK(hipGetLastError());
  return ECCKD_OK;
}

int ecckd_invert_permutation_dev(ecckd_ctx* ctx, size_t n, const int32_t* d_perm, int32_t* d_inverse) {
  ECCKD_REQUIRE(ctx && (n == 0 || (d_perm && d_inverse)), "ecckd_invert_permutation_dev: NULL argument");
  if (n == 0) return ECCKD_OK;
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  DevBuf flag(ctx);
  ECCKD_HIP_CHECK(flag.alloc(sizeof(int)));
  ECCKD_HIP_CHECK(hipMemsetAsync(flag.p, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(k_invert_perm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, d_perm, d_inverse,
                     (int*)flag.p);
  int h = 0;
  ECCKD_HIP_CHECK(hipMemcpyAsync(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  ECCKD_REQUIRE(h == 0, "ecckd_invert_permutation_dev: entries outside 0..n-1");
  return ECCKD_OK;
}

int ecckd_regroup_rank_by_wavenumber_dev(ecckd_ctx* ctx, size_t nwav, const double* d_wavenumber, int32_t* d_rank,
                                         size_t rank_lo, size_t rank_hi, int nsub, const double* h_wn_bound,
                                         int64_t* h_count) {
  ECCKD_REQUIRE(ctx && d_wavenumber && d_rank && h_wn_bound, "ecckd_regroup_rank_by_wavenumber_dev: NULL argument");
  ECCKD_REQUIRE(nsub >= 1 && nsub <= MAX_GROUPS, "ecckd_regroup_rank_by_wavenumber_dev: 1..%d groups supported, got %d",
                MAX_GROUPS, nsub);
  ECCKD_REQUIRE(rank_lo <= rank_hi && rank_hi < nwav, "ecckd_regroup_rank_by_wavenumber_dev: rank range [%zu,%zu] outside 0..%zu",
                rank_lo, rank_hi, nwav);
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  const size_t m = rank_hi - rank_lo + 1;
  GroupBounds gb;
  gb.n = nsub;
  for (int q = 0; q <= nsub; ++q) gb.b[q] = h_wn_bound[q];
  DevBuf key(ctx), newpos(ctx), count(ctx);
  ECCKD_HIP_CHECK(key.alloc(m * sizeof(double)));
  ECCKD_HIP_CHECK(newpos.alloc(m * sizeof(int32_t)));
  ECCKD_HIP_CHECK(count.alloc((MAX_GROUPS + 1) * sizeof(unsigned long long)));
  ECCKD_HIP_CHECK(hipMemsetAsync(count.p, 0, (MAX_GROUPS + 1) * sizeof(unsigned long long), ctx->stream));
  // a rank that is not a permutation leaves holes: fill with the "no group" key so they are counted
  const unsigned eblocks = (unsigned)((nwav + 255) / 256);
  hipLaunchKernelGGL(k_regroup_key, dim3(eblocks), dim3(256), 0, ctx->stream, nwav, d_wavenumber, d_rank,
                     (long long)rank_lo, (long long)rank_hi, gb, (double*)key.p, (unsigned long long*)count.p);
  ECCKD_HIP_CHECK(hipGetLastError());
  unsigned long long h_cnt[MAX_GROUPS + 1];
  ECCKD_HIP_CHECK(hipMemcpyAsync(h_cnt, count.p, sizeof(h_cnt), hipMemcpyDeviceToHost, ctx->stream));
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  unsigned long long assigned = 0;
  for (int q = 0; q < nsub; ++q) {
    assigned += h_cnt[q];
    if (h_count) h_count[q] = (int64_t)h_cnt[q];
  }
  if (assigned != m) return ecckd::fail(ECCKD_PARAMETER_ERROR, "Failed to account for all wavenumbers in split");
  const int64_t b0 = 0, b1 = (int64_t)m - 1;
  ECCKD_CHECK(ecckd_stable_argsort_bands_dev(ctx, m, (const double*)key.p, 1, &b0, &b1, (int32_t*)newpos.p, nullptr));
  hipLaunchKernelGGL(k_regroup_apply, dim3(eblocks), dim3(256), 0, ctx->stream, nwav, d_rank, (long long)rank_lo,
                     (long long)rank_hi, (const int32_t*)newpos.p);
  ECCKD_HIP_CHECK(hipGetLastError());
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return ECCKD_OK;
}

int ecckd_subband_setup_dev(ecckd_ctx* ctx, size_t nwav, const double* d_wavenumber, int32_t* d_rank, size_t ibegin,
                            size_t iend, double g_split, double band_bound1, double band_bound2, int nboundary,
                            const double* h_boundary, int* nsubband, int64_t* h_isubband1, int64_t* h_isubband2,
                            int64_t* iupperindex) {
  ECCKD_REQUIRE(ctx && nsubband && h_isubband1 && h_isubband2 && iupperindex && (nboundary == 0 || h_boundary),
                "ecckd_subband_setup_dev: NULL argument");
  ECCKD_REQUIRE(ibegin <= iend && iend < nwav, "ecckd_subband_setup_dev: band [%zu,%zu] outside the spectrum", ibegin, iend);
  *nsubband = 0;
  *iupperindex = -1;
  // :800-802 the band is split only if g_split > 0 and a boundary lies strictly inside it
  std::vector<double> inside;
  for (int q = 0; q < nboundary; ++q)
    if (h_boundary[q] > band_bound1 && h_boundary[q] < band_bound2) inside.push_back(h_boundary[q]);
  if (!(g_split > 0.0) || inside.empty()) return ECCKD_OK;
  const long long irank1 = (long long)ibegin, irank3 = (long long)iend;
  long long irank2 = irank3;
  *iupperindex = irank3;
  if (g_split < 1.0) irank2 = (long long)((double)irank1 + g_split * (double)(irank3 - irank1));  // :813-815 (int truncation)
  const int nsub = 1 + (int)inside.size();
  std::vector<double> wn_bound(nsub + 1);
  wn_bound[0] = band_bound1;
  wn_bound[nsub] = band_bound2 + 1.0;  // :824
  for (int q = 0; q < nsub - 1; ++q) wn_bound[q + 1] = inside[q];
  std::vector<int64_t> count(nsub);
  ECCKD_CHECK(ecckd_regroup_rank_by_wavenumber_dev(ctx, nwav, d_wavenumber, d_rank, (size_t)irank1, (size_t)irank2, nsub,
                                                   wn_bound.data(), count.data()));
  long long first = irank1;
  for (int q = 0; q < nsub; ++q) {
    h_isubband1[q] = first;
    h_isubband2[q] = first + count[q] - 1;
    first = h_isubband2[q] + 1;
  }
  *nsubband = nsub;
  return ECCKD_OK;
}

int ecckd_gas_median_sorting_variable(ecckd_gas* g, const double* d_sorting_variable_sorted, int n, const int64_t* h_ind1,
                                      const int64_t* h_ind2, double* h_median) {
  ECCKD_REQUIRE(g && d_sorting_variable_sorted && (n == 0 || (h_ind1 && h_ind2 && h_median)),
                "ecckd_gas_median_sorting_variable: NULL argument");
  if (n <= 0) return ECCKD_OK;
  ecckd_ctx* ctx = g->ctx;
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  for (int k = 0; k < n; ++k)
    ECCKD_REQUIRE(h_ind1[k] >= 0 && h_ind1[k] <= h_ind2[k] && (size_t)h_ind2[k] < g->n,
                  "ecckd_gas_median_sorting_variable: interval %d [%lld,%lld] outside the spectrum", k,
                  (long long)h_ind1[k], (long long)h_ind2[k]);
  // weight: surface Planck function (LW, :1405) or reordered solar irradiance (SW, :1408)
  const double* weight = g->do_sw ? g->ssi : g->planck_hl + (size_t)g->nlay * g->n;
  const size_t ntiles = (g->n + MED_TILE - 1) / MED_TILE;
  const size_t ts_bytes = ecckd_align_up(ntiles * sizeof(double), 256), idx_bytes = ecckd_align_up((size_t)n * 2 * sizeof(long long), 256);
  ECCKD_CHECK(ecckd::ensure_scratch(ctx, ts_bytes + idx_bytes + (size_t)n * sizeof(double)));
  double* d_ts = (double*)ctx->scratch;
  long long* d_i1 = (long long*)((char*)ctx->scratch + ts_bytes);
  long long* d_i2 = d_i1 + n;
  double* d_out = (double*)((char*)ctx->scratch + ts_bytes + idx_bytes);
  std::vector<long long> tmp(2 * (size_t)n);
  for (int k = 0; k < n; ++k) { tmp[k] = h_ind1[k]; tmp[n + k] = h_ind2[k]; }
  ECCKD_HIP_CHECK(hipMemcpyAsync(d_i1, tmp.data(), tmp.size() * sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_median_tile_sums, dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, g->n, weight, d_ts);
  hipLaunchKernelGGL(k_median_sorting, dim3(n), dim3(256), 0, ctx->stream, d_i1, d_i2, weight, d_ts, d_sorting_variable_sorted, d_out);
  ECCKD_HIP_CHECK(hipGetLastError());
  ECCKD_HIP_CHECK(hipMemcpyAsync(h_median, d_out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return ECCKD_OK;
}

}  // extern "C"

// ecckd_find_g_gases_begin / _add / _wait: the searches in flight
struct ecckd_gas_search_job {
  double tolerance_tolerance = 0.02;
  int max_iterations = 60;
  int width = 0;                       // gases at a time; <= 0: what the host's cores allow
  std::vector<ecckd_gas_search*> req;
  std::deque<std::string> message;     // (a deque: the searches in flight keep pointers to their own slot while others are added)
  std::vector<std::thread> threads;
  std::mutex mutex;
  std::condition_variable cv;
  int running = 0, threads_running = 0;
  // timing window (ecckd_profile_enable)
  ecckd_ctx* ctx0 = nullptr;
  bool timed = false;
  hipEvent_t w0 = nullptr, w1 = nullptr;
  double points_before = 0.0;
};

namespace {

// Host threads that wait on each other by spinning (band searches on their batcher, a batcher on its searches).  While there
// are no more of them than cores they spin with a pause; with several multi-band gases searched side by side
// (ecckd_find_g_gases) there are more, and every turn of a wait loop gives the core away so that the thread waited for runs.
std::atomic<int> g_spinning_threads{0};
inline bool oversubscribed() {
  static const int cores = ecckd::host_cores();
  return g_spinning_threads.load(std::memory_order_relaxed) > cores;
}
struct SpinningThread {
  SpinningThread() { g_spinning_threads.fetch_add(1, std::memory_order_relaxed); }
  ~SpinningThread() { g_spinning_threads.fetch_sub(1, std::memory_order_relaxed); }
};

// Merges the error evaluations that several band searches (one host thread each) ask for at the same time into one
// ecckd_calc_error_multi call.  A search posts its batch in its own slot and spins on it; the thread that started the
// searches (ecckd_find_g_bands_ex) watches the slots and runs the merged batch when EVERY search that is still going has
// posted one.  Only that thread talks to the device: a thread's first HIP call costs milliseconds of per-thread set-up,
// which with one fresh thread per band and gas used to be most of a narrow-band job's wall time.
class BandBatcher {
 public:
  BandBatcher(ecckd_gas* gas, int nsearch) : gas_(gas), slots_(nsearch), active_(nsearch) {}

  // called by search `id`: post the request, wait for its errors
  int evaluate(int id, size_t ibegin, size_t npoints, double albedo, int n, const double* b1, const double* b2, double* e) {
    Slot& s = slots_[id];
    s.ibegin = ibegin; s.npoints = npoints; s.albedo = albedo; s.n = n; s.b1 = b1; s.b2 = b2; s.e = e;
    s.state.store(PENDING, std::memory_order_release);
    // (spin, then give the core away now and then: with several gases searched side by side there can be more searches than
    // cores, and the thread that serves the batch must get to run)
    for (unsigned spins = 1; s.state.load(std::memory_order_acquire) != DONE; ++spins) {
      if ((spins & 0xfff) == 0 || ((spins & 0x1f) == 0 && oversubscribed())) std::this_thread::yield(); else _mm_pause();
    }
    s.state.store(IDLE, std::memory_order_relaxed);
    if (s.rc != ECCKD_OK) ecckd::fail(s.rc, "%s", s.message.c_str());   // the message was recorded on the thread that ran the batch
    return s.rc;
  }

  // search `id` is over (converged, failed, or gone on to its post-processing): the others no longer wait for it
  void leave(int id) {
    slots_[id].state.store(LEFT, std::memory_order_release);
    active_.fetch_sub(1, std::memory_order_acq_rel);
  }

  // on the starting thread: serve batches until every search has left
  void serve() {
    std::vector<size_t> ib, np;
    std::vector<double> b1, b2, alb, err;
    std::vector<int> who;
    unsigned idle = 0;
    while (active_.load(std::memory_order_acquire) > 0) {
      // a round is complete when no search is between two requests
      bool complete = true;
      who.clear();
      for (size_t i = 0; i < slots_.size(); ++i) {
        const int st = slots_[i].state.load(std::memory_order_acquire);
        if (st == PENDING) who.push_back((int)i);
        else if (st != LEFT) { complete = false; break; }
      }
      if (!complete || who.empty()) {
        ++idle;
        if ((idle & 0xfff) == 0 || ((idle & 0x1f) == 0 && oversubscribed())) std::this_thread::yield(); else _mm_pause();
        continue;
      }
      idle = 0;
      // laid out by band (slot order = band order): the batch is the same from run to run
      ib.clear(); np.clear(); b1.clear(); b2.clear(); alb.clear();
      for (int i : who) {
        const Slot& s = slots_[i];
        for (int k = 0; k < s.n; ++k) {
          ib.push_back(s.ibegin); np.push_back(s.npoints); alb.push_back(s.albedo); b1.push_back(s.b1[k]); b2.push_back(s.b2[k]);
        }
      }
      err.resize(b1.size());
      const int rc = ecckd_calc_error_multi(gas_, (int)b1.size(), ib.data(), np.data(), alb.data(), b1.data(), b2.data(), err.data());
      const std::string message = rc == ECCKD_OK ? std::string() : std::string(ecckd_last_error());
      size_t off = 0;
      for (int i : who) {
        Slot& s = slots_[i];
        if (rc == ECCKD_OK) std::copy(err.begin() + off, err.begin() + off + s.n, s.e);
        off += s.n;
        s.rc = rc;
        s.message = message;
        s.state.store(DONE, std::memory_order_release);
      }
    }
  }

 private:
  enum { IDLE = 0, PENDING = 1, DONE = 2, LEFT = 3 };
  struct Slot {
    std::atomic<int> state{IDLE};
    size_t ibegin = 0, npoints = 0;
    double albedo = 0.0;      // shortwave: the band's surface albedo
    int n = 0;
    const double *b1 = nullptr, *b2 = nullptr;
    double* e = nullptr;
    int rc = ECCKD_OK;
    std::string message;
  };

  ecckd_gas* gas_;
  std::vector<Slot> slots_;
  std::atomic<int> active_;
};

struct BatcherRef { BandBatcher* batcher; int id; };
thread_local BatcherRef tl_batcher = {nullptr, -1};
std::mutex g_band_device_mutex;

}  // namespace

namespace ecckd {
bool host_oversubscribed() { return oversubscribed(); }
}

extern "C" {

int ecckd_find_g_band_ex(ecckd_gas* g, size_t ibegin, size_t iend, double heating_rate_tolerance, double tolerance_tolerance,
                         int max_iterations, const ecckd_band_options* opt, int* ng, double* bounds, double* error,
                         int64_t* rank1, int64_t* rank2, int capacity, int* status, double* comp_cost) {
  ECCKD_REQUIRE(g && opt && ng && bounds && error && status, "ecckd_find_g_band_ex: NULL argument");
  ECCKD_REQUIRE(iend >= ibegin && iend < g->n, "ecckd_find_g_band_ex: band [%zu,%zu] outside the spectrum", ibegin, iend);
  const size_t npoints = iend - ibegin + 1;
  const double cost0 = g->total_comp_cost;
  int rc_eval = ECCKD_OK;
  BandBatcher* const batcher = tl_batcher.batcher;    // set when this search is one of several running side by side
  const int batcher_id = tl_batcher.id;
  double local_cost = 0.0;
  ecckd::PartitionSearch ps([&](int n, const double* b1, const double* b2, double* e) {
    if (batcher) {
      for (int k = 0; k < n; ++k) local_cost += b2[k] - b1[k];
      rc_eval = batcher->evaluate(batcher_id, ibegin, npoints, opt->band_albedo, n, b1, b2, e);
    } else {
      rc_eval = ecckd_calc_error_batch(g, ibegin, npoints, n, b1, b2, e);
    }
    return rc_eval;
  });
  // CkdEquipartition::init_lw / init_sw (find_g_points.cpp:230-233, :256-261) + :1180-1181
  ps.set_resolution(1.0 / (double)npoints);
  ps.set_minimize_frac_range(true);
  ps.set_partition_max_iterations(max_iterations);
  ps.set_partition_tolerance(tolerance_tolerance);
  // CkdEquipartition::lower_index / upper_index (:282-287)
  auto lower_index = [&](double b) { return (long long)std::ceil(b * (double)(npoints - 1)); };
  auto upper_index = [&](double b) { return (long long)std::floor(b * (double)(npoints - 1)); };

  std::vector<double> b, e;
  int n = 10;
  int st = 0;
  if (opt->nsubband > 1) {
    // ---- :1186-1229 one search per sub-band of the optically thin part, then the overarching rest ----
    ECCKD_REQUIRE(opt->isubband1 && opt->isubband2, "ecckd_find_g_band_ex: sub-band ranks missing");
    const double denom = (double)(opt->iupperindex - opt->isubband1[0]);
    n = 0;
    for (int jsub = 0; jsub < opt->nsubband; ++jsub) {
      std::vector<double> sb, se;
      int nsubg = 4;
      const double g_start = (double)(opt->isubband1[jsub] - opt->isubband1[0]) / denom;
      const double g_end = (double)(opt->isubband2[jsub] - opt->isubband1[0]) / denom;
      st = ps.equipartition_e(heating_rate_tolerance, g_start, g_end, nsubg, sb, se);
      if (ps.evaluator_status()) return rc_eval ? rc_eval : ECCKD_PROCESSING_ERROR;
      b.insert(b.begin() + n, sb.begin(), sb.end());
      e.insert(e.end(), se.begin(), se.end());
      n += nsubg;
    }
    if (opt->g_split < 1.0) {
      std::vector<double> sb, se;
      int nsubg = 4;
      const double g_start = (double)(opt->isubband2[opt->nsubband - 1] - opt->isubband1[0]) / denom;
      st = ps.equipartition_e(heating_rate_tolerance, g_start, 1.0, nsubg, sb, se);
      if (ps.evaluator_status()) return rc_eval ? rc_eval : ECCKD_PROCESSING_ERROR;
      if (n + nsubg < opt->min_g_points) {
        nsubg = opt->min_g_points - n;
        sb.resize(nsubg + 1);
        se.resize(nsubg);
        for (int i = 0; i <= nsubg; ++i)
          sb[i] = opt->g_split + (1.0 - opt->g_split) * std::sqrt((double)i / (double)nsubg);
        st = ps.equipartition_n(nsubg, sb.data(), se.data());
        if (ps.evaluator_status()) return rc_eval ? rc_eval : ECCKD_PROCESSING_ERROR;
      }
      b.insert(b.begin() + n, sb.begin(), sb.end());
      e.insert(e.end(), se.begin(), se.end());
      n += nsubg;
    }
    b.resize(n + 1);
  } else {
    st = ps.equipartition_e(heating_rate_tolerance, 0.0, 1.0, n, b, e);
    const int min_g = opt->min_g_points, max_g = opt->max_g_points > 0 ? opt->max_g_points : 0x7fffffff;
    if (!ps.evaluator_status() && (n < min_g || n > max_g)) {
      // :1232-1257 restart from bounds sqrt(i/ng)
      n = (n < min_g) ? min_g : max_g;
      b.resize(n + 1);
      e.resize(n);
      for (int i = 0; i <= n; ++i) b[i] = std::sqrt((double)i / (double)n);
      st = ps.equipartition_n(n, b.data(), e.data());
    }
    if (ps.evaluator_status()) return rc_eval ? rc_eval : ECCKD_PROCESSING_ERROR;
  }

  // ---- :1265-1383 dissect the base g point by wavenumber and / or absorption ----
  const int nwavsplit = opt->nbase_wn_bound >= 2 ? opt->nbase_wn_bound - 1 : 1;
  const double base_split = opt->base_split == 0.0 ? 1.0 : opt->base_split;  // zero-initialised struct = no split
  if (base_split != 1.0 || nwavsplit > 1) {
    int nabssplit = 1;
    if (base_split > 1.0) {
      nabssplit = (int)base_split;
      if (nabssplit == 1) return ecckd::fail(ECCKD_PARAMETER_ERROR, "Positive values of base_split must be at least 2");
    } else {
      nabssplit = 2 + (int)(base_split * n);  // always split into at least two (:1285)
    }
    const int nsplit = nwavsplit * nabssplit;
    std::vector<long long> iwav1(nwavsplit), iwav2(nwavsplit);
    iwav1[0] = (long long)ibegin;
    iwav2[nwavsplit - 1] = (long long)iend;
    if (nwavsplit > 1) {
      ECCKD_REQUIRE(opt->base_wn_bound && opt->d_wavenumber && opt->d_rank && opt->nwav > 0,
                    "ecckd_find_g_band_ex: wavenumber split of the base g point needs wavenumber and rank");
      const long long ind2 = upper_index(b[1]) + (long long)ibegin;
      // :1317 "iwav1(0) = 0": the new ranks are counted from 0, so the split is only consistent for a
      // band that starts at rank 0; any other band ends in the reference's :1335-1338 error
      if (ibegin != 0) return ecckd::fail(ECCKD_PARAMETER_ERROR, "Failed to account for all wavenumbers in split");
      std::vector<int64_t> count(nwavsplit);
      {
        // the regrouping runs on the context's stream with the context's sort scratch: one search at a time.  (The other
        // searches of this gas never touch the device themselves; two of them, or two
        // searches could get here together (other bands of this gas, other gases searched side by side)
        std::lock_guard<std::mutex> device_lock(g_band_device_mutex);
        ECCKD_CHECK(ecckd_regroup_rank_by_wavenumber_dev(g->ctx, opt->nwav, opt->d_wavenumber, opt->d_rank, 0, (size_t)ind2,
                                                         nwavsplit, opt->base_wn_bound, count.data()));
      }
      iwav1[0] = 0;
      for (int q = 0; q < nwavsplit; ++q) {
        if (q > 0) iwav1[q] = iwav2[q - 1] + 1;
        iwav2[q] = iwav1[q] + count[q] - 1;
      }
    }
    const double upper_bound = b[1];
    double lower_bound_local = b[0];
    e[0] = -1.0;  // first error is now incorrect (:1352)
    int ibnd = 0;
    for (int iw = 0; iw < nwavsplit; ++iw) {
      const double upper_bound_local = upper_bound * (double)iwav2[iw] / (double)iwav2[nwavsplit - 1];
      for (int ia = 0; ia < nabssplit; ++ia) {
        if (ia < nabssplit - 1 || iw < nwavsplit - 1) {
          b.insert(b.begin() + ibnd + 1,
                   lower_bound_local + (upper_bound_local - lower_bound_local) * (double)(ia + 1) / (double)nabssplit);
          e.insert(e.begin() + ibnd, -1.0);
          ++ibnd;
        }
      }
      lower_bound_local = upper_bound_local;
    }
    n += nsplit - 1;
  }

  for (int i = 0; i < n; ++i)
    if (!(b[i + 1] - b[i] > 0.0)) return ecckd::fail(ECCKD_PARAMETER_ERROR, "Bounds are not monotonically increasing");  // :1390-1393

  *status = st;
  *ng = n;
  if (comp_cost) *comp_cost = batcher ? local_cost : g->total_comp_cost - cost0;
  ECCKD_REQUIRE(n <= capacity, "ecckd_find_g_band_ex: %d g points exceed the caller's capacity %d", n, capacity);
  std::memcpy(bounds, b.data(), (size_t)(n + 1) * sizeof(double));
  std::memcpy(error, e.data(), (size_t)n * sizeof(double));
  for (int i = 0; i < n; ++i) {
    if (rank1) rank1[i] = lower_index(b[i]) + (long long)ibegin;       // :1397
    if (rank2) rank2[i] = upper_index(b[i + 1]) + (long long)ibegin;   // :1398
  }
  return ECCKD_OK;
}

int ecckd_find_g_band(ecckd_gas* g, size_t ibegin, size_t iend, double heating_rate_tolerance, double tolerance_tolerance,
                      int max_iterations, int min_g_points, int max_g_points, int* ng, double* bounds, double* error,
                      int capacity, int* status, double* comp_cost) {
  ecckd_band_options opt;
  std::memset(&opt, 0, sizeof(opt));
  opt.min_g_points = min_g_points;
  opt.max_g_points = max_g_points;
  opt.base_split = 1.0;
  return ecckd_find_g_band_ex(g, ibegin, iend, heating_rate_tolerance, tolerance_tolerance, max_iterations, &opt, ng,
                              bounds, error, nullptr, nullptr, capacity, status, comp_cost);
}

int ecckd_find_g_bands_ex(ecckd_gas* g, int nband, const size_t* ibegin, const size_t* iend, const double* heating_rate_tolerance,
                          double tolerance_tolerance, int max_iterations, const ecckd_band_options* opt, int* ng, double* bounds,
                          double* error, int64_t* rank1, int64_t* rank2, int capacity, int* status, double* comp_cost) {
  ECCKD_REQUIRE(g && nband > 0 && ibegin && iend && heating_rate_tolerance && opt && ng && bounds && error && status && capacity > 0,
                "ecckd_find_g_bands_ex: bad argument");
  if (nband == 1 && !g->do_sw)
    return ecckd_find_g_band_ex(g, ibegin[0], iend[0], heating_rate_tolerance[0], tolerance_tolerance, max_iterations, &opt[0], ng,
                                bounds, error, rank1, rank2, capacity, status, comp_cost);
  BandBatcher batcher(g, nband);
  std::vector<int> rc(nband, ECCKD_OK);
  std::vector<std::string> message(nband);
  std::vector<std::thread> threads;
  threads.reserve(nband);
  for (int b = 0; b < nband; ++b) {
    threads.emplace_back([&, b] {
      SpinningThread counted;
      tl_batcher = {&batcher, b};
      struct Leave {   // whatever way the search ends, the others must stop waiting for it
        BandBatcher& bb;
        int id;
        ~Leave() { bb.leave(id); }
      } leave{batcher, b};
      rc[b] = ecckd_find_g_band_ex(g, ibegin[b], iend[b], heating_rate_tolerance[b], tolerance_tolerance, max_iterations, &opt[b],
                                   &ng[b], bounds + (size_t)b * (capacity + 1), error + (size_t)b * capacity,
                                   rank1 ? rank1 + (size_t)b * capacity : nullptr, rank2 ? rank2 + (size_t)b * capacity : nullptr,
                                   capacity, &status[b], comp_cost ? &comp_cost[b] : nullptr);
      if (rc[b] != ECCKD_OK) message[b] = ecckd_last_error();
      tl_batcher = {nullptr, -1};
    });
  }
  {
    SpinningThread counted;
    batcher.serve();
  }
  for (std::thread& t : threads) t.join();
  for (int b = 0; b < nband; ++b)
    if (rc[b] != ECCKD_OK) return ecckd::fail(rc[b], "band %d: %s", b, message[b].c_str());
  return ECCKD_OK;
}

// find_g_points.cpp:655-1266, the gas loop: the band searches of SEVERAL gases side by side on one device.  The reference
// searches gas after gas; the searches are independent of each other (each gas has its own prepared rows; the shared Planck
// matrix is only read), and a single search cannot fill the chip: 86 % of its error batches are one or two intervals that
// are bound by launch and memory latency (~35 us each, a few per cent of the CUs), and every batch depends on the one before.
// So every gas gets a host thread and a lane of its own - HIP stream, pinned result slots, timing events - and runs EXACTLY
// the launch trains it runs alone: an interval's error does not depend on what else is on the device, so every search takes
// the decisions it takes alone (same g points, same errors to the last bit), and one gas's latency-bound batches run in the
// shadow of another gas's whole-partition passes.  (Merging the gases' requests into ONE launch train per round, as the
// bands of one gas are merged by BandBatcher, would make every single-interval request last as long as the longest pass of
// the round - ~1 ms instead of ~35 us on each of the ~2 000 dependent steps of a search.)
int ecckd_find_g_gases_begin(double tolerance_tolerance, int max_iterations, int max_concurrent, ecckd_gas_search_job** out) {
  ECCKD_REQUIRE(out, "ecckd_find_g_gases_begin: NULL argument");
  ecckd_gas_search_job* job = new ecckd_gas_search_job();
  job->tolerance_tolerance = tolerance_tolerance;
  job->max_iterations = max_iterations;
  job->width = max_concurrent;
  if (const char* e = std::getenv("ECCKD_GASES_SIDE_BY_SIDE")) job->width = std::max(1, std::atoi(e));
  *out = job;
  return ECCKD_OK;
}

int ecckd_find_g_gases_add(ecckd_gas_search_job* job, ecckd_gas_search* r) {
  ECCKD_REQUIRE(job && r, "ecckd_find_g_gases_add: NULL argument");
  const int k = (int)job->req.size();
  ECCKD_REQUIRE(r->gas && r->nband > 0 && r->ibegin && r->iend && r->heating_rate_tolerance && r->opt && r->ng && r->bounds && r->error &&
                r->status && r->capacity > 0, "ecckd_find_g_gases: request %d is incomplete", k);
  for (int j = 0; j < k; ++j) ECCKD_REQUIRE(job->req[j]->gas != r->gas, "ecckd_find_g_gases: requests %d and %d name the same gas", j, k);
  ECCKD_REQUIRE(r->gas->lane == nullptr, "ecckd_find_g_gases: gas %d is already being searched", k);
  r->rc = ECCKD_OK;
  ecckd_gas* const g = r->gas;
  ecckd_ctx* const ctx = g->ctx;
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  // what the preparation left on the context's stream must have landed before the lane's stream reads it
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (job->req.empty()) {
    // with the timing switched on (ecckd_profile_enable): the window in which the searches run, between two events on the first
    // gas's context stream (every lane is synchronised before the second one is recorded), and the points the gases' sweeps
    // process in it - side by side the sweeps of several streams overlap, so bytes over THIS time, not bytes per launch over a
    // launch's duration, say how busy the memory system was
    job->ctx0 = ctx;
    job->timed = ctx->profile && ctx->pev0;
    if (job->timed) {
      ECCKD_HIP_CHECK(hipEventCreate(&job->w0));
      ECCKD_HIP_CHECK(hipEventCreate(&job->w1));
      ECCKD_HIP_CHECK(hipEventRecord(job->w0, ctx->stream));
    }
  }
  job->points_before += g->points_evaluated;
  job->req.push_back(r);
  job->message.emplace_back();
  std::string* const msg = &job->message.back();
  // a gas with several bands runs a thread per band and one that serves their batches; they wait for each other by spinning
  // and yield the core when there are more of them than cores (oversubscribed()), so the limit is generous: sixteen threads per
  // core (width <= 0: what that allows).  Measured on a 16-core share: the 32-band x 3-gas shortwave job 31.2 ms with one gas at a
  // time (four threads per core), 25.4 ms with all three (99 threads); the 13-band x 8-gas longwave job 165 ms either way.
  const int cores = 16 * ecckd::host_cores();
  const int per_gas = (r->nband > 1 || g->do_sw) ? r->nband + 1 : 1;
  // a search that re-ranks its base g point by wavenumber (ecckd_regroup_rank_by_wavenumber_dev) works on the context's stream
  // with the context's scratch, which the caller may be using for the next gas's preparation: such a gas is searched here and now
  bool needs_context = false;
  for (int b = 0; b < r->nband; ++b) needs_context |= r->opt[b].nbase_wn_bound > 2;
  auto search = [job, r, msg] {
    r->rc = ecckd_find_g_bands_ex(r->gas, r->nband, r->ibegin, r->iend, r->heating_rate_tolerance, job->tolerance_tolerance,
                                  job->max_iterations, r->opt, r->ng, r->bounds, r->error, r->rank1, r->rank2, r->capacity, r->status,
                                  r->comp_cost);
    if (r->rc != ECCKD_OK) *msg = ecckd_last_error();
  };
  if (job->width == 1 || needs_context) {
    search();
    return ECCKD_OK;      // (a failure is reported by _wait, like the others')
  }
  job->threads.emplace_back([job, r, msg, g, ctx, per_gas, cores, search] {
    {
      // at most `width` gases at a time, and not more threads than cores
      std::unique_lock<std::mutex> lock(job->mutex);
      job->cv.wait(lock, [&] {
        return job->running == 0 || ((job->width <= 0 || job->running < job->width) && job->threads_running + per_gas <= cores);
      });
      job->running += 1;
      job->threads_running += per_gas;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) { r->rc = ECCKD_UNEXPECTED_EXCEPTION; *msg = "hipSetDevice failed"; }
    else if (!(g->lane = ecckd::lane_acquire(ctx))) { r->rc = ECCKD_UNEXPECTED_EXCEPTION; *msg = ecckd_last_error(); }
    else {
      g->pinned_dev_of = nullptr;       // the device alias of the pinned slots belongs to the buffer of the lane
      search();
      (void)hipStreamSynchronize(g->lane->stream);
      ecckd::lane_release(ctx, g->lane);
      g->lane = nullptr;
      g->pinned_dev_of = nullptr;
      g->pinned = nullptr;
      g->pinned_bytes = 0;
    }
    {
      std::lock_guard<std::mutex> lock(job->mutex);
      job->running -= 1;
      job->threads_running -= per_gas;
    }
    job->cv.notify_all();
  });
  return ECCKD_OK;
}

int ecckd_find_g_gases_wait(ecckd_gas_search_job* job) {
  ECCKD_REQUIRE(job, "ecckd_find_g_gases_wait: NULL argument");
  for (std::thread& t : job->threads) t.join();
  if (job->timed) {
    float ms = 0.f;
    const bool ok = hipEventRecord(job->w1, job->ctx0->stream) == hipSuccess && hipEventSynchronize(job->w1) == hipSuccess &&
                    hipEventElapsedTime(&ms, job->w0, job->w1) == hipSuccess;
    (void)hipEventDestroy(job->w0);
    (void)hipEventDestroy(job->w1);
    if (ok) {
      double points_after = 0.0;
      for (ecckd_gas_search* r : job->req) points_after += r->gas->points_evaluated;
      job->ctx0->stat_gases.ms += ms;
      job->ctx0->stat_gases.units += points_after - job->points_before;
      job->ctx0->stat_gases.calls += 1;
    }
  }
  int rc = ECCKD_OK;
  for (size_t k = 0; k < job->req.size() && rc == ECCKD_OK; ++k)
    if (job->req[k]->rc != ECCKD_OK) rc = ecckd::fail(job->req[k]->rc, "gas %zu: %s", k, job->message[k].c_str());
  delete job;
  return rc;
}

int ecckd_find_g_gases(int ngas, ecckd_gas_search* req, double tolerance_tolerance, int max_iterations, int max_concurrent) {
  ECCKD_REQUIRE(ngas > 0 && req, "ecckd_find_g_gases: bad argument");
  ecckd_gas_search_job* job = nullptr;
  ECCKD_CHECK(ecckd_find_g_gases_begin(tolerance_tolerance, max_iterations, ngas == 1 ? 1 : max_concurrent, &job));
  int rc = ECCKD_OK;
  std::string message;
  for (int k = 0; k < ngas && rc == ECCKD_OK; ++k) {
    rc = ecckd_find_g_gases_add(job, &req[k]);
    if (rc != ECCKD_OK) message = ecckd_last_error();
  }
  const int rc_wait = ecckd_find_g_gases_wait(job);      // the searches already started are waited for whatever happened
  if (rc != ECCKD_OK) return ecckd::fail(rc, "%s", message.c_str());
  return rc_wait;
}

}  // extern "C"
