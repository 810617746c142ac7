// radix_sort.hip - K3: stable per-band argsort of the f64 sorting key on gfx950.
//
// Replaces reference src/ecckd/reorder_spectrum.cpp:262-300 (std::stable_sort of
// an index vector with `key[i1] < key[i2]`, single thread, then the rank
// scatter).  A least-significant-digit radix sort is stable by construction, so
// the permutation is the unique one std::stable_sort produces for keys without
// NaNs; -0.0 is canonicalised to +0.0 (they compare equal under `<`), NaN keys
// sort after everything else (undefined behaviour in the reference).
//
// Layout: 64-bit order-preserving key image + 32-bit index payload, ping-pong
// buffers in the context scratch.  8 passes of 8 bits; each pass is
//   (1) per-tile digit histogram   (LDS integer atomics)
//   (2) exclusive scan of the digit-major (digit, tile) table (one block per digit row)
//   (3) stable scatter: each 64-wide wave ranks its elements with ballot
//       matching (wave64: one 64-bit peer mask per element), waves of a block
//       are ordered through an LDS count table.
// Algorithmic traffic per pass: 8 B (histogram) + 12 B read + 12 B written.
#include "common.hpp"

#include <cstdlib>
#include <cstring>

namespace {

constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int SORT_THREADS = 256;
constexpr int SORT_WAVES = SORT_THREADS / 64;
constexpr int SORT_ITEMS = 16;
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;  // 4096 elements per block

__device__ __forceinline__ unsigned long long key_to_sortable(double k) {
  if (k != k) return ~0ull;          // NaN last
  if (k == 0.0) k = 0.0;             // -0.0 -> +0.0
  unsigned long long b = (unsigned long long)__double_as_longlong(k);
  return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

// Several bands in ONE sort: the spectrum is cut into segments (bands, and the gaps between them) that are contiguous in
// the input; after the eight passes over the key a ninth pass sorts by segment number, which - the sort being stable -
// puts every band's points, ordered by key, back into the band's own index range.  Points of a gap carry the key 0 and
// therefore keep their places.  The segment of an element is looked up from its original index (the payload).
constexpr int MAX_SEG = 120;
struct SegTable {
  int n;                         // number of segments
  unsigned begin[MAX_SEG];       // first index of segment q (begin[0] = 0)
  unsigned char is_band[MAX_SEG];
};

__device__ __forceinline__ unsigned seg_of(const SegTable& st, unsigned j) {
  unsigned q = 0;
  for (int t = 1; t < st.n; ++t) q += (st.begin[t] <= j) ? 1u : 0u;
  return q;
}

// digit of an element in the pass `shift`: eight bits of the key, or (shift == 64) its segment
__device__ __forceinline__ unsigned digit_of(unsigned long long key, unsigned idx, int shift, const SegTable& st) {
  return shift < 64 ? (unsigned)(key >> shift) & (RADIX - 1) : seg_of(st, idx);
}

__global__ void __launch_bounds__(256)
k_sort_prepare_segments(size_t n, const double* __restrict__ key, SegTable st, unsigned long long* __restrict__ skey,
                        unsigned* __restrict__ sidx) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  skey[i] = st.is_band[seg_of(st, (unsigned)i)] ? key_to_sortable(key[i]) : 0ull;
  sidx[i] = (unsigned)i;
}

__global__ void __launch_bounds__(256)
k_sort_prepare(size_t off, size_t n, const double* __restrict__ key,
               unsigned long long* __restrict__ skey, unsigned* __restrict__ sidx) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  skey[i] = key_to_sortable(key[off + i]);
  sidx[i] = (unsigned)(off + i);
}

// (1) tile histogram -> tile_hist[digit * ntiles + tile]
__global__ void __launch_bounds__(SORT_THREADS)
k_sort_hist(size_t n, int shift, const unsigned long long* __restrict__ skey, const unsigned* __restrict__ sidx, SegTable st,
            unsigned* __restrict__ tile_hist, unsigned ntiles) {
  __shared__ unsigned s_hist[RADIX];
  const int tid = threadIdx.x;
  s_hist[tid] = 0;  // SORT_THREADS == RADIX
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * SORT_TILE;
#pragma unroll
  for (int it = 0; it < SORT_ITEMS; ++it) {
    size_t i = base + (size_t)it * SORT_THREADS + tid;
    if (i < n) {
      unsigned d = digit_of(skey[i], shift < 64 ? 0u : sidx[i], shift, st);
      atomicAdd(&s_hist[d], 1u);
    }
  }
  __syncthreads();
  tile_hist[(size_t)tid * ntiles + blockIdx.x] = s_hist[tid];
}

// (2) one block per digit: exclusive scan of that digit's row of per-tile counts
// (in place, coalesced) and the row total -> digit_total[digit].  The scan over
// the 256 digit totals is done by every scatter block in LDS.
__global__ void __launch_bounds__(256)
k_sort_scan_rows(unsigned* __restrict__ tile_hist, unsigned ntiles, unsigned* __restrict__ digit_total) {
  __shared__ unsigned s_wave[4];
  __shared__ unsigned s_carry;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  unsigned* row = tile_hist + (size_t)blockIdx.x * ntiles;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (unsigned base = 0; base < ntiles; base += 256) {
    const unsigned i = base + tid;
    const unsigned v = (i < ntiles) ? row[i] : 0u;
    // inclusive scan inside the wave
    unsigned x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      unsigned y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    unsigned woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_wave[w];
    const unsigned carry = s_carry;
    if (i < ntiles) row[i] = carry + woff + x - v;
    __syncthreads();
    if (tid == 255) s_carry = carry + woff + x;
    __syncthreads();
  }
  if (tid == 0) digit_total[blockIdx.x] = s_carry;
}

// (3) stable scatter.  Wave w of a block owns the contiguous sub-tile
// [w*1024, (w+1)*1024) of the block's tile as 16 chunks of 64 consecutive
// elements, so (chunk, lane) order is input order.
__global__ void __launch_bounds__(SORT_THREADS)
k_sort_scatter(size_t n, int shift, const unsigned long long* __restrict__ skey_in,
               const unsigned* __restrict__ sidx_in, unsigned long long* __restrict__ skey_out,
               unsigned* __restrict__ sidx_out, const unsigned* __restrict__ tile_offs,
               unsigned ntiles, const unsigned* __restrict__ digit_total, SegTable st) {
  __shared__ unsigned s_cnt[SORT_WAVES][RADIX];  // per-wave digit counts, then running offsets
  __shared__ unsigned s_dig[RADIX];              // exclusive scan of the digit totals
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  for (int w = 0; w < SORT_WAVES; ++w) s_cnt[w][tid] = 0;
  {
    // exclusive scan of 256 digit totals: wave scan + 4 wave carries
    const unsigned v0 = digit_total[tid];
    unsigned x = v0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      unsigned y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    s_dig[tid] = x;  // inclusive within the wave
    __syncthreads();
    unsigned woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_dig[w * 64 + 63];
    __syncthreads();
    s_dig[tid] = woff + x - v0;
  }
  __syncthreads();

  const size_t wave_base = (size_t)blockIdx.x * SORT_TILE + (size_t)wave * (SORT_ITEMS * 64);
  unsigned long long k[SORT_ITEMS];
  unsigned v[SORT_ITEMS];
#pragma unroll
  for (int c = 0; c < SORT_ITEMS; ++c) {
    size_t i = wave_base + (size_t)c * 64 + lane;
    bool valid = i < n;
    k[c] = valid ? skey_in[i] : ~0ull;
    v[c] = valid ? sidx_in[i] : 0u;
    if (valid) atomicAdd(&s_cnt[wave][digit_of(k[c], v[c], shift, st)], 1u);
  }
  __syncthreads();
  // thread `tid` owns digit `tid`: turn counts into per-wave start offsets
  {
    unsigned run = s_dig[tid] + tile_offs[(size_t)tid * ntiles + blockIdx.x];
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) {
      unsigned cnt = s_cnt[w][tid];
      s_cnt[w][tid] = run;
      run += cnt;
    }
  }
  __syncthreads();

  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int c = 0; c < SORT_ITEMS; ++c) {
    size_t i = wave_base + (size_t)c * 64 + lane;
    bool valid = i < n;
    unsigned d = digit_of(k[c], v[c], shift, st);
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RADIX_BITS; ++b) {
      unsigned long long vote = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? vote : ~vote;
    }
    unsigned rank = (unsigned)__popcll(peers & lt_mask);
    unsigned count = (unsigned)__popcll(peers);
    volatile unsigned* wcnt = s_cnt[wave];
    unsigned start = 0;
    if (valid) start = wcnt[d];
    // all peers have read the running offset before the last peer bumps it:
    // LDS operations of one wave execute in issue order.
    __builtin_amdgcn_wave_barrier();
    if (valid && rank == count - 1) wcnt[d] = start + count;
    __builtin_amdgcn_wave_barrier();
    if (valid) {
      size_t pos = (size_t)start + rank;
      skey_out[pos] = k[c];
      sidx_out[pos] = v[c];
    }
  }
}

// (3') the same scatter with the tile put in order in LDS first.  In (3) every lane stores its own element straight to its
// final position: a wave's 64 consecutive inputs go to up to 64 different buckets, so a tile of 4096 keys is written as 2 x 4096
// stores that each touch one cache line for 8 (4) bytes.  Here the block first places its elements at their rank WITHIN the tile
// (digit-major, input order inside a digit: the same stable order) in LDS, then consecutive threads copy consecutive LDS entries
// out: the ~16 keys a bucket receives from a tile leave as one 128-byte run (64 bytes of indices), 16 times fewer write
// transactions.  The position of every element is the one (3) computes; the output is identical bit for bit.
__global__ void __launch_bounds__(SORT_THREADS)
k_sort_scatter_lds(size_t n, int shift, const unsigned long long* __restrict__ skey_in,
                   const unsigned* __restrict__ sidx_in, unsigned long long* __restrict__ skey_out,
                   unsigned* __restrict__ sidx_out, const unsigned* __restrict__ tile_offs,
                   unsigned ntiles, const unsigned* __restrict__ digit_total, SegTable st) {
  __shared__ unsigned long long s_key[SORT_TILE];
  __shared__ unsigned s_idx[SORT_TILE];
  __shared__ unsigned s_cnt[SORT_WAVES][RADIX];  // per-wave digit counts, then running LOCAL offsets
  __shared__ unsigned s_gbase[RADIX];            // global position of the tile's first element of a digit
  __shared__ unsigned s_lstart[RADIX];           // its position in the tile's digit-major order
  __shared__ unsigned s_scan[RADIX];
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  for (int w = 0; w < SORT_WAVES; ++w) s_cnt[w][tid] = 0;
  unsigned dig_excl;
  {
    // exclusive scan of 256 digit totals: wave scan + 4 wave carries
    const unsigned v0 = digit_total[tid];
    unsigned x = v0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      unsigned y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    s_scan[tid] = x;  // inclusive within the wave
    __syncthreads();
    unsigned woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_scan[w * 64 + 63];
    dig_excl = woff + x - v0;
  }
  __syncthreads();

  const size_t tile_base = (size_t)blockIdx.x * SORT_TILE;
  const size_t wave_base = tile_base + (size_t)wave * (SORT_ITEMS * 64);
  unsigned long long k[SORT_ITEMS];
  unsigned v[SORT_ITEMS];
#pragma unroll
  for (int c = 0; c < SORT_ITEMS; ++c) {
    size_t i = wave_base + (size_t)c * 64 + lane;
    bool valid = i < n;
    k[c] = valid ? skey_in[i] : ~0ull;
    v[c] = valid ? sidx_in[i] : 0u;
    if (valid) atomicAdd(&s_cnt[wave][digit_of(k[c], v[c], shift, st)], 1u);
  }
  __syncthreads();
  // thread `tid` owns digit `tid`: the tile's count of it, the exclusive scan of those counts over the digits (the tile's
  // digit-major order), per-wave local start offsets
  {
    unsigned cw[SORT_WAVES];
    unsigned tot = 0;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) { cw[w] = s_cnt[w][tid]; tot += cw[w]; }
    unsigned x = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      unsigned y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    s_scan[tid] = x;
    __syncthreads();
    unsigned woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_scan[w * 64 + 63];
    const unsigned lstart = woff + x - tot;
    s_lstart[tid] = lstart;
    s_gbase[tid] = dig_excl + tile_offs[(size_t)tid * ntiles + blockIdx.x];
    unsigned run = lstart;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) {
      s_cnt[w][tid] = run;
      run += cw[w];
    }
  }
  __syncthreads();

  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int c = 0; c < SORT_ITEMS; ++c) {
    size_t i = wave_base + (size_t)c * 64 + lane;
    bool valid = i < n;
    unsigned d = digit_of(k[c], v[c], shift, st);
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RADIX_BITS; ++b) {
      unsigned long long vote = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? vote : ~vote;
    }
    unsigned rank = (unsigned)__popcll(peers & lt_mask);
    unsigned count = (unsigned)__popcll(peers);
    volatile unsigned* wcnt = s_cnt[wave];
    unsigned start = 0;
    if (valid) start = wcnt[d];
    // all peers have read the running offset before the last peer bumps it:
    // LDS operations of one wave execute in issue order.
    __builtin_amdgcn_wave_barrier();
    if (valid && rank == count - 1) wcnt[d] = start + count;
    __builtin_amdgcn_wave_barrier();
    if (valid) {
      const unsigned lpos = start + rank;
      s_key[lpos] = k[c];
      s_idx[lpos] = v[c];
    }
  }
  __syncthreads();
  const unsigned nvalid = (unsigned)((n - tile_base) < (size_t)SORT_TILE ? (n - tile_base) : (size_t)SORT_TILE);
#pragma unroll
  for (int it = 0; it < SORT_ITEMS; ++it) {
    const unsigned i = (unsigned)it * SORT_THREADS + tid;
    if (i < nvalid) {
      const unsigned long long kk = s_key[i];
      const unsigned vv = s_idx[i];
      const unsigned d = digit_of(kk, vv, shift, st);
      const size_t pos = (size_t)s_gbase[d] + (i - s_lstart[d]);
      skey_out[pos] = kk;
      sidx_out[pos] = vv;
    }
  }
}

// final: ordered_index[off+i] = idx[i]; rank[idx[i]] = off+i  (reorder_spectrum.cpp:297-300)
__global__ void __launch_bounds__(256)
k_sort_finish(size_t off, size_t n, const unsigned* __restrict__ sidx, int32_t* __restrict__ rank,
              int32_t* __restrict__ ordered) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned src = sidx[i];
  if (ordered) ordered[off + i] = (int32_t)src;
  rank[src] = (int32_t)(off + i);
}

__global__ void __launch_bounds__(256)
k_iota(size_t n, int32_t* __restrict__ a, int32_t* __restrict__ b) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  a[i] = (int32_t)i;
  if (b) b[i] = (int32_t)i;
}

}  // namespace

extern "C" int ecckd_stable_argsort_bands_dev(ecckd_ctx* ctx, size_t nwav, const double* d_key,
                                              int nband, const int64_t* h_band_begin,
                                              const int64_t* h_band_end, int32_t* d_rank,
                                              int32_t* d_ordered_index) {
  ECCKD_REQUIRE(ctx, "ecckd_stable_argsort_bands_dev: ctx is NULL");
  ECCKD_REQUIRE(d_key && d_rank && h_band_begin && h_band_end && nband > 0,
                "ecckd_stable_argsort_bands_dev: bad argument");
  ECCKD_REQUIRE(nwav < (size_t)0x7fffffff, "ecckd_stable_argsort_bands_dev: nwav exceeds int32 range");
  if (nwav == 0) return ECCKD_OK;
  size_t nmax = 0;
  for (int b = 0; b < nband; ++b) {
    if (h_band_end[b] < h_band_begin[b]) continue;  // empty band
    ECCKD_REQUIRE(h_band_begin[b] >= 0 && (size_t)h_band_end[b] < nwav,
                  "ecckd_stable_argsort_bands_dev: band %d range [%lld,%lld] outside [0,%zu)", b,
                  (long long)h_band_begin[b], (long long)h_band_end[b], nwav);
    size_t n = (size_t)(h_band_end[b] - h_band_begin[b] + 1);
    if (n > nmax) nmax = n;
  }
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_iota, dim3((unsigned)((nwav + 255) / 256)), dim3(256), 0, ctx->stream, nwav, d_rank,
                     d_ordered_index);
  ECCKD_HIP_CHECK(hipGetLastError());
  if (nmax == 0) return ECCKD_OK;

  const size_t ntiles_max = (nmax + SORT_TILE - 1) / SORT_TILE;
  const size_t key_bytes = ecckd_align_up(nmax * sizeof(unsigned long long), 256);
  const size_t idx_bytes = ecckd_align_up(nmax * sizeof(unsigned), 256);
  const size_t hist_bytes = ecckd_align_up((size_t)RADIX * ntiles_max * sizeof(unsigned), 256) + 1024;
  ECCKD_CHECK(ecckd::ensure_scratch(ctx, 2 * key_bytes + 2 * idx_bytes + hist_bytes));
  char* p = (char*)ctx->scratch;
  unsigned long long* keys[2] = {(unsigned long long*)p, (unsigned long long*)(p + key_bytes)};
  p += 2 * key_bytes;
  unsigned* idxs[2] = {(unsigned*)p, (unsigned*)(p + idx_bytes)};
  p += 2 * idx_bytes;
  unsigned* hist = (unsigned*)p;
  unsigned* digit_total = (unsigned*)(p + hist_bytes - 1024);

  SegTable none;
  std::memset(&none, 0, sizeof none);
  const bool direct = std::getenv("ECCKD_SORT_DIRECT") != nullptr;     // A/B knob: every lane stores to its final position itself
  // several bands, ascending and disjoint (reorder_spectrum.cpp:277-289 builds them so): one nine-pass sort of the whole
  // spectrum instead of one eight-pass sort per band
  bool one_sort = nband > 1 && 2 * nband + 1 <= MAX_SEG && std::getenv("ECCKD_SORT_PER_BAND") == nullptr;
  long long prev_end = -1;
  for (int b = 0; b < nband && one_sort; ++b) {
    if (h_band_end[b] < h_band_begin[b]) continue;
    if (h_band_begin[b] <= prev_end) one_sort = false;
    prev_end = h_band_end[b];
  }
  if (one_sort) {
    SegTable st;
    std::memset(&st, 0, sizeof st);
    long long pos = 0;
    auto add = [&](long long begin, bool band) { st.begin[st.n] = (unsigned)begin; st.is_band[st.n] = band ? 1 : 0; ++st.n; };
    for (int b = 0; b < nband; ++b) {
      if (h_band_end[b] < h_band_begin[b]) continue;
      if (h_band_begin[b] > pos) add(pos, false);
      add(h_band_begin[b], true);
      pos = h_band_end[b] + 1;
    }
    if (pos < (long long)nwav) add(pos, false);
    const size_t n = nwav;
    const unsigned ntiles = (unsigned)((n + SORT_TILE - 1) / SORT_TILE);
    const unsigned eblocks = (unsigned)((n + 255) / 256);
    const size_t kb = ecckd_align_up(n * sizeof(unsigned long long), 256), ib = ecckd_align_up(n * sizeof(unsigned), 256);
    const size_t hb = ecckd_align_up((size_t)RADIX * ntiles * sizeof(unsigned), 256) + 1024;
    ECCKD_CHECK(ecckd::ensure_scratch(ctx, 2 * kb + 2 * ib + hb));
    char* q = (char*)ctx->scratch;
    unsigned long long* ks[2] = {(unsigned long long*)q, (unsigned long long*)(q + kb)};
    q += 2 * kb;
    unsigned* is[2] = {(unsigned*)q, (unsigned*)(q + ib)};
    q += 2 * ib;
    unsigned* hs = (unsigned*)q;
    unsigned* dt = (unsigned*)(q + hb - 1024);
    hipLaunchKernelGGL(k_sort_prepare_segments, dim3(eblocks), dim3(256), 0, ctx->stream, n, d_key, st, ks[0], is[0]);
    int cur = 0;
    for (int pass = 0; pass <= 64 / RADIX_BITS; ++pass) {
      const int shift = pass * RADIX_BITS;      // the last one, 64: by segment
      hipLaunchKernelGGL(k_sort_hist, dim3(ntiles), dim3(SORT_THREADS), 0, ctx->stream, n, shift, ks[cur], is[cur], st, hs, ntiles);
      hipLaunchKernelGGL(k_sort_scan_rows, dim3(RADIX), dim3(256), 0, ctx->stream, hs, ntiles, dt);
      if (direct)
        hipLaunchKernelGGL(k_sort_scatter, dim3(ntiles), dim3(SORT_THREADS), 0, ctx->stream, n, shift, ks[cur], is[cur], ks[cur ^ 1],
                           is[cur ^ 1], hs, ntiles, dt, st);
      else
        hipLaunchKernelGGL(k_sort_scatter_lds, dim3(ntiles), dim3(SORT_THREADS), 0, ctx->stream, n, shift, ks[cur], is[cur], ks[cur ^ 1],
                           is[cur ^ 1], hs, ntiles, dt, st);
      cur ^= 1;
    }
    hipLaunchKernelGGL(k_sort_finish, dim3(eblocks), dim3(256), 0, ctx->stream, (size_t)0, n, is[cur], d_rank, d_ordered_index);
    ECCKD_HIP_CHECK(hipGetLastError());
    return ECCKD_OK;
  }

  for (int b = 0; b < nband; ++b) {
    if (h_band_end[b] < h_band_begin[b]) continue;
    const size_t off = (size_t)h_band_begin[b];
    const size_t n = (size_t)(h_band_end[b] - h_band_begin[b] + 1);
    const unsigned ntiles = (unsigned)((n + SORT_TILE - 1) / SORT_TILE);
    const unsigned eblocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_sort_prepare, dim3(eblocks), dim3(256), 0, ctx->stream, off, n, d_key, keys[0], idxs[0]);
    int cur = 0;
    for (int pass = 0; pass < 64 / RADIX_BITS; ++pass) {
      const int shift = pass * RADIX_BITS;
      hipLaunchKernelGGL(k_sort_hist, dim3(ntiles), dim3(SORT_THREADS), 0, ctx->stream, n, shift, keys[cur], idxs[cur], none,
                         hist, ntiles);
      hipLaunchKernelGGL(k_sort_scan_rows, dim3(RADIX), dim3(256), 0, ctx->stream, hist, ntiles, digit_total);
      if (direct)
        hipLaunchKernelGGL(k_sort_scatter, dim3(ntiles), dim3(SORT_THREADS), 0, ctx->stream, n, shift,
                           keys[cur], idxs[cur], keys[cur ^ 1], idxs[cur ^ 1], hist, ntiles, digit_total, none);
      else
        hipLaunchKernelGGL(k_sort_scatter_lds, dim3(ntiles), dim3(SORT_THREADS), 0, ctx->stream, n, shift,
                           keys[cur], idxs[cur], keys[cur ^ 1], idxs[cur ^ 1], hist, ntiles, digit_total, none);
      cur ^= 1;
    }
    hipLaunchKernelGGL(k_sort_finish, dim3(eblocks), dim3(256), 0, ctx->stream, off, n, idxs[cur], d_rank,
                       d_ordered_index);
    ECCKD_HIP_CHECK(hipGetLastError());
  }
  return ECCKD_OK;
}
