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
// keys per thread: a block's tile is SORT_THREADS * ITEMS keys (a template argument of the pass kernels; sort_items() picks it)

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

// k_sort_prepare with the tile histograms of the first pass (the key's lowest digit) taken on the way: the first k_sort_hist
// launch - another read of every key - is not needed.  grid = tiles of 256 * ITEMS keys.
template <int SORT_ITEMS>
__global__ void __launch_bounds__(SORT_THREADS)
k_sort_prepare_hist(size_t off, size_t n, const double* __restrict__ key, unsigned long long* __restrict__ skey,
                    unsigned* __restrict__ sidx, unsigned* __restrict__ tile_hist, unsigned ntiles) {
  constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;
  __shared__ unsigned s_hist[RADIX];
  const int tid = threadIdx.x;
  s_hist[tid] = 0;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * SORT_TILE;
#pragma unroll
  for (int it = 0; it < SORT_ITEMS; ++it) {
    const size_t i = base + (size_t)it * SORT_THREADS + tid;
    if (i < n) {
      const unsigned long long k = key_to_sortable(key[off + i]);
      skey[i] = k;
      sidx[i] = (unsigned)(off + i);
      atomicAdd(&s_hist[(unsigned)k & (RADIX - 1)], 1u);
    }
  }
  __syncthreads();
  tile_hist[(size_t)tid * ntiles + blockIdx.x] = s_hist[tid];
}

// (1) tile histogram -> tile_hist[digit * ntiles + tile]
template <int SORT_ITEMS>
__global__ void __launch_bounds__(SORT_THREADS)
k_sort_hist(size_t n, int shift, const unsigned long long* __restrict__ skey, const unsigned* __restrict__ sidx, SegTable st,
            unsigned* __restrict__ tile_hist, unsigned ntiles) {
  // eight copies of the histogram, a lane adds to copy (lane & 7): the high digits of a sorting key take a handful of values, and
  // 64 lanes adding to the same few LDS words are served one after the other (the passes over the exponent bytes took twice as
  // long as those over the mantissa)
  constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;
  constexpr int COPIES = 8;
  __shared__ unsigned s_hist[COPIES][RADIX];
  const int tid = threadIdx.x;
#pragma unroll
  for (int c = 0; c < COPIES; ++c) s_hist[c][tid] = 0;  // SORT_THREADS == RADIX
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * SORT_TILE;
  unsigned* mine = s_hist[tid & (COPIES - 1)];
#pragma unroll
  for (int it = 0; it < SORT_ITEMS; ++it) {
    size_t i = base + (size_t)it * SORT_THREADS + tid;
    if (i < n) {
      unsigned d = digit_of(skey[i], shift < 64 ? 0u : sidx[i], shift, st);
      atomicAdd(&mine[d], 1u);
    }
  }
  __syncthreads();
  unsigned total = 0;
#pragma unroll
  for (int c = 0; c < COPIES; ++c) total += s_hist[c][tid];
  tile_hist[(size_t)tid * ntiles + blockIdx.x] = total;
}

// (2) one block per digit: exclusive scan of that digit's row of per-tile counts (in place) and the row total ->
// digit_total[digit].  Every thread takes eight consecutive counts (the loads of a round are in flight together), the block
// scans the threads' sums once per round of 2 048 counts.  The scan over the 256 digit totals is done by every scatter block
// in LDS.
__global__ void __launch_bounds__(256)
k_sort_scan_rows(unsigned* __restrict__ tile_hist, unsigned ntiles, unsigned* __restrict__ digit_total) {
  __shared__ unsigned s_wave[4];
  constexpr int PER = 8;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  unsigned* row = tile_hist + (size_t)blockIdx.x * ntiles;
  unsigned carry = 0;
  for (unsigned base = 0; base < ntiles; base += 256 * PER) {
    const unsigned first = base + (unsigned)tid * PER;
    unsigned v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) v[j] = (first + j < ntiles) ? row[first + j] : 0u;
    unsigned mine = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) mine += v[j];
    unsigned x = mine;   // inclusive scan of the threads' sums inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      unsigned y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    __syncthreads();     // (the previous round's s_wave has been read)
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    unsigned woff = 0, all = 0;
    for (int w = 0; w < 4; ++w) { if (w < wave) woff += s_wave[w]; all += s_wave[w]; }
    unsigned run = carry + woff + x - mine;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      if (first + j < ntiles) row[first + j] = run;
      run += v[j];
    }
    carry += all;
  }
  if (tid == 0) digit_total[blockIdx.x] = carry;
}

// (3) stable scatter.  Wave w of a block owns the contiguous sub-tile
// [w*1024, (w+1)*1024) of the block's tile as 16 chunks of 64 consecutive
// elements, so (chunk, lane) order is input order.
template <int SORT_ITEMS>
__global__ void __launch_bounds__(SORT_THREADS)
k_sort_scatter(size_t n, int shift, const unsigned long long* __restrict__ skey_in,
               const unsigned* __restrict__ sidx_in, unsigned long long* __restrict__ skey_out,
               unsigned* __restrict__ sidx_out, const unsigned* __restrict__ tile_offs,
               unsigned ntiles, const unsigned* __restrict__ digit_total, SegTable st) {
  constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;
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
// FINAL (the last pass of a sort): the sorted arrays are not written; an element's position IS its rank, stored straight into
// rank[] (and its index into ordered_index[]) - what k_sort_finish does in a launch of its own after the direct scatter.
template <int SORT_ITEMS, bool FINAL>
__global__ void __launch_bounds__(SORT_THREADS)
k_sort_scatter_lds(size_t n, int shift, const unsigned long long* __restrict__ skey_in,
                   const unsigned* __restrict__ sidx_in, unsigned long long* __restrict__ skey_out,
                   unsigned* __restrict__ sidx_out, const unsigned* __restrict__ tile_offs,
                   unsigned ntiles, const unsigned* __restrict__ digit_total, SegTable st,
                   size_t off, int32_t* __restrict__ rank_out, int32_t* __restrict__ ordered_out) {
  constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;
  __shared__ unsigned long long s_key[SORT_TILE];
  __shared__ unsigned s_idx[SORT_TILE];
  __shared__ unsigned s_cnt[SORT_WAVES][RADIX];  // per-wave digit counts, then running LOCAL offsets
  __shared__ unsigned s_gbase[RADIX];            // global position of the tile's first element of a digit
  __shared__ unsigned s_lstart[RADIX];           // its position in the tile's digit-major order
  __shared__ unsigned s_scan[RADIX];
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  // everything this block needs from memory leaves in one go: the digit totals, the tile's offsets, its keys and indices
  const unsigned v0 = digit_total[tid];
  const unsigned toff = tile_offs[(size_t)tid * ntiles + blockIdx.x];
  const size_t tile_base = (size_t)blockIdx.x * SORT_TILE;
  const size_t wave_base = tile_base + (size_t)wave * (SORT_ITEMS * 64);
  unsigned long long k[SORT_ITEMS];
  unsigned v[SORT_ITEMS];
#pragma unroll
  for (int c = 0; c < SORT_ITEMS; ++c) {
    const size_t i = wave_base + (size_t)c * 64 + lane;
    const bool valid = i < n;
    k[c] = valid ? skey_in[i] : ~0ull;
    v[c] = valid ? sidx_in[i] : 0u;
  }
  for (int w = 0; w < SORT_WAVES; ++w) s_cnt[w][tid] = 0;
  unsigned dig_excl;
  {
    // exclusive scan of 256 digit totals: wave scan + 4 wave carries
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

  // Ranks inside the wave, once.  The lanes of a chunk of 64 consecutive elements that hold the same digit find each other
  // with ballots (`peers`); the LAST of them adds their number to the wave's count of that digit and gets back what the
  // earlier chunks of this wave had put there - LDS operations of one wave execute in issue order, the sixteen atomics are in
  // flight together - and hands it to its peers: wl = the element's place among the wave's elements of its digit.
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  unsigned before[SORT_ITEMS];      // (leader only) the wave's earlier elements of the digit
  unsigned rl[SORT_ITEMS];          // rank among the peers | leader's lane << 8
#pragma unroll
  for (int c = 0; c < SORT_ITEMS; ++c) {
    const size_t i = wave_base + (size_t)c * 64 + lane;
    const bool valid = i < n;
    const unsigned d = digit_of(k[c], v[c], shift, st);
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RADIX_BITS; ++b) {
      const unsigned long long vote = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? vote : ~vote;
    }
    const unsigned rank = (unsigned)__popcll(peers & lt_mask);
    const unsigned count = (unsigned)__popcll(peers);
    const unsigned leader = valid ? 63u - (unsigned)__clzll((long long)peers) : (unsigned)lane;
    before[c] = 0u;
    if (valid && leader == (unsigned)lane) before[c] = atomicAdd(&s_cnt[wave][d], count);
    rl[c] = rank | (leader << 8);
  }
  unsigned wl[SORT_ITEMS];
#pragma unroll
  for (int c = 0; c < SORT_ITEMS; ++c) wl[c] = (unsigned)__shfl((int)before[c], (int)(rl[c] >> 8), 64) + (rl[c] & 0xffu);
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
    s_gbase[tid] = dig_excl + toff;
    unsigned run = lstart;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) {
      s_cnt[w][tid] = run;
      run += cw[w];
    }
  }
  __syncthreads();

#pragma unroll
  for (int c = 0; c < SORT_ITEMS; ++c) {
    const size_t i = wave_base + (size_t)c * 64 + lane;
    if (i < n) {
      const unsigned lpos = s_cnt[wave][digit_of(k[c], v[c], shift, st)] + wl[c];
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
      if (FINAL) {
        rank_out[vv] = (int32_t)(off + pos);                     // reorder_spectrum.cpp:297-300
        if (ordered_out) ordered_out[off + pos] = (int32_t)vv;
      } else {
        skey_out[pos] = kk;
        sidx_out[pos] = vv;
      }
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

// keys per thread of the pass kernels: 15 (3 840-key tiles: the scatter's LDS, 12 B per key + 7 KB, then lets three blocks share a
// CU) unless ECCKD_SORT_ITEMS says 8, 12, 15, 16 or 20 (an A/B knob of tools/sort_probe.py; 7.2e6 keys in one band / thirteen:
// 0.82 / 0.86, 0.83 / 0.92, 0.78 / 0.87, 0.82 / 0.97, 0.84 / 0.94 ms)
int sort_items() {
  static const int items = [] {
    const char* e = std::getenv("ECCKD_SORT_ITEMS");
    const int v = e ? std::atoi(e) : 15;
    return (v == 8 || v == 12 || v == 15 || v == 16 || v == 20) ? v : 15;
  }();
  return items;
}

// one pass: tile histograms (unless the caller has them already), their scan, the scatter; the LAST pass of a sort stores the
// ranks itself (final_rank != nullptr: rank[index] = off + position, ordered[off + position] = index)
struct SortPassArgs {
  size_t n; int shift; bool direct, have_hist;
  const unsigned long long* kin; const unsigned* iin; unsigned long long* kout; unsigned* iout;
  unsigned* hist; unsigned* digit_total;
  size_t off; int32_t* final_rank; int32_t* final_ordered;
};
template <int ITEMS>
void sort_pass_t(hipStream_t stream, const SortPassArgs& a, const SegTable& st) {
  const unsigned ntiles = (unsigned)((a.n + (size_t)SORT_THREADS * ITEMS - 1) / ((size_t)SORT_THREADS * ITEMS));
  if (!a.have_hist)
    hipLaunchKernelGGL((k_sort_hist<ITEMS>), dim3(ntiles), dim3(SORT_THREADS), 0, stream, a.n, a.shift, a.kin, a.iin, st, a.hist, ntiles);
  hipLaunchKernelGGL(k_sort_scan_rows, dim3(RADIX), dim3(256), 0, stream, a.hist, ntiles, a.digit_total);
  if (a.direct)
    hipLaunchKernelGGL((k_sort_scatter<ITEMS>), dim3(ntiles), dim3(SORT_THREADS), 0, stream, a.n, a.shift, a.kin, a.iin, a.kout, a.iout, a.hist, ntiles,
                       a.digit_total, st);
  else if (a.final_rank)
    hipLaunchKernelGGL((k_sort_scatter_lds<ITEMS, true>), dim3(ntiles), dim3(SORT_THREADS), 0, stream, a.n, a.shift, a.kin, a.iin, a.kout, a.iout, a.hist,
                       ntiles, a.digit_total, st, a.off, a.final_rank, a.final_ordered);
  else
    hipLaunchKernelGGL((k_sort_scatter_lds<ITEMS, false>), dim3(ntiles), dim3(SORT_THREADS), 0, stream, a.n, a.shift, a.kin, a.iin, a.kout, a.iout, a.hist,
                       ntiles, a.digit_total, st, (size_t)0, (int32_t*)nullptr, (int32_t*)nullptr);
}
void sort_pass(hipStream_t stream, const SortPassArgs& a, const SegTable& st) {
  switch (sort_items()) {
    case 8: sort_pass_t<8>(stream, a, st); break;
    case 12: sort_pass_t<12>(stream, a, st); break;
    case 16: sort_pass_t<16>(stream, a, st); break;
    case 20: sort_pass_t<20>(stream, a, st); break;
    default: sort_pass_t<15>(stream, a, st);
  }
}
template <int ITEMS>
void sort_prepare_hist_t(hipStream_t stream, size_t off, size_t n, const double* key, unsigned long long* skey, unsigned* sidx, unsigned* hist) {
  const unsigned ntiles = (unsigned)((n + (size_t)SORT_THREADS * ITEMS - 1) / ((size_t)SORT_THREADS * ITEMS));
  hipLaunchKernelGGL((k_sort_prepare_hist<ITEMS>), dim3(ntiles), dim3(SORT_THREADS), 0, stream, off, n, key, skey, sidx, hist, ntiles);
}
void sort_prepare_hist(hipStream_t stream, size_t off, size_t n, const double* key, unsigned long long* skey, unsigned* sidx, unsigned* hist) {
  switch (sort_items()) {
    case 8: sort_prepare_hist_t<8>(stream, off, n, key, skey, sidx, hist); break;
    case 12: sort_prepare_hist_t<12>(stream, off, n, key, skey, sidx, hist); break;
    case 16: sort_prepare_hist_t<16>(stream, off, n, key, skey, sidx, hist); break;
    case 20: sort_prepare_hist_t<20>(stream, off, n, key, skey, sidx, hist); break;
    default: sort_prepare_hist_t<15>(stream, off, n, key, skey, sidx, hist);
  }
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
  // points outside every band keep their places (rank = index): written first, unless the sort below gives EVERY point its rank
  // (one band over the whole spectrum - the fsck structure -, or the one sort of all bands and gaps together)
  auto identity = [&]() {
    hipLaunchKernelGGL(k_iota, dim3((unsigned)((nwav + 255) / 256)), dim3(256), 0, ctx->stream, nwav, d_rank, d_ordered_index);
  };
  if (nmax == 0) {
    identity();
    ECCKD_HIP_CHECK(hipGetLastError());
    return ECCKD_OK;
  }

  const size_t SORT_TILE = (size_t)SORT_THREADS * sort_items();
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
    const int npass = 64 / RADIX_BITS + 1;      // the last one, shift 64: by segment
    for (int pass = 0; pass < npass; ++pass) {
      const bool last = pass == npass - 1 && !direct;
      const SortPassArgs a{n, pass * RADIX_BITS, direct, false, ks[cur], is[cur], ks[cur ^ 1], is[cur ^ 1], hs, dt,
                           0, last ? d_rank : nullptr, last ? d_ordered_index : nullptr};
      sort_pass(ctx->stream, a, st);
      cur ^= 1;
    }
    if (direct) hipLaunchKernelGGL(k_sort_finish, dim3(eblocks), dim3(256), 0, ctx->stream, (size_t)0, n, is[cur], d_rank, d_ordered_index);
    ECCKD_HIP_CHECK(hipGetLastError());
    return ECCKD_OK;
  }

  bool whole = false;
  for (int b = 0; b < nband; ++b) whole |= h_band_begin[b] == 0 && h_band_end[b] == (int64_t)nwav - 1;
  if (!whole) identity();
  for (int b = 0; b < nband; ++b) {
    if (h_band_end[b] < h_band_begin[b]) continue;
    const size_t off = (size_t)h_band_begin[b];
    const size_t n = (size_t)(h_band_end[b] - h_band_begin[b] + 1);
    const unsigned eblocks = (unsigned)((n + 255) / 256);
    if (direct) hipLaunchKernelGGL(k_sort_prepare, dim3(eblocks), dim3(256), 0, ctx->stream, off, n, d_key, keys[0], idxs[0]);
    else sort_prepare_hist(ctx->stream, off, n, d_key, keys[0], idxs[0], hist);
    int cur = 0;
    const int npass = 64 / RADIX_BITS;
    for (int pass = 0; pass < npass; ++pass) {
      const bool last = pass == npass - 1 && !direct;
      const SortPassArgs a{n, pass * RADIX_BITS, direct, pass == 0 && !direct, keys[cur], idxs[cur], keys[cur ^ 1], idxs[cur ^ 1], hist, digit_total,
                           off, last ? d_rank : nullptr, last ? d_ordered_index : nullptr};
      sort_pass(ctx->stream, a, none);
      cur ^= 1;
    }
    if (direct) hipLaunchKernelGGL(k_sort_finish, dim3(eblocks), dim3(256), 0, ctx->stream, off, n, idxs[cur], d_rank, d_ordered_index);
    ECCKD_HIP_CHECK(hipGetLastError());
  }
  return ECCKD_OK;
}
