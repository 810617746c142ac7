// inflate.hip - zlib / DEFLATE (RFC 1950, RFC 1951) streams inflated on the device, one wavefront per stream.
//
// The CKDMIP spectra are NetCDF-4 files whose chunks went through HDF5's shuffle and deflate filters
// (OutputDataFile.cpp:350-359 writes the reference's own files the same way); the reference leaves the inflation to the HDF5
// library on the one reading thread, and its documentation names that reading as where the wall-clock time goes
// (doc/ecckd_documentation.tex:226-229, :526-528).  Here the RAW chunks travel over PCIe (a half to a third of the bytes)
// and are inflated where the values are needed; a second kernel undoes the shuffle filter, converts to the requested type
// and places every chunk's part of the requested box (nc_stream.hip).
//
// A DEFLATE stream is sequential: the wavefront decodes it as ONE thread of control - every lane runs the same decoder on
// the same state, the values that steer it are made wave-uniform (readfirstlane) so that the bit buffer, the table look-ups'
// results and the branches live on the scalar unit - and uses its 64 lanes where the format has parallel work: fetching
// the input in 512-byte pieces, filling the Huffman look-up tables, copying the bytes of a match (a match of length L at
// distance D reads only bytes that lie BEFORE the current position, also when D < L: byte i comes from position
// pos - D + i mod D).  The last 32 KB of output (the DEFLATE window) are kept in LDS; every output byte also goes straight
// to its place in device memory.  LDS per wave ~37 KB: four streams per CU, 1 024 in flight on the chip.
#include "common.hpp"

#include <cstring>
#include <vector>

namespace {

typedef unsigned char u8;
typedef unsigned short u16;
typedef unsigned int u32;
typedef unsigned long long u64;

constexpr int WINDOW = 32768;
constexpr int IN_BYTES = 1024;       // two halves of 512 bytes
constexpr int LIT_BITS = 10, DIST_BITS = 8;
constexpr size_t IN_SLACK = 2048;    // bytes behind a stream that must be readable (zero or anything): fetches end below
                                     // round_up(in_bytes, 512) + 1024

struct StreamDesc {
  u64 in_off, in_bytes;    // zlib stream in the input buffer; in_off is a multiple of 16
  u64 out_off, out_bytes;  // where its bytes go in the output buffer, and how many there must be
};

enum : int { INF_OK = 0, INF_BAD_HEADER = 1, INF_BAD_BLOCK = 2, INF_BAD_CODE = 3, INF_BAD_DISTANCE = 4, INF_OVERRUN = 5,
             INF_SHORT = 6, INF_INPUT = 7, INF_CHECKSUM = 8 };

__constant__ u16 c_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ u8 c_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ u16 c_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097,
                                    6145, 8193, 12289, 16385, 24577};
__constant__ u8 c_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ u8 c_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ __forceinline__ u32 uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }

// One Huffman code: primary look-up table of 2^BITS entries (symbol << 4 | length, 0 = longer than BITS or unused) plus the
// canonical description for the longer codes (count per length, symbols sorted by length then value: puff.c's decode).
template <int BITS, int MAXSYM>
struct Huff {
  u16 table[1 << BITS];
  u16 count[16];
  u16 symbol[MAXSYM];
};

struct Lds {
  u8 window[WINDOW];
  u32 in[IN_BYTES / 4];
  Huff<LIT_BITS, 288> lit;
  Huff<DIST_BITS, 32> dist;
  Huff<7, 19> cl;
  u64 st_buf;      // the bit reader and the output position while a rare, out-of-line step runs (save_state / load_state)
  u32 st_cnt, st_ipos, st_opos, st_limit;
  u8 len[384];     // code lengths: [0, 19) the code-length code's, [32, 32 + 286 + 30) the block's (fixed block: [0, 318))
};

// Build a code from the lengths len[0..n).  Cooperative: the wave's lanes fill the replicated table entries.  Returns false
// for an over-subscribed set of lengths, and for an incomplete one as zlib does (inftrees.c: `left > 0 && (type == CODES ||
// max != 1)`): `incomplete` = 0 never (the code-length code), 1 only if no code is longer than one bit (a literal/length or
// distance code of a single symbol) or there is no code at all (a block without matches), 2 always (the fixed block's
// 30 five-bit distance codes).
// Rare (once per DEFLATE block): kept out of line so that the symbol loop stays small and keeps its registers.
__device__ __noinline__ bool huff_build(u16* table, int bits, u16* count, u16* symbol, const u8* len, int n, int lane,
                                        int incomplete = 0) {
  for (int i = lane; i < (1 << bits); i += 64) table[i] = 0;
  __builtin_amdgcn_wave_barrier();
  // every lane runs the same serial passes over the (at most 288) symbols: the results are wave-uniform
  u32 cnt[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) cnt[b] = 0;
#pragma unroll 1
  for (int s = 0; s < n; ++s) {
    const u32 l = uni(len[s]);
#pragma unroll
    for (int b = 1; b < 16; ++b) cnt[b] += (l == (u32)b) ? 1u : 0u;
  }
  int left = 1;
  u32 offs[16], next[16];
  offs[1] = 0;
  u32 code = 0;
#pragma unroll
  for (int b = 1; b < 16; ++b) {
    left = (left << 1) - (int)cnt[b];
    if (left < 0) return false;
    if (b < 15) offs[b + 1] = offs[b] + cnt[b];
    code = (code + (b > 1 ? cnt[b - 1] : 0u)) << 1;
    next[b] = code;
  }
  if (left > 0 && incomplete < 2) {                // incomplete
    u32 longer = 0;                                // codes of more than one bit
#pragma unroll
    for (int b = 2; b < 16; ++b) longer += cnt[b];
    if (incomplete == 0 || longer != 0) return false;
  }
  if (lane == 0) {
    count[0] = 0;
#pragma unroll
    for (int b = 1; b < 16; ++b) count[b] = (u16)cnt[b];
  }
#pragma unroll 1
  for (int s = 0; s < n; ++s) {
    const u32 l = uni(len[s]);
    if (l == 0) continue;
    u32 c = 0, o = 0;
#pragma unroll
    for (int b = 1; b < 16; ++b)
      if (l == (u32)b) { c = next[b]++; o = offs[b]++; }
    if (lane == 0) symbol[o] = (u16)s;
    if (l <= (u32)bits) {
      const u32 rev = __builtin_bitreverse32(c) >> (32 - l);          // the stream presents the code's most significant bit first
      const u32 reps = 1u << ((u32)bits - l);
      for (u32 j = (u32)lane; j < reps; j += 64) table[rev | (j << l)] = (u16)((s << 4) | l);
    }
  }
  __builtin_amdgcn_wave_barrier();
  return true;
}

// a code longer than the primary table's bits: canonical decoding, one bit at a time (puff.c) -> (symbol << 8) | length, or ~0
__device__ __noinline__ u32 decode_long(const u16* count, const u16* symbol, u32 bits) {
  int code = 0, first = 0, index = 0;
#pragma unroll 1
  for (int l = 1; l <= 15; ++l) {
    code |= (int)(bits & 1u);
    bits >>= 1;
    const int c = (int)uni(count[l]);
    if (code - c < first) return (uni(symbol[index + (code - first)]) << 8) | (u32)l;
    index += c;
    first += c;
    first <<= 1;
    code <<= 1;
  }
  return ~0u;
}

// the 512 input bytes [first, first + 512) into their half of the LDS window; `src` is 16-byte aligned, first a multiple of 512
__device__ __forceinline__ void fetch_half(Lds& L, const u8* src, u32 first, int lane) {
  const u64 v = *reinterpret_cast<const u64*>(src + first + (u32)lane * 8);
  reinterpret_cast<u64*>(L.in)[((first >> 3) & (IN_BYTES / 8 - 1)) + lane] = v;
  __builtin_amdgcn_wave_barrier();
}

// Bit reader: buf holds cnt valid bits (wave-uniform); the next input word is at byte ipos (a multiple of 4).  The LDS window
// holds the 512-byte half that contains ipos; the half behind it is fetched when ipos enters a half.
struct Bits {
  u64 buf;
  u32 cnt;
  u32 ipos;
  u32 limit;      // no input is fetched from byte `limit` on (a multiple of 512 behind the stream's end, inside the slack): a damaged
                  // stream whose codes keep consuming the zero padding is fed zeros until it runs into one of the decoder's checks
};

__device__ __forceinline__ void refill(Lds& L, Bits& b, const u8* src, int lane) {
  if (b.cnt <= 32) {
    if ((b.ipos & 511u) == 0 && b.ipos + 512 < b.limit) fetch_half(L, src, b.ipos + 512, lane);
    const u32 w = b.ipos < b.limit ? uni(L.in[(b.ipos >> 2) & (IN_BYTES / 4 - 1)]) : 0u;
    b.buf |= (u64)w << b.cnt;
    b.cnt += 32;
    b.ipos += 4;
  }
}

// The rare steps run out of line; handing them the bit reader by reference would pin it in scratch memory for the whole kernel,
// so it travels through LDS instead.
__device__ __forceinline__ void save_state(Lds& L, const Bits& b, u32 opos, int lane) {
  if (lane == 0) { L.st_buf = b.buf; L.st_cnt = b.cnt; L.st_ipos = b.ipos; L.st_opos = opos; L.st_limit = b.limit; }
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void load_state(Lds& L, Bits& b, u32& opos) {
  __builtin_amdgcn_wave_barrier();
  const u32 lo = uni((u32)L.st_buf), hi = uni((u32)(L.st_buf >> 32));
  b.buf = ((u64)hi << 32) | lo;
  b.cnt = uni(L.st_cnt);
  b.ipos = uni(L.st_ipos);
  b.limit = uni(L.st_limit);
  opos = uni(L.st_opos);
}

// continue reading at byte `at` (behind a stored block)
__device__ __forceinline__ void reposition(Lds& L, Bits& b, const u8* src, u32 at, int lane) {
  const u32 word = at & ~3u;
  if ((word & ~511u) < b.limit) fetch_half(L, src, word & ~511u, lane);
  if ((word & 511u) && (word & ~511u) + 512 < b.limit) fetch_half(L, src, (word & ~511u) + 512, lane);   // else refill() fetches it with the first word
  b.buf = 0; b.cnt = 0; b.ipos = word;
  refill(L, b, src, lane);
  const u32 drop = 8 * (at & 3u);
  b.buf >>= drop;
  b.cnt -= drop;
}

__device__ __forceinline__ u32 take(Bits& b, u32 n) {   // n <= 16; the caller has refilled
  const u32 v = (u32)b.buf & ((1u << n) - 1u);
  b.buf >>= n;
  b.cnt -= n;
  return v;
}

// one symbol; returns the symbol or -1.  The caller has refilled (> 32 bits valid, or the zero padding behind the stream).
template <int BITS, int MAXSYM>
__device__ __forceinline__ int decode(const Huff<BITS, MAXSYM>& h, Bits& b) {
  const u32 e = uni(h.table[(u32)b.buf & ((1u << BITS) - 1u)]);
  u32 l = e & 15u, sym = e >> 4;
  if (__builtin_expect(e == 0, 0)) {
    const u32 r = uni(decode_long(h.count, h.symbol, (u32)b.buf));     // a function's result arrives in a vector register
    if (r == ~0u) return -1;
    l = r & 255u;
    sym = r >> 8;
  }
  b.buf >>= l;
  b.cnt -= l;
  return (int)sym;
}

// The block header of a dynamic-Huffman block (RFC 1951 3.2.7) -> the two codes' tables.  Rare: out of line.
__device__ __noinline__ int read_dynamic_header(Lds& L, const u8* src, int lane) {
  Bits b;
  u32 opos;
  load_state(L, b, opos);
  refill(L, b, src, lane);
  const int nlen = (int)take(b, 5) + 257, ndist = (int)take(b, 5) + 1, ncode = (int)take(b, 4) + 4;
  if (nlen > 286 || ndist > 30) return INF_BAD_BLOCK;
  if (lane < 19) L.len[lane] = 0;
  __builtin_amdgcn_wave_barrier();
#pragma unroll 1
  for (int i = 0; i < ncode; ++i) {
    refill(L, b, src, lane);
    const u32 v = take(b, 3);
    if (lane == 0) L.len[c_cl_order[i]] = (u8)v;
  }
  __builtin_amdgcn_wave_barrier();
  if (!uni(huff_build(L.cl.table, 7, L.cl.count, L.cl.symbol, L.len, 19, lane))) return INF_BAD_CODE;
  // the code lengths of the literal/length and distance codes, run-length coded; they are decoded into L.len + 32 so that
  // the code-length code's own lengths stay where L.cl was built from
  u8* lens = L.len + 32;
  int i = 0;
  u32 prev = 0;
#pragma unroll 1
  while (i < nlen + ndist) {
    refill(L, b, src, lane);
    const int sym = decode(L.cl, b);
    if (sym < 0) return INF_BAD_CODE;
    if (sym < 16) {
      if (lane == 0) lens[i] = (u8)sym;
      prev = (u32)sym;
      ++i;
    } else {
      u32 rep, val = 0;
      if (sym == 16) { if (i == 0) return INF_BAD_CODE; val = prev; rep = 3 + take(b, 2); }
      else if (sym == 17) rep = 3 + take(b, 3);
      else rep = 11 + take(b, 7);
      if (i + (int)rep > nlen + ndist) return INF_BAD_CODE;
      for (u32 j = (u32)lane; j < rep; j += 64) lens[i + j] = (u8)val;
      prev = val;
      i += (int)rep;
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (uni(lens[256]) == 0) return INF_BAD_CODE;       // no end-of-block code
  if (!uni(huff_build(L.lit.table, LIT_BITS, L.lit.count, L.lit.symbol, lens, nlen, lane, 1))) return INF_BAD_CODE;
  if (!uni(huff_build(L.dist.table, DIST_BITS, L.dist.count, L.dist.symbol, lens + nlen, ndist, lane, 1))) return INF_BAD_CODE;
  save_state(L, b, opos, lane);
  return INF_OK;
}

__device__ __noinline__ void fixed_tables(Lds& L, int lane) {
  for (int i = lane; i < 288; i += 64) L.len[i] = (u8)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
  if (lane < 30) L.len[288 + lane] = 5;
  __builtin_amdgcn_wave_barrier();
  huff_build(L.lit.table, LIT_BITS, L.lit.count, L.lit.symbol, L.len, 288, lane);
  huff_build(L.dist.table, DIST_BITS, L.dist.count, L.dist.symbol, L.len + 288, 30, lane, 2);     // (30 five-bit codes: incomplete by design)
}

// a stored block: to the byte boundary, LEN, NLEN, then LEN bytes as they are
__device__ __noinline__ int stored_block(Lds& L, const u8* src, u32 in_bytes, u8* dst, u32 out_bytes, int lane) {
  Bits b;
  u32 opos;
  load_state(L, b, opos);
  take(b, b.cnt & 7u);
  refill(L, b, src, lane);
  const u32 len = take(b, 16);
  refill(L, b, src, lane);
  const u32 nlen = take(b, 16);
  if ((len ^ 0xffffu) != nlen) return INF_BAD_BLOCK;
  const u32 from = b.ipos - b.cnt / 8;      // the buffer holds whole bytes now
  if ((u64)from + len > in_bytes) return INF_INPUT;
  if ((u64)opos + len > out_bytes) return INF_OVERRUN;
  for (u32 i = (u32)lane; i < len; i += 64) {
    const u8 v = src[from + i];
    dst[opos + i] = v;
    L.window[(opos + i) & (WINDOW - 1)] = v;
  }
  __builtin_amdgcn_wave_barrier();
  opos += len;
  reposition(L, b, src, from + len, lane);
  save_state(L, b, opos, lane);
  return INF_OK;
}

__global__ void __launch_bounds__(64)
k_inflate(int nstreams, const u8* __restrict__ in, const StreamDesc* __restrict__ desc, u8* __restrict__ out, int* __restrict__ status) {
  __shared__ Lds L;
  const int s = blockIdx.x;
  if (s >= nstreams) return;
  const int lane = threadIdx.x;
  const StreamDesc d = desc[s];
  const u8* src = in + d.in_off;
  u8* dst = out + d.out_off;
  int err = INF_OK;
  if (d.in_bytes >= 0xfff00000ull || d.out_bytes >= 0xfff00000ull) err = INF_INPUT;    // positions are 32-bit
  const u32 out_bytes = (u32)d.out_bytes, in_bytes = (u32)d.in_bytes;
  u32 opos = 0;

  Bits b;
  b.buf = 0; b.cnt = 0; b.ipos = 0;
  b.limit = ((in_bytes + 511u) & ~511u) + 512u;        // < in_bytes + IN_SLACK: every fetch stays inside the stream's slack
  fetch_half(L, src, 0, lane);
  refill(L, b, src, lane);
  // RFC 1950: CMF (method 8, window <= 32 KB), FLG (check bits, no preset dictionary)
  {
    const u32 cmf = take(b, 8), flg = take(b, 8);
    if ((cmf & 15u) != 8u || (cmf >> 4) > 7u || ((cmf << 8) | flg) % 31u != 0u || (flg & 32u)) err = INF_BAD_HEADER;
  }
  bool last = false;
  while (!err && !last) {
    refill(L, b, src, lane);
    last = take(b, 1) != 0;
    const u32 type = take(b, 2);
    if (type == 0) {
      save_state(L, b, opos, lane);
      err = (int)uni((u32)stored_block(L, src, in_bytes, dst, out_bytes, lane));
      load_state(L, b, opos);
      continue;
    }
    if (type == 3) { err = INF_BAD_BLOCK; break; }
    if (type == 1) fixed_tables(L, lane);
    else {
      save_state(L, b, opos, lane);
      err = (int)uni((u32)read_dynamic_header(L, src, lane));
      load_state(L, b, opos);
    }
    if (err) break;
    // ---- the block's symbols ----
#pragma unroll 1
    for (;;) {
      refill(L, b, src, lane);
      int sym = decode(L.lit, b);
      if (sym < 256) {
        if (sym < 0) { err = INF_BAD_CODE; break; }
        if (opos >= out_bytes) { err = INF_OVERRUN; break; }
        if (lane == 0) {
          dst[opos] = (u8)sym;
          L.window[opos & (WINDOW - 1)] = (u8)sym;
        }
        ++opos;
        continue;
      }
      if (sym == 256) break;
      sym -= 257;
      if (sym >= 29) { err = INF_BAD_CODE; break; }
      const u32 len = (u32)c_len_base[sym] + take(b, c_len_extra[sym]);
      refill(L, b, src, lane);
      const int dsym = decode(L.dist, b);
      if (dsym < 0 || dsym >= 30) { err = INF_BAD_CODE; break; }
      const u32 dist = (u32)c_dist_base[dsym] + take(b, c_dist_extra[dsym]);
      if (dist > opos) { err = INF_BAD_DISTANCE; break; }
      if ((u64)opos + len > out_bytes) { err = INF_OVERRUN; break; }
      __builtin_amdgcn_wave_barrier();            // the literals before this match are in the window
      // byte i of the match is the byte at pos - dist + (i mod dist): all sources lie before pos
      if (dist >= len) {
        for (u32 i = (u32)lane; i < len; i += 64) {
          const u8 v = L.window[(opos - dist + i) & (WINDOW - 1)];
          L.window[(opos + i) & (WINDOW - 1)] = v;
          dst[opos + i] = v;
        }
      } else {
        u8 v[5];                                  // len <= 258: at most five bytes per lane; read all before writing any
        int k = 0;
        for (u32 i = (u32)lane; i < len; i += 64, ++k) v[k] = L.window[(opos - dist + i % dist) & (WINDOW - 1)];
        __builtin_amdgcn_wave_barrier();
        k = 0;
        for (u32 i = (u32)lane; i < len; i += 64, ++k) {
          L.window[(opos + i) & (WINDOW - 1)] = v[k];
          dst[opos + i] = v[k];
        }
      }
      __builtin_amdgcn_wave_barrier();
      opos += len;
    }
  }
  if (!err && opos != out_bytes) err = INF_SHORT;
  if (!err && b.ipos - b.cnt / 8 > in_bytes) err = INF_INPUT;      // the decoder ran into the padding behind the stream
  // RFC 1950: the stream ends with the Adler-32 of the inflated bytes, big-endian, directly behind the last block (the bit
  // reader is byte-aligned there); nothing may follow it - what zlib, the HDF5 filter and the host decoder check
  if (!err) {
    const u32 at = b.ipos - b.cnt / 8;             // first byte not consumed by the blocks
    if (at + 4u != in_bytes) err = INF_INPUT;
    else {
      __builtin_amdgcn_wave_barrier();
      // s1 = 1 + sum b_i, s2 = n + sum (n - i) b_i (mod 65521): every lane takes a contiguous slice, the slices' sums are
      // combined over the wave; 64-bit partial sums stay far below 2^64 for slices of up to 2^26 bytes
      const u32 n = out_bytes;
      const u32 per = (n + 63u) / 64u;
      const u32 i0 = min(n, (u32)lane * per), i1 = min(n, i0 + per);
      u64 a = 0, w = 0;                            // sum of bytes, sum of (i1 - i) * byte over the slice
      for (u32 i = i0; i < i1; ++i) {
        const u64 v = dst[i];
        a += v;
        w += (u64)(i1 - i) * v;
        if (((i - i0) & 0xffffu) == 0xffffu) { a %= 65521u; w %= 65521u; }
      }
      a %= 65521u; w %= 65521u;
      // contribution of the slice to s2: bytes of the slice count (n - i) = (n - i1) + (i1 - i) times
      u64 s2 = (w + (u64)((n - i1) % 65521u) * a) % 65521u;
      u64 s1 = a;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        s1 += (u64)__shfl_down((unsigned long long)s1, off, 64);
        s2 += (u64)__shfl_down((unsigned long long)s2, off, 64);
      }
      s1 = (1u + s1) % 65521u;
      s2 = ((u64)(n % 65521u) + s2) % 65521u;
      const u32 want = ((u32)src[at] << 24) | ((u32)src[at + 1] << 16) | ((u32)src[at + 2] << 8) | (u32)src[at + 3];
      const u32 got = ((u32)s2 << 16) | (u32)s1;
      if (uni(lane == 0 ? (got != want ? 1u : 0u) : 0u)) err = INF_CHECKSUM;
    }
  }
  if (lane == 0) status[s] = err;
}

}  // namespace

namespace ecckd {

size_t inflate_in_slack() { return IN_SLACK; }
size_t inflate_desc_bytes() { return sizeof(StreamDesc); }

// streams described by d_desc (device array of {in_off, in_bytes, out_off, out_bytes}, in_off multiples of 16, IN_SLACK readable
// bytes behind every stream) -> d_out; d_status[s] = 0 or the reason stream s is not a valid zlib stream of out_bytes bytes
int inflate_launch(ecckd_ctx* ctx, hipStream_t stream, int nstreams, const void* d_in, const void* d_desc, void* d_out, int* d_status) {
  if (nstreams <= 0) return ECCKD_OK;
  hipLaunchKernelGGL(k_inflate, dim3((unsigned)nstreams), dim3(64), 0, stream, nstreams, (const u8*)d_in, (const StreamDesc*)d_desc,
                     (u8*)d_out, d_status);
  ECCKD_HIP_CHECK(hipGetLastError());
  (void)ctx;
  return ECCKD_OK;
}

}  // namespace ecckd

// Host-pointer convenience (tests, small inputs): nstreams zlib streams, stream s at h_in + h_in_off[s] .. h_in_off[s + 1],
// expected to inflate to exactly h_out_bytes[s] bytes, which are written one stream after the other to h_out.
extern "C" int ecckd_inflate(ecckd_ctx* ctx, int nstreams, const void* h_in, const unsigned long long* h_in_off,
                             const unsigned long long* h_out_bytes, void* h_out, int* h_status) {
  ECCKD_REQUIRE(ctx && nstreams >= 0 && (nstreams == 0 || (h_in && h_in_off && h_out_bytes && h_out && h_status)), "ecckd_inflate: bad argument");
  if (nstreams == 0) return ECCKD_OK;
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  std::vector<StreamDesc> desc((size_t)nstreams);
  u64 in_total = 0, out_total = 0;
  for (int s = 0; s < nstreams; ++s) {
    ECCKD_REQUIRE(h_in_off[s + 1] >= h_in_off[s], "ecckd_inflate: stream offsets must not decrease");
    desc[s].in_off = in_total;
    desc[s].in_bytes = h_in_off[s + 1] - h_in_off[s];
    desc[s].out_off = out_total;
    desc[s].out_bytes = h_out_bytes[s];
    in_total = (in_total + desc[s].in_bytes + IN_SLACK + 15) & ~(u64)15;
    out_total += h_out_bytes[s];
  }
  std::vector<u8> staged((size_t)in_total, 0);
  for (int s = 0; s < nstreams; ++s) std::memcpy(staged.data() + desc[s].in_off, (const u8*)h_in + h_in_off[s], (size_t)desc[s].in_bytes);
  void *d_in = nullptr, *d_desc = nullptr, *d_out = nullptr, *d_status = nullptr;
  int rc = ECCKD_OK;
  auto cleanup = [&]() {
    (void)hipStreamSynchronize(ctx->stream);
    if (d_in) (void)hipFree(d_in);
    if (d_desc) (void)hipFree(d_desc);
    if (d_out) (void)hipFree(d_out);
    if (d_status) (void)hipFree(d_status);
  };
#define TRY(x) do { rc = (x); if (rc != ECCKD_OK) { cleanup(); return rc; } } while (0)
  TRY(ecckd_dev_alloc(ctx, staged.size(), &d_in));
  TRY(ecckd_dev_alloc(ctx, desc.size() * sizeof(StreamDesc), &d_desc));
  TRY(ecckd_dev_alloc(ctx, out_total ? out_total : 16, &d_out));
  TRY(ecckd_dev_alloc(ctx, (size_t)nstreams * sizeof(int), &d_status));
  TRY(ecckd_h2d(ctx, d_in, staged.data(), staged.size()));
  TRY(ecckd_h2d(ctx, d_desc, desc.data(), desc.size() * sizeof(StreamDesc)));
  TRY(ecckd::inflate_launch(ctx, ctx->stream, nstreams, d_in, d_desc, d_out, (int*)d_status));
  if (out_total) TRY(ecckd_d2h(ctx, h_out, d_out, out_total));
  TRY(ecckd_d2h(ctx, h_status, d_status, (size_t)nstreams * sizeof(int)));
#undef TRY
  cleanup();
  return ECCKD_OK;
}
