// fast_inflate.cpp - a zlib-stream (RFC 1950 / 1951) decoder for the worker threads of the NetCDF-4 read path (nc_stream.hip).
//
// The chunks of a CKDMIP spectrum are shuffled FLOATs deflated at a low level; undoing that with the zlib that sits beside the
// HDF5 library runs at ~0.4 GB/s per thread and is what bounds the read once the raw chunks no longer come through one thread.
// This decoder does the same work with a 64-bit bit buffer that is refilled once per length/distance pair, two-level decode
// tables whose entries carry base value and extra-bit count, and matches copied eight bytes at a time.  It checks what zlib
// checks (header, block structure, distances, exact output length, Adler-32 of the output) and answers false for anything it
// does not take or that fails a check: the caller then gives the chunk to zlib, so a wrong answer would need a corrupted stream
// that is structurally valid AND has the right checksum.
#include "fast_inflate.hpp"

#include <cstdint>
#include <cstring>
#include <emmintrin.h>

namespace ecckd {
namespace {

constexpr int LIT_BITS = 10, DIST_BITS = 8, SUB_BITS_LIT = 15 - LIT_BITS, SUB_BITS_DIST = 15 - DIST_BITS;
enum : uint32_t { T_LITERAL = 0, T_LENGTH = 1, T_END = 2, T_SUB = 3, T_INVALID = 4 };
// entry: bits 0-3 code length (T_SUB: unused), 4-6 type, 8-12 extra bits, 16-31 value (literal, base, or subtable offset)
inline uint32_t entry(uint32_t type, uint32_t len, uint32_t extra, uint32_t value) { return len | (type << 4) | (extra << 8) | (value << 16); }
inline uint32_t e_len(uint32_t e) { return e & 15; }
inline uint32_t e_type(uint32_t e) { return (e >> 4) & 7; }
inline uint32_t e_extra(uint32_t e) { return (e >> 8) & 31; }
inline uint32_t e_value(uint32_t e) { return e >> 16; }

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Tables {
  uint32_t lit[(1 << LIT_BITS) + 288 * (1 << SUB_BITS_LIT)];
  uint32_t dist[(1 << DIST_BITS) + 32 * (1 << SUB_BITS_DIST)];
};

inline uint32_t reverse_bits(uint32_t code, int len) {
  uint32_t r = 0;
  for (int i = 0; i < len; ++i) { r = (r << 1) | (code & 1); code >>= 1; }
  return r;
}

// canonical Huffman code of `lens[0..n)` into a two-level table with `primary` index bits; `make` turns a symbol into an
// entry without its length.  False for an over-subscribed code and, as in zlib's inflate_table, for an incomplete one unless it
// consists of a single one-bit code (a lone distance code); the unused half of that one leaves T_INVALID entries.
template <typename Make>
bool build(uint32_t* table, int primary, int sub_bits, const uint8_t* lens, int n, Make make) {
  int count[16] = {};
  for (int i = 0; i < n; ++i) ++count[lens[i]];
  count[0] = 0;
  uint32_t next[16];
  uint32_t code = 0;
  long long left = 1;
  int longest = 0;
  for (int l = 1; l <= 15; ++l) {
    left = (left << 1) - count[l];
    if (left < 0) return false;
    if (count[l]) longest = l;
    code = (code + (uint32_t)count[l - 1]) << 1;
    next[l] = code;
  }
  if (left > 0 && longest != 1) return false;
  const uint32_t psize = 1u << primary;
  for (uint32_t i = 0; i < psize; ++i) table[i] = entry(T_INVALID, 0, 0, 0);
  uint32_t used = psize;                        // subtables are handed out behind the primary table
  for (int s = 0; s < n; ++s) {
    const int l = lens[s];
    if (!l) continue;
    const uint32_t rev = reverse_bits(next[l]++, l);
    const uint32_t e = make(s);
    if (e_type(e) == T_INVALID) { if (l <= primary) for (uint32_t i = rev; i < psize; i += 1u << l) table[i] = e; continue; }
    if (l <= primary) {
      const uint32_t full = e | (uint32_t)l;
      for (uint32_t i = rev; i < psize; i += 1u << l) table[i] = full;
    } else {
      const uint32_t pre = rev & (psize - 1);
      if (e_type(table[pre]) != T_SUB) {
        table[pre] = entry(T_SUB, 0, 0, used);
        for (uint32_t i = 0; i < (1u << sub_bits); ++i) table[used + i] = entry(T_INVALID, 0, 0, 0);
        used += 1u << sub_bits;
      }
      uint32_t* sub = table + e_value(table[pre]);
      const int rest = l - primary;
      const uint32_t full = e | (uint32_t)l;
      for (uint32_t i = rev >> primary; i < (1u << sub_bits); i += 1u << rest) sub[i] = full;
    }
  }
  return true;
}

inline uint32_t make_lit(int s) {
  if (s < 256) return entry(T_LITERAL, 0, 0, (uint32_t)s);
  if (s == 256) return entry(T_END, 0, 0, 0);
  if (s < 286) return entry(T_LENGTH, 0, kLenExtra[s - 257], kLenBase[s - 257]);
  return entry(T_INVALID, 0, 0, 0);
}
inline uint32_t make_dist(int s) {
  if (s < 30) return entry(T_LENGTH, 0, kDistExtra[s], kDistBase[s]);
  return entry(T_INVALID, 0, 0, 0);
}

inline uint64_t load64(const uint8_t* p) { uint64_t v; std::memcpy(&v, p, 8); return v; }

struct Reader {
  const uint8_t* in;
  const uint8_t* end;
  uint64_t buf = 0;
  int cnt = 0;              // valid bits in buf
  bool overrun = false;     // bits were asked for behind the end of the input
  // at least 56 valid bits, or everything there is
  inline void refill() {
    if (end - in >= 8) {
      buf |= load64(in) << cnt;
      in += (63 - cnt) >> 3;
      cnt |= 56;
    } else {
      while (cnt <= 56 && in < end) { buf |= (uint64_t)*in++ << cnt; cnt += 8; }
    }
  }
  inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
  inline void drop(int n) { if (n > cnt) { overrun = true; cnt = 0; buf = 0; } else { buf >>= n; cnt -= n; } }
  inline uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
};

uint32_t adler32_sse2(const uint8_t* p, size_t n) {
  uint32_t s1 = 1, s2 = 0;
  const __m128i zero = _mm_setzero_si128();
  const __m128i w_lo = _mm_set_epi16(9, 10, 11, 12, 13, 14, 15, 16), w_hi = _mm_set_epi16(1, 2, 3, 4, 5, 6, 7, 8);
  while (n >= 16) {
    size_t blocks = n / 16;
    if (blocks > 5552 / 16) blocks = 5552 / 16;       // the sums stay below 2^32 for 5552 bytes
    n -= blocks * 16;
    __m128i v_s1 = zero, v_s2 = zero, v_ps = zero;     // v_ps: sum of the s1 values before each block
    for (size_t b = 0; b < blocks; ++b, p += 16) {
      const __m128i x = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p));
      v_ps = _mm_add_epi32(v_ps, v_s1);
      v_s1 = _mm_add_epi32(v_s1, _mm_sad_epu8(x, zero));
      const __m128i lo = _mm_unpacklo_epi8(x, zero), hi = _mm_unpackhi_epi8(x, zero);
      v_s2 = _mm_add_epi32(v_s2, _mm_add_epi32(_mm_madd_epi16(lo, w_lo), _mm_madd_epi16(hi, w_hi)));
    }
    uint32_t t[4];
    _mm_storeu_si128(reinterpret_cast<__m128i*>(t), v_s1);
    const uint32_t bytes_sum = t[0] + t[2];
    _mm_storeu_si128(reinterpret_cast<__m128i*>(t), v_ps);
    const uint64_t prefix = (uint64_t)t[0] + t[2];
    _mm_storeu_si128(reinterpret_cast<__m128i*>(t), v_s2);
    const uint64_t weighted = (uint64_t)t[0] + t[1] + t[2] + t[3];
    // s2 after the blocks: s2 + 16 * (blocks * s1 + sum of the partial byte sums before each block) + the weighted bytes
    s2 = (uint32_t)((s2 + 16ull * ((uint64_t)blocks * s1 + prefix) + weighted) % 65521);
    s1 = (uint32_t)(((uint64_t)s1 + bytes_sum) % 65521);
  }
  for (; n; --n) { s1 += *p++; s2 += s1; }
  return ((s2 % 65521) << 16) | (s1 % 65521);
}

}  // namespace

bool fast_inflate_zlib(void* dst_v, size_t dst_len, const void* src_v, size_t src_len) {
  uint8_t* const dst = static_cast<uint8_t*>(dst_v);
  const uint8_t* const src = static_cast<const uint8_t*>(src_v);
  if (src_len < 6) return false;
  // RFC 1950: method 8, window <= 32K, no preset dictionary, header check
  if ((src[0] & 15) != 8 || (src[0] >> 4) > 7 || (src[1] & 32) || ((src[0] << 8) | src[1]) % 31) return false;
  Reader r{src + 2, src + src_len - 4};
  uint8_t* out = dst;
  uint8_t* const out_end = dst + dst_len;
  static thread_local Tables T;
  bool last = false;
  while (!last) {
    r.refill();
    last = r.take(1) != 0;
    const uint32_t type = r.take(2);
    if (type == 3) return false;
    if (type == 0) {
      // stored: to the byte boundary, LEN, ~LEN, the bytes
      r.drop(r.cnt & 7);
      r.refill();
      if (r.cnt < 32) return false;
      const uint32_t len = r.take(16), nlen = r.take(16);
      if ((len ^ nlen) != 0xffff) return false;
      // hand back the whole bytes still in the bit buffer
      const uint8_t* p = r.in - (r.cnt >> 3);
      if ((r.cnt & 7) != 0 || (size_t)(r.end - p) < len || (size_t)(out_end - out) < len) return false;
      std::memcpy(out, p, len);
      out += len;
      r.in = p + len;
      r.buf = 0;
      r.cnt = 0;
      continue;
    }
    if (type == 1) {
      uint8_t lens[288 + 32];
      int i = 0;
      for (; i < 144; ++i) lens[i] = 8;
      for (; i < 256; ++i) lens[i] = 9;
      for (; i < 280; ++i) lens[i] = 7;
      for (; i < 288; ++i) lens[i] = 8;
      for (i = 0; i < 32; ++i) lens[288 + i] = 5;
      if (!build(T.lit, LIT_BITS, SUB_BITS_LIT, lens, 288, make_lit) || !build(T.dist, DIST_BITS, SUB_BITS_DIST, lens + 288, 32, make_dist))
        return false;
    } else {
      const int hlit = (int)r.take(5) + 257, hdist = (int)r.take(5) + 1, hclen = (int)r.take(4) + 4;
      if (hlit > 286 || hdist > 30) return false;
      uint8_t cl[19] = {};
      for (int i = 0; i < hclen; ++i) { r.refill(); cl[kClOrder[i]] = (uint8_t)r.take(3); }
      uint32_t cltab[1 << 7];
      {
        // one-level table of the code-length code (at most 7 bits)
        int count[8] = {};
        for (int i = 0; i < 19; ++i) ++count[cl[i]];
        count[0] = 0;
        uint32_t next[8], code = 0;
        long long left = 1;
        for (int l = 1; l <= 7; ++l) {
          left = (left << 1) - count[l];
          if (left < 0) return false;
          code = (code + (uint32_t)count[l - 1]) << 1;
          next[l] = code;
        }
        if (left > 0) return false;            // the code-length code must be complete (zlib: "invalid code lengths set")
        for (uint32_t& e : cltab) e = entry(T_INVALID, 0, 0, 0);
        for (int s = 0; s < 19; ++s) {
          const int l = cl[s];
          if (!l) continue;
          const uint32_t rev = reverse_bits(next[l]++, l);
          for (uint32_t i = rev; i < 128; i += 1u << l) cltab[i] = entry(T_LITERAL, (uint32_t)l, 0, (uint32_t)s);
        }
      }
      uint8_t lens[286 + 30 + 138];
      int have = 0;
      const int want = hlit + hdist;
      while (have < want) {
        r.refill();
        const uint32_t e = cltab[r.peek(7)];
        if (e_type(e) == T_INVALID) return false;
        r.drop((int)e_len(e));
        const uint32_t s = e_value(e);
        if (s < 16) { lens[have++] = (uint8_t)s; continue; }
        int rep;
        uint8_t v = 0;
        if (s == 16) { if (!have) return false; v = lens[have - 1]; rep = 3 + (int)r.take(2); }
        else if (s == 17) rep = 3 + (int)r.take(3);
        else rep = 11 + (int)r.take(7);
        if (have + rep > want) return false;
        std::memset(lens + have, v, (size_t)rep);
        have += rep;
      }
      if (r.overrun || lens[256] == 0) return false;
      if (!build(T.lit, LIT_BITS, SUB_BITS_LIT, lens, hlit, make_lit) || !build(T.dist, DIST_BITS, SUB_BITS_DIST, lens + hlit, hdist, make_dist))
        return false;
    }
    // ---- the symbols of the block ----
    for (;;) {
      // Fast iterations while eight bytes of input and 300 of output room are there: after the refill 56 bits are valid, a
      // length/distance pair takes at most 48 and literals are taken while 15 remain, so no request can run dry, and a
      // match (258 bytes at most) with its eight-byte overshoot fits.  The state lives in locals for the compiler's sake.
      if (r.end - r.in >= 8 && out_end - out >= 300) {
        uint64_t buf = r.buf;
        int cnt = r.cnt;
        const uint8_t* in = r.in;
        bool done = false, bad = false;
        do {
          buf |= load64(in) << cnt;
          in += (63 - cnt) >> 3;
          cnt |= 56;
          uint32_t e = T.lit[buf & ((1u << LIT_BITS) - 1)];
          if (e_type(e) == T_LITERAL) {
            for (;;) {
              buf >>= e_len(e);
              cnt -= (int)e_len(e);
              *out++ = (uint8_t)e_value(e);
              if (cnt < 15) break;
              e = T.lit[buf & ((1u << LIT_BITS) - 1)];
              if (e_type(e) != T_LITERAL) break;
            }
            continue;
          }
          if (e_type(e) == T_SUB) e = T.lit[e_value(e) + ((buf >> LIT_BITS) & ((1u << SUB_BITS_LIT) - 1))];
          buf >>= e_len(e);
          cnt -= (int)e_len(e);
          if (e_type(e) == T_LITERAL) { *out++ = (uint8_t)e_value(e); continue; }
          if (e_type(e) == T_END) { done = true; break; }
          if (e_type(e) != T_LENGTH) { bad = true; break; }
          const uint32_t len = e_value(e) + (uint32_t)(buf & ((1u << e_extra(e)) - 1));
          buf >>= e_extra(e);
          cnt -= (int)e_extra(e);
          uint32_t d = T.dist[buf & ((1u << DIST_BITS) - 1)];
          if (e_type(d) == T_SUB) d = T.dist[e_value(d) + ((buf >> DIST_BITS) & ((1u << SUB_BITS_DIST) - 1))];
          if (e_type(d) != T_LENGTH) { bad = true; break; }
          buf >>= e_len(d);
          cnt -= (int)e_len(d);
          const uint32_t dist = e_value(d) + (uint32_t)(buf & ((1u << e_extra(d)) - 1));
          buf >>= e_extra(d);
          cnt -= (int)e_extra(d);
          if (dist > (size_t)(out - dst)) { bad = true; break; }
          const uint8_t* from = out - dist;
          uint8_t* o = out;
          out += len;
          if (dist >= 8) {
            do { std::memcpy(o, from, 8); o += 8; from += 8; } while (o < out);
          } else if (dist == 1) {
            std::memset(o, *from, len);
          } else {
            do { *o++ = *from++; } while (o < out);
          }
        } while (r.end - in >= 8 && out_end - out >= 300);
        r.buf = buf;
        r.cnt = cnt;
        r.in = in;
        if (bad) return false;
        if (done) break;
        continue;
      }
      r.refill();
      uint32_t e = T.lit[r.peek(LIT_BITS)];
      if (e_type(e) == T_SUB) e = T.lit[e_value(e) + ((r.buf >> LIT_BITS) & ((1u << SUB_BITS_LIT) - 1))];
      if (e_type(e) == T_LITERAL) {
        if (out >= out_end) return false;
        r.drop((int)e_len(e));
        *out++ = (uint8_t)e_value(e);
        // a second and a third literal from the same refill (15 bits each at most, 56 were there)
        e = T.lit[r.peek(LIT_BITS)];
        if (e_type(e) != T_LITERAL || out >= out_end) continue;
        r.drop((int)e_len(e));
        *out++ = (uint8_t)e_value(e);
        e = T.lit[r.peek(LIT_BITS)];
        if (e_type(e) != T_LITERAL || out >= out_end) continue;
        r.drop((int)e_len(e));
        *out++ = (uint8_t)e_value(e);
        continue;
      }
      if (e_type(e) == T_END) { r.drop((int)e_len(e)); break; }
      if (e_type(e) != T_LENGTH) return false;
      r.drop((int)e_len(e));
      const uint32_t len = e_value(e) + r.take((int)e_extra(e));
      uint32_t d = T.dist[r.peek(DIST_BITS)];
      if (e_type(d) == T_SUB) d = T.dist[e_value(d) + ((r.buf >> DIST_BITS) & ((1u << SUB_BITS_DIST) - 1))];
      if (e_type(d) != T_LENGTH) return false;
      r.drop((int)e_len(d));
      const uint32_t dist = e_value(d) + r.take((int)e_extra(d));
      if (r.overrun || dist > (size_t)(out - dst) || len > (size_t)(out_end - out)) return false;
      const uint8_t* from = out - dist;
      if (dist >= 8 && (size_t)(out_end - out) >= len + 8) {
        // eight bytes at a time, up to seven past the match (room checked)
        uint8_t* o = out;
        uint8_t* const stop = out + len;
        do { std::memcpy(o, from, 8); o += 8; from += 8; } while (o < stop);
      } else if (dist == 1) {
        std::memset(out, *from, len);
      } else {
        for (uint32_t i = 0; i < len; ++i) out[i] = from[i];
      }
      out += len;
    }
    if (r.overrun) return false;
  }
  // everything written, nothing left but the checksum (whole bytes of lookahead are handed back first)
  if (out != out_end) return false;
  const uint8_t* p = r.in - (r.cnt >> 3);
  if (p != r.end) return false;
  const uint32_t want = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
  return adler32_sse2(dst, dst_len) == want;
}

}  // namespace ecckd

extern "C" int ecckd_inflate_host(const void* in, size_t in_bytes, void* out, size_t out_bytes, int* ok) {
  if (!ok || (in_bytes && !in) || (out_bytes && !out)) return 1 /* ECCKD_PARAMETER_ERROR */;
  *ok = ecckd::fast_inflate_zlib(out, out_bytes, in, in_bytes) ? 1 : 0;
  return 0;
}
