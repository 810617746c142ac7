// ASan/UBSan harness for csrc/fast_inflate.cpp: valid streams must decode to the input; damaged / truncated / mis-sized ones must
// be refused or - if accepted - give exactly what zlib gives.  Buffers are heap blocks of the exact size, so any overrun trips ASan.
// build: g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -std=c++17 -Iecckd_amd/csrc tools/fuzz_fast_inflate.cpp \
//        ecckd_amd/csrc/fast_inflate.cpp -lz -o /tmp/fuzz_fast_inflate && ASAN_OPTIONS=detect_leaks=0 /tmp/fuzz_fast_inflate 6000
// (round 2: 6 000 valid streams decoded, 36 000 damaged ones: 35 959 refused, 41 accepted identically to zlib, no report)
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "fast_inflate.hpp"
using ecckd::fast_inflate_zlib;
int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
  std::mt19937_64 rng(12345);
  long ok_valid = 0, refused = 0, accepted_damaged = 0;
  for (int r = 0; r < rounds; ++r) {
    const size_t n = rng() % 70000;
    std::vector<unsigned char> data(n);
    const int kind = rng() % 5;
    for (size_t i = 0; i < n; ++i) {
      switch (kind) {
        case 0: data[i] = (unsigned char)rng(); break;
        case 1: data[i] = (unsigned char)(i % 7); break;
        case 2: data[i] = (unsigned char)((rng() % 100 < 90) ? 0 : rng()); break;
        case 3: data[i] = (unsigned char)(i > 300 && rng() % 4 ? data[i - 1 - rng() % 300] : rng()); break;
        default: data[i] = (unsigned char)(rng() % 3);
      }
    }
    const int level = rng() % 10, strategy = rng() % 5, wbits = 9 + rng() % 7;
    z_stream zs{};
    deflateInit2(&zs, level, Z_DEFLATED, wbits, 1 + rng() % 9, strategy);
    std::vector<unsigned char> comp(2 * n + 4096);
    zs.next_in = data.data(); zs.avail_in = (uInt)n; zs.next_out = comp.data(); zs.avail_out = (uInt)comp.size();
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { printf("harness: deflate did not finish\n"); return 2; }
    comp.resize(zs.total_out);
    deflateEnd(&zs);
    {
      unsigned char* src = (unsigned char*)malloc(comp.size() ? comp.size() : 1);
      memcpy(src, comp.data(), comp.size());
      unsigned char* dst = (unsigned char*)malloc(n ? n : 1);
      if (!fast_inflate_zlib(dst, n, src, comp.size()) || memcmp(dst, data.data(), n)) { printf("FAIL valid r=%d n=%zu level=%d strategy=%d wbits=%d\n", r, n, level, strategy, wbits); return 1; }
      ++ok_valid;
      free(src); free(dst);
    }
    for (int k = 0; k < 6; ++k) {
      std::vector<unsigned char> bad = comp;
      size_t out_n = n;
      const int what = rng() % 4;
      if (what == 0 && !bad.empty()) bad[rng() % bad.size()] ^= (unsigned char)(1u << (rng() % 8));
      else if (what == 1 && !bad.empty()) bad.resize(rng() % bad.size());
      else if (what == 2) out_n = n ? n - 1 - rng() % (n < 9 ? n : 9) % n : 1;
      else { for (int q = 0; q < 4 && !bad.empty(); ++q) bad[rng() % bad.size()] = (unsigned char)rng(); }
      unsigned char* src = (unsigned char*)malloc(bad.size() ? bad.size() : 1);
      memcpy(src, bad.data(), bad.size());
      unsigned char* dst = (unsigned char*)malloc(out_n ? out_n : 1);
      const bool ok = fast_inflate_zlib(dst, out_n, src, bad.size());
      if (ok) {
        std::vector<unsigned char> ref(out_n ? out_n : 1);
        uLongf len = (uLongf)out_n;
        const int zr = uncompress(ref.data(), &len, bad.data(), (uLong)bad.size());
        if (zr != Z_OK || len != out_n || memcmp(ref.data(), dst, out_n)) { printf("FAIL accepted a stream zlib does not decode the same way (r=%d what=%d)\n", r, what); return 1; }
        ++accepted_damaged;
      } else ++refused;
      free(src); free(dst);
    }
  }
  printf("valid streams decoded: %ld; damaged refused: %ld, damaged accepted identically to zlib: %ld\n", ok_valid, refused, accepted_damaged);
  return 0;
}
