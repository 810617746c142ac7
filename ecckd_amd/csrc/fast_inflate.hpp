// fast_inflate.hpp - host-side zlib-stream decoder of the NetCDF-4 read path's worker threads (fast_inflate.cpp)
#pragma once
#include <cstddef>

namespace ecckd {

// true: `src` is a zlib stream (RFC 1950) that inflates to exactly dst_len bytes with the right Adler-32, and dst holds them.
// false: not taken or a check failed (dst then holds rubbish): the caller falls back to zlib.  Thread-safe.
bool fast_inflate_zlib(void* dst, size_t dst_len, const void* src, size_t src_len);

}  // namespace ecckd
