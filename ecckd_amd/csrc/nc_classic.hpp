// nc_classic.hpp - what the streaming reader (nc_stream.hip) needs to know about an open classic file.
#pragma once
#include <cstdint>

struct ecckd_nc;

namespace ecckd {

struct NcSlice {
  int fd = -1;             // file descriptor of the open file
  uint64_t offset = 0;     // byte offset of the first element
  uint64_t count = 0;      // elements
  int nc_type = 0;         // external type (NC_FLOAT = 5, NC_DOUBLE = 6, ...)
  bool contiguous = false; // stored in one piece (a fixed-size variable of a classic file); false: HDF5 or a record variable
};

// Where one index of the slowest dimension (slice >= 0) or the whole variable (slice < 0) lies in the file.
int nc_locate_slice(ecckd_nc* f, const char* name, long long slice, NcSlice* out);

struct H5File;
// the HDF5 backend of a NetCDF-4 file opened for reading, or NULL for a classic file
H5File* nc_h5_handle(ecckd_nc* f);

}  // namespace ecckd
