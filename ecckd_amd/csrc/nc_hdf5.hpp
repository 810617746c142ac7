// nc_hdf5.hpp - the NetCDF-4 (HDF5) read backend behind ecckd_nc_* (csrc/nc_hdf5.cpp); classic files: nc_classic.cpp
#pragma once
#include <cstddef>

namespace ecckd {

struct H5File;
bool h5_is_hdf5(const unsigned char* magic8);
int h5_open(const char* path, H5File** out);
void h5_close(H5File* h);
int h5_inq_dim(H5File* h, const char* name, size_t* len);
int h5_inq_var(H5File* h, const char* name, int* exists, int* nc_type, int* ndims, size_t* shape, int shape_capacity);
int h5_read_double(H5File* h, const char* name, long long slice, double* out, size_t capacity);
// FLOAT / DOUBLE variable (one slice or all) as float (out_type 4) or double (8) with the chunks inflated by worker threads;
// *handled = false if the variable's layout is not one this path takes apart (the caller then uses h5_read_double)
int h5_read_real_parallel(H5File* h, const char* name, long long slice, int out_type, void* out, size_t capacity, bool* handled);
int h5_read_att_text(H5File* h, const char* var, const char* att, int* exists, char* out, size_t capacity);
int h5_read_att_double(H5File* h, const char* var, const char* att, int* nelems, double* out, size_t capacity);

}  // namespace ecckd
