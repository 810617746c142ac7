// nc_hdf5.hpp - the NetCDF-4 (HDF5) read backend behind ecckd_nc_* (csrc/nc_hdf5.cpp); classic files: nc_classic.cpp
#pragma once
#include <cstddef>
#include <string>
#include <vector>

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
// The raw (still filtered) chunks of such a variable, for a consumer that inflates them itself (nc_stream.hip: on the device).
struct H5ChunkPlan {
  int nd = 0, shuffle = 0;
  size_t ts = 0;                       // bytes per stored value (4 or 8, little-endian IEEE)
  unsigned long long dims[8] = {}, cdims[8] = {}, lo[8] = {}, hi[8] = {};   // dataset shape, chunk shape, requested box [lo, hi)
  size_t nchunks = 0, chunk_elems = 0, total = 0;                            // chunks that meet the box, values per chunk, values in the box
};
struct H5ChunkReader;
int h5_chunks_open(H5File* h, const char* name, long long slice, size_t capacity, H5ChunkReader** out, H5ChunkPlan* plan);
int h5_chunks_next(H5ChunkReader* r, unsigned long long* off, size_t* bytes, int* unwritten);
int h5_chunks_read(H5ChunkReader* r, void* dst, int* deflated, int* shuffled);
// the address of the chunk in the file instead of its bytes (~0: unknown), for readers that pread it themselves
int h5_chunks_locate(H5ChunkReader* r, unsigned long long* addr, int* deflated, int* shuffled, bool advance = true);
bool h5_can_locate(H5File* h);
const char* h5_path(H5File* h);
void h5_chunks_close(H5ChunkReader* r);
bool h5_has_zlib(H5File* h);
bool h5_inflate_host(H5File* h, void* dst, size_t dst_len, const void* src, size_t src_len);
int h5_read_att_text(H5File* h, const char* var, const char* att, int* exists, char* out, size_t capacity);
int h5_read_att_double(H5File* h, const char* var, const char* att, int* nelems, double* out, size_t capacity);

// ---- writing (nc_hdf5_write.cpp): a NetCDF-4 file laid out through the HDF5 library and its high-level library ----
struct H5WDim { std::string name; unsigned long long len = 0; };
struct H5WAtt { std::string name; int nc_type = 0; std::string text; std::vector<double> values; };   // NC_CHAR: text; else values
struct H5WVar { std::string name; int nc_type = 0; std::vector<int> dimids; std::vector<H5WAtt> atts; bool deflate = false; };
struct H5Writer;
bool h5w_available(const char** why);
int h5w_create(const char* path, const std::vector<H5WDim>& dims, const std::vector<H5WVar>& vars, const std::vector<H5WAtt>& gatts,
               H5Writer** out);
int h5w_write(H5Writer* w, int varindex, long long slice /* < 0: the whole variable */, const double* data, size_t count);
int h5w_close(H5Writer* w);

}  // namespace ecckd
