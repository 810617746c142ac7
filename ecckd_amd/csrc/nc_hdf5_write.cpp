// nc_hdf5_write.cpp - the NetCDF-4 (HDF5) WRITE backend behind ecckd_nc_create / ecckd_nc_def_* / ecckd_nc_write_*:
// what the reference's OutputDataFile produces for files named *.h5 / *.hdf (src/tools/OutputDataFile.cpp:84-157:
// nc_create(NC_NETCDF4)), including deflate_variable = shuffle + deflate level 2 (:345-359), which write_order applies to
// `wavenumber` and find_g_points to the per-wavenumber g-point variables.
//
// The image has no NetCDF library; the file is laid out by hand as the NetCDF-4 format specification describes it, through
// the system's HDF5 library and its high-level library (dimension scales), both loaded at run time like the read side
// (nc_hdf5.cpp): root-group datasets; every dimension a DIMENSION SCALE - the coordinate variable of that name where there
// is one, otherwise a data-less dataset whose NAME attribute reads "This is a netCDF dimension but not a netCDF variable."
// followed by the length -; every variable attached to the scales of its dimensions; `_Netcdf4Dimid` on every scale;
// link / attribute creation order tracked (the order variables and attributes are listed in); NC_CHAR attributes as
// fixed-length null-terminated scalar strings, numeric attributes as 1-D arrays; little-endian IEEE / two's-complement
// file types; contiguous layout unless a variable is deflated (then chunks of at most 2^18 values along the last dimension).
// A deflated variable's chunks are converted, shuffled and deflated by worker threads (the zlib found at run time, the call
// the library's own filter makes: compress2 at level 2) while the caller goes on to the next variable, and handed to the library
// as finished chunks when the file is closed (H5Dwrite_chunk, HDF5 >= 1.10.3): the library's filter pipeline is one thread at
// ~40 MB/s - 4.6 s for the six 7.2e6-value variables of an ordering
// file, 0.19 s for the same file in the classic format.  Without H5Dwrite_chunk or zlib, or with ECCKD_H5_SERIAL_WRITE=1, the
// library's pipeline does it.
// Unpinned: there is no NetCDF library here to read the result back with; the tests read it through the HDF5 library
// (structure, filters, attributes) and through this repository's own reader.
#include "common.hpp"
#include "nc_hdf5.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <exception>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace ecckd {

namespace {
typedef long long hid_t;
typedef int herr_t;
typedef unsigned long long hsize_t;

struct WApi {
  void* lib = nullptr;
  void* hl = nullptr;
  herr_t (*H5open)();
  herr_t (*H5Eset_auto2)(hid_t, void*, void*);
  hid_t (*H5Fcreate)(const char*, unsigned, hid_t, hid_t);
  herr_t (*H5Fclose)(hid_t);
  hid_t (*H5Pcreate)(hid_t);
  herr_t (*H5Pclose)(hid_t);
  herr_t (*H5Pset_chunk)(hid_t, int, const hsize_t*);
  herr_t (*H5Pset_shuffle)(hid_t);
  herr_t (*H5Pset_deflate)(hid_t, unsigned);
  herr_t (*H5Pset_link_creation_order)(hid_t, unsigned);
  herr_t (*H5Pset_attr_creation_order)(hid_t, unsigned);
  hid_t (*H5Screate)(int);
  hid_t (*H5Screate_simple)(int, const hsize_t*, const hsize_t*);
  herr_t (*H5Sselect_hyperslab)(hid_t, int, const hsize_t*, const hsize_t*, const hsize_t*, const hsize_t*);
  herr_t (*H5Sclose)(hid_t);
  hid_t (*H5Dcreate2)(hid_t, const char*, hid_t, hid_t, hid_t, hid_t, hid_t);
  herr_t (*H5Dwrite)(hid_t, hid_t, hid_t, hid_t, hid_t, const void*);
  herr_t (*H5Dclose)(hid_t);
  hid_t (*H5Tcopy)(hid_t);
  herr_t (*H5Tset_size)(hid_t, size_t);
  herr_t (*H5Tset_strpad)(hid_t, int);
  herr_t (*H5Tclose)(hid_t);
  hid_t (*H5Acreate2)(hid_t, const char*, hid_t, hid_t, hid_t, hid_t);
  herr_t (*H5Awrite)(hid_t, hid_t, const void*);
  herr_t (*H5Aclose)(hid_t);
  int (*H5Zfilter_avail)(int);
  herr_t (*H5Dwrite_chunk)(hid_t, hid_t, unsigned, const hsize_t*, size_t, const void*) = nullptr;   // 1.10.3 on
  int (*z_compress2)(unsigned char*, unsigned long*, const unsigned char*, unsigned long, int) = nullptr;
  unsigned long (*z_compressBound)(unsigned long) = nullptr;
  herr_t (*H5DSset_scale)(hid_t, const char*);
  herr_t (*H5DSattach_scale)(hid_t, hid_t, unsigned);
  hid_t p_file_create, p_dataset_create;
  hid_t t_f32le, t_f64le, t_i32le, t_i16le, t_i8le, t_u8le, t_f32be, t_native_double, t_native_int, t_c_s1;
  bool ok = false;
  std::string why;
};

WApi& wapi() {
  static WApi a;
  static std::once_flag once;
  std::call_once(once, [] {
    std::vector<std::string> names;
    if (const char* e = std::getenv("ECCKD_HDF5_LIB")) names.push_back(e);
    for (const char* n : {"libhdf5.so", "libhdf5_serial.so", "libhdf5.so.103", "libhdf5.so.200", "/opt/conda/lib/libhdf5.so"}) names.push_back(n);
    std::string used;
    for (const std::string& n : names) {
      a.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL);     // GLOBAL: the high-level library resolves against this one
      if (a.lib) { used = n; break; }
    }
    if (!a.lib) { a.why = "the HDF5 shared library could not be loaded (set ECCKD_HDF5_LIB)"; return; }
    std::vector<std::string> hl;
    if (const char* e = std::getenv("ECCKD_HDF5_HL_LIB")) hl.push_back(e);
    const size_t slash = used.find_last_of('/');
    if (slash != std::string::npos) hl.push_back(used.substr(0, slash + 1) + "libhdf5_hl.so");
    for (const char* n : {"libhdf5_hl.so", "libhdf5_serial_hl.so", "libhdf5_hl.so.100", "libhdf5_hl.so.200", "/opt/conda/lib/libhdf5_hl.so"}) hl.push_back(n);
    for (const std::string& n : hl) {
      a.hl = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (a.hl) break;
    }
    if (!a.hl) { a.why = "the HDF5 high-level library (dimension scales) could not be loaded (set ECCKD_HDF5_HL_LIB)"; return; }
    bool good = true;
#define LOADW(f) do { void* p = dlsym(a.lib, #f); if (!p) { good = false; a.why = std::string("HDF5 symbol missing: ") + #f; } \
                      a.f = reinterpret_cast<decltype(a.f)>(p); } while (0)
    LOADW(H5open); LOADW(H5Eset_auto2); LOADW(H5Fcreate); LOADW(H5Fclose); LOADW(H5Pcreate); LOADW(H5Pclose); LOADW(H5Pset_chunk);
    LOADW(H5Pset_shuffle); LOADW(H5Pset_deflate); LOADW(H5Pset_link_creation_order); LOADW(H5Pset_attr_creation_order);
    LOADW(H5Screate); LOADW(H5Screate_simple); LOADW(H5Sselect_hyperslab); LOADW(H5Sclose); LOADW(H5Dcreate2); LOADW(H5Dwrite);
    LOADW(H5Dclose); LOADW(H5Tcopy); LOADW(H5Tset_size); LOADW(H5Tset_strpad); LOADW(H5Tclose); LOADW(H5Acreate2); LOADW(H5Awrite);
    LOADW(H5Aclose); LOADW(H5Zfilter_avail);
#undef LOADW
    a.H5DSset_scale = reinterpret_cast<decltype(a.H5DSset_scale)>(dlsym(a.hl, "H5DSset_scale"));
    a.H5DSattach_scale = reinterpret_cast<decltype(a.H5DSattach_scale)>(dlsym(a.hl, "H5DSattach_scale"));
    if (!a.H5DSset_scale || !a.H5DSattach_scale) { good = false; a.why = "H5DSset_scale / H5DSattach_scale missing"; }
    if (!good) return;
    a.H5open();
    a.H5Eset_auto2(0, nullptr, nullptr);
    bool ids = true;
    auto id = [&](const char* name) -> hid_t {
      void* p = dlsym(a.lib, name);
      if (!p) { ids = false; a.why = std::string("HDF5 symbol missing: ") + name; return -1; }
      return *reinterpret_cast<hid_t*>(p);
    };
    a.p_file_create = id("H5P_CLS_FILE_CREATE_ID_g");
    a.p_dataset_create = id("H5P_CLS_DATASET_CREATE_ID_g");
    a.t_f32le = id("H5T_IEEE_F32LE_g"); a.t_f64le = id("H5T_IEEE_F64LE_g"); a.t_i32le = id("H5T_STD_I32LE_g");
    a.t_i16le = id("H5T_STD_I16LE_g"); a.t_i8le = id("H5T_STD_I8LE_g"); a.t_u8le = id("H5T_STD_U8LE_g");
    a.t_f32be = id("H5T_IEEE_F32BE_g"); a.t_native_double = id("H5T_NATIVE_DOUBLE_g"); a.t_native_int = id("H5T_NATIVE_INT_g");
    a.t_c_s1 = id("H5T_C_S1_g");
    if (!ids) return;
    if (a.H5Zfilter_avail(1) <= 0) { a.why = "this HDF5 library has no deflate filter"; return; }
    a.H5Dwrite_chunk = reinterpret_cast<decltype(a.H5Dwrite_chunk)>(dlsym(a.lib, "H5Dwrite_chunk"));
    for (const char* zn : {"libz.so.1", "libz.so", "/opt/conda/lib/libz.so.1"}) {
      if (void* z = dlopen(zn, RTLD_NOW | RTLD_LOCAL)) {
        a.z_compress2 = reinterpret_cast<decltype(a.z_compress2)>(dlsym(z, "compress2"));
        a.z_compressBound = reinterpret_cast<decltype(a.z_compressBound)>(dlsym(z, "compressBound"));
        if (a.z_compress2 && a.z_compressBound) break;
        a.z_compress2 = nullptr;
      }
    }
    a.ok = true;
  });
  return a;
}

enum { NC_BYTE = 1, NC_CHAR = 2, NC_SHORT = 3, NC_INT = 4, NC_FLOAT = 5, NC_DOUBLE = 6, NC_UBYTE = 7 };
constexpr hsize_t kChunkValues = (hsize_t)1 << 18;      // values per chunk of a deflated variable (along its last dimension)

size_t file_type_size(int nc_type) {
  switch (nc_type) {
    case NC_SHORT: return 2;
    case NC_INT: case NC_FLOAT: return 4;
    case NC_DOUBLE: return 8;
    default: return 1;
  }
}

// `n` doubles as little-endian file values of type t (x86: native order), the conversion the library makes on H5Dwrite:
// to float by rounding, to the integer types by truncation with the ends of the type's range as limits
template <typename T> inline T clamp_to(double v, double lo, double hi) { return (T)(v < lo ? lo : v > hi ? hi : v); }
void to_file_values(unsigned char* out, int t, const double* v, size_t n) {
  switch (t) {
    case NC_DOUBLE: std::memcpy(out, v, n * 8); return;
    case NC_FLOAT: { float* o = reinterpret_cast<float*>(out); for (size_t i = 0; i < n; ++i) o[i] = (float)v[i]; return; }
    case NC_INT: { int32_t* o = reinterpret_cast<int32_t*>(out); for (size_t i = 0; i < n; ++i) o[i] = clamp_to<int32_t>(v[i], -2147483648.0, 2147483647.0); return; }
    case NC_SHORT: { int16_t* o = reinterpret_cast<int16_t*>(out); for (size_t i = 0; i < n; ++i) o[i] = clamp_to<int16_t>(v[i], -32768.0, 32767.0); return; }
    case NC_BYTE: { int8_t* o = reinterpret_cast<int8_t*>(out); for (size_t i = 0; i < n; ++i) o[i] = clamp_to<int8_t>(v[i], -128.0, 127.0); return; }
    default: for (size_t i = 0; i < n; ++i) out[i] = clamp_to<uint8_t>(v[i], 0.0, 255.0);
  }
}

// the shuffle filter: byte b of every value, then byte b + 1 of every value ...
void shuffle_bytes(unsigned char* out, const unsigned char* in, size_t n, size_t ts) {
  if (ts == 1) { std::memcpy(out, in, n); return; }
  for (size_t b = 0; b < ts; ++b) {
    unsigned char* o = out + b * n;
    const unsigned char* q = in + b;
    for (size_t i = 0; i < n; ++i) o[i] = q[i * ts];
  }
}

hid_t file_type(const WApi& a, int nc_type) {
  switch (nc_type) {
    case NC_BYTE: return a.t_i8le;
    case NC_UBYTE: case NC_CHAR: return a.t_u8le;
    case NC_SHORT: return a.t_i16le;
    case NC_INT: return a.t_i32le;
    case NC_FLOAT: return a.t_f32le;
    case NC_DOUBLE: return a.t_f64le;
    default: return -1;
  }
}

int put_att(WApi& a, hid_t loc, const H5WAtt& att, const std::string& where) {
  hid_t space, type, id;
  if (att.nc_type == NC_CHAR) {
    type = a.H5Tcopy(a.t_c_s1);
    a.H5Tset_size(type, att.text.size() > 0 ? att.text.size() : 1);
    a.H5Tset_strpad(type, 0 /* H5T_STR_NULLTERM */);
    space = a.H5Screate(0 /* H5S_SCALAR */);
    id = a.H5Acreate2(loc, att.name.c_str(), type, space, 0, 0);
    const std::string text = att.text.empty() ? std::string(1, '\0') : att.text;
    const bool bad = id < 0 || a.H5Awrite(id, type, text.data()) < 0;
    if (id >= 0) a.H5Aclose(id);
    a.H5Sclose(space);
    a.H5Tclose(type);
    if (bad) return fail(ECCKD_PROCESSING_ERROR, "%s: attribute \"%s\" could not be written", where.c_str(), att.name.c_str());
    return ECCKD_OK;
  }
  const hsize_t n = att.values.size();
  type = file_type(a, att.nc_type);
  if (type < 0) return fail(ECCKD_PARAMETER_ERROR, "%s: attribute \"%s\" has a type NetCDF-4 output does not write", where.c_str(), att.name.c_str());
  space = a.H5Screate_simple(1, &n, nullptr);
  id = a.H5Acreate2(loc, att.name.c_str(), type, space, 0, 0);
  const bool bad = id < 0 || a.H5Awrite(id, a.t_native_double, att.values.data()) < 0;
  if (id >= 0) a.H5Aclose(id);
  a.H5Sclose(space);
  if (bad) return fail(ECCKD_PROCESSING_ERROR, "%s: attribute \"%s\" could not be written", where.c_str(), att.name.c_str());
  return ECCKD_OK;
}
}  // namespace

struct H5Writer {
  std::string path;
  hid_t file = -1;
  std::vector<hid_t> var_ids;
  std::vector<hid_t> dimonly_ids;
  std::vector<H5WVar> vars;
  std::vector<H5WDim> dims;
  // deflated variables on their way through the worker threads (queue_deflated_chunks)
  struct Chunk {
    int var = 0;
    size_t type_size = 1;
    hsize_t offset[32] = {0};
    std::vector<unsigned char> raw, z;      // file values of the whole chunk; shuffled + deflated
    int state = 0;                          // 0 queued, 1 deflated, -1 failed
  };
  std::deque<Chunk> queue;                  // (a deque: the workers hold pointers to its elements while more are queued)
  size_t next_job = 0, finished = 0, queued_bytes = 0;
  bool stop = false;
  std::mutex m;
  std::condition_variable cv, cv_done;
  std::vector<std::thread> pool;
};

bool h5w_available(const char** why) {
  WApi& a = wapi();
  if (why) *why = a.why.c_str();
  return a.ok;
}

int h5w_create(const char* path, const std::vector<H5WDim>& dims, const std::vector<H5WVar>& vars, const std::vector<H5WAtt>& gatts,
               H5Writer** out) {
  WApi& a = wapi();
  if (!a.ok) return fail(ECCKD_PROCESSING_ERROR, "Cannot write NetCDF-4/HDF-5 file \"%s\": %s", path, a.why.c_str());
  const unsigned order = 1u | 2u;      // H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED
  const hid_t fcpl = a.H5Pcreate(a.p_file_create);
  a.H5Pset_link_creation_order(fcpl, order);
  a.H5Pset_attr_creation_order(fcpl, order);
  const hid_t file = a.H5Fcreate(path, 2u /* H5F_ACC_TRUNC */, fcpl, 0);
  a.H5Pclose(fcpl);
  if (file < 0) return fail(ECCKD_PARAMETER_ERROR, "cannot open %s for writing", path);
  H5Writer* w = new H5Writer;
  w->path = path;
  w->file = file;
  w->vars = vars;
  w->dims = dims;
  int rc = ECCKD_OK;
  // the coordinate variable of a dimension: the 1-D variable of its name over it
  std::vector<int> coord(dims.size(), -1);
  for (size_t v = 0; v < vars.size(); ++v)
    if (vars[v].dimids.size() == 1 && vars[v].name == dims[vars[v].dimids[0]].name) coord[vars[v].dimids[0]] = (int)v;
  // ---- the variables, in the order they were defined ----
  for (size_t v = 0; v < vars.size() && rc == ECCKD_OK; ++v) {
    const H5WVar& var = vars[v];
    const int nd = (int)var.dimids.size();
    hsize_t shape[32], chunk[32];
    for (int k = 0; k < nd; ++k) { shape[k] = dims[var.dimids[k]].len; chunk[k] = 1; }
    const hid_t space = nd == 0 ? a.H5Screate(0) : a.H5Screate_simple(nd, shape, nullptr);
    const hid_t dcpl = a.H5Pcreate(a.p_dataset_create);
    a.H5Pset_attr_creation_order(dcpl, order);
    if (var.deflate && nd > 0) {
      chunk[nd - 1] = shape[nd - 1] < kChunkValues ? shape[nd - 1] : kChunkValues;
      a.H5Pset_chunk(dcpl, nd, chunk);
      a.H5Pset_shuffle(dcpl);
      a.H5Pset_deflate(dcpl, 2);            // nc_def_var_deflate(ncid, varid, 1, 1, 2), OutputDataFile.cpp:356
    }
    const hid_t type = file_type(a, var.nc_type);
    // a variable that shares its name with a dimension it is not the coordinate variable of (g_point(wavenumber) beside the
    // dimension g_point): the plain name belongs to the dimension's scale, the variable is stored as _nc4_non_coord_<name>
    std::string stored = var.name;
    for (size_t k = 0; k < dims.size(); ++k)
      if (dims[k].name == var.name && coord[k] != (int)v) stored = "_nc4_non_coord_" + var.name;
    const hid_t d = type < 0 ? -1 : a.H5Dcreate2(file, stored.c_str(), type, space, 0, dcpl, 0);
    a.H5Pclose(dcpl);
    a.H5Sclose(space);
    w->var_ids.push_back(d);
    if (d < 0) { rc = fail(ECCKD_PROCESSING_ERROR, "%s: variable \"%s\" could not be created", path, var.name.c_str()); break; }
    for (const H5WAtt& att : var.atts) {
      rc = put_att(a, d, att, w->path + ": " + var.name);
      if (rc != ECCKD_OK) break;
    }
  }
  // ---- dimensions: coordinate variables become scales, the others get a data-less scale dataset ----
  std::vector<hid_t> scale(dims.size(), -1);
  for (size_t k = 0; k < dims.size() && rc == ECCKD_OK; ++k) {
    if (coord[k] >= 0) {
      scale[k] = w->var_ids[coord[k]];
      if (a.H5DSset_scale(scale[k], dims[k].name.c_str()) < 0) rc = fail(ECCKD_PROCESSING_ERROR, "%s: dimension scale \"%s\"", path, dims[k].name.c_str());
    } else {
      const hsize_t len = dims[k].len;
      const hid_t space = a.H5Screate_simple(1, &len, nullptr);
      const hid_t dcpl = a.H5Pcreate(a.p_dataset_create);
      a.H5Pset_attr_creation_order(dcpl, order);
      const hid_t d = a.H5Dcreate2(file, dims[k].name.c_str(), a.t_f32be, space, 0, dcpl, 0);
      a.H5Pclose(dcpl);
      a.H5Sclose(space);
      if (d < 0) { rc = fail(ECCKD_PROCESSING_ERROR, "%s: dimension \"%s\" could not be created", path, dims[k].name.c_str()); break; }
      w->dimonly_ids.push_back(d);
      scale[k] = d;
      char name[96];
      std::snprintf(name, sizeof name, "This is a netCDF dimension but not a netCDF variable.%10d", (int)len);
      if (a.H5DSset_scale(d, name) < 0) rc = fail(ECCKD_PROCESSING_ERROR, "%s: dimension scale \"%s\"", path, dims[k].name.c_str());
    }
    if (rc == ECCKD_OK) {
      H5WAtt id;
      id.name = "_Netcdf4Dimid";
      id.nc_type = NC_INT;
      id.values.assign(1, (double)k);
      // (a scalar in files the NetCDF library writes; a one-element array reads the same)
      const hid_t space = a.H5Screate(0);
      const hid_t at = a.H5Acreate2(scale[k], id.name.c_str(), a.t_i32le, space, 0, 0);
      const int v = (int)k;
      if (at < 0 || a.H5Awrite(at, a.t_native_int, &v) < 0) rc = fail(ECCKD_PROCESSING_ERROR, "%s: _Netcdf4Dimid of \"%s\"", path, dims[k].name.c_str());
      if (at >= 0) a.H5Aclose(at);
      a.H5Sclose(space);
    }
  }
  // ---- every variable attached to the scales of its dimensions (a coordinate variable is not attached to itself) ----
  for (size_t v = 0; v < vars.size() && rc == ECCKD_OK; ++v)
    for (size_t k = 0; k < vars[v].dimids.size() && rc == ECCKD_OK; ++k) {
      const int dimid = vars[v].dimids[k];
      if (coord[dimid] == (int)v) continue;
      if (a.H5DSattach_scale(w->var_ids[v], scale[dimid], (unsigned)k) < 0)
        rc = fail(ECCKD_PROCESSING_ERROR, "%s: \"%s\" could not be attached to dimension \"%s\"", path, vars[v].name.c_str(), dims[dimid].name.c_str());
    }
  for (const H5WAtt& att : gatts) {
    if (rc != ECCKD_OK) break;
    rc = put_att(a, file, att, w->path);
  }
  if (rc != ECCKD_OK) { h5w_close(w); return rc; }
  *out = w;
  return ECCKD_OK;
}

namespace {
// ---- deflated variables: chunks built by worker threads, written when the file is closed -----------------------------------
// h5w_write converts its values to the file type, cuts them into the dataset's chunks and queues those; the writer's worker
// threads shuffle and deflate them while the caller goes on (the next variable's chunks join the same queue: the six variables of
// an ordering file keep every core busy); drain() waits for the queue and hands the finished chunks to the library from the
// calling thread, in the order they were queued (the library is not thread-safe).
void pool_worker(H5Writer* w) {
  WApi& a = wapi();
  std::vector<unsigned char> shuffled;
  for (;;) {
    H5Writer::Chunk* c = nullptr;
    {
      std::unique_lock<std::mutex> lock(w->m);
      w->cv.wait(lock, [&] { return w->stop || w->next_job < w->queue.size(); });
      if (w->next_job >= w->queue.size()) return;      // stop, nothing left
      c = &w->queue[w->next_job++];
    }
    const size_t nbytes = c->raw.size();
    shuffled.resize(nbytes);
    shuffle_bytes(shuffled.data(), c->raw.data(), nbytes / c->type_size, c->type_size);
    unsigned long size = a.z_compressBound((unsigned long)nbytes);
    c->z.resize(size);
    const int zrc = a.z_compress2(c->z.data(), &size, shuffled.data(), (unsigned long)nbytes, 2);
    c->z.resize(zrc == 0 ? size : 0);
    std::vector<unsigned char>().swap(c->raw);
    {
      std::lock_guard<std::mutex> lock(w->m);
      c->state = zrc == 0 ? 1 : -1;
      ++w->finished;
    }
    w->cv_done.notify_all();
  }
}

int drain(H5Writer* w) {
  WApi& a = wapi();
  {
    std::unique_lock<std::mutex> lock(w->m);
    w->cv_done.wait(lock, [&] { return w->finished == w->queue.size(); });
  }
  int rc = ECCKD_OK;
  for (H5Writer::Chunk& c : w->queue) {
    if (rc != ECCKD_OK) break;
    const H5WVar& var = w->vars[c.var];
    if (c.state != 1) { rc = fail(ECCKD_PROCESSING_ERROR, "%s: deflate of \"%s\" failed", w->path.c_str(), var.name.c_str()); break; }
    if (a.H5Dwrite_chunk(w->var_ids[c.var], 0, 0u, c.offset, c.z.size(), c.z.data()) < 0)
      rc = fail(ECCKD_PROCESSING_ERROR, "%s: write of \"%s\" failed", w->path.c_str(), var.name.c_str());
  }
  {
    std::lock_guard<std::mutex> lock(w->m);
    w->queue.clear();
    w->next_job = 0;
    w->finished = 0;
    w->queued_bytes = 0;
  }
  return rc;
}

void stop_pool(H5Writer* w) {
  {
    std::lock_guard<std::mutex> lock(w->m);
    w->stop = true;
  }
  w->cv.notify_all();
  for (std::thread& t : w->pool) t.join();
  w->pool.clear();
}

// Queues the chunks of a deflated variable that the flat run data[0 .. count) covers - whole rows of the last dimension, from
// row `row0` of the variable on.  Returns 1 when the direct path does not apply (the caller then writes through the library's
// pipeline), ECCKD_OK or an error code otherwise.
int queue_deflated_chunks(WApi& a, H5Writer* w, int varindex, size_t row0, const double* data, size_t count) {
  const H5WVar& var = w->vars[varindex];
  const int nd = (int)var.dimids.size();
  if (!var.deflate || nd == 0 || !a.H5Dwrite_chunk || !a.z_compress2 || std::getenv("ECCKD_H5_SERIAL_WRITE")) return 1;
  const size_t last = w->dims[var.dimids[nd - 1]].len;
  if (last == 0 || count % last != 0) return 1;
  if (w->queued_bytes > ((size_t)1 << 30)) ECCKD_CHECK(drain(w));       // bounded memory for variables written slice after slice
  const size_t nrows = count / last;
  const size_t cv = (size_t)std::min<hsize_t>(last, kChunkValues);     // values per chunk (the dataset's chunk shape, h5w_create)
  const size_t per_row = (last + cv - 1) / cv;
  const size_t nchunks = nrows * per_row;
  const size_t ts = file_type_size(var.nc_type);
  std::vector<H5Writer::Chunk> fresh(nchunks);
  for (size_t c = 0; c < nchunks; ++c) {
    H5Writer::Chunk& ch = fresh[c];
    const size_t first = (c % per_row) * cv;
    const size_t n = std::min(cv, last - first);
    ch.var = varindex;
    ch.type_size = ts;
    ch.raw.assign(cv * ts, 0);                                         // an edge chunk is stored whole, zero behind its values
    to_file_values(ch.raw.data(), var.nc_type, data + (c / per_row) * last + first, n);
    size_t row = row0 + c / per_row;
    for (int k = nd - 2; k >= 0; --k) {
      const size_t len = w->dims[var.dimids[k]].len;
      ch.offset[k] = row % len;
      row /= len;
    }
    ch.offset[nd - 1] = (hsize_t)first;
  }
  if (w->pool.empty()) {
    // (a host that cannot start another thread: the library's own pipeline writes this variable)
    const int nthreads = std::max(1, std::min(16, host_cores()));
    try {
      for (int t = 0; t < nthreads; ++t) w->pool.emplace_back(pool_worker, w);
    } catch (const std::exception&) {
      if (w->pool.empty()) return 1;
    }
  }
  {
    std::lock_guard<std::mutex> lock(w->m);
    for (H5Writer::Chunk& ch : fresh) w->queue.push_back(std::move(ch));
    w->queued_bytes += nchunks * cv * ts;
  }
  w->cv.notify_all();
  return ECCKD_OK;
}
}  // namespace

int h5w_write(H5Writer* w, int varindex, long long slice, const double* data, size_t count) {
  WApi& a = wapi();
  const H5WVar& var = w->vars[varindex];
  const hid_t d = w->var_ids[varindex];
  const int nd = (int)var.dimids.size();
  herr_t e;
  if (var.deflate && nd > 0) {
    size_t rows_per_slice = 1;
    for (int k = 1; k + 1 < nd; ++k) rows_per_slice *= w->dims[var.dimids[k]].len;
    const int rc = queue_deflated_chunks(a, w, varindex, slice < 0 || nd == 1 ? 0 : (size_t)slice * rows_per_slice, data, count);
    if (rc != 1) return rc;
  }
  if (slice < 0 || nd == 0) {
    const hsize_t n = count;
    const hid_t mem = nd == 0 ? a.H5Screate(0) : a.H5Screate_simple(1, &n, nullptr);
    // memory: a flat run of doubles; file: the whole dataset (same number of elements)
    if (nd <= 1) {
      e = a.H5Dwrite(d, a.t_native_double, 0, 0, 0, data);
    } else {
      hsize_t shape[32];
      for (int k = 0; k < nd; ++k) shape[k] = w->dims[var.dimids[k]].len;
      const hid_t fsp = a.H5Screate_simple(nd, shape, nullptr);
      e = a.H5Dwrite(d, a.t_native_double, mem, fsp, 0, data);
      a.H5Sclose(fsp);
    }
    a.H5Sclose(mem);
  } else {
    hsize_t shape[32], start[32], cnt[32];
    for (int k = 0; k < nd; ++k) { shape[k] = w->dims[var.dimids[k]].len; start[k] = 0; cnt[k] = shape[k]; }
    start[0] = (hsize_t)slice;
    cnt[0] = 1;
    const hid_t fsp = a.H5Screate_simple(nd, shape, nullptr);
    a.H5Sselect_hyperslab(fsp, 0 /* H5S_SELECT_SET */, start, nullptr, cnt, nullptr);
    const hsize_t n = count;
    const hid_t mem = a.H5Screate_simple(1, &n, nullptr);
    e = a.H5Dwrite(d, a.t_native_double, mem, fsp, 0, data);
    a.H5Sclose(mem);
    a.H5Sclose(fsp);
  }
  if (e < 0) return fail(ECCKD_PROCESSING_ERROR, "%s: write of \"%s\" failed", w->path.c_str(), var.name.c_str());
  return ECCKD_OK;
}

int h5w_close(H5Writer* w) {
  if (!w) return ECCKD_OK;
  WApi& a = wapi();
  const int rc_chunks = drain(w);
  stop_pool(w);
  for (hid_t d : w->var_ids) if (d >= 0) a.H5Dclose(d);
  for (hid_t d : w->dimonly_ids) if (d >= 0) a.H5Dclose(d);
  const herr_t e = w->file >= 0 ? a.H5Fclose(w->file) : 0;
  const std::string path = w->path;
  delete w;
  if (rc_chunks != ECCKD_OK) return rc_chunks;
  if (e < 0) return fail(ECCKD_PROCESSING_ERROR, "%s: close failed", path.c_str());
  return ECCKD_OK;
}

}  // namespace ecckd
