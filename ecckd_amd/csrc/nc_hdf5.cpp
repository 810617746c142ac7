// nc_hdf5.cpp - reading NetCDF-4 files (HDF5 containers: the CKDMIP spectra, everything the reference's scripts
// name *.h5) through the system's HDF5 C library, loaded at run time.
//
// The image ships no NetCDF library and no HDF5 development package, but an HDF5 1.10 shared library is present
// (/opt/conda/lib).  It is dlopen'ed on first use - ECCKD_HDF5_LIB, then the usual sonames - so the product has no
// link-time dependency and classic files keep working where the library is missing.  Only what DataFile does with a
// NetCDF-4 file on the hot path is provided (src/tools/DataFileEngineNetcdf.cpp): existence and shape of a variable,
// whole-variable or one-slice reads converted to double (nc_get_vara_double, :593-599; the HDF5 filters - shuffle,
// deflate - run inside H5Dread), numeric and text attributes, dimension lengths.  Writing stays classic: the
// NetCDF library the reference links detects the format from the file's first bytes, not from its name, so a classic
// file called *.h5 is read back correctly by both sides.
#include <dlfcn.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.hpp"
#include "nc_hdf5.hpp"
#include "fast_inflate.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

namespace {

typedef int64_t hid_t;
typedef unsigned long long hsize_t;
typedef int herr_t;
typedef int htri_t;

struct Api {
  void* lib = nullptr;
  herr_t (*H5open)();
  herr_t (*H5Eset_auto2)(hid_t, void*, void*);
  hid_t (*H5Fopen)(const char*, unsigned, hid_t);
  herr_t (*H5Fclose)(hid_t);
  htri_t (*H5Lexists)(hid_t, const char*, hid_t);
  hid_t (*H5Dopen2)(hid_t, const char*, hid_t);
  herr_t (*H5Dclose)(hid_t);
  hid_t (*H5Dget_space)(hid_t);
  hid_t (*H5Dget_type)(hid_t);
  herr_t (*H5Dread)(hid_t, hid_t, hid_t, hid_t, hid_t, void*);
  int (*H5Sget_simple_extent_ndims)(hid_t);
  int (*H5Sget_simple_extent_dims)(hid_t, hsize_t*, hsize_t*);
  long long (*H5Sget_simple_extent_npoints)(hid_t);
  herr_t (*H5Sselect_hyperslab)(hid_t, int, const hsize_t*, const hsize_t*, const hsize_t*, const hsize_t*);
  hid_t (*H5Screate_simple)(int, const hsize_t*, const hsize_t*);
  herr_t (*H5Sclose)(hid_t);
  int (*H5Tget_class)(hid_t);
  size_t (*H5Tget_size)(hid_t);
  int (*H5Tget_sign)(hid_t);
  htri_t (*H5Tis_variable_str)(hid_t);
  hid_t (*H5Tcopy)(hid_t);
  herr_t (*H5Tset_size)(hid_t, size_t);
  herr_t (*H5Tclose)(hid_t);
  htri_t (*H5Aexists_by_name)(hid_t, const char*, const char*, hid_t);
  hid_t (*H5Aopen_by_name)(hid_t, const char*, const char*, hid_t, hid_t);
  hid_t (*H5Aget_type)(hid_t);
  hid_t (*H5Aget_space)(hid_t);
  herr_t (*H5Aread)(hid_t, hid_t, void*);
  herr_t (*H5Aclose)(hid_t);
  herr_t (*H5get_libversion)(unsigned*, unsigned*, unsigned*);
  // optional (1.10.3+ / 1.10.5+): raw chunk access for the parallel inflate path
  hid_t (*H5Dget_create_plist)(hid_t) = nullptr;
  int (*H5Pget_layout)(hid_t) = nullptr;
  int (*H5Pget_chunk)(hid_t, int, hsize_t*) = nullptr;
  int (*H5Pget_nfilters)(hid_t) = nullptr;
  int (*H5Pget_filter2)(hid_t, unsigned, unsigned*, size_t*, unsigned*, size_t, char*, unsigned*) = nullptr;
  herr_t (*H5Pclose)(hid_t) = nullptr;
  herr_t (*H5Dread_chunk)(hid_t, hid_t, const hsize_t*, unsigned*, void*) = nullptr;
  herr_t (*H5Dget_chunk_storage_size)(hid_t, const hsize_t*, hsize_t*) = nullptr;
  herr_t (*H5Dget_chunk_info_by_coord)(hid_t, const hsize_t*, unsigned*, unsigned long long* /* haddr_t */, hsize_t*) = nullptr;   // 1.10.5 on
  int (*H5Tget_order)(hid_t) = nullptr;
  int (*z_uncompress)(unsigned char*, unsigned long*, const unsigned char*, unsigned long) = nullptr;   // zlib, loaded beside
  herr_t (*H5free_memory)(void*) = nullptr;   // optional (1.8.13+): releases what the library allocated for variable-length strings
  hid_t* native_double = nullptr;   // H5T_NATIVE_DOUBLE_g
  hid_t* c_s1 = nullptr;            // H5T_C_S1_g
  std::string error;
};

Api* api() {
  static Api a;
  static bool tried = false;
  if (tried) return a.lib ? &a : nullptr;
  tried = true;
  std::vector<std::string> names;
  if (const char* e = std::getenv("ECCKD_HDF5_LIB")) names.push_back(e);
  for (const char* n : {"libhdf5.so", "libhdf5_serial.so", "libhdf5.so.103", "libhdf5.so.200", "libhdf5_serial.so.103",
                        "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so"})
    names.push_back(n);
  for (const std::string& n : names) {
    a.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (a.lib) break;
  }
  if (!a.lib) { a.error = "no HDF5 shared library found (set ECCKD_HDF5_LIB)"; return nullptr; }
  bool ok = true;
  auto sym = [&](const char* name) -> void* {
    void* p = dlsym(a.lib, name);
    if (!p) { ok = false; a.error = std::string("HDF5 library lacks ") + name; }
    return p;
  };
#define LOAD(f) a.f = reinterpret_cast<decltype(a.f)>(sym(#f))
  LOAD(H5open); LOAD(H5Eset_auto2); LOAD(H5Fopen); LOAD(H5Fclose); LOAD(H5Lexists); LOAD(H5Dopen2); LOAD(H5Dclose);
  LOAD(H5Dget_space); LOAD(H5Dget_type); LOAD(H5Dread); LOAD(H5Sget_simple_extent_ndims); LOAD(H5Sget_simple_extent_dims);
  LOAD(H5Sget_simple_extent_npoints); LOAD(H5Sselect_hyperslab); LOAD(H5Screate_simple); LOAD(H5Sclose); LOAD(H5Tget_class);
  LOAD(H5Tget_size); LOAD(H5Tget_sign); LOAD(H5Tis_variable_str); LOAD(H5Tcopy); LOAD(H5Tset_size); LOAD(H5Tclose);
  LOAD(H5Aexists_by_name); LOAD(H5Aopen_by_name); LOAD(H5Aget_type); LOAD(H5Aget_space); LOAD(H5Aread); LOAD(H5Aclose);
  LOAD(H5get_libversion);
#undef LOAD
  a.H5free_memory = reinterpret_cast<decltype(a.H5free_memory)>(dlsym(a.lib, "H5free_memory"));
#define OPT(f) a.f = reinterpret_cast<decltype(a.f)>(dlsym(a.lib, #f))
  OPT(H5Dget_create_plist); OPT(H5Pget_layout); OPT(H5Pget_chunk); OPT(H5Pget_nfilters); OPT(H5Pget_filter2); OPT(H5Pclose);
  OPT(H5Dread_chunk); OPT(H5Dget_chunk_storage_size); OPT(H5Tget_order); OPT(H5Dget_chunk_info_by_coord);
#undef OPT
  for (const char* zn : {"libz.so.1", "libz.so", "/opt/conda/lib/libz.so.1"}) {
    if (void* z = dlopen(zn, RTLD_NOW | RTLD_LOCAL)) {
      a.z_uncompress = reinterpret_cast<decltype(a.z_uncompress)>(dlsym(z, "uncompress"));
      if (a.z_uncompress) break;
    }
  }
  if (ok) {
    // the ABI declared above (64-bit hid_t, H5P_DEFAULT = H5S_ALL = 0) is that of HDF5 1.10 and later; 1.8 has a 32-bit hid_t
    unsigned maj = 0, min = 0, rel = 0;
    if (a.H5get_libversion(&maj, &min, &rel) < 0 || maj < 1 || (maj == 1 && min < 10)) {
      ok = false;
      a.error = "HDF5 library version " + std::to_string(maj) + "." + std::to_string(min) + "." + std::to_string(rel) +
                " is older than 1.10 (set ECCKD_HDF5_LIB to a newer one)";
    }
  }
  a.native_double = static_cast<hid_t*>(sym("H5T_NATIVE_DOUBLE_g"));
  a.c_s1 = static_cast<hid_t*>(sym("H5T_C_S1_g"));
  if (!ok) { dlclose(a.lib); a.lib = nullptr; return nullptr; }
  a.H5open();
  a.H5Eset_auto2(0, nullptr, nullptr);   // failures are reported through ecckd_last_error, not HDF5's stack dump
  return &a;
}

// NetCDF external type of an HDF5 datatype (classic numbering, CDF-5 for the rest)
int nc_type_of(Api* a, hid_t t) {
  const int cls = a->H5Tget_class(t);
  const size_t size = a->H5Tget_size(t);
  if (cls == 1) return size == 4 ? 5 : 6;                       // H5T_FLOAT
  if (cls == 3) return 2;                                       // H5T_STRING -> char
  if (cls == 0) {                                               // H5T_INTEGER
    const bool uns = a->H5Tget_sign(t) == 0;
    switch (size) {
      case 1: return uns ? 7 : 1;
      case 2: return uns ? 8 : 3;
      case 4: return uns ? 9 : 4;
      default: return uns ? 11 : 10;
    }
  }
  return 0;
}

}  // namespace

namespace ecckd {

struct H5File {
  Api* a = nullptr;
  hid_t file = -1;
  std::string path;
};

bool h5_is_hdf5(const unsigned char* magic8) {
  static const unsigned char sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
  return std::memcmp(magic8, sig, 8) == 0;
}

int h5_open(const char* path, H5File** out) {
  Api* a = api();
  if (!a) {
    return fail(ECCKD_PARAMETER_ERROR, "%s is a NetCDF-4 / HDF5 file and %s", path, "no usable HDF5 shared library was found (set ECCKD_HDF5_LIB)");
  }
  const hid_t f = a->H5Fopen(path, 0u /* H5F_ACC_RDONLY */, 0);
  if (f < 0) return fail(ECCKD_PARAMETER_ERROR, "cannot open %s as an HDF5 file", path);
  H5File* h = new H5File;
  h->a = a; h->file = f; h->path = path;
  *out = h;
  return ECCKD_OK;
}

void h5_close(H5File* h) {
  if (!h) return;
  if (h->file >= 0) h->a->H5Fclose(h->file);
  delete h;
}

// A variable that shares its name with a dimension it is not the coordinate variable of (the g-points file: dimension
// "g_point" = number of g points, variable "g_point"(wavenumber)) is stored under "_nc4_non_coord_<name>" (NetCDF-4 format);
// the plain name is then the dimension's scale dataset.
static std::string dataset_of_variable(H5File* h, const char* name) {
  const std::string alt = std::string("_nc4_non_coord_") + name;
  return h->a->H5Lexists(h->file, alt.c_str(), 0) > 0 ? alt : std::string(name);
}

static int inq_dataset(H5File* h, const char* name, int* exists, int* nc_type, int* ndims, size_t* shape, int shape_capacity);

int h5_inq_var(H5File* h, const char* name, int* exists, int* nc_type, int* ndims, size_t* shape, int shape_capacity) {
  const std::string ds = dataset_of_variable(h, name);
  int want_type = 0;
  return inq_dataset(h, ds.c_str(), exists, nc_type ? nc_type : &want_type, ndims, shape, shape_capacity);
}

static int inq_dataset(H5File* h, const char* name, int* exists, int* nc_type, int* ndims, size_t* shape, int shape_capacity) {
  Api* a = h->a;
  *exists = a->H5Lexists(h->file, name, 0) > 0 ? 1 : 0;
  if (!*exists) return ECCKD_OK;
  const hid_t d = a->H5Dopen2(h->file, name, 0);
  if (d < 0) { *exists = 0; return ECCKD_OK; }   // a group, not a variable
  const hid_t sp = a->H5Dget_space(d), t = a->H5Dget_type(d);
  // a dimension without a coordinate variable is a data-less scale dataset whose NAME says so (NetCDF-4 format): it has a
  // length (ecckd_nc_inq_dim) but is no variable
  if (nc_type && a->H5Aexists_by_name(h->file, name, "NAME", 0) > 0) {
    char text[80] = "";
    int ex = 0;
    if (h5_read_att_text(h, name, "NAME", &ex, text, sizeof text) == ECCKD_OK && ex &&
        std::strncmp(text, "This is a netCDF dimension but not a netCDF variable.", 53) == 0) {
      *exists = 0;
      a->H5Tclose(t); a->H5Sclose(sp); a->H5Dclose(d);
      return ECCKD_OK;
    }
  }
  const int nd = a->H5Sget_simple_extent_ndims(sp);
  hsize_t dims[32] = {};
  if (nd > 0) a->H5Sget_simple_extent_dims(sp, dims, nullptr);
  if (nc_type) *nc_type = nc_type_of(a, t);
  if (ndims) *ndims = nd < 0 ? 0 : nd;
  int rc = ECCKD_OK;
  if (shape) {
    if (nd > shape_capacity) rc = fail(ECCKD_PARAMETER_ERROR, "%s: \"%s\" has %d dimensions", h->path.c_str(), name, nd);
    else for (int k = 0; k < nd; ++k) shape[k] = (size_t)dims[k];
  }
  a->H5Tclose(t); a->H5Sclose(sp); a->H5Dclose(d);
  return rc;
}

int h5_inq_dim(H5File* h, const char* name, size_t* len) {
  int exists = 0, nd = 0;
  size_t shape[32];
  ECCKD_CHECK(inq_dataset(h, name, &exists, nullptr, &nd, shape, 32));   // a NetCDF-4 dimension is a (scale) dataset of its name
  if (!exists || nd != 1) return fail(ECCKD_PARAMETER_ERROR, "%s: no dimension \"%s\"", h->path.c_str(), name);
  *len = shape[0];
  return ECCKD_OK;
}

int h5_read_double(H5File* h, const char* var_name, long long slice, double* out, size_t capacity) {
  Api* a = h->a;
  const std::string ds = dataset_of_variable(h, var_name);
  const char* name = ds.c_str();
  if (a->H5Lexists(h->file, name, 0) <= 0) return fail(ECCKD_PARAMETER_ERROR, "%s: no variable \"%s\"", h->path.c_str(), name);
  const hid_t d = a->H5Dopen2(h->file, name, 0);
  if (d < 0) return fail(ECCKD_PARAMETER_ERROR, "%s: cannot open variable \"%s\"", h->path.c_str(), name);
  const hid_t sp = a->H5Dget_space(d);
  const int nd = a->H5Sget_simple_extent_ndims(sp);
  hsize_t dims[32] = {};
  if (nd > 0) a->H5Sget_simple_extent_dims(sp, dims, nullptr);
  hsize_t total = 1;
  for (int k = 0; k < nd; ++k) total *= dims[k];
  int rc = ECCKD_OK;
  hid_t mem = 0 /* H5S_ALL */, fsel = 0;
  if (slice >= 0) {
    if (nd < 1 || (hsize_t)slice >= dims[0]) {
      rc = fail(ECCKD_PARAMETER_ERROR, "%s: slice %lld of \"%s\" outside 0..%llu", h->path.c_str(), slice, name, nd < 1 ? 0ull : dims[0]);
    } else {
      hsize_t start[32] = {}, count[32];
      start[0] = (hsize_t)slice;
      count[0] = 1;
      total = 1;
      for (int k = 1; k < nd; ++k) { count[k] = dims[k]; total *= dims[k]; }
      a->H5Sselect_hyperslab(sp, 0 /* H5S_SELECT_SET */, start, nullptr, count, nullptr);
      mem = a->H5Screate_simple(1, &total, nullptr);
      fsel = sp;
    }
  }
  if (rc == ECCKD_OK && total > capacity)
    rc = fail(ECCKD_PARAMETER_ERROR, "ecckd_nc_read_double: \"%s\" needs %llu values, buffer holds %zu", name, total, capacity);
  if (rc == ECCKD_OK && total > 0 && a->H5Dread(d, *a->native_double, mem, fsel, 0, out) < 0)
    rc = fail(ECCKD_PROCESSING_ERROR, "%s: reading \"%s\" failed (missing filter plug-in?)", h->path.c_str(), name);
  if (mem > 0) a->H5Sclose(mem);
  a->H5Sclose(sp); a->H5Dclose(d);
  return rc;
}

// One slice (or all) of a chunked FLOAT / DOUBLE variable with the shuffle + deflate filters of NetCDF-4
// (OutputDataFile.cpp:350-359 writes them so; the CKDMIP spectra are stored that way): the calling thread pulls the RAW chunks
// out of the file (H5Dread_chunk: no filter pipeline, the library is not thread-safe), worker threads inflate, unshuffle and
// place them - the HDF5 library would do all of that on the one calling thread.  *handled = false: the layout is not
// one this path takes apart (contiguous, other filters, big-endian, missing chunks, old library); the caller falls back to H5Dread.
int h5_read_real_parallel(H5File* h, const char* var_name, long long slice, int out_type, void* out, size_t capacity, bool* handled) {
  Api* a = h->a;
  const std::string ds = dataset_of_variable(h, var_name);
  const char* name = ds.c_str();
  *handled = false;
  if (!a->H5Dread_chunk || !a->H5Dget_chunk_storage_size || !a->H5Dget_create_plist || !a->H5Pget_layout || !a->H5Pget_chunk ||
      !a->H5Pget_nfilters || !a->H5Pget_filter2 || !a->H5Pclose || !a->H5Tget_order || !a->z_uncompress ||
      std::getenv("ECCKD_NO_PARALLEL_INFLATE"))
    return ECCKD_OK;
  if (a->H5Lexists(h->file, name, 0) <= 0) return ECCKD_OK;
  const hid_t d = a->H5Dopen2(h->file, name, 0);
  if (d < 0) return ECCKD_OK;
  const hid_t sp = a->H5Dget_space(d), t = a->H5Dget_type(d), pl = a->H5Dget_create_plist(d);
  const int nd = a->H5Sget_simple_extent_ndims(sp);
  hsize_t dims[32] = {}, cdims[32] = {};
  if (nd > 0) a->H5Sget_simple_extent_dims(sp, dims, nullptr);
  const size_t ts = a->H5Tget_size(t);
  bool ok = nd >= 1 && nd <= 8 && a->H5Tget_class(t) == 1 && (ts == 4 || ts == 8) && a->H5Tget_order(t) == 0 /* little-endian */ &&
            a->H5Pget_layout(pl) == 2 /* H5D_CHUNKED */ && a->H5Pget_chunk(pl, nd, cdims) == nd;
  // filter pipeline: [shuffle,] deflate  (ids 2, 1), in that order
  int i_shuffle = -1, i_deflate = -1;
  if (ok) {
    const int nf = a->H5Pget_nfilters(pl);
    for (int i = 0; i < nf && ok; ++i) {
      unsigned flags = 0, cfg = 0, cd[8];
      size_t ncd = 8;
      char fname[8];
      const int id = a->H5Pget_filter2(pl, (unsigned)i, &flags, &ncd, cd, sizeof fname, fname, &cfg);
      if (id == 2 && i_shuffle < 0 && i_deflate < 0) i_shuffle = i;
      else if (id == 1 && i_deflate < 0) i_deflate = i;
      else ok = false;
    }
    ok = ok && i_deflate >= 0;
  }
  hsize_t lo[8] = {}, hi[8] = {};          // requested box [lo, hi) in dataset coordinates
  size_t total = 1;
  if (ok) {
    for (int k = 0; k < nd; ++k) { lo[k] = 0; hi[k] = dims[k]; }
    if (slice >= 0) {
      if ((hsize_t)slice >= dims[0]) ok = false;
      else { lo[0] = (hsize_t)slice; hi[0] = lo[0] + 1; }
    }
    for (int k = 0; k < nd; ++k) total *= (size_t)(hi[k] - lo[k]);
    ok = ok && total <= capacity && total > 0;
  }
  int rc = ECCKD_OK;
  if (ok) {
    // chunks that intersect the box
    hsize_t c0[8], c1[8];
    size_t nchunks = 1, chunk_elems = 1;
    for (int k = 0; k < nd; ++k) {
      c0[k] = lo[k] / cdims[k];
      c1[k] = (hi[k] - 1) / cdims[k];
      nchunks *= (size_t)(c1[k] - c0[k] + 1);
      chunk_elems *= (size_t)cdims[k];
    }
    struct Job { hsize_t off[8]; std::vector<unsigned char> raw; unsigned mask; };
    const unsigned nworkers = (unsigned)std::max(1, std::min(16, ecckd::host_cores()));
    std::vector<Job> jobs(nchunks);
    std::atomic<size_t> produced{0}, next{0};
    std::atomic<int> bad{0};
    const size_t out_ts = (size_t)out_type;
    auto worker = [&]() {
      std::vector<unsigned char> plain(chunk_elems * ts), unshuf(chunk_elems * ts);
      for (;;) {
        const size_t j = next.fetch_add(1);
        if (j >= nchunks) return;
        while (produced.load(std::memory_order_acquire) <= j && !bad.load()) std::this_thread::yield();
        if (bad.load()) return;
        Job& job = jobs[j];
        const unsigned char* data = job.raw.data();
        if (!(job.mask & (1u << i_deflate))) {
          unsigned long n = (unsigned long)plain.size();
          if (a->z_uncompress(plain.data(), &n, job.raw.data(), (unsigned long)job.raw.size()) != 0 || n != plain.size()) { bad.store(1); return; }
          data = plain.data();
        } else if (job.raw.size() != plain.size()) { bad.store(1); return; }
        if (i_shuffle >= 0 && !(job.mask & (1u << i_shuffle))) {       // byte b of element e sits at b * nelems + e
          for (size_t b = 0; b < ts; ++b) {
            const unsigned char* src = data + b * chunk_elems;
            for (size_t e = 0; e < chunk_elems; ++e) unshuf[e * ts + b] = src[e];
          }
          data = unshuf.data();
        }
        // copy the part of the chunk that lies in the box, innermost dimension contiguous
        hsize_t b0[8], b1[8];
        for (int k = 0; k < nd; ++k) { b0[k] = std::max(lo[k], job.off[k]); b1[k] = std::min(hi[k], job.off[k] + cdims[k]); }
        const size_t run = (size_t)(b1[nd - 1] - b0[nd - 1]);
        hsize_t idx[8];
        for (int k = 0; k < nd; ++k) idx[k] = b0[k];
        for (;;) {
          size_t src_e = 0, dst_e = 0;
          for (int k = 0; k < nd; ++k) { src_e = src_e * (size_t)cdims[k] + (size_t)(idx[k] - job.off[k]); dst_e = dst_e * (size_t)(hi[k] - lo[k]) + (size_t)(idx[k] - lo[k]); }
          if (ts == 4 && out_ts == 4) std::memcpy((float*)out + dst_e, data + src_e * 4, run * 4);
          else if (ts == 8 && out_ts == 8) std::memcpy((double*)out + dst_e, data + src_e * 8, run * 8);
          else if (ts == 4) { const float* sf = (const float*)(data + src_e * 4); for (size_t e = 0; e < run; ++e) ((double*)out)[dst_e + e] = (double)sf[e]; }
          else { const double* sd = (const double*)(data + src_e * 8); for (size_t e = 0; e < run; ++e) ((float*)out)[dst_e + e] = (float)sd[e]; }
          int k = nd - 2;
          for (; k >= 0; --k) { if (++idx[k] < b1[k]) break; idx[k] = b0[k]; }
          if (k < 0) break;
        }
        std::vector<unsigned char>().swap(job.raw);
      }
    };
    const auto t_start = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (unsigned w = 0; w < nworkers; ++w) pool.emplace_back(worker);
    // producer: raw chunks in row-major chunk order
    hsize_t ci[8];
    for (int k = 0; k < nd; ++k) ci[k] = c0[k];
    for (size_t j = 0; j < nchunks && !bad.load(); ++j) {
      Job& job = jobs[j];
      for (int k = 0; k < nd; ++k) job.off[k] = ci[k] * cdims[k];
      hsize_t bytes = 0;
      if (a->H5Dget_chunk_storage_size(d, job.off, &bytes) < 0 || bytes == 0) { bad.store(2); break; }   // an unwritten chunk: leave it to H5Dread (fill value)
      job.raw.resize((size_t)bytes);
      job.mask = 0;
      if (a->H5Dread_chunk(d, 0, job.off, &job.mask, job.raw.data()) < 0) { bad.store(3); break; }
      produced.store(j + 1, std::memory_order_release);
      for (int k = nd - 1; k >= 0; --k) { if (++ci[k] <= c1[k]) break; ci[k] = c0[k]; }
    }
    if (bad.load()) produced.store(nchunks, std::memory_order_release);
    const auto t_read = std::chrono::steady_clock::now();
    for (std::thread& th : pool) th.join();
    if (std::getenv("ECCKD_H5_TIMES"))
      std::fprintf(stderr, "%s \"%s\": %zu chunks, raw chunks read in %.1f ms, inflated and placed by %u threads %.1f ms later\n", h->path.c_str(), name,
                   nchunks, 1e3 * std::chrono::duration<double>(t_read - t_start).count(), nworkers,
                   1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_read).count());
    if (bad.load() == 1 || bad.load() == 3) rc = fail(ECCKD_PROCESSING_ERROR, "%s: a chunk of \"%s\" could not be read or inflated", h->path.c_str(), name);
    else if (bad.load() == 0) *handled = true;
  }
  a->H5Pclose(pl); a->H5Tclose(t); a->H5Sclose(sp); a->H5Dclose(d);
  return rc;
}

// ---- the same raw chunks for a consumer that inflates them elsewhere (nc_stream.hip: on the device) ----
struct H5ChunkReader {
  H5File* h = nullptr;
  hid_t d = -1, sp = -1, t = -1, pl = -1;
  H5ChunkPlan plan;
  hsize_t c0[8] = {}, c1[8] = {}, ci[8] = {};
  size_t next = 0;
  hsize_t off[8] = {};
  int i_shuffle = -1, i_deflate = -1;
};

void h5_chunks_close(H5ChunkReader* r) {
  if (!r) return;
  Api* a = r->h->a;
  if (r->pl >= 0) a->H5Pclose(r->pl);
  if (r->t >= 0) a->H5Tclose(r->t);
  if (r->sp >= 0) a->H5Sclose(r->sp);
  if (r->d >= 0) a->H5Dclose(r->d);
  delete r;
}

// *out = nullptr (and ECCKD_OK) if the variable's layout is not [shuffle,] deflate over little-endian FLOAT / DOUBLE chunks
int h5_chunks_open(H5File* h, const char* var_name, long long slice, size_t capacity, H5ChunkReader** out, H5ChunkPlan* plan) {
  Api* a = h->a;
  const std::string ds = dataset_of_variable(h, var_name);
  const char* name = ds.c_str();
  *out = nullptr;
  if (!a->H5Dread_chunk || !a->H5Dget_chunk_storage_size || !a->H5Dget_create_plist || !a->H5Pget_layout || !a->H5Pget_chunk ||
      !a->H5Pget_nfilters || !a->H5Pget_filter2 || !a->H5Pclose || !a->H5Tget_order)
    return ECCKD_OK;
  if (a->H5Lexists(h->file, name, 0) <= 0) return ECCKD_OK;
  H5ChunkReader* r = new H5ChunkReader;
  r->h = h;
  r->d = a->H5Dopen2(h->file, name, 0);
  if (r->d < 0) { delete r; return ECCKD_OK; }
  r->sp = a->H5Dget_space(r->d); r->t = a->H5Dget_type(r->d); r->pl = a->H5Dget_create_plist(r->d);
  H5ChunkPlan& P = r->plan;
  P = H5ChunkPlan();
  const int nd = a->H5Sget_simple_extent_ndims(r->sp);
  hsize_t dims[32] = {}, cdims[32] = {};
  if (nd > 0) a->H5Sget_simple_extent_dims(r->sp, dims, nullptr);
  const size_t ts = a->H5Tget_size(r->t);
  bool ok = nd >= 1 && nd <= 8 && a->H5Tget_class(r->t) == 1 && (ts == 4 || ts == 8) && a->H5Tget_order(r->t) == 0 &&
            a->H5Pget_layout(r->pl) == 2 && a->H5Pget_chunk(r->pl, nd, cdims) == nd;
  if (ok) {
    const int nf = a->H5Pget_nfilters(r->pl);
    for (int i = 0; i < nf && ok; ++i) {
      unsigned flags = 0, cfg = 0, cd[8];
      size_t ncd = 8;
      char fname[8];
      const int id = a->H5Pget_filter2(r->pl, (unsigned)i, &flags, &ncd, cd, sizeof fname, fname, &cfg);
      if (id == 2 && r->i_shuffle < 0 && r->i_deflate < 0) r->i_shuffle = i;
      else if (id == 1 && r->i_deflate < 0) r->i_deflate = i;
      else ok = false;
    }
    ok = ok && r->i_deflate >= 0;
  }
  if (ok) {
    P.nd = nd; P.ts = ts; P.shuffle = r->i_shuffle >= 0 ? 1 : 0;
    P.total = 1; P.nchunks = 1; P.chunk_elems = 1;
    for (int k = 0; k < nd; ++k) { P.dims[k] = dims[k]; P.cdims[k] = cdims[k]; P.lo[k] = 0; P.hi[k] = dims[k]; }
    if (slice >= 0) {
      if ((hsize_t)slice >= dims[0]) ok = false;
      else { P.lo[0] = (unsigned long long)slice; P.hi[0] = P.lo[0] + 1; }
    }
  }
  if (ok) {
    for (int k = 0; k < nd; ++k) {
      P.total *= (size_t)(P.hi[k] - P.lo[k]);
      r->c0[k] = P.lo[k] / cdims[k];
      r->c1[k] = (P.hi[k] - 1) / cdims[k];
      r->ci[k] = r->c0[k];
      P.nchunks *= (size_t)(r->c1[k] - r->c0[k] + 1);
      P.chunk_elems *= (size_t)cdims[k];
    }
    ok = P.total > 0 && P.total <= capacity;
  }
  if (!ok) { h5_chunks_close(r); return ECCKD_OK; }
  *plan = P;
  *out = r;
  return ECCKD_OK;
}

// the next chunk (row-major chunk order): its origin in dataset coordinates and its stored size; *bytes = 0 behind the last
// chunk or for a chunk that was never written (the caller then leaves the variable to the HDF5 library, which knows the
// fill value)
int h5_chunks_next(H5ChunkReader* r, unsigned long long* off, size_t* bytes, int* unwritten) {
  Api* a = r->h->a;
  *bytes = 0;
  *unwritten = 0;
  if (r->next >= r->plan.nchunks) return ECCKD_OK;
  for (int k = 0; k < r->plan.nd; ++k) { r->off[k] = r->ci[k] * r->plan.cdims[k]; off[k] = r->off[k]; }
  hsize_t n = 0;
  if (a->H5Dget_chunk_storage_size(r->d, r->off, &n) < 0 || n == 0) { *unwritten = 1; return ECCKD_OK; }
  *bytes = (size_t)n;
  return ECCKD_OK;
}

// zlib's uncompress as loaded beside the HDF5 library (thread-safe); false if there is none or the stream is not dst_len bytes
bool h5_inflate_host(H5File* h, void* dst, size_t dst_len, const void* src, size_t src_len) {
  Api* a = h->a;
  // the in-tree decoder first (fast_inflate.cpp: ~3x zlib on shuffled FLOATs; ECCKD_ZLIB_INFLATE: zlib alone); whatever it
  // does not take or does not pass its checks goes to zlib
  const bool zlib_only = std::getenv("ECCKD_ZLIB_INFLATE") != nullptr;   // (per call: probes and tests switch it)
  if (!zlib_only && fast_inflate_zlib(dst, dst_len, src, src_len)) return true;
  if (!a->z_uncompress) return false;
  unsigned long n = (unsigned long)dst_len;
  return a->z_uncompress((unsigned char*)dst, &n, (const unsigned char*)src, (unsigned long)src_len) == 0 && n == dst_len;
}
bool h5_has_zlib(H5File* h) { return h->a->z_uncompress != nullptr; }

// the bytes of the chunk h5_chunks_next announced, as stored (no filter is undone); *deflated / *shuffled: which filters were
// applied to this chunk
int h5_chunks_read(H5ChunkReader* r, void* dst, int* deflated, int* shuffled) {
  Api* a = r->h->a;
  unsigned mask = 0;
  if (a->H5Dread_chunk(r->d, 0, r->off, &mask, dst) < 0)
    return fail(ECCKD_PROCESSING_ERROR, "%s: a chunk could not be read", r->h->path.c_str());
  *deflated = !(mask & (1u << r->i_deflate));
  *shuffled = r->i_shuffle >= 0 && !(mask & (1u << r->i_shuffle));
  ++r->next;
  for (int k = r->plan.nd - 1; k >= 0; --k) { if (++r->ci[k] <= r->c1[k]) break; r->ci[k] = r->c0[k]; }
  return ECCKD_OK;
}

// where the chunk h5_chunks_next announced lies in the file (instead of its bytes: the caller reads them itself, from any
// thread); *addr = ~0 if the library cannot tell (before 1.10.5) or the chunk has no address.  advance: on to the next chunk, like
// h5_chunks_read (false: the same chunk can still be read through the library).
int h5_chunks_locate(H5ChunkReader* r, unsigned long long* addr, int* deflated, int* shuffled, bool advance) {
  Api* a = r->h->a;
  *addr = ~0ull;
  unsigned mask = 0;
  unsigned long long at = ~0ull;
  hsize_t size = 0;
  if (a->H5Dget_chunk_info_by_coord && a->H5Dget_chunk_info_by_coord(r->d, r->off, &mask, &at, &size) >= 0 && size > 0) *addr = at;
  *deflated = !(mask & (1u << r->i_deflate));
  *shuffled = r->i_shuffle >= 0 && !(mask & (1u << r->i_shuffle));
  if (!advance) return ECCKD_OK;
  ++r->next;
  for (int k = r->plan.nd - 1; k >= 0; --k) { if (++r->ci[k] <= r->c1[k]) break; r->ci[k] = r->c0[k]; }
  return ECCKD_OK;
}
bool h5_can_locate(H5File* h) { return h->a->H5Dget_chunk_info_by_coord != nullptr; }
const char* h5_path(H5File* h) { return h->path.c_str(); }

static hid_t open_att(H5File* h, const char* var, const char* att) {
  Api* a = h->a;
  const std::string ds = (var && var[0]) ? dataset_of_variable(h, var) : std::string("/");
  const char* obj = ds.c_str();
  if (obj[0] != '/' && a->H5Lexists(h->file, obj, 0) <= 0) return -2;
  if (a->H5Aexists_by_name(h->file, obj, att, 0) <= 0) return -1;
  return a->H5Aopen_by_name(h->file, obj, att, 0, 0);
}

int h5_read_att_text(H5File* h, const char* var, const char* att, int* exists, char* out, size_t capacity) {
  Api* a = h->a;
  const hid_t at = open_att(h, var, att);
  if (at == -2) return fail(ECCKD_PARAMETER_ERROR, "%s: no variable \"%s\"", h->path.c_str(), var);
  *exists = at >= 0 ? 1 : 0;
  if (at < 0) return ECCKD_OK;
  int rc = ECCKD_OK;
  const hid_t t = a->H5Aget_type(at);
  std::string text;
  if (a->H5Tget_class(t) != 3) {
    rc = fail(ECCKD_PARAMETER_ERROR, "attribute \"%s\" is not text", att);
  } else if (a->H5Tis_variable_str(t) > 0) {       // NC_STRING: the library allocates
    char* p = nullptr;
    const hid_t mt = a->H5Tcopy(*a->c_s1);
    a->H5Tset_size(mt, (size_t)-1 /* H5T_VARIABLE */);
    if (a->H5Aread(at, mt, &p) < 0) rc = fail(ECCKD_PROCESSING_ERROR, "%s: reading attribute \"%s\" failed", h->path.c_str(), att);
    if (p) { text = p; if (a->H5free_memory) a->H5free_memory(p); else std::free(p); }
    a->H5Tclose(mt);
  } else {                                         // NC_CHAR: fixed-length string (possibly an array of them)
    const size_t size = a->H5Tget_size(t);
    const hid_t sp = a->H5Aget_space(at);
    const long long n = a->H5Sget_simple_extent_npoints(sp);
    a->H5Sclose(sp);
    std::vector<char> buf(size * (size_t)(n > 0 ? n : 1) + 1, '\0');
    if (a->H5Aread(at, t, buf.data()) < 0) rc = fail(ECCKD_PROCESSING_ERROR, "%s: reading attribute \"%s\" failed", h->path.c_str(), att);
    text.assign(buf.data(), strnlen(buf.data(), buf.size() - 1));
  }
  a->H5Tclose(t); a->H5Aclose(at);
  if (rc != ECCKD_OK || !out) return rc;
  if (text.size() + 1 > capacity) return fail(ECCKD_PARAMETER_ERROR, "attribute \"%s\" needs %zu bytes", att, text.size() + 1);
  std::memcpy(out, text.c_str(), text.size() + 1);
  return ECCKD_OK;
}

int h5_read_att_double(H5File* h, const char* var, const char* att, int* nelems, double* out, size_t capacity) {
  Api* a = h->a;
  const hid_t at = open_att(h, var, att);
  if (at == -2) return fail(ECCKD_PARAMETER_ERROR, "%s: no variable \"%s\"", h->path.c_str(), var);
  *nelems = -1;
  if (at < 0) return ECCKD_OK;
  const hid_t sp = a->H5Aget_space(at);
  const long long n = a->H5Sget_simple_extent_npoints(sp);
  a->H5Sclose(sp);
  *nelems = (int)n;
  int rc = ECCKD_OK;
  if (out) {
    if ((size_t)n > capacity) rc = fail(ECCKD_PARAMETER_ERROR, "attribute \"%s\" has %lld values", att, n);
    else if (a->H5Aread(at, *a->native_double, out) < 0) rc = fail(ECCKD_PARAMETER_ERROR, "attribute \"%s\" is not numeric", att);
  }
  a->H5Aclose(at);
  return rc;
}

}  // namespace ecckd
