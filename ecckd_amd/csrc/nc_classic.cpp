// nc_classic.cpp - a self-contained reader / writer for the NetCDF CLASSIC on-disk formats
// (CDF-1, CDF-2 "64-bit offset", CDF-5 "64-bit data"), which is what the reference produces and
// consumes for files named *.nc / *.cdf (src/tools/DataFile.cpp:88-96, OutputDataFile.cpp:84-157).
// The image has no NetCDF library.  NetCDF-4 files (HDF5 containers) are recognised by their signature and READ
// through nc_hdf5.cpp (the system's HDF5 library, loaded at run time); files named *.h5 / *.hdf are WRITTEN as NetCDF-4 through
// nc_hdf5_write.cpp where that library (and its high-level library) can be loaded, classic otherwise.
//
// Layout (NetCDF classic format specification): big-endian throughout;
//   header = magic numrecs dim_list gatt_list var_list;  lists = tag(4) nelems [entries] or ABSENT (two zeros);
//   name = nelems bytes padded to 4;  att = name nc_type nelems values padded to 4;
//   var = name ndims dimids vatt_list nc_type vsize begin;  CDF-2 widens `begin` to 8 bytes, CDF-5 also the counts.
//   Fixed-size variables are stored contiguously at `begin`; record variables (slowest dimension of
//   length 0 = unlimited) are interleaved record by record.
// Reads convert any external type to double, as the reference does through nc_get_vara_double
// (src/tools/DataFileEngineNetcdf.cpp:593-599); `slice` selects one index of the slowest dimension
// like DataFile::read(M, "v", j) (:582-590).
#include "common.hpp"
#include "nc_hdf5.hpp"
#include "nc_classic.hpp"

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {

enum { NC_BYTE = 1, NC_CHAR = 2, NC_SHORT = 3, NC_INT = 4, NC_FLOAT = 5, NC_DOUBLE = 6,
       NC_UBYTE = 7, NC_USHORT = 8, NC_UINT = 9, NC_INT64 = 10, NC_UINT64 = 11 };
constexpr uint32_t TAG_DIM = 0x0A, TAG_VAR = 0x0B, TAG_ATT = 0x0C;

size_t type_size(int t) {
  switch (t) {
    case NC_BYTE: case NC_CHAR: case NC_UBYTE: return 1;
    case NC_SHORT: case NC_USHORT: return 2;
    case NC_INT: case NC_UINT: case NC_FLOAT: return 4;
    case NC_DOUBLE: case NC_INT64: case NC_UINT64: return 8;
    default: return 0;
  }
}
size_t pad4(size_t n) { return (n + 3) & ~(size_t)3; }

struct Att {
  std::string name;
  int type = 0;
  std::vector<unsigned char> raw;   // external (big-endian) bytes, unpadded
  size_t nelems = 0;
};
struct Dim { std::string name; uint64_t len = 0; };
struct Var {
  std::string name;
  std::vector<int> dimids;
  std::vector<Att> atts;
  int type = 0;
  uint64_t vsize = 0, begin = 0;
  bool record = false;
  bool deflate = false;    // NetCDF-4 output only: deflate_variable (OutputDataFile.cpp:345-359)
  bool past_end = false;   // the header places (part of) its data beyond the end of the file: refused when asked for, see open
};

// big-endian decode of one element to double
double decode(const unsigned char* p, int t) {
  switch (t) {
    case NC_BYTE: return (double)(int8_t)p[0];
    case NC_CHAR: case NC_UBYTE: return (double)p[0];
    case NC_SHORT: return (double)(int16_t)((p[0] << 8) | p[1]);
    case NC_USHORT: return (double)(uint16_t)((p[0] << 8) | p[1]);
    case NC_INT: return (double)(int32_t)(((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]);
    case NC_UINT: return (double)(((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]);
    case NC_FLOAT: { uint32_t u = ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; float f; std::memcpy(&f, &u, 4); return (double)f; }
    case NC_DOUBLE: case NC_INT64: case NC_UINT64: {
      uint64_t u = 0;
      for (int i = 0; i < 8; ++i) u = (u << 8) | p[i];
      if (t == NC_DOUBLE) { double d; std::memcpy(&d, &u, 8); return d; }
      return t == NC_INT64 ? (double)(int64_t)u : (double)u;
    }
  }
  return 0.0;
}
// encode one double as external type t (C conversion rules, like nc_put_vara_double)
void encode(unsigned char* p, int t, double v) {
  auto put = [&](uint64_t u, int nbytes) { for (int i = nbytes - 1; i >= 0; --i) { p[i] = (unsigned char)(u & 0xFF); u >>= 8; } };
  switch (t) {
    case NC_BYTE: put((uint64_t)(int64_t)(int8_t)v, 1); break;
    case NC_CHAR: case NC_UBYTE: put((uint64_t)(uint8_t)v, 1); break;
    case NC_SHORT: put((uint64_t)(int64_t)(int16_t)v, 2); break;
    case NC_USHORT: put((uint64_t)(uint16_t)v, 2); break;
    case NC_INT: put((uint64_t)(int64_t)(int32_t)v, 4); break;
    case NC_UINT: put((uint64_t)(uint32_t)v, 4); break;
    case NC_FLOAT: { float f = (float)v; uint32_t u; std::memcpy(&u, &f, 4); put(u, 4); break; }
    case NC_DOUBLE: { uint64_t u; std::memcpy(&u, &v, 8); put(u, 8); break; }
    case NC_INT64: put((uint64_t)(int64_t)v, 8); break;
    case NC_UINT64: put((uint64_t)v, 8); break;
  }
}

// Runs of one type at a time: the per-element switch of encode() / decode() held the tools' 1-D variables (an order file is
// six arrays of 7.2e6 values) at ~0.5 GB/s; the four types those files use get loops the compiler vectorises.  Same C conversion
// rules as encode() / decode().
void encode_run(unsigned char* p, int t, const double* v, size_t n) {
  switch (t) {
    case NC_DOUBLE:
      for (size_t i = 0; i < n; ++i) { uint64_t u; std::memcpy(&u, &v[i], 8); u = __builtin_bswap64(u); std::memcpy(p + 8 * i, &u, 8); }
      return;
    case NC_FLOAT:
      for (size_t i = 0; i < n; ++i) { const float f = (float)v[i]; uint32_t u; std::memcpy(&u, &f, 4); u = __builtin_bswap32(u); std::memcpy(p + 4 * i, &u, 4); }
      return;
    case NC_INT:
      for (size_t i = 0; i < n; ++i) { uint32_t u = (uint32_t)(int32_t)v[i]; u = __builtin_bswap32(u); std::memcpy(p + 4 * i, &u, 4); }
      return;
    case NC_SHORT:
      for (size_t i = 0; i < n; ++i) { uint16_t u = (uint16_t)(int16_t)v[i]; u = __builtin_bswap16(u); std::memcpy(p + 2 * i, &u, 2); }
      return;
    default:
      for (size_t i = 0; i < n; ++i) encode(p + i * type_size(t), t, v[i]);
  }
}
void decode_run(const unsigned char* p, int t, double* out, size_t n) {
  switch (t) {
    case NC_DOUBLE:
      for (size_t i = 0; i < n; ++i) { uint64_t u; std::memcpy(&u, p + 8 * i, 8); u = __builtin_bswap64(u); std::memcpy(&out[i], &u, 8); }
      return;
    case NC_FLOAT:
      for (size_t i = 0; i < n; ++i) { uint32_t u; std::memcpy(&u, p + 4 * i, 4); u = __builtin_bswap32(u); float x; std::memcpy(&x, &u, 4); out[i] = (double)x; }
      return;
    case NC_INT:
      for (size_t i = 0; i < n; ++i) { uint32_t u; std::memcpy(&u, p + 4 * i, 4); out[i] = (double)(int32_t)__builtin_bswap32(u); }
      return;
    case NC_SHORT:
      for (size_t i = 0; i < n; ++i) { uint16_t u; std::memcpy(&u, p + 2 * i, 2); out[i] = (double)(int16_t)__builtin_bswap16(u); }
      return;
    default:
      for (size_t i = 0; i < n; ++i) out[i] = decode(p + i * type_size(t), t);
  }
}

}  // namespace

struct ecckd_nc {
  ecckd::H5File* h5 = nullptr;   // set for a NetCDF-4 file opened for reading; the classic fields are then unused
  bool netcdf4 = false;          // a file being WRITTEN as NetCDF-4 (its name ends in .h5 / .hdf, OutputDataFile.cpp:84-111)
  ecckd::H5Writer* h5w = nullptr;
  FILE* fp = nullptr;
  bool writing = false, defining = false;
  int version = 1;
  uint64_t numrecs = 0, recsize = 0;
  std::vector<Dim> dims;
  std::vector<Att> gatts;
  std::vector<Var> vars;
  std::string path;

  const Var* find(const char* name) const {
    for (const Var& v : vars) if (v.name == name) return &v;
    return nullptr;
  }
  Var* find(const char* name) {
    for (Var& v : vars) if (v.name == name) return &v;
    return nullptr;
  }
  // number of elements of one "slice" (everything below the slowest dimension) and the slowest length
  void shape_of(const Var& v, std::vector<uint64_t>& shape) const {
    shape.clear();
    for (size_t k = 0; k < v.dimids.size(); ++k) {
      uint64_t len = dims[v.dimids[k]].len;
      if (k == 0 && v.record) len = numrecs;
      shape.push_back(len);
    }
  }
};

namespace {

struct Reader {
  FILE* fp;
  bool ok = true;
  explicit Reader(FILE* f) : fp(f) {}
  uint32_t u32() { unsigned char b[4]; if (fread(b, 1, 4, fp) != 4) { ok = false; return 0; } return ((uint32_t)b[0] << 24) | (b[1] << 16) | (b[2] << 8) | b[3]; }
  uint64_t u64() { uint64_t hi = u32(); uint64_t lo = u32(); return (hi << 32) | lo; }
  uint64_t count(int version) { return version == 5 ? u64() : u32(); }
  std::string name(int version) {
    uint64_t n = count(version);
    if (!ok || n > (1u << 20)) { ok = false; return std::string(); }
    std::string s(pad4(n), '\0');
    if (n && fread(&s[0], 1, pad4(n), fp) != pad4(n)) ok = false;
    s.resize(n);
    return s;
  }
  bool atts(int version, std::vector<Att>& out) {
    uint32_t tag = u32();
    uint64_t n = count(version);
    if (!ok) return false;
    if (tag == 0 && n == 0) return true;
    if (tag != TAG_ATT) return false;
    for (uint64_t i = 0; i < n; ++i) {
      Att a;
      a.name = name(version);
      a.type = (int)u32();
      a.nelems = count(version);
      const size_t bytes = a.nelems * type_size(a.type);
      if (!ok || type_size(a.type) == 0 || bytes > ((size_t)1 << 30)) return false;
      a.raw.resize(pad4(bytes));
      if (bytes && fread(a.raw.data(), 1, pad4(bytes), fp) != pad4(bytes)) return false;
      a.raw.resize(bytes);
      out.push_back(a);
    }
    return ok;
  }
};

int parse_header(ecckd_nc* f) {
  Reader r(f->fp);
  unsigned char magic[4];
  if (fread(magic, 1, 4, f->fp) != 4 || magic[0] != 'C' || magic[1] != 'D' || magic[2] != 'F' ||
      !(magic[3] == 1 || magic[3] == 2 || magic[3] == 5))
    return ecckd::fail(ECCKD_PARAMETER_ERROR, "%s is not a NetCDF classic file (CDF-1/2/5); NetCDF-4/HDF5 is not supported",
                       f->path.c_str());
  f->version = magic[3];
  f->numrecs = r.count(f->version);
  const bool streaming = f->version == 5 ? f->numrecs == UINT64_MAX : f->numrecs == 0xFFFFFFFFu;   // recomputed from the file size below
  if (streaming) f->numrecs = 0;
  {
    uint32_t tag = r.u32();
    uint64_t n = r.count(f->version);
    if (!(tag == 0 && n == 0)) {
      if (tag != TAG_DIM) return ecckd::fail(ECCKD_PARAMETER_ERROR, "%s: corrupt dimension list", f->path.c_str());
      for (uint64_t i = 0; i < n; ++i) {
        Dim d;
        d.name = r.name(f->version);
        d.len = r.count(f->version);
        if (!r.ok) return ecckd::fail(ECCKD_PARAMETER_ERROR, "%s: corrupt or truncated dimension list", f->path.c_str());   // (a damaged count must not spin)
        f->dims.push_back(d);
      }
    }
  }
  if (!r.atts(f->version, f->gatts)) return ecckd::fail(ECCKD_PARAMETER_ERROR, "%s: corrupt global attribute list", f->path.c_str());
  {
    uint32_t tag = r.u32();
    uint64_t n = r.count(f->version);
    if (!(tag == 0 && n == 0)) {
      if (tag != TAG_VAR) return ecckd::fail(ECCKD_PARAMETER_ERROR, "%s: corrupt variable list", f->path.c_str());
      for (uint64_t i = 0; i < n; ++i) {
        Var v;
        v.name = r.name(f->version);
        uint64_t nd = r.count(f->version);
        if (!r.ok || nd > 64) return ecckd::fail(ECCKD_PARAMETER_ERROR, "%s: corrupt variable header", f->path.c_str());
        for (uint64_t k = 0; k < nd; ++k) {
          const uint64_t id = r.count(f->version);
          if (id >= f->dims.size()) return ecckd::fail(ECCKD_PARAMETER_ERROR, "%s: bad dimension id", f->path.c_str());
          v.dimids.push_back((int)id);
        }
        if (!r.atts(f->version, v.atts)) return ecckd::fail(ECCKD_PARAMETER_ERROR, "%s: corrupt attribute list of %s", f->path.c_str(), v.name.c_str());
        v.type = (int)r.u32();
        v.vsize = r.count(f->version);
        v.begin = (f->version == 1) ? r.u32() : r.u64();
        v.record = !v.dimids.empty() && f->dims[v.dimids[0]].len == 0;
        if (!r.ok) return ecckd::fail(ECCKD_PARAMETER_ERROR, "%s: truncated header", f->path.c_str());
        if (type_size(v.type) == 0) return ecckd::fail(ECCKD_PARAMETER_ERROR, "%s: variable %s has unknown type %d", f->path.c_str(), v.name.c_str(), v.type);
        f->vars.push_back(v);
      }
    }
  }
  if (!r.ok) return ecckd::fail(ECCKD_PARAMETER_ERROR, "%s: truncated header", f->path.c_str());
  // record size: sum of the record variables' (padded) slabs; a single record variable is not padded
  int nrec = 0;
  for (const Var& v : f->vars) if (v.record) ++nrec;
  f->recsize = 0;
  for (const Var& v : f->vars) {
    if (!v.record) continue;
    uint64_t slab = type_size(v.type);
    for (size_t k = 1; k < v.dimids.size(); ++k) slab *= f->dims[v.dimids[k]].len;
    f->recsize += (nrec == 1) ? slab : pad4(slab);
  }
  // A variable whose data the header places beyond the end of the file (a damaged dimension length or offset, a cut-off
  // file, a NC_NOFILL file whose last variable was never written) is refused WHEN IT IS ASKED FOR - ecckd_nc_inq_var, which
  // sizes the callers' buffers, the readers and nc_locate_slice -; the other variables of the file stay readable, as they are
  // through the NetCDF library.
  uint64_t file_size = 0;
  {
    const long at = std::ftell(f->fp);
    if (at < 0 || std::fseek(f->fp, 0, SEEK_END) != 0) return ecckd::fail(ECCKD_PARAMETER_ERROR, "%s: cannot be measured", f->path.c_str());
    file_size = (uint64_t)std::ftell(f->fp);
    std::fseek(f->fp, at, SEEK_SET);
  }
  if (streaming) {
    // numrecs = 0xFFFFFFFF ("indeterminate", written by a streaming producer): as many whole records as the file holds
    uint64_t first = UINT64_MAX;
    for (const Var& v : f->vars) if (v.record) first = std::min(first, v.begin);
    f->numrecs = (f->recsize > 0 && first <= file_size) ? (file_size - first) / f->recsize : 0;
  }
  for (Var& v : f->vars) {
    unsigned __int128 bytes = type_size(v.type);
    for (size_t k = v.record ? 1 : 0; k < v.dimids.size(); ++k) bytes *= f->dims[v.dimids[k]].len, bytes = bytes > ((unsigned __int128)1 << 100) ? ((unsigned __int128)1 << 100) : bytes;
    unsigned __int128 last = (unsigned __int128)v.begin + bytes;
    if (v.record && f->numrecs > 0) last = (unsigned __int128)v.begin + (unsigned __int128)(f->numrecs - 1) * f->recsize + bytes;
    v.past_end = (v.record && f->numrecs == 0) ? v.begin > file_size : last > file_size;
  }
  return ECCKD_OK;
}

int refuse_past_end(const ecckd_nc* f, const Var& v) {
  return ecckd::fail(ECCKD_PARAMETER_ERROR, "%s: variable %s extends past the end of the file (damaged header or truncated file)",
                     f->path.c_str(), v.name.c_str());
}

void put_u32(std::vector<unsigned char>& b, uint32_t v) { for (int i = 3; i >= 0; --i) b.push_back((unsigned char)(v >> (8 * i))); }
void put_u64(std::vector<unsigned char>& b, uint64_t v) { put_u32(b, (uint32_t)(v >> 32)); put_u32(b, (uint32_t)v); }
void put_count(std::vector<unsigned char>& b, int version, uint64_t v) { if (version == 5) put_u64(b, v); else put_u32(b, (uint32_t)v); }
void put_name(std::vector<unsigned char>& b, int version, const std::string& s) {
  put_count(b, version, s.size());
  b.insert(b.end(), s.begin(), s.end());
  b.resize(b.size() + (pad4(s.size()) - s.size()), 0);
}
void put_atts(std::vector<unsigned char>& b, int version, const std::vector<Att>& atts) {
  if (atts.empty()) { put_u32(b, 0); put_count(b, version, 0); return; }
  put_u32(b, TAG_ATT);
  put_count(b, version, atts.size());
  for (const Att& a : atts) {
    put_name(b, version, a.name);
    put_u32(b, (uint32_t)a.type);
    put_count(b, version, a.nelems);
    b.insert(b.end(), a.raw.begin(), a.raw.end());
    b.resize(b.size() + (pad4(a.raw.size()) - a.raw.size()), 0);
  }
}
std::vector<unsigned char> build_header(const ecckd_nc* f) {
  std::vector<unsigned char> b = {'C', 'D', 'F', (unsigned char)f->version};
  put_count(b, f->version, 0);   // numrecs: no record variables are written
  if (f->dims.empty()) { put_u32(b, 0); put_count(b, f->version, 0); }
  else {
    put_u32(b, TAG_DIM);
    put_count(b, f->version, f->dims.size());
    for (const Dim& d : f->dims) { put_name(b, f->version, d.name); put_count(b, f->version, d.len); }
  }
  put_atts(b, f->version, f->gatts);
  if (f->vars.empty()) { put_u32(b, 0); put_count(b, f->version, 0); }
  else {
    put_u32(b, TAG_VAR);
    put_count(b, f->version, f->vars.size());
    for (const Var& v : f->vars) {
      put_name(b, f->version, v.name);
      put_count(b, f->version, v.dimids.size());
      for (int id : v.dimids) put_count(b, f->version, (uint64_t)id);
      put_atts(b, f->version, v.atts);
      put_u32(b, (uint32_t)v.type);
      put_count(b, f->version, f->version == 5 ? v.vsize : (v.vsize > 0xFFFFFFFFull ? 0xFFFFFFFFull : v.vsize));
      if (f->version == 1) put_u32(b, (uint32_t)v.begin); else put_u64(b, v.begin);
    }
  }
  return b;
}

Att make_text_att(const char* name, const char* text) {
  Att a;
  a.name = name;
  a.type = NC_CHAR;
  a.nelems = std::strlen(text);
  a.raw.assign((const unsigned char*)text, (const unsigned char*)text + a.nelems);
  return a;
}

}  // namespace

extern "C" {

int ecckd_nc_open(const char* path, ecckd_nc** out) {
  ECCKD_REQUIRE(path && out, "ecckd_nc_open: NULL argument");
  *out = nullptr;
  FILE* fp = std::fopen(path, "rb");
  if (!fp) return ecckd::fail(ECCKD_PARAMETER_ERROR, "cannot open %s for reading", path);
  unsigned char magic[8] = {};
  if (std::fread(magic, 1, 8, fp) == 8 && ecckd::h5_is_hdf5(magic)) {
    std::fclose(fp);
    ecckd::H5File* h5 = nullptr;
    ECCKD_CHECK(ecckd::h5_open(path, &h5));
    ecckd_nc* f = new ecckd_nc;
    f->h5 = h5;
    f->path = path;
    *out = f;
    return ECCKD_OK;
  }
  std::rewind(fp);
  ecckd_nc* f = new ecckd_nc;
  f->fp = fp;
  f->path = path;
  int rc = parse_header(f);
  if (rc != ECCKD_OK) { std::fclose(fp); delete f; return rc; }
  *out = f;
  return ECCKD_OK;
}

int ecckd_nc_close(ecckd_nc* f) {
  if (!f) return ECCKD_OK;
  if (f->h5) { ecckd::h5_close(f->h5); delete f; return ECCKD_OK; }
  int rc = ECCKD_OK;
  if (f->netcdf4) {
    if (f->defining) rc = ecckd::fail(ECCKD_PROCESSING_ERROR, "%s closed while still in define mode", f->path.c_str());
    const int rc2 = ecckd::h5w_close(f->h5w);
    delete f;
    return rc != ECCKD_OK ? rc : rc2;
  }
  if (f->writing && f->defining) rc = ecckd::fail(ECCKD_PROCESSING_ERROR, "%s closed while still in define mode", f->path.c_str());
  if (f->fp && std::fclose(f->fp) != 0) rc = ecckd::fail(ECCKD_PROCESSING_ERROR, "error closing %s", f->path.c_str());
  delete f;
  return rc;
}

int ecckd_nc_inq_dim(ecckd_nc* f, const char* name, size_t* len) {
  ECCKD_REQUIRE(f && name && len, "ecckd_nc_inq_dim: NULL argument");
  if (f->h5) return ecckd::h5_inq_dim(f->h5, name, len);
  for (const Dim& d : f->dims)
    if (d.name == name) { *len = (size_t)(d.len == 0 ? f->numrecs : d.len); return ECCKD_OK; }
  return ecckd::fail(ECCKD_PARAMETER_ERROR, "%s: no dimension \"%s\"", f->path.c_str(), name);
}

int ecckd_nc_inq_var(ecckd_nc* f, const char* name, int* exists, int* nc_type, int* ndims, size_t* shape, int shape_capacity) {
  ECCKD_REQUIRE(f && name && exists, "ecckd_nc_inq_var: NULL argument");
  if (f->h5) return ecckd::h5_inq_var(f->h5, name, exists, nc_type, ndims, shape, shape_capacity);
  const Var* v = f->find(name);
  *exists = v ? 1 : 0;
  if (!v) return ECCKD_OK;
  if (v->past_end) return refuse_past_end(f, *v);        // its shape must not size anybody's buffer
  std::vector<uint64_t> sh;
  f->shape_of(*v, sh);
  if (nc_type) *nc_type = v->type;
  if (ndims) *ndims = (int)sh.size();
  if (shape) {
    ECCKD_REQUIRE((int)sh.size() <= shape_capacity, "ecckd_nc_inq_var: %s has %zu dimensions", name, sh.size());
    for (size_t k = 0; k < sh.size(); ++k) shape[k] = (size_t)sh[k];
  }
  return ECCKD_OK;
}

}  // extern "C" (reopened below)

namespace ecckd {
H5File* nc_h5_handle(ecckd_nc* f) { return f ? f->h5 : nullptr; }

int nc_locate_slice(ecckd_nc* f, const char* name, long long slice, NcSlice* out) {
  ECCKD_REQUIRE(f && name && out && !f->writing, "nc_locate_slice: bad argument");
  *out = NcSlice();
  if (f->h5) return ECCKD_OK;                 // not contiguous: the caller reads through the HDF5 layer
  const Var* v = f->find(name);
  if (!v) return ecckd::fail(ECCKD_PARAMETER_ERROR, "%s: no variable \"%s\"", f->path.c_str(), name);
  if (v->past_end) return refuse_past_end(f, *v);
  std::vector<uint64_t> sh;
  f->shape_of(*v, sh);
  uint64_t per_slice = 1;
  for (size_t k = 1; k < sh.size(); ++k) per_slice *= sh[k];
  const uint64_t nslice = sh.empty() ? 1 : sh[0];
  if (slice >= 0)
    ECCKD_REQUIRE(!sh.empty() && (uint64_t)slice < nslice, "%s: slice %lld of \"%s\" outside 0..%llu", f->path.c_str(), slice, name,
                  (unsigned long long)nslice);
  out->nc_type = v->type;
  out->count = slice >= 0 ? per_slice : per_slice * nslice;
  if (v->record) return ECCKD_OK;              // interleaved with the other record variables
  out->fd = fileno(f->fp);
  out->offset = v->begin + (slice >= 0 ? (uint64_t)slice * per_slice * type_size(v->type) : 0);
  out->contiguous = true;
  return ECCKD_OK;
}
}  // namespace ecckd

extern "C" {

int ecckd_nc_read_double(ecckd_nc* f, const char* name, long long slice, double* out, size_t capacity) {
  ECCKD_REQUIRE(f && name && out && !f->writing, "ecckd_nc_read_double: bad argument");
  if (f->h5) {
    bool handled = false;        // big chunked FLOAT / DOUBLE variables: the chunks inflated by worker threads
    ECCKD_CHECK(ecckd::h5_read_real_parallel(f->h5, name, slice, 8, out, capacity, &handled));
    if (handled) return ECCKD_OK;
    return ecckd::h5_read_double(f->h5, name, slice, out, capacity);
  }
  const Var* v = f->find(name);
  if (!v) return ecckd::fail(ECCKD_PARAMETER_ERROR, "%s: no variable \"%s\"", f->path.c_str(), name);
  if (v->past_end) return refuse_past_end(f, *v);
  std::vector<uint64_t> sh;
  f->shape_of(*v, sh);
  uint64_t per_slice = 1;
  for (size_t k = 1; k < sh.size(); ++k) per_slice *= sh[k];
  const uint64_t nslice = sh.empty() ? 1 : sh[0];
  uint64_t s0 = 0, s1 = nslice;
  if (slice >= 0) {
    ECCKD_REQUIRE(!sh.empty() && (uint64_t)slice < nslice, "%s: slice %lld of \"%s\" outside 0..%llu", f->path.c_str(), slice, name,
                  (unsigned long long)nslice);
    s0 = (uint64_t)slice;
    s1 = s0 + 1;
  }
  const uint64_t total = (s1 - s0) * per_slice;
  ECCKD_REQUIRE(total <= capacity, "ecckd_nc_read_double: \"%s\" needs %llu values, buffer holds %zu", name,
                (unsigned long long)total, capacity);
  const size_t ts = type_size(v->type);
  size_t o = 0;
  // a fixed-size variable lies in one piece: the requested slices are one contiguous run (a 1-D variable is "slices" of
  // ONE element each - seeking to every one of them made reading a 4e6-point grid take seconds); a record variable is
  // interleaved with the others record by record
  const uint64_t nrun = v->record ? (s1 - s0) : 1;
  const uint64_t run_elems = v->record ? per_slice : total;
  // a long run of a file opened for reading (the 1-D variables of an ordering or g-points file: 7.2e6 values each): cut into
  // up to eight ranges that a thread each reads (pread at the range's own offset) and decodes - one thread manages ~1 GB/s
  const size_t piece = (size_t)1 << 20;
  if (nrun == 1 && run_elems >= 4 * piece) {
    const int fd = ::fileno(f->fp);
    // (one record of a record variable: the records lie recsize apart, not per_slice * ts)
    const uint64_t at = v->record ? v->begin + s0 * f->recsize : v->begin + s0 * per_slice * ts;
    const size_t nthreads = (size_t)std::min<uint64_t>(8, run_elems / (2 * piece));
    const size_t share = (size_t)((run_elems + nthreads - 1) / nthreads);
    std::vector<char> good(nthreads, 1);
    std::vector<std::thread> pool;
    const int vtype = v->type;
    for (size_t t = 0; t < nthreads; ++t) {
      const size_t first = t * share, n_values = first < run_elems ? (size_t)std::min<uint64_t>(share, run_elems - first) : 0;
      pool.emplace_back([&, t, first, n_values] {
        std::vector<unsigned char> b(std::min(n_values, piece) * ts);
        size_t i = 0;
        while (i < n_values) {
          const size_t n = std::min(n_values - i, piece);
          size_t done = 0;
          while (done < n * ts) {
            const ssize_t got = ::pread(fd, b.data() + done, n * ts - done, (off_t)(at + (uint64_t)(first + i) * ts + done));
            if (got <= 0) { good[t] = 0; return; }
            done += (size_t)got;
          }
          decode_run(b.data(), vtype, out + first + i, n);
          i += n;
        }
      });
    }
    for (auto& th : pool) th.join();
    for (char gd : good) if (!gd) return ecckd::fail(ECCKD_PROCESSING_ERROR, "%s: short read of \"%s\"", f->path.c_str(), name);
    return ECCKD_OK;
  }
  std::vector<unsigned char> buf((size_t)std::min<uint64_t>(std::max<uint64_t>(run_elems, 1), (uint64_t)1 << 20) * ts);
  for (uint64_t r = 0; r < nrun; ++r) {
    const uint64_t off = v->record ? v->begin + (s0 + r) * f->recsize : v->begin + s0 * per_slice * ts;
    if (fseeko(f->fp, (off_t)off, SEEK_SET) != 0) return ecckd::fail(ECCKD_PROCESSING_ERROR, "%s: seek failed", f->path.c_str());
    uint64_t left = run_elems;
    while (left > 0) {
      const size_t n = (size_t)std::min<uint64_t>(left, buf.size() / ts);
      if (fread(buf.data(), ts, n, f->fp) != n) return ecckd::fail(ECCKD_PROCESSING_ERROR, "%s: short read of \"%s\"", f->path.c_str(), name);
      decode_run(buf.data(), v->type, out + o, n);
      o += n;
      left -= n;
    }
  }
  return ECCKD_OK;
}

// var == NULL or "" selects the global attributes (scope "_global_", DataFileEngine.h:28)
static const Att* find_att(ecckd_nc* f, const char* var, const char* att, int* rc) {
  *rc = ECCKD_OK;
  const std::vector<Att>* list = &f->gatts;
  if (var && var[0]) {
    const Var* v = f->find(var);
    if (!v) { *rc = ecckd::fail(ECCKD_PARAMETER_ERROR, "%s: no variable \"%s\"", f->path.c_str(), var); return nullptr; }
    list = &v->atts;
  }
  for (const Att& a : *list) if (a.name == att) return &a;
  return nullptr;
}

int ecckd_nc_read_att_text(ecckd_nc* f, const char* var, const char* att, int* exists, char* out, size_t capacity) {
  ECCKD_REQUIRE(f && att && exists, "ecckd_nc_read_att_text: NULL argument");
  if (f->h5) return ecckd::h5_read_att_text(f->h5, var, att, exists, out, capacity);
  int rc;
  const Att* a = find_att(f, var, att, &rc);
  if (rc != ECCKD_OK) return rc;
  *exists = a ? 1 : 0;
  if (!a || !out) return ECCKD_OK;
  ECCKD_REQUIRE(a->type == NC_CHAR, "attribute \"%s\" is not text", att);
  ECCKD_REQUIRE(a->raw.size() + 1 <= capacity, "attribute \"%s\" needs %zu bytes", att, a->raw.size() + 1);
  std::memcpy(out, a->raw.data(), a->raw.size());
  out[a->raw.size()] = '\0';
  return ECCKD_OK;
}

int ecckd_nc_read_att_double(ecckd_nc* f, const char* var, const char* att, int* nelems, double* out, size_t capacity) {
  ECCKD_REQUIRE(f && att && nelems, "ecckd_nc_read_att_double: NULL argument");
  if (f->h5) return ecckd::h5_read_att_double(f->h5, var, att, nelems, out, capacity);
  int rc;
  const Att* a = find_att(f, var, att, &rc);
  if (rc != ECCKD_OK) return rc;
  *nelems = a ? (int)a->nelems : -1;
  if (!a || !out) return ECCKD_OK;
  ECCKD_REQUIRE(a->nelems <= capacity, "attribute \"%s\" has %zu values", att, a->nelems);
  const size_t ts = type_size(a->type);
  for (size_t i = 0; i < a->nelems; ++i) out[i] = decode(a->raw.data() + i * ts, a->type);
  return ECCKD_OK;
}

// ---- writing: define dimensions / variables / attributes, ecckd_nc_enddef, then write whole variables ----
int ecckd_nc_create(const char* path, ecckd_nc** out) {
  ECCKD_REQUIRE(path && out, "ecckd_nc_create: NULL argument");
  *out = nullptr;
  // The reference picks the format by the file name (OutputDataFile.cpp:84-111): .nc / .cdf -> classic, .h5 / .hdf -> NetCDF-4.
  // NetCDF-4 needs the HDF5 library and its high-level library at run time (nc_hdf5_write.cpp); where they are missing, or with
  // ECCKD_CLASSIC_OUTPUT set, the file is written in the classic format under the name asked for (the NetCDF library and this
  // repository's reader key on the content, not the name).
  const std::string name(path);
  const size_t dot = name.find_last_of('.');
  const std::string ext = dot == std::string::npos ? std::string() : name.substr(dot + 1);
  const bool wants4 = (ext == "h5" || ext == "hdf") && std::getenv("ECCKD_CLASSIC_OUTPUT") == nullptr;
  if (wants4 && ecckd::h5w_available(nullptr)) {
    ecckd_nc* f = new ecckd_nc;
    f->path = path;
    f->netcdf4 = true;
    f->writing = f->defining = true;
    *out = f;
    return ECCKD_OK;
  }
  FILE* fp = std::fopen(path, "wb");
  if (!fp) return ecckd::fail(ECCKD_PARAMETER_ERROR, "cannot open %s for writing", path);
  ecckd_nc* f = new ecckd_nc;
  f->fp = fp;
  f->path = path;
  f->writing = f->defining = true;
  *out = f;
  return ECCKD_OK;
}

int ecckd_nc_def_dim(ecckd_nc* f, const char* name, size_t len, int* dimid) {
  ECCKD_REQUIRE(f && f->defining && name && len > 0, "ecckd_nc_def_dim: bad argument (record dimensions are not written)");
  Dim d;
  d.name = name;
  d.len = len;
  f->dims.push_back(d);
  if (dimid) *dimid = (int)f->dims.size() - 1;
  return ECCKD_OK;
}

int ecckd_nc_def_var(ecckd_nc* f, const char* name, int nc_type, int ndims, const int* dimids, int* varid) {
  ECCKD_REQUIRE(f && f->defining && name && type_size(nc_type) > 0 && ndims >= 0 && (ndims == 0 || dimids), "ecckd_nc_def_var: bad argument");
  Var v;
  v.name = name;
  v.type = nc_type;
  for (int k = 0; k < ndims; ++k) {
    ECCKD_REQUIRE(dimids[k] >= 0 && dimids[k] < (int)f->dims.size(), "ecckd_nc_def_var: bad dimension id %d", dimids[k]);
    v.dimids.push_back(dimids[k]);
  }
  f->vars.push_back(v);
  if (varid) *varid = (int)f->vars.size() - 1;
  return ECCKD_OK;
}

int ecckd_nc_put_att_text(ecckd_nc* f, const char* var, const char* att, const char* text) {
  ECCKD_REQUIRE(f && f->defining && att && text, "ecckd_nc_put_att_text: bad argument");
  std::vector<Att>* list = &f->gatts;
  if (var && var[0]) {
    Var* v = f->find(var);
    if (!v) return ecckd::fail(ECCKD_PARAMETER_ERROR, "ecckd_nc_put_att_text: no variable \"%s\"", var);
    list = &v->atts;
  }
  for (Att& a : *list) if (a.name == att) { a = make_text_att(att, text); return ECCKD_OK; }
  list->push_back(make_text_att(att, text));
  return ECCKD_OK;
}

int ecckd_nc_put_att_double(ecckd_nc* f, const char* var, const char* att, int nc_type, int n, const double* values) {
  ECCKD_REQUIRE(f && f->defining && att && values && n > 0 && type_size(nc_type) > 0 && nc_type != NC_CHAR, "ecckd_nc_put_att_double: bad argument");
  std::vector<Att>* list = &f->gatts;
  if (var && var[0]) {
    Var* v = f->find(var);
    if (!v) return ecckd::fail(ECCKD_PARAMETER_ERROR, "ecckd_nc_put_att_double: no variable \"%s\"", var);
    list = &v->atts;
  }
  Att a;
  a.name = att;
  a.type = nc_type;
  a.nelems = (size_t)n;
  a.raw.resize((size_t)n * type_size(nc_type));
  for (int i = 0; i < n; ++i) encode(a.raw.data() + (size_t)i * type_size(nc_type), nc_type, values[i]);
  list->push_back(a);
  return ECCKD_OK;
}

// deflate_variable (OutputDataFile.cpp:345-359): shuffle + deflate level 2 in a NetCDF-4 file, nothing in a classic one
int ecckd_nc_deflate_var(ecckd_nc* f, const char* name) {
  ECCKD_REQUIRE(f && f->defining && name, "ecckd_nc_deflate_var: bad argument");
  Var* v = f->find(name);
  if (!v) return ecckd::fail(ECCKD_PARAMETER_ERROR, "ecckd_nc_deflate_var: no variable \"%s\"", name);
  v->deflate = true;
  return ECCKD_OK;
}

int ecckd_nc_is_netcdf4(ecckd_nc* f, int* is_netcdf4) {
  ECCKD_REQUIRE(f && is_netcdf4, "ecckd_nc_is_netcdf4: NULL argument");
  *is_netcdf4 = (f->netcdf4 || f->h5) ? 1 : 0;
  return ECCKD_OK;
}

int ecckd_nc_enddef(ecckd_nc* f) {
  ECCKD_REQUIRE(f && f->writing && f->defining, "ecckd_nc_enddef: not in define mode");
  if (f->netcdf4) {
    auto conv = [](const Att& a) {
      ecckd::H5WAtt o;
      o.name = a.name;
      o.nc_type = a.type;
      if (a.type == NC_CHAR) o.text.assign((const char*)a.raw.data(), a.raw.size());
      else for (size_t i = 0; i < a.nelems; ++i) o.values.push_back(decode(a.raw.data() + i * type_size(a.type), a.type));
      return o;
    };
    std::vector<ecckd::H5WDim> dims;
    for (const Dim& d : f->dims) { ecckd::H5WDim o; o.name = d.name; o.len = d.len; dims.push_back(o); }
    std::vector<ecckd::H5WVar> vars;
    for (const Var& v : f->vars) {
      ecckd::H5WVar o;
      o.name = v.name; o.nc_type = v.type; o.dimids = v.dimids; o.deflate = v.deflate;
      for (const Att& a : v.atts) o.atts.push_back(conv(a));
      vars.push_back(o);
    }
    std::vector<ecckd::H5WAtt> gatts;
    for (const Att& a : f->gatts) gatts.push_back(conv(a));
    ECCKD_CHECK(ecckd::h5w_create(f->path.c_str(), dims, vars, gatts, &f->h5w));
    f->defining = false;
    return ECCKD_OK;
  }
  // sizes first: they decide the format variant (CDF-1: every offset < 2 GiB; CDF-2: every variable < 4 GiB; else CDF-5)
  uint64_t data_bytes = 0, max_var = 0;
  for (Var& v : f->vars) {
    uint64_t nel = 1;
    for (int id : v.dimids) nel *= f->dims[id].len;
    v.vsize = pad4(nel * type_size(v.type));
    data_bytes += v.vsize;
    max_var = std::max(max_var, v.vsize);
  }
  bool big_dim = false;
  for (const Dim& d : f->dims) big_dim = big_dim || d.len > 0xFFFFFFFFull;
  for (int version : {1, 2, 5}) {
    f->version = version;
    const size_t hdr = build_header(f).size();   // offsets do not change the header size within a variant
    if (version == 1 && hdr + data_bytes < ((uint64_t)1 << 31) && !big_dim) break;
    if (version == 2 && max_var < ((uint64_t)1 << 32) - 4 && !big_dim) break;
  }
  uint64_t off = build_header(f).size();
  for (Var& v : f->vars) { v.begin = off; off += v.vsize; }
  const std::vector<unsigned char> hdr = build_header(f);
  if (std::fwrite(hdr.data(), 1, hdr.size(), f->fp) != hdr.size()) return ecckd::fail(ECCKD_PROCESSING_ERROR, "%s: header write failed", f->path.c_str());
  // reserve the data section so that unwritten variables read back as zeros
  if (off > hdr.size()) {
    if (fseeko(f->fp, (off_t)(off - 1), SEEK_SET) != 0 || std::fputc(0, f->fp) == EOF)
      return ecckd::fail(ECCKD_PROCESSING_ERROR, "%s: cannot extend file to %llu bytes", f->path.c_str(), (unsigned long long)off);
  }
  f->defining = false;
  return ECCKD_OK;
}

}  // extern "C"

namespace {
// `count` values of variable v to the file from byte offset `at`: encoded in pieces of 2^20 values; a long run (the 1-D variables
// of an ordering or g-points file: 7.2e6 values each) is cut into up to eight ranges that are encoded and written by a thread
// each (pwrite at the range's own offset), since one thread encodes and copies into the page cache at ~1 GB/s only.
int write_values(ecckd_nc* f, const Var& v, const char* name, uint64_t at, const double* data, size_t count) {
  const size_t ts = type_size(v.type);
  const size_t piece = (size_t)1 << 20;
  auto write_range = [&](size_t first, size_t n_values, int fd) -> bool {
    std::vector<unsigned char> buf(std::min(n_values, piece) * ts);
    size_t i = 0;
    while (i < n_values) {
      const size_t n = std::min(n_values - i, piece);
      encode_run(buf.data(), v.type, data + first + i, n);
      size_t done = 0;
      while (done < n * ts) {
        const ssize_t w = ::pwrite(fd, buf.data() + done, n * ts - done, (off_t)(at + (uint64_t)(first + i) * ts + done));
        if (w <= 0) return false;
        done += (size_t)w;
      }
      i += n;
    }
    return true;
  };
  if (std::fflush(f->fp) != 0) return ecckd::fail(ECCKD_PROCESSING_ERROR, "%s: flush failed", f->path.c_str());
  const int fd = ::fileno(f->fp);
  const size_t nthreads = std::min<size_t>(8, count / (2 * piece));
  bool ok = true;
  if (nthreads < 2) {
    ok = write_range(0, count, fd);
  } else {
    std::vector<std::thread> pool;
    std::vector<char> good(nthreads, 1);
    const size_t share = (count + nthreads - 1) / nthreads;
    for (size_t t = 0; t < nthreads; ++t) {
      const size_t first = t * share, n_values = first < count ? std::min(share, count - first) : 0;
      pool.emplace_back([&, t, first, n_values] { if (n_values && !write_range(first, n_values, fd)) good[t] = 0; });
    }
    for (auto& th : pool) th.join();
    for (char gd : good) ok = ok && gd;
  }
  if (!ok) return ecckd::fail(ECCKD_PROCESSING_ERROR, "%s: short write of \"%s\"", f->path.c_str(), name);
  return ECCKD_OK;
}
}  // namespace

extern "C" {

int ecckd_nc_write_double(ecckd_nc* f, const char* name, const double* data, size_t count) {
  ECCKD_REQUIRE(f && f->writing && !f->defining && name && data, "ecckd_nc_write_double: bad argument or still in define mode");
  const Var* v = f->find(name);
  if (!v) return ecckd::fail(ECCKD_PARAMETER_ERROR, "%s: no variable \"%s\"", f->path.c_str(), name);
  uint64_t nel = 1;
  for (int id : v->dimids) nel *= f->dims[id].len;
  ECCKD_REQUIRE(count == nel, "ecckd_nc_write_double: \"%s\" has %llu elements, %zu given", name, (unsigned long long)nel, count);
  if (f->netcdf4) return ecckd::h5w_write(f->h5w, (int)(v - f->vars.data()), -1, data, count);
  return write_values(f, *v, name, v->begin, data, count);
}

// one index of the slowest dimension of a (fixed-size) variable: what lets a tool write a (column, level, wavenumber)
// matrix column by column without holding all columns
int ecckd_nc_write_slice_double(ecckd_nc* f, const char* name, size_t slice, const double* data, size_t count) {
  ECCKD_REQUIRE(f && f->writing && !f->defining && name && data, "ecckd_nc_write_slice_double: bad argument or still in define mode");
  const Var* v = f->find(name);
  if (!v) return ecckd::fail(ECCKD_PARAMETER_ERROR, "%s: no variable \"%s\"", f->path.c_str(), name);
  ECCKD_REQUIRE(!v->dimids.empty() && !v->record, "ecckd_nc_write_slice_double: \"%s\" is a scalar or a record variable", name);
  uint64_t per = 1;
  for (size_t k = 1; k < v->dimids.size(); ++k) per *= f->dims[v->dimids[k]].len;
  ECCKD_REQUIRE(slice < f->dims[v->dimids[0]].len && count == per, "ecckd_nc_write_slice_double: \"%s\" slice %zu / %zu values do not fit",
                name, slice, count);
  if (f->netcdf4) return ecckd::h5w_write(f->h5w, (int)(v - f->vars.data()), (long long)slice, data, count);
  return write_values(f, *v, name, v->begin + (uint64_t)slice * per * type_size(v->type), data, count);
}


// write_order (write_order.cpp:24-143): the reordering file that find_g_points and create_look_up_table read.
// Same variable names, external types, long_name / units / comment attributes and global attributes
// (title, molecule, history, config); `deflate_variable` has no counterpart in the classic format.
int ecckd_write_order_file(const char* path, const char* molecule, const char* config_str, const char* history, int nband,
                           const double* band_bound1, const double* band_bound2, size_t nwav, const double* wavenumber,
                           const double* d_wavenumber, const int16_t* iband, const int32_t* rank,
                           const double* column_optical_depth, const double* sorting_variable) {
  ECCKD_REQUIRE(path && nband > 0 && band_bound1 && band_bound2 && nwav > 0 && wavenumber && d_wavenumber && iband && rank &&
                sorting_variable, "ecckd_write_order_file: NULL argument");
  ecckd_nc* f = nullptr;
  ECCKD_CHECK(ecckd_nc_create(path, &f));
  int rc = ECCKD_OK;
#define NCTRY(x) do { if (rc == ECCKD_OK) rc = (x); } while (0)
  int d_band = 0, d_wav = 0;
  NCTRY(ecckd_nc_def_dim(f, "band", (size_t)nband, &d_band));
  NCTRY(ecckd_nc_def_dim(f, "wavenumber", nwav, &d_wav));
  auto var = [&](const char* name, int type, int dim, const char* long_name, const char* units, const char* comment) {
    NCTRY(ecckd_nc_def_var(f, name, type, 1, &dim, nullptr));
    NCTRY(ecckd_nc_put_att_text(f, name, "long_name", long_name));
    if (units) NCTRY(ecckd_nc_put_att_text(f, name, "units", units));
    if (comment) NCTRY(ecckd_nc_put_att_text(f, name, "comment", comment));
  };
  var("wavenumber1_band", NC_FLOAT, d_band, "Lower wavenumber bound of band", "cm-1", nullptr);
  var("wavenumber2_band", NC_FLOAT, d_band, "Upper wavenumber bound of band", "cm-1", nullptr);
  var("wavenumber", NC_DOUBLE, d_wav, "Wavenumber", "cm-1", nullptr);
  var("d_wavenumber", NC_FLOAT, d_wav, "Wavenumber interval", "cm-1", nullptr);
  var("band_number", NC_SHORT, d_wav, "Band number", nullptr,
      "This variable indicates the number of the band (0 based) that each wavenumber is in, with -1 indicating a wavenumber not considered.");
  var("rank", NC_INT, d_wav, "Rank when reordered", nullptr,
      "This variable indicates the place of each wavenumber after reordering, with 0 indicating the least optically thick.\n"
      "rank(i) provides the rank of wavenumber i.");
  if (column_optical_depth) var("column_optical_depth", NC_FLOAT, d_wav, "Column optical depth", nullptr, nullptr);
  const bool is_cloud = molecule && std::strcmp(molecule, "cloud") == 0;
  var("sorting_variable", NC_FLOAT, d_wav, "Variable used to sort spectrum", nullptr,
      is_cloud ? "This variable is equal to the approximate cloud absorptance in the optically thick limit."
               : "This variable is equal to log(surface pressure) minus log(pressure of peak heating/cooling),\n"
                 "but for column optical depths less than a threshold, set to column optical depth minus the threshold.");
  if (molecule && molecule[0]) {
    std::string upper(molecule);
    for (char& c : upper) c = (char)std::toupper((unsigned char)c);
    NCTRY(ecckd_nc_put_att_text(f, nullptr, "title", ("Optimal reordering of the absorption spectrum of " + upper).c_str()));
    NCTRY(ecckd_nc_put_att_text(f, nullptr, "molecule", molecule));
  } else {
    NCTRY(ecckd_nc_put_att_text(f, nullptr, "title", "Optimal reordering of the absorption spectrum of a gas"));
  }
  if (history) NCTRY(ecckd_nc_put_att_text(f, nullptr, "history", history));
  NCTRY(ecckd_nc_put_att_text(f, nullptr, "config", config_str ? config_str : ""));
  // write_order.cpp:59-94 deflates every per-wavenumber variable (a NetCDF-4 file only)
  for (const char* v : {"wavenumber", "d_wavenumber", "band_number", "rank", "sorting_variable"}) NCTRY(ecckd_nc_deflate_var(f, v));
  if (column_optical_depth) NCTRY(ecckd_nc_deflate_var(f, "column_optical_depth"));
  NCTRY(ecckd_nc_enddef(f));
  NCTRY(ecckd_nc_write_double(f, "wavenumber1_band", band_bound1, (size_t)nband));
  NCTRY(ecckd_nc_write_double(f, "wavenumber2_band", band_bound2, (size_t)nband));
  NCTRY(ecckd_nc_write_double(f, "wavenumber", wavenumber, nwav));
  NCTRY(ecckd_nc_write_double(f, "d_wavenumber", d_wavenumber, nwav));
  {
    std::vector<double> tmp(nwav);
    for (size_t j = 0; j < nwav; ++j) tmp[j] = (double)iband[j];
    NCTRY(ecckd_nc_write_double(f, "band_number", tmp.data(), nwav));
    for (size_t j = 0; j < nwav; ++j) tmp[j] = (double)rank[j];
    NCTRY(ecckd_nc_write_double(f, "rank", tmp.data(), nwav));
  }
  if (column_optical_depth) NCTRY(ecckd_nc_write_double(f, "column_optical_depth", column_optical_depth, nwav));
  NCTRY(ecckd_nc_write_double(f, "sorting_variable", sorting_variable, nwav));
#undef NCTRY
  if (rc != ECCKD_OK) { f->defining = false; (void)ecckd_nc_close(f); return rc; }
  return ecckd_nc_close(f);
}

}  // extern "C"
