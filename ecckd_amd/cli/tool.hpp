// Shared host code of the command-line tools (reorder_spectrum, find_g_points, ...).
//
// The tools are the process-level boundary of the reference: `exe [key=value ...] [file.cfg]`,
// NetCDF files in and out, exit code 0 or one of src/include/EsaExitCodes.h.  Everything here sits ABOVE the
// C ABI of include/ecckd_hip.h - the tools never touch HIP themselves.
#pragma once
#include <algorithm>
#include <climits>
#include <map>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <string>
#include <functional>
#include <vector>

#include "../../include/ecckd_hip.h"

namespace tool {

struct Fatal {
  int code;
  std::string msg;
};

[[noreturn]] inline void fail(int code, const char* fmt, ...) {
  char buf[2048];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  throw Fatal{code, buf};
}

inline void ck(int rc) {
  if (rc != ECCKD_OK) {
    const char* m = ecckd_last_error();
    throw Fatal{rc, m ? m : ""};
  }
}

// ---- logging (src/include/Logging.h: LOG goes to stdout, errors to stderr) ----
inline int& log_level() { static int lvl = 2; return lvl; }
inline void set_log_level(const std::string& s) {
  if (s == "error" || s == "0") log_level() = 0;
  else if (s == "warning" || s == "1") log_level() = 1;
  else if (s == "info" || s == "2") log_level() = 2;
  else log_level() = 3;
}
// ECCKD_LOG_TIMES=1: every LOG line starts with the seconds since the tool started (where does the wall time go?)
inline double seconds_since_start() {
  static const std::clock_t unused = std::clock();
  (void)unused;
  static const auto t0 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
inline bool log_times() { static const bool on = std::getenv("ECCKD_LOG_TIMES") != nullptr; return on; }
#define LOG(...) do { if (tool::log_level() >= 2) { if (tool::log_times()) std::printf("[%8.3f] ", tool::seconds_since_start()); \
                                                    std::printf(__VA_ARGS__); std::fflush(stdout); } } while (0)
#define WARN(...) do { if (tool::log_level() >= 1) { std::fprintf(stderr, "*** Warning: "); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } } while (0)

// `return done(0);` at the end of a tool's work: the device side of the process (context, streams, pinned buffers, device
// memory) is then left to the operating system instead of being taken down piece by piece, and run() ends the process without
// the HIP runtime's own exit handlers - 30-50 ms of a process whose fixed cost is ~0.3 s (tools/startup_probe.py); the files
// are closed by their destructors as always.  ECCKD_NO_FAST_EXIT keeps the orderly teardown.
inline bool& leaving() { static bool v = false; return v; }
inline int done(int rc) {
  static const bool orderly = std::getenv("ECCKD_NO_FAST_EXIT") != nullptr;
  leaving() = !orderly;
  if (log_times() && log_level() >= 2) { std::printf("[%8.3f] Output written\n", seconds_since_start()); std::fflush(stdout); }
  return rc;
}

// ---- configuration (DataFile config(argc, argv)) ----
class Config {
 public:
  Config(int argc, const char* const* argv) { ck(ecckd_cfg_from_args(argc, argv, &c_)); }
  ~Config() { ecckd_cfg_destroy(c_); }
  Config(const Config&) = delete;
  Config& operator=(const Config&) = delete;

  bool exist(const std::string& name, const char* scope = nullptr) const {
    int e = 0;
    ck(ecckd_cfg_exists(c_, scope, name.c_str(), &e));
    return e != 0;
  }
  // DataFile::read semantics: the variable is left untouched and false is returned when the key is absent
  bool read(int& x, const std::string& name, const char* scope = nullptr) const {
    int f = 0;
    ck(ecckd_cfg_get_int(c_, scope, name.c_str(), &x, &f));
    return f != 0;
  }
  bool read(double& x, const std::string& name, const char* scope = nullptr) const {
    int f = 0;
    ck(ecckd_cfg_get_real(c_, scope, name.c_str(), &x, &f));
    return f != 0;
  }
  bool read(bool& x, const std::string& name, const char* scope = nullptr) const {
    int v = 0;
    ck(ecckd_cfg_get_boolean(c_, scope, name.c_str(), &v));
    x = v != 0;
    return true;
  }
  bool read(std::string& s, const std::string& name, const char* scope = nullptr, int index = -1) const {
    size_t len = 0;
    int f = 0;
    ck(ecckd_cfg_get_string(c_, scope, name.c_str(), index, nullptr, 0, &len, &f));
    if (!f) return false;
    std::vector<char> buf(len + 1);
    ck(ecckd_cfg_get_string(c_, scope, name.c_str(), index, buf.data(), buf.size(), &len, &f));
    s.assign(buf.data(), len);
    return true;
  }
  // the idx-th element of a numeric list (DataFile::read(Real&, name, j): rc_assign_real_element)
  bool read_element(double& x, const std::string& name, int idx, const char* scope = nullptr) const {
    std::string s;
    if (!read(s, name, scope, idx)) return false;
    char* end = nullptr;
    const double v = std::strtod(s.c_str(), &end);
    if (end == s.c_str()) return false;
    x = v;
    return true;
  }
  bool read(std::vector<double>& v, const std::string& name, const char* scope = nullptr) const {
    int n = 0;
    ck(ecckd_cfg_get_real_vector(c_, scope, name.c_str(), nullptr, 0, &n));
    if (n == 0) return false;
    v.resize(n);
    ck(ecckd_cfg_get_real_vector(c_, scope, name.c_str(), v.data(), n, &n));
    return true;
  }
  bool read(std::vector<int>& v, const std::string& name, const char* scope = nullptr) const {
    int n = 0;
    ck(ecckd_cfg_get_int_vector(c_, scope, name.c_str(), nullptr, 0, &n));
    if (n == 0) return false;
    v.resize(n);
    ck(ecckd_cfg_get_int_vector(c_, scope, name.c_str(), v.data(), n, &n));
    return true;
  }
  int count(const std::string& name, const char* scope = nullptr) const {
    int c = 0;
    ck(ecckd_cfg_size(c_, scope, name.c_str(), &c, nullptr, nullptr));
    return c;
  }
  std::vector<std::string> read_list(const std::string& name, const char* scope = nullptr) const {
    std::vector<std::string> out;
    std::string s;
    for (int i = 0; read(s, name, scope, i); ++i) out.push_back(s);
    return out;
  }
  // config.read(config_str): the whole configuration as stored in the `config` attribute
  std::string str() const {
    size_t len = 0;
    ck(ecckd_cfg_sprint(c_, nullptr, 0, &len));
    std::vector<char> buf(len + 1);
    ck(ecckd_cfg_sprint(c_, buf.data(), buf.size(), &len));
    return std::string(buf.data(), len);
  }

 private:
  ecckd_cfg* c_ = nullptr;
};

// ---- file search path (src/tools/file_manager.cpp:21-123) ----
class SearchPath {
 public:
  SearchPath() {
    if (const char* p = std::getenv("OPTSYN_PATH")) split(p, dirs_, false);
    else dirs_.push_front(".");
  }
  void append(const std::string& path) { split(path, dirs_, false); }
  void prepend(const std::string& path) {
    std::deque<std::string> tmp;
    split(path, tmp, false);
    for (auto it = tmp.rbegin(); it != tmp.rend(); ++it) dirs_.push_front(*it);
  }
  void configure(const Config& config) {
    std::string p;
    if (config.read(p, "prepend_path")) prepend(p);
    if (config.read(p, "append_path")) append(p);
  }
  // the full path of an existing file; throws CANNOT_OPEN_MANDATORY_FILE otherwise
  std::string find(const std::string& name) const {
    if (!name.empty() && name[0] == '/') {
      if (readable(name)) return name;
    } else {
      for (const std::string& d : dirs_)
        if (readable(d + "/" + name)) return d + "/" + name;
    }
    fail(ECCKD_CANNOT_OPEN_MANDATORY_FILE, "Cannot find \"%s\" in the search path", name.c_str());
  }

 private:
  static bool readable(const std::string& p) {
    FILE* f = std::fopen(p.c_str(), "r");
    if (!f) return false;
    std::fclose(f);
    return true;
  }
  static void split(std::string path, std::deque<std::string>& out, bool) {
    while (!path.empty()) {
      const size_t end = path.find(':');
      if (end == std::string::npos) { out.push_back(path); break; }
      if (end > 0) out.push_back(path.substr(0, end));
      path = path.substr(end + 1);
    }
  }
  std::deque<std::string> dirs_;
};

// ---- NetCDF (classic) files ----
class NcIn {
 public:
  explicit NcIn(const std::string& path) : path_(path) { ck(ecckd_nc_open(path.c_str(), &f_)); }
  ~NcIn() { if (f_) ecckd_nc_close(f_); }
  NcIn(const NcIn&) = delete;
  NcIn& operator=(const NcIn&) = delete;

  bool exist(const std::string& var, std::vector<size_t>* shape = nullptr, int* type = nullptr) const {
    int e = 0, t = 0, nd = 0;
    size_t sh[8] = {};
    ck(ecckd_nc_inq_var(f_, var.c_str(), &e, &t, &nd, sh, 8));
    if (e && shape) shape->assign(sh, sh + nd);
    if (e && type) *type = t;
    return e != 0;
  }
  std::vector<size_t> shape(const std::string& var) const {
    std::vector<size_t> s;
    if (!exist(var, &s)) fail(ECCKD_PARAMETER_ERROR, "Variable \"%s\" not found in %s", var.c_str(), path_.c_str());
    return s;
  }
  // the whole variable (slice < 0) or one index of its slowest dimension
  std::vector<double> read(const std::string& var, long long slice = -1) const {
    std::vector<size_t> s = shape(var);
    size_t n = 1;
    for (size_t k = (slice >= 0 ? 1 : 0); k < s.size(); ++k) n *= s[k];
    std::vector<double> out(n);
    ck(ecckd_nc_read_double(f_, var.c_str(), slice, out.data(), n));
    return out;
  }
  double read_scalar(const std::string& var) const { return read(var).at(0); }
  bool att_text(const std::string& att, std::string& out, const char* var = nullptr) const {
    int e = 0;
    std::vector<char> buf(1 << 20);
    ck(ecckd_nc_read_att_text(f_, var, att.c_str(), &e, buf.data(), buf.size()));
    if (!e) return false;
    out = buf.data();
    return true;
  }
  const std::string& path() const { return path_; }
  ecckd_nc* handle() const { return f_; }

 private:
  std::string path_;
  ecckd_nc* f_ = nullptr;
};

enum NcType { NC_BYTE_T = 1, NC_CHAR_T = 2, NC_SHORT_T = 3, NC_INT_T = 4, NC_FLOAT_T = 5, NC_DOUBLE_T = 6 };

class NcOut {
 public:
  explicit NcOut(const std::string& path) { ck(ecckd_nc_create(path.c_str(), &f_)); }
  ~NcOut() { if (f_) ecckd_nc_close(f_); }
  NcOut(const NcOut&) = delete;
  NcOut& operator=(const NcOut&) = delete;
  void dim(const std::string& name, size_t len) {
    int id = 0;
    ck(ecckd_nc_def_dim(f_, name.c_str(), len, &id));
    dims_.push_back({name, id});
  }
  void var(const std::string& name, int type, const std::vector<std::string>& dims = {}, const char* long_name = nullptr,
           const char* units = nullptr) {
    std::vector<int> ids;
    for (const std::string& d : dims) {
      int id = -1;
      for (auto& p : dims_) if (p.first == d) id = p.second;
      if (id < 0) fail(ECCKD_PARAMETER_ERROR, "Dimension \"%s\" not defined", d.c_str());
      ids.push_back(id);
    }
    int vid = 0;
    ck(ecckd_nc_def_var(f_, name.c_str(), type, (int)ids.size(), ids.data(), &vid));
    if (long_name) att(long_name, "long_name", name.c_str());
    if (units) att(units, "units", name.c_str());
  }
  void att(const std::string& text, const std::string& name, const char* var = nullptr) {
    ck(ecckd_nc_put_att_text(f_, var, name.c_str(), text.c_str()));
  }
  // OutputDataFile::deflate_variable (:345-359): shuffle + deflate level 2 where the file is NetCDF-4 (*.h5 / *.hdf), nothing otherwise
  void deflate(const std::string& name) { ck(ecckd_nc_deflate_var(f_, name.c_str())); }
  void end_define() { ck(ecckd_nc_enddef(f_)); }
  void write(const std::string& name, const std::vector<double>& v) { ck(ecckd_nc_write_double(f_, name.c_str(), v.data(), v.size())); }
  void write_slice(const std::string& name, size_t slice, const std::vector<double>& v) {
    ck(ecckd_nc_write_slice_double(f_, name.c_str(), slice, v.data(), v.size()));
  }
  template <class T>
  void write_as_double(const std::string& name, const std::vector<T>& v) {
    std::vector<double> d(v.begin(), v.end());
    write(name, d);
  }
  void close() { if (f_) { ck(ecckd_nc_close(f_)); f_ = nullptr; } }

 private:
  ecckd_nc* f_ = nullptr;
  std::vector<std::pair<std::string, int>> dims_;
};

// "<date>: <command line>" - the line OutputDataFile::append_history adds (OutputDataFile.cpp:1009-1048)
inline std::string history_line(int argc, const char* const* argv) {
  char stamp[64];
  const std::time_t t = std::time(nullptr);
  std::strftime(stamp, sizeof stamp, "%a %b %e %H:%M:%S %Y", std::gmtime(&t));
  std::string s = std::string(stamp) + ":";
  for (int i = 0; i < argc; ++i) { s += " "; s += argv[i]; }
  return s;
}

// ---- one column of a CKDMIP spectral file (read_spectrum.cpp:20-87) ----
struct Spectrum {
  int ncol = 0, nlay = 0;
  size_t nwav = 0;
  std::vector<double> pressure_hl, temperature_hl, wavenumber_cm_1, d_wavenumber_cm_1, vmr_fl;
  double reference_surface_vmr = -1.0;
  std::string molecule;
  std::vector<double> optical_depth;   // [nlay][nwav]; empty if not requested
  bool od_is_float = false;            // stored as FLOAT in the file: can be shipped to the device as f32 without loss
};

inline void read_od_meta(const NcIn& f, int iprofile, int nlay, double& reference_surface_vmr, std::vector<double>& vmr_fl,
                         std::string& molecule) {
  reference_surface_vmr = f.exist("reference_surface_mole_fraction") ? f.read_scalar("reference_surface_mole_fraction") : -1.0;
  std::vector<size_t> sh;
  if (f.exist("mole_fraction_fl", &sh) && sh.size() == 2) vmr_fl = f.read("mole_fraction_fl", iprofile);
  else vmr_fl.assign(nlay, -1.0);
  molecule.clear();
  if (!f.att_text("constituent_id", molecule)) f.att_text("molecules", molecule);
}

inline Spectrum read_spectrum(const std::string& path, int iprofile, bool want_od = true) {
  NcIn f(path);
  Spectrum s;
  s.ncol = (int)f.shape("pressure_hl").at(0);
  s.pressure_hl = f.read("pressure_hl", iprofile);
  s.nlay = (int)s.pressure_hl.size() - 1;
  if (f.exist("temperature_hl")) s.temperature_hl = f.read("temperature_hl", iprofile);
  s.wavenumber_cm_1 = f.read("wavenumber");
  s.nwav = s.wavenumber_cm_1.size();
  if (f.exist("d_wavenumber")) {
    s.d_wavenumber_cm_1 = f.read("d_wavenumber");
  } else {   // :55-65
    const size_t n = s.nwav;
    s.d_wavenumber_cm_1.assign(n, 0.0);
    for (size_t i = 1; i + 1 < n; ++i) s.d_wavenumber_cm_1[i] = 0.5 * (s.wavenumber_cm_1[i + 1] - s.wavenumber_cm_1[i - 1]);
    if (n > 2) { s.d_wavenumber_cm_1[0] = 0.5 * s.d_wavenumber_cm_1[1]; s.d_wavenumber_cm_1[n - 1] = 0.5 * s.d_wavenumber_cm_1[n - 2]; }
  }
  read_od_meta(f, iprofile, s.nlay, s.reference_surface_vmr, s.vmr_fl, s.molecule);
  if (want_od) {
    int type = 0;
    f.exist("optical_depth", nullptr, &type);
    s.od_is_float = type == NC_FLOAT_T;
    s.optical_depth = f.read("optical_depth", iprofile);
    if (s.optical_depth.size() != (size_t)s.nlay * s.nwav)
      fail(ECCKD_PARAMETER_ERROR, "optical_depth in %s is not (column, level, wavenumber)", path.c_str());
  }
  return s;
}

// ---- device ----
class Device {
 public:
  Device() {
    // one process per GPU: ECCKD_DEVICE, else the LOCAL_RANK a launcher (torchrun --no-python, mpirun wrappers) hands out
    int dev = 0;
    if (const char* e = std::getenv("ECCKD_DEVICE")) dev = std::atoi(e);
    else if (const char* l = std::getenv("LOCAL_RANK")) dev = std::atoi(l);
    ck(ecckd_init(dev, &ctx_));
  }
  ~Device() {
    if (leaving()) {                       // the work is done: let every queued operation finish, leave the rest to the exit
      if (ctx_) (void)ecckd_synchronize(ctx_);
      return;
    }
    for (auto& kv : od_cache_) ecckd_dev_free(ctx_, kv.second.ptr);
    if (ctx_) ecckd_destroy(ctx_);
  }
  Device(const Device&) = delete;
  Device& operator=(const Device&) = delete;
  ecckd_ctx* ctx() const { return ctx_; }

  // Optical-depth slices that stay on the device for the life of the process, keyed by file and profile: find_g_points reads
  // the spectrum of every gas once as the target and once more in the background of every other gas (the reference reads
  // the files again each time, read_merged_spectrum.cpp:63-100).  ECCKD_OD_CACHE_GB: the budget (default 64, 0 = off);
  // slices beyond it are read as before and owned by their reader.
  struct CachedOd { void* ptr; size_t bytes; int type; };
  const CachedOd* od_cache_find(const std::string& key) const {
    auto it = od_cache_.find(key);
    return it == od_cache_.end() ? nullptr : &it->second;
  }
  void enable_od_cache() { od_cache_on_ = true; }     // for a tool that reads the same slices again (find_g_points)
  // Room for one more slice in the cache?  The budget is ECCKD_OD_CACHE_GB (default 64) per process, and never more than
  // what leaves a quarter of the device's memory - as free as the driver reports it NOW, so that several processes sharing a
  // GPU see each other - to the gas preparations (a prepared gas holds ~4x the bytes of its FLOAT slice).
  bool od_cache_room(size_t bytes) const {
    if (!od_cache_on_) return false;
    double gb = 64.0;
    if (const char* e = std::getenv("ECCKD_OD_CACHE_GB")) gb = std::atof(e);
    if ((double)(od_cache_bytes_ + bytes) > gb * 1073741824.0) return false;
    size_t free_b = 0, total_b = 0;
    if (ecckd_mem_info(ctx_, &free_b, &total_b) != ECCKD_OK) return false;
    return (double)bytes <= (double)free_b - 0.25 * (double)total_b;
  }
  void od_cache_put(const std::string& key, void* ptr, size_t bytes, int type) const {
    od_cache_[key] = CachedOd{ptr, bytes, type};
    od_cache_bytes_ += bytes;
  }

 private:
  ecckd_ctx* ctx_ = nullptr;
  mutable std::map<std::string, CachedOd> od_cache_;
  mutable size_t od_cache_bytes_ = 0;
  bool od_cache_on_ = false;
};

class DevBuf {
 public:
  DevBuf() = default;
  DevBuf(const Device& d, size_t bytes) { alloc(d, bytes); }
  ~DevBuf() { if (!leaving()) release(); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : ctx_(o.ctx_), p_(o.p_), bytes_(o.bytes_), owned_(o.owned_) { o.p_ = nullptr; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); ctx_ = o.ctx_; p_ = o.p_; bytes_ = o.bytes_; owned_ = o.owned_; o.p_ = nullptr; }
    return *this;
  }
  // a view of memory that something else owns (the Device's cache of optical-depth slices): release() only forgets it
  static DevBuf view(ecckd_ctx* ctx, void* p, size_t bytes) {
    DevBuf b;
    b.ctx_ = ctx; b.p_ = p; b.bytes_ = bytes; b.owned_ = false;
    return b;
  }
  void alloc(const Device& d, size_t bytes) {
    release();
    ctx_ = d.ctx();
    bytes_ = bytes;
    owned_ = true;
    ck(ecckd_dev_alloc(ctx_, bytes ? bytes : 8, &p_));
  }
  void release() { if (p_ && owned_) ecckd_dev_free(ctx_, p_); p_ = nullptr; owned_ = true; }
  size_t bytes() const { return bytes_; }
  template <class T> void upload(const Device& d, const std::vector<T>& v) {
    alloc(d, v.size() * sizeof(T));
    if (!v.empty()) ck(ecckd_h2d(ctx_, p_, v.data(), v.size() * sizeof(T)));
  }
  template <class T> std::vector<T> download() const {
    std::vector<T> v(bytes_ / sizeof(T));
    if (!v.empty()) ck(ecckd_d2h(ctx_, v.data(), p_, bytes_));
    return v;
  }
  void* ptr() const { return p_; }
  template <class T> T* as() const { return static_cast<T*>(p_); }
  bool empty() const { return p_ == nullptr; }

 private:
  ecckd_ctx* ctx_ = nullptr;
  void* p_ = nullptr;
  size_t bytes_ = 0;
  bool owned_ = true;
};

// optical depth to the device: FLOAT as stored when the file holds FLOAT, DOUBLE otherwise
struct DevOd {
  DevBuf buf;
  int type = ECCKD_F64;
};
inline DevOd upload_od(const Device& d, const std::vector<double>& od, bool as_float) {
  DevOd out;
  if (as_float) {
    std::vector<float> f(od.begin(), od.end());
    out.buf.upload(d, f);
    out.type = ECCKD_F32;
  } else {
    out.buf.upload(d, od);
  }
  return out;
}

// optical_depth of one profile straight from the file into device memory (ecckd_nc_read_dev: the bytes are streamed through
// pinned buffers and decoded on the device; FLOAT stays FLOAT), instead of file -> doubles on the host -> floats -> upload
inline DevOd read_od_dev(const Device& d, const NcIn& f, int iprofile, int nlay, size_t nwav) {
  std::vector<size_t> sh;
  int type = 0;
  if (!f.exist("optical_depth", &sh, &type)) fail(ECCKD_PARAMETER_ERROR, "Variable \"optical_depth\" not found in %s", f.path().c_str());
  size_t n = 1;
  for (size_t k = 1; k < sh.size(); ++k) n *= sh[k];
  if (sh.size() != 3 || n != (size_t)nlay * nwav)
    fail(ECCKD_PARAMETER_ERROR, "optical_depth in %s is not (column, level, wavenumber)", f.path().c_str());
  DevOd out;
  out.type = type == NC_FLOAT_T ? ECCKD_F32 : ECCKD_F64;
  const size_t bytes = n * (size_t)out.type;
  char resolved[PATH_MAX];
  const std::string key = std::string(realpath(f.path().c_str(), resolved) ? resolved : f.path().c_str()) + "#" + std::to_string(iprofile);
  if (const Device::CachedOd* hit = d.od_cache_find(key)) {
    if (hit->bytes == bytes && hit->type == out.type) {
      LOG("    (optical depths already on the device)\n");
      out.buf = DevBuf::view(d.ctx(), hit->ptr, bytes);
      return out;
    }
  }
  if (d.od_cache_room(bytes)) {
    void* p = nullptr;
    if (ecckd_dev_alloc(d.ctx(), bytes, &p) == ECCKD_OK) {     // no room after all: read into a buffer the caller owns, below
      const int rc = ecckd_nc_read_dev(d.ctx(), f.handle(), "optical_depth", iprofile, out.type, p, n);
      if (rc != ECCKD_OK) { ecckd_dev_free(d.ctx(), p); ck(rc); }
      d.od_cache_put(key, p, bytes, out.type);
      out.buf = DevBuf::view(d.ctx(), p, bytes);
      return out;
    }
  }
  out.buf.alloc(d, bytes);
  ck(ecckd_nc_read_dev(d.ctx(), f.handle(), "optical_depth", iprofile, out.type, out.buf.ptr(), n));
  return out;
}

// ---- read_merged_spectrum (read_merged_spectrum.cpp:20-185): keys <prefix>input / scaling / conc ----
struct Merged {
  Spectrum first;            // grid, pressures, temperatures of the first file (optical_depth released)
  std::string molecules;
  DevBuf d_od;               // DOUBLE [nlay][nwav] unless `single` is set
  DevOd single;              // a single unscaled input is kept as stored
  bool is_single = false;
  const void* od_ptr() const { return is_single ? single.buf.ptr() : d_od.ptr(); }
  int od_type() const { return is_single ? single.type : ECCKD_F64; }
  std::vector<std::vector<double>> vmr_fl;   // one row per constituent (:153-165)
};

inline Merged read_merged_spectrum(const Device& dev, const Config& config, const SearchPath& paths, int iprofile,
                                   const std::string& prefix) {
  Merged m;
  const std::vector<std::string> files = config.read_list(prefix + "input");
  if (files.empty()) fail(ECCKD_PARAMETER_ERROR, "Unable to read input file names in %sinput", prefix.c_str());
  // a concentration file to scale the spectra to (:47-61): pressure_fl and <molecule>_mole_fraction_fl of one profile
  std::string conc_file_name;
  int iprof_conc = -1;
  std::vector<double> pressure_conc;
  std::string conc_path;
  if (config.read(conc_file_name, prefix + "conc_input")) {
    if (!config.read(iprof_conc, prefix + "iprofile"))
      fail(ECCKD_PARAMETER_ERROR, "Concentration file specified without profile number in \"iprofile\"");
    conc_path = paths.find(conc_file_name);
    pressure_conc = NcIn(conc_path).read("pressure_fl", iprof_conc);
  }
  for (size_t ibg = 0; ibg < files.size(); ++ibg) {
    double scaling = -1.0, conc = -1.0;
    config.read_element(scaling, prefix + "scaling", (int)ibg);
    config.read_element(conc, prefix + "conc", (int)ibg);
    const std::string path = paths.find(files[ibg]);
    LOG("  Reading %s\n", path.c_str());
    Spectrum s;
    DevOd od;
    if (ibg == 0) {
      s = read_spectrum(path, iprofile, false);
      m.molecules = s.molecule;
      od = read_od_dev(dev, NcIn(path), iprofile, s.nlay, s.nwav);
    } else {
      NcIn f(path);
      s.nlay = m.first.nlay;
      s.nwav = m.first.nwav;
      od = read_od_dev(dev, f, iprofile, s.nlay, s.nwav);
      read_od_meta(f, iprofile, s.nlay, s.reference_surface_vmr, s.vmr_fl, s.molecule);
      if (s.molecule.empty())
        fail(ECCKD_PARAMETER_ERROR, "Found neither \"constituent_id\" nor \"molecules\" amongst the global attributes");
      m.molecules += " " + s.molecule;
    }
    const std::vector<double>& p_hl = ibg == 0 ? s.pressure_hl : m.first.pressure_hl;
    std::vector<double> profile(s.nlay), vmr_out(s.nlay), conc_req;
    if (iprof_conc >= 0) {   // :104-116
      const std::string mol = s.molecule.substr(0, s.molecule.find(' '));
      conc_req = NcIn(conc_path).read(mol + "_mole_fraction_fl", iprof_conc);
      if (conc_req.size() != pressure_conc.size()) fail(ECCKD_PARAMETER_ERROR, "%s: %s_mole_fraction_fl does not match pressure_fl", conc_path.c_str(), mol.c_str());
    }
    ck(ecckd_merge_scaling(s.nlay, p_hl.data(), scaling, conc, s.reference_surface_vmr, s.vmr_fl.data(), (int)conc_req.size(),
                           conc_req.empty() ? nullptr : pressure_conc.data(), conc_req.empty() ? nullptr : conc_req.data(),
                           profile.data(), vmr_out.data()));
    bool unscaled = true;
    for (double v : profile) unscaled = unscaled && v == 1.0;
    if (iprof_conc >= 0) LOG("    Scaling to target concentration profile\n");
    else if (!unscaled) LOG("    Scaling by %g\n", profile[0]);
    m.vmr_fl.push_back(vmr_out);
    if (files.size() == 1 && unscaled) {
      m.single = std::move(od);
      m.is_single = true;
    } else {
      if (ibg == 0) m.d_od.alloc(dev, (size_t)s.nlay * s.nwav * sizeof(double));
      ck(ecckd_merge_spectrum_dev(dev.ctx(), s.nlay, s.nwav, od.buf.ptr(), od.type, s.nwav, profile.data(), ibg == 0 ? 1 : 0,
                                  m.d_od.as<double>(), s.nwav));
      ck(ecckd_synchronize(dev.ctx()));
    }
    if (ibg == 0) m.first = std::move(s);
  }
  return m;
}

// ---- several processes (one per GPU): RANK / WORLD_SIZE of the launcher; contiguous shares of a task table ----
inline int env_int(const char* name, int fallback) {
  const char* e = std::getenv(name);
  return e && *e ? std::atoi(e) : fallback;
}
// tasks [begin, end) of `rank` when `ntasks` are cut into `world` contiguous shares whose sizes differ by at most one
// (ecckd_amd/shard.py::deal_tasks)
inline void deal_tasks(int ntasks, int rank, int world, int& begin, int& end) {
  const int base = ntasks / world, extra = ntasks % world;
  begin = rank * base + std::min(rank, extra);
  end = begin + base + (rank < extra ? 1 : 0);
}

// what a tool wants done when it ends with an error (find_g_points under a launcher: leave a failure marker for process 0)
inline std::function<void(int)>& on_failure() { static std::function<void(int)> f; return f; }
inline void note_failure(int code) {
  if (on_failure()) {
    try { on_failure()(code); } catch (...) {}
  }
}
// FNV-1a over a byte string, chained: the run identity the part files of a several-process find_g_points carry
inline uint64_t fnv1a(const void* data, size_t n, uint64_t h = 0xcbf29ce484222325ull) {
  const unsigned char* p = static_cast<const unsigned char*>(data);
  for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
  return h;
}
inline uint64_t fnv1a(const std::string& s, uint64_t h = 0xcbf29ce484222325ull) { return fnv1a(s.data(), s.size() + 1, h); }

// A tool that fails may have searches of other gases in flight on their own threads and streams (find_g_points): the process ends
// at once with the reference's exit code, without unwinding statics under those threads.
inline int failed(int code) {
  std::fflush(stdout);
  std::fflush(stderr);
  std::_Exit(code);
  return code;
}

// ---- main wrapper: exit codes like THROW(code) (Logging.h:115-117) ----
template <class Body>
int run(int argc, char** argv, Body body) {
  try {
    Config config(argc, argv);
    std::string lvl;
    if (config.read(lvl, "log_level")) set_log_level(lvl);
    const int rc = body(config);
    if (leaving()) {
      std::fflush(stdout);
      std::fflush(stderr);
      std::_Exit(rc);
    }
    return rc;
  } catch (const Fatal& f) {
    std::fprintf(stderr, "*** Error: %s\n", f.msg.c_str());
    note_failure(f.code ? f.code : 1);
    return failed(f.code ? f.code : 1);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "*** Error: %s\n", e.what());
    note_failure(ECCKD_UNEXPECTED_EXCEPTION);
    return failed(ECCKD_UNEXPECTED_EXCEPTION);
  }
}

}  // namespace tool
