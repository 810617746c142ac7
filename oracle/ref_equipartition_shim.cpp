// ref_equipartition_shim.cpp - C-callable wrapper around the REFERENCE's own
// Equipartition class (TEST INFRASTRUCTURE).  This file is ours; it is compiled
// together with /root/reference/src/ecckd/equipartition.cpp (read where it
// lies, never copied) into oracle/_ref/libequipartition_ref.so by
// oracle/Makefile.  The result pins the partition search: the
// product's own host-side search must make the same calc_error calls and
// return the same bounds.
#include <vector>
#include <cstring>
#include <cmath>
#include <iostream>
#include "equipartition.h"  // -I/root/reference/src/ecckd

typedef double (*refep_calc_error_cb)(double bound1, double bound2, void* user);

namespace {
class CallbackEquipartition : public Equipartition {
public:
  CallbackEquipartition(refep_calc_error_cb cb, void* user) : cb_(cb), user_(user) {}
  ep_real calc_error(ep_real bound1, ep_real bound2) { return cb_(bound1, bound2, user_); }
private:
  refep_calc_error_cb cb_;
  void* user_;
};
}

extern "C" {

void* refep_create(refep_calc_error_cb cb, void* user) {
  return new CallbackEquipartition(cb, user);
}
void refep_destroy(void* h) { delete static_cast<CallbackEquipartition*>(h); }

void refep_set_verbose(void* h, int v) { static_cast<CallbackEquipartition*>(h)->set_verbose(v); }
void refep_set_partition_max_iterations(void* h, int n) {
  static_cast<CallbackEquipartition*>(h)->set_partition_max_iterations(n);
}
void refep_set_line_search_max_iterations(void* h, int n) {
  static_cast<CallbackEquipartition*>(h)->set_line_search_max_iterations(n);
}
void refep_set_partition_tolerance(void* h, double t) {
  static_cast<CallbackEquipartition*>(h)->set_partition_tolerance(t);
}
void refep_set_cubic_interpolation(void* h, int c) {
  static_cast<CallbackEquipartition*>(h)->set_cubic_interpolation(c != 0);
}
void refep_set_resolution(void* h, double r) { static_cast<CallbackEquipartition*>(h)->set_resolution(r); }
void refep_set_parallel(void* h, int p) { static_cast<CallbackEquipartition*>(h)->set_parallel(p != 0); }
void refep_set_minimize_frac_range(void* h, int m) {
  static_cast<CallbackEquipartition*>(h)->set_minimize_frac_range(m != 0);
}

int refep_equipartition_n(void* h, int ni, double* bounds, double* error) {
  return static_cast<int>(static_cast<CallbackEquipartition*>(h)->equipartition_n(ni, bounds, error));
}

// bounds_out has room for cap+1 values, error_out for cap; returns status, or
// -1 if the answer does not fit.
int refep_equipartition_e(void* h, double target_error, double bound0, double boundn,
                          int* ni, double* bounds_out, double* error_out, int cap) {
  std::vector<ep_real> bounds, error;
  int n = *ni;
  EpStatus st = static_cast<CallbackEquipartition*>(h)->equipartition_e(target_error, bound0, boundn,
                                                                        n, bounds, error);
  *ni = n;
  if (st == EP_INPUT_ERROR) return static_cast<int>(st);
  if (n > cap) return -1;
  std::memcpy(bounds_out, bounds.data(), sizeof(double) * (n + 1));
  std::memcpy(error_out, error.data(), sizeof(double) * n);
  return static_cast<int>(st);
}

const char* refep_status_string(int status) { return ep_status_string(static_cast<EpStatus>(status)); }

}  // extern "C"
