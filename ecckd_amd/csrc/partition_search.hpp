// partition_search.hpp - host-side equal-error partition search with a BATCHED
// error evaluator.
//
// Restates the algorithm of reference src/ecckd/equipartition.{h,cpp} (class
// Equipartition): the same sequence of interval-error evaluations and the same
// floating-point expressions for every new bound, so that, given the same error
// function, it returns the bounds the reference returns (pinned in
// tests/test_partition_search.py against the reference file compiled into
// oracle/_ref).  What differs is the evaluation interface: the reference calls a
// virtual calc_error(bound1, bound2) once per interval, from OpenMP threads
// (equipartition.h:95-116); here every group of independent evaluations is ONE
// call of `evaluate(n, b1[], b2[], err[])`, which the GPU path maps to one
// batched launch sequence (K5).
#pragma once

#include <functional>
#include <vector>

namespace ecckd {

// equipartition.h:32-40
enum PartitionStatus {
  PS_SUCCESS = 0,
  PS_MAX_ITERATIONS_REACHED,
  PS_FAILED_TO_CONVERGE,
  PS_RESOLUTION_LIMIT_REACHED,
  PS_NO_PROGRESS,
  PS_FAILURE,
  PS_INPUT_ERROR
};

const char* partition_status_string(int status);

class PartitionSearch {
public:
  // Evaluate n intervals [b1[i], b2[i]]; return non-zero to abort the search
  // (the reference's calc_error throws, find_g_points.cpp:298-313).
  using Evaluator = std::function<int(int n, const double* b1, const double* b2, double* err)>;

  explicit PartitionSearch(Evaluator ev) : evaluate_(std::move(ev)) {}

  // setters mirror equipartition.h:118-174
  void set_partition_max_iterations(int n) { partition_max_iterations_ = n; }
  void set_line_search_max_iterations(int n) { line_search_max_iterations_ = n; }
  void set_partition_tolerance(double t) { partition_tolerance_ = t; }
  void set_cubic_interpolation(bool c) { cubic_interpolation_ = c; }
  void set_resolution(double r) { resolution_ = r; }
  void set_minimize_frac_range(bool m) { minimize_frac_range_ = m; }

  // equipartition.cpp:348-566.  bounds[ni+1] in/out, error[ni] out.
  int equipartition_n(int ni, double* bounds, double* error);

  // equipartition.cpp:574-634.
  int equipartition_e(double target_error, double bound0, double boundn, int& ni,
                      std::vector<double>& bounds, std::vector<double>& error);

  // equipartition.h:98-116 (one batched call)
  int calc_error_all(int ni, const double* bounds, double* error);

  // equipartition.cpp:132-160
  double cost_function(int ni, const double* error) const;

  // non-zero once an evaluation has failed; the search then unwinds with PS_FAILURE
  int evaluator_status() const { return eval_status_; }

  // Decision trace (audits only; tests/test_decision_trace_gpu.py): every floating-point comparison that steers the search
  // is reported as (site, lhs, rhs, outcome) in the order it is taken.  The sites are numbered in partition_search.cpp.
  using Trace = std::function<void(int site, double lhs, double rhs, int taken)>;
  void set_trace(Trace t) { trace_ = std::move(t); }

private:
  double calc_error(double b1, double b2);
  double next_bound_above(double target_error, double bound1, double boundn, double bound2_test,
                          double* error_test);
  double next_bound_below(double target_error, double bound0, double bound2, double bound1_test,
                          double* error_test);
  int equipartition_2(double* bounds, double* error);
  int line_search(int ni, double* bounds, double* newbounds, double* error);

  bool decide(int site, double lhs, double rhs, bool taken) const {
    if (trace_) trace_(site, lhs, rhs, taken ? 1 : 0);
    return taken;
  }
  bool decide_lt(int site, double a, double b) const { return decide(site, a, b, a < b); }
  bool decide_gt(int site, double a, double b) const { return decide(site, a, b, a > b); }
  bool decide_eq(int site, double a, double b) const { return decide(site, a, b, a == b); }

  Evaluator evaluate_;
  Trace trace_;
  int eval_status_ = 0;
  // defaults of equipartition.h:191-205
  double next_bound_error_tolerance_ = 0.05;
  double partition_tolerance_ = 0.05;
  double resolution_ = 0.0;
  int next_bound_max_iterations_ = 20;
  int partition_max_iterations_ = 20;
  int line_search_max_iterations_ = 10;
  bool cubic_interpolation_ = false;
  bool minimize_frac_range_ = true;
  bool errors_up_to_date_ = false;
};

}  // namespace ecckd
