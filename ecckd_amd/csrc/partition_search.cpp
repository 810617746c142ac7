// partition_search.cpp - see partition_search.hpp.  Host-only code.
//
// Parity notes (all pinned against the reference file compiled into oracle/_ref):
//  * errors_up_to_date_ is carried across calls exactly as the reference's member
//    (equipartition.h:207), including its quirks: after a failed line search the
//    error array still holds the errors of the LAST trial bounds, and the first
//    pairwise step of a shuffle is the only one that re-evaluates
//    (equipartition.cpp:207-210, :499-531); the resolution-limit return leaves the
//    flag set (:479-492).
//  * next_bound_above tests `error_low > 2*max_error` and its un-bracketed
//    extrapolation is max(high, high - ...) (equipartition.cpp:779-789): kept.
//  * std::max(a, b) is (a < b) ? b : a, which matters when b is NaN (0/0 on the
//    first un-bracketed step of next_bound_below, :704-706).
#include "partition_search.hpp"

#include <cmath>
#include <limits>

namespace ecckd {

// the comparisons that steer the search, numbered for the decision trace (PartitionSearch::set_trace): 1 line search;
// 2-10 equipartition_2; 12-15 equipartition_n; 16-17 equipartition_e; 18-26 next_bound_below; 27-35 next_bound_above
#define LT(site, a, b) decide_lt((site), (a), (b))
#define GT(site, a, b) decide_gt((site), (a), (b))
#define EQ(site, a, b) decide_eq((site), (a), (b))

namespace {

// abound <- ascale*abound + bscale*bbound, ni+1 values (equipartition.cpp:23-28)
inline void blend(int ni, double* a, const double* b, double ascale, double bscale) {
  for (int i = 0; i <= ni; ++i) a[i] = ascale * a[i] + bscale * b[i];
}
}  // namespace

const char* partition_status_string(int status) {
  switch (status) {
    case PS_SUCCESS: return "Converged";
    case PS_MAX_ITERATIONS_REACHED: return "Maximum iterations reached";
    case PS_RESOLUTION_LIMIT_REACHED: return "Resolution limit reached";
    case PS_FAILED_TO_CONVERGE: return "Failed to converge";
    case PS_NO_PROGRESS: return "No progress made";
    case PS_FAILURE: return "Unspecified failure";
    case PS_INPUT_ERROR: return "Input error";
    default: return "Unknown convergence status";
  }
}

double PartitionSearch::calc_error(double b1, double b2) {
  double e = std::numeric_limits<double>::quiet_NaN();
  if (eval_status_) return e;
  int rc = evaluate_(1, &b1, &b2, &e);
  if (rc) eval_status_ = rc;
  return e;
}

int PartitionSearch::calc_error_all(int ni, const double* bounds, double* error) {
  if (eval_status_) return eval_status_;
  int rc = evaluate_(ni, bounds, bounds + 1, error);
  if (rc) eval_status_ = rc;
  return rc;
}

double PartitionSearch::cost_function(int ni, const double* error) const {
  double mean = 0.0;
  double lo = +std::numeric_limits<double>::infinity();
  double hi = -std::numeric_limits<double>::infinity();
  for (int i = 0; i < ni; ++i) {
    mean += error[i];
    if (error[i] < lo) lo = error[i];
    if (error[i] > hi) hi = error[i];
  }
  mean /= ni;
  if (minimize_frac_range_) return (hi - lo) / mean;
  double chi2 = 0.0;
  for (int i = 0; i < ni; ++i) chi2 += (error[i] - mean) * (error[i] - mean);
  return std::sqrt(chi2 / ni) / mean;
}

int PartitionSearch::line_search(int ni, double* bounds, double* newbounds, double* error) {
  if (!errors_up_to_date_) {
    calc_error_all(ni, bounds, error);
    errors_up_to_date_ = true;
  }
  const double start_cost = cost_function(ni, error);
  blend(ni, newbounds, bounds, 0.5, 0.5);
  for (int left = line_search_max_iterations_; left > 0; --left) {
    calc_error_all(ni, newbounds, error);
    errors_up_to_date_ = false;
    if (eval_status_) return PS_FAILURE;
    const double trial_cost = cost_function(ni, error);
    if (LT(1, trial_cost, start_cost)) {
      blend(ni, bounds, newbounds, 0.0, 1.0);
      errors_up_to_date_ = true;
      return PS_SUCCESS;
    }
    blend(ni, newbounds, bounds, 0.5, 0.5);
  }
  return PS_NO_PROGRESS;
}

int PartitionSearch::equipartition_2(double* bounds, double* error) {
  if (!errors_up_to_date_) {
    calc_error_all(2, bounds, error);
    errors_up_to_date_ = true;
  }
  double bound_left = bounds[0], bound_right = bounds[2];
  double ediff_left = 0.0, ediff_right = 0.0;
  double frac_error = 0.5 * std::fabs(error[1] - error[0]) / (error[0] + error[1]);
  const double frac_error_orig = frac_error;
  double nb[3] = {bounds[0], bounds[1], bounds[2]};
  double ne[2] = {error[0], error[1]};
  int left = partition_max_iterations_;

  if (GT(2, error[0], error[1])) {
    // the middle bound is too far right: march it left until the sign flips
    bound_right = bounds[1];
    ediff_right = error[1] - error[0];
    while (left) {
      nb[1] = (-ediff_right * nb[0] + (ne[0] + ediff_right) * nb[1]) / ne[0];
      calc_error_all(2, nb, ne);
      if (eval_status_) return PS_FAILURE;
      if (LT(3, ne[0], ne[1])) {
        bound_left = nb[1];
        ediff_left = ne[1] - ne[0];
        break;
      }
      ediff_right = ne[1] - ne[0];
      --left;
    }
  } else {
    bound_left = bounds[1];
    ediff_left = error[1] - error[0];
    while (left) {
      nb[1] = (ediff_left * nb[2] + (ne[1] - ediff_left) * nb[1]) / ne[1];
      calc_error_all(2, nb, ne);
      if (eval_status_) return PS_FAILURE;
      if (GT(4, ne[0], ne[1])) {
        bound_right = nb[1];
        ediff_right = ne[1] - ne[0];
        break;
      }
      ediff_left = ne[1] - ne[0];
      --left;
    }
  }

  bool stalled = false;
  double prev_frac_error = frac_error;
  while (left) {
    // regula falsi between the two bracketing positions (ediff_right < 0)
    if (stalled) nb[1] = 0.5 * (bound_right + bound_left);
    else nb[1] = (ediff_left * bound_right - ediff_right * bound_left) / (ediff_left - ediff_right);
    calc_error_all(2, nb, ne);
    if (eval_status_) return PS_FAILURE;
    const double ediff = ne[1] - ne[0];
    frac_error = 0.5 * std::fabs(ediff) / (ne[0] + ne[1]);
    if (LT(5, frac_error, partition_tolerance_) && LT(6, frac_error, frac_error_orig)) {
      bounds[1] = nb[1];
      error[0] = ne[0];
      error[1] = ne[1];
      errors_up_to_date_ = true;
      return PS_SUCCESS;
    } else if (EQ(7, frac_error, prev_frac_error)) {
      if (stalled) break;
      stalled = true;
    }
    if (LT(8, ediff, 0.0)) {
      ediff_right = ediff;
      bound_right = nb[1];
    } else {
      ediff_left = ediff;
      bound_left = nb[1];
    }
    prev_frac_error = frac_error;
    --left;
  }

  if (!LT(9, frac_error, frac_error_orig)) return PS_NO_PROGRESS;
  bounds[1] = nb[1];
  error[0] = ne[0];
  error[1] = ne[1];
  errors_up_to_date_ = true;
  if (LT(10, bound_right - bound_left, resolution_)) return PS_RESOLUTION_LIMIT_REACHED;
  if (!left) return PS_MAX_ITERATIONS_REACHED;
  return PS_SUCCESS;
}

int PartitionSearch::equipartition_n(int ni, double* bounds_out, double* error) {
  if (ni == 2) return equipartition_2(bounds_out, error);
  int status = PS_SUCCESS;
  int shuffles_left = partition_max_iterations_ / 2;
  for (int i = 0; i < ni; ++i)
    if (bounds_out[i + 1] <= bounds_out[i]) return PS_INPUT_ERROR;

  std::vector<double> bounds(bounds_out, bounds_out + ni + 1);
  std::vector<double> cum(ni + 1), trial(ni + 1);
  int left = partition_max_iterations_;

  while (left > 0) {
    if (!errors_up_to_date_) {
      calc_error_all(ni, bounds.data(), error);
      errors_up_to_date_ = true;
    }
    if (eval_status_) { status = PS_FAILURE; break; }
    if (LT(12, cost_function(ni, error), partition_tolerance_)) break;

    // move every interior bound to where the cumulative error (piecewise linear
    // or cubic in the bound) reaches an equal share
    cum[0] = 0.0;
    for (int i = 0; i < ni; ++i) cum[i + 1] = cum[i] + error[i];
    const double share = cum[ni] / ni;
    trial[0] = bounds[0];
    trial[ni] = bounds[ni];
    int k = 0;
    for (int j = 1; j < ni; ++j) {
      const double target = share * j;
      while (k + 1 < ni && LT(13, cum[k + 1], target)) ++k;
      if (cubic_interpolation_) {
        const double u = (target - cum[k]) / (cum[k + 1] - cum[k]);
        const double u2 = u * u;
        const double u3 = u2 * u;
        const double grad = (bounds[k + 1] - bounds[k]) / (cum[k + 1] - cum[k]);
        const double grad0 = (k == 0) ? grad : (bounds[k + 1] - bounds[k - 1]) / (cum[k + 1] - cum[k - 1]);
        const double grad1 = (k == ni - 1) ? grad : (bounds[k + 2] - bounds[k]) / (cum[k + 2] - cum[k]);
        trial[j] = (2.0 * u3 - 3.0 * u2 + 1) * bounds[k] + (u3 - 2.0 * u2 + u) * grad0 +
                   (-2.0 * u3 + 3.0 * u2) * bounds[k + 1] + (u3 - u2) * grad1;
      } else {
        trial[j] = ((cum[k + 1] - target) * bounds[k] + (target - cum[k]) * bounds[k + 1]) /
                   (cum[k + 1] - cum[k]);
      }
    }

    if (resolution_ > 0.0) {
      bool moved = false;
      for (int i = 1; i < ni; ++i) {
        if (GT(14, std::fabs(trial[i] - bounds[i]), resolution_)) { moved = true; break; }
      }
      if (!moved) {
        for (int i = 0; i <= ni; ++i) bounds_out[i] = bounds[i];
        return PS_RESOLUTION_LIMIT_REACHED;
      }
    }

    int ls = line_search(ni, bounds.data(), trial.data(), error);
    if (eval_status_) { status = PS_FAILURE; break; }
    if (ls != PS_SUCCESS) {
      status = PS_FAILED_TO_CONVERGE;
      int stuck = 0;
      if (ni > 2 && shuffles_left > 0) {
        // pairwise sweeps, direction alternating with the shuffle count
        if (shuffles_left % 2) {
          for (int i = 0; i < ni - 1; ++i)
            if (equipartition_2(&bounds[i], &error[i]) == PS_NO_PROGRESS) ++stuck;
          for (int i = ni - 3; i >= 0; --i)
            if (equipartition_2(&bounds[i], &error[i]) == PS_NO_PROGRESS) ++stuck;
        } else {
          for (int i = ni - 2; i >= 0; --i)
            if (equipartition_2(&bounds[i], &error[i]) == PS_NO_PROGRESS) ++stuck;
          for (int i = 1; i < ni - 1; ++i)
            if (equipartition_2(&bounds[i], &error[i]) == PS_NO_PROGRESS) ++stuck;
        }
        --shuffles_left;
        if (eval_status_) { status = PS_FAILURE; break; }
        if (LT(15, cost_function(ni, error), partition_tolerance_)) {
          status = PS_SUCCESS;
          break;
        } else if (stuck >= ni * 2 - 3) {
          status = PS_FAILED_TO_CONVERGE;
        } else {
          status = PS_SUCCESS;  // some pair improved: go round again
        }
      }
      if (status != PS_SUCCESS) break;
    }
    --left;
  }

  for (int i = 0; i <= ni; ++i) bounds_out[i] = bounds[i];
  if (left == 0) status = PS_MAX_ITERATIONS_REACHED;
  errors_up_to_date_ = false;
  if (eval_status_) return PS_FAILURE;
  return status;
}

int PartitionSearch::equipartition_e(double target_error, double bound0, double boundn, int& ni,
                                     std::vector<double>& bounds, std::vector<double>& error) {
  if (boundn <= bound0) return PS_INPUT_ERROR;

  // topmost interval first: search downwards from boundn
  double upper_error = -1.0;
  const double upper_bound =
      next_bound_below(target_error, bound0, boundn, 0.05 * bound0 + 0.95 * boundn, &upper_error);
  if (eval_status_) return PS_FAILURE;
  if (EQ(16, upper_bound, bound0)) {
    ni = 1;
    bounds.assign({bound0, boundn});
    error.assign({upper_error});
    return PS_SUCCESS;
  }

  bounds.assign({bound0});
  error.clear();
  // then fill upwards from bound0 until the topmost interval is met
  size_t i = 0;
  while (LT(17, bounds[i], upper_bound)) {
    double e = -1.0;
    double b = next_bound_above(target_error, bounds[i], upper_bound, 0.25 * bounds[i] + 0.75 * upper_bound, &e);
    if (eval_status_) return PS_FAILURE;
    error.push_back(e);
    bounds.push_back(b);
    ++i;
    if (i > 1000000) return PS_FAILURE;  // no progress: the reference would not terminate
  }
  error.push_back(upper_error);
  bounds.push_back(boundn);
  ni = (int)error.size();

  errors_up_to_date_ = true;
  return equipartition_n(ni, bounds.data(), error.data());
}

double PartitionSearch::next_bound_below(double target_error, double bound0, double bound2,
                                         double bound1_test, double* error_test_value) {
  const double max_error = target_error;
  const double min_error = target_error * (1.0 - next_bound_error_tolerance_);
  double lo = bound0, hi = bound2;        // bracket on bound1
  double error_lo = -1.0, error_hi = 0.0; // errors at the bracket ends (lo -> larger error)
  double error_test = (*error_test_value < 0.0) ? calc_error(bound1_test, bound2) : *error_test_value;

  for (int left = next_bound_max_iterations_;
       left > 0 && (GT(18, error_test, max_error) || LT(19, error_test, min_error)); --left) {
    if (eval_status_) break;
    if (GT(20, error_test, target_error)) {
      lo = bound1_test;
      error_lo = error_test;
    } else {
      hi = bound1_test;
      error_hi = error_test;
    }
    if (EQ(21, lo, hi)) break;
    if (GT(22, error_lo, 0.0)) {
      bound1_test = ((target_error - error_hi) * lo + (error_lo - target_error) * hi) / (error_lo - error_hi);
      if (EQ(23, error_hi, 0.0)) {
        bound1_test = 0.5 * (bound1_test + hi);
      } else if (LT(24, error_test, min_error) && GT(25, error_lo, 2.0 * max_error)) {
        bound1_test = 0.75 * bound1_test + 0.25 * lo;
      }
    } else {
      const double extrapolated = hi - 0.5 * target_error * (bound2 - hi) / error_hi;
      bound1_test = LT(26, lo, extrapolated) ? extrapolated : lo;      // std::max(lo, extrapolated)
    }
    error_test = calc_error(bound1_test, bound2);
  }
  *error_test_value = error_test;
  return bound1_test;
}

double PartitionSearch::next_bound_above(double target_error, double bound1, double boundn,
                                         double bound2_test, double* error_test_value) {
  const double max_error = target_error;
  const double min_error = target_error * (1.0 - next_bound_error_tolerance_);
  double lo = bound1, hi = boundn;        // bracket on bound2
  double error_lo = 0.0, error_hi = -1.0;
  double error_test = (*error_test_value < 0.0) ? calc_error(bound1, bound2_test) : *error_test_value;

  for (int left = next_bound_max_iterations_;
       left > 0 && (GT(27, error_test, max_error) || LT(28, error_test, min_error)); --left) {
    if (eval_status_) break;
    if (GT(29, error_test, target_error)) {
      hi = bound2_test;
      error_hi = error_test;
    } else {
      lo = bound2_test;
      error_lo = error_test;
    }
    if (EQ(30, lo, hi)) break;
    if (GT(31, error_hi, 0.0)) {
      bound2_test = ((target_error - error_lo) * hi + (error_hi - target_error) * lo) / (error_hi - error_lo);
      if (EQ(32, error_lo, 0.0)) {
        bound2_test = 0.5 * (bound2_test + lo);
      } else if (LT(33, error_test, min_error) && GT(34, error_lo, 2.0 * max_error)) {
        bound2_test = 0.75 * bound2_test + 0.25 * hi;
      }
    } else {
      const double extrapolated = hi - 0.5 * target_error * (lo - bound1) / error_lo;
      bound2_test = LT(35, hi, extrapolated) ? extrapolated : hi;      // std::max(hi, extrapolated)
    }
    error_test = calc_error(bound1, bound2_test);
  }
  *error_test_value = error_test;
  return bound2_test;
}

}  // namespace ecckd
