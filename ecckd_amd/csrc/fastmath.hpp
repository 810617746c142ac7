// fastmath.hpp - fp64 device helpers shared by the hot kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace ecckd {

// exp(y), any sign.  Cody-Waite reduction y = k ln2 + r, |r| <= ln2/2, degree-12 Taylor/Horner
// in FMA, scaled with v_ldexp_f64.  < 1 ulp over the finite range; below -745 flushes to 0,
// above 709 overflows to +inf like libm.
__device__ __forceinline__ double exp_fast(double y) {
  const double kf = __builtin_rint(y * 1.4426950408889634074);
  double r = __builtin_fma(kf, -6.93147180369123816490e-01, y);
  r = __builtin_fma(kf, -1.90821492927058770002e-10, r);
  double p = 2.08767569878680989792e-09;                 // 1/12!
  p = __builtin_fma(p, r, 2.50521083854417187751e-08);   // 1/11!
  p = __builtin_fma(p, r, 2.75573192239858906526e-07);   // 1/10!
  p = __builtin_fma(p, r, 2.75573192239858906526e-06);   // 1/9!
  p = __builtin_fma(p, r, 2.48015873015873015873e-05);   // 1/8!
  p = __builtin_fma(p, r, 1.98412698412698412698e-04);   // 1/7!
  p = __builtin_fma(p, r, 1.38888888888888888889e-03);   // 1/6!
  p = __builtin_fma(p, r, 8.33333333333333333333e-03);   // 1/5!
  p = __builtin_fma(p, r, 4.16666666666666666667e-02);   // 1/4!
  p = __builtin_fma(p, r, 1.66666666666666666667e-01);   // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  const int k = (int)fmin(fmax(kf, -1100.0), 1100.0);
  return __builtin_amdgcn_ldexp(p, k);
}

// a / b for b in the normal range: v_rcp_f64 seed + 2 Newton steps + 1 residual correction
// (relative error < 1 ulp; no denormal/inf handling - callers guarantee a sane b).
__device__ __forceinline__ double div_fast(double a, double b) {
  double y = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  const double q = a * y;
  const double res = __builtin_fma(-b, q, a);
  return __builtin_fma(res, y, q);
}

}  // namespace ecckd
