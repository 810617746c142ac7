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

// ---- the same exp with its constants pinned in scalar registers ------------------------------------
// gfx9-family VOP3 cannot encode a 64-bit literal, so every polynomial coefficient has to come from a
// register.  In a kernel that already fills its vector registers (K1: one wave per SIMD, whole column in
// VGPR/AGPRs) the compiler re-materialises each coefficient next to every use - two s_mov_b32 or two
// v_mov_b32 plus a destructive v_fmac per Horner step, ~37 of 181 instructions per layer in K1.  An empty
// asm with an "s" constraint makes the value opaque: it is built once and stays in an SGPR pair, and each
// Horner step is a single v_fma_f64 with a scalar addend.
__device__ __forceinline__ double sgpr_pin(double v) {
  asm volatile("" : "+s"(v));
  return v;
}

struct ExpConsts {
  double log2e, ln2hi, ln2lo, c12, c11, c10, c9, c8, c7, c6, c5, c4, c3;
};

__device__ __forceinline__ ExpConsts exp_consts() {
  ExpConsts k;
  k.log2e = sgpr_pin(1.4426950408889634074);
  k.ln2hi = sgpr_pin(-6.93147180369123816490e-01);
  k.ln2lo = sgpr_pin(-1.90821492927058770002e-10);
  k.c12 = sgpr_pin(2.08767569878680989792e-09);
  k.c11 = sgpr_pin(2.50521083854417187751e-08);
  k.c10 = sgpr_pin(2.75573192239858906526e-07);
  k.c9 = sgpr_pin(2.75573192239858906526e-06);
  k.c8 = sgpr_pin(2.48015873015873015873e-05);
  k.c7 = sgpr_pin(1.98412698412698412698e-04);
  k.c6 = sgpr_pin(1.38888888888888888889e-03);
  k.c5 = sgpr_pin(8.33333333333333333333e-03);
  k.c4 = sgpr_pin(4.16666666666666666667e-02);
  k.c3 = sgpr_pin(1.66666666666666666667e-01);
  return k;
}

// Bit-identical to exp_fast for every finite argument: the exponent clamp of exp_fast only bounds what
// v_cvt_i32_f64 (which saturates) and v_ldexp_f64 (which overflows to inf / underflows to 0) do anyway.
__device__ __forceinline__ double exp_fast_s(double y, const ExpConsts& k) {
  const double kf = __builtin_rint(y * k.log2e);
  double r = __builtin_fma(kf, k.ln2hi, y);
  r = __builtin_fma(kf, k.ln2lo, r);
  double p = __builtin_fma(k.c12, r, k.c11);
  p = __builtin_fma(p, r, k.c10);
  p = __builtin_fma(p, r, k.c9);
  p = __builtin_fma(p, r, k.c8);
  p = __builtin_fma(p, r, k.c7);
  p = __builtin_fma(p, r, k.c6);
  p = __builtin_fma(p, r, k.c5);
  p = __builtin_fma(p, r, k.c4);
  p = __builtin_fma(p, r, k.c3);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  int e;
  asm("v_cvt_i32_f64_e32 %0, %1" : "=v"(e) : "v"(kf));
  return __builtin_amdgcn_ldexp(p, e);
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
