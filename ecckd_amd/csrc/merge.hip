// merge.hip - the arithmetic of read_spectrum / read_merged_spectrum on the device:
// the wavenumber spacing derived from the grid (read_spectrum.cpp:55-65) and the merged optical depth
// sum_gas scaling(level) * optical_depth (read_merged_spectrum.cpp:117-166).  File access is the
// caller's business; these functions take the arrays the reader produced (FLOAT as stored in the CKDMIP
// files, or DOUBLE) and keep the merged matrix resident in HBM for find_g_points.
#include "common.hpp"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

__global__ void __launch_bounds__(256)
k_derive_dwn(size_t n, const double* __restrict__ wn, double* __restrict__ dwn) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  // interior: half the distance between the neighbours; the two ends: half of their neighbour's value
  auto interior = [&](size_t i) { return 0.5 * (wn[i + 1] - wn[i - 1]); };
  double v;
  if (j == 0) v = 0.5 * interior(1);
  else if (j == n - 1) v = 0.5 * interior(n - 2);
  else v = interior(j);
  dwn[j] = v;
}

// merged[l][j] (+)= od[l][j] * scale[l]; product and sum rounded separately, as the expression
// `optical_depth += od * spread<1>(scaling_profile, n)` evaluates without contraction
template <typename T>
__global__ void __launch_bounds__(256)
k_merge(size_t n, size_t od_stride, size_t out_stride, const T* __restrict__ od, const double* __restrict__ scale,
        int first, double* __restrict__ merged) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int l = blockIdx.y;
  if (j >= n) return;
  const double v = __dmul_rn((double)od[(size_t)l * od_stride + j], scale[l]);
  double* o = merged + (size_t)l * out_stride + j;
  *o = first ? v : __dadd_rn(*o, v);
}

}  // namespace

extern "C" {

int ecckd_derive_d_wavenumber_dev(ecckd_ctx* ctx, size_t nwav, const double* d_wavenumber, double* d_d_wavenumber) {
  ECCKD_REQUIRE(ctx && d_wavenumber && d_d_wavenumber, "ecckd_derive_d_wavenumber_dev: NULL argument");
  ECCKD_REQUIRE(nwav >= 3, "ecckd_derive_d_wavenumber_dev: at least 3 wavenumbers needed, got %zu", nwav);
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_derive_dwn, dim3((unsigned)((nwav + 255) / 256)), dim3(256), 0, ctx->stream, nwav, d_wavenumber,
                     d_d_wavenumber);
  ECCKD_HIP_CHECK(hipGetLastError());
  return ECCKD_OK;
}

int ecckd_merge_scaling(int nlay, const double* h_pressure_hl, double scaling, double conc, double reference_surface_vmr,
                        const double* h_vmr_fl_one_gas, int nconc, const double* h_pressure_conc,
                        const double* h_conc_req, double* h_scaling_profile, double* h_vmr_fl_out) {
  ECCKD_REQUIRE(nlay > 0 && h_pressure_hl && h_scaling_profile, "ecckd_merge_scaling: bad argument");
  if (nconc > 0) {
    // a requested concentration profile (read_merged_spectrum.cpp:117-131): interp in pressure with the
    // ends clamped, then the ratio to the file's own profile
    ECCKD_REQUIRE(h_pressure_conc && h_conc_req && h_vmr_fl_one_gas, "ecckd_merge_scaling: concentration profile arrays missing");
    for (int l = 0; l < nlay; ++l) {
      const double pfl = 0.5 * (h_pressure_hl[l] + h_pressure_hl[l + 1]);
      double c;
      if (nconc == 1 || pfl < h_pressure_conc[0]) c = h_conc_req[0];
      else if (pfl > h_pressure_conc[nconc - 1]) c = h_conc_req[nconc - 1];
      else {
        int j = 0;
        while (j < nconc - 2 && pfl > h_pressure_conc[j + 1]) ++j;
        const double w = (pfl - h_pressure_conc[j]) / (h_pressure_conc[j + 1] - h_pressure_conc[j]);
        c = (1.0 - w) * h_conc_req[j] + w * h_conc_req[j + 1];
      }
      h_scaling_profile[l] = c / h_vmr_fl_one_gas[l];
      if (h_vmr_fl_out) h_vmr_fl_out[l] = c;
    }
    return ECCKD_OK;
  }
  // scalar rules, :132-147
  if (conc == 0.0) scaling = 0.0;
  else if (conc > 0.0) {
    if (reference_surface_vmr < 0.0)
      return ecckd::fail(ECCKD_PARAMETER_ERROR, "Attempt to specify concentration when no reference_surface_mole_fraction present");
    scaling = conc / reference_surface_vmr;
  } else if (scaling < 0.0) scaling = 1.0;
  for (int l = 0; l < nlay; ++l) {
    h_scaling_profile[l] = scaling;
    // :158-164: the mole fraction is scaled too (an absent profile, -1, scales with it as in the reference)
    if (h_vmr_fl_out) h_vmr_fl_out[l] = (scaling != 1.0) ? (h_vmr_fl_one_gas ? h_vmr_fl_one_gas[l] : -1.0) * scaling
                                                         : (h_vmr_fl_one_gas ? h_vmr_fl_one_gas[l] : -1.0);
  }
  return ECCKD_OK;
}

int ecckd_merge_spectrum_dev(ecckd_ctx* ctx, int nlay, size_t nwav, const void* d_od, int od_type, size_t od_stride,
                             const double* h_scaling_profile, int first, double* d_merged, size_t merged_stride) {
  ECCKD_REQUIRE(ctx && d_od && h_scaling_profile && d_merged && nlay > 0 && nlay < 65536,
                "ecckd_merge_spectrum_dev: bad argument");
  ECCKD_REQUIRE(od_type == ECCKD_F32 || od_type == ECCKD_F64, "ecckd_merge_spectrum_dev: od_type must be ECCKD_F32 or ECCKD_F64");
  ECCKD_REQUIRE(od_stride >= nwav && merged_stride >= nwav, "ecckd_merge_spectrum_dev: row stride shorter than nwav");
  if (nwav == 0) return ECCKD_OK;
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));
  ECCKD_CHECK(ecckd::ensure_scratch(ctx, (size_t)nlay * sizeof(double)));
  double* d_scale = (double*)ctx->scratch;
  ECCKD_HIP_CHECK(hipMemcpyAsync(d_scale, h_scaling_profile, (size_t)nlay * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  const dim3 grid((unsigned)((nwav + 255) / 256), (unsigned)nlay);
  if (od_type == ECCKD_F32)
    hipLaunchKernelGGL(k_merge<float>, grid, dim3(256), 0, ctx->stream, nwav, od_stride, merged_stride, (const float*)d_od,
                       d_scale, first, d_merged);
  else
    hipLaunchKernelGGL(k_merge<double>, grid, dim3(256), 0, ctx->stream, nwav, od_stride, merged_stride, (const double*)d_od,
                       d_scale, first, d_merged);
  ECCKD_HIP_CHECK(hipGetLastError());
  // h_scaling_profile is pageable: the copy has been staged, but keep the call synchronous so the caller may reuse it
  ECCKD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return ECCKD_OK;
}

}  // extern "C"
