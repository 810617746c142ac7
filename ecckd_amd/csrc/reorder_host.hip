// reorder_host.hip - host-pointer wrapper of the whole reorder hot path
// (reference src/ecckd/reorder_spectrum.cpp:111-300): stage one column to the
// device, K1/K2 key, K3 per-band stable sort, copy the results back.
#include "common.hpp"

#include <vector>

static int reorder_impl(ecckd_ctx* ctx, int nlay, size_t nwav, const double* h_pressure_hl,
                        const double* h_wavenumber, const double* h_d_wavenumber,
                        const void* h_od, const void* d_od_in, int od_type, const double* h_ssi, double thr,
                        int nband, const double* h_band_bound1, const double* h_band_bound2,
                        double* h_key, double* h_col_od, int16_t* h_iband, int32_t* h_rank) {
  ECCKD_REQUIRE(ctx, "ecckd_reorder_spectrum: ctx is NULL");
  ECCKD_REQUIRE(nlay > 0 && nwav > 0, "ecckd_reorder_spectrum: empty spectrum (nlay=%d, nwav=%zu)", nlay, nwav);
  ECCKD_REQUIRE(h_pressure_hl && h_wavenumber && h_d_wavenumber && (h_od || d_od_in) && h_key && h_col_od && h_rank,
                "ecckd_reorder_spectrum: NULL array argument");
  ECCKD_REQUIRE(od_type == ECCKD_F32 || od_type == ECCKD_F64, "ecckd_reorder_spectrum: od_type must be 4 or 8");
  ECCKD_REQUIRE(nband > 0 && h_band_bound1 && h_band_bound2,
                "ecckd_reorder_spectrum: Failure to interpret wavenumber1 and wavenumber2 as a list of band boundaries");
  ECCKD_HIP_CHECK(hipSetDevice(ctx->device));

  std::vector<int64_t> bb(nband), be(nband);
  ECCKD_CHECK(ecckd_band_ranges(nwav, h_wavenumber, nband, h_band_bound1, h_band_bound2, h_iband, bb.data(), be.data()));

  const size_t od_bytes = (size_t)nlay * nwav * (size_t)od_type;
  void *d_od = nullptr, *d_wn = nullptr, *d_dwn = nullptr, *d_key = nullptr, *d_col = nullptr, *d_rank = nullptr;
  int rc = ECCKD_OK;
  auto cleanup = [&]() {
    (void)hipStreamSynchronize(ctx->stream);
    if (d_od && !d_od_in) (void)hipFree(d_od);
    if (d_wn) (void)hipFree(d_wn);
    if (d_dwn) (void)hipFree(d_dwn);
    if (d_key) (void)hipFree(d_key);
    if (d_col) (void)hipFree(d_col);
    if (d_rank) (void)hipFree(d_rank);
  };
#define TRY(x) do { rc = (x); if (rc != ECCKD_OK) { cleanup(); return rc; } } while (0)
  if (d_od_in) d_od = const_cast<void*>(d_od_in);
  else TRY(ecckd_dev_alloc(ctx, od_bytes, &d_od));
  TRY(ecckd_dev_alloc(ctx, nwav * 8, &d_wn));
  TRY(ecckd_dev_alloc(ctx, nwav * 8, &d_dwn));
  TRY(ecckd_dev_alloc(ctx, nwav * 8, &d_key));
  TRY(ecckd_dev_alloc(ctx, nwav * 8, &d_col));
  TRY(ecckd_dev_alloc(ctx, nwav * 4, &d_rank));
  if (!d_od_in) TRY(ecckd_h2d(ctx, d_od, h_od, od_bytes));
  if (!h_ssi) {
    std::vector<double> t_hl(nlay + 1);
    TRY(ecckd_idealised_temperature(nlay + 1, h_pressure_hl, t_hl.data()));
    TRY(ecckd_h2d(ctx, d_wn, h_wavenumber, nwav * 8));
    TRY(ecckd_h2d(ctx, d_dwn, h_d_wavenumber, nwav * 8));
    TRY(ecckd_reorder_key_lw_dev(ctx, nlay, nwav, h_pressure_hl, t_hl.data(), (const double*)d_wn,
                                 (const double*)d_dwn, d_od, od_type, nwav, thr, (double*)d_key, (double*)d_col));
  } else {
    // the SW key does not depend on ssi (reorder_spectrum.cpp:224-228)
    TRY(ecckd_reorder_key_sw_dev(ctx, nlay, nwav, h_pressure_hl, d_od, od_type, nwav, thr, (double*)d_key,
                                 (double*)d_col));
  }
  TRY(ecckd_stable_argsort_bands_dev(ctx, nwav, (const double*)d_key, nband, bb.data(), be.data(),
                                     (int32_t*)d_rank, nullptr));
  TRY(ecckd_d2h(ctx, h_key, d_key, nwav * 8));
  TRY(ecckd_d2h(ctx, h_col_od, d_col, nwav * 8));
  TRY(ecckd_d2h(ctx, h_rank, d_rank, nwav * 4));
#undef TRY
  cleanup();
  return ECCKD_OK;
}

extern "C" int ecckd_reorder_spectrum(ecckd_ctx* ctx, int nlay, size_t nwav, const double* h_pressure_hl,
                                      const double* h_wavenumber, const double* h_d_wavenumber,
                                      const void* h_od, int od_type, const double* h_ssi, double thr,
                                      int nband, const double* h_band_bound1, const double* h_band_bound2,
                                      double* h_key, double* h_col_od, int16_t* h_iband, int32_t* h_rank) {
  ECCKD_REQUIRE(h_od, "ecckd_reorder_spectrum: NULL array argument");
  return reorder_impl(ctx, nlay, nwav, h_pressure_hl, h_wavenumber, h_d_wavenumber, h_od, nullptr, od_type, h_ssi, thr, nband,
                      h_band_bound1, h_band_bound2, h_key, h_col_od, h_iband, h_rank);
}

// the same with the optical depths already in device memory (read there by ecckd_nc_read_dev)
extern "C" int ecckd_reorder_spectrum_od_dev(ecckd_ctx* ctx, int nlay, size_t nwav, const double* h_pressure_hl,
                                             const double* h_wavenumber, const double* h_d_wavenumber,
                                             const void* d_od, int od_type, const double* h_ssi, double thr,
                                             int nband, const double* h_band_bound1, const double* h_band_bound2,
                                             double* h_key, double* h_col_od, int16_t* h_iband, int32_t* h_rank) {
  ECCKD_REQUIRE(d_od, "ecckd_reorder_spectrum_od_dev: NULL array argument");
  return reorder_impl(ctx, nlay, nwav, h_pressure_hl, h_wavenumber, h_d_wavenumber, nullptr, d_od, od_type, h_ssi, thr, nband,
                      h_band_bound1, h_band_bound2, h_key, h_col_od, h_iband, h_rank);
}
