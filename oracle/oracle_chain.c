/* oracle_chain.c - the CPU side of the benchmark (TEST INFRASTRUCTURE, see ecckd_oracle.h): the chain
 *   reorder_spectrum (key, per-band stable sort)  ->  find_g_points (gas preparation, one partition search per band)
 * for ONE gas, composed from the restated pieces exactly as the reference's main() functions compose them
 * (reorder_spectrum.cpp:111-300, find_g_points.cpp:872-1266), with the partition search done by the REFERENCE's own
 * Equipartition class (oracle/_ref/libequipartition_ref.so, loaded at run time) over a plain C callback - no Python
 * between the search and calc_error.  OpenMP at the reference's sites only: planck_function.cpp:50 (orc_planck_function)
 * and equipartition.h:101 (the search's calc_error_all, set_parallel as find_g_points.cpp:231 does).
 * bench.py times this on a bounded sample of its workload as `cpu_baseline`. */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ecckd_oracle.h"

typedef double (*refep_cb)(double, double, void*);
typedef struct {
  void* lib;
  void* (*create)(refep_cb, void*);
  void (*destroy)(void*);
  void (*set_verbose)(void*, int);
  void (*set_partition_max_iterations)(void*, int);
  void (*set_partition_tolerance)(void*, double);
  void (*set_resolution)(void*, double);
  void (*set_parallel)(void*, int);
  void (*set_minimize_frac_range)(void*, int);
  int (*equipartition_e)(void*, double, double, double, int*, double*, double*, int);
} refep_api;

static int load_refep(const char* path, refep_api* a) {
  memset(a, 0, sizeof *a);
  a->lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!a->lib) return 1;
#define SYM(f, name) *(void**)(&a->f) = dlsym(a->lib, name); if (!a->f) return 2
  SYM(create, "refep_create"); SYM(destroy, "refep_destroy"); SYM(set_verbose, "refep_set_verbose");
  SYM(set_partition_max_iterations, "refep_set_partition_max_iterations");
  SYM(set_partition_tolerance, "refep_set_partition_tolerance"); SYM(set_resolution, "refep_set_resolution");
  SYM(set_parallel, "refep_set_parallel"); SYM(set_minimize_frac_range, "refep_set_minimize_frac_range");
  SYM(equipartition_e, "refep_equipartition_e");
#undef SYM
  return 0;
}

typedef struct {
  orc_ckd_equipartition eq;   /* band-local views */
  double comp_cost;
  int failed;
} band_ctx;

/* the virtual calc_error of CkdEquipartition (find_g_points.cpp:291-405); the reference calls it from OpenMP threads and
 * lets total_comp_cost race (:320); here the counter is kept exact */
static double calc_error_cb(double b1, double b2, void* user) {
  band_ctx* c = (band_ctx*)user;
  orc_ckd_equipartition local = c->eq;      /* per-call copy: orc_ckd_calc_error adds to total_comp_cost */
  local.total_comp_cost = 0.0;
  int status = 0;
  const double e = orc_ckd_calc_error(&local, b1, b2, &status);
#pragma omp atomic
  c->comp_cost += local.total_comp_cost;
  if (status) c->failed = status;
  return e;
}

static double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

int orc_find_g_lw_chain_ex(const char* refep_path, int nlay, size_t nwav, const double* pressure_hl,
                           const double* temperature_hl, const double* wn, const double* dwn, const float* od32,
                           const float* bg32, double threshold_optical_depth, int nband, const double* band_bound1,
                           const double* band_bound2, int method, double flux_weight, double min_pressure,
                           const double* tolerance, double tolerance_tolerance, int max_iterations, int parallel,
                           int* ng, double* comp_cost, int* status, double* seconds, int32_t* rank, int capacity,
                           int64_t* rank1, int64_t* rank2, double* error_out, double* median_out, double* key_out,
                           double* planck_io, int planck_mode);

/* One gas, longwave.  Inputs in ORIGINAL wavenumber order, FLOAT optical depths as in the CKDMIP files (widened to double
 * as DataFile::read does, DataFileEngineNetcdf.cpp:593-599).  Outputs: ng[nband], comp_cost[nband], status[nband],
 * seconds[3] = {reorder, preparation, search}, rank[nwav].  Returns 0, or a non-zero code. */
int orc_find_g_lw_chain(const char* refep_path, int nlay, size_t nwav, const double* pressure_hl,
                        const double* temperature_hl, const double* wn, const double* dwn, const float* od32,
                        const float* bg32, double threshold_optical_depth, int nband, const double* band_bound1,
                        const double* band_bound2, int method, double flux_weight, double min_pressure,
                        const double* tolerance, double tolerance_tolerance, int max_iterations, int parallel,
                        int* ng, double* comp_cost, int* status, double* seconds, int32_t* rank) {
  return orc_find_g_lw_chain_ex(refep_path, nlay, nwav, pressure_hl, temperature_hl, wn, dwn, od32, bg32, threshold_optical_depth,
                                nband, band_bound1, band_bound2, method, flux_weight, min_pressure, tolerance, tolerance_tolerance,
                                max_iterations, parallel, ng, comp_cost, status, seconds, rank, 0, NULL, NULL, NULL, NULL, NULL, NULL, 0);
}

/* The same with the per-g-point results the rest of find_g_points needs (:1396-1409): rank1 / rank2 (first and last sorted
 * index), error and Planck-weighted median sorting variable of every g point, bands one after the other (capacity entries);
 * key_out[nwav] = the sorting variable (what write_order stores).  planck_mode: 0 = the Planck matrix of this gas's own
 * ordering; 1 = the same, and copied to planck_io[(nlay+1)*nwav] (the FIRST gas of a find_g_points run); 2 = planck_io is
 * used INSTEAD - the reference evaluates the matrix for the first gas only and keeps it for the later ones
 * (find_g_points.cpp:529, :970-984). */
/* A merged background of several gases is a DOUBLE matrix (read_merged_spectrum.cpp:135-166 sums scaling * optical depth in
 * double): handed over here for the NEXT call of orc_find_g_lw_chain_ex, which then ignores its FLOAT background argument. */
static const double* g_background64 = NULL;
void orc_chain_set_background64(const double* bg64) { g_background64 = bg64; }

int orc_find_g_lw_chain_ex(const char* refep_path, int nlay, size_t nwav, const double* pressure_hl,
                           const double* temperature_hl, const double* wn, const double* dwn, const float* od32,
                           const float* bg32, double threshold_optical_depth, int nband, const double* band_bound1,
                           const double* band_bound2, int method, double flux_weight, double min_pressure,
                           const double* tolerance, double tolerance_tolerance, int max_iterations, int parallel,
                           int* ng, double* comp_cost, int* status, double* seconds, int32_t* rank, int capacity,
                           int64_t* rank1, int64_t* rank2, double* error_out, double* median_out, double* key_out,
                           double* planck_io, int planck_mode) {
  refep_api ep;
  const double* bg64 = g_background64;
  g_background64 = NULL;
  if (load_refep(refep_path, &ep)) return 10;
  const size_t nhl = (size_t)nlay + 1;
  const size_t mat = (size_t)nlay * nwav;
  int rc = 0;
  double* od = (double*)malloc(mat * sizeof(double));
  double* bg = (double*)malloc(mat * sizeof(double));
  double* key = (double*)malloc(nwav * sizeof(double));
  double* col = (double*)malloc(nwav * sizeof(double));
  int32_t* iband = (int32_t*)malloc(nwav * sizeof(int32_t));
  int32_t* ordered = (int32_t*)malloc(nwav * sizeof(int32_t));
  double* t_ideal = (double*)malloc(nhl * sizeof(double));
  double *od_s = NULL, *bg_s = NULL, *tot_s = NULL, *wn_s = NULL, *dwn_s = NULL, *planck = NULL, *fdn = NULL, *fup = NULL,
         *hr = NULL, *metric = NULL, *ones = NULL, *lw = NULL, *fds = NULL, *fut = NULL;
  if (!od || !bg || !key || !col || !iband || !ordered || !t_ideal) { rc = 11; goto done; }
  for (size_t i = 0; i < mat; ++i) od[i] = (double)od32[i];
  for (size_t i = 0; i < mat; ++i) bg[i] = bg64 ? bg64[i] : (bg32 ? (double)bg32[i] : 0.0);

  /* ---- reorder_spectrum.cpp:111-300 ---- */
  double t0 = now_s();
  orc_idealised_temperature((int)nhl, pressure_hl, t_ideal);
  if (orc_reorder_key(nlay, nwav, pressure_hl, t_ideal, wn, dwn, od, NULL, threshold_optical_depth, key, col)) { rc = 12; goto done; }
  orc_stable_argsort_bands(nwav, wn, key, nband, band_bound1, band_bound2, iband, ordered, rank);
  double t1 = now_s();
  seconds[0] = t1 - t0;

  /* ---- find_g_points.cpp:872-1150: gather-reorder, Planck function, radiative transfer, heating rate, metric ---- */
  od_s = (double*)malloc(mat * sizeof(double)); bg_s = (double*)malloc(mat * sizeof(double));
  tot_s = (double*)malloc(mat * sizeof(double)); metric = (double*)malloc(mat * sizeof(double));
  wn_s = (double*)malloc(nwav * sizeof(double)); dwn_s = (double*)malloc(nwav * sizeof(double));
  planck = (double*)malloc(nhl * nwav * sizeof(double)); fdn = (double*)malloc(nhl * nwav * sizeof(double));
  fup = (double*)malloc(nhl * nwav * sizeof(double)); hr = (double*)malloc(mat * sizeof(double));
  ones = (double*)malloc(nwav * sizeof(double)); lw = (double*)malloc((size_t)nlay * sizeof(double));
  fds = (double*)malloc(nwav * sizeof(double)); fut = (double*)malloc(nwav * sizeof(double));
  if (!od_s || !bg_s || !tot_s || !metric || !wn_s || !dwn_s || !planck || !fdn || !fup || !hr || !ones || !lw || !fds || !fut) { rc = 11; goto done; }
  for (size_t i = 0; i < nwav; ++i) {          /* ireorder(irank) = index (:779-780); columns gathered (:899, :927-929) */
    const size_t j = (size_t)ordered[i];
    wn_s[i] = wn[j]; dwn_s[i] = dwn[j]; ones[i] = 1.0;
  }
  for (int l = 0; l < nlay; ++l)
    for (size_t i = 0; i < nwav; ++i) {
      const size_t j = (size_t)ordered[i];
      od_s[(size_t)l * nwav + i] = od[(size_t)l * nwav + j];
      bg_s[(size_t)l * nwav + i] = bg[(size_t)l * nwav + j];
      tot_s[(size_t)l * nwav + i] = bg[(size_t)l * nwav + j] + od[(size_t)l * nwav + j];
    }
  if (planck_mode == 2 && planck_io) memcpy(planck, planck_io, nhl * nwav * sizeof(double));
  else orc_planck_function((int)nhl, temperature_hl, nwav, wn_s, dwn_s, planck);                  /* :970-979 */
  if (planck_mode == 1 && planck_io) memcpy(planck_io, planck, nhl * nwav * sizeof(double));
  orc_radiative_transfer_lw(nlay, nwav, planck, tot_s, ones, planck + (size_t)nlay * nwav, fdn, fup);   /* :993-995 */
  orc_heating_rate(nlay, nwav, pressure_hl, fdn, fup, hr);                                       /* :1041 */
  memcpy(fds, fdn + (size_t)nlay * nwav, nwav * sizeof(double));                                 /* :1044-1053 */
  memcpy(fut, fup, nwav * sizeof(double));
  orc_layer_weight(nlay, pressure_hl, min_pressure, lw);                                         /* :1093-1099 */
  orc_metric(method, mat, od_s, metric);                                                         /* :1119-1150 */
  double t2 = now_s();
  seconds[1] = t2 - t1;

  /* ---- the bands (:1152-1266) ---- */
  int nout = 0;
  double* key_s = NULL;
  if (key_out) {
    memcpy(key_out, key, nwav * sizeof(double));
    key_s = (double*)malloc(nwav * sizeof(double));
    for (size_t i = 0; i < nwav; ++i) key_s[i] = key[ordered[i]];
  }
  for (int b = 0; b < nband; ++b) {
    long first = -1, last = -1;
    for (size_t i = 0; i < nwav; ++i)
      if (iband[ordered[i]] == b) { if (first < 0) first = (long)i; last = (long)i; }
    ng[b] = 0; comp_cost[b] = 0.0; status[b] = -1;
    if (first < 0) continue;
    band_ctx c;
    memset(&c, 0, sizeof c);
    c.eq.do_sw = 0; c.eq.method = method; c.eq.nlay = nlay; c.eq.npoints = (size_t)(last - first + 1); c.eq.stride = nwav;
    c.eq.flux_weight = flux_weight; c.eq.layer_weight = lw; c.eq.pressure_hl = pressure_hl;
    c.eq.surf_emissivity = ones + first; c.eq.surf_planck = planck + (size_t)nlay * nwav + first;
    c.eq.flux_dn_surf = fds + first; c.eq.flux_up_toa = fut + first; c.eq.planck_hl = planck + first;
    c.eq.bg_od = bg_s + first; c.eq.metric = metric + first; c.eq.hr = hr + first;
    void* h = ep.create(calc_error_cb, &c);
    ep.set_verbose(h, 0);
    ep.set_resolution(h, 1.0 / (double)c.eq.npoints);          /* CkdEquipartition::init_lw, find_g_points.cpp:230-233 */
    ep.set_minimize_frac_range(h, 1);
    ep.set_parallel(h, parallel);
    ep.set_partition_max_iterations(h, max_iterations);        /* :1180-1181 */
    ep.set_partition_tolerance(h, tolerance_tolerance);
    enum { CAP = 4096 };
    double* bounds = (double*)malloc((CAP + 1) * sizeof(double));
    double* err = (double*)malloc(CAP * sizeof(double));
    int n = 10;
    status[b] = ep.equipartition_e(h, tolerance[b], 0.0, 1.0, &n, bounds, err, CAP);
    ng[b] = n;
    comp_cost[b] = c.comp_cost;
    if (c.failed) rc = 20 + c.failed;
    if (rank1 && status[b] >= 0) {
      const double np1 = (double)(c.eq.npoints - 1);
      for (int k = 0; k < n && nout < capacity; ++k, ++nout) {
        rank1[nout] = (int64_t)ceil(bounds[k] * np1) + first;                 /* :1397-1398 */
        rank2[nout] = (int64_t)floor(bounds[k + 1] * np1) + first;
        error_out[nout] = err[k];
        if (median_out && key_s)
          median_out[nout] = orc_median_sorting_variable(key_s, planck + (size_t)nlay * nwav, (size_t)rank1[nout], (size_t)rank2[nout]);
      }
    }
    free(bounds); free(err);
    ep.destroy(h);
  }
  seconds[2] = now_s() - t2;
  free(key_s);

done:
  free(od); free(bg); free(key); free(col); free(iband); free(ordered); free(t_ideal); free(od_s); free(bg_s); free(tot_s);
  free(wn_s); free(dwn_s); free(planck); free(fdn); free(fup); free(hr); free(metric); free(ones); free(lw); free(fds); free(fut);
  if (ep.lib) dlclose(ep.lib);
  return rc;
}
