// find_g_points [key=value ...] [file.cfg]
//
// Drop-in for the reference executable of the same name (src/ecckd/find_g_points.cpp:407-1664, without the
// cloud pseudo-gas of :541-652, which the shipped configurations leave commented out): for every gas in
// `gases` and every band of its reordering file, partition the reordered spectrum into g points whose heating-
// rate error is within `heating_rate_tolerance`, then overlap the gases' g points and write the g-points file.
// Keys (:443-523, :655-771): output, ssi, iprofile, heating_rate_tolerance, tolerance_tolerance, max_iterations,
// averaging_method, flux_weight, min_pressure, max_no_rayleigh_wavenumber, gases, prepend_path, append_path,
// log_level; per gas <gas>.input / scaling / conc, <gas>.background_input / _scaling / _conc, reordering_input,
// min_scaling, max_scaling, g_split + subband_wavenumber_boundary, base_split, base_wavenumber_boundary,
// min_g_points, max_g_points.
// All nwav-sized work runs on the GPU through include/ecckd_hip.h; this file is the driver around it.
//
// Several processes, one per GPU (RANK / WORLD_SIZE / LOCAL_RANK from the launcher, e.g. torchrun --no-python): the (gas, band)
// searches are independent problems (:655, :1152); every process takes a contiguous share of the task table, reads and prepares
// only the gases of which it searches a band, and leaves its per-band results (a few numbers per g point and the band's slice
// of the - possibly re-ranked - rank) in "<output>.part<rank>"; rank 0 collects the parts, does what follows the gas loop
// (:1452-1660) and prints the final cost (sum of the g points' errors).  The g-points file does not depend on WORLD_SIZE.
#include <fstream>
#include <thread>
#include <unistd.h>
#include <algorithm>
#include <cctype>

#include <future>
#include <list>
#include "tool.hpp"

using namespace tool;

namespace {

struct GasResult {   // SingleGasData (single_gas_data.h:24-80)
  std::string molecule;
  std::vector<int> n_g_points;        // per band
  std::vector<int> band_number;       // per gas g point
  std::vector<int64_t> rank1, rank2;
  std::vector<double> error, sorting_variable;
  std::vector<int> g_min, g_max;      // per merged g point
  DevBuf d_g_point;                   // int32 [nwav]
};

// what one (gas, band) search hands back
struct BandResult {
  int gas = 0, band = 0, ng = 0, status = 0;
  double comp_cost = 0.0;
  int64_t ibegin = 0, iend = -1;
  std::vector<int64_t> rank1, rank2;
  std::vector<double> error, median;
  std::vector<int32_t> rank_slice;     // rank of the wavenumbers ibegin..iend after the search (sub-band / base-split re-ranking)
};

template <class T> void put(std::ofstream& f, const T& v) { f.write(reinterpret_cast<const char*>(&v), sizeof v); }
template <class T> void put_vec(std::ofstream& f, const std::vector<T>& v) {
  const uint64_t n = v.size();
  put(f, n);
  if (n) f.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(n * sizeof(T)));
}
template <class T> bool get(std::ifstream& f, T& v) { return (bool)f.read(reinterpret_cast<char*>(&v), sizeof v); }
template <class T> bool get_vec(std::ifstream& f, std::vector<T>& v) {
  uint64_t n = 0;
  if (!get(f, n) || n > ((uint64_t)1 << 33)) return false;
  v.resize((size_t)n);
  return n == 0 || (bool)f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(n * sizeof(T)));
}

// A part file carries the identity of its run: a hash of the configuration text, the launcher's rendezvous (MASTER_ADDR,
// MASTER_PORT, TORCHELASTIC_RUN_ID, ECCKD_RUN_ID), WORLD_SIZE and - known once the ordering files are read - the numbers of
// gases, bands and wavenumbers.  Process 0 refuses a part whose identity is not its own (a stale file of an aborted earlier
// run, ordering files that disagree between the processes).
uint64_t run_identity(const Config& config, int world) {
  uint64_t h = fnv1a(config.str());
  for (const char* name : {"MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "ECCKD_RUN_ID"}) {
    const char* e = std::getenv(name);
    h = fnv1a(std::string(e ? e : ""), h);
  }
  return fnv1a(&world, sizeof world, h);
}
uint64_t job_identity(uint64_t run_id, int ngas, int nband, uint64_t nwav) {
  uint64_t h = fnv1a(&ngas, sizeof ngas, run_id);
  h = fnv1a(&nband, sizeof nband, h);
  return fnv1a(&nwav, sizeof nwav, h);
}

void write_part(const std::string& path, const std::vector<BandResult>& res, uint64_t identity) {
  const std::string tmp = path + ".tmp";
  {
    std::ofstream f(tmp, std::ios::binary);
    if (!f) fail(ECCKD_PROCESSING_ERROR, "Cannot write %s", tmp.c_str());
    const uint64_t magic = 0x45434b4450415254ull, n = res.size();   // "ECKDPART"
    put(f, magic); put(f, identity); put(f, n);
    for (const BandResult& r : res) {
      put(f, r.gas); put(f, r.band); put(f, r.ng); put(f, r.status); put(f, r.comp_cost); put(f, r.ibegin); put(f, r.iend);
      put_vec(f, r.rank1); put_vec(f, r.rank2); put_vec(f, r.error); put_vec(f, r.median); put_vec(f, r.rank_slice);
    }
    if (!f) fail(ECCKD_PROCESSING_ERROR, "Short write of %s", tmp.c_str());
  }
  if (std::rename(tmp.c_str(), path.c_str()) != 0) fail(ECCKD_PROCESSING_ERROR, "Cannot rename %s", tmp.c_str());   // complete or absent
}

// the marker a process other than 0 leaves when it ends with an error: "<output>.part<r>.failed" = [run identity, exit code]
void write_failure_marker(const std::string& path, uint64_t run_id, int code) {
  std::ofstream f(path, std::ios::binary);
  put(f, run_id); put(f, code);
}

std::vector<BandResult> read_part(const std::string& path, double timeout_s, uint64_t run_id, uint64_t identity) {
  const auto t0 = std::chrono::steady_clock::now();
  const std::string failed = path + ".failed";
  while (access(path.c_str(), R_OK) != 0) {
    if (access(failed.c_str(), R_OK) == 0) {
      std::ifstream m(failed, std::ios::binary);
      uint64_t id = 0; int code = 0;
      if (get(m, id) && get(m, code) && id == run_id) {
        m.close();
        std::remove(failed.c_str());
        fail(ECCKD_PROCESSING_ERROR, "The process that was to write %s ended with exit code %d", path.c_str(), code);
      }                                         // a marker of another run: not ours to act on
    }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
      fail(ECCKD_PROCESSING_ERROR, "Timed out after %g s waiting for %s (was the process of that rank started?)", timeout_s, path.c_str());
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
  }
  std::ifstream f(path, std::ios::binary);
  uint64_t magic = 0, id = 0, n = 0;
  if (!get(f, magic) || magic != 0x45434b4450415254ull || !get(f, id) || !get(f, n)) fail(ECCKD_PROCESSING_ERROR, "%s is not a find_g_points part file", path.c_str());
  if (id != identity)
    fail(ECCKD_PROCESSING_ERROR, "%s belongs to another run (configuration, launcher rendezvous, WORLD_SIZE or the numbers of gases / bands / "
         "wavenumbers differ from this process's): remove it, and start every process of a run with the same configuration", path.c_str());
  std::vector<BandResult> res((size_t)n);
  for (BandResult& r : res) {
    const bool ok = get(f, r.gas) && get(f, r.band) && get(f, r.ng) && get(f, r.status) && get(f, r.comp_cost) && get(f, r.ibegin) && get(f, r.iend) &&
                    get_vec(f, r.rank1) && get_vec(f, r.rank2) && get_vec(f, r.error) && get_vec(f, r.median) && get_vec(f, r.rank_slice);
    if (!ok) fail(ECCKD_PROCESSING_ERROR, "%s is truncated", path.c_str());
  }
  return res;
}

template <class T>
std::vector<T> per_band(const Config& config, const std::string& gas, const char* key, int nband, T fill, bool* present = nullptr) {
  std::vector<T> raw, out(nband, fill);
  const bool have = config.read(raw, key, gas.c_str());
  if (present) *present = have;
  if (have)
    for (int b = 0; b < std::min<int>(nband, (int)raw.size()); ++b) out[b] = raw[b];
  return out;
}

}  // namespace

int main(int argc, char** argv) {
  return run(argc, argv, [&](Config& config) -> int {
    SearchPath paths;
    paths.configure(config);
    std::string output, ssi_file_name;
    if (!config.read(output, "output")) fail(ECCKD_PARAMETER_ERROR, "\"output\" file not specified");
    const double cos_sza = 0.5;               // REFERENCE_COS_SZA, calc_cost_function_sw.h:20
    const double reference_albedo = 0.15;     // :469
    std::vector<double> ssi;
    const bool do_sw = config.read(ssi_file_name, "ssi");
    if (do_sw) {
      LOG("Assuming shortwave spectral region (ssi provided)\nReading %s\n", ssi_file_name.c_str());
      NcIn f(paths.find(ssi_file_name));
      ssi = f.read("solar_spectral_irradiance");
    } else {
      LOG("Assuming longwave spectral region (ssi not provided)\n");
    }
    if (config.exist("cloud")) fail(ECCKD_PARAMETER_ERROR, "The cloud pseudo-gas is not supported by this tool");
    int iprofile = 0;
    config.read(iprofile, "iprofile");
    std::vector<double> tolerance_in;
    if (!config.read(tolerance_in, "heating_rate_tolerance")) fail(ECCKD_PARAMETER_ERROR, "heating_rate_tolerance not defined");
    double tolerance_tolerance = 0.02, flux_weight = 0.02, min_pressure = 0.0, max_no_rayleigh_wavenumber = 10000.0;
    int max_iterations = 60;
    std::string averaging_method = "linear";
    config.read(tolerance_tolerance, "tolerance_tolerance");
    config.read(max_iterations, "max_iterations");
    config.read(averaging_method, "averaging_method");
    config.read(flux_weight, "flux_weight");
    config.read(min_pressure, "min_pressure");
    config.read(max_no_rayleigh_wavenumber, "max_no_rayleigh_wavenumber");
    static const char* const methods[] = {"linear", "transmission", "transmission-2", "square-root", "logarithmic",
                                          "total-transmission", "transmission-3", "transmission-10", "hybrid-logarithmic-transmission-3"};
    int method = -1;
    for (int k = 0; k < 9; ++k) if (averaging_method == methods[k]) method = k;
    if (method < 0) fail(ECCKD_PARAMETER_ERROR, "Averaging method \"%s\" not understood", averaging_method.c_str());

    const int world = std::max(1, env_int("WORLD_SIZE", 1)), my_rank = env_int("RANK", 0);
    if (my_rank < 0 || my_rank >= world) fail(ECCKD_PARAMETER_ERROR, "RANK=%d outside WORLD_SIZE=%d", my_rank, world);
    const uint64_t run_id = run_identity(config, world);
    if (world > 1) {
      // before anything that can fail (device start-up, the gas list, the files): never a stale part or marker of an earlier
      // run under this process's name, and from here on a failure of this process is visible to process 0
      const std::string my_part = output + ".part" + std::to_string(my_rank);
      std::remove(my_part.c_str());
      std::remove((my_part + ".tmp").c_str());
      std::remove((my_part + ".failed").c_str());
      if (my_rank != 0) on_failure() = [my_part, run_id](int code) { write_failure_marker(my_part + ".failed", run_id, code); };
    }
    Device dev;
    dev.enable_od_cache();       // a gas's spectrum is the target once and part of the other gases' backgrounds: read once
    const std::vector<std::string> gas_list = config.read_list("gases");
    const int ngas = (int)gas_list.size();
    if (ngas == 0) fail(ECCKD_PARAMETER_ERROR, "No gases specified in \"gases\"");
    if (world > 1) {
      LOG("Process %d of %d: the (gas, band) searches are dealt in contiguous shares\n", my_rank, world);
    }
    std::vector<BandResult> results;                     // the searches of this process
    std::vector<std::vector<int32_t>> order_rank(ngas);  // process 0: the rank of every gas as its ordering file has it
    std::vector<double> band_bound1, band_bound2, wavenumber;
    int nband = 0;
    size_t nwav = 0;
    ecckd_gas* first_lw_gas = nullptr;
    const double* d_planck_first = nullptr;
    DevBuf d_planck_rebuilt;

    // the ordering file of a gas (:669-683) and the sub-bands of the optically thin part of `bands` (:788-870), which re-rank
    // d_rank before anything is reordered
    struct Ordering {
      std::vector<int32_t> rank;
      std::vector<int> iband;
      std::vector<double> sorting_variable, subband_wavenumber_boundary, g_split;
      std::vector<int64_t> ibegin, iend;   // first / last index of every band: the members of a band are contiguous in rank
      bool have_g_split = false;
      int nsb = 1;
      std::vector<int> nsubband;
      std::vector<int64_t> iupperindex, isubband1, isubband2;
      DevBuf d_rank, d_wn;
    };
    auto read_ordering = [&](const std::string& gas_str, Ordering& o) {
      const char* scope = gas_str.c_str();
      std::string reordering_input;
      if (!config.read(reordering_input, "reordering_input", scope)) fail(ECCKD_PARAMETER_ERROR, "No reordering_input found");
      LOG("Reading %s\n", reordering_input.c_str());
      {
        // the four per-wavenumber variables side by side, each through a handle of its own: a 7.2e6-point variable costs
        // ~20 ms, most of it the first touch of the 58 MB it is decoded into
        // (a classic file only: a NetCDF-4 one is read through the HDF5 library, which must not be entered by two threads)
        const std::string order_path = paths.find(reordering_input);
        bool classic = false;
        if (FILE* fp = std::fopen(order_path.c_str(), "rb")) {
          char magic[4] = {0, 0, 0, 0};
          classic = std::fread(magic, 1, 4, fp) == 4 && magic[0] == 'C' && magic[1] == 'D' && magic[2] == 'F';
          std::fclose(fp);
        }
        const std::launch how = classic ? std::launch::async : std::launch::deferred;
        auto read_var = [order_path](const char* name) { NcIn g(order_path); return g.read(name); };
        auto f_rank = std::async(how, [&] { std::vector<double> r = read_var("rank"); return std::vector<int32_t>(r.begin(), r.end()); });
        auto f_band = std::async(how, [&] { std::vector<double> b = read_var("band_number"); return std::vector<int>(b.begin(), b.end()); });
        auto f_sort = std::async(how, [&] { return read_var("sorting_variable"); });
        NcIn f(order_path);
        band_bound1 = f.read("wavenumber1_band");
        band_bound2 = f.read("wavenumber2_band");
        wavenumber = f.read("wavenumber");
        o.rank = f_rank.get();
        o.iband = f_band.get();
        o.sorting_variable = f_sort.get();
      }
      nband = (int)band_bound1.size();
      nwav = o.rank.size();
      o.g_split = per_band<double>(config, gas_str, "g_split", nband, -1.0, &o.have_g_split);
      if (o.have_g_split && !config.read(o.subband_wavenumber_boundary, "subband_wavenumber_boundary", scope))
        fail(ECCKD_PARAMETER_ERROR, "g_split must be accompanied by subband_wavenumber_boundary");
      o.ibegin.assign(nband, -1);
      o.iend.assign(nband, -1);
      for (size_t i = 0; i < nwav; ++i) {
        const int b = o.iband[i];
        if (b < 0 || b >= nband) continue;
        if (o.ibegin[b] < 0) o.ibegin[b] = (int64_t)i;
        o.iend[b] = (int64_t)i;
      }
      o.nsb = (int)o.subband_wavenumber_boundary.size() + 1;
      o.nsubband.assign(nband, 0);
      o.iupperindex.assign(nband, -1);
      o.isubband1.assign((size_t)nband * o.nsb, -1);
      o.isubband2.assign((size_t)nband * o.nsb, -1);
    };
    auto subband_setup = [&](Ordering& o, const std::vector<char>& bands) {
      o.d_rank.upload(dev, o.rank);
      o.d_wn.upload(dev, wavenumber);
      if (!o.have_g_split) return;
      for (int b = 0; b < nband; ++b)
        if (bands[b] && o.g_split[b] > 0.0 && o.ibegin[b] >= 0)
          ck(ecckd_subband_setup_dev(dev.ctx(), nwav, o.d_wn.as<double>(), o.d_rank.as<int32_t>(), o.ibegin[b], o.iend[b], o.g_split[b],
                                     band_bound1[b], band_bound2[b], (int)o.subband_wavenumber_boundary.size(),
                                     o.subband_wavenumber_boundary.data(), &o.nsubband[b], &o.isubband1[(size_t)b * o.nsb],
                                     &o.isubband2[(size_t)b * o.nsb], &o.iupperindex[b]));
    };

    // The searches of a gas start as soon as the gas is prepared and run - a host thread and a HIP stream per gas
    // (ecckd_find_g_gases_begin / _add / _wait) - while the next gas's files are read, merged and prepared: the reference's loop
    // (:655) takes gas after gas, but a search is a chain of small dependent evaluations that cannot fill the device, and the
    // reading of the next gas needs the host.  Every search takes the decisions it takes alone.  Extension keys:
    // gases_side_by_side = n (gases searched at a time; 1 = the reference's order, 0 = what the host has cores for),
    // sequential_bands (one band at a time, which also means gas after gas).
    struct GasJob {
      int gi = 0;
      Ordering ord;
      ecckd_gas* gas = nullptr;
      DevBuf d_sv_sorted;
      std::vector<int> mine;
      std::vector<double> band_albedo;
      std::vector<ecckd_band_options> opts;
      std::vector<std::vector<double>> wn_bounds;
      std::vector<size_t> ib, ie;
      std::vector<double> tol, comp_costs, bounds, error;
      std::vector<int> ngs, statuses;
      std::vector<int64_t> r1, r2;
      ecckd_gas_search req;
      bool searched = false;
    };
    const int band_capacity = 1024;
    bool sequential_bands = false;                 // extension key: the reference's one-band-at-a-time order of evaluation
    config.read(sequential_bands, "sequential_bands");
    int gases_side_by_side = 0;
    config.read(gases_side_by_side, "gases_side_by_side");
    if (sequential_bands) gases_side_by_side = 1;
    std::list<GasJob> pending;
    ecckd_gas_search_job* search_job = nullptr;

    // what follows a gas's searches (:1396-1414): log, median sorting variable of every g point, the ranks as the searches left them
    auto finish_gas = [&](GasJob& gj) {
      const int nmine = (int)gj.mine.size();
      std::string Gas = gas_list[gj.gi];
      std::transform(Gas.begin(), Gas.end(), Gas.begin(), ::toupper);
      LOG("*** G POINTS OF %s\n", Gas.c_str());
      for (int m = 0; m < nmine; ++m) {
        const int b = gj.mine[m];
        LOG("  Band %d: %g-%g cm-1\n", b, band_bound1[b], band_bound2[b]);
        const size_t o = (size_t)m * band_capacity;
        const int ng = gj.ngs[m];
        LOG("    %s: %d g points, computational cost = %g\n", ecckd_partition_status_string(gj.statuses[m]), ng, gj.comp_costs[m]);
        BandResult br;
        br.gas = gj.gi; br.band = b; br.ng = ng; br.status = gj.statuses[m]; br.comp_cost = gj.comp_costs[m];
        br.ibegin = gj.ord.ibegin[b]; br.iend = gj.ord.iend[b];
        br.median.resize(ng);
        ck(ecckd_gas_median_sorting_variable(gj.gas, gj.d_sv_sorted.as<double>(), ng, &gj.r1[o], &gj.r2[o], br.median.data()));
        br.rank1.assign(gj.r1.begin() + o, gj.r1.begin() + o + ng);
        br.rank2.assign(gj.r2.begin() + o, gj.r2.begin() + o + ng);
        br.error.assign(gj.error.begin() + o, gj.error.begin() + o + ng);
        for (int k = 0; k < ng; ++k)
          LOG("    g point %d: ranks %lld-%lld, error %g K d-1\n", k, (long long)gj.r1[o + k], (long long)gj.r2[o + k], gj.error[o + k]);
        // the ranks of this band as the search left them (sub-bands and base splits re-rank inside a band)
        br.rank_slice.resize((size_t)(br.iend - br.ibegin + 1));
        ck(ecckd_d2h(dev.ctx(), br.rank_slice.data(), gj.ord.d_rank.as<int32_t>() + br.ibegin, br.rank_slice.size() * sizeof(int32_t)));
        results.push_back(std::move(br));
      }
      if (gj.gas != first_lw_gas) ck(ecckd_gas_destroy(gj.gas));
      gj.gas = nullptr;
      LOG("\n");
    };
    // wait for the searches in flight and finish their gases, in the order of the gas list
    auto flush = [&]() {
      if (search_job) {
        ecckd_gas_search_job* j = search_job;
        search_job = nullptr;
        ck(ecckd_find_g_gases_wait(j));
      }
      for (GasJob& gj : pending) finish_gas(gj);
      pending.clear();
    };

    for (int gi = 0; gi < ngas; ++gi) {
      const std::string& gas_str = gas_list[gi];
      std::string Gas = gas_str;
      std::transform(Gas.begin(), Gas.end(), Gas.begin(), ::toupper);
      LOG("*** FINDING G POINTS FOR %s\n", Gas.c_str());
      const char* scope = gas_str.c_str();
      double min_scaling = 1.0, max_scaling = 1.0;
      config.read(min_scaling, "min_scaling", scope);
      config.read(max_scaling, "max_scaling", scope);
      min_scaling = std::min(0.5, min_scaling);   // :666-667
      max_scaling = std::max(2.5, max_scaling);

      // all gases of the run stay resident until their searches are over: when the device runs short (3 x nlay + 8 rows of
      // doubles per gas, and the spectra it is made from while it is prepared), the searches in flight are finished first
      if (!pending.empty() && nwav > 0) {
        size_t free_b = 0, total_b = 0;
        const double need = 6.0 * 60.0 * 8.0 * (double)nwav;
        ck(ecckd_mem_info(dev.ctx(), &free_b, &total_b));
        if ((double)free_b < need) {           // (blocks parked in the library's allocator count as used: hand them back first)
          ck(ecckd_trim_cache(dev.ctx()));
          ck(ecckd_mem_info(dev.ctx(), &free_b, &total_b));
        }
        if ((double)free_b < need) flush();
      }

      // ---- ordering (:669-683) ----
      pending.emplace_back();
      GasJob& gj = pending.back();
      gj.gi = gi;
      Ordering& ord = gj.ord;
      read_ordering(gas_str, ord);
      if (my_rank == 0) order_rank[gi] = ord.rank;

      // ---- the bands of this gas that fall to this process ----
      int task_begin = 0, task_end = 0;
      deal_tasks(ngas * nband, my_rank, world, task_begin, task_end);
      std::vector<int> mine;
      std::vector<char> is_mine(nband, 0);
      for (int b = 0; b < nband; ++b)
        if (gi * nband + b >= task_begin && gi * nband + b < task_end) {
          mine.push_back(b);
          is_mine[b] = 1;
        }
      if (mine.empty()) {
        LOG("  (searched by other processes)\n\n");
        pending.pop_back();
        continue;
      }
      const int nmine = (int)mine.size();

      // ---- band-specific configuration (:687-771) ----
      std::vector<double> base_wavenumber_boundary;
      config.read(base_wavenumber_boundary, "base_wavenumber_boundary", scope);
      bool have_base_split = false;
      std::vector<double> base_split = per_band<double>(config, gas_str, "base_split", nband, 1.0, &have_base_split);
      if (have_base_split && ord.have_g_split) fail(ECCKD_PARAMETER_ERROR, "Cannot use both g_split and base_split");
      std::vector<int> min_g_points = per_band<int>(config, gas_str, "min_g_points", nband, 1);
      std::vector<int> max_g_points = per_band<int>(config, gas_str, "max_g_points", nband, 256);
      std::vector<double> band_albedo(nband, 0.0);
      double no_rayleigh_limit = -1.0e300;
      for (int b = 0; b < nband; ++b)
        if (band_bound2[b] <= max_no_rayleigh_wavenumber) {
          band_albedo[b] = reference_albedo;
          no_rayleigh_limit = std::max(no_rayleigh_limit, band_bound2[b]);
        }
      std::vector<double> tolerance(nband);   // :762-771
      if ((int)tolerance_in.size() == nband) tolerance = tolerance_in;
      else if (tolerance_in.size() == 1) tolerance.assign(nband, tolerance_in[0]);
      else fail(ECCKD_PARAMETER_ERROR, "heating_rate_tolerance must have either one element or one per band (%d)", nband);

      // ---- sub-bands (:788-870).  The first longwave gas lends its Planck matrix to the later ones, whichever process searches
      // them: its ordering is prepared in every band, so that the matrix does not depend on how the tasks were dealt ----
      subband_setup(ord, (gi == 0 && !do_sw) ? std::vector<char>(nband, 1) : is_mine);
      DevBuf& d_rank = ord.d_rank;
      DevBuf& d_wn = ord.d_wn;
      DevBuf d_dwn;
      const std::vector<int>& nsubband = ord.nsubband;
      const int nsb = ord.nsb;

      // ---- background and target optical depths (:872-915) ----
      Merged bg;
      const bool have_bg = config.exist(gas_str + ".background_input");
      if (have_bg) {
        LOG("Generating background optical depth\n");
        bg = read_merged_spectrum(dev, config, paths, iprofile, gas_str + ".background_");
      }
      LOG("Generating target optical depth\n");
      Merged target = read_merged_spectrum(dev, config, paths, iprofile, gas_str + ".");
      const Spectrum& s = target.first;
      if (s.nwav != nwav) fail(ECCKD_PARAMETER_ERROR, "%s: %zu spectral points, the reordering file has %zu", gas_str.c_str(), s.nwav, nwav);
      if (have_bg && bg.first.nwav != nwav) fail(ECCKD_PARAMETER_ERROR, "%s: background spectra on a different grid", gas_str.c_str());
      wavenumber = s.wavenumber_cm_1;
      d_wn.upload(dev, s.wavenumber_cm_1);
      d_dwn.upload(dev, s.d_wavenumber_cm_1);

      // ---- gas preparation on the device (:917-1150) ----
      LOG("  Reordering and computing the reference fluxes and heating rates\n");
      ecckd_gas* gas = nullptr;
      DevBuf d_ssi, d_albedo;
      if (!do_sw) {
        if (s.temperature_hl.empty()) fail(ECCKD_PARAMETER_ERROR, "temperature_hl missing from the spectrum of %s", gas_str.c_str());
        if (gi > 0 && !d_planck_first) {
          // The reference evaluates the Planck function once, on the FIRST gas's reordered grid (:529, :970-984).  This process
          // does not search the first gas: the matrix is rebuilt from that gas's ordering file and temperatures, bit for bit.
          const std::string first = gas_list[0];
          std::vector<double> bb1 = band_bound1, bb2 = band_bound2, wn_now = wavenumber;
          Ordering o0;
          read_ordering(first, o0);
          subband_setup(o0, std::vector<char>(nband, 1));
          const std::vector<std::string> files0 = config.read_list(first + ".input");
          if (files0.empty()) fail(ECCKD_PARAMETER_ERROR, "%s.input not found", first.c_str());
          const Spectrum s0 = read_spectrum(paths.find(files0[0]), iprofile, false);
          if (s0.nwav != nwav || s0.nlay != s.nlay || s0.temperature_hl.empty())
            fail(ECCKD_PARAMETER_ERROR, "The spectrum of %s does not match that of %s", first.c_str(), gas_str.c_str());
          DevBuf d_wn0, d_dwn0;
          d_wn0.upload(dev, s0.wavenumber_cm_1);
          d_dwn0.upload(dev, s0.d_wavenumber_cm_1);
          d_planck_rebuilt.alloc(dev, (size_t)(s0.nlay + 1) * nwav * sizeof(double));
          ck(ecckd_planck_hl_sorted_dev(dev.ctx(), s0.nlay, nwav, s0.temperature_hl.data(), d_wn0.as<double>(), d_dwn0.as<double>(),
                                        o0.d_rank.as<int32_t>(), d_planck_rebuilt.as<double>()));
          ck(ecckd_synchronize(dev.ctx()));
          d_planck_first = d_planck_rebuilt.as<double>();
          band_bound1 = bb1; band_bound2 = bb2; wavenumber = wn_now;
          nband = (int)band_bound1.size();
          nwav = ord.rank.size();
        }
        ck(ecckd_gas_create_lw(dev.ctx(), s.nlay, nwav, s.pressure_hl.data(), s.temperature_hl.data(), d_wn.as<double>(),
                               d_dwn.as<double>(), d_rank.as<int32_t>(), have_bg ? bg.od_ptr() : nullptr, have_bg ? bg.od_type() : 0,
                               target.od_ptr(), target.od_type(), nwav, method, flux_weight, min_pressure, d_planck_first, &gas));
        if (gi == 0) {
          // the first gas stays alive and lends its matrix
          first_lw_gas = gas;
          size_t rows = 0, cols = 0;
          ck(ecckd_gas_view(gas, "planck_hl", &d_planck_first, &rows, &cols));
        }
      } else {
        if (ssi.size() != nwav) fail(ECCKD_PARAMETER_ERROR, "solar_spectral_irradiance has %zu points, the spectrum %zu", ssi.size(), nwav);
        std::vector<double> albedo(nwav, 0.0);   // :919-923
        for (size_t i = 0; i < nwav; ++i) if (wavenumber[i] < no_rayleigh_limit) albedo[i] = reference_albedo;
        d_ssi.upload(dev, ssi);
        d_albedo.upload(dev, albedo);
        ck(ecckd_gas_create_sw(dev.ctx(), s.nlay, nwav, s.pressure_hl.data(), d_ssi.as<double>(), d_albedo.as<double>(),
                               d_rank.as<int32_t>(), have_bg ? bg.od_ptr() : nullptr, have_bg ? bg.od_type() : 0, target.od_ptr(),
                               target.od_type(), nwav, method, flux_weight, min_pressure, cos_sza, min_scaling, max_scaling, &gas));
      }
      bg = Merged();
      target.d_od.release();
      target.single.buf.release();

      // sorting variable in sorted order (:781): sorted[r] = orig[ireorder[r]]
      gj.gas = gas;
      gj.mine = mine;
      gj.band_albedo = band_albedo;
      DevBuf d_ireorder(dev, nwav * sizeof(int32_t)), d_sv;
      gj.d_sv_sorted.alloc(dev, nwav * sizeof(double));
      d_sv.upload(dev, ord.sorting_variable);
      ck(ecckd_invert_permutation_dev(dev.ctx(), nwav, d_rank.as<int32_t>(), d_ireorder.as<int32_t>()));
      ck(ecckd_gather_f64_dev(dev.ctx(), nwav, d_sv.as<double>(), d_ireorder.as<int32_t>(), gj.d_sv_sorted.as<double>()));
      ck(ecckd_synchronize(dev.ctx()));

      // ---- the bands (:1152-1414) ----
      // Options of every band first: the bands are then searched side by side (ecckd_find_g_bands_ex); a shortwave band brings
      // its albedo (init_sw(..., band_albedo(jband), ...), :1177) with it.
      gj.opts.resize(nmine);
      gj.wn_bounds.resize(nmine);
      gj.ib.resize(nmine); gj.ie.resize(nmine); gj.tol.resize(nmine);
      for (int m = 0; m < nmine; ++m) {
        const int b = mine[m];
        if (ord.ibegin[b] < 0) fail(ECCKD_PARAMETER_ERROR, "Band %d contains no wavenumbers", b);
        gj.ib[m] = (size_t)ord.ibegin[b];
        gj.ie[m] = (size_t)ord.iend[b];
        gj.tol[m] = tolerance[b];
        ecckd_band_options& opt = gj.opts[m];
        std::memset(&opt, 0, sizeof opt);
        opt.min_g_points = min_g_points[b];
        opt.max_g_points = max_g_points[b];
        if (nsubband[b] > 1) {
          opt.nsubband = nsubband[b];
          opt.isubband1 = &ord.isubband1[(size_t)b * nsb];
          opt.isubband2 = &ord.isubband2[(size_t)b * nsb];
          opt.iupperindex = ord.iupperindex[b];
          opt.g_split = ord.g_split[b];
        }
        opt.base_split = base_split[b];
        opt.band_albedo = do_sw ? band_albedo[b] : 0.0;
        std::vector<double>& wn_bound = gj.wn_bounds[m];
        std::vector<double> interior;
        for (double w : base_wavenumber_boundary) if (w > band_bound1[b] && w < band_bound2[b]) interior.push_back(w);
        if (base_split[b] != 1.0 || !interior.empty()) {   // :1268-1301
          wn_bound.push_back(band_bound1[b]);
          wn_bound.insert(wn_bound.end(), interior.begin(), interior.end());
          wn_bound.push_back(band_bound2[b] + 1.0);
          opt.nbase_wn_bound = (int)wn_bound.size();
          opt.base_wn_bound = wn_bound.data();
          opt.d_wavenumber = d_wn.as<double>();
          opt.d_rank = d_rank.as<int32_t>();
          opt.nwav = nwav;
        }
      }
      gj.ngs.assign(nmine, 0); gj.statuses.assign(nmine, 0);
      gj.comp_costs.assign(nmine, 0.0);
      gj.bounds.assign((size_t)nmine * (band_capacity + 1), 0.0); gj.error.assign((size_t)nmine * band_capacity, 0.0);
      gj.r1.assign((size_t)nmine * band_capacity, 0); gj.r2.assign((size_t)nmine * band_capacity, 0);
      if (sequential_bands) {
        // one band at a time, here and now
        for (int m = 0; m < nmine; ++m) {
          const size_t o = (size_t)m * band_capacity;
          if (do_sw) ck(ecckd_gas_set_band_albedo(gas, band_albedo[mine[m]]));
          ck(ecckd_find_g_band_ex(gas, gj.ib[m], gj.ie[m], gj.tol[m], tolerance_tolerance, max_iterations, &gj.opts[m], &gj.ngs[m],
                                  &gj.bounds[(size_t)m * (band_capacity + 1)], &gj.error[o], &gj.r1[o], &gj.r2[o], band_capacity, &gj.statuses[m],
                                  &gj.comp_costs[m]));
        }
        flush();
      } else {
        // the bands of this gas side by side, next to the gases whose searches are still running
        ecckd_gas_search& rq = gj.req;
        std::memset(&rq, 0, sizeof rq);
        rq.gas = gas; rq.nband = nmine; rq.ibegin = gj.ib.data(); rq.iend = gj.ie.data(); rq.heating_rate_tolerance = gj.tol.data();
        rq.opt = gj.opts.data(); rq.ng = gj.ngs.data(); rq.bounds = gj.bounds.data(); rq.error = gj.error.data();
        rq.rank1 = gj.r1.data(); rq.rank2 = gj.r2.data(); rq.capacity = band_capacity; rq.status = gj.statuses.data();
        rq.comp_cost = gj.comp_costs.data();
        if (!search_job) ck(ecckd_find_g_gases_begin(tolerance_tolerance, max_iterations, gases_side_by_side, &search_job));
        ck(ecckd_find_g_gases_add(search_job, &rq));
        if (gases_side_by_side == 1) flush();
      }
      LOG("\n");
    }
    flush();
    if (first_lw_gas) ck(ecckd_gas_destroy(first_lw_gas));
    d_planck_rebuilt.release();

    // ---- several processes: the others leave their searches for process 0 ----
    double my_cost = 0.0;
    for (const BandResult& r : results) for (double e : r.error) my_cost += e;
    if (world > 1) LOG("Process %d: %zu searches, sum of the g points' errors %.17g K d-1\n", my_rank, results.size(), my_cost);
    const uint64_t identity = job_identity(run_id, ngas, nband, (uint64_t)nwav);
    if (my_rank != 0) {
      write_part(output + ".part" + std::to_string(my_rank), results, identity);
      return done(0);
    }
    // extension key: how long process 0 waits for a part that has not appeared, in seconds.  A process that FAILS through an
    // error path says so through its marker at once; one that is killed (a signal, the out-of-memory killer, a HIP abort) leaves
    // none, so the wait is what bounds that case - the reference-sized default stays an hour
    double part_timeout = 3600.0;
    config.read(part_timeout, "part_timeout");
    for (int r = 1; r < world; ++r) {
      const std::string path = output + ".part" + std::to_string(r);
      std::vector<BandResult> part = read_part(path, part_timeout, run_id, identity);
      for (BandResult& br : part) results.push_back(std::move(br));
      std::remove(path.c_str());
    }
    std::vector<const BandResult*> by_task((size_t)ngas * nband, nullptr);
    for (const BandResult& br : results) {
      if (br.gas < 0 || br.gas >= ngas || br.band < 0 || br.band >= nband || by_task[(size_t)br.gas * nband + br.band])
        fail(ECCKD_PROCESSING_ERROR, "Search of gas %d, band %d reported twice or out of range", br.gas, br.band);
      by_task[(size_t)br.gas * nband + br.band] = &br;
    }
    double final_cost = 0.0;
    std::vector<GasResult> gases(ngas);
    for (int gi = 0; gi < ngas; ++gi) {
      GasResult& res = gases[gi];
      res.molecule = gas_list[gi];
      std::vector<int32_t>& rank = order_rank[gi];
      for (int b = 0; b < nband; ++b) {
        const BandResult* br = by_task[(size_t)gi * nband + b];
        if (!br) fail(ECCKD_PROCESSING_ERROR, "No process searched band %d of %s", b, gas_list[gi].c_str());
        if (br->ibegin < 0 || br->iend >= (int64_t)nwav || (int64_t)br->rank_slice.size() != br->iend - br->ibegin + 1)
          fail(ECCKD_PROCESSING_ERROR, "Rank slice of band %d of %s does not fit", b, gas_list[gi].c_str());
        std::copy(br->rank_slice.begin(), br->rank_slice.end(), rank.begin() + br->ibegin);
        res.n_g_points.push_back(br->ng);
        for (int k = 0; k < br->ng; ++k) {
          res.band_number.push_back(b);
          res.rank1.push_back(br->rank1[k]);
          res.rank2.push_back(br->rank2[k]);
          res.error.push_back(br->error[k]);
          res.sorting_variable.push_back(br->median[k]);
          final_cost += br->error[k];
        }
      }
      // SingleGasData::store_g_points (single_gas_data.h:56-62) with the (possibly re-ranked) ranks
      const int ngp = (int)res.rank1.size();
      std::vector<int32_t> gr1(res.rank1.begin(), res.rank1.end()), gr2(res.rank2.begin(), res.rank2.end());
      DevBuf d_rank;
      d_rank.upload(dev, rank);
      res.d_g_point.alloc(dev, nwav * sizeof(int32_t));
      ck(ecckd_gas_g_point_dev(dev.ctx(), nwav, d_rank.as<int32_t>(), ngp, gr1.data(), gr2.data(), res.d_g_point.as<int32_t>()));
      ck(ecckd_synchronize(dev.ctx()));
      std::vector<int32_t>().swap(rank);
    }
    results.clear();
    if (world > 1) LOG("Final cost (sum of the g points' errors over all processes): %.17g K d-1\n", final_cost);

    // ---- spectral overlap of the gases (:1452-1483) ----
    LOG("*** COMPUTING SPECTRAL OVERLAP OF GASES\n");
    std::vector<int> n_g_points, gas_offset;
    std::vector<double> sorting_all;
    int capacity = 1;
    for (const GasResult& g : gases) {
      n_g_points.insert(n_g_points.end(), g.n_g_points.begin(), g.n_g_points.end());
      gas_offset.push_back((int)sorting_all.size());
      sorting_all.insert(sorting_all.end(), g.sorting_variable.begin(), g.sorting_variable.end());
      capacity += (int)g.rank1.size();
    }
    int ng = 0;
    std::vector<int> band_number(capacity), g_min((size_t)ngas * capacity), g_max((size_t)ngas * capacity);
    ck(ecckd_overlap_g_points(ngas, nband, n_g_points.data(), gas_offset.data(), sorting_all.data(), capacity, &ng, band_number.data(),
                              g_min.data(), g_max.data()));
    band_number.resize(ng);
    std::vector<const int32_t*> d_gp;
    for (int k = 0; k < ngas; ++k) {
      gases[k].g_min.assign(g_min.begin() + (size_t)k * capacity, g_min.begin() + (size_t)k * capacity + ng);
      gases[k].g_max.assign(g_max.begin() + (size_t)k * capacity, g_max.begin() + (size_t)k * capacity + ng);
      d_gp.push_back(gases[k].d_g_point.as<int32_t>());
    }
    DevBuf d_g_point(dev, nwav * sizeof(int32_t));
    int64_t n_unassigned = 0;
    ck(ecckd_merge_g_points_dev(dev.ctx(), nwav, ngas, d_gp.data(), ng, capacity, g_min.data(), g_max.data(), d_g_point.as<int32_t>(),
                                &n_unassigned));
    std::vector<int32_t> g_point = d_g_point.download<int32_t>();
    if (n_unassigned > 0) WARN("%lld wavenumbers are not assigned to a g point", (long long)n_unassigned);
    LOG("%d g points in total\n", ng);

    // ---- the g-points file (:1485-1660) ----
    LOG("Writing %s\n", output.c_str());
    NcOut file(output);
    file.dim("band", nband);
    if (ng > 0) file.dim("g_point", ng);
    std::string molecule_list;
    for (const GasResult& g : gases) {
      file.dim(g.molecule + "_g_point", g.rank1.size());
      molecule_list += (molecule_list.empty() ? "" : " ") + g.molecule;
    }
    file.dim("wavenumber", nwav);
    file.var("n_gases", NC_INT_T, {}, "Number of gases treated");
    file.var("wavenumber1_band", NC_FLOAT_T, {"band"}, "Lower wavenumber bound of band", "cm-1");
    file.var("wavenumber2_band", NC_FLOAT_T, {"band"}, "Upper wavenumber bound of band", "cm-1");
    file.var("band_number", NC_SHORT_T, {"g_point"}, "Band number of each g point");
    if (do_sw) file.var("solar_irradiance", NC_FLOAT_T, {"g_point"}, "Solar irradiance across each g point", "W m-2");
    for (const GasResult& g : gases) {
      const std::string m = g.molecule, d = m + "_g_point";
      file.var(m + "_n_g_points", NC_INT_T, {"band"}, "Number of g points in each band");
      file.var(m + "_band_number", NC_SHORT_T, {d}, "Band number of each g point");
      file.var(m + "_rank1", NC_INT_T, {d}, "Rank of first wavenumber in each g point");
      file.var(m + "_rank2", NC_INT_T, {d}, "Rank of last wavenumber in each g point");
      file.var(m + "_error", NC_FLOAT_T, {d}, "Root-mean-squared heating-rate error of each g point", "K d-1");
      file.var(m + "_sorting_variable", NC_FLOAT_T, {d}, "Median sorting variable of each g point");
      file.var(m + "_g_min", NC_INT_T, {"g_point"}, "First single-gas g point contributing to each merged g point");
      file.var(m + "_g_max", NC_INT_T, {"g_point"}, "Last single-gas g point contributing to each merged g point");
    }
    file.var("wavenumber", NC_DOUBLE_T, {"wavenumber"}, "Wavenumber", "cm-1");
    file.var("g_point", NC_SHORT_T, {"wavenumber"}, "G point of each wavenumber");
    file.deflate("g_point");                                               // find_g_points.cpp:1580
    for (const GasResult& g : gases) {
      file.var(g.molecule + "_g_point", NC_SHORT_T, {"wavenumber"}, "Single-gas g point of each wavenumber");
      file.deflate(g.molecule + "_g_point");                               // :1587
    }
    file.att(do_sw ? "Definition of the spectral intervals of a shortwave CKD model"
                   : "Definition of the spectral intervals of a longwave CKD model", "title");
    file.att(molecule_list, "constituent_id");
    file.att(history_line(argc, argv), "history");
    file.att(config.str(), "config");
    file.end_define();
    file.write("n_gases", {(double)ngas});
    file.write("wavenumber1_band", band_bound1);
    file.write("wavenumber2_band", band_bound2);
    file.write_as_double("band_number", band_number);
    if (do_sw) {   // :1620-1633
      std::vector<double> solar(ng, 0.0);
      for (size_t i = 0; i < nwav; ++i) if (g_point[i] >= 0) solar[g_point[i]] += ssi[i];
      int nbad = 0;
      for (double v : solar) if (v <= 0.0) ++nbad;
      if (nbad) WARN("%d g points have no solar irradiance", nbad);
      file.write("solar_irradiance", solar);
    }
    for (GasResult& g : gases) {
      const std::string m = g.molecule;
      file.write_as_double(m + "_n_g_points", g.n_g_points);
      file.write_as_double(m + "_band_number", g.band_number);
      file.write_as_double(m + "_rank1", g.rank1);
      file.write_as_double(m + "_rank2", g.rank2);
      file.write(m + "_error", g.error);
      file.write(m + "_sorting_variable", g.sorting_variable);
      file.write_as_double(m + "_g_min", g.g_min);
      file.write_as_double(m + "_g_max", g.g_max);
      file.write_as_double(m + "_g_point", g.d_g_point.download<int32_t>());
    }
    file.write("wavenumber", wavenumber);
    file.write_as_double("g_point", g_point);
    file.close();
    return done(0);
  });
}
