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
#include <algorithm>
#include <cctype>

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

    Device dev;
    std::vector<GasResult> gases;
    std::vector<double> band_bound1, band_bound2, wavenumber;
    int nband = 0;
    size_t nwav = 0;
    ecckd_gas* first_lw_gas = nullptr;
    const double* d_planck_first = nullptr;

    for (const std::string& gas_str : config.read_list("gases")) {
      std::string Gas = gas_str;
      std::transform(Gas.begin(), Gas.end(), Gas.begin(), ::toupper);
      LOG("*** FINDING G POINTS FOR %s\n", Gas.c_str());
      const char* scope = gas_str.c_str();
      double min_scaling = 1.0, max_scaling = 1.0;
      config.read(min_scaling, "min_scaling", scope);
      config.read(max_scaling, "max_scaling", scope);
      min_scaling = std::min(0.5, min_scaling);   // :666-667
      max_scaling = std::max(2.5, max_scaling);

      // ---- ordering (:669-683) ----
      std::string reordering_input;
      if (!config.read(reordering_input, "reordering_input", scope)) fail(ECCKD_PARAMETER_ERROR, "No reordering_input found");
      LOG("Reading %s\n", reordering_input.c_str());
      std::vector<int32_t> rank;
      std::vector<int> iband;
      std::vector<double> sorting_variable;
      {
        NcIn f(paths.find(reordering_input));
        std::vector<double> r = f.read("rank"), b = f.read("band_number");
        rank.assign(r.begin(), r.end());
        iband.assign(b.begin(), b.end());
        band_bound1 = f.read("wavenumber1_band");
        band_bound2 = f.read("wavenumber2_band");
        sorting_variable = f.read("sorting_variable");
        wavenumber = f.read("wavenumber");
      }
      nband = (int)band_bound1.size();
      nwav = rank.size();

      // ---- band-specific configuration (:687-771) ----
      std::vector<double> base_wavenumber_boundary, subband_wavenumber_boundary;
      config.read(base_wavenumber_boundary, "base_wavenumber_boundary", scope);
      bool have_g_split = false;
      std::vector<double> g_split = per_band<double>(config, gas_str, "g_split", nband, -1.0, &have_g_split);
      if (have_g_split && !config.read(subband_wavenumber_boundary, "subband_wavenumber_boundary", scope))
        fail(ECCKD_PARAMETER_ERROR, "g_split must be accompanied by subband_wavenumber_boundary");
      bool have_base_split = false;
      std::vector<double> base_split = per_band<double>(config, gas_str, "base_split", nband, 1.0, &have_base_split);
      if (have_base_split && have_g_split) fail(ECCKD_PARAMETER_ERROR, "Cannot use both g_split and base_split");
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

      // first / last sorted index of every band: the members of a band are contiguous in rank
      std::vector<int64_t> ibegin(nband, -1), iend(nband, -1);
      for (size_t i = 0; i < nwav; ++i) {
        const int b = iband[i];
        if (b < 0 || b >= nband) continue;
        if (ibegin[b] < 0) ibegin[b] = (int64_t)i;
        iend[b] = (int64_t)i;
      }

      DevBuf d_rank, d_wn, d_dwn;
      d_rank.upload(dev, rank);
      d_wn.upload(dev, wavenumber);

      // ---- sub-bands of the optically thin part of a band (:788-870): re-ranks d_rank before anything is reordered ----
      std::vector<int> nsubband(nband, 0);
      std::vector<int64_t> iupperindex(nband, -1);
      const int nsb = (int)subband_wavenumber_boundary.size() + 1;
      std::vector<int64_t> isubband1((size_t)nband * nsb, -1), isubband2((size_t)nband * nsb, -1);
      if (have_g_split)
        for (int b = 0; b < nband; ++b)
          if (g_split[b] > 0.0 && ibegin[b] >= 0)
            ck(ecckd_subband_setup_dev(dev.ctx(), nwav, d_wn.as<double>(), d_rank.as<int32_t>(), ibegin[b], iend[b], g_split[b],
                                       band_bound1[b], band_bound2[b], (int)subband_wavenumber_boundary.size(),
                                       subband_wavenumber_boundary.data(), &nsubband[b], &isubband1[(size_t)b * nsb],
                                       &isubband2[(size_t)b * nsb], &iupperindex[b]));

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
        ck(ecckd_gas_create_lw(dev.ctx(), s.nlay, nwav, s.pressure_hl.data(), s.temperature_hl.data(), d_wn.as<double>(),
                               d_dwn.as<double>(), d_rank.as<int32_t>(), have_bg ? bg.od_ptr() : nullptr, have_bg ? bg.od_type() : 0,
                               target.od_ptr(), target.od_type(), nwav, method, flux_weight, min_pressure, d_planck_first, &gas));
        if (!first_lw_gas) {
          // The reference evaluates the Planck function once, on the FIRST gas's reordered grid, and keeps using
          // that matrix for the later gases (:529, :970-984).  Reproduced: the first gas stays alive and lends it.
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
      DevBuf d_ireorder(dev, nwav * sizeof(int32_t)), d_sv, d_sv_sorted(dev, nwav * sizeof(double));
      d_sv.upload(dev, sorting_variable);
      ck(ecckd_invert_permutation_dev(dev.ctx(), nwav, d_rank.as<int32_t>(), d_ireorder.as<int32_t>()));
      ck(ecckd_gather_f64_dev(dev.ctx(), nwav, d_sv.as<double>(), d_ireorder.as<int32_t>(), d_sv_sorted.as<double>()));
      ck(ecckd_synchronize(dev.ctx()));

      // ---- the bands (:1152-1414) ----
      GasResult res;
      res.molecule = gas_str;
      const int capacity = 1024;
      // Options of every band first: the bands are then searched side by side (ecckd_find_g_bands_ex); a shortwave band brings
      // its albedo (init_sw(..., band_albedo(jband), ...), :1177) with it.
      std::vector<ecckd_band_options> opts(nband);
      std::vector<std::vector<double>> wn_bounds(nband);
      for (int b = 0; b < nband; ++b) {
        if (ibegin[b] < 0) fail(ECCKD_PARAMETER_ERROR, "Band %d contains no wavenumbers", b);
        ecckd_band_options& opt = opts[b];
        std::memset(&opt, 0, sizeof opt);
        opt.min_g_points = min_g_points[b];
        opt.max_g_points = max_g_points[b];
        if (nsubband[b] > 1) {
          opt.nsubband = nsubband[b];
          opt.isubband1 = &isubband1[(size_t)b * nsb];
          opt.isubband2 = &isubband2[(size_t)b * nsb];
          opt.iupperindex = iupperindex[b];
          opt.g_split = g_split[b];
        }
        opt.base_split = base_split[b];
        opt.band_albedo = do_sw ? band_albedo[b] : 0.0;
        std::vector<double>& wn_bound = wn_bounds[b];
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
      std::vector<int> ngs(nband, 0), statuses(nband, 0);
      std::vector<double> comp_costs(nband, 0.0);
      std::vector<double> bounds((size_t)nband * (capacity + 1)), error((size_t)nband * capacity);
      std::vector<int64_t> r1((size_t)nband * capacity), r2((size_t)nband * capacity);
      bool sequential_bands = false;                 // extension key: the reference's one-band-at-a-time order of evaluation
      config.read(sequential_bands, "sequential_bands");
      const bool side_by_side = nband > 1 && !sequential_bands;
      if (side_by_side) {
        std::vector<size_t> ib(ibegin.begin(), ibegin.end()), ie(iend.begin(), iend.end());
        ck(ecckd_find_g_bands_ex(gas, nband, ib.data(), ie.data(), tolerance.data(), tolerance_tolerance, max_iterations, opts.data(),
                                 ngs.data(), bounds.data(), error.data(), r1.data(), r2.data(), capacity, statuses.data(), comp_costs.data()));
      }
      for (int b = 0; b < nband; ++b) {
        LOG("  Band %d: %g-%g cm-1\n", b, band_bound1[b], band_bound2[b]);
        const size_t o = (size_t)b * capacity;
        if (!side_by_side) {
          if (do_sw) ck(ecckd_gas_set_band_albedo(gas, band_albedo[b]));
          ck(ecckd_find_g_band_ex(gas, (size_t)ibegin[b], (size_t)iend[b], tolerance[b], tolerance_tolerance, max_iterations, &opts[b], &ngs[b],
                                  &bounds[(size_t)b * (capacity + 1)], &error[o], &r1[o], &r2[o], capacity, &statuses[b], &comp_costs[b]));
        }
        const int ng = ngs[b];
        LOG("    %s: %d g points, computational cost = %g\n", ecckd_partition_status_string(statuses[b]), ng, comp_costs[b]);
        std::vector<double> med(ng);
        ck(ecckd_gas_median_sorting_variable(gas, d_sv_sorted.as<double>(), ng, &r1[o], &r2[o], med.data()));
        res.n_g_points.push_back(ng);
        for (int k = 0; k < ng; ++k) {
          res.band_number.push_back(b);
          res.rank1.push_back(r1[o + k]);
          res.rank2.push_back(r2[o + k]);
          res.error.push_back(error[o + k]);
          res.sorting_variable.push_back(med[k]);
          LOG("    g point %d: ranks %lld-%lld, error %g K d-1\n", k, (long long)r1[o + k], (long long)r2[o + k], error[o + k]);
        }
      }
      if (gas != first_lw_gas) ck(ecckd_gas_destroy(gas));
      // SingleGasData::store_g_points (single_gas_data.h:56-62) with the (possibly re-ranked) ranks
      const int ngp = (int)res.rank1.size();
      std::vector<int32_t> gr1(res.rank1.begin(), res.rank1.end()), gr2(res.rank2.begin(), res.rank2.end());
      res.d_g_point.alloc(dev, nwav * sizeof(int32_t));
      ck(ecckd_gas_g_point_dev(dev.ctx(), nwav, d_rank.as<int32_t>(), ngp, gr1.data(), gr2.data(), res.d_g_point.as<int32_t>()));
      ck(ecckd_synchronize(dev.ctx()));
      gases.push_back(std::move(res));
      LOG("\n");
    }
    if (first_lw_gas) ck(ecckd_gas_destroy(first_lw_gas));
    const int ngas = (int)gases.size();
    if (ngas == 0) fail(ECCKD_PARAMETER_ERROR, "No gases specified in \"gases\"");

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
    for (const GasResult& g : gases) file.var(g.molecule + "_g_point", NC_SHORT_T, {"wavenumber"}, "Single-gas g point of each wavenumber");
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
    return 0;
  });
}
