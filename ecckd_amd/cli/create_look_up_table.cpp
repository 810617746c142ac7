// create_look_up_table [key=value ...] [file.cfg]
//
// Drop-in for the reference executable of the same name (src/ecckd/create_look_up_table.cpp:27-606): averages
// the high-resolution optical depths of every gas to the g points of a g-points file, for every temperature
// column (and mole fraction) of the training spectra, and writes the raw CKD-definition file.
// Keys (:38-82, :243-246, :259, :364): input (g-points file), output, ssi, temperature_stride, averaging_method,
// gases, prepend_path, append_path, log_level; per gas conc_dependence (none | linear | lut | relative-linear),
// input (+ scaling / conc for "none": read_merged_spectrum), reference_conc.
// base_wavenumber_boundary (:63-65, :160-223) splits the base g point of the bands it falls into by wavenumber; when that
// or the removal of empty g points changes the numbering, the g point of every wavenumber is stored in the output.
// The averaging, the g-point fractions and the Planck look-up table run on the GPU (ecckd_gmap_*).
#include <algorithm>

#include "ckd_file.hpp"

using namespace tool;

int main(int argc, char** argv) {
  return run(argc, argv, [&](Config& config) -> int {
    SearchPath paths;
    paths.configure(config);
    std::string input, output, ssi_file_name;
    if (!config.read(output, "output")) fail(ECCKD_PARAMETER_ERROR, "\"output\" file not specified");
    if (!config.read(input, "input")) fail(ECCKD_PARAMETER_ERROR, "\"input\" file not specified");
    std::vector<double> base_wavenumber_boundary;
    config.read(base_wavenumber_boundary, "base_wavenumber_boundary");
    std::vector<double> ssi, ssi_wavenumber;
    double tsi = -1.0;
    const bool do_sw = config.read(ssi_file_name, "ssi");
    if (do_sw) {
      NcIn f(paths.find(ssi_file_name));
      ssi = f.read("solar_spectral_irradiance");
      tsi = f.read_scalar("total_solar_irradiance");
      ssi_wavenumber = f.read("wavenumber");
    }

    CkdFile model;
    std::vector<int32_t> g_point;
    std::vector<int> band_number;
    {
      LOG("Reading %s\n", input.c_str());
      NcIn f(paths.find(input));
      if (!f.exist("g_point")) fail(ECCKD_PARAMETER_ERROR, "\"g_point\" not found in \"%s\"", input.c_str());
      for (double v : f.read("g_point")) g_point.push_back((int32_t)v);
      model.wavenumber1_band = f.read("wavenumber1_band");
      model.wavenumber2_band = f.read("wavenumber2_band");
      for (double v : f.read("band_number")) band_number.push_back((int)v);
      if (f.exist("solar_irradiance")) { model.solar_irradiance = f.read("solar_irradiance"); model.is_sw = true; }
      f.att_text("history", model.history);
      f.att_text("config", model.config);
    }
    const size_t nwav = g_point.size();
    int ng = *std::max_element(g_point.begin(), g_point.end()) + 1;

    // ---- g points that occupy none of the spectrum are removed (:108-168) ----
    {
      std::vector<char> present(ng, 0);
      for (int32_t g : g_point) if (g >= 0) present[g] = 1;
      std::vector<int> kept;
      for (int g = 0; g < ng; ++g) {
        if (present[g]) kept.push_back(g);
        else WARN("g point %d occupies none of the spectrum: removing", g);
      }
      if ((int)kept.size() != ng) {
        std::vector<int> lookup(ng, -1);
        for (size_t k = 0; k < kept.size(); ++k) lookup[kept[k]] = (int)k;
        for (int32_t& g : g_point) {
          if (g < 0) { std::fprintf(stderr, "*** Error: Some unassigned spectral points after mapping\n"); return 1; }   // THROW(1), :146-149
          g = lookup[g];
        }
        band_number.assign(kept.begin(), kept.end());     // sic (:142): the old g-point index, not the band
        if (model.is_sw) {
          std::vector<double> s;
          for (int g : kept) s.push_back(model.solar_irradiance[g]);
          model.solar_irradiance = s;
        }
        ng = (int)kept.size();
        model.save_g_points = true;      // :575-577: the numbering no longer matches the g-points file
      }
    }

    // ---- base g points split by wavenumber (:160-223); the high-resolution wavenumbers are those of the ssi file ----
    if (!base_wavenumber_boundary.empty()) {
      if (!do_sw) fail(ECCKD_PARAMETER_ERROR, "base_wavenumber_boundary needs the \"ssi\" file (its wavenumbers and irradiances, :198-214)");
      if (ssi_wavenumber.size() != nwav) fail(ECCKD_PARAMETER_ERROR, "the ssi file has %zu wavenumbers, the g-points file %zu", ssi_wavenumber.size(), nwav);
      const int nband = (int)model.wavenumber1_band.size();
      for (int iband = 0; iband < nband; ++iband) {
        std::vector<double> inner;
        for (double w : base_wavenumber_boundary) if (w > model.wavenumber1_band[iband] && w < model.wavenumber2_band[iband]) inner.push_back(w);
        if (inner.empty()) continue;
        const int m = (int)inner.size();
        LOG("Splitting base g-point of band %d into %d\n", iband, m + 1);
        int ig = -1;
        for (int g = 0; g < ng; ++g) if (band_number[g] == iband) { ig = g; break; }
        if (ig < 0) fail(ECCKD_PARAMETER_ERROR, "band %d has no g points", iband);
        const int new_ng = ng + m;
        std::vector<int> new_band(new_ng, iband);
        std::vector<double> new_solar(new_ng, 0.0);
        for (int g = 0; g <= ig; ++g) new_band[g] = band_number[g];
        for (int g = 0; g < ig; ++g) new_solar[g] = model.solar_irradiance[g];
        for (int g = ig + 1; g < ng; ++g) { new_band[g + m] = band_number[g]; new_solar[g + m] = model.solar_irradiance[g]; }
        std::vector<double> bounds;
        bounds.push_back(model.wavenumber1_band[iband]);
        bounds.insert(bounds.end(), inner.begin(), inner.end());
        bounds.push_back(model.wavenumber2_band[iband]);
        for (size_t i = 0; i < nwav; ++i) {
          const int g = g_point[i];
          if (g > ig) g_point[i] = g + m;
          else if (g == ig) {
            const double w = ssi_wavenumber[i];
            for (int k = 0; k <= m; ++k) if (w >= bounds[k] && w < bounds[k + 1]) { g_point[i] = ig + k; break; }
          }
        }
        for (size_t i = 0; i < nwav; ++i) if (g_point[i] >= ig && g_point[i] <= ig + m) new_solar[g_point[i]] += ssi[i];
        band_number = new_band;
        model.solar_irradiance = new_solar;
        ng = new_ng;
      }
      model.save_g_points = true;
    }
    model.ng = ng;
    model.band_number = band_number;

    int temperature_stride = 1;
    config.read(temperature_stride, "temperature_stride");
    std::string averaging_method = "transmission";
    config.read(averaging_method, "averaging_method");
    static const char* const methods[] = {"linear", "transmission", "transmission-2", "square-root", "logarithmic",
                                          "total-transmission", "transmission-3", "transmission-10", "hybrid-logarithmic-transmission-3"};
    int method = -1;
    for (int k = 0; k < 9; ++k) if (averaging_method == methods[k]) method = k;
    // average_optical_depth_to_g_point (average_optical_depth.cpp:43-133) has no total-transmission branch
    if (method < 0 || averaging_method == "total-transmission")
      fail(ECCKD_PARAMETER_ERROR, "averaging_method \"%s\" not understood", averaging_method.c_str());

    Device dev;
    ecckd_gmap* gmap = nullptr;
    DevBuf d_g_point, d_wn, d_dwn, d_ssi;
    d_g_point.upload(dev, g_point);
    if (do_sw) {
      if (ssi.size() != nwav) fail(ECCKD_PARAMETER_ERROR, "solar_spectral_irradiance has %zu points, the g-points file %zu", ssi.size(), nwav);
      d_ssi.upload(dev, ssi);
    }
    int nlay = 0;

    // one temperature column of one gas -> rows [nlay][ng] of the three tables; returns the number of columns in the file
    auto column = [&](const Spectrum& s, const void* d_od, int od_type, double ref_vmr, GasTable& gas, size_t offset,
                      std::vector<double>& t_fl) {
      if (!gmap) {
        if (s.nwav != nwav) fail(ECCKD_PARAMETER_ERROR, "spectra have %zu points, the g-points file %zu", s.nwav, nwav);
        if (model.save_g_points) {
          model.wavenumber_hr = s.wavenumber_cm_1;
          model.g_point_hr.assign(g_point.begin(), g_point.end());
        }
        d_wn.upload(dev, s.wavenumber_cm_1);
        d_dwn.upload(dev, s.d_wavenumber_cm_1);
        ck(ecckd_gmap_create(dev.ctx(), nwav, d_g_point.as<int32_t>(), ng, d_wn.as<double>(), d_dwn.as<double>(), &gmap));
        nlay = s.nlay;
        model.np = nlay;
        model.pressure.resize(nlay);
        for (int l = 0; l < nlay; ++l) model.pressure[l] = 0.5 * (s.pressure_hl[l] + s.pressure_hl[l + 1]);
      }
      t_fl.resize(nlay);   // :310-311
      for (int l = 0; l < nlay; ++l)
        t_fl[l] = 0.5 * (s.temperature_hl[l] * s.pressure_hl[l] + s.temperature_hl[l + 1] * s.pressure_hl[l + 1]) / model.pressure[l];
      LOG(do_sw ? "  Solar-spectrum-weighted averaging optical depths for each g point\n"
                : "  Planck-weighted averaging optical depths for each g point\n");
      ck(ecckd_average_to_gpoints(gmap, nlay, s.pressure_hl.data(), do_sw ? nullptr : t_fl.data(), do_sw ? d_ssi.as<double>() : nullptr,
                                  d_od, od_type, nwav, method, ref_vmr, &gas.molar_abs[offset], &gas.min_molar_abs[offset],
                                  &gas.max_molar_abs[offset]));
    };

    for (const std::string& gas_str : config.read_list("gases")) {
      LOG("Creating look-up table for %s (gas number %zu)\n  Averaging method = %s\n", upper(gas_str).c_str(), model.gases.size(),
          averaging_method.c_str());
      const char* scope = gas_str.c_str();
      GasTable gas;
      gas.name = gas_str;
      std::string dep;
      if (!config.read(dep, "conc_dependence", scope)) fail(ECCKD_PARAMETER_ERROR, "%s.conc_dependence not found in configuration", scope);
      if (dep == "none") gas.conc = CONC_NONE;
      else if (dep == "linear") gas.conc = CONC_LINEAR;
      else if (dep == "lut") gas.conc = CONC_LUT;
      else if (dep == "relative-linear") gas.conc = CONC_RELATIVE_LINEAR;
      else fail(ECCKD_PARAMETER_ERROR, "conc_dependence \"%s\" not understood", dep.c_str());
      std::vector<std::string> files = config.read_list("input", scope);
      if (files.empty()) fail(ECCKD_PARAMETER_ERROR, "%s.input not found", scope);
      if (gas.conc == CONC_RELATIVE_LINEAR && !config.read(gas.reference_vmr, "reference_conc", scope))
        fail(ECCKD_PARAMETER_ERROR, "%s.reference_conc must be provided if conc_dependence is relative-linear", scope);
      const size_t nconc = gas.conc == CONC_LUT ? files.size() : 1;
      int ncol = 1;
      std::vector<double> t_fl;
      for (size_t iconc = 0; iconc < nconc; ++iconc) {
        for (int icol = 0; icol < ncol; ++icol) {
          const int iprofile = icol * temperature_stride;
          Spectrum first;
          Merged merged;
          DevOd od;
          double ref_vmr = 1.0;   // :283
          const void* d_od = nullptr;
          int od_type = ECCKD_F64;
          if (gas.conc == CONC_NONE) {
            LOG("  Reading temperature profile %d for %s\n", iprofile, scope);
            merged = read_merged_spectrum(dev, config, paths, iprofile, gas_str + ".");
            d_od = merged.od_ptr();
            od_type = merged.od_type();
            if (icol == 0) {   // :311-313
              gas.composite_molecules = merged.molecules;
              gas.composite_vmr.clear();
              for (const std::vector<double>& row : merged.vmr_fl) gas.composite_vmr.insert(gas.composite_vmr.end(), row.begin(), row.end());
            }
          } else {
            const std::string path = paths.find(files[iconc]);
            LOG("  Reading temperature profile %d from %s\n", iprofile, path.c_str());
            first = read_spectrum(path, iprofile, false);
            ref_vmr = first.reference_surface_vmr;
            if (gas.conc == CONC_LUT && ref_vmr < 0.0)
              fail(ECCKD_PARAMETER_ERROR, "Invalid reference_surface_vmr for constructing VMR-dependent look-up table");
            od = read_od_dev(dev, NcIn(path), iprofile, first.nlay, first.nwav);
            d_od = od.buf.ptr();
            od_type = od.type;
          }
          const Spectrum& s = gas.conc == CONC_NONE ? merged.first : first;
          ncol = (s.ncol + temperature_stride - 1) / temperature_stride;
          if (iconc == 0 && icol == 0) {
            const size_t total = nconc * (size_t)ncol * s.nlay * ng;
            gas.molar_abs.assign(total, 0.0);
            gas.min_molar_abs.assign(total, 0.0);
            gas.max_molar_abs.assign(total, 0.0);
            model.nt = ncol;
            model.temperature.assign((size_t)ncol * s.nlay, 0.0);
          }
          column(s, d_od, od_type, ref_vmr, gas, (iconc * ncol + icol) * (size_t)s.nlay * ng, t_fl);
          std::copy(t_fl.begin(), t_fl.end(), model.temperature.begin() + (size_t)icol * nlay);   // the last gas's, as the reference keeps it
          if (gas.conc == CONC_LUT && icol == 0) gas.vmr.push_back(ref_vmr);
        }
      }
      model.gases.push_back(std::move(gas));
    }
    if (!gmap) fail(ECCKD_PARAMETER_ERROR, "No gases specified in \"gases\"");

    // ---- fraction of the spectrum contributing to each g point (:506-548) ----
    LOG("Computing fraction of spectrum contributing to each g-point\n");
    const int dwav = do_sw ? 50 : 10;
    const int startwav = (int)(std::floor(*std::min_element(model.wavenumber1_band.begin(), model.wavenumber1_band.end()) / dwav) * dwav);
    const int endwav = (int)(std::ceil(*std::max_element(model.wavenumber2_band.begin(), model.wavenumber2_band.end()) / dwav) * dwav);
    LOG("  using wavenumber grid %d-%d cm-1 with %d cm-1 resolution\n", startwav, endwav, dwav);
    const int nint = (endwav - startwav) / dwav;
    model.wavenumber1.resize(nint);
    model.wavenumber2.resize(nint);
    for (int i = 0; i < nint; ++i) { model.wavenumber1[i] = startwav + (double)i * dwav; model.wavenumber2[i] = startwav + (double)(i + 1) * dwav; }
    model.gpoint_fraction.resize((size_t)ng * nint);
    ck(ecckd_gpoint_fraction(gmap, nint, model.wavenumber1.data(), model.wavenumber2.data(), model.gpoint_fraction.data()));

    if (model.is_sw) {
      if (!do_sw) fail(ECCKD_PARAMETER_ERROR, "The g-points file is shortwave but no \"ssi\" file was given");
      // solar irradiance in each interval (:556-561) and the Rayleigh coefficient of each g point
      // (CkdModel::calc_rayleigh_molar_scat, ckd_model.h:368-385; rayleigh_scattering.h:25-43)
      model.solar_spectral_irradiance.assign(nint, 0.0);
      for (size_t i = 0; i < nwav; ++i) {
        const double w = ssi_wavenumber[i];
        const int k = (int)std::ceil((w - startwav) / dwav) - 1;          // wavenumber1 < w <= wavenumber2
        if (k >= 0 && k < nint && w > model.wavenumber1[k] && w <= model.wavenumber2[k]) model.solar_spectral_irradiance[k] += ssi[i];
        else for (int j = 0; j < nint; ++j) if (w > model.wavenumber1[j] && w <= model.wavenumber2[j]) { model.solar_spectral_irradiance[j] += ssi[i]; break; }
      }
      model.reference_total_solar_irradiance = tsi;
      const double molar_column = 1.0e5 / (9.80665 * 0.001 * 28.970);
      std::vector<double> trans_hr(nint);
      for (int i = 0; i < nint; ++i) {
        const double um = 10000.0 / (0.5 * (model.wavenumber1[i] + model.wavenumber2[i]));
        const double xs = um < 0.5 ? 3.01577e-32 * std::pow(um, -(3.55212 + 1.35579 * um + 0.11563 / um))
                                   : 4.01061e-32 * std::pow(um, -(3.99668 + 0.00110298 * um + 0.0271393 / um));
        trans_hr[i] = std::exp(-molar_column * xs * 6.02214076e23 / 0.5);
      }
      model.rayleigh_molar_scattering.resize(ng);
      for (int g = 0; g < ng; ++g) {
        double num = 0.0, den = 0.0;
        for (int i = 0; i < nint; ++i) {
          const double gf = model.gpoint_fraction[(size_t)g * nint + i];
          num += gf * (model.solar_spectral_irradiance[i] * trans_hr[i]);
          den += gf * model.solar_spectral_irradiance[i];
        }
        model.rayleigh_molar_scattering[g] = -std::log(std::max(1.0e-14, num / den)) * 0.5 / molar_column;
      }
    } else {
      LOG("Generating Planck-function look-up table\n");
      for (int t = 120; t <= 350; ++t) model.temperature_planck.push_back(t);   // :581
      model.planck_function.resize(model.temperature_planck.size() * ng);
      ck(ecckd_planck_lut(gmap, (int)model.temperature_planck.size(), model.temperature_planck.data(), model.planck_function.data()));
    }
    ck(ecckd_gmap_destroy(gmap));

    LOG("Writing %s\n", output.c_str());
    write_ckd(output, model, history_line(argc, argv), config.str());
    return done(0);
  });
}
