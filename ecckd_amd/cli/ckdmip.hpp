// ckdmip.hpp - bin/ckdmip_lw and bin/ckdmip_sw: stand-ins for the EXTERNAL CKDMIP tools that the reference's scripts call (they are
// not part of ecCKD; test/config.h:32-35), restricted to what those scripts use and to the no-scattering radiative transfer
// the reference itself contains.
//
// ckdmip_{lw,sw} [--config file.nam] [--scenario name] [--merge-only] [--column-range a b] [--ssi file]
//                {[--scale s | --conc c | --const c] spectrum-file}... [--ckd optical-depth-file] --output file
//
//   --merge-only   test/merge_well_mixed_lw.sh:28-31, :46-49, :60-63 (merge_well_mixed_sw.sh:35-81): the optical depths of
//                  several gas files added up, each scaled as requested, written as one spectrum file that read_spectrum /
//                  read_merged_spectrum read (what src/ecckd/merge_spectra.cpp:20-156 does too);
//   (default)      test/run_lw_lbl_evaluation.sh:286-323, test/run_sw_lbl_evaluation.sh:70-260: the line-by-line training
//                  fluxes of every column - the gases' optical depths added up with their scalings, radiative transfer per
//                  wavenumber, fluxes summed over the bands of the namelist - in the variables LblFluxes::read expects
//                  (lbl_fluxes.cpp:60-133).  Longwave: Planck function + two-stream.  Shortwave: direct beam and the
//                  upwelling flux reflected by the surface (radiative_transfer_norayleigh_sw, radiative_transfer_sw.cpp:45-77)
//                  for every cos_solar_zenith_angle of the namelist; a Rayleigh spectrum file among the inputs is one more
//                  absorber, as the reference's own forward model treats Rayleigh scattering (solve_adept.cpp:34) - the real
//                  tool scatters, which is why the reference masks the upwelling fluxes it cannot represent
//                  (LblFluxes::mask_rayleigh_up);
//   --ckd file     test/run_ckd_lw.sh:133-137, test/run_ckd_sw.sh:125-128: radiative transfer on the g-point optical depths
//                  that run_ckd wrote, fluxes per column.
// A file may be preceded by  --scale s  (optical depth times s),  --conc c  (scaled so that the file's reference surface
// mole fraction becomes c)  or  --const c  (a mole fraction c at every level: each level scaled by c / its own).
// Namelist (&longwave_config / &shortwave_config): band_wavenumber1 / band_wavenumber2, nspectralstride (1 only), nangle (0:
// classic two-stream, diffusivity 1.66; N = 1..16: N Gauss-Legendre zenith angles per hemisphere, as test/run_ckd_lw.sh:28
// and test/copy_to_ckdmip_lw.sh:32 ask for with NANGLE=4), surf_albedo, cos_solar_zenith_angle, do_write_spectral_boundary_fluxes (the
// spectral fluxes at the surface and the top of the atmosphere per wavenumber, with the wavenumber grid); the
// *_name keys must keep their CKDMIP defaults.  Spectra are streamed from the files into HBM (ecckd_nc_read_dev) and merged
// there; the per-wavenumber radiative transfer is ecckd_lbl_band_fluxes_lw / _sw.  Output is a classic NetCDF file whatever
// its name.
#pragma once
#include <algorithm>
#include <cctype>
#include <fstream>
#include <sstream>

#include "tool.hpp"

using namespace tool;

namespace {

struct GasArg {
  std::string path;
  enum { NONE, SCALE, CONC, CONST } mode = NONE;
  double value = 1.0;
};

struct Namelist {
  std::vector<double> band1, band2;
  int nspectralstride = 1, nangle = 0;
  bool boundary_fluxes = false;
  double surf_albedo = 0.15;                       // test/run_sw_lbl_evaluation.sh sets it
  std::vector<double> mu0;                         // cos_solar_zenith_angle
};

std::string lower(std::string s) { std::transform(s.begin(), s.end(), s.begin(), ::tolower); return s; }

// Fortran namelist, as far as the scripts' files go: `key = v, v, ...` items, `!` comments, `key(a:b)` index ranges ignored
Namelist read_namelist(const std::string& path) {
  Namelist nl;
  std::ifstream in(path);
  if (!in) fail(ECCKD_PARAMETER_ERROR, "Cannot open namelist %s", path.c_str());
  std::string text, line;
  while (std::getline(in, line)) {
    const size_t c = line.find('!');
    if (c != std::string::npos) line.erase(c);
    text += line + "\n";
  }
  // split into key = value chunks
  std::vector<std::pair<std::string, std::string>> items;
  size_t pos = 0;
  std::string key;
  while (true) {
    const size_t eq = text.find('=', pos);
    if (eq == std::string::npos) break;
    // key = the token before '=' (back to the previous separator)
    size_t k0 = text.find_last_of(",\n&", eq);
    k0 = k0 == std::string::npos ? 0 : k0 + 1;
    std::string k = text.substr(k0, eq - k0);
    k.erase(std::remove_if(k.begin(), k.end(), ::isspace), k.end());
    const size_t paren = k.find('(');
    if (paren != std::string::npos) k.erase(paren);
    // value = up to the next "key =" or the closing '/'
    size_t next_eq = text.find('=', eq + 1);
    size_t end = text.size();
    if (next_eq != std::string::npos) {
      size_t nk = text.find_last_of(",\n", next_eq);
      if (nk != std::string::npos && nk > eq) end = nk;
    }
    std::string v = text.substr(eq + 1, end - eq - 1);
    const size_t slash = v.find('/');
    if (slash != std::string::npos && v.find('"') == std::string::npos) v.erase(slash);
    items.push_back({lower(k), v});
    pos = eq + 1;
  }
  auto numbers = [](const std::string& v) {
    std::vector<double> out;
    std::string t = v;
    std::replace(t.begin(), t.end(), ',', ' ');
    std::istringstream ss(t);
    double x;
    while (ss >> x) out.push_back(x);
    return out;
  };
  auto truth = [](const std::string& v) { const std::string t = lower(v); return t.find("true") != std::string::npos || t.find(".t") != std::string::npos; };
  for (auto& kv : items) {
    if (kv.first == "band_wavenumber1") nl.band1 = numbers(kv.second);
    else if (kv.first == "band_wavenumber2") nl.band2 = numbers(kv.second);
    else if (kv.first == "nspectralstride") nl.nspectralstride = (int)numbers(kv.second).at(0);
    else if (kv.first == "nangle") nl.nangle = (int)numbers(kv.second).at(0);
    else if (kv.first == "do_write_spectral_boundary_fluxes") nl.boundary_fluxes = truth(kv.second);
    else if (kv.first == "surf_albedo") nl.surf_albedo = numbers(kv.second).at(0);
    else if (kv.first == "cos_solar_zenith_angle") nl.mu0 = numbers(kv.second);
  }
  return nl;
}

}  // namespace

inline int ckdmip_main(int argc, char** argv, bool sw) {
  try {
    std::vector<GasArg> gases;
    std::string config_file, scenario, output, ckd_file, ssi_file;
    bool merge_only = false;
    long col_a = -1, col_b = -1;
    GasArg pending;
    for (int i = 1; i < argc; ++i) {
      const std::string a = argv[i];
      auto need = [&](int n) { if (i + n >= argc) fail(ECCKD_PARAMETER_ERROR, "%s needs %d argument(s)", a.c_str(), n); };
      if (a == "--config") { need(1); config_file = argv[++i]; }
      else if (a == "--scenario") { need(1); scenario = argv[++i]; }
      else if (a == "--output") { need(1); output = argv[++i]; }
      else if (a == "--ckd") { need(1); ckd_file = argv[++i]; }
      else if (a == "--ssi") { need(1); ssi_file = argv[++i]; }
      else if (a == "--merge-only") merge_only = true;
      else if (a == "--column-range") { need(2); col_a = std::atol(argv[++i]); col_b = std::atol(argv[++i]); }
      else if (a == "--scale") { need(1); pending.mode = GasArg::SCALE; pending.value = std::atof(argv[++i]); }
      else if (a == "--conc") { need(1); pending.mode = GasArg::CONC; pending.value = std::atof(argv[++i]); }
      else if (a == "--const") { need(1); pending.mode = GasArg::CONST; pending.value = std::atof(argv[++i]); }
      else if (a.rfind("--", 0) == 0) fail(ECCKD_PARAMETER_ERROR, "Argument \"%s\" not understood", a.c_str());
      else { pending.path = a; gases.push_back(pending); pending = GasArg(); }
    }
    if (output.empty()) fail(ECCKD_PARAMETER_ERROR, "\"--output\" file not specified");
    Namelist nl;
    if (!config_file.empty()) nl = read_namelist(config_file);
    if (nl.nspectralstride != 1) fail(ECCKD_PARAMETER_ERROR, "nspectralstride = %d is not supported (1 only)", nl.nspectralstride);
    if (nl.nangle < 0 || nl.nangle > 16) fail(ECCKD_PARAMETER_ERROR, "nangle = %d outside 0..16", nl.nangle);
    if (sw && nl.nangle != 0) WARN("nangle = %d has no meaning for the direct solar beam: ignored", nl.nangle);
    const std::string history = history_line(argc, argv);

    // ---------------------------------------------------------------------------------------------------------------
    if (!ckd_file.empty()) {   // radiative transfer on a CKD model's optical depths (test/run_ckd_lw.sh:133-137)
      Device dev;
      NcIn f(ckd_file);
      const std::vector<size_t> sh = f.shape("optical_depth");
      if (sh.size() != 3) fail(ECCKD_PARAMETER_ERROR, "optical_depth in %s is not (column, level, g_point)", ckd_file.c_str());
      const int ncol = (int)sh[0], nlay = (int)sh[1], ng = (int)sh[2];
      if (sw) {
        if (!f.exist("incoming_sw")) fail(ECCKD_PARAMETER_ERROR, "%s holds no incoming_sw: not a shortwave optical-depth file", ckd_file.c_str());
        std::vector<double> mu0 = nl.mu0;
        if (mu0.empty()) mu0 = {0.5};
        const int nmu = (int)mu0.size();
        NcOut out(output);
        out.dim("column", ncol); out.dim("mu0", nmu); out.dim("half_level", nlay + 1); out.dim("g_point", ng);
        out.var("pressure_hl", NC_FLOAT_T, {"column", "half_level"}, "Pressure at half levels", "Pa");
        out.var("mu0", NC_FLOAT_T, {"mu0"}, "Cosine of solar zenith angle", "1");
        out.var("flux_up_sw", NC_FLOAT_T, {"column", "mu0", "half_level"}, "Upwelling shortwave flux", "W m-2");
        out.var("flux_dn_sw", NC_FLOAT_T, {"column", "mu0", "half_level"}, "Downwelling shortwave flux", "W m-2");
        out.var("flux_dn_direct_sw", NC_FLOAT_T, {"column", "mu0", "half_level"}, "Downwelling direct shortwave flux", "W m-2");
        out.var("spectral_flux_up_sw", NC_FLOAT_T, {"column", "mu0", "half_level", "g_point"}, "Upwelling shortwave flux per g point", "W m-2");
        out.var("spectral_flux_dn_direct_sw", NC_FLOAT_T, {"column", "mu0", "half_level", "g_point"}, "Downwelling direct shortwave flux per g point", "W m-2");
        if (!scenario.empty()) out.att(scenario, "scenario");
        out.att(history, "history");
        out.end_define();
        out.write("mu0", mu0);
        const size_t nhg = (size_t)(nlay + 1) * ng;
        const bool have_ray = f.exist("rayleigh_optical_depth");
        for (int c = 0; c < ncol; ++c) {
          std::vector<double> od = f.read("optical_depth", c);
          if (have_ray) {                       // one more absorber, as in the reference's forward model (solve_adept.cpp:34)
            const std::vector<double> ray = f.read("rayleigh_optical_depth", c);
            for (size_t i = 0; i < od.size(); ++i) od[i] += ray[i];
          }
          const std::vector<double> incoming = f.read("incoming_sw", c);
          std::vector<double> dn_all(nmu * nhg), up_all(nmu * nhg), bdn((size_t)nmu * (nlay + 1), 0.0), bup((size_t)nmu * (nlay + 1), 0.0);
          for (int m = 0; m < nmu; ++m) {
            // radiative_transfer_sw.cpp:45-77 per g point, on the device (ecckd_rt_sw_gpoints)
            ck(ecckd_rt_sw_gpoints(dev.ctx(), 1, nlay, ng, mu0[m], nl.surf_albedo, incoming.data(), od.data(), &dn_all[m * nhg], &up_all[m * nhg]));
            for (int i = 0; i <= nlay; ++i)
              for (int g = 0; g < ng; ++g) {
                bdn[(size_t)m * (nlay + 1) + i] += dn_all[m * nhg + (size_t)i * ng + g];
                bup[(size_t)m * (nlay + 1) + i] += up_all[m * nhg + (size_t)i * ng + g];
              }
          }
          out.write_slice("pressure_hl", c, f.read("pressure_hl", c));
          out.write_slice("flux_dn_direct_sw", c, bdn); out.write_slice("flux_dn_sw", c, bdn); out.write_slice("flux_up_sw", c, bup);
          out.write_slice("spectral_flux_dn_direct_sw", c, dn_all); out.write_slice("spectral_flux_up_sw", c, up_all);
        }
        out.close();
        return done(0);
      }
      if (!f.exist("planck_hl")) fail(ECCKD_PARAMETER_ERROR, "%s holds no planck_hl: not a longwave optical-depth file", ckd_file.c_str());
      NcOut out(output);
      out.dim("column", ncol); out.dim("half_level", nlay + 1); out.dim("g_point", ng);
      out.var("pressure_hl", NC_FLOAT_T, {"column", "half_level"}, "Pressure at half levels", "Pa");
      out.var("flux_up_lw", NC_FLOAT_T, {"column", "half_level"}, "Upwelling longwave flux", "W m-2");
      out.var("flux_dn_lw", NC_FLOAT_T, {"column", "half_level"}, "Downwelling longwave flux", "W m-2");
      out.var("spectral_flux_up_lw", NC_FLOAT_T, {"column", "half_level", "g_point"}, "Upwelling longwave flux per g point", "W m-2");
      out.var("spectral_flux_dn_lw", NC_FLOAT_T, {"column", "half_level", "g_point"}, "Downwelling longwave flux per g point", "W m-2");
      if (!scenario.empty()) out.att(scenario, "scenario");
      out.att(history, "history");
      out.end_define();
      const size_t nhg = (size_t)(nlay + 1) * ng;
      for (int c = 0; c < ncol; ++c) {
        const std::vector<double> od = f.read("optical_depth", c), planck = f.read("planck_hl", c);
        std::vector<double> dn(nhg), up(nhg), bdn(nlay + 1, 0.0), bup(nlay + 1, 0.0);
        // radiative_transfer_lw.cpp:27-60 per g point (two-stream or nangle Gauss-Legendre angles), on the device
        ck(ecckd_rt_lw_gpoints(dev.ctx(), nl.nangle, 1, nlay, ng, planck.data(), od.data(), dn.data(), up.data()));
        for (int i = 0; i <= nlay; ++i)
          for (int g = 0; g < ng; ++g) { bdn[i] += dn[(size_t)i * ng + g]; bup[i] += up[(size_t)i * ng + g]; }
        out.write_slice("pressure_hl", c, f.read("pressure_hl", c));
        out.write_slice("flux_dn_lw", c, bdn); out.write_slice("flux_up_lw", c, bup);
        out.write_slice("spectral_flux_dn_lw", c, dn); out.write_slice("spectral_flux_up_lw", c, up);
      }
      out.close();
      return done(0);
    }

    if (gases.empty()) fail(ECCKD_PARAMETER_ERROR, "No spectrum files given");
    // ---- grids and profiles from the first file; every other file must share them ----
    Spectrum first = read_spectrum(gases[0].path, 0, false);
    const int nlay = first.nlay, ncol_file = first.ncol;
    const size_t nwav = first.nwav;
    int c0 = 0, c1 = ncol_file - 1;
    if (col_a >= 1) { c0 = (int)col_a - 1; c1 = std::min<long>(col_b, ncol_file) - 1; }   // 1-based, inclusive, like the Fortran tool
    const int ncol = c1 - c0 + 1;
    if (ncol <= 0) fail(ECCKD_PARAMETER_ERROR, "Empty column range");
    LOG("%d gas file(s), %d column(s), %d layers, %zu spectral points\n", (int)gases.size(), ncol, nlay, nwav);

    Device dev;
    DevBuf d_merged(dev, (size_t)nlay * nwav * sizeof(double));
    DevBuf d_wn, d_dwn;
    d_wn.upload(dev, first.wavenumber_cm_1);
    d_dwn.upload(dev, first.d_wavenumber_cm_1);
    DevBuf d_ssi, d_albedo;
    std::vector<double> mu0 = nl.mu0;
    if (sw && !merge_only) {
      if (ssi_file.empty()) fail(ECCKD_PARAMETER_ERROR, "\"--ssi\" file not specified");
      NcIn fs(ssi_file);
      const std::vector<double> ssi = fs.read("solar_spectral_irradiance");
      if (ssi.size() != nwav) fail(ECCKD_PARAMETER_ERROR, "solar_spectral_irradiance has %zu points, the spectra %zu", ssi.size(), nwav);
      d_ssi.upload(dev, ssi);
      d_albedo.upload(dev, std::vector<double>(nwav, nl.surf_albedo));
      if (mu0.empty()) fail(ECCKD_PARAMETER_ERROR, "cos_solar_zenith_angle missing from the namelist");
    }
    const int nmu = (int)mu0.size();
    std::string ids;
    std::vector<std::string> molecules;
    for (const GasArg& g : gases) {
      NcIn f(g.path);
      std::string m;
      if (!f.att_text("constituent_id", m)) f.att_text("molecules", m);
      molecules.push_back(m.substr(0, m.find(' ')));
      ids += (ids.empty() ? "" : " ") + molecules.back();
    }

    // bands of the namelist -> wavenumber index ranges
    std::vector<int64_t> bbegin, bend;
    int nband = 0;
    if (!merge_only) {
      if (nl.band1.empty() || nl.band1.size() != nl.band2.size())
        fail(ECCKD_PARAMETER_ERROR, "band_wavenumber1 / band_wavenumber2 missing from the namelist");
      nband = (int)nl.band1.size();
      bbegin.resize(nband); bend.resize(nband);
      ck(ecckd_band_ranges(nwav, first.wavenumber_cm_1.data(), nband, nl.band1.data(), nl.band2.data(), nullptr, bbegin.data(), bend.data()));
    }

    NcOut out(output);
    out.dim("column", ncol); out.dim("half_level", nlay + 1); out.dim("level", nlay);
    out.var("pressure_hl", NC_FLOAT_T, {"column", "half_level"}, "Pressure at half levels", "Pa");
    out.var("temperature_hl", NC_FLOAT_T, {"column", "half_level"}, "Temperature at half levels", "K");
    if (merge_only) {
      out.dim("wavenumber", nwav);
      out.var("wavenumber", NC_DOUBLE_T, {"wavenumber"}, "Wavenumber", "cm-1");
      out.var("optical_depth", NC_FLOAT_T, {"column", "level", "wavenumber"}, "Layer optical depth");
      out.att(ids, "molecules");
      out.att("composite", "constituent_id");
    } else if (sw) {
      out.dim("gas", gases.size()); out.dim("mu0", nmu); out.dim("band_sw", nband);
      out.var("mole_fraction_fl", NC_FLOAT_T, {"column", "gas", "level"}, "Mole fraction at full levels", "1");
      out.var("mu0", NC_FLOAT_T, {"mu0"}, "Cosine of solar zenith angle", "1");
      out.var("flux_up_sw", NC_FLOAT_T, {"column", "mu0", "half_level"}, "Upwelling shortwave flux", "W m-2");
      out.var("flux_dn_sw", NC_FLOAT_T, {"column", "mu0", "half_level"}, "Downwelling shortwave flux", "W m-2");
      out.var("flux_dn_direct_sw", NC_FLOAT_T, {"column", "mu0", "half_level"}, "Downwelling direct shortwave flux", "W m-2");
      out.var("band_wavenumber1_sw", NC_FLOAT_T, {"band_sw"}, "Lower bound wavenumber for shortwave band", "cm-1");
      out.var("band_wavenumber2_sw", NC_FLOAT_T, {"band_sw"}, "Upper bound wavenumber for shortwave band", "cm-1");
      out.var("band_flux_up_sw", NC_FLOAT_T, {"column", "mu0", "half_level", "band_sw"}, "Upwelling shortwave flux in bands", "W m-2");
      out.var("band_flux_dn_direct_sw", NC_FLOAT_T, {"column", "mu0", "half_level", "band_sw"}, "Downwelling direct shortwave flux in bands", "W m-2");
      out.att(ids, "constituent_id");
      if (nl.boundary_fluxes) {     // what LblFluxes::read maps to g points (lbl_fluxes.cpp:183-246)
        out.dim("wavenumber", nwav);
        out.var("wavenumber", NC_DOUBLE_T, {"wavenumber"}, "Wavenumber", "cm-1");
        out.var("spectral_flux_dn_direct_surf_sw", NC_FLOAT_T, {"column", "mu0", "wavenumber"}, "Spectral direct shortwave flux at the surface", "W m-2");
        out.var("spectral_flux_up_toa_sw", NC_FLOAT_T, {"column", "mu0", "wavenumber"}, "Spectral upwelling shortwave flux at top of atmosphere", "W m-2");
      }
    } else {
      out.dim("gas", gases.size()); out.dim("band_lw", nband);
      out.var("mole_fraction_fl", NC_FLOAT_T, {"column", "gas", "level"}, "Mole fraction at full levels", "1");
      out.var("flux_up_lw", NC_FLOAT_T, {"column", "half_level"}, "Upwelling longwave flux", "W m-2");
      out.var("flux_dn_lw", NC_FLOAT_T, {"column", "half_level"}, "Downwelling longwave flux", "W m-2");
      out.var("band_wavenumber1_lw", NC_FLOAT_T, {"band_lw"}, "Lower bound wavenumber for longwave band", "cm-1");
      out.var("band_wavenumber2_lw", NC_FLOAT_T, {"band_lw"}, "Upper bound wavenumber for longwave band", "cm-1");
      out.var("band_flux_up_lw", NC_FLOAT_T, {"column", "half_level", "band_lw"}, "Upwelling longwave flux in bands", "W m-2");
      out.var("band_flux_dn_lw", NC_FLOAT_T, {"column", "half_level", "band_lw"}, "Downwelling longwave flux in bands", "W m-2");
      out.att(ids, "constituent_id");
      if (nl.boundary_fluxes) {     // lbl_fluxes.cpp:301-325
        out.dim("wavenumber", nwav);
        out.var("wavenumber", NC_DOUBLE_T, {"wavenumber"}, "Wavenumber", "cm-1");
        out.var("spectral_flux_dn_surf_lw", NC_FLOAT_T, {"column", "wavenumber"}, "Spectral downwelling longwave flux at the surface", "W m-2");
        out.var("spectral_flux_up_toa_lw", NC_FLOAT_T, {"column", "wavenumber"}, "Spectral upwelling longwave flux at top of atmosphere", "W m-2");
      }
    }
    if (!scenario.empty()) out.att(scenario, "scenario");
    out.att(history, "history");
    out.end_define();
    if (merge_only) out.write("wavenumber", first.wavenumber_cm_1);
    else if (sw) { out.write("band_wavenumber1_sw", nl.band1); out.write("band_wavenumber2_sw", nl.band2); out.write("mu0", mu0); }
    else { out.write("band_wavenumber1_lw", nl.band1); out.write("band_wavenumber2_lw", nl.band2); }
    DevBuf d_bnd_dn, d_bnd_up;
    if (!merge_only && nl.boundary_fluxes) {
      out.write("wavenumber", first.wavenumber_cm_1);
      d_bnd_dn.alloc(dev, nwav * sizeof(double));
      d_bnd_up.alloc(dev, nwav * sizeof(double));
    }
    double* const p_bnd_dn = nl.boundary_fluxes && !merge_only ? d_bnd_dn.as<double>() : nullptr;
    double* const p_bnd_up = nl.boundary_fluxes && !merge_only ? d_bnd_up.as<double>() : nullptr;

    for (int c = c0; c <= c1; ++c) {
      const Spectrum col = c == 0 ? first : read_spectrum(gases[0].path, c, false);
      std::vector<double> vmr_all;   // [gas][level]
      for (size_t ig = 0; ig < gases.size(); ++ig) {
        const GasArg& g = gases[ig];
        NcIn f(g.path);
        double ref = -1.0;
        std::vector<double> vmr;
        std::string mol;
        read_od_meta(f, c, nlay, ref, vmr, mol);
        std::vector<double> profile(nlay, 1.0), vmr_out(nlay, -1.0);
        if (g.mode == GasArg::SCALE) {
          for (int l = 0; l < nlay; ++l) { profile[l] = g.value; vmr_out[l] = vmr[l] >= 0.0 ? vmr[l] * g.value : -1.0; }
        } else if (g.mode == GasArg::CONC) {
          ck(ecckd_merge_scaling(nlay, col.pressure_hl.data(), -1.0, g.value, ref, vmr.data(), 0, nullptr, nullptr, profile.data(), vmr_out.data()));
        } else if (g.mode == GasArg::CONST) {
          for (int l = 0; l < nlay; ++l) {
            if (!(vmr[l] > 0.0)) fail(ECCKD_PARAMETER_ERROR, "--const needs mole_fraction_fl in %s", g.path.c_str());
            profile[l] = g.value / vmr[l];
            vmr_out[l] = g.value;
          }
        } else {
          vmr_out = vmr;
        }
        vmr_all.insert(vmr_all.end(), vmr_out.begin(), vmr_out.end());
        DevOd od = read_od_dev(dev, f, c, nlay, nwav);
        ck(ecckd_merge_spectrum_dev(dev.ctx(), nlay, nwav, od.buf.ptr(), od.type, nwav, profile.data(), ig == 0 ? 1 : 0,
                                    d_merged.as<double>(), nwav));
        ck(ecckd_synchronize(dev.ctx()));
      }
      const size_t oc = (size_t)(c - c0);
      out.write_slice("pressure_hl", oc, col.pressure_hl);
      if (!col.temperature_hl.empty()) out.write_slice("temperature_hl", oc, col.temperature_hl);
      if (merge_only) {
        out.write_slice("optical_depth", oc, d_merged.download<double>());
      } else if (sw) {
        const size_t nhl = (size_t)nlay + 1;
        std::vector<double> tdn((size_t)nmu * nhl * nband), tup((size_t)nmu * nhl * nband), sdn((size_t)nmu * nhl, 0.0), sup((size_t)nmu * nhl, 0.0);
        std::vector<double> bdn((size_t)nband * nhl), bup((size_t)nband * nhl);
        std::vector<double> all_dn, all_up;     // (mu0, wavenumber) boundary fluxes of this column
        for (int m = 0; m < nmu; ++m) {
          ck(ecckd_lbl_band_fluxes_sw_ex(dev.ctx(), nlay, nwav, mu0[m], d_ssi.as<double>(), d_albedo.as<double>(), d_merged.ptr(), ECCKD_F64, nwav,
                                         nband, bbegin.data(), bend.data(), bdn.data(), bup.data(), p_bnd_dn, p_bnd_up));
          if (p_bnd_dn) {
            const std::vector<double> a = d_bnd_dn.download<double>(), b = d_bnd_up.download<double>();
            all_dn.insert(all_dn.end(), a.begin(), a.end());
            all_up.insert(all_up.end(), b.begin(), b.end());
          }
          for (int b = 0; b < nband; ++b)
            for (size_t i = 0; i < nhl; ++i) {
              tdn[((size_t)m * nhl + i) * nband + b] = bdn[(size_t)b * nhl + i];
              tup[((size_t)m * nhl + i) * nband + b] = bup[(size_t)b * nhl + i];
              sdn[(size_t)m * nhl + i] += bdn[(size_t)b * nhl + i];
              sup[(size_t)m * nhl + i] += bup[(size_t)b * nhl + i];
            }
        }
        out.write_slice("mole_fraction_fl", oc, vmr_all);
        out.write_slice("band_flux_dn_direct_sw", oc, tdn); out.write_slice("band_flux_up_sw", oc, tup);
        out.write_slice("flux_dn_direct_sw", oc, sdn); out.write_slice("flux_dn_sw", oc, sdn); out.write_slice("flux_up_sw", oc, sup);
        if (p_bnd_dn) { out.write_slice("spectral_flux_dn_direct_surf_sw", oc, all_dn); out.write_slice("spectral_flux_up_toa_sw", oc, all_up); }
      } else {
        if (col.temperature_hl.empty()) fail(ECCKD_PARAMETER_ERROR, "temperature_hl missing from %s", gases[0].path.c_str());
        std::vector<double> bdn((size_t)nband * (nlay + 1)), bup((size_t)nband * (nlay + 1));
        ck(ecckd_lbl_band_fluxes_lw_angles(dev.ctx(), nl.nangle, nlay, nwav, col.temperature_hl.data(), d_wn.as<double>(), d_dwn.as<double>(),
                                       d_merged.ptr(), ECCKD_F64, nwav, nband, bbegin.data(), bend.data(), bdn.data(), bup.data(),
                                       p_bnd_dn, p_bnd_up));
        // [band][level] -> (half_level, band) and the broadband sums
        std::vector<double> tdn((size_t)(nlay + 1) * nband), tup((size_t)(nlay + 1) * nband), sdn(nlay + 1, 0.0), sup(nlay + 1, 0.0);
        for (int b = 0; b < nband; ++b)
          for (int i = 0; i <= nlay; ++i) {
            tdn[(size_t)i * nband + b] = bdn[(size_t)b * (nlay + 1) + i];
            tup[(size_t)i * nband + b] = bup[(size_t)b * (nlay + 1) + i];
            sdn[i] += bdn[(size_t)b * (nlay + 1) + i];
            sup[i] += bup[(size_t)b * (nlay + 1) + i];
          }
        out.write_slice("mole_fraction_fl", oc, vmr_all);
        out.write_slice("band_flux_dn_lw", oc, tdn); out.write_slice("band_flux_up_lw", oc, tup);
        out.write_slice("flux_dn_lw", oc, sdn); out.write_slice("flux_up_lw", oc, sup);
        if (p_bnd_dn) {
          out.write_slice("spectral_flux_dn_surf_lw", oc, d_bnd_dn.download<double>());
          out.write_slice("spectral_flux_up_toa_lw", oc, d_bnd_up.download<double>());
        }
      }
      LOG("  column %d done\n", c + 1);
    }
    out.close();
    return done(0);
  } catch (const Fatal& f) {
    std::fprintf(stderr, "*** Error: %s\n", f.msg.c_str());
    return f.code ? f.code : 1;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "*** Error: %s\n", e.what());
    return ECCKD_UNEXPECTED_EXCEPTION;
  }
}
