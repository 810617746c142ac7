// The CKD-definition file (CkdModel::read ckd_model.cpp:32-286, CkdModel::write :290-641), classic NetCDF:
// what create_look_up_table writes and optimize_lut / scale_lut / run_ckd read and write.
#pragma once
#include <algorithm>

#include "tool.hpp"

namespace tool {

enum ConcDependence { CONC_NONE = 0, CONC_LINEAR = 1, CONC_LUT = 2, CONC_RELATIVE_LINEAR = 3 };

struct GasTable {   // SingleGasData<false> (single_gas_data.h)
  std::string name;
  int conc = CONC_NONE;
  std::vector<double> vmr;            // LUT: mole fractions of the table
  double reference_vmr = 0.0;         // relative-linear
  std::vector<double> molar_abs, min_molar_abs, max_molar_abs;   // [nconc][nt][np][ng] (nconc only for LUT)
  // conc_dependence none: the gases merged into this one and their mole fractions [ngas][np] (ckd_model.cpp:431-438)
  std::string composite_molecules;
  std::vector<double> composite_vmr;
  size_t nconc() const { return conc == CONC_LUT ? vmr.size() : 1; }
};

struct CkdFile {
  int ng = 0, nt = 0, np = 0;
  bool is_sw = false;
  std::string model_id;
  std::vector<double> pressure, temperature;                         // [np], [nt][np]
  std::vector<double> temperature_planck, planck_function;           // LW: [ntp], [ntp][ng]
  std::vector<double> solar_irradiance, rayleigh_molar_scattering;   // SW: [ng]
  std::vector<double> solar_spectral_irradiance;                     // SW: per interval of wavenumber1/2
  double reference_total_solar_irradiance = -1.0;
  std::vector<double> wavenumber1, wavenumber2, gpoint_fraction;     // [nwav], [nwav], [ng][nwav]
  std::vector<double> wavenumber1_band, wavenumber2_band;
  std::vector<int> band_number;                                      // [ng]
  std::vector<GasTable> gases;
  std::string history, config;                                       // of the file this model came from
  // the g point of every high-resolution wavenumber, stored by create_look_up_table when it changed the numbering of the
  // g-points file (CkdModel::save_g_points, ckd_model.h:314-318); read back by optimize_lut / scale_lut in preference to
  // "gpointfile" (read_g_points, :324-333).  `save_g_points` is set only by the tool that creates them (:471).
  std::vector<double> wavenumber_hr;
  std::vector<int> g_point_hr;
  bool save_g_points = false;
};

static const char* const K_NAME = "molar_absorption_coeff";   // constants.h:21

inline std::string upper(std::string s) {
  for (char& c : s) c = (char)std::toupper((unsigned char)c);
  return s;
}

inline void write_ckd(const std::string& path, const CkdFile& m, const std::string& history_line_, const std::string& config_str,
                      const std::string& summary = "") {
  NcOut f(path);
  const size_t nwav = m.wavenumber1.size();
  f.dim("temperature", m.nt);
  f.dim("pressure", m.np);
  f.dim("g_point", m.ng);
  if (!m.is_sw) f.dim("temperature_planck", m.temperature_planck.size());
  f.dim("wavenumber", nwav);
  f.dim("band", m.wavenumber1_band.size());
  if (m.save_g_points) f.dim("wavenumber_hr", m.wavenumber_hr.size());
  f.var("n_gases", NC_INT_T, {}, "Number of gases treated");
  f.att("The gases are listed in the global attribute \"constituent_id\".", "comment", "n_gases");
  f.var("temperature", NC_FLOAT_T, {"temperature", "pressure"}, "Temperature", "K");
  f.var("pressure", NC_FLOAT_T, {"pressure"}, "Pressure", "Pa");
  if (m.is_sw) {
    if (m.reference_total_solar_irradiance > 0.0)
      f.var("reference_total_solar_irradiance", NC_FLOAT_T, {}, "Reference total solar irradiance", "W m-2");
    f.var("solar_irradiance", NC_FLOAT_T, {"g_point"}, "Solar irradiance across each g point", "W m-2");
    if (!m.solar_spectral_irradiance.empty())
      f.var("solar_spectral_irradiance", NC_FLOAT_T, {"wavenumber"}, "Solar irradiance in each spectral interval", "W m-2");
  } else {
    f.var("temperature_planck", NC_FLOAT_T, {"temperature_planck"}, "Temperature for Planck function look-up table", "K");
    f.var("planck_function", NC_FLOAT_T, {"temperature_planck", "g_point"}, "Planck function look-up table", "W m-2");
  }
  f.var("wavenumber1", NC_FLOAT_T, {"wavenumber"}, "Lower wavenumber bound of spectral interval", "cm-1");
  f.var("wavenumber2", NC_FLOAT_T, {"wavenumber"}, "Upper wavenumber bound of spectral interval", "cm-1");
  f.var("gpoint_fraction", NC_FLOAT_T, {"g_point", "wavenumber"}, "Fraction of spectrum contributing to each g-point");
  f.var("wavenumber1_band", NC_FLOAT_T, {"band"}, "Lower wavenumber bound of band", "cm-1");
  f.var("wavenumber2_band", NC_FLOAT_T, {"band"}, "Upper wavenumber bound of band", "cm-1");
  f.var("band_number", NC_SHORT_T, {"g_point"}, "Band number of each g point");
  if (m.save_g_points) {
    f.var("wavenumber_hr", NC_DOUBLE_T, {"wavenumber_hr"}, "High-resolution wavenumber", "cm-1");
    f.var("g_point", NC_SHORT_T, {"wavenumber_hr"}, "G point");
  }
  if (m.is_sw && !m.rayleigh_molar_scattering.empty())
    f.var("rayleigh_molar_scattering_coeff", NC_FLOAT_T, {"g_point"}, "Rayleigh molar scattering coefficient in each g-point",
          "m2 mol-1");
  if (!m.model_id.empty()) f.att(m.model_id, "model_id");
  std::string names;
  for (const GasTable& g : m.gases) names += (names.empty() ? "" : " ") + g.name;
  f.att(names, "constituent_id");
  for (const GasTable& g : m.gases) {
    const std::string mol = g.name, Mol = upper(g.name), code = mol + "_conc_dependence_code";
    f.var(code, NC_SHORT_T, {}, (Mol + " concentration dependence code").c_str());
    f.att("0: No dependence of absorption on concentration (background gases)\n"
          "1: Absorption varies linearly with concentration\n"
          "2: Look-up table for concentration-dependence of absorption\n"
          "3: Linear dependence on concentration minus a reference value", "definition", code.c_str());
    std::vector<std::string> dims = {"temperature", "pressure", "g_point"};
    if (g.conc == CONC_LUT) {
      f.dim(mol + "_mole_fraction", g.vmr.size());
      f.var(mol + "_mole_fraction", NC_FLOAT_T, {mol + "_mole_fraction"}, (Mol + " mole fraction for look-up table").c_str(), "1");
      dims.insert(dims.begin(), mol + "_mole_fraction");
    }
    if (g.conc == CONC_RELATIVE_LINEAR)
      f.var(mol + "_reference_mole_fraction", NC_FLOAT_T, {}, ("Reference mole fraction of " + Mol).c_str(), "1");
    const std::string what = g.conc == CONC_NONE ? std::string("background gases") : Mol;
    const std::string k = mol + "_" + K_NAME;
    f.var(k, NC_FLOAT_T, dims, ("Molar absorption coefficient of " + what).c_str(), "m2 mol-1");
    if (!g.min_molar_abs.empty() && !g.max_molar_abs.empty()) {
      f.var(k + "_min", NC_FLOAT_T, dims, ("Minimum molar absorption coefficient of " + what).c_str(), "m2 mol-1");
      f.var(k + "_max", NC_FLOAT_T, dims, ("Maximum molar absorption coefficient of " + what).c_str(), "m2 mol-1");
    }
    if (g.conc == CONC_NONE && !g.composite_vmr.empty()) {
      f.dim(mol + "_gas", g.composite_vmr.size() / m.np);
      f.var(mol + "_mole_fraction", NC_FLOAT_T, {mol + "_gas", "pressure"}, ("Mole fractions of the gases that make up " + Mol).c_str(), "1");
      f.att("The gases that make up " + Mol + " are listed in the global attribute \"" + mol + "_constituent_id\".", "comment",
            (mol + "_mole_fraction").c_str());
      f.att(g.composite_molecules, mol + "_constituent_id");
    }
  }
  // the history of the g-points file, then this command (CkdModel::write :603-616)
  std::string history = m.history;
  if (!history.empty() && history.back() != '\n') history += "\n";
  f.att(history + history_line_, "history");
  f.att(config_str, "config");
  if (!summary.empty()) f.att(summary, "summary");
  f.end_define();
  f.write("n_gases", {(double)m.gases.size()});
  f.write("temperature", m.temperature);
  f.write("pressure", m.pressure);
  if (m.is_sw) {
    if (m.reference_total_solar_irradiance > 0.0) f.write("reference_total_solar_irradiance", {m.reference_total_solar_irradiance});
    f.write("solar_irradiance", m.solar_irradiance);
    if (!m.solar_spectral_irradiance.empty()) f.write("solar_spectral_irradiance", m.solar_spectral_irradiance);
    if (!m.rayleigh_molar_scattering.empty()) f.write("rayleigh_molar_scattering_coeff", m.rayleigh_molar_scattering);
  } else {
    f.write("temperature_planck", m.temperature_planck);
    f.write("planck_function", m.planck_function);
  }
  f.write("wavenumber1", m.wavenumber1);
  f.write("wavenumber2", m.wavenumber2);
  f.write("gpoint_fraction", m.gpoint_fraction);
  f.write("wavenumber1_band", m.wavenumber1_band);
  f.write("wavenumber2_band", m.wavenumber2_band);
  f.write_as_double("band_number", m.band_number);
  if (m.save_g_points) {
    f.write("wavenumber_hr", m.wavenumber_hr);
    f.write_as_double("g_point", m.g_point_hr);
  }
  for (const GasTable& g : m.gases) {
    const std::string mol = g.name, k = mol + "_" + K_NAME;
    f.write(mol + "_conc_dependence_code", {(double)g.conc});
    if (g.conc == CONC_LUT) f.write(mol + "_mole_fraction", g.vmr);
    if (g.conc == CONC_RELATIVE_LINEAR) f.write(mol + "_reference_mole_fraction", {g.reference_vmr});
    if (g.conc == CONC_NONE && !g.composite_vmr.empty()) f.write(mol + "_mole_fraction", g.composite_vmr);
    f.write(k, g.molar_abs);
    if (!g.min_molar_abs.empty() && !g.max_molar_abs.empty()) {
      f.write(k + "_min", g.min_molar_abs);
      f.write(k + "_max", g.max_molar_abs);
    }
  }
  f.close();
}

inline CkdFile read_ckd(const std::string& path) {
  NcIn f(path);
  CkdFile m;
  m.is_sw = f.exist("solar_irradiance");
  if (m.is_sw) {
    m.solar_irradiance = f.read("solar_irradiance");
    if (f.exist("solar_spectral_irradiance")) m.solar_spectral_irradiance = f.read("solar_spectral_irradiance");
    if (f.exist("reference_total_solar_irradiance")) m.reference_total_solar_irradiance = f.read_scalar("reference_total_solar_irradiance");
    if (f.exist("rayleigh_molar_scattering_coeff")) m.rayleigh_molar_scattering = f.read("rayleigh_molar_scattering_coeff");
  } else {
    m.temperature_planck = f.read("temperature_planck");
    m.planck_function = f.read("planck_function");
  }
  std::vector<size_t> sh = f.shape("temperature");
  m.nt = (int)sh.at(0);
  m.np = (int)sh.at(1);
  m.temperature = f.read("temperature");
  m.pressure = f.read("pressure");
  m.wavenumber1 = f.read("wavenumber1");
  m.wavenumber2 = f.read("wavenumber2");
  m.gpoint_fraction = f.read("gpoint_fraction");
  m.ng = (int)f.shape("gpoint_fraction").at(0);
  m.wavenumber1_band = f.read("wavenumber1_band");
  m.wavenumber2_band = f.read("wavenumber2_band");
  for (double b : f.read("band_number")) m.band_number.push_back((int)b);
  if (f.exist("g_point")) {   // ckd_model.cpp:72-75
    m.wavenumber_hr = f.read("wavenumber_hr");
    for (double g : f.read("g_point")) m.g_point_hr.push_back((int)g);
  }
  f.att_text("model_id", m.model_id);
  f.att_text("history", m.history);
  f.att_text("config", m.config);
  std::string ids;
  if (!f.att_text("constituent_id", ids)) fail(ECCKD_PARAMETER_ERROR, "%s: no constituent_id attribute", path.c_str());
  size_t pos = 0;
  while (pos < ids.size()) {
    size_t e = ids.find(' ', pos);
    if (e == std::string::npos) e = ids.size();
    if (e > pos) {
      GasTable g;
      g.name = ids.substr(pos, e - pos);
      g.conc = (int)f.read_scalar(g.name + "_conc_dependence_code");
      const std::string k = g.name + "_" + K_NAME;
      g.molar_abs = f.read(k);
      if (g.conc == CONC_LUT) g.vmr = f.read(g.name + "_mole_fraction");
      if (g.conc == CONC_RELATIVE_LINEAR) g.reference_vmr = f.read_scalar(g.name + "_reference_mole_fraction");
      if (g.conc == CONC_NONE && f.exist(g.name + "_mole_fraction")) {   // :193-195
        g.composite_vmr = f.read(g.name + "_mole_fraction");
        f.att_text(g.name + "_constituent_id", g.composite_molecules);
      }
      if (f.exist(k + "_min")) { g.min_molar_abs = f.read(k + "_min"); g.max_molar_abs = f.read(k + "_max"); }
      m.gases.push_back(std::move(g));
    }
    pos = e + 1;
  }
  if ((int)f.read_scalar("n_gases") != (int)m.gases.size())
    fail(ECCKD_PARAMETER_ERROR, "%s: n_gases does not match constituent_id", path.c_str());
  return m;
}

// ---- the library's view of a CKD definition (ecckd_opt_model): pointers into a CkdFile that must outlive it ----
struct ModelView {   // ecckd_opt_model over a CkdFile (pointers into it)
  ecckd_opt_model m;
  std::vector<ecckd_opt_gas> gases;
  std::vector<double> log_pressure;
  std::vector<int> iband;
};

inline void make_model(const CkdFile& f, const std::vector<std::string>& active, const std::vector<int>& iband, ModelView& v) {
  std::memset(&v.m, 0, sizeof v.m);
  v.log_pressure.resize(f.np);
  for (int i = 0; i < f.np; ++i) v.log_pressure[i] = std::log(f.pressure[i]);
  v.iband = iband;
  v.gases.resize(f.gases.size());
  for (size_t i = 0; i < f.gases.size(); ++i) {
    const GasTable& g = f.gases[i];
    ecckd_opt_gas& o = v.gases[i];
    std::memset(&o, 0, sizeof o);
    o.conc_dependence = g.conc;
    o.is_active = active.empty() || std::find(active.begin(), active.end(), g.name) != active.end();
    o.nconc = g.conc == CONC_LUT ? (int)g.vmr.size() : 0;
    o.vmr = g.conc == CONC_LUT ? g.vmr.data() : nullptr;
    o.reference_vmr = g.reference_vmr;
    o.molar_abs = g.molar_abs.data();
    o.min_molar_abs = g.min_molar_abs.empty() ? nullptr : g.min_molar_abs.data();
    o.max_molar_abs = g.max_molar_abs.empty() ? nullptr : g.max_molar_abs.data();
  }
  v.m.ng = f.ng; v.m.nt = f.nt; v.m.np = f.np;
  v.m.log_pressure = v.log_pressure.data();
  v.m.temperature = f.temperature.data();
  v.m.ntp = (int)f.temperature_planck.size();
  v.m.temperature_planck = f.is_sw ? nullptr : f.temperature_planck.data();
  v.m.planck_function = f.is_sw ? nullptr : f.planck_function.data();
  v.m.iband_per_g = v.iband.data();
  v.m.ngas = (int)v.gases.size();
  v.m.gases = v.gases.data();
  v.m.solar_irradiance = f.is_sw ? f.solar_irradiance.data() : nullptr;
  v.m.rayleigh_molar_scattering = f.is_sw && !f.rayleigh_molar_scattering.empty() ? f.rayleigh_molar_scattering.data() : nullptr;
}

}  // namespace tool
