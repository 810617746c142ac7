// optimize_lut [key=value ...] [file.cfg]
//
// Drop-in for the reference executable of the same name (src/ecckd/optimize_lut.cpp:27-330): refines the molar
// absorption coefficients of a CKD definition so that its fluxes and heating rates match line-by-line training
// fluxes, and writes the optimised definition.
// Keys (:35-157, :204, :245, :248): input, output, gases (the gases to optimise; none = all), training_input,
// relative_to, band_mapping, model_id, flux_weight, flux_profile_weight, broadband_weight, pressure_weight_power,
// prior_error, min_prior_error, max_prior_error, prior_error_scaling, temperature_corr, pressure_corr, conc_corr,
// convergence_criterion, max_iterations, negative_od_penalty, bounded_minimization, max_no_rayleigh_wavenumber,
// prepend_path, append_path, log_level.
// spectral_boundary_weight / erythemal_weight with "gpointfile": the high-resolution surface / top-of-atmosphere
// fluxes of the training files are summed per g point on the GPU (ecckd_gmap_sum_rows, lbl_fluxes.cpp:180-246,
// :300-325); g points stored inside the CKD file take precedence over "gpointfile" (CkdModel::read_g_points).
// Not handled: rayleigh_prior_error > 0 (Rayleigh scattering as an optimised pseudo-gas).
// The cost function, its gradient and the L-BFGS iteration run on the GPU (ecckd_opt_*); adept::Minimizer is
// replaced by the library's own L-BFGS, so iteration counts differ from the reference's.
#include <algorithm>

#include "ckd_file.hpp"

using namespace tool;

namespace {

// one training file as LblFluxes holds it (lbl_fluxes.cpp:52-397), columns replicated per solar zenith angle
struct Scene {
  bool is_sw = false, have_band = false;
  int ncol = 0, nlay = 0, nband = 0, nfilegas = 0;
  std::vector<double> pressure_hl, temperature_hl, vmr_file;   // [ncol][nlay+1], [ncol][nfilegas][nlay]
  std::vector<std::string> file_gases;
  std::vector<double> flux_dn, flux_up;                        // [ncol][nlay+1][nband]
  std::vector<double> wn1, wn2, mu0, albedo;
  double tsi = 0.0;
  // mapped onto the model
  std::vector<double> vmr_fl, relative_dn, relative_up;
  std::vector<int> gas_present, iband_per_g;
  // high-resolution boundary fluxes summed per g point [ncol][ng], erythemal weights [ng] (shortwave)
  std::vector<double> flux_dn_surf_g, flux_up_toa_g, erythemal, boundary_weights;
};

struct GPointMap {   // the g point of every wavenumber ("gpointfile"), sorted once on the device
  ecckd_gmap* gmap = nullptr;
  size_t nwav = 0;
  int ng = 0;
};

// sum the file's narrow bands into the wide bands of band_mapping (:150-176, :272-296)
void map_bands(std::vector<double>& a, size_t nrow, int nb_file, const std::vector<int>& bm, int nb_new) {
  std::vector<double> out(nrow * nb_new, 0.0);
  for (size_t r = 0; r < nrow; ++r)
    for (int j = 0; j < nb_file; ++j)
      if (bm[j] >= 0) out[r * nb_new + bm[j]] += a[r * nb_file + j];
  a.swap(out);
}

Scene read_lbl_fluxes(const std::string& path, const std::vector<int>& band_mapping, const Device& dev, const GPointMap& gp) {
  NcIn f(path);
  Scene s;
  std::vector<size_t> sh = f.shape("pressure_hl");
  const int ncol_file = (int)sh.at(0);
  s.nlay = (int)sh.at(1) - 1;
  const int nhl = s.nlay + 1;
  s.is_sw = f.exist("mu0");
  std::vector<double> p = f.read("pressure_hl"), t = f.read("temperature_hl"), vmr = f.read("mole_fraction_fl");
  s.nfilegas = (int)f.shape("mole_fraction_fl").at(1);
  std::string ids;
  f.att_text("constituent_id", ids);
  for (size_t pos = 0; pos < ids.size();) {
    size_t e = ids.find(' ', pos);
    if (e == std::string::npos) e = ids.size();
    if (e > pos) {
      std::string g = ids.substr(pos, e - pos);
      s.file_gases.push_back(g.substr(0, g.find('-')));       // "h2o-no-continuum" -> "h2o"
    }
    pos = e + 1;
  }
  const int index_sza[3] = {0, 2, 4};                           // :82
  const int nsza = s.is_sw ? 3 : 1;
  s.ncol = ncol_file * nsza;
  auto repeat = [&](const std::vector<double>& a, size_t per_col) {
    std::vector<double> out;
    out.reserve(a.size() * nsza);
    for (int c = 0; c < ncol_file; ++c)
      for (int k = 0; k < nsza; ++k) out.insert(out.end(), a.begin() + c * per_col, a.begin() + (c + 1) * per_col);
    return out;
  };
  s.pressure_hl = repeat(p, nhl);
  s.temperature_hl = repeat(t, nhl);
  s.vmr_file = repeat(vmr, (size_t)s.nfilegas * s.nlay);
  std::string dn_name, up_name, w1_name, w2_name;
  if (s.is_sw) {
    std::vector<double> mu0_all = f.read("mu0");
    const int nsza_file = (int)mu0_all.size();
    if (nsza_file < 5) fail(ECCKD_PARAMETER_ERROR, "%s: %d solar zenith angles, the angles 0, 2 and 4 are used (lbl_fluxes.cpp:82)", path.c_str(), nsza_file);
    for (int c = 0; c < ncol_file; ++c) for (int k = 0; k < nsza; ++k) s.mu0.push_back(mu0_all[index_sza[k]]);
    // columns x selected zenith angles of a (column, sza, ...) variable
    auto pick = [&](const std::vector<double>& a, size_t per) {
      std::vector<double> out;
      out.reserve((size_t)s.ncol * per);
      for (int c = 0; c < ncol_file; ++c)
        for (int k = 0; k < nsza; ++k) {
          const size_t off = ((size_t)c * nsza_file + index_sza[k]) * per;
          out.insert(out.end(), a.begin() + off, a.begin() + off + per);
        }
      return out;
    };
    std::vector<double> bb = pick(f.read("flux_dn_direct_sw"), nhl);
    s.tsi = bb[0] / s.mu0[0];                                   // :116
    if (f.exist("spectral_flux_dn_direct_sw")) { dn_name = "spectral_flux_dn_direct_sw"; up_name = "spectral_flux_up_sw"; }
    else if (f.exist("band_flux_dn_direct_sw")) {
      dn_name = "band_flux_dn_direct_sw"; up_name = "band_flux_up_sw"; w1_name = "band_wavenumber1_sw"; w2_name = "band_wavenumber2_sw";
      s.have_band = true;
    } else fail(ECCKD_PARAMETER_ERROR, "%s: no spectral or band fluxes", path.c_str());
    {
      const std::vector<size_t> fs = f.shape(dn_name);      // (column, mu0, half_level, band)
      if (fs.size() != 4 || (int)fs[0] != ncol_file || (int)fs[1] != nsza_file || (int)fs[2] != nhl || f.shape(up_name) != fs)
        fail(ECCKD_PARAMETER_ERROR, "%s: %s is not (column, mu0, half_level, band) on the file's grid", path.c_str(), dn_name.c_str());
    }
    s.nband = (int)f.shape(dn_name).back();
    s.flux_dn = pick(f.read(dn_name), (size_t)nhl * s.nband);
    s.flux_up = pick(f.read(up_name), (size_t)nhl * s.nband);
  } else {
    if (f.exist("spectral_flux_up_lw")) { dn_name = "spectral_flux_dn_lw"; up_name = "spectral_flux_up_lw"; }
    else if (f.exist("band_flux_up_lw")) {
      dn_name = "band_flux_dn_lw"; up_name = "band_flux_up_lw"; w1_name = "band_wavenumber1_lw"; w2_name = "band_wavenumber2_lw";
      s.have_band = true;
    } else fail(ECCKD_PARAMETER_ERROR, "%s: no spectral or band fluxes", path.c_str());
    {
      const std::vector<size_t> fs = f.shape(dn_name);      // (column, half_level, band)
      if (fs.size() != 3 || (int)fs[0] != ncol_file || (int)fs[1] != nhl || f.shape(up_name) != fs)
        fail(ECCKD_PARAMETER_ERROR, "%s: %s is not (column, half_level, band) on the file's grid", path.c_str(), dn_name.c_str());
    }
    s.nband = (int)f.shape(dn_name).back();
    s.flux_dn = f.read(dn_name);
    s.flux_up = f.read(up_name);
  }
  if (s.have_band) {
    s.wn1 = f.read(w1_name);
    s.wn2 = f.read(w2_name);
    if (!band_mapping.empty()) {
      if ((int)band_mapping.size() != s.nband) fail(ECCKD_PARAMETER_ERROR, "band_mapping has %zu entries, %s has %d bands", band_mapping.size(), path.c_str(), s.nband);
      const int nb_new = *std::max_element(band_mapping.begin(), band_mapping.end()) + 1;
      map_bands(s.flux_dn, (size_t)s.ncol * nhl, s.nband, band_mapping, nb_new);
      map_bands(s.flux_up, (size_t)s.ncol * nhl, s.nband, band_mapping, nb_new);
      std::vector<double> w1(nb_new, 1.0e300), w2(nb_new, -1.0e300);
      for (int j = 0; j < s.nband; ++j) if (band_mapping[j] >= 0) {
        w1[band_mapping[j]] = std::min(w1[band_mapping[j]], s.wn1[j]);
        w2[band_mapping[j]] = std::max(w2[band_mapping[j]], s.wn2[j]);
      }
      s.wn1 = w1; s.wn2 = w2; s.nband = nb_new;
    }
  }
  // high-resolution fluxes at the boundaries -> sums over the wavenumbers of each g point (:180-246, :300-325)
  const std::string hi_dn = s.is_sw ? "spectral_flux_dn_direct_surf_sw" : "spectral_flux_dn_surf_lw";
  const std::string hi_up = s.is_sw ? "spectral_flux_up_toa_sw" : "spectral_flux_up_toa_lw";
  if (f.exist(hi_dn) && f.exist(hi_up)) {
    if (!gp.gmap) {
      WARN("Surface/TOA spectral fluxes ignored because g-point file not provided");
    } else {
      LOG("  Mapping high-resolution boundary fluxes to g-points\n");
      const size_t nwav = f.shape(hi_dn).back();
      if (nwav != gp.nwav) fail(ECCKD_PARAMETER_ERROR, "%s: %zu spectral points, the g-point file has %zu", path.c_str(), nwav, gp.nwav);
      const int nsza_file = s.is_sw ? (int)f.shape(hi_dn).at(1) : 1;
      s.flux_dn_surf_g.assign((size_t)s.ncol * gp.ng, 0.0);
      s.flux_up_toa_g.assign((size_t)s.ncol * gp.ng, 0.0);
      DevBuf d_rows(dev, (size_t)2 * nsza * nwav * sizeof(double));
      std::vector<double> rows((size_t)2 * nsza * nwav), sums((size_t)2 * nsza * gp.ng);
      for (int c = 0; c < ncol_file; ++c) {
        const std::vector<double> dn = f.read(hi_dn, c), up = f.read(hi_up, c);      // LW: [nwav]; SW: [nsza_file][nwav]
        for (int k = 0; k < nsza; ++k) {
          const size_t off = (size_t)(s.is_sw ? index_sza[k] : 0) * nwav;
          if (off + nwav > dn.size() || nsza_file <= (s.is_sw ? index_sza[k] : 0)) fail(ECCKD_PARAMETER_ERROR, "%s: unexpected shape of %s", path.c_str(), hi_dn.c_str());
          std::copy(dn.begin() + off, dn.begin() + off + nwav, rows.begin() + (size_t)k * nwav);
          std::copy(up.begin() + off, up.begin() + off + nwav, rows.begin() + (size_t)(nsza + k) * nwav);
        }
        ck(ecckd_h2d(dev.ctx(), d_rows.ptr(), rows.data(), rows.size() * sizeof(double)));
        ck(ecckd_gmap_sum_rows(gp.gmap, 2 * nsza, d_rows.ptr(), ECCKD_F64, nwav, sums.data()));
        for (int k = 0; k < nsza; ++k) {
          std::copy(sums.begin() + (size_t)k * gp.ng, sums.begin() + (size_t)(k + 1) * gp.ng, s.flux_dn_surf_g.begin() + ((size_t)c * nsza + k) * gp.ng);
          std::copy(sums.begin() + (size_t)(nsza + k) * gp.ng, sums.begin() + (size_t)(nsza + k + 1) * gp.ng,
                    s.flux_up_toa_g.begin() + ((size_t)c * nsza + k) * gp.ng);
        }
      }
      if (s.is_sw) {   // :198-230
        s.erythemal.resize(gp.ng);
        ck(ecckd_gmap_erythemal_spectrum(gp.gmap, s.erythemal.data()));
      }
    }
  }
  if (s.is_sw) {   // effective spectral albedo (:147-148, :166-167): surface sums over every column
    s.albedo.assign(s.nband, 0.0);
    std::vector<double> up(s.nband, 0.0), dn(s.nband, 0.0);
    for (int c = 0; c < s.ncol; ++c)
      for (int b = 0; b < s.nband; ++b) {
        up[b] += s.flux_up[((size_t)c * nhl + s.nlay) * s.nband + b];
        dn[b] += s.flux_dn[((size_t)c * nhl + s.nlay) * s.nband + b];
      }
    for (int b = 0; b < s.nband; ++b) s.albedo[b] = up[b] / dn[b];
  }
  return s;
}

// CkdModel::iband_per_g (ckd_model.h:287-306)
std::vector<int> iband_per_g(const CkdFile& m, const std::vector<double>& wn1, const std::vector<double>& wn2) {
  const size_t nwav = m.wavenumber1.size();
  std::vector<int> iband(m.ng, -1);
  for (size_t ib = 0; ib < wn1.size(); ++ib)
    for (int g = 0; g < m.ng; ++g) {
      double weight = 0.0;
      for (size_t i = 0; i < nwav; ++i)
        if (m.wavenumber1[i] >= wn1[ib] && m.wavenumber2[i] <= wn2[ib]) weight += m.gpoint_fraction[(size_t)g * nwav + i];
      if (weight > 0.05 && (weight < 0.95 || weight > 1.05)) { std::fprintf(stderr, "*** Error: G-points do not lie entirely within requested bands\n"); throw Fatal{1, "g points straddle bands"}; }
      if (weight > 0.5) iband[g] = (int)ib;
    }
  for (int b : iband) if (b < 0) throw Fatal{1, "Some g-points not inside a band"};
  return iband;
}

// LblFluxes::make_gas_mapping: the file's gases onto the model's
void map_gases(Scene& s, const CkdFile& m) {
  const size_t ngas = m.gases.size();
  s.vmr_fl.assign((size_t)s.ncol * ngas * s.nlay, 0.0);
  s.gas_present.assign(ngas, 0);
  for (size_t i = 0; i < ngas; ++i) {
    auto it = std::find(s.file_gases.begin(), s.file_gases.end(), m.gases[i].name);
    if (it == s.file_gases.end()) continue;
    const size_t j = it - s.file_gases.begin();
    s.gas_present[i] = 1;
    for (int c = 0; c < s.ncol; ++c)
      std::copy(s.vmr_file.begin() + ((size_t)c * s.nfilegas + j) * s.nlay, s.vmr_file.begin() + ((size_t)c * s.nfilegas + j + 1) * s.nlay,
                s.vmr_fl.begin() + ((size_t)c * ngas + i) * s.nlay);
  }
}

ecckd_opt_scene scene_view(const Scene& s) {
  ecckd_opt_scene o;
  std::memset(&o, 0, sizeof o);
  o.ncol = s.ncol; o.nlay = s.nlay; o.nband = s.nband;
  o.pressure_hl = s.pressure_hl.data();
  o.temperature_hl = s.temperature_hl.data();
  o.vmr_fl = s.vmr_fl.data();
  o.gas_present = s.gas_present.data();
  o.flux_dn = s.flux_dn.data();
  o.flux_up = s.flux_up.data();
  if (s.is_sw) { o.mu0 = s.mu0.data(); o.tsi = s.tsi; o.albedo = s.albedo.data(); }
  if (!s.relative_dn.empty()) { o.relative_flux_dn = s.relative_dn.data(); o.relative_flux_up = s.relative_up.data(); }
  if (!s.flux_dn_surf_g.empty()) { o.spectral_flux_dn_surf = s.flux_dn_surf_g.data(); o.spectral_flux_up_toa = s.flux_up_toa_g.data(); }
  if (!s.boundary_weights.empty()) o.spectral_boundary_weights = s.boundary_weights.data();
  return o;
}

}  // namespace

int main(int argc, char** argv) {
  return run(argc, argv, [&](Config& config) -> int {
    SearchPath paths;
    paths.configure(config);
    std::string input, output;
    if (!config.read(input, "input")) fail(ECCKD_PARAMETER_ERROR, "\"input\" file not specified");
    if (!config.read(output, "output")) fail(ECCKD_PARAMETER_ERROR, "\"output\" file not specified");
    std::vector<std::string> gas_list = config.read_list("gases");
    LOG("Optimizing coefficients of:");
    if (gas_list.empty()) LOG(" ALL GASES\n");
    else { for (auto& g : gas_list) LOG(" %s", g.c_str()); LOG("\n"); }

    ecckd_opt_config oc;
    oc.flux_weight = 0.02; oc.flux_profile_weight = 0.0; oc.broadband_weight = 0.5; oc.spectral_boundary_weight = 0.0;
    oc.negative_od_penalty = 1.0e4; oc.pressure_weight_power = 0.5;
    oc.prior_error = -1.0; oc.min_prior_error = -1.0; oc.max_prior_error = -1.0; oc.prior_error_scaling = 1.0;
    oc.pressure_corr = 0.5; oc.temperature_corr = 0.5; oc.conc_corr = 0.5; oc.cap_relative_linear = 0.8;   // :185
    double rayleigh_prior_error = 0.0, erythemal_weight = 0.0, convergence_criterion = 0.02, max_no_rayleigh_wavenumber = 10000.0;
    int max_iterations = 3000;
    bool is_bounded = true;
    std::string model_id;
    config.read(oc.flux_weight, "flux_weight");
    config.read(oc.flux_profile_weight, "flux_profile_weight");
    config.read(oc.broadband_weight, "broadband_weight");
    config.read(oc.spectral_boundary_weight, "spectral_boundary_weight");
    config.read(erythemal_weight, "erythemal_weight");
    config.read(oc.pressure_weight_power, "pressure_weight_power");
    config.read(oc.prior_error, "prior_error");
    config.read(oc.min_prior_error, "min_prior_error");
    config.read(oc.max_prior_error, "max_prior_error");
    config.read(oc.prior_error_scaling, "prior_error_scaling");
    config.read(rayleigh_prior_error, "rayleigh_prior_error");
    config.read(oc.temperature_corr, "temperature_corr");
    config.read(oc.pressure_corr, "pressure_corr");
    config.read(oc.conc_corr, "conc_corr");
    config.read(convergence_criterion, "convergence_criterion");
    config.read(model_id, "model_id");
    config.read(max_no_rayleigh_wavenumber, "max_no_rayleigh_wavenumber");
    config.read(max_iterations, "max_iterations");
    config.read(oc.negative_od_penalty, "negative_od_penalty");
    if (config.exist("bounded_minimization")) config.read(is_bounded, "bounded_minimization");
    bool remove_min_max = false;                                            // :243-244
    if (config.exist("remove_min_max")) config.read(remove_min_max, "remove_min_max");
    if (rayleigh_prior_error > 0.0) fail(ECCKD_PARAMETER_ERROR, "rayleigh_prior_error > 0 (optimised Rayleigh scattering) is not supported by this tool");
    std::vector<int> band_mapping;
    if (config.exist("band_mapping")) config.read(band_mapping, "band_mapping");

    LOG("Reading %s\n", input.c_str());
    CkdFile model = read_ckd(paths.find(input));
    model.model_id = model_id;
    for (const std::string& g : gas_list) {
      bool found = false;
      for (const GasTable& t : model.gases) found = found || t.name == g;
      if (!found) WARN("gas \"%s\" is not in %s", g.c_str(), input.c_str());
    }

    // ---- the g point of every wavenumber, for the high-resolution boundary fluxes (:166-182) ----
    Device dev;
    GPointMap gp;
    DevBuf d_g_point, d_wn, d_dwn;
    std::string gpoint_filename;
    std::vector<int32_t> g_point;
    std::vector<double> wn;
    if (!model.g_point_hr.empty()) {   // stored by create_look_up_table (CkdModel::read_g_points, :166)
      g_point.assign(model.g_point_hr.begin(), model.g_point_hr.end());
      wn = model.wavenumber_hr;
    } else if (config.read(gpoint_filename, "gpointfile")) {
      NcIn f(paths.find(gpoint_filename));
      for (double v : f.read("g_point")) g_point.push_back((int32_t)v);
      if (model.ng != *std::max_element(g_point.begin(), g_point.end()) + 1)
        fail(ECCKD_PARAMETER_ERROR, "Number of g-points in %s does not match number in %s", input.c_str(), gpoint_filename.c_str());
      wn = f.read("wavenumber");
    }
    if (!g_point.empty()) {
      std::vector<double> dwn(wn.size(), 0.0);
      for (size_t i = 1; i + 1 < wn.size(); ++i) dwn[i] = 0.5 * (wn[i + 1] - wn[i - 1]);
      if (wn.size() > 2) { dwn[0] = 0.5 * dwn[1]; dwn[wn.size() - 1] = 0.5 * dwn[wn.size() - 2]; }
      gp.nwav = g_point.size();
      gp.ng = model.ng;
      d_g_point.upload(dev, g_point);
      d_wn.upload(dev, wn);
      d_dwn.upload(dev, dwn);
      ck(ecckd_gmap_create(dev.ctx(), gp.nwav, d_g_point.as<int32_t>(), gp.ng, d_wn.as<double>(), d_dwn.as<double>(), &gp.gmap));
    }

    auto load = [&](const std::string& name) {
      const std::string path = paths.find(name);
      LOG("Reading %s\n", path.c_str());
      Scene s = read_lbl_fluxes(path, band_mapping, dev, gp);
      if (s.is_sw && erythemal_weight > 0.0 && !s.erythemal.empty()) {   // solve_adept.cpp:182
        s.boundary_weights.resize(s.erythemal.size());
        for (size_t g = 0; g < s.erythemal.size(); ++g) s.boundary_weights[g] = erythemal_weight * s.erythemal[g];
      }
      if (s.is_sw != model.is_sw) fail(ECCKD_PARAMETER_ERROR, "%s and the CKD model are not for the same spectral region", path.c_str());
      if (s.have_band) s.iband_per_g = iband_per_g(model, s.wn1, s.wn2);          // :273-276
      else s.iband_per_g = model.band_number;
      if (s.is_sw && s.have_band) {   // LblFluxes::mask_rayleigh_up (lbl_fluxes.cpp:415-429), :278-283
        for (int b = 0; b < s.nband; ++b)
          if (s.wn2[b] > max_no_rayleigh_wavenumber) {
            s.albedo[b] = 0.0;
            for (size_t r = 0; r < (size_t)s.ncol * (s.nlay + 1); ++r) s.flux_up[r * s.nband + b] = 0.0;
          }
      }
      map_gases(s, model);
      return s;
    };

    // ---- optional "relative_to" scene: CKD fluxes at the initial coefficients (:204-236) ----
    std::string relative_to_file;
    Scene rel;
    std::vector<double> rel_flux;   // [ncol][2][nlay+1][ng]
    const bool have_rel = config.read(relative_to_file, "relative_to");
    if (have_rel) {
      LOG("Fluxes will be fitted relative to those of %s\n", relative_to_file.c_str());
      rel = load(relative_to_file);
      ModelView mv;
      make_model(model, gas_list, rel.iband_per_g, mv);
      ecckd_opt_scene sv = scene_view(rel);
      ecckd_opt* ro = nullptr;
      ck(ecckd_opt_create(dev.ctx(), &mv.m, 1, &sv, &oc, &ro));
      std::vector<double> x0(ecckd_opt_nx(ro));
      ck(ecckd_opt_initial_state(ro, x0.data(), nullptr, nullptr));
      rel_flux.resize((size_t)rel.ncol * 2 * (rel.nlay + 1) * model.ng);
      ck(ecckd_opt_forward_ex(ro, x0.data(), 1 /* od = value(aod): no clamp, :231-234 */, nullptr, rel_flux.data()));
      ck(ecckd_opt_destroy(ro));
    }

    // ---- training scenes (:238-300) ----
    std::vector<Scene> scenes;
    for (const std::string& name : config.read_list("training_input")) {
      Scene s = load(name);
      if (have_rel) {
        if (s.ncol != rel.ncol || s.nlay != rel.nlay || s.nband != rel.nband)
          fail(ECCKD_PARAMETER_ERROR, "%s does not match the shape of the relative_to fluxes", name.c_str());
        for (size_t i = 0; i < s.flux_dn.size(); ++i) { s.flux_dn[i] -= rel.flux_dn[i]; s.flux_up[i] -= rel.flux_up[i]; }   // LblFluxes::subtract
        const size_t per = (size_t)(s.nlay + 1) * model.ng;
        s.relative_dn.resize((size_t)s.ncol * per);
        s.relative_up.resize((size_t)s.ncol * per);
        for (int c = 0; c < s.ncol; ++c) {
          std::copy(rel_flux.begin() + ((size_t)c * 2) * per, rel_flux.begin() + ((size_t)c * 2 + 1) * per, s.relative_dn.begin() + c * per);
          std::copy(rel_flux.begin() + ((size_t)c * 2 + 1) * per, rel_flux.begin() + ((size_t)c * 2 + 2) * per, s.relative_up.begin() + c * per);
        }
      }
      scenes.push_back(std::move(s));
    }
    if (scenes.empty()) fail(ECCKD_PARAMETER_ERROR, "\"training_input\" not specified");

    ModelView mv;
    make_model(model, gas_list, scenes.back().iband_per_g, mv);
    std::vector<ecckd_opt_scene> views;
    for (const Scene& s : scenes) views.push_back(scene_view(s));
    ecckd_opt* opt = nullptr;
    ck(ecckd_opt_create(dev.ctx(), &mv.m, (int)views.size(), views.data(), &oc, &opt));
    const size_t nx = ecckd_opt_nx(opt);
    LOG("Optimizing %zu coefficients against %zu training file(s)\n", nx, scenes.size());
    std::vector<double> x(nx);
    int status = 0, niter = 0;
    double J = 0.0, gnorm = 0.0;
    // report_progress (solve_adept.cpp:295-299) and the Timer of :214-231, whose table goes to stderr when it is destroyed
    ck(ecckd_opt_set_progress(opt, [](int it, double cost, double gn, void*) {
      LOG("Iteration %d: cost function = %g, gradient norm = %g\n", it, cost, gn);
    }, nullptr));
    LOG(is_bounded ? "  Minimization is bounded\n" : "  Minimization is unbounded\n");
    ck(ecckd_opt_minimize(opt, max_iterations, convergence_criterion, is_bounded ? 1 : 0, x.data(), &status, &niter, &J, &gnorm));
    static const char* const status_str[] = {"Converged", "Initial state", "Maximum iterations reached", "Failed to converge",
                                             "Direction-finding failure", "Bound reached", "Invalid cost function", "Invalid gradient",
                                             "Invalid bounds"};
    LOG("Minimizer status: %s after %d iterations, cost function %g, gradient norm %g\n",
        status >= 0 && status < 9 ? status_str[status] : "unknown", niter, J, gnorm);
    for (size_t i = 0; i < model.gases.size(); ++i) ck(ecckd_opt_coefficients(opt, x.data(), (int)i, model.gases[i].molar_abs.data()));
    {
      double t_min = 0.0, t_prior = 0.0, t_rt = 0.0;   // Timer::print (Timer.h:59-69)
      ck(ecckd_opt_timings(opt, &t_min, &t_prior, &t_rt));
      std::fprintf(stderr, "3 activities:\n%10g s: minimizer\n%10g s: a-priori\n%10g s: radiative transfer\n%10g s: Total\n", t_min, t_prior,
                   t_rt, t_min + t_prior + t_rt);
    }
    ck(ecckd_opt_destroy(opt));
    if (gp.gmap) ck(ecckd_gmap_destroy(gp.gmap));

    if (remove_min_max)   // ckd_model.save_min_max(false), :308-310: the <gas>_molar_absorption_coeff_min / _max tables are not written
      for (GasTable& t : model.gases) { t.min_molar_abs.clear(); t.max_molar_abs.clear(); }
    LOG("Writing %s\n", output.c_str());
    write_ckd(output, model, history_line(argc, argv), config.str());
    if (status >= 6) {   // :315-319
      std::fprintf(stderr, "*** Error: minimizer returned an anomalous status\n");
      return 1;
    }
    return done(0);
  });
}
