// run_ckd [key=value ...] [file.cfg]
//
// Drop-in for the reference executable of the same name (src/ecckd/run_ckd.cpp:27-373): evaluates a CKD
// definition on the profiles of a CKDMIP-style concentration file - optical depth per g point (total and per
// gas), Planck function or incoming solar flux, and the fluxes of the reference's own no-scattering radiative
// transfer - and writes them with the reference's variable names.
// Keys (:50-90): ckd_model, input, output, gases (restricts the gas list), co2_scaling, ch4_scaling, n2o_scaling,
// cfc11_scaling, cfc12_scaling, write_od_only, tsi, prepend_path, append_path, log_level.
// All of the arithmetic is ecckd_run_ckd (include/ecckd_hip.h) on the GPU.
#include "ckd_file.hpp"

using namespace tool;

int main(int argc, char** argv) {
  return run(argc, argv, [&](Config& config) -> int {
    SearchPath paths;
    paths.configure(config);
    std::string ckd_file, input_file, output_file;
    if (!config.read(ckd_file, "ckd_model")) fail(ECCKD_PARAMETER_ERROR, "\"ckd_model\" not specified");
    if (!config.read(input_file, "input")) fail(ECCKD_PARAMETER_ERROR, "\"input\" file not specified");
    if (!config.read(output_file, "output")) fail(ECCKD_PARAMETER_ERROR, "\"output\" file not specified");
    const std::vector<std::string> gas_list = config.read_list("gases");
    const char* const scaled[] = {"co2", "ch4", "n2o", "cfc11", "cfc12"};
    double scaling[5] = {-1.0, -1.0, -1.0, -1.0, -1.0};
    for (int k = 0; k < 5; ++k) config.read(scaling[k], std::string(scaled[k]) + "_scaling");
    bool write_od_only = false;
    config.read(write_od_only, "write_od_only");
    double tsi = 1361.0;
    config.read(tsi, "tsi");

    CkdFile model = read_ckd(paths.find(ckd_file));
    const std::string dom = model.is_sw ? "sw" : "lw";
    const int ngas = (int)model.gases.size(), ng = model.ng;

    LOG("Reading %s\n", input_file.c_str());
    NcIn input(paths.find(input_file));
    std::vector<size_t> sh = input.shape("pressure_hl");
    const int ncol = (int)sh.at(0), nhl = (int)sh.at(1), nlay = nhl - 1;
    std::vector<double> pressure_hl = input.read("pressure_hl"), temperature_hl = input.read("temperature_hl");

    // ---- concentrations of the gases in model order; the gas list and the scalings (:270-307) ----
    std::vector<double> vmr((size_t)ncol * ngas * nlay, 1.0);
    std::vector<int> present(ngas, 1);
    for (int i = 0; i < ngas; ++i) {
      const std::string& mol = model.gases[i].name;
      if (!gas_list.empty() && std::find(gas_list.begin(), gas_list.end(), mol) == gas_list.end()) {
        LOG("  Skipping %s\n", mol.c_str());
        present[i] = 0;
        continue;
      }
      const std::string var = mol + "_mole_fraction_fl";
      if (!input.exist(var)) {
        if (model.gases[i].conc != CONC_NONE)
          fail(ECCKD_PARAMETER_ERROR, "%s not found in %s but the absorption of %s depends on its concentration", var.c_str(), input_file.c_str(), mol.c_str());
        LOG("  Computing optical depth of %s assuming no concentration dependence\n", mol.c_str());
        continue;
      }
      std::vector<double> v = input.read(var);
      double f = -1.0;
      for (int k = 0; k < 5; ++k) if (mol == scaled[k]) f = scaling[k];
      if (f >= 0.0) { LOG("  Computing optical depth of %s from concentration scaled by %g\n", mol.c_str(), f); for (double& x : v) x *= f; }
      else LOG("  Computing optical depth of %s\n", mol.c_str());
      if (v.size() != (size_t)ncol * nlay) fail(ECCKD_PARAMETER_ERROR, "%s is not (column, level)", var.c_str());
      for (int c = 0; c < ncol; ++c) std::copy(v.begin() + (size_t)c * nlay, v.begin() + (size_t)(c + 1) * nlay, vmr.begin() + ((size_t)c * ngas + i) * nlay);
    }

    Device dev;
    ModelView mv;
    make_model(model, {}, model.band_number, mv);
    std::vector<double> mu0(ncol, 0.5);   // REFERENCE_COS_SZA (:358)
    ecckd_opt_scene sc;
    std::memset(&sc, 0, sizeof sc);
    sc.ncol = ncol; sc.nlay = nlay; sc.nband = (int)model.wavenumber1_band.size();
    sc.pressure_hl = pressure_hl.data();
    sc.temperature_hl = temperature_hl.data();
    sc.vmr_fl = vmr.data();
    sc.gas_present = present.data();
    if (model.is_sw) { sc.mu0 = mu0.data(); sc.tsi = tsi; }

    const size_t ncell = (size_t)ncol * nlay * ng, nlev = (size_t)ncol * nhl * ng;
    std::vector<double> od(ncell), ray(model.is_sw ? ncell : 0), planck(nlev), flux(2 * nlev);
    ck(ecckd_run_ckd(dev.ctx(), &mv.m, &sc, od.data(), model.is_sw ? ray.data() : nullptr, planck.data(), flux.data()));

    LOG("Writing %s\n", output_file.c_str());
    NcOut file(output_file);
    file.dim("column", ncol);
    file.dim("level", nlay);
    file.dim("half_level", nhl);
    file.dim("g_point", ng);
    file.var("pressure_hl", NC_FLOAT_T, {"column", "half_level"}, "Pressure", "Pa");
    file.var("optical_depth", NC_FLOAT_T, {"column", "level", "g_point"},
             model.is_sw ? "Layer optical depth due to molecular absorption" : "Layer optical depth");
    if (!write_od_only)
      for (const GasTable& g : model.gases)
        file.var(g.name + "_optical_depth", NC_FLOAT_T, {"column", "level", "g_point"}, (g.name + " optical depth").c_str());
    if (!model.is_sw) {
      file.var("planck_hl", NC_FLOAT_T, {"column", "half_level", "g_point"}, "Planck function", "W m-2");
    } else {
      file.var("incoming_sw", NC_FLOAT_T, {"column", "g_point"}, "Incoming shortwave flux at top-of-atmosphere in direction of sun", "W m-2");
      file.var("rayleigh_optical_depth", NC_FLOAT_T, {"column", "level", "g_point"}, "Layer optical depth due to Rayleigh scattering");
    }
    if (!write_od_only) {
      if (!model.is_sw) {
        file.var("planck_surf", NC_FLOAT_T, {"column", "g_point"}, "Planck function at surface", "W m-2");
        file.var("spectral_flux_up_lw", NC_FLOAT_T, {"column", "half_level", "g_point"}, "Spectral upwelling longwave flux", "W m-2");
        file.var("spectral_flux_dn_lw", NC_FLOAT_T, {"column", "half_level", "g_point"}, "Spectral downwelling longwave flux", "W m-2");
        file.var("flux_up_lw", NC_FLOAT_T, {"column", "half_level"}, "Upwelling longwave flux", "W m-2");
        file.var("flux_dn_lw", NC_FLOAT_T, {"column", "half_level"}, "Downwelling longwave flux", "W m-2");
      } else {
        file.var("spectral_flux_dn_direct_sw", NC_FLOAT_T, {"column", "half_level", "g_point"}, "Spectral downwelling direct shortwave flux", "W m-2");
        file.var("flux_dn_direct_sw", NC_FLOAT_T, {"column", "half_level"}, "Downwelling direct shortwave flux", "W m-2");
      }
    }
    file.att("Spectral optical depth from ecCKD gas optics scheme", "title");
    if (!model.model_id.empty()) file.att(model.model_id, "model_id");
    file.att(history_line(argc, argv), "history");
    for (const char* a : {"experiment", "experiment_id", "sub_experiment", "sub_experiment_id"}) {
      std::string v;
      if (input.att_text(a, v) && !v.empty()) file.att(v, a);
    }
    file.end_define();
    file.write("pressure_hl", pressure_hl);
    file.write("optical_depth", od);
    if (!write_od_only) {   // one evaluation per gas for the "<gas>_optical_depth" variables; absent gases stay zero
      std::vector<double> od_gas(ncell);
      for (int i = 0; i < ngas; ++i) {
        if (!present[i]) { std::fill(od_gas.begin(), od_gas.end(), 0.0); }
        else {
          std::vector<int> one(ngas, 0);
          one[i] = 1;
          sc.gas_present = one.data();
          ck(ecckd_run_ckd(dev.ctx(), &mv.m, &sc, od_gas.data(), nullptr, nullptr, nullptr));
        }
        file.write(model.gases[i].name + "_optical_depth", od_gas);
      }
    }
    auto level = [&](const std::vector<double>& a, int lev) {   // one half level of a [ncol][nhl][ng] array
      std::vector<double> out((size_t)ncol * ng);
      for (int c = 0; c < ncol; ++c) std::copy(a.begin() + ((size_t)c * nhl + lev) * ng, a.begin() + ((size_t)c * nhl + lev + 1) * ng, out.begin() + (size_t)c * ng);
      return out;
    };
    auto part = [&](int which) {   // [ncol][2][nhl][ng] -> [ncol][nhl][ng]
      std::vector<double> out(nlev);
      for (int c = 0; c < ncol; ++c) std::copy(flux.begin() + ((size_t)c * 2 + which) * nhl * ng, flux.begin() + ((size_t)c * 2 + which + 1) * nhl * ng, out.begin() + (size_t)c * nhl * ng);
      return out;
    };
    auto broadband = [&](const std::vector<double>& a) {   // sum over g points, ascending (sum(flux, 2))
      std::vector<double> out((size_t)ncol * nhl, 0.0);
      for (size_t r = 0; r < out.size(); ++r) for (int g = 0; g < ng; ++g) out[r] += a[r * ng + g];
      return out;
    };
    if (!model.is_sw) {
      file.write("planck_hl", planck);
      if (!write_od_only) {
        file.write("planck_surf", level(planck, nlay));
        const std::vector<double> dn = part(0), up = part(1);
        file.write("spectral_flux_up_lw", up);
        file.write("spectral_flux_dn_lw", dn);
        file.write("flux_up_lw", broadband(up));
        file.write("flux_dn_lw", broadband(dn));
      }
    } else {
      file.write("rayleigh_optical_depth", ray);
      file.write("incoming_sw", level(planck, 0));
      if (!write_od_only) {
        const std::vector<double> dn = part(0);
        file.write("spectral_flux_dn_direct_sw", dn);
        file.write("flux_dn_direct_sw", broadband(dn));
      }
    }
    file.close();
    return done(0);
  });
}
