// scale_lut [key=value ...] [file.cfg]
//
// Drop-in for the reference executable of the same name (src/ecckd/scale_lut.cpp:23-192; step 4b of
// test/do_all_sw.sh): scales the absorption coefficients of a shortwave CKD definition so that, for one reference
// profile, the direct-beam transmission of every layer and g point equals the line-by-line one.
// Keys (:27-104): input (CKD definition), output, gpointfile (g point of every wavenumber), lblfile (line-by-line
// fluxes with spectral_flux_dn_direct_sw), prepend_path, append_path, log_level.
// The per-g-point sums of the line-by-line flux (ecckd_gmap_sum_rows) and the scaling itself (ecckd_scale_lut) run
// on the GPU.  G points stored inside the CKD file take precedence over "gpointfile" (CkdModel::read_g_points).
#include <sstream>

#include "ckd_file.hpp"

using namespace tool;

int main(int argc, char** argv) {
  return run(argc, argv, [&](Config& config) -> int {
    SearchPath paths;
    paths.configure(config);
    std::string input, output, gpoint_filename, lbl_filename;
    if (!config.read(input, "input")) fail(ECCKD_PARAMETER_ERROR, "\"input\" file not specified");
    if (!config.read(output, "output")) fail(ECCKD_PARAMETER_ERROR, "\"output\" file not specified");
    CkdFile model = read_ckd(paths.find(input));
    const int ng = model.ng, ngas_model = (int)model.gases.size();
    std::vector<int32_t> g_point;
    std::vector<double> wn;
    if (!model.g_point_hr.empty()) {   // stored by create_look_up_table (CkdModel::read_g_points, :52)
      g_point.assign(model.g_point_hr.begin(), model.g_point_hr.end());
      wn = model.wavenumber_hr;
    } else {
      if (!config.read(gpoint_filename, "gpointfile")) fail(ECCKD_PARAMETER_ERROR, "gpointfile not provided");
      NcIn f(paths.find(gpoint_filename));
      wn = f.read("wavenumber");
      for (double v : f.read("g_point")) g_point.push_back((int32_t)v);
      if (ng != *std::max_element(g_point.begin(), g_point.end()) + 1)
        fail(ECCKD_PARAMETER_ERROR, "Number of g-points in %s does not match number in %s", input.c_str(), gpoint_filename.c_str());
    }
    if (!config.read(lbl_filename, "lblfile")) fail(ECCKD_PARAMETER_ERROR, "lblfile not provided");

    // ---- the first profile / zenith angle of the line-by-line file (:83-112) ----
    LOG("Reading %s\n", lbl_filename.c_str());
    NcIn lbl(paths.find(lbl_filename));
    const int imu0 = 0;
    const double mu0 = lbl.read("mu0").at(imu0);
    std::string molecules_str;
    lbl.att_text("constituent_id", molecules_str);
    std::vector<double> pressure_hl = lbl.read("pressure_hl", imu0), temperature_hl = lbl.read("temperature_hl", imu0);
    std::vector<double> mole_fraction = lbl.read("mole_fraction_fl", imu0);        // [ngas_file][nz]
    const int nz = (int)pressure_hl.size() - 1;
    const int ngas_file = (int)(mole_fraction.size() / nz);
    std::vector<size_t> fsh = lbl.shape("spectral_flux_dn_direct_sw");
    const size_t nwav = fsh.back();
    if (nwav != g_point.size()) fail(ECCKD_PARAMETER_ERROR, "%s: %zu spectral points, the g-point file has %zu", lbl_filename.c_str(), nwav, g_point.size());
    std::vector<double> flux = lbl.read("spectral_flux_dn_direct_sw", imu0);        // [nz+1][nwav]
    if (flux.size() != (size_t)(nz + 1) * nwav) fail(ECCKD_PARAMETER_ERROR, "%s: spectral_flux_dn_direct_sw is not (column, half_level, wavenumber)", lbl_filename.c_str());

    // ---- which model gases the file provides (:137-182): "composite" always, names cut at the first hyphen ----
    std::vector<double> vmr((size_t)ngas_model * nz, 0.0);
    std::vector<int> present(ngas_model, 0);
    auto index_of = [&](const std::string& mol) { for (int i = 0; i < ngas_model; ++i) if (model.gases[i].name == mol) return i; return -1; };
    if (index_of("composite") >= 0) present[index_of("composite")] = 1;
    std::stringstream ms(molecules_str);
    for (int igas = 0; igas < ngas_file; ++igas) {
      std::string molecule;
      std::getline(ms, molecule, ' ');
      const size_t hy = molecule.find('-');
      if (hy != std::string::npos) { LOG("  Renaming %s to %s\n", molecule.c_str(), molecule.substr(0, hy).c_str()); molecule = molecule.substr(0, hy); }
      const int gi = index_of(molecule);
      if (gi < 0) { LOG("  Gas %d: %s not found\n", igas, molecule.c_str()); continue; }
      LOG("  Gas %d: %s\n", igas, molecule.c_str());
      present[gi] = 1;
      std::copy(mole_fraction.begin() + (size_t)igas * nz, mole_fraction.begin() + (size_t)(igas + 1) * nz, vmr.begin() + (size_t)gi * nz);
    }

    Device dev;
    LOG("Computing optimal layer optical depths in each g point\n");
    std::vector<double> dwn(nwav, 0.0);
    for (size_t i = 1; i + 1 < nwav; ++i) dwn[i] = 0.5 * (wn[i + 1] - wn[i - 1]);
    if (nwav > 2) { dwn[0] = 0.5 * dwn[1]; dwn[nwav - 1] = 0.5 * dwn[nwav - 2]; }
    DevBuf d_g_point, d_wn, d_dwn, d_flux;
    d_g_point.upload(dev, g_point);
    d_wn.upload(dev, wn);
    d_dwn.upload(dev, dwn);
    d_flux.upload(dev, flux);
    ecckd_gmap* gmap = nullptr;
    ck(ecckd_gmap_create(dev.ctx(), nwav, d_g_point.as<int32_t>(), ng, d_wn.as<double>(), d_dwn.as<double>(), &gmap));
    std::vector<double> sums((size_t)(nz + 1) * ng);
    ck(ecckd_gmap_sum_rows(gmap, nz + 1, d_flux.ptr(), ECCKD_F64, nwav, sums.data()));
    ck(ecckd_gmap_destroy(gmap));

    LOG("Running CKD model\nScaling coefficients in CKD look-up tables\n");
    ModelView mv;
    make_model(model, {}, model.band_number, mv);
    std::vector<std::vector<double>> scaled(ngas_model);
    std::vector<double*> outs(ngas_model);
    for (int i = 0; i < ngas_model; ++i) { scaled[i].resize(model.gases[i].molar_abs.size()); outs[i] = scaled[i].data(); }
    ck(ecckd_scale_lut(dev.ctx(), &mv.m, nz, pressure_hl.data(), temperature_hl.data(), vmr.data(), present.data(), mu0, sums.data(),
                       nullptr, outs.data()));
    for (int i = 0; i < ngas_model; ++i) model.gases[i].molar_abs = scaled[i];

    LOG("Writing %s\n", output.c_str());
    write_ckd(output, model, history_line(argc, argv), config.str());
    return done(0);
  });
}
