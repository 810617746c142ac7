// reorder_spectrum [key=value ...] [file.cfg]
//
// Drop-in for the reference executable of the same name (src/ecckd/reorder_spectrum.cpp:30-310): reads one
// column of a spectral optical-depth file, ranks the wavenumbers of every band by the height of peak cooling
// (longwave) or by the height where the optical depth from the top reaches a threshold (shortwave, `ssi`
// given) and writes the reordering file find_g_points consumes.  Keys: input, output, ssi, iprofile,
// threshold_optical_depth, molecule, wavenumber1, wavenumber2, log_level (:54-83, :232-234).
// The sweep and the per-band stable sort run on the GPU through ecckd_reorder_spectrum (include/ecckd_hip.h).
#include <algorithm>
#include <future>
#include <thread>

#include "tool.hpp"

using namespace tool;

int main(int argc, char** argv) {
  return run(argc, argv, [&](Config& config) -> int {
    std::string input, output, ssi_file_name;
    double threshold_optical_depth = 0.5;
    if (!config.read(input, "input")) fail(ECCKD_PARAMETER_ERROR, "\"input\" file not specified");
    if (!config.read(output, "output")) fail(ECCKD_PARAMETER_ERROR, "\"output\" file not specified");
    const bool do_sw = config.read(ssi_file_name, "ssi");
    LOG(do_sw ? "Assuming shortwave spectral region (ssi provided)\n" : "Assuming longwave spectral region (ssi not provided)\n");
    int iprofile = 0;
    config.read(iprofile, "iprofile");
    config.read(threshold_optical_depth, "threshold_optical_depth");

    SearchPath paths;
    LOG("Reading %s\n", input.c_str());
    // the host-side part of the file (coordinates, 1-D variables) is read while the HIP runtime starts: the two take
    // ~0.05 and ~0.2 s and need nothing of each other
    const std::string input_path = paths.find(input);
    auto spectrum_read = std::async(std::launch::async, [&] { return read_spectrum(input_path, iprofile, false); });
    Device dev;
    Spectrum s = spectrum_read.get();
    // the per-wavenumber results (160 MB at 7.2e6 points): allocated and touched by a second thread while the optical depths
    // stream in - first-touch page faults, 60 ms of a tool that runs 0.4 s
    std::vector<double> key, col;
    std::vector<int16_t> iband;
    std::vector<int32_t> rank;
    std::thread results_ready([&, n = s.nwav] { key.resize(n); col.resize(n); iband.resize(n); rank.resize(n); });
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } results_joiner{results_ready};
    DevOd d_od = read_od_dev(dev, NcIn(paths.find(input)), iprofile, s.nlay, s.nwav);   // file -> pinned buffers -> HBM, decoded there
    std::string molecule = s.molecule;
    config.read(molecule, "molecule");
    LOG("%d layers\n%zu spectral points\n", s.nlay, s.nwav);

    std::vector<double> ssi;
    if (do_sw) {
      LOG("Reading %s\n", ssi_file_name.c_str());
      NcIn f(paths.find(ssi_file_name));
      ssi = f.read("solar_spectral_irradiance");
      if (ssi.size() != s.nwav) fail(ECCKD_PARAMETER_ERROR, "solar_spectral_irradiance has %zu points, the spectrum %zu", ssi.size(), s.nwav);
    }

    std::vector<double> band_bound1, band_bound2;
    if (config.exist("wavenumber1")) {
      config.read(band_bound1, "wavenumber1");
      config.read(band_bound2, "wavenumber2");
    } else {   // :236-241
      band_bound1 = {std::max(0.0, s.wavenumber_cm_1.front() - s.d_wavenumber_cm_1.front())};
      band_bound2 = {s.wavenumber_cm_1.back() + s.d_wavenumber_cm_1.back()};
    }
    const int nband = (int)band_bound1.size();
    if (nband <= 0 || band_bound2.size() != band_bound1.size())
      fail(ECCKD_PARAMETER_ERROR, "Failure to interpret wavenumber1 and wavenumber2 as a list of band boundaries");
    if (nband == 1) LOG("Treating the entire spectrum as one band\n");
    else LOG("Splitting the spectrum into %d bands\n", nband);
    LOG(do_sw ? "Sorting by peak heating\n" : "Sorting by peak cooling\n");

    results_ready.join();
    ck(ecckd_reorder_spectrum_od_dev(dev.ctx(), s.nlay, s.nwav, s.pressure_hl.data(), s.wavenumber_cm_1.data(),
                                     s.d_wavenumber_cm_1.data(), d_od.buf.ptr(), d_od.type, do_sw ? ssi.data() : nullptr,
                                     threshold_optical_depth, nband, band_bound1.data(), band_bound2.data(), key.data(), col.data(),
                                     iband.data(), rank.data()));
    // the file stores the bounds clamped to the range of the data (:268-273), membership used the unclamped ones
    std::vector<double> clamp1 = band_bound1, clamp2 = band_bound2;
    clamp1.front() = std::max(s.wavenumber_cm_1.front(), band_bound1.front());
    clamp2.back() = std::min(s.wavenumber_cm_1.back(), band_bound2.back());
    for (int b = 0; b < nband; ++b) LOG("  Band %d: %g-%g cm-1\n", b, clamp1[b], clamp2[b]);

    LOG("Writing %s\n", output.c_str());
    const std::string history = history_line(argc, argv);
    ck(ecckd_write_order_file(output.c_str(), molecule.c_str(), config.str().c_str(), history.c_str(), nband, clamp1.data(),
                              clamp2.data(), s.nwav, s.wavenumber_cm_1.data(), s.d_wavenumber_cm_1.data(), iband.data(),
                              rank.data(), col.data(), key.data()));
    return done(0);
  });
}
