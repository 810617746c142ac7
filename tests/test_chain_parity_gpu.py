"""The chain of test/do_all_lw.sh through the command-line tools (every hand-over a NetCDF file) against the oracle chain that
rounds where the reference's files round (tools/e2e_bench.py: FLOAT sorting variable in the ordering / g-points files,
write_order.cpp:45-139; FLOAT tables, temperatures, pressures and mole fractions in the CKD-definition file,
ckd_model.cpp:318-326, :418-445), at 2^16 points, 4 gases (water vapour as a look-up table), 13 bands:

  * identical g-point maps;
  * the raw models' heating rates within 1e-6 K/day (RMS, plot/calc_hr_error.m weights) - evaluated in DOUBLE from the two
    chains' models: run_ckd's own file holds FLOAT fluxes (run_ckd.cpp:221-230), and 3e-5 W m-2 of rounding on a 400 W m-2 flux
    is 0.03 K/day in a top layer of 1 Pa, so through that file the bound is the file's, not the chain's (asserted as such);
  * where a residual stays, the coefficient that carries it is named in the assertion message.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def test_tools_chain_against_the_oracle_chain_with_the_references_hand_overs(ctx, tmp_path):
    import e2e_bench as e2e
    from ecckd_amd import ncio
    os.environ.setdefault("OMP_NUM_THREADS", str(min(16, len(os.sched_getaffinity(0)))))
    n, nlay = 1 << 16, 54
    d = str(tmp_path)
    saved = dict(e2e.OPT)
    e2e.OPT["max_iterations"] = 8            # the optimised model only has to exist; its trajectory is unpinned
    try:
        inp = e2e.make_inputs(ctx, d, n, nlay, nlines=1500)
        c_secs, c_out = e2e.cpu_chain(ctx, d, inp)
        s_secs, s_out = e2e.gpu_chain(d, inp)
    finally:
        e2e.OPT.update(saved)
    p1 = inp["p1"]
    gp_tools = ncio.read_g_points(os.path.join(d, "gpoints.nc"))["g_point"]
    assert np.array_equal(gp_tools, c_out["g_point"]), "%d wavenumbers fall into another g point" % int((gp_tools != c_out["g_point"]).sum())
    # the raw models, table by table
    mt = ncio.read_ckd_model(os.path.join(d, "raw_ckd.nc"))
    mc = c_out["models"]["raw"]
    tables = e2e.compare_models(mt, mc)
    assert np.array_equal(mt["temperature"], mc["temperature"]) and np.array_equal(mt["log_pressure"], mc["log_pressure"])
    assert np.allclose(mt["planck_function"], mc["planck_function"], rtol=2e-7, atol=0.0)       # one FLOAT rounding at most
    # heating rates of the evaluation profiles from the two models, in double
    ft = e2e.oracle_fluxes_of_model(mt, inp, c_out["cfg"])
    fc = e2e.oracle_fluxes_of_model(mc, inp, c_out["cfg"])
    hr_t, hr_c = e2e.hr_k_per_day(p1, *ft), e2e.hr_k_per_day(p1, *fc)
    rms = e2e.hr_rms_difference(p1, hr_t, hr_c)
    col, lay = np.unravel_index(int(np.argmax(np.abs(hr_t - hr_c))), hr_t.shape)
    print("raw models: heating-rate RMS difference %.3e K/day (largest %.3e K/day in profile %d, layer %d); tables: %s"
          % (rms, float(np.abs(hr_t - hr_c).max()), col, lay, tables))
    assert rms <= 1.0e-6, ("raw heating rates %.3e K/day apart; largest difference in profile %d layer %d; tables %s" % (rms, col, lay, tables))
    # through run_ckd's FLOAT file the bound is the file's rounding: both chains rounded alike agree to a few FLOAT roundings of
    # the fluxes, not to 1e-6 K/day
    hr_file_t = e2e.hr_k_per_day(p1, *s_out["raw"])
    hr_file_c = e2e.hr_k_per_day(p1, *c_out["raw"])
    flux_ulps = np.abs(s_out["raw"][0] - c_out["raw"][0]) / np.maximum(np.spacing(c_out["raw"][0].astype(np.float32)).astype(np.float64), 1e-300)
    assert flux_ulps.max() <= 1.0, "a FLOAT flux of the two files differs by more than one rounding"
    through_files = e2e.hr_rms_difference(p1, hr_file_t, hr_file_c)
    print("through the FLOAT flux files: %.3e K/day (%d of %d fluxes one FLOAT rounding apart)"
          % (through_files, int((flux_ulps > 0).sum()), flux_ulps.size))
    assert through_files <= 5.0e-3
