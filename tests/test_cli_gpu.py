"""The command-line tools (bin/reorder_spectrum, bin/find_g_points) as a user of the reference would run them:
`exe [key=value ...] [file.cfg]` on NetCDF files, exit code 0 or one of EsaExitCodes.h.  The tools are C++ above
the C ABI and never see Python or torch; their output files must hold exactly what the host mirrors in
ecckd_amd.pipeline produce (which tests/test_pipeline_gpu.py ties to the CPU oracle and the reference search)."""
import os
import subprocess

import numpy as np
import pytest
from scipy.io import netcdf_file

from ecckd_amd import synthetic as syn
from test_pipeline_gpu import NLAY, _write_spectrum

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin")


def run_tool(name, *args, cwd=None):
    exe = os.path.join(BIN, name)
    if not os.path.exists(exe):                      # a fresh checkout: build the library and the tools first
        import sys
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    assert os.path.exists(exe), f"{exe} not built (python -c 'import __graft_entry__ as g; g.build()')"
    return subprocess.run([exe, *[str(a) for a in args]], cwd=cwd, capture_output=True, text=True, timeout=600)


def _nc(path):
    return netcdf_file(str(path), "r", mmap=False)


def _make_lw_files(d, nwav=12000):
    p = syn.pressure_grid(NLAY)
    t_hl = syn.temperature_profile(p)
    wn, _ = syn.wavenumber_grid(nwav)
    for g, (seed, scale, vmr) in {"h2o": (41, 30.0, 5e-3), "co2": (43, 8.0, 4e-4)}.items():
        od = syn.optical_depth(np, p, wn, syn.SEED_BASE + seed, nlines=40, column_scale=scale, dtype="float32")
        _write_spectrum(d / f"{g}.nc", g, p, t_hl, wn, od, vmr)
    return wn


LW_CFG = """# written by tests/test_cli_gpu.py
append_path "{d}"
iprofile 0
averaging_method "transmission"
tolerance_tolerance 0.02
flux_weight 0.02
max_iterations 40
heating_rate_tolerance 0.08
gases h2o co2

\\begin h2o
  input h2o.nc
  reordering_input order_h2o.nc
  background_input "co2.nc"
\\end h2o
\\begin co2
  input co2.nc
  reordering_input order_co2.nc
  background_input co2_bg_is_h2o.nc
  min_g_points 2 1
\\end co2
"""


def test_reorder_and_find_g_points_lw(ctx, tmp_path):
    from ecckd_amd import ncio, pipeline
    d = tmp_path
    _make_lw_files(d)
    os.symlink(d / "h2o.nc", d / "co2_bg_is_h2o.nc")
    b1, b2 = np.array([0.0, 1300.0]), np.array([1300.0, 3260.0])
    for g in ("h2o", "co2"):
        r = run_tool("reorder_spectrum", f"input={d}/{g}.nc", f"output={d}/order_{g}.nc", "wavenumber1=0 1300",
                     "wavenumber2=1300 3260", "iprofile=0")
        assert r.returncode == 0, r.stderr
        assert "Splitting the spectrum into 2 bands" in r.stdout
        ref = pipeline.reorder_spectrum(ctx, d / f"{g}.nc", d / f"pyorder_{g}.nc", b1, b2)
        got, exp = ncio.read_order(d / f"order_{g}.nc"), ncio.read_order(d / f"pyorder_{g}.nc")
        for k in ("rank", "band_number", "sorting_variable", "wavenumber", "wavenumber1_band", "wavenumber2_band"):
            assert np.array_equal(got[k], exp[k]), (g, k)
        assert got["molecule"] == g and np.array_equal(got["rank"], ref["rank"])
        f = _nc(d / f"order_{g}.nc")
        assert b"wavenumber1={0 1300}" in f.config and b"reorder_spectrum" in f.history
        assert f.variables["rank"].typecode() == "i" and f.variables["band_number"].typecode() == "h"
        f.close()

    (d / "find_g.cfg").write_text(LW_CFG.format(d=d))
    r = run_tool("find_g_points", d / "find_g.cfg", f"output={d}/gpoints.nc", cwd="/")
    assert r.returncode == 0, r.stderr + r.stdout
    assert "*** FINDING G POINTS FOR H2O" in r.stdout and "*** COMPUTING SPECTRAL OVERLAP OF GASES" in r.stdout
    exp = pipeline.find_g_points(ctx, [dict(name="h2o", input=d / "h2o.nc", reordering_input=d / "order_h2o.nc",
                                            background=[dict(path=d / "co2.nc")]),
                                       dict(name="co2", input=d / "co2.nc", reordering_input=d / "order_co2.nc",
                                            background=[dict(path=d / "h2o.nc")], min_g_points=[2, 1])],
                                 b1, b2, 0.08, tolerance_tolerance=0.02, max_iterations=40)
    f = _nc(d / "gpoints.nc")
    v = f.variables
    assert int(v["n_gases"][...]) == 2 and f.constituent_id == b"h2o co2"
    assert np.array_equal(v["g_point"][:], exp["g_point"]) and np.array_equal(v["band_number"][:], exp["band_number"])
    for k, g in enumerate(("h2o", "co2")):
        e = exp["gases"][k]
        assert np.array_equal(v[g + "_n_g_points"][:], e["n_g_points"])
        assert np.array_equal(v[g + "_rank1"][:], e["rank1"]) and np.array_equal(v[g + "_rank2"][:], e["rank2"])
        assert np.array_equal(v[g + "_band_number"][:], e["band_number"])
        assert np.array_equal(v[g + "_g_min"][:], e["g_min"][:exp["ng"]]) and np.array_equal(v[g + "_g_max"][:], e["g_max"][:exp["ng"]])
        assert np.array_equal(v[g + "_g_point"][:], e["g_point"])
        assert np.array_equal(v[g + "_error"][:], np.asarray(e["error"], dtype=np.float32))
        assert np.array_equal(v[g + "_sorting_variable"][:], np.asarray(e["sorting_variable"], dtype=np.float32))
    assert v["co2_n_g_points"][0] >= 2
    assert b"h2o.background_input=co2.nc" in f.config
    f.close()
    back = ncio.read_g_points(d / "gpoints.nc")          # what create_look_up_table reads next
    assert np.array_equal(back["g_point"], exp["g_point"])
    # bands side by side (the default for a longwave run) against the reference's one-band-at-a-time order (sequential_bands=1):
    # same g points; the interval errors agree to rounding (the chunking of the sums follows the batch).  Two side-by-side
    # runs are bit-identical.
    r = run_tool("find_g_points", d / "find_g.cfg", f"output={d}/gpoints_seq.nc", "sequential_bands=1", cwd="/")
    assert r.returncode == 0, r.stderr + r.stdout
    r = run_tool("find_g_points", d / "find_g.cfg", f"output={d}/gpoints_again.nc", cwd="/")
    assert r.returncode == 0, r.stderr + r.stdout
    a, s, again = ncio.read_g_points(d / "gpoints.nc"), ncio.read_g_points(d / "gpoints_seq.nc"), ncio.read_g_points(d / "gpoints_again.nc")
    assert np.array_equal(a["g_point"], s["g_point"]) and np.array_equal(a["g_point"], again["g_point"])
    fa, fs, fg = _nc(d / "gpoints.nc"), _nc(d / "gpoints_seq.nc"), _nc(d / "gpoints_again.nc")
    for g in ("h2o", "co2"):
        for k in ("_rank1", "_rank2", "_n_g_points", "_g_min", "_g_max"):
            assert np.array_equal(fa.variables[g + k][:], fs.variables[g + k][:]), (g, k)
        assert np.allclose(fa.variables[g + "_error"][:], fs.variables[g + "_error"][:], rtol=1e-6)
        assert np.array_equal(fa.variables[g + "_error"][:], fg.variables[g + "_error"][:])
    fa.close(); fs.close(); fg.close()


def _run_ranks(world, *args, cwd=None):
    """`world` find_g_points processes as a launcher (torchrun --no-python, one per GPU) would start them; here they share GPU 0."""
    exe = os.path.join(BIN, "find_g_points")
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), ECCKD_DEVICE="0")
        procs.append(subprocess.Popen([exe, *[str(a) for a in args]], cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err + out
    return [o for o, _ in outs]


def test_orderly_teardown_gives_the_same_files(ctx, tmp_path, monkeypatch):
    """The tools leave their device resources to the process exit (done() in cli/tool.hpp: std::_Exit once the files are closed);
    ECCKD_NO_FAST_EXIT keeps the orderly teardown: the same output, byte for byte, and the same log either way."""
    d = tmp_path
    _make_lw_files(d, nwav=6000)
    args = ("input=h2o.nc", "wavenumber1=0 1300", "wavenumber2=1300 3260")
    fast = run_tool("reorder_spectrum", *args, "output=order_fast.nc", cwd=d)
    monkeypatch.setenv("ECCKD_NO_FAST_EXIT", "1")
    slow = run_tool("reorder_spectrum", *args, "output=order_slow.nc", cwd=d)
    assert fast.returncode == 0 and slow.returncode == 0, fast.stderr + slow.stderr
    assert fast.stdout.replace("order_fast", "X") == slow.stdout.replace("order_slow", "X")
    a, b = (d / "order_fast.nc").read_bytes(), (d / "order_slow.nc").read_bytes()
    assert len(a) == len(b)
    # the history attribute carries the command line (the output's name): compare the variables
    fa, fb = _nc(d / "order_fast.nc"), _nc(d / "order_slow.nc")
    for k in fa.variables:
        assert np.array_equal(fa.variables[k][...], fb.variables[k][...]), k
    fa.close(); fb.close()


@pytest.mark.parametrize("world", [2, 3])
def test_find_g_points_several_processes(ctx, tmp_path, world):
    """The (gas, band) searches dealt to `world` processes: the g-points file is the one a single process writes, whatever the
    share of each - including the process that searches only the second gas and rebuilds the first gas's Planck matrix (with
    that gas's sub-band re-ranking), and the bands whose rank the search changed (g_split; only in the second band: the first band's
    lower bound, a FLOAT in the ordering file, lies above the first wavenumber of this grid, which ends the reference's sub-band
    set-up - and this one - with "Failed to account for all wavenumbers in split")."""
    d = tmp_path
    _make_lw_files(d)
    os.symlink(d / "h2o.nc", d / "co2_bg_is_h2o.nc")
    for g in ("h2o", "co2"):
        r = run_tool("reorder_spectrum", f"input={d}/{g}.nc", f"output={d}/order_{g}.nc", "wavenumber1=0 1300", "wavenumber2=1300 3260")
        assert r.returncode == 0, r.stderr
    cfg = LW_CFG.format(d=d).replace("  background_input \"co2.nc\"\n",
                                     "  background_input \"co2.nc\"\n  g_split 0 0.7\n  subband_wavenumber_boundary 2000 2600\n")
    assert "g_split" in cfg
    (d / "find_g.cfg").write_text(cfg)
    r = run_tool("find_g_points", d / "find_g.cfg", f"output={d}/one.nc", cwd="/")
    assert r.returncode == 0, r.stderr + r.stdout
    outs = _run_ranks(world, d / "find_g.cfg", f"output={d}/many.nc", "part_timeout=300", cwd="/")
    _same_files(d / "one.nc", d / "many.nc")
    assert not [f for f in os.listdir(d) if ".part" in f]                   # the parts are collected and removed
    assert "Final cost" in outs[0] and "COMPUTING SPECTRAL OVERLAP" in outs[0]
    assert all("COMPUTING SPECTRAL OVERLAP" not in o for o in outs[1:])
    f = _nc(d / "one.nc")
    total = float(np.sum(f.variables["h2o_error"][:].astype(np.float64)) + np.sum(f.variables["co2_error"][:].astype(np.float64)))
    f.close()
    final = float(outs[0].split("Final cost")[1].split(":")[1].split()[0])
    assert final == pytest.approx(total, rel=1e-6)                           # the file holds FLOAT errors
    if world == 3:
        assert "(searched by other processes)" in outs[1]                    # process 1 searches co2 only


@pytest.mark.parametrize("variant", ["plain", "g_split", "sequential_bands"])
def test_find_g_points_gases_side_by_side_write_the_file_of_gas_after_gas(ctx, tmp_path, variant):
    """bin/find_g_points searches a gas from the moment it is prepared while it reads and prepares the next one
    (ecckd_find_g_gases_begin / _add / _wait, a HIP stream per gas); gases_side_by_side=1 is the reference's gas loop
    (find_g_points.cpp:655).  Both must write the same g-points file - also with sub-bands (the re-ranking of a band happens
    before the gas is prepared) and with sequential_bands, which implies gas after gas."""
    d = tmp_path
    _make_lw_files(d)
    os.symlink(d / "h2o.nc", d / "co2_bg_is_h2o.nc")
    for g in ("h2o", "co2"):
        r = run_tool("reorder_spectrum", f"input={d}/{g}.nc", f"output={d}/order_{g}.nc", "wavenumber1=0 1300", "wavenumber2=1300 3260")
        assert r.returncode == 0, r.stderr
    cfg = LW_CFG.format(d=d)
    if variant == "g_split":
        cfg = cfg.replace("  background_input \"co2.nc\"\n", "  background_input \"co2.nc\"\n  g_split 0 0.7\n  subband_wavenumber_boundary 2000 2600\n")
        assert "g_split" in cfg
    (d / "find_g.cfg").write_text(cfg)
    extra = ["sequential_bands=1"] if variant == "sequential_bands" else []
    r = run_tool("find_g_points", d / "find_g.cfg", f"output={d}/side.nc", *extra, cwd="/")
    assert r.returncode == 0, r.stderr + r.stdout
    assert "*** G POINTS OF H2O" in r.stdout and "*** G POINTS OF CO2" in r.stdout
    r = run_tool("find_g_points", d / "find_g.cfg", f"output={d}/after.nc", "gases_side_by_side=1", cwd="/")
    assert r.returncode == 0, r.stderr + r.stdout
    _same_files(d / "side.nc", d / "after.nc", skip=("history", "config"))


def test_find_g_points_refuses_stale_parts_and_sees_failed_peers(ctx, tmp_path):
    """a part file carries the identity of its run (configuration text, launcher rendezvous, WORLD_SIZE, numbers of
    gases / bands / wavenumbers); process 0 refuses one that is not its own instead of building the g-points file from another
    run's data, a process removes its stale part before anything that can fail, and a process that ends with an error leaves a
    marker that process 0 acts on at once instead of waiting for the time-out."""
    d = tmp_path
    _make_lw_files(d)
    os.symlink(d / "h2o.nc", d / "co2_bg_is_h2o.nc")
    for g in ("h2o", "co2"):
        r = run_tool("reorder_spectrum", f"input={d}/{g}.nc", f"output={d}/order_{g}.nc", "wavenumber1=0 1300", "wavenumber2=1300 3260")
        assert r.returncode == 0, r.stderr
    (d / "find_g.cfg").write_text(LW_CFG.format(d=d))
    exe = os.path.join(BIN, "find_g_points")

    def start(rank, *extra, env=None):
        e = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), ECCKD_DEVICE="0", MASTER_PORT="29999")
        e.update(env or {})
        return subprocess.Popen([exe, str(d / "find_g.cfg"), f"output={d}/many.nc", "part_timeout=120", *extra], cwd="/", env=e,
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)

    # (1) a part left by ANOTHER run (here: another tolerance on the command line = another configuration text): refused
    p1 = start(1, "heating_rate_tolerance=0.5")
    out, err = p1.communicate(timeout=600)
    assert p1.returncode == 0, err + out
    assert (d / "many.nc.part1").exists()
    p0 = start(0)
    out, err = p0.communicate(timeout=600)
    assert p0.returncode == 148 and "belongs to another run" in err, err + out
    assert not (d / "many.nc").exists()
    # (2) the stale part is gone as soon as the process of that rank starts, even when it then fails at device start-up; its
    # marker ends process 0's wait at once (well inside the 120 s time-out)
    import time
    t0 = time.perf_counter()
    p0 = start(0)
    p1 = start(1, env={"ECCKD_DEVICE": "63"})
    out1, err1 = p1.communicate(timeout=600)
    out0, err0 = p0.communicate(timeout=600)
    assert p1.returncode != 0
    assert p0.returncode == 148 and "ended with exit code" in err0, err0 + out0
    assert time.perf_counter() - t0 < 60.0
    assert not (d / "many.nc.part1").exists() and not (d / "many.nc").exists()
    # (3) and the run as it should be: same rendezvous on both, parts collected, markers gone
    p0, p1 = start(0), start(1)
    (out0, err0), (out1, err1) = p0.communicate(timeout=600), p1.communicate(timeout=600)
    assert p0.returncode == 0 and p1.returncode == 0, err0 + err1
    assert (d / "many.nc").exists() and not [f for f in os.listdir(d) if ".part" in f]


def test_find_g_points_sw(ctx, tmp_path):
    from ecckd_amd import pipeline
    d = tmp_path
    nwav, lo, hi = 16000, 250.0, 50000.0
    p = syn.pressure_grid(NLAY)
    t_hl = syn.temperature_profile(p)
    wn, dwn = syn.wavenumber_grid(nwav, lo, hi)
    ssi = syn.solar_spectral_irradiance(wn, dwn)
    w = netcdf_file(str(d / "ssi.nc"), "w", version=2)
    w.createDimension("wavenumber", nwav)
    w.createVariable("solar_spectral_irradiance", "d", ("wavenumber",))[:] = ssi
    w.close()
    for g, (seed, scale, vmr) in {"h2o": (61, 5.0, 5e-3), "o3": (67, 1.5, 1e-6)}.items():
        od = syn.optical_depth(np, p, wn, syn.SEED_BASE + seed, nlines=40, column_scale=scale, dtype="float32", lo=lo, hi=hi)
        _write_spectrum(d / f"{g}.nc", g, p, t_hl, wn, od, vmr)
        r = run_tool("reorder_spectrum", f"input={g}.nc", f"output=order_{g}.nc", "ssi=ssi.nc", "wavenumber1=250 10000",
                     "wavenumber2=10000 50000", cwd=d)
        assert r.returncode == 0, r.stderr
        assert "Assuming shortwave spectral region" in r.stdout
    cfg = ("ssi ssi.nc\naveraging_method transmission\nheating_rate_tolerance 0.03\nmax_iterations 40\ngases h2o o3\n"
           "\\begin h2o\n input h2o.nc\n reordering_input order_h2o.nc\n background_input o3.nc\n\\end h2o\n"
           "\\begin o3\n input o3.nc\n reordering_input order_o3.nc\n background_input h2o.nc\n max_scaling 3.0\n\\end o3\n")
    (d / "sw.cfg").write_text(cfg)
    # the scripts override the averaging method on the command line (test/find_g_points_sw.sh:26)
    r = run_tool("find_g_points", "averaging_method=total-transmission", "sw.cfg", "output=gpoints_sw.nc", cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    f = _nc(d / "order_h2o.nc")
    b1, b2 = f.variables["wavenumber1_band"][:].astype(np.float64), f.variables["wavenumber2_band"][:].astype(np.float64)
    f.close()
    assert b1[0] == np.float32(wn[0]) and b2[-1] == np.float32(wn[-1])       # clamped to the data (reorder_spectrum.cpp:268-273)
    specs = [dict(name="h2o", input=d / "h2o.nc", reordering_input=d / "order_h2o.nc", background=[dict(path=d / "o3.nc")]),
             dict(name="o3", input=d / "o3.nc", reordering_input=d / "order_o3.nc", background=[dict(path=d / "h2o.nc")],
                  max_scaling=3.0)]
    exp = pipeline.find_g_points(ctx, specs, b1, b2, 0.03, averaging_method="total-transmission", max_iterations=40, ssi=ssi)
    f = _nc(d / "gpoints_sw.nc")
    v = f.variables
    assert np.array_equal(v["g_point"][:], exp["g_point"])
    for k, g in enumerate(("h2o", "o3")):
        assert np.array_equal(v[g + "_rank1"][:], exp["gases"][k]["rank1"]) and np.array_equal(v[g + "_rank2"][:], exp["gases"][k]["rank2"])
    solar = np.array([ssi[exp["g_point"] == ig].sum() for ig in range(exp["ng"])])
    assert np.allclose(v["solar_irradiance"][:], solar, rtol=1e-6)
    assert b"shortwave" in f.title
    f.close()
    # two processes (the four (gas, band) searches dealt two and two): the same file
    _run_ranks(2, "averaging_method=total-transmission", "sw.cfg", "output=gpoints_sw_2.nc", "part_timeout=300", cwd=d)
    _same_files(d / "gpoints_sw.nc", d / "gpoints_sw_2.nc")
    # the band loop one band after the other (the reference's order) finds the same g points
    r = run_tool("find_g_points", "averaging_method=total-transmission", "sw.cfg", "output=gpoints_sw_seq.nc", "sequential_bands=1", cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    fa, fs = _nc(d / "gpoints_sw.nc"), _nc(d / "gpoints_sw_seq.nc")
    assert np.array_equal(fa.variables["g_point"][:], fs.variables["g_point"][:])
    for g in ("h2o", "o3"):
        for k in ("_rank1", "_rank2", "_n_g_points"):
            assert np.array_equal(fa.variables[g + k][:], fs.variables[g + k][:]), (g, k)
        assert np.allclose(fa.variables[g + "_error"][:], fs.variables[g + "_error"][:], rtol=1e-6)
    fa.close(); fs.close()


def test_exit_codes(tmp_path):
    """THROW(code) -> process exit code (Logging.h:115-117, EsaExitCodes.h): the scripts run under `set -e`."""
    r = run_tool("reorder_spectrum", "output=x.nc", cwd=tmp_path)
    assert r.returncode == 147 and "\"input\" file not specified" in r.stderr          # PARAMETER_ERROR
    r = run_tool("reorder_spectrum", "missing.cfg", cwd=tmp_path)
    assert r.returncode == 139                                                          # CANNOT_OPEN_MANDATORY_FILE
    r = run_tool("reorder_spectrum", "input=nothing.nc", "output=x.nc", cwd=tmp_path)
    assert r.returncode == 139 and "nothing.nc" in r.stderr
    (tmp_path / "bad.cfg").write_text("output g.nc\ngases h2o\n")
    r = run_tool("find_g_points", "bad.cfg", cwd=tmp_path)
    assert r.returncode == 147 and "heating_rate_tolerance not defined" in r.stderr
    # Rayleigh scattering as an OPTIMISED pseudo-gas (rayleigh_prior_error > 0; commented out in the shipped scripts,
    # test/optimize_lut_sw.sh) is a documented refusal, not a silent difference: the reference adds that prior to the cost but
    # not to the gradient (ckd_model.cpp:869-874) and leaves the bounds of those state elements unset (:136-137,
    # solve_adept.cpp:346-347) - there is no well-defined minimisation to reproduce ("Known differences")
    r = run_tool("optimize_lut", "input=raw.nc", "output=opt.nc", "rayleigh_prior_error=0.5", cwd=tmp_path)
    assert r.returncode == 147 and "rayleigh_prior_error > 0" in r.stderr
    r = run_tool("optimize_lut", "input=raw.nc", "output=opt.nc", "rayleigh_prior_error=0", cwd=tmp_path)
    assert r.returncode != 147 or "rayleigh_prior_error" not in r.stderr                # 0 (the scripts' value) is accepted


def _write_columns(path, gas, p1, t, wn, seed, scale, vmr, lo=0.0, hi=3260.0):
    ncol, nlay = t.shape[0], p1.size - 1
    w = netcdf_file(str(path), "w", version=2)
    for d, n in (("column", ncol), ("half_level", nlay + 1), ("level", nlay), ("wavenumber", wn.size)):
        w.createDimension(d, n)
    od = np.stack([syn.optical_depth(np, p1, wn, syn.SEED_BASE + seed, nlines=30, column_scale=scale * (1 + 0.1 * c), dtype="float32",
                                     lo=lo, hi=hi) for c in range(ncol)])
    w.createVariable("pressure_hl", "d", ("column", "half_level"))[:] = np.tile(p1, (ncol, 1))
    w.createVariable("temperature_hl", "d", ("column", "half_level"))[:] = t
    w.createVariable("wavenumber", "d", ("wavenumber",))[:] = wn
    w.createVariable("mole_fraction_fl", "d", ("column", "level"))[:] = np.full((ncol, nlay), vmr)
    w.createVariable("optical_depth", "f", ("column", "level", "wavenumber"))[:] = od
    w.createVariable("reference_surface_mole_fraction", "d", ())[...] = vmr
    w.constituent_id = gas
    w.close()


def _write_columns_from(path, gas, p1, temps, wn, od, vmr):
    """A spectral file with one column per temperature profile, all sharing the optical depth `od` (nlay, nwav)."""
    ncol, nlay = len(temps), p1.size - 1
    w = netcdf_file(str(path), "w", version=2)
    for d, n in (("column", ncol), ("half_level", nlay + 1), ("level", nlay), ("wavenumber", wn.size)):
        w.createDimension(d, n)
    w.createVariable("pressure_hl", "d", ("column", "half_level"))[:] = np.tile(p1, (ncol, 1))
    w.createVariable("temperature_hl", "d", ("column", "half_level"))[:] = np.stack(temps)
    w.createVariable("wavenumber", "d", ("wavenumber",))[:] = wn
    w.createVariable("mole_fraction_fl", "d", ("column", "level"))[:] = np.full((ncol, nlay), vmr)
    w.createVariable("optical_depth", "f", ("column", "level", "wavenumber"))[:] = np.tile(np.asarray(od, dtype=np.float32)[None], (ncol, 1, 1))
    w.createVariable("reference_surface_mole_fraction", "d", ())[...] = vmr
    w.constituent_id = gas
    w.close()


def _same_files(a, b, rtol=0.0, skip=()):
    fa, fb = _nc(a), _nc(b)
    assert set(fa.variables) == set(fb.variables), set(fa.variables) ^ set(fb.variables)
    for k, va in fa.variables.items():
        vb = fb.variables[k]
        assert va.typecode() == vb.typecode() and va.shape == vb.shape and va.dimensions == vb.dimensions, k
        if k in skip:
            continue
        if rtol:
            assert np.allclose(va[...], vb[...], rtol=rtol, atol=0.0), k
        else:
            assert np.array_equal(va[...], vb[...]), k
    ida, idb = fa.constituent_id, fb.constituent_id
    fa.close(); fb.close()
    assert ida == idb


@pytest.mark.parametrize("is_sw", [False, True])
def test_create_look_up_table(ctx, tmp_path, is_sw):
    from ecckd_amd import ncio, pipeline
    d = tmp_path
    rs = np.random.RandomState(7)
    nlay, nwav, ncol = 12, 6000, 3
    lo, hi = (250.0, 50000.0) if is_sw else (0.0, 3260.0)
    p1 = syn.pressure_grid(nlay)
    wn, dwn = syn.wavenumber_grid(nwav, lo, hi)
    t = np.stack([syn.temperature_profile(p1) + 15.0 * (c - 1) for c in range(ncol)])
    for name, seed, scale, vmr in (("o2", 1, 0.5, 0.209), ("n2", 2, 0.2, 0.781), ("co2", 3, 8.0, 4e-4), ("ch4", 4, 2.0, 1.8e-6),
                                   ("h2o_a", 5, 3.0, 1e-3), ("h2o_b", 6, 30.0, 1e-2)):
        _write_columns(d / f"{name}.nc", name.split("_")[0], p1, t, wn, seed, scale, vmr, lo, hi)
    g_point = rs.randint(0, 7, nwav).astype(np.int32)
    g_point[g_point == 5] = 6                                   # g point 5 occupies none of the spectrum
    band_number = np.array([0, 0, 0, 1, 1, 1, 1])
    b1, b2 = ([lo, 10000.0], [10000.0, hi]) if is_sw else ([0.0, 1300.0], [1300.0, 3260.0])
    ssi = solar = None
    if is_sw:
        ssi = syn.solar_spectral_irradiance(wn, dwn)
        solar = np.array([ssi[g_point == g].sum() for g in range(7)])
        w = netcdf_file(str(d / "ssi.nc"), "w", version=2)
        w.createDimension("wavenumber", nwav)
        w.createVariable("solar_spectral_irradiance", "d", ("wavenumber",))[:] = ssi
        w.createVariable("wavenumber", "d", ("wavenumber",))[:] = wn
        w.createVariable("total_solar_irradiance", "d", ())[...] = 1361.0
        w.close()
    ncio.write_g_points(d / "gpoints.nc", b1, b2, band_number, [], wn, g_point, solar_irradiance=solar, config_str="tolerance=0.1",
                        history="earlier: find_g_points")
    # a concentration file for the well-mixed pair (read_merged_spectrum.cpp:47-61, :104-116): other pressures, two profiles
    w = netcdf_file(str(d / "conc.nc"), "w", version=2)
    w.createDimension("column", 2); w.createDimension("level", 5)
    pc = np.geomspace(50.0, 9.0e4, 5)
    w.createVariable("pressure_fl", "d", ("column", "level"))[:] = np.stack([pc, pc])
    w.createVariable("o2_mole_fraction_fl", "d", ("column", "level"))[:] = np.stack([np.full(5, 0.3), np.linspace(0.18, 0.22, 5)])
    w.createVariable("n2_mole_fraction_fl", "d", ("column", "level"))[:] = np.stack([np.full(5, 0.5), np.linspace(0.70, 0.80, 5)])
    w.close()
    cfg = ("input gpoints.nc\noutput raw.nc\ngases composite co2 ch4 h2o\n"
           "\\begin composite\n conc_dependence none\n input \"o2.nc n2.nc\"\n conc_input conc.nc\n iprofile 1\n\\end composite\n"
           "\\begin co2\n conc_dependence linear\n input co2.nc\n\\end co2\n"
           "\\begin ch4\n conc_dependence relative-linear\n input ch4.nc\n reference_conc 1.8e-6\n\\end ch4\n"
           "\\begin h2o\n conc_dependence lut\n input \"h2o_a.nc\nh2o_b.nc\"\n\\end h2o\n")
    (d / "lut.cfg").write_text(cfg)
    # shortwave: also split the base g point of the upper band at 20 000 cm-1 (create_lut_sw.sh:187 does this for one band structure)
    extra = ["ssi=ssi.nc", "base_wavenumber_boundary=20000"] if is_sw else []
    r = run_tool("create_look_up_table", "lut.cfg", *extra, cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "occupies none of the spectrum" in r.stderr

    gases = [dict(name="composite", conc="none", inputs=[dict(path=d / "o2.nc"), dict(path=d / "n2.nc")],
                  conc_input=dict(path=d / "conc.nc", iprofile=1)),
             dict(name="co2", conc="linear", inputs=[d / "co2.nc"]),
             dict(name="ch4", conc="relative-linear", inputs=[d / "ch4.nc"], reference_conc=1.8e-6),
             dict(name="h2o", conc="lut", inputs=[d / "h2o_a.nc", d / "h2o_b.nc"])]
    model = pipeline.create_look_up_table(ctx, g_point, band_number, b1, b2, gases, ssi=ssi, solar_irradiance=solar,
                                          base_wavenumber_boundary=[20000.0] if is_sw else None, ssi_wavenumber=wn if is_sw else None)
    ng_out = 7 if is_sw else 6                                   # 7 in the file - 1 empty (+ 1 from the split)
    assert model["ng"] == ng_out and model["save_g_points"] and model["g_point_hr"].max() == ng_out - 1
    if is_sw:
        model["reference_total_solar_irradiance"] = 1361.0
        assert model["rayleigh_molar_scattering"].shape == (ng_out,) and np.all(model["rayleigh_molar_scattering"] > 0)
        # After the removal of the empty g point the "band numbers" are the OLD g indices [0, 1, 2, 3, 4, 6] (:142), so the
        # first g point "of band 1" is g point 1: it is the one divided at 20 000 cm-1 into 1 and 2, the rest move up by one.
        old1 = (g_point == 1)
        assert np.array_equal(model["g_point_hr"][old1], np.where(wn[old1] < 20000.0, 1, 2))
        assert np.array_equal(model["g_point_hr"][g_point == 3], np.full((g_point == 3).sum(), 4))
        assert np.array_equal(model["iband_per_g"], [0, 1, 1, 2, 3, 4, 6])
        assert np.allclose(model["solar_irradiance"][[1, 2]], [ssi[old1 & (wn < 20000.0)].sum(), ssi[old1 & (wn >= 20000.0)].sum()])
        # shorter wavelengths scatter more: the coefficient of a g point made of the upper band only is larger
    ncio.write_ckd_model(str(d / "py.nc"), model)
    _same_files(d / "raw.nc", d / "py.nc", skip=("rayleigh_molar_scattering_coeff",))
    if is_sw:
        fa, fb = _nc(d / "raw.nc"), _nc(d / "py.nc")
        assert np.allclose(fa.variables["rayleigh_molar_scattering_coeff"][:], fb.variables["rayleigh_molar_scattering_coeff"][:], rtol=1e-6)
        assert float(fa.variables["reference_total_solar_irradiance"][...]) == 1361.0
        fa.close(); fb.close()
    f = _nc(d / "raw.nc")
    assert f.history.startswith(b"earlier: find_g_points\n") and b"create_look_up_table lut.cfg" in f.history
    assert b"composite.conc_input=conc.nc" in f.config
    f.close()
    back = ncio.read_ckd_model(str(d / "raw.nc"))              # what optimize_lut / run_ckd read next
    assert [g["name"] for g in back["gases"]] == ["composite", "co2", "ch4", "h2o"] and back["ng"] == ng_out
    assert back["gases"][0]["composite_molecules"] == "o2 n2" and back["gases"][0]["composite_vmr"].shape == (2, nlay)
    assert np.array_equal(back["g_point_hr"], model["g_point_hr"]) and np.array_equal(back["wavenumber_hr"], wn)


def test_optimize_lut(ctx, tmp_path):
    """bin/optimize_lut against the host mirror on the same files: the same library calls in the same order, so the
    optimised coefficients in the two output files are identical."""
    from ecckd_amd import ncio, pipeline
    from test_pipeline_gpu import make_optimize_files
    d = tmp_path
    model, truth, scenes, paths, ib, names = make_optimize_files(ctx, d)
    cfg = ("input raw.nc\noutput opt.nc\ntraining_input \"lbl0.nc\nlbl1.nc\"\ngases composite h2o co2 ch4\nmodel_id cli-test\n"
           "flux_weight 0.2\nflux_profile_weight 0.05\nbroadband_weight 0.4\nprior_error 4.0\npressure_corr 0.95\n"
           "temperature_corr 0.95\nconc_corr 0.9\nmax_iterations 40\nconvergence_criterion 0.0\n")
    (d / "opt.cfg").write_text(cfg)
    r = run_tool("optimize_lut", "opt.cfg", cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "Optimizing coefficients of: composite h2o co2 ch4" in r.stdout and "Minimizer status" in r.stdout
    # the reference's progress line per iteration (solve_adept.cpp:295-299) and its three timed activities (:216-218)
    its = [l for l in r.stdout.splitlines() if l.startswith("Iteration ") and "cost function = " in l and "gradient norm = " in l]
    assert len(its) >= 40 and its[0].startswith("Iteration 0:")
    assert all(k in r.stderr for k in ("3 activities:", "s: minimizer", "s: a-priori", "s: radiative transfer", "s: Total"))
    raw = ncio.read_ckd_model(str(d / "raw.nc"), active_gases=["composite", "h2o", "co2", "ch4"])
    opt_model, res = pipeline.optimize_lut(ctx, raw, paths, max_iterations=40, flux_weight=0.2, flux_profile_weight=0.05,
                                           broadband_weight=0.4, prior_error=4.0, pressure_corr=0.95, temperature_corr=0.95,
                                           conc_corr=0.9, min_prior_error=-1.0, max_prior_error=-1.0)
    ncio.write_ckd_model(str(d / "py_opt.nc"), opt_model, model_id="cli-test")
    _same_files(d / "opt.nc", d / "py_opt.nc")
    f = _nc(d / "opt.nc")
    assert f.model_id == b"cli-test" and b"training_input={lbl0.nc lbl1.nc}" in f.config
    f.close()
    # and relative to a reference scene (optimize_lut.cpp:204-254): the tool accepts it and still lowers the misfit
    (d / "rel.cfg").write_text(cfg.replace("output opt.nc", "output opt_rel.nc").replace('"lbl0.nc\nlbl1.nc"', "lbl1.nc") + "relative_to lbl0.nc\n")
    r = run_tool("optimize_lut", "rel.cfg", cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    opt_rel, _ = pipeline.optimize_lut(ctx, raw, [paths[1]], relative_to=paths[0], max_iterations=40, flux_weight=0.2,
                                       flux_profile_weight=0.05, broadband_weight=0.4, prior_error=4.0, pressure_corr=0.95,
                                       temperature_corr=0.95, conc_corr=0.9, min_prior_error=-1.0, max_prior_error=-1.0)
    ncio.write_ckd_model(str(d / "py_rel.nc"), opt_rel, model_id="cli-test")
    _same_files(d / "opt_rel.nc", d / "py_rel.nc")
    # remove_min_max=1 as the shipped scripts pass it (test/optimize_lut_lw.sh, optimize_lut.cpp:243-244, :308-310): the
    # <gas>_molar_absorption_coeff_min / _max tables of the input are left out of the output, everything else is unchanged
    f = _nc(d / "opt.nc")
    assert "h2o_molar_absorption_coeff_min" in f.variables and "co2_molar_absorption_coeff_max" in f.variables
    f.close()
    r = run_tool("optimize_lut", "opt.cfg", "output=opt_nomm.nc", "remove_min_max=1", cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    f = _nc(d / "opt_nomm.nc")
    assert not [v for v in f.variables if v.endswith("_min") or v.endswith("_max")]
    f.close()
    stripped, _ = pipeline.optimize_lut(ctx, raw, paths, max_iterations=40, flux_weight=0.2, flux_profile_weight=0.05,
                                        broadband_weight=0.4, prior_error=4.0, pressure_corr=0.95, temperature_corr=0.95,
                                        conc_corr=0.9, min_prior_error=-1.0, max_prior_error=-1.0, remove_min_max=True)
    ncio.write_ckd_model(str(d / "py_nomm.nc"), stripped, model_id="cli-test")
    _same_files(d / "opt_nomm.nc", d / "py_nomm.nc")
    fa, fb = _nc(d / "opt.nc"), _nc(d / "opt_nomm.nc")
    for v in fb.variables:
        assert np.array_equal(fa.variables[v][...], fb.variables[v][...]), v
    fa.close(); fb.close()
    # exit codes
    r = run_tool("optimize_lut", "input=raw.nc", "output=x.nc", cwd=d)
    assert r.returncode == 147 and "training_input" in r.stderr


def test_optimize_lut_boundary_fluxes(ctx, tmp_path):
    """The scripts' longwave options (test/optimize_lut_lw.sh:55, :339): spectral_boundary_weight with a gpointfile -
    the high-resolution surface / TOA fluxes of the training files are summed per g point on the device."""
    import torch
    from ecckd_amd import api, ncio, pipeline
    from test_pipeline_gpu import make_optimize_files
    d = tmp_path
    model, truth, scenes, paths, ib, names = make_optimize_files(ctx, d, boundary=True)
    cfg = ("input raw.nc\noutput opt.nc\ntraining_input \"lbl0.nc lbl1.nc\"\nprior_error 4.0\nbroadband_weight 0.8\n"
           "flux_profile_weight 0.2\ntemperature_corr 0.95\npressure_corr 0.95\nconc_corr 0.95\nspectral_boundary_weight 0.1\n"
           "max_iterations 30\nconvergence_criterion 0.0\n")
    (d / "opt.cfg").write_text(cfg)
    r = run_tool("optimize_lut", "opt.cfg", "gpointfile=gpoints.nc", cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "Mapping high-resolution boundary fluxes to g-points" in r.stdout and "ALL GASES" in r.stdout
    gp = ncio.read_g_points(str(d / "gpoints.nc"))
    wn = gp["wavenumber"]
    dwn = np.empty_like(wn); dwn[1:-1] = 0.5 * (wn[2:] - wn[:-2]); dwn[0] = 0.5 * dwn[1]; dwn[-1] = 0.5 * dwn[-2]
    dev = ctx.device
    gmap = api.GPointMap(ctx, torch.as_tensor(gp["g_point"], device=dev), int(gp["g_point"].max()) + 1, torch.as_tensor(wn, device=dev),
                         torch.as_tensor(dwn, device=dev))
    raw = ncio.read_ckd_model(str(d / "raw.nc"))
    kw = dict(max_iterations=30, flux_weight=0.02, flux_profile_weight=0.2, broadband_weight=0.8, prior_error=4.0, pressure_corr=0.95,
              temperature_corr=0.95, conc_corr=0.95, min_prior_error=-1.0, max_prior_error=-1.0)
    with_b, _ = pipeline.optimize_lut(ctx, raw, paths, gmap=gmap, spectral_boundary_weight=0.1, **kw)
    without, _ = pipeline.optimize_lut(ctx, raw, paths, **kw)
    gmap.close()
    ncio.write_ckd_model(str(d / "py.nc"), with_b)
    _same_files(d / "opt.nc", d / "py.nc")
    assert any(not np.array_equal(a["molar_abs"], b["molar_abs"]) for a, b in zip(with_b["gases"], without["gases"]))   # the term acts
    # g points stored in the CKD file (CkdModel::save_g_points / read_g_points) replace the g-point file
    stored = dict(raw, save_g_points=True, wavenumber_hr=wn, g_point_hr=gp["g_point"])
    ncio.write_ckd_model(str(d / "raw_with_g.nc"), stored)
    r = run_tool("optimize_lut", "opt.cfg", "input=raw_with_g.nc", "output=opt3.nc", cwd=d)
    assert r.returncode == 0 and "Mapping high-resolution boundary fluxes to g-points" in r.stdout, r.stderr + r.stdout
    _same_files(d / "opt3.nc", d / "py.nc")                      # and the output does not carry them on (ckd_model.h:471)
    # without the g-point file the boundary fluxes are ignored with a warning, as in the reference (lbl_fluxes.cpp:300-305)
    r = run_tool("optimize_lut", "opt.cfg", "output=opt2.nc", cwd=d)
    assert r.returncode == 0 and "ignored because g-point file not provided" in r.stderr
    ncio.write_ckd_model(str(d / "py2.nc"), without)
    _same_files(d / "opt2.nc", d / "py2.nc")


def test_run_ckd(ctx, tmp_path):
    """bin/run_ckd on a CKDMIP-style concentration file against api.run_ckd (which tests/test_run_ckd_gpu.py ties to the
    oracle): every variable run_ckd.cpp writes, the gas list, a concentration scaling, write_od_only."""
    from ecckd_amd import api, ncio
    from test_pipeline_gpu import make_optimize_files
    d = tmp_path
    model, truth, scenes, paths, ib, names = make_optimize_files(ctx, d)
    sc = dict(scenes[0], gas_present=None)
    ncol, nhl = sc["pressure_hl"].shape
    w = netcdf_file(str(d / "conc.nc"), "w", version=2)
    for dim, n in (("column", ncol), ("half_level", nhl), ("level", nhl - 1)):
        w.createDimension(dim, n)
    w.createVariable("pressure_hl", "d", ("column", "half_level"))[:] = sc["pressure_hl"]
    w.createVariable("temperature_hl", "d", ("column", "half_level"))[:] = sc["temperature_hl"]
    for i, g in enumerate(model["gases"]):
        if g["conc"] != "none":                                   # the well-mixed composite has no concentration variable
            w.createVariable(names[i] + "_mole_fraction_fl", "d", ("column", "level"))[:] = sc["vmr_fl"][:, i, :]
    w.experiment = "synthetic profiles"
    w.close()
    back = ncio.read_ckd_model(str(d / "raw.nc"))
    r = run_tool("run_ckd", "ckd_model=raw.nc", "input=conc.nc", "output=out.nc", cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "assuming no concentration dependence" in r.stdout
    exp = api.run_ckd(ctx, back, sc)
    f = _nc(d / "out.nc")
    assert set(exp) == set(f.variables), set(exp) ^ set(f.variables)
    for k, v in exp.items():
        assert f.variables[k].typecode() == "f"
        assert np.allclose(f.variables[k][...], v, rtol=2e-6, atol=1e-30), k
    assert f.experiment == b"synthetic profiles" and b"run_ckd ckd_model=raw.nc" in f.history
    f.close()
    # gas list + scaling + write_od_only (:68-90, :270-307)
    ico2 = names.index("co2")
    r = run_tool("run_ckd", "ckd_model=raw.nc", "input=conc.nc", "output=out2.nc", "gases=composite co2", "co2_scaling=2", "write_od_only=1", cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    exp2 = api.run_ckd(ctx, back, sc, gases=["composite", "co2"], scalings={ico2: 2.0}, per_gas=False)
    f = _nc(d / "out2.nc")
    assert set(f.variables) == {"pressure_hl", "optical_depth", "planck_hl"}
    assert np.allclose(f.variables["optical_depth"][...], exp2["optical_depth"], rtol=2e-6, atol=1e-30)
    assert not np.allclose(exp2["optical_depth"], exp["optical_depth"])
    f.close()
    r = run_tool("run_ckd", "input=conc.nc", "output=x.nc", cwd=d)
    assert r.returncode == 147 and "ckd_model" in r.stderr


def test_scale_lut(ctx, tmp_path):
    """bin/scale_lut (step 4b of test/do_all_sw.sh) against api.scale_lut + GPointMap.sum_rows on the same files."""
    import torch
    import ckd_synth
    from ecckd_amd import api, ncio
    d = tmp_path
    model = ckd_synth.make_model_sw(seed=8)
    ng, names = model["ng"], [g["name"] for g in model["gases"]]
    ib, nband = model["iband_per_g"], model["nband"]
    model.update(wavenumber1=250.0 + np.arange(ng) * 50.0, wavenumber2=250.0 + np.arange(1, ng + 1) * 50.0, gpoint_fraction=np.eye(ng),
                 wavenumber1_band=np.array([250.0 + 50.0 * np.nonzero(ib == b)[0][0] for b in range(nband)]),
                 wavenumber2_band=np.array([250.0 + 50.0 * (np.nonzero(ib == b)[0][-1] + 1) for b in range(nband)]))
    ncio.write_ckd_model(str(d / "raw_sw.nc"), model)
    K = 4
    wn = 250.0 + (np.arange(ng * K) + 0.5) * (50.0 / K)
    g_point = np.repeat(np.arange(ng), K)
    ncio.write_g_points(str(d / "gpoints_sw.nc"), model["wavenumber1_band"], model["wavenumber2_band"], ib, [], wn, g_point,
                        solar_irradiance=model["solar_irradiance"])
    scene = ckd_synth.make_scenes(model, nscene=1, ncol=1, nlay=20)[0]
    p, T, vmr = scene["pressure_hl"][0], scene["temperature_hl"][0], scene["vmr_fl"][0]
    file_gases = [n for n in names if n not in ("composite", "ch4")]          # ch4 is not in the line-by-line file
    rs = np.random.RandomState(3)
    mu0 = 0.5
    flux = np.empty((p.size, ng))
    flux[0] = mu0 * rs.uniform(5.0, 30.0, ng)
    for l in range(p.size - 1):
        flux[l + 1] = flux[l] * np.exp(-rs.uniform(0.01, 0.4, ng) / mu0)
    flux[12:, 3] = 0.0                                                         # one beam extinguished half way down
    share = np.array([0.4, 0.1, 0.3, 0.2])
    hi = (flux[:, :, None] * share[None, None, :]).reshape(p.size, ng * K)
    w = netcdf_file(str(d / "lbl_sw.nc"), "w", version=2)
    for dim, n in (("column", 1), ("half_level", p.size), ("level", p.size - 1), ("gas", len(file_gases)), ("sza", 5), ("wavenumber", ng * K)):
        w.createDimension(dim, n)
    w.createVariable("mu0", "d", ("sza",))[:] = [mu0, 0.4, 0.3, 0.2, 0.1]
    w.createVariable("pressure_hl", "d", ("column", "half_level"))[:] = p[None]
    w.createVariable("temperature_hl", "d", ("column", "half_level"))[:] = T[None]
    w.createVariable("mole_fraction_fl", "d", ("column", "gas", "level"))[:] = np.stack([vmr[names.index(n)] for n in file_gases])[None]
    w.createVariable("spectral_flux_dn_direct_sw", "d", ("column", "half_level", "wavenumber"))[:] = hi[None]
    w.constituent_id = " ".join(n + ("-no-continuum" if n == "h2o" else "") for n in file_gases)
    w.close()
    r = run_tool("scale_lut", "input=raw_sw.nc", "output=scaled.nc", "gpointfile=gpoints_sw.nc", "lblfile=lbl_sw.nc", cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "Renaming h2o-no-continuum to h2o" in r.stdout

    back = ncio.read_ckd_model(str(d / "raw_sw.nc"))
    dev = ctx.device
    dwn = np.empty_like(wn); dwn[1:-1] = 0.5 * (wn[2:] - wn[:-2]); dwn[0] = 0.5 * dwn[1]; dwn[-1] = 0.5 * dwn[-2]
    gm = api.GPointMap(ctx, torch.as_tensor(g_point.astype(np.int32), device=dev), ng, torch.as_tensor(wn, device=dev), torch.as_tensor(dwn, device=dev))
    sums = gm.sum_rows(torch.as_tensor(hi, device=dev))
    gm.close()
    present = np.array([1 if (n == "composite" or n in file_gases) else 0 for n in names], dtype=np.int32)
    vm = np.where(present[:, None] == 1, vmr, 0.0)
    outs, scaling = api.scale_lut(ctx, back, sums, p, T, vm, present, mu0)
    assert np.all(scaling[11:, 3] == 1.0) and np.abs(scaling - 1.0).max() > 0.05
    scaled = dict(back, gases=[dict(g, molar_abs=o) for g, o in zip(back["gases"], outs)])
    ncio.write_ckd_model(str(d / "py_scaled.nc"), scaled)
    _same_files(d / "scaled.nc", d / "py_scaled.nc")
    r = run_tool("scale_lut", "input=raw_sw.nc", "output=x.nc", "lblfile=lbl_sw.nc", cwd=d)
    assert r.returncode == 147 and "gpointfile not provided" in r.stderr


@pytest.mark.parametrize("ext", ["nc", "h5"])
def test_do_all_lw_with_the_tools(ctx, oracle, tmp_path, ext):
    """test/do_all_lw.sh with the binaries only: reorder_spectrum -> find_g_points -> create_look_up_table -> optimize_lut ->
    run_ckd (raw and optimised definitions), every hand-over a NetCDF file, every option a config key.  The line-by-line
    training fluxes (external ckdmip_lw in the reference's scripts) come from the LBL stand-in.  Judged like the scripts' own
    evaluation: heating-rate RMS error against the line-by-line fluxes (plot/calc_hr_error.m).
    ext: the ordering and g-points files named *.nc (classic) or, as the scripts name them (test/reorder_spectrum_lw.sh,
    test/find_g_points_lw.sh), *.h5 - written as NetCDF-4 with deflated per-wavenumber variables and read back by the next tool."""
    from test_pipeline_gpu import hr_error_against_lbl, make_do_all_inputs
    if ext == "h5":
        import h5_fixture
        if not h5_fixture.available():
            pytest.skip("no HDF5 shared library with the deflate filter in this environment")
    d = tmp_path
    inp = make_do_all_inputs(ctx, d)
    ok = lambda r: (r.returncode == 0, r.stderr + r.stdout)
    for g in ("h2o", "co2"):
        r = run_tool("reorder_spectrum", f"input=present_{g}.nc", f"output=order_{g}.{ext}", "wavenumber1=0 1300", "wavenumber2=1300 3260", cwd=d)
        assert ok(r)[0], ok(r)[1]
    (d / "find_g.cfg").write_text(
        "heating_rate_tolerance 0.3\nmax_iterations 30\naveraging_method transmission\ngases h2o co2\n"
        f"\\begin h2o\n input present_h2o.nc\n reordering_input order_h2o.{ext}\n background_input present_co2.nc\n\\end h2o\n"
        f"\\begin co2\n input present_co2.nc\n reordering_input order_co2.{ext}\n background_input present_h2o.nc\n\\end co2\n")
    r = run_tool("find_g_points", "find_g.cfg", f"output=gpoints.{ext}", cwd=d)
    assert ok(r)[0], ok(r)[1]
    magic = b"\x89HDF\r\n\x1a\n" if ext == "h5" else b"CDF"
    for name in ("order_h2o", "order_co2", "gpoints"):
        assert open(d / f"{name}.{ext}", "rb").read(len(magic)) == magic
    (d / "lut.cfg").write_text(
        f"input gpoints.{ext}\noutput raw_ckd.nc\ngases h2o co2\n"
        "\\begin h2o\n conc_dependence lut\n input \"ideal_h2o.nc ideal_h2o_x4.nc\"\n\\end h2o\n"
        "\\begin co2\n conc_dependence linear\n input ideal_co2.nc\n\\end co2\n")
    r = run_tool("create_look_up_table", "lut.cfg", cwd=d)
    assert ok(r)[0], ok(r)[1]
    r = run_tool("optimize_lut", "input=raw_ckd.nc", "output=ckd.nc", "training_input=lbl.nc", "max_iterations=80", "flux_weight=0.2",
                 "flux_profile_weight=0.05", "broadband_weight=0.5", "prior_error=8.0", "convergence_criterion=0", "model_id=do_all_lw", cwd=d)
    assert ok(r)[0], ok(r)[1]
    # evaluation profiles for run_ckd: the training file's own columns
    ncol, nlay = inp["ncol"], inp["nlay"]
    w = netcdf_file(str(d / "eval.nc"), "w", version=2)
    for dim, n in (("column", ncol), ("half_level", nlay + 1), ("level", nlay)):
        w.createDimension(dim, n)
    w.createVariable("pressure_hl", "d", ("column", "half_level"))[:] = np.tile(inp["p1"], (ncol, 1))
    w.createVariable("temperature_hl", "d", ("column", "half_level"))[:] = inp["T"]
    for i, g in enumerate(("h2o", "co2")):
        w.createVariable(g + "_mole_fraction_fl", "d", ("column", "level"))[:] = inp["vmr"][:, i, :]
    w.close()
    errs = {}
    for tag, ckd in (("raw", "raw_ckd.nc"), ("optimised", "ckd.nc")):
        r = run_tool("run_ckd", f"ckd_model={ckd}", "input=eval.nc", f"output=fluxes_{tag}.nc", cwd=d)
        assert ok(r)[0], ok(r)[1]
        f = _nc(d / f"fluxes_{tag}.nc")
        errs[tag] = hr_error_against_lbl(oracle, inp, f.variables["flux_dn_lw"][...].astype(np.float64), f.variables["flux_up_lw"][...].astype(np.float64))
        f.close()
    print("heating-rate RMS error (K/day), tools only:", errs)
    assert np.isfinite(errs["raw"]) and errs["optimised"] < 0.9 * errs["raw"] and errs["optimised"] < 1.0
    f = _nc(d / "ckd.nc")
    assert f.model_id == b"do_all_lw" and f.history.count(b"\n") >= 2          # find_g_points, create_look_up_table, optimize_lut lines
    f.close()


def test_tools_read_netcdf4_inputs(ctx, tmp_path):
    """The scripts' file names end in .h5 (NetCDF-4).  Inputs in that format are read through the HDF5 library; since round 4
    outputs named *.h5 are WRITTEN as NetCDF-4 too (OutputDataFile.cpp:84-157; the per-wavenumber variables deflated as
    write_order.cpp:59-94 and find_g_points.cpp:1580-1587 ask), so this chain hands NetCDF-4 files from tool to tool:
    spectrum.h5 -> reorder_spectrum -> order.h5 -> find_g_points -> gpoints.h5."""
    import h5_fixture as h5
    from ecckd_amd import ncio
    if not h5.available():
        pytest.skip("no HDF5 shared library with the deflate filter in this environment")
    d = tmp_path
    nwav = 6000
    p = syn.pressure_grid(NLAY)
    t_hl = syn.temperature_profile(p)
    wn, _ = syn.wavenumber_grid(nwav)
    od = syn.optical_depth(np, p, wn, syn.SEED_BASE + 71, nlines=40, column_scale=20.0, dtype="float32")
    _write_spectrum(d / "h2o.nc", "h2o", p, t_hl, wn, od, 5e-3)
    h5.write(d / "h2o.h5", {
        "pressure_hl": (p[None], "f8", None, None), "temperature_hl": (t_hl[None], "f8", None, None),
        "wavenumber": (wn, "f8", (1000,), None), "mole_fraction_fl": (np.full((1, NLAY), 5e-3), "f8", None, None),
        "reference_surface_mole_fraction": (5e-3, "f8", None, None),
        "optical_depth": (od[None], "f4", (1, NLAY, 512), None)}, {"constituent_id": "h2o"})
    for ext in ("nc", "h5"):
        r = run_tool("reorder_spectrum", f"input=h2o.{ext}", f"output=order_{ext}.h5", "wavenumber1=0 1300", "wavenumber2=1300 3260", cwd=d)
        assert r.returncode == 0, r.stderr + r.stdout
    a, b = ncio.read_order(d / "order_nc.h5"), ncio.read_order(d / "order_h5.h5")
    for k in ("rank", "band_number", "sorting_variable", "wavenumber"):
        assert np.array_equal(a[k], b[k]), k
    assert b["molecule"] == "h2o"
    cfg = ("heating_rate_tolerance 0.1\nmax_iterations 30\naveraging_method transmission\ngases h2o\n"
           "\\begin h2o\n input h2o.h5\n reordering_input order_h5.h5\n\\end h2o\n")
    (d / "g.cfg").write_text(cfg)
    r = run_tool("find_g_points", "g.cfg", "output=gpoints.h5", cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    (d / "g2.cfg").write_text(cfg.replace("input h2o.h5", "input h2o.nc").replace("order_h5.h5", "order_nc.h5"))
    r = run_tool("find_g_points", "g2.cfg", "output=gpoints_nc.h5", cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    assert np.array_equal(ncio.read_g_points(d / "gpoints.h5")["g_point"], ncio.read_g_points(d / "gpoints_nc.h5")["g_point"])


def test_optimize_lut_shortwave(ctx, tmp_path):
    """bin/optimize_lut on a shortwave definition: zenith-angle selection (columns x [0, 2, 4] of the file's five angles),
    total solar irradiance and effective band albedo from the training file, upwelling masked above
    max_no_rayleigh_wavenumber, "h2o-no-continuum" mapped to h2o - identical to the host mirror on the same files."""
    import ckd_synth
    from ecckd_amd import api, ncio, pipeline
    d = tmp_path
    model = ckd_synth.make_model_sw(seed=8)
    ng, names = model["ng"], [g["name"] for g in model["gases"]]
    ib, nband = model["iband_per_g"], model["nband"]
    first = np.array([np.nonzero(ib == b)[0][0] for b in range(nband)])
    last = np.array([np.nonzero(ib == b)[0][-1] for b in range(nband)])
    model.update(wavenumber1=250.0 + np.arange(ng) * 400.0, wavenumber2=250.0 + np.arange(1, ng + 1) * 400.0, gpoint_fraction=np.eye(ng),
                 wavenumber1_band=250.0 + 400.0 * first, wavenumber2_band=250.0 + 400.0 * (last + 1))
    ncio.write_ckd_model(str(d / "raw_sw.nc"), model)
    raw = ncio.read_ckd_model(str(d / "raw_sw.nc"))
    truth = dict(raw, gases=[dict(g, molar_abs=g["molar_abs"] * np.exp(0.2 * np.random.RandomState(i).normal(size=g["molar_abs"].shape)))
                             for i, g in enumerate(raw["gases"])])
    ncol, nlay, mu0_file = 3, 12, np.array([0.9, 0.7, 0.5, 0.3, 0.1])
    base = ckd_synth.make_scenes(raw, nscene=1, ncol=ncol, nlay=nlay)[0]
    file_gases = [n for n in names if n != "composite"]
    dn_b = np.empty((ncol, 5, nlay + 1, nband))
    for k, mu in enumerate(mu0_file):
        sc = dict(base, gas_present=None, mu0=np.full(ncol, mu), tsi=1361.0)
        fl = api.run_ckd(ctx, truth, sc, per_gas=False)["spectral_flux_dn_direct_sw"]       # run_ckd itself uses mu0 = 0.5 (:358) ...
        dn_b[:, k] = np.stack([fl[..., ib == b].sum(-1) for b in range(nband)], axis=-1) * (mu / 0.5)   # ... any positive profile will do
    up_b = 0.1 * dn_b[:, :, -1:, :] * np.linspace(0.5, 1.0, nlay + 1)[None, None, :, None]
    w = netcdf_file(str(d / "lbl_sw.nc"), "w", version=2)
    for dim, n in (("column", ncol), ("mu0", 5), ("half_level", nlay + 1), ("level", nlay), ("gas", len(file_gases)), ("band", nband)):
        w.createDimension(dim, n)
    vmr = np.stack([base["vmr_fl"][:, names.index(n), :] for n in file_gases], axis=1)
    for name, dims, a in (("mu0", ("mu0",), mu0_file), ("pressure_hl", ("column", "half_level"), base["pressure_hl"]),
                          ("temperature_hl", ("column", "half_level"), base["temperature_hl"]),
                          ("mole_fraction_fl", ("column", "gas", "level"), vmr),
                          ("flux_dn_direct_sw", ("column", "mu0", "half_level"), dn_b.sum(-1)), ("flux_up_sw", ("column", "mu0", "half_level"), up_b.sum(-1)),
                          ("band_flux_dn_direct_sw", ("column", "mu0", "half_level", "band"), dn_b),
                          ("band_flux_up_sw", ("column", "mu0", "half_level", "band"), up_b),
                          ("band_wavenumber1_sw", ("band",), model["wavenumber1_band"]), ("band_wavenumber2_sw", ("band",), model["wavenumber2_band"])):
        w.createVariable(name, "d", dims)[:] = a
    w.constituent_id = " ".join(n + ("-no-continuum" if n == "h2o" else "") for n in file_gases)
    w.close()
    limit = float(model["wavenumber2_band"][nband // 2])                 # bands above this lose their upwelling
    r = run_tool("optimize_lut", "input=raw_sw.nc", "output=opt_sw.nc", "training_input=lbl_sw.nc", "prior_error=2.0", "broadband_weight=0.4",
                 "flux_weight=0.3", "flux_profile_weight=0.05", "temperature_corr=0.8", "pressure_corr=0.8", "conc_corr=0.8",
                 "max_iterations=25", "convergence_criterion=0", f"max_no_rayleigh_wavenumber={limit}", cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    opt, res = pipeline.optimize_lut(ctx, raw, [str(d / "lbl_sw.nc")], max_iterations=25, max_no_rayleigh_wavenumber=limit, prior_error=2.0,
                                     broadband_weight=0.4, flux_weight=0.3, flux_profile_weight=0.05, temperature_corr=0.8, pressure_corr=0.8,
                                     conc_corr=0.8, min_prior_error=-1.0, max_prior_error=-1.0)
    assert res["status"] in (0, 2, 3) and res["iterations"] > 3
    ncio.write_ckd_model(str(d / "py_sw.nc"), opt)
    _same_files(d / "opt_sw.nc", d / "py_sw.nc")
    assert any(not np.array_equal(a["molar_abs"], b["molar_abs"]) for a, b in zip(opt["gases"], raw["gases"]))


def test_do_all_sw_with_the_tools(ctx, tmp_path):
    """test/do_all_sw.sh with the binaries only: reorder_spectrum (ssi) -> find_g_points (total-transmission) ->
    create_look_up_table (solar weights, Rayleigh coefficient) -> scale_lut -> optimize_lut -> run_ckd.  The line-by-line
    direct-beam fluxes (external ckdmip_sw in the reference's scripts) come from the LBL stand-in.  Judged by the direct
    flux profile of run_ckd against the line-by-line one at the reference zenith angle."""
    import torch
    from ecckd_amd import api
    d = tmp_path
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=ctx.device)
    nlay, nwav, lo, hi = 16, 8000, 250.0, 50000.0
    p1 = syn.pressure_grid(nlay)
    wn, dwn = syn.wavenumber_grid(nwav, lo, hi)
    ssi = syn.solar_spectral_irradiance(wn, dwn)
    w = netcdf_file(str(d / "ssi.nc"), "w", version=2)
    w.createDimension("wavenumber", nwav)
    w.createVariable("solar_spectral_irradiance", "d", ("wavenumber",))[:] = ssi
    w.createVariable("wavenumber", "d", ("wavenumber",))[:] = wn
    w.createVariable("total_solar_irradiance", "d", ())[...] = ssi.sum()
    w.close()
    base = {"h2o": (syn.optical_depth(np, p1, wn, syn.SEED_BASE + 81, nlines=60, column_scale=3.0, dtype="float32", lo=lo, hi=hi), 5e-3),
            "o3": (syn.optical_depth(np, p1, wn, syn.SEED_BASE + 83, nlines=30, column_scale=0.8, dtype="float32", lo=lo, hi=hi), 1e-6)}
    t0 = syn.temperature_profile(p1)
    temps = [t0 - 20.0, t0, t0 + 20.0]
    for g, (od, vmr) in base.items():
        _write_columns_from(d / f"present_{g}.nc", g, p1, [t0], wn, od, vmr)
        _write_columns_from(d / f"ideal_{g}.nc", g, p1, temps, wn, od, vmr)
    _write_columns_from(d / "ideal_h2o_x4.nc", "h2o", p1, temps, wn, base["h2o"][0] * np.float32(4.0), base["h2o"][1] * 4.0)

    # line-by-line direct fluxes: 3 columns x 5 zenith angles, 2 bands; no Rayleigh scattering in this toy atmosphere
    b1, b2 = np.array([lo, 10000.0]), np.array([10000.0, hi])
    ncol, mu0s = 3, np.array([0.9, 0.7, 0.5, 0.3, 0.1])
    amount = {"h2o": np.array([0.7, 1.5, 3.0]), "o3": np.array([1.0, 2.0, 0.5])}
    begin = [int(np.nonzero((wn >= a) & (wn < b + (b == hi)))[0][0]) for a, b in zip(b1, b2)]
    end = [int(np.nonzero((wn >= a) & (wn < b + (b == hi)))[0][-1]) for a, b in zip(b1, b2)]
    dn = np.empty((ncol, 5, nlay + 1, 2))
    ods = [sum(base[g][0].astype(np.float64) * amount[g][c] for g in base) for c in range(ncol)]
    for c in range(ncol):
        for k, mu in enumerate(mu0s):
            dn[c, k] = api.lbl_band_fluxes_sw(ctx, mu, dev(ssi), dev(ods[c]), begin, end)[0].T
    hires = ssi[None, :] * 0.9 * np.exp(-np.concatenate([np.zeros((1, nwav)), np.cumsum(ods[0], axis=0)]) / 0.9)     # column 0, mu0 = 0.9
    w = netcdf_file(str(d / "lbl_sw.nc"), "w", version=2)
    for dim, n in (("column", ncol), ("mu0", 5), ("half_level", nlay + 1), ("level", nlay), ("gas", 2), ("band", 2), ("wavenumber", nwav)):
        w.createDimension(dim, n)
    vmr = np.stack([np.stack([np.full(nlay, base[g][1] * amount[g][c]) for g in ("h2o", "o3")]) for c in range(ncol)])
    for name, dims, a in (("mu0", ("mu0",), mu0s), ("pressure_hl", ("column", "half_level"), np.tile(p1, (ncol, 1))),
                          ("temperature_hl", ("column", "half_level"), np.tile(t0, (ncol, 1))), ("mole_fraction_fl", ("column", "gas", "level"), vmr),
                          ("flux_dn_direct_sw", ("column", "mu0", "half_level"), dn.sum(-1)), ("flux_up_sw", ("column", "mu0", "half_level"), 0.0 * dn.sum(-1)),
                          ("band_flux_dn_direct_sw", ("column", "mu0", "half_level", "band"), dn), ("band_flux_up_sw", ("column", "mu0", "half_level", "band"), 0.0 * dn),
                          ("band_wavenumber1_sw", ("band",), b1), ("band_wavenumber2_sw", ("band",), b2)):
        w.createVariable(name, "d", dims)[:] = a
    w.constituent_id = "h2o o3"
    w.close()
    # the line-by-line file scale_lut reads (one profile, per-wavenumber direct flux); a training file must not carry a
    # variable of that name with this shape - the tool says so instead of mis-reading it
    w = netcdf_file(str(d / "lbl_hires.nc"), "w", version=2)
    for dim, n in (("column", 1), ("mu0", 1), ("half_level", nlay + 1), ("level", nlay), ("gas", 2), ("wavenumber", nwav)):
        w.createDimension(dim, n)
    w.createVariable("mu0", "d", ("mu0",))[:] = [0.9]
    w.createVariable("pressure_hl", "d", ("column", "half_level"))[:] = p1[None]
    w.createVariable("temperature_hl", "d", ("column", "half_level"))[:] = t0[None]
    w.createVariable("mole_fraction_fl", "d", ("column", "gas", "level"))[:] = vmr[:1]
    w.createVariable("spectral_flux_dn_direct_sw", "d", ("column", "half_level", "wavenumber"))[:] = hires[None]
    w.constituent_id = "h2o o3"
    w.close()

    ok = lambda r: (r.returncode == 0, r.stderr + r.stdout)
    for g in base:
        r = run_tool("reorder_spectrum", f"input=present_{g}.nc", f"output=order_{g}.nc", "ssi=ssi.nc", "wavenumber1=250 10000", "wavenumber2=10000 50000", cwd=d)
        assert ok(r)[0], ok(r)[1]
    (d / "g.cfg").write_text(
        "ssi ssi.nc\nheating_rate_tolerance 0.06\nmax_iterations 30\naveraging_method total-transmission\ngases h2o o3\n"
        "\\begin h2o\n input present_h2o.nc\n reordering_input order_h2o.nc\n background_input present_o3.nc\n\\end h2o\n"
        "\\begin o3\n input present_o3.nc\n reordering_input order_o3.nc\n background_input present_h2o.nc\n\\end o3\n")
    r = run_tool("find_g_points", "g.cfg", "output=gpoints.nc", cwd=d)
    assert ok(r)[0], ok(r)[1]
    (d / "lut.cfg").write_text(
        "input gpoints.nc\noutput raw.nc\nssi ssi.nc\naveraging_method transmission-3\ngases h2o o3\n"      # create_lut_sw.sh:22
        "\\begin h2o\n conc_dependence lut\n input \"ideal_h2o.nc ideal_h2o_x4.nc\"\n\\end h2o\n"
        "\\begin o3\n conc_dependence linear\n input ideal_o3.nc\n\\end o3\n")
    r = run_tool("create_look_up_table", "lut.cfg", cwd=d)
    assert ok(r)[0], ok(r)[1]
    r = run_tool("scale_lut", "input=raw.nc", "output=scaled.nc", "gpointfile=gpoints.nc", "lblfile=lbl_hires.nc", cwd=d)
    assert ok(r)[0], ok(r)[1]
    r = run_tool("optimize_lut", "input=scaled.nc", "output=x.nc", "training_input=lbl_hires.nc", cwd=d)
    assert r.returncode == 147 and "solar zenith angles" in r.stderr, r.stderr            # not a training file
    r = run_tool("optimize_lut", "input=scaled.nc", "output=ckd.nc", "training_input=lbl_sw.nc", "prior_error=2.0", "broadband_weight=0.4", "flux_weight=0.3",
                 "flux_profile_weight=0.05", "max_iterations=80", "convergence_criterion=0", cwd=d)
    assert ok(r)[0], ok(r)[1]
    w = netcdf_file(str(d / "eval.nc"), "w", version=2)
    for dim, n in (("column", ncol), ("half_level", nlay + 1), ("level", nlay)):
        w.createDimension(dim, n)
    w.createVariable("pressure_hl", "d", ("column", "half_level"))[:] = np.tile(p1, (ncol, 1))
    w.createVariable("temperature_hl", "d", ("column", "half_level"))[:] = np.tile(t0, (ncol, 1))
    for i, g in enumerate(("h2o", "o3")):
        w.createVariable(g + "_mole_fraction_fl", "d", ("column", "level"))[:] = vmr[:, i, :]
    w.close()
    truth = dn[:, 2].sum(-1)                                    # mu0 = 0.5 = REFERENCE_COS_SZA, what run_ckd evaluates (:358)
    rms = {}
    for tag in ("raw", "scaled", "ckd"):
        r = run_tool("run_ckd", f"ckd_model={tag}.nc", "input=eval.nc", f"output=flux_{tag}.nc", f"tsi={ssi.sum()}", cwd=d)
        assert ok(r)[0], ok(r)[1]
        f = _nc(d / f"flux_{tag}.nc")
        got = f.variables["flux_dn_direct_sw"][...].astype(np.float64)
        assert np.all(f.variables["rayleigh_optical_depth"][...] > 0)
        f.close()
        assert np.allclose(got[:, 0], 0.5 * ssi.sum(), rtol=1e-5)                   # top of atmosphere: mu0 * tsi
        rms[tag] = float(np.sqrt(np.mean((got - truth) ** 2)))
    print("direct-flux RMS error (W m-2), tools only:", rms)
    assert np.isfinite(rms["raw"]) and rms["ckd"] < 0.9 * rms["raw"]


def test_ckdmip_lw_stand_in(ctx, tmp_path):
    """bin/ckdmip_lw, the stand-in for the external CKDMIP tool of the reference's scripts: --merge-only
    (test/merge_well_mixed_lw.sh:28-63) against the library's merge, the line-by-line band fluxes
    (test/run_lw_lbl_evaluation.sh:286-323) against ecckd_lbl_band_fluxes_lw and through LblFluxes::read's mirror, and the
    flux evaluation of a CKD model's optical depths (test/run_ckd_lw.sh:133-137) against run_ckd's own fluxes."""
    import torch
    from ecckd_amd import api, ncio
    from test_pipeline_gpu import make_do_all_inputs
    d = tmp_path
    inp = make_do_all_inputs(ctx, d)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=ctx.device)
    nlay, wn = inp["nlay"], inp["wn"]
    # ---- merge-only: h2o as it is + co2 scaled to 8e-4 at the surface (its file says 4e-4) ----
    r = run_tool("ckdmip_lw", "--merge-only", "ideal_h2o.nc", "--conc", "8e-4", "ideal_co2.nc", "--output", "merged.nc", cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    for col in (0, 2):
        m = ncio.read_spectrum(d / "merged.nc", col)
        h, c = ncio.read_spectrum(d / "ideal_h2o.nc", col), ncio.read_spectrum(d / "ideal_co2.nc", col)
        want = (h["optical_depth"] + 2.0 * c["optical_depth"]).astype(np.float32)
        assert np.allclose(m["optical_depth"], want, rtol=2e-7, atol=0) and np.array_equal(m["wavenumber_cm_1"], wn)
        assert np.allclose(m["temperature_hl"], h["temperature_hl"], rtol=1e-6) and m["molecule"] == "composite"
    # ---- line-by-line band fluxes of the three idealised columns, co2 at a constant 6e-4, h2o scaled by 0.5 ----
    (d / "lw.nam").write_text("&longwave_config\noptical_depth_name = \"optical_depth\",\nnspectralstride = 1,\nnangle = 0, ! classic\n"
                              "do_write_spectral_boundary_fluxes = false,\nband_wavenumber1(1:2) = 0, 1300,\n"
                              "band_wavenumber2(1:2) = 1300, 3260,\niverbose = 3\n/\n")
    r = run_tool("ckdmip_lw", "--config", "lw.nam", "--scenario", "test-1", "--scale", "0.5", "ideal_h2o.nc", "--const", "6e-4", "ideal_co2.nc",
                 "--output", "lbl_tool.nc", cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    f = _nc(d / "lbl_tool.nc")
    assert f.scenario == b"test-1" and f.constituent_id == b"h2o co2"
    begin = [int(np.nonzero((wn >= a) & (wn < b + (b == 3260.0)))[0][0]) for a, b in zip(*inp["bands"])]
    end = [int(np.nonzero((wn >= a) & (wn < b + (b == 3260.0)))[0][-1]) for a, b in zip(*inp["bands"])]
    dwn = ncio.read_spectrum(d / "ideal_h2o.nc")["d_wavenumber_cm_1"]
    for col in range(3):
        h, c = ncio.read_spectrum(d / "ideal_h2o.nc", col), ncio.read_spectrum(d / "ideal_co2.nc", col)
        od = 0.5 * h["optical_depth"] + (6e-4 / c["vmr_fl"])[:, None] * c["optical_depth"]
        dn, up = api.lbl_band_fluxes_lw(ctx, h["temperature_hl"], dev(wn), dev(dwn), dev(od), begin, end)
        assert np.allclose(f.variables["band_flux_dn_lw"][col], dn.T, rtol=3e-7, atol=1e-30)
        assert np.allclose(f.variables["band_flux_up_lw"][col], up.T, rtol=3e-7)
        assert np.allclose(f.variables["flux_up_lw"][col], up.sum(0), rtol=3e-7)
        assert np.allclose(f.variables["mole_fraction_fl"][col], np.stack([0.5 * h["vmr_fl"], np.full(nlay, 6e-4)]), rtol=2e-7)
    f.close()
    s = ncio.read_lbl_fluxes(d / "lbl_tool.nc", ["h2o", "co2"], ctx=ctx)           # what optimize_lut reads (lbl_fluxes.cpp:52-397)
    assert s["have_band_fluxes"] and s["flux_dn"].shape == (3, nlay + 1, 2) and np.array_equal(s["band_wavenumber2"], [1300.0, 3260.0])
    # ---- the same with do_write_spectral_boundary_fluxes: the spectral fluxes at the surface and the top (lbl_fluxes.cpp:301-325) ----
    (d / "lw_b.nam").write_text((d / "lw.nam").read_text().replace("do_write_spectral_boundary_fluxes = false", "do_write_spectral_boundary_fluxes = true"))
    r = run_tool("ckdmip_lw", "--config", "lw_b.nam", "--scale", "0.5", "ideal_h2o.nc", "--const", "6e-4", "ideal_co2.nc", "--output", "lbl_b.nc", cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    f, g = _nc(d / "lbl_tool.nc"), _nc(d / "lbl_b.nc")
    assert np.array_equal(g.variables["wavenumber"][...], wn)
    for name in ("band_flux_dn_lw", "band_flux_up_lw", "flux_dn_lw", "flux_up_lw"):
        assert np.array_equal(f.variables[name][...], g.variables[name][...]), name
    for col in range(3):
        h, c = ncio.read_spectrum(d / "ideal_h2o.nc", col), ncio.read_spectrum(d / "ideal_co2.nc", col)
        od = 0.5 * h["optical_depth"] + (6e-4 / c["vmr_fl"])[:, None] * c["optical_depth"]
        _, _, sdn, tup = api.lbl_band_fluxes_lw(ctx, h["temperature_hl"], dev(wn), dev(dwn), dev(od), begin, end, boundary=True)
        assert np.allclose(g.variables["spectral_flux_dn_surf_lw"][col], sdn.cpu().numpy(), rtol=3e-7, atol=1e-37)
        assert np.allclose(g.variables["spectral_flux_up_toa_lw"][col], tup.cpu().numpy(), rtol=3e-7, atol=1e-37)
        # summed over the spectrum they are the broadband fluxes at those levels
        assert np.isclose(g.variables["spectral_flux_dn_surf_lw"][col].astype(np.float64).sum(), g.variables["flux_dn_lw"][col][-1], rtol=1e-5)
        assert np.isclose(g.variables["spectral_flux_up_toa_lw"][col].astype(np.float64).sum(), g.variables["flux_up_lw"][col][0], rtol=1e-5)
    # ... and through LblFluxes::read's mirror with a g-points map: the boundary fluxes summed per g point (:308-325)
    ngp = 5
    gp = (np.arange(wn.size) * 7919 % ngp).astype(np.int32)
    gmap = api.GPointMap(ctx, dev(gp), ngp, dev(wn), dev(dwn))
    sb = ncio.read_lbl_fluxes(d / "lbl_b.nc", ["h2o", "co2"], gmap=gmap, ctx=ctx)
    for col in range(3):
        want_dn = np.array([g.variables["spectral_flux_dn_surf_lw"][col].astype(np.float64)[gp == k].sum() for k in range(ngp)])
        want_up = np.array([g.variables["spectral_flux_up_toa_lw"][col].astype(np.float64)[gp == k].sum() for k in range(ngp)])
        assert np.allclose(sb["spectral_flux_dn_surf"][col], want_dn, rtol=1e-10) and np.allclose(sb["spectral_flux_up_toa"][col], want_up, rtol=1e-10)
    gmap.close()
    f.close(); g.close()
    # ---- nangle = 4, what test/run_ckd_lw.sh:28,83 and test/copy_to_ckdmip_lw.sh:32 write into the namelist: four Gauss-Legendre
    # zenith angles per hemisphere instead of the two-stream diffusivity; same file layout, fluxes equal to the library's quadrature
    (d / "lw4.nam").write_text((d / "lw.nam").read_text().replace("nangle = 0, ! classic", "nangle = 4,"))
    r = run_tool("ckdmip_lw", "--config", "lw4.nam", "--scale", "0.5", "ideal_h2o.nc", "--const", "6e-4", "ideal_co2.nc", "--output", "lbl_4.nc", cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    f, g = _nc(d / "lbl_tool.nc"), _nc(d / "lbl_4.nc")
    for col in range(3):
        h, c = ncio.read_spectrum(d / "ideal_h2o.nc", col), ncio.read_spectrum(d / "ideal_co2.nc", col)
        od = 0.5 * h["optical_depth"] + (6e-4 / c["vmr_fl"])[:, None] * c["optical_depth"]
        dn4, up4 = api.lbl_band_fluxes_lw(ctx, h["temperature_hl"], dev(wn), dev(dwn), dev(od), begin, end, nangle=4)
        assert np.allclose(g.variables["band_flux_dn_lw"][col], dn4.T, rtol=3e-7, atol=1e-30)
        assert np.allclose(g.variables["band_flux_up_lw"][col], up4.T, rtol=3e-7)
        # the two-stream fluxes are an approximation of the same integral: close, not equal
        two, four = f.variables["flux_up_lw"][col].astype(np.float64), g.variables["flux_up_lw"][col].astype(np.float64)
        assert 0.0 < np.max(np.abs(two - four) / four) < 0.03
    f.close(); g.close()
    # ---- errors: more angles than the library integrates, a namelist without bands ----
    (d / "bad.nam").write_text("&longwave_config\nnangle = 40,\nband_wavenumber1(1:1) = 0,\nband_wavenumber2(1:1) = 3260\n/\n")
    r = run_tool("ckdmip_lw", "--config", "bad.nam", "ideal_h2o.nc", "--output", "x.nc", cwd=d)
    assert r.returncode == 147 and "nangle" in r.stderr
    r = run_tool("ckdmip_lw", "ideal_h2o.nc", "--output", "x.nc", cwd=d)
    assert r.returncode == 147 and "band_wavenumber1" in r.stderr
    # ---- --ckd: fluxes from the optical depths run_ckd wrote ----
    from test_pipeline_gpu import make_optimize_files
    e = tmp_path / "ckd"
    e.mkdir()
    model, truth, scenes, paths, ib, names = make_optimize_files(ctx, e)
    sc = scenes[0]
    ncol, nhl = sc["pressure_hl"].shape
    w = netcdf_file(str(e / "conc.nc"), "w", version=2)
    for dim, n in (("column", ncol), ("half_level", nhl), ("level", nhl - 1)):
        w.createDimension(dim, n)
    w.createVariable("pressure_hl", "d", ("column", "half_level"))[:] = sc["pressure_hl"]
    w.createVariable("temperature_hl", "d", ("column", "half_level"))[:] = sc["temperature_hl"]
    for i, g in enumerate(model["gases"]):
        if g["conc"] != "none":
            w.createVariable(names[i] + "_mole_fraction_fl", "d", ("column", "level"))[:] = sc["vmr_fl"][:, i, :]
    w.close()
    r = run_tool("run_ckd", "ckd_model=raw.nc", "input=conc.nc", "output=od.nc", cwd=e)
    assert r.returncode == 0, r.stderr + r.stdout
    r = run_tool("ckdmip_lw", "--scenario", "present", "--ckd", "od.nc", "--output", "fluxes.nc", cwd=e)
    assert r.returncode == 0, r.stderr + r.stdout
    a, b = _nc(e / "od.nc"), _nc(e / "fluxes.nc")
    for k in ("flux_dn_lw", "flux_up_lw"):          # run_ckd's own fluxes come from the unrounded optical depths: FLOAT agreement
        assert np.allclose(b.variables[k][...], a.variables[k][...], rtol=2e-5, atol=1e-4), k
    assert b.variables["spectral_flux_dn_lw"].shape == a.variables["optical_depth"].shape[:1] + (nhl, a.variables["optical_depth"].shape[2])
    # ... and with the namelist of test/run_ckd_lw.sh:76-90 (NANGLE=4): the quadrature on the g points' optical depths,
    # checked against the same four angles evaluated with numpy
    (e / "ckd4.nam").write_text("&longwave_config\noptical_depth_name = \"optical_depth\",\nnangle = 4,\niverbose = 3\n/\n")
    r = run_tool("ckdmip_lw", "--config", "ckd4.nam", "--scenario", "present", "--ckd", "od.nc", "--output", "fluxes4.nc", cwd=e)
    assert r.returncode == 0, r.stderr + r.stdout
    c4 = _nc(e / "fluxes4.nc")
    od = a.variables["optical_depth"][...].astype(np.float64)
    pl = a.variables["planck_hl"][...].astype(np.float64) if "planck_hl" in a.variables else None
    mu, wq = api.gauss_legendre_01(4)
    if pl is not None:
        want_dn = np.zeros_like(pl); want_up = np.zeros_like(pl)
        for m, w_ in zip(mu, wq):
            eps = 1.0 - np.exp(-od / m)
            fac = np.where(eps > 1e-5, 1.0 - eps * m / np.where(od > 0, od, 1.0), 0.5 * eps)
            dn_ = np.zeros_like(pl); up_ = np.zeros_like(pl)
            for l in range(nhl - 1):
                dn_[:, l + 1] = dn_[:, l] * (1 - eps[:, l]) + pl[:, l] * (eps[:, l] - fac[:, l]) + pl[:, l + 1] * fac[:, l]
            up_[:, -1] = pl[:, -1]
            for l in range(nhl - 2, -1, -1):
                up_[:, l] = up_[:, l + 1] * (1 - eps[:, l]) + pl[:, l + 1] * (eps[:, l] - fac[:, l]) + pl[:, l] * fac[:, l]
            want_dn += 2 * w_ * m * dn_; want_up += 2 * w_ * m * up_
        assert np.allclose(c4.variables["spectral_flux_dn_lw"][...], want_dn, rtol=3e-5, atol=1e-6)
        assert np.allclose(c4.variables["spectral_flux_up_lw"][...], want_up, rtol=3e-5, atol=1e-6)
    diff = np.abs(c4.variables["flux_up_lw"][...].astype(np.float64) - b.variables["flux_up_lw"][...].astype(np.float64))
    assert 0.0 < diff.max() < 0.03 * np.abs(b.variables["flux_up_lw"][...]).max()
    a.close(); b.close(); c4.close()


def test_ckdmip_sw_stand_in(ctx, tmp_path):
    """bin/ckdmip_sw, the stand-in for the external CKDMIP shortwave tool of the reference's scripts: --merge-only
    (test/merge_well_mixed_sw.sh:35-81), the line-by-line band fluxes for every zenith angle of the namelist
    (test/run_sw_lbl_evaluation.sh) against ecckd_lbl_band_fluxes_sw and through LblFluxes::read's mirror (five angles of which
    the reference keeps three, lbl_fluxes.cpp:88), and the flux evaluation of a CKD model's optical depths
    (test/run_ckd_sw.sh:125-128) against the closed form."""
    import torch
    from ecckd_amd import api, ncio
    d = tmp_path
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=ctx.device)
    nlay, nwav, lo, hi = 16, 8000, 250.0, 50000.0
    p1 = syn.pressure_grid(nlay)
    wn, dwn = syn.wavenumber_grid(nwav, lo, hi)
    ssi = syn.solar_spectral_irradiance(wn, dwn)
    w = netcdf_file(str(d / "ssi.nc"), "w", version=2)
    w.createDimension("wavenumber", nwav)
    w.createVariable("solar_spectral_irradiance", "d", ("wavenumber",))[:] = ssi
    w.close()
    t0 = syn.temperature_profile(p1)
    base = {"h2o": (syn.optical_depth(np, p1, wn, syn.SEED_BASE + 81, nlines=60, column_scale=3.0, dtype="float32", lo=lo, hi=hi), 5e-3),
            "o3": (syn.optical_depth(np, p1, wn, syn.SEED_BASE + 83, nlines=30, column_scale=0.8, dtype="float32", lo=lo, hi=hi), 1e-6)}
    for g, (od, vmr) in base.items():
        _write_columns_from(d / f"ideal_{g}.nc", g, p1, [t0 - 20.0, t0, t0 + 20.0], wn, od, vmr)
    # ---- merge-only ----
    r = run_tool("ckdmip_sw", "--merge-only", "ideal_h2o.nc", "--scale", "2", "ideal_o3.nc", "--output", "merged.nc", cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    m = ncio.read_spectrum(d / "merged.nc", 1)
    assert np.allclose(m["optical_depth"], (base["h2o"][0] + 2.0 * base["o3"][0]).astype(np.float32), rtol=2e-7) and m["molecule"] == "composite"
    # ---- line-by-line band fluxes: 3 columns x 5 zenith angles, 2 bands ----
    (d / "sw.nam").write_text("&shortwave_config\noptical_depth_name = \"optical_depth\",\nsurf_albedo = 0.15,\nuse_mu0_dimension = true,\n"
                              "cos_solar_zenith_angle(1:5) = 0.1, 0.3, 0.5, 0.7, 0.9,\nnspectralstride = 1,\n"
                              "band_wavenumber1(1:2) = 250, 10000,\nband_wavenumber2(1:2) = 10000, 50000,\niverbose = 3\n/\n")
    r = run_tool("ckdmip_sw", "--config", "sw.nam", "--scenario", "present", "--ssi", "ssi.nc", "ideal_h2o.nc", "--scale", "0.5", "ideal_o3.nc",
                 "--output", "lbl_sw_tool.nc", cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    f = _nc(d / "lbl_sw_tool.nc")
    mu0s = np.array([0.1, 0.3, 0.5, 0.7, 0.9])
    assert np.allclose(f.variables["mu0"][:], mu0s) and f.constituent_id == b"h2o o3" and f.scenario == b"present"
    b1, b2 = np.array([lo, 10000.0]), np.array([10000.0, hi])
    begin = [int(np.nonzero((wn >= a) & (wn < b + (b == hi)))[0][0]) for a, b in zip(b1, b2)]
    end = [int(np.nonzero((wn >= a) & (wn < b + (b == hi)))[0][-1]) for a, b in zip(b1, b2)]
    od = base["h2o"][0].astype(np.float64) + 0.5 * base["o3"][0].astype(np.float64)
    alb = dev(np.full(nwav, 0.15))
    for k, mu in enumerate(mu0s):
        dn, up = api.lbl_band_fluxes_sw(ctx, mu, dev(ssi), dev(od), begin, end, albedo=alb)
        for col in range(3):                       # the three columns share the optical depths (only the temperatures differ)
            assert np.allclose(f.variables["band_flux_dn_direct_sw"][col, k], dn.T, rtol=3e-7, atol=1e-30)
            assert np.allclose(f.variables["band_flux_up_sw"][col, k], up.T, rtol=3e-7, atol=1e-30)
            assert np.allclose(f.variables["flux_dn_direct_sw"][col, k], dn.sum(0), rtol=3e-7)
        assert f.variables["flux_dn_direct_sw"][0, k, 0] == pytest.approx(mu * ssi.sum(), rel=1e-6)        # top of atmosphere
        assert np.all(up[:, -1] <= 0.15 * dn[:, -1] * (1 + 1e-12))
    assert np.allclose(f.variables["mole_fraction_fl"][1], np.stack([np.full(nlay, 5e-3), np.full(nlay, 0.5e-6)]), rtol=2e-7)
    f.close()
    s = ncio.read_lbl_fluxes(d / "lbl_sw_tool.nc", ["h2o", "o3"], ctx=ctx)           # what optimize_lut reads (lbl_fluxes.cpp:52-133)
    assert s["is_sw"] and s["have_band_fluxes"] and s["flux_dn"].shape == (9, nlay + 1, 2)      # 3 columns x the angles 0, 2, 4
    assert np.allclose(s["mu0"], np.tile([0.1, 0.5, 0.9], 3)) and s["tsi"] == pytest.approx(ssi.sum(), rel=1e-6)
    # ---- the same with do_write_spectral_boundary_fluxes: (column, mu0, wavenumber) fluxes at the boundaries (lbl_fluxes.cpp:183-246) ----
    (d / "sw_b.nam").write_text((d / "sw.nam").read_text().replace("nspectralstride = 1,", "nspectralstride = 1,\ndo_write_spectral_boundary_fluxes = true,"))
    r = run_tool("ckdmip_sw", "--config", "sw_b.nam", "--ssi", "ssi.nc", "ideal_h2o.nc", "--scale", "0.5", "ideal_o3.nc", "--output", "lbl_sw_b.nc", cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    f, g = _nc(d / "lbl_sw_tool.nc"), _nc(d / "lbl_sw_b.nc")
    assert np.array_equal(g.variables["wavenumber"][...], wn)
    for name in ("band_flux_dn_direct_sw", "band_flux_up_sw", "flux_up_sw"):
        assert np.array_equal(f.variables[name][...], g.variables[name][...]), name
    assert g.variables["spectral_flux_dn_direct_surf_sw"].shape == (3, 5, nwav)
    for k, mu in enumerate(mu0s):
        _, _, sdn, tup = api.lbl_band_fluxes_sw(ctx, mu, dev(ssi), dev(od), begin, end, albedo=alb, boundary=True)
        for col in (0, 2):
            assert np.allclose(g.variables["spectral_flux_dn_direct_surf_sw"][col, k], sdn.cpu().numpy(), rtol=3e-7, atol=1e-37)
            assert np.allclose(g.variables["spectral_flux_up_toa_sw"][col, k], tup.cpu().numpy(), rtol=3e-7, atol=1e-37)
        assert np.isclose(g.variables["spectral_flux_dn_direct_surf_sw"][1, k].astype(np.float64).sum(), g.variables["flux_dn_direct_sw"][1, k][-1], rtol=1e-5)
        assert np.isclose(g.variables["spectral_flux_up_toa_sw"][1, k].astype(np.float64).sum(), g.variables["flux_up_sw"][1, k][0], rtol=1e-5)
    f.close(); g.close()
    # ---- errors ----
    r = run_tool("ckdmip_sw", "--config", "sw.nam", "ideal_h2o.nc", "--output", "x.nc", cwd=d)
    assert r.returncode == 147 and "--ssi" in r.stderr
    # ---- --ckd: direct beam and reflected flux on g-point optical depths (the file run_ckd writes for a shortwave model) ----
    rs = np.random.RandomState(5)
    ncol, ng = 2, 7
    odg = rs.uniform(0.0, 0.4, (ncol, nlay, ng))
    ray = rs.uniform(0.0, 0.05, (ncol, nlay, ng))
    inc = rs.uniform(10.0, 300.0, (ncol, ng))
    w = netcdf_file(str(d / "od_sw.nc"), "w", version=2)
    for dim, n in (("column", ncol), ("half_level", nlay + 1), ("level", nlay), ("g_point", ng)):
        w.createDimension(dim, n)
    w.createVariable("pressure_hl", "d", ("column", "half_level"))[:] = np.tile(p1, (ncol, 1))
    w.createVariable("optical_depth", "d", ("column", "level", "g_point"))[:] = odg
    w.createVariable("rayleigh_optical_depth", "d", ("column", "level", "g_point"))[:] = ray
    w.createVariable("incoming_sw", "d", ("column", "g_point"))[:] = inc
    w.close()
    r = run_tool("ckdmip_sw", "--config", "sw.nam", "--ckd", "od_sw.nc", "--output", "fluxes_sw.nc", cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    f = _nc(d / "fluxes_sw.nc")
    tau = np.concatenate([np.zeros((ncol, 1, ng)), np.cumsum(odg + ray, axis=1)], axis=1)                # to the top of every half level
    for k, mu in enumerate(mu0s):
        dn = mu * inc[:, None, :] * np.exp(-tau / mu)
        up = 0.15 * dn[:, -1:, :] * np.exp(-2.0 * (tau[:, -1:, :] - tau))
        assert np.allclose(f.variables["spectral_flux_dn_direct_sw"][:, k], dn, rtol=3e-6)
        assert np.allclose(f.variables["spectral_flux_up_sw"][:, k], up, rtol=3e-6)
        assert np.allclose(f.variables["flux_up_sw"][:, k], up.sum(-1), rtol=3e-6)
    f.close()
