"""Where a tool's wall time goes: runs the find_g_points / create_look_up_table stages of tools/e2e_chain.py on synthetic files
with ECCKD_LOG_TIMES=1 and prints their time-stamped logs."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import e2e_chain as e
from ecckd_amd import api
nwav = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
d = tempfile.mkdtemp(prefix="ecckd_times_")
with api.Context(0) as ctx:
    inp = e.make_inputs(ctx, d, nwav, 30)
env = dict(os.environ, ECCKD_LOG_TIMES="1")
def run(name, *args):
    t0 = time.perf_counter()
    r = subprocess.run([os.path.join(ROOT, "bin", name), *args], cwd=d, capture_output=True, text=True, env=env)
    print(f"=== {name} {' '.join(args)}: {time.perf_counter() - t0:.3f} s (exit {r.returncode})")
    lines = r.stdout.splitlines()
    keep = [l for l in lines if "g point " not in l]
    print("\n".join(keep[:80]))
    if r.returncode: print(r.stderr[-1500:])
for g in e.GASES:
    run("reorder_spectrum", f"input=present_{g}.nc", f"output=order_{g}.nc", "wavenumber1=0 1300", "wavenumber2=1300 3260")
open(os.path.join(d, "find_g.cfg"), "w").write(
    "heating_rate_tolerance 0.1\nmax_iterations 30\ntolerance_tolerance 0.02\nflux_weight 0.02\naveraging_method transmission\ngases h2o co2\n"
    "\\begin h2o\n input present_h2o.nc\n reordering_input order_h2o.nc\n background_input present_co2.nc\n\\end h2o\n"
    "\\begin co2\n input present_co2.nc\n reordering_input order_co2.nc\n background_input present_h2o.nc\n\\end co2\n")
run("find_g_points", "find_g.cfg", "output=gpoints.nc")
open(os.path.join(d, "lut.cfg"), "w").write(
    "input gpoints.nc\noutput raw_ckd.nc\ngases h2o co2\n\\begin h2o\n conc_dependence lut\n input \"ideal_h2o.nc ideal_h2o_x4.nc\"\n\\end h2o\n"
    "\\begin co2\n conc_dependence linear\n input ideal_co2.nc\n\\end co2\n")
run("create_look_up_table", "lut.cfg")
