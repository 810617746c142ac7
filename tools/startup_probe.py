"""What one tool process costs before and after its work: bin/reorder_spectrum on a 4 096-point spectrum (the work itself is
microseconds), wall time of the process against the time between its first and last log line; the same with LD_BIND_NOW and
with the usage message only (no device)."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import e2e_chain as e
from ecckd_amd import api
d = tempfile.mkdtemp(prefix="ecckd_start_")
with api.Context(0) as ctx:
    e.make_inputs(ctx, d, 4096, 30)
exe = os.path.join(ROOT, "bin", "reorder_spectrum")
args = ["input=present_h2o.nc", "output=order_h2o.nc", "wavenumber1=0 1300", "wavenumber2=1300 3260"]
def wall(cmd, env=None, n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        r = subprocess.run(cmd, cwd=d, capture_output=True, text=True, env=env)
        ts.append(time.perf_counter() - t0)
    return min(ts), sorted(ts)[len(ts) // 2], r
lo, med, r = wall([exe])
print(f"usage message only (no device): min {lo:.3f} s, median {med:.3f} s (exit {r.returncode})")
env = dict(os.environ, ECCKD_LOG_TIMES="1")
lo, med, r = wall([exe] + args, env)
last = [l for l in r.stdout.splitlines() if l.startswith("[")][-1]
print(f"4 096-point spectrum: min {lo:.3f} s, median {med:.3f} s (exit {r.returncode}); last log line: {last.strip()}")
for k, v in (("AMD_LOG_LEVEL", "0"), ("HIP_VISIBLE_DEVICES", "0"), ("HSA_ENABLE_SDMA", "0"), ("GPU_MAX_HW_QUEUES", "1")):
    lo, med, r = wall([exe] + args, dict(env, **{k: v}))
    print(f"  with {k}={v}: min {lo:.3f} s, median {med:.3f} s")
print("kept work directory:", d)
