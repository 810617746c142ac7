"""Reading one column of a classic NetCDF spectrum (what the tools write and read between the stages) into device memory
with 1 ... 6 pread threads (ECCKD_READ_THREADS is read once per process, hence one child process per setting).
usage: python tools/classic_read_probe.py [nwav]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
nwav = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "child" else 4_194_304
path = "/tmp/probe_classic.nc"
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from ecckd_amd import api, ncio
    ctx = api.Context(0)
    ts = []
    for rep in range(5):
        f = ncio.NcFile(path)
        ctx.synchronize(); t0 = time.perf_counter()
        out = f.read_dev(ctx, "optical_depth", 0)
        ctx.synchronize(); ts.append(time.perf_counter() - t0)
        f.close()
    nbytes = out.numel() * out.element_size()
    best = min(ts)
    print(f"ECCKD_READ_THREADS={os.environ.get('ECCKD_READ_THREADS', 'default')}: {best * 1e3:.1f} ms = {nbytes / best / 1e9:.2f} GB/s ({nbytes / 1e6:.0f} MB); "
          "the reads in order: " + " ".join(f"{t * 1e3:.0f}" for t in ts) + " ms")
    sys.exit(0)
import numpy as np
from scipy.io import netcdf_file
nlay = 30
od = (np.arange(nlay * nwav, dtype=np.float32).reshape(nlay, nwav) % 977) * 1e-3
f = netcdf_file(path, "w", version=2)
f.createDimension("column", 1); f.createDimension("level", nlay); f.createDimension("wavenumber", nwav)
v = f.createVariable("optical_depth", "f4", ("column", "level", "wavenumber")); v[0] = od
f.close()
for n in ("1", "2", "4", "6", "4", "6"):
    r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, ECCKD_READ_THREADS=n), capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-500:])
