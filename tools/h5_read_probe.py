"""Reading one column of a NetCDF-4 (HDF5: chunked, shuffle + deflate 2) spectrum into device memory: where the time goes.
usage: python tools/h5_read_probe.py [nwav] [chunk_points]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import h5_fixture as h5
from ecckd_amd import api, ncio, synthetic as syn

nwav = int(sys.argv[1]) if len(sys.argv) > 1 else 4_194_304
cpts = int(sys.argv[2]) if len(sys.argv) > 2 else 262_144
nlay = 54
path = "/tmp/probe_spectrum.h5"
p = syn.pressure_grid(nlay)
wn, _ = syn.wavenumber_grid(nwav)
t0 = time.perf_counter()
od = syn.optical_depth_lines(np, p, wn[: 1 << 19], syn.SEED_BASE + 5, nlines=2000, dtype="float32")
od = np.tile(od, (1, nwav // od.shape[1] + 1))[:, :nwav] * np.linspace(0.5, 2.0, nwav, dtype=np.float32)[None, :]
print(f"spectrum made in {time.perf_counter() - t0:.1f} s", flush=True)
t0 = time.perf_counter()
h5.write(path, {"optical_depth": (od[None], "f4", (1, 1, cpts), None)})
print(f"file written in {time.perf_counter() - t0:.1f} s: {os.path.getsize(path) / 1e6:.0f} MB for {od.nbytes / 1e6:.0f} MB", flush=True)
ctx = api.Context(0)
import torch
for label, env in (("worker threads read and inflate, device places (default)", {}),
                   ("the same with zlib instead of the in-tree decoder", {"ECCKD_ZLIB_INFLATE": "1"}),
                   ("calling thread reads the raw chunks, worker threads inflate with zlib (the path before)", {"ECCKD_H5_SERIAL_READ": "1", "ECCKD_ZLIB_INFLATE": "1"}),
                   ("device inflates and places", {"ECCKD_GPU_INFLATE": "1"}),
                   ("worker threads inflate and place, one upload", {"ECCKD_NO_DEVICE_PLACE": "1"}),
                   ("HDF5 library alone", {"ECCKD_NO_PARALLEL_INFLATE": "1"})):
    for k in ("ECCKD_NO_PARALLEL_INFLATE", "ECCKD_GPU_INFLATE", "ECCKD_NO_DEVICE_PLACE", "ECCKD_H5_SERIAL_READ", "ECCKD_ZLIB_INFLATE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    os.environ["ECCKD_H5_TIMES"] = "1"
    for rep in range(2):
        f = ncio.NcFile(path)
        ctx.synchronize(); t0 = time.perf_counter()
        out = f.read_dev(ctx, "optical_depth", 0)
        ctx.synchronize(); dt = time.perf_counter() - t0
        f.close()
    ok = bool(torch.equal(out.cpu(), torch.from_numpy(od)))
    print(f"{label}: {dt * 1e3:.0f} ms = {od.nbytes / dt / 1e9:.2f} GB/s of values, identical: {ok}", flush=True)
