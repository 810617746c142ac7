"""ecckd_nc_read_dev: a variable of a NetCDF file straight into device memory (pinned chunks, copy stream, decoding on the
device) against the element-wise host reader (nc_get_vara_double semantics, DataFileEngineNetcdf.cpp:593-608): FLOAT and
DOUBLE variables larger than several chunks, one slice and the whole variable, widening and narrowing, the fallbacks
(integer type, record variable) and the tools' use of it (tests/test_cli_gpu.py runs them on files)."""
import numpy as np
import pytest
import torch
from scipy.io import netcdf_file

pytestmark = pytest.mark.gpu


def test_streamed_read_matches_host_reader(ctx, tmp_path, monkeypatch):
    from ecckd_amd import ncio
    rs = np.random.RandomState(1)
    ncol, nlay, nwav = 2, 3, 2_000_003            # 24 MB per FLOAT slice: two 16-MB chunks, the second one ragged
    od = (rs.standard_normal((ncol, nlay, nwav)) * 1e3).astype(np.float32)
    od[0, 0, :5] = [0.0, -0.0, np.float32(1e-40), np.inf, -1.5]
    wn = np.cumsum(rs.uniform(0.1, 1.0, nwav))
    path = tmp_path / "s.nc"
    w = netcdf_file(str(path), "w", version=2)
    for d, n in (("column", ncol), ("level", nlay), ("wavenumber", nwav)):
        w.createDimension(d, n)
    w.createVariable("optical_depth", "f", ("column", "level", "wavenumber"))[:] = od
    w.createVariable("wavenumber", "d", ("wavenumber",))[:] = wn
    w.createVariable("count", "i", ("level",))[:] = [3, -7, 2**31 - 1]
    w.close()
    with ncio.NcFile(path) as f:
        for idx in (0, 1):
            got = f.read_dev(ctx, "optical_depth", idx)
            assert got.dtype == torch.float32 and got.shape == (nlay, nwav)
            assert np.array_equal(got.cpu().numpy().view(np.uint32), od[idx].view(np.uint32))          # bit for bit, -0.0 and denormals too
            wide = f.read_dev(ctx, "optical_depth", idx, dtype=torch.float64)
            assert np.array_equal(wide.cpu().numpy(), f.read("optical_depth", idx))
        whole = f.read_dev(ctx, "optical_depth")
        assert whole.shape == (ncol, nlay, nwav) and np.array_equal(whole.cpu().numpy().view(np.uint32), od.view(np.uint32))
        assert np.array_equal(f.read_dev(ctx, "wavenumber").cpu().numpy(), wn)
        assert np.array_equal(f.read_dev(ctx, "wavenumber", dtype=torch.float32).cpu().numpy(), wn.astype(np.float32))
        assert np.array_equal(f.read_dev(ctx, "count").cpu().numpy(), [3.0, -7.0, 2.0**31 - 1])           # integer type: host path
        monkeypatch.setenv("ECCKD_NO_STREAMED_READ", "1")                                               # the fallback gives the same
        assert np.array_equal(f.read_dev(ctx, "optical_depth", 1).cpu().numpy().view(np.uint32), od[1].view(np.uint32))


def test_netcdf4_variable_to_device(ctx, tmp_path):
    """A NetCDF-4 (HDF5) spectrum read straight to the device: the chunks are inflated by worker threads into the requested
    type (FLOAT stays FLOAT) and uploaded once."""
    import os, sys
    sys.path.insert(0, os.path.dirname(__file__))
    import h5_fixture as h5
    from ecckd_amd import ncio
    if not h5.available():
        pytest.skip("no HDF5 shared library with the deflate filter in this environment")
    rs = np.random.RandomState(2)
    od = (rs.standard_normal((2, 9, 70_001)) * 5).astype(np.float32)
    h5.write(tmp_path / "s.h5", {"optical_depth": (od, "f4", (1, 9, 8192), None)})
    with ncio.NcFile(tmp_path / "s.h5") as f:
        got = f.read_dev(ctx, "optical_depth", 1)
        assert got.dtype == torch.float32 and np.array_equal(got.cpu().numpy().view(np.uint32), od[1].view(np.uint32))
        assert np.array_equal(f.read_dev(ctx, "optical_depth", 0, dtype=torch.float64).cpu().numpy(), od[0].astype(np.float64))


@pytest.mark.parametrize("mode", ["device inflate", "host inflate", "host inflate, serial read", "host inflate, zlib", "host only"])
def test_netcdf4_chunks_to_the_device(ctx, tmp_path, monkeypatch, mode):
    """The NetCDF-4 read path: raw chunks read (pread at the addresses the library reports; ECCKD_H5_SERIAL_READ=1: by the calling
    thread through the library) and inflated by worker threads (csrc/fast_inflate.cpp, then zlib for what it refuses;
    ECCKD_ZLIB_INFLATE=1: zlib alone) into pinned slots (default) or on the device
    (ECCKD_GPU_INFLATE=1: csrc/inflate.hip, one wavefront per chunk), then unshuffled, converted and placed by k_place_chunks;
    or everything on the host and one upload (ECCKD_NO_DEVICE_PLACE=1).  Chunk shapes that do not divide the variable, chunks
    spanning several slices, DOUBLE storage, a 1-D variable, special values, both output types: the same bits every way."""
    import os, sys
    sys.path.insert(0, os.path.dirname(__file__))
    import h5_fixture as h5
    from ecckd_amd import ncio
    if not h5.available():
        pytest.skip("no HDF5 shared library with the deflate filter in this environment")
    if mode == "device inflate":
        monkeypatch.setenv("ECCKD_GPU_INFLATE", "1")
    elif mode == "host only":
        monkeypatch.setenv("ECCKD_NO_DEVICE_PLACE", "1")
    elif mode == "host inflate, serial read":
        monkeypatch.setenv("ECCKD_H5_SERIAL_READ", "1")
    elif mode == "host inflate, zlib":
        monkeypatch.setenv("ECCKD_ZLIB_INFLATE", "1")
    rs = np.random.RandomState(0)
    od = np.exp(np.cumsum(rs.normal(0, 0.05, (3, 17, 50_003)), axis=-1)).astype(np.float32)
    od[1, 3, :7] = [0.0, -0.0, np.float32(1e-42), np.inf, -np.inf, 1.0, -1.0]
    od[2, :, 1000:30_000] = 0.0                                                           # long runs: matches at distance 1
    x64 = rs.standard_normal((4, 5, 3001))
    noise = rs.standard_normal((2, 3, 20_000)).astype(np.float32)                       # incompressible mantissas
    path = tmp_path / "chunky.h5"
    h5.write(path, {"optical_depth": (od, "f4", (2, 5, 4096), None), "x64": (x64, "f8", (1, 2, 777), None),
                    "flat": (od[0, 0], "f4", (1000,), None), "noise": (noise, "f4", (1, 1, 20_000), None)})
    with ncio.NcFile(path) as f:
        for k in range(3):
            got = f.read_dev(ctx, "optical_depth", k)
            assert got.dtype == torch.float32 and np.array_equal(got.cpu().numpy().view(np.uint32), od[k].view(np.uint32)), k
        assert np.array_equal(f.read_dev(ctx, "optical_depth", dtype=torch.float64).cpu().numpy().view(np.uint64),
                              od.astype(np.float64).view(np.uint64))
        assert np.array_equal(f.read_dev(ctx, "x64", 3).cpu().numpy(), x64[3])
        assert np.array_equal(f.read_dev(ctx, "x64").cpu().numpy(), x64)
        assert np.array_equal(f.read_dev(ctx, "x64", 1, dtype=torch.float32).cpu().numpy(), x64[1].astype(np.float32))
        assert np.array_equal(f.read_dev(ctx, "flat").cpu().numpy(), od[0, 0])
        assert np.array_equal(f.read_dev(ctx, "noise", 1).cpu().numpy(), noise[1])
