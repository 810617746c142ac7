"""Throughput of the device inflate kernel on chunks like those of a NetCDF-4 spectrum file (shuffle + deflate level 2,
OutputDataFile.cpp:350-359): `nchunks` zlib streams of `chunk_kb` KB of byte-shuffled floats in one launch.
run under:  rocprofv3 --kernel-trace --stats -d /tmp/inf -o inf -- python3 tools/inflate_probe.py [nchunks] [chunk_kb]"""
import os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ecckd_amd import api, synthetic as syn

nchunks = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
chunk_kb = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n = chunk_kb * 1024 // 4
p = syn.pressure_grid(54)
wn, _ = syn.wavenumber_grid(n * 8)
od = syn.optical_depth_lines(np, p[:3], wn, syn.SEED_BASE + 5, nlines=2000, dtype="float32")      # two layers of a line spectrum
rows = od.reshape(-1)[: (od.size // n) * n].reshape(-1, n)
distinct = []
for r in rows[:16]:
    distinct.append(zlib.compress(np.ascontiguousarray(r, dtype="<f4").view(np.uint8).reshape(n, 4).T.copy().tobytes(), 2))
ratio = sum(len(d) for d in distinct) / (len(distinct) * n * 4)
streams = [distinct[k % len(distinct)] for k in range(nchunks)]
ctx = api.Context(0)
t0 = time.perf_counter()
out, status = api.inflate(ctx, streams, [n * 4] * nchunks)
dt = time.perf_counter() - t0
assert not status.any()
ref = zlib.decompress(distinct[3])
assert out[3] == ref and out[3 + len(distinct)] == ref
t0 = time.perf_counter()
for d in distinct:
    zlib.decompress(d)
host = (time.perf_counter() - t0) / len(distinct)
print(f"{nchunks} chunks x {chunk_kb} KB, compressed to {100 * ratio:.0f} %: whole call {dt * 1e3:.0f} ms (staging, copies, kernel); "
      f"one host core inflates a chunk in {host * 1e3:.2f} ms = {n * 4 / host / 1e9:.2f} GB/s")
