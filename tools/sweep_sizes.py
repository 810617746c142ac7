"""Efficiency of the longwave sweep kernel by launch size, from the per-launch log the library writes when ECCKD_SWEEP_LOG
is set (bench.py --steps 1): share of the kernel's time and achieved algorithmic bandwidth per size class.
usage: ECCKD_SWEEP_LOG=/tmp/sweeps.txt python bench.py --steps 1 --warmup 0 --no-cpu --no-lut-opt --no-sw; python tools/sweep_sizes.py /tmp/sweeps.txt"""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1])
nint, pts, chunks, ms = a.T
bpp = 872.0
tot = ms.sum()
print(f"{len(a)} launches, {tot:.1f} ms, {pts.sum() * bpp / tot / 1e9:.2f} TB/s overall")
edges = [0, 4e3, 16e3, 49152, 98304, 2e5, 4e5, 8e5, 1.6e6, 3.2e6, 8e6]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (pts > lo) & (pts <= hi)
    if m.any():
        print(f"points {int(lo):8d}-{int(hi):8d}: {m.sum():5d} launches  {100 * ms[m].sum() / tot:5.1f}% of the time  mean {1e3 * ms[m].mean():7.1f} us  "
              f"{pts[m].sum() * bpp / ms[m].sum() / 1e9:5.2f} TB/s  mean intervals {nint[m].mean():.1f}  mean chunks {chunks[m].mean():.0f}")
