"""Where a search batch's time goes, from a rocprofv3 --kernel-trace database (rocpd SQLite): per batch of the longwave
search the three kernels (interval sums + fit, sweep, cost) and the three gaps (sums -> sweep, sweep -> cost, cost -> next
batch's sums), grouped by the number of intervals in the batch (grid.y of the sums kernel).
usage: python tools/batch_breakdown.py <results.db>"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
gy = "grid_y" if "grid_y" in cols else ("grid_size_y" if "grid_size_y" in cols else None)
wy = "workgroup_y" if "workgroup_y" in cols else ("workgroup_size_y" if "workgroup_size_y" in cols else None)
sel = "name,start,end" + ("," + gy if gy else "") + ("," + wy if wy else "") + ("," + gx if gx else "")
rows = cur.execute(f"select {sel} from kernels order by start").fetchall()
own = [r for r in rows if "k_interval_sums_fit_lw" in r[0] or "k_rt_lw_bb" in r[0] or "k_cost_lw" in r[0]]
batches = []
i = 0
while i + 2 < len(own):
    a, c, d = own[i], own[i + 1], own[i + 2]
    if "k_interval_sums_fit_lw" in a[0] and "k_rt_lw_bb" in c[0] and "k_cost_lw" in d[0]:
        nint = (a[3] // (a[4] if wy and a[4] else 1)) if gy else 0
        nxt = own[i + 3][1] if i + 3 < len(own) and "k_interval_sums_fit_lw" in own[i + 3][0] else None
        batches.append((nint, a[2] - a[1], c[1] - a[2], c[2] - c[1], d[1] - c[2], d[2] - d[1], (nxt - d[2]) if nxt else None,
                        (c[5] // 256) if gx and len(c) > 5 else 0))
        i += 3
    else:
        i += 1
print(f"{len(batches)} batches; columns in us: sums | gap | sweep | gap | cost | gap to the next batch (host turnaround)")
groups = collections.defaultdict(list)
for b in batches:
    n = b[0]
    key = n if n <= 4 else (8 if n <= 8 else (16 if n <= 16 else (32 if n <= 32 else 64)))
    groups[key].append(b)
tot = [0.0] * 6
for key in sorted(groups):
    g = groups[key]
    m = [sum(b[j] for b in g) / len(g) / 1e3 for j in range(1, 6)]
    host = [b[6] for b in g if b[6] is not None and b[6] < 5e6]
    mh = sum(host) / max(len(host), 1) / 1e3
    print(f"nint<={key:3d}: {len(g):6d} batches  sums {m[0]:7.2f}  gap {m[1]:6.2f}  sweep {m[2]:8.2f}  gap {m[3]:6.2f}  cost {m[4]:6.2f}  next {mh:7.2f}")
    for j in range(5):
        tot[j] += sum(b[j + 1] for b in g)
    tot[5] += sum(host)
s = sum(tot)
print("share of the search time: " + "  ".join(f"{n} {100 * t / s:.1f}%" for n, t in zip(("sums", "gap", "sweep", "gap", "cost", "turnaround"), tot)))
print(f"per batch: {s / len(batches) / 1e3:.1f} us")

# single-interval batches by the number of sweep blocks (chunks of 128 points below 768 blocks, i.e. below 98 304 points)
print("single-interval batches by sweep blocks: count, mean sweep us, points/us if the chunks are 128 points")
one = [b for b in batches if b[0] == 1]
for lo, hi in ((1, 8), (9, 32), (33, 96), (97, 192), (193, 384), (385, 640), (641, 767), (768, 768), (769, 10 ** 9)):
    g = [b for b in one if lo <= b[7] <= hi]
    if g:
        ms = sum(b[3] for b in g) / len(g) / 1e3
        mb = sum(b[7] for b in g) / len(g)
        print(f"  blocks {lo:4d}-{hi if hi < 10**9 else 0:4d}: {len(g):5d}  sweep {ms:7.2f} us  mean blocks {mb:6.1f}  -> {mb * 128 / ms / 1e3:6.2f} e9 points/s (only if < 768 blocks)")
two = [b for b in batches if b[0] == 2]
print("two-interval batches by sweep blocks:")
for lo, hi in ((1, 64), (65, 256), (257, 768), (769, 1200), (1201, 1535), (1536, 10 ** 9)):
    g = [b for b in two if lo <= b[7] <= hi]
    if g:
        print(f"  blocks {lo:4d}-{hi if hi < 10**9 else 0:4d}: {len(g):5d}  sweep {sum(b[3] for b in g) / len(g) / 1e3:7.2f} us  mean blocks {sum(b[7] for b in g) / len(g):6.1f}")
