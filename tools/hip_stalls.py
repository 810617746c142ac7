"""Longest HIP API calls of the last bench step in a rocprofv3 (--kernel-trace --hip-trace) database: where the host stalls."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
k = cur.execute("select name,start,end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(k) if 'reorder_key_lw' in r[0]]
nback = int(sys.argv[2]) if len(sys.argv) > 2 else 2
t0 = k[idx[-nback]][1]
rows = cur.execute("select name,start,end from regions where start >= ? order by start", (t0,)).fetchall()
big = sorted(((r[2] - r[1], r[0], (r[1] - t0) / 1e6) for r in rows), reverse=True)
for d, n, t in big[:30]:
    print(f"{d/1e6:9.3f} ms  {n:44s} at {t:8.2f} ms")
import collections
tot = collections.Counter(); cnt = collections.Counter()
for r in rows:
    tot[r[0]] += r[2] - r[1]; cnt[r[0]] += 1
print("--- totals")
for n, d in tot.most_common(15):
    print(f"{d/1e6:9.3f} ms  {cnt[n]:6d}  {n}")
