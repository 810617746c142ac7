"""Per-kernel averages of an optimize_lut kernel trace, by region (grid size of K8a), and one iteration's timeline."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
agg = collections.defaultdict(list)
region = "?"
for r in rows:
    n = name(r)
    if "forward_adjoint" in n:
        region = "sw" if ("true>" in n or int(r["Grid_Size_X"]) > 600000) else "lw"
    agg[(region, n)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    if len(v) > 5:
        print("%-4s %-46s n=%4d  avg %7.1f us" % (k[0], k[1], len(v), sum(v) / len(v)))
for tag in ("false>", "true>"):
    idx = [i for i, r in enumerate(rows) if "forward_adjoint" in r["Kernel_Name"] and tag in r["Kernel_Name"]]
    if len(idx) < 12:
        continue
    i0 = idx[10]
    t0 = int(rows[i0 - 5]["Start_Timestamp"])
    print()
    for r in rows[i0 - 5:i0 + 12]:
        print("  %-46s %7.1f us  @ %7.1f" % (name(r), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, (int(r["Start_Timestamp"]) - t0) / 1e3))
