"""Top kernels of a rocprofv3 *_kernel_stats.csv.  Usage: python tools/prof_top.py <kernel_stats.csv> [n]"""
import csv
import sys

n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in list(csv.DictReader(open(sys.argv[1])))[:n]:
    print(f'{r["Name"][:90]:90s} calls={r["Calls"]:>6s} total_ms={float(r["TotalDurationNs"]) / 1e6:9.2f} '
          f'avg_us={float(r["AverageNs"]) / 1e3:9.1f} {r["Percentage"]}%')
