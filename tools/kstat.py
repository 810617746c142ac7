"""Average duration of the kernels whose name contains a substring, from a rocprofv3 *_kernel_stats.csv.
Usage: python tools/kstat.py <substring> <kernel_stats.csv> [...]"""
import csv
import sys

for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        if sys.argv[1] in r["Name"]:
            print(f"{path}: {r['Name'][:70]} calls={r['Calls']} avg_us={float(r['AverageNs']) / 1e3:.1f}")
