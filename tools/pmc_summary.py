"""Summarise rocprofv3 counter-collection CSVs for one kernel: average counter value per launch.
Usage: python tools/pmc_summary.py <kernel-substring> <counter_collection.csv> [...]"""
import csv
import sys
from collections import defaultdict


def main():
    pat = sys.argv[1]
    for path in sys.argv[2:]:
        sums, counts = defaultdict(float), defaultdict(int)
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if pat not in row.get("Kernel_Name", ""):
                    continue
                name = row["Counter_Name"]
                sums[name] += float(row["Counter_Value"])
                counts[name] += 1
        for name in sorted(sums):
            print(f"{path}: {name}: launches={counts[name]} avg={sums[name] / counts[name]:.6g}")


if __name__ == "__main__":
    main()
