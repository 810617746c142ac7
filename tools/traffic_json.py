"""HBM traffic of the dominant kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected separately as
MI355X_MICROARCH.md prescribes) and the bench line of one of those runs.
Usage: python tools/traffic_json.py <kernel-substring> <fetch_counter_collection.csv> <write_counter_collection.csv> <bench.json> > out.json"""
import csv
import json
import sys


def avg(path, pat, counter):
    tot, n = 0.0, 0
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if pat in row.get("Kernel_Name", "") and row["Counter_Name"] == counter:
                tot += float(row["Counter_Value"]); n += 1
    return tot / max(n, 1), n


pat, fetch_csv, write_csv, bench = sys.argv[1:5]
fetch_kb, nf = avg(fetch_csv, pat, "FETCH_SIZE")
write_kb, nw = avg(write_csv, pat, "WRITE_SIZE")
line = json.loads(open(bench).read().strip().splitlines()[-1])
# mean points per launch over ALL launches (the counters are averaged over all of them; roofline.points_per_launch is the mean
# of the launches timed with events, every N-th one)
ppl = line.get("search", {}).get("points_per_batch") or line["roofline"]["points_per_launch"]
corrected = 2.0 * fetch_kb * 1024.0 + write_kb * 1024.0
print(json.dumps({
    "kernel": pat, "launches_fetch_pass": nf, "launches_write_pass": nw, "points_per_launch": ppl, "fetch_kb": fetch_kb,
    "write_kb": write_kb, "corrected_bytes_per_launch": corrected, "corrected_bytes_per_point": corrected / ppl,
    "algorithmic_bytes_per_point": line["roofline"]["algorithmic_bytes_per_point"],
    "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --steps 1 "
            "--warmup 0 --no-cpu --no-lut-opt --no-sw`; gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream "
            "(MI355X_MICROARCH.md, HBM section): traffic = 2*FETCH_SIZE + WRITE_SIZE (KB = 1024 B)"}, indent=1))
