#!/usr/bin/env python3
"""HBM bytes per L-BFGS iteration of the LUT optimisation from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
tools/lut_opt_probe.py: per kernel the counter summed over its launches, and the sum over the optimiser's kernels divided by the
iterations of the run.  FETCH_SIZE / WRITE_SIZE are in KB (1024 B); traffic = 2 x FETCH + WRITE as MI355X_MICROARCH.md prescribes
for gfx950 (the raw sums are printed too: the optimiser's reads are gathers of 8-byte rows served mostly by the L2s, for which
the factor 2 of wide streaming reads is an upper bound).
    python tools/opt_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <lut_opt_probe.json>"""
import csv
import json
import re
import sys
from collections import defaultdict


def sums(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            m = re.search(r"(k_[A-Za-z0-9_]+)", row.get("Kernel_Name", ""))
            if not m:
                continue
            tot[m.group(1)] += float(row["Counter_Value"])
            n[m.group(1)] += 1
    return tot, n


fetch_csv, write_csv, probe = sys.argv[1:4]
line = json.loads(open(probe).read().strip().splitlines()[-1])
its = line["iterations"]
F, nF = sums(fetch_csv, "FETCH_SIZE")
W, nW = sums(write_csv, "WRITE_SIZE")
kernels = sorted(k for k in F if k.startswith(("k_opt", "k_lb")))
per = {k: {"launches": nF[k], "fetch_kb_per_iteration": F[k] / its, "write_kb_per_iteration": W.get(k, 0.0) / its} for k in kernels}
fetch = sum(F[k] for k in kernels) * 1024.0 / its
write = sum(W.get(k, 0.0) for k in kernels) * 1024.0 / its
print(json.dumps({"iterations": its, "nx": line["nx"], "fetch_bytes_per_iteration": fetch, "write_bytes_per_iteration": write,
                  "corrected_bytes_per_iteration": 2.0 * fetch + write,
                  "algorithmic_bytes_per_iteration": line["roofline"]["algorithmic_bytes_per_iteration"], "kernels": per,
                  "note": "HBM traffic only: the 265 + 221 MB of row gathers per evaluation are served by the L2s (88 % hits, "
                          "profiles/r03_pmc_opt_kernels.md); 2 x FETCH_SIZE + WRITE_SIZE, KB = 1024 B"}, indent=1))
