#!/usr/bin/env python3
"""Per kernel of a rocprofv3 --kernel-trace CSV: launches, average and summed duration, and the UNION of the time in which at
least one launch of it was executing - what replaces "launches x average duration" as the device time of a kernel whose launches
overlap on several streams (the side-by-side searches of ecckd_find_g_gases).  With --bytes-per-launch-total B the bytes over
the union are printed for the kernel matching --kernel.
    python tools/kernel_union.py <kernel_trace.csv> [--kernel k_rt_lw_bb_mirror] [--total-bytes 1.48e13]"""
import argparse
import csv
import re
import json
import sys
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--kernel", default="k_rt_lw_bb_mirror")
    ap.add_argument("--total-bytes", type=float, default=None)
    args = ap.parse_args()
    spans = defaultdict(list)
    with open(args.csv, newline="") as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name") or row.get("kernel_name") or ""
            a = int(row.get("Start_Timestamp") or row.get("start_timestamp"))
            b = int(row.get("End_Timestamp") or row.get("end_timestamp"))
            # "void (anonymous namespace)::k_rt_lw_bb_mirror<54, false>(unsigned long, ...)" -> "k_rt_lw_bb_mirror<54, false>"
            m = re.search(r"(k_[A-Za-z0-9_]+(?:<[^()]*?>)?)\(", name)
            short = m.group(1) if m else name.split("(")[0].replace("void ", "")[:80]
            spans[short].append((a, b))
    out = {}
    for name, sp in spans.items():
        sp.sort()
        total = sum(b - a for a, b in sp)
        union, cur_a, cur_b = 0, sp[0][0], sp[0][1]
        streams_busy_max = 0
        for a, b in sp[1:]:
            if a <= cur_b:
                cur_b = max(cur_b, b)
            else:
                union += cur_b - cur_a
                cur_a, cur_b = a, b
        union += cur_b - cur_a
        out[name] = {"launches": len(sp), "avg_us": total / len(sp) / 1e3, "sum_ms": total / 1e6, "union_ms": union / 1e6,
                     "overlap_factor": total / max(union, 1)}
    rows = sorted(out.items(), key=lambda kv: -kv[1]["sum_ms"])
    res = {"kernels": {k: v for k, v in rows[:25]}}
    if args.total_bytes is not None:
        for k, v in rows:
            if args.kernel in k:
                res.setdefault("bytes_over_union", {})[k] = {"GB_per_s": args.total_bytes / (v["union_ms"] * 1e-3) / 1e9, "union_ms": v["union_ms"]}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    sys.exit(main())
