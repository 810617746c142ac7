#!/usr/bin/env python3
"""The FSCK find_g_points job of BASELINE configs[1] on one device: six gases (composite, h2o, o3, co2, ch4, n2o), each with
a background MERGED from several spectra in double (read_merged_spectrum.cpp:135-166) - searched gas after gas and side by
side (ecckd_find_g_gases), the same prepared gases both times.  Prints both times, the per-gas results and whether the two
runs agree bit for bit.

    python tools/gases_probe.py [--nwav 7200000] [--tolerance 0.0161] [--widths 1,6] [--out profiles/....json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nwav", type=int, default=7_200_000)
    ap.add_argument("--nlay", type=int, default=54)
    ap.add_argument("--tolerance", type=float, default=0.0161)
    ap.add_argument("--tolerance-tolerance", type=float, default=0.01)
    ap.add_argument("--max-iterations", type=int, default=60)
    ap.add_argument("--nlines", type=int, default=12000)
    ap.add_argument("--ngas", type=int, default=6)
    ap.add_argument("--widths", default="1,6")
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    from ecckd_amd import api, fsck_job
    ctx = api.Context(0)
    job = fsck_job.FsckJob(ctx, args.nwav, args.nlay, ngas=args.ngas, nlines=args.nlines)
    t0 = time.perf_counter()
    gases = job.prepare()
    ctx.synchronize()
    t_prep = time.perf_counter() - t0
    out = {"nwav": args.nwav, "nlay": args.nlay, "gases": job.names, "background_files": job.background_names,
           "preparation_ms": t_prep * 1e3, "sweep_bytes_per_point": [g.sweep_bytes_per_point() for g in gases], "runs": []}
    ref = None
    for width in [int(w) for w in args.widths.split(",")]:
        for rep in range(args.repeat):
            for g in gases:
                g.reset_memo()
            ctx.synchronize()
            t0 = time.perf_counter()
            res = job.search(gases, args.tolerance, args.tolerance_tolerance, args.max_iterations, max_concurrent=width)
            dt = time.perf_counter() - t0
            swept = [g.eval_stats()["points_evaluated"] / args.nwav for g in gases]
            sig = [(r[0]["status"], tuple(r[0]["rank1"]), tuple(r[0]["rank2"]), r[0]["error"].tobytes()) for r in res]
            same = None if ref is None else bool(sig == ref)
            if ref is None:
                ref = sig
            out["runs"].append({"gases_side_by_side": width, "ms": dt * 1e3, "passes_swept": swept, "ng": [len(r[0]["error"]) for r in res],
                                "status": [r[0]["status"] for r in res], "identical_to_first_run": same,
                                "points_per_s": args.nwav * sum(swept) / dt})
            print(json.dumps(out["runs"][-1]), flush=True)
    for g in gases:
        g.close()
    job.close()
    print(json.dumps(out))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
