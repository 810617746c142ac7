#!/usr/bin/env python3
"""The longwave (or --sw) LUT optimisation of bench.py on its own: N L-BFGS iterations at nx = 3.05e5, for kernel traces and
counter passes (tools/round_profiles.sh).  Prints the bench's lut_opt block."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=100)
    ap.add_argument("--sw", action="store_true")
    args = ap.parse_args()
    import bench
    from ecckd_amd import api
    with api.Context(0) as ctx:
        print(json.dumps(bench.lut_opt_bench(ctx, args.iterations, sw=args.sw)))


if __name__ == "__main__":
    main()
