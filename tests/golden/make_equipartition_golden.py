#!/usr/bin/env python3
"""Generate tests/golden/equipartition_known_answers.json.

Known answers of the reference's only test-like program, src/ecckd/test_equipartition.cpp:
partition exp(linspace(-2, 10, 1e6)) into 16 equal-error intervals with linear then cubic
interpolation (:50-85).  Its calc_error (:28-34) is restated below with numpy; the SEARCH is
the reference's own equipartition.cpp compiled into oracle/_ref (needs /root/reference at
build time).  Run from the repo root:  python tests/golden/make_equipartition_golden.py
"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402


def make_error_fn(npoints):
    values = np.exp(np.linspace(-2.0, 10.0, npoints))

    def calc_error(b1, b2):
        i1 = int(math.ceil(b1 * (npoints - 1)))
        i2 = int(math.floor(b2 * (npoints - 1)))
        # sum in index order (np.cumsum is sequential; np.sum would be pairwise)
        s = float(np.cumsum(values[i1:i2 + 1])[-1])
        return abs(s - (i2 - i1 + 1) * values[(i1 + i2) // 2])

    return calc_error


def main():
    npoints = 1000000
    out = {"npoints": npoints, "cases": []}
    fn = make_error_fn(npoints)
    # test_equipartition.cpp:57-61: the object (and its errors_up_to_date flag) persists across both runs
    ep = pyoracle.RefEquipartition(fn, resolution=1.0 / npoints, partition_tolerance=0.001,
                                   partition_max_iterations=200, line_search_max_iterations=15)
    for cubic in (0, 1):
        ep.r.refep_set_cubic_interpolation(ep.h, cubic)
        ncall0 = len(ep.calls)
        st, b, e = ep.equipartition_n(np.linspace(0.0, 1.0, 17))
        calls = ep.calls[ncall0:]
        out["cases"].append({
            "cubic": cubic, "status": st, "bounds": [float(x) for x in b], "error": [float(x) for x in e],
            "n_calls": len(calls), "comp_cost": float(sum(c[1] - c[0] for c in calls)),
            "first_calls": [[c[0], c[1]] for c in calls[:40]],
        })
        print("cubic", cubic, "status", pyoracle.EP_STATUS[st], "calls", len(calls))
    with open(os.path.join(ROOT, "tests", "golden", "equipartition_known_answers.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
