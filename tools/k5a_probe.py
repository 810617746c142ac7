"""Durations of the interval-sums kernel (K5a) by number and kind of intervals.
run:  rocprofv3 --kernel-trace -d /tmp/k5a -o k5a -- python3 tools/k5a_probe.py run ; python3 tools/k5a_probe.py report /tmp/k5a/k5a_results.db"""
import os, sys
import numpy as np
CASES = [(n, kind) for kind in ("partition", "tiny", "medium") for n in (1, 2, 8, 16, 45)]
REPS = 4
if sys.argv[1] == "run":
    os.environ["ECCKD_NO_ERROR_MEMO"] = "1"
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from ecckd_amd import api, synthetic as syn
    ctx = api.Context(0); dev = ctx.device
    nwav, nlay = 7_200_000, 54
    p = syn.pressure_grid(nlay)
    wn_h, dwn_h = syn.wavenumber_grid(nwav)
    wn, dwn = torch.as_tensor(wn_h, device=dev), torch.as_tensor(dwn_h, device=dev)
    od = syn.optical_depth(torch, p, wn, syn.SEED_BASE + 1, nlines=32, column_scale=30.0, device=dev, chunk=1 << 20)
    bg = syn.optical_depth(torch, p, wn, syn.SEED_BASE + 1001, nlines=24, column_scale=3.0, zero_fraction=0.0, device=dev, chunk=1 << 20)
    key, _ = api.reorder_key_lw(ctx, p, api.idealised_temperature(p), wn, dwn, od, 0.5)
    rnk, _ = api.stable_argsort_bands(ctx, key, [0], [nwav - 1], want_ordered=False)
    gas = api.GasLW(ctx, p, syn.temperature_profile(p), wn, dwn, rnk, od, bg, "transmission", 0.0)
    rs = np.random.RandomState(1)
    for n, kind in CASES:
        for rep in range(REPS):
            if kind == "partition":
                cuts = np.sort(rs.uniform(0.02, 0.98, n - 1)) if n > 1 else np.array([])
                b1, b2 = np.concatenate([[0.0], cuts]), np.concatenate([cuts, [1.0]])
            else:
                length = 3000.0 if kind == "tiny" else 200000.0
                start = np.sort(rs.uniform(0.0, 0.9, n))
                b1, b2 = start, start + length / nwav
            gas.calc_error_batch(0, nwav, b1, b2)
    gas.close()
else:
    import sqlite3
    db = sqlite3.connect(sys.argv[2])
    rows = db.execute("select name,start,end,grid_y from kernels where name like '%k_interval_sums_fit_lw%' order by start").fetchall()
    cost = db.execute("select name,start,end from kernels where name like '%k_cost_lw%' order by start").fetchall()
    sweep = db.execute("select name,start,end from kernels where name like '%k_rt_lw_bb%' order by start").fetchall()
    assert len(rows) == len(CASES) * REPS, (len(rows), len(CASES) * REPS)
    i = 0
    for n, kind in CASES:
        d = [(rows[i + r][2] - rows[i + r][1]) / 1e3 for r in range(REPS)]
        c = [(cost[i + r][2] - cost[i + r][1]) / 1e3 for r in range(REPS)]
        s = [(sweep[i + r][2] - sweep[i + r][1]) / 1e3 for r in range(REPS)]
        print(f"{kind:10s} n={n:3d}: sums {np.mean(d[1:]):6.1f} us (first {d[0]:6.1f})   sweep {np.mean(s[1:]):7.1f}   cost {np.mean(c[1:]):6.1f}")
        i += REPS
