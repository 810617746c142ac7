"""Audit of the g-point search at full width (VERDICT r02, next #1a): WHY do the device chain and the CPU chain end at different
g-point maps at 4.2e6 points when every interval error agrees to 1e-10?

The same host search (csrc/partition_search.cpp, bit-identical to the reference's equipartition.cpp for equal errors:
tests/test_partition_search.py) is run twice over the SAME reordered spectrum of 2^22 points of the headline generator
(synthetic.optical_depth_lines):
  A  over the device's interval errors (ecckd_calc_error_batch), the gas prepared on the device;
  B  over the oracle's (oracle_find_g.c: CkdEquipartition::calc_error, find_g_points.cpp:291-405), the gas prepared by the
     oracle on the host.
Both record their event streams: every request (bounds, index range find_g_points.cpp:282-287, error) and every comparison
that steers the search (site, lhs, rhs, outcome).  The streams are walked in lockstep:

  * while the index ranges of the requests are identical the two errors are the same interval evaluated twice, from two
    INDEPENDENT preparations of the gas (device K4 / oracle).  Asserted: (i) over the device's OWN rows the oracle reproduces
    the device's errors within the stated tolerance, rtol 1e-9 + 1e-10 K/d (sampled requests); (ii) over its own rows within
    ten times that - the relative differences reach 2e-5, but only on intervals whose error is ~1e-6 K/d (nothing absorbs
    there): in K/d the two never differ by more than 1e-10;
  * the FIRST difference must be a knife edge: either a comparison whose margin |lhs - rhs| is below 1e-8 of its operands, or
    an index rounding (lower = ceil(b (n - 1)), upper = floor(b (n - 1))) where the two bounds agree to < 1e-3 of an index step
    and straddle an integer;
  * after it the streams stay aligned as long as the comparisons' outcomes agree; there the index ranges may differ by the few
    points the knife edge moved and the errors agree as far as those points allow (1e-4); the first comparison whose outcome
    differs must again be a knife edge relative to THAT perturbation;
  * the margin histogram of run A's comparisons is printed: how many decisions of a search sit within 1e-9 ... 1e-3 of a flip.
"""
import math
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import os
# The cases (points, layers, heating-rate tolerance in K/d, expected status of the device-driven search, the multiple of the
# stated tolerance rtol 1e-9 + 1e-10 K/d within which the oracle over the DEVICE's rows must reproduce the device's errors):
#  * 2^22 points, 30 layers, 0.0161 (the fsck tolerance of test/do_all_lw.sh:59-60): converges with 33 g points after 1 639
#    requests, ~4 min with the two oracle-driven searches - the k_rt_lw_bb_mirror<30,...> sweep;
#  * 54 layers - the <54,...> kernels the bench times - in the regime the headline bench's searches end in: the search runs
#    into its 60 iterations (status 2, "Maximum iterations reached").  In the suite: 2^19 points at 0.04 K/d (2 946 requests,
#    ~1 min; device- and oracle-driven searches part at an index rounding 1 783 requests in - two bounds 2.6e-4 index steps
#    apart on either side of an integer - and end at the same 18 g points).  Run by hand and committed
#    (profiles/r04_decision_trace_54layers_2e20_status2.txt, ECCKD_AUDIT_CASE=1048576,54,0.0161,2, 4.4 min): 2^20 points at the
#    fsck tolerance, 15 243 requests, 43 207 intervals, 38 721 comparisons - device- and oracle-driven searches take the SAME
#    decisions throughout and end at the same 37 g points, the oracle over the device's rows reproduces the errors to 3e-12.
#    (Round 3 at 2^22 points, 30 layers and 0.013 K/d, by hand: profiles/r03_decision_trace_tol0.013.txt.)
# ECCKD_AUDIT_CASE=nwav,nlay,tol[,status] runs another case by hand.
CASES = [
    pytest.param(1 << 22, 30, 0.0161, 0, 1.0, id="2^22 points, 30 layers, converging"),
    pytest.param(1 << 19, 54, 0.04, 2, 1.0, id="2^19 points, 54 layers, 60 iterations reached"),
]
if os.environ.get("ECCKD_AUDIT_CASE"):
    _c = os.environ["ECCKD_AUDIT_CASE"].split(",")
    CASES = [pytest.param(int(_c[0]), int(_c[1]), float(_c[2]), int(_c[3]) if len(_c) > 3 else None, 1.0, id="by hand")]
TOL_TOL = 0.01          # tolerance_tolerance of test/find_g_points_lw.sh
MAX_IT = 60


def _index_range(b1, b2, n):
    """find_g_points.cpp:282-317"""
    lo = int(math.ceil(b1 * (n - 1)))
    hi = int(math.floor(b2 * (n - 1)))
    if hi < lo:
        hi = lo
    return lo, hi


def _margin(lhs, rhs):
    s = max(abs(lhs), abs(rhs))
    return abs(lhs - rhs) / s if s > 0 else 0.0


@pytest.mark.parametrize("NWAV,NLAY,TOL,expected_status,same_rows_multiple", CASES)
def test_first_divergence_of_device_and_oracle_searches_is_a_knife_edge(ctx, oracle, monkeypatch, capsys, NWAV, NLAY, TOL, expected_status,
                                                                        same_rows_multiple):
    from ecckd_amd import api, synthetic as syn
    monkeypatch.setenv("ECCKD_NO_ERROR_MEMO", "1")                  # every request evaluated, none answered from the memo
    dev = ctx.device
    n = NWAV
    p = syn.pressure_grid(NLAY)
    wn_h, dwn_h = syn.wavenumber_grid(n)
    wn = torch.as_tensor(wn_h, device=dev)
    od = syn.optical_depth_lines(torch, p, wn, syn.SEED_BASE + 1, nlines=12000, device=dev)
    bg = syn.optical_depth_lines(torch, p, wn, syn.SEED_BASE + 1001, nlines=4000, column_scale=3.0, zero_fraction=0.0, nclusters=5,
                                 device=dev)
    t_ideal = api.idealised_temperature(p)
    t_hl = syn.temperature_profile(p)
    dwn = torch.as_tensor(dwn_h, device=dev)
    key, _ = api.reorder_key_lw(ctx, p, t_ideal, wn, dwn, od, 0.5)
    rank, _ = api.stable_argsort_bands(ctx, key, [0], [n - 1], want_ordered=False)
    gas = api.GasLW(ctx, p, t_hl, wn, dwn, rank, od, bg, "transmission", flux_weight=0.0)

    # ---- the oracle's own preparation of the same ordering (find_g_points.cpp:891-1150), on the host ----
    rank_h = rank.cpu().numpy().astype(np.int64)
    ireorder = np.empty(n, dtype=np.int64)
    ireorder[rank_h] = np.arange(n)
    od_host = od.cpu().numpy().astype(np.float64)
    bg_host = bg.cpu().numpy().astype(np.float64)
    od_s, bg_s = od_host[:, ireorder], bg_host[:, ireorder]
    del od, bg
    wn_s, dwn_s = wn_h[ireorder], dwn_h[ireorder]
    planck = oracle.planck_function(t_hl, wn_s, dwn_s)
    surf_planck = oracle.planck_function([t_hl[-1]], wn_s, dwn_s)[0]
    tot = bg_s + od_s
    fdn, fup = oracle.radiative_transfer_lw(planck, tot, np.ones(n), surf_planck)
    del tot
    hr = oracle.heating_rate(p, fdn, fup)
    fds, fut = fdn[-1].copy(), fup[0].copy()
    del fdn, fup
    metric = oracle.metric("transmission", od_s)
    del od_s
    eq = oracle.CkdEquipartitionLW("transmission", 0.0, oracle.layer_weight(p, 0.0), p, np.ones(n), surf_planck, fds, fut,
                                   planck, bg_s, metric, hr)
    pool = ThreadPoolExecutor(max_workers=16)

    seen = {}

    def oracle_one(ab):
        # (an interval asked for again is the same number again: the oracle is deterministic; keeps the run to minutes)
        r = _index_range(ab[0], ab[1], n)
        if r not in seen:
            seen[r] = eq.calc_error(ab[0], ab[1])
        return seen[r]

    def oracle_errors(b1, b2):
        # the intervals of one call are independent (the reference runs them in OpenMP threads, equipartition.h:100-104)
        return list(pool.map(oracle_one, zip(b1, b2)))

    def device_errors(b1, b2):
        return list(gas.calc_error_batch(0, n, np.asarray(b1), np.asarray(b2)))

    def run(fn):
        ps = api.PartitionSearch(fn, resolution=1.0 / n, partition_tolerance=TOL_TOL, partition_max_iterations=MAX_IT, trace=True)
        st, b, e = ps.equipartition_e(TOL)
        return st, b, e, ps.events

    st_a, b_a, e_a, ev_a = run(device_errors)
    st_b, b_b, e_b, ev_b = run(oracle_errors)

    # ---- run C: the oracle chain proper - its OWN ordering too (reorder_spectrum.cpp:111-300 by the oracle) ----
    od_h = od_host          # (nlay, n) float64, unsorted
    key_o, _, st_key = oracle.reorder_key(p, t_ideal, wn_h, dwn_h, od_h, None, 0.5)
    assert st_key == 0
    _, _, rank_o = oracle.stable_argsort_bands(wn_h, key_o, [0.0], [3260.0])
    rank_o = np.asarray(rank_o, dtype=np.int64)
    moved = np.nonzero(rank_o != rank_h)[0]
    key_d = key.cpu().numpy()
    key_rel = float(np.max(np.abs(key_d - key_o) / np.maximum(np.abs(key_o), 1e-3)))
    # a wavenumber whose rank differs sits among neighbours whose keys agree to the keys' own tolerance: a knife edge of the sort
    moved_margin = 0.0
    if moved.size:
        inv_o = np.empty(n, dtype=np.int64); inv_o[rank_o] = np.arange(n)
        for j in moved[:2000]:
            lo_r, hi_r = sorted((int(rank_o[j]), int(rank_h[j])))
            ks = key_o[inv_o[lo_r:hi_r + 1]]
            moved_margin = max(moved_margin, float((ks.max() - ks.min()) / max(abs(ks).max(), 1e-3)))
    ir_o = np.empty(n, dtype=np.int64); ir_o[rank_o] = np.arange(n)
    od_s2, bg_s2 = od_h[:, ir_o], bg_host[:, ir_o]
    wn_s2, dwn_s2 = wn_h[ir_o], dwn_h[ir_o]
    planck2 = oracle.planck_function(t_hl, wn_s2, dwn_s2)
    surf2 = oracle.planck_function([t_hl[-1]], wn_s2, dwn_s2)[0]
    fdn2, fup2 = oracle.radiative_transfer_lw(planck2, bg_s2 + od_s2, np.ones(n), surf2)
    hr2 = oracle.heating_rate(p, fdn2, fup2)
    fds2, fut2 = fdn2[-1].copy(), fup2[0].copy()
    del fdn2, fup2
    metric2 = oracle.metric("transmission", od_s2)
    del od_s2
    eq2 = oracle.CkdEquipartitionLW("transmission", 0.0, oracle.layer_weight(p, 0.0), p, np.ones(n), surf2, fds2, fut2, planck2, bg_s2,
                                    metric2, hr2)
    seen2 = {}

    def oracle2_one(ab):
        r = _index_range(ab[0], ab[1], n)
        if r not in seen2:
            seen2[r] = eq2.calc_error(ab[0], ab[1])
        return seen2[r]

    st_c, b_c, e_c, ev_c = run(lambda b1, b2: list(pool.map(oracle2_one, zip(b1, b2))))
    del eq2, planck2, hr2, metric2, bg_s2
    idx = lambda b: [_index_range(b[i], b[i + 1], n) for i in range(len(b) - 1)]
    same_boundaries_c = len(b_a) == len(b_c) and idx(b_a) == idx(b_c)
    # the g point of every wavenumber (SingleGasData::store_g_points): how many wavenumbers end up in another g point
    gp_differ = None
    if len(b_a) == len(b_c):
        upper_a = np.array([r[1] for r in idx(b_a)]); upper_c = np.array([r[1] for r in idx(b_c)])
        gp_a = np.searchsorted(upper_a, rank_h); gp_c = np.searchsorted(upper_c, rank_o)
        gp_differ = int((gp_a != gp_c).sum())

    # ---- where the two preparations differ, and the oracle over the DEVICE's rows on a sample of run A's requests ----
    d_hr, d_planck = gas.view("hr"), gas.view("planck_hl")
    d_fds, d_fut = gas.view("flux_dn_surf")[0].copy(), gas.view("flux_up_toa")[0].copy()
    hr_scale = np.abs(hr).max(axis=1, keepdims=True)
    # the boundary fluxes against the scale they live on, the column's surface Planck flux: on a near-transparent column the
    # downwelling flux at the surface is a sum of emissions of ~1e-8 of that scale, each the difference 1 - exp(-x) of nearly
    # equal numbers, and its RELATIVE difference between two correct evaluations reaches 4e-8 although nothing differs by more
    # than a rounding of the scale
    prep_diff = dict(planck=float(np.max(np.abs(d_planck - planck) / np.maximum(np.abs(planck), 1e-300))),
                     hr_rel_to_layer_max=float(np.max(np.abs(d_hr - hr) / hr_scale)),
                     hr_layer_sums_rel=float(np.max(np.abs(d_hr.sum(1) - hr.sum(1)) / np.maximum(np.abs(hr.sum(1)), 1e-300))),
                     flux_dn_surf=float(np.max(np.abs(d_fds - fds) / np.maximum(np.abs(fds), 1e-300))),
                     flux_dn_surf_rel_to_surface_planck=float(np.max(np.abs(d_fds - fds) / np.maximum(surf_planck, 1e-300))),
                     flux_up_toa=float(np.max(np.abs(d_fut - fut) / np.maximum(np.abs(fut), 1e-300))),
                     flux_up_toa_rel_to_surface_planck=float(np.max(np.abs(d_fut - fut) / np.maximum(surf_planck, 1e-300))))
    eq_dev = oracle.CkdEquipartitionLW("transmission", 0.0, oracle.layer_weight(p, 0.0), p, np.ones(n), d_planck[-1].copy(), d_fds, d_fut,
                                       d_planck, bg_s, metric, d_hr)
    reqs = [(ev[1][k], ev[2][k], ev[3][k]) for ev in ev_a if ev[0] == "req" for k in range(len(ev[1]))]
    rs = np.random.RandomState(3)
    sample = [reqs[i] for i in rs.choice(len(reqs), min(48, len(reqs)), replace=False)]
    same_rows = list(pool.map(lambda r: eq_dev.calc_error(r[0], r[1]), sample))
    # (stated tolerance of the interval errors: rtol 1e-9 + 1e-10 K/d, tests/test_find_g_gpu.py)
    worst_same_rows = max(abs(a - r[2]) / (abs(r[2]) + 0.1) for a, r in zip(same_rows, sample))
    worst_same_rows_tol = max(abs(a - r[2]) / (1e-9 * abs(r[2]) + 1e-10) for a, r in zip(same_rows, sample))
    del d_hr, d_planck, eq_dev
    gas.close()
    pool.shutdown()

    # ---- lockstep walk ----
    nreq = sum(1 for ev in ev_a if ev[0] == "req")
    ndec = sum(1 for ev in ev_a if ev[0] == "dec")
    identical = True            # index ranges identical so far
    first_index_flip = None     # (event position, interval, xA, xB)
    first_outcome_flip = None   # (event position, site, (lhsA, rhsA), (lhsB, rhsB))
    worst_err_identical = 0.0   # max relative error difference over requests with identical index ranges
    worst_abs_identical = 0.0   # ... in K/d
    worst_tol_identical = 0.0   # ... as a multiple of the stated tolerance rtol 1e-9 + 1e-10 K/d
    worst_err_shifted = 0.0     # ... over aligned requests after the index ranges began to differ
    max_shift = 0               # largest index difference of aligned requests
    aligned_req = aligned_dec = 0
    for pos, (a, b) in enumerate(zip(ev_a, ev_b)):
        if a[0] != b[0]:
            first_outcome_flip = first_outcome_flip or (pos, -1, None, None)       # the streams' structure differs
            break
        if a[0] == "req":
            if len(a[1]) != len(b[1]):
                first_outcome_flip = first_outcome_flip or (pos, -1, None, None)
                break
            aligned_req += 1
            for k in range(len(a[1])):
                ra, rb = _index_range(a[1][k], a[2][k], n), _index_range(b[1][k], b[2][k], n)
                rel = abs(a[3][k] - b[3][k]) / max(abs(b[3][k]), 1e-300)
                if ra == rb and identical:
                    worst_err_identical = max(worst_err_identical, rel)
                    worst_abs_identical = max(worst_abs_identical, abs(a[3][k] - b[3][k]))
                    worst_tol_identical = max(worst_tol_identical, abs(a[3][k] - b[3][k]) / (1e-9 * abs(b[3][k]) + 1e-10))
                elif ra == rb:
                    worst_err_shifted = max(worst_err_shifted, rel)
                else:
                    if identical:
                        identical = False
                        # which end moved, and how far apart the two (continuous) bounds are in index units
                        end = 1 if ra[0] != rb[0] else 2
                        xa, xb = a[end][k] * (n - 1), b[end][k] * (n - 1)
                        first_index_flip = (pos, k, xa, xb)
                    max_shift = max(max_shift, abs(ra[0] - rb[0]), abs(ra[1] - rb[1]))
                    worst_err_shifted = max(worst_err_shifted, rel)
        else:
            aligned_dec += 1
            if a[1] != b[1] or a[4] != b[4]:
                first_outcome_flip = (pos, a[1], (a[2], a[3]), (b[2], b[3]))
                break
            if identical:
                # same interval errors to 1e-9 -> the operands of the comparison agree (bounds are interpolated from errors)
                # (differences of two nearly equal errors lose digits: measured against the errors' own scale)
                sc = max(abs(a[2]), abs(a[3]), TOL)
                assert abs(a[2] - b[2]) <= 1e-3 * sc and abs(a[3] - b[3]) <= 1e-3 * sc, (pos, a, b)

    # ---- the margin histogram of run A's decisions ----
    margins = np.array([_margin(ev[2], ev[3]) for ev in ev_a if ev[0] == "dec" and not (ev[2] == 0.0 and ev[3] == 0.0)])
    edges = [0.0, 1e-12, 1e-9, 1e-8, 1e-6, 1e-4, 1e-2, 1.0 + 1e-12]
    hist = np.histogram(np.minimum(margins, 1.0), bins=edges)[0]
    # index roundings: distance of b (n - 1) to the integer it is rounded at, in index units, for every request of run A
    fr = []
    for ev in ev_a:
        if ev[0] == "req":
            for k in range(len(ev[1])):
                for x in (ev[1][k] * (n - 1), ev[2][k] * (n - 1)):
                    fr.append(min(x - math.floor(x), math.ceil(x) - x))
    fr = np.array(fr)
    near = [(fr < t).sum() for t in (1e-6, 1e-4, 1e-2)]
    # the line search (equipartition.cpp:162-196, site 1): how often is the first trial of a line search accepted?  (What a
    # speculative evaluation of the NEXT halving beside the current one could save depends on this.)
    ls_trials = [ev[4] for ev in ev_a if ev[0] == "dec" and ev[1] == 1]
    n_ls = first_ok = 0
    run_len = 0
    for taken in ls_trials:
        run_len += 1
        if taken:
            n_ls += 1
            first_ok += run_len == 1
            run_len = 0
    n_ls_failed = 1 if run_len else 0
    with capsys.disabled():
        print("\n[decision trace] line searches of run A: %d accepted (%d at the first trial), %d trials in all, %d ran out of trials"
              % (n_ls, first_ok, len(ls_trials), n_ls_failed))
        print("[decision trace] n = %d points, ng = %d (device) / %d (oracle), status %d / %d" % (n, len(e_a), len(e_b), st_a, st_b))
        print("[decision trace] run A: %d requests (%d intervals), %d comparisons; aligned with run B: %d requests, %d comparisons"
              % (nreq, len(fr) // 2, ndec, aligned_req, aligned_dec))
        print("[decision trace] margin |lhs - rhs| / max of run A's comparisons: " +
              ", ".join("%s..%s: %d" % ("%g" % edges[i], "%g" % min(edges[i + 1], 1.0), hist[i]) for i in range(len(hist))))
        print("[decision trace] index roundings of run A within 1e-6 / 1e-4 / 1e-2 of an integer: %d / %d / %d of %d" % (*near, fr.size))
        print("[decision trace] the two preparations (device K4 vs oracle), max rel. difference: " +
              ", ".join("%s %.2e" % kv for kv in prep_diff.items()))
        print("[decision trace] interval errors, oracle over the DEVICE's rows vs device: max difference %.2e of (|error| + 0.1 K/d) = %.2f x "
              "the stated tolerance (rtol 1e-9 + 1e-10 K/d) (%d sampled requests)" % (worst_same_rows, worst_same_rows_tol, len(sample)))
        print("[decision trace] interval errors, device vs oracle over its OWN rows: max rel. difference %.2e, max abs. %.2e K/d, "
              "%.2f x the stated tolerance (rtol 1e-9 + 1e-10 K/d) over the requests with identical index ranges; %.2e rel. after the "
              "ranges began to differ (largest shift %d points)"
              % (worst_err_identical, worst_abs_identical, worst_tol_identical, worst_err_shifted, max_shift))
        print("[decision trace] reorder: sorting keys device vs oracle max rel. difference %.2e; %d of %d wavenumbers get another rank, "
              "each inside a run of keys that agree to %.2e" % (key_rel, moved.size, n, moved_margin))
        print("[decision trace] run C (oracle chain with its OWN ordering): ng = %d, status %d, g-point index boundaries %s run A's; "
              "%s wavenumbers fall into another g point" % (len(e_c), st_c, "identical to" if same_boundaries_c else "DIFFERENT from", gp_differ))
        print("[decision trace] first index rounding that differs: %s" % (first_index_flip,))
        print("[decision trace] first comparison whose outcome differs: %s" % (first_outcome_flip,))
        print("[decision trace] final bounds: max |difference| = %.3g index steps" %
              (np.max(np.abs(b_a - b_b)) * (n - 1) if len(b_a) == len(b_b) else float("nan")))

    # ---- the assertions ----
    assert nreq > 200 and aligned_req > 50
    if expected_status is not None:
        assert st_a == expected_status, "the device-driven search ended with status %d" % st_a
    # the two preparations of the gas (device K4, oracle) agree to rounding on the scale of every quantity
    assert prep_diff["planck"] <= 1e-11 and prep_diff["hr_rel_to_layer_max"] <= 1e-11
    assert prep_diff["flux_dn_surf_rel_to_surface_planck"] <= 2e-12 and prep_diff["flux_up_toa_rel_to_surface_planck"] <= 2e-12
    # same rows in, same error out: the stated tolerance (rtol 1e-9 + 1e-10 K/d) holds on the converging searches (3e-11
    # measured); the non-converging one asks for intervals of a few points of the most opaque end of the spectrum, where the
    # transmission fit takes -ln(1 - mean) of a mean within 1e-16 of one (average_optical_depth.cpp:43-133, find_g_points.cpp:
    # 64-68): 2.4e-9 measured there
    assert worst_same_rows_tol <= same_rows_multiple, "%.2f x the stated tolerance" % worst_same_rows_tol
    # two independent preparations: the large RELATIVE differences (4e-5 ... 1e-4) sit on intervals whose error is ~1e-6 K/d
    # (nothing absorbs there); in K/d the two never differ by more than 2e-8 (north_star's heating-rate tolerance: 1e-6 K/d)
    assert worst_abs_identical <= 2e-8
    if first_index_flip is not None:
        pos, k, xa, xb = first_index_flip
        # a knife edge: the two bounds agree to a small fraction of an index step yet round to different integers
        assert abs(xa - xb) < 1e-3, first_index_flip
        assert math.floor(min(xa, xb)) != math.floor(max(xa, xb)) or math.ceil(min(xa, xb)) != math.ceil(max(xa, xb))
        # from there on the intervals differ by a few points: errors still agree to what a few points of 4e6 can change
        assert max_shift <= 64 and worst_err_shifted <= 1e-2
    if first_outcome_flip is not None and first_outcome_flip[2] is not None:
        pos, site, (la, ra), (lb, rb) = first_outcome_flip
        scale = max(abs(la), abs(ra), 1e-300)
        # either the comparison sat closer to its threshold than the perturbation the streams had accumulated by then, or it is
        # the direct consequence of the index rounding that flipped a few events before it (measured at tolerance 0.013: the
        # rounding moves one search's interval by ONE point, the other search repeats its interval, and three events later the
        # stall test `frac_error == prev_frac_error` of equipartition_2 (equipartition.cpp:276-280) is true for one of them only)
        pert = max(worst_err_shifted, worst_err_identical, 1e-9)          # (the errors of the two runs differ by this much)
        near_rounding = first_index_flip is not None and 0 <= pos - first_index_flip[0] <= 10
        assert near_rounding or abs(la - ra) <= 50.0 * pert * scale, (first_outcome_flip, first_index_flip, pert)
    # the reorder's own knife edges: keys agree to their tolerance (tests/test_reorder_gpu.py: 1e-9), a rank moves only inside a
    # run of keys that close; and the oracle chain with its own ordering ends at the same g points up to those wavenumbers
    assert key_rel <= 1e-9 and moved_margin <= 1e-9
    assert abs(len(e_a) - len(e_c)) <= 1
    if same_boundaries_c:
        assert gp_differ <= moved.size
    if first_index_flip is None and first_outcome_flip is None:
        assert len(b_a) == len(b_b) and idx(b_a) == idx(b_b)
    else:
        # the PATHS of the two searches part at the knife edge; the answers stay together: the same number of g points (one
        # more or less at worst), and - measured, not asserted - the same index boundaries at tolerance 0.013
        assert abs(len(e_a) - len(e_b)) <= 1
