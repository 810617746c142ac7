"""The product's host-side partition search (ecckd_partition_*, row a13) against
 (1) the committed known answers of the reference's test_equipartition.cpp, and
 (2) the reference's own equipartition.cpp compiled into oracle/_ref, on identical error
     functions: same evaluation sequence, bit-identical bounds and errors.
CPU only: the search is host code; the GPU enters only through the batched callback."""
import json
import math
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "equipartition_known_answers.json")


def _flat_calls(ps):
    out = []
    for b1, b2, e in ps.calls:
        out.extend(zip(b1, b2, e))
    return out


def _exp_error_fn(npoints):
    values = np.exp(np.linspace(-2.0, 10.0, npoints))

    def calc_error(b1, b2):
        i1 = int(math.ceil(b1 * (npoints - 1)))
        i2 = int(math.floor(b2 * (npoints - 1)))
        s = float(np.cumsum(values[i1:i2 + 1])[-1])
        return abs(s - (i2 - i1 + 1) * values[(i1 + i2) // 2])
    return calc_error


def test_known_answers_of_test_equipartition():
    from ecckd_amd import api
    gold = json.load(open(GOLD))
    npoints = gold["npoints"]
    fn = _exp_error_fn(npoints)
    ps = api.PartitionSearch(lambda b1, b2: [fn(a, b) for a, b in zip(b1, b2)], resolution=1.0 / npoints,
                             partition_tolerance=0.001, partition_max_iterations=200,
                             line_search_max_iterations=15)
    for case in gold["cases"]:
        ps.lib.ecckd_partition_configure(ps.handle, 1.0 / npoints, 0.001, 200, 15, case["cubic"], 1)
        n0 = len(_flat_calls(ps))
        st, b, e = ps.equipartition_n(np.linspace(0.0, 1.0, 17))
        calls = _flat_calls(ps)[n0:]
        assert st == case["status"]
        assert b.tolist() == case["bounds"]          # bit-identical
        assert e.tolist() == case["error"]
        assert len(calls) == case["n_calls"]
        assert [[c[0], c[1]] for c in calls[:40]] == case["first_calls"]
        assert sum(c[1] - c[0] for c in calls) == pytest.approx(case["comp_cost"], rel=1e-12)


def _rough_error(seed, n=20000):
    """A deterministic, non-smooth interval error like the CKD one: |sum - n*mid| over a rough series."""
    rs = np.random.RandomState(seed)
    values = np.exp(np.linspace(-3.0, 6.0, n)) * (1.0 + 0.3 * rs.uniform(size=n))

    def calc_error(b1, b2):
        i1 = int(math.ceil(b1 * (n - 1)))
        i2 = max(i1, int(math.floor(b2 * (n - 1))))
        seg = values[i1:i2 + 1]
        return float(np.sqrt(np.cumsum((seg - seg.mean()) ** 2)[-1])) + 1e-9
    return calc_error, n


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("tol_frac", [0.08, 0.02])
def test_equipartition_e_matches_reference_build(oracle, seed, tol_frac):
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    from ecckd_amd import api
    fn, n = _rough_error(seed)
    target = fn(0.0, 1.0) * tol_frac
    ref = oracle.RefEquipartition(fn, resolution=1.0 / n, partition_tolerance=0.02, partition_max_iterations=60)
    rst, rb, re = ref.equipartition_e(target)
    ps = api.PartitionSearch(lambda b1, b2: [fn(a, b) for a, b in zip(b1, b2)], resolution=1.0 / n,
                             partition_tolerance=0.02, partition_max_iterations=60)
    st, b, e = ps.equipartition_e(target)
    assert st == rst
    assert b.tolist() == rb.tolist()
    assert e.tolist() == re.tolist()
    assert [(c[0], c[1]) for c in _flat_calls(ps)] == [(c[0], c[1]) for c in ref.calls]
    assert len(b) > 3


@pytest.mark.parametrize("ni", [2, 3, 7, 16])
def test_equipartition_n_matches_reference_build(oracle, ni):
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    from ecckd_amd import api
    fn, n = _rough_error(10 + ni, n=5000)
    start = np.sqrt(np.arange(ni + 1) / ni)           # find_g_points.cpp:1243-1245
    for frac_range in (True, False):
        ref = oracle.RefEquipartition(fn, resolution=1.0 / n, partition_tolerance=0.01,
                                      partition_max_iterations=30, minimize_frac_range=frac_range)
        rst, rb, re = ref.equipartition_n(start)
        ps = api.PartitionSearch(lambda b1, b2: [fn(a, b) for a, b in zip(b1, b2)], resolution=1.0 / n,
                                 partition_tolerance=0.01, partition_max_iterations=30,
                                 minimize_frac_range=frac_range)
        st, b, e = ps.equipartition_n(start)
        assert (st, b.tolist(), e.tolist()) == (rst, rb.tolist(), re.tolist())
        assert [(c[0], c[1]) for c in _flat_calls(ps)] == [(c[0], c[1]) for c in ref.calls]


def test_input_errors_and_single_interval(oracle):
    from ecckd_amd import api
    ps = api.PartitionSearch(lambda b1, b2: [1.0] * len(b1))
    st, b, e = ps.equipartition_n([0.0, 0.5, 0.5, 1.0])   # not strictly increasing
    assert st == 6                                         # EP_INPUT_ERROR
    st, b, e = ps.equipartition_e(1.0, 1.0, 0.0)
    assert st == 6
    # error of the whole domain already below target -> one interval (equipartition.cpp:596-605)
    ps2 = api.PartitionSearch(lambda b1, b2: [0.5 * (y - x) for x, y in zip(b1, b2)])
    st, b, e = ps2.equipartition_e(10.0)
    if oracle.ref_lib() is not None:
        ref = oracle.RefEquipartition(lambda x, y: 0.5 * (y - x))
        rst, rb, re = ref.equipartition_e(10.0)
        assert (st, b.tolist(), e.tolist()) == (rst, rb.tolist(), re.tolist())


def test_callback_failure_aborts_search():
    from ecckd_amd import api, EcckdError

    def boom(b1, b2):
        raise ValueError("bad interval")
    ps = api.PartitionSearch(boom)
    with pytest.raises(EcckdError) as e:
        ps.equipartition_e(1.0)
    assert e.value.code == 148


def test_decision_trace_reports_without_changing_the_search():
    """ecckd_partition_set_trace: the comparisons that steer the search are reported in order; results are the same bits."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from ecckd_amd import api
    x = np.exp(np.linspace(-2, 10, 200000))
    cum = np.concatenate([[0.0], np.cumsum(x)])

    def err(b1, b2):
        n = x.size
        return [float(cum[min(n, int(np.floor(b * n)) + 0)] - cum[int(np.ceil(a * n))]) / cum[-1] for a, b in zip(b1, b2)]

    plain = api.PartitionSearch(err, partition_max_iterations=30)
    traced = api.PartitionSearch(err, partition_max_iterations=30, trace=True)
    sa, ba, ea = plain.equipartition_e(0.07)
    sb, bb, eb = traced.equipartition_e(0.07)
    assert sa == sb and np.array_equal(ba, bb) and np.array_equal(ea, eb)
    assert plain.events is None
    kinds = [e[0] for e in traced.events]
    assert kinds.count("req") == len(traced.calls) and kinds.count("dec") > kinds.count("req")
    sites = {e[1] for e in traced.events if e[0] == "dec"}
    assert sites <= set(range(1, 36)) and {16, 17, 18, 27} <= sites          # equipartition_e and both next_bound searches
    for e in traced.events:
        if e[0] == "dec" and e[1] in (1, 3, 5, 6, 8, 9, 10, 12, 13, 15, 17, 19, 24, 26, 33, 35):     # the "<" sites
            assert e[4] == int(e[2] < e[3])
