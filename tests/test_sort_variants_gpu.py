"""K3's A/B knobs are read once per process, so the driver-run suite only ever sees the default path.  Each variant - one sort
per band, the direct scatter, other tile sizes - sorts the same keys in a process of its own and must give the ranks
std::stable_sort gives (reorder_spectrum.cpp:262-300): ties, negative zero, negative keys, points outside every band, bands whose
lengths are not multiples of a tile."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import numpy as np, torch, sys
sys.path.insert(0, %r)
from ecckd_amd import api
ctx = api.Context(0)
rs = np.random.RandomState(3)
n = 200_003
key = rs.standard_normal(n) * np.exp(rs.uniform(-20, 5, n))
key[rs.randint(0, n, 5000)] = 0.0
key[rs.randint(0, n, 500)] = -0.0
key[rs.randint(0, n, 20000)] = np.round(key[rs.randint(0, n, 20000)], 2)          # many ties
cases = {"one band": ([0], [n - 1]), "three bands and gaps": ([17, 60_000, 150_000], [59_999, 120_500, n - 2]),
         "thirteen bands": (list(range(0, n - 13, n // 13))[:13], [b + n // 13 - 1 for b in list(range(0, n - 13, n // 13))[:13]])}
for name, (bb, be) in cases.items():
    want_rank = np.arange(n, dtype=np.int64)
    want_ord = np.arange(n, dtype=np.int64)
    for b, e in zip(bb, be):
        o = np.argsort(key[b:e + 1], kind="stable") + b
        want_ord[b:e + 1] = o
        want_rank[o] = np.arange(b, e + 1)
    rank, ordered = api.stable_argsort_bands(ctx, torch.as_tensor(key, device=ctx.device), bb, be)
    assert np.array_equal(rank.cpu().numpy(), want_rank), name
    assert np.array_equal(ordered.cpu().numpy(), want_ord), name
    rank2, none = api.stable_argsort_bands(ctx, torch.as_tensor(key, device=ctx.device), bb, be, want_ordered=False)
    assert none is None and np.array_equal(rank2.cpu().numpy(), want_rank), name
print("ranks ok")
""" % ROOT


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"ECCKD_SORT_PER_BAND": "1"}, {"ECCKD_SORT_DIRECT": "1"}, {"ECCKD_SORT_ITEMS": "8"},
                                 {"ECCKD_SORT_ITEMS": "20", "ECCKD_SORT_PER_BAND": "1"}, {"ECCKD_SORT_ITEMS": "12", "ECCKD_SORT_DIRECT": "1"}],
                         ids=lambda e: "+".join(f"{k[11:]}={v}" for k, v in e.items()) or "default")
def test_every_sort_variant_gives_the_stable_ranks(env):
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ranks ok" in r.stdout, r.stdout + r.stderr
