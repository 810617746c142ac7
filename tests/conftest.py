import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "oracle") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def ctx():
    """One ecckd context on cuda:0 for the whole GPU session (fails loudly without the .so)."""
    from ecckd_amd import api
    c = api.Context(0)
    yield c
    c.close()


def make_lw_case(nwav, nlay=54, seed=1, dtype="float32", lo=0.0, hi=3260.0, nlines=48, column_scale=30.0):
    """Synthetic LW column: (pressure_hl, wavenumber, d_wavenumber, optical_depth)."""
    from ecckd_amd import synthetic as syn
    p = syn.pressure_grid(nlay)
    wn, dwn = syn.wavenumber_grid(nwav, lo, hi)
    od = syn.optical_depth(np, p, wn, syn.SEED_BASE + seed, nlines=nlines, dtype=dtype,
                           column_scale=column_scale, lo=lo, hi=hi)
    return p, wn, dwn, od
