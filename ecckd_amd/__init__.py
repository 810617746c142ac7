"""ecckd_amd - MI355X (gfx950) implementation of ecCKD's spectral-integration hot path.

The product is the C-ABI shared library ``libecckd_hip.so`` (``include/ecckd_hip.h``),
built from the hand-written HIP sources in ``ecckd_amd/csrc``.  This Python package is a
thin ctypes binding over that ABI used by the tests and ``bench.py``; PyTorch appears only
as the owner of device memory.  There is no CPU fallback: importing ``ecckd_amd.api`` on a
machine without the built library, or creating a context without a gfx950 device, fails.
"""
from ._lib import EcckdError, load_library, library_path  # noqa: F401

__all__ = ["EcckdError", "load_library", "library_path"]
