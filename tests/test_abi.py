"""The C-ABI library loads and exports every symbol include/ecckd_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ecckd_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ecckd_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_declared_symbols():
    import __graft_entry__ as g
    g.build()
    from ecckd_amd import _lib
    lib = ctypes.CDLL(_lib.library_path())
    names = _declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ecckd_hip.h but not exported"
    # the ctypes table used by the bindings covers the header
    assert set(names) == set(_lib.SIGNATURES), sorted(set(names) ^ set(_lib.SIGNATURES))


def test_host_only_entry_points():
    import numpy as np
    from ecckd_amd import api
    p = np.array([1.0, 1.0e5, 1.0e3])
    t = api.idealised_temperature(p)
    assert np.allclose(t, [173.15, 288.15, 242.15], rtol=1e-14)
    wn = np.linspace(0.0, 100.0, 101)
    iband, bb, be = api.band_ranges(wn, [10.0, 50.0], [50.0, 90.0])
    assert (bb.tolist(), be.tolist()) == ([10, 50], [49, 90])  # last band closed on the right
    assert iband[9] == -1 and iband[10] == 0 and iband[50] == 1 and iband[90] == 1 and iband[91] == -1


def test_no_cpu_fallback():
    """Without a HIP device the product must fail loudly, never compute on the CPU."""
    import torch
    from ecckd_amd import api, EcckdError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(EcckdError):
        api.Context(0)


def test_product_does_not_import_oracle():
    import glob
    for f in glob.glob(os.path.join(ROOT, "ecckd_amd", "**", "*"), recursive=True):
        if os.path.isfile(f) and f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
            src = open(f).read()
            # no import, include, link or dlopen of anything under oracle/ (comments may cite it)
            for needle in ("pyoracle", "ecckd_oracle.h", "libecckd_oracle", "libequipartition_ref", "orc_"):
                assert needle not in src, (f, needle)


def test_struct_mirrors_match_the_header():
    """Every ctypes mirror of a struct of include/ecckd_hip.h against the C compiler's layout: size and the offset of
    every field (a field added to the header only, or in another order, shows up here and not as a wrong answer on the GPU)."""
    import ctypes as C
    import shutil
    import subprocess
    import tempfile
    from ecckd_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mirrors = {"ecckd_band_options": _lib.BandOptions, "ecckd_gas_search": _lib.GasSearch, "ecckd_opt_gas": _lib.OptGas, "ecckd_opt_model": _lib.OptModel,
               "ecckd_opt_scene": _lib.OptScene, "ecckd_opt_config": _lib.OptConfig}
    lines = []
    for cname, cls in mirrors.items():
        lines.append(f' printf("%zu", sizeof({cname}));\n')
        lines += [f' printf(" %zu", offsetof({cname}, {name}));\n' for name, _ in cls._fields_]
        lines.append(' printf("\\n");\n')
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "ecckd_hip.h"\nint main(void) {\n' + "".join(lines) + " return 0; }\n"
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "layout.c"), "w") as f:
            f.write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), os.path.join(d, "layout.c"), "-o", os.path.join(d, "layout")])
        rows = subprocess.check_output([os.path.join(d, "layout")]).decode().strip().split("\n")
    for (cname, cls), row in zip(mirrors.items(), rows):
        out = [int(x) for x in row.split()]
        assert out[0] == C.sizeof(cls), cname
        assert out[1:] == [getattr(cls, name).offset for name, _ in cls._fields_], cname
