"""The C-ABI library builds, loads and exports every symbol include/suave_hip.h declares (no compute calls)."""

import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "suave_hip.h")).read()
    return sorted(set(re.findall(r"\b(suave_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from suave_amd import lib

    path = lib.lib_path()
    assert os.path.exists(path), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    handle = ctypes.CDLL(path)
    names = _declared()
    assert "suave_elbo_backward" in names and "suave_decode_sample" in names
    for name in names:
        assert hasattr(handle, name), f"{name} declared in include/suave_hip.h but not exported"
    assert set(lib.EXPORTS) == set(names)


def test_ctypes_struct_sizes_match_header():
    from suave_amd import plan

    assert ctypes.sizeof(plan.ColT) == 48
    assert ctypes.sizeof(plan.LayerT) == 20
    assert ctypes.sizeof(plan.BatchT) == 56
    assert ctypes.sizeof(plan.NoiseT) == 16 + 16 * 8 + 16
    assert ctypes.sizeof(plan.ModelT) == 11 * 4 + (2 * 8 + 2) * 20 + 8 * 4
