"""The C-ABI library loads (no GPU needed) and exports every symbol include/asd_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "asd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(asd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from scaledreamer_amd import _lib

    names = _declared_symbols()
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/asd_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == names


def test_host_side_grid_meta_matches_oracle(oracle):
    from scaledreamer_amd import _lib

    for args in [(16, 2, 19, 16, 1.447269237440378), (4, 2, 19, 4, 4.0), (16, 2, 19, 16, 1.0)]:
        a = _lib.make_grid_meta(*args)
        b = oracle.grid_meta(*args)
        assert bytes(a) == bytes(b)
    assert _lib.make_grid_meta(16, 2, 19, 16, 1.447269237440378).n_params == 12_599_920


def test_struct_layouts_match_between_binding_and_oracle(oracle):
    from scaledreamer_amd import _lib

    for a, b in [(_lib.GridMeta, oracle.GridMeta), (_lib.FieldCfg, oracle.FieldCfg), (_lib.MarchCfg, oracle.MarchCfg)]:
        assert ctypes.sizeof(a) == ctypes.sizeof(b)
        assert [(n, t) for n, t in a._fields_] == [(n, t) for n, t in b._fields_]


def test_errors_are_reported_not_swallowed():
    from scaledreamer_amd import _lib

    m = _lib.GridMeta()
    assert _lib.lib().asd_grid_meta_init(ctypes.byref(m), 17, 2, 19, 16, 1.5) == 0
    assert b"levels" in _lib.lib().asd_last_error()
