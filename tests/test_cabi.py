"""CPU: the C-ABI shared library loads (no GPU needed for dlopen) and exports exactly the entry
points include/semstereo_hip.h declares, with the argument counts the ctypes binding uses.
No compute call is made here."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_entry_points():
    text = open(os.path.join(ROOT, "include", "semstereo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"(?:int|const char\*)\s+(ss_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        decls[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return decls


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    from semstereo_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        ge.build()
    return _lib.load()


def test_header_declares_something():
    d = declared_entry_points()
    assert len(d) >= 20 and "ss_gwc_volume_fwd" in d and "ss_conv3d_fwd" in d


def test_every_declared_symbol_is_exported(lib):
    for name in declared_entry_points():
        assert hasattr(lib, name), f"{name} is declared in include/semstereo_hip.h but not exported"


def test_binding_matches_header(lib):
    from semstereo_amd import _lib
    decl = declared_entry_points()
    assert set(_lib.EXPORTS) == set(decl)
    for name, argtypes in _lib._SIGNATURES.items():
        assert len(argtypes) == decl[name], (name, len(argtypes), decl[name])


def test_abi_version_and_status_strings(lib):
    from semstereo_amd import _lib
    assert lib.ss_abi_version() == _lib.ABI_VERSION
    assert lib.ss_status_string(0) == b"ok"
    assert b"invalid" in lib.ss_status_string(-1)
    assert b"not supported" in lib.ss_status_string(-2)


def test_invalid_arguments_are_rejected_before_any_launch(lib):
    """NULL pointers / bad sizes return SS_ERR_INVALID without touching a device."""
    assert lib.ss_gwc_volume_fwd(None, None, None, 1, 8, 4, 4, -2, 4, 2, 0, None) == -1
    assert lib.ss_conv3d_fwd(None, None, None, None, None, None, None, 1, 1, 1, 1, 1, 1, 3, 1, 0, None) == -1
    assert lib.ss_regression_topk_fwd(None, None, None, 1, 4, 2, 2, 2, None) == -1
