"""CPU: the C-ABI shared library loads and exports every symbol include/asis_hip.h declares; the ctypes
binding covers all of them; calls that need no GPU behave."""
import ctypes
import os
import re

import pytest

from adaptersis_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "asis_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(asis_[a-z0-9_]+)\s*\(", src)))


def test_library_is_built_in_tree():
    assert os.path.exists(_lib.LIB_PATH), "run `python -m adaptersis_amd.build` (or __graft_entry__.build())"
    assert os.path.dirname(_lib.LIB_PATH).startswith(ROOT)


def test_every_declared_symbol_is_exported_and_bound():
    names = header_functions()
    assert len(names) > 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/asis_hip.h but not exported: {missing}"
    unbound = [n for n in names if n not in _lib.SIGNATURES and n != "asis_last_error"]
    assert not unbound, f"exported but not bound in adaptersis_amd/_lib.py: {unbound}"
    extra = [n for n in _lib.SIGNATURES if n not in names]
    assert not extra, f"bound but not declared in the header: {extra}"


def test_no_gpu_needed_for_identity_calls():
    lib = _lib.lib()
    assert lib.asis_version() >= 100
    assert lib.asis_gemm_tiles_m(129) == 2
    assert lib.asis_wgrad_splits(1 << 20, 64, 576) >= 1
    assert isinstance(lib.asis_last_error(), bytes)


def test_argument_errors_are_reported_without_a_gpu():
    lib = _lib.lib()
    d = _lib.GemmDesc()
    rc = lib.asis_gemm(None, ctypes.byref(d))
    assert rc == -1 and b"null operand" in lib.asis_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc, "asis_gemm")


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libasis_hip.so")
    with pytest.raises(_lib.AsisError, match="no fallback"):
        _lib.lib()
