"""The C-ABI library loads on a CPU-only host and exports every symbol include/arco_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_header_symbols():
    from arco_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "arco_hip.h")).read()
    declared = set(re.findall(r"\b(arco_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)


def test_missing_library_fails_loudly(monkeypatch):
    from arco_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libarco_hip.so")
    import pytest
    with pytest.raises(RuntimeError):
        _lib.load()
