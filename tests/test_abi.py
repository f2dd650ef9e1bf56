"""The C-ABI shared library loads on a machine without a GPU and exports exactly what
include/dgp_hip.h declares; argument validation works without touching a device."""
import ctypes as C
import os
import re

import pytest

from discontinuum_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dgp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dgp_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 24
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/dgp_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes SIGNATURES and the header disagree"


def test_queries_and_argument_validation_without_a_device():
    lib = _lib.load()
    assert lib.dgp_version() >= 1
    assert lib.dgp_padded_n(300) == 384 and lib.dgp_padded_n(8192) == 8192
    assert lib.dgp_model_ntheta(_lib.MODEL_LOADEST, 3) == 11
    assert lib.dgp_model_ntheta(_lib.MODEL_LOADEST, 2) == 9
    assert lib.dgp_model_ntheta(_lib.MODEL_RATING, 2) == 16
    assert lib.dgp_model_ntheta(_lib.MODEL_RATING, 3) < 0
    h = C.c_void_p()
    assert lib.dgp_plan_create(7, _lib.F64, 100, 2, C.byref(h)) == -2  # DGP_E_MODEL
    assert lib.dgp_plan_create(_lib.MODEL_LOADEST, 5, 100, 2, C.byref(h)) == -1  # bad dtype
    assert b"dtype" in lib.dgp_last_error()
    assert lib.dgp_plan_create(_lib.MODEL_LOADEST, _lib.F64, 1000, 3, C.byref(h)) == 0
    nbytes = lib.dgp_plan_workspace_bytes(h)
    assert nbytes > 3 * 1024 * 1024 * 8  # three 1024^2 fp64 matrices
    assert lib.dgp_fit_step(h, None, None, None, None, None, None, None) == -3  # no workspace yet
    assert lib.dgp_plan_destroy(h) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.DGPLibraryError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under discontinuum_amd/ may reference it."""
    pkg = os.path.join(ROOT, "discontinuum_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
