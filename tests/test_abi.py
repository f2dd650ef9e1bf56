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


def test_plan_options_without_a_device():
    """dgp_plan_set_option / dgp_plan_get_option are host-side bookkeeping: defaults, round trips, range checks, the
    float32-only refinement switch, unknown keys; and the diagnostic entry points reject bad arguments before any launch."""
    lib = _lib.load()
    h64, h32 = C.c_void_p(), C.c_void_p()
    assert lib.dgp_plan_create(_lib.MODEL_LOADEST, _lib.F64, 1000, 3, C.byref(h64)) == 0
    assert lib.dgp_plan_create(_lib.MODEL_RATING, _lib.F32, 1000, 2, C.byref(h32)) == 0
    v = C.c_int64()

    def get(h, key):
        assert lib.dgp_plan_get_option(h, key, C.byref(v)) == 0
        return v.value

    if not any(os.environ.get(k) for k in ("DGP_LAUUM64", "DGP_SYRK_SLOTS", "DGP_TRTRI_SMALL", "DGP_SYRK_ORDER", "DGP_LAUUM_ORDER", "DGP_NO_REFINE")):
        assert [get(h64, k) for k in range(6)] == [1000, 512, 1024, 0, 0, 0]  # refinement off for float64 plans
        assert get(h32, _lib.OPT_REFINE) == 1
    for key, val in ((_lib.OPT_LAUUM64_MAX_TILES, 0), (_lib.OPT_SYRK_SLOTS, 4), (_lib.OPT_TRTRI_SMALL, 0), (_lib.OPT_SYRK_ORDER, 8),
                     (_lib.OPT_LAUUM_ORDER, 4)):
        assert lib.dgp_plan_set_option(h64, key, val) == 0 and get(h64, key) == val
    assert lib.dgp_plan_set_option(h32, _lib.OPT_REFINE, 0) == 0 and get(h32, _lib.OPT_REFINE) == 0
    assert lib.dgp_plan_set_option(h64, _lib.OPT_REFINE, 1) == -1 and b"float32" in lib.dgp_last_error()
    assert lib.dgp_plan_set_option(h64, _lib.OPT_REFINE, 0) == 0
    assert lib.dgp_plan_set_option(h64, _lib.OPT_SYRK_SLOTS, 0) == -1
    assert lib.dgp_plan_set_option(h64, _lib.OPT_LAUUM_ORDER, 65) == -1
    assert get(h64, _lib.OPT_CHAIN_YIELD) == 1  # (default; DGP_CHAIN_YIELD=0 in the environment would say 0)
    assert lib.dgp_plan_set_option(h64, _lib.OPT_CHAIN_YIELD, 0) == 0 and get(h64, _lib.OPT_CHAIN_YIELD) == 0
    assert get(h64, _lib.OPT_FUSED_GRAD) == 0  # (default; DGP_FUSED_GRAD=1 in the environment would say 1)
    assert lib.dgp_plan_set_option(h64, _lib.OPT_FUSED_GRAD, 1) == 0 and get(h64, _lib.OPT_FUSED_GRAD) == 1
    assert get(h64, _lib.OPT_GROUP_GEMM) == 0 and lib.dgp_plan_set_option(h64, _lib.OPT_GROUP_GEMM, 1) == 0 and get(h64, _lib.OPT_GROUP_GEMM) == 1
    assert lib.dgp_plan_set_option(h64, 99, 1) == -1 and lib.dgp_plan_get_option(h64, 99, C.byref(v)) == -1
    assert lib.dgp_plan_set_option(None, 0, 1) == -1
    assert lib.dgp_debug_clock_probe(None, 16, 0.1, None) == -1
    assert lib.dgp_debug_tile_gemm(0, 1, 1, 1, None, 16, None, 16, 128, None, 128, 1, 1, 0, 0, None) == -1
    assert lib.dgp_plan_destroy(h64) == 0 and lib.dgp_plan_destroy(h32) == 0


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
