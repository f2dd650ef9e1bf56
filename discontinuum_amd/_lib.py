"""ctypes binding of ``libdgp_hip.so`` (C ABI declared in ``include/dgp_hip.h``).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C discontinuum_amd/csrc``.
There is no CPU fallback: if the shared object is missing or does not load, importing the symbols
raises ``DGPLibraryError`` and every engine entry point fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdgp_hip.so")

F64, F32 = 0, 1
MODEL_LOADEST, MODEL_RATING = 0, 1
OUT_NLL, OUT_QUAD, OUT_LOGDET, OUT_INFO, OUT_DTHETA, OUT_SUM_DR, OUT_DR_W0, OUT_SUM_DNOISE, OUT_LEN = 0, 1, 2, 3, 4, 28, 29, 31, 32
BUF_XT, BUF_A, BUF_T, BUF_S, BUF_Z, BUF_ALPHA = range(6)
BUF_SCAL = 7
OPT_LAUUM64_MAX_TILES, OPT_SYRK_SLOTS, OPT_TRTRI_SMALL, OPT_REFINE, OPT_SYRK_ORDER, OPT_LAUUM_ORDER, OPT_CHAIN_YIELD, OPT_FUSED_GRAD, OPT_GROUP_GEMM = range(9)
TIME_GRAM, TIME_POTRF, TIME_SYRK_SUM, TIME_SYRK_N, TIME_TRTRI, TIME_LAUUM, TIME_SOLVE, TIME_GRAD, TIME_SYRK_FLOP, TIME_COUNT = range(10)


class DGPLibraryError(RuntimeError):
    pass


class DGPError(RuntimeError):
    def __init__(self, code, where, msg):
        super().__init__(f"{where} failed with code {code}: {msg}")
        self.code = code


_vp, _i, _i64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
_dp = C.POINTER(C.c_double)

# name -> (restype, argtypes); mirrors include/dgp_hip.h one to one
SIGNATURES = {
    "dgp_version": (_i, []),
    "dgp_last_error": (C.c_char_p, []),
    "dgp_model_ntheta": (_i, [_i, _i]),
    "dgp_composite_define": (_i, [C.POINTER(_i), _i, C.POINTER(_i)]),
    "dgp_padded_n": (_i64, [_i64]),
    "dgp_plan_create": (_i, [_i, _i, _i64, _i, C.POINTER(_vp)]),
    "dgp_plan_destroy": (_i, [_vp]),
    "dgp_plan_workspace_bytes": (_sz, [_vp]),
    "dgp_plan_set_workspace": (_i, [_vp, _vp, _sz]),
    "dgp_plan_set_lookahead": (_i, [_vp, _i]),
    "dgp_plan_set_option": (_i, [_vp, _i, _i64]),
    "dgp_plan_get_option": (_i, [_vp, _i, C.POINTER(_i64)]),
    "dgp_plan_set_batch": (_i, [_vp, _i]),
    "dgp_plan_batch": (_i, [_vp]),
    "dgp_plan_set_site_sizes": (_i, [_vp, C.POINTER(C.c_int64), _vp]),
    "dgp_plan_set_dr_weights": (_i, [_vp, _vp]),
    "dgp_dist_last_error": (C.c_char_p, []),
    "dgp_dist_create": (_i, [_i, _i, _i64, _i, _i, _i, _i, C.POINTER(_vp)]),
    "dgp_dist_destroy": (_i, [_vp]),
    "dgp_dist_padded_n": (_i64, [_vp]),
    "dgp_dist_groups": (_i, [_vp]),
    "dgp_dist_slab_columns": (_i64, [_vp]),
    "dgp_dist_workspace_bytes": (_sz, [_vp]),
    "dgp_dist_panel_elems": (_sz, [_vp, _i]),
    "dgp_dist_set_workspace": (_i, [_vp, _vp, _sz]),
    "dgp_dist_set_inputs": (_i, [_vp, _vp, _vp]),
    "dgp_dist_gram": (_i, [_vp, _dp, _vp, _vp]),
    "dgp_dist_factor": (_i, [_vp, _i, _vp, _vp]),
    "dgp_dist_update": (_i, [_vp, _i, _vp, _i, _i, _vp]),
    "dgp_dist_invert": (_i, [_vp, _i, _vp, _vp]),
    "dgp_dist_status": (_i, [_vp, _vp, _vp]),
    "dgp_dist_solve_partial": (_i, [_vp, _vp, _vp, _vp]),
    "dgp_dist_alpha_partial": (_i, [_vp, _vp, _vp, _vp]),
    "dgp_dist_residual": (_i, [_vp, _dp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dgp_dist_pack_inverse": (_i, [_vp, _i, _vp, _vp]),
    "dgp_dist_product": (_i, [_vp, _i, _vp, _vp]),
    "dgp_dist_grad_partial": (_i, [_vp, _dp, _vp, _vp, _vp, _vp]),
    "dgp_dist_slab": (_i, [_vp, _i, C.POINTER(_vp)]),
    "dgp_plan_buffer": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_i64)]),
    "dgp_plan_site_stride_bytes": (_sz, [_vp]),
    "dgp_set_inputs": (_i, [_vp, _vp, _vp]),
    "dgp_fit_step": (_i, [_vp, _dp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dgp_factorize": (_i, [_vp, _dp, _vp, _vp, _vp, _vp]),
    "dgp_predict_workspace_bytes": (_sz, [_vp, _i64]),
    "dgp_predict": (_i, [_vp, _dp, _vp, _i64, _vp, _sz, _vp, _vp, _vp]),
    "dgp_posterior_cov": (_i, [_vp, _dp, _vp, _i64, _vp, _sz, _vp, _vp, _vp]),
    "dgp_sample_draws": (_i, [_i, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "dgp_mean_vjp_workspace_bytes": (_sz, [_vp, _i64]),
    "dgp_predict_mean": (_i, [_vp, _dp, _vp, _i64, _vp, _sz, _vp, _vp]),
    "dgp_mean_vjp": (_i, [_vp, _dp, _vp, _i64, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "dgp_plan_set_timing": (_i, [_vp, _i]),
    "dgp_plan_get_timing": (_i, [_vp, _dp]),
    "dgp_stage_gram": (_i, [_vp, _dp, _vp, _vp]),
    "dgp_stage_potrf": (_i, [_vp, _vp]),
    "dgp_stage_trtri": (_i, [_vp, _vp]),
    "dgp_stage_lauum": (_i, [_vp, _vp]),
    "dgp_stage_solve": (_i, [_vp, _vp, _vp]),
    "dgp_stage_grad": (_i, [_vp, _dp, _vp, _vp]),
    "dgp_cross_gram": (_i, [_vp, _dp, _vp, _i64, _vp, _vp, _vp]),
    "dgp_debug_clock_probe": (_i, [_vp, _i, C.c_double, _vp]),
    "dgp_debug_tile_gemm": (_i, [_i, _i, _i, _i, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _i, _i, _i, _i, _vp]),
}

_lib = None


def load():
    """Load (once) and return the ctypes handle with typed signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DGPLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C discontinuum_amd/csrc` (there is no CPU fallback)"
        )
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise DGPLibraryError(f"could not load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, where):
    if rc != 0:
        lib = load()
        err = lib.dgp_dist_last_error if where.startswith("dgp_dist_") else lib.dgp_last_error
        raise DGPError(rc, where, err().decode("utf-8", "replace"))
