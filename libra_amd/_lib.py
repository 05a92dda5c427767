"""ctypes loader for the gfx950 C-ABI library (include/libra_hip.h).

There is NO fallback: if ``libra_amd/lib/liblibra_hip.so`` is missing or a symbol is absent the
import fails loudly.  Build it with ``make -C libra_amd/csrc`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liblibra_hip.so")

OK, ERR_SHAPE, ERR_ALIGN, ERR_LAUNCH = 0, -1, -2, -3

_P, _I64, _F, _I = C.c_void_p, C.c_int64, C.c_float, C.c_int

GEMM_MULTI_MAX = 12          # include/libra_hip.h LIBRA_GEMM_MULTI_MAX


class GemmProblem(C.Structure):
    """include/libra_hip.h `libra_gemm_problem`: one GEMM of a multi-problem launch (libra_gemm_bf16_multi)."""
    _fields_ = [("A", _P), ("lda", _I64), ("B", _P), ("ldb", _I64), ("C", _P), ("ldc", _I64),
                ("M", _I64), ("N", _I64), ("K", _I64),
                ("bias", _P), ("resid", _P), ("ldr", _I64), ("aux", _P), ("ldaux", _I64), ("preact", _P), ("ldpre", _I64),
                ("alpha", _F), ("flags", C.c_int32), ("alpha_cols", _I64),
                ("a_rows", _P), ("a_phys_rows", _I64), ("c_rows", _P), ("splitk", _I64), ("slab", _P), ("wait_on", _I64)]


# name -> argtypes, exactly the prototypes of include/libra_hip.h
SIGNATURES = {
    "libra_hip_abi_version": [],
    "libra_gemm_bf16_nt": [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _P, _P, _I64, _P, _I64, _P, _I64,
                           _F, _I64, _I, _P],
    "libra_gemm_bf16_nt_routed": [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _P, _P, _I64, _P, _I64, _P, _I64,
                                  _F, _I64, _I, _P, _I64, _P, _P],
    "libra_gemm_bf16_nt_tile": [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _P, _P, _I64, _P, _I64, _P, _I64,
                                _F, _I64, _I, _P, _I64, _P, _I, _P],
    "libra_gemm_bf16_nt_grouped": [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _I64, _F, _I64, _I, _P, _I64, _P, _I, _P],
    "libra_gemm_bf16_multi": [C.POINTER(GemmProblem), _I64, _P, _P],
    "libra_gemm_swiglu_skinny": [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _P, _I64, _P],
    "libra_gemm_splitk_plan": [_I64, _I64, _I64],
    "libra_gemm_splitk_workspace_bytes": [_I64, _I64, _I64],
    "libra_gemm_bf16_nt_splitk": [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _I64, _I, _P, C.c_size_t, _P],
    "libra_gemm_bf16_nt_splitk_routed": [_P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, _I64, _I, _P, _I64, _P, _I64, _P, _P,
                                         C.c_size_t, _P],
    "libra_colsum_workspace_bytes": [_I64, _I64],
    "libra_colsum_bf16": [_P, _I64, _I64, _I64, _P, _P, C.c_size_t, _P],
    "libra_layernorm_bwd_workspace_bytes": [_I64, _I64],
    "libra_layernorm_fwd": [_P, _P, _P, _P, _P, _P, _I64, _I64, _F, _P],
    "libra_layernorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _I64, _I64, _P],
    "libra_patch_im2col": [_P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _P],
    "libra_patch_col2im": [_P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _P],
    "libra_vit_embed_ln": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _F, _P],
    "libra_vit_attn_fwd": [_P, _I64, _P, _I64, _P, _P, _I64, _I64, _I64, _F, _P],
    "libra_vit_attn_delta": [_P, _P, _P, _I64, _P, _I64, _I64, _I64, _P],
    "libra_vit_attn_bwd": [_P, _I64, _P, _I64, _P, _P, _P, _I64, _I64, _I64, _I64, _F, _P],
    "libra_feature_select": [_P, _I64, _P, _I64, _I64, _I64, _P],
    "libra_feature_select_bwd": [_P, _P, _P, _I64, _I64, _I64, _I64, _P],
    "libra_lfq_encode": [_P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _P],
    "libra_rmsnorm_routed_fwd": [_P, _I64, _P, _P, _P, _P, _I64, _P, _I64, _I64, _F, _P],
    "libra_rope_bridge": [_P, _I64, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _I64, _I64, _I64, _I64, _P],
    "libra_rope_bridge_pos": [_P, _I64, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _I64, _I64, _P, _I64, _I64, _P],
    "libra_rope_bridge_pos_append": [_P, _I64, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _I64, _I64, _P, _I64, _I64,
                                     _P, _P, _P, _P, _I64, _I64, _P, _P],
    "libra_kv_cache_append": [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _P, _P, _P, _I64, _I64, _P, _I64, _I64, _P],
    "libra_bridge_attn_decode_workspace_bytes": [_I64, _I64],
    "libra_bridge_attn_decode": [_P, _I64, _P, _P, _P, _P, _I64, _I64, _P, _I64, _P, _P, _P, _P, _I64, _I64, _I64, _F, _P,
                                 C.c_size_t, _P],
    "libra_bridge_attn_fwd": [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _P, _P, _P, _I64, _P, _P, _I64, _I64, _I64, _F, _P],
    "libra_bridge_attn_bwd": [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _P, _I64, _P, _I64, _P, _P, _P, _P, _P, _I64,
                              _P, _P, _P, _P, _I64, _I64, _I64, _I64, _F, _P, _P],
    "libra_swiglu": [_P, _P, _I64, _P, _I64, _I64, _I64, _P],
    "libra_gather_rows": [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _I64, _P],
    "libra_copy_rows": [_P, _I64, _I64, _P, _I64, _P, _I64, _I64, _P],
    "libra_ce_rows": [_P, _I64, _I64, _P, _I64, _P, _I64, _P],
    "libra_ce_rows_bwd": [_P, _I64, _I64, _P, _P, _I64, _F, _F, _P, _P, _I64, _I64, _P],
    "libra_rmsnorm_routed_bwd": [_P, _I64, _P, _I64, _P, _P, _P, _P, _P, _I64, _P, _I64, _I64, _I64, _P],
    "libra_rmsnorm_wgrad_workspace_bytes": [_I64, _I64],
    "libra_rmsnorm_routed_wgrad": [_P, _I64, _P, _I64, _P, _P, _P, _P, _P, C.c_size_t, _I64, _I64, _P, _I64, _P],
    "libra_swiglu_bwd": [_P, _I64, _P, _P, _I64, _P, _P, _I64, _I64, _I64, _P],
    "libra_rope_bridge_bwd": [_P, _P, _P, _P, _P, _I64, _P, _P, _I64, _P, _I64, _P, _I64, _P, _P, _P, _P, _P, _P, _I64,
                              _I64, _I64, _I64, _P, _I64, _P],
    "libra_f32_to_bf16": [_P, _P, _I64, _P],
    "libra_add_bf16": [_P, _P, _P, _I64, _P],
    "libra_lfq_codes": [_P, _P, _I64, _I64, _I64, _I64, _P],
    "libra_groupnorm_workspace_bytes": [_I64, _I64, _I64],
    "libra_groupnorm_affine": [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _F, _P, C.c_size_t, _P],
    "libra_conv_gather": [_P, _P, _I64, _P, _P, _I, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _F, _F, _P],
    "libra_softmax_rows": [_P, _I64, _I64, _I64, _F, _P],
    "libra_resample_h_u8": [_P, _I64, _I64, _I64, _I64, _I, _I, _I, _P, _P, _I64, _P, _I64, _I64, _I64, _P],
    "libra_resample_v_u8_norm": [_P, _I64, _I64, _P, _P, _I64, _I64, _I64, _I64, _P, _P, _I64, _I64, _P],
    "libra_adamw_step": [_P, _P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _F, _F, _F, _P, _F, _P],
    "libra_rank_outer_wgrad_workspace_bytes": [_I64, _I64, _I64],
    "libra_rank_outer_wgrad": [_P, _I64, _P, _I64, _I64, _P, _P, _P, _I64, _I, _I64, _I64, _P, C.c_size_t, _P],
    "libra_sumsq_workspace_bytes": [_I64],
    "libra_sumsq_bf16": [_P, _I64, _P, _I, _P, C.c_size_t, _P],
    "libra_stream_create_cu_reserved": [C.c_int32, _P, _P],
    "libra_stream_destroy": [_P],
    "libra_set_cu_budget": [C.c_int32],
    "libra_get_cu_count": [],
    "libra_stream_create_cu_mask": [_P, C.c_int32, _P],
    "libra_debug_cu_map": [_P, C.c_int32, C.c_int32, _P],
}

ABI_VERSION = 13


class LibraHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU / PyTorch fallback exists). "
            "Build it with `make -C libra_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.argtypes = argtypes
        fn.restype = C.c_size_t if name.endswith("_workspace_bytes") else C.c_int
    v = lib.libra_hip_abi_version()
    if v != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {v}, host expects {ABI_VERSION}; rebuild it")
    return lib


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = load()
    return _lib


def check(rc: int, what: str):
    """Map C-ABI error codes onto the exception types the reference raises for the same conditions
    (ValueError for shape mismatches, modeling_libra.py:374-403)."""
    if rc == OK:
        return
    if rc == ERR_SHAPE:
        raise ValueError(f"{what}: unsupported or inconsistent shape")
    if rc == ERR_ALIGN:
        raise ValueError(f"{what}: null / misaligned pointer or leading dimension")
    raise LibraHipError(f"{what}: HIP launch failed (rc={rc})")
