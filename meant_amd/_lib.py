"""ctypes binding of libmeant_hip.so (the C ABI declared in include/meant_hip.h).

The product path has exactly one backend.  If the shared library is missing this module
raises at import: there is no CPU or PyTorch-eager fallback (by design -- see DESIGN.md).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MEANT_LIB_PATH: load another build of the same ABI (A/B measurements of one kernel against an older build)
LIB_PATH = os.environ.get("MEANT_LIB_PATH") or os.path.join(_HERE, "libmeant_hip.so")

F32, BF16 = 0, 1
RAW_F32, RAW_BF16, RAW_F64, RAW_U8 = 0, 1, 2, 3      # meant_raw_dtype (input pipeline)
EPI_NONE, EPI_GELU, EPI_RESIDUAL, EPI_SIGMOID = 0, 1, 2, 4

_p, _i, _i64, _f, _u64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64, C.c_size_t

# name -> (restype, argtypes); mirrors include/meant_hip.h one to one
SIGNATURES = {
    "meant_version": (_i, []),
    "meant_last_error": (C.c_char_p, []),
    "meant_num_cus": (_i, []),
    "meant_rmsnorm_fwd": (_i, [_p, _p, _p, _p, _i64, _i64, _f, _f, _u64, _i, _p]),
    "meant_rmsnorm_bwd_ws": (_sz, [_i64, _i64]),
    "meant_rmsnorm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i64, _f, _f, _u64, _p, _p, _i, _p, _sz, _p]),
    "meant_rmsnorm_pooled_ok": (_i, [_i64, _i64, _i64]),
    "meant_rmsnorm_fwd_pooled": (_i, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i, _i, _f, _f, _u64, _i, _p]),
    "meant_rmsnorm_bwd_pooled": (_i, [_p, _i, _p, _p, _p, _p, _p, _i64, _i64, _i64, _f, _f, _u64, _p, _i, _p, _i, _p, _sz, _p]),
    "meant_layernorm_fwd": (_i, [_p, _p, _p, _p, _p, _i64, _i64, _f, _i, _p]),
    "meant_layernorm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i, _p, _sz, _p]),
    "meant_linear_fwd": (_i, [_p, _i64, _p, _p, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i, _i, _p]),
    "meant_linear_bwd_dx": (_i, [_p, _i64, _p, _p, _i64, _i64, _i64, _i64, _i, _p]),
    "meant_linear_bwd_dw_ws": (_sz, [_i64, _i64, _i64, _i]),
    "meant_linear_bwd_dw": (_i, [_p, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _i, _p, _sz, _p]),
    "meant_set_option": (_i, [C.c_char_p, _i]),
    "meant_get_option": (_i, [C.c_char_p, C.POINTER(_i)]),
    "meant_route_count": (_i64, [C.c_char_p]),
    "meant_route_reset": (None, []),
    "meant_debug_nt_steals": (_i64, []),
    "meant_gemm_f32_strided": (_i, [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, _p, _p, _p, _f, _i, _p]),
    "meant_rotary_qk": (_i, [_p, _i64, _i64, _i, _i, _i, _p, _p, _p, _p, _i, _i, _p]),
    "meant_attn_ws": (_sz, [_i64, _i64, _i, _i, _i]),
    "meant_attn_fwd_ws": (_sz, [_i64, _i64, _i, _i, _i]),
    "meant_attn_drop_ws": (_sz, [_i64, _i64, _i, _i, _i]),
    "meant_attn_drop_fwd": (_i, [_p, _p, _p, _p, _i64, _i64, _i, _i, _f, _i, _f, _u64, _i, _p, _sz, _p]),
    "meant_attn_drop_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i, _i, _f, _i, _f, _u64, _i, _p, _sz, _p]),
    "meant_attn_fwd": (_i, [_p, _p, _p, _p, _i64, _i64, _i, _i, _f, _i, _i, _p, _sz, _p]),
    "meant_attn_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i, _i, _f, _i, _p, _p, _p, _p, _i, _i, _p, _sz, _p]),
    "meant_qkv_proj_fwd": (_i, [_p, _i64, _p, _p, _p, _i64, _i64, _i64, _i, _i, _i, _p, _p, _p, _p, _i, _p]),
    "meant_gather_rows": (_i, [_p, _p, _p, _p, _i64, _i64, _i, _p]),
    "meant_gather_rows_rot": (_i, [_p, _p, _p, _i64, _i64, _i, _i, _i, _p, _p, _p, _p, _i, _p]),
    "meant_group_scatter": (_i, [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i, _p]),
    "meant_attn_cls_fwd": (_i, [_p, _p, _i64, _p, _p, _i64, _i64, _i, _i, _f, _i, _p]),
    "meant_attn_cls_bwd": (_i, [_p, _p, _i64, _p, _i64, _p, _p, _p, _i64, _i64, _i, _i, _f, _i, _p]),
    "meant_token_shift": (_i, [_p, _p, _i64, _i64, _i64, _i64, _i, _i, _p]),
    "meant_dropout": (_i, [_p, _p, _i64, _f, _u64, _i, _p]),
    "meant_temporal_attn_fwd": (_i, [_p, _p, _p, _p, _i64, _i, _i, _i, _f, _i, _p]),
    "meant_temporal_attn_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _f, _i, _p]),
    "meant_patchify": (_i, [_p, _i, _p, _i64, _i, _i, _i, _i, _i, _p]),
    "meant_patchify_raw": (_i, [_p, _i, _f, _f, _p, _i64, _i, _i, _i, _i, _i, _p]),
    "meant_meanpool_fwd": (_i, [_p, _p, _i64, _i64, _i64, _i64, _i64, _i, _i, _p]),
    "meant_meanpool_bwd": (_i, [_p, _i64, _i64, _p, _i64, _i64, _i64, _i, _i, _p]),
    "meant_add_rowvec": (_i, [_p, _p, _p, _i64, _i64, _i64, _i, _p]),
    "meant_add_rowvec_bwd": (_i, [_p, _p, _i64, _i64, _i64, _i, _p]),
    "meant_gelu_bwd": (_i, [_p, _p, _p, _i64, _i, _p]),
    "meant_sigmoid_bwd": (_i, [_p, _p, _p, _i64, _i, _p]),
    "meant_geglu_fwd": (_i, [_p, _p, _i64, _i64, _i, _p]),
    "meant_geglu_bwd": (_i, [_p, _p, _p, _i64, _i64, _i, _p]),
    "meant_add": (_i, [_p, _p, _p, _i64, _i, _p]),
    "meant_cast": (_i, [_p, _i, _p, _i, _i64, _p]),
    "meant_transpose2d": (_i, [_p, _i, _p, _i, _i64, _i64, _p]),
    "meant_embedding_fwd": (_i, [_p, _p, _p, _i64, _i64, _i64, _i, _p]),
    "meant_embedding_bwd": (_i, [_p, _p, _p, _i64, _i64, _i64, _i, _p]),
    "meant_ce_probs": (_i, [_p, _p, _p, _p, _i64, _i, _p]),
    "meant_softmax_ce_fwd": (_i, [_p, _i64, _p, _i64, _i64, _i64, _p, _p, _i, _p]),
    "meant_softmax_ce_bwd": (_i, [_p, _i64, _p, _p, _i64, _i64, _i64, _p, _p, _i, _p]),
    "meant_sumsq_f32": (_i, [_p, _i64, _p, _p]),
    "meant_adamw_f32": (_i, [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _i64, _p, _f, _f, _p]),
    "meant_embedding_bwd_sorted": (_i, [_p, _p, _p, _p, _i64, _i64, _i64, _i, _p]),
    "meant_embedding_bwd_sorted_range": (_i, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i, _p]),
    "meant_rmsnorm_partial_fwd": (_i, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _f, _i, _p]),
    "meant_rmsnorm_partial_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _f, _i, _p, _sz, _p]),
    "meant_rmsnorm_stats": (_i, [_p, _p, _i64, _i64, _f, _i, _p]),
    "meant_rmsnorm_bwd_chain": (_i, [_p, _i, _p, _p, _p, _p, _p, _i64, _i64, _i64, _f, _f, _u64, _p, _p, _p, _f, _i64, _p, _p, _i, _p, _sz, _p]),
    "meant_colscale": (_i, [_p, _p, _p, _i64, _i64, _p]),
    "meant_colscale_bwd": (_i, [_p, _p, _p, _p, _p, _i64, _i64, _p]),
    "meant_linear_fwd_rowscale": (_i, [_p, _i64, _p, _p, _p, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i, _i, _p]),
    "meant_linear_bwd_dx_norm": (_i, [_p, _i64, _p, _p, _i64, _p, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _i, _p]),
}


class MeantLibraryMissing(ImportError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise MeantLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C meant_amd/csrc`).  meant_amd has no CPU / eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here == header and library out of sync
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def set_option(name: str, value: int) -> None:
    check(lib.meant_set_option(name.encode(), int(value)), "set_option")


def get_option(name: str) -> int:
    v = _i(0)
    check(lib.meant_get_option(name.encode(), C.byref(v)), "get_option")
    return v.value


def route_count(route: str) -> int:
    n = lib.meant_route_count(route.encode())
    if n < 0:
        raise KeyError(route)
    return n


def route_reset() -> None:
    lib.meant_route_reset()


class MeantHipError(RuntimeError):
    pass


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib.meant_last_error().decode("utf-8", "replace")
        raise MeantHipError(f"{what or 'libmeant_hip'} failed (status {rc}): {msg}")
