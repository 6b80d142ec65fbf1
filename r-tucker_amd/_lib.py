"""ctypes binding of ``librtucker_hip.so`` (C ABI: ``include/rtucker_hip.h``).

The library is built in-tree by ``__graft_entry__.build()`` /
``r-tucker_amd/csrc/build.sh`` into ``r-tucker_amd/lib/``.  Loading is lazy and
LOUD: if the shared object is missing or a symbol is absent, an exception is raised;
there is no Python/CPU fallback for any entry point.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# R_TUCKER_AMD_LIB: another build of the same ABI (A/B comparisons of kernel variants on one box)
LIB_PATH = os.environ.get("R_TUCKER_AMD_LIB") or os.path.join(_HERE, "lib", "librtucker_hip.so")

RTK_OK = 0
RTK_F32, RTK_BF16 = 0, 1
RTK_SCORE_SIGMOID = 1
RTK_SCORE_EXACT_F32 = 2
RTK_SCORE_SIGMOID_FAST = 4
RTK_SCORE_OUT_BF16 = 8
RTK_SCORE_KERNEL_CG, RTK_SCORE_KERNEL_WS, RTK_SCORE_KERNEL_V3 = 0x100, 0x200, 0x300   # kernel hints (A/B, tests)

_p, _i, _i64, _sz, _u = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_uint

# name -> (restype, argtypes); mirrors include/rtucker_hip.h one to one
SIGNATURES = {
    "rtk_version": (_i, []),
    "rtk_last_error_string": (C.c_char_p, []),
    "rtk_workspace_bytes": (_sz, [_i, _i64, _i64, _i, _i, _i]),
    "rtk_packed_query_bytes": (_sz, [_i, _i64, _i]),
    "rtk_read_error_flag": (_i, [_p, _p, C.POINTER(C.c_uint32)]),
    "rtk_query_vectors_f32": (_i, [_p, _i, _i, _i, _p, _i64, _p, _i64, _p, _p, _i64, _p, _p, _p, _sz, _p]),
    "rtk_score_f32": (_i, [_p, _i64, _i, _p, _i64, _p, _i64, _u, _p]),
    "rtk_score_packed_f32": (_i, [_p, _i64, _i, _p, _i64, _p, _i64, _u, _p]),
    "rtk_score_1vN_f32": (_i, [_p, _i, _i, _i, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _u, _p, _sz, _p]),
    "rtk_query_vectors_bf16": (_i, [_p, _i, _i, _i, _p, _i64, _p, _i64, _p, _p, _i64, _p, _p, _p, _sz, _p]),
    "rtk_score_packed_bf16": (_i, [_p, _i64, _i, _p, _i64, _p, _i64, _u, _p]),
    "rtk_score_1vN_bf16": (_i, [_p, _i, _i, _i, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _u, _p, _sz, _p]),
    "rtk_gemm_f32": (_i, [_p, _i, _i64, _p, _i, _i64, _p, _i64, _i64, _i64, _i64, _u, _p]),
    "rtk_gemm_f32_splitk_workspace_bytes": (_sz, [_i64, _i64, _i]),
    "rtk_gemm_f32_splitk": (_i, [_p, _i, _i64, _p, _i, _i64, _p, _i64, _i64, _i64, _i64, _i, _p, _sz, _p]),
    "rtk_gemm_sf16_splitk": (_i, [_p, _i, _i64, _p, _p, _i, _i64, _p, _p, _i64, _i64, _i64, _i64, _i, _p, _sz, _p]),
    "rtk_absmax_f32": (_i, [_p, _i64, _i64, _i64, _p, _p]),
    "rtk_query_bwd_workspace_bytes": (_sz, [_i64, _i, _i, _i]),
    "rtk_query_vectors_bwd_f32": (_i, [_p, _i, _i, _i, _p, _i64, _p, _i64, _p, _p, _i64, _p, _p, _p, _p, _p, _sz, _p]),
    "rtk_pack_query_vectors": (_i, [_p, _i64, _i, _i, _p, _p]),
    "rtk_relation_tables_bytes": (_sz, [_i64, _i, _i]),
    "rtk_relation_tables_workspace_bytes": (_sz, [_i, _i64, _i, _i, _i]),
    "rtk_from_tables_workspace_bytes": (_sz, [_i64, _i64]),
    "rtk_relation_tables_f32": (_i, [_p, _i, _i, _i, _p, _i64, _p, _p, _sz, _p]),
    "rtk_relation_tables_bf16": (_i, [_p, _i, _i, _i, _p, _i64, _p, _p, _sz, _p]),
    "rtk_query_vectors_from_tables_f32": (_i, [_p, _i64, _i, _i, _p, _i64, _p, _p, _i64, _p, _p, _p, _sz, _p]),
    "rtk_query_vectors_from_tables_bf16": (_i, [_p, _i64, _i, _i, _p, _i64, _p, _p, _i64, _p, _p, _p, _sz, _p]),
    "rtk_query_vectors_from_tables_part_f32": (_i, [_p, _i64, _i, _i, _p, _i64, _p, _p, _i64, _i, _i, _p, _p, _sz, _p]),
    "rtk_query_vectors_from_tables_part_bf16": (_i, [_p, _i64, _i, _i, _p, _i64, _p, _p, _i64, _i, _i, _p, _p, _sz, _p]),
    "rtk_sigmoid_grad_f32": (_i, [_p, _p, _p, _i64, _p]),
    "rtk_sigmoid_grad_rows_f32": (_i, [_p, _i64, _p, _i64, _p, _i64, _i64, _i64, _p]),
    "rtk_filtered_rank_f32": (_i, [_p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _p]),
    "rtk_bce_rows_f32": (_i, [_p, _i64, _i64, _i64, _p, _p, _p, C.c_float, _p, _p]),
    "rtk_bce_grad_f32": (_i, [_p, _i64, _i64, _i64, _p, _p, _p, C.c_float, _p, C.c_float, _p]),
    "rtk_rank_metrics_f64": (_i, [_p, _p, _i64, _p, _p]),
    "rtk_rank_metrics_scaled_f64": (_i, [_p, _p, _i64, C.c_double, _p, _p]),
    "rtk_score_bce_partials": (_i, []),
    "rtk_score_packed_bce_f32": (_i, [_p, _i64, _i, _p, _i64, _p, _i64, C.c_float, _p, _p]),
    "rtk_bce_patch_pos_f32": (_i, [_p, _i64, _i64, _i64, _p, _p, _p, C.c_float, _p, _p, _i, _p, _p]),
    "rtk_comm_unique_id": (_i, [_p]),
    "rtk_comm_init": (_i, [_i, _i, _p, C.POINTER(_p)]),
    "rtk_allgather_scores": (_i, [_p, _p, _sz, _p]),
    "rtk_comm_destroy": (_i, [_p]),
    "rtk_timer_create": (_i, [C.POINTER(C.c_void_p)]),
    "rtk_timer_arm": (_i, [_p]),
    "rtk_timer_elapsed_ms": (_i, [_p, C.POINTER(C.c_float)]),
    "rtk_timer_destroy": (_i, [_p]),
    "rtk_gram_factor_f64": (_i, [_p, _i64, _i, _i, C.c_double, C.c_double, _p, _p, _p]),
    "rtk_target_scores_f32": (_i, [_p, _i64, _i64, _i64, _i64, _p, _p, _p]),
    "rtk_filtered_rank_partial_f32": (_i, [_p, _i64, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _p]),
}

_lib = None


class RTuckerHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle with typed entry points."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RTuckerHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or r-tucker_amd/csrc/build.sh).  There is no CPU fallback for the scoring path.")
    import torch  # noqa: F401  -- torch's HIP runtime must be the one in the process before ours binds to it
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RTuckerHipError(f"{LIB_PATH} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != RTK_OK:
        msg = load().rtk_last_error_string().decode(errors="replace")
        # reference behaviour on bad shapes is a torch RuntimeError
        raise RuntimeError(f"{what} failed with status {rc}: {msg}")
