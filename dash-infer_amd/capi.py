"""ctypes binding of include/dashinfer_hip.h (plumbing; every signature mirrors the header).

Fails loudly when lib/libdashinfer_hip.so is missing - there is no fallback path.
"""
import ctypes as C
import os
import re

from . import LIB_PATH, REPO_ROOT

# enums (include/dashinfer_hip.h)
F32, F16, BF16 = 0, 1, 2
KV_NONE, KV_I8, KV_U4 = 0, 1, 2
ACT = {None: 0, "none": 0, "tanh": 1, "gelu_erf": 2, "gelu_tanh": 3, "relu": 4, "silu": 5, "sigmoid": 6}
KV = {"none": KV_NONE, "i8": KV_I8, "u4": KV_U4}
SUCCESS, PARAM_ERROR, MEMORY_ERROR, RUNTIME_ERROR, EXCEED_LIMIT_ERROR, INVALID_CALL_ERROR = 0, 2, 4, 5, 7, 8
SA_SUCCESS, SA_HIP_ERROR, SA_RUNTIME_ERROR, SA_PARAM_ERROR, SA_EXCEED_LIMIT_ERROR = 0, 1, 2, 3, 4

vp, sz, i32, f32 = C.c_void_p, C.c_size_t, C.c_int, C.c_float

_SIGS = {
    "dihip_version": (C.c_char_p, []),
    "dihip_last_error": (C.c_char_p, []),
    "dihip_device_info": (i32, [C.POINTER(i32), C.POINTER(i32), C.c_char_p, sz]),
    "dihip_gemm_lowp_packed_weight_bytes": (sz, [i32, i32, i32]),
    "dihip_gemm_lowp_packed_sz_bytes": (sz, [i32, i32, i32]),
    "dihip_gemm_lowp_pack": (i32, [vp, i32, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "dihip_gemm_lowp_workspace_bytes": (sz, [i32, i32, i32, i32, i32]),
    "dihip_gemm_lowp_sync_bytes": (sz, []),
    "dihip_gemm_a16w8": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, sz, vp, i32]),
    "dihip_gemm_a16w4": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, sz, vp, i32]),
    "dihip_fused_norm_gemm": (i32, [vp, i32, vp, vp, f32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, sz, vp, i32]),
    "dihip_fused_norm_swiglu": (i32, [vp, i32, vp, vp, f32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp, i32]),
    "dihip_fused_gemm_addto": (i32, [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp, i32]),
    "dihip_fused_norm_swiglu_ex": (i32, [vp, i32, vp, vp, f32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp, i32, i32]),
    "dihip_fused_gemm_addto_ex": (i32, [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp, i32, i32]),
    "dihip_fused_gemm_addto_norm": (i32, [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp, i32, i32, vp, f32, vp, i32]),
    "dihip_prenorm_gemm": (i32, [vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, sz, vp, i32]),
    "dihip_prenorm_swiglu": (i32, [vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp, i32, i32]),
    "dihip_fused_gemm_addto_prenorm": (i32, [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp, i32, i32, vp, f32, vp, i32, vp, sz, vp]),
    "dihip_prenorm_gemm_rowsq": (i32, [vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, sz, vp, i32, vp, i32, f32]),
    "dihip_prenorm_swiglu_rowsq": (i32, [vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp, i32, i32, vp, i32, f32]),
    "dihip_prenorm_rowsq_supported": (i32, [i32, i32, i32, i32, i32, i32, i32, i32]),
    "dihip_rowsq_bytes": (sz, []),
    "dihip_gemm_prefill_tail_parts": (i32, [i32, i32, i32, i32, i32, i32]),
    "dihip_gemm_lowp_prefers_frag": (i32, [i32, i32, i32, i32, i32, i32]),
    "dihip_moe_route": (i32, [vp, vp, i32, i32, i32, vp, vp, i32]),
    "dihip_rmsnorm_rows": (i32, [vp, vp, vp, vp, f32, i32, i32, i32]),
    "dihip_moe_route_ep": (i32, [vp, vp, i32, i32, i32, vp, vp, i32, i32, i32]),
    "dihip_calc_expert": (i32, [vp, vp, vp, vp, i32, i32, i32]),
    "dihip_moe_shared_combine": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32]),
    "dihip_moe_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "dihip_moe_experts": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, sz, i32]),
    "dihip_moe_experts_ex": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, sz, i32, i32]),
    "dihip_moe_route_grouped": (i32, [vp, vp, i32, i32, i32, vp, vp, i32, i32, i32, i32, i32, vp, sz]),
    "dihip_moe_combine": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32]),
    "dihip_moe_router_gate": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32]),
    "dihip_act_frag_bytes": (sz, [i32, i32]),
    "dihip_act_to_frag": (i32, [vp, vp, vp, i32, i32, i32]),
    "dihip_act_from_frag": (i32, [vp, vp, vp, i32, i32, i32]),
    "dihip_span_bytes": (sz, [i32, i32, i32, i32, i32]),
    "dihip_kv_append": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32]),
    "dihip_rope_kv_append": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32]),
    "dihip_kv_context_copy": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32]),
    "dihip_kv_prefix_gather": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32]),
    "dihip_span_gather": (i32, [vp, vp, vp, i32, sz]),
    "dihip_span_scatter": (i32, [vp, vp, vp, i32, sz]),
    "dihip_span_attn_create_handle": (i32, [C.POINTER(vp), i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(i32), i32]),
    "dihip_span_attn_destroy_handle": (i32, [vp]),
    "dihip_span_attn_host_workspace_bytes": (i32, [C.POINTER(sz), vp]),
    "dihip_span_attn_device_workspace_bytes": (i32, [C.POINTER(sz), vp]),
    "dihip_span_attn_run": (i32, [vp, vp, vp, vp, vp, sz, vp, sz, f32, vp, vp]),
    "dihip_span_attn_decode_workspace_bytes": (sz, [i32, i32, i32, i32, i32]),
    "dihip_span_attn_decode": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, sz, vp]),
    "dihip_span_attn_decode_ex": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, sz, vp, i32]),
    "dihip_span_attn_decode_sync": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, sz, vp, sz, i32]),
    "dihip_span_attn_sync_bytes": (sz, [i32, i32]),
    "dihip_rope_table": (i32, [vp, vp, vp, i32, i32]),
    "dihip_span_attn_fused_workspace_bytes": (sz, [i32, i32, i32, i32, i32]),
    "dihip_span_attn_decode_fused": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, sz]),
    "dihip_span_attn_decode_fused_sync": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, sz, vp, sz]),
    "dihip_span_attn_decode_step": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, sz, vp, sz, i32]),
    "dihip_span_attn_merge_partials": (i32, [vp, vp, vp, i32, i32, i32, i32]),
    "dihip_decode_attn_block_supported": (i32, [i32] * 10),
    "dihip_decode_attn_block_sync_bytes": (sz, [i32, i32, i32]),
    "dihip_decode_attn_block_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "dihip_decode_attn_block_status_async": (i32, [vp, vp, vp]),
    "dihip_decode_attn_block_reset": (i32, [vp, vp, sz]),
    "dihip_decode_attn_block_prepare": (i32, [vp, vp, sz, i32, i32, i32, i32]),
    "dihip_decode_mlp_block_supported": (i32, [i32] * 6),
    "dihip_decode_mlp_block_sync_bytes": (sz, [i32]),
    "dihip_decode_mlp_block": (i32, [vp, i32, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz]),
    "dihip_decode_attn_block": (i32, [vp, i32, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32,
                                      i32, i32, f32, vp, sz, vp, sz]),
    "dihip_prefill_attn": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, i32]),
    "dihip_rmsnorm": (i32, [vp, vp, vp, vp, f32, i32, i32, i32]),
    "dihip_rope_qk": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32]),
    "dihip_binary_add": (i32, [vp, vp, vp, vp, sz, i32]),
    "dihip_silu_mul": (i32, [vp, vp, vp, vp, sz, i32]),
    "dihip_binary_mul": (i32, [vp, vp, vp, vp, sz, i32]),
    "dihip_unary": (i32, [vp, vp, vp, sz, i32, i32]),
    "dihip_unary_glu": (i32, [vp, vp, vp, sz, sz, i32, i32]),
    "dihip_embedding_ft": (i32, [vp, vp, vp, vp, i32, i32, i32, i32]),
    "dihip_cast_to_f32": (i32, [vp, vp, vp, sz, i32]),
    "dihip_dense_packed_weight_bytes": (sz, [i32, i32]),
    "dihip_dense_pack": (i32, [vp, vp, i32, i32, i32, vp]),
    "dihip_dense_workspace_bytes": (sz, [i32, i32, i32]),
    "dihip_gemm_a16w16": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, sz, vp, i32]),
    "dihip_lm_head": (i32, [vp, vp, vp, vp, f32, vp, i32, i32, i32, vp, sz, vp, i32]),
    "dihip_argmax": (i32, [vp, vp, vp, i32, i32, vp, sz]),
    "dihip_argmax_advance": (i32, [vp, vp, vp, i32, i32, vp, sz, vp, vp]),
    "dihip_sample": (i32, [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dihip_sample_rows": (i32, [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32]),
    "dihip_logits_processor_workspace_bytes": (sz, [i32, i32]),
    "dihip_logits_processor": (i32, [vp, vp, i32, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz]),
    "dihip_logprobs_workspace_bytes": (sz, [i32, i32, i32]),
    "dihip_logprobs": (i32, [vp, vp, i32, i32, vp, i32, i32, vp, vp, vp, vp, sz]),
    "dihip_logits_processor_rows": (i32, [vp, vp, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz]),
    "dihip_logprobs_records": (i32, [vp, vp, i32, i32, vp, i32, i32, vp, vp, i32, i32, vp, sz]),
    "dihip_argmax_partial": (i32, [vp, vp, vp, i32, i32, i32, vp, sz]),
    "dihip_argmax_merge": (i32, [vp, vp, vp, i32, i32]),
    "dihip_embedding": (i32, [vp, vp, vp, vp, i32, i32, i32]),
    "dihip_embedding_v": (i32, [vp, vp, vp, vp, i32, i32, i32, i32]),
    "dihip_increment_u32": (i32, [vp, vp, i32]),
    "dihip_prefetch": (i32, [vp, C.POINTER(vp), C.POINTER(sz), i32, i32]),
    "dihip_rccl_unique_id": (i32, [vp]),
    "dihip_rccl_comm_init_rank": (i32, [C.POINTER(vp), i32, vp, i32]),
    "dihip_rccl_comm_destroy": (i32, [vp]),
    "dihip_allreduce_sum": (i32, [vp, vp, vp, vp, sz, i32]),
    "dihip_allgather_bytes": (i32, [vp, vp, vp, vp, sz]),
    "dihip_allgather_rows": (i32, [vp, vp, vp, vp, vp, i32, sz, i32]),
    "dihip_gather_rows_transpose": (i32, [vp, vp, vp, i32, i32, sz]),
    "dihip_p2p_ar_buffer_bytes": (sz, []),
    "dihip_p2p_ar_max_bytes": (sz, []),
    "dihip_p2p_ar_alloc": (i32, [C.POINTER(vp)]),
    "dihip_p2p_ar_free": (i32, [vp]),
    "dihip_ipc_get_handle": (i32, [vp, vp]),
    "dihip_ipc_open_handle": (i32, [vp, C.POINTER(vp)]),
    "dihip_ipc_close_handle": (i32, [vp]),
    "dihip_p2p_ar_create": (i32, [C.POINTER(vp), i32, i32, C.POINTER(vp)]),
    "dihip_p2p_ar_destroy": (i32, [vp]),
    "dihip_p2p_ar_set_timeout": (i32, [vp, C.c_ulonglong, i32]),
    "dihip_p2p_ar_error": (i32, [vp, vp]),
    "dihip_p2p_allreduce_sum": (i32, [vp, vp, vp, vp, sz, i32]),
    "dihip_debug_set_trace": (i32, [vp, sz]),
    "dihip_debug_attn_plan": (i32, [i32, i32, i32, i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]),
    "dihip_debug_gemv_plan": (i32, [i32, i32, i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32),
                              C.POINTER(sz)]),
}

_lib = None


def header_symbols():
    """Every function declared in include/dashinfer_hip.h."""
    text = open(os.path.join(REPO_ROOT, "include", "dashinfer_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dihip_[a-z0-9_]+)\s*\(", text)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C dash-infer_amd/csrc -j8` "
                "(or __graft_entry__.build()).  There is no CPU fallback.")
        try:  # torch ships its own libamdhip64: it must be the HIP runtime of the process, so
            import torch  # noqa: F401  load it BEFORE our library resolves libamdhip64.so
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


class DihipError(RuntimeError):
    def __init__(self, code, what):
        super().__init__(f"{what}: status {code}: {lib().dihip_last_error().decode()}")
        self.code = code


def check(code, what="dihip call"):
    if code != 0:
        raise DihipError(code, what)
    return code
