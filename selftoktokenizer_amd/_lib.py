"""ctypes loader for libselftok_hip.so (the C ABI of include/selftok_hip.h).

No fallback: if the library is missing or a symbol is absent this raises -- the product path must
fail loudly rather than silently run something else.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SELFTOK_HIP_LIB") or os.path.join(_HERE, "libselftok_hip.so")   # override: ablation builds (tools/)

_vp, _i, _f, _sz, _l = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_long

# name -> (restype, argtypes); must list every symbol declared in include/selftok_hip.h
SIGNATURES = {
    "selftok_version": (_i, []),
    "selftok_last_error": (C.c_char_p, []),
    "selftok_vq_workspace_bytes": (_sz, [_i, _i]),
    "selftok_vq_encode_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "selftok_vq_packed_bytes": (_sz, [_i, _i]),
    "selftok_vq_pack_codebook": (_i, [_vp, _vp, _i, _i, _vp]),
    "selftok_vq_encode_packed_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "selftok_vq_argmax_partial_packed_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "selftok_vq_finalize_packed": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "selftok_vq_ema_accumulate_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "selftok_vq_tpc_update_f32": (_i, [_vp, _vp, _i, _i, _i, _f, _i, _vp]),
    "selftok_vq_softmax_workspace_bytes": (_sz, [_i]),
    "selftok_vq_softmax_stats_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "selftok_vq_softmax_backward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "selftok_code_gather_ln_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp]),
    "selftok_residual_ln_mod_f32": (_i, [_vp] * 7 + [_i, _i, _i, _l, _l, _l, _l, _f, _vp]),
    "selftok_bias_gelu_f32": (_i, [_vp, _vp, _l, _i, _vp]),
    "selftok_silu_f32": (_i, [_vp, _vp, _l, _vp]),
    "selftok_add_rows_f32": (_i, [_vp, _vp, _vp, _i, _l, _vp]),
    "selftok_timestep_embed_f32": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp]),
    "selftok_patchify_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "selftok_unpatchify_cfg_euler_f32": (_i, [_vp] * 5 + [_i, _i, _i, _i, _f, _f, _vp]),
    "selftok_rmsnorm_f32": (_i, [_vp, _vp, _vp, _l, _i, _f, _vp]),
    "selftok_rotary_f32": (_i, [_vp, _vp, _vp, _l, _i, _i, _f, _vp]),
    "selftok_attn_f32": (_i, [_vp, _vp]),
    "selftok_residual_ln_mod_split": (_i, [_vp] * 8 + [_i, _i, _i, _l, _l, _l, _l, _f, _vp]),
    "selftok_linear_f16x2_packed_bytes": (_sz, [_i, _i]),
    "selftok_linear_f16x2_pack_weight": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "selftok_linear_f16x2_f32": (_i, [_vp, _l, _vp, _vp, _vp, _l, _i, _i, _i, _i, _vp, _vp]),
    "selftok_split_f16x2_bytes": (_sz, [_l, _i]),
    "selftok_split_f16x2_f32": (_i, [_vp, _l, _vp, _l, _i, _vp, _vp]),
    "selftok_linear_f16x2_split": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _vp, _vp]),
    "selftok_linear_f16x2_split_residual": (_i, [_vp, _vp, _vp, _vp, _l, _vp, _l, _l, _i, _vp, _l, _i, _i, _i, _vp, _vp]),
    "selftok_linear_f16x2_splitk_workspace_bytes": (_sz, [_i, _i, _i]),
    "selftok_linear_f16x2_split_k": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "selftok_linear_f16x2_split_residual_k": (_i, [_vp, _vp, _vp, _vp, _l, _vp, _l, _l, _i, _vp, _l, _i, _i, _i, _i, _vp, _vp, _vp]),
    "selftok_groupnorm_silu_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "selftok_latent_process_in": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _f, _vp]),
    "selftok_latent_process_out": (_i, [_vp, _vp, _l, _f, _f, _vp]),
    "selftok_clamp01_bf16": (_i, [_vp, _l, _vp]),
    "selftok_conv2d_packed_bytes": (_sz, [_i, _i, _i, _i]),
    "selftok_conv2d_pack_weight_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "selftok_conv2d_nhwc_bf16": (_i, [_vp, _vp, _vp, _vp, _vp] + [_i] * 11 + [_vp]),
    "selftok_groupnorm_nhwc_workspace_bytes": (_sz, [_i, _i, _i]),
    "selftok_groupnorm_silu_nhwc_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "selftok_vx_conv2d_bf16": (_i, [_vp] * 5 + [_i] * 9 + [_vp]),
    "selftok_vx_groupnorm_workspace_bytes": (_sz, [_i, _i, _i]),
    "selftok_vx_groupnorm_bf16": (_i, [_vp] * 7 + [_i, _i, _i, _i, C.c_double, _vp]),
    "selftok_vx_silu_table_bf16": (_i, [_vp, _vp]),
    "selftok_vx_attention_workspace_bytes": (_sz, [_i, _i, _i]),
    "selftok_vx_attention_bf16": (_i, [_vp] * 5 + [_i, _i, _i, _vp]),
    "selftok_vx_expf_f32": (_i, [_vp, _vp, _l, _vp]),
    "selftok_ex_linear_f32": (_i, [_vp, _l, _vp, _vp, _vp, _l, _i, _vp, _l, _i, _vp, _l, _l, _i, _i, _i, _vp]),
    "selftok_linear_f32_workspace_bytes": (_sz, [_l, _i, _i, _i]),
    "selftok_linear_f32": (_i, [_vp, _l, _vp, _vp, _vp, _l, _i, _vp, _l, _i, _vp, _l, _l, _i, _i, _i, _vp, _sz, _vp]),
    "selftok_ex_layernorm_mod_f32": (_i, [_vp, _l, _vp, _l, _vp, _vp, _l, _i, _vp, _vp, _vp, _l, _i, _f, _vp]),
    "selftok_ex_res_layernorm_mod_f32": (_i, [_vp, _l, _vp, _l, _vp, _vp, _l, _i, _vp, _l, _vp, _l, _vp, _vp, _l, _i, _l, _i, _f, _vp]),
    "selftok_ex_unary_f32": (_i, [_vp, _vp, _l, _i, _vp]),
    "selftok_ex_attention_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "selftok_ex_attention_f32": (_i, [_vp, _l, _vp, _vp, _l, _i, _i, _i, _vp, _vp, _l, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "selftok_ex_attention_fused_supported": (_i, [_i, _i, _i]),
    "selftok_ex_attention_fused_f32": (_i, [_vp, _l, _vp, _vp, _l, _i, _i, _i, _vp, _vp, _l, _i, _vp, _i, _i, _i, _i, _vp]),
}


class AttnSeg(C.Structure):
    _fields_ = [("q", _vp), ("k", _vp), ("v", _vp), ("o", _vp), ("len", _i),
                ("q_rs", _l), ("k_rs", _l), ("v_rs", _l), ("o_rs", _l),
                ("q_bs", _l), ("k_bs", _l), ("v_bs", _l), ("o_bs", _l)]


class AttnDesc(C.Structure):
    _fields_ = [("seg", AttnSeg * 2), ("B", _i), ("H", _i), ("head_dim", _i), ("kvis", _vp),
                ("seg0_sees_seg1", _i), ("scale", _f), ("mode", _i), ("overflow", _vp),
                ("o_blk", _vp * 2)]

_lib = None


class SelftokHipError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SelftokHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().selftok_last_error().decode("utf-8", "replace")
        raise SelftokHipError(f"{what} failed (rc={rc}): {msg}")
