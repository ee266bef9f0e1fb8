"""ctypes binding + block driver of oracle/libencoder_exact.so (TEST INFRASTRUCTURE): the fp32 Q-Former encoder in the exact
summation orders / transcendental polynomials of the reference's torch-CPU run (see oracle/encoder_exact.c).  numpy fp32 arrays;
the element-wise glue (`x * (1 + scale) + shift`, `q + gate * y`) is numpy: separately rounded IEEE operations, like torch's."""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
ENC_DEPTH, ENC_HIDDEN, ENC_QDIM, ENC_HEADS, ENC_QHEADS = 16, 64, 512, 4, 8


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libencoder_exact.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "encoder_exact.c")):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _LIB = C.CDLL(path)
        for n in ("xe_sleef_expf", "xe_sleef_tanhf", "xe_gelu_tanh1", "xe_silu1", "xe_exp_u20", "xe_expf"):
            getattr(_LIB, n).restype = C.c_float
            getattr(_LIB, n).argtypes = [C.c_float]
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def linear(x, w, b=None):
    """F.linear in MKL's order: x [..., K], w [N, K]"""
    x = _f32(x); w = _f32(w); b = _f32(b) if b is not None else None
    K = x.shape[-1]; N = w.shape[0]; M = x.size // K
    out = np.empty(x.shape[:-1] + (N,), np.float32)
    lib().xe_linear(_p(x), _p(w), _p(b), _p(out), C.c_long(M), N, K)
    return out


def layernorm(x, gamma=None, beta=None, eps=1e-6, want_stats=False):
    x = _f32(x); N = x.shape[-1]; rows = x.size // N
    y = np.empty_like(x)
    st = np.empty((rows, 2), np.float32) if want_stats else None
    lib().xe_layernorm(_p(x), _p(y), _p(_f32(gamma)) if gamma is not None else None, _p(_f32(beta)) if beta is not None else None,
                       C.c_long(rows), N, C.c_float(eps), _p(st))
    return (y, st) if want_stats else y


def gelu_tanh(x):
    x = _f32(x); y = np.empty_like(x)
    lib().xe_gelu_tanh(_p(x), _p(y), C.c_long(x.size))
    return y


def silu(x):
    x = _f32(x); y = np.empty_like(x)
    lib().xe_silu(_p(x), _p(y), C.c_long(x.size))
    return y


def patch_embed(x, w, b, bias_first=0):
    """x [B,C,H,W], w [OC,C,2,2] -> [B, H/2*W/2, OC]"""
    x = _f32(x); w = _f32(w); b = _f32(b)
    B, Cc, H, W = x.shape; OC = w.shape[0]
    y = np.empty((B, (H // 2) * (W // 2), OC), np.float32)
    lib().xe_patch_embed(_p(x), _p(w), _p(b), _p(y), B, Cc, H, W, OC, int(bias_first))
    return y


def attention(q, k1, v1, heads, k2=None, v2=None, valid1=None, slots1=None):
    """`valid1` / `slots1`: the first segment occupies slots1 key positions of which the first valid1 are visible (k1 / v1 hold >= valid1 rows)"""
    """q [B,Tq,H*D] (a strided view of a fused projection is fine: last dim contiguous), k1/v1 [B,Tk1,H*D], optional second segment"""
    def rs(a):
        assert a.strides[-1] == 4 and a.strides[0] == a.shape[1] * a.strides[1], "rows must be equally strided"
        return a.strides[1] // 4
    B, Tq, HD = q.shape; D = HD // heads
    out = np.empty((B, Tq, HD), np.float32)
    Tk2 = 0 if k2 is None else k2.shape[1]
    slots = k1.shape[1] if slots1 is None else int(slots1)
    lib().xe_attention_masked(_p(q), C.c_long(rs(q)), _p(k1), _p(v1), C.c_long(rs(k1)), slots, k1.shape[1] if valid1 is None else int(valid1), k1.shape[1],
                              _p(k2), _p(v2), C.c_long(rs(k2) if k2 is not None else 0), Tk2, _p(out), B, heads, Tq, D)
    return out


# ---- host-side, input-independent pieces: torch-CPU's own cos / sin / exp (models.py:56-74) ---------------------------------------
def timestep_embedding(t, dim=256):
    import torch
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = torch.as_tensor(t)[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1).numpy()


def encoder_tables(sd, K, pos_emb):
    """[K, 6*512] adaLN tables per block: adaLN_modulation(SiLU(t_embedder(pos)))  (modules.py:312-318).  `pos_emb` [K, 256]: the sinusoidal
    embedding of the positions 1000 + 8 k -- `timestep_embedding(positions)` on the build container (MKL VML: host dependent), or the
    table the product ships (selftoktokenizer_amd/data/encoder_pos_sincos.npy), which is that very array."""
    g = lambda k: sd[k].detach().cpu().numpy() if hasattr(sd[k], "detach") else sd[k]
    emb = _f32(pos_emb)[:K]
    out = []
    for i in range(ENC_DEPTH):
        p = f"encoder.blocks.{i}"
        h = linear(emb, g(p + ".t_embedder.mlp.0.weight"), g(p + ".t_embedder.mlp.0.bias"))
        h = linear(silu(h), g(p + ".t_embedder.mlp.2.weight"), g(p + ".t_embedder.mlp.2.bias"))
        out.append(linear(silu(h), g(p + ".adaLN_modulation.1.weight"), g(p + ".adaLN_modulation.1.bias")))
    return out


def crop_pos(pos, h, w):
    grid = int(round(math.sqrt(pos.shape[1])))
    top, left = (grid - h) // 2, (grid - w) // 2
    return pos.reshape(grid, grid, -1)[top:top + h, left:left + w].reshape(1, h * w, -1)


def encoder_features(sd, x0, pos_emb, tables=None, bias_first=0, trace=None, pre_norm=False):
    """x0 [B,16,h,w] fp32 -> pre-quantizer features z [B,K,16] with the reference's bits (B >= 8: see encoder_exact.c)"""
    g = lambda k: sd[k].detach().cpu().numpy() if hasattr(sd[k], "detach") else sd[k]
    x0 = _f32(x0)
    B, _, H, W = x0.shape
    K = g("encoder.query_tokens").shape[1]
    tables = tables or encoder_tables(sd, K, pos_emb)
    x = patch_embed(x0, g("encoder.x_embedder.proj.weight"), g("encoder.x_embedder.proj.bias"), bias_first) + crop_pos(g("encoder.pos_embed"), H // 2, W // 2)
    q = np.broadcast_to(g("encoder.query_tokens"), (B, K, ENC_QDIM)).copy()
    Hd, Q = ENC_HIDDEN, ENC_QDIM
    for i in range(ENC_DEPTH):
        p = f"encoder.blocks.{i}"
        t = tables[i]
        sh_msa, sc_msa, g_msa, sh_mlp, sc_mlp, g_mlp = [t[None, :, j * Q:(j + 1) * Q] for j in range(6)]
        xn = layernorm(x)
        qn = layernorm(q) * (np.float32(1) + sc_msa) + sh_msa
        qkv = linear(xn, g(p + ".attn.qkv.weight"), g(p + ".attn.qkv.bias"))
        kvx = linear(xn, g(p + ".attn.to_query_kv.weight"), g(p + ".attn.to_query_kv.bias"))
        qq = linear(qn, g(p + ".attn.query_linear.weight"), g(p + ".attn.query_linear.bias"))
        xa = attention(qkv[..., :Hd], qkv[..., Hd:2 * Hd], qkv[..., 2 * Hd:], ENC_HEADS)
        qa = attention(qq[..., :Q], kvx[..., :Q], kvx[..., Q:], ENC_QHEADS, qq[..., Q:2 * Q], qq[..., 2 * Q:])
        xa = linear(xa, g(p + ".attn.proj.weight"), g(p + ".attn.proj.bias"))
        qa = linear(qa, g(p + ".attn.query_proj.weight"), g(p + ".attn.query_proj.bias"))
        x = x + xa
        h = gelu_tanh(linear(layernorm(x), g(p + ".mlp.fc1.weight"), g(p + ".mlp.fc1.bias")))
        x = x + linear(h, g(p + ".mlp.fc2.weight"), g(p + ".mlp.fc2.bias"))
        q = q + g_msa * qa
        h = gelu_tanh(linear(layernorm(q) * (np.float32(1) + sc_mlp) + sh_mlp, g(p + ".q_mlp.fc1.weight"), g(p + ".q_mlp.fc1.bias")))
        q = q + g_mlp * linear(h, g(p + ".q_mlp.fc2.weight"), g(p + ".q_mlp.fc2.bias"))
        if trace is not None:
            trace.append((x.copy(), q.copy()))
    if pre_norm:                                     # models_ours.py:219-220: outs = self.final_layer_norm(outs)
        q = layernorm(q, g("encoder.final_layer_norm.weight"), g("encoder.final_layer_norm.bias"))
    return linear(q, g("encoder.quantizer.project_in.weight"), g("encoder.quantizer.project_in.bias"))
