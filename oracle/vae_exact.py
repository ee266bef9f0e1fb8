"""ctypes binding + layer driver of oracle/libvae_exact.so (TEST INFRASTRUCTURE): the bf16 SD3-VAE encoder with the exact summation
orders of the reference's torch-CPU run (see oracle/vae_exact.c).  Tensors are numpy uint16 arrays of bf16 bit patterns, NHWC."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
SILU_TABLE = os.path.join(os.path.dirname(_HERE), "tests", "golden", "silu_bf16_table.npy")


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libvae_exact.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "vae_exact.c")):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _LIB = C.CDLL(path)
        _LIB.vx_expf.restype = C.c_float
        _LIB.vx_expf.argtypes = [C.c_float]
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u16(a):
    a = np.ascontiguousarray(a)
    assert a.dtype == np.uint16, a.dtype
    return a


def bf16_bits(t) -> np.ndarray:
    """torch bf16 tensor -> uint16 array"""
    import torch
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def bits_to_torch(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(torch.bfloat16)


def silu_table() -> np.ndarray:
    return np.load(SILU_TABLE)


def conv_order(ic: int, kh: int, stride: int, width: int = 256) -> int:
    """the chunk order oneDNN's AMX kernel uses for a layer of the SD3-VAE encoder (probed per layer shape at 128 / 256 / 320 px,
    tools/probe_cpu_bf16/): a stride-2 layer runs channel-block major (order 3) iff its input is at least 102 pixels wide -- at 256 px the
    128- and 256-channel Downsample layers (256 and 128 wide; the 512-channel one is 64 wide)"""
    if ic < 32:
        return 2
    if stride == 2 and width >= 102:
        return 3
    return 0


def conv2d(x, w, b, stride=1, pad=1, residual=None, order=None, out_hw=None):
    """x [B,H,W,IC]; w [OC,KH,KW,IC]; b [OC] (uint16 bf16 bits).  stride 2 = the Downsample layer: pad 0, the zero row / column that
    F.pad adds at the bottom / right is the bounds check."""
    x, w, b = _u16(x), _u16(w), _u16(b)
    B, H, W, IC = x.shape
    OC, KH, KW, _ = w.shape
    if out_hw is None:
        OH, OW = ((H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1) if stride == 1 else (H // 2, W // 2)
    else:
        OH, OW = out_hw
    y = np.empty((B, OH, OW, OC), dtype=np.uint16)
    r = None if residual is None else _u16(residual)
    rc = lib().vx_conv2d_nhwc(_p(x), _p(w), _p(b), _p(r), _p(y), B, H, W, IC, OC, KH, KW, stride, pad, OH, OW,
                              conv_order(IC, KH, stride, W) if order is None else order)
    assert rc == 0
    return y


def group_norm(x, gamma, beta, groups=32, eps=1e-6, silu=None, want_stats=False):
    x = _u16(x)
    B, H, W, Cn = x.shape
    y = np.empty_like(x)
    stats = np.empty((B, groups, 2), dtype=np.float32) if want_stats else None
    rc = lib().vx_group_norm_nhwc(_p(x), _p(_u16(gamma)), _p(_u16(beta)), _p(y), B, C.c_int64(H * W), Cn, groups, C.c_double(eps),
                                  _p(None if silu is None else _u16(silu)), _p(stats))
    assert rc == 0
    return (y, stats) if want_stats else y


def attention(q, k, v):
    q, k, v = _u16(q), _u16(k), _u16(v)
    B, T, Cd = q.shape
    o = np.empty_like(q)
    assert lib().vx_attention(_p(q), _p(k), _p(v), _p(o), B, T, Cd) == 0
    return o


def expf(x: float) -> float:
    return float(lib().vx_expf(C.c_float(x)))


def add_bf16(a, b):
    fa = (a.astype(np.uint32) << 16).view(np.float32)
    fb = (b.astype(np.uint32) << 16).view(np.float32)
    s = (fa + fb).view(np.uint32)
    return ((s + 0x7FFF + ((s >> 16) & 1)) >> 16).astype(np.uint16)


def pack_weights(vsd):
    """diffusers-layout VAE state dict (torch tensors) -> {name: uint16 arrays}; conv weights as [OC,KH,KW,IC]"""
    import torch
    out = {}
    for k, v in vsd.items():
        if not (k.startswith("encoder.") or k.startswith("decoder.")):
            continue
        t = v.detach().cpu().to(torch.bfloat16)
        if t.dim() == 4:
            t = t.permute(0, 2, 3, 1)
        elif t.dim() == 2:                               # diffusers stores the attention projections as Linear [O, I]
            t = t.reshape(t.shape[0], 1, 1, t.shape[1])
        out[k] = bf16_bits(t.contiguous())
    return out


def encode_moments(pw, img_bits, trace=None):
    """pw = pack_weights(vsd); img_bits [B,256,256,3] uint16 -> moments [B,32,32,32] uint16 (NHWC; channels 0..15 = the mean)"""
    tab = silu_table()

    def conv(name, x, **kw):
        y = conv2d(x, pw[name + ".weight"], pw[name + ".bias"], **kw)
        if trace is not None:
            trace.append((name, y))
        return y

    def gn(name, x, act=True):
        y = group_norm(x, pw[name + ".weight"], pw[name + ".bias"], silu=tab if act else None)
        if trace is not None:
            trace.append((name, y))
        return y

    def res(p, x):
        h = conv(p + ".conv1", gn(p + ".norm1", x))
        sc = conv(p + ".conv_shortcut", x, pad=0) if (p + ".conv_shortcut.weight") in pw else x
        return conv(p + ".conv2", gn(p + ".norm2", h), residual=sc)

    h = conv("encoder.conv_in", img_bits)
    for lvl in range(4):
        for j in range(2):
            h = res(f"encoder.down_blocks.{lvl}.resnets.{j}", h)
        if lvl != 3:
            h = conv(f"encoder.down_blocks.{lvl}.downsamplers.0.conv", h, stride=2, pad=0)
    h = res("encoder.mid_block.resnets.0", h)
    p = "encoder.mid_block.attentions.0"
    B, H, W, Cn = h.shape
    n = gn(p + ".group_norm", h, act=False)
    q, k, v = (conv(p + s, n, pad=0).reshape(B, H * W, Cn) for s in (".to_q", ".to_k", ".to_v"))
    a = attention(q, k, v).reshape(B, H, W, Cn)
    if trace is not None:
        trace.append((p + ".sdpa", a))
    h = conv(p + ".to_out.0", a, pad=0, residual=h)
    h = res("encoder.mid_block.resnets.1", h)
    return conv("encoder.conv_out", gn("encoder.conv_norm_out", h))


def upsample2x(x):
    """F.interpolate(scale_factor=2, mode='nearest') on an NHWC array (Upsample, sd3_impls.py:308-311): a copy"""
    return np.ascontiguousarray(np.repeat(np.repeat(x, 2, axis=1), 2, axis=2))


def decode(pw, z_bits, trace=None):
    """pw = pack_weights(vsd) (incl. the decoder.* keys); z_bits [B,32,32,16] uint16 (NHWC bf16 latents after process_out) -> pixels
    [B,256,256,3] uint16, before norm_ip: `VAEDecoder.forward` (sd3_impls.py:427-444) in the summation orders of the reference's torch-CPU run.
    Every convolution of the decoder -- incl. the three that read a nearest-upsampled input, conv_in (16 channels: one 16-channel chunk per
    tap) and conv_out (3 output channels) -- sums its chunks in (kh, kw, channel-block) order (tools/probe_cpu_bf16/check_decoder_convs.py,
    fprev_conv_ic16.py: 0 mismatches against F.conv2d on every layer shape); GroupNorm / SiLU / attention as in the encoder."""
    tab = silu_table()

    def conv(name, x, **kw):
        w = pw[name + ".weight"]
        # order 1 (channel-block major into the one running total) once a 3x3 layer's bf16 input or output reaches 2^31 bytes: 64 images of 256 channels
        # at 256 x 256 (tools/probe_cpu_bf16/check_conv_batch64.py)
        big = w.shape[1] == 3 and 2 * x.shape[0] * x.shape[1] * x.shape[2] * max(w.shape[0], w.shape[3]) >= 2 ** 31
        y = conv2d(x, w, pw[name + ".bias"], order=1 if big else 0, **kw)
        if trace is not None:
            trace.append((name, y))
        return y

    def gn(name, x, act=True):
        y = group_norm(x, pw[name + ".weight"], pw[name + ".bias"], silu=tab if act else None)
        if trace is not None:
            trace.append((name, y))
        return y

    def res(p, x):
        h = conv(p + ".conv1", gn(p + ".norm1", x))
        sc = conv(p + ".conv_shortcut", x, pad=0) if (p + ".conv_shortcut.weight") in pw else x
        return conv(p + ".conv2", gn(p + ".norm2", h), residual=sc)

    h = conv("decoder.conv_in", z_bits)
    h = res("decoder.mid_block.resnets.0", h)
    p = "decoder.mid_block.attentions.0"
    B, H, W, Cn = h.shape
    n = gn(p + ".group_norm", h, act=False)
    q, k, v = (conv(p + s, n, pad=0).reshape(B, H * W, Cn) for s in (".to_q", ".to_k", ".to_v"))
    a = attention(q, k, v).reshape(B, H, W, Cn)
    h = conv(p + ".to_out.0", a, pad=0, residual=h)
    h = res("decoder.mid_block.resnets.1", h)
    for lvl in range(4):
        for j in range(3):
            h = res(f"decoder.up_blocks.{lvl}.resnets.{j}", h)
        if lvl != 3:
            h = conv(f"decoder.up_blocks.{lvl}.upsamplers.0.conv", upsample2x(h))
    return conv("decoder.conv_out", gn("decoder.conv_norm_out", h))
