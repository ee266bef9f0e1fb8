"""not gpu: the ctypes stubs INTEGRATION.md shows for the round-6 entry points are EXECUTED -- the text of the document, verbatim, bound to the CPU twin of the C ABI
(oracle/libselftok_cpu.so: same symbols, host pointers, `stream` ignored) -- and compared with the oracle: a stub with a transposed argument
or a wrong stride would otherwise only show when a maintainer pastes it."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import encoder_exact as EX
from selftoktokenizer_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TWIN = os.path.join(ROOT, "oracle", "libselftok_cpu.so")


def _blocks(text, heading):
    sec = text[text.index(heading):]
    sec = sec[:sec.index("\n### ", 5)] if "\n### " in sec[5:] else sec
    return re.findall(r"```python\n(.*?)```", sec, flags=re.S)


def _rand(seed, shape, scale=1.0):
    return (synth.hash_normalish(seed, shape) * scale).float().contiguous()


@pytest.fixture(scope="module")
def stubs():
    if not os.path.exists(TWIN):
        pytest.skip("oracle/libselftok_cpu.so not built (python __graft_entry__.py / make -C oracle)")
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    pre = _blocks(text, "## 2. Kernel by kernel")[0]
    assert 'ctypes.CDLL("libselftok_hip.so")' in pre and "def _s():" in pre
    pre = pre.replace('ctypes.CDLL("libselftok_hip.so")', f'ctypes.CDLL({TWIN!r})')                       # the ONE line section 2.7 tells a maintainer to change
    pre = pre.replace("def _s(): return torch.cuda.current_stream().cuda_stream", "def _s(): return None")   # no stream on the CPU twin
    ns = {}
    exec(pre, ns)
    exec(_blocks(text, "### 2.6d'' Round 6")[0], ns)
    for heading in ("### 2.1 VQ nearest code", "### 2.2 Code gather", "### 2.3 LayerNorm + modulate", "### 2.4 Joint attention", "### 2.6d The Q-Former encoder"):
        exec(_blocks(text, heading)[0], ns)
    ns16 = dict(ns)                                              # section 2.4b defines its own `linear` (the f16x2 arithmetic): a namespace of its own
    exec(_blocks(text, "### 2.4b fp32-equivalent")[0], ns16)
    ns["f16x2"] = ns16
    nsv = dict(ns)                                               # section 2.6c (the exact VAE): its own `sdpa`; the SiLU table is made on the device the stubs run on
    exec(_blocks(text, "### 2.6c The VAE encoder")[0].replace('device="cuda"', 'device="cpu"'), nsv)
    ns["vae"] = nsv
    return ns


def test_linear_wide_stub(stubs):
    x, w, b = _rand(1, (300, 1536), 1.1), _rand(2, (256, 1536), 0.03), _rand(3, (256,), 0.2)
    out = stubs["linear_wide"](x, w, b)
    ref = EX.linear(x.numpy(), w.numpy(), b.numpy())
    assert out.shape == (300, 256) and int((out.numpy().view(np.uint32) != ref.view(np.uint32)).sum()) == 0


def test_sdpa_joint_stub(stubs):
    B, H, D, Tq, vis, slots, Tk2 = 2, 3, 64, 70, 50, 512, 256
    HD = H * D
    q, ctx, img = _rand(4, (B, Tq, HD), 1.2), _rand(5, (B, vis, 2 * HD), 1.2), _rand(6, (B, Tk2, 2 * HD), 1.2)
    out = stubs["sdpa_joint"](q, ctx[..., :HD], ctx[..., HD:], H, img[..., :HD], img[..., HD:], slots, vis)
    ref = EX.attention(q.numpy(), ctx[..., :HD].numpy(), ctx[..., HD:].numpy(), H, img[..., :HD].numpy(), img[..., HD:].numpy(), valid1=vis, slots1=slots)
    assert int((out.numpy().view(np.uint32) != ref.view(np.uint32)).sum()) == 0


def test_residual_then_norm_mod_stub(stubs):
    B, T, N = 3, 40, 1536
    x, lin, bias = _rand(7, (B, T, N), 1.5), _rand(8, (B, T, N), 1.0), _rand(9, (N,), 0.2)
    tab = _rand(10, (B, 3 * N), 0.6)
    gate, shift, scale = tab[:, :N], tab[:, N:2 * N], tab[:, 2 * N:]
    x0 = x.clone()
    out = stubs["residual_then_norm_mod"](x, lin, bias, gate, T, shift, scale)
    x1 = x0.numpy() + gate.numpy()[:, None, :] * (lin.numpy() + bias.numpy())                  # three separately rounded fp32 operations (sd3/mmdit.py:485-487)
    want = EX.layernorm(x1.reshape(B * T, N)).reshape(B, T, N) * (np.float32(1) + scale.numpy()[:, None, :]) + shift.numpy()[:, None, :]
    assert int((x.numpy().view(np.uint32) != x1.view(np.uint32)).sum()) == 0, "x must have been updated in place"
    assert int((out.numpy().view(np.uint32) != want.view(np.uint32)).sum()) == 0


def test_round5_encoder_stubs(stubs):
    """section 2.6d: linear / norm_mod / sdpa of the exact Q-Former encoder"""
    lin = torch.nn.Linear(512, 192)
    with torch.no_grad():
        lin.weight.copy_(_rand(11, (192, 512), 0.04)); lin.bias.copy_(_rand(12, (192,), 0.2))
    x = _rand(13, (2, 300, 512), 1.1)
    ref = EX.linear(x.numpy().reshape(-1, 512), lin.weight.detach().numpy(), lin.bias.detach().numpy()).reshape(2, 300, 192)
    assert int((stubs["linear"](x, lin).numpy().view(np.uint32) != ref.view(np.uint32)).sum()) == 0
    tab = _rand(14, (300, 2 * 512), 0.5)
    want = EX.layernorm(x.numpy().reshape(-1, 512)).reshape(2, 300, 512) * (np.float32(1) + tab[:, 512:].numpy()[None]) + tab[:, :512].numpy()[None]
    got = stubs["norm_mod"](x, tab[:, :512], tab[:, 512:])
    assert int((got.numpy().view(np.uint32) != want.view(np.uint32)).sum()) == 0
    H = 8
    q, kv, kv2 = _rand(15, (2, 70, 512), 1.2), _rand(16, (2, 256, 1024), 1.2), _rand(17, (2, 64, 1024), 1.2)             # key count a multiple of 16
    ref = EX.attention(q.numpy(), kv[..., :512].numpy(), kv[..., 512:].numpy(), H, kv2[..., :512].numpy(), kv2[..., 512:].numpy())
    got = stubs["sdpa"](q, kv[..., :512], kv[..., 512:], H, kv2[..., :512], kv2[..., 512:])
    assert int((got.numpy().view(np.uint32) != ref.view(np.uint32)).sum()) == 0


def test_round1_stubs(stubs):
    """sections 2.1 - 2.3: VQ nearest code, code gather + LayerNorm(16), residual + LayerNorm + modulate"""
    from oracle import clib
    cb = torch.nn.functional.normalize(_rand(18, (1, 32768, 16)), dim=-1)
    packed = stubs["pack"](cb)
    z = _rand(19, (3, 100, 16), 1.3)
    ids = stubs["nearest_code"](z, packed, 32768)
    want, _ = clib.vq_encode(z.reshape(-1, 16).numpy(), cb[0].numpy())
    assert np.array_equal(ids.numpy().reshape(-1), want)
    ln = torch.nn.LayerNorm(16, eps=1e-6)
    with torch.no_grad():
        ln.weight.copy_(_rand(20, (16,), 0.3) + 1); ln.bias.copy_(_rand(21, (16,), 0.2))
    got = stubs["codes_ln"](ids, cb[0].contiguous(), ln)
    assert float((got - ln(cb[0][ids]).detach()).abs().max()) < 2e-6
    B, T, Hd = 2, 50, 512
    x, y, tab = _rand(22, (B, T, Hd), 1.4), _rand(23, (B, T, Hd), 1.0), _rand(24, (T, 6 * Hd), 0.5)
    gate, shift, scale = tab[:, 2 * Hd:3 * Hd], tab[:, 3 * Hd:4 * Hd], tab[:, 4 * Hd:5 * Hd]
    xo, n = stubs["residual_ln_mod"](x, y, gate, shift, scale, False)
    x1 = x + gate[None] * y
    assert float((xo - x1).abs().max()) < 1e-6
    assert float((n - (torch.nn.functional.layer_norm(x1, (Hd,), eps=1e-6) * (1 + scale[None]) + shift[None])).abs().max()) < 2e-5


def test_joint_attention_and_f16x2_linear_stubs(stubs):
    """sections 2.4 (the descriptor struct of selftok_attn_f32, prefix visibility, context_see_xt) and 2.4b (packed f16x2 Linear)"""
    B, Hh, n_ctx, n_x = 2, 3, 40, 64
    D = Hh * 64
    cq, xq = _rand(30, (B, n_ctx, 3 * D), 1.1), _rand(31, (B, n_x, 3 * D), 1.1)
    kvis = torch.tensor([25, 39], dtype=torch.int32)
    oc, ox = stubs["joint_attention"](cq, xq, Hh, kvis, True)
    def heads(t):
        return t.reshape(B, -1, Hh, 64).transpose(1, 2)
    q = heads(torch.cat([cq[..., :D], xq[..., :D]], 1)); k = heads(torch.cat([cq[..., D:2 * D], xq[..., D:2 * D]], 1)); v = heads(torch.cat([cq[..., 2 * D:], xq[..., 2 * D:]], 1))
    vis = torch.cat([torch.arange(n_ctx)[None] <= kvis[:, None], torch.ones(B, n_x, dtype=torch.bool)], 1)             # key visibility [B, n_ctx + n_x]
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=vis[:, None, None, :]).transpose(1, 2).reshape(B, n_ctx + n_x, D)
    assert float((ox - ref[:, n_ctx:]).abs().max()) < 2e-5
    for b in range(B):                                           # context rows beyond k are not written: compare the live ones
        live = int(kvis[b]) + 1
        assert float((oc[b, :live] - ref[b, :live]).abs().max()) < 2e-5
    f = stubs["f16x2"]
    lin = torch.nn.Linear(256, 128)
    with torch.no_grad():
        lin.weight.copy_(_rand(32, (128, 256), 0.05)); lin.bias.copy_(_rand(33, (128,), 0.2))
    flag = torch.zeros(1, dtype=torch.int32)
    packed = f["pack_linear"](lin, flag)
    x = _rand(34, (3, 50, 256), 1.0)
    got = f["linear"](x, packed, lin.bias.detach(), 128, False, flag)
    want = (x.double() @ lin.weight.detach().double().t() + lin.bias.detach().double()).float()
    assert int(flag.item()) == 0 and float((got - want).abs().max()) < 5e-6


def test_exact_vae_stubs(stubs):
    """section 2.6c: convolution (stride 1 with the ResnetBlock residual, the Downsample layer), GroupNorm + SiLU, the AttnBlock attention -- bf16 bits against oracle/vae_exact"""
    from oracle import vae_exact as VX
    v = stubs["vae"]
    bf = lambda seed, shape, scale=1.0: _rand(seed, shape, scale).to(torch.bfloat16)
    x, res = bf(40, (1, 16, 16, 128)), bf(41, (1, 16, 16, 128))

    class Conv:                                                   # what the stub reads of a conv module: weight_nhwc, bias
        def __init__(self, seed, cout, cin):
            self.weight_nhwc, self.bias = bf(seed, (cout, 3, 3, cin), 0.03), bf(seed + 1, (cout,), 0.1)
    c1 = Conv(42, 128, 128)
    got = v["conv"](x, c1, residual=res)
    want = VX.conv2d(VX.bf16_bits(x), VX.bf16_bits(c1.weight_nhwc), VX.bf16_bits(c1.bias), residual=VX.bf16_bits(res))
    assert np.array_equal(VX.bf16_bits(got), want)
    xd = bf(44, (1, 32, 32, 128))
    got = v["conv"](xd, c1, stride=2)
    want = VX.conv2d(VX.bf16_bits(xd), VX.bf16_bits(c1.weight_nhwc), VX.bf16_bits(c1.bias), stride=2, pad=0)
    assert got.shape == (1, 16, 16, 128) and np.array_equal(VX.bf16_bits(got), want)
    norm = torch.nn.GroupNorm(32, 128, eps=1e-6)
    with torch.no_grad():
        norm.weight.copy_(_rand(45, (128,), 0.3) + 1); norm.bias.copy_(_rand(46, (128,), 0.2))
    norm = norm.to(torch.bfloat16)
    got = v["norm_swish"](x, norm)
    want = VX.group_norm(VX.bf16_bits(x), VX.bf16_bits(norm.weight), VX.bf16_bits(norm.bias), silu=VX.silu_table())
    assert np.array_equal(VX.bf16_bits(got), want)
    q, k, vv = bf(47, (1, 64, 128)), bf(48, (1, 64, 128)), bf(49, (1, 64, 128))
    assert np.array_equal(VX.bf16_bits(v["sdpa"](q, k, vv)), VX.attention(VX.bf16_bits(q), VX.bf16_bits(k), VX.bf16_bits(vv)))
