"""-m gpu: every fused HIP kernel (through the C ABI) against a plain torch fp32 reference of the same op."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from selftoktokenizer_amd import ops, synth

pytestmark = pytest.mark.gpu


def U(seed, shape, lo=-1.0, hi=1.0):
    return synth.hash_uniform(seed, shape, lo, hi)


def ln(x):
    return F.layer_norm(x, (x.shape[-1],), None, None, 1e-6)


@pytest.mark.parametrize("H", [64, 512, 1536])
@pytest.mark.parametrize("mode", ["token", "sample", "plain"])
def test_residual_ln_mod(H, mode):
    B, T = 3, 37
    x, y = U(1, (B, T, H), -2, 2), U(2, (B, T, H), -2, 2)
    rows = T if mode != "sample" else B
    table = U(3, (rows, 6 * H), -0.5, 0.5)
    xc, yc, tc = x.cuda(), y.cuda(), table.cuda()
    if mode == "plain":
        xo, n = ops.residual_ln_mod(xc, y=yc)
        ref_x = x + y
        ref_n = ln(ref_x)
    else:
        sh, sc, g = table[:, 0:H], table[:, H:2 * H], table[:, 2 * H:3 * H]
        xo, n = ops.residual_ln_mod(xc, y=yc, gate=tc[:, 2 * H:3 * H], shift=tc[:, 0:H], scale=tc[:, H:2 * H],
                                    per_sample=(mode == "sample"))
        ax = 0 if mode == "token" else 1
        ref_x = x + g.unsqueeze(ax) * y
        ref_n = ln(ref_x) * (1 + sc.unsqueeze(ax)) + sh.unsqueeze(ax)
    torch.testing.assert_close(xo.cpu(), ref_x, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(n.cpu(), ref_n, rtol=2e-5, atol=2e-5)
    # LN only (no residual)
    _, n2 = ops.residual_ln_mod(xc)
    torch.testing.assert_close(n2.cpu(), ln(x), rtol=2e-5, atol=2e-5)
    # split-activation output (what the f16x2 Linear consumes): exactly the split of the fp32 output, same x'
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    if mode == "plain":
        xo_s, n_s = ops.residual_ln_mod(xc, y=yc, split=True, overflow=flag)
    else:
        xo_s, n_s = ops.residual_ln_mod(xc, y=yc, gate=tc[:, 2 * H:3 * H], shift=tc[:, 0:H], scale=tc[:, H:2 * H],
                                        per_sample=(mode == "sample"), split=True, overflow=flag)
    assert n_s.dtype == torch.float16 and n_s.shape == (B, T, H)
    assert torch.equal(xo_s, xo) and torch.equal(n_s.planes(), ops.split_f16x2(n).planes()) and int(flag.item()) == 0


def test_bias_gelu_silu_addrows():
    h, b = U(4, (130, 2048), -4, 4), U(5, (2048,), -0.1, 0.1)
    out = ops.bias_gelu_(h.cuda().clone(), b.cuda())
    torch.testing.assert_close(out.cpu(), F.gelu(h + b, approximate="tanh"), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ops.silu(h.cuda()).cpu(), F.silu(h), rtol=1e-5, atol=1e-6)
    x, t = U(6, (5, 256, 64)), U(7, (256, 64))
    torch.testing.assert_close(ops.add_rows_(x.cuda().clone(), t.cuda()).cpu(), x + t)


def test_timestep_embed():
    from oracle import model as OM
    half = 128
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
    t = torch.tensor([1.0, 0.62, 0.02, 0.0, 0.98])
    got = ops.timestep_embed(t.cuda(), freqs.cuda(), 1000.0).cpu()
    torch.testing.assert_close(got, OM.timestep_embedding(t * 1000.0), rtol=0, atol=2e-6)
    pos = (1000 + 8 * torch.arange(512)).float()
    got = ops.timestep_embed(pos.cuda(), freqs.cuda(), 1.0).cpu()
    torch.testing.assert_close(got, OM.timestep_embedding(pos), rtol=0, atol=2e-6)


def test_patchify_matches_conv():
    x = U(8, (3, 16, 32, 32))
    w, b = U(9, (64, 16, 2, 2), -0.2, 0.2), U(10, (64,), -0.1, 0.1)
    ref = F.conv2d(x, w, b, stride=2).flatten(2).transpose(1, 2)
    p = ops.patchify(x.cuda())
    got = torch.addmm(b.cuda(), p.reshape(-1, 64), w.reshape(64, 64).t().cuda()).reshape(3, 256, 64)
    torch.testing.assert_close(got.cpu(), ref, rtol=1e-5, atol=1e-5)


def test_unpatchify_cfg_euler():
    from oracle import model as OM
    B = 2
    yc, yu, x = U(11, (B, 256, 64)), U(12, (B, 256, 64)), U(13, (B, 16, 32, 32))
    dt = float(torch.tensor(0.62) - torch.tensor(0.6))
    xn, v = ops.unpatchify_cfg_euler(yc.cuda(), x.cuda(), dt, want_v=True)
    vref = OM.unpatchify(yc, 16, 16)
    assert torch.equal(v.cpu(), vref)
    torch.testing.assert_close(xn.cpu(), x - dt * vref, rtol=0, atol=1e-7)
    xn2, _ = ops.unpatchify_cfg_euler(yc.cuda(), x.cuda(), dt, y_uncond=yu.cuda(), cfg_scale=3.0)
    uref = OM.unpatchify(yu, 16, 16)
    torch.testing.assert_close(xn2.cpu(), x - dt * (uref + 3.0 * (vref - uref)), rtol=0, atol=1e-6)


def test_rmsnorm_rotary():
    """SURVEY 8a rows a31 / a32 against vectors produced by the REFERENCE's own RMSNorm class and apply_rotary_emb
    (tests/golden/rmsnorm_rotary.npz, tools/oracle/gen_golden.py rmsnorm_rotary); inputs regenerate from synth by seed."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rmsnorm_rotary.npz"))
    x, w = U(14, (7, 24, 64), -2, 2), U(15, (64,), 0.9, 1.1)
    torch.testing.assert_close(ops.rmsnorm(x.cuda(), w.cuda()).cpu(), torch.from_numpy(g["rms_affine"]), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ops.rmsnorm(x.cuda(), None).cpu(), torch.from_numpy(g["rms_plain"]), rtol=1e-5, atol=1e-6)
    x2 = U(18, (5, 3, 256), -4, 4)
    torch.testing.assert_close(ops.rmsnorm(x2.cuda(), None, eps=1e-5).cpu(), torch.from_numpy(g["rms_256"]), rtol=1e-5, atol=1e-6)
    t, f = U(16, (2, 3, 10, 32), -2, 2), U(17, (10, 32), -3, 3)
    torch.testing.assert_close(ops.rotary(t.cuda(), f.cuda()).cpu(), torch.from_numpy(g["rot_full"]), rtol=1e-5, atol=1e-5)
    f16 = U(19, (10, 16), -3, 3)
    torch.testing.assert_close(ops.rotary(t.cuda(), f16.cuda(), start_index=8).cpu(), torch.from_numpy(g["rot_partial_start8"]), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(ops.rotary(t.cuda(), f.cuda(), scale=0.5).cpu(), torch.from_numpy(g["rot_scaled"]), rtol=1e-5, atol=1e-5)
    with pytest.raises(ValueError):
        ops.rotary(t.cuda(), f.cuda(), start_index=8)


def _ref_joint_attention(ctx_qkv, x_qkv, H, kvis, see):
    """reference semantics: cat + bool mask + SDPA (sd3/mmdit.py:508-553, 1041-1094)"""
    from oracle import model as OM
    B, Kc, _ = ctx_qkv.shape
    nx = x_qkv.shape[1]
    qkv = torch.cat([ctx_qkv, x_qkv], dim=1)
    S = Kc + nx
    q, k, v = qkv.reshape(B, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    if kvis is None:
        mask = torch.ones(B, Kc, dtype=torch.bool)
    else:
        mask = torch.arange(Kc)[None] <= kvis[:, None]
    am = OM.joint_mask(mask, nx, see)
    a = F.scaled_dot_product_attention(q, k, v, attn_mask=am)
    return a.transpose(1, 2).reshape(B, S, H * 64)


ATTN_MODES = [0, ops.ATTN_F16X2]      # fp32-input MFMA kernel / f16x2-split kernel (csrc/attention.hip)


@pytest.mark.parametrize("mode", ATTN_MODES)
@pytest.mark.parametrize("see", [True, False])
@pytest.mark.parametrize("kv", [None, [511, 19], [300, 0]])
def test_joint_attention_vs_masked_sdpa(see, kv, mode):
    B, H, Kc, nx = 2, 3, 512, 256
    ctx = U(20, (B, Kc, 3 * H * 64), -1.5, 1.5)
    xs = U(21, (B, nx, 3 * H * 64), -1.5, 1.5)
    kvis = None if kv is None else torch.tensor(kv)
    ref = _ref_joint_attention(ctx, xs, H, kvis, see)
    cc, xc = ctx.cuda(), xs.cuda()
    D = H * 64
    o_c = torch.zeros(B, Kc, D, device="cuda")
    o_x = torch.zeros(B, nx, D, device="cuda")
    ops.attention((cc[..., :D], cc[..., D:2 * D], cc[..., 2 * D:], o_c), (xc[..., :D], xc[..., D:2 * D], xc[..., 2 * D:], o_x),
                  H, 64, kvis=None if kvis is None else kvis.int().cuda(), seg0_sees_seg1=see, mode=mode)
    torch.testing.assert_close(o_x.cpu(), ref[:, Kc:], rtol=2e-5, atol=2e-5)
    for b in range(B):
        live = Kc if kvis is None else int(kvis[b]) + 1
        torch.testing.assert_close(o_c[b, :live].cpu(), ref[b, :live], rtol=2e-5, atol=2e-5)
        assert torch.count_nonzero(o_c[b, live:]) == 0     # dead context rows are not written
    if mode == ops.ATTN_F16X2:      # outputs written as split activations (for the proj Linear): exactly the split of the fp32 outputs
        s_c = ops.SplitAct((B, Kc, D), "cuda", zero=True)
        s_x = ops.SplitAct((B, nx, D), "cuda", zero=True)
        ops.attention((cc[..., :D], cc[..., D:2 * D], cc[..., 2 * D:], s_c), (xc[..., :D], xc[..., D:2 * D], xc[..., 2 * D:], s_x),
                      H, 64, kvis=None if kvis is None else kvis.int().cuda(), seg0_sees_seg1=see, mode=mode)
        assert torch.equal(s_x.planes(), ops.split_f16x2(o_x).planes()) and torch.equal(s_c.planes(), ops.split_f16x2(o_c).planes())


@pytest.mark.parametrize("mode", ATTN_MODES)
def test_encoder_query_attention(mode):
    """queries attend to cat(to_query_kv(x), query_kv): 8 heads x 64, no mask (modules.py:255-266)"""
    B, N, K, H = 2, 256, 512, 8
    kv = U(22, (B, N, 2 * H * 64), -1.5, 1.5)
    qq = U(23, (B, K, 3 * H * 64), -1.5, 1.5)
    D = H * 64
    k2 = torch.cat([kv[..., :D], qq[..., D:2 * D]], dim=1)
    v2 = torch.cat([kv[..., D:], qq[..., 2 * D:]], dim=1)
    hd = lambda t: t.reshape(B, -1, H, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(hd(qq[..., :D]), hd(k2), hd(v2)).transpose(1, 2).reshape(B, K, D)
    kvc, qc = kv.cuda(), qq.cuda()
    o = torch.empty(B, K, D, device="cuda")
    ops.attention((None, kvc[..., :D], kvc[..., D:], None), (qc[..., :D], qc[..., D:2 * D], qc[..., 2 * D:], o), H, 64, mode=mode)
    torch.testing.assert_close(o.cpu(), ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("mode", ATTN_MODES)
def test_attention_ragged_lengths(mode):
    """K = 1024-token tokenizer length and a non-multiple-of-32 segment"""
    B, H = 1, 2
    for Kc, nx in ((1024, 256), (77, 45)):
        ctx, xs = U(24, (B, Kc, 3 * H * 64)), U(25, (B, nx, 3 * H * 64))
        ref = _ref_joint_attention(ctx, xs, H, None, True)
        cc, xc = ctx.cuda(), xs.cuda()
        D = H * 64
        o_c, o_x = torch.empty(B, Kc, D, device="cuda"), torch.empty(B, nx, D, device="cuda")
        ops.attention((cc[..., :D], cc[..., D:2 * D], cc[..., 2 * D:], o_c), (xc[..., :D], xc[..., D:2 * D], xc[..., 2 * D:], o_x), H, 64, mode=mode)
        torch.testing.assert_close(torch.cat([o_c, o_x], 1).cpu(), ref, rtol=2e-5, atol=2e-5)


def test_attention_f16x2_accuracy_gate_and_range_flag():
    """the f16x2-split attention may stand in for the fp32-MFMA kernel only if it is as close to the exact (fp64) softmax
    attention; operands beyond the fp16 range must raise the flag (bit 2), ordinary ones must not"""
    B, H, Kc, nx = 2, 4, 358, 256
    D = H * 64
    g = torch.Generator().manual_seed(7)
    ctx = torch.randn(B, Kc, 3 * D, generator=g) * 1.5
    xs = torch.randn(B, nx, 3 * D, generator=g) * 1.5
    qkv = torch.cat([ctx, xs], 1).double()
    q, k, v = qkv.reshape(B, Kc + nx, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, Kc + nx, D)
    cc, xc = ctx.cuda(), xs.cuda()
    errs = {}
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    for mode in ATTN_MODES:
        o_c, o_x = torch.empty(B, Kc, D, device="cuda"), torch.empty(B, nx, D, device="cuda")
        ops.attention((cc[..., :D], cc[..., D:2 * D], cc[..., 2 * D:], o_c), (xc[..., :D], xc[..., D:2 * D], xc[..., 2 * D:], o_x), H, 64,
                      mode=mode, overflow=flag)
        e = torch.cat([o_c, o_x], 1).cpu().double() - ref
        errs[mode] = (float(e.abs().max()), float(e.pow(2).mean().sqrt()))
    print("attention error vs fp64 (max, rms): fp32-MFMA", errs[0], " f16x2", errs[ops.ATTN_F16X2])
    assert int(flag.item()) == 0
    assert errs[ops.ATTN_F16X2][1] <= 2.0 * errs[0][1] + 1e-8 and errs[ops.ATTN_F16X2][0] <= 4.0 * errs[0][0] + 1e-7
    xc2 = xc.clone()
    xc2[0, 5, D + 7] = 1.0e5                                   # one key element beyond the fp16 range
    o_c, o_x = torch.empty(B, Kc, D, device="cuda"), torch.empty(B, nx, D, device="cuda")
    ops.attention((cc[..., :D], cc[..., D:2 * D], cc[..., 2 * D:], o_c), (xc2[..., :D], xc2[..., D:2 * D], xc2[..., 2 * D:], o_x), H, 64,
                  mode=ops.ATTN_F16X2, overflow=flag)
    assert int(flag.item()) & 4


def test_attention_head_dim16():
    B, N, H = 3, 256, 4
    qkv = U(26, (B, N, 3 * H * 16), -2, 2)
    q, k, v = qkv.reshape(B, N, 3, H, 16).permute(2, 0, 3, 1, 4)
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, H * 16)
    c = qkv.cuda()
    D = H * 16
    o = torch.empty(B, N, D, device="cuda")
    ops.attention(None, (c[..., :D], c[..., D:2 * D], c[..., 2 * D:], o), H, 16)
    torch.testing.assert_close(o.cpu(), ref, rtol=2e-5, atol=2e-5)


def test_groupnorm_silu_bf16():
    for C, HW in ((128, 64), (512, 32)):
        x = U(27, (2, C, HW, HW), -3, 3).bfloat16()
        w, b = U(28, (C,), 0.9, 1.1).bfloat16(), U(29, (C,), -0.1, 0.1).bfloat16()
        ref = F.silu(F.group_norm(x, 32, w, b, 1e-6))
        got = ops.groupnorm_silu(x.cuda(), w.cuda(), b.cuda()).cpu()
        d = (got.float() - ref.float()).abs()
        # bf16 outputs: allow one bf16 ulp (2^-8 relative) on a tiny fraction of elements
        assert float(d.max()) <= 0.03 and float((d > 0).float().mean()) < 0.02
        ref2 = F.group_norm(x, 32, w, b, 1e-6)
        got2 = ops.groupnorm_silu(x.cuda(), w.cuda(), b.cuda(), silu_act=False).cpu()
        assert float((got2.float() - ref2.float()).abs().max()) <= 0.03


def test_latent_format_and_clamp():
    from oracle import model as OM
    m = U(30, (2, 32, 32, 32), -3, 3).bfloat16()
    ref = OM.process_in(m[:, :16]).to(torch.float32)
    assert torch.equal(ops.latent_process_in(m.cuda()).cpu(), ref)
    z = U(31, (2, 16, 32, 32), -4, 4)
    assert torch.equal(ops.latent_process_out(z.cuda()).cpu(), OM.process_out(z).to(torch.bfloat16))
    img = U(32, (2, 3, 64, 64), -1.5, 1.5).bfloat16()
    assert torch.equal(ops.clamp01_(img.cuda().clone()).cpu(), OM.norm_ip(img))
