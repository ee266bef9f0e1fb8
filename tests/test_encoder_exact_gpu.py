"""-m gpu: csrc/encoder_exact.hip against its bit-for-bit CPU twin oracle/encoder_exact.c (itself pinned on the CPU to torch's own
fp32 ops and, end to end, to the pre-quantizer features of the REFERENCE pipeline runs: tests/test_encoder_exact_cpu.py), and the
whole encoder against the reference's goldens.  Everything here is BIT-EXACT: a single differing fp32 element fails."""
import os

import numpy as np
import pytest
import torch

from oracle import encoder_exact as EX
from selftoktokenizer_amd import ops, synth, weights as W
from selftoktokenizer_amd.schedule import DiTiCont

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _same(gpu: torch.Tensor, ref: np.ndarray, what: str, allow_nan_payload: bool = False):
    g = gpu.detach().cpu().numpy()
    assert g.shape == ref.shape, (g.shape, ref.shape)
    bad = g.view(np.uint32) != ref.view(np.uint32)
    bad &= ~((g == 0) & (ref == 0))                       # +0 / -0: equal values (torch's own kernels differ in the sign of an exact zero)
    if allow_nan_payload:
        bad &= ~(np.isnan(g) & np.isnan(ref))
    n = int(bad.sum())
    if n:
        i = np.argwhere(bad)[0]
        raise AssertionError(f"{what}: {n} of {g.size} fp32 elements differ from the oracle; first at {tuple(i)}: {g[tuple(i)]!r} vs {ref[tuple(i)]!r}")


def _rand(seed, shape, scale=1.0, shift=0.0):
    return (synth.hash_normalish(seed, shape) * scale + shift).float().contiguous()


# ---- transcendental building blocks ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("kind,cname", [("gelu_tanh", "xe_gelu_tanh1"), ("silu", "xe_silu1"), ("sleef_expf", "xe_sleef_expf"), ("sleef_tanhf", "xe_sleef_tanhf"),
                                         ("exp_u20", "xe_exp_u20")])
def test_unary_bit_patterns(kind, cname):
    """2^22 evenly spaced fp32 bit patterns (every exponent, both signs, inf / NaN) + a dense sweep of the working range"""
    bits = np.concatenate([np.arange(0, 2 ** 32, 2 ** 10, dtype=np.uint64).astype(np.uint32),
                           np.linspace(-12, 12, 1 << 20, dtype=np.float32).view(np.uint32),
                           np.array([0, 0x80000000, 0x7f800000, 0xff800000, 0x7fc00000, 0x00000001, 0x80000001, 0x7f7fffff, 0xff7fffff], dtype=np.uint32)])
    x = np.ascontiguousarray(bits.view(np.float32))
    fn = getattr(EX.lib(), cname)
    ref = np.array([fn(float(v)) for v in x[::64]], dtype=np.float32)            # ctypes scalar calls: a 1/64 sample ...
    out = ops.ex_unary(torch.from_numpy(x).cuda(), kind)
    torch.cuda.synchronize()
    _same(out[::64], ref, kind + " (scalar sample)", allow_nan_payload=True)
    if kind in ("gelu_tanh", "silu"):                                            # ... and everything through the vectorised entry points
        full = EX.gelu_tanh(x) if kind == "gelu_tanh" else EX.silu(x)
        _same(out, full, kind, allow_nan_payload=True)


# ---- Linear in MKL's order ------------------------------------------------------------------------------------------------------
LINEARS = [("attn.qkv 64->192", 64, 192), ("to_query_kv 64->1024", 64, 1024), ("query_linear 512->1536", 512, 1536), ("proj 64->64", 64, 64),
           ("mlp.fc1 64->256", 64, 256), ("mlp.fc2 256->64", 256, 64), ("query_proj 512->512", 512, 512), ("q_mlp.fc1 512->2048", 512, 2048),
           ("q_mlp.fc2 2048->512", 2048, 512), ("project_in 512->16", 512, 16), ("t_embedder 256->512", 256, 512), ("adaLN 512->3072", 512, 3072)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,K,N", LINEARS, ids=[c[0] for c in LINEARS])
def test_linear_mkl_order(name, K, N):
    M = 1024
    x = _rand(0xE0 + K, (M, K), 1.2, 0.05)
    w = _rand(0xE1 + N, (N, K), (1.0 / K) ** 0.5)
    b = _rand(0xE2, (N,), 0.2)
    ref = EX.linear(x.numpy(), w.numpy(), b.numpy())
    out = ops.ex_linear(x.cuda(), w.cuda(), b.cuda())
    torch.cuda.synchronize()
    _same(out, ref, name)


@pytest.mark.gpu
def test_linear_epilogues_and_views():
    """GELU, `res + gate * y` with per-token tables, a column slice of a fused projection as input, rows not a multiple of the tile"""
    M, K, N, T = 2 * 512 + 64, 512, 2048, 512
    x3 = _rand(0xF0, (M, 3 * K), 1.1)
    w = _rand(0xF1, (N, K), (1.0 / K) ** 0.5)
    b = _rand(0xF2, (N,), 0.2)
    xs = x3[:, K:2 * K]
    ref = EX.gelu_tanh(EX.linear(np.ascontiguousarray(xs.numpy()), w.numpy(), b.numpy()))
    out = ops.ex_linear(x3.cuda()[:, K:2 * K], w.cuda(), b.cuda(), gelu=True)
    _same(out, ref, "fc1 + GELU on a column slice")
    w2 = _rand(0xF3, (K, N), (1.0 / N) ** 0.5)
    b2 = _rand(0xF4, (K,), 0.2)
    res = _rand(0xF5, (M, K))
    table = _rand(0xF6, (T, 6 * K), 0.7)
    y = EX.linear(ref, w2.numpy(), b2.numpy())
    gate = table[:, 5 * K:6 * K].numpy()
    ref2 = res.numpy() + gate[np.arange(M) % T] * y
    out2 = ops.ex_linear(out, w2.cuda(), b2.cuda(), res=res.cuda(), gate=table.cuda()[:, 5 * K:6 * K], gate_mod=T)
    _same(out2, ref2, "q + gate * fc2(h)")
    ref3 = res.numpy() + y
    out3 = ops.ex_linear(out, w2.cuda(), b2.cuda(), res=res.cuda())
    torch.cuda.synchronize()
    _same(out3, ref3, "x + fc2(h)")


# ---- LayerNorm -------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("N,affine,mod", [(64, False, False), (512, False, True), (16, True, False), (1536, False, True)])
def test_layernorm_aten_order(N, affine, mod):
    rows, T = 3 * 512, 512
    x = _rand(0xA0 + N, (rows, N), 2.0, 0.3)
    g = _rand(0xA1, (N,), 0.1, 1.0) if affine else None
    b = _rand(0xA2, (N,), 0.1) if affine else None
    ref, st = EX.layernorm(x.numpy(), None if g is None else g.numpy(), None if b is None else b.numpy(), want_stats=True)
    table = _rand(0xA3, (T, 6 * N), 0.5)
    if mod:
        sh, sc = table[:, 3 * N:4 * N].numpy(), table[:, 4 * N:5 * N].numpy()
        ref = ref * (np.float32(1) + sc[np.arange(rows) % T]) + sh[np.arange(rows) % T]
    tc = table.cuda()
    out, stats = ops.ex_layernorm_mod(x.cuda(), shift=tc[:, 3 * N:4 * N] if mod else None, scale=tc[:, 4 * N:5 * N] if mod else None,
                                      gamma=None if g is None else g.cuda(), beta=None if b is None else b.cuda(), want_stats=True)
    torch.cuda.synchronize()
    _same(stats, st, f"LayerNorm({N}) mean / rstd")
    _same(out, ref, f"LayerNorm({N})")


@pytest.mark.gpu
@pytest.mark.parametrize("per_sample,bias_last,with_gate", [(False, True, True), (True, False, True), (False, False, False)])
def test_residual_update_fused_into_layernorm(per_sample, bias_last, with_gate):
    """round 6: x' = x + gate * (Linear(.) [+ bias last]) inside the LayerNorm + modulate pass = the Linear's res + gate epilogue followed by ex_layernorm_mod, bit for bit
    (the MMDiT's per-token context tables and per-sample image tables, sd3/mmdit.py:485-496), incl. x' written in place"""
    B, T, N, K = 3, 200, 1536, 1536
    x = _rand(0xD0, (B, T, N), 1.5).cuda()
    a = _rand(0xD1, (B, T, K), 1.0).cuda()
    w = _rand(0xD2, (N, K), (1.0 / K) ** 0.5).cuda()
    b = _rand(0xD3, (N,), 0.2).cuda()
    tab = _rand(0xD4, (B if per_sample else T, 6 * N), 0.6).cuda()
    gate = tab[:, 2 * N:3 * N] if with_gate else None
    gm = (-T if per_sample else T) if with_gate else 0
    sh, sc = tab[:, 3 * N:4 * N], tab[:, 4 * N:5 * N]
    x1 = ops.ex_linear(a, w, b, res=x, gate=gate, gate_mod=gm, bias_last=bias_last, kernel="xe")
    n1 = ops.ex_layernorm_mod(x1, shift=sh, scale=sc, per_sample=per_sample)
    y = ops.ex_linear(a, w, None if bias_last else b, kernel="xe")
    x2, n2 = ops.ex_res_layernorm_mod(x, y, lin_bias=b if bias_last else None, gate=gate, gate_mod=gm, shift=sh, scale=sc, per_sample=per_sample)
    torch.cuda.synchronize()
    _same(x2, x1.cpu().numpy(), "x'")
    _same(n2, n1.cpu().numpy(), "LayerNorm(x') modulated")
    xi = x.clone()
    x3, n3 = ops.ex_res_layernorm_mod(xi, y, lin_bias=b if bias_last else None, gate=gate, gate_mod=gm, shift=sh, scale=sc, per_sample=per_sample, x_out=xi)
    assert x3.data_ptr() == xi.data_ptr()
    _same(x3, x1.cpu().numpy(), "x' in place")
    _same(n3, n1.cpu().numpy(), "LayerNorm(x') modulated, in place")


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1536, 1024, 2048])
def test_wide_row_layernorm_equals_the_oracle(N):
    """round 6: rows >= 2048 of N % 512 == 0 columns run on xe_lnw_kernel (a row's chunks spread over N / 512 waves, the cascade replayed from LDS): against the CPU oracle
    (ATen's bits) -- gamma / beta + statistics, a per-token modulation table, and the fused residual form against the separately rounded fp32 operations; a ragged last
    workgroup (rows % 8 != 0); and the same bits as the 8-threads-per-row kernel that serves fewer rows"""
    rows, T = 2051, 293                                        # 7 x 293 rows
    x = _rand(0xE0 + N, (rows, N), 1.5, 0.1)
    g, b = _rand(0xE1, (N,), 0.3, 1.0), _rand(0xE2, (N,), 0.2)
    ref, st = EX.layernorm(x.numpy(), g.numpy(), b.numpy(), want_stats=True)
    out, stats = ops.ex_layernorm_mod(x.cuda(), gamma=g.cuda(), beta=b.cuda(), want_stats=True)
    _same(out, ref, f"wide-row LayerNorm N = {N}")
    _same(stats, st, "mean / rstd")
    _same(ops.ex_layernorm_mod(x[:1000].cuda(), gamma=g.cuda(), beta=b.cuda()), ref[:1000], "the narrow kernel on the same rows")
    tab = _rand(0xE3, (T, 2 * N), 0.6)
    sh, sc = tab[:, :N], tab[:, N:]
    plain = EX.layernorm(x.numpy())
    idx = np.arange(rows) % T
    want = plain * (np.float32(1) + sc.numpy()[idx]) + sh.numpy()[idx]
    tc = tab.cuda()
    _same(ops.ex_layernorm_mod(x.cuda(), shift=tc[:, :N], scale=tc[:, N:]), want, "modulated, per-token table")
    lin, bias, gate = _rand(0xE4, (rows, N), 1.0), _rand(0xE5, (N,), 0.2), _rand(0xE6, (T, N), 0.7)
    x1 = x.numpy() + gate.numpy()[idx] * (lin.numpy() + bias.numpy())                      # three separately rounded fp32 operations
    want2 = EX.layernorm(x1) * (np.float32(1) + sc.numpy()[idx]) + sh.numpy()[idx]
    x2, n2 = ops.ex_res_layernorm_mod(x.cuda(), lin.cuda(), lin_bias=bias.cuda(), gate=gate.cuda(), gate_mod=T, shift=tc[:, :N], scale=tc[:, N:])
    _same(x2, x1, "x' (fused residual update)")
    _same(n2, want2, "LayerNorm(x') modulated")


# ---- attention -------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name,B,H,Tq,Tk1,Tk2,D", [("latent self-attention 4 x 16", 2, 4, 256, 256, 0, 16), ("query attention 8 x 64, 256 + 512 keys", 2, 8, 512, 256, 512, 64),
                                                  ("K = 1024 tokenizer: 256 + 1024 keys", 1, 8, 1024, 256, 1024, 64), ("128 px: 64 + 512 keys", 1, 8, 512, 64, 512, 64),
                                                  ("320 px: 400 keys (MKL halves of 200)", 1, 4, 400, 400, 0, 16), ("320 px: 400 + 512 keys", 1, 8, 512, 400, 512, 64)])
def test_attention_flash_order(name, B, H, Tq, Tk1, Tk2, D):
    HD = H * D
    qq = _rand(0xB0 + Tq, (B, Tq, 3 * HD), 1.4)                     # fused projections, as the encoder holds them
    kvx = _rand(0xB1 + Tk1, (B, Tk1, 2 * HD), 1.4)
    if Tk2:
        ref = EX.attention(qq[..., :HD].numpy(), kvx[..., :HD].numpy(), kvx[..., HD:].numpy(), H, qq[:, :Tk2, HD:2 * HD].numpy(), qq[:, :Tk2, 2 * HD:].numpy())
        qc, kc = qq.cuda(), kvx.cuda()
        out = ops.ex_attention(qc[..., :HD], kc[..., :HD], kc[..., HD:], H, qc[:, :Tk2, HD:2 * HD], qc[:, :Tk2, 2 * HD:])
    else:
        ref = EX.attention(qq[..., :HD].numpy(), qq[..., HD:2 * HD].numpy(), qq[..., 2 * HD:].numpy(), H)
        qc = qq.cuda()
        out = ops.ex_attention(qc[..., :HD], qc[..., HD:2 * HD], qc[..., 2 * HD:], H)
    torch.cuda.synchronize()
    _same(out, ref, name)


@pytest.mark.gpu
@pytest.mark.parametrize("slots,valid,Tk2,Tq", [(512, 512, 256, 512), (512, 358, 256, 358), (512, 358, 256, 256), (512, 100, 256, 100), (512, 20, 256, 256), (512, 0, 256, 256),
                                               (512, 300, 0, 300), (1024, 750, 256, 750), (1024, 300, 256, 256), (1024, 1024, 256, 200)])
def test_attention_fused_equals_unfused_with_a_prefix_mask(slots, valid, Tk2, Tq):
    """the MMDiT's joint attention (sd3/mmdit.py:508-553 with the bool prefix mask of models_ours.py:353): `slots` context key slots of which the first `valid` are
    visible (ragged tiles, fully masked tiles and kv blocks, no visible context key at all = cfg_inference), then Tk2 image keys (0: the context rows of the renderer
    see no image key); Tq rows that are no multiple of the 128-row tile.  One kernel (round 6) vs scores GEMM -> row pass -> P V GEMM (round 5): the same bits."""
    B, H, D = 2, 3, 64
    HD = H * D
    q = _rand(0xC0 + Tq, (B, Tq, 3 * HD), 1.3).cuda()
    ctx = _rand(0xC1 + valid, (B, max(valid, 1), 3 * HD), 1.3).cuda()
    x = _rand(0xC2 + Tk2, (B, max(Tk2, 1), 3 * HD), 1.3).cuda()
    k1, v1 = (ctx[..., HD:2 * HD], ctx[..., 2 * HD:]) if valid else (None, None)
    k2, v2 = (x[..., HD:2 * HD], x[..., 2 * HD:]) if Tk2 else (None, None)
    a = ops.ex_attention(q[..., :HD], k1, v1, H, k2, v2, slots1=slots, kernel="fused")
    b = ops.ex_attention(q[..., :HD], k1, v1, H, k2, v2, slots1=slots, kernel="unfused")
    torch.cuda.synchronize()
    assert bool(torch.isfinite(a).all())
    _same(a, b.cpu().numpy(), f"fused vs unfused, {valid} of {slots} context keys + {Tk2}, {Tq} rows")


@pytest.mark.gpu
def test_unfused_attention_runs_a_large_batch_in_slices(monkeypatch):
    """the unfused route's score workspace is bounded -- above `ops.EX_ATTENTION_WS_LIMIT` the batch runs in slices (rows are independent: the same bits)"""
    B, H, D, Tq, valid, Tk2 = 5, 3, 64, 200, 130, 256
    HD = H * D
    q = _rand(0xD0, (B, Tq, 3 * HD), 1.3).cuda()
    ctx = _rand(0xD1, (B, valid, 3 * HD), 1.3).cuda()
    x = _rand(0xD2, (B, Tk2, 3 * HD), 1.3).cuda()
    args = (q[..., :HD], ctx[..., HD:2 * HD], ctx[..., 2 * HD:], H, x[..., HD:2 * HD], x[..., 2 * HD:])
    whole = ops.ex_attention(*args, slots1=512, kernel="unfused")
    per = int(ops._lib.load().selftok_ex_attention_workspace_bytes(1, H, Tq, 512 + Tk2, D))
    monkeypatch.setattr(ops, "EX_ATTENTION_WS_LIMIT", 2 * per)                   # slices of 2, 2, 1 samples
    sliced = ops.ex_attention(*args, slots1=512, kernel="unfused")
    torch.cuda.synchronize()
    _same(sliced, whole.cpu().numpy(), "unfused attention, batch in slices of two")


# ---- the whole encoder against the REFERENCE's own runs --------------------------------------------------------------------------
@pytest.fixture(scope="module")
def encoder():
    from selftoktokenizer_amd.encoder import QformerEncoderGPU
    shapes = {k: v for k, v in W.expected_shapes(512).items() if k.startswith("encoder.")}
    return QformerEncoderGPU(W.synthetic_state_dict(shapes), torch.device("cuda", torch.cuda.current_device()), 512, mode="exact")


def test_position_table_is_the_reference_formula():
    """the shipped position table (the reference's `timestep_embedding` of 1000 + 8 k on the build container) against the formula evaluated
    in fp64 here: MKL's VML is not correctly rounded (3 of the 128 frequencies are 1 ulp off -> up to 4e-4 in their cos / sin columns, the other
    columns within a few 1e-8), but it IS the formula -- a corrupted or mis-indexed table fails"""
    import math
    from selftoktokenizer_amd.encoder import encoder_pos_embedding
    t = encoder_pos_embedding(1024).numpy()
    k = np.arange(128, dtype=np.float32)
    freqs = np.exp((np.float32(-math.log(10000)) * k / np.float32(128)).astype(np.float64))
    args = (1000 + 8 * np.arange(1024)).astype(np.float32)[:, None] * freqs.astype(np.float32)[None]
    want = np.concatenate([np.cos(args.astype(np.float64)), np.sin(args.astype(np.float64))], axis=1)
    d = np.abs(t.astype(np.float64) - want).max(axis=0)
    assert int((d > 3e-7).sum()) <= 6 and float(d.max()) < 1e-3, (np.sort(d)[-8:])
    assert torch.equal(encoder_pos_embedding(512), encoder_pos_embedding(1024)[:512])


@pytest.mark.gpu
def test_features_equal_reference_16_images(encoder):
    g = np.load(os.path.join(GOLD, "pipeline_b16.npz"))
    x0 = torch.from_numpy(g["x0_bf16"]).view(torch.bfloat16).float().cuda()
    z = encoder.features(x0)
    torch.cuda.synchronize()
    _same(z, g["z"], "pre-quantizer features vs the reference pipeline's 16-image run")
    ids = encoder(x0)[1].cpu().numpy()
    assert np.array_equal(ids, g["tokens"].astype(np.int64))


@pytest.mark.gpu
def test_pre_norm_features_and_ids_equal_the_reference():
    """encoder_config.pre_norm = True (models_ours.py:219-220; no shipped config sets it; implemented in round 6): the reference Encoder built with the flag on
    (tests/golden/encoder_prenorm_b8.npz, the first 8 latents of its 64-image run): exact mode features 0 differing bits, ids equal; fast mode within 1e-4; the
    pipeline takes the flag from the YAML's encoder_config"""
    from selftoktokenizer_amd.encoder import QformerEncoderGPU
    from selftoktokenizer_amd.config import default_config
    from selftoktokenizer_amd.pipeline import SelftokPipeline
    g, g64 = np.load(os.path.join(GOLD, "encoder_prenorm_b8.npz")), np.load(os.path.join(GOLD, "encode_b64.npz"))
    x0 = torch.from_numpy(g64["x0_bf16"][:8]).view(torch.bfloat16).float().cuda()
    shapes = {k: v for k, v in W.expected_shapes(512).items() if k.startswith("encoder.")}
    dev = torch.device("cuda", torch.cuda.current_device())
    enc = QformerEncoderGPU(W.synthetic_state_dict(shapes), dev, 512, mode="exact", pre_norm=True)
    z = enc.features(x0)
    _same(z, g["z"], "pre_norm features vs the reference's")
    assert np.array_equal(enc(x0)[1].cpu().numpy(), g["ids"].astype(np.int64))
    assert float((z.cpu() - torch.from_numpy(g64["z"][:8])).abs().max()) > 0.1                     # the flag matters
    fast = QformerEncoderGPU(W.synthetic_state_dict(shapes), dev, 512, mode="fast", pre_norm=True)
    assert float((fast.features(x0).cpu() - torch.from_numpy(g["z"])).abs().max()) < 1e-4
    cfg = default_config(512)
    cfg.tokenizer.params.encoder_config.pre_norm = True
    pipe = SelftokPipeline(cfg, None, None, device="cuda", state_dict=W.synthetic_state_dict(W.expected_shapes(512), device="cuda"),
                           vae_state_dict=W.synthetic_vae_state_dict(device="cuda"), verbose=False)
    assert pipe.model.encoder.pre_norm and np.array_equal(pipe.model.encoder(x0[:2])[1].cpu().numpy(), g["ids"][:2].astype(np.int64))


@pytest.mark.gpu
def test_features_and_ids_equal_reference_64_images_and_batch_invariance(encoder):
    """BASELINE configs[1]'s batch: the reference's `encoding` on 64 images in ONE batch (tests/golden/encode_b64.npz): features bit-equal,
    ids 32768 / 32768; the same 64 latents as 4 x 16, 8 x 8 and 64 x 1 give IDENTICAL features (every kernel is row-independent)"""
    g = np.load(os.path.join(GOLD, "encode_b64.npz"))
    x0 = torch.from_numpy(g["x0_bf16"]).view(torch.bfloat16).float().cuda()
    z = encoder.features(x0)
    _same(z, g["z"], "pre-quantizer features vs the reference's B = 64 run")
    ids = encoder(x0)[1]
    assert np.array_equal(ids.cpu().numpy(), g["tokens"].astype(np.int64))
    for gsz in (16, 8, 1):
        zg = torch.cat([encoder.features(x0[i:i + gsz]) for i in range(0, 64, gsz)])
        assert torch.equal(zg, z), f"features depend on the batch size ({64 // gsz} x {gsz})"
    assert torch.equal(torch.cat([encoder(x0[i:i + 8])[1] for i in range(0, 64, 8)]), ids)
