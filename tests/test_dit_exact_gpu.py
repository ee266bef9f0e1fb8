"""-m gpu: the exact-order MMDiT mode (`gemm="exact"`, round 5): every Linear / LayerNorm / GELU / SiLU / attention of `MMDiT.forward`
(sd3/mmdit.py:992-1101) as the sequence of fp32 operations torch-CPU executes for the reference.  Kernels against the C twin
(oracle/encoder_exact.c: the masked joint attention, per-sample tables), one forward against the reference's own (tests/golden/dit_forward_b16.npz:
crc32 of the image stream after every joint block and of the velocity), and the 16-image pipeline run end to end: final latents, pixels and PSNR
EQUAL to the reference's, bit for bit."""
import os
import zlib

import numpy as np
import pytest
import torch

from oracle import encoder_exact as EX
from selftoktokenizer_amd import ops, synth, weights as W
from selftoktokenizer_amd.config import default_config

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _same(gpu, ref, what):
    g = gpu.detach().cpu().numpy()
    bad = (g.view(np.uint32) != ref.view(np.uint32)) & ~((g == 0) & (ref == 0))
    assert int(bad.sum()) == 0, f"{what}: {int(bad.sum())} of {g.size} fp32 elements differ from the oracle"


def _rand(seed, shape, scale=1.0):
    return (synth.hash_normalish(seed, shape) * scale).float().contiguous()


@pytest.mark.parametrize("valid,see_x", [(512, True), (301, True), (20, True), (256, True), (257, False), (0, True)])
def test_joint_attention_with_prefix_mask(valid, see_x):
    """context rows and image rows of the joint attention: K = 512 context slots of which `valid` are visible (held as `valid` rows), 256 image keys;
    24 heads x 64 as in the MMDiT (2 heads here)"""
    B, H, D, K, nx = 2, 2, 64, 512, 256
    HD = H * D
    cq = _rand(0x30 + valid, (B, max(valid, 1), 3 * HD), 1.3)[:, :valid].contiguous() if valid else None
    xq = _rand(0x31, (B, nx, 3 * HD), 1.3)
    xk, xv = xq[..., HD:2 * HD], xq[..., 2 * HD:]
    xc = xq.cuda()
    if valid:
        cc = cq.cuda()
        ref_x = EX.attention(xq[..., :HD].numpy(), cq[..., HD:2 * HD].numpy(), cq[..., 2 * HD:].numpy(), H, xk.numpy(), xv.numpy(), valid1=valid, slots1=K)
        out_x = ops.ex_attention(xc[..., :HD], cc[..., HD:2 * HD], cc[..., 2 * HD:], H, xc[..., HD:2 * HD], xc[..., 2 * HD:], slots1=K)
        _same(out_x, ref_x, f"image rows, {valid} of {K} context keys visible")
        k2 = (xk.numpy(), xv.numpy()) if see_x else (None, None)
        ref_c = EX.attention(cq[..., :HD].numpy(), cq[..., HD:2 * HD].numpy(), cq[..., 2 * HD:].numpy(), H, k2[0], k2[1], valid1=valid, slots1=K)
        out_c = ops.ex_attention(cc[..., :HD], cc[..., HD:2 * HD], cc[..., 2 * HD:], H, xc[..., HD:2 * HD] if see_x else None, xc[..., 2 * HD:] if see_x else None, slots1=K)
        _same(out_c, ref_c, f"context rows, {valid} of {K} visible, see_x={see_x}")
    else:       # cfg_inference: no context key visible at all -- the first kv block is fully masked
        dummy = np.zeros((B, 16, HD), np.float32)
        ref_x = EX.attention(xq[..., :HD].numpy(), dummy, dummy, H, xk.numpy(), xv.numpy(), valid1=0, slots1=K)
        out_x = ops.ex_attention(xc[..., :HD], None, None, H, xc[..., HD:2 * HD], xc[..., 2 * HD:], slots1=K)
        _same(out_x, ref_x, "image rows, no context key visible")


def test_per_sample_tables():
    """the image stream's modulation ('t_emb': one [6H] row per sample): LayerNorm + modulate and the gated residual epilogue index their tables by sample"""
    B, T, N = 3, 256, 1536
    x = _rand(0x40, (B, T, N), 2.0)
    tab = _rand(0x41, (B, 6 * N), 0.5)
    ref = EX.layernorm(x.numpy()) * (np.float32(1) + tab[:, None, N:2 * N].numpy()) + tab[:, None, 0:N].numpy()
    tc = tab.cuda()
    out = ops.ex_layernorm_mod(x.cuda(), shift=tc[:, 0:N], scale=tc[:, N:2 * N], per_sample=True)
    _same(out, ref, "LayerNorm + per-sample modulate")
    w, b = _rand(0x42, (512, N), N ** -0.5), _rand(0x43, (512,), 0.1)
    y = EX.linear(ref, w.numpy(), b.numpy())
    res = _rand(0x44, (B, T, 512))
    g = _rand(0x45, (B, 6 * 512), 0.7)
    ref2 = res.numpy() + g[:, None, 2 * 512:3 * 512].numpy() * y
    out2 = ops.ex_linear(out, w.cuda(), b.cuda(), res=res.cuda(), gate=g.cuda()[:, 2 * 512:3 * 512], gate_mod=-T)
    _same(out2, ref2, "x + gate[sample] * Linear")


@pytest.fixture(scope="module")
def models():
    from selftoktokenizer_amd.encoder import QformerEncoderGPU
    from selftoktokenizer_amd.mmdit import MMDiTGPU
    sd = W.synthetic_state_dict(W.expected_shapes(512), device="cuda")
    dev = torch.device("cuda", torch.cuda.current_device())
    enc = QformerEncoderGPU(sd, dev, 512, mode="exact")
    dit = MMDiTGPU(sd, dev, 512, gemm="exact")
    return sd, enc, dit


def test_forward_16_images_equals_the_reference_block_by_block(models):
    """MMDiT.forward at B = 16, three scheduled timesteps (k = 511, 302 ..): the image stream after EVERY joint block and the velocity have the
    reference's crc32 (tests/golden/dit_forward_b16.npz)"""
    from selftoktokenizer_amd.pipeline import _Flow
    sd, enc, dit = models
    g = np.load(os.path.join(GOLD, "dit_forward_b16.npz"))
    B = 16
    ehs = enc.codes_ln(torch.from_numpy(synth.synthetic_token_ids(B)).cuda())
    x = synth.synthetic_noise(B, device="cuda")
    flow = _Flow(50, 1.0, dit.device)
    ctx0 = dit.embed_context(ehs)
    for j, i in enumerate(g["steps"]):
        k = int(g[f"k_{j}"])
        tf = flow.t_freq_exact[i:i + 1].expand(B, -1).contiguous()
        trace = []
        dit._trace = trace
        try:
            y = dit.velocity_tokens(x, tf, ctx0, k + 1, True)
        finally:
            dit._trace = None
        _, v = ops.unpatchify_cfg_euler(y, None, 0.0, C=16, hp=16, wp=16)
        crcs = np.array([zlib.crc32(t.contiguous().cpu().numpy().tobytes()) for t in trace], dtype=np.uint32)
        first = np.nonzero(crcs != g[f"xcrc_{j}"])[0]
        if first.size:
            b = int(first[0])
            print(f"step {i} (k = {k}): first differing block {b}; head ours {trace[b][0, 0, :8].cpu().numpy()} reference {g[f'xhead_{j}'][b]}")
        sub = v[:, :, ::4, ::4].contiguous().cpu().numpy()
        print(f"step {i} (k = {k}): velocity sub-sample max abs diff vs the reference {np.abs(sub - g[f'vsub_{j}']).max():.3e}")
        assert first.size == 0, f"step {i}: the image stream differs from the reference's from block {int(first[0])} on"
        assert zlib.crc32(v.contiguous().cpu().numpy().tobytes()) == int(g[f"vcrc_{j}"])


def test_pipeline_16_images_pixels_equal_the_reference_bit_for_bit():
    """the reference pipeline's own 16-image run (pipeline_b16.npz, decode_b16.npz): with gemm='exact' (+ the exact VAE and encoder) the final latents
    of the 50-step loop, the decoded pixels and hence the PSNR of every image EQUAL the reference's"""
    from mimogpt.infer.SelftokPipeline import SelftokPipeline
    from selftoktokenizer_amd import evaluate as E
    g, gd = np.load(os.path.join(GOLD, "pipeline_b16.npz")), np.load(os.path.join(GOLD, "decode_b16.npz"))
    sd = W.synthetic_state_dict(W.expected_shapes(512), device="cuda")
    pipe = SelftokPipeline(default_config(512), None, None, device="cuda", state_dict=sd, vae_state_dict=W.synthetic_vae_state_dict(device="cuda"), verbose=False, gemm="exact")
    imgs = synth.synthetic_images(16, device="cuda")
    ids = pipe.encoding(imgs)
    assert np.array_equal(ids.cpu().numpy(), g["tokens"].astype(np.int64))
    rec, lat = pipe.decoding(ids.cpu().numpy(), noise=synth.synthetic_noise(16), return_latent=True)
    d = (lat.cpu() - torch.from_numpy(g["lat"])).abs().max()
    print(f"\nfinal latents after 50 exact-order steps vs the reference: max abs diff {float(d):.3e}; differing elements {int((lat.cpu() != torch.from_numpy(g['lat'])).sum())}")
    assert torch.equal(lat.cpu(), torch.from_numpy(g["lat"])), "final latents differ from the reference's"
    bits = rec.cpu().view(torch.int16).numpy().view(np.uint16)
    crc = np.array([zlib.crc32(np.ascontiguousarray(bits[i]).tobytes()) for i in range(16)], dtype=np.uint32)
    assert np.array_equal(crc, gd["crc"]), "pixels differ from the reference's"
    psnr = E.psnr_each(rec, imgs)
    assert np.array_equal(psnr, g["psnr_ref"]), np.abs(psnr - g["psnr_ref"]).max()
    print("pixels of all 16 images equal the reference's; PSNR identical:", psnr[:4])


@pytest.mark.parametrize("R", [128, 320])
def test_pipeline_at_other_image_sizes_equals_the_reference_bit_for_bit(R):
    """`datasize` = 128 / 320 (`enable_enc_variable_size`: cropped position embeddings, 64 / 400 image tokens; models_ours.py:183-202, SelftokPipeline.py:262):
    the reference pipeline's own 16-image run at that size (tests/golden/res{R}_b16.npz, tools/oracle/gen_golden.py res{R}) -- VAE latents, ids from
    pixels, the final latents of the 50-step loop, every image's pixels and PSNR EQUAL the reference's with the three exact modes.  oneDNN's chunk
    order, ATen's GroupNorm cascade and the flash kernel's kv blocks all depend on the layer sizes: DESIGN.md section 15.9."""
    from mimogpt.infer.SelftokPipeline import SelftokPipeline
    from selftoktokenizer_amd import evaluate as E
    g = np.load(os.path.join(GOLD, f"res{R}_b16.npz"))
    sd = W.synthetic_state_dict(W.expected_shapes(512), device="cuda")
    pipe = SelftokPipeline(default_config(512), None, None, datasize=R, device="cuda", state_dict=sd, vae_state_dict=W.synthetic_vae_state_dict(device="cuda"), verbose=False,
                           gemm="exact")
    assert pipe.vae.mode == "exact"
    imgs = synth.synthetic_images(16, size=R, device="cuda")
    x0 = pipe.encode_latents(imgs)
    ref = torch.from_numpy(g["x0_bf16"]).view(torch.bfloat16).float()
    bad = int((x0.cpu() != ref).sum())
    assert bad == 0, f"{R} px: {bad} of {ref.numel()} VAE latent elements differ from the reference pipeline's"
    ids = pipe.encoding(imgs)
    flips = int((ids.cpu().numpy() != g["tokens"].astype(np.int64)).sum())
    print(f"\n{R} px: VAE latents bit-equal; token ids from pixels vs the reference: {ids.numel() - flips} / {ids.numel()}")
    assert flips == 0
    assert torch.equal(pipe.encode_latents(imgs[5:6]), x0[5:6])                        # batch independent
    noise = synth.hash_normalish(0xA0 + R, (16, 16, R // 8, R // 8), "cpu")
    rec, lat = pipe.decoding(ids.cpu().numpy(), noise=noise, return_latent=True)
    diff = int((lat.cpu() != torch.from_numpy(g["lat"])).sum())
    print(f"{R} px: final latents after 50 exact-order steps: {diff} differing elements")
    assert diff == 0
    assert tuple(rec.shape) == (16, 3, R, R)
    bits = rec.cpu().view(torch.int16).numpy().view(np.uint16)
    crc = np.array([zlib.crc32(np.ascontiguousarray(bits[i]).tobytes()) for i in range(16)], dtype=np.uint32)
    assert np.array_equal(crc, g["crc"]), f"{R} px: pixels differ from the reference's"
    assert np.array_equal(E.psnr_each(rec, imgs), g["psnr_ref"])


def test_renderer_16_images_equals_the_reference_bit_for_bit():
    """`decoding_with_renderer` (SelftokPipeline.py:296-322; MMDiT_Renderer.forward sd3/mmdit.py:1511-1620: mask_token + positional_embedding, t = 1000,
    context rows cannot see the image rows, one pass) with gemm='exact': the reference's own 16-row run (tests/golden/renderer_b16.npz,
    tools/oracle/gen_golden.py renderer16) -- the one-pass latent and every image's pixels EQUAL the reference's"""
    from mimogpt.infer.SelftokPipeline import SelftokPipeline
    g = np.load(os.path.join(GOLD, "renderer_b16.npz"))
    sd = W.synthetic_state_dict(W.expected_shapes(512, renderer=True), device="cuda")
    pipe = SelftokPipeline(default_config(512, renderer=True), None, None, device="cuda", state_dict=sd, vae_state_dict=W.synthetic_vae_state_dict(device="cuda"),
                           verbose=False, gemm="exact")
    assert pipe.model.model.gemm == "exact" and pipe.model.model.renderer
    rec, lat = pipe.decoding_with_renderer(g["ids"], return_latent=True)
    diff = int((lat.cpu() != torch.from_numpy(g["latent"])).sum())
    print(f"\nrenderer, 16 id rows: {diff} of {lat.numel()} latent elements differ from the reference's")
    assert diff == 0
    bits = rec.cpu().view(torch.int16).numpy().view(np.uint16)
    crc = np.array([zlib.crc32(np.ascontiguousarray(bits[i]).tobytes()) for i in range(16)], dtype=np.uint32)
    assert np.array_equal(crc, g["crc"]), "renderer pixels differ from the reference's"
    # rows are independent: 3 of the 16 alone give the same latents (the reference would not: MKL's path below 16 rows)
    _, lat3 = pipe.decoding_with_renderer(g["ids"][4:7], return_latent=True)
    assert torch.equal(lat3, lat[4:7])


def test_pipeline_k1024_16_images_equals_the_reference_bit_for_bit():
    """BASELINE configs[2] end to end (K = 1024; stage split 384,368,144,96,32): the reference pipeline's own 16-image run
    (tests/golden/k1024_pipe_b16.npz, tools/oracle/gen_golden.py k1024_pipe16) -- ids from pixels, final latents of the 50-step loop, pixels, PSNR"""
    from mimogpt.infer.SelftokPipeline import SelftokPipeline
    from selftoktokenizer_amd import evaluate as E
    g = np.load(os.path.join(GOLD, "k1024_pipe_b16.npz"))
    sd = W.synthetic_state_dict(W.expected_shapes(1024), device="cuda")
    pipe = SelftokPipeline(default_config(1024), None, None, device="cuda", state_dict=sd, vae_state_dict=W.synthetic_vae_state_dict(device="cuda"), verbose=False, gemm="exact")
    imgs = synth.synthetic_images(16, device="cuda")
    ids = pipe.encoding(imgs)
    flips = int((ids.cpu().numpy() != g["tokens"].astype(np.int64)).sum())
    print(f"\nK = 1024: token ids from pixels vs the reference: {ids.numel() - flips} / {ids.numel()}")
    assert flips == 0
    rec, lat = pipe.decoding(ids.cpu().numpy(), noise=synth.synthetic_noise(16), return_latent=True)
    diff = int((lat.cpu() != torch.from_numpy(g["lat"])).sum())
    print(f"K = 1024: final latents after 50 exact-order steps: {diff} differing elements")
    assert diff == 0
    bits = rec.cpu().view(torch.int16).numpy().view(np.uint16)
    crc = np.array([zlib.crc32(np.ascontiguousarray(bits[i]).tobytes()) for i in range(16)], dtype=np.uint32)
    assert np.array_equal(crc, g["crc"]), "K = 1024: pixels differ from the reference's"
    assert np.array_equal(E.psnr_each(rec, imgs), g["psnr_ref"])


def test_pipeline_64_images_in_one_batch_equals_the_reference_bit_for_bit():
    """BASELINE configs[1] at its configured batch: the reference pipeline's own run of 64 images in ONE batch (tests/golden/pipeline_b64.npz,
    tools/oracle/gen_golden.py pipeline64: ~3 h of CPU) -- a crc32 of every image's final latent after 50 steps and of its bf16 pixels, PSNR of every image"""
    from mimogpt.infer.SelftokPipeline import SelftokPipeline
    from selftoktokenizer_amd import evaluate as E
    g, g64 = np.load(os.path.join(GOLD, "pipeline_b64.npz")), np.load(os.path.join(GOLD, "encode_b64.npz"))
    sd = W.synthetic_state_dict(W.expected_shapes(512), device="cuda")
    pipe = SelftokPipeline(default_config(512), None, None, device="cuda", state_dict=sd, vae_state_dict=W.synthetic_vae_state_dict(device="cuda"), verbose=False, gemm="exact")
    imgs = synth.synthetic_images(64, device="cuda")
    ids = pipe.encoding(imgs)
    assert np.array_equal(ids.cpu().numpy(), g64["tokens"].astype(np.int64))
    rec, lat = pipe.decoding(ids.cpu().numpy(), noise=synth.synthetic_noise(64), return_latent=True)
    l = lat.float().cpu().contiguous().numpy()
    lat_ok = np.array([zlib.crc32(l[i].tobytes()) == int(g["lat_crc"][i]) for i in range(64)])
    print(f"\n64 images in one batch: final latents of {int(lat_ok.sum())} / 64 images equal the reference's; first four: {int((l[:4] != g['lat4']).sum())} differing elements")
    assert lat_ok.all(), np.nonzero(~lat_ok)[0]
    bits = rec.cpu().view(torch.int16).numpy().view(np.uint16)
    crc = np.array([zlib.crc32(np.ascontiguousarray(bits[i]).tobytes()) for i in range(64)], dtype=np.uint32)
    assert np.array_equal(crc, g["crc"]), "pixels differ from the reference's"
    assert np.array_equal(E.psnr_each(rec, imgs), g["psnr_ref"])


def test_guided_steps_16_images_equal_the_reference(models):
    """classifier-free guidance (sd3/rectified_flow.py:280-289: MMDiT.cfg_inference -- integer-floored timestep, no context key visible -- and the conditional
    call without context_see_xt, mixed as u + s (c - u)): the latents after one and two guided steps at B = 16 have the reference's crc32 (tests/golden/cfg_b16.npz)"""
    from selftoktokenizer_amd.pipeline import _Flow
    from selftoktokenizer_amd.schedule import DiTiCont
    sd, enc, dit = models
    g = np.load(os.path.join(GOLD, "cfg_b16.npz"))
    B = 16
    cfgp = default_config(512).tokenizer.params
    flow = _Flow(50, 1.0, dit.device)
    ktab = DiTiCont(1000, 512, cfgp.stages, cfgp.k_per_stage).to_indices(flow.t_long)
    ehs = enc.codes_ln(torch.from_numpy(synth.synthetic_token_ids(B, first_index=11)).cuda())
    noise = synth.synthetic_noise(B, first_index=11)
    for steps in (1, 2):
        lat = flow.p_sample_loop(dit, noise, ehs, ktab, context_see_xt=True, uncond_scale=float(g["scale"]), max_steps=steps)
        sub = lat[:, :, ::4, ::4].contiguous().cpu().numpy()
        print(f"\nafter {steps} guided step(s): sub-sample max abs diff vs the reference {np.abs(sub - g[f'sub_{steps}']).max():.3e}")
        assert zlib.crc32(lat.contiguous().cpu().numpy().tobytes()) == int(g[f"crc_{steps}"]), f"latents after {steps} guided step(s) differ from the reference's"


def test_k1024_tokenizer_equals_the_reference_at_16_images():
    """BASELINE configs[2] (K = 1024: 1280 keys in the Q-Former's query attention, 1024 context slots in the joint attention): the reference's own
    ImageTokenizer(k = 1024) at B = 16 (tests/golden/k1024_b16.npz) -- encoder features and ids from 16 latents and one MMDiT.forward, bit for bit"""
    from selftoktokenizer_amd.encoder import QformerEncoderGPU
    from selftoktokenizer_amd.mmdit import MMDiTGPU
    from selftoktokenizer_amd.pipeline import _Flow
    g = np.load(os.path.join(GOLD, "k1024_b16.npz"))
    sd = W.synthetic_state_dict(W.expected_shapes(1024), device="cuda")
    dev = torch.device("cuda", torch.cuda.current_device())
    enc = QformerEncoderGPU(sd, dev, 1024, mode="exact")
    B = 16
    x0 = synth.synthetic_latents(B, first_index=5).to(torch.bfloat16).float().cuda()
    z = enc.features(x0)
    _same(z[:2], g["z2"], "K = 1024 encoder features (first two images)")
    assert zlib.crc32(z.contiguous().cpu().numpy().tobytes()) == int(g["zcrc"])
    outs_q, ids = enc(x0)
    assert np.array_equal(ids.cpu().numpy(), g["ids"].astype(np.int64))
    dit = MMDiTGPU(sd, dev, 1024, gemm="exact")
    flow = _Flow(50, 1.0, dev)
    i, k = int(g["step"]), int(g["k"])
    x = synth.synthetic_noise(B, first_index=5, device="cuda")
    y = dit.velocity_tokens(x, flow.t_freq_exact[i:i + 1].expand(B, -1).contiguous(), dit.embed_context(outs_q), k + 1, True)
    _, v = ops.unpatchify_cfg_euler(y, None, 0.0, C=16, hp=16, wp=16)
    print(f"\nK = 1024, step {i} (k = {k}): velocity sub-sample max abs diff vs the reference {np.abs(v[:, :, ::4, ::4].cpu().numpy() - g['vsub']).max():.3e}")
    assert zlib.crc32(v.contiguous().cpu().numpy().tobytes()) == int(g["vcrc"])
