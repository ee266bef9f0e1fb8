"""-m gpu: SelftokPipeline (drop-in API, through the `mimogpt.infer` import path) end to end against the
reference's own pipeline run on the same synthetic weights (tests/golden/pipeline_b1.npz)."""
import os

import numpy as np
import pytest
import torch

from selftoktokenizer_amd import synth, weights as W
from selftoktokenizer_amd.config import default_config

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def pipe():
    from mimogpt.infer.SelftokPipeline import SelftokPipeline
    sd = W.synthetic_state_dict(W.expected_shapes(512), device="cuda")
    p = SelftokPipeline(default_config(512), ckpt_path=None, sd3_path=None, device="cuda", state_dict=sd,
                        vae_state_dict=W.synthetic_vae_state_dict(device="cuda"))
    p.verbose = False
    return p


def test_api_surface(pipe):
    assert pipe.K == 512 and pipe._steps == 50
    assert hasattr(pipe, "vae") and hasattr(pipe, "flow") and hasattr(pipe.model, "encoder") and hasattr(pipe.model, "model")
    assert pipe.flow.timestep_map.shape == (50,)
    # the read-only nn.Module surface users of the reference touch on pipe.model / pipe.vae (SelftokPipeline.py:163, 200-208)
    sd_ref = W.expected_shapes(512)
    sd = pipe.model.state_dict()
    assert set(sd) <= set(sd_ref) and all(tuple(sd[k].shape) == tuple(sd_ref[k]) for k in sd) and len(sd) > 900
    assert "quantizer._codebook.embed" in pipe.model.encoder.state_dict() and "joint_blocks.0.x_block.attn.qkv.weight" in pipe.model.model.state_dict()
    assert pipe.model.eval() is pipe.model and pipe.model.set_eval() is pipe.model and pipe.vae.eval() is pipe.vae
    assert pipe.vae.to("cuda") is pipe.vae and pipe.model.to(pipe.device) is pipe.model
    assert sum(p.numel() for p in pipe.model.model.parameters()) > 2.0e9 and all(p.is_cuda for p in pipe.vae.parameters())
    with pytest.raises(NotImplementedError):
        pipe.model.train()
    with pytest.raises(NotImplementedError):
        pipe.vae.to("cpu")
    with pytest.raises(ValueError):
        from mimogpt.infer.SelftokPipeline import SelftokPipeline
        SelftokPipeline(default_config(512), None, None, model_type="sdxl", device="cuda")


def test_encoding_vs_reference(pipe):
    """one image from pixels against the reference pipeline's ids: 512 / 512 (exact-order VAE encoder since round 4, exact-order Q-Former
    encoder since round 5: no flip is tolerated)."""
    g = np.load(os.path.join(GOLD, "pipeline_b1.npz"))
    g16 = np.load(os.path.join(GOLD, "pipeline_b16.npz"))
    assert np.array_equal(g["tokens"][0], g16["tokens"][0].astype(np.int64))        # the B = 16 reference run agrees on image 0
    tokens = pipe.encoding(synth.synthetic_images(1), device="cuda")
    assert tokens.dtype == torch.int64 and tokens.is_cuda and tuple(tokens.shape) == (1, 512)
    mism = tokens.cpu().numpy() != g["tokens"]
    gaps = g16["gap"][0][mism[0]]
    print("e2e token-id exact match vs reference pipeline (bf16 VAE upstream):", 1 - mism.mean(), "reference gaps of the flips:", gaps)
    assert mism.sum() == 0


@pytest.mark.parametrize("gemm", ["fp32", "f16x2"])
def test_decoding_vs_reference(pipe, gemm):
    """50-step decode against the reference pipeline's own run, with the MMDiT Linears on hipBLASLt fp32 and on the f16x2
    split kernel: the same gates hold for both (the split arithmetic is the more accurate of the two)."""
    assert pipe.set_gemm(gemm) == gemm
    try:
        _decoding_vs_reference(pipe)
    finally:
        pipe.set_gemm("fp32")


def _decoding_vs_reference(pipe):
    g = np.load(os.path.join(GOLD, "pipeline_b1.npz"))
    noise = synth.synthetic_noise(1)
    trace = []
    real = pipe.flow.p_sample_loop

    def traced(*a, **k):
        return real(*a, trace=trace, **k)
    pipe.flow.p_sample_loop = traced
    try:
        rec = pipe.decoding(g["tokens"], device="cuda", noise=noise)
    finally:
        pipe.flow.p_sample_loop = real
    assert rec.dtype == torch.bfloat16 and tuple(rec.shape) == (1, 3, 256, 256)
    assert float(rec.min()) >= 0.0 and float(rec.max()) <= 1.0
    for j, step in enumerate(g["lat_steps"]):          # golden lats[j] = DiT input at step `step` = latent after `step` Euler steps
        err = float((trace[int(step) - 1].cpu() - torch.from_numpy(g["lats"][j])).abs().max())
        print(f"latent after {int(step)} steps: max abs err {err:.3e}")
        assert err < 2e-5                  # measured (round 6): 2.4e-7 after one step .. 2.4e-6 after 49, both arithmetics
    ref = torch.from_numpy(g["rec_bf16"]).view(torch.bfloat16).float()
    mse = float(((rec.float().cpu() - ref) ** 2).mean())
    psnr = 10 * np.log10(1.0 / max(mse, 1e-12))
    print("pixel PSNR vs reference pipeline output:", psnr)
    assert psnr > 38.0
    # the north star's "within 1e-3 PSNR of the reference": reconstruction PSNR against the ORIGINAL image, ours vs reference's
    orig = (synth.synthetic_images(1) + 1.0) / 2.0
    p_ref = 10 * np.log10(1.0 / float(((ref - orig) ** 2).mean()))
    p_our = 10 * np.log10(1.0 / float(((rec.float().cpu() - orig) ** 2).mean()))
    print(f"reconstruction PSNR vs original: reference {p_ref:.5f} dB, ours {p_our:.5f} dB, delta {abs(p_ref - p_our):.2e} dB")
    # End to end this delta is the bf16 VAE decoder's implementation noise, not the tokenizer/DiT path.  Rounds 1-2 measured 5e-4 ..
    # 2.2e-3 dB here (MIOpen solver search + a separate bf16 bias add per convolution) under a 5e-3 gate; with the bias inside the
    # accumulation and the deterministic GEMM algorithm (vae.py) it is bit-stable from run to run and 16 images give mean 2.9e-4 /
    # max 8.8e-4 dB -- the spread between two CPU implementations of the same decoder is 3.9e-4 / 1.09e-3 (test_parity16_gpu.py)
    assert abs(p_ref - p_our) < 1e-3       # the north star's bound (measured: 4.8e-4 fp32, 2.6e-4 f16x2, parity decoder)
    # ... and the north star's 1e-3 dB criterion is checked where it is meaningful: the reference's latent and ours (both after
    # 49 of the 50 steps) through the SAME decoder in ONE batch, so that only the latent difference remains.
    both = torch.cat([torch.from_numpy(g["lats"][-1]).cuda(), trace[int(g["lat_steps"][-1]) - 1]])
    px = pipe._to_pixels(both).float().cpu()
    q_ref = 10 * np.log10(1.0 / float(((px[0:1] - orig) ** 2).mean()))
    q_our = 10 * np.log10(1.0 / float(((px[1:2] - orig) ** 2).mean()))
    print(f"same-decoder PSNR vs original: reference latent {q_ref:.6f} dB, our latent {q_our:.6f} dB, delta {abs(q_ref - q_our):.2e} dB")
    assert abs(q_ref - q_our) < 1e-3       # the north star's bound (measured 1.4e-4; the metric's floor for latents that differ by 2e-6 reaches ~1e-3 dB on single images)


def test_decode_is_deterministic_and_batch_independent(pipe):
    ids = synth.synthetic_token_ids(3)
    noise = synth.synthetic_noise(3)
    a, la = pipe.decoding(ids, noise=noise, return_latent=True, max_steps=3)
    b, lb = pipe.decoding(ids[1:2], noise=noise[1:2], return_latent=True, max_steps=3)
    torch.testing.assert_close(la[1:2], lb, rtol=1e-4, atol=1e-4)
    a2, la2 = pipe.decoding(ids, noise=noise, return_latent=True, max_steps=3)
    assert torch.equal(la, la2)                      # tokenizer/DiT path is bit-deterministic run to run
    # the bf16 VAE convolutions (MIOpen) are not: allow bf16-ulp noise on the pixels
    assert float((a.float() - a2.float()).abs().max()) < 0.05


@pytest.mark.parametrize("gemm", ["fp32", "f16x2"])
def test_cfg_vs_reference(pipe, gemm):
    """classifier-free guidance against the reference's own sampler (golden cfg_b1.npz = RectifiedFlow.sample_one_step with
    cfg_scale = 2 calling MMDiT.cfg_inference and the conditional forward without context_see_xt): latents after one and two
    guided steps, and the unconditional / conditional velocities alone at schedule entry 30 (k = 375)."""
    g = np.load(os.path.join(GOLD, "cfg_b1.npz"))
    assert pipe.set_gemm(gemm) == gemm
    try:
        noise = synth.synthetic_noise(1, first_index=11)
        trace = []
        real = pipe.flow.p_sample_loop
        pipe.flow.p_sample_loop = lambda *a, **k: real(*a, trace=trace, **k)
        try:
            pipe.decoding(g["ids"], noise=noise, max_steps=2, uncond_scale=float(g["scale"]))
        finally:
            pipe.flow.p_sample_loop = real
        for n in (1, 2):
            err = float((trace[n - 1].cpu() - torch.from_numpy(g[f"lat_after_{n}"])).abs().max())
            print(f"[{gemm}] guided latent after {n} steps: max abs err {err:.3e}")
            assert err < 1e-5
        dit, enc = pipe.model.model, pipe.model.encoder
        i, k = int(g["index"]), int(g["k"])
        assert int(pipe.k_table[i]) == k
        x = noise.cuda()
        ctx0 = dit.embed_context(enc.codes_ln(torch.from_numpy(g["ids"]).cuda()))
        yc = dit.velocity_tokens(x, pipe.flow.t_freq[i:i + 1].contiguous(), ctx0, k + 1, False)
        yu = dit.velocity_tokens(x, pipe.flow.t_freq_uncond[i:i + 1].contiguous(), ctx0, 0, False)
        from selftoktokenizer_amd import ops
        for name, y in (("v_cond", yc), ("v_uncond", yu)):
            v = ops.unpatchify_cfg_euler(y, C=16, hp=16, wp=16)[1]
            err = float((v.cpu() - torch.from_numpy(g[name])).abs().max())
            print(f"[{gemm}] {name} at schedule entry {i}: max abs err {err:.3e}")
            assert err < 5e-5
    finally:
        pipe.set_gemm("fp32")


def test_decoding_accepts_every_integer_wire_format(pipe):
    """tokens.py advertises uint16; torch.from_numpy also takes int16/int32 arrays: all must decode like int64"""
    ids = synth.synthetic_token_ids(1)
    noise = synth.synthetic_noise(1)
    _, ref = pipe.decoding(ids, noise=noise, return_latent=True, max_steps=1)
    for dt in (np.uint16, np.int32, np.uint32):
        _, lat = pipe.decoding(ids.astype(dt), noise=noise, return_latent=True, max_steps=1)
        assert torch.equal(lat, ref), dt
    _, lat = pipe.decoding(torch.from_numpy(ids).to(torch.int16).clamp(min=0), noise=noise, return_latent=True, max_steps=1)
    assert lat.shape == ref.shape
    with pytest.raises(TypeError):
        pipe.decoding(ids.astype(np.float32), noise=noise, max_steps=1)


@pytest.mark.parametrize("prefix_k", [20, 376])
def test_partial_prefix_decode_vs_oracle(pipe, prefix_k):
    """decode from the first prefix_k tokens only (the reference loop's super_mask hook with a prefix mask; README.md:241):
    equals the oracle's p_sample_loop with mask * super_mask, and does not depend on the ids beyond the prefix"""
    from oracle import model as OM, schedule as OS
    from selftoktokenizer_amd import tokens as T
    ids = synth.synthetic_token_ids(1, first_index=4)
    noise = synth.synthetic_noise(1, first_index=4)
    _, lat = pipe.decoding(ids, noise=noise, return_latent=True, max_steps=2, prefix_k=prefix_k)
    padded, k = T.pad_prefix(T.from_ar_order(T.to_ar_order(ids))[:, :prefix_k], 512)
    assert k == prefix_k
    _, lat_pad = pipe.decoding(padded, noise=noise, return_latent=True, max_steps=2, prefix_k=k)
    assert torch.equal(lat, lat_pad)                       # invisible ids are never read
    _, lat_full = pipe.decoding(ids, noise=noise, return_latent=True, max_steps=2)
    assert not torch.equal(lat, lat_full)
    dit_sd = {k_: v.cpu() for k_, v in pipe.model.model.w.items()}
    dit_sd.update({k_: v.cpu() for k_, v in pipe.model.encoder.w.items() if "final_layer_norm3" in k_})
    dit_sd["encoder.quantizer._codebook.embed"] = pipe.model.encoder.codebook.cpu()[None]
    stg, kps = OS.parse_stages("200,400,600,800,1000", "192,184,72,48,16")
    ref = OM.decode_latent(dit_sd, torch.from_numpy(ids), noise, stg, kps, 50, max_steps=2, prefix_k=prefix_k)
    err = float((lat.cpu() - ref).abs().max())
    print(f"prefix_k={prefix_k}: latent after 2 steps vs oracle max abs err {err:.3e}")
    assert err < 1e-4
    with pytest.raises(ValueError):
        pipe.decoding(ids, noise=noise, prefix_k=513)


def test_k1024_vs_reference_and_renderer_config():
    """BASELINE configs[2] (1024-token tokenizer; stage split assumed, the reference ships no 1024 config) pinned to the REFERENCE's
    own ImageTokenizer(k=1024) run (golden k1024_b1.npz: encoder features + ids, one MMDiT.forward at k = 750), for both GEMM
    arithmetics; then shapes/finiteness of the full K=1024 pipeline and of configs[3] (one-step renderer)."""
    from mimogpt.infer.SelftokPipeline import SelftokPipeline
    g = np.load(os.path.join(GOLD, "k1024_b1.npz"))
    vsd = W.synthetic_vae_state_dict(device="cuda")
    sd = W.synthetic_state_dict(W.expected_shapes(1024), device="cuda")
    p = SelftokPipeline(default_config(1024), None, None, device="cuda", state_dict=sd, vae_state_dict=vsd)
    p.verbose = False
    assert p.K == 1024 and int(p.k_table[0]) == 1023
    x0 = synth.synthetic_latents(1, first_index=5, device="cuda")
    z = p.model.encoder.features(x0).cpu()
    err = float((z - torch.from_numpy(g["z"])).abs().max())
    _, ids = p.model.encoder(x0, d=None)
    mism = ids.cpu().numpy() != g["ids"]
    print(f"K=1024 encoder z max abs err vs reference {err:.3e}; id mismatches {int(mism.sum())} / 1024, gaps {g['gap'][mism]}")
    # the golden is the reference's B = 1 run, whose MKL path differs from its B >= 8 runs (PINNING.json: encode64): the exact-order encoder reproduces the
    # B >= 8 bits (test_dit_exact_gpu.py::test_k1024_tokenizer_equals_the_reference_at_16_images: 0 bits), against B = 1 it measures 2.1e-6 and 0 flipped ids
    assert err < 2e-5 and int(mism.sum()) == 0
    x = synth.synthetic_noise(1, first_index=5, device="cuda")
    t = torch.full((1,), float(g["t"]), device="cuda")
    mask = torch.arange(1024, device="cuda")[None] <= int(g["k"])
    ehs = torch.from_numpy(g["outs_q"]).cuda()
    for gemm in ("fp32", "f16x2"):
        assert p.set_gemm(gemm) == gemm
        v, _ = p.model.model(x, t, encoder_hidden_states=ehs, mask=mask, context_see_xt=True)
        e = float((v.cpu() - torch.from_numpy(g["v"])).abs().max())
        print(f"K=1024 MMDiT.forward (k={int(g['k'])}) [{gemm}] max abs err vs reference {e:.3e}")
        assert e < 5e-5                    # measured 1.5e-5 in both arithmetics
    p.set_gemm("fp32")
    tok = p.encoding(synth.synthetic_images(2))
    assert tuple(tok.shape) == (2, 1024)
    rec, lat = p.decoding(tok.cpu().numpy(), noise=synth.synthetic_noise(2), max_steps=2, return_latent=True)
    assert tuple(rec.shape) == (2, 3, 256, 256) and bool(torch.isfinite(lat).all())
    # the FULL-SIZE leg of configs[2] (B = 64 x 1024 tokens; two of the 50 sampler steps keep the test short): same per-image results
    # as the B = 2 call up to GEMM-tiling noise, in both arithmetics
    tok64 = p.encoding(synth.synthetic_images(64, device="cuda"))
    assert tuple(tok64.shape) == (64, 1024) and torch.equal(tok64[:2], tok)      # the exact-order encoder is batch invariant: B = 2 and B = 64 give the same ids
    ids64 = tok64.cpu().numpy()
    ids64[:2] = tok.cpu().numpy()
    for gemm in ("fp32", "f16x2"):
        assert p.set_gemm(gemm) == gemm
        _, lat64 = p.decoding(ids64, noise=synth.synthetic_noise(64), max_steps=2, return_latent=True)
        assert tuple(lat64.shape) == (64, 16, 32, 32) and bool(torch.isfinite(lat64).all())
        d = float((lat64[:2] - lat).abs().max())
        print(f"configs[2] full size (B=64, K=1024) [{gemm}]: latents of images 0-1 vs the B=2 fp32 call, max abs diff {d:.3e}")
        assert d < 2e-4
    del p, sd
    torch.cuda.empty_cache()
    sd = W.synthetic_state_dict(W.expected_shapes(512, renderer=True), device="cuda")
    r = SelftokPipeline(default_config(512, renderer=True), None, None, device="cuda", state_dict=sd, vae_state_dict=vsd)
    r.verbose = False
    rec, lat = r.decoding_with_renderer(synth.synthetic_token_ids(2), return_latent=True)
    assert tuple(rec.shape) == (2, 3, 256, 256) and float(rec.min()) >= 0 and float(rec.max()) <= 1
    # the FULL-SIZE leg of configs[3]: B = 256 through the one-step renderer, both arithmetics
    for gemm in ("fp32", "f16x2"):
        assert r.set_gemm(gemm) == gemm
        rec256, lat256 = r.decoding_with_renderer(synth.synthetic_token_ids(256), return_latent=True)
        assert tuple(rec256.shape) == (256, 3, 256, 256) and float(rec256.min()) >= 0 and float(rec256.max()) <= 1
        d = float((lat256[:2] - lat).abs().max())
        print(f"configs[3] full size (B=256 renderer) [{gemm}]: latents of images 0-1 vs the B=2 fp32 call, max abs diff {d:.3e}")
        assert d < 2e-4
    # ... and its ENCODE leg at B = 256: 256 images in one call through the exact VAE + encoder; the first 16 are the images of the
    # reference pipeline's run, and an exact-order path is batch-invariant by construction: their ids must be the reference's, whatever rides along
    imgs = synth.synthetic_images(256, device="cuda")
    ids256 = r.encoding(imgs)
    assert tuple(ids256.shape) == (256, 512) and ids256.dtype == torch.int64
    g16 = np.load(os.path.join(GOLD, "pipeline_b16.npz"))
    assert np.array_equal(ids256[:16].cpu().numpy(), g16["tokens"].astype(np.int64)), "ids of the first 16 of 256 images differ from the reference's 16-image run"
    g64 = np.load(os.path.join(GOLD, "encode_b64.npz"))
    assert np.array_equal(ids256[:64].cpu().numpy(), g64["tokens"].astype(np.int64))
    rec256, _ = r.decoding_with_renderer(ids256.cpu().numpy(), return_latent=True)
    assert tuple(rec256.shape) == (256, 3, 256, 256)


def test_ema_decoder_option(pipe):
    """ema_decoder=True takes the DiT from state_dict['ema_state_dict'] (keys without the 'model.' prefix), reference :193-198"""
    from mimogpt.infer.SelftokPipeline import SelftokPipeline
    sd = {k: v for k, v in pipe.model.encoder.w.items()}
    sd["encoder.quantizer._codebook.embed"] = pipe.model.encoder.codebook[None]
    sd["ema_state_dict"] = {k[len("model."):]: v for k, v in pipe.model.model.w.items()}
    p2 = SelftokPipeline(default_config(512), None, None, device="cuda", state_dict=sd, vae_state_dict=pipe.vae.w, ema_decoder=True)
    p2.verbose = False
    ids, noise = synth.synthetic_token_ids(1), synth.synthetic_noise(1)
    _, la = pipe.decoding(ids, noise=noise, return_latent=True, max_steps=2)
    _, lb = p2.decoding(ids, noise=noise, return_latent=True, max_steps=2)
    assert torch.equal(la, lb)


@pytest.mark.parametrize("datasize", [128, 320])
def test_other_resolutions_vs_oracle(pipe, datasize):
    """enable_enc_variable_size: pos-embed cropping and ragged attention tiles at latent 16x16 / 40x40
    (reference: cropped_pos_embed models_ours.py:183-202, sd3/mmdit.py:878-896); oracle on the same weights."""
    from oracle import model as OM, schedule as OS
    lat = datasize // 8
    x0 = synth.hash_normalish(0xD5 + datasize, (1, 16, lat, lat), "cuda")
    z = pipe.model.encoder.features(x0).cpu()
    enc_sd = {k: v.cpu() for k, v in pipe.model.encoder.w.items()}
    z_ref = OM.encoder_features(enc_sd, x0.cpu())
    assert float((z - z_ref).abs().max()) < 3e-4
    # two sampler steps on a lat x lat latent vs the oracle
    dit_sd = {k: v.cpu() for k, v in pipe.model.model.w.items()}
    dit_sd.update({k: v for k, v in enc_sd.items() if "final_layer_norm3" in k})
    dit_sd["encoder.quantizer._codebook.embed"] = pipe.model.encoder.codebook.cpu()[None]
    ids = synth.synthetic_token_ids(1, first_index=3)
    noise = synth.hash_normalish(0xA0 + datasize, (1, 16, lat, lat), "cpu")
    old = pipe.datasize
    pipe.datasize = datasize
    try:
        rec, latent = pipe.decoding(ids, noise=noise, return_latent=True, max_steps=2)
    finally:
        pipe.datasize = old
    stg, kps = OS.parse_stages("200,400,600,800,1000", "192,184,72,48,16")
    ref = OM.decode_latent(dit_sd, torch.from_numpy(ids), noise, stg, kps, 50, max_steps=2)
    assert tuple(rec.shape) == (1, 3, datasize, datasize)
    assert float((latent.cpu() - ref).abs().max()) < 2e-3


def test_hipgraph_replay_equals_eager(pipe):
    """the sampler loop captured into a hipGraph (use_graph=True) replays bit-identically, also on new inputs"""
    ids, noise = synth.synthetic_token_ids(2), synth.synthetic_noise(2)
    _, eager = pipe.decoding(ids, noise=noise, return_latent=True, max_steps=3)
    _, g1 = pipe.decoding(ids, noise=noise, return_latent=True, max_steps=3, use_graph=True)     # capture + replay
    assert torch.equal(eager, g1)
    ids2, noise2 = synth.synthetic_token_ids(2, first_index=5), synth.synthetic_noise(2, first_index=5)
    _, eager2 = pipe.decoding(ids2, noise=noise2, return_latent=True, max_steps=3)
    _, g2 = pipe.decoding(ids2, noise=noise2, return_latent=True, max_steps=3, use_graph=True)   # replay only
    assert torch.equal(eager2, g2)


@pytest.mark.parametrize("gemm", ["fp32", "f16x2"])
def test_single_image_decode_remembers_the_timestep_modulations(pipe, gemm):
    """one image: t_embedder + the 26 adaLN Linears of a step depend on the scheduled timestep alone and are remembered across steps'
    calls (MMDiTGPU._step_modulations) -- same tensors the step would compute: latents bit-equal with the table off, on (filling),
    on (reading), and through a hipGraph captured after the table was filled; CFG's second branch (floor(t 1000)) has its own entries"""
    dit = pipe.model.model
    assert pipe.set_gemm(gemm) == gemm
    ids, noise = synth.synthetic_token_ids(1, first_index=3), synth.synthetic_noise(1, first_index=3)
    try:
        dit._mod_cache.clear()
        keep, dit.MOD_CACHE_MAX = dit.MOD_CACHE_MAX, 0
        _, off = pipe.decoding(ids, noise=noise, return_latent=True, max_steps=4)
        _, off_cfg = pipe.decoding(ids, noise=noise, return_latent=True, max_steps=4, uncond_scale=2.0)
        assert len(dit._mod_cache) == 0
        dit.MOD_CACHE_MAX = keep
        _, fill = pipe.decoding(ids, noise=noise, return_latent=True, max_steps=4)
        assert len(dit._mod_cache) == 4
        _, read = pipe.decoding(ids, noise=noise, return_latent=True, max_steps=4)
        _, graph = pipe.decoding(ids, noise=noise, return_latent=True, max_steps=4, use_graph=True)
        _, cfg = pipe.decoding(ids, noise=noise, return_latent=True, max_steps=4, uncond_scale=2.0)
        assert len(dit._mod_cache) == 8
        assert torch.equal(off, fill) and torch.equal(off, read) and torch.equal(off, graph) and torch.equal(off_cfg, cfg)
        n = len(dit._mod_cache)
        pipe.decoding(synth.synthetic_token_ids(2), noise=synth.synthetic_noise(2), max_steps=2)      # B > 1: computed per step, nothing stored
        assert len(dit._mod_cache) == n
        # the captured graph read the remembered tensors -- it must keep them alive.  Drop the table, churn the allocator so that
        # freed blocks would be reused and overwritten, replay: still the same latents
        dit._mod_cache.clear()
        junk = [torch.full((1 << 18,), float("nan"), device="cuda") for _ in range(64)]
        torch.cuda.synchronize()
        _, replay = pipe.decoding(ids, noise=noise, return_latent=True, max_steps=4, use_graph=True)
        del junk
        assert torch.equal(off, replay), "a graph replay read freed modulation tensors"
        # a full table evicts its OLDEST entry only
        dit.MOD_CACHE_MAX = 2
        pipe.decoding(ids, noise=noise, return_latent=True, max_steps=4)
        assert len(dit._mod_cache) == 2
    finally:
        dit.MOD_CACHE_MAX = keep
        pipe.set_gemm("fp32")


def test_full_batch_size_independence(pipe):
    """BASELINE configs[1] batch (B=64): every image's tokens / latent must not depend on its batch-mates.
    The exact / parity VAE (csrc/vae_exact.hip, csrc/conv.hip + the fp64-statistics GroupNorm: no library, no solver choice, fixed summation order per output
    element) is batch independent BY CONSTRUCTION: its latents at B = 64 and in 8 chunks of 8 must be bit-identical.  Behind it the
    round 5: the Q-Former encoder is row-independent too (csrc/encoder_exact.hip: own fixed-order GEMM instead of hipBLASLt, which tiles
    M = 64*768 and M = 8*768 differently): ids are IDENTICAL for every batching.  The 'fast' encoder (library GEMMs) is measured beside it."""
    B = 64
    imgs = synth.synthetic_images(B, device="cuda")
    assert pipe.vae.mode in ("exact", "parity")            # both are batch independent by construction
    x0 = pipe.encode_latents(imgs)
    x0_chunks = torch.cat([pipe.encode_latents(imgs[i:i + 8]) for i in range(0, B, 8)])
    assert torch.equal(x0, x0_chunks), "the parity VAE's latents depend on the batch size"
    assert torch.equal(x0[5:6], pipe.encode_latents(imgs[5:6]))
    tok_all = pipe.encoding(imgs)
    tok_chunks = torch.cat([pipe.encoding(imgs[i:i + 8]) for i in range(0, B, 8)])
    assert pipe.model.encoder.mode == "exact" and torch.equal(tok_all, tok_chunks), "ids depend on the batch size"
    assert torch.equal(tok_all[5:6], pipe.encoding(imgs[5:6]))
    # the 'fast' encoder (hipBLASLt GEMMs, rounds 1-3 kernels) on the same latents: fp32 GEMM tiling noise, measured and kept in the log
    from selftoktokenizer_amd.encoder import QformerEncoderGPU
    fast = QformerEncoderGPU(pipe.model.encoder.state_dict(prefix="encoder."), pipe.device, pipe.K, mode="fast")
    t_all = fast(x0, d=None)[1]
    t_chunks = torch.cat([fast(x0[i:i + 8], d=None)[1] for i in range(0, B, 8)])
    match32 = float((t_all == t_chunks).float().mean())
    print(f"'fast' encoder: token match B=64 vs 8x8 from identical latents {match32:.6f}; vs the exact encoder at B=64 {float((t_all == tok_all).float().mean()):.6f}")
    assert match32 >= 0.999 and float((t_all == tok_all).float().mean()) >= 0.999
    noise = synth.synthetic_noise(B)
    ids = tok_all.cpu().numpy()
    _, lat_all = pipe.decoding(ids, noise=noise, return_latent=True, max_steps=1)
    _, lat_8 = pipe.decoding(ids[8:16], noise=noise[8:16], return_latent=True, max_steps=1)
    assert float((lat_all[8:16] - lat_8).abs().max()) < 1e-4
    # the id matrix that N ranks would all-gather is just the concatenation of the shards
    from selftoktokenizer_amd.dist import shard_range
    lo, hi = shard_range(B, 3, 8)
    assert torch.equal(tok_all[lo:hi], pipe.model.encoder(x0[lo:hi], d=None)[1])


@pytest.mark.parametrize("gemm", ["fp32", "f16x2"])
def test_sampler_options_vs_reference(pipe, gemm):
    """two dormant branches of the reference's sampler, against its own RectifiedFlow.sample_one_step on the real MMDiT
    (tests/golden/sampler_options_b1.npz): `parameterization: x0` (euler_step, sd3/rectified_flow.py:305-307) and a NON-prefix
    `super_mask` (p_sample_loop's mask * super_mask, :226-227) with a hash-random visibility pattern."""
    g = np.load(os.path.join(GOLD, "sampler_options_b1.npz"))
    noise = synth.synthetic_noise(1, first_index=21)
    assert pipe.set_gemm(gemm) == gemm
    real = pipe.flow.p_sample_loop
    try:
        for name, kw, param in (("x0", {}, "x0"), ("supermask", {"super_mask": g["super_mask"]}, "velocity")):
            trace = []
            pipe.flow.p_sample_loop = lambda *a, **k: real(*a, trace=trace, **k)
            pipe.flow.parameterization = param
            pipe.decoding(g["ids"], noise=noise, max_steps=2, **kw)
            for n in (1, 2):
                err = float((trace[n - 1].cpu() - torch.from_numpy(g[f"{name}_after_{n}"])).abs().max())
                print(f"[{gemm}] {name}: latent after {n} steps vs the reference, max abs err {err:.3e}")
                assert err < 2e-5
    finally:
        pipe.flow.p_sample_loop = real
        pipe.flow.parameterization = "velocity"
        pipe.set_gemm("fp32")
    # the visibility pattern is honoured exactly: ids at masked positions are never read ...
    ids2 = g["ids"].copy()
    ids2[:, ~g["super_mask"]] = (ids2[:, ~g["super_mask"]] + 12345) % 32768
    _, la = pipe.decoding(g["ids"], noise=noise, max_steps=2, super_mask=g["super_mask"], return_latent=True)
    _, lb = pipe.decoding(ids2, noise=noise, max_steps=2, super_mask=g["super_mask"], return_latent=True)
    assert torch.equal(la, lb)
    # ... a prefix pattern equals prefix_k, an all-true pattern equals no mask, and per-sample patterns are refused
    pre = np.arange(512) < 100
    _, lc = pipe.decoding(g["ids"], noise=noise, max_steps=2, super_mask=pre, return_latent=True)
    _, ld = pipe.decoding(g["ids"], noise=noise, max_steps=2, prefix_k=100, return_latent=True)
    torch.testing.assert_close(lc, ld, rtol=0, atol=2e-5)          # (block 0's context QKV runs as a GEMM over 100 instead of 512 rows per sample)
    _, le = pipe.decoding(g["ids"], noise=noise, max_steps=2, super_mask=np.ones(512, bool), return_latent=True)
    _, lf = pipe.decoding(g["ids"], noise=noise, max_steps=2, return_latent=True)
    torch.testing.assert_close(le, lf, rtol=0, atol=2e-5)
    # a pattern PER SAMPLE: decoded in groups of equal pattern = the samples decoded one by one
    ids3, noise3 = np.repeat(g["ids"], 3, 0), synth.synthetic_noise(3)
    masks3 = np.stack([pre, g["super_mask"], pre])
    _, l3 = pipe.decoding(ids3, noise=noise3, max_steps=2, super_mask=masks3, return_latent=True)
    for b in range(3):
        _, l1 = pipe.decoding(ids3[b:b + 1], noise=noise3[b:b + 1], max_steps=2, super_mask=masks3[b], return_latent=True)
        torch.testing.assert_close(l3[b:b + 1], l1, rtol=0, atol=2e-5)
    with pytest.raises(NotImplementedError):       # the sampler itself still takes one pattern per call
        pipe.flow.p_sample_loop(pipe.model.model, noise3[:2], pipe._codes(ids3[:2]), pipe.k_table, max_steps=1, super_mask=np.stack([pre, ~pre]))
    with pytest.raises(ValueError):
        pipe.decoding(g["ids"], noise=noise, max_steps=1, super_mask=np.ones(100, bool))
    # a non-prefix pattern inside a hipGraph capture (the mask is resolved on the host BEFORE the capture)
    _, lg = pipe.decoding(g["ids"], noise=noise, max_steps=2, super_mask=g["super_mask"], return_latent=True, use_graph=True)
    assert torch.equal(lg, la)
    _, lg2 = pipe.decoding(ids2, noise=noise, max_steps=2, super_mask=g["super_mask"], return_latent=True, use_graph=True)    # replay
    assert torch.equal(lg2, la)
