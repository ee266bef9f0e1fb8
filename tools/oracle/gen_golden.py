"""Generate tests/golden/*.npz by running the REFERENCE (read-only import, build container only) on the
hash-generated synthetic weights/inputs, and report how oracle/ compares on the same inputs.

    python tools/oracle/gen_golden.py [stage ...]      stages: keys vq schedule encoder dit vae pipeline pipeline16 encode64 renderer cfg k1024 vqtrain rmsnorm_rotary sampler_options

A golden file holds only data: inputs that cannot be regenerated from a seed, and the reference's
outputs.  Weights/images/noise are regenerated from selftoktokenizer_amd.synth by name/seed.
The generator's findings are appended to tests/golden/PINNING.json (max abs diffs oracle vs reference).
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_harness as H  # noqa: E402
from selftoktokenizer_amd import synth, weights as W  # noqa: E402
from oracle import clib, model as OM, schedule as OS  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CFG_256 = "/root/reference/configs/res256/256-eval.yml"
CFG_RND = "/root/reference/configs/renderer/renderer-eval.yml"
REPORT = {}


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def report(name, **kv):
    REPORT[name] = {k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in kv.items()}
    print(f"[pin] {name}: {REPORT[name]}", flush=True)


def maxdiff(a, b):
    a = torch.as_tensor(a).float()
    b = torch.as_tensor(b).float()
    return float((a - b).abs().max())


_tok = {}


def tokenizer(cfg_path=CFG_256):
    if cfg_path not in _tok:
        _tok.clear()
        t0 = time.time()
        cfg = H.load_cfg(cfg_path)
        model, ref_sd = H.build_tokenizer(cfg)
        sd = {k: v for k, v in model.state_dict().items()}
        print(f"[ref] built {cfg_path} in {time.time() - t0:.1f}s", flush=True)
        _tok[cfg_path] = (cfg, model, sd)
    return _tok[cfg_path]


# ---------------------------------------------------------------------------------------------
def stage_keys():
    out = {}
    for name, path, rnd in (("k512", CFG_256, False), ("renderer", CFG_RND, True)):
        cfg, model, sd = tokenizer(path)
        ref = {k: list(v.shape) for k, v in sd.items()}
        mine = W.expected_shapes(512, renderer=rnd)
        ref_nd = {k: v for k, v in ref.items() if not k.startswith("diffusion.")}
        assert set(ref_nd) == set(mine), (set(ref_nd) ^ set(mine))
        assert all(tuple(ref_nd[k]) == tuple(mine[k]) for k in mine)
        out[name] = ref
    with open(os.path.join(GOLD, "state_dict_keys.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    report("state_dict_keys", k512=len(out["k512"]), renderer=len(out["renderer"]))


def stage_vq():
    """reference VectorQuantize eval forward on small inputs incl. edge rows"""
    cfg, model, sd = tokenizer(CFG_256)
    vq = model.encoder.quantizer
    cb = sd["encoder.quantizer._codebook.embed"][0]
    z = synth.synthetic_vq_rows(1024, seed=0x901D).reshape(2, 512, 16).clone()
    z[0, 0] = cb[77] * 3.0
    z[0, 1] = 0.0
    z[0, 2, 5] = float("nan")
    z[0, 3, 0] = float("inf")
    z[0, 4] = -cb[0]
    z[0, 5] = cb[31000] * 0.5 + cb[12] * 1e-4
    # feed z *after* project_in: call the codebook path the way VectorQuantize.forward does (:854, :876)
    from mimogpt.models.selftok.vector_quantize_pytorch import l2norm
    with torch.no_grad():
        xn = l2norm(z)
        quantize, ids, dist, _ = vq._codebook(xn)
        best = dist.reshape(2, 512, -1).max(-1).values
    ids_o, best_o = clib.vq_encode(z.reshape(-1, 16).numpy(), cb.numpy())
    ok_ids = bool(np.array_equal(ids_o.reshape(2, 512), ids.numpy()))
    nan = np.isnan(best.numpy().reshape(-1))
    ok_best = bool(np.array_equal(bits(best_o)[~nan], bits(best.numpy().reshape(-1))[~nan]))
    report("vq", ids_equal=ok_ids, best_bits_equal=ok_best, n=1024)
    assert ok_ids and ok_best
    # full VectorQuantize.forward (project_in + logging path) must return the same ids from features
    feats = synth.hash_uniform(0xFEA7, (2, 512, 512), -1.0, 1.0)
    with torch.no_grad():
        q2, ids2, _, _ = vq(feats)
        z2 = vq.project_in(feats)
    ids2_o, _ = clib.vq_encode(z2.reshape(-1, 16).numpy(), cb.numpy())
    assert np.array_equal(ids2_o.reshape(2, 512), ids2.numpy())
    np.savez_compressed(os.path.join(GOLD, "vq_small.npz"), z=z.numpy(), ids=ids.numpy(), best_bits=bits(best.numpy()),
                        z_proj=z2.numpy(), ids_proj=ids2.numpy())


def stage_schedule():
    H.install()                      # stand-alone run: make sure `mimogpt` resolves to the reference, not to the repo's shim
    from mimogpt.models.selftok.sd3.rectified_flow import RectifiedFlow
    from mimogpt.models.selftok.diti_utils import DiTi_cont
    out = {}
    for steps in (50, 100):
        flow = RectifiedFlow(steps, 1.0, None, val_schedule="uniform", shift=1.0, schedule="log_norm",
                             parameterization="velocity", m=0.0, s=1.0, force_recon=False, is_eval=True)
        mine = OS.make_schedule(steps)
        for k in ("scheduled_t", "scheduled_t_prev", "timestep_map"):
            assert np.array_equal(bits(getattr(flow, k).numpy()), bits(mine[k])), (steps, k)
        t_long = torch.stack([torch.tensor([flow.timestep_map[i]]).long()[0] for i in range(steps)]).numpy()
        assert np.array_equal(t_long, mine["t_long"])
        out[f"scheduled_t_{steps}"] = bits(flow.scheduled_t.numpy())
        out[f"scheduled_t_prev_{steps}"] = bits(flow.scheduled_t_prev.numpy())
        out[f"timestep_map_{steps}"] = bits(flow.timestep_map.numpy())
        out[f"t_long_{steps}"] = t_long
    configs = {"k512": ("200,400,600,800,1000", "192,184,72,48,16", 512), "renderer": ("1000", "512", 512),
               "k1024_assumed": ("200,400,600,800,1000", "384,368,144,96,32", 1024)}
    all_t = torch.arange(0, 1001).long()
    for name, (st, kp, K) in configs.items():
        diti = DiTi_cont(1000, K, st, kp)
        ref_all = diti.to_indices(all_t).numpy()
        stg, kps = OS.parse_stages(st, kp)
        mine_all = OS.diti_indices(all_t.numpy(), stg, kps, K)
        assert np.array_equal(ref_all, mine_all), name
        out[f"diti_{name}"] = ref_all
        k50 = diti.to_indices(torch.from_numpy(out["t_long_50"])).numpy()
        out[f"k50_{name}"] = k50
        assert np.array_equal(k50, OS.k_table(50, stg, kps, K))
    np.savez_compressed(os.path.join(GOLD, "schedule.npz"), **out)
    report("schedule", ok=True, k50_k512=out["k50_k512"].tolist())


def stage_encoder():
    cfg, model, sd = tokenizer(CFG_256)
    x0 = synth.synthetic_latents(2)
    cap = {}
    hk = model.encoder.quantizer.project_in.register_forward_hook(lambda m, i, o: cap.__setitem__("z", o.detach().clone()))
    with torch.no_grad():
        outs_q, ids = model.encoder(x0, d=None)
    hk.remove()
    z_o = OM.encoder_features(sd, x0)
    ids_o = OM.vq_ids(sd, z_o)
    match = float((ids_o == ids).float().mean())
    report("encoder", z_maxdiff=maxdiff(z_o, cap["z"]), ids_match=match, z_absmax=float(cap["z"].abs().max()))
    # gap (top1-top2) of the reference for every token: tells how fragile each id is
    cb = sd["encoder.quantizer._codebook.embed"][0]
    xn = torch.nn.functional.normalize(cap["z"].reshape(-1, 16), dim=-1)
    top2 = (xn @ cb.T).topk(2, dim=-1).values
    gap = (top2[:, 0] - top2[:, 1]).reshape(2, 512)
    np.savez_compressed(os.path.join(GOLD, "encoder_b2.npz"), z=cap["z"].numpy(), ids=ids.numpy(),
                        outs_q=outs_q.numpy(), gap=gap.numpy())


def stage_encoder_prenorm(B=8):
    """encoder_config.pre_norm = True (models_ours.py:219-220: `outs = self.final_layer_norm(outs)` before the quantizer; no shipped config sets it --
    VERDICT r5 "missing" item 4): the reference Encoder built with the flag on, on the first B latents of its own 64-image run (encode_b64.npz)."""
    cfg = H.load_cfg(CFG_256)
    cfg.tokenizer.params.encoder_config.pre_norm = True
    model, _ = H.build_tokenizer(cfg)
    assert model.encoder.pre_norm is True
    sd = {k: v for k, v in model.state_dict().items()}
    g = np.load(os.path.join(GOLD, "encode_b64.npz"))
    x0 = torch.from_numpy(g["x0_bf16"][:B]).view(torch.bfloat16).float()
    cap = {}
    hk = model.encoder.quantizer.project_in.register_forward_hook(lambda m, i, o: cap.__setitem__("z", o.detach().clone()))
    with torch.no_grad():
        outs_q, ids = model.encoder(x0, d=None)
    hk.remove()
    z = cap["z"]
    z_o = OM.encoder_features(sd, x0, pre_norm=True)
    sys.path.insert(0, "/root/repo")
    from oracle import encoder_exact as EX
    from selftoktokenizer_amd.encoder import encoder_pos_embedding
    z_x = EX.encoder_features({k: v for k, v in sd.items() if k.startswith("encoder.")}, x0.numpy(), encoder_pos_embedding(512).numpy(), pre_norm=True)
    off = float((torch.from_numpy(g["z"][:B]) - z).abs().max())
    report("encoder_prenorm", images=B, z_maxdiff_torch_oracle=maxdiff(z_o, z), z_bits_differing_exact_oracle=int((z_x.view(np.uint32) != z.numpy().view(np.uint32)).sum()),
           ids_match_exact_oracle=float((OM.vq_ids(sd, torch.from_numpy(z_x)) == ids).float().mean()), z_absmax=float(z.abs().max()),
           z_maxdiff_to_pre_norm_off=off, ids_differing_from_pre_norm_off=int((ids.numpy().astype(np.int16) != g["tokens"][:B]).sum()))
    np.savez_compressed(os.path.join(GOLD, "encoder_prenorm_b8.npz"), z=z.numpy(), ids=ids.numpy().astype(np.int16))


def stage_dit():
    cfg, model, sd = tokenizer(CFG_256)
    ids = torch.from_numpy(synth.synthetic_token_ids(1))
    with torch.no_grad():
        codes = model.encoder.quantizer.get_output_from_indices(ids)
        ehs = model.encoder.final_layer_norm3(codes.reshape(1, -1, 16))
    assert maxdiff(OM.codes_from_ids(sd, ids), ehs) < 1e-6
    x = synth.synthetic_noise(1)
    out = {"ehs": ehs.numpy()}
    for name, tval, k in (("a", 0.62, 375), ("b", 0.02, 19), ("c", 1.0, 511)):
        t = torch.full((1,), tval)
        mask = torch.arange(512)[None] <= k
        with torch.no_grad():
            v, _ = model.model(x, t, encoder_hidden_states=ehs, mask=mask, context_see_xt=True)
        v_o = OM.dit_forward(sd, x, t, ehs, mask, True)
        report(f"dit_{name}", v_maxdiff=maxdiff(v, v_o), v_absmax=float(v.abs().max()))
        out[f"v_{name}"] = v.numpy()
        out[f"t_{name}"] = np.float32(tval)
        out[f"k_{name}"] = np.int64(k)
    np.savez_compressed(os.path.join(GOLD, "dit_forward_b1.npz"), **out)


class _MirrorVAE:
    """diffusers' AutoencoderKL surface (`encode(x)[0].mode()`, `decode(z)[0]`) over the in-repo SDVAE mirror."""

    def __init__(self, vae):
        self.vae = vae

    @classmethod
    def from_pretrained(cls, path, subfolder=None):
        from mimogpt.models.selftok.sd3.sd3_impls import SDVAE
        with H.fast_init():
            vae = SDVAE(dtype=torch.bfloat16, device="cpu")
        vsd = W.synthetic_vae_state_dict()
        ldm = {}
        ref_sd = vae.state_dict()
        for k, v in vsd.items():
            k2 = W.diffusers_to_ldm_key(k)
            ldm[k2] = v.reshape(ref_sd[k2].shape)
        vae.load_state_dict(ldm, strict=True)
        return cls(vae)

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def encode(self, x, return_dict=False):
        class _D:
            def __init__(s, h):
                s.h = h

            def mode(s):
                return s.h.chunk(2, dim=1)[0]
        with torch.no_grad():
            return (_D(self.vae.encoder(x)),)

    def decode(self, z, return_dict=False):
        with torch.no_grad():
            return (self.vae.decoder(z),)


def stage_vae():
    H.install()
    vae = _MirrorVAE.from_pretrained(None)
    vsd = W.synthetic_vae_state_dict()
    img = synth.synthetic_images(1).to(torch.bfloat16)
    mean = vae.encode(img)[0].mode()
    mean_o = OM.vae_encode_mean(vsd, img)
    z = synth.synthetic_latents(1).to(torch.bfloat16)
    rec = vae.decode(z)[0]
    rec_o = OM.vae_decode(vsd, z)
    report("vae", enc_maxdiff=maxdiff(mean, mean_o), dec_maxdiff=maxdiff(rec, rec_o),
           enc_absmax=float(mean.float().abs().max()), dec_absmax=float(rec.float().abs().max()))
    np.savez_compressed(os.path.join(GOLD, "vae_b1.npz"), mean=mean.float().numpy(), rec=rec.float().numpy())


def stage_pipeline():
    """the real SelftokPipeline class end to end, B=1, 50 steps (takes a few minutes on 8 cores)"""
    H.install()
    import mimogpt.infer.SelftokPipeline as SP
    cfg = H.load_cfg(CFG_256)
    SP.AutoencoderKL = _MirrorVAE
    shapes = W.expected_shapes(512)
    real_load = torch.load
    torch.load = lambda *a, **k: W.synthetic_state_dict(shapes)
    try:
        with H.fast_init():
            pipe = SP.SelftokPipeline(cfg=cfg, ckpt_path="synthetic", sd3_path="synthetic", datasize=256, device="cpu")
    finally:
        torch.load = real_load
    sd = dict(pipe.model.state_dict())
    vsd = W.synthetic_vae_state_dict()
    images = synth.synthetic_images(1)
    t0 = time.time()
    tokens = pipe.encoding(images, device="cpu")
    print(f"[ref] encoding {time.time() - t0:.1f}s", flush=True)
    tok_o = OM.pipeline_encode(sd, vsd, images)
    report("pipeline_encode", ids_match=float((tok_o == tokens).float().mean()))
    # decode with hash noise instead of torch.randn, capturing the DiT input latent of a few steps
    noise = synth.synthetic_noise(1)
    real_randn = torch.randn
    torch.randn = lambda *a, **k: noise.clone()
    xs = []
    hk = pipe.model.model.register_forward_pre_hook(lambda m, args: xs.append(args[0].detach().clone()))
    t0 = time.time()
    try:
        rec = pipe.decoding(tokens.numpy(), device="cpu")
    finally:
        torch.randn = real_randn
        hk.remove()
    print(f"[ref] decoding {time.time() - t0:.1f}s", flush=True)
    assert len(xs) == 50
    stg, kps = OS.parse_stages(cfg.tokenizer.params.stages, cfg.tokenizer.params.k_per_stage)
    trace = []
    t0 = time.time()
    rec_o, lat_o = OM.pipeline_decode(sd, vsd, tokens.numpy(), noise, stg, kps, 50)
    print(f"[oracle] decoding {time.time() - t0:.1f}s", flush=True)
    mse = float(((rec.float() - rec_o.float()) ** 2).mean())
    report("pipeline_decode", pix_maxdiff=maxdiff(rec, rec_o), psnr_oracle_vs_ref=(10 * np.log10(1.0 / mse) if mse > 0 else 999.0))
    keep = [1, 2, 5, 10, 25, 49]
    np.savez_compressed(os.path.join(GOLD, "pipeline_b1.npz"), tokens=tokens.numpy(),
                        rec_bf16=rec.view(torch.int16).numpy(), lat_steps=np.array(keep),
                        lats=np.stack([xs[i].numpy() for i in keep]))


def stage_pipeline16(B=16):
    """The reference SelftokPipeline end to end on B = 16 synthetic images (one batch; 50 steps: ~45 min on 8 cores), to
    CHARACTERISE the end-to-end parity through the bf16 VAE over more than one image:
      reference side : VAE latents x0 (bf16-exact), pre-quantizer features z, token ids, top-1/top-2 gap and runner-up id of
                       every token, the final latent of the 50-step loop, reconstruction PSNR vs the original per image;
      second CPU implementation ("*_oracle" keys): the same images through oracle/ on this CPU with the VAE attention projections in
                       diffusers' own formulation (F.linear; OM.VAE_ATTN_PROJ = "linear") -- ids, and its VAE decode of the reference's
                       final latents -> PSNR = the spread between two CPU implementations of the same bf16 VAE (ldm-style mirror with
                       1x1-conv attention vs the diffusers layout) that the GPU's deviation is judged against.  The oracle's DEFAULT
                       formulation (1x1 convolutions, like the mirror) is bit-identical to the reference: asserted below."""
    H.install()
    import mimogpt.infer.SelftokPipeline as SP
    cfg = H.load_cfg(CFG_256)
    SP.AutoencoderKL = _MirrorVAE
    shapes = W.expected_shapes(512)
    real_load = torch.load
    torch.load = lambda *a, **k: W.synthetic_state_dict(shapes)
    try:
        with H.fast_init():
            pipe = SP.SelftokPipeline(cfg=cfg, ckpt_path="synthetic", sd3_path="synthetic", datasize=256, device="cpu")
    finally:
        torch.load = real_load
    sd = dict(pipe.model.state_dict())
    vsd = W.synthetic_vae_state_dict()
    images = synth.synthetic_images(B)
    cap = {}
    hk1 = pipe.model.encoder.register_forward_pre_hook(lambda m, args: cap.setdefault("x0", args[0].detach().clone()))
    hk2 = pipe.model.encoder.quantizer.project_in.register_forward_hook(lambda m, i, o: cap.setdefault("z", o.detach().clone()))
    t0 = time.time()
    tokens = pipe.encoding(images, device="cpu")
    hk1.remove(); hk2.remove()
    print(f"[ref] encoding B={B} {time.time() - t0:.1f}s", flush=True)
    x0, z = cap["x0"], cap["z"]
    assert torch.equal(x0, x0.to(torch.bfloat16).float())            # process_in runs in bf16: the latents are bf16-exact
    cb = sd["encoder.quantizer._codebook.embed"][0]
    xn = torch.nn.functional.normalize(z.reshape(-1, 16), dim=-1)
    top2 = (xn @ cb.T).topk(2, dim=-1)
    gap = (top2.values[:, 0] - top2.values[:, 1]).reshape(B, 512)
    id2 = top2.indices[:, 1].reshape(B, 512)
    assert bool((top2.indices[:, 0].reshape(B, 512) == tokens).float().mean() > 0.999)
    t0 = time.time()
    assert torch.equal(OM.process_in(OM.vae_encode_mean(vsd, images.to(torch.bfloat16))).to(torch.float32), x0), "default oracle VAE != mirror"
    assert torch.equal(OM.pipeline_encode(sd, vsd, images), tokens), "default oracle ids != reference"
    OM.VAE_ATTN_PROJ = "linear"
    try:
        tok_o = OM.pipeline_encode(sd, vsd, images)
        x0_o = OM.process_in(OM.vae_encode_mean(vsd, images.to(torch.bfloat16))).to(torch.float32)
    finally:
        OM.VAE_ATTN_PROJ = "conv"
    z_o = OM.encoder_features(sd, x0_o)
    print(f"[oracle] encoding B={B} {time.time() - t0:.1f}s", flush=True)
    mism_o = (tok_o != tokens)
    report("pipeline16_encode", images=B, oracle_default_equals_reference=True,
           linear_variant_ids_match=float((~mism_o).float().mean()), linear_variant_flips=int(mism_o.sum()),
           linear_variant_flip_gaps=[round(float(v), 8) for v in gap[mism_o]], x0_maxdiff_linear_variant_vs_ref=maxdiff(x0_o, x0),
           x0_rms_linear_variant_vs_ref=float((x0_o - x0).pow(2).mean().sqrt()), z_maxdiff_linear_variant_vs_ref=maxdiff(z_o, z))
    # ---- 50-step decode of the reference (hash noise instead of torch.randn) ----
    noise = synth.synthetic_noise(B)
    real_randn = torch.randn
    torch.randn = lambda *a, **k: noise.clone()
    xs = []
    hk = pipe.model.model.register_forward_pre_hook(lambda m, args: xs.append(args[0].detach().clone()))
    real_loop = pipe.flow.p_sample_loop

    def loop(*a, **k):
        cap["lat"] = real_loop(*a, **k)
        return cap["lat"]
    pipe.flow.p_sample_loop = loop
    t0 = time.time()
    try:
        rec = pipe.decoding(tokens.numpy(), device="cpu")
    finally:
        torch.randn = real_randn
        hk.remove()
        pipe.flow.p_sample_loop = real_loop
    print(f"[ref] decoding B={B} {time.time() - t0:.1f}s", flush=True)
    lat = cap["lat"].detach().float()
    orig = (images + 1.0) / 2.0

    def psnr_each(r):
        mse = ((r.float() - orig) ** 2).reshape(B, -1).double().mean(dim=1)
        return (10.0 * torch.log10(1.0 / mse)).numpy()
    p_ref = psnr_each(rec)
    # the oracle's sampler reproduces the reference's latents bit for bit (checked on the first two steps of this very batch), so
    # the oracle's pixels = its VAE decode of the reference's final latents
    stg, kps = OS.parse_stages(cfg.tokenizer.params.stages, cfg.tokenizer.params.k_per_stage)
    trace = []
    OM.decode_latent(sd, tokens, noise, stg, kps, 50, trace=trace, max_steps=2)
    d2 = max(maxdiff(trace[0], xs[1]), maxdiff(trace[1], xs[2]))
    assert torch.equal(OM.norm_ip(OM.vae_decode(vsd, OM.process_out(lat).to(torch.bfloat16))), rec), "default oracle decoder != mirror"
    OM.VAE_ATTN_PROJ = "linear"
    try:
        rec_o = OM.norm_ip(OM.vae_decode(vsd, OM.process_out(lat).to(torch.bfloat16)))
    finally:
        OM.VAE_ATTN_PROJ = "conv"
    p_or = psnr_each(rec_o)
    d = np.abs(p_or - p_ref)
    report("pipeline16_decode", images=B, oracle_latents_first2steps_maxdiff=d2, oracle_default_decoder_equals_reference=True,
           psnr_ref_mean=float(p_ref.mean()), psnr_delta_linear_variant_vs_ref_mean=float(d.mean()),
           psnr_delta_linear_variant_vs_ref_max=float(d.max()), psnr_delta_linear_variant_vs_ref_each=[round(float(v), 6) for v in d])
    np.savez_compressed(os.path.join(GOLD, "pipeline_b16.npz"), tokens=tokens.numpy().astype(np.int16), id2=id2.numpy().astype(np.int16),
                        gap=gap.numpy(), x0_bf16=x0.to(torch.bfloat16).view(torch.int16).numpy(), z=z.numpy(), lat=lat.numpy(),
                        psnr_ref=p_ref, psnr_oracle=p_or, tokens_oracle=tok_o.numpy().astype(np.int16),
                        x0_oracle_bf16=x0_o.to(torch.bfloat16).view(torch.int16).numpy())


def stage_encode64(B=64):
    """The reference `SelftokPipeline.encoding` on B = 64 synthetic images IN ONE BATCH -- BASELINE configs[1]'s encode leg
    (SelftokPipeline.py:210-225).  Stores the VAE latents (bf16-exact), the pre-quantizer features, the ids and
    the top-1/top-2 gap + runner-up of every token.  Also answers, on the REFERENCE side, whether its ids depend on how the 64
    images are batched (1x64 vs 4x16 vs 8x8 vs 64x1): `ref_split_flips` (MKL picks its kernel by M)."""
    H.install()
    import mimogpt.infer.SelftokPipeline as SP
    cfg = H.load_cfg(CFG_256)
    SP.AutoencoderKL = _MirrorVAE
    shapes = W.expected_shapes(512)
    real_load = torch.load
    torch.load = lambda *a, **k: W.synthetic_state_dict(shapes)
    try:
        with H.fast_init():
            pipe = SP.SelftokPipeline(cfg=cfg, ckpt_path="synthetic", sd3_path="synthetic", datasize=256, device="cpu")
    finally:
        torch.load = real_load
    sd = dict(pipe.model.state_dict())
    images = synth.synthetic_images(B)

    def run(imgs):
        cap = {}
        hk1 = pipe.model.encoder.register_forward_pre_hook(lambda m, args: cap.setdefault("x0", args[0].detach().clone()))
        hk2 = pipe.model.encoder.quantizer.project_in.register_forward_hook(lambda m, i, o: cap.setdefault("z", o.detach().clone()))
        try:
            tok = pipe.encoding(imgs, device="cpu")
        finally:
            hk1.remove(); hk2.remove()
        return tok, cap["x0"], cap["z"]
    t0 = time.time()
    tokens, x0, z = run(images)
    print(f"[ref] encoding B={B} {time.time() - t0:.1f}s", flush=True)
    assert torch.equal(x0, x0.to(torch.bfloat16).float())
    cb = sd["encoder.quantizer._codebook.embed"][0]
    xn = torch.nn.functional.normalize(z.reshape(-1, 16), dim=-1)
    top2 = (xn @ cb.T).topk(2, dim=-1)
    gap = (top2.values[:, 0] - top2.values[:, 1]).reshape(B, 512)
    id2 = top2.indices[:, 1].reshape(B, 512)
    assert bool((top2.indices[:, 0].reshape(B, 512) == tokens).float().mean() > 0.999)
    # the reference against itself under other batchings of the same 64 images
    split = {}
    for g in (16, 8, 1):
        toks, zs, xs_ = [], [], []
        for i in range(0, B, g):
            t, x_, z_ = run(images[i:i + g])
            toks.append(t); zs.append(z_); xs_.append(x_)
        t_g, z_g, x_g = torch.cat(toks), torch.cat(zs), torch.cat(xs_)
        split[g] = dict(flips=int((t_g != tokens).sum()), x0_equal=bool(torch.equal(x_g, x0)), z_bits_equal=bool(torch.equal(z_g, z)),
                        z_maxdiff=maxdiff(z_g, z))
        print(f"[ref] {B // g} x {g}: {split[g]}", flush=True)
    # pipeline_b16's ids are the reference's B=16 run of the first 16 images
    b16 = np.load(os.path.join(GOLD, "pipeline_b16.npz"))
    report("encode64", images=B, ref_split_flips={str(k): v["flips"] for k, v in split.items()},
           ref_split_x0_equal={str(k): v["x0_equal"] for k, v in split.items()},
           ref_split_z_bits_equal={str(k): v["z_bits_equal"] for k, v in split.items()},
           ref_split_z_maxdiff={str(k): v["z_maxdiff"] for k, v in split.items()},
           first16_equal_pipeline_b16=bool(np.array_equal(tokens[:16].numpy().astype(np.int16), b16["tokens"])),
           min_gap=float(gap.min()), tokens_gap_below_1e_5=int((gap < 1e-5).sum()), tokens_gap_below_1e_4=int((gap < 1e-4).sum()))
    np.savez_compressed(os.path.join(GOLD, "encode_b64.npz"), tokens=tokens.numpy().astype(np.int16), id2=id2.numpy().astype(np.int16),
                        gap=gap.numpy(), x0_bf16=x0.to(torch.bfloat16).view(torch.int16).numpy(), z=z.numpy())


def stage_decode16():
    """the REFERENCE's VAE decode (the in-repo mirror the pipeline run executes, `self.vae.decode(z)[0]` + norm_ip, SelftokPipeline.py:285-292) of
    the 16 final latents stored in pipeline_b16.npz, in one batch as the pipeline does: a crc32 of every image's bf16 pixels (pixels themselves
    would be 6 MB) + their mean, so that a decoder can be checked for BIT equality with the reference's on 16 images."""
    import zlib
    H.install()
    import mimogpt.infer.SelftokPipeline as SP
    vae = _MirrorVAE.from_pretrained(None)
    g = np.load(os.path.join(GOLD, "pipeline_b16.npz"))
    lat = torch.from_numpy(g["lat"])
    from mimogpt.models.selftok.sd3.sd3_impls import SD3LatentFormat
    z = SD3LatentFormat().process_out(lat).to(torch.bfloat16)
    rec = vae.decode(z)[0]
    SP.norm_ip(rec, -1, 1)                                                              # in place, as the pipeline (:135-137, 292)
    bits = rec.contiguous().view(torch.int16).numpy().view(np.uint16)
    crc = np.array([zlib.crc32(np.ascontiguousarray(bits[i]).tobytes()) for i in range(bits.shape[0])], dtype=np.uint32)
    orig = (synth.synthetic_images(16) + 1.0) / 2.0
    mse = ((rec.float() - orig) ** 2).reshape(16, -1).double().mean(dim=1)
    psnr = (10.0 * torch.log10(1.0 / mse)).numpy()
    assert np.allclose(psnr, g["psnr_ref"], atol=1e-9), (psnr, g["psnr_ref"])          # the same pixels the pipeline run produced
    vsd = W.synthetic_vae_state_dict()
    from oracle import vae_exact as VX
    px = VX.decode(VX.pack_weights(vsd), VX.bf16_bits(z[:2].permute(0, 2, 3, 1)))
    mine = OM.norm_ip(VX.bits_to_torch(px).permute(0, 3, 1, 2).contiguous())
    ok = bool(torch.equal(mine, rec[:2]))
    report("decode16", images=16, oracle_vae_exact_decode_equals_reference_first2=ok, pixel_mean=float(rec.float().mean()))
    np.savez_compressed(os.path.join(GOLD, "decode_b16.npz"), crc=crc, mean=rec.float().reshape(16, -1).double().mean(dim=1).numpy(),
                        head=bits.reshape(16, -1)[:, :256].copy())


def stage_res(R, B=16):
    """The reference SelftokPipeline end to end at another image size (`datasize` = R, `enable_enc_variable_size`; models_ours.py:183-202,
    SelftokPipeline.py:262) on B = 16 synthetic R x R images in one batch: VAE latents (bf16-exact), pre-quantizer features, ids, the 50-step
    loop's final latents, a crc32 of every reconstructed image's bf16 pixels and its PSNR -- what `vae_mode='exact'`, `encoder_mode='exact'`
    and `gemm='exact'` are checked against at 128 / 320 px."""
    import zlib
    H.install()
    import mimogpt.infer.SelftokPipeline as SP
    cfg = H.load_cfg(CFG_256)
    SP.AutoencoderKL = _MirrorVAE
    shapes = W.expected_shapes(512)
    real_load = torch.load
    torch.load = lambda *a, **k: W.synthetic_state_dict(shapes)
    try:
        with H.fast_init():
            pipe = SP.SelftokPipeline(cfg=cfg, ckpt_path="synthetic", sd3_path="synthetic", datasize=R, device="cpu")
    finally:
        torch.load = real_load
    images = synth.synthetic_images(B, size=R)
    cap = {}
    hk1 = pipe.model.encoder.register_forward_pre_hook(lambda m, args: cap.setdefault("x0", args[0].detach().clone()))
    hk2 = pipe.model.encoder.quantizer.project_in.register_forward_hook(lambda m, i, o: cap.setdefault("z", o.detach().clone()))
    t0 = time.time()
    tokens = pipe.encoding(images, device="cpu")
    hk1.remove(); hk2.remove()
    print(f"[ref] encoding {R} px B={B} {time.time() - t0:.1f}s", flush=True)
    x0, z = cap["x0"], cap["z"]
    assert torch.equal(x0, x0.to(torch.bfloat16).float()) and tuple(x0.shape) == (B, 16, R // 8, R // 8)
    sd = dict(pipe.model.state_dict())
    cb = sd["encoder.quantizer._codebook.embed"][0]
    xn = torch.nn.functional.normalize(z.reshape(-1, 16), dim=-1)
    top2 = (xn @ cb.T).topk(2, dim=-1)
    gap = (top2.values[:, 0] - top2.values[:, 1]).reshape(B, 512)
    np.savez_compressed(os.path.join(GOLD, f"res{R}_b16.npz"), tokens=tokens.numpy().astype(np.int16), gap=gap.numpy(),
                        x0_bf16=x0.to(torch.bfloat16).view(torch.int16).numpy(), z=z.numpy())          # the encode half first: the decode takes most of an hour
    noise = synth.hash_normalish(0xA0 + R, (B, 16, R // 8, R // 8), "cpu")
    real_randn = torch.randn
    torch.randn = lambda *a, **k: noise.clone()
    real_loop = pipe.flow.p_sample_loop

    def loop(*a, **k):
        cap["lat"] = real_loop(*a, **k)
        return cap["lat"]
    pipe.flow.p_sample_loop = loop
    t0 = time.time()
    try:
        rec = pipe.decoding(tokens.numpy(), device="cpu")
    finally:
        torch.randn = real_randn
        pipe.flow.p_sample_loop = real_loop
    print(f"[ref] decoding {R} px B={B} {time.time() - t0:.1f}s", flush=True)
    lat = cap["lat"].detach().float()
    assert tuple(rec.shape) == (B, 3, R, R)
    bits = rec.to(torch.bfloat16).contiguous().view(torch.int16).numpy().view(np.uint16)
    assert torch.equal(rec.to(torch.bfloat16).float(), rec.float())                      # the pipeline's pixels are bf16 values
    crc = np.array([zlib.crc32(np.ascontiguousarray(bits[i]).tobytes()) for i in range(B)], dtype=np.uint32)
    orig = (images + 1.0) / 2.0
    mse = ((rec.float() - orig) ** 2).reshape(B, -1).double().mean(dim=1)
    psnr = (10.0 * torch.log10(1.0 / mse)).numpy()
    report(f"res{R}", images=B, min_gap=float(gap.min()), psnr_ref_mean=float(psnr.mean()), pixel_mean=float(rec.float().mean()))
    np.savez_compressed(os.path.join(GOLD, f"res{R}_b16.npz"), tokens=tokens.numpy().astype(np.int16), gap=gap.numpy(),
                        x0_bf16=x0.to(torch.bfloat16).view(torch.int16).numpy(), lat=lat.numpy(), crc=crc, psnr_ref=psnr,
                        head=bits.reshape(B, -1)[:, :256].copy())


def stage_renderer16(B=16):
    """The reference `SelftokPipeline.decoding_with_renderer` (SelftokPipeline.py:296-322; MMDiT_Renderer.forward sd3/mmdit.py:1511-1620) on B = 16 id
    rows in one batch with the renderer config: the one-pass latent and a crc32 of every image's bf16 pixels -- what `gemm='exact'` on the renderer
    is checked against (16 rows: below that MKL takes another path for the conditioning Linears)."""
    import zlib
    H.install()
    import mimogpt.infer.SelftokPipeline as SP
    cfg = H.load_cfg(CFG_RND)
    SP.AutoencoderKL = _MirrorVAE
    shapes = W.expected_shapes(512, renderer=True)
    real_load = torch.load
    torch.load = lambda *a, **k: W.synthetic_state_dict(shapes)
    try:
        with H.fast_init():
            pipe = SP.SelftokPipeline(cfg=cfg, ckpt_path="synthetic", sd3_path="synthetic", datasize=256, device="cpu")
    finally:
        torch.load = real_load
    ids = synth.synthetic_token_ids(B)
    cap = {}
    def keep(m, i, o):                      # (a hook that RETURNS something replaces the module's output)
        cap["lat"] = o[0].detach().clone()
    hk = pipe.model.model.register_forward_hook(keep)
    t0 = time.time()
    rec = pipe.decoding_with_renderer(ids, device="cpu")
    hk.remove()
    print(f"[ref] decoding_with_renderer B={B} {time.time() - t0:.1f}s", flush=True)
    lat = cap["lat"].float()
    assert tuple(rec.shape) == (B, 3, 256, 256) and rec.dtype == torch.bfloat16
    bits = rec.contiguous().view(torch.int16).numpy().view(np.uint16)
    crc = np.array([zlib.crc32(np.ascontiguousarray(bits[i]).tobytes()) for i in range(B)], dtype=np.uint32)
    report("renderer16", images=B, latent_absmax=float(lat.abs().max()), pixel_mean=float(rec.float().mean()))
    np.savez_compressed(os.path.join(GOLD, "renderer_b16.npz"), ids=ids, latent=lat.numpy(), crc=crc, head=bits.reshape(B, -1)[:, :256].copy())


def stage_k1024_pipe16(B=16):
    """BASELINE configs[2] end to end: the reference SelftokPipeline built with k = 1024 (stage split ASSUMED 384,368,144,96,32 as in stage_k1024: the
    reference ships no 1024-token config) on B = 16 synthetic images in one batch -- ids from pixels, the final latents of the 50-step loop (1024 context
    slots + 256 image tokens per sample), a crc32 of every image's bf16 pixels and its PSNR: what the three exact modes are checked against at K = 1024."""
    import zlib
    H.install()
    import mimogpt.infer.SelftokPipeline as SP
    cfg = H.load_cfg(CFG_256)
    cfg.tokenizer.params.k = 1024
    cfg.tokenizer.params.k_per_stage = "384,368,144,96,32"
    SP.AutoencoderKL = _MirrorVAE
    shapes = W.expected_shapes(1024)
    real_load = torch.load
    torch.load = lambda *a, **k: W.synthetic_state_dict(shapes)
    try:
        with H.fast_init():
            pipe = SP.SelftokPipeline(cfg=cfg, ckpt_path="synthetic", sd3_path="synthetic", datasize=256, device="cpu")
    finally:
        torch.load = real_load
    images = synth.synthetic_images(B)
    t0 = time.time()
    tokens = pipe.encoding(images, device="cpu")
    print(f"[ref] encoding K=1024 B={B} {time.time() - t0:.1f}s", flush=True)
    assert tuple(tokens.shape) == (B, 1024)
    np.savez_compressed(os.path.join(GOLD, "k1024_pipe_b16.npz"), tokens=tokens.numpy().astype(np.int16))
    noise = synth.synthetic_noise(B)
    real_randn = torch.randn
    torch.randn = lambda *a, **k: noise.clone()
    cap = {}
    real_loop = pipe.flow.p_sample_loop

    def loop(*a, **k):
        cap["lat"] = real_loop(*a, **k)
        return cap["lat"]
    pipe.flow.p_sample_loop = loop
    t0 = time.time()
    try:
        rec = pipe.decoding(tokens.numpy(), device="cpu")
    finally:
        torch.randn = real_randn
        pipe.flow.p_sample_loop = real_loop
    print(f"[ref] decoding K=1024 B={B} {time.time() - t0:.1f}s", flush=True)
    lat = cap["lat"].detach().float()
    bits = rec.to(torch.bfloat16).contiguous().view(torch.int16).numpy().view(np.uint16)
    crc = np.array([zlib.crc32(np.ascontiguousarray(bits[i]).tobytes()) for i in range(B)], dtype=np.uint32)
    orig = (images + 1.0) / 2.0
    mse = ((rec.float() - orig) ** 2).reshape(B, -1).double().mean(dim=1)
    psnr = (10.0 * torch.log10(1.0 / mse)).numpy()
    report("k1024_pipe16", images=B, psnr_ref_mean=float(psnr.mean()), pixel_mean=float(rec.float().mean()))
    np.savez_compressed(os.path.join(GOLD, "k1024_pipe_b16.npz"), tokens=tokens.numpy().astype(np.int16), lat=lat.numpy(), crc=crc, psnr_ref=psnr)


def stage_pipeline64(B=64):
    """BASELINE configs[1] at its configured batch: the reference SelftokPipeline end to end on B = 64 synthetic images IN ONE BATCH (encode + 50-step
    decode + VAE decode; ~3 h on 8 cores) -- a crc32 of every image's final latent and of its bf16 pixels, the latents of the first four images in full,
    the PSNR of every image.  The ids must be encode_b64.npz's.  What `bench.py`'s exact leg and the GPU suite check the timed batch against."""
    import zlib
    H.install()
    import mimogpt.infer.SelftokPipeline as SP
    cfg = H.load_cfg(CFG_256)
    SP.AutoencoderKL = _MirrorVAE
    shapes = W.expected_shapes(512)
    real_load = torch.load
    torch.load = lambda *a, **k: W.synthetic_state_dict(shapes)
    try:
        with H.fast_init():
            pipe = SP.SelftokPipeline(cfg=cfg, ckpt_path="synthetic", sd3_path="synthetic", datasize=256, device="cpu")
    finally:
        torch.load = real_load
    images = synth.synthetic_images(B)
    t0 = time.time()
    tokens = pipe.encoding(images, device="cpu")
    print(f"[ref] encoding B={B} {time.time() - t0:.1f}s", flush=True)
    g64 = np.load(os.path.join(GOLD, "encode_b64.npz"))
    assert np.array_equal(tokens.numpy().astype(np.int16), g64["tokens"])
    noise = synth.synthetic_noise(B)
    real_randn = torch.randn
    torch.randn = lambda *a, **k: noise.clone()
    cap = {}
    real_loop = pipe.flow.p_sample_loop

    def loop(*a, **k):
        cap["lat"] = real_loop(*a, **k)
        return cap["lat"]
    pipe.flow.p_sample_loop = loop
    t0 = time.time()
    try:
        rec = pipe.decoding(tokens.numpy(), device="cpu")
    finally:
        torch.randn = real_randn
        pipe.flow.p_sample_loop = real_loop
    print(f"[ref] decoding B={B} {time.time() - t0:.1f}s", flush=True)
    lat = cap["lat"].detach().float().contiguous()
    bits = rec.to(torch.bfloat16).contiguous().view(torch.int16).numpy().view(np.uint16)
    crc = np.array([zlib.crc32(np.ascontiguousarray(bits[i]).tobytes()) for i in range(B)], dtype=np.uint32)
    lat_crc = np.array([zlib.crc32(lat[i].numpy().tobytes()) for i in range(B)], dtype=np.uint32)
    orig = (images + 1.0) / 2.0
    mse = ((rec.float() - orig) ** 2).reshape(B, -1).double().mean(dim=1)
    psnr = (10.0 * torch.log10(1.0 / mse)).numpy()
    b16 = np.load(os.path.join(GOLD, "pipeline_b16.npz"))
    same16 = bool(np.array_equal(lat[:16].numpy(), b16["lat"]))
    report("pipeline64", images=B, psnr_ref_mean=float(psnr.mean()), first16_latents_equal_the_b16_run=same16)
    np.savez_compressed(os.path.join(GOLD, "pipeline_b64.npz"), lat_crc=lat_crc, crc=crc, psnr_ref=psnr, lat4=lat[:4].numpy(), first16_latents_equal_the_b16_run=np.bool_(same16))


def stage_config():
    """the hot-path keys of the reference's two shipped YAMLs (configs/res256/256-eval.yml, configs/renderer/renderer-eval.yml) as the REFERENCE's own
    `parse_args_from_yaml` (infer_utils.py:165-168) returns them: the values `selftoktokenizer_amd.config.default_config` must reproduce (VERDICT r4
    item 2).  Only the keys default_config carries are stored (values, not YAML text)."""
    H.install()
    from mimogpt.infer.infer_utils import parse_args_from_yaml
    from selftoktokenizer_amd.config import default_config

    absent = {}

    def pick(ref, mine, where, name):
        out = {}
        for k, v in mine.items():
            if k not in ref:                              # a key the YAML leaves to the code's default (e.g. context_see_xt: kwargs.get(..., False), image_tokenizer.py:158)
                absent.setdefault(name, []).append(where + k)
                continue
            out[k] = pick(ref[k], v, where + k + ".", name) if isinstance(v, dict) else ref[k]
        return out
    out = {}
    for name, path, rnd in (("k512", CFG_256, False), ("renderer", CFG_RND, True)):
        ref = parse_args_from_yaml(path)
        out[name] = pick(ref, default_config(512, renderer=rnd), "", name)
    out["absent_in_the_reference_yaml"] = absent
    with open(os.path.join(GOLD, "config_hotpath.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    report("config", absent=absent)


def stage_dit4():
    """the reference MMDiT.forward at B = 16 (>= 16 rows in the conditioning Linears and >= 1024 in the token Linears: the regime where MKL's K-blocking
    is row-count independent -- below 16 rows sgemm takes another path, probed -- the one the B = 16 / 64 pipeline runs live in) at three scheduled
    timesteps, for the exact-order MMDiT mode: crc32 + a sub-sampled copy of the velocity, and a crc32 / head of the image stream after every joint
    block (forward hooks), so that a first differing bit can be located"""
    import zlib
    cfg, model, sd = tokenizer(CFG_256)
    B = 16
    ids = torch.from_numpy(synth.synthetic_token_ids(B))
    with torch.no_grad():
        codes = model.encoder.quantizer.get_output_from_indices(ids)
        ehs = model.encoder.final_layer_norm3(codes.reshape(B, -1, 16))
    x = synth.synthetic_noise(B)
    from selftoktokenizer_amd.schedule import FlowSchedule, DiTiCont
    p = cfg.tokenizer.params
    fs = FlowSchedule(50, 1.0)
    ktab = DiTiCont(1000, 512, p.stages, p.k_per_stage).to_indices(fs.t_long)
    out = {"steps": np.array([0, 25, 49])}
    for j, i in enumerate((0, 25, 49)):
        t = torch.full((B,), float(fs.scheduled_t[i]))
        k = int(ktab[i])
        mask = (torch.arange(512)[None] <= k).expand(B, -1)
        crcs, heads = [], []
        def hook(m, a, o):                                  # (a hook that returns a value replaces the output: return None)
            crcs.append(zlib.crc32(o[1].contiguous().numpy().tobytes()))
            heads.append(o[1][0, 0, :8].numpy().copy())
        hooks = [blk.register_forward_hook(hook) for blk in model.model.joint_blocks]
        with torch.no_grad():
            v, _ = model.model(x, t, encoder_hidden_states=ehs, mask=mask, context_see_xt=True)
        for h in hooks:
            h.remove()
        out[f"vcrc_{j}"] = np.uint32(zlib.crc32(v.contiguous().numpy().tobytes())); out[f"vsub_{j}"] = v[:, :, ::4, ::4].contiguous().numpy(); out[f"k_{j}"] = np.int64(k); out[f"xcrc_{j}"] = np.array(crcs, dtype=np.uint32); out[f"xhead_{j}"] = np.stack(heads)
        report(f"dit4_{j}", step=i, k=k, v_absmax=float(v.abs().max()))
    np.savez_compressed(os.path.join(GOLD, "dit_forward_b16.npz"), **out)


def stage_renderer():
    cfg, model, sd = tokenizer(CFG_RND)
    ids = torch.from_numpy(synth.synthetic_token_ids(1, first_index=7))
    with torch.no_grad():
        codes = model.encoder.quantizer.get_output_from_indices(ids)
        ehs = model.encoder.final_layer_norm3(codes.reshape(1, -1, 16))
        out, _ = model.model(y=None, encoder_hidden_states=ehs)
    out_o = OM.renderer_forward(sd, ehs)
    report("renderer", maxdiff=maxdiff(out, out_o), absmax=float(out.abs().max()))
    np.savez_compressed(os.path.join(GOLD, "renderer_b1.npz"), ids=ids.numpy(), latent=out.numpy())


def _ref_flow(steps=50):
    H.install()
    from mimogpt.models.selftok.sd3.rectified_flow import RectifiedFlow
    return RectifiedFlow(steps, 1.0, None, val_schedule="uniform", shift=1.0, schedule="log_norm",
                         parameterization="velocity", m=0.0, s=1.0, force_recon=False, is_eval=True)


def stage_cfg():
    """classifier-free guidance exactly as the reference's sampler runs it (sd3/rectified_flow.py:258-294 calling
    MMDiT.cfg_inference sd3/mmdit.py:1117-1163): two sampler steps at uncond_scale = 2 from the first schedule entries, plus the
    unconditional / conditional velocities alone at a mid-schedule entry (k = 375)."""
    cfg, model, sd = tokenizer(CFG_256)
    flow = _ref_flow()
    diti = model.diti if hasattr(model, "diti") else None
    from mimogpt.models.selftok.diti_utils import DiTi_cont
    diti = DiTi_cont(1000, 512, cfg.tokenizer.params.stages, cfg.tokenizer.params.k_per_stage)
    ids = torch.from_numpy(synth.synthetic_token_ids(1, first_index=11))
    with torch.no_grad():
        codes = model.encoder.quantizer.get_output_from_indices(ids)
        ehs = model.encoder.final_layer_norm3(codes.reshape(1, -1, 16))
    x = synth.synthetic_noise(1, first_index=11)
    stg, kps = OS.parse_stages(cfg.tokenizer.params.stages, cfg.tokenizer.params.k_per_stage)
    sch = OS.make_schedule(50)
    tables = OM.dit_ctx_tables(sd, 512)
    out = {"ids": ids.numpy(), "scale": np.float32(2.0)}
    xr, xo = x.clone(), x.clone()
    for i in (0, 1):
        t = torch.tensor([flow.scheduled_t[i]] * 1)
        k = diti.to_indices(torch.tensor([flow.timestep_map[i]]).long())
        mask = model.encoder.get_encoder_mask(x, k)
        kw = dict(encoder_hidden_states=ehs, mask=mask, context_see_xt=True)
        with torch.no_grad():
            xr, _ = flow.sample_one_step(model.model, xr, t, index=i, model_kwargs=kw, cfg_scale=2.0)
        xo = OM.sample_one_step(sd, xo, i, ehs, mask, sch, tables, cfg_scale=2.0)
        report(f"cfg_step{i}", lat_maxdiff=maxdiff(xr, xo), k=int(k[0]))
        out[f"lat_after_{i + 1}"] = xr.numpy()
    i = 30
    t = torch.tensor([flow.scheduled_t[i]])
    k = diti.to_indices(torch.tensor([flow.timestep_map[i]]).long())
    mask = model.encoder.get_encoder_mask(x, k)
    with torch.no_grad():
        vu = model.model.cfg_inference(x, t, None, None, mask=torch.zeros(mask.size(), dtype=torch.int), shape=512)
        vc, _ = model.model(x, t, None, ehs, mask=mask, shape=512)
    vu_o = OM.cfg_uncond_forward(sd, x, t, 512, tables)
    vc_o = OM.dit_forward(sd, x, t, ehs, mask, False, tables)
    report("cfg_velocities", uncond_maxdiff=maxdiff(vu, vu_o), cond_maxdiff=maxdiff(vc, vc_o), k=int(k[0]), index=i)
    out.update(v_uncond=vu.numpy(), v_cond=vc.numpy(), index=np.int64(i), k=np.int64(int(k[0])))
    np.savez_compressed(os.path.join(GOLD, "cfg_b1.npz"), **out)


def stage_cfg16():
    """classifier-free guidance at B = 16 (the row regime of MKL's K-blocking the exact-order MMDiT mode reproduces): two guided sampler steps of the
    reference (sd3/rectified_flow.py:258-294 -> MMDiT.cfg_inference + MMDiT.forward without context_see_xt) from the first schedule entries; crc32 + a
    sub-sampled copy of the latents after each step"""
    import zlib
    cfg, model, sd = tokenizer(CFG_256)
    flow = _ref_flow()
    from mimogpt.models.selftok.diti_utils import DiTi_cont
    diti = DiTi_cont(1000, 512, cfg.tokenizer.params.stages, cfg.tokenizer.params.k_per_stage)
    B = 16
    ids = torch.from_numpy(synth.synthetic_token_ids(B, first_index=11))
    with torch.no_grad():
        codes = model.encoder.quantizer.get_output_from_indices(ids)
        ehs = model.encoder.final_layer_norm3(codes.reshape(B, -1, 16))
    x = synth.synthetic_noise(B, first_index=11)
    out = {"scale": np.float32(2.0)}
    xr = x.clone()
    for i in (0, 1):
        t = torch.tensor([flow.scheduled_t[i]] * B)
        k = diti.to_indices(torch.tensor([flow.timestep_map[i]] * B).long())
        mask = model.encoder.get_encoder_mask(x, k)
        kw = dict(encoder_hidden_states=ehs, mask=mask, context_see_xt=True)
        with torch.no_grad():
            xr, _ = flow.sample_one_step(model.model, xr, t, index=i, model_kwargs=kw, cfg_scale=2.0)
        out[f"crc_{i + 1}"] = np.uint32(zlib.crc32(xr.contiguous().numpy().tobytes())); out[f"sub_{i + 1}"] = xr[:, :, ::4, ::4].contiguous().numpy()
        report(f"cfg16_step{i}", k=int(k[0]), absmax=float(xr.abs().max()))
    np.savez_compressed(os.path.join(GOLD, "cfg_b16.npz"), **out)


def stage_sampler_options():
    """two dormant branches of the reference's sampler, from the reference's own RectifiedFlow.sample_one_step on the real MMDiT:
    (a) `parameterization: x0` (euler_step, sd3/rectified_flow.py:305-307): two steps from the first schedule entries;
    (b) a NON-prefix `super_mask` (p_sample_loop's mask * super_mask, :226-227): two velocity steps with a hash-random visibility
        pattern over the 512 tokens (the step mask arange(K) <= k times the pattern)."""
    cfg, model, sd = tokenizer(CFG_256)
    H.install()
    from mimogpt.models.selftok.sd3.rectified_flow import RectifiedFlow
    from mimogpt.models.selftok.diti_utils import DiTi_cont
    diti = DiTi_cont(1000, 512, cfg.tokenizer.params.stages, cfg.tokenizer.params.k_per_stage)
    ids = torch.from_numpy(synth.synthetic_token_ids(1, first_index=21))
    with torch.no_grad():
        codes = model.encoder.quantizer.get_output_from_indices(ids)
        ehs = model.encoder.final_layer_norm3(codes.reshape(1, -1, 16))
    x = synth.synthetic_noise(1, first_index=21)
    sch = OS.make_schedule(50)
    tables = OM.dit_ctx_tables(sd, 512)
    sup = (synth.hash_u32(0x5A5A, 512) % 3 != 0)                 # ~2/3 of the tokens visible, no structure
    out = {"ids": ids.numpy(), "super_mask": sup.numpy()}
    for name, param, smask in (("x0", "x0", None), ("supermask", "velocity", sup)):
        flow = RectifiedFlow(50, 1.0, None, val_schedule="uniform", shift=1.0, schedule="log_norm", parameterization=param, m=0.0, s=1.0,
                             force_recon=False, is_eval=True)
        xr, xo = x.clone(), x.clone()
        for i in (0, 1):
            t = torch.tensor([flow.scheduled_t[i]])
            k = diti.to_indices(torch.tensor([flow.timestep_map[i]]).long())
            mask = model.encoder.get_encoder_mask(x, k)
            if smask is not None:
                mask = mask * smask[None]
            kw = dict(encoder_hidden_states=ehs, mask=mask, context_see_xt=True)
            with torch.no_grad():
                xr, _ = flow.sample_one_step(model.model, xr, t, index=i, model_kwargs=kw, cfg_scale=1.0)
            xo = OM.sample_one_step(sd, xo, i, ehs, mask.bool(), sch, tables, parameterization=param)
            report(f"sampler_{name}_step{i}", lat_maxdiff=maxdiff(xr, xo), k=int(k[0]), visible=int(mask.sum()))
            out[f"{name}_after_{i + 1}"] = xr.numpy()
    np.savez_compressed(os.path.join(GOLD, "sampler_options_b1.npz"), **out)


def stage_k1024():
    """BASELINE configs[2]: the reference's own ImageTokenizer built with k = 1024 (query_tokens / context_pos_embed grow, stage
    split ASSUMED 384,368,144,96,32 -- the reference ships no 1024 config): encoder features + ids, and one MMDiT.forward."""
    cfg = H.load_cfg(CFG_256)
    cfg.tokenizer.params.k = 1024
    cfg.tokenizer.params.k_per_stage = "384,368,144,96,32"
    t0 = time.time()
    model, ref_sd = H.build_tokenizer(cfg)
    sd = dict(model.state_dict())
    print(f"[ref] built k=1024 tokenizer in {time.time() - t0:.1f}s", flush=True)
    mine = W.expected_shapes(1024)
    ref_nd = {k: tuple(v.shape) for k, v in sd.items() if not k.startswith("diffusion.")}
    assert set(ref_nd) == set(mine) and all(tuple(mine[k]) == ref_nd[k] for k in mine)
    x0 = synth.synthetic_latents(1, first_index=5)
    cap = {}
    hk = model.encoder.quantizer.project_in.register_forward_hook(lambda m, i, o: cap.__setitem__("z", o.detach().clone()))
    with torch.no_grad():
        outs_q, ids = model.encoder(x0, d=None)
    hk.remove()
    z_o = OM.encoder_features(sd, x0)
    ids_o = OM.vq_ids(sd, z_o)
    report("k1024_encoder", z_maxdiff=maxdiff(z_o, cap["z"]), ids_match=float((ids_o == ids).float().mean()))
    cb = sd["encoder.quantizer._codebook.embed"][0]
    xn = torch.nn.functional.normalize(cap["z"].reshape(-1, 16), dim=-1)
    top2 = (xn @ cb.T).topk(2, dim=-1).values
    gap = (top2[:, 0] - top2[:, 1]).reshape(1, 1024)
    x = synth.synthetic_noise(1, first_index=5)
    tval, k = 0.62, 750
    t = torch.full((1,), tval)
    mask = torch.arange(1024)[None] <= k
    with torch.no_grad():
        v, _ = model.model(x, t, encoder_hidden_states=outs_q, mask=mask, context_see_xt=True)
    v_o = OM.dit_forward(sd, x, t, outs_q, mask, True)
    report("k1024_dit", v_maxdiff=maxdiff(v, v_o), v_absmax=float(v.abs().max()))
    np.savez_compressed(os.path.join(GOLD, "k1024_b1.npz"), z=cap["z"].numpy(), ids=ids.numpy(), gap=gap.numpy(), outs_q=outs_q.numpy(),
                        v=v.numpy(), t=np.float32(tval), k=np.int64(k))


def stage_k1024_16():
    """BASELINE configs[2] in the exact-order regime: the reference's ImageTokenizer(k = 1024) at B = 16 -- encoder features + ids from 16 latents (crc32, and the
    features of two images in full) and one MMDiT.forward at a scheduled timestep (crc32 + sub-sample): for the exact-order encoder / MMDiT at K = 1024"""
    import zlib
    cfg = H.load_cfg(CFG_256)
    cfg.tokenizer.params.k = 1024
    cfg.tokenizer.params.k_per_stage = "384,368,144,96,32"
    model, ref_sd = H.build_tokenizer(cfg)
    B = 16
    x0 = synth.synthetic_latents(B, first_index=5).to(torch.bfloat16).float()
    cap = {}
    hk = model.encoder.quantizer.project_in.register_forward_hook(lambda m, i, o: cap.__setitem__("z", o.detach().clone()))
    with torch.no_grad():
        outs_q, ids = model.encoder(x0, d=None)
    hk.remove()
    from selftoktokenizer_amd.schedule import FlowSchedule, DiTiCont
    fs = FlowSchedule(50, 1.0)
    ktab = DiTiCont(1000, 1024, cfg.tokenizer.params.stages, cfg.tokenizer.params.k_per_stage).to_indices(fs.t_long)
    i = 25
    x = synth.synthetic_noise(B, first_index=5)
    t = torch.full((B,), float(fs.scheduled_t[i]))
    k = int(ktab[i])
    mask = (torch.arange(1024)[None] <= k).expand(B, -1)
    with torch.no_grad():
        v, _ = model.model(x, t, encoder_hidden_states=outs_q, mask=mask, context_see_xt=True)
    report("k1024_16", k=k, step=i, v_absmax=float(v.abs().max()))
    np.savez_compressed(os.path.join(GOLD, "k1024_b16.npz"), zcrc=np.uint32(zlib.crc32(cap["z"].contiguous().numpy().tobytes())), z2=cap["z"][:2].numpy(),
                        ids=ids.numpy().astype(np.int16), vcrc=np.uint32(zlib.crc32(v.contiguous().numpy().tobytes())), vsub=v[:, :, ::4, ::4].contiguous().numpy(),
                        step=np.int64(i), k=np.int64(k))


def stage_vqtrain():
    """training-side codebook maintenance: the reference's own CosineSimCodebook in train() mode, three EMA steps (dead-code expiry
    off so that every number is deterministic), then the dead-code mask / sampling weights of a fourth state, and one k-means
    iteration from given seeds.  Pins oracle/vq_train.py."""
    H.install()
    import torch.distributed as dist
    from oracle import vq_train as VT
    from mimogpt.models.selftok.vector_quantize_pytorch import CosineSimCodebook, gumbel_sample, kmeans, l2norm
    from functools import partial
    if not dist.is_initialized():                                  # the training forward calls distributed.get_world_size()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
        dist.init_process_group("gloo", rank=0, world_size=1)
    C, D, K, B, decay = 2048, 16, 32, 16, 0.99
    gs = partial(gumbel_sample, stochastic=False, reinmax=False, straight_through=False)
    cb = CosineSimCodebook(dim=D, codebook_size=C, kmeans_init=False, decay=decay, threshold_ema_dead_code=0, use_ddp=False,
                           gumbel_sample=gs, sample_codebook_temp=1.0, smart_re_K=K)
    embed0 = l2norm(synth.hash_normalish(0xC0DEB00C, (C, D)))
    cb.embed.data.copy_(embed0[None]); cb.embed_avg.data.copy_(embed0[None])
    cb.train()
    st = VT.new_state(embed0, K)
    out = {"embed0": embed0.numpy(), "decay": np.float32(decay)}
    worst = 0.0
    for step in range(3):
        z = synth.hash_normalish(0x7A11 + step, (B, K, D))
        x = l2norm(z)
        with torch.no_grad():
            _, ids, _, _ = cb(x)
        ids_o = VT.train_step(st, x, decay)
        assert torch.equal(ids_o, ids), step
        for name in ("embed", "embed_avg", "cluster_size", "timestep_p_over_c"):
            ref = getattr(cb, name)[0]
            worst = max(worst, maxdiff(ref, st[name]))
            out[f"{name}_{step}"] = ref.numpy().copy()
        out[f"ids_{step}"] = ids.numpy()
        out[f"delta_embed_{step}"] = np.float32(float(cb.delta_embed))
    report("vqtrain_ema", steps=3, worst_maxdiff=worst)
    w_ref = cb.compute_timestep_weight()[0]
    out["timestep_weight"] = w_ref.numpy()
    thr, reset = VT.scaled_thresholds(0.2, 0.2, B, K, 1, C)
    out["thr_abs"], out["reset_abs"] = np.float32(thr), np.float32(reset)
    out["expired"] = (cb.cluster_size[0] < thr).numpy()
    report("vqtrain_weights", tw_maxdiff=maxdiff(w_ref, VT.timestep_weight(st)), expired=int(out["expired"].sum()),
           expired_equal=bool(torch.equal(VT.expired_codes(st, thr), torch.from_numpy(out["expired"]))))
    # one k-means iteration from given seeds (kmeans samples its seeds at random: inject them)
    samples = l2norm(synth.hash_normalish(0x5EED5, (1, 4096, D)))
    seeds = samples[:, :256].clone()
    means, bins = kmeans(samples, 256, num_iters=1, use_cosine_sim=True, sample_fn=lambda s, n: seeds)
    m_o, b_o = VT.kmeans_iteration(samples[0], seeds[0])
    report("vqtrain_kmeans", means_maxdiff=maxdiff(means[0], m_o), bins_equal=bool(torch.equal(bins[0], b_o)))
    out["kmeans_means"], out["kmeans_bins"] = means[0].numpy(), bins[0].numpy()
    np.savez_compressed(os.path.join(GOLD, "vqtrain.npz"), **out)


def stage_vq_entropy():
    """entropy regularisers of the training forward (vector_quantize_pytorch.py:1006-1031): the reference's OWN calc_entropy /
    calc_ema_entropy / get_group_perplexity on the `distances` its CosineSimCodebook returns in train() mode (after that forward's
    timestep_p_over_c update), and torch autograd through them for d(diversity_loss)/dz.  Pins oracle/vq_train.py:entropy_terms.
    (VectorQuantize.forward itself cannot reach these lines: it passes min_ref= to calc_entropy, a TypeError.)"""
    H.install()
    import torch.distributed as dist
    from oracle import vq_train as VT
    from mimogpt.models.selftok.vector_quantize_pytorch import CosineSimCodebook, gumbel_sample, l2norm, calc_entropy, calc_ema_entropy
    from functools import partial
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
        dist.init_process_group("gloo", rank=0, world_size=1)
    C, D, K, B = 2048, 16, 128, 6
    gs = partial(gumbel_sample, stochastic=False, reinmax=False, straight_through=False)
    cb = CosineSimCodebook(dim=D, codebook_size=C, kmeans_init=False, decay=0.99, threshold_ema_dead_code=0, use_ddp=False,
                           gumbel_sample=gs, sample_codebook_temp=1.0, smart_re_K=K)
    embed0 = l2norm(synth.hash_normalish(0xC0DEB00C, (C, D)))
    cb.embed.data.copy_(embed0[None]); cb.embed_avg.data.copy_(embed0[None])
    cb.train()
    with torch.enable_grad():                                        # the generator runs with autograd off
        return _vq_entropy_cases(cb, embed0, l2norm, calc_entropy, calc_ema_entropy, VT, C, D, K, B)


def _vq_entropy_cases(cb, embed0, l2norm, calc_entropy, calc_ema_entropy, VT, C, D, K, B):
    out = {}
    worst = 0.0
    for case, (dw, ratio, reg, seed) in enumerate(((0.1, 0.7, [0.25, 0.5], 0xE17), (1.0, 0.4, [0.001, 0.9], 0xE18))):
        z = synth.hash_normalish(seed, (B, K, D)).requires_grad_(True)
        _, ids, distances, _ = cb(l2norm(z), freeze_codebook=True)                 # [1, B, K, C]; updates timestep_p_over_c first (:568-578)
        tpc = cb.timestep_p_over_c[0].clone()
        scaled = distances * 10.0
        e_max, e_min = calc_entropy(scaled.flatten(end_dim=-2))
        c_ent, g_ent = calc_ema_entropy(scaled, tpc, ratio_d=1. - ratio)
        perp = cb.get_group_perplexity().mean()
        frac = perp / C
        w = 0.5 if frac < reg[0] else max((0.5 - 0.5 / (reg[1] - reg[0]) * (frac - reg[0])), 0.0)
        loss = -dw * w * (0.5 * (c_ent + g_ent))
        (gz,) = torch.autograd.grad(loss, z, retain_graph=True)
        (gz_max,) = torch.autograd.grad(-dw * e_max, z)                              # the smart_re_K == 0 branch (:1028)
        o = VT.entropy_terms(z, embed0, tpc, dw, True, ratio, reg)
        (gz_o,) = torch.autograd.grad(o["diversity_loss"], z)
        ref = dict(entropy_to_max=e_max, entropy_to_min=e_min, codebook_entropy=c_ent, group_entropy=g_ent, perplexity=perp, diversity_loss=loss)
        for k, v in ref.items():
            worst = max(worst, maxdiff(v.detach(), o[k].detach()))
            out[f"{k}_{case}"] = np.float32(float(v))
        worst = max(worst, maxdiff(gz, gz_o) / float(gz.abs().max()))
        out[f"tpc_{case}"], out[f"grad_z_{case}"], out[f"grad_z_entropy_to_max_{case}"] = tpc.numpy(), gz.numpy(), gz_max.numpy()
        out[f"ids_{case}"] = ids.numpy()
        out[f"args_{case}"] = np.asarray([dw, ratio, reg[0], reg[1], float(w)], np.float64)
        out[f"seed_{case}"] = np.int64(seed)
    out["embed0_seed"] = np.int64(0xC0DEB00C)
    report("vq_entropy", cases=2, worst_oracle_vs_reference=worst)
    np.savez_compressed(os.path.join(GOLD, "vq_entropy.npz"), **out)


def stage_rmsnorm_rotary():
    """the two optional / off-path element-wise ops of SURVEY 8a (a31, a32), from the reference's OWN classes: RMSNorm (modules.py:
    49-95; inactive in the shipped configs, qk_norm unset) with and without the learnable scale, and apply_rotary_emb
    (utils/rotary_embedding_torch.py:37-53; no call site in the reference) incl. the partial-rotation (start_index) and scale forms.
    Inputs are regenerated from synth by seed; the file holds the reference's outputs only."""
    H.install()
    from mimogpt.models.selftok.modules import RMSNorm
    from mimogpt.utils.rotary_embedding_torch import apply_rotary_emb
    out = {}
    x = synth.hash_uniform(14, (7, 24, 64), -2.0, 2.0)
    w = synth.hash_uniform(15, (64,), 0.9, 1.1)
    with H.fast_init():
        rn = RMSNorm(64, elementwise_affine=True, eps=1e-6)
    with torch.no_grad():
        rn.weight.copy_(w)
        out["rms_affine"] = rn(x).numpy()
        with H.fast_init():
            out["rms_plain"] = RMSNorm(64, elementwise_affine=False, eps=1e-6)(x).numpy()
        x2 = synth.hash_uniform(18, (5, 3, 256), -4.0, 4.0)                      # the widest row the kernel takes (dim <= 256), eps 1e-5
        with H.fast_init():
            r2 = RMSNorm(256, elementwise_affine=False, eps=1e-5)
        out["rms_256"] = r2(x2).numpy()
        t = synth.hash_uniform(16, (2, 3, 10, 32), -2.0, 2.0)
        f = synth.hash_uniform(17, (10, 32), -3.0, 3.0)
        out["rot_full"] = apply_rotary_emb(f, t).numpy()
        f16 = synth.hash_uniform(19, (10, 16), -3.0, 3.0)
        out["rot_partial_start8"] = apply_rotary_emb(f16, t, start_index=8).numpy()   # rotates features 8..23 only
        out["rot_scaled"] = apply_rotary_emb(f, t, scale=0.5).numpy()
    np.savez_compressed(os.path.join(GOLD, "rmsnorm_rotary.npz"), **out)
    report("rmsnorm_rotary", arrays=sorted(out), rms_absmax=float(np.abs(out["rms_affine"]).max()), rot_absmax=float(np.abs(out["rot_full"]).max()))


STAGES = dict(encoder_prenorm=stage_encoder_prenorm, pipeline64=stage_pipeline64, k1024_pipe16=stage_k1024_pipe16, renderer16=stage_renderer16, res128=lambda: stage_res(128), res320=lambda: stage_res(320), k1024_16=stage_k1024_16, cfg16=stage_cfg16, dit4=stage_dit4, config=stage_config, decode16=stage_decode16, encode64=stage_encode64, vq_entropy=stage_vq_entropy, rmsnorm_rotary=stage_rmsnorm_rotary, sampler_options=stage_sampler_options, keys=stage_keys, vq=stage_vq, schedule=stage_schedule, encoder=stage_encoder, dit=stage_dit,
              vae=stage_vae, pipeline=stage_pipeline, pipeline16=stage_pipeline16, renderer=stage_renderer, cfg=stage_cfg, k1024=stage_k1024, vqtrain=stage_vqtrain)

if __name__ == "__main__":
    torch.set_grad_enabled(False)
    os.makedirs(GOLD, exist_ok=True)
    names = sys.argv[1:] or ["vq", "schedule", "encoder", "dit", "vae", "pipeline", "keys", "renderer", "cfg", "k1024"]
    for n in names:
        t0 = time.time()
        STAGES[n]()
        print(f"[stage] {n} done in {time.time() - t0:.1f}s", flush=True)
    path = os.path.join(GOLD, "PINNING.json")
    old = json.load(open(path)) if os.path.exists(path) else {}
    old.update(REPORT)
    json.dump(old, open(path, "w"), indent=1, sort_keys=True)
