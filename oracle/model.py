"""oracle/model.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

torch-CPU fp32 restatement of the floating-point part of the Selftok hot path, written from the
reference's call stacks over the reference's flat checkpoint layout.
Each function cites the reference file:line it follows (paths relative to
mimogpt/models/selftok/ unless stated).  Pinned against the reference itself (imported in the
build container, tools/oracle/gen_golden.py) through tests/golden/*.npz.

What is *deliberately not* restated is the reference's redundant work, which does not change any
result: the 50 encoder passes per decode that only produce `arange(K) <= k`
(sd3/rectified_flow.py:215), the per-call recomputation of the input-independent adaLN tables
(modules.py:312-318, sd3/mmdit.py:446-458), the one-hot/perplexity logging in the quantizer
(vector_quantize_pytorch.py:136,957-975).

VAE: the arithmetic of record is diffusers==0.32.2 AutoencoderKL (absent offline); this file
follows the in-repo architectural mirror sd3/sd3_impls.py:215-474 with diffusers key names.
"parity unpinned" w.r.t. diffusers; pinned against the mirror only.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import clib, schedule

SD = Dict[str, torch.Tensor]

ENC_DEPTH = 16
DIT_DEPTH = 24


# ----------------------------------------------------------------------------------------------
# small pieces
# ----------------------------------------------------------------------------------------------

def lin(sd: SD, name: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def ln(x: torch.Tensor) -> torch.Tensor:
    """nn.LayerNorm(elementwise_affine=False, eps=1e-6) (modules.py:104-106; sd3/mmdit.py:389,404,628)"""
    return F.layer_norm(x, (x.shape[-1],), None, None, 1e-6)


def timestep_embedding(t: torch.Tensor, dim: int = 256, max_period: int = 10000) -> torch.Tensor:
    """models.py:56-74 / sd3/mmdit.py:156-175: [cos(t f_i), sin(t f_i)], f_i = exp(-ln(1e4) i/half)"""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def t_embedder(sd: SD, prefix: str, t: torch.Tensor) -> torch.Tensor:
    """TimestepEmbedder.forward: Linear -> SiLU -> Linear (models.py:76-79; sd3/mmdit.py:177-183)"""
    h = lin(sd, prefix + ".mlp.0", timestep_embedding(t))
    return lin(sd, prefix + ".mlp.2", F.silu(h))


def adaln(sd: SD, prefix: str, c: torch.Tensor) -> torch.Tensor:
    """adaLN_modulation = Sequential(SiLU, Linear) (modules.py:297-300; sd3/mmdit.py:428-430)"""
    return lin(sd, prefix + ".adaLN_modulation.1", F.silu(c))


def mlp(sd: SD, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """fc1 -> GELU(tanh) -> fc2 (timm Mlp, modules.py:109,293; sd3/other_impls.py:65-90)"""
    return lin(sd, prefix + ".fc2", F.gelu(lin(sd, prefix + ".fc1", x), approximate="tanh"))


def heads(x: torch.Tensor, n: int) -> torch.Tensor:
    B, L, C = x.shape
    return x.view(B, L, n, C // n).transpose(1, 2)


def unheads(x: torch.Tensor) -> torch.Tensor:
    B, H, L, D = x.shape
    return x.transpose(1, 2).reshape(B, L, H * D)


def crop_pos(pos: torch.Tensor, grid: int, h: int, w: int) -> torch.Tensor:
    """cropped_pos_embed: centre h x w window of the grid x grid table (models_ours.py:183-202; sd3/mmdit.py:878-896)"""
    top, left = (grid - h) // 2, (grid - w) // 2
    p = pos.reshape(1, grid, grid, -1)[:, top:top + h, left:left + w, :]
    return p.reshape(1, h * w, -1)


def patch_embed(sd: SD, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """PatchEmbed.forward: conv k=2 s=2 -> flatten -> NLC (sd3/mmdit.py:66-75)"""
    y = F.conv2d(x, sd[prefix + ".proj.weight"], sd[prefix + ".proj.bias"], stride=2)
    return y.flatten(2).transpose(1, 2)


# ----------------------------------------------------------------------------------------------
# encoder (Enc-Qformer-Uni-XL/2, 'dual')  -- models_ours.py:204-257, 315-343; modules.py:165-327
# ----------------------------------------------------------------------------------------------

def encoder_tables(sd: SD, K: int):
    """per-block [K, 6*512] adaLN tables from positions 1000+8k (modules.py:312-318). Input independent."""
    pos = schedule.get_position(torch.arange(K))
    out = []
    for i in range(ENC_DEPTH):
        p = f"encoder.blocks.{i}"
        out.append(adaln(sd, p, t_embedder(sd, p + ".t_embedder", pos)))
    return out


def dual_block(sd: SD, i: int, x: torch.Tensor, q: torch.Tensor, table: torch.Tensor):
    """DualBlock.forward + DualAttention uni branch (modules.py:310-327, 165-176, 216-274)"""
    p = f"encoder.blocks.{i}"
    B, N, C = x.shape
    sh_msa, sc_msa, g_msa, sh_mlp, sc_mlp, g_mlp = table.chunk(6, dim=1)
    xn = ln(x)
    qn = ln(q) * (1 + sc_msa.unsqueeze(0)) + sh_msa.unsqueeze(0)
    # latent stream self-attention: 4 heads x 16
    qkv = lin(sd, p + ".attn.qkv", xn).reshape(B, N, 3, 4, 16).permute(2, 0, 3, 1, 4)
    xa = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
    # query stream attends to [to_query_kv(x), query kv]: 8 heads x 64, no mask
    Kq = q.shape[1]
    qqkv = lin(sd, p + ".attn.query_linear", qn).reshape(B, Kq, 3, 8, 64).permute(2, 0, 3, 1, 4)
    kv = lin(sd, p + ".attn.to_query_kv", xn).reshape(B, N, 2, 8, 64).permute(2, 0, 3, 1, 4)
    k2 = torch.cat([kv[0], qqkv[1]], dim=2)
    v2 = torch.cat([kv[1], qqkv[2]], dim=2)
    qa = F.scaled_dot_product_attention(qqkv[0], k2, v2)
    xa = lin(sd, p + ".attn.proj", xa.transpose(1, 2).reshape(B, N, C))
    qa = lin(sd, p + ".attn.query_proj", qa.transpose(1, 2).reshape(B, Kq, q.shape[2]))
    x = x + xa
    x = x + mlp(sd, p + ".mlp", ln(x))
    q = q + g_msa.unsqueeze(0) * qa
    q = q + g_mlp.unsqueeze(0) * mlp(sd, p + ".q_mlp", ln(q) * (1 + sc_mlp.unsqueeze(0)) + sh_mlp.unsqueeze(0))
    return x, q


def encoder_features(sd: SD, x0: torch.Tensor, tables=None, pre_norm: bool = False) -> torch.Tensor:
    """x0 [B,16,32,32] fp32 (process_in'ed VAE mean) -> pre-quantizer features z [B,K,16]
    (Encoder.forward up to and incl. quantizer.project_in: models_ours.py:204-221, vq:844).  pre_norm: `outs = self.final_layer_norm(outs)`
    before the quantizer (models_ours.py:219-220; encoder_config.pre_norm, False in the shipped configs)"""
    K = sd["encoder.query_tokens"].shape[1]
    tables = tables or encoder_tables(sd, K)
    B, _, H, W = x0.shape
    grid = int(round(math.sqrt(sd["encoder.pos_embed"].shape[1])))
    x = patch_embed(sd, "encoder.x_embedder", x0) + crop_pos(sd["encoder.pos_embed"], grid, H // 2, W // 2)
    q = sd["encoder.query_tokens"].expand(B, -1, -1)
    for i in range(ENC_DEPTH):
        x, q = dual_block(sd, i, x, q, tables[i])
    if pre_norm:
        q = F.layer_norm(q, (q.shape[-1],), sd["encoder.final_layer_norm.weight"], sd["encoder.final_layer_norm.bias"], 1e-6)
    return lin(sd, "encoder.quantizer.project_in", q)


def vq_ids(sd: SD, z: torch.Tensor) -> torch.Tensor:
    """l2norm + cosine argmax (vector_quantize_pytorch.py:854,561,135) via the bit-exact C oracle"""
    cb = sd["encoder.quantizer._codebook.embed"][0]
    ids, _ = clib.vq_encode(z.reshape(-1, 16).numpy(), cb.numpy())
    return torch.from_numpy(ids).reshape(z.shape[:-1])


def encode_latents(sd: SD, x0: torch.Tensor, tables=None) -> torch.Tensor:
    """`_, tokens = model.encoder(x_0, d=None)` (SelftokPipeline.py:221) -> int64 [B,K]"""
    return vq_ids(sd, encoder_features(sd, x0, tables))


def codes_from_ids(sd: SD, ids: torch.Tensor) -> torch.Tensor:
    """get_output_from_indices + final_layer_norm3 (SelftokPipeline.py:236-240)"""
    cb = sd["encoder.quantizer._codebook.embed"][0]
    codes = cb[ids]
    return F.layer_norm(codes, (16,), sd["encoder.final_layer_norm3.weight"], sd["encoder.final_layer_norm3.bias"], 1e-6)


# ----------------------------------------------------------------------------------------------
# MMDiT decoder -- sd3/mmdit.py:441-553 (blocks), 609-645 (final layer), 992-1101 (forward)
# ----------------------------------------------------------------------------------------------

def dit_ctx_tables(sd: SD, K: int):
    """per-block context-stream tables [K, 6*1536] = adaLN(t_embedder(1000+8k)) (sd3/mmdit.py:446-458);
    the last block's context stream is pre_only and modulated by c instead (:476-483) -> None."""
    pos = schedule.get_position(torch.arange(K))
    out = []
    for i in range(DIT_DEPTH - 1):
        p = f"model.joint_blocks.{i}.context_block"
        out.append(adaln(sd, p, t_embedder(sd, p + ".t_embedder", pos)))
    out.append(None)
    return out


def joint_mask(mask: torch.Tensor, n_x: int, context_see_xt: bool) -> torch.Tensor:
    """bool [B,1,K+n_x,K+n_x] exactly as MMDiT.forward builds it (sd3/mmdit.py:1041-1094):
    context rows see (mask | x iff context_see_xt); image rows see (mask | all x)."""
    B, K = mask.shape
    ones = torch.ones(B, n_x, dtype=torch.bool)
    ctx_row = torch.cat([mask.bool(), ones if context_see_xt else torch.zeros_like(ones)], dim=1)
    img_row = torch.cat([mask.bool(), ones], dim=1)
    ctx = ctx_row[:, None, None, :].expand(B, 1, K, K + n_x)
    img = img_row[:, None, None, :].expand(B, 1, n_x, K + n_x)
    return torch.cat([ctx, img], dim=2)


def joint_blocks(sd: SD, ctx: torch.Tensor, x: torch.Tensor, c: torch.Tensor, amask, tables):
    H = DIT_DEPTH
    for i in range(DIT_DEPTH):
        pc, px = f"model.joint_blocks.{i}.context_block", f"model.joint_blocks.{i}.x_block"
        last = i == DIT_DEPTH - 1
        # context pre-attention
        if not last:
            c_sh1, c_sc1, c_g1, c_sh2, c_sc2, c_g2 = tables[i].chunk(6, dim=1)           # [K,1536] each
            cqkv = lin(sd, pc + ".attn.qkv", ln(ctx) * (1 + c_sc1.unsqueeze(0)) + c_sh1.unsqueeze(0))
        else:
            c_sh1, c_sc1 = adaln(sd, pc, c).chunk(2, dim=1)                               # [B,1536]
            cqkv = lin(sd, pc + ".attn.qkv", ln(ctx) * (1 + c_sc1.unsqueeze(1)) + c_sh1.unsqueeze(1))
        # x pre-attention
        x_sh1, x_sc1, x_g1, x_sh2, x_sc2, x_g2 = adaln(sd, px, c).chunk(6, dim=1)
        xqkv = lin(sd, px + ".attn.qkv", ln(x) * (1 + x_sc1.unsqueeze(1)) + x_sh1.unsqueeze(1))
        # joint attention over cat(context, x) (block_mixing, sd3/mmdit.py:508-553)
        qkv = torch.cat([cqkv, xqkv], dim=1)
        B, S, _ = qkv.shape
        q, k, v = qkv.reshape(B, S, 3, H, 64).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(q, k, v, attn_mask=amask)
        a = a.transpose(1, 2).reshape(B, S, H * 64)
        Kc = ctx.shape[1]
        ca, xa = a[:, :Kc], a[:, Kc:]
        # post-attention
        if not last:
            ctx = ctx + c_g1.unsqueeze(0) * lin(sd, pc + ".attn.proj", ca)
            ctx = ctx + c_g2.unsqueeze(0) * mlp(sd, pc + ".mlp", ln(ctx) * (1 + c_sc2.unsqueeze(0)) + c_sh2.unsqueeze(0))
        x = x + x_g1.unsqueeze(1) * lin(sd, px + ".attn.proj", xa)
        x = x + x_g2.unsqueeze(1) * mlp(sd, px + ".mlp", ln(x) * (1 + x_sc2.unsqueeze(1)) + x_sh2.unsqueeze(1))
    return x


def final_layer(sd: SD, x: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    """FinalLayer.forward (sd3/mmdit.py:641-645)"""
    sh, sc = adaln(sd, "model.final_layer", c).chunk(2, dim=1)
    return lin(sd, "model.final_layer.linear", ln(x) * (1 + sc.unsqueeze(1)) + sh.unsqueeze(1))


def unpatchify(x: torch.Tensor, h: int, w: int, p: int = 2, c: int = 16) -> torch.Tensor:
    """'nhwpqc->nchpwq' (sd3/mmdit.py:898-916)"""
    B = x.shape[0]
    x = x.reshape(B, h, w, p, p, c)
    return torch.einsum("nhwpqc->nchpwq", x).reshape(B, c, h * p, w * p)


def dit_forward(sd: SD, x: torch.Tensor, t: torch.Tensor, ehs: torch.Tensor, mask: torch.Tensor,
                context_see_xt: bool = True, tables=None) -> torch.Tensor:
    """MMDiT.forward (sd3/mmdit.py:992-1101): x [B,16,32,32], t [B] in [0,1], ehs [B,K,16], mask [B,K] bool"""
    K = ehs.shape[1]
    tables = tables or dit_ctx_tables(sd, K)
    B, _, Hh, Ww = x.shape
    t = t * 1000.0
    xe = patch_embed(sd, "model.x_embedder", x) + crop_pos(sd["model.pos_embed"], 192, Hh // 2, Ww // 2)
    c = t_embedder(sd, "model.t_embedder", t)
    ctx = lin(sd, "model.context_embedder", ehs) + sd["model.context_pos_embed"]
    amask = joint_mask(mask, xe.shape[1], context_see_xt)
    xo = joint_blocks(sd, ctx, xe, c, amask, tables)
    return unpatchify(final_layer(sd, xo, c), Hh // 2, Ww // 2)


def cfg_uncond_forward(sd: SD, x: torch.Tensor, t: torch.Tensor, K: int, tables=None) -> torch.Tensor:
    """MMDiT.cfg_inference(x, t, None, None, mask=zeros[B,K], shape=K) (sd3/mmdit.py:1117-1163), the unconditional branch of
    classifier-free guidance: integer-floored timestep, all-zero context WITHOUT context_embedder / context_pos_embed,
    sd3_cond_pooling is the string 'None' in the shipped configs (no pooled y), and ONE mask row for every query -- no
    context key visible, every image key visible (context rows therefore see the image tokens here)."""
    tables = tables or dit_ctx_tables(sd, K)
    B, _, Hh, Ww = x.shape
    t_int = torch.floor(t * 1000).int().clamp(0, 999)
    xe = patch_embed(sd, "model.x_embedder", x) + crop_pos(sd["model.pos_embed"], 192, Hh // 2, Ww // 2)
    c = t_embedder(sd, "model.t_embedder", t_int)
    ctx = torch.zeros(B, K, xe.shape[-1], dtype=xe.dtype)
    row = torch.cat([torch.zeros(B, K), torch.ones(B, xe.shape[1])], dim=1).bool()
    amask = row[:, None, None, :].repeat(1, 1, K + xe.shape[1], 1)
    xo = joint_blocks(sd, ctx, xe, c, amask, tables)
    return unpatchify(final_layer(sd, xo, c), Hh // 2, Ww // 2)


def sample_one_step(sd: SD, x: torch.Tensor, i: int, ehs: torch.Tensor, mask: torch.Tensor, sch, tables=None,
                    cfg_scale: float = 1.0, context_see_xt: bool = True, parameterization: str = "velocity") -> torch.Tensor:
    """RectifiedFlow.sample_one_step + euler_step (sd3/rectified_flow.py:258-309) for schedule entry i; `parameterization` 'velocity'
    (shipped configs) or 'x0' (:305-307: the model output is the clean latent, x_prev = v + a_prev (x - v) / a_t).
    cfg_scale != 1: out = u + s (c - u) with u = cfg_inference(...) and c = model(x, t, None, context, mask=) -- that call does
    not forward context_see_xt, which therefore falls back to False (sd3/mmdit.py:1012)."""
    B = x.shape[0]
    a_t = torch.tensor(sch["scheduled_t"][i])
    a_prev = torch.tensor(sch["scheduled_t_prev"][i])
    t = torch.full((B,), float(sch["scheduled_t"][i]), dtype=torch.float32)
    if cfg_scale == 1.0:
        v = dit_forward(sd, x, t, ehs, mask, context_see_xt, tables)
    else:
        u = cfg_uncond_forward(sd, x, t, ehs.shape[1], tables)
        c = dit_forward(sd, x, t, ehs, mask, False, tables)
        v = u + cfg_scale * (c - u)
    if parameterization == "x0":
        return v + a_prev * (x - v) / a_t
    return x - (a_t - a_prev) * v


def renderer_forward(sd: SD, ehs: torch.Tensor, tables=None) -> torch.Tensor:
    """MMDiT_Renderer.forward(y=None, encoder_hidden_states=ehs) (sd3/mmdit.py:1511-1620)"""
    B, K, _ = ehs.shape
    tables = tables or dit_ctx_tables(sd, K)
    n = sd["model.positional_embedding"].shape[0]
    g = int(round(math.sqrt(n)))
    x = sd["model.mask_token"].repeat(B, n, 1) + sd["model.positional_embedding"]
    t = torch.ones(B) * 1000.0
    c = t_embedder(sd, "model.t_embedder", t)
    ctx = lin(sd, "model.context_embedder", ehs) + sd["model.context_pos_embed"]
    amask = joint_mask(torch.ones(B, K, dtype=torch.bool), n, context_see_xt=False)
    xo = joint_blocks(sd, ctx, x, c, amask, tables)
    return unpatchify(final_layer(sd, xo, c), g, g)


def decode_latent(sd: SD, ids: torch.Tensor, noise: torch.Tensor, stages, k_per_stage, num_steps: int = 50,
                  tables=None, trace: Optional[list] = None, max_steps: Optional[int] = None, uncond_scale: float = 1.0,
                  prefix_k: Optional[int] = None, super_mask: Optional[torch.Tensor] = None,
                  parameterization: str = "velocity") -> torch.Tensor:
    """SelftokPipeline.decoding up to pred_x0 (SelftokPipeline.py:232-282) + p_sample_loop / euler_step
    (sd3/rectified_flow.py:165-256, 258-309).  The pipeline never forwards uncond_scale (cfg_scale == 1); `uncond_scale` exposes
    p_sample_loop's own argument.  `prefix_k`: p_sample_loop's `super_mask` = the first prefix_k tokens (mask * super_mask,
    rectified_flow.py:226-227) -- decoding from a partial token prefix, README.md:241."""
    B, K = ids.shape
    ehs = codes_from_ids(sd, ids)                       # mask at timestep_map[0] (k=K-1) is all-true: ehs * 1
    sch = schedule.make_schedule(num_steps)
    ks = schedule.diti_indices(sch["t_long"], stages, k_per_stage, K)
    tables = tables or dit_ctx_tables(sd, K)
    x = noise.float()
    steps = num_steps if max_steps is None else min(max_steps, num_steps)
    for i in range(steps):
        mask = (torch.arange(K)[None, :] <= int(ks[i])).expand(B, K)
        if prefix_k is not None:
            mask = mask & (torch.arange(K)[None, :] < int(prefix_k))
        if super_mask is not None:                       # mask = mask * super_mask (rectified_flow.py:226-227), any visibility pattern
            mask = mask & torch.as_tensor(super_mask).bool().reshape(-1, K)
        x = sample_one_step(sd, x, i, ehs, mask, sch, tables, cfg_scale=uncond_scale, context_see_xt=True,
                            parameterization=parameterization)
        if trace is not None:
            trace.append(x.clone())
    return x


# ----------------------------------------------------------------------------------------------
# SD3 VAE (diffusers key names; arithmetic per the in-repo mirror sd3/sd3_impls.py:215-474)
# ----------------------------------------------------------------------------------------------

def _gn(vsd: SD, name: str, x: torch.Tensor) -> torch.Tensor:
    return F.group_norm(x, 32, vsd[name + ".weight"], vsd[name + ".bias"], 1e-6)


def _conv(vsd: SD, name: str, x: torch.Tensor, stride=1, padding=1) -> torch.Tensor:
    return F.conv2d(x, vsd[name + ".weight"], vsd[name + ".bias"], stride=stride, padding=padding)


def _resnet(vsd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """ResnetBlock.forward (sd3_impls.py:244-254)"""
    h = _conv(vsd, p + ".conv1", F.silu(_gn(vsd, p + ".norm1", x)))
    h = _conv(vsd, p + ".conv2", F.silu(_gn(vsd, p + ".norm2", h)))
    if (p + ".conv_shortcut.weight") in vsd:
        x = _conv(vsd, p + ".conv_shortcut", x, padding=0)
    return x + h


VAE_ATTN_PROJ = "conv"     # "linear": diffusers' own formulation of the attention projections (F.linear) -- kept as the second CPU
#                             implementation whose spread against the reference the GPU's deviation is judged by (gen_golden pipeline16)


def _vae_attn(vsd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """AttnBlock.forward: single head over h*w tokens (sd3_impls.py:274-284).  diffusers stores q/k/v/out as Linear weights [C, C]; the
    reference's mirror applies them as 1x1 convolutions (:257-271) -- on the CPU in bf16 a fused-bias 1x1 convolution and `F.linear`
    round differently (27 % of the projected elements differ by an ulp), so the projections are applied as 1x1 convolutions here,
    which makes this block, and with it the whole VAE restatement, bit-identical to the mirror (tests/golden/PINNING.json: vae)."""
    B, C, H, W = x.shape
    if VAE_ATTN_PROJ == "linear":
        h = _gn(vsd, p + ".group_norm", x).reshape(B, C, H * W).transpose(1, 2)
        q = F.linear(h, vsd[p + ".to_q.weight"], vsd[p + ".to_q.bias"])
        k = F.linear(h, vsd[p + ".to_k.weight"], vsd[p + ".to_k.bias"])
        v = F.linear(h, vsd[p + ".to_v.weight"], vsd[p + ".to_v.bias"])
        a = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
        a = F.linear(a, vsd[p + ".to_out.0.weight"], vsd[p + ".to_out.0.bias"])
        return x + a.transpose(1, 2).reshape(B, C, H, W)
    h = _gn(vsd, p + ".group_norm", x)

    def proj(name, t):
        return F.conv2d(t, vsd[p + name + ".weight"].reshape(C, C, 1, 1), vsd[p + name + ".bias"])
    q, k, v = (proj(n, h).reshape(B, C, H * W).transpose(1, 2).contiguous()[:, None] for n in (".to_q", ".to_k", ".to_v"))
    a = F.scaled_dot_product_attention(q, k, v)[:, 0].transpose(1, 2).reshape(B, C, H, W)
    return x + proj(".to_out.0", a)


def vae_encode_mean(vsd: SD, img: torch.Tensor) -> torch.Tensor:
    """`vae.encode(images)[0].mode()` (SelftokPipeline.py:215) = first 16 of the 32 moment channels
    (VAEEncoder.forward, sd3_impls.py:359-377)"""
    h = _conv(vsd, "encoder.conv_in", img)
    for lvl in range(4):
        for j in range(2):
            h = _resnet(vsd, f"encoder.down_blocks.{lvl}.resnets.{j}", h)
        if lvl != 3:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            h = _conv(vsd, f"encoder.down_blocks.{lvl}.downsamplers.0.conv", h, stride=2, padding=0)
    h = _resnet(vsd, "encoder.mid_block.resnets.0", h)
    h = _vae_attn(vsd, "encoder.mid_block.attentions.0", h)
    h = _resnet(vsd, "encoder.mid_block.resnets.1", h)
    h = _conv(vsd, "encoder.conv_out", F.silu(_gn(vsd, "encoder.conv_norm_out", h)))
    return h[:, :16]


def vae_decode(vsd: SD, z: torch.Tensor) -> torch.Tensor:
    """`vae.decode(z)[0]` (SelftokPipeline.py:288) (VAEDecoder.forward, sd3_impls.py:427-444)"""
    h = _conv(vsd, "decoder.conv_in", z)
    h = _resnet(vsd, "decoder.mid_block.resnets.0", h)
    h = _vae_attn(vsd, "decoder.mid_block.attentions.0", h)
    h = _resnet(vsd, "decoder.mid_block.resnets.1", h)
    for lvl in range(4):
        for j in range(3):
            h = _resnet(vsd, f"decoder.up_blocks.{lvl}.resnets.{j}", h)
        if lvl != 3:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(vsd, f"decoder.up_blocks.{lvl}.upsamplers.0.conv", h)
    return _conv(vsd, "decoder.conv_out", F.silu(_gn(vsd, "decoder.conv_norm_out", h)))


SD3_SCALE, SD3_SHIFT = 1.5305, 0.0609


def process_in(z):   # SD3LatentFormat.process_in (sd3_impls.py:140-141); runs in the VAE dtype (bf16)
    return (z - SD3_SHIFT) * SD3_SCALE


def process_out(z):  # SD3LatentFormat.process_out (sd3_impls.py:143-144); runs in fp32
    return (z / SD3_SCALE) + SD3_SHIFT


def norm_ip(img: torch.Tensor) -> torch.Tensor:
    """norm_ip(recons, -1, 1) (SelftokPipeline.py:135-137): in-place clamp, then (x+1)/2"""
    img = img.clone()
    img.clamp_(min=-1, max=1)
    img.sub_(-1).div_(max(1 - (-1), 1e-5))
    return img


# ----------------------------------------------------------------------------------------------
# pipeline-level restatement (SelftokPipeline.py:210-322)
# ----------------------------------------------------------------------------------------------

def pipeline_encode(sd: SD, vsd: SD, images: torch.Tensor, tables=None) -> torch.Tensor:
    x0 = vae_encode_mean(vsd, images.to(torch.bfloat16))
    x0 = process_in(x0).to(torch.float32)
    return encode_latents(sd, x0, tables)


def pipeline_decode(sd: SD, vsd: SD, ids, noise, stages, k_per_stage, num_steps=50, tables=None, max_steps=None):
    lat = decode_latent(sd, torch.as_tensor(ids), noise, stages, k_per_stage, num_steps, tables, max_steps=max_steps)
    rec = vae_decode(vsd, process_out(lat).to(torch.bfloat16))
    return norm_ip(rec), lat


def pipeline_decode_renderer(sd: SD, vsd: SD, ids, tables=None):
    lat = renderer_forward(sd, codes_from_ids(sd, torch.as_tensor(ids)), tables)
    rec = vae_decode(vsd, process_out(lat).to(torch.bfloat16))
    return norm_ip(rec), lat
