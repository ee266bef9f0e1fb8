"""State-dict contract of the Selftok hot path + synthetic / real checkpoint loading.

The drop-in pipeline consumes the *reference's* flat checkpoint layout
(`torch.load(ckpt)` -> {key: tensor}; reference: mimogpt/infer/SelftokPipeline.py:190-195,
key families listed in SURVEY.md section 3.1).  `expected_shapes()` is our own declaration of
that contract (pinned against the reference's `state_dict()` by tools/oracle/gen_golden.py ->
tests/golden/state_dict_keys.json).  `synthetic_state_dict()` fills any {key: shape} table
with hash-generated values (no weights are reachable offline).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch

from . import synth

Shape = Tuple[int, ...]

# architecture constants of the two shipped presets (reference model_zoo.py:22-60,177-180)
ENC_HIDDEN = 64      # latent-stream width of Enc-Qformer-Uni-XL/2
ENC_HEADS = 4
ENC_DEPTH = 16
ENC_QDIM = 512       # query-stream width
ENC_QHEADS = 8
DIT_DEPTH = 24
DIT_HIDDEN = 64 * DIT_DEPTH  # 1536
DIT_HEADS = DIT_DEPTH
CODE_DIM = 16
CODEBOOK = 32768
POS_MAX_DIT = 192
FREQ_DIM = 256


def _lin(d, name, out_f, in_f, bias=True):
    d[name + ".weight"] = (out_f, in_f)
    if bias:
        d[name + ".bias"] = (out_f,)


def encoder_shapes(K: int = 512, latent: int = 32, codebook: int = CODEBOOK) -> Dict[str, Shape]:
    """`encoder.*` keys: QformerEncoder 'dual' (reference models_ours.py:43-95,268-311)."""
    d: Dict[str, Shape] = {}
    pm = 2 * latent
    d["encoder.pos_embed"] = (1, pm * pm, ENC_HIDDEN)
    d["encoder.query_tokens"] = (1, K, ENC_QDIM)
    d["encoder.x_embedder.proj.weight"] = (ENC_HIDDEN, 16, 2, 2)
    d["encoder.x_embedder.proj.bias"] = (ENC_HIDDEN,)
    for i in range(ENC_DEPTH):
        p = f"encoder.blocks.{i}."
        _lin(d, p + "attn.qkv", 3 * ENC_HIDDEN, ENC_HIDDEN)
        _lin(d, p + "attn.to_query_kv", 2 * ENC_QDIM, ENC_HIDDEN)
        _lin(d, p + "attn.query_linear", 3 * ENC_QDIM, ENC_QDIM)
        _lin(d, p + "attn.proj", ENC_HIDDEN, ENC_HIDDEN)
        _lin(d, p + "attn.query_proj", ENC_QDIM, ENC_QDIM)
        _lin(d, p + "mlp.fc1", 4 * ENC_HIDDEN, ENC_HIDDEN)
        _lin(d, p + "mlp.fc2", ENC_HIDDEN, 4 * ENC_HIDDEN)
        _lin(d, p + "q_mlp.fc1", 4 * ENC_QDIM, ENC_QDIM)
        _lin(d, p + "q_mlp.fc2", ENC_QDIM, 4 * ENC_QDIM)
        _lin(d, p + "adaLN_modulation.1", 6 * ENC_QDIM, ENC_QDIM)
        _lin(d, p + "t_embedder.mlp.0", ENC_QDIM, FREQ_DIM)
        _lin(d, p + "t_embedder.mlp.2", ENC_QDIM, ENC_QDIM)
    for n, w in (("final_layer_norm", ENC_QDIM), ("final_layer_norm2", CODE_DIM), ("final_layer_norm3", CODE_DIM)):
        d[f"encoder.{n}.weight"] = (w,)
        d[f"encoder.{n}.bias"] = (w,)
    q = "encoder.quantizer."
    _lin(d, q + "project_in", CODE_DIM, ENC_QDIM)
    d[q + "_codebook.initted"] = (1,)
    d[q + "_codebook.cluster_size"] = (1, codebook)
    d[q + "_codebook.cluster_size_wo_react"] = (1, codebook)
    d[q + "_codebook.embed_avg"] = (1, codebook, CODE_DIM)
    d[q + "_codebook.timestep_p_over_c"] = (1, K, codebook)
    d[q + "_codebook.tpc_initted"] = (1,)
    d[q + "_codebook.embed"] = (1, codebook, CODE_DIM)
    d[q + "continuous"] = (1,)
    d[q + "steps"] = (1,)
    d[q + "count"] = (1, codebook)
    return d


def dit_shapes(K: int = 512, renderer: bool = False, latent: int = 32) -> Dict[str, Shape]:
    """`model.*` keys: MMDiT_XL / MMDiT_XL_Renderer (reference mmdit.py:648-825,1166-1340)."""
    d: Dict[str, Shape] = {}
    H = DIT_HIDDEN
    if renderer:
        g = latent // 2
        d["model.positional_embedding"] = (g * g, H)
        d["model.mask_token"] = (1, 1, H)
    else:
        d["model.x_embedder.proj.weight"] = (H, 16, 2, 2)
        d["model.x_embedder.proj.bias"] = (H,)
    d["model.pos_embed"] = (1, POS_MAX_DIT * POS_MAX_DIT, H)
    d["model.context_pos_embed"] = (1, K, H)
    _lin(d, "model.t_embedder.mlp.0", H, FREQ_DIM)
    _lin(d, "model.t_embedder.mlp.2", H, H)
    _lin(d, "model.y_embedder.mlp.0", H, CODE_DIM)
    _lin(d, "model.y_embedder.mlp.2", H, H)
    _lin(d, "model.context_embedder", H, CODE_DIM)
    for i in range(DIT_DEPTH):
        last = i == DIT_DEPTH - 1
        for stream in ("context_block", "x_block"):
            p = f"model.joint_blocks.{i}.{stream}."
            pre_only = last and stream == "context_block"
            _lin(d, p + "attn.qkv", 3 * H, H)
            if not pre_only:
                _lin(d, p + "attn.proj", H, H)
                _lin(d, p + "mlp.fc1", 4 * H, H)
                _lin(d, p + "mlp.fc2", H, 4 * H)
            _lin(d, p + "adaLN_modulation.1", (2 if pre_only else 6) * H, H)
            if stream == "context_block":
                _lin(d, p + "t_embedder.mlp.0", H, FREQ_DIM)
                _lin(d, p + "t_embedder.mlp.2", H, H)
    _lin(d, "model.final_layer.linear", 2 * 2 * 16, H)
    _lin(d, "model.final_layer.adaLN_modulation.1", 2 * H, H)
    return d


def expected_shapes(K: int = 512, renderer: bool = False, latent: int = 32) -> Dict[str, Shape]:
    d = encoder_shapes(K, latent)
    d.update(dit_shapes(K, renderer, latent))
    return d


# ----------------------------------------------------------------------------
# sin-cos tables (float64 numpy then cast: reference models.py:305-352, mmdit.py:91-135)
# ----------------------------------------------------------------------------

def sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d(embed_dim: int, grid_size: int) -> np.ndarray:
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid_size, grid_size)
    eh = sincos_1d(embed_dim // 2, grid[0])
    ew = sincos_1d(embed_dim // 2, grid[1])
    return np.concatenate([eh, ew], axis=1)


def context_pos_embed(K: int, dim: int = DIT_HIDDEN) -> torch.Tensor:
    """sin-cos over positions 1000+8k (reference mmdit.py:812-822, diti_utils.py:109)."""
    pos = 1000 + np.arange(K, dtype=np.float32) * 8
    return torch.from_numpy(sincos_1d(dim, pos)).float().unsqueeze(0)


# ----------------------------------------------------------------------------
# synthetic state dict
# ----------------------------------------------------------------------------

def _synth_tensor(name: str, shape: Shape, device) -> torch.Tensor:
    seed = synth.name_seed(name)
    leaf = name.rsplit(".", 1)[-1]
    if name.startswith("diffusion."):
        return None
    if name == "encoder.pos_embed":
        g = int(round(math.sqrt(shape[1])))
        return torch.from_numpy(sincos_2d(shape[2], g)).float().unsqueeze(0).to(device)
    if name == "model.context_pos_embed":
        return context_pos_embed(shape[1], shape[2]).to(device)
    if name == "model.pos_embed":
        return synth.hash_uniform(seed, shape, -0.5, 0.5, device)
    if name == "encoder.query_tokens":
        return synth.hash_uniform(seed, shape, -1.0, 1.0, device)
    if name in ("model.positional_embedding", "model.mask_token"):
        return synth.hash_uniform(seed, shape, -0.5, 0.5, device)
    if name.endswith("_codebook.embed") or name.endswith("_codebook.embed_avg"):
        raw = synth.hash_normalish(synth.name_seed("codebook"), shape, "cpu")
        return torch.nn.functional.normalize(raw, p=2, dim=-1).to(device)
    if leaf in ("initted", "tpc_initted"):
        return torch.ones(shape, device=device)
    if leaf in ("cluster_size", "cluster_size_wo_react", "count", "continuous", "steps"):
        return torch.zeros(shape, device=device)
    if leaf == "timestep_p_over_c":
        return torch.full(shape, 1.0 / shape[-1], device=device)
    if leaf == "weight" and len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        a = math.sqrt(3.0 / fan_in)
        return synth.hash_uniform(seed, shape, -a, a, device)
    if leaf == "weight":  # norm scale
        return synth.hash_uniform(seed, shape, 0.9, 1.1, device)
    if leaf == "bias":
        return synth.hash_uniform(seed, shape, -0.1, 0.1, device)
    raise KeyError(f"no synthetic rule for {name} {shape}")


def synthetic_state_dict(shapes: Dict[str, Shape], device="cpu") -> Dict[str, torch.Tensor]:
    out = {}
    for name, shape in shapes.items():
        t = _synth_tensor(name, tuple(shape), device)
        if t is not None:
            out[name] = t
    return out


# ----------------------------------------------------------------------------
# SD3 VAE (diffusers AutoencoderKL key names; reference SelftokPipeline.py:162-163)
# ----------------------------------------------------------------------------

def _res(d, p, cin, cout):
    d[p + ".norm1.weight"] = (cin,); d[p + ".norm1.bias"] = (cin,)
    d[p + ".conv1.weight"] = (cout, cin, 3, 3); d[p + ".conv1.bias"] = (cout,)
    d[p + ".norm2.weight"] = (cout,); d[p + ".norm2.bias"] = (cout,)
    d[p + ".conv2.weight"] = (cout, cout, 3, 3); d[p + ".conv2.bias"] = (cout,)
    if cin != cout:
        d[p + ".conv_shortcut.weight"] = (cout, cin, 1, 1); d[p + ".conv_shortcut.bias"] = (cout,)


def _mid(d, p, c):
    _res(d, p + ".resnets.0", c, c)
    a = p + ".attentions.0"
    d[a + ".group_norm.weight"] = (c,); d[a + ".group_norm.bias"] = (c,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        d[f"{a}.{n}.weight"] = (c, c); d[f"{a}.{n}.bias"] = (c,)
    _res(d, p + ".resnets.1", c, c)


def vae_shapes() -> Dict[str, Shape]:
    """stabilityai SD3 VAE: block_out_channels (128,256,512,512), 2 layers/block, latent 16, no quant convs."""
    d: Dict[str, Shape] = {}
    ch = (128, 256, 512, 512)
    d["encoder.conv_in.weight"] = (128, 3, 3, 3); d["encoder.conv_in.bias"] = (128,)
    cin = 128
    for L, cout in enumerate(ch):
        for j in range(2):
            _res(d, f"encoder.down_blocks.{L}.resnets.{j}", cin, cout)
            cin = cout
        if L != 3:
            d[f"encoder.down_blocks.{L}.downsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            d[f"encoder.down_blocks.{L}.downsamplers.0.conv.bias"] = (cout,)
    _mid(d, "encoder.mid_block", 512)
    d["encoder.conv_norm_out.weight"] = (512,); d["encoder.conv_norm_out.bias"] = (512,)
    d["encoder.conv_out.weight"] = (32, 512, 3, 3); d["encoder.conv_out.bias"] = (32,)
    d["decoder.conv_in.weight"] = (512, 16, 3, 3); d["decoder.conv_in.bias"] = (512,)
    _mid(d, "decoder.mid_block", 512)
    cin = 512
    for L, cout in enumerate((512, 512, 256, 128)):
        for j in range(3):
            _res(d, f"decoder.up_blocks.{L}.resnets.{j}", cin, cout)
            cin = cout
        if L != 3:
            d[f"decoder.up_blocks.{L}.upsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            d[f"decoder.up_blocks.{L}.upsamplers.0.conv.bias"] = (cout,)
    d["decoder.conv_norm_out.weight"] = (128,); d["decoder.conv_norm_out.bias"] = (128,)
    d["decoder.conv_out.weight"] = (3, 128, 3, 3); d["decoder.conv_out.bias"] = (3,)
    return d


def synthetic_vae_state_dict(device="cpu", dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    sd = {}
    for name, shape in vae_shapes().items():
        sd[name] = _synth_tensor("vae." + name, shape, device).to(dtype)
    return sd


def diffusers_to_ldm_key(key: str) -> str:
    """diffusers AutoencoderKL key -> the ldm/SDVAE-mirror key (reference sd3_impls.py:215-474 module names;
    mapping hints at SelftokPipeline.py:48-53).  Attention q/k/v/out are 1x1 convs there ([C,C,1,1])."""
    import re
    k = key
    k = re.sub(r"^encoder\.down_blocks\.(\d)\.resnets\.(\d)\.", r"encoder.down.\1.block.\2.", k)
    k = re.sub(r"^encoder\.down_blocks\.(\d)\.downsamplers\.0\.conv\.", r"encoder.down.\1.downsample.conv.", k)
    m = re.match(r"^decoder\.up_blocks\.(\d)\.(resnets\.(\d)|upsamplers\.0\.conv)\.(.*)$", k)
    if m:
        lvl = 3 - int(m.group(1))
        if m.group(3) is not None:
            k = f"decoder.up.{lvl}.block.{m.group(3)}.{m.group(4)}"
        else:
            k = f"decoder.up.{lvl}.upsample.conv.{m.group(4)}"
    k = k.replace("mid_block.resnets.0.", "mid.block_1.").replace("mid_block.resnets.1.", "mid.block_2.")
    k = k.replace("mid_block.attentions.0.group_norm.", "mid.attn_1.norm.")
    k = k.replace("mid_block.attentions.0.to_q.", "mid.attn_1.q.").replace("mid_block.attentions.0.to_k.", "mid.attn_1.k.")
    k = k.replace("mid_block.attentions.0.to_v.", "mid.attn_1.v.").replace("mid_block.attentions.0.to_out.0.", "mid.attn_1.proj_out.")
    k = k.replace("conv_norm_out.", "norm_out.").replace("conv_shortcut.", "nin_shortcut.")
    return k


# ----------------------------------------------------------------------------
# real checkpoints
# ----------------------------------------------------------------------------

def load_state(model_shapes: Dict[str, Shape], state_dict: Dict[str, torch.Tensor], prefix: str = "", init_method=None):
    """The reference's `load_state(model, state_dict, prefix, init_method)` filter (SelftokPipeline.py:46-83) over a shape table
    instead of an nn.Module: strip `prefix`, keep only keys the model has, drop keys whose shape differs, and -- for the SD3
    pretrain prefix 'model.diffusion_model.' -- the reference's exclusion lists (context_embedder, every context_block, and by
    init_method the final layer / the x_block attention).  Returns (kept, missing, unexpected, shape_mismatched):
    `missing` = model keys without a tensor, `unexpected` = always [] (foreign keys are filtered out before loading, as the
    reference does), `shape_mismatched` = keys dropped for their shape."""
    def strip(k):
        return k.replace(prefix, "") if prefix else k
    if prefix == "model.diffusion_model.":
        excluded = ["context_embedder.bias", "context_embedder.weight"]
        if init_method == 1:
            excluded += ["final_layer.adaLN_modulation.1.bias", "final_layer.adaLN_modulation.1.weight",
                         "final_layer.linear.bias", "final_layer.linear.weight"]
        kept = {strip(k): v for k, v in state_dict.items()
                if strip(k) in model_shapes and strip(k) not in excluded and "context_block" not in k
                and not (init_method == 2 and "x_block.attn" in k)}
    else:
        kept = {strip(k): v for k, v in state_dict.items() if strip(k) in model_shapes}
    bad = [k for k, v in kept.items() if tuple(v.shape) != tuple(model_shapes[k])]
    for k in bad:
        kept.pop(k)
    missing = [k for k in model_shapes if k not in kept]
    return kept, missing, [], bad


def check_tokenizer_state_dict(sd: Dict[str, torch.Tensor], K: int, renderer: bool = False, ema: bool = False) -> None:
    """What `ImageTokenizer.load_state_dict(state_dict, strict=False)` (SelftokPipeline.py:195) accepts, made explicit for the
    keys the encode/decode path reads.  torch raises on a size mismatch even with strict=False -> RuntimeError here too.  A MISSING
    key is silently tolerated by the reference, which then runs on randomly initialised parameters; this build refuses instead
    (RuntimeError naming the keys).  Unexpected keys (optimizer state, diffusion.* buffers, ...) are ignored like the reference.
    `ema`: also require the strict `ema.load_state_dict(state_dict['ema_state_dict'])` contract of :193-194 -- exactly the
    MMDiT keys without the 'model.' prefix, missing or unexpected keys are errors."""
    shapes = expected_shapes(K, renderer=renderer)
    need = {k: v for k, v in shapes.items() if not k.startswith("encoder.quantizer.") or k.endswith("_codebook.embed")
            or ".project_in." in k}
    mism = [f"{k}: checkpoint {tuple(sd[k].shape)} vs model {tuple(v)}" for k, v in need.items() if k in sd and tuple(sd[k].shape) != tuple(v)]
    if mism:
        raise RuntimeError("size mismatch for " + "; ".join(mism[:8]) + (" ..." if len(mism) > 8 else ""))
    # parameters of the reference model that the encode / decode path never reads (the class-label embedder of the MMDiT, two
    # LayerNorms of the encoder that only the training forward uses): an inference-only / pruned checkpoint without them loads in
    # the reference (strict=False) and must load here too
    unused = ("model.y_embedder.", "encoder.final_layer_norm.", "encoder.final_layer_norm2.")
    missing = [k for k in need if k not in sd and not (ema and k.startswith("model.")) and not k.startswith(unused)]
    if missing:
        raise RuntimeError(f"{len(missing)} tokenizer parameters are missing from the checkpoint (the reference would silently keep random "
                           f"initial values, strict=False): {missing[:6]}{' ...' if len(missing) > 6 else ''}")
    if ema:
        if "ema_state_dict" not in sd:
            raise KeyError("ema_state_dict")                       # what state_dict['ema_state_dict'] raises in the reference
        dit = {k[len("model."):]: v for k, v in shapes.items() if k.startswith("model.")}
        e = sd["ema_state_dict"]
        miss = [k for k in dit if k not in e]
        unexp = [k for k in e if k not in dit]
        bad = [k for k in dit if k in e and tuple(e[k].shape) != tuple(dit[k])]
        if miss or unexp or bad:
            raise RuntimeError(f"Error(s) in loading ema_state_dict (strict): missing {miss[:4]}, unexpected {unexp[:4]}, size mismatch {bad[:4]}")


def load_tokenizer_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    """`torch.load(ckpt, map_location='cpu')` flat state dict (reference SelftokPipeline.py:190)."""
    sd = torch.load(path, map_location="cpu")
    if isinstance(sd, dict) and "state_dict" in sd and not any(k.startswith("encoder.") for k in sd):
        sd = sd["state_dict"]
    return sd


VAE_FILENAMES = ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors", "diffusion_pytorch_model.bin",
                 "diffusion_pytorch_model.fp16.bin")


def load_vae_checkpoint(sd3_path: str) -> Dict[str, torch.Tensor]:
    """<sd3_path>/vae/diffusion_pytorch_model.{safetensors,fp16.safetensors,bin} in diffusers layout (what
    `AutoencoderKL.from_pretrained(sd3_path, subfolder="vae")` reads, reference SelftokPipeline.py:162), or a single-file ldm
    checkpoint with `first_stage_model.` keys (reference set_sd3_vae + load_state, :46-83, :115-121).  The result is checked
    against the 244-entry SD3-VAE table: anything missing or mis-shaped is reported here, not as a KeyError deep inside the VAE."""
    import os
    shapes = vae_shapes()
    out = None
    if os.path.isdir(sd3_path):
        for name in VAE_FILENAMES:
            cand = os.path.join(sd3_path, "vae", name)
            if os.path.exists(cand):
                if name.endswith(".safetensors"):
                    from safetensors.torch import load_file
                    raw = load_file(cand)
                else:
                    raw = torch.load(cand, map_location="cpu")
                out = {k: v for k, v in raw.items() if k in shapes}
                break
        if out is None:
            raise FileNotFoundError(f"no VAE weights under {os.path.join(sd3_path, 'vae')}: looked for {', '.join(VAE_FILENAMES)}")
    else:
        raw = torch.load(sd3_path, map_location="cpu")
        ldm_shapes = {diffusers_to_ldm_key(k): k for k in shapes}
        out = {}
        for k, v in raw.items():
            k2 = k.replace("first_stage_model.", "")
            if k2 in ldm_shapes:
                tgt = ldm_shapes[k2]
                if v.numel() == int(torch.tensor(shapes[tgt]).prod()):          # attention q/k/v/out are 1x1 convs in the ldm layout
                    out[tgt] = v.reshape(shapes[tgt])
    return out


def check_vae_state_dict(vsd: Dict[str, torch.Tensor]) -> None:
    shapes = vae_shapes()
    missing = [k for k in shapes if k not in vsd]
    bad = [f"{k}: {tuple(vsd[k].shape)} vs {tuple(shapes[k])}" for k in shapes if k in vsd and tuple(vsd[k].shape) != tuple(shapes[k])]
    if missing or bad:
        raise RuntimeError(f"SD3 VAE checkpoint does not match the 244-tensor stabilityai layout: {len(missing)} missing "
                           f"{missing[:5]}{' ...' if len(missing) > 5 else ''}; size mismatch {bad[:5]}")
