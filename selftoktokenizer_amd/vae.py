"""SD3 VAE (diffusers AutoencoderKL layout, bf16) on MI355X.

The reference calls the third-party `diffusers.AutoencoderKL` (SelftokPipeline.py:162-163, 215, 288, 316);
this module exposes the same surface (`encode(x)[0].mode()`, `decode(z)[0]`) over the same checkpoint keys
(`<sd3_path>/vae/diffusion_pytorch_model.safetensors`) without diffusers.  Topology follows the in-repo
architectural mirror mimogpt/models/selftok/sd3/sd3_impls.py:215-474.  Convolutions and the single-head mid
attention run through PyTorch-ROCm (MIOpen / SDPA); GroupNorm+SiLU is our fused HIP epilogue.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from . import ops


class _Posterior:
    def __init__(self, moments):
        self.moments = moments

    def mode(self):
        return self.moments[:, : self.moments.shape[1] // 2]


class AutoencoderKLGPU:
    def __init__(self, vsd: Dict[str, torch.Tensor], device, dtype=torch.bfloat16):
        assert dtype == torch.bfloat16, "the HIP GroupNorm+SiLU epilogue is bf16 (the reference runs the VAE in bf16)"
        self.device, self.dtype = device, dtype
        self.w = {k: v.to(device=device, dtype=dtype).contiguous() for k, v in vsd.items()}

    # diffusers-ish plumbing so the pipeline code reads like the reference
    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def _gn_silu(self, name, x, act=True):
        return ops.groupnorm_silu(x.contiguous(), self.w[name + ".weight"], self.w[name + ".bias"], 32, 1e-6, act)

    def _conv(self, name, x, stride=1, padding=1):
        return F.conv2d(x, self.w[name + ".weight"], self.w[name + ".bias"], stride=stride, padding=padding)

    def _res(self, p, x):
        h = self._conv(p + ".conv1", self._gn_silu(p + ".norm1", x))
        h = self._conv(p + ".conv2", self._gn_silu(p + ".norm2", h))
        if (p + ".conv_shortcut.weight") in self.w:
            x = self._conv(p + ".conv_shortcut", x, padding=0)
        return x + h

    def _attn(self, p, x):
        B, C, H, W = x.shape
        h = self._gn_silu(p + ".group_norm", x, act=False).reshape(B, C, H * W).transpose(1, 2)
        q = F.linear(h, self.w[p + ".to_q.weight"], self.w[p + ".to_q.bias"])
        k = F.linear(h, self.w[p + ".to_k.weight"], self.w[p + ".to_k.bias"])
        v = F.linear(h, self.w[p + ".to_v.weight"], self.w[p + ".to_v.bias"])
        a = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
        a = F.linear(a, self.w[p + ".to_out.0.weight"], self.w[p + ".to_out.0.bias"])
        return x + a.transpose(1, 2).reshape(B, C, H, W)

    @torch.no_grad()
    def encode_moments(self, img: torch.Tensor) -> torch.Tensor:
        h = self._conv("encoder.conv_in", img.to(self.device, self.dtype))
        for lvl in range(4):
            for j in range(2):
                h = self._res(f"encoder.down_blocks.{lvl}.resnets.{j}", h)
            if lvl != 3:
                h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
                h = self._conv(f"encoder.down_blocks.{lvl}.downsamplers.0.conv", h, stride=2, padding=0)
        h = self._res("encoder.mid_block.resnets.0", h)
        h = self._attn("encoder.mid_block.attentions.0", h)
        h = self._res("encoder.mid_block.resnets.1", h)
        return self._conv("encoder.conv_out", self._gn_silu("encoder.conv_norm_out", h))

    def encode(self, img, return_dict=False):
        return (_Posterior(self.encode_moments(img)),)

    @torch.no_grad()
    def decode(self, z, return_dict=False):
        h = self._conv("decoder.conv_in", z.to(self.device, self.dtype))
        h = self._res("decoder.mid_block.resnets.0", h)
        h = self._attn("decoder.mid_block.attentions.0", h)
        h = self._res("decoder.mid_block.resnets.1", h)
        for lvl in range(4):
            for j in range(3):
                h = self._res(f"decoder.up_blocks.{lvl}.resnets.{j}", h)
            if lvl != 3:
                h = F.interpolate(h, scale_factor=2.0, mode="nearest")
                h = self._conv(f"decoder.up_blocks.{lvl}.upsamplers.0.conv", h)
        return (self._conv("decoder.conv_out", self._gn_silu("decoder.conv_norm_out", h)),)
