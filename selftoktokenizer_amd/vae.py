"""SD3 VAE (diffusers AutoencoderKL layout, bf16) on MI355X.

The reference calls the third-party `diffusers.AutoencoderKL` (SelftokPipeline.py:162-163, 215, 288, 316);
this module exposes the same surface (`encode(x)[0].mode()`, `decode(z)[0]`) over the same checkpoint keys
(`<sd3_path>/vae/diffusion_pytorch_model.safetensors`) without diffusers.  Topology follows the in-repo
architectural mirror mimogpt/models/selftok/sd3/sd3_impls.py:215-474.

Default (`mode="parity"`): channels-last activations, every convolution through our implicit-GEMM kernel (csrc/conv.hip: fp32 accumulation
of bf16 products with the bias inside, ONE rounding -- the reference's CPU arithmetic -- with ResnetBlock's residual add, Upsample's
nearest 2x and Downsample's padding fused), GroupNorm+SiLU by our fp64-statistics kernel, the single-head mid attention (round 4) by the
exact-order kernels of csrc/vae_exact.hip -- no torch arithmetic is left in the VAE at 256 x 256 (`_attn_tokens`, the torch-op formulation
that rounds where the CPU flash kernel rounds, remains for other token counts and the 'miopen' / 'fast' modes).  No MIOpen, bit-stable by construction, 103 ms per 64 images (encode +
decode) against 331 ms for the MIOpen route below and 185 ms for MIOpen's fastest (inaccurate) solvers.

`mode="exact"` (round 4 encoder, round 5 decoder): every reduction in the ORDER of the reference's torch-CPU run -- oneDNN's AMX convolution
chunks, ATen's GroupNorm cascade, SiLU table, flash-attention row pass (csrc/vae_exact.hip; orders probed and restated in
oracle/vae_exact.c) -- so the ENCODER's latents, and with them the token ids from pixels, are the reference's BIT FOR BIT
(tests/test_vae_exact_gpu.py: 16 images of the reference pipeline's own run, 8192 / 8192 ids), and the DECODER's pixels equal the reference's
decode of the same latents (tests/golden/decode_b16.npz).  It runs on the fp32 matrix cores (a prescribed order cannot use a bf16 MFMA's
internal one): ~5x the `parity` kernels' time.
The two halves are independent (round 6: `mode` = the encoder's arithmetic, `decode_mode` = the decoder's, default the same): token ids need the
exact ENCODER; pixels meet the north star's 1e-3 dB with the `parity` DECODER too (the 50-step fp32 sampler upstream is not bit-reproducible
outside gemm='exact'), so the pipeline's default is exact encode + parity decode and gemm='exact' selects the exact decoder.
The exact decoder follows the reference's batch dependence: a 3x3 layer whose bf16 input or output reaches 2^31 bytes (64 images at 256 x 256)
switches to oneDNN's order 1 (`_x_conv`), so its pixels at 64 images per call differ from those at 48 -- as the reference's do; probed at 48 / 64
images per call, an extrapolation of that threshold above 64.

`mode="miopen"` keeps the route rounds 1-3 used to reach the same arithmetic through PyTorch-ROCm, `mode="fast"` the rounds 1-2 arithmetic.
Two properties of PyTorch-ROCm's bf16 convolution path matter for parity with the reference's CPU run and are handled by `mode="miopen"`
(measured: tools/probe_vae_modes.py, profiles/r3_vae_modes.txt; tests/test_parity16_gpu.py):

  * BIAS.  `F.conv2d(x, w, b)` on ROCm runs MIOpen's convolution (fp32 accumulate, result rounded to bf16) and then ADDS the bias as
    a second bf16 op -- two roundings per convolution, where the reference's CPU convolution (oneDNN) adds the bias to the fp32
    accumulator and rounds once.  That alone moves the VAE latents by rms 0.0148 (one bf16 ulp at |x| ~ 2) against the reference,
    4x the spread between two CPU implementations, and was the cause of two thirds of the end-to-end token flips.  Here the
    bias rides INSIDE the accumulation: every convolution input gets 8 extra channels (one of ones, seven of zeros: channel
    counts stay multiples of 8) and the weight gets the bias in the centre tap of the ones channel -- bias * 1.0 is exact in the fp32
    accumulator, the centre tap never reads padding, the result is rounded once.
  * SOLVER CHOICE.  MIOpen's default Find benchmarks every applicable solver the first time a shape is seen (60 s of GPU time per
    fresh process) and may pick solvers whose results are not bit-stable from run to run.  The convolutions here run under
    `torch.backends.cudnn.flags(deterministic=True)` (PyTorch then asks for MIOpen's GEMM algorithm: im2col + GEMM, fp32
    accumulation in a fixed order) with MIOPEN_FIND_MODE=FAST (no benchmarking): latents, ids and pixels are bit-identical between
    calls and between processes, first calls take 0.1 - 0.3 s, steady state is 1.7x the fastest solver's (the VAE is ~1 % of a
    50-step decode).
"""
from __future__ import annotations

import os
from typing import Dict

import torch
import torch.nn.functional as F

from . import ops
from .modsurface import ModuleSurface

ONES_PAD = 8          # extra input channels per convolution: [ones, 0, 0, 0, 0, 0, 0, 0]


class _Posterior:
    def __init__(self, moments):
        self.moments = moments

    def mode(self):
        return self.moments[:, : self.moments.shape[1] // 2]


def _bits_f32(u: int) -> float:
    return float(torch.tensor(u if u < (1 << 31) else u - (1 << 32), dtype=torch.int32).view(torch.float32))


_LOG2E, _LN_FLT_MIN, _LN_FLT_MAX = _bits_f32(0x3fb8aa3b), _bits_f32(0xc2aeac50), _bits_f32(0x42b17218)
_FEXP_C = [float(torch.tensor(c, dtype=torch.float32)) for c in (0.00010703434948458272, 0.30354260500649682, -0.22433836478672356, -0.079204240219773236)]


def fexp_u20(x: torch.Tensor) -> torch.Tensor:
    """exp(x) exactly as torch's CPU flash-attention kernel evaluates it for bf16 / fp16 inputs: `Vectorized<float>::fexp_u20()`
    (ATen/cpu/vec/vec512/vec512_float.h in the torch 2.10 wheel; Malossi et al., "Fast Exponential Computation on SIMD Architectures"):
    2^(x log2 e) with the fractional part corrected by a degree-3 polynomial and the result assembled in the exponent / mantissa bits
    by a float -> int truncation -- relative error up to 1.05e-4.  The reference's CPU run computes the un-normalised softmax
    probabilities of the VAE's attention block with it, and at bf16 output resolution that error decides 15 % of the block's output
    roundings: with the accurate exp the block agrees with the CPU in 83 % of its elements, with this one in 99.6 % (measured on
    the CPU itself).  fp32 multiply / subtract as the AVX-512 code does them; its FMAs are reproduced through fp64 (a 24 x 24-bit
    product is exact there, the sum is rounded to fp32 once)."""
    src = x * _LOG2E
    frac = src - torch.floor(src)
    fd = frac.double()
    res = (fd * _FEXP_C[3] + _FEXP_C[2]).float()
    res = (fd * res.double() + _FEXP_C[1]).float()
    res = (fd * res.double() + _FEXP_C[0]).float()
    src = src - res
    ci = (src.double() * float(2 ** 23) + float(2 ** 23) * 127.0).float().to(torch.int32)        # cvttps: truncation
    ci = torch.where(x < _LN_FLT_MIN, torch.zeros_like(ci), ci)
    ci = torch.where(x > _LN_FLT_MAX, torch.full_like(ci, 0x7F800000), ci)
    return ci.view(torch.float32)


class _Deterministic:
    """torch.backends.cudnn.deterministic (= MIOpen's GEMM algorithm on ROCm) for the duration of a VAE call, then restored"""

    def __init__(self, on: bool):
        self.on = on

    def __enter__(self):
        self.prev = torch.backends.cudnn.deterministic
        torch.backends.cudnn.deterministic = bool(self.on)

    def __exit__(self, *a):
        torch.backends.cudnn.deterministic = self.prev
        return False


class AutoencoderKLGPU(ModuleSurface):
    _sd_prefix = ""
    MODES = ("exact", "parity", "miopen", "fast")

    def __init__(self, vsd: Dict[str, torch.Tensor], device, dtype=torch.bfloat16, mode: str = "parity", decode_mode: str = None):
        """`mode`: arithmetic of `encode`; `decode_mode`: arithmetic of `decode` (default: the same; `set_decode_mode` switches it later).
        mode 'parity' (default since the end of round 3): channels-last, every convolution through the implicit-GEMM kernel of
        csrc/conv.hip (fp32 accumulation incl. the bias, one rounding; residual add, nearest upsample and Downsample's padding fused),
        GroupNorm by the fp64-statistics kernel: the reference's CPU arithmetic, bit-stable by construction, no MIOpen.  mode 'miopen':
        the same arithmetic as rounds 1-3 reached it -- bias inside the accumulation + MIOpen's GEMM algorithm (module docstring).  mode 'fast': the
        rounds 1-2 arithmetic -- `F.conv2d(x, w, b)` with whatever solver MIOpen's (non-benchmarking) search picks: 1.8x faster
        convolutions, but 40 instead of 11 flipped tokens per 8192 and a 6e-3 instead of 3e-4 dB PSNR delta against the reference, and
        latents that are not bit-stable from call to call.  For throughput runs that do not compare against the reference."""
        assert dtype == torch.bfloat16, "the HIP GroupNorm+SiLU epilogue is bf16 (the reference runs the VAE in bf16)"
        decode_mode = decode_mode or mode
        for m in (mode, decode_mode):
            if m not in self.MODES:
                raise ValueError(f"VAE mode {m!r}: expected one of {self.MODES}")
        if (mode in ("miopen", "fast")) != (decode_mode in ("miopen", "fast")) or (mode in ("miopen", "fast") and mode != decode_mode):
            raise ValueError(f"VAE modes {mode!r} (encode) / {decode_mode!r} (decode): the MIOpen routes ('miopen', 'fast') cover both halves or neither")
        self.mode = mode                      # the ENCODER's arithmetic (token ids depend on it)
        self.decode_mode = decode_mode        # the DECODER's
        self.device, self.dtype = device, dtype
        # no benchmarking Find (see the module docstring); a value the caller exported wins.  MIOpen reads it when it first searches.
        os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
        self.deterministic = mode == "miopen"
        self.w = {k: v.to(device=device, dtype=dtype).contiguous() for k, v in vsd.items()}
        # convolution weights with the bias folded in: [O, C + 8, kh, kw], bias in the centre tap of channel C
        self.wb = {}
        for k, v in self.w.items():
            if mode == "miopen" and k.endswith(".weight") and v.dim() == 4:
                O, C, kh, kw = v.shape
                wb = torch.zeros(O, C + ONES_PAD, kh, kw, device=device, dtype=dtype)
                wb[:, :C] = v
                wb[:, C, kh // 2, kw // 2] = self.w[k[:-len("weight")] + "bias"]
                self.wb[k[:-len(".weight")]] = wb.contiguous()
        self._tails = {}
        # native path: packed weight images of csrc/conv.hip (built on first use of mode 'parity')
        self.pc = {}
        self.xw = {}
        native = mode in ("exact", "parity")
        if native:
            # the checkpoint's tensors, permuted to [Cout, k, k, Cin] (no packing): the exact-order kernels read 64 contiguous bytes per
            # (output channel, tap, 32-channel block).  diffusers stores the attention projections as Linear [O, I] = a 1x1 convolution.
            # 'exact': the whole encoder; both modes: the mid-block attention of encoder AND decoder (AttnBlock, sd3_impls.py:274-284)
            # runs on these kernels -- GroupNorm, q / k / v / out projections and the flash-kernel row pass in the reference's order
            for k, v in self.w.items():
                if k.endswith(".weight") and v.dim() in (2, 4) and (".attentions." in k or (mode == "exact" and not k.startswith("decoder."))):
                    v4 = v.reshape(v.shape[0], v.shape[1], 1, 1) if v.dim() == 2 else v
                    self.xw[k[:-len(".weight")]] = v4.permute(0, 2, 3, 1).contiguous()
            self.silu_table = ops.vx_silu_table(device)
            if decode_mode == "exact":
                self._prepare_exact_decoder()
        if native:
            for k, v in self.w.items():
                if k.endswith(".weight") and v.dim() == 4:
                    name = k[:-len(".weight")]
                    self.pc[name] = ops.PackedConv(v, self.w[name + ".bias"])

    def _prepare_exact_decoder(self):
        """the decoder's weights in the exact-order kernels' layout ([Cout, k, k, Cin]), once"""
        if getattr(self, "xb_out", None) is not None:
            return
        for k, v in self.w.items():
            if k.startswith("decoder.") and k.endswith(".weight") and v.dim() in (2, 4) and k[:-len(".weight")] not in self.xw:
                v4 = v.reshape(v.shape[0], v.shape[1], 1, 1) if v.dim() == 2 else v
                self.xw[k[:-len(".weight")]] = v4.permute(0, 2, 3, 1).contiguous()
        # the decoder's two odd layers on the 32-channel-chunk kernel: conv_in (16 input channels = one 16-channel chunk per tap in
        # oneDNN) with zero input channels 16..31, conv_out (3 output channels) with 29 zero output channels -- a zero product
        # leaves an fp32 chain's bits unchanged, so the padded layers compute the unpadded layers' bits
        wi = self.xw["decoder.conv_in"]                                               # [512,3,3,16]
        self.xw["decoder.conv_in"] = F.pad(wi, (0, 32 - wi.shape[3])).contiguous()
        wo = self.xw["decoder.conv_out"]                                              # [3,3,3,128]
        self.xw["decoder.conv_out"] = F.pad(wo, (0, 0, 0, 0, 0, 0, 0, 32 - wo.shape[0])).contiguous()
        self.xb_out = F.pad(self.w["decoder.conv_out.bias"], (0, 32 - wo.shape[0])).contiguous()

    def set_decode_mode(self, decode_mode: str) -> str:
        """switch the decoder between 'parity' and 'exact' (native modes only); weights of the new mode are prepared on first use"""
        if decode_mode not in ("parity", "exact") or self.mode not in ("parity", "exact"):
            raise ValueError(f"set_decode_mode({decode_mode!r}) with encoder mode {self.mode!r}: only the native modes 'parity' / 'exact' can be mixed")
        if decode_mode == "exact":
            self._prepare_exact_decoder()
        self.decode_mode = decode_mode
        return decode_mode

    # ---- native channels-last path (mode 'parity') ----------------------------------------------------------------------------
    def _n_gn(self, name, x, act=True):
        """GroupNorm [+ SiLU] of the channels-last path.  Where the exact-order kernels apply (C % 128 == 0, H*W a multiple of 1024: every
        layer of the VAE at 256 x 256) they are used in BOTH modes -- ATen's own statistics, `bf16(fma(scale, x, bias))` and torch's SiLU
        by table cost the same 1 ms on the largest layer as the fp64-statistics kernel with its expf + division, and remove one source of
        deviation from the reference (the decoder's remaining one is the summation order inside its bf16-MFMA convolutions)."""
        B, C = x.shape[0], x.shape[-1]
        hw = x.numel() // max(B * C, 1)
        rp = 4096 if hw >= 4096 else 1024
        if self.xw and B > 0 and C % 128 == 0 and hw % rp == 0 and ((hw // rp) & (hw // rp - 1)) == 0:
            return self._x_gn(name, x, act)
        return ops.groupnorm_silu_nhwc(x, self.w[name + ".weight"], self.w[name + ".bias"], 32, 1e-6, act)

    def _n_res(self, p, x):
        h = ops.conv2d_nhwc(self._n_gn(p + ".norm1", x), self.pc[p + ".conv1"])
        sc = ops.conv2d_nhwc(x, self.pc[p + ".conv_shortcut"]) if (p + ".conv_shortcut") in self.pc else x
        return ops.conv2d_nhwc(self._n_gn(p + ".norm2", h), self.pc[p + ".conv2"], residual=sc)          # x + h: second rounding in the epilogue

    def _n_attn(self, p, x):
        """AttnBlock on channels-last tokens, every step a HIP kernel in the reference's CPU order (csrc/vae_exact.hip): GroupNorm with
        ATen's statistics, the four projections as 1x1 convolutions in oneDNN's chunk order, the attention as ATen's flash kernel
        evaluates it (T = 1024 tokens: the VAE at 256 x 256).  Other token counts keep the torch-op formulation of `_attn_tokens`."""
        B, H, W, C = x.shape
        exact_here = (self.decode_mode if p.startswith("decoder.") else self.mode) == "exact"
        if not self.xw or (H * W != 1024 and not (exact_here and (H * W) % 32 == 0)):
            h = self._n_gn(p + ".group_norm", x, act=False).reshape(B, H * W, C)
            return x + self._attn_tokens(p, h).reshape(B, H, W, C)
        n = self._x_gn(p + ".group_norm", x, act=False)
        q, k, v = (self._x_conv(p + s, n).reshape(B, H * W, C) for s in (".to_q", ".to_k", ".to_v"))
        a = ops.vx_attention(q, k, v).reshape(B, H, W, C)
        return self._x_conv(p + ".to_out.0", a, residual=x)

    def _n_encode_moments(self, img):
        x = img.to(self.device, self.dtype).permute(0, 2, 3, 1)
        h = F.pad(x, (0, 8 - x.shape[-1])).contiguous()                                # 3 -> 8 channels: 16-byte pixels
        h = ops.conv2d_nhwc(h, self.pc["encoder.conv_in"])
        for lvl in range(4):
            for j in range(2):
                h = self._n_res(f"encoder.down_blocks.{lvl}.resnets.{j}", h)
            if lvl != 3:
                h = ops.conv2d_nhwc(h, self.pc[f"encoder.down_blocks.{lvl}.downsamplers.0.conv"], stride=2)
        h = self._n_res("encoder.mid_block.resnets.0", h)
        h = self._n_attn("encoder.mid_block.attentions.0", h)
        h = self._n_res("encoder.mid_block.resnets.1", h)
        h = ops.conv2d_nhwc(self._n_gn("encoder.conv_norm_out", h), self.pc["encoder.conv_out"])
        return h.permute(0, 3, 1, 2).contiguous()

    # ---- exact-order encoder (mode 'exact'): every reduction in the reference's torch-CPU order -------------------------------------
    EXACT_SIZES = (128, 256, 320)          # square image sizes whose every layer shape was probed against oneDNN (tools/probe_cpu_bf16/) and has goldens

    @staticmethod
    def _x_order(cin: int, stride: int, width: int) -> int:
        """the chunk order oneDNN's AMX convolution kernel uses for a layer of the VAE (probed per layer shape at 128 / 256 / 320 px,
        tools/probe_cpu_bf16/, profiles/r5_cpu_bf16_resolution_orders.txt; include/selftok_hip.h): conv_in is one 27-element chunk; a stride-2
        (Downsample) layer whose INPUT is at least 102 pixels wide runs channel-block major with private partial sums -- whatever its height,
        batch and channel count (at 256 px: the 128- and 256-channel ones; at 128 px only the first; at 320 px the same two as at 256);
        everything else, and every layer of the decoder, (kh, kw, channel-block)"""
        if cin < 32:
            return 2
        return 3 if (stride == 2 and width >= 102) else 0

    @classmethod
    def _x_check_size(cls, h: int, w: int):
        if h != w or h not in cls.EXACT_SIZES:
            raise NotImplementedError(f"vae_mode='exact' reproduces oneDNN's summation orders as probed for the VAE's layer shapes at {cls.EXACT_SIZES} px square "
                                      f"images (got {h} x {w}); use vae_mode='parity' for other sizes")

    def _x_conv(self, name, x, stride=1, residual=None, upsample=False, bias=None):
        w = self.xw[name]
        order = 0 if name.startswith("decoder.") else self._x_order(w.shape[3], stride, x.shape[2])       # every decoder layer: (kh, kw, channel-block) chunks
        if name.startswith("decoder.") and w.shape[1] == 3:
            # ... until the convolution's bf16 input or output reaches 2^31 bytes -- 64 images of 256 channels at 256 x 256, the reference's own batch of
            # BASELINE configs[1]: oneDNN then walks the chunks channel-block major into the one running total (order 1; probed layer by layer,
            # tools/probe_cpu_bf16/check_conv_batch64.py: the reference's decode of 64 images differs from its decode of 48 in exactly these two layers)
            B, Hin, Win = x.shape[0], x.shape[1] * (2 if upsample else 1), x.shape[2] * (2 if upsample else 1)
            if 2 * B * Hin * Win * max(w.shape[3], w.shape[0]) >= 2 ** 31:
                order = 1
        return ops.vx_conv2d(x, w, self.w[name + ".bias"] if bias is None else bias, stride=stride, residual=residual, order=order, upsample=upsample)

    def _x_gn(self, name, x, act=True):
        return ops.vx_groupnorm(x, self.w[name + ".weight"], self.w[name + ".bias"], silu_table=self.silu_table if act else None, groups=32, eps=1e-6)

    def _x_res(self, p, x):
        h = self._x_conv(p + ".conv1", self._x_gn(p + ".norm1", x))
        sc = self._x_conv(p + ".conv_shortcut", x) if (p + ".conv_shortcut") in self.xw else x
        return self._x_conv(p + ".conv2", self._x_gn(p + ".norm2", h), residual=sc)

    def _x_encode_moments(self, img):
        self._x_check_size(int(img.shape[-2]), int(img.shape[-1]))
        x = img.to(self.device, self.dtype).permute(0, 2, 3, 1)
        h = F.pad(x, (0, 8 - x.shape[-1])).contiguous()                                # 3 -> 8 channels: 16-byte pixels
        h = self._x_conv("encoder.conv_in", h)
        for lvl in range(4):
            for j in range(2):
                h = self._x_res(f"encoder.down_blocks.{lvl}.resnets.{j}", h)
            if lvl != 3:
                h = self._x_conv(f"encoder.down_blocks.{lvl}.downsamplers.0.conv", h, stride=2)
        h = self._x_res("encoder.mid_block.resnets.0", h)
        h = self._n_attn("encoder.mid_block.attentions.0", h)
        h = self._x_res("encoder.mid_block.resnets.1", h)
        h = self._x_conv("encoder.conv_out", self._x_gn("encoder.conv_norm_out", h))
        return h.permute(0, 3, 1, 2).contiguous()

    def _x_decode(self, z):
        """`VAEDecoder.forward` (sd3_impls.py:427-444) with every reduction in the reference's torch-CPU order (round 5): pixels equal the
        reference's decode of the same latents bit for bit (tests/golden/decode_b16.npz, vae_b1.npz; CPU twin oracle/vae_exact.py decode)"""
        self._x_check_size(8 * int(z.shape[-2]), 8 * int(z.shape[-1]))
        h = z.to(self.device, self.dtype).permute(0, 2, 3, 1)
        h = F.pad(h, (0, 32 - h.shape[-1])).contiguous()                               # 16 -> 32 channels (zeros): one chunk per tap
        h = self._x_conv("decoder.conv_in", h)
        h = self._x_res("decoder.mid_block.resnets.0", h)
        h = self._n_attn("decoder.mid_block.attentions.0", h)
        h = self._x_res("decoder.mid_block.resnets.1", h)
        for lvl in range(4):
            for j in range(3):
                h = self._x_res(f"decoder.up_blocks.{lvl}.resnets.{j}", h)
            if lvl != 3:
                h = self._x_conv(f"decoder.up_blocks.{lvl}.upsamplers.0.conv", h, upsample=True)
        h = self._x_conv("decoder.conv_out", self._x_gn("decoder.conv_norm_out", h), bias=self.xb_out)     # [B,H,W,32], channels 3.. are padding
        return h[..., :3].permute(0, 3, 1, 2).contiguous()

    def _n_decode(self, z):
        h = z.to(self.device, self.dtype).permute(0, 2, 3, 1).contiguous()
        h = ops.conv2d_nhwc(h, self.pc["decoder.conv_in"])
        h = self._n_res("decoder.mid_block.resnets.0", h)
        h = self._n_attn("decoder.mid_block.attentions.0", h)
        h = self._n_res("decoder.mid_block.resnets.1", h)
        for lvl in range(4):
            for j in range(3):
                h = self._n_res(f"decoder.up_blocks.{lvl}.resnets.{j}", h)
            if lvl != 3:
                h = ops.conv2d_nhwc(h, self.pc[f"decoder.up_blocks.{lvl}.upsamplers.0.conv"], upsample=True)
        h = ops.conv2d_nhwc(self._n_gn("decoder.conv_norm_out", h), self.pc["decoder.conv_out"])     # [B,H,W,4], channel 3 is padding
        return h[..., :3].permute(0, 3, 1, 2).contiguous()

    def _gn_silu(self, name, x, act=True):
        return ops.groupnorm_silu(x.contiguous(), self.w[name + ".weight"], self.w[name + ".bias"], 32, 1e-6, act)

    def _tail(self, x):
        """the 8 extra channels [B, 8, H, W] (ones, then zeros) for an input of x's batch / spatial shape (cached per shape)"""
        key = (x.shape[0], x.shape[2], x.shape[3])
        t = self._tails.get(key)
        if t is None:
            t = torch.zeros(x.shape[0], ONES_PAD, x.shape[2], x.shape[3], device=x.device, dtype=x.dtype)
            t[:, 0] = 1.0
            if len(self._tails) > 32:
                self._tails.clear()
            self._tails[key] = t
        return t

    def _conv(self, name, x, stride=1, padding=1):
        """conv2d with the bias inside the fp32 accumulation (module docstring): one rounding to bf16, as the reference's CPU conv"""
        if self.mode != "miopen":
            return F.conv2d(x, self.w[name + ".weight"], self.w[name + ".bias"], stride=stride, padding=padding)
        return F.conv2d(torch.cat((x, self._tail(x)), dim=1), self.wb[name], None, stride=stride, padding=padding)

    def _flags(self):
        return _Deterministic(self.deterministic)

    def _res(self, p, x):
        h = self._conv(p + ".conv1", self._gn_silu(p + ".norm1", x))
        h = self._conv(p + ".conv2", self._gn_silu(p + ".norm2", h))
        if (p + ".conv_shortcut.weight") in self.w:
            x = self._conv(p + ".conv_shortcut", x, padding=0)
        return x + h

    def _attn(self, p, x):
        """AttnBlock (sd3_impls.py:274-284): one head of C = 512 channels over H*W tokens.  The reference's CPU run projects with fused-bias
        1x1 convolutions and calls the CPU flash kernel (fp32 scores and softmax, un-normalised probabilities rounded to bf16 for the
        P V product, fp32 accumulate, one rounding of the output).  PyTorch-ROCm's bf16 SDPA has no fused kernel for head_dim 512 and
        falls back to the math path, which rounds the SCORES to bf16 (2 % error on a probability at |s| ~ 10).  Here every product
        accumulates in fp32 on bf16-exact operands and is rounded where the reference rounds: 2 GFLOP per image, 1 ms at B = 64.
        (Teacher-forced against the CPU run, tools/probe_vae_layers.py: every convolution and GroupNorm of vae.py agrees with the CPU in
        > 99.95 % of its output elements; this block is the one whose formulation matters.)"""
        B, C, H, W = x.shape
        h = self._gn_silu(p + ".group_norm", x, act=False).reshape(B, C, H * W).transpose(1, 2)              # [B, T, C]
        return x + self._attn_tokens(p, h).transpose(1, 2).reshape(B, C, H, W)

    def _attn_tokens(self, p, h):
        """the block after its GroupNorm, on tokens: h [B, T, C] bf16 -> [B, T, C] bf16 (before the residual add)"""
        B, T, C = h.shape
        f = torch.float32
        h = h.to(f)                                              # bf16-exact values

        def lin(name, t):                                        # fp32 accumulate + bias, ONE rounding to bf16 (a fused-bias bf16 Linear)
            return torch.baddbmm(self.w[name + ".bias"].to(f), t, self.w[name + ".weight"].to(f).t().expand(B, -1, -1)).to(self.dtype)
        q, k, v = lin(p + ".to_q", h).to(f), lin(p + ".to_k", h).to(f), lin(p + ".to_v", h).to(f)
        # online softmax over key blocks of 512, the way torch's CPU flash kernel walks them (ATen FlashAttentionKernel.cpp,
        # kvSplitSize = 512): the un-normalised probabilities -- from the kernel's own fast exp, fexp_u20 -- are rounded to bf16
        # relative to the RUNNING maximum of their block, the row sums come from the fp32 values, the accumulator is rescaled by
        # exp(old max - new max) (accurate exp), the output is dst * (1 / sum): same rounding points, so the result agrees with the
        # reference's CPU run except at rounding ties of the two GEMMs' fp32 sums (99.6 % of the elements, measured on the CPU)
        scale, KV = C ** -0.5, 512
        m = l = acc = None
        for n0 in range(0, T, KV):
            sc = torch.bmm(q, k[:, n0:n0 + KV].transpose(1, 2)) * scale
            bm = sc.amax(dim=-1, keepdim=True)
            m_new = bm if m is None else torch.maximum(m, bm)
            pr = fexp_u20(sc - m_new)                          # the CPU kernel's fast exp for reduced types, see fexp_u20
            ps = pr.sum(dim=-1, keepdim=True)
            pv = torch.bmm(pr.to(self.dtype).to(f), v[:, n0:n0 + KV])
            if m is None:
                l, acc = ps, pv
            else:
                al = torch.exp(m - m_new)
                l, acc = ps + al * l, acc * al + pv
            m = m_new
        a = (acc * (1.0 / l)).to(self.dtype)                 # dst * sum_reciprocal, as the CPU kernel
        return lin(p + ".to_out.0", a.to(f))

    @torch.no_grad()
    def encode_moments(self, img: torch.Tensor) -> torch.Tensor:
        if self.mode == "exact":
            return self._x_encode_moments(img)
        if self.mode == "parity":
            return self._n_encode_moments(img)
        with self._flags():
            return self._encode_moments(img)

    def _encode_moments(self, img: torch.Tensor) -> torch.Tensor:
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
        if self.decode_mode == "exact":
            if z.shape[0] > 64 and not getattr(self, "_warned_big_exact_decode", False):
                import warnings
                self._warned_big_exact_decode = True            # say it once instead of extrapolating silently
                warnings.warn(f"exact VAE decode of {z.shape[0]} images in one call: oneDNN's order switch at 2^31-byte activations was probed at 48 / 64 images per call; "
                              "above 64 the layer set that switches is an extrapolation (no reference run to pin it) -- decode in calls of <= 64 images for pinned pixels")
            return (self._x_decode(z),)
        if self.decode_mode == "parity":
            return (self._n_decode(z),)
        with self._flags():
            return self._decode(z)

    def _decode(self, z):
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
