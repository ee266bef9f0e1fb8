"""`SelftokPipeline` -- the drop-in host API of the reference (mimogpt/infer/SelftokPipeline.py:153-322),
re-authored for MI355X: same constructor, `encoding`, `decoding`, `decoding_with_renderer`,
`NormalizeToTensor`, same checkpoint layouts, same dtypes at every hand-off (bf16 VAE <-> fp32 tokenizer),
same RNG source for the decode noise (global CPU torch generator, :264).

Public attributes users touch in the reference are kept: `.model` (with `.encoder`, `.model`, `.diti`),
`.vae`, `.flow`, `.diti`, `.K`, `._steps`, `.cfg_scale`, `.start`.

Nothing here falls back to a CPU path: without libselftok_hip.so / a GPU the constructor raises.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, gemm_tune, ops, weights as W
from .encoder import QformerEncoderGPU, sinusoid_host
from .mmdit import MMDiTGPU
from .modsurface import ModuleSurface
from .schedule import DiTiCont, FlowSchedule
from .vae import AutoencoderKLGPU

SD3_SCALE, SD3_SHIFT = 1.5305, 0.0609     # SD3LatentFormat (sd3/sd3_impls.py:136-138)
DEFAULT_GEMM = "fp32"                     # see MMDiTGPU.set_gemm
DEFAULT_VAE_DECODE = "parity"             # decoder arithmetic outside gemm='exact' (see SelftokPipeline.__init__: vae_decode_mode)


class NormalizeToTensor(object):
    """PIL/ndarray HWC uint8 -> float tensor CHW in [-1,1] (reference SelftokPipeline.py:85-97)."""

    def __init__(self, reshape=True):
        self.reshape = reshape

    def __call__(self, image):
        image = np.array(image).astype(np.float32)
        image = (image / 127.5 - 1.0).astype(np.float32)
        if self.reshape:
            image = np.reshape(image, (image.shape[0], image.shape[1], -1))
        return torch.from_numpy(image.transpose((2, 0, 1)))


def norm_ip(img, low=-1, high=1):
    """in-place clamp + rescale to [0,1] (reference :135-137); bf16 device tensors use the HIP kernel."""
    assert (low, high) == (-1, 1)
    return ops.clamp01_(img)


class _Tokenizer(ModuleSurface):
    """`pipe.model` : the ImageTokenizer surface (image_tokenizer.py:58-159) the pipeline and users touch: `.encoder`, `.model`
    (the MMDiT), `.diti`, `.k`, and the read-only Module surface (`state_dict()` has the reference checkpoint's keys)."""

    def __init__(self, encoder, model, diti):
        self.encoder, self.model, self.diti = encoder, model, diti
        self.k = diti.K
        self.device = encoder.device

    def _flat_weights(self):
        d = dict(self.encoder._flat_weights())
        d.update(self.model._flat_weights())
        return d


class _Flow(FlowSchedule):
    """`pipe.flow`: RectifiedFlow(50, start, cut_of_k, val_schedule='uniform', shift=1.0, ...) buffers as numpy,
    plus the device-side sampler (p_sample_loop / sample_one_step / euler_step, sd3/rectified_flow.py:165-309)."""

    def __init__(self, num_steps, start, device, parameterization: str = "velocity"):
        super().__init__(num_steps, start)
        self.device = device
        if parameterization not in ("velocity", "x0"):
            raise ValueError(f"parameterization {parameterization!r}: expected 'velocity' or 'x0'")
        self.parameterization = parameterization                              # euler_step (sd3/rectified_flow.py:301-309)
        # sinusoidal embedding of t*1000 for the scheduled timesteps, evaluated with the reference's CPU arithmetic
        t1000 = torch.from_numpy(self.scheduled_t) * 1000.0
        self.t_freq = sinusoid_host(t1000).to(device)                        # [steps,256]
        # cfg_inference embeds floor(t*1000).int().clamp(0,999) instead (sd3/mmdit.py:1126)
        t_unc = torch.floor(torch.from_numpy(self.scheduled_t) * 1000).int().clamp(0, 999)
        self.t_freq_uncond = sinusoid_host(t_unc).to(device)
        # gemm='exact': the reference's own bits of these two tables (torch.cos / sin = MKL VML there: host dependent, shipped as data for the
        # default schedule -- 50 steps from t = 1; tools/oracle/gen_pos_table.py).  Another schedule falls back to this host's evaluation.
        self.t_freq_exact = self.t_freq_uncond_exact = None
        if num_steps == 50 and float(start) == 1.0:
            import os
            tab = torch.from_numpy(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "flow50_t_sincos.npy")))
            self.t_freq_exact, self.t_freq_uncond_exact = tab[0].to(device).contiguous(), tab[1].to(device).contiguous()

    def resolve_super_mask(self, super_mask, K: int):
        """[K] (or batch-uniform [B,K]) visibility pattern -> (device int64 index of the visible tokens, their positions as numpy)"""
        sm = torch.as_tensor(super_mask).cpu()
        if sm.dim() == 2:
            if not bool((sm == sm[:1]).all()):      # SelftokPipeline._sample splits such a batch into groups of equal pattern before it gets here
                raise NotImplementedError("p_sample_loop takes ONE visibility pattern per call (decode the samples in groups of equal pattern)")
            sm = sm[0]
        if sm.numel() != K:
            raise ValueError(f"super_mask has {sm.numel()} entries, the tokenizer has K = {K} tokens")
        vis_pos = np.nonzero(sm.reshape(-1).bool().numpy())[0].astype(np.int64)
        return torch.from_numpy(vis_pos).to(self.device), vis_pos

    @torch.no_grad()
    def p_sample_loop(self, dit: MMDiTGPU, noise: torch.Tensor, ehs: torch.Tensor, k_table: np.ndarray,
                      context_see_xt: bool = True, uncond_scale: float = 1.0, max_steps: Optional[int] = None,
                      trace: Optional[list] = None, prefix_k: Optional[int] = None, super_mask=None, visible=None) -> torch.Tensor:
        """`prefix_k`: the reference loop's `super_mask` (rectified_flow.py:226-227, mask = mask * super_mask) for the prefix mask
        arange(K) < prefix_k -- only the first prefix_k tokens are ever visible (decode from a partial token sequence).
        `super_mask`: the same hook for ANY visibility pattern over the K tokens ([K] bool / 0-1, the same for every sample): the visible
        tokens are gathered once (MMDiTGPU.gather_context) and the step mask is a prefix of that list.  `visible`: the same pattern
        already resolved by `resolve_super_mask` -- what a caller that captures this loop in a hipGraph passes (the resolution reads
        the mask on the host and uploads an index tensor, neither of which may happen while a stream is capturing)."""
        B = noise.shape[0]
        x = noise.to(self.device).float().contiguous()
        hp, wp = x.shape[-2] // 2, x.shape[-1] // 2
        ctx0 = dit.embed_context(ehs)                                         # step independent
        tables, vis_pos = None, None
        if visible is None and super_mask is not None:
            visible = self.resolve_super_mask(super_mask, ctx0.shape[1])
        if visible is not None:
            if dit.gemm == "exact" and not np.array_equal(visible[1], np.arange(len(visible[1]))):
                # the exact mode keeps every context key at its POSITION in the reference's key sequence (kv blocks of 512, MKL's K-blocks); gathering the
                # visible tokens of a non-prefix pattern moves them.  Prefix patterns (the sampler's own masks, prefix_k) are exact.
                raise NotImplementedError("gemm='exact' reproduces the reference's bits for prefix visibility patterns; decode a non-prefix super_mask with gemm='fp32' / 'f16x2'")
            idx_dev, vis_pos = visible                                        # device index tensor (made outside any capture), host positions
            ctx0, tables, _ = dit.gather_context(ctx0, index=idx_dev)
        cqkv0 = dit.block0_context_qkv(ctx0, tables) if ctx0.shape[1] > 0 else None   # block 0's context QKV is step independent too
        steps = self.num_timesteps if max_steps is None else min(max_steps, self.num_timesteps)
        for i in range(steps):
            n_live = int(k_table[i]) + 1                                      # mask = arange(K) <= k  (models_ours.py:353)
            if prefix_k is not None:
                n_live = min(n_live, int(prefix_k))
            if vis_pos is not None:                                           # visible tokens at positions < n_live: a prefix of the gathered list
                n_live = int(np.searchsorted(vis_pos, n_live, side="left"))
            exact = dit.gemm == "exact" and self.t_freq_exact is not None
            tf = (self.t_freq_exact if exact else self.t_freq)[i:i + 1].expand(B, -1).contiguous()
            t_name = float(self.scheduled_t[i])                               # names the embedded timestep (MMDiTGPU._step_modulations)
            if uncond_scale == 1.0:
                y = dit.velocity_tokens(x, tf, ctx0, n_live, context_see_xt, cqkv0, tables, t_key=("t", t_name))
                yu = None
            else:
                # CFG branch (rectified_flow.py:280-289): the conditional call omits context_see_xt (-> False) and
                # the unconditional one sees no context token at all
                y = dit.velocity_tokens(x, tf, ctx0, n_live, False, cqkv0, tables, t_key=("t", t_name))
                tfu = (self.t_freq_uncond_exact if exact else self.t_freq_uncond)[i:i + 1].expand(B, -1).contiguous()
                yu = dit.velocity_tokens(x, tfu, ctx0, 0, False, t_key=("floor", t_name))   # cfg_inference: no context key visible at all
            if self.parameterization == "x0":
                # the model output is the clean latent: x_prev = v + a_prev (x - v) / a_t  (euler_step, rectified_flow.py:305-307);
                # the CFG mix rides in the unpatchify kernel, the update is the reference's own chain of fp32 element-wise ops
                _, v = ops.unpatchify_cfg_euler(y, None, 0.0, y_uncond=yu, cfg_scale=uncond_scale, C=x.shape[1], hp=hp, wp=wp)
                x = v + float(self.scheduled_t_prev[i]) * (x - v) / float(self.scheduled_t[i])
            else:
                x, _ = ops.unpatchify_cfg_euler(y, x, float(self.dt[i]), y_uncond=yu, cfg_scale=uncond_scale, C=x.shape[1], hp=hp, wp=wp)
            if trace is not None:
                trace.append(x.clone())
        return x


def _on_own_device(fn):
    """run a public entry point with the pipeline's GPU as the current device: every HIP launch goes to the current device's
    stream, and a process may hold several pipelines on different GPUs or change the current device between calls"""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        with torch.cuda.device(self.device):
            return fn(self, *a, **k)
    return wrapped


class SelftokPipeline():
    def __init__(self, cfg, ckpt_path, sd3_path, datasize=256, start=1.0, cfg_scale=1, model_type='sd3',
                 dtype=torch.bfloat16, ema_decoder=False, device=None, state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 vae_state_dict: Optional[Dict[str, torch.Tensor]] = None, verbose: bool = True, gemm: Optional[str] = None,
                 vae_mode: Optional[str] = None, tune_gemm: Optional[bool] = None, encoder_mode: Optional[str] = None,
                 vae_encode_mode: Optional[str] = None, vae_decode_mode: Optional[str] = None):
        """cfg: parse_args_from_yaml(...) ; ckpt_path: tokenizer .pth ; sd3_path: diffusers SD3 folder (…/vae/…).
        `state_dict` / `vae_state_dict` (extensions) bypass the files, e.g. with weights.synthetic_state_dict().
        `gemm` (extension): arithmetic of the MMDiT block Linears, 'fp32' (hipBLASLt fp32) or 'f16x2' (fp32-equivalent
        split GEMM on the f16 matrix cores, csrc/gemm_split.hip); default DEFAULT_GEMM.
        `vae_encode_mode` / `vae_decode_mode` (extensions; `vae_mode` sets both): arithmetic of the two halves of the SD3 VAE (vae.AutoencoderKLGPU).
        'exact': every reduction in the summation ORDER of the reference's torch-CPU run (csrc/vae_exact.hip, fp32 matrix cores) -- encoder: latents and
        token ids from pixels equal the reference's bit for bit; decoder: pixels equal the reference's decode of the same latents, incl. its batch
        dependence (a 3x3 layer whose bf16 activation reaches 2^31 bytes -- 64 images per call at 256 x 256 -- takes oneDNN's order 1: probed at 48 / 64
        images per call, extrapolated above).  'parity': every convolution / GroupNorm through csrc/conv.hip -- fp32 accumulation with the bias inside, one
        rounding, the reference's CPU arithmetic in its own summation order; no MIOpen, bit-stable, batch independent, 5x faster.  'miopen' / 'fast': the
        rounds 1-3 routes through MIOpen (both halves).  Defaults (round 6): encode 'exact' at the probed sizes (128 / 256 / 320 px; token ids need it),
        'parity' elsewhere; decode 'parity' (pixels within the north star's 1e-3 dB of the reference either way), and 'exact' while gemm == 'exact'
        (the mode whose pixels ARE the reference's) unless `vae_decode_mode` / `vae_mode` was given.
        `encoder_mode` (extension): 'exact' (default: the Q-Former encoder in the summation order of every reduction and the polynomial of every
        transcendental torch-CPU executes for the reference, csrc/encoder_exact.hip -- pre-quantizer features and token ids equal those of the
        reference's runs at 8 <= B <= 64 images per call bit for bit, and are the same here for every batching; the reference itself takes other MKL paths at
        B = 1 and for Linears below 16 / 512 rows, which this mode does not follow: DESIGN 15.2, 15.8) or 'fast' (hipBLASLt GEMMs + the fused rounds 1-3 kernels: features within 6e-5, ~2x faster encoder).
        No environment variable is read: every knob is a constructor argument.
        `tune_gemm` (extension, OPT-IN): pick hipBLASLt's kernel for the fp32 block Linears by a ~4 s measurement per batch size
        (gemm_tune.py) -- up front through `pipe.tune_linears(batch)`, or at the first decode of a batch size.  TunableOp is enabled
        (tuning off) only inside this pipeline's own sampler calls and the caller's torch.cuda.tunable flags are restored; on another
        hipBLASLt build than the one the candidate kernels were found with it does nothing, loudly.  Default off."""
        _lib.load()                                                           # fail loudly if the HIP library is missing
        self.tune_gemm = bool(tune_gemm)
        self.gemm_tune_report = None
        if device is None:
            device = "cuda"
        if not torch.cuda.is_available():
            raise _lib.SelftokHipError("SelftokPipeline needs an MI355X (ROCm) device; there is no CPU path")
        if model_type != 'sd3':
            raise ValueError(f"Unsupported MODEL_TYPE: {model_type}. Expected 'sd3'")
        self.cfg, self.datasize, self.model_type, self.dtype = cfg, datasize, model_type, dtype
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.SelftokHipError(f"SelftokPipeline needs a GPU device, got {self.device}")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(self.device):                                  # our launches use the current device's stream
            self._build(cfg, ckpt_path, sd3_path, start, cfg_scale, dtype, ema_decoder, state_dict, vae_state_dict, verbose, gemm, vae_mode, encoder_mode,
                        vae_encode_mode, vae_decode_mode)

    def _build(self, cfg, ckpt_path, sd3_path, start, cfg_scale, dtype, ema_decoder, state_dict, vae_state_dict, verbose, gemm, vae_mode=None, encoder_mode=None,
               vae_encode_mode=None, vae_decode_mode=None):
        p = cfg.tokenizer.params
        p.noise_schedule_config.is_eval = cfg.common.is_eval
        # configuration knobs the reference honours but this hot path does not implement: refuse, never ignore silently
        if p.get("diffusion_type", "flow") != "flow":
            raise NotImplementedError("diffusion_type != 'flow' (the Gaussian-diffusion sampler is outside the hot path)")
        nsc = p.noise_schedule_config
        self.parameterization = str(nsc.get("parameterization", "velocity"))
        if self.parameterization not in ("velocity", "x0"):
            raise ValueError(f"noise_schedule_config.parameterization {self.parameterization!r}: expected 'velocity' or 'x0'")
        cut = p.get("cut_of_k", None)
        if cut and float(cut) < 1:
            raise NotImplementedError("cut_of_k < 1 (context padding, rectified_flow.py:216-224) is not implemented; the shipped configs do not set it")
        pre_norm = bool(p.get("encoder_config", {}).get("pre_norm", False))        # models_ours.py:219-220 (round 6; False in the shipped configs)
        K = int(p.k)
        renderer = "Renderer" in str(p.model)
        self.diti = DiTiCont(1000, K, p.stages, p.k_per_stage)
        self.K = K
        self.context_see_xt = bool(p.get("context_see_xt", False))

        vsd = vae_state_dict if vae_state_dict is not None else W.load_vae_checkpoint(sd3_path)
        W.check_vae_state_dict(vsd)
        probed = int(self.datasize) in AutoencoderKLGPU.EXACT_SIZES
        enc_mode = vae_encode_mode or vae_mode or ("exact" if probed else "parity")
        self._vae_decode_explicit = (vae_decode_mode or vae_mode) is not None
        dec_mode = vae_decode_mode or vae_mode or ("exact" if (probed and (gemm or DEFAULT_GEMM) == "exact" and enc_mode in ("exact", "parity")) else
                                                   (enc_mode if enc_mode in ("miopen", "fast") else DEFAULT_VAE_DECODE))
        self.vae = AutoencoderKLGPU(vsd, self.device, dtype, mode=enc_mode, decode_mode=dec_mode)

        self.verbose = verbose
        self._say("Loading all...")
        sd = state_dict if state_dict is not None else W.load_tokenizer_checkpoint(ckpt_path)
        W.check_tokenizer_state_dict(sd, K, renderer=renderer, ema=bool(ema_decoder))   # load_state_dict(strict=False) / strict EMA load
        self.ema_decoder = ema_decoder
        dit_sd = sd
        if ema_decoder:   # reference :193-194: EMA copy of the DiT under 'ema_state_dict' (keys without the 'model.' prefix)
            dit_sd = {"model." + k: v for k, v in sd["ema_state_dict"].items()}       # strict contract checked above: bare MMDiT keys only
        encoder = QformerEncoderGPU(sd, self.device, K, mode=encoder_mode or "exact", pre_norm=pre_norm)
        dit = MMDiTGPU(dit_sd, self.device, K, renderer=renderer)
        dit.set_gemm(gemm or DEFAULT_GEMM)
        self.model = _Tokenizer(encoder, dit, self.diti)

        self.count = 0
        self.count_cfg = 0
        self.start = start
        self.cfg_scale = cfg_scale
        self.cut_of_k = p.get("cut_of_k", None) or None
        self._steps = 50
        self.flow = _Flow(self._steps, self.start, self.device, self.parameterization)
        self.k_table = self.diti.to_indices(self.flow.t_long)                # k for each of the 50 steps
        self.cond_vary = True
        self.saved_images = 8
        self._graphs = {}        # (B, latent, steps, scale) -> (hipGraph, static noise, static ehs, static output)

    def _say(self, msg):
        if self.verbose:          # the reference prints these progress lines unconditionally (:192,212,223,230,292,299,320)
            print(msg)

    # ------------------------------------------------------------------------------------------------
    @_on_own_device
    @torch.no_grad()
    def encode_latents(self, images: torch.Tensor) -> torch.Tensor:
        """VAE mean -> SD3LatentFormat.process_in -> fp32 (reference :214-218)"""
        moments = self.vae.encode_moments(images.to(dtype=self.dtype, device=self.device))
        return ops.latent_process_in(moments.contiguous(), moments.shape[1] // 2, SD3_SHIFT, SD3_SCALE)

    @_on_own_device
    @torch.no_grad()
    def encoding(self, images, device=None):
        self._say("Begin encoding.")
        x_0 = self.encode_latents(images)
        _, tokens = self.model.encoder(x_0, d=None)
        self._say('End encoding.')
        return tokens

    @_on_own_device
    @torch.no_grad()
    def _codes(self, idx) -> torch.Tensor:
        if isinstance(idx, np.ndarray):
            if not np.issubdtype(idx.dtype, np.integer):
                raise TypeError(f"token ids must be integers, got {idx.dtype} (the reference indexes the codebook with them)")
            idx = np.ascontiguousarray(idx if idx.dtype in (np.int64, np.int32) else idx.astype(np.int64))   # uint16 / int16 / uint8 wire formats
            token_idx = torch.from_numpy(idx).to(self.device)
        else:
            if idx.is_floating_point() or idx.dtype == torch.bool:
                raise TypeError(f"token ids must be integers, got {idx.dtype}")
            token_idx = idx.to(self.device)
        B = token_idx.shape[0]
        return self.model.encoder.codes_ln(token_idx.reshape(B, -1))          # get_output_from_indices + final_layer_norm3

    @_on_own_device
    @torch.no_grad()
    def _to_pixels(self, pred_x0: torch.Tensor) -> torch.Tensor:
        z = ops.latent_process_out(pred_x0, SD3_SHIFT, SD3_SCALE)             # process_out + .to(bf16) (:285-287)
        recons = self.vae.decode(z)[0].contiguous()
        return norm_ip(recons, -1, 1)

    @_on_own_device
    def tune_linears(self, batch: int, renderer: Optional[bool] = None, guided: bool = False):
        """(extension) measure hipBLASLt's candidate kernels for the fp32 block Linears of a decode at `batch` images now (~4 s, once
        per batch size; gemm_tune.py) and switch `tune_gemm` on, so that this pipeline's sampler calls run under TunableOp with the
        winners.  `guided`: a CFG decode (two MMDiT passes per step, rows of 2 x batch).  Returns the report ({(N, K): (kernel or None,
        ms default, ms chosen)}), or None in f16x2 mode / on another hipBLASLt build."""
        self.tune_gemm = True
        tokens = (self.datasize // 16) ** 2
        renderer = self.model.model.renderer if renderer is None else renderer
        self._tune_linears(int(batch) * (2 if guided else 1), [self.K - 1] if renderer else self.k_table, tokens)
        return self.gemm_tune_report

    @_on_own_device
    def _tune_linears(self, B: int, k_table, image_tokens: int) -> None:
        """fp32 mode: choose hipBLASLt's kernels for this batch size's block Linears once (gemm_tune.py)"""
        if self.tune_gemm and self.model.model.gemm == "fp32":
            rows, reps = gemm_tune.step_row_counts(B, k_table, image_tokens)
            self.gemm_tune_report = gemm_tune.autotune_linears(rows, self.device, reps=reps, verbose=self.verbose)

    def _tunable_scope(self):
        """TunableOp on for the duration of one of OUR sampler calls when kernels were installed; the caller's flags come back after"""
        import contextlib
        return gemm_tune.enabled() if (self.tune_gemm and self.gemm_tune_report and self.model.model.gemm == "fp32") else contextlib.nullcontext()

    def set_gemm(self, mode: str) -> str:
        """switch the MMDiT between 'fp32', 'f16x2' and 'exact' (see MMDiTGPU.set_gemm); returns the mode in force.  Unless the decoder's arithmetic was
        chosen explicitly (`vae_decode_mode` / `vae_mode`), 'exact' brings the exact-order VAE decoder with it (the mode whose pixels are the reference's
        bit for bit) and the other modes return to the default decoder."""
        got = self.model.model.set_gemm(mode)
        if not self._vae_decode_explicit and self.vae.mode in ("exact", "parity"):
            want = "exact" if (got == "exact" and int(self.datasize) in AutoencoderKLGPU.EXACT_SIZES) else DEFAULT_VAE_DECODE
            if self.vae.decode_mode != want:
                self.vae.set_decode_mode(want)
        return got

    @torch.no_grad()
    def _checked(self, run):
        """run() -> latent.  In 'f16x2' GEMM mode an activation outside the fp16 range (|a| >= 65504) invalidates the
        result (sticky device flag, one host read per call): redo the call on the fp32 library GEMMs."""
        dit = self.model.model
        out = run()
        if dit.gemm == "f16x2" and int(dit.overflow.item()) != 0:
            print("[selftok] f16x2 GEMM: activation outside the fp16 range -> recomputing this call with fp32 GEMMs")
            dit.overflow.zero_()
            dit.set_gemm("fp32")
            try:
                out = run()
            finally:
                dit.set_gemm("f16x2")              # the split weights are kept: no re-pack
        return out

    @torch.no_grad()
    def _sample(self, xt, ehs, max_steps, uncond_scale, use_graph, prefix_k=None, super_mask=None):
        """the 50-step loop, optionally replayed from a hipGraph captured once per (batch, latent size) -- the loop is
        ~21k kernel launches; at small batch the host cannot issue them as fast as the GPU retires them."""
        if super_mask is not None:
            sm = torch.as_tensor(super_mask).cpu()
            if sm.dim() == 2 and sm.shape[0] == xt.shape[0] and not bool((sm == sm[:1]).all()):
                # a visibility pattern PER SAMPLE (the reference's `mask * super_mask` with a [B, K] tensor, rectified_flow.py:226-227):
                # samples are independent, so the batch is decoded in groups of equal pattern (a group's context is gathered once)
                rows = sm.reshape(sm.shape[0], -1).bool().numpy()
                out = torch.empty(xt.shape, dtype=torch.float32, device=self.device)
                seen = {}
                for b in range(rows.shape[0]):
                    seen.setdefault(rows[b].tobytes(), []).append(b)
                for idx in seen.values():
                    sel = torch.as_tensor(idx)
                    out[sel.to(self.device)] = self._sample(xt[sel], ehs[sel.to(ehs.device)], max_steps, uncond_scale, use_graph, prefix_k, rows[idx[0]])
                return out
        if not use_graph:
            return self.flow.p_sample_loop(self.model.model, xt, ehs, self.k_table, context_see_xt=True,
                                           uncond_scale=uncond_scale, max_steps=max_steps, prefix_k=prefix_k, super_mask=super_mask)
        visible = None if super_mask is None else self.flow.resolve_super_mask(super_mask, self.K)     # host read + upload: before the capture
        sm_key = None if visible is None else visible[1].tobytes()
        key = (tuple(xt.shape), tuple(ehs.shape), max_steps, float(uncond_scale), self.model.model.gemm, self.model.model.PRESPLIT, self.model.model.SPLITK, prefix_k, sm_key)
        if key not in self._graphs:
            s_noise = torch.empty(xt.shape, dtype=torch.float32, device=self.device)
            s_ehs = torch.empty_like(ehs)
            s_noise.copy_(xt); s_ehs.copy_(ehs)
            run = lambda: self.flow.p_sample_loop(self.model.model, s_noise, s_ehs, self.k_table, context_see_xt=True,
                                                  uncond_scale=uncond_scale, max_steps=max_steps, prefix_k=prefix_k, visible=visible)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):       # warm-up outside capture (hipBLASLt / allocator warm)
                run()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            dit = self.model.model
            dit._capture_refs = []               # remembered modulations the capture reads: kept alive with the graph (MMDiTGPU._step_modulations)
            try:
                with torch.cuda.graph(g):
                    s_out = run()
                refs = dit._capture_refs
            finally:
                dit._capture_refs = None
            self._graphs[key] = (g, s_noise, s_ehs, s_out, refs)
        g, s_noise, s_ehs, s_out = self._graphs[key][:4]
        s_noise.copy_(xt.to(self.device)); s_ehs.copy_(ehs)
        g.replay()
        return s_out.clone()

    @_on_own_device
    @torch.no_grad()
    def decoding(self, idx, device=None, noise: Optional[torch.Tensor] = None, return_latent: bool = False,
                 max_steps: Optional[int] = None, uncond_scale: float = 1.0, use_graph: bool = False,
                 prefix_k: Optional[int] = None, super_mask=None):
        """idx: np.ndarray int64 [B,K] -> bf16 [B,3,H,W] in [0,1] (reference :227-294).  `noise` (extension) replaces the
        `torch.randn` draw from the global CPU generator (:264); `uncond_scale` exposes the dormant CFG branch
        (p_sample_loop's argument of that name).  `prefix_k` (extension): decode from the first prefix_k tokens only -- the
        reference loop's `super_mask` hook with a prefix mask (rectified_flow.py:226-227; README.md:241: an AR model emits the
        sequence in reverse order, `tokens.from_ar_order` restores it and `tokens.pad_prefix` pads a partial one to [B,K]).
        `super_mask` (extension): the same hook with any visibility pattern over the K tokens ([K] bool / 0-1 array)."""
        self._say("Begin decoding.")
        if prefix_k is not None and not (0 <= int(prefix_k) <= self.K):
            raise ValueError(f"prefix_k must be in [0, {self.K}]")
        outs_q = self._codes(idx)
        B = outs_q.shape[0]
        # t_mapped = timestep_map[0] -> k = K-1 -> enc_mask all true -> encoder_hidden_states = outs_q (:243-252)
        k0 = int(self.diti.to_indices(self.flow.t_long[:1])[0])
        ehs = outs_q if k0 >= self.K - 1 else outs_q * (torch.arange(self.K, device=self.device) <= k0)[None, :, None]
        latent_dim = self.datasize // 8
        xt = noise if noise is not None else torch.randn(B, 16, latent_dim, latent_dim)
        self._tune_linears(B * (2 if uncond_scale != 1.0 else 1), self.k_table, (latent_dim // 2) ** 2)
        with self._tunable_scope():
            pred_x0 = self._checked(lambda: self._sample(xt, ehs, max_steps, uncond_scale, use_graph, prefix_k, super_mask))
        recons = self._to_pixels(pred_x0)
        self._say('End decoding.')
        return (recons, pred_x0) if return_latent else recons

    @_on_own_device
    @torch.no_grad()
    def decoding_with_renderer(self, idx, device=None, return_latent: bool = False):
        """one MMDiT_Renderer pass instead of the 50-step loop (reference :296-322)"""
        self._say("Begin decoding with Renderer.")
        outs_q = self._codes(idx)
        self._tune_linears(outs_q.shape[0], [self.K - 1], (self.datasize // 16) ** 2)
        with self._tunable_scope():
            pred_x0 = self._checked(lambda: self.model.model(y=None, encoder_hidden_states=outs_q)[0])
        recons = self._to_pixels(pred_x0)
        self._say('End decoding with Renderer.')
        return (recons, pred_x0) if return_latent else recons
