"""SD3-style MMDiT decoder (2.09 B params) of Selftok on MI355X: host orchestration of hipBLASLt fp32 GEMMs
and our HIP kernels (fused residual+LayerNorm+adaLN, bias+GELU, MFMA two-segment attention, patchify,
unpatchify+CFG+Euler).

Counterpart of the reference's `MMDiT.forward` / `MMDiT_Renderer.forward` / `cfg_inference`
(mimogpt/models/selftok/sd3/mmdit.py:992-1101, 1511-1620, 1117-1163) with `JointBlock` ->
`block_mixing` -> `DismantledBlock` (:441-606) and `FinalLayer` (:609-645), over the reference's
checkpoint keys (`model.*`).

Design points (none changes a result):
  * context-stream adaLN tables are functions of token position only -> computed once per model
    (the reference recomputes 23 x [K,1536]->[K,9216] GEMMs every call: 27 % of a step);
  * the attention mask of the reference is `context key j visible iff j <= k_b`; within one sampler
    step k is the same for the whole batch, so the context stream is simply TRUNCATED to its k+1 live
    tokens (dead rows can be read by nobody and the model returns only the image stream) and no mask
    exists at all; a per-sample `kvis` path remains for callers that mix timesteps in a batch;
  * context embedding (Linear 16->1536 + pos-embed) does not depend on the step -> once per decode.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .modsurface import ModuleSurface
from .encoder import sinusoid_host
from .schedule import DiTiCont
from .weights import DIT_DEPTH, DIT_HEADS, DIT_HIDDEN, POS_MAX_DIT


class MMDiTGPU(ModuleSurface):
    _sd_prefix = "model."
    GEMM_MODES = ("fp32", "f16x2", "exact")
    PRESPLIT = True     # f16x2 mode: producers (LN-modulate, attention, fc1+GELU) hand the next Linear its input already split
    EXACT_FUSED_RESIDUAL_LN = True   # gemm='exact': `x + gate * Linear(.)` inside the LayerNorm pass that follows (ops.ex_res_layernorm_mod) instead of the Linear's epilogue
    SPLITK = True       # f16x2 mode, <= ops.SPLITK_MAX_ROWS rows (one .. four images): several work-groups per output tile (ops.f16x2_ksplit)

    def __init__(self, sd: Dict[str, torch.Tensor], device, K: int, renderer: bool = False, gemm: str = "fp32"):
        self.device, self.K, self.renderer = device, K, renderer
        self.gemm = "fp32"
        self._packed = {}                                                   # linear name -> f16x2-split weight image
        self._mod_cache = {}                                                # (timestep name, gemm mode) -> modulations of a single-image step
        self._capture_refs = None                                           # list while a caller captures a hipGraph (see _step_modulations)
        self._trace = None                                                  # tests: a list that receives the image stream after every joint block
        self.overflow = torch.zeros(1, dtype=torch.int32, device=device)    # sticky fp16-range flag of the split GEMMs
        self.w = {k: v.to(device=device, dtype=torch.float32).contiguous() for k, v in sd.items() if k.startswith("model.")}
        H = DIT_HIDDEN
        if not renderer:
            self.pe_w = self.w["model.x_embedder.proj.weight"].reshape(H, -1).t().contiguous()      # [64,1536]
        self._pos_cache = {}
        half = 128
        self.freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half).to(device)
        # context adaLN tables [K, 6H] for blocks 0..22 (block 23's context stream is pre_only: modulated by c)
        pos_emb = sinusoid_host(torch.from_numpy(DiTiCont.get_position(np.arange(K))).to(torch.int64)).to(device)
        self.ctx_tables = []
        for i in range(DIT_DEPTH - 1):
            p = f"model.joint_blocks.{i}.context_block"
            h = self.lin(p + ".t_embedder.mlp.0", pos_emb)
            h = self.lin(p + ".t_embedder.mlp.2", ops.silu(h))
            self.ctx_tables.append(self.lin(p + ".adaLN_modulation.1", ops.silu(h)).contiguous())
        self.context_pos_embed = self.w["model.context_pos_embed"][0].contiguous()                    # [K,H]
        self._tables_fast, self._tables_exact = self.ctx_tables, None
        if gemm != "fp32":
            self.set_gemm(gemm)

    # ---- GEMM arithmetic of the block Linears ---------------------------------------------------
    def set_gemm(self, mode: str) -> str:
        """'fp32': hipBLASLt fp32 GEMMs (PyTorch-ROCm) + the fp32-input-MFMA attention kernel.  'f16x2': the qkv / proj / fc1 / fc2
        Linears of the 24 joint blocks (99.6 % of the decode FLOPs) run on ops.linear_f16x2 and the joint attention on
        attn64_f16x2_kernel -- fp32-equivalent split arithmetic on the f16 matrix cores, measured MORE accurate against fp64 than the
        fp32 kernels they replace (tests/test_gemm_gpu.py, tests/test_kernels_gpu.py).  Weights are split once, here.  If a weight
        is outside the fp16 range the mode stays 'fp32'; activations outside it raise `self.overflow` (checked once per decode
        call by the pipeline, which then recomputes in 'fp32').  Returns the mode in force."""
        if mode not in self.GEMM_MODES:
            raise ValueError(f"gemm mode {mode!r}: expected one of {self.GEMM_MODES}")
        if mode == "exact":
            self._build_exact()
        self.ctx_tables = self._tables_exact if mode == "exact" else self._tables_fast
        if mode == "f16x2" and not self._packed:
            flag = torch.zeros(1, dtype=torch.int32, device=self.device)
            packed = {}
            for name, w in self.w.items():
                if (name.endswith(".weight") and ".joint_blocks." in name and w.dim() == 2
                        and any(t in name for t in (".attn.qkv.", ".attn.proj.", ".mlp.fc1.", ".mlp.fc2."))
                        and ops.linear_f16x2_supported(w.shape[0], w.shape[1])):
                    packed[name[:-len(".weight")]] = ops.linear_f16x2_pack(w, flag)
            if int(flag.item()) != 0:
                print("[selftok] f16x2 GEMM mode refused: a weight is outside the fp16 range; staying on fp32 GEMMs")
                mode = "fp32"
            else:
                self._packed = packed
        self.gemm = mode
        return mode

    # ---- 'exact': every Linear / LayerNorm / GELU / SiLU / attention as the sequence of fp32 operations torch-CPU executes for the reference ----
    def _build_exact(self):
        """the input-independent pieces of the exact mode, once: the context adaLN tables through the exact Linear / SiLU from the reference's
        position table (selftoktokenizer_amd/data), the PatchEmbed convolution as a Linear over the (kh, kw, ic)-ordered patch"""
        if self._tables_exact is not None:
            return
        from .encoder import encoder_pos_embedding
        pos_emb = encoder_pos_embedding(self.K).to(self.device)
        w = self.w
        ex = lambda n, t: ops.ex_linear(t, w[n + ".weight"], w[n + ".bias"])
        silu = lambda t: ops.ex_unary(t.contiguous(), "silu")
        tabs = []
        for i in range(DIT_DEPTH - 1):
            p = f"model.joint_blocks.{i}.context_block"
            h = ex(p + ".t_embedder.mlp.2", silu(ex(p + ".t_embedder.mlp.0", pos_emb)))
            tabs.append(ex(p + ".adaLN_modulation.1", silu(h)).contiguous())
        self._tables_exact = tabs
        if self.renderer:
            # MMDiT_Renderer.forward (sd3/mmdit.py:1511-1620): no PatchEmbed (mask_token + positional_embedding), t = 1000 for every sample -- its
            # sinusoid is the first row of the sampler's table (scheduled_t[0] * 1000 = 1000.0: the same function of the same input; torch.cos / sin are
            # MKL VML on the reference host, shipped as data: tools/oracle/gen_pos_table.py)
            import os
            tab = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "flow50_t_sincos.npy"))
            self._t1000_exact = torch.from_numpy(tab[0, 0:1].copy()).to(self.device)
            return
        self.pe_w_exact = w["model.x_embedder.proj.weight"].permute(0, 2, 3, 1).reshape(DIT_HIDDEN, -1).contiguous()
        self._pe_perm = torch.arange(64, device=self.device).reshape(16, 2, 2).permute(1, 2, 0).reshape(-1)

    def _silu(self, t):
        return ops.ex_unary(t.contiguous(), "silu") if self.gemm == "exact" else ops.silu(t)

    def lin(self, name, x, gelu: bool = False, out_split: bool = False):
        """x fp32 [..., K], or a split activation (ops.SplitAct) when `self._pre(name)`; out_split: return the split form for the
        next Linear (only with a split input)."""
        w, b = self.w[name + ".weight"], self.w[name + ".bias"]
        if self.gemm == "exact":
            assert not out_split and not isinstance(x, ops.SplitAct)
            return ops.ex_linear(x, w, b, gelu=gelu)
        if isinstance(x, ops.SplitAct):
            assert name in self._packed, f"split activation handed to Linear {name!r}, which has no f16x2-split weight (set_gemm('f16x2') packs the block Linears)"
            return ops.linear_f16x2_split(x, self._packed[name], b, w.shape[0], gelu=gelu, overflow=self.overflow, out_split=out_split,
                                          ksplit=self._ksplit(x.rows, w))
        assert not out_split
        if self.gemm == "f16x2" and name in self._packed:
            return ops.linear_f16x2(x, self._packed[name], b, w.shape[0], gelu=gelu, overflow=self.overflow)
        if gelu:
            return ops.linear_gelu(x, w, b)
        return F.linear(x, w, b)

    def _ksplit(self, rows: int, w) -> int:
        return ops.f16x2_ksplit(rows, w.shape[0], w.shape[1]) if self.SPLITK else 1

    def _pre(self, name) -> bool:
        """does Linear `name` take its input as a split activation?"""
        return self.PRESPLIT and self.gemm == "f16x2" and name in self._packed

    def _ln(self, consumer, x, **kw):
        """residual_ln_mod whose normalised output feeds Linear `consumer`: split form if that Linear takes it"""
        if self.gemm == "exact":
            return x, ops.ex_layernorm_mod(x, shift=kw.get("shift"), scale=kw.get("scale"), per_sample=bool(kw.get("per_sample", False)))
        return ops.residual_ln_mod(x, split=self._pre(consumer), overflow=self.overflow, **kw)

    def _res_ln(self, consumer, x, lin_name, lin_in, *, gate, gate_per_sample, split=None, **ln_kw):
        """x' = x + gate * Linear(lin_in);  n = LN(x') * (1 + scale) + shift  ->  (x', n).
        With a split input the residual update rides in the Linear's epilogue (one [B,T,H] fp32 round trip less) and the LN kernel
        only normalises; otherwise residual_ln_mod does both from the stored Linear output.  Same bits either way."""
        if self.gemm == "exact":
            # x' = x + gate * Linear(lin_in): the product and the sum rounded separately in the Linear's epilogue (`x + gate.unsqueeze(..) * post_attention(attn)`,
            # sd3/mmdit.py:485-496); a per-token gate table is indexed by row % tokens, a per-sample one by row / tokens
            wl, bl = self.w[lin_name + ".weight"], self.w[lin_name + ".bias"]
            rows = x.shape[1]
            # attn.proj reads a SLICE of the concatenated attention output in the reference (non-contiguous -> at::linear = matmul + add_(bias): bias last)
            bias_last = lin_name.endswith(".attn.proj")
            if self.EXACT_FUSED_RESIDUAL_LN:
                # round 6: the Linear keeps its plain epilogue (the matrix pipe does not wait for 64 operand loads per lane) and the residual update rides in the
                # LayerNorm pass behind it -- the same fp32 operations in the same order
                y = ops.ex_linear(lin_in, wl, None if bias_last else bl)
                return ops.ex_res_layernorm_mod(x, y, lin_bias=bl if bias_last else None, gate=gate, gate_mod=(-rows if gate_per_sample else rows),
                                                shift=ln_kw.get("shift"), scale=ln_kw.get("scale"), per_sample=bool(ln_kw.get("per_sample", False)))
            x = ops.ex_linear(lin_in, wl, bl, res=x, gate=gate, gate_mod=(-rows if gate_per_sample else rows), bias_last=bias_last)
            n = ops.ex_layernorm_mod(x, shift=ln_kw.get("shift"), scale=ln_kw.get("scale"), per_sample=bool(ln_kw.get("per_sample", False)))
            return x, n
        split = self._pre(consumer) if split is None else split
        if isinstance(lin_in, ops.SplitAct):
            assert lin_name in self._packed, f"split activation handed to Linear {lin_name!r}, which has no f16x2-split weight"
            w, b = self.w[lin_name + ".weight"], self.w[lin_name + ".bias"]
            x = ops.linear_f16x2_split_residual(lin_in, self._packed[lin_name], b, w.shape[0], x, gate=gate, gate_per_sample=gate_per_sample,
                                                overflow=self.overflow, ksplit=self._ksplit(lin_in.rows, w))
            _, n = ops.residual_ln_mod(x, split=split, overflow=self.overflow, **ln_kw)
            return x, n
        return ops.residual_ln_mod(x, y=self.lin(lin_name, lin_in), gate=gate, gate_per_sample=gate_per_sample, split=split,
                                   overflow=self.overflow, **ln_kw)

    def _pos_bias(self, h: int, w: int) -> torch.Tensor:
        key = (h, w)
        if key not in self._pos_cache:
            pe = self.w["model.pos_embed"]
            top, left = (POS_MAX_DIT - h) // 2, (POS_MAX_DIT - w) // 2
            crop = pe.reshape(POS_MAX_DIT, POS_MAX_DIT, -1)[top:top + h, left:left + w].reshape(h * w, -1)
            self._pos_cache[key] = (crop + self.w["model.x_embedder.proj.bias"]).contiguous()
        return self._pos_cache[key]

    # ---- conditioning ---------------------------------------------------------------------------
    @torch.no_grad()
    def embed_context(self, ehs: torch.Tensor) -> torch.Tensor:
        """context_embedder(ehs) + context_pos_embed  (sd3/mmdit.py:1026) -> [B,K,1536]; step independent"""
        ctx = self.lin("model.context_embedder", ehs.contiguous())
        return ops.add_rows_(ctx, self.context_pos_embed[: ctx.shape[1]].contiguous())

    @torch.no_grad()
    def time_embed(self, t_freq: torch.Tensor) -> torch.Tensor:
        """c = t_embedder.mlp(sinusoid)  (sd3/mmdit.py:177-183, 1022) ; t_freq [B,256]"""
        h = self.lin("model.t_embedder.mlp.0", t_freq)
        return self.lin("model.t_embedder.mlp.2", self._silu(h))

    # ---- the 24 joint blocks + final layer ---------------------------------------------------------
    @torch.no_grad()
    def block0_context_qkv(self, ctx0: torch.Tensor, tables=None) -> torch.Tensor:
        """QKV of the context stream in block 0: LN(ctx0)*(1+scale)+shift -> Linear.  Depends on the tokens only (not on
        x_t or t), so the sampler computes it once per decode instead of once per step (the reference recomputes it)."""
        H = DIT_HIDDEN
        t0 = (tables or self.ctx_tables)[0][: ctx0.shape[1]]
        _, cn = self._ln("model.joint_blocks.0.context_block.attn.qkv", ctx0, shift=t0[:, 0:H], scale=t0[:, H:2 * H])
        return self.lin("model.joint_blocks.0.context_block.attn.qkv", cn)

    @torch.no_grad()
    def modulations(self, c: torch.Tensor, has_ctx: bool = True):
        """the 26 adaLN_modulation Linears of one model evaluation (sd3/mmdit.py:430-470, 641-645): functions of c = t_embedder(t) alone.
        -> ([24 x [B,6H]] image stream, [B,2H] last context block or None, [B,2H] final layer)"""
        sc = self._silu(c)                                   # every adaLN_modulation starts with SiLU(c)
        mods_x = [self.lin(f"model.joint_blocks.{i}.x_block.adaLN_modulation.1", sc) for i in range(DIT_DEPTH)]   # [B,6H]
        mods_c_last = self.lin(f"model.joint_blocks.{DIT_DEPTH - 1}.context_block.adaLN_modulation.1", sc) if has_ctx else None  # [B,2H]
        mods_f = self.lin("model.final_layer.adaLN_modulation.1", sc)                                              # [B,2H]
        return mods_x, mods_c_last, mods_f

    MOD_CACHE_MAX = 256     # entries (one per scheduled timestep and branch; 0.9 MB each); 0: off

    def _step_modulations(self, t_freq: torch.Tensor, t_key):
        """`modulations(time_embed(t_freq))` of a sampler step, remembered per timestep for single-image calls: the reference evaluates
        t_embedder and the adaLN Linears at every step of every call (28 M = 1 GEMMs streaming 1.4 GB of weights: 27 of the 360 ms of a
        one-image decode), but they depend on the scheduled timestep alone.  B == 1 only: there the remembered tensors ARE what the
        step would compute (same shapes, same kernels, bit for bit); at larger B the GEMMs are 0.1 % of the step.  Nothing is stored
        while a stream is capturing (tensors made during a capture belong to the graph's pool); the warm-up pass before a capture
        fills the table, the capture then reads it -- and a capturing caller sets `self._capture_refs = []` first: every remembered
        tensor the capture reads is appended there, and the caller keeps that list with the graph (SelftokPipeline._graphs), so an
        eviction or `_mod_cache.clear()` can never free memory a captured graph still reads.  When the table is full the
        OLDEST entry goes (dict order), not the whole table."""
        key = None if (t_key is None or t_freq.shape[0] != 1 or self.MOD_CACHE_MAX <= 0) else (t_key, self.gemm)
        capturing = t_freq.is_cuda and torch.cuda.is_current_stream_capturing()
        if key is not None and key in self._mod_cache:
            mods = self._mod_cache[key]
            if capturing:
                if self._capture_refs is None:        # a capture nobody announced: the graph must own what it reads
                    return self.modulations(self.time_embed(t_freq), True)
                self._capture_refs.append(mods)
            return mods
        mods = self.modulations(self.time_embed(t_freq), True)
        if key is not None and not capturing:
            while len(self._mod_cache) >= self.MOD_CACHE_MAX:
                self._mod_cache.pop(next(iter(self._mod_cache)))
            self._mod_cache[key] = mods
        return mods

    @torch.no_grad()
    def core(self, xe: torch.Tensor, c: Optional[torch.Tensor], ctx: Optional[torch.Tensor], seg0_sees_seg1: bool = True,
             kvis: Optional[torch.Tensor] = None, cqkv0: Optional[torch.Tensor] = None, tables=None, mods=None) -> torch.Tensor:
        """xe [B,n_x,H] embedded image tokens, c [B,H], ctx [B,n_ctx,H] live context tokens (or None / n_ctx = 0)
        -> FinalLayer output [B,n_x,64] (before unpatchify).  `tables`: per-block context adaLN tables whose row j belongs to context
        row j (default: the position tables; `gather_context` returns the rows of a visibility pattern).  `mods`: the result of
        `modulations(c, ...)` if the caller already has it (c is then unused)."""
        H, NH = DIT_HIDDEN, DIT_HEADS
        B, nx, _ = xe.shape
        n = 0 if ctx is None else ctx.shape[1]
        has_ctx = n > 0
        amode = ops.ATTN_F16X2 if self.gemm == "f16x2" else 0   # 'f16x2': the joint attention runs as split products too
        mods_x, mods_c_last, mods_f = mods if mods is not None else self.modulations(c, has_ctx)
        tab = [t[:n] for t in (tables or self.ctx_tables)]
        x = xe
        blk = "model.joint_blocks.{}.{}_block.{}".format

        def attn_out(rows, consumer, zero=False):   # attention output buffer: split planes if the proj Linear takes them
            if self._pre(consumer) and amode:
                return ops.SplitAct((B, rows, H), x.device, zero=zero)
            return (torch.zeros if zero else torch.empty)(B, rows, H, device=x.device)

        _, xn = self._ln(blk(0, "x", "attn.qkv"), x, shift=mods_x[0][:, 0:H], scale=mods_x[0][:, H:2 * H], per_sample=True)
        if has_ctx and cqkv0 is None:
            _, cn = self._ln(blk(0, "context", "attn.qkv"), ctx, shift=tab[0][:, 0:H], scale=tab[0][:, H:2 * H])
        for i in range(DIT_DEPTH):
            pc, px = f"model.joint_blocks.{i}.context_block", f"model.joint_blocks.{i}.x_block"
            last = i == DIT_DEPTH - 1
            xqkv = self.lin(px + ".attn.qkv", xn)                                  # [B,nx,3H]
            if self.gemm == "exact":
                # the joint attention as ATen's fp32 flash kernel evaluates `attention(q, k, v, heads, mask)` (sd3/other_impls.py:37-45) on the
                # FULL key sequence [K context slots | image tokens] with the prefix mask: the n live context keys keep their positions (kv blocks
                # of 512, MKL's K-blocks of 256 inside), the masked ones contribute exact zeros -- the same bits, not the truncated sequence's
                if kvis is not None:
                    raise NotImplementedError("gemm='exact': one visibility prefix per call (the sampler's case); a per-sample `kvis` needs gemm='fp32' / 'f16x2'")
                xk, xv = xqkv[..., H:2 * H], xqkv[..., 2 * H:]
                if has_ctx:
                    cqkv = cqkv0[:, :n].contiguous() if (i == 0 and cqkv0 is not None) else self.lin(pc + ".attn.qkv", cn)
                    ck, cv = cqkv[..., H:2 * H], cqkv[..., 2 * H:]
                    if not last:
                        oc = ops.ex_attention(cqkv[..., :H], ck, cv, NH, xk if seg0_sees_seg1 else None, xv if seg0_sees_seg1 else None, slots1=self.K)
                    ox = ops.ex_attention(xqkv[..., :H], ck, cv, NH, xk, xv, slots1=self.K)
                else:
                    ox = ops.ex_attention(xqkv[..., :H], None, None, NH, xk, xv, slots1=self.K)        # cfg_inference: every context key masked
            ox = ox if self.gemm == "exact" else attn_out(nx, px + ".attn.proj")
            seg1 = (xqkv[..., :H], xqkv[..., H:2 * H], xqkv[..., 2 * H:], ox)
            if self.gemm == "exact":
                pass
            elif has_ctx:
                # [B,n,3H]; block 0's is step-invariant and may come precomputed (a strided [:, :n] view is fine)
                cqkv = cqkv0[:, :n] if (i == 0 and cqkv0 is not None) else self.lin(pc + ".attn.qkv", cn)
                if last:   # pre_only context block: keys/values only, its attention output is discarded (sd3/mmdit.py:544-547)
                    seg0 = (None, cqkv[..., H:2 * H], cqkv[..., 2 * H:], None)
                else:
                    oc = attn_out(n, pc + ".attn.proj", zero=kvis is not None)
                    seg0 = (cqkv[..., :H], cqkv[..., H:2 * H], cqkv[..., 2 * H:], oc)
                ops.attention(seg0, seg1, NH, 64, kvis=kvis, seg0_sees_seg1=seg0_sees_seg1, mode=amode, overflow=self.overflow)
            else:
                ops.attention(None, seg1, NH, 64, mode=amode, overflow=self.overflow)
            # ---- context stream post-attention (sd3/mmdit.py:485-496, 'pos_emb') ----
            if has_ctx and not last:
                t = tab[i]
                ctx, cn2 = self._res_ln(pc + ".mlp.fc1", ctx, pc + ".attn.proj", oc, gate=t[:, 2 * H:3 * H], gate_per_sample=False,
                                        shift=t[:, 3 * H:4 * H], scale=t[:, 4 * H:5 * H])
                h = self.lin(pc + ".mlp.fc1", cn2, gelu=True, out_split=isinstance(cn2, ops.SplitAct) and self._pre(pc + ".mlp.fc2"))
                nq = blk(i + 1, "context", "attn.qkv")
                if i + 1 < DIT_DEPTH - 1:
                    tn = tab[i + 1]
                    ctx, cn = self._res_ln(nq, ctx, pc + ".mlp.fc2", h, gate=t[:, 5 * H:6 * H], gate_per_sample=False,
                                           shift=tn[:, 0:H], scale=tn[:, H:2 * H])
                else:      # next block is the pre_only one: modulated per sample by c (sd3/mmdit.py:476-483)
                    ctx, cn = self._res_ln(nq, ctx, pc + ".mlp.fc2", h, gate=t[:, 5 * H:6 * H], gate_per_sample=False,
                                           shift=mods_c_last[:, 0:H], scale=mods_c_last[:, H:2 * H], per_sample=True)
            # ---- image stream post-attention ('t_emb') ----
            mx = mods_x[i]
            x, xn2 = self._res_ln(px + ".mlp.fc1", x, px + ".attn.proj", ox, gate=mx[:, 2 * H:3 * H], gate_per_sample=True,
                                  shift=mx[:, 3 * H:4 * H], scale=mx[:, 4 * H:5 * H], per_sample=True)
            h = self.lin(px + ".mlp.fc1", xn2, gelu=True, out_split=isinstance(xn2, ops.SplitAct) and self._pre(px + ".mlp.fc2"))
            if not last:
                mn = mods_x[i + 1]
                x, xn = self._res_ln(blk(i + 1, "x", "attn.qkv"), x, px + ".mlp.fc2", h, gate=mx[:, 5 * H:6 * H], gate_per_sample=True,
                                     shift=mn[:, 0:H], scale=mn[:, H:2 * H], per_sample=True)
            else:          # FinalLayer: LN + modulate(shift, scale = adaLN(c).chunk(2)) + Linear (sd3/mmdit.py:641-645); its Linear takes fp32
                x, xn = self._res_ln(None, x, px + ".mlp.fc2", h, gate=mx[:, 5 * H:6 * H], gate_per_sample=True, split=False,
                                     shift=mods_f[:, 0:H], scale=mods_f[:, H:2 * H], per_sample=True)
            if self._trace is not None:
                self._trace.append(x.clone())
        return self.lin("model.final_layer.linear", xn)

    # ---- reference-shaped entry points ---------------------------------------------------------------
    @torch.no_grad()
    def embed_image(self, x: torch.Tensor) -> torch.Tensor:
        """x_embedder(x) + cropped_pos_embed (sd3/mmdit.py:1000)"""
        B, _, Hh, Ww = x.shape
        if self.gemm == "exact":       # conv k2 s2 = ONE 64-tap chain in (kh, kw, ic) order + bias, then + pos (two roundings): oracle/encoder_exact.c
            patch = ops.patchify(x)[..., self._pe_perm].contiguous()
            return ops.ex_linear(patch, self.pe_w_exact, self.w["model.x_embedder.proj.bias"], res=self._pos_only(Hh // 2, Ww // 2), res_mod=patch.shape[1])
        xe = torch.matmul(ops.patchify(x), self.pe_w)
        return ops.add_rows_(xe, self._pos_bias(Hh // 2, Ww // 2))

    def _pos_only(self, h: int, w: int) -> torch.Tensor:
        key = ("pos", h, w)
        if key not in self._pos_cache:
            pe = self.w["model.pos_embed"]
            top, left = (POS_MAX_DIT - h) // 2, (POS_MAX_DIT - w) // 2
            self._pos_cache[key] = pe.reshape(POS_MAX_DIT, POS_MAX_DIT, -1)[top:top + h, left:left + w].reshape(h * w, -1).contiguous()
        return self._pos_cache[key]

    @torch.no_grad()
    def velocity_tokens(self, x, t_freq, ctx0, n_live: int, context_see_xt: bool = True, cqkv0=None, tables=None, t_key=None):
        """one model evaluation inside the sampler: returns the FinalLayer tokens [B,256,64].  `t_key`: a hashable name of the
        timestep t_freq embeds (the same for every sample), see `_step_modulations`."""
        mods = self._step_modulations(t_freq, t_key)
        ctx = ctx0[:, :n_live].contiguous() if n_live < ctx0.shape[1] else ctx0
        return self.core(self.embed_image(x), None, ctx if n_live > 0 else None, context_see_xt, cqkv0=cqkv0 if n_live > 0 else None,
                         tables=tables, mods=mods)

    @torch.no_grad()
    def gather_context(self, ctx0: torch.Tensor, visible: torch.Tensor = None, index: torch.Tensor = None):
        """A visibility pattern over the context tokens that is not a prefix (the reference's `super_mask`, rectified_flow.py:226-227).
        A masked context token is a key nobody can attend to, and its own row feeds nothing but its keys / values in later blocks
        (the model returns the image stream only): dropping the masked rows is exact.  Returns the visible rows of ctx0 in token order,
        the matching rows of the 23 position tables, and the sorted visible positions -- with them the sampler's step mask
        `arange(K) <= k` is again a PREFIX (of the visible list) and the truncated-context path applies unchanged.
        `index`: the sorted visible positions as a device tensor, instead of the mask (no `nonzero` = no host synchronisation: legal
        inside a hipGraph capture)."""
        idx = index if index is not None else torch.nonzero(visible.to(self.device).reshape(-1).bool())[:, 0]
        return ctx0[:, idx].contiguous(), [t[idx].contiguous() for t in self.ctx_tables], idx

    @torch.no_grad()
    def __call__(self, x=None, t=None, y=None, encoder_hidden_states=None, **kwargs):
        """MMDiT.forward(x, t, y=None, encoder_hidden_states, mask=, context_see_xt=) -> (v [B,16,h,w], drop_ids)
        and MMDiT_Renderer.forward(y=None, encoder_hidden_states=) -> (latent, drop_ids)."""
        ehs = encoder_hidden_states
        B = ehs.shape[0]
        if self.renderer:
            g = int(round(math.sqrt(self.w["model.positional_embedding"].shape[0])))
            xe = (self.w["model.mask_token"].expand(B, g * g, -1) + self.w["model.positional_embedding"]).contiguous()
            if self.gemm == "exact":
                t_freq = self._t1000_exact.expand(B, -1).contiguous()
            else:
                t_freq = sinusoid_host(torch.full((B,), 1000.0)).to(self.device)       # t = ones*1000 (sd3/mmdit.py:1525)
            out = self.core(xe, self.time_embed(t_freq), self.embed_context(ehs), kwargs.get("context_see_xt", False))
            _, v = ops.unpatchify_cfg_euler(out, C=16, hp=g, wp=g)
            return v, torch.zeros(B, dtype=torch.bool)
        mask = kwargs.get("mask", None)
        see = kwargs.get("context_see_xt", False)
        Hh, Ww = x.shape[-2:]
        exact = self.gemm == "exact"
        t_freq = self._t_freq_exact(t) if exact else ops.timestep_embed(t.to(self.device).float(), self.freqs, 1000.0)
        ctx = self.embed_context(ehs)
        kvis = None
        if mask is not None:
            m = mask.to(self.device).bool()
            # the reference's masks are prefixes (arange(K) <= k); anything else is outside the hot path
            cnt = m.sum(dim=1)
            if not bool((m == (torch.arange(m.shape[1], device=m.device)[None] < cnt[:, None])).all()):
                raise NotImplementedError("MMDiTGPU.__call__: `mask` must be a prefix mask (arange(K) <= k, models_ours.py:353); decode other visibility "
                                          "patterns through SelftokPipeline.decoding(super_mask=)")
            if exact:
                # gemm='exact' keeps every context key at its position in the reference's key sequence and takes ONE visibility prefix per call (the
                # sampler's case): a batch-uniform mask is that prefix; a per-sample one would need the per-sample key walk the exact attention does not have
                if not bool((cnt == cnt[0]).all()):
                    raise NotImplementedError("gemm='exact': the mask must be the same prefix for every sample of the call (decode mixed prefixes in groups, or "
                                              "with gemm='fp32' / 'f16x2')")
                n_live = int(cnt[0])
                ctx = ctx[:, :n_live].contiguous() if n_live > 0 else None
            else:
                kvis = (cnt - 1).to(torch.int32).contiguous()
        out = self.core(self.embed_image(x.to(self.device).float()), self.time_embed(t_freq), ctx, see, kvis)
        _, v = ops.unpatchify_cfg_euler(out, C=16, hp=Hh // 2, wp=Ww // 2)
        return v, torch.zeros(B, dtype=torch.bool)

    def _t_freq_exact(self, t: torch.Tensor) -> torch.Tensor:
        """gemm='exact': sinusoidal embedding of t * 1000 with the reference's bits.  The reference evaluates it with torch.cos / sin on the CPU (MKL VML on
        its host class: closed source, not correctly rounded, host dependent), so the rows of the default sampler schedule ship as data
        (data/flow50_t_sincos.npy, tools/oracle/gen_pos_table.py) and are used when t is one of those 50 timesteps; any other t is evaluated with the same
        torch-CPU formula on THIS host -- the reference's bits only on a host of its class (documented deviation of this entry point; the pipeline's own
        sampler always hits the table)."""
        import os
        from .schedule import FlowSchedule
        if getattr(self, "_flow50", None) is None:
            tab = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "flow50_t_sincos.npy"))
            self._flow50 = (FlowSchedule(50, 1.0).scheduled_t.copy(), torch.from_numpy(tab[0].copy()))
        sched, tab = self._flow50
        tc = t.detach().float().cpu()
        rows = sinusoid_host(tc * 1000.0)
        tn = tc.numpy()
        for b in range(tn.shape[0]):
            hit = np.nonzero(sched == tn[b])[0]
            if hit.size:
                rows[b] = tab[int(hit[0])]
        return rows.to(self.device).contiguous()
