"""Training-side maintenance of the cosine-similarity code book on MI355X.

Counterpart of `CosineSimCodebook.forward` in training mode, `expire_codes_` / `replace` / `change_code`, `compute_timestep_weight`
and `kmeans` of the reference (mimogpt/models/selftok/vector_quantize_pytorch.py:276-307, 443-451, 479-523, 536-611) for the
tokenizer's single code book (32768 x 16, decay 0.99, smart reactivation over the K token positions).  What differs by design:

  * ids come from the bit-exact HIP argmax (ops.vq_encode) -- no [N, C] score matrix;
  * `bins` / `embed_sum` are scattered from the ids by selftok_vq_ema_accumulate_f32 -- no [N, C] one-hot, no second N x C x D
    contraction (the reference moves 4 B * N * C = 4.3 GB of one-hot per step at N = C = 32768);
  * multi-GPU: `bins` [C] and `embed_sum` [C,16] are all-reduced over RCCL exactly as the reference does (:588, :594), but the dense
    [K, C] `batch_t_p_over_c` all-reduce (:573, 64 MiB per step) is replaced by an all-gather of the token ids (128 KiB per rank):
    every rank applies the identical sparse update, the result is the same mean over the global batch;
  * dead codes are replaced by batch vectors drawn with torch.multinomial on the device (the reference draws with numpy on the
    host, :166-168); the statistics are the same, the random stream is not.

The entropy regularisers of VectorQuantize.forward in training (:1006-1031) are here too, forward AND backward, without the
[B, K, C] probability tensor the reference materialises (csrc/vq_entropy.hip): `softmax_colmean` is a torch.autograd.Function whose
backward is one more fused pass, so `CodebookEMA.entropy_regularisers(z, ...)` returns a diversity loss that back-propagates into
whatever produced z.  The commitment loss / straight-through estimator are plain element-wise torch on [B, K, 16] and need no kernel.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from . import dist as D, ops


def l2norm(t: torch.Tensor) -> torch.Tensor:
    return F.normalize(t, p=2, dim=-1)


class _SoftmaxColMean(torch.autograd.Function):
    """(mean_b softmax_c(scale <l2norm(z[b,k]), e_c>) [K,C], row entropies [B*K]) with a fused backward for the first output
    (csrc/vq_entropy.hip); the row entropies are a logged quantity in the reference (`deterministic_entropy`, :1030) and carry no
    gradient here."""

    @staticmethod
    def forward(ctx, z, codebook, scale):
        rows, colmean = ops.vq_softmax_stats(z, codebook, scale)
        ctx.save_for_backward(z, codebook, rows)
        ctx.scale = scale
        ent = rows[:, 1].clone()
        ctx.mark_non_differentiable(ent)
        return colmean, ent

    @staticmethod
    def backward(ctx, g_colmean, _g_ent):
        z, codebook, rows = ctx.saved_tensors
        # the softmax backward p (g - sum_c p g) does not change when a constant is added to a row of g; an entropy's g = -(1 + log ap)
        # is such a constant plus a small variation: remove the row means first, or A - t m cancels 3-4 digits in fp32
        g = g_colmean - g_colmean.mean(dim=1, keepdim=True)
        return ops.vq_softmax_backward(z, codebook, rows, g, ctx.scale).view_as(z), None, None


def softmax_colmean(z: torch.Tensor, codebook: torch.Tensor, scale: float = 10.0):
    """z [B,K,16] (pre-norm), codebook [C,16] (treated as a constant, `embed.detach()` :559) -> (ap_k [K,C], H(p_n) [B*K])"""
    return _SoftmaxColMean.apply(z, codebook.detach(), float(scale))


class CodebookEMA:
    """state + update rule of the reference's CosineSimCodebook buffers (`embed`, `embed_avg`, `cluster_size`,
    `cluster_size_wo_react`, `timestep_p_over_c`, `tpc_initted`) for one code book, on one GPU of a data-parallel group."""

    def __init__(self, embed: torch.Tensor, K: int, decay: float = 0.99, eps: float = 1e-5, threshold_ema_dead_code: float = 0.2,
                 reset_cluster_size: Optional[float] = 0.2):
        assert embed.is_cuda and embed.dim() == 2 and embed.shape[1] == 16, "code book [C,16] on the GPU"
        self.C, self.K, self.decay, self.eps = embed.shape[0], K, decay, eps
        self.embed = embed.detach().float().contiguous().clone()
        self.embed_avg = self.embed.clone()                                            # :385
        self.cluster_size = torch.zeros(self.C, device=embed.device)
        self.cluster_size_wo_react = torch.zeros(self.C, device=embed.device)
        self.timestep_p_over_c = torch.full((K, self.C), 1.0 / self.C, device=embed.device)   # :387-391
        self.tpc_initted = False
        self.threshold_rel = threshold_ema_dead_code
        self.reset_rel = threshold_ema_dead_code if reset_cluster_size is None else reset_cluster_size   # :371
        self.threshold_abs = self.reset_abs = None                                      # fixed at the first step (:536-542)
        self.delta_embed = torch.zeros((), device=embed.device)
        self._packed = None

    # ---- the step -----------------------------------------------------------------------------------------------------
    def _codebook_packed(self):
        if self._packed is None and self.C % 32 == 0:
            self._packed = ops.vq_pack_codebook(self.embed)
        return self._packed

    @torch.no_grad()
    def step(self, z: torch.Tensor, freeze_codebook: bool = False, generator: Optional[torch.Generator] = None) -> Tuple[torch.Tensor, torch.Tensor, int]:
        """z [B,K,16] = project_in output of this rank's batch (pre-norm, as the eval path).  Returns (quantize [B,K,16], ids [B,K],
        number of re-activated codes).  One training forward of the code book: :544-611."""
        B, K, Dm = z.shape
        assert K == self.K
        world = D.world_size()
        if self.threshold_abs is None:                                                   # relative -> absolute thresholds (:536-542)
            ratio = B * K * world / self.C
            self.threshold_abs, self.reset_abs = ratio * self.threshold_rel, ratio * self.reset_rel
        pk = self._codebook_packed()
        ids = ops.vq_encode(z, pk, packed=True) if pk is not None else ops.vq_encode(z, self.embed)      # argmax, ties -> lowest index
        quantize = ops.code_gather_ln(ids, self.embed)                                  # batched_embedding (:580); eval-mode value
        # timestep_p_over_c: mean one-hot per token position over the GLOBAL batch (:568-578)
        ids_all = D.all_gather_ids(ids)
        w = 1.0 - (self.decay if self.tpc_initted else 0.3)
        ops.vq_tpc_update_(self.timestep_p_over_c, ids_all, w)
        self.tpc_initted = True
        n_react = 0
        if not freeze_codebook:
            bins, embed_sum = ops.vq_ema_accumulate(z, ids, self.C)
            D.all_reduce_sum_(bins)                                                      # :588
            self.cluster_size.lerp_(bins, 1 - self.decay)                                # ema_inplace (:590-591)
            self.cluster_size_wo_react.lerp_(bins, 1 - self.decay)
            D.all_reduce_sum_(embed_sum)                                                 # :594
            self.embed_avg.lerp_(embed_sum, 1 - self.decay)
            tot = self.cluster_size.sum(dim=-1, keepdim=True)
            smoothed = (self.cluster_size + self.eps) / (tot + self.C * self.eps) * tot  # laplace_smoothing * sum (:598)
            embed_normalized = l2norm(self.embed_avg / smoothed[:, None])
            self.delta_embed = F.mse_loss(self.embed, embed_normalized, reduction="sum")
            self.embed = l2norm(embed_normalized).contiguous()                           # :608
            self._packed = None
            n_react = self.expire_codes_(z, generator)
        return quantize, ids, n_react

    # ---- entropy regularisers (:1006-1031) -------------------------------------------------------------------------------
    def entropy_regularisers(self, z: torch.Tensor, diversity_weight: float, smart_re_K: bool = True, ema_entropy_ratio: float = 0.7,
                             reg=(0.25, 0.5)) -> Dict[str, torch.Tensor]:
        """calc_entropy (:89-100) + calc_ema_entropy (:109-118) + the perplexity-ramped weight (:1019-1026) on this rank's batch,
        z [B,K,16] = the features `step` quantised (call it after `step`: the reference reads timestep_p_over_c after that forward's
        update).  `diversity_loss` is differentiable with respect to z; the other entries are the reference's log values.  The
        reference's own call site raises TypeError (`calc_entropy(..., min_ref=...)`, :1008-1010): this is the call without the
        stray keyword."""
        ap_k, row_entropy = softmax_colmean(z, self.embed, 10.0)
        out = {"entropy_to_min": row_entropy.mean()}
        ap = ap_k.mean(dim=0)                                                            # mean over all B*K rows
        out["entropy_to_max"] = -(ap * torch.log(ap)).sum(dim=-1)
        if smart_re_K:
            tpc = self.timestep_p_over_c
            ratio_d = 1.0 - ema_entropy_ratio                                            # :1015
            ema_p = tpc * (1 - ratio_d) + ap_k * ratio_d
            out["codebook_entropy"] = (-(ema_p * torch.log(ema_p)).sum(dim=-1)).mean()
            grp = torch.stack([t.mean(dim=0) for t in ema_p.tensor_split(64, dim=0)], dim=0)
            out["group_entropy"] = (-(grp * torch.log(grp)).sum(dim=-1)).mean()
            entropy = 0.5 * (out["codebook_entropy"] + out["group_entropy"])
            out["perplexity"] = torch.exp(-torch.sum(tpc * torch.log(tpc + 1e-10), dim=-1)).mean()     # get_group_perplexity (:456-459)
            frac = float(out["perplexity"]) / self.C                                     # one host read, as the reference's `if frac < reg[0]`
            w = 0.5 if frac < reg[0] else max(0.5 - 0.5 / (reg[1] - reg[0]) * (frac - reg[0]), 0.0)
            out["codebook_ent_weight"] = torch.tensor(w, device=z.device)
            out["diversity_loss"] = -diversity_weight * w * entropy
        else:
            out["diversity_loss"] = -diversity_weight * out["entropy_to_max"]
        return out

    # ---- dead codes ------------------------------------------------------------------------------------------------------
    def expired_codes(self) -> torch.Tensor:
        return self.cluster_size < self.threshold_abs                                    # :509

    def timestep_weight(self) -> torch.Tensor:
        """compute_timestep_weight (:443-451): positions whose code usage is concentrated get sampled more"""
        ap = self.timestep_p_over_c
        perplexity = torch.exp(-torch.sum(ap * torch.log(ap + 1e-10), dim=-1))
        w = 1 / perplexity
        w = w / w.max() * 10.0
        return w.softmax(dim=-1)

    def change_code(self, indices: torch.Tensor, new_codes: torch.Tensor) -> None:
        self.embed[indices] = new_codes                                                  # :483-486
        self.embed_avg[indices] = new_codes * self.reset_abs
        self.cluster_size[indices] = self.reset_abs
        self._packed = None

    def expire_codes_(self, z: torch.Tensor, generator: Optional[torch.Generator] = None) -> int:
        """expire_codes_ + replace (:488-523): dead codes are replaced by batch vectors sampled with the smart-reactivation weights.
        Data parallel as the reference's sample_vectors_distributed (:249-265): rank r draws its share n // world (+1 for the first
        n % world ranks) from ITS batch and the shares are all-gathered in rank order, so the replacements come from the global
        batch and every rank installs the same vectors.  (The dead-code mask is a function of the all-reduced cluster sizes: the
        same on every rank.)"""
        if self.threshold_rel == 0:
            return 0
        mask = self.expired_codes()
        n = int(mask.sum().item())               # one host read per training step, as the reference's `if not torch.any(...)` (:510)
        if n == 0:
            return 0
        samples = l2norm(z.float()).reshape(-1, z.shape[-1])                             # replace(): batch_samples = l2norm(...)
        b = samples.shape[0] // self.K
        p = (self.timestep_weight() / b)[None, :].expand(b, -1).reshape(-1)              # :491-496
        world, rank = D.world_size(), D.rank()
        shares = [n // world + (1 if r < n % world else 0) for r in range(world)]
        mine = shares[rank]
        if mine > 0:
            pick = torch.multinomial(p, mine, replacement=True, generator=generator)
            local = samples[pick]
        else:
            local = samples[:0]
        new_codes = D.all_gather_rows(local.contiguous(), shares)
        self.change_code(mask.nonzero()[:, 0], new_codes)
        return n

    # ---- k-means initialisation ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def kmeans_iteration(self, samples: torch.Tensor, means: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """one iteration of kmeans(use_cosine_sim=True) (:283-305): assignment by the argmax kernel, means by the scatter kernel.
        samples [n,16] unit-norm on this rank, means [C,16]; bins and sums are all-reduced like `kmeans_all_reduce_fn`."""
        Cm = means.shape[0]
        ids = ops.vq_encode(samples, means.contiguous(), prenormed=True)
        bins, sums = ops.vq_ema_accumulate(samples, ids, Cm, prenormed=True)
        D.all_reduce_sum_(bins)
        zero = bins == 0
        new_means = sums / bins.masked_fill(zero, 1)[:, None]
        D.all_reduce_sum_(new_means)                                                     # the reference all-reduces the per-rank MEANS (:300)
        new_means = l2norm(new_means)
        return torch.where(zero[:, None], means, new_means), bins
