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

The differentiable part of the reference's VectorQuantize.forward in training (commitment / entropy losses, straight-through
estimator) needs autograd through the encoder and stays outside this inference-centred hot path.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from . import dist as D, ops


def l2norm(t: torch.Tensor) -> torch.Tensor:
    return F.normalize(t, p=2, dim=-1)


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
