"""oracle/vq_train.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

torch-CPU restatement of the training-side codebook maintenance of the reference's cosine-similarity VQ
(mimogpt/models/selftok/vector_quantize_pytorch.py): the EMA update of cluster sizes / embedding sums and the
re-normalised codebook (:583-611), the per-token-position code statistics `timestep_p_over_c` (:568-578), the dead-code
test of `expire_codes_` (:505-523) with its replacement bookkeeping `change_code` (:479-486), the smart-reactivation
sampling weights `compute_timestep_weight` (:443-451) and one k-means iteration (:276-307).  Single code book (h = 1).
Pinned against the reference's own CosineSimCodebook in training mode by tools/oracle/gen_golden.py vqtrain
(tests/golden/vqtrain.npz).  What is random in the reference (which batch vectors replace dead codes, the k-means seeds) is
an INPUT here.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

State = Dict[str, torch.Tensor]


def l2norm(t: torch.Tensor) -> torch.Tensor:
    return F.normalize(t, p=2, dim=-1)          # vector_quantize_pytorch.py:47-48


def new_state(embed: torch.Tensor, K: int) -> State:
    """buffers of CosineSimCodebook.__init__ (:382-399) for a given initial code book [C,D] and smart_re_K = K"""
    C = embed.shape[0]
    return {"embed": embed.clone(), "embed_avg": embed.clone(), "cluster_size": torch.zeros(C), "cluster_size_wo_react": torch.zeros(C),
            "timestep_p_over_c": torch.ones(K, C) / C, "tpc_initted": torch.tensor(False)}


def scaled_thresholds(threshold: float, reset: Optional[float], batch: int, tokens: int, world: int, C: int):
    """first training call: the relative dead-code threshold becomes absolute (:536-542)"""
    ratio = batch * tokens * world / C
    return ratio * threshold, ratio * (threshold if reset is None else reset)


def train_step(st: State, x: torch.Tensor, decay: float, eps: float = 1e-5, world: int = 1, all_reduce=None) -> torch.Tensor:
    """one CosineSimCodebook.forward in training mode up to (not including) expire_codes_ (:544-611).
    x [B,K,D] is already l2-normalised (VectorQuantize.forward :854).  Updates `st` in place, returns ids [B,K].
    `all_reduce(t)` sums t over ranks in place (distributed.all_reduce, :573,:588,:594); None = single rank."""
    B, K, D = x.shape
    C = st["embed"].shape[0]
    flat = x.reshape(-1, D).float()
    dist = flat @ st["embed"].t()                                        # :561
    ids = dist.argmax(dim=-1)                                            # gumbel_sample, stochastic=False (:125-136)
    onehot = F.one_hot(ids, C).to(flat.dtype)
    # ---- timestep_p_over_c (:568-578) ----
    batch_tpc = onehot.reshape(B, K, C).mean(dim=0)
    if all_reduce is not None:
        all_reduce(batch_tpc)
    batch_tpc = batch_tpc / world
    d = decay if bool(st["tpc_initted"]) else 0.3
    st["timestep_p_over_c"].lerp_(batch_tpc, 1 - d)                      # ema_inplace (:66-72)
    st["tpc_initted"] = torch.tensor(True)
    # ---- EMA of cluster sizes and embedding sums, re-normalised code book (:583-608) ----
    bins = onehot.sum(dim=0)
    if all_reduce is not None:
        all_reduce(bins)
    st["cluster_size"].lerp_(bins, 1 - decay)
    st["cluster_size_wo_react"].lerp_(bins, 1 - decay)
    embed_sum = (flat.t() @ onehot).t().contiguous()                     # einsum('h n d, h n c -> h c d')
    if all_reduce is not None:
        all_reduce(embed_sum)
    st["embed_avg"].lerp_(embed_sum, 1 - decay)
    cs = st["cluster_size"]
    smoothed = (cs + eps) / (cs.sum(dim=-1, keepdim=True) + C * eps) * cs.sum(dim=-1, keepdim=True)   # laplace_smoothing (:161-163) * sum
    embed_normalized = l2norm(st["embed_avg"] / smoothed[:, None])
    st["delta_embed"] = F.mse_loss(st["embed"], embed_normalized, reduction="sum")
    st["embed"] = l2norm(embed_normalized)
    return ids.reshape(B, K)


def expired_codes(st: State, threshold_abs: float) -> torch.Tensor:
    """expire_codes_ (:505-515): codes whose EMA cluster size fell below the absolute threshold"""
    return st["cluster_size"] < threshold_abs


def timestep_weight(st: State) -> torch.Tensor:
    """compute_timestep_weight (:443-451): sampling weights over the K token positions for smart reactivation"""
    ap = st["timestep_p_over_c"]
    perplexity = torch.exp(-torch.sum(ap * torch.log(ap + 1e-10), dim=-1))
    w = 1 / perplexity
    w = w / w.max() * 10.0
    return w.softmax(dim=-1)


def change_code(st: State, indices: torch.Tensor, new_codes: torch.Tensor, reset_abs: float) -> None:
    """change_code (:479-486): install replacement vectors for dead codes"""
    st["embed"][indices] = new_codes
    st["embed_avg"][indices] = new_codes * reset_abs
    st["cluster_size"][indices] = reset_abs


def kmeans_iteration(samples: torch.Tensor, means: torch.Tensor, all_reduce=None):
    """one iteration of kmeans(..., use_cosine_sim=True) (:283-305) for one code book: samples [n,D], means [C,D]"""
    C = means.shape[0]
    buckets = (samples @ means.t()).argmax(dim=-1)
    bins = torch.zeros(C, dtype=torch.int64).scatter_add_(0, buckets, torch.ones_like(buckets))
    if all_reduce is not None:
        all_reduce(bins)
    zero = bins == 0
    new_means = torch.zeros_like(means).index_add_(0, buckets, samples)
    new_means = new_means / bins.masked_fill(zero, 1)[:, None]
    if all_reduce is not None:
        all_reduce(new_means)
    new_means = l2norm(new_means)
    return torch.where(zero[:, None], means, new_means), bins


def entropy_terms(z: torch.Tensor, embed: torch.Tensor, tpc: torch.Tensor, diversity_weight: float, smart_re_K: bool = True,
                  ema_entropy_ratio: float = 0.7, reg=(0.25, 0.5)) -> Dict[str, torch.Tensor]:
    """the entropy regularisers of VectorQuantize.forward in training (:1006-1031), on the materialised [1, B, K, C] score tensor as
    the reference computes them: z [B,K,D] pre-norm (autograd-capable), embed [C,D] (detached, :559), tpc = timestep_p_over_c [K,C].
      calc_entropy (:89-100) on the 10x scaled scores of all B*K rows: entropy_to_max = H(mean_n p), entropy_to_min = mean_n H(p_n)
      calc_ema_entropy (:109-118): ema_p = tpc (1 - ratio_d) + mean_b p ratio_d with ratio_d = 1 - ema_entropy_ratio (:1015);
          entropy of every token position, and of the means over 64 groups of positions (tensor_split(64))
      get_group_perplexity (:456-459), the weight ramp between reg[0] and reg[1] (:1021-1024), diversity_loss (:1026 / :1028).
    The reference's own call passes `min_ref=` to calc_entropy, which takes no such argument (TypeError, :1008-1010): the value
    restated is the call without it."""
    C = embed.shape[0]
    distances = torch.einsum("h n d, h c d -> h n c", l2norm(z.float()).reshape(1, -1, z.shape[-1]), embed.detach()[None]).reshape(1, *z.shape[:2], C)
    scaled = distances * 10.0
    p = scaled.flatten(end_dim=-2).softmax(dim=-1)
    ap = p.mean(dim=0)
    out = {"entropy_to_max": -(ap * torch.log(ap)).sum(dim=-1), "entropy_to_min": (-(p * torch.log(p)).sum(dim=-1)).mean()}
    if smart_re_K:
        ratio_d = 1.0 - ema_entropy_ratio
        apk = scaled.softmax(dim=-1)[0].mean(dim=0)
        ema_p = tpc * (1 - ratio_d) + apk * ratio_d
        out["codebook_entropy"] = (-(ema_p * torch.log(ema_p)).sum(dim=-1)).mean()
        grp = torch.stack([t.mean(dim=0) for t in ema_p.tensor_split(64, dim=0)], dim=0)
        out["group_entropy"] = (-(grp * torch.log(grp)).sum(dim=-1)).mean()
        entropy = 0.5 * (out["codebook_entropy"] + out["group_entropy"])
        out["perplexity"] = torch.exp(-torch.sum(tpc * torch.log(tpc + 1e-10), dim=-1)).mean()
        frac = float(out["perplexity"]) / C
        w = 0.5 if frac < reg[0] else max(0.5 - 0.5 / (reg[1] - reg[0]) * (frac - reg[0]), 0.0)
        out["codebook_ent_weight"] = torch.tensor(w)
        out["diversity_loss"] = -diversity_weight * w * entropy
    else:
        out["diversity_loss"] = -diversity_weight * out["entropy_to_max"]
    return out
