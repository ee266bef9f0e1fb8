"""Reconstruction-quality evaluation of a tokenizer checkpoint: images -> `encoding` -> `decoding` (50-step) and / or
`decoding_with_renderer` -> per-image PSNR against the pre-processed original, batch-sharded over the ranks of a torchrun job.

What it reproduces: the PSNR column of the reference's README table (README.md:89-94, 256 x 256: tokenizer_512_ckpt 21.86 dB / with
renderer 24.14 dB, tokenizer_1024_ckpt 23.06 dB / with renderer 26.30 dB) with the reference's own data flow (test.py:24-43: Resize -> CenterCrop ->
NormalizeToTensor -> encoding -> np.save / np.load -> decoding).  The published weights are not reachable offline; the harness is
pinned instead to the reference pipeline's own run on the synthetic weights (tests/golden/pipeline_b16.npz: psnr_ref of 16 images,
tests/test_eval_gpu.py), so the moment a checkpoint is reachable ONE command gives the README's number:

    python tools/eval_psnr.py --images <dir> --yml-path configs/res256/256-eval.yml --pretrained tokenizer_512_ckpt.pth \\
                              --sd3_pretrained <stable-diffusion-3-medium-diffusers> [--renderer-yml ... --renderer-pretrained ...]
"""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import dist as D

IMG_EXT = (".jpg", ".jpeg", ".png", ".bmp", ".webp", ".JPEG", ".JPG", ".PNG")


def list_images(folder: str) -> List[str]:
    """every image file under `folder` (recursive), sorted: the order defines the sharding, so it must be the same on every rank"""
    out = []
    for root, _, files in os.walk(folder):
        out += [os.path.join(root, f) for f in files if f.endswith(IMG_EXT)]
    return sorted(out)


def psnr_each(recon: torch.Tensor, original: torch.Tensor) -> np.ndarray:
    """recon [B,3,H,W] in [0,1] (any float dtype / device), original [B,3,H,W] in [-1,1] -> PSNR in dB per image (fp64 mean of the squared error,
    peak 1.0) -- the arithmetic the reference pipeline run's `psnr_ref` golden was made with"""
    o = (original.detach().float().cpu() + 1.0) / 2.0
    mse = ((recon.detach().float().cpu() - o) ** 2).reshape(o.shape[0], -1).double().mean(dim=1)
    return (10.0 * torch.log10(1.0 / mse)).numpy()


def evaluate(pipe, load_batch: Callable[[int, int], torch.Tensor], n_images: int, batch: int = 64, decoders: Sequence[str] = ("diffusion",),
             noise_fn: Optional[Callable[[int, int], torch.Tensor]] = None, seed: Optional[int] = 1234, renderer_pipe=None, verbose: bool = False) -> Dict:
    """PSNR of `n_images` images through `pipe` (and `renderer_pipe` for the one-step decoder), this rank's contiguous shard of them
    (dist.shard_range) in batches of `batch`; per-image values are gathered to every rank (one all_gather_into_tensor per decoder at the end).
    load_batch(lo, hi) -> float tensor [hi-lo, 3, H, W] in [-1, 1] (host or device).  noise_fn(lo, hi) -> [hi-lo, 16, h, w] replaces the
    reference's `torch.randn` draw (global CPU generator, seeded here with `seed` + rank when given)."""
    world, rank = D.world_size(), D.rank()       # the dist module's view: sharding and the gather below must agree
    lo, hi = D.shard_range(n_images, rank, world)
    counts = [D.shard_range(n_images, r, world)[1] - D.shard_range(n_images, r, world)[0] for r in range(world)]
    if seed is not None:
        torch.manual_seed(int(seed) + rank)
    vals = {d: [] for d in decoders}
    ids_all = []
    for b0 in range(lo, hi, batch):
        b1 = min(b0 + batch, hi)
        imgs = load_batch(b0, b1).to(pipe.device)
        tokens = pipe.encoding(imgs, device=pipe.device)
        ids = tokens.detach().cpu().numpy()                       # the reference round-trips the ids through a host .npy (test.py:38-39)
        ids_all.append(ids)
        for d in decoders:
            if d == "diffusion":
                kw = {} if noise_fn is None else {"noise": noise_fn(b0, b1)}
                rec = pipe.decoding(ids, device=pipe.device, **kw)
            elif d == "renderer":
                rec = (renderer_pipe or pipe).decoding_with_renderer(ids, device=pipe.device)
            else:
                raise ValueError(f"decoder {d!r}: expected 'diffusion' or 'renderer'")
            vals[d].append(psnr_each(rec, imgs))
        if verbose and rank == 0:
            print(f"[eval] images {b0}..{b1 - 1} of shard {lo}..{hi - 1}: " + ", ".join(f"{d} {np.concatenate(vals[d]).mean():.4f} dB so far" for d in decoders), flush=True)
    out = {"images": int(n_images), "ranks": world, "batch": int(batch), "shard": [int(lo), int(hi)]}
    for d in decoders:
        mine = torch.from_numpy(np.concatenate(vals[d]) if vals[d] else np.zeros(0)).to(torch.float64)
        allv = D.all_gather_rows(mine.to(pipe.device), counts).cpu().numpy()      # returns `mine` when there is no exchange to perform
        out[d] = {"psnr_mean_dB": float(allv.mean()) if allv.size else float("nan"), "psnr_each_dB": [round(float(v), 6) for v in allv]}
    out["token_ids_first_image"] = ids_all[0][0, :8].tolist() if ids_all else []
    return out


def folder_loader(paths: Sequence[str], size: int) -> Callable[[int, int], torch.Tensor]:
    from . import preprocess

    def load(lo: int, hi: int) -> torch.Tensor:
        return torch.stack([preprocess.load_image(p, size) for p in paths[lo:hi]])
    return load
