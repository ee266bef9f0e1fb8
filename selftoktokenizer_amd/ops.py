"""Thin torch-tensor front end of the C ABI (include/selftok_hip.h).

Every function takes CUDA(=HIP) tensors, passes `data_ptr()`s and the current torch stream to
libselftok_hip.so, and returns torch tensors.  PyTorch is only the allocator / stream owner here.
"""
from __future__ import annotations

import torch

from . import _lib

IDS_I32 = 1
PRENORMED = 2


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.SelftokHipError("selftok HIP ops need device tensors (there is no CPU fallback)")


def _p(t):
    return None if t is None else t.data_ptr()


def vq_pack_codebook(codebook: torch.Tensor) -> torch.Tensor:
    """[C,16] fp32 -> MFMA-fragment-ordered copy (one-time, the codebook is a constant)."""
    _need_cuda(codebook)
    cb = codebook.contiguous().float()
    C, D = cb.shape
    packed = torch.empty_like(cb)
    _lib.check(_lib.load().selftok_vq_pack_codebook(_p(cb), _p(packed), C, D, _stream()), "selftok_vq_pack_codebook")
    return packed


def vq_encode(z: torch.Tensor, codebook: torch.Tensor, *, packed: bool = False, return_best: bool = False,
              ids_dtype=torch.int64, prenormed: bool = False):
    """z [...,16] fp32 (pre-norm) , codebook [C,16] (raw, or packed if packed=True) -> ids [...]"""
    _need_cuda(z, codebook)
    lib = _lib.load()
    zz = z.contiguous().float().reshape(-1, z.shape[-1])
    N, D = zz.shape
    C = codebook.shape[0]
    ids = torch.empty(N, dtype=ids_dtype, device=z.device)
    best = torch.empty(N, dtype=torch.float32, device=z.device) if return_best else None
    ws = torch.empty(lib.selftok_vq_workspace_bytes(N, C), dtype=torch.uint8, device=z.device)
    flags = (IDS_I32 if ids_dtype == torch.int32 else 0) | (PRENORMED if prenormed else 0)
    fn = lib.selftok_vq_encode_packed_f32 if packed else lib.selftok_vq_encode_f32
    _lib.check(fn(_p(zz), _p(codebook), _p(ids), _p(best), _p(ws), N, C, D, flags, _stream()),
               "selftok_vq_encode_packed_f32" if packed else "selftok_vq_encode_f32")
    ids = ids.reshape(z.shape[:-1])
    if return_best:
        return ids, best.reshape(z.shape[:-1])
    return ids


def code_gather_ln(ids: torch.Tensor, codebook: torch.Tensor, ln_w=None, ln_b=None, eps: float = 1e-6) -> torch.Tensor:
    """ids [...] (int64/int32) -> LayerNorm16(codebook[ids]) [...,16]"""
    _need_cuda(ids, codebook)
    flat = ids.contiguous().reshape(-1)
    n = flat.numel()
    C, D = codebook.shape
    out = torch.empty(n, D, dtype=torch.float32, device=ids.device)
    flags = IDS_I32 if flat.dtype == torch.int32 else 0
    _lib.check(_lib.load().selftok_code_gather_ln_f32(_p(flat), _p(codebook), _p(ln_w), _p(ln_b), _p(out), n, C, D,
                                                      eps, flags, _stream()), "selftok_code_gather_ln_f32")
    return out.reshape(*ids.shape, D)
