"""Platform-independent synthetic data for the Selftok hot path.

There are no published weights or datasets reachable offline, so every parity
test, the golden-vector generator (tools/oracle/) and bench.py draw their
weights / images / noise from the integer-hash generator below.  Only integer
arithmetic and exactly-representable fp32 values are used, so torch-CPU,
torch-ROCm and numpy all produce bit-identical tensors (SURVEY.md section 8d:
"integer hash -> exact fp32 values; do not rely on torch.randn bit-stability").

Nothing here is an oracle: it only manufactures inputs.
"""
from __future__ import annotations

import math
import zlib

import numpy as np
import torch

_M32 = 0xFFFFFFFF


def name_seed(name: str) -> int:
    """Stable 32-bit seed of a tensor name (crc32; same on every platform)."""
    return zlib.crc32(name.encode("utf-8")) & _M32


def _mix32(h: torch.Tensor) -> torch.Tensor:
    # murmur3 finaliser on the low 32 bits of an int64 tensor (wrap-around
    # int64 multiplies keep the low 32 bits exact).
    h = h ^ (h >> 16)
    h = (h * 0x85EBCA6B) & _M32
    h = h ^ (h >> 13)
    h = (h * 0xC2B2AE35) & _M32
    h = h ^ (h >> 16)
    return h


def hash_u32(seed: int, n: int, device="cpu", offset: int = 0) -> torch.Tensor:
    """n pseudo-random 32-bit values (as int64) for indices offset..offset+n-1."""
    idx = torch.arange(offset, offset + n, dtype=torch.int64, device=device)
    h = (idx * 0x9E3779B1 + (seed & _M32)) & _M32
    h = _mix32(h)
    h = (h + 0x7F4A7C15 + ((seed * 0x632BE5AB) & _M32)) & _M32
    return _mix32(h)


def hash_uniform(seed: int, shape, lo=-1.0, hi=1.0, device="cpu") -> torch.Tensor:
    """fp32 tensor ~ U[lo, hi): u = (h>>8)*2^-24 is exact; one rounding in the affine map."""
    n = int(np.prod(shape)) if len(shape) else 1
    out = torch.empty(n, dtype=torch.float32, device=device)
    chunk = (1 << 24) if str(device).startswith("cuda") else (1 << 18)   # cache-resident chunks on CPU (8x faster)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        h = hash_u32(seed, m, device=device, offset=s)
        u = (h >> 8).to(torch.float32) * (1.0 / 16777216.0)
        out[s:s + m] = u
    # (hi-lo)*u and +lo are two individually rounded fp32 ops: identical on CPU/GPU
    out = out * float(np.float32(hi - lo))
    out = out + float(np.float32(lo))
    return out.reshape(shape)


def hash_normalish(seed: int, shape, device="cpu") -> torch.Tensor:
    """Approximately N(0,1) fp32 without transcendentals: Irwin-Hall sum of 4 x 16-bit
    uniforms taken from two hash words, centred and scaled.  All integer until the last
    multiply, hence bit-identical everywhere."""
    n = int(np.prod(shape))
    h0 = hash_u32(seed, n, device=device)
    h1 = hash_u32(seed ^ 0x5BD1E995, n, device=device)
    s = (h0 & 0xFFFF) + (h0 >> 16) + (h1 & 0xFFFF) + (h1 >> 16)  # 0 .. 4*65535
    s = (s - 2 * 65535).to(torch.float32)  # exact integer in fp32
    # var of one 16-bit uniform = (65536^2-1)/12 ; sum of 4
    scale = 1.0 / math.sqrt(4.0 * (65536.0 ** 2 - 1.0) / 12.0)
    return (s * float(np.float32(scale))).reshape(shape)


# ----------------------------------------------------------------------------
# synthetic inputs
# ----------------------------------------------------------------------------

IMAGE_SEED = 0x5E1F70C


def synthetic_images(batch: int, size: int = 256, device="cpu", first_index: int = 0) -> torch.Tensor:
    """[B,3,size,size] fp32 uniform in [-1,1); image i uses seed IMAGE_SEED+first_index+i."""
    imgs = [hash_uniform(IMAGE_SEED + first_index + i, (3, size, size), -1.0, 1.0, device) for i in range(batch)]
    return torch.stack(imgs)


def synthetic_latents(batch: int, device="cpu", first_index: int = 0) -> torch.Tensor:
    """[B,16,32,32] fp32 approx-normal latents (stand-in for process_in(VAE mean))."""
    zs = [hash_normalish(0x1A7E17 + first_index + i, (16, 32, 32), device) for i in range(batch)]
    return torch.stack(zs)


def synthetic_noise(batch: int, latent: int = 32, device="cpu", first_index: int = 0) -> torch.Tensor:
    """[B,16,latent,latent] decode-start noise; hash-normal instead of torch.randn so the
    same noise exists on every box."""
    zs = [hash_normalish(0x2015E + first_index + i, (16, latent, latent), device) for i in range(batch)]
    return torch.stack(zs)


def synthetic_token_ids(batch: int, K: int = 512, codebook_size: int = 32768, first_index: int = 0) -> np.ndarray:
    """[B,K] int64 ids (host numpy, the dtype/device `decoding` takes)."""
    rows = [(hash_u32(0x70CE5 + first_index + i, K) % codebook_size).numpy() for i in range(batch)]
    return np.stack(rows).astype(np.int64)


def synthetic_vq_rows(n: int, dim: int = 16, device="cpu", seed: int = 0xC0DE) -> torch.Tensor:
    """[n,dim] fp32 pre-normalisation encoder features for the VQ micro-benchmark."""
    return hash_normalish(seed, (n, dim), device)
