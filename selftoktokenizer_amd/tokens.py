"""Token file / wire formats either side of the hot path.

The reference hands tokens around as `np.save` of an int64 `[B,K]` array (test.py:38-39) -- 8 bytes per 15-bit id.
Ids are < 32768 = 2^15, so they also fit uint16 (4x smaller) or a dense 15-bit stream (4.27x smaller); both are
lossless and round-trip to the reference's int64 layout.  `reverse_for_ar` implements the README's note for AR
training ("decode the sequence reversely", README.md:241): tokens are ordered from the most detailed (index 0,
visible only at small t) to the coarsest, an AR model consumes them in reverse.
"""
from __future__ import annotations

import numpy as np

CODEBOOK_BITS = 15


def save_reference_npy(path: str, tokens) -> None:
    """exactly what the reference script writes: int64 [B,K] .npy"""
    np.save(path, np.asarray(tokens, dtype=np.int64))


def load_reference_npy(path: str) -> np.ndarray:
    t = np.load(path)
    if t.dtype != np.int64:
        t = t.astype(np.int64)
    return t


def to_uint16(tokens) -> np.ndarray:
    t = np.asarray(tokens)
    if t.size and (t.min() < 0 or t.max() >= (1 << 16)):
        raise ValueError("token id out of uint16 range")
    return t.astype(np.uint16)


def pack15(tokens) -> bytes:
    """dense little-endian bit stream, 15 bits per id, row-major; header-less (shape travels separately)."""
    t = np.asarray(tokens, dtype=np.int64).reshape(-1)
    if t.size and (t.min() < 0 or t.max() >= (1 << CODEBOOK_BITS)):
        raise ValueError("token id does not fit 15 bits")
    bits = ((t[:, None] >> np.arange(CODEBOOK_BITS)) & 1).astype(np.uint8).reshape(-1)
    return np.packbits(bits, bitorder="little").tobytes()


def unpack15(buf: bytes, shape) -> np.ndarray:
    n = int(np.prod(shape))
    bits = np.unpackbits(np.frombuffer(buf, dtype=np.uint8), bitorder="little")[: n * CODEBOOK_BITS]
    vals = (bits.reshape(n, CODEBOOK_BITS).astype(np.int64) << np.arange(CODEBOOK_BITS)).sum(axis=1)
    return vals.reshape(shape)


def reverse_for_ar(tokens) -> np.ndarray:
    """[B,K] -> [B,K] with the token axis reversed (coarse-to-fine order for AR models)."""
    return np.ascontiguousarray(np.asarray(tokens)[:, ::-1])


def to_ar_order(tokens) -> np.ndarray:
    """tokenizer order -> the order an AR model is trained on / emits (README.md:241: reversed)"""
    return reverse_for_ar(tokens)


def from_ar_order(tokens) -> np.ndarray:
    """what an AR model emitted (coarse -> fine) -> tokenizer order, ready for `SelftokPipeline.decoding`"""
    return reverse_for_ar(tokens)


def pad_prefix(prefix, K: int, fill: int = 0):
    """[B,k] tokenizer-order prefix (k <= K) -> (int64 [B,K] padded with `fill`, k): the arguments of
    `SelftokPipeline.decoding(idx, prefix_k=k)`.  The padding ids are never visible (mask * super_mask)."""
    t = np.asarray(prefix)
    if t.ndim != 2 or t.shape[1] > K:
        raise ValueError(f"prefix must be [B,k] with k <= {K}")
    out = np.full((t.shape[0], K), fill, dtype=np.int64)
    out[:, : t.shape[1]] = t
    return out, int(t.shape[1])


def prefix_mask(K: int, k) -> np.ndarray:
    """visibility mask of a partial decode that uses only tokens 0..k (reference get_encoder_mask, models_ours.py:345-353)"""
    k = np.asarray(k).reshape(-1, 1)
    return np.arange(K)[None, :] <= k
