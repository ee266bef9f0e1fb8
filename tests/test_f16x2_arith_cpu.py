"""not gpu: the f16x2 split ARITHMETIC (csrc/gemm_split.hip header) emulated in numpy -- independent of any kernel.
x = x0 + x1 2^-11 with x0 = fp16(x), x1 = fp16((x - x0) 2^11);  a.w ~= a0 w0 + 2^-11 (a0 w1 + a1 w0), every product exact in fp32
(11 x 11 significand bits), high and low terms accumulated separately in fp32.  Claims checked: the representation keeps 22
significand bits, the three-term product is at least as close to the exact (fp64) product as a plain fp32 GEMM, values below the
fp16 normal range survive through the scaled low part, and out-of-range operands are detectable (inf in the high part)."""
import numpy as np
import pytest


def split(x):
    x0 = x.astype(np.float16)
    x1 = ((x - x0.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    return x0.astype(np.float32), x1.astype(np.float32)


def split_matmul(a, w):
    a0, a1 = split(a)
    w0, w1 = split(w)
    hi = a0 @ w0.T                              # fp32 accumulate of exact fp16 x fp16 products
    lo = a0 @ w1.T + a1 @ w0.T                  # lives 2^11 larger: loses nothing against the high sum
    return hi + lo * np.float32(1.0 / 2048.0)


def test_split_keeps_22_bits():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-8, 8, 200000))).astype(np.float32)
    x = x[np.abs(x) < 6.0e4]
    x0, x1 = split(x)
    err = np.abs((x0.astype(np.float64) + x1.astype(np.float64) / 2048.0) - x.astype(np.float64))
    normal = np.abs(x) >= 2.0 ** -14              # fp16 normal range of the high part
    assert (err[normal] / np.abs(x[normal].astype(np.float64))).max() <= 2.0 ** -21      # 11 + 11 bits, two roundings
    assert err[~normal].max() <= 2.0 ** -35       # below it: half a subnormal step of the low part, scaled back by 2^-11
    with np.errstate(over="ignore", invalid="ignore"):
        assert np.isinf(split(np.array([7.0e4], np.float32))[0]).all()   # beyond fp16: shows up as inf, i.e. detectable


@pytest.mark.parametrize("M,N,K", [(64, 96, 1536), (48, 64, 6144)])
def test_three_term_product_not_worse_than_fp32_gemm(M, N, K):
    rng = np.random.default_rng(M + N + K)
    a = (rng.standard_normal((M, K)) * (1.0 + 3.0 * rng.random((1, K)))).astype(np.float32)
    w = ((rng.random((N, K)) * 2 - 1) * np.sqrt(3.0 / K)).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    err_split = np.sqrt(np.mean((split_matmul(a, w).astype(np.float64) - ref) ** 2))
    err_fp32 = np.sqrt(np.mean(((a @ w.T).astype(np.float64) - ref) ** 2))
    # the dropped a1 w1 term is <= 2^-22 relative per product and zero-mean; numpy's fp32 GEMM (blocked accumulation) is the yardstick
    assert err_split <= 1.5 * err_fp32 + 1e-9, (err_split, err_fp32)
    assert err_split <= 2.0 ** -20 * np.sqrt(np.mean(ref ** 2)) * 4


def test_tiny_magnitudes_ride_in_the_scaled_low_part():
    rng = np.random.default_rng(3)
    for scale in (1e-3, 1e-5, 1e-7):
        a = (rng.standard_normal((32, 256)) * scale).astype(np.float32)
        w = rng.standard_normal((16, 256)).astype(np.float32)
        ref = a.astype(np.float64) @ w.astype(np.float64).T
        err = np.abs(split_matmul(a, w).astype(np.float64) - ref).max()
        # fp16 flushes |x| < 6e-8 and is subnormal below 6e-5: the x 2^11 low part keeps the product accurate to ~1e-4 relative even at 1e-7
        assert err <= np.abs(ref).max() * (3e-6 if scale >= 1e-5 else 3e-3), (scale, err, np.abs(ref).max())
