"""GPU micro-benchmark: f16x2-split Linear (csrc/gemm_split.hip) vs the fp32 library GEMM at the MMDiT shapes
(B = 64: context rows 64 x 358 (mean live tokens), image rows 64 x 256)."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import ops  # noqa: E402


def bench(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def main():
    torch.manual_seed(0)
    shapes = [(64 * 358, 4608, 1536), (64 * 358, 1536, 1536), (64 * 358, 6144, 1536), (64 * 358, 1536, 6144),
              (64 * 256, 4608, 1536), (64 * 256, 6144, 1536), (64 * 256, 1536, 6144), (64 * 512, 4608, 1536)]
    if len(sys.argv) > 1:
        shapes = shapes[: int(sys.argv[1])]
    for M, N, K in shapes:
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.02
        b = torch.randn(N, device="cuda")
        packed = ops.linear_f16x2_pack(w)
        flop = 2.0 * M * N * K
        t_lib = bench(lambda: F.linear(a, w, b))
        t_s = bench(lambda: ops.linear_f16x2(a, packed, b, N))
        a_s = ops.split_f16x2(a)
        same = torch.equal(ops.linear_f16x2_split(a_s, packed, b, N), ops.linear_f16x2(a, packed, b, N))
        t_p = bench(lambda: ops.linear_f16x2_split(a_s, packed, b, N))
        t_po = bench(lambda: ops.linear_f16x2_split(a_s, packed, b, N, gelu=True, out_split=True))
        print(f"M={M} N={N} K={K}: fp32 library {t_lib * 1e3:.3f} ms ({flop / t_lib / 1e12:.0f} TF) | f16x2 split {t_s * 1e3:.3f} ms "
              f"({flop / t_s / 1e12:.0f} TF-equiv, {3 * flop / t_s / 1e12:.0f} TF f16 MFMA = {3 * flop / t_s / 2.5e15:.2f} of peak) | x{t_lib / t_s:.2f}"
              f" | pre-split A {t_p * 1e3:.3f} ms ({3 * flop / t_p / 2.5e15:.2f} of peak, bit-equal {same}), +GELU+split out {t_po * 1e3:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
