"""The fp32 block Linears of the MMDiT step through PyTorch-ROCm (hipBLASLt): achieved TFLOP/s per shape, default heuristic against
PyTorch's TunableOp selection (PYTORCH_TUNABLEOP_ENABLED=1 in a second process).  Usage: python tools/bench_fp32_gemm_shapes.py [rows...]"""
import os
import sys

import torch
import torch.nn.functional as F


def ms(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


rows = [int(v) for v in sys.argv[1:]] or [16384, 22976, 64 * 100, 64 * 500]
H = 1536
print("tunableop:", os.environ.get("PYTORCH_TUNABLEOP_ENABLED", "0"), flush=True)
tot = {}
for M in rows:
    a1, a4 = torch.randn(M, H, device="cuda"), torch.randn(M, 4 * H, device="cuda")
    for name, a, N, K in (("qkv", a1, 3 * H, H), ("proj", a1, H, H), ("fc1", a1, 4 * H, H), ("fc2", a4, H, 4 * H)):
        w, b = torch.randn(N, K, device="cuda") * 0.02, torch.randn(N, device="cuda")
        t = ms(lambda: F.linear(a, w, b))
        fl = 2.0 * M * N * K
        tot[M] = tot.get(M, 0.0) + t
        print(f"M={M:6d} {name:4s} [{M},{K}]x[{K},{N}]: {t:7.3f} ms  {fl / t * 1e-9:6.1f} TFLOP/s  ({fl / t * 1e-9 / 157.3:.3f} of the fp32 matrix peak)", flush=True)
print("sum per M (ms):", {k: round(v, 3) for k, v in tot.items()})
