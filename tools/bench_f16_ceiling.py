"""GPU: what the vendor library sustains on the f16 matrix cores of THIS box with random data (the practical, power-limited ceiling the
f16x2 kernels are up against): plain fp16 / bf16 GEMMs with the same number of MFMAs as the f16x2 Linear at the MMDiT shapes
(K tripled: the split product issues 3 MFMAs per fp32 product)."""
import os
import subprocess
import sys
import time

import torch

def bench(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n

try:
    print(subprocess.run(["rocm-smi", "--showpower", "--showmaxpower", "--showclocks"], capture_output=True, text=True, timeout=60).stdout[-1500:])
except Exception as e:  # noqa: BLE001
    print("rocm-smi:", e)
for dt in (torch.float16, torch.bfloat16):
    for zero in (False, True):
        for M, N, K in ((22912, 4608, 3 * 1536), (22912, 6144, 3 * 1536), (16384, 1536, 3 * 6144), (8192, 8192, 8192)):
            a = (torch.zeros if zero else torch.randn)(M, K, device="cuda", dtype=dt)
            w = (torch.zeros if zero else torch.randn)(N, K, device="cuda", dtype=dt)
            t = bench(lambda: torch.nn.functional.linear(a, w))
            print(f"{str(dt):15s} {'zeros ' if zero else 'random'} M={M} N={N} K={K}: {t * 1e3:.3f} ms = {2.0 * M * N * K / t / 1e12:.0f} TF ({2.0 * M * N * K / t / 2.5e15:.2f} of 2.5 PF)", flush=True)
