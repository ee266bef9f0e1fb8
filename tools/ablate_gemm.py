"""GPU: timing-only ablations of linear_f16x2_kernel (tools/microbench/libselftok_gemm_ablate.so, built with
-DSELFTOK_GEMM_ABLATE): which part of the k-loop bounds it?  mask: 1 no split/ds_write, 2 no row loads, 4 no weight DMA,
8 no MFMA, 16 no fragment reads."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SELFTOK_HIP_LIB"] = os.path.join(ROOT, "tools", "microbench", "libselftok_gemm_ablate.so")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from selftoktokenizer_amd import ops  # noqa: E402

M, N, K = 22912, 4608, 1536
a = torch.randn(M, K, device="cuda")
w = torch.randn(N, K, device="cuda") * 0.02
b = torch.randn(N, device="cuda")
packed = ops.linear_f16x2_pack(w)
flop = 2.0 * M * N * K
names = {0: "full", 1: "no split/ds_write", 2: "no row loads", 3: "no split, no row loads", 4: "no weight DMA", 7: "no staging at all",
         8: "no MFMA", 16: "no fragment reads", 24: "no MFMA, no fragment reads (staging only)", 23: "MFMA only", 31: "empty loop"}
for abl in (0, 1, 2, 3, 4, 7, 16, 23, 8, 24, 31):
    os.environ["SELFTOK_GEMM_ABL"] = str(abl)
    for _ in range(3):
        ops.linear_f16x2(a, packed, b, N)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        ops.linear_f16x2(a, packed, b, N)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    print(f"ABL={abl:2d} {names[abl]:45s} {dt * 1e3:7.3f} ms  ({3 * flop / dt / 1e12:6.0f} TF f16-MFMA-equivalent)", flush=True)
