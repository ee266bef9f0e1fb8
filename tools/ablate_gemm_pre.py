"""GPU: timing-only ablations of linear_f16x2_pre_kernel (pre-split activations, ping-pong schedule), built with
-DSELFTOK_GEMM_ABLATE into tools/microbench/libselftok_gemm_ablate.so.  mask: 1 no activation DMA, 4 no weight DMA, 8 no MFMA,
16 no fragment reads, 32 no barriers, 64 no static priority for waves 4-7; 1000 = the single-phase (non-ping-pong) loop."""
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
xs = ops.split_f16x2(a)
flop = 2.0 * M * N * K
names = {0: "full (ping-pong, static priority)", 1000: "full (single-phase loop)", 1: "no activation DMA", 4: "no weight DMA", 5: "no DMA at all",
         8: "no MFMA", 16: "no fragment reads", 21: "MFMA + barriers only", 24: "DMA + barriers only", 32: "no barriers",
         64: "no static priority", 13: "barriers + fragment reads only", 37: "MFMA + fragment reads, no DMA, no barriers"}
for abl in (0, 0, 1000, 64, 1, 4, 5, 16, 21, 8, 24, 13, 32, 37):
    os.environ["SELFTOK_GEMM_ABL"] = str(abl)
    for _ in range(3):
        ops.linear_f16x2_split(xs, packed, b, N)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        ops.linear_f16x2_split(xs, packed, b, N)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    print(f"ABL={abl:4d} {names[abl]:45s} {dt * 1e3:7.3f} ms  ({3 * flop / dt / 1e12:6.0f} TF f16-MFMA-equivalent)", flush=True)
