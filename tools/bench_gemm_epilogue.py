"""GPU: cost of the epilogue variants of the f16x2 Linear kernels (GELU, split-activation outputs) at the fc1 shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selftoktokenizer_amd import ops
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
M, N, K = 22912, 6144, 1536
a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02; b = torch.randn(N, device="cuda")
packed = ops.linear_f16x2_pack(w); xs = ops.split_f16x2(a)
for rep in range(2):
    for gelu in (False, True):
        for osplit in (False, True):
            t = bench(lambda: ops.linear_f16x2_split(xs, packed, b, N, gelu=gelu, out_split=osplit))
            print(f"pre-split kernel gelu={gelu} out_split={osplit}: {t*1e3:.3f} ms", flush=True)
    for gelu in (False, True):
        t = bench(lambda: ops.linear_f16x2(a, packed, b, N, gelu=gelu))
        print(f"fp32-A kernel gelu={gelu}: {t*1e3:.3f} ms", flush=True)
