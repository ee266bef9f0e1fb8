"""the exact-order fp32 GEMM (selftok_ex_linear_f32, csrc/encoder_exact.hip) at the MMDiT's and the Q-Former's Linear shapes: TFLOP/s and fraction of the fp32
matrix peak (157.3), next to hipBLASLt's fp32 GEMM on the same operands.  Usage (GPU box): python tools/bench_ex_gemm.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import ops


def ms(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for name, M, K, N in (("qkv  [64*358] 1536->4608", 22912, 1536, 4608), ("proj [64*358] 1536->1536", 22912, 1536, 1536), ("fc1  [64*256] 1536->6144", 16384, 1536, 6144),
                      ("fc2  [64*256] 6144->1536", 16384, 6144, 1536), ("enc query_linear [64*512] 512->1536", 32768, 512, 1536), ("enc q_mlp.fc2 [64*512] 2048->512", 32768, 2048, 512),
                      ("enc to_query_kv [64*256] 64->1024", 16384, 64, 1024)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
    t = ms(lambda: ops.ex_linear(x, w, b))
    tl = ms(lambda: torch.nn.functional.linear(x, w, b))
    fl = 2.0 * M * N * K
    print(f"{name:40s} exact {t:8.3f} ms {fl / t * 1e-9:6.1f} TF ({fl / t * 1e-9 / 157.3:.3f})   hipBLASLt {tl:8.3f} ms {fl / tl * 1e-9:6.1f} TF ({fl / tl * 1e-9 / 157.3:.3f})", flush=True)
