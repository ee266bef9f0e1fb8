"""workload for tools/pmc_sq.sh: the exact-order fp32 GEMM at the MMDiT's qkv / fc2 shapes (5 launches each)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import ops
for M, K, N in ((22912, 1536, 4608), (16384, 6144, 1536)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
    for _ in range(5):
        ops.ex_linear(x, w, b)
torch.cuda.synchronize()
