"""workload for tools/pmc_sgemm.sh: the three fp32 GEMMs at the MMDiT's qkv shape with whole rounds of tiles (M = 16384), 5 launches each:
hipBLASLt (F.linear, kernel Cijk_...), xe_gemm128_kernel (ex_linear) and sg_gemm_kernel (linear_f32, free order + MKL order)"""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import ops
M, K, N = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (16384, 1536, 4608)
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
for _ in range(5):
    F.linear(x, w, b)
for _ in range(5):
    ops.ex_linear(x, w, b, out=out, kernel="xe")
for _ in range(5):
    ops.linear_f32(x, w, b, out=out)
for _ in range(5):
    ops.linear_f32(x, w, b, out=out, mkl_order=True)
torch.cuda.synchronize()
