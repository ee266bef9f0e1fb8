"""hipBLASLt fp32 GEMM rates for the MMDiT shapes: F.linear (weight [N,K]) vs mm with a pre-transposed [K,N] weight."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
def t(f, n=8):
    for _ in range(3): f()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
for M in (64 * 256, 64 * 358, 64 * 512, 768):
    for name, N, K in (("qkv", 4608, 1536), ("proj", 1536, 1536), ("fc1", 6144, 1536), ("fc2", 1536, 6144)):
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02; b = torch.randn(N, device="cuda")
        wt = w.t().contiguous()
        fl = 2.0 * M * N * K
        a = t(lambda: F.linear(x, w, b)); c = t(lambda: torch.addmm(b, x, wt))
        print(json.dumps({"M": M, "op": name, "N": N, "K": K, "linear_TF": round(fl / a / 1e9, 1), "mm_pretransposed_TF": round(fl / c / 1e9, 1)}), flush=True)
