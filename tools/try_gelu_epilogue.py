import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from selftoktokenizer_amd import ops, synth
M, K, N = 64 * 358, 1536, 6144
x = synth.hash_uniform(1, (M, K), -2, 2, "cuda"); w = synth.hash_uniform(2, (N, K), -0.04, 0.04, "cuda"); b = synth.hash_uniform(3, (N,), -0.1, 0.1, "cuda")
ref = F.gelu(F.linear(x.double(), w.double(), b.double()), approximate="tanh").float()
def mine():
    h = torch.matmul(x, w.t()); ops.bias_gelu_(h, b); return h
def fused():
    return torch._addmm_activation(b, x, w.t(), use_gelu=True)
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
a, c = mine(), fused()
print(json.dumps({"mine_err": float((a - ref).abs().max()), "fused_err": float((c - ref).abs().max()), "mine_vs_fused": float((a - c).abs().max()),
                  "mine_ms": round(t(mine), 3), "fused_ms": round(t(fused), 3), "gemm_only_ms": round(t(lambda: F.linear(x, w, b)), 3)}))
