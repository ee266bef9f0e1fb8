import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
K = 1536
for M in (16384, 22912, 32768, 64 * 100):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(4608, K, device="cuda") * 0.02; b = torch.randn(4608, device="cuda")
    ws = [w[i * 1536:(i + 1) * 1536] for i in range(3)]; bs = [b[i * 1536:(i + 1) * 1536] for i in range(3)]
    fused = t(lambda: F.linear(x, w, b))
    split = t(lambda: [F.linear(x, ws[i], bs[i]) for i in range(3)])
    two = t(lambda: (F.linear(x, w[:3072], b[:3072]), F.linear(x, w[3072:], b[3072:])))
    fl = 2.0 * M * 4608 * K
    print(json.dumps({"M": M, "fused_ms": round(fused, 3), "split3_ms": round(split, 3), "split2_ms": round(two, 3),
                      "fused_TF": round(fl / fused / 1e9, 1), "split3_TF": round(fl / split / 1e9, 1), "split2_TF": round(fl / two / 1e9, 1)}))
