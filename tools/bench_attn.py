"""Time the joint attention kernel at the C2 shape (B=64, 24 heads, n context + 256 image tokens)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selftoktokenizer_amd import ops
B, H = 64, 24
D = H * 64
for n, mode in [(n, m) for n in (512, 358, 20) for m in (0, ops.ATTN_F16X2)]:
    ctx = torch.randn(B, n, 3 * D, device="cuda"); xs = torch.randn(B, 256, 3 * D, device="cuda")
    oc = torch.empty(B, n, D, device="cuda"); ox = torch.empty(B, 256, D, device="cuda")
    f = lambda: ops.attention((ctx[..., :D], ctx[..., D:2*D], ctx[..., 2*D:], oc), (xs[..., :D], xs[..., D:2*D], xs[..., 2*D:], ox), H, 64, mode=mode)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    S = n + 256
    print(json.dumps({"mode": "f16x2" if mode else "fp32-mfma", "n_ctx": n, "ms": round(ms, 4), "TFLOPs": round(4.0 * B * H * S * S * 64 / ms / 1e9, 1)}))
