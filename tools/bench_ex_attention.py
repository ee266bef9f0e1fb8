"""tools: the exact-order joint attention of the MMDiT (B = 64, 24 heads of 64) fused (round 6) vs unfused (round 5) at three points of the sampler
(context keys visible: 512, 358 = the mean, 100), both query streams.  MFMA work of the fused kernel = 3 units (scores twice + P V)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import ops
B, H, D, K = 64, 24, 64, 512
HD = H * D


def t_ms(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for n in (512, 358, 100):
    ctx = torch.randn(B, n, 3 * HD, device="cuda")
    x = torch.randn(B, 256, 3 * HD, device="cuda")
    for name, q in (("context rows", ctx), ("image rows", x)):
        Tq = q.shape[1]
        args = (q[..., :HD], ctx[..., HD:2 * HD], ctx[..., 2 * HD:], H, x[..., HD:2 * HD], x[..., 2 * HD:])
        f = t_ms(lambda: ops.ex_attention(*args, slots1=K, kernel="fused"))
        u = t_ms(lambda: ops.ex_attention(*args, slots1=K, kernel="unfused"))
        fl = 2.0 * B * H * Tq * (n + 256) * D * 2          # scores + P V once
        print(f"visible context {n:3d}, {name:12s} ({Tq:3d} x {n + 256}): fused {f:.3f} ms = {3 * fl / 2 / f / 1e9:.1f} TF of MFMA work ({3 * fl / 2 / f / 1e9 / 157.3:.3f} of the fp32 peak; "
              f"useful {fl / f / 1e9:.1f} TF)   unfused {u:.3f} ms (useful {fl / u / 1e9:.1f} TF)", flush=True)
