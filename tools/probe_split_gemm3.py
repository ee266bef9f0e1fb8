"""GPU probe 3: f16x2 split with ONE accumulator (natural-scale residuals, weights pre-multiplied by 2^10) -- would allow 128x64
per-wave tiles at 128 accumulator registers.  Emulated through hipBLASLt (K-concatenated fp16 operands, fp32 accumulate)."""
import torch

torch.manual_seed(0)
dev = "cuda"
for K in (1536, 6144):
    for kind, sc in (("ln", 1.0), ("small", 0.05)):
        a = torch.randn(2048, K, device=dev) * (1.0 + 3.0 * torch.rand(1, K, device=dev)) * sc
        w = (torch.rand(K, 1536, device=dev) * 2 - 1) * (3.0 / K) ** 0.5
        ref = a.double() @ w.double()
        e32 = (a @ w).double() - ref
        for a_scale in (1.0, 16.0):
            as_ = a * a_scale
            ws = w * 1024.0
            a0 = as_.half(); a1 = (as_ - a0.float()).half()
            w0 = ws.half(); w1 = (ws - w0.float()).half()
            one = torch.mm(torch.cat([a0, a0, a1], 1), torch.cat([w0, w1, w0], 0), out_dtype=torch.float32) / (1024.0 * a_scale)
            hi = torch.mm(a0, w0, out_dtype=torch.float32)
            lo = torch.mm(torch.cat([a0, a1], 1), torch.cat([w1, w0], 0), out_dtype=torch.float32)
            two = (hi + lo) / (1024.0 * a_scale)
            e1, e2 = one.double() - ref, two.double() - ref
            print(f"K={K} {kind} a_scale={a_scale:g}: fp32 lib rms {float(e32.pow(2).mean().sqrt()):.3e} | one accumulator rms {float(e1.pow(2).mean().sqrt()):.3e} max {float(e1.abs().max()):.3e}"
                  f" | hi/lo accumulators (unscaled lo) rms {float(e2.pow(2).mean().sqrt()):.3e}")
