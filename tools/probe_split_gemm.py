"""GPU probe (round 2): can an fp32 GEMM be replaced by a 3-way bf16 split on the bf16 matrix cores?

a = a0 + a1 + a2 (each bf16, 24 significand bits together), same for b; the fp32-equivalent product keeps the six
cross terms of order <= 2:  a0b0 + a0b1 + a1b0 + a0b2 + a1b1 + a2b0, accumulated in fp32 by the MFMA.
Measured here through hipBLASLt (torch.mm(..., out_dtype=fp32) on K-concatenated bf16 operands) to learn
  (i) the error of 3 / 6 terms against an fp64 product, next to hipBLASLt's own fp32 GEMM error;
  (ii) the bf16 GEMM rate at the DiT shapes, i.e. the ceiling a hand-written split kernel competes with.
"""
import sys
import time

import torch


def split3(x):
    x0 = x.to(torch.bfloat16)
    r = x - x0.float()
    x1 = r.to(torch.bfloat16)
    r = r - x1.float()
    x2 = r.to(torch.bfloat16)
    return x0, x1, x2


def cat_terms(a, b, terms):
    """a [M,K] fp32, b [K,N] fp32 -> bf16 [M, T*K], [T*K, N] for the list of (i,j) plane pairs"""
    A, Bp = split3(a), split3(b)
    return torch.cat([A[i] for i, _ in terms], dim=1).contiguous(), torch.cat([Bp[j] for _, j in terms], dim=0).contiguous()


T6 = [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]
T3 = [(0, 0), (0, 1), (1, 0)]


def bench(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def main():
    dev = "cuda"
    torch.manual_seed(0)
    print("== numerics (M=2048, N=1536) ==")
    for K in (1536, 6144):
        a = torch.randn(2048, K, device=dev) * (1.0 + 3.0 * torch.rand(1, K, device=dev))        # LN+modulate like activations
        w = (torch.rand(K, 1536, device=dev) * 2 - 1) * (3.0 / K) ** 0.5
        ref = a.double() @ w.double()
        scale = float(ref.abs().max())
        f32 = a @ w
        e32 = (f32.double() - ref)
        print(f"K={K} |ref|max {scale:.3f}  fp32 hipBLASLt: max {float(e32.abs().max()):.3e} rms {float(e32.pow(2).mean().sqrt()):.3e}")
        for name, terms in (("bf16x3 (3 terms)", T3), ("bf16x3 (6 terms)", T6)):
            A, Bm = cat_terms(a, w, terms)
            out = torch.mm(A, Bm, out_dtype=torch.float32)
            e = out.double() - ref
            print(f"K={K} {name}: max {float(e.abs().max()):.3e} rms {float(e.pow(2).mean().sqrt()):.3e}")
        # low-order planes accumulated separately, then added (checks whether the accumulator rounding of tiny terms matters)
        A0, A1, A2 = split3(a)
        B0, B1, B2 = split3(w)
        hi = torch.mm(A0, B0, out_dtype=torch.float32)
        lo = torch.mm(torch.cat([A0, A1, A0, A1, A2], 1), torch.cat([B1, B0, B2, B1, B0], 0), out_dtype=torch.float32)
        e = (hi + lo).double() - ref
        print(f"K={K} 6 terms, hi/lo separate accumulators: max {float(e.abs().max()):.3e} rms {float(e.pow(2).mean().sqrt()):.3e}")
        e = hi.double() - ref
        print(f"K={K} plain bf16 (1 term): max {float(e.abs().max()):.3e} rms {float(e.pow(2).mean().sqrt()):.3e}")

    print("== throughput at the DiT shapes (B=64: M = 64*(358+256) = 39296 rows) ==")
    M = 39296
    for (N, K) in ((4608, 1536), (1536, 1536), (6144, 1536), (1536, 6144)):
        a = torch.randn(M, K, device=dev)
        w = torch.randn(K, N, device=dev) * 0.02
        flop = 2.0 * M * N * K
        t32 = bench(lambda: a @ w)
        line = f"N={N} K={K}: fp32 {t32 * 1e3:.3f} ms {flop / t32 / 1e12:.1f} TF"
        for name, terms in (("x3", T3), ("x6", T6)):
            A, Bm = cat_terms(a, w, terms)
            t = bench(lambda: torch.mm(A, Bm, out_dtype=torch.float32))
            line += f" | bf16{name} {t * 1e3:.3f} ms = {flop / t / 1e12:.1f} TF-equiv ({len(terms) * flop / t / 1e12:.0f} TF bf16)"
        tsplit = bench(lambda: split3(a))
        line += f" | torch split3(A) {tsplit * 1e3:.3f} ms"
        print(line, flush=True)


if __name__ == "__main__":
    sys.exit(main())
