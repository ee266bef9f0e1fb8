"""tools: the exact-order LayerNorm kernels at the MMDiT's shapes (B = 64: 22912 context rows, 16384 image rows, 1536 columns): plain + modulate, and with the residual
update fused in (selftok_ex_res_layernorm_mod_f32); GB/s over the algorithmic bytes.   python tools/bench_ex_ln.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import ops

g = torch.Generator(device="cuda").manual_seed(2)
N = 1536


def t_us(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
    return best


for rows, T in ((22912, 358), (16384, 256)):
    B = rows // T
    x = torch.randn(rows, N, device="cuda", generator=g)
    lin = torch.randn(rows, N, device="cuda", generator=g)
    bias = torch.randn(N, device="cuda", generator=g)
    gate = torch.randn(B, N, device="cuda", generator=g)
    shift = torch.randn(B, N, device="cuda", generator=g)
    scale = torch.randn(B, N, device="cuda", generator=g)
    t0 = t_us(lambda: ops.ex_layernorm_mod(x.view(B, T, N), shift=shift, scale=scale, per_sample=True))
    xo = torch.empty_like(x)
    t1 = t_us(lambda: ops.ex_res_layernorm_mod(x.view(B, T, N), lin.view(B, T, N), lin_bias=bias, gate=gate, gate_mod=-T, shift=shift, scale=scale, per_sample=True, x_out=xo.view(B, T, N)))
    print(f"rows {rows}: LayerNorm + modulate {t0:7.1f} us ({rows * N * 8 / t0 / 1e3:7.1f} GB/s over 8 B per element)   with the residual update fused {t1:7.1f} us "
          f"({rows * N * 16 / t1 / 1e3:7.1f} GB/s over 16 B per element)", flush=True)
