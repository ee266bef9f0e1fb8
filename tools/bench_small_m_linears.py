"""tools: the block Linears at ONE image's row counts (BASELINE configs[0]: image stream 256 rows, context stream k + 1 <= 512 rows): hipBLASLt's default choice
against csrc/gemm_fp32.hip in free order (256-row tiles + K-split tail round).   python tools/bench_small_m_linears.py"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import ops

g = torch.Generator(device="cuda").manual_seed(3)


def t_us(fn, reps=50):
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


for M in (21, 64, 128, 200, 256, 257, 384, 512, 513, 768, 1024, 2048):
    tl = ts = 0.0
    for name, (N, K) in zip(("qkv", "proj", "fc1", "fc2"), ((4608, 1536), (1536, 1536), (6144, 1536), (1536, 6144))):
        x = torch.randn(M, K, device="cuda", generator=g)
        w = torch.randn(N, K, device="cuda", generator=g) * 0.03
        b = torch.randn(N, device="cuda", generator=g)
        out = torch.empty(M, N, device="cuda")
        lib = t_us(lambda: F.linear(x, w, b))
        best = (1e9, 0)
        for split in (0, 2, 4, 8):
            try:
                t = t_us(lambda: ops.linear_f32(x, w, b, out=out, split=split))
            except Exception:
                continue
            best = min(best, (t, split))
        err = float((ops.linear_f32(x, w, b) - F.linear(x, w, b)).abs().max())
        fl = 2.0 * M * N * K
        tl += lib; ts += best[0]
        print(f"M={M:5d} {name:5s}: hipBLASLt {lib:7.1f} us ({fl / lib / 1e6:6.1f} TF)   sg free {best[0]:7.1f} us ({fl / best[0] / 1e6:6.1f} TF, best forced split {best[1]})  max |diff| {err:.1e}", flush=True)
    print(f"M={M:5d} sum: hipBLASLt {tl:7.1f} us, sg {ts:7.1f} us", flush=True)
