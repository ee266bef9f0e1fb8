"""tools: csrc/gemm_fp32.hip in free order at ONE image's row counts with the tail split forced beyond the planner's 8 units (a 512 MiB workspace): is a deeper K-split
what the 256-row-tile kernel lacks at small M?   python tools/bench_small_m_splits.py"""
import ctypes, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import _lib, ops

lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(3)
ws = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")


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


def sg(x, w, b, out, split):
    M, K = x.shape
    N = w.shape[0]
    rc = lib.selftok_linear_f32(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), None, 0, 0, None, 0, 0, out.data_ptr(), N, M, N, K, (split & 0xFF) << 8,
                                ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError(lib.selftok_last_error().decode())


for M in (64, 256, 512, 768):
    for name, (N, K) in zip(("qkv", "proj", "fc1", "fc2"), ((4608, 1536), (1536, 1536), (6144, 1536), (1536, 6144))):
        x = torch.randn(M, K, device="cuda", generator=g)
        w = torch.randn(N, K, device="cuda", generator=g) * 0.03
        b = torch.randn(N, device="cuda", generator=g)
        out = torch.empty(M, N, device="cuda")
        lib_us = t_us(lambda: F.linear(x, w, b))
        res = []
        for split in (0, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96):
            if split and (K // 32) % split:
                continue
            try:
                sg(x, w, b, out, split)
                err = float((out - F.linear(x, w, b)).abs().max())
                res.append((t_us(lambda: sg(x, w, b, out, split)), split, err))
            except Exception as e:
                res.append((1e9, split, str(e)[:40]))
        fl = 2.0 * M * N * K
        best = min(res)
        print(f"M={M:4d} {name:5s}: hipBLASLt {lib_us:6.1f} us ({fl / lib_us / 1e6:6.1f} TF) | sg best {best[0]:6.1f} us at split {best[1]} ({fl / best[0] / 1e6:6.1f} TF) | " +
              " ".join(f"{s}:{t:.0f}" for t, s, _ in res if t < 1e8), flush=True)
