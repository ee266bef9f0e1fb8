"""tools: the LDS-DMA staged fp32-MFMA Linear (csrc/gemm_fp32.hip) -- correctness (MKL order: bit-equal to ex_linear; free order: error vs fp64 next to
hipBLASLt's) and time at the MMDiT's block-Linear shapes against hipBLASLt (F.linear) and xe_gemm128 (ex_linear).
    python tools/bench_sgemm.py [rows ...]        (default rows: 22912 = 64 x 358 context rows, 16384 = 64 x 256 image rows)"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import ops  # noqa: E402

PEAK = 157.3e12


def t_ms(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    rows = [int(a) for a in sys.argv[1:]] or [22912, 16384]
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(7)
    shapes = [("qkv", 4608, 1536), ("proj", 1536, 1536), ("fc1", 6144, 1536), ("fc2", 1536, 6144)]
    # ---- correctness at a ragged row count (not a multiple of 256) incl. the epilogues and every forced split ----
    M = 1000
    for name, N, K in shapes:
        x = torch.randn(M, K, device=dev, generator=g)
        w = torch.randn(N, K, device=dev, generator=g) * 0.03
        b = torch.randn(N, device=dev, generator=g)
        res = torch.randn(M, N, device=dev, generator=g)
        gate = torch.randn(250, N, device=dev, generator=g)
        ref = ops.ex_linear(x, w, b, kernel="xe")
        nblk = (K // 32 + 11) // 12
        for split in (0, nblk):
            got = ops.linear_f32(x, w, b, mkl_order=True, split=split)
            assert torch.equal(got, ref), (name, "mkl", split, float((got - ref).abs().max()))
        ref2 = ops.ex_linear(x, w, b, res=res, gate=gate, gate_mod=250, bias_last=True, kernel="xe")
        got2 = ops.linear_f32(x, w, b, mkl_order=True, res=res, gate=gate, gate_mod=250, bias_last=True, split=nblk)
        assert torch.equal(got2, ref2), (name, "mkl epilogue")
        if name == "fc1":                   # GELU: a second launch over out (contiguous out, no res / gate)
            assert torch.equal(ops.linear_f32(x, w, b, mkl_order=True, gelu=True, split=nblk), ops.ex_linear(x, w, b, gelu=True, kernel="xe")), (name, "mkl + GELU")
        ref3 = ops.ex_linear(x, w, b, res=res, gate=gate[:4], gate_mod=-250, kernel="xe")
        got3 = ops.linear_f32(x, w, b, mkl_order=True, res=res, gate=gate[:4], gate_mod=-250)
        assert torch.equal(got3, ref3), (name, "mkl per-sample gate")
        r64 = x.double() @ w.double().t() + b.double()
        e_lib = float((F.linear(x, w, b).double() - r64).pow(2).mean().sqrt())
        for split in (0, 2, 4, 8):
            got = ops.linear_f32(x, w, b, split=split)
            e = float((got.double() - r64).pow(2).mean().sqrt())
            assert e < 1.5 * e_lib + 1e-9, (name, "free", split, e, e_lib)
        print(f"{name:5s} M={M}: MKL order bit-equal to ex_linear (planned + forced split, epilogues); free order rms err {e:.3e} (hipBLASLt {e_lib:.3e})", flush=True)
    # ---- time ----
    for M in rows:
        for name, N, K in shapes:
            x = torch.randn(M, K, device=dev, generator=g)
            w = torch.randn(N, K, device=dev, generator=g) * 0.03
            b = torch.randn(N, device=dev, generator=g)
            fl = 2.0 * M * N * K
            out = torch.empty(M, N, device=dev)
            ms = {"hipBLASLt": t_ms(lambda: F.linear(x, w, b)),
                  "xe_gemm128 (MKL order)": t_ms(lambda: ops.ex_linear(x, w, b, out=out)),
                  "sg free": t_ms(lambda: ops.linear_f32(x, w, b, out=out)),
                  "sg free, tail unsplit": t_ms(lambda: ops.linear_f32(x, w, b, out=out, use_workspace=False)),
                  "sg free, flat priority": t_ms(lambda: ops.linear_f32(x, w, b, out=out, _flags=1 << 16)),
                  "sg MKL": t_ms(lambda: ops.linear_f32(x, w, b, out=out, mkl_order=True)),
                  "sg MKL, tail unsplit": t_ms(lambda: ops.linear_f32(x, w, b, out=out, mkl_order=True, use_workspace=False))}
            # the epilogues the exact MMDiT runs (sd3/mmdit.py:485-496): proj = x + gate * (y + bias last), fc1 = GELU, fc2 = x + gate * y; per-token gate table
            T = 358 if M % 358 == 0 else 256
            if name != "qkv":
                res = torch.randn(M, N, device=dev, generator=g)
                tab = torch.randn(T, N, device=dev, generator=g)
                kw = dict(gelu=True) if name == "fc1" else dict(res=res, gate=tab, gate_mod=T, bias_last=(name == "proj"))
                ms["xe_gemm128 + model epilogue"] = t_ms(lambda: ops.ex_linear(x, w, b, out=out, kernel="xe", **kw))
                ms["sg MKL + model epilogue"] = t_ms(lambda: ops.ex_linear(x, w, b, out=out, kernel="sg", **kw))
                ms["auto + model epilogue"] = t_ms(lambda: ops.ex_linear(x, w, b, out=out, **kw))
            tiles = ((M + 127) // 128) * (N // 128)
            print(f"{name:5s} [{M},{K}]x[{K},{N}] tiles {tiles} = {tiles / 512:.2f} rounds: " + "  ".join(f"{k} {v:.3f} ms ({fl / v / 1e9 / PEAK * 1e12:.3f})" for k, v in ms.items()), flush=True)


if __name__ == "__main__":
    main()
