"""tools: would csrc/gemm_fp32.hip in FREE order (one ascending chain per output; `selftok_linear_f32(flags = 0)`) beat the tuned hipBLASLt kernels anywhere in the
fp32 headline step?  Every row count of a B = 64 decode (context stream 64 (k + 1) rows for the 50 scheduled k, image stream 16384 rows) x the four block-Linear
families, the pipeline's own tuned F.linear against ops.linear_f32, weighted by how often the step runs each shape.   python tools/sweep_fp32_linear_vs_sg.py"""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import gemm_tune, ops, synth, weights as W
from selftoktokenizer_amd.config import default_config
from selftoktokenizer_amd.pipeline import SelftokPipeline

B = 64
sd = W.synthetic_state_dict(W.expected_shapes(512), device="cuda")
pipe = SelftokPipeline(default_config(512), None, None, device="cuda", state_dict=sd, vae_state_dict=W.synthetic_vae_state_dict(device="cuda"), verbose=False, tune_gemm=True)
pipe.tune_linears(B)
ctx_rows = [B * (int(k) + 1) for k in pipe.k_table]
g = torch.Generator(device="cuda").manual_seed(3)


def t_ms(fn, reps=6):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(2):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


tot_lib = tot_best = 0.0
with gemm_tune.enabled():
    for name, (N, K) in zip(("qkv", "proj", "fc1", "fc2"), gemm_tune.FAMILIES):
        w = torch.randn(N, K, device="cuda", generator=g) * 0.03
        b = torch.randn(N, device="cuda", generator=g)
        fam_lib = fam_best = 0.0
        wins = []
        for M, count in [(B * 256, 50)] + [(m, 1) for m in sorted(set(ctx_rows))]:
            count = count if M == B * 256 else ctx_rows.count(M)
            x = torch.randn(M, K, device="cuda", generator=g)
            out = torch.empty(M, N, device="cuda")
            lib = t_ms(lambda: F.linear(x, w, b))
            sg = t_ms(lambda: ops.linear_f32(x, w, b, out=out)) if M >= 256 else 1e9
            fl = 2.0 * M * N * K
            fam_lib += count * lib; fam_best += count * min(lib, sg)
            if sg < lib:
                wins.append((M, lib, sg))
            print(f"{name:5s} M={M:6d} x{count:2d}: tuned hipBLASLt {lib:7.3f} ms ({fl / lib / 1e9 / 157.3:.3f})   sg free {sg:7.3f} ms ({fl / sg / 1e9 / 157.3:.3f}){'   <-- sg' if sg < lib else ''}", flush=True)
        print(f"== {name}: per sampler pass of one block stream pair: library {fam_lib:.1f} ms, best-of-two {fam_best:.1f} ms ({100 * (1 - fam_best / fam_lib):.2f} % less); sg wins at {len(wins)} row counts", flush=True)
        tot_lib += fam_lib; tot_best += fam_best
print(f"== all four families, 24 blocks: library {24 * tot_lib / 1e3:.2f} s per 64 images, best-of-two {24 * tot_best / 1e3:.2f} s: {100 * (1 - tot_best / tot_lib):.2f} % of the Linear time, "
      f"~{100 * 0.91 * (1 - tot_best / tot_lib):.2f} % of the step")
