"""Kernel times of the entropy-regulariser passes (csrc/vq_entropy.hip) at the tokenizer's training shape B x K = 64 x 512 rows against
C = 32768 codes: the two forward reductions and the backward, HIP events over 10 launches each.
Usage (GPU box): python tools/bench_vq_entropy.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import ops, synth  # noqa: E402
from selftoktokenizer_amd.vq_train import l2norm  # noqa: E402


def ms(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    K, C, D = 512, 32768, 16
    z = (synth.hash_normalish(0x5EED, (B, K, D)) * 3.0).cuda()
    cb = l2norm(synth.hash_normalish(0xC0DE, (C, D))).cuda()
    rows, cm = ops.vq_softmax_stats(z, cb)
    g = torch.randn(K, C, device="cuda")
    t_rows = ms(lambda: ops.vq_softmax_stats(z, cb, colmean=False))
    t_both = ms(lambda: ops.vq_softmax_stats(z, cb))
    t_bwd = ms(lambda: ops.vq_softmax_backward(z, cb, rows, g))
    N = B * K
    score_flop = 2.0 * N * C * D
    print(f"N = {N} rows x C = {C} codes (one materialised fp32 [N, C] tensor = {4e-9 * N * C:.2f} GB)")
    print(f"  row statistics (S, H)        {t_rows:7.3f} ms   {score_flop / t_rows * 1e-9:7.1f} TFLOP/s of score FMAs, {N * C / t_rows * 1e-6:6.1f} G exp/s")
    print(f"  + column means [K, C]        {t_both - t_rows:7.3f} ms   {score_flop / (t_both - t_rows) * 1e-9:7.1f} TFLOP/s")
    print(f"  backward (3 x 16 + 1 sums)   {t_bwd:7.3f} ms   {(score_flop + 4.0 * N * C * D) / t_bwd * 1e-9:7.1f} TFLOP/s (score + two 16-vector accumulations)")
    print(f"  forward + backward           {t_both + t_bwd:7.3f} ms;  HBM time of ONE pass over one materialised tensor at 8 TB/s: {4.0 * N * C / 8e12 * 1e3:.3f} ms")


if __name__ == "__main__":
    main()
