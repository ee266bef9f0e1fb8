"""Launch shapes of the implicit-GEMM convolution (csrc/conv.hip), tune build only (tools/build_tune.sh + SELFTOK_HIP_LIB), SELFTOK_CONV_VARIANT:
  0  conv_nhwc_bf16_kernel<4,2,2>: 8 waves, 256 px x 128 ch per workgroup (64 x 64 per wave), one barrier per tap
  1  conv_nhwc_bf16_kernel<2,2,2>: 4 waves, 128 px x 128 ch (64 x 64 per wave)
  2  conv_nhwc_bf16_kernel<2,4,1>: 8 waves, 128 px x 128 ch (64 x 32 per wave, 120 VGPRs: two workgroups per CU)
  9  conv3x3_rows_kernel<2>: as 2 with one barrier per kernel row, two weight buffers (64 KB LDS: two workgroups per CU)
  8  conv3x3_rows_kernel<1>: one weight buffer (41 KB LDS, 78 VGPRs: three workgroups per CU) -- the product's choice
Alternating runs, median of 3 x 5 launches; the outputs of all variants must be bit-identical."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import ops  # noqa: E402


def ms(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "1", "2"]
shapes = ((256, 128, 128, 3, False), (256, 256, 128, 3, False), (128, 256, 256, 3, True), (128, 256, 256, 3, False), (64, 512, 512, 3, False), (32, 512, 512, 3, False), (256, 256, 128, 1, False))
for (H, Cin, Cout, ks, up) in shapes:
    x = torch.randn(B, H, H, Cin, device="cuda").to(torch.bfloat16)
    pc = ops.PackedConv(torch.randn(Cout, Cin, ks, ks, device="cuda").to(torch.bfloat16) * 0.02, torch.randn(Cout, device="cuda").to(torch.bfloat16))
    Ho = H * 2 if up else H
    fl = 2.0 * B * Ho * Ho * Cout * Cin * ks * ks
    res = {v: [] for v in variants}
    ref = None
    for rep in range(3):
        for v in variants:
            os.environ["SELFTOK_CONV_VARIANT"] = v
            res[v].append(ms(lambda: ops.conv2d_nhwc(x, pc, upsample=up)))
            out = ops.conv2d_nhwc(x, pc, upsample=up)
            if ref is None:
                ref = out
            else:
                assert torch.equal(out, ref), "variants must be bit-identical (same accumulation order per output)"
    line = "  ".join(f"v{v}: {statistics.median(res[v]):7.3f} ms {fl / statistics.median(res[v]) * 1e-9:6.0f} TF/s" for v in variants)
    print(f"[{B},{H},{H},{Cin}]->{Cout} k{ks}{' up' if up else ''}: {line}", flush=True)
