"""GPU: the split-activation producers against their fp32-output forms at the decode step's shapes (B = 64)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from selftoktokenizer_amd import ops  # noqa: E402


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


B, H, NH = 64, 1536, 24
for n in (358, 256):
    x, y = torch.randn(B, n, H, device="cuda"), torch.randn(B, n, H, device="cuda")
    tab = torch.randn(n, 6 * H, device="cuda")
    kw = dict(y=y, gate=tab[:, 2 * H:3 * H], shift=tab[:, 3 * H:4 * H], scale=tab[:, 4 * H:5 * H])
    t0 = bench(lambda: ops.residual_ln_mod(x, **kw))
    t1 = bench(lambda: ops.residual_ln_mod(x, split=True, **kw))
    t2 = bench(lambda: ops.residual_ln_mod(x, shift=kw["shift"], scale=kw["scale"]))
    t3 = bench(lambda: ops.residual_ln_mod(x, split=True, shift=kw["shift"], scale=kw["scale"]))
    print(f"residual_ln_mod [{B},{n},{H}]: residual+LN fp32 out {t0:.3f} ms, split out {t1:.3f} ms | LN only fp32 out {t2:.3f} ms, split out {t3:.3f} ms", flush=True)
n = 358
cq, xq = torch.randn(B, n, 3 * H, device="cuda"), torch.randn(B, 256, 3 * H, device="cuda")
for split in (False, True):
    oc = ops.SplitAct((B, n, H), "cuda") if split else torch.empty(B, n, H, device="cuda")
    ox = ops.SplitAct((B, 256, H), "cuda") if split else torch.empty(B, 256, H, device="cuda")
    seg0 = (cq[..., :H], cq[..., H:2 * H], cq[..., 2 * H:], oc)
    seg1 = (xq[..., :H], xq[..., H:2 * H], xq[..., 2 * H:], ox)
    t = bench(lambda: ops.attention(seg0, seg1, NH, 64, mode=ops.ATTN_F16X2))
    print(f"attn64_f16x2 S={n}+256 split_out={split}: {t:.3f} ms", flush=True)
