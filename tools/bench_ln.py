"""GPU: residual_ln_mod variants (tune build: SELFTOK_LN_VARIANT 0 = one row per wave (rounds 1-2), 1 = walk; SELFTOK_LN_R rows per wave;
SELFTOK_LN_NT non-temporal stores; SELFTOK_LN_HOIST bit 0 shift/scale, bit 1 gate kept in registers along the walk) at the decode step's shapes, stand-alone (cold
tensors, 4 x 141 MB > Infinity Cache) -- TB/s of algorithmic bytes (4 tensors + the table once)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SELFTOK_HIP_LIB"] = os.path.join(ROOT, "tools", "microbench", "libselftok_tune.so")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from selftoktokenizer_amd import ops  # noqa: E402


def ev(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


B, H = 64, 1536
cases = []
for n, per_sample in ((358, False), (256, True), (512, False)):
    x, y = torch.randn(B, n, H, device="cuda"), torch.randn(B, n, H, device="cuda")
    tab = torch.randn(B if per_sample else n, 6 * H, device="cuda")
    kw = dict(y=y, gate=tab[:, 2 * H:3 * H], shift=tab[:, 3 * H:4 * H], scale=tab[:, 4 * H:5 * H], per_sample=per_sample)
    cases.append((f"[{B},{n},{H}] {'per-sample' if per_sample else 'per-token'} resid+LN (2r+2w)", x, kw, 4))
    cases.append((f"[{B},{n},{H}] {'per-sample' if per_sample else 'per-token'} LN only (1r+1w)", x, dict(shift=kw["shift"], scale=kw["scale"], per_sample=per_sample), 2))
    cases.append((f"[{B},{n},{H}] {'per-sample' if per_sample else 'per-token'} LN only, split out", x, dict(shift=kw["shift"], scale=kw["scale"], per_sample=per_sample, split=True), 2))
ref = {}
configs = [(0, 1, 0, 3)] + [(1, r, nt, hoist) for hoist in (3, 1, 0) for r in (2, 4, 8) for nt in (0, 1)]
for name, x, kw, ntens in cases:
    by = ntens * x.numel() * 4
    line = []
    for v, r, nt, hoist in configs:
        os.environ.update(SELFTOK_LN_VARIANT=str(v), SELFTOK_LN_R=str(r), SELFTOK_LN_NT=str(nt), SELFTOK_LN_HOIST=str(hoist))
        out = ops.residual_ln_mod(x, **kw)
        o = out[1].data if isinstance(out[1], ops.SplitAct) else out[1]
        key = name
        if v == 0:
            ref[key] = (o.clone(), out[0].clone() if kw.get("y") is not None else None)
        else:
            assert torch.equal(o, ref[key][0]), (name, v, r, nt)
            if ref[key][1] is not None:
                assert torch.equal(out[0], ref[key][1])
        ms = ev(lambda: ops.residual_ln_mod(x, **kw))
        line.append(f"v{v} R{r} nt{nt} h{hoist}: {by / ms / 1e9:.2f}")
    print(name + "  TB/s\n    " + " | ".join(line), flush=True)
