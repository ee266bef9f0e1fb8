"""GPU probe 2 (round 2): fp16 x 2 split (3 MFMA terms) as an fp32-equivalent GEMM.

a = a0 + a1 * 2^-11 with a0 = fp16(a), a1 = fp16((a - a0) * 2^11)   (same for the weight)
a.b ~= a0 b0 + 2^-11 (a0 b1 + a1 b0)          [dropped: a1 b1 2^-22]
hi and lo terms accumulate in separate fp32 accumulators.  Emulated through hipBLASLt (torch.mm out_dtype=fp32).
Also runs the MMDiT golden cases with every big Linear replaced by the emulation, to see the end-to-end effect.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

S = 2048.0


def split2(x):
    x0 = x.to(torch.float16)
    x1 = ((x - x0.float()) * S).to(torch.float16)
    return x0, x1


def mm_split(a, w_t):
    """a [M,K] fp32, w_t [K,N] fp32"""
    a0, a1 = split2(a)
    b0, b1 = split2(w_t)
    hi = torch.mm(a0, b0, out_dtype=torch.float32)
    lo = torch.mm(torch.cat([a0, a1], 1), torch.cat([b1, b0], 0), out_dtype=torch.float32)
    return hi + lo * (1.0 / S)


def main():
    dev = "cuda"
    torch.manual_seed(0)
    print("== numerics (M=2048, N=1536) ==")
    for K in (1536, 6144):
        for kind in ("ln", "wide"):
            a = torch.randn(2048, K, device=dev) * (1.0 + 3.0 * torch.rand(1, K, device=dev))
            if kind == "wide":        # heavy-tailed activations: a few huge entries + many tiny ones
                a = a * torch.exp(3.0 * torch.randn(2048, K, device=dev))
            w = (torch.rand(K, 1536, device=dev) * 2 - 1) * (3.0 / K) ** 0.5
            ref = a.double() @ w.double()
            den = float((a.double().abs() @ w.double().abs()).mean())
            e32 = (a @ w).double() - ref
            es = mm_split(a, w).double() - ref
            print(f"K={K} {kind}: |a|max {float(a.abs().max()):.1f} sum|ab| {den:.3f} | fp32 hipBLASLt max {float(e32.abs().max()):.3e} rms {float(e32.pow(2).mean().sqrt()):.3e}"
                  f" | fp16x2 max {float(es.abs().max()):.3e} rms {float(es.pow(2).mean().sqrt()):.3e}")

    print("== MMDiT golden cases with split linears ==")
    from selftoktokenizer_amd import synth, weights as W
    from selftoktokenizer_amd.encoder import QformerEncoderGPU
    from selftoktokenizer_amd.mmdit import MMDiTGPU
    import torch.nn.functional as F
    sd = W.synthetic_state_dict(W.expected_shapes(512), device="cuda")
    enc = QformerEncoderGPU(sd, torch.device("cuda"), 512)
    dit = MMDiTGPU(sd, torch.device("cuda"), 512)
    g = np.load(os.path.join(ROOT, "tests", "golden", "dit_forward_b1.npz"))
    ids = torch.from_numpy(synth.synthetic_token_ids(1)).cuda()
    ehs = enc.codes_ln(ids)
    x = synth.synthetic_noise(1, device="cuda")

    def run(case):
        t = torch.full((1,), float(g[f"t_{case}"]), device="cuda")
        mask = (torch.arange(512, device="cuda")[None] <= int(g[f"k_{case}"]))
        v, _ = dit(x, t, encoder_hidden_states=ehs, mask=mask, context_see_xt=True)
        return v.cpu()

    base = {c: run(c) for c in "abc"}
    real_lin = MMDiTGPU.lin
    from selftoktokenizer_amd import ops
    real_lg = ops.linear_gelu

    def lin_split(self, name, xx):
        w, b = self.w[name + ".weight"], self.w[name + ".bias"]
        if xx.shape[-1] < 1024 or xx.numel() // xx.shape[-1] < 64:
            return F.linear(xx, w, b)
        y = mm_split(xx.reshape(-1, xx.shape[-1]), w.t()) + b
        return y.reshape(*xx.shape[:-1], w.shape[0])

    def lg_split(xx, w, b):
        y = mm_split(xx.reshape(-1, xx.shape[-1]), w.t()) + b
        return F.gelu(y, approximate="tanh").reshape(*xx.shape[:-1], w.shape[0])

    MMDiTGPU.lin = lin_split
    ops.linear_gelu = lg_split
    import selftoktokenizer_amd.mmdit as MM
    MM.ops.linear_gelu = lg_split
    for c in "abc":
        v = run(c)
        ref = torch.from_numpy(g[f"v_{c}"])
        print(f"case {c}: fp32-lib err vs reference {float((base[c] - ref).abs().max()):.3e} | split err vs reference {float((v - ref).abs().max()):.3e}"
              f" | split vs fp32-lib {float((v - base[c]).abs().max()):.3e}  (|v|max {float(ref.abs().max()):.2f})")
    MMDiTGPU.lin = real_lin
    ops.linear_gelu = real_lg


if __name__ == "__main__":
    main()
