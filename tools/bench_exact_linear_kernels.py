"""tools: same-box A/B of the exact-order MMDiT's Linear kernels: `ex_linear` on csrc/gemm_fp32.hip (auto dispatch) vs on xe_gemm128 (round 5), B = 64, K = 512,
the first `steps` sampler steps of a decode + the last `steps` (short context).   python tools/bench_exact_linear_kernels.py [steps]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import ops, synth, weights as W
from selftoktokenizer_amd.config import default_config
from selftoktokenizer_amd.pipeline import SelftokPipeline

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = 64
sd = W.synthetic_state_dict(W.expected_shapes(512), device="cuda")
pipe = SelftokPipeline(default_config(512), None, None, device="cuda", state_dict=sd, vae_state_dict=W.synthetic_vae_state_dict(device="cuda"), verbose=False, gemm="exact")
ids = synth.synthetic_token_ids(B)
noise = synth.synthetic_noise(B)
ehs = pipe._codes(ids)


def run(k_table):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lat = pipe.flow.p_sample_loop(pipe.model.model, noise, ehs, k_table, context_see_xt=True, max_steps=steps)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, lat


res = {}
for name, min_rows, attn in (("sg (csrc/gemm_fp32.hip)", 256, "auto"), ("xe_gemm128 (round 5)", 10 ** 9, "auto"), ("sg again", 256, "auto"),
                             ("sg, fc1 + GELU on xe_gemm128", -256, "auto"), ("xe_gemm128 again", 10 ** 9, "auto"),
                             ("sg, attention unfused (round 5)", 256, "unfused"), ("sg, attention fused", 256, "auto")):
    ops.EX_LINEAR_SG_MIN_ROWS = abs(min_rows)
    ops.EX_LINEAR_GELU_ON_XE = min_rows < 0
    ops.EX_ATTENTION_DEFAULT = attn
    for label, kt in (("first steps (context 512..)", pipe.k_table), ("last steps (context ..20)", pipe.k_table[-steps:])):
        run(kt)
        t, lat = run(kt)
        res[(name, label)] = (t, lat)
        print(f"{name:28s} {label:30s}: {1e3 * t / steps:8.1f} ms per sampler step", flush=True)
a, b = res[("sg (csrc/gemm_fp32.hip)", "first steps (context 512..)")][1], res[("xe_gemm128 (round 5)", "first steps (context 512..)")][1]
print("latents bit-equal between the two Linear kernels:", bool(torch.equal(a, b)))
c, d = res[("sg, attention unfused (round 5)", "first steps (context 512..)")][1], res[("sg, attention fused", "first steps (context 512..)")][1]
print("latents bit-equal between the fused and the unfused attention:", bool(torch.equal(c, d)))
