"""The exact-order VAE (csrc/vae_exact.hip) at the other probed image sizes: encode / decode time per B images next to the `parity` mode.
Usage (GPU box): python tools/bench_vae_exact_sizes.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import synth, weights as W  # noqa: E402
from selftoktokenizer_amd.vae import AutoencoderKLGPU  # noqa: E402


def ms(fn, n=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sd = W.synthetic_vae_state_dict()
for R in (128, 256, 320):
    img = synth.synthetic_images(B, size=R).to(torch.bfloat16).cuda()
    z = synth.hash_normalish(0xBE + R, (B, 16, R // 8, R // 8)).to(torch.bfloat16).cuda()
    for mode in ("exact", "parity"):
        vae = AutoencoderKLGPU(sd, torch.device("cuda"), mode=mode)
        print(f"{R} px  vae[{mode:6s}] B={B}: encode {ms(lambda: vae.encode(img)[0].mode()):8.1f} ms   decode {ms(lambda: vae.decode(z)[0]):8.1f} ms", flush=True)
