"""The VAE in its three modes (selftoktokenizer_amd/vae.py): encode + decode of B images at 256 x 256, HIP events, and the convolution kernel
alone on the decoder's heaviest layers.  Usage (GPU box): python tools/bench_vae.py [B] [modes...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import ops, synth, weights as W  # noqa: E402
from selftoktokenizer_amd.vae import AutoencoderKLGPU  # noqa: E402


def ms(fn, n=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    modes = sys.argv[2:] or ["parity", "miopen", "fast"]
    sd = W.synthetic_vae_state_dict()
    img = synth.synthetic_images(B).to(torch.bfloat16).cuda()
    lat = synth.synthetic_latents(B).to(torch.bfloat16).cuda()
    for mode in modes:
        vae = AutoencoderKLGPU(sd, torch.device("cuda"), mode=mode)
        te = ms(lambda: vae.encode(img)[0].mode())
        td = ms(lambda: vae.decode(lat)[0])
        print(f"vae[{mode:7s}] B={B}: encode {te:8.1f} ms  decode {td:8.1f} ms  sum {te + td:8.1f} ms", flush=True)
    print("convolution kernel alone (selftok_conv2d_nhwc_bf16):")
    for (H, Cin, Cout, up, name) in ((256, 128, 128, False, "up3 resnet conv"), (128, 256, 256, True, "up2 upsampler (reads 128^2, writes 256^2)"), (128, 256, 256, False, "up2 resnet conv"),
                                     (64, 512, 512, False, "up1 resnet conv"), (32, 512, 512, False, "mid resnet conv")):
        x = torch.randn(B, H, H, Cin, device="cuda").to(torch.bfloat16)
        pc = ops.PackedConv(torch.randn(Cout, Cin, 3, 3, device="cuda").to(torch.bfloat16) * 0.02, torch.randn(Cout, device="cuda").to(torch.bfloat16))
        t = ms(lambda: ops.conv2d_nhwc(x, pc, upsample=up), n=5)
        Ho = H * 2 if up else H
        fl = 2.0 * B * Ho * Ho * Cout * Cin * 9
        print(f"  {name:44s} [{B},{H},{H},{Cin}] -> {Cout}: {t:7.3f} ms  {fl / t * 1e-9:7.1f} TFLOP/s  ({fl / t * 1e-9 / 2500:.3f} of the bf16 peak)", flush=True)


if __name__ == "__main__":
    main()
