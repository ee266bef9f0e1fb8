"""The exact-order VAE encoder (csrc/vae_exact.hip) at B images: encode time next to the `parity` encoder, and the convolution kernel alone
on the encoder's layer shapes as a fraction of the fp32 matrix peak (157.3 TFLOP/s).  Usage (GPU box): python tools/bench_vae_exact.py [B]"""
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
    sd = W.synthetic_vae_state_dict()
    img = synth.synthetic_images(B).to(torch.bfloat16).cuda()
    z = synth.synthetic_latents(B).to(torch.bfloat16).cuda()
    for mode in ("exact", "parity"):
        vae = AutoencoderKLGPU(sd, torch.device("cuda"), mode=mode)
        print(f"vae[{mode:6s}] B={B}: encode {ms(lambda: vae.encode(img)[0].mode()):8.1f} ms   decode {ms(lambda: vae.decode(z)[0]):8.1f} ms", flush=True)
    print("selftok_vx_conv2d_bf16 alone (fp32 MFMA, AMX chunk order):")
    tot_fl = tot_t = 0.0
    for (name, cin, cout, H, k, stride, order, count) in (("128->128 3x3 @256", 128, 128, 256, 3, 1, 0, 4), ("Downsample 128", 128, 128, 256, 3, 2, 3, 1), ("128->256 @128", 128, 256, 128, 3, 1, 0, 1),
                                                          ("256->256 @128", 256, 256, 128, 3, 1, 0, 3), ("Downsample 256", 256, 256, 128, 3, 2, 3, 1), ("256->512 @64", 256, 512, 64, 3, 1, 0, 1),
                                                          ("512->512 @64", 512, 512, 64, 3, 1, 0, 3), ("Downsample 512", 512, 512, 64, 3, 2, 0, 1), ("512->512 @32", 512, 512, 32, 3, 1, 0, 8),
                                                          ("1x1 512 @32 (q,k,v,out)", 512, 512, 32, 1, 1, 0, 4)):
        x = torch.randn(B, H, H, cin, device="cuda").to(torch.bfloat16)
        w = (torch.randn(cout, k, k, cin, device="cuda") * 0.02).to(torch.bfloat16)
        b = torch.randn(cout, device="cuda").to(torch.bfloat16)
        t = ms(lambda: ops.vx_conv2d(x, w, b, stride=stride, order=order), n=3)
        Ho = H // stride
        fl = 2.0 * B * Ho * Ho * cout * cin * k * k
        tot_fl += fl * count; tot_t += t * count
        print(f"  {name:28s} x{count}: {t:8.3f} ms  {fl / t * 1e-9:6.1f} TFLOP/s  ({fl / t * 1e-9 / 157.3:.3f} of the fp32 matrix peak)", flush=True)
    print(f"  all convolutions of one encode: {tot_t:.1f} ms, {tot_fl / tot_t * 1e-9:.1f} TFLOP/s ({tot_fl / tot_t * 1e-9 / 157.3:.3f})")
    x = torch.randn(B, 256, 256, 128, device="cuda").to(torch.bfloat16)
    g, bb = torch.ones(128, device="cuda").to(torch.bfloat16), torch.zeros(128, device="cuda").to(torch.bfloat16)
    tab = ops.vx_silu_table("cuda")
    t = ms(lambda: ops.vx_groupnorm(x, g, bb, silu_table=tab))
    print(f"  GroupNorm + SiLU [{B},256,256,128]: {t:.3f} ms ({3 * x.numel() * 2 / t * 1e-6:.0f} GB/s over 2 reads + 1 write)")
    q = torch.randn(B, 1024, 512, device="cuda").to(torch.bfloat16)
    t = ms(lambda: ops.vx_attention(q, q, q))
    print(f"  attention [{B},1024,512]: {t:.3f} ms ({4.0 * B * 1024 * 1024 * 512 / t * 1e-9:.1f} TFLOP/s)")


if __name__ == "__main__":
    main()
