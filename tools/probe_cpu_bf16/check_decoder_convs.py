"""which of the encoder's chunk orders (oracle/vae_exact.c: 0 = (kh, kw, ic-block) sequential, 3 = ic-block-major with private partial sums) does
oneDNN use for the convolution shapes of the SD3-VAE DECODER (sd3_impls.py:380-444) on this CPU?  random bf16 data, mismatching outputs per order.
    python check_decoder_convs.py [B] [name-filter|-] [R]        (R = image resolution, default 256: every feature map scales by R / 256)"""
import os, sys, time
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import vae_exact as VX
from selftoktokenizer_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
only = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
R = int(sys.argv[3]) if len(sys.argv) > 3 else 256
SHAPES = [("conv_in 16->512 @32", 16, 512, 32, 3, False), ("512->512 @32", 512, 512, 32, 3, False), ("up 512->512 @32->64", 512, 512, 32, 3, True),
          ("512->512 @64", 512, 512, 64, 3, False), ("up 512->512 @64->128", 512, 512, 64, 3, True), ("512->256 @128", 512, 256, 128, 3, False),
          ("shortcut 512->256 1x1 @128", 512, 256, 128, 1, False), ("256->256 @128", 256, 256, 128, 3, False), ("up 256->256 @128->256", 256, 256, 128, 3, True),
          ("256->128 @256", 256, 128, 256, 3, False), ("shortcut 256->128 1x1 @256", 256, 128, 256, 1, False), ("128->128 @256", 128, 128, 256, 3, False),
          ("conv_out 128->3 @256", 128, 3, 256, 3, False)]
for name, cin, cout, H, k, up in SHAPES:
    if only and only not in name: continue
    H = H * R // 256
    x = (synth.hash_normalish(0x70 + cin + H, (B, cin, H, H)) * 1.2 + 0.05).to(torch.bfloat16)
    w = (synth.hash_normalish(0x71 + cout, (cout, cin, k, k)) * (1.0 / (cin * k * k)) ** 0.5).to(torch.bfloat16)
    b = (synth.hash_normalish(0x72, (cout,)) * 0.1).to(torch.bfloat16)
    with torch.no_grad():
        xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
        ref = F.conv2d(xin, w, b, padding=k // 2)
    refb = VX.bf16_bits(ref.permute(0, 2, 3, 1))
    xb = VX.bf16_bits(xin.permute(0, 2, 3, 1)); wb = VX.bf16_bits(w.permute(0, 2, 3, 1)); bb = VX.bf16_bits(b)
    res = {}
    for order in ((0, 3) if cin >= 32 else (0,)):
        if cin < 32: break
        t0 = time.time()
        y = VX.conv2d(xb, wb, bb, pad=k // 2, order=order)
        res[order] = int((y != refb).sum())
    print(f"R={R} {name:32s} B={B}: mismatching outputs of {refb.size} per order {res}", flush=True)
