"""does oneDNN's chunk order for the SD3-VAE ENCODER's convolutions depend on the image resolution?  The 256 x 256 assignment (oracle/vae_exact.py
conv_order: conv_in one 27-element chunk; the 128- / 256-channel Downsample layers channel-block major = order 3; everything else order 0) tried at
another resolution R: mismatching bf16 outputs against F.conv2d per order.
    python check_encoder_convs_res.py R [B]"""
import os, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import vae_exact as VX
from selftoktokenizer_amd import synth
R = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
SHAPES = [("conv_in", 3, 128, R, 3, 1), ("128@R", 128, 128, R, 3, 1), ("down128", 128, 128, R, 3, 2), ("128->256@R/2", 128, 256, R // 2, 3, 1), ("256@R/2", 256, 256, R // 2, 3, 1),
          ("down256", 256, 256, R // 2, 3, 2), ("256->512@R/4", 256, 512, R // 4, 3, 1), ("512@R/4", 512, 512, R // 4, 3, 1), ("down512", 512, 512, R // 4, 3, 2),
          ("512@R/8", 512, 512, R // 8, 3, 1), ("attn1x1@R/8", 512, 512, R // 8, 1, 1), ("conv_out", 512, 32, R // 8, 3, 1)]
for name, cin, cout, H, k, stride in SHAPES:
    x = (synth.hash_normalish(0x11 + cin + H, (B, cin, H, H)) * 1.2 + 0.05).to(torch.bfloat16)
    w = (synth.hash_normalish(0x12 + cout, (cout, cin, k, k)) * (1.0 / (cin * k * k)) ** 0.5).to(torch.bfloat16)
    b = (synth.hash_normalish(0x13, (cout,)) * 0.1).to(torch.bfloat16)
    with torch.no_grad():
        ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2) if stride == 2 else F.conv2d(x, w, b, padding=k // 2)
    refb = VX.bf16_bits(ref.permute(0, 2, 3, 1))
    xb = VX.bf16_bits(x.permute(0, 2, 3, 1)); wb = VX.bf16_bits(w.permute(0, 2, 3, 1)); bb = VX.bf16_bits(b)
    res = {}
    for order in ((2,) if cin < 32 else (0, 3)):
        y = VX.conv2d(xb, wb, bb, stride=stride, pad=(1 if k == 3 and stride == 1 else 0), order=order)
        res[order] = int((y != refb).sum())
    print(f"R={R} B={B} {name:14s} 256-px assignment: order {VX.conv_order(cin, k, stride)}; mismatches of {refb.size} per order {res}", flush=True)
