"""does torch-CPU's bf16 F.conv2d give the same bits for an image inside a batch of 64 as inside a batch of 32?  The reference pipeline's decode of 64 images
in one call (BASELINE configs[1]) differs from its own decode of the same latents at 16 / 32 / 48 images per call in half of the bf16 pixels
(tools/oracle/gen_golden.py pipeline64; whole-decoder sweep: equal up to B = 48, different at B = 64).  Layer by layer: only the convolutions that touch the
[64, 256, 256, 256] bf16 activation -- exactly 2^31 bytes -- change; everything smaller is batch independent.
    python check_conv_batch64.py"""
import sys, os, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from selftoktokenizer_amd import synth
SHAPES = [("up 256->256 @128->256 (output = 2^31 bytes)", 256, 256, 128, 3, True), ("256->128 @256 (input = 2^31 bytes)", 256, 128, 256, 3, False),
          ("shortcut 256->128 1x1 @256 (input = 2^31 bytes)", 256, 128, 256, 1, False), ("128->128 @256 (2^30 bytes)", 128, 128, 256, 3, False),
          ("512->256 @128 (2^30 bytes in)", 512, 256, 128, 3, False)]
for name, cin, cout, H, k, up in SHAPES:
    x = (synth.hash_normalish(0x70 + cin + H, (64, cin, H, H)) * 1.2 + 0.05).to(torch.bfloat16)
    w = (synth.hash_normalish(0x71 + cout, (cout, cin, k, k)) * (1.0 / (cin * k * k)) ** 0.5).to(torch.bfloat16)
    b = (synth.hash_normalish(0x72, (cout,)) * 0.1).to(torch.bfloat16)
    t0 = time.time()
    with torch.no_grad():
        xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
        full = F.conv2d(xin, w, b, padding=k // 2)
        half = torch.cat([F.conv2d(xin[:32], w, b, padding=k // 2), F.conv2d(xin[32:], w, b, padding=k // 2)])
    d = int((full != half).sum())
    print(f"{name:52s}: batch of 64 vs two batches of 32: {d} of {full.numel()} outputs differ ({time.time() - t0:.0f} s)", flush=True)
