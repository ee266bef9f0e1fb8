"""time the Q-Former encoder (features + VQ) at B = 64 in 'exact' and 'fast' mode; report feature / id differences between the two
    python tools/bench_encoder_modes.py [B]"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import weights as W, synth
from selftoktokenizer_amd.encoder import QformerEncoderGPU

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
shapes = {k: v for k, v in W.expected_shapes(512).items() if k.startswith("encoder.")}
sd = W.synthetic_state_dict(shapes)
dev = torch.device("cuda", 0)
x0 = synth.synthetic_latents(B).float().to(torch.bfloat16).float().cuda()
res = {}
for mode in ("exact", "fast"):
    enc = QformerEncoderGPU(sd, dev, 512, mode=mode)
    for _ in range(2):
        z = enc.features(x0); ids = enc(x0)[1]
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        ids = enc(x0)[1]
    torch.cuda.synchronize()
    res[mode] = (z, ids, (time.time() - t0) / 5)
    print(f"{mode}: {res[mode][2] * 1e3:.2f} ms per encoder call at B = {B}", flush=True)
ze, ie, _ = res["exact"]; zf, i_f, _ = res["fast"]
print(f"fast vs exact: features max abs diff {float((ze - zf).abs().max()):.3e}, ids differing {int((ie != i_f).sum())} of {ie.numel()}")
