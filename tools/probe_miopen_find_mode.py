"""GPU: first-call and steady-state time of the bf16 VAE (B = 64) under MIOPEN_FIND_MODE=$1 (unset = MIOpen's default, which on a
fresh box spends ~60 s of GPU time in naive reference convolutions during the first call of every shape)."""
import json
import os
import sys
import time

if len(sys.argv) > 1 and sys.argv[1] != "default":
    os.environ["MIOPEN_FIND_MODE"] = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from selftoktokenizer_amd import synth, weights as W  # noqa: E402
from selftoktokenizer_amd.vae import AutoencoderKLGPU  # noqa: E402

B = 64
vae = AutoencoderKLGPU(W.synthetic_vae_state_dict(device="cuda"), torch.device("cuda"))
img = synth.synthetic_images(B, device="cuda").bfloat16()
z = synth.synthetic_latents(B, device="cuda").bfloat16()
t0 = time.time(); vae.encode_moments(img); torch.cuda.synchronize(); first_e = time.time() - t0
t0 = time.time(); vae.decode(z); torch.cuda.synchronize(); first_d = time.time() - t0


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


print(json.dumps({"MIOPEN_FIND_MODE": os.environ.get("MIOPEN_FIND_MODE", "default"), "first_enc_s": round(first_e, 2), "first_dec_s": round(first_d, 2),
                  "enc_ms": round(timeit(lambda: vae.encode_moments(img)), 1), "dec_ms": round(timeit(lambda: vae.decode(z)), 1)}), flush=True)
