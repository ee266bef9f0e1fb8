"""Tiny end-to-end invocation used by __graft_entry__.smoke(): B=1 encode + 2 decode steps + VAE on cuda:0.
(The oracle comparison for the VQ kernel lives in smoke() itself; the full-model parity lives in tests/.)"""
import torch

from . import synth, weights as W
from .config import default_config
from .pipeline import SelftokPipeline


def run():
    sd = W.synthetic_state_dict(W.expected_shapes(512), device="cuda")
    pipe = SelftokPipeline(default_config(512), None, None, device="cuda", state_dict=sd,
                           vae_state_dict=W.synthetic_vae_state_dict(device="cuda"))
    pipe.verbose = False
    tokens = pipe.encoding(synth.synthetic_images(1), device="cuda")
    assert tuple(tokens.shape) == (1, 512) and tokens.dtype == torch.int64
    assert int(tokens.min()) >= 0 and int(tokens.max()) < 32768
    rec = pipe.decoding(tokens.cpu().numpy(), device="cuda", noise=synth.synthetic_noise(1), max_steps=2)
    assert tuple(rec.shape) == (1, 3, 256, 256) and rec.dtype == torch.bfloat16
    assert bool(torch.isfinite(rec.float()).all()) and float(rec.min()) >= 0.0 and float(rec.max()) <= 1.0
    # the f16x2 arithmetic (split GEMM + split attention) must land on the same latents as the fp32 kernels
    _, lat32 = pipe.decoding(tokens.cpu().numpy(), noise=synth.synthetic_noise(1), max_steps=2, return_latent=True)
    assert pipe.set_gemm("f16x2") == "f16x2"
    _, lat16 = pipe.decoding(tokens.cpu().numpy(), noise=synth.synthetic_noise(1), max_steps=2, return_latent=True)
    pipe.set_gemm("fp32")
    diff = float((lat32 - lat16).abs().max())
    assert diff < 1e-5 and int(pipe.model.model.overflow.item()) == 0, diff
    torch.cuda.synchronize()
    print("[smoke] pipeline ok: encode -> 512 ids, 2-step decode -> pixels in [0,1]; f16x2 vs fp32 latents max diff %.1e" % diff)
