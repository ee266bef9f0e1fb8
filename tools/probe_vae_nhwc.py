"""GPU probe: do MIOpen's bf16 convolutions of the SD3 VAE run faster from channels_last tensors (no NCHW<->NHWC transposes)?
Times VAE decode / encode at B=64 (and B=256 for the renderer config) with torch's own group_norm as the stand-in epilogue in
both memory formats, next to the product path (NCHW + our GroupNorm+SiLU kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from selftoktokenizer_amd import synth, weights as W
from selftoktokenizer_amd.vae import AutoencoderKLGPU

vsd = W.synthetic_vae_state_dict(device="cuda")
vae = AutoencoderKLGPU(vsd, torch.device("cuda"))


def timeit(fn, n=3):
    fn(); fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


class Alt(AutoencoderKLGPU):
    def __init__(self, base, cl):
        self.device, self.dtype = base.device, base.dtype
        self.cl = cl
        self.w = {k: (v.contiguous(memory_format=torch.channels_last) if (cl and v.dim() == 4) else v) for k, v in base.w.items()}

    def _gn_silu(self, name, x, act=True):
        y = F.group_norm(x, 32, self.w[name + ".weight"], self.w[name + ".bias"], 1e-6)
        return F.silu(y) if act else y


for B in (64,):
    z = synth.synthetic_latents(B, device="cuda").to(torch.bfloat16)
    img = synth.synthetic_images(B, device="cuda").to(torch.bfloat16)
    print(f"B={B} product (NCHW + HIP GroupNorm+SiLU): decode {timeit(lambda: vae.decode(z)) * 1e3:.1f} ms, encode {timeit(lambda: vae.encode_moments(img)) * 1e3:.1f} ms", flush=True)
    for cl in (False, True):
        a = Alt(vae, cl)
        zz = z.contiguous(memory_format=torch.channels_last) if cl else z
        ii = img.contiguous(memory_format=torch.channels_last) if cl else img
        print(f"B={B} torch group_norm, channels_last={cl}: decode {timeit(lambda: a.decode(zz)) * 1e3:.1f} ms, encode {timeit(lambda: a.encode_moments(ii)) * 1e3:.1f} ms", flush=True)
