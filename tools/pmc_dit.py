"""B=64 encode + 2 sampler steps for rocprofv3 --pmc passes (weights = torch uniform noise: counters do not care)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selftoktokenizer_amd import weights as W
from selftoktokenizer_amd.config import default_config
from selftoktokenizer_amd.pipeline import SelftokPipeline

def fake(shapes, dtype=torch.float32):
    sd = {}
    for k, s in shapes.items():
        if len(s) >= 2:
            a = math.sqrt(3.0 / max(1, int(torch.tensor(s[1:]).prod())))
            sd[k] = torch.empty(s, device="cuda", dtype=dtype).uniform_(-a, a)
        else:
            sd[k] = torch.empty(s, device="cuda", dtype=dtype).uniform_(0.9, 1.1)
    return sd

sd = fake(W.expected_shapes(512))
sd["encoder.quantizer._codebook.embed"] = torch.nn.functional.normalize(torch.randn(1, 32768, 16, device="cuda"), dim=-1)
pipe = SelftokPipeline(default_config(512), None, None, device="cuda", state_dict=sd, vae_state_dict=fake(W.vae_shapes(), torch.bfloat16))
pipe.verbose = False
B = 64
tok = pipe.encoding(torch.rand(B, 3, 256, 256, device="cuda") * 2 - 1)
pipe.decoding(tok, noise=torch.randn(B, 16, 32, 32), max_steps=2)
torch.cuda.synchronize()
