"""B=64 sampler steps only (no VAE/encode): ms per step for steps at several context lengths."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selftoktokenizer_amd import synth, weights as W
from selftoktokenizer_amd.config import default_config
from selftoktokenizer_amd.pipeline import SelftokPipeline
sd = W.synthetic_state_dict(W.expected_shapes(512), device="cuda")
pipe = SelftokPipeline(default_config(512), None, None, device="cuda", state_dict=sd, vae_state_dict=W.synthetic_vae_state_dict(device="cuda"), verbose=False)
B = 64
ids = torch.from_numpy(synth.synthetic_token_ids(B)).cuda(); noise = synth.synthetic_noise(B, device="cuda")
ehs = pipe.model.encoder.codes_ln(ids)
def run(n):
    return pipe.flow.p_sample_loop(pipe.model.model, noise, ehs, pipe.k_table, max_steps=n)
ref = run(4); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); out = run(12); e.record(); torch.cuda.synchronize()
print(json.dumps({"ms_per_step_first12": round(s.elapsed_time(e) / 12, 2),
                  "checksum": float(out.double().abs().sum())}))
