import sys, time, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools/oracle")
import ref_harness as H
import torch
t0=time.time()
cfg = H.load_cfg("/root/reference/configs/res256/256-eval.yml")
from selftoktokenizer_amd import weights as W
H.install()
from mimogpt.models.selftok.image_tokenizer import ImageTokenizer
cfg.tokenizer.params.noise_schedule_config.is_eval = True
with H.fast_init():
    m = ImageTokenizer(**cfg.tokenizer.params)
print("built", time.time()-t0)
ref = {k: tuple(v.shape) for k,v in m.state_dict().items() if not k.startswith("diffusion.")}
mine = W.expected_shapes(512)
print(len(ref), len(mine))
print("missing in mine:", [k for k in ref if k not in mine][:20])
print("extra in mine:", [k for k in mine if k not in ref][:20])
print("shape diff:", [(k, ref[k], mine[k]) for k in ref if k in mine and ref[k]!=mine[k]][:20])
print([k for k in m.state_dict() if k.startswith("diffusion.")])
