"""The reference's test.py flow on MI355X: image -> tokens (.npy) -> reconstruction (.png).

    python tools/run_image.py --image some.jpg [--yml-path cfg.yml --pretrained tokenizer_512_ckpt.pth --sd3_pretrained <sd3 dir>]

Without --pretrained the hash-generated synthetic weights are used (no checkpoints are reachable offline), which
exercises the whole path but reconstructs noise."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mimogpt.infer.infer_utils import parse_args_from_yaml
from mimogpt.infer.SelftokPipeline import SelftokPipeline
from selftoktokenizer_amd import preprocess, weights as W
from selftoktokenizer_amd.config import default_config

ap = argparse.ArgumentParser()
ap.add_argument("--image", required=True)
ap.add_argument("--yml-path", default=None)
ap.add_argument("--pretrained", default=None)
ap.add_argument("--sd3_pretrained", default=None)
ap.add_argument("--data_size", type=int, default=256)
ap.add_argument("--out", default=".")
a = ap.parse_args()
cfg = parse_args_from_yaml(a.yml_path) if a.yml_path else default_config(512)
kw = {}
if a.pretrained is None:
    kw = dict(state_dict=W.synthetic_state_dict(W.expected_shapes(int(cfg.tokenizer.params.k)), device="cuda"),
              vae_state_dict=W.synthetic_vae_state_dict(device="cuda"))
model = SelftokPipeline(cfg=cfg, ckpt_path=a.pretrained, sd3_path=a.sd3_pretrained, datasize=a.data_size, device="cuda", **kw)
images = torch.stack([preprocess.load_image(a.image, a.data_size)]).to("cuda")
tokens = model.encoding(images, device="cuda")
np.save(os.path.join(a.out, "token.npy"), tokens.detach().cpu().numpy())
tokens = np.load(os.path.join(a.out, "token.npy"))
images = model.decoding(tokens, device="cuda")
for b in range(len(images)):
    preprocess.save_image(images[b], os.path.join(a.out, f"re_{b}_{a.data_size}_2.png"))
print("tokens[0,:8] =", tokens[0, :8], "->", os.path.join(a.out, f"re_0_{a.data_size}_2.png"))
