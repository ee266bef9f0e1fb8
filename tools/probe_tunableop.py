"""Which GEMM problems does one step issue, in PyTorch TunableOp's own naming?  Runs encode + 3 decode steps at B = 64 with TunableOp enabled,
tuning off and `record_untuned` on; prints the recorded (operator, problem key) list.  GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import synth, weights as W  # noqa: E402
from selftoktokenizer_amd.config import default_config  # noqa: E402
from selftoktokenizer_amd.pipeline import SelftokPipeline  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/untuned.csv"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda")
sd = W.synthetic_state_dict(W.expected_shapes(512), device=dev)
pipe = SelftokPipeline(default_config(512), None, None, device=dev, state_dict=sd, vae_state_dict=W.synthetic_vae_state_dict(device=dev), verbose=False)
img = synth.synthetic_images(B, device=dev)
tun = torch.cuda.tunable
tun.enable(True)
tun.tuning_enable(False)
tun.record_untuned_enable(True)
os.environ["PYTORCH_TUNABLEOP_UNTUNED_FILENAME"] = out
tok = pipe.encoding(img)
pipe.decoding(tok.cpu().numpy(), max_steps=3)
torch.cuda.synchronize()
print("validators:", tun.get_validators())
print("results in memory:", len(tun.get_results()))
for cand in (out, out.replace(".csv", "0.csv"), "tunableop_untuned0.csv"):
    if os.path.exists(cand):
        print("==", cand)
        print(open(cand).read())
