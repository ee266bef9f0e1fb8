"""One image through encode + the 50-step decode (eager), for `rocprofv3 --kernel-trace --stats`: where a B = 1 step spends its time."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import synth, weights as W  # noqa: E402
from selftoktokenizer_amd.config import default_config  # noqa: E402
from selftoktokenizer_amd.pipeline import SelftokPipeline  # noqa: E402

dev = torch.device("cuda")
sd = W.synthetic_state_dict(W.expected_shapes(512), device=dev)
pipe = SelftokPipeline(default_config(512), None, None, device=dev, state_dict=sd, vae_state_dict=W.synthetic_vae_state_dict(device=dev), verbose=False)
img = synth.synthetic_images(1, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
if len(sys.argv) > 2:
    print("gemm mode:", pipe.set_gemm(sys.argv[2]), flush=True)
graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
for i in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tok = pipe.encoding(img)
    pipe.decoding(tok.cpu().numpy(), use_graph=graph)
    torch.cuda.synchronize()
    print(f"pass {i}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
