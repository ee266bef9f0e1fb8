"""B = 1 latency of encode + 50-step decode in f16x2 mode: single-pass block Linears vs the small-M split-K entry points, eager and hipGraph."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import synth, weights as W  # noqa: E402
from selftoktokenizer_amd.config import default_config  # noqa: E402
from selftoktokenizer_amd.pipeline import SelftokPipeline  # noqa: E402

dev = torch.device("cuda")
sd = W.synthetic_state_dict(W.expected_shapes(512), device=dev)
pipe = SelftokPipeline(default_config(512), None, None, device=dev, state_dict=sd, vae_state_dict=W.synthetic_vae_state_dict(device=dev), verbose=False)
img = synth.synthetic_images(1, device=dev)
tok = pipe.encoding(img).cpu().numpy()
noise = synth.synthetic_noise(1)
lat = {}
for gemm, splitk in (("fp32", False), ("f16x2", False), ("f16x2", True)):
    pipe.set_gemm(gemm)
    pipe.model.model.SPLITK = splitk
    for graph in (False, True):
        ts = []
        for i in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            t = pipe.encoding(img)
            rec, l = pipe.decoding(t.cpu().numpy(), noise=noise, use_graph=graph, return_latent=True)
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        lat[gemm, splitk, graph] = l
        print(json.dumps({"gemm": gemm, "split_k": splitk, "hipgraph": graph, "encode_plus_decode_ms": round(float(np.median(ts[1:])), 1)}), flush=True)
print("final latent, f16x2 split-K vs single pass: max abs diff", float((lat["f16x2", True, False] - lat["f16x2", False, False]).abs().max()),
      "| vs fp32 GEMMs:", float((lat["f16x2", True, False] - lat["fp32", False, False]).abs().max()),
      "| single pass vs fp32:", float((lat["f16x2", False, False] - lat["fp32", False, False]).abs().max()),
      "| graph == eager:", bool(torch.equal(lat["f16x2", True, True], lat["f16x2", True, False])))
