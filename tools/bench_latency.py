"""B=1 (and small-batch) decode latency: eager launches vs the captured hipGraph."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selftoktokenizer_amd import synth, weights as W
from selftoktokenizer_amd.config import default_config
from selftoktokenizer_amd.pipeline import SelftokPipeline
sd = W.synthetic_state_dict(W.expected_shapes(512), device="cuda")
pipe = SelftokPipeline(default_config(512), None, None, device="cuda", state_dict=sd, vae_state_dict=W.synthetic_vae_state_dict(device="cuda"))
pipe.verbose = False
for B in (1, 4):
    ids, noise = synth.synthetic_token_ids(B), synth.synthetic_noise(B)
    img = synth.synthetic_images(B, device="cuda")
    for graph in (False, True):
        pipe.decoding(ids, noise=noise, use_graph=graph)       # warm / capture
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2):
            pipe.decoding(ids, noise=noise, use_graph=graph)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
        print(json.dumps({"B": B, "hipgraph": graph, "decode_s": round(dt, 3), "images_per_s": round(B / dt, 3)}), flush=True)
    pipe.encoding(img); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): pipe.encoding(img)
    torch.cuda.synchronize(); print(json.dumps({"B": B, "encode_s": round((time.perf_counter() - t0) / 3, 4)}), flush=True)
