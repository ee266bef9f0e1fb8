"""Time the parts of the pipeline on the GPU (HIP events), B=64."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selftoktokenizer_amd import synth, weights as W
from selftoktokenizer_amd.vae import AutoencoderKLGPU

def timeit(fn, n=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

B = int(os.environ.get("B", "64"))
mode = sys.argv[1] if len(sys.argv) > 1 else "vae"
if mode == "vae":
    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        vae = AutoencoderKLGPU(W.synthetic_vae_state_dict(device="cuda"), torch.device("cuda"))
        img = synth.synthetic_images(B, device="cuda").bfloat16()
        z = synth.synthetic_latents(B, device="cuda").bfloat16()
        t0 = time.time(); vae.encode_moments(img); torch.cuda.synchronize(); first_e = time.time() - t0
        t0 = time.time(); vae.decode(z); torch.cuda.synchronize(); first_d = time.time() - t0
        te = timeit(lambda: vae.encode_moments(img))
        td = timeit(lambda: vae.decode(z))
        print(json.dumps({"cudnn.benchmark": bench, "B": B, "first_enc_s": round(first_e, 2), "first_dec_s": round(first_d, 2),
                          "enc_ms": round(te, 1), "dec_ms": round(td, 1),
                          "enc_TFs": round(0.271 * B / te * 1e3, 1), "dec_TFs": round(0.620 * B / td * 1e3, 1)}), flush=True)
