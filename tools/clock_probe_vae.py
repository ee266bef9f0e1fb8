"""tools: the shader clock beside the exact-order VAE (xconv_kernel: fp32-input MFMAs of the K = 1 form, as csrc/gemm_fp32.hip) and beside the parity VAE (bf16 MFMAs):
the probe of tools/clock_probe.py on a second stream while 64 images are encoded back to back.   python tools/clock_probe_vae.py"""
import ctypes, os, subprocess, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from selftoktokenizer_amd import synth, weights as W
from selftoktokenizer_amd.vae import AutoencoderKLGPU

so = "/tmp/clock_probe.so"
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "microbench", "clock_probe.hip")], check=True)
lib = ctypes.CDLL(so)
lib.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
N, GAP = 3000, 8
buf = torch.zeros(2 * N, dtype=torch.int64, device="cuda")
probe_stream = torch.cuda.Stream(priority=-1)
vsd = W.synthetic_vae_state_dict(device="cuda")
img = synth.synthetic_images(64, device="cuda")
for mode in ("exact", "parity"):
    vae = AutoencoderKLGPU(vsd, torch.device("cuda"), torch.bfloat16, mode=mode)
    enc = lambda: vae.encode(img)
    for _ in range(3):
        enc()
    buf.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    enc()
    assert lib.clock_probe_launch(buf.data_ptr(), N, GAP, probe_stream.cuda_stream) == 0
    e0.record()
    for _ in range(4):
        enc()
    e1.record(); torch.cuda.synchronize()
    a = buf.cpu().numpy().reshape(N, 2).astype(np.float64)
    rt, sc = a[:, 0], a[:, 1]
    k = N // 10
    win = [(sc[min(i + k, N - 1)] - sc[i]) / max(1.0, (rt[min(i + k, N - 1)] - rt[i])) * 0.1 for i in range(0, N - k, k)]
    print(f"VAE encode of 64 images, mode {mode:7s}: {e0.elapsed_time(e1) / 4:7.1f} ms per call; probe span {(rt[-1] - rt[0]) / 1e5:6.1f} ms; clock GHz per tenth: " + " ".join(f"{c:.3f}" for c in win), flush=True)
