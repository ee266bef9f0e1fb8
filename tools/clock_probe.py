"""tools: the shader clock the chip holds while one fp32 Linear kernel runs back to back, read by a one-wave probe kernel on a second stream
(tools/microbench/clock_probe.hip: s_memtime against the 100 MHz s_memrealtime).   python tools/clock_probe.py"""
import ctypes, os, subprocess, sys
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from selftoktokenizer_amd import ops

src = os.path.join(ROOT, "tools", "microbench", "clock_probe.hip")
so = "/tmp/clock_probe.so"
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, src], check=True)
lib = ctypes.CDLL(so)
lib.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

N, GAP = 3000, 2                 # ~ 3000 samples, one every ~10 us at 2 x s_sleep(127) (64 x 127 cycles each)
buf = torch.zeros(2 * N, dtype=torch.int64, device="cuda")
probe_stream = torch.cuda.Stream(priority=-1)
M, Nn, K = 16384, 6144, 1536
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(M, K, device="cuda", generator=g)
w = torch.randn(Nn, K, device="cuda", generator=g) * 0.03
b = torch.randn(Nn, device="cuda", generator=g)
out = torch.empty(M, Nn, device="cuda")
xh, wh = x.half(), w.half()
fl = 2.0 * M * Nn * K
cases = [("idle", None), ("hipBLASLt F.linear fp32", lambda: F.linear(x, w, b)), ("sg free order", lambda: ops.linear_f32(x, w, b, out=out)),
         ("sg MKL order", lambda: ops.linear_f32(x, w, b, mkl_order=True, out=out)), ("xe_gemm128 (MKL order)", lambda: ops.ex_linear(x, w, b, out=out, kernel="xe")),
         ("hipBLASLt f16", lambda: F.linear(xh, wh))]
for rep in range(2):
    for name, fn in cases:
        reps = 0
        if fn is not None:
            for _ in range(30):          # settle: ~70 ms of the kernel before the probe starts
                fn()
        buf.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if fn is not None:
            for _ in range(10):
                fn()
        assert lib.clock_probe_launch(buf.data_ptr(), N, GAP, probe_stream.cuda_stream) == 0
        if fn is not None:
            e0.record()
            for _ in range(40):
                fn(); reps += 1
            e1.record()
        torch.cuda.synchronize()
        a = buf.cpu().numpy().reshape(N, 2).astype(np.float64)
        rt, sc = a[:, 0], a[:, 1]
        span_ms = (rt[-1] - rt[0]) / 1e5
        k = max(1, N // 10)
        # windows of N/10 samples: clock = d(shader cycles) / d(100 MHz ticks) * 100 MHz
        win = [(sc[min(i + k, N - 1)] - sc[i]) / max(1.0, (rt[min(i + k, N - 1)] - rt[i])) * 0.1 for i in range(0, N - k, k)]
        ms = e0.elapsed_time(e1) / reps if reps else 0.0
        rate = f"{ms:6.3f} ms {fl / ms / 1e9:7.1f} TF ({fl / ms / 1e9 / 157.3:.3f} of 157.3)" if reps else " " * 40
        print(f"{name:26s} {rate}  probe span {span_ms:6.1f} ms  clock GHz per tenth: " + " ".join(f"{c:.3f}" for c in win), flush=True)
