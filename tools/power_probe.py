"""tools: board power and shader clock while one fp32 Linear kernel runs back to back (round 6: does the chip throttle under this repo's fp32-MFMA kernels and not under
the vendor's?).  Samples the amdgpu hwmon files (power1_average / power1_input in uW, freq1_input in Hz) every 50 ms from a thread; falls back to `rocm-smi`.
python tools/power_probe.py [seconds per kernel]"""
import glob, os, subprocess, sys, threading, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import ops

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0


def hwmon_files():
    out = {}
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for key, names in (("power_uW", ("power1_average", "power1_input")), ("sclk_Hz", ("freq1_input",)), ("mclk_Hz", ("freq2_input",)), ("temp_mC", ("temp1_input", "temp2_input"))):
            for n in names:
                p = os.path.join(h, n)
                if key not in out and os.path.exists(p):
                    try:
                        open(p).read(); out[key] = p
                    except OSError:
                        pass
    return out


FILES = hwmon_files()
print("hwmon:", FILES, flush=True)


def sample():
    s = {}
    for k, p in FILES.items():
        try:
            s[k] = float(open(p).read().strip())
        except (OSError, ValueError):
            pass
    return s


def smi():
    try:
        return subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:          # noqa: BLE001
        return f"rocm-smi failed: {e}"


def run(name, fn, flops):
    fn(); torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            samples.append(sample()); time.sleep(0.05)
    th = threading.Thread(target=poll); th.start()
    n, t0 = 0, time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < secs:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    mid = smi() if not FILES else ""
    stop.set(); th.join()
    ms = e0.elapsed_time(e1) / n
    tail = samples[len(samples) // 3:]              # after the clocks settled
    mean = lambda k: sum(s[k] for s in tail if k in s) / max(1, sum(1 for s in tail if k in s))
    print(f"{name:34s} {ms:7.3f} ms  {flops / ms / 1e9:6.1f} TF ({flops / ms / 1e9 / 157.3:.3f})  power {mean('power_uW') / 1e6:7.1f} W  sclk {mean('sclk_Hz') / 1e6:7.0f} MHz  "
          f"mclk {mean('mclk_Hz') / 1e6:6.0f} MHz  temp {mean('temp_mC') / 1e3:5.1f} C  ({len(tail)} samples)", flush=True)
    if mid:
        print(mid)


print(smi())
M, N, K = 16384, 6144, 1536
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(M, K, device="cuda", generator=g)
w = torch.randn(N, K, device="cuda", generator=g) * 0.03
b = torch.randn(N, device="cuda", generator=g)
out = torch.empty(M, N, device="cuda")
fl = 2.0 * M * N * K
for rep in range(2):
    run("idle (1 tiny kernel per loop)", lambda: out[:1].zero_(), 0.0)
    run("hipBLASLt F.linear", lambda: F.linear(x, w, b), fl)
    run("sg free order", lambda: ops.linear_f32(x, w, b, out=out), fl)
    run("sg MKL order", lambda: ops.linear_f32(x, w, b, mkl_order=True, out=out), fl)
    run("xe_gemm128 (MKL order)", lambda: ops.ex_linear(x, w, b, out=out, kernel="xe"), fl)
    xh, wh = x.half(), w.half()
    run("hipBLASLt f16 (4x the flops/clk)", lambda: F.linear(xh, wh), fl)
