"""B = 1 decode: the fp32 block Linears at the row counts a single image issues (image stream 256 rows, context stream k + 1 <= 513 rows),
hipBLASLt's default choice against TunableOp's exhaustive search, and the weight-streaming bound of each shape (fp32 weights once from HBM
at 8 TB/s).  GPU box.  Usage: python tools/probe_b1_gemm.py [rows...]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def ms(fn, n=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


rows = [int(v) for v in sys.argv[1:]] or [256, 64, 257, 513]
H = 1536
tun = torch.cuda.tunable
res = {}
for phase in ("default", "tuned"):
    if phase == "tuned":
        tun.enable(True)
        tun.tuning_enable(True)
        tun.set_max_tuning_iterations(20)
        tun.set_max_tuning_duration(20)
        tun.set_filename("/tmp/b1_tunableop.csv")
    for M in rows:
        a1, a4 = torch.randn(M, H, device="cuda"), torch.randn(M, 4 * H, device="cuda")
        for name, a, N, K in (("qkv", a1, 3 * H, H), ("proj", a1, H, H), ("fc1", a1, 4 * H, H), ("fc2", a4, H, 4 * H)):
            w, b = torch.randn(N, K, device="cuda") * 0.02, torch.randn(N, device="cuda")
            # rotate through 8 weight copies so the weights come from HBM as in the model (24 blocks x 8 Linears = 8.3 GB per step), not from L2 / MALL
            ws = [w.clone() for _ in range(8)]
            i = [0]

            def f():
                i[0] = (i[0] + 1) & 7
                return F.linear(a, ws[i[0]], b)
            F.linear(a, w, b)           # (tuned phase: the search happens here, on this key)
            res[(M, name, phase)] = ms(f)
            del ws
for M in rows:
    for name, N, K in (("qkv", 3 * H, H), ("proj", H, H), ("fc1", 4 * H, H), ("fc2", H, 4 * H)):
        t0, t1 = res[(M, name, "default")], res[(M, name, "tuned")]
        bound = (N * K * 4 + M * (N + K) * 4) / 8e12 * 1e3
        print(f"M={M:4d} {name:4s} N={N:5d} K={K:5d}: default {t0 * 1e3:7.1f} us  tuned {t1 * 1e3:7.1f} us  HBM bound {bound * 1e3:5.1f} us  "
              f"({2.0 * M * N * K / t1 * 1e-9:6.1f} TFLOP/s tuned)", flush=True)
    print(f"M={M}: sum default {sum(res[(M, n, 'default')] for n in ('qkv', 'proj', 'fc1', 'fc2')) * 1e3:.1f} us, tuned "
          f"{sum(res[(M, n, 'tuned')] for n in ('qkv', 'proj', 'fc1', 'fc2')) * 1e3:.1f} us")
print("selected:")
for r in tun.get_results():
    print("  ", r)
