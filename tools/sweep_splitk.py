"""Small-M f16x2 Linear: time of (partial launch + finish launch) per split factor, at the model's four shapes and the row counts of one
image, replayed from a hipGraph of 20 calls (eager timing of 5-20 us kernels measures the host).  GPU box."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import ops  # noqa: E402

H = 1536
rows = [int(v) for v in sys.argv[1:]] or [256, 257, 513, 1024]
REP = 20


def graph_ms(fn):
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / REP)
    return best


for M in rows:
    for name, N, K in (("qkv", 3 * H, H), ("proj", H, H), ("fc1", 4 * H, H), ("fc2", H, 4 * H)):
        a = torch.randn(1, M, K, device="cuda")
        w, b = torch.randn(N, K, device="cuda") * 0.02, torch.randn(N, device="cuda")
        # eight weight copies in rotation: the model streams 8.3 GB of weights per step, nothing stays in L2 / MALL between uses
        packs = [ops.linear_f16x2_pack(w) for _ in range(8)]
        xs = ops.split_f16x2(a)
        kt = K // 32
        res = {}
        for s in [v for v in (1, 2, 3, 4, 6, 8, 12, 16, 24) if kt % v == 0 and kt // v >= 2]:
            i = [0]

            def f():
                i[0] = (i[0] + 1) & 7
                return ops.linear_f16x2_split(xs, packs[i[0]], b, N, ksplit=s)
            res[s] = round(1e3 * graph_ms(f), 1)
        best = min(res, key=res.get)
        print(json.dumps({"M": M, "linear": name, "us_by_ksplit": res, "best": best, "heuristic": ops.f16x2_ksplit(M, N, K)}), flush=True)
        del packs
