"""Every candidate kernel of gemm_tune.py against hipBLASLt's default on the block-Linear shapes, with long timing (20 launches after 5):
how much the best candidate gains, and whether gemm_tune's own short probe would have found it.  GPU box."""
import os
import sys
import tempfile

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import gemm_tune as G  # noqa: E402

rows = [int(v) for v in sys.argv[1:]] or [16384, 22976]
tun = torch.cuda.tunable
tun.enable(True)
tun.tuning_enable(False)
dev = torch.device("cuda")
with tempfile.TemporaryDirectory() as td:
    for (N, K) in G.FAMILIES:
        w, b = torch.randn(N, K, device=dev) * 0.02, torch.randn(N, device=dev)
        for M in rows:
            res = {}
            for rnd in range(2):
                for ci, cand in enumerate((None,) + G.CANDIDATES):
                    Mp = M + 1 + 2 * ci
                    if cand is not None and rnd == 0:
                        p = os.path.join(td, f"p{N}_{K}_{M}_{ci}.csv")
                        G._write(p, [(G._key(N, Mp, K), cand)])
                        tun.read_file(p)
                    a = torch.randn(Mp, K, device=dev)
                    res[cand] = min(res.get(cand, 1e9), G._time(lambda: F.linear(a, w, b), n=20, warm=5))
                    del a
            fl = 2.0 * M * N * K
            best = min(res, key=res.get)
            line = " ".join(f"{(c or 'default').replace('Gemm_Hipblaslt_', '')}:{fl / t * 1e-9 / 157.3:.3f}" for c, t in res.items())
            print(f"[{M},{K}]x[{K},{N}]  best {(best or 'default')} {fl / res[best] * 1e-9 / 157.3:.3f} vs default {fl / res[None] * 1e-9 / 157.3:.3f} | {line}", flush=True)
