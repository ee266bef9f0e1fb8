"""not gpu: the product's host modules (encoder, MMDiT, sampler) run END TO END on the CPU when their C-ABI calls go to the CPU twin
(oracle/twin_host.py) -- the path bench.py times as `cpu_baseline.twin`.  Checked against the reference pipeline's golden run."""
import os

import numpy as np
import pytest
import torch

from oracle import twin_host as TH
from selftoktokenizer_amd import ops, synth, weights as W

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_host_modules_on_the_cpu_twin_reproduce_the_reference_run():
    full = os.environ.get("SELFTOK_TWIN_FULL") == "1"          # + one sampler step through the MMDiT (2 minutes: 8.7 GB of hash-generated weights on the CPU)
    shapes = W.expected_shapes(512)
    if not full:
        shapes = {k: v for k, v in shapes.items() if k.startswith("encoder.")}
    sd = W.synthetic_state_dict(shapes, device="cpu")
    g16 = np.load(os.path.join(GOLD, "pipeline_b16.npz"))
    g1 = np.load(os.path.join(GOLD, "pipeline_b1.npz"))
    x0 = torch.from_numpy(g16["x0_bf16"][:1]).view(torch.bfloat16).float()
    with TH.on_cpu_twin():
        enc = TH.build_encoder(sd)
        _, ids = enc(x0, d=None)
        flips = ids.numpy() != g16["tokens"][:1].astype(np.int64)
        assert flips.sum() <= 1 and (g16["gap"][:1][flips] < 1e-4).all()            # fp32 summation order of the CPU twin's attention: at most a near-tie
        if full:
            _, dit, flow, ktab = TH.build(sd)
            ehs = enc.codes_ln(torch.from_numpy(g1["tokens"]))
            lat = flow.p_sample_loop(dit, synth.synthetic_noise(1), ehs, ktab, context_see_xt=True, max_steps=1)
            ref = torch.from_numpy(g1["lats"][list(g1["lat_steps"]).index(1)])
            assert float((lat - ref).abs().max()) < 2e-5
    # outside the block the product is itself again: CPU tensors are refused
    with pytest.raises(Exception):
        ops.clamp01_(torch.zeros(4, dtype=torch.bfloat16))
