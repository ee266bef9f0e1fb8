"""selftoktokenizer_amd/data/encoder_pos_sincos.npy: the sinusoidal embedding of the token positions 1000 + 8 k (k < 1024) as the
REFERENCE evaluates it on the build container's CPU -- `timestep_embedding` (mimogpt/models/selftok/models.py:56-74) = torch.exp /
torch.cos / torch.sin on fp32 CPU tensors, which ATen hands to MKL's VML (vsExp / vsCos / vsSin).  VML is closed source, dispatches by CPU
vendor (the same call returns other bits on an AMD host: 5 % of the entries differ from the correctly rounded value here, another set
there), and cannot be restated; the table is input- and weight-independent, so it ships as DATA.  Run in the build container only.
    python tools/oracle/gen_pos_table.py"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
import ref_harness as H  # noqa: E402,F401  (stubs)

H.install()
from mimogpt.models.selftok.models import TimestepEmbedder  # noqa: E402

K = 1024
pos = 1000 + 8 * torch.arange(K)
table = TimestepEmbedder.timestep_embedding(pos, 256).float().numpy()
half = 128
freqs = torch.exp(-math.log(10000) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
args = pos[:, None].float() * freqs[None]
own = torch.cat([torch.cos(args), torch.sin(args)], dim=-1).numpy()
assert np.array_equal(table.view(np.uint32), own.view(np.uint32)), "the reference's timestep_embedding is not the formula this script documents"
out = os.path.join(ROOT, "selftoktokenizer_amd", "data", "encoder_pos_sincos.npy")
np.save(out, table)
print("wrote", out, table.shape, "cpu capability", torch.backends.cpu.get_cpu_capability())


# ---- the MMDiT's timestep embeddings of the DEFAULT sampler schedule (50 steps, start = 1.0): `t_embedder.timestep_embedding(t * 1000)` (sd3/mmdit.py:156-175,
# 999, 1022) and cfg_inference's `floor(t * 1000).int().clamp(0, 999)` (:1126) -- the same MKL-VML arithmetic, [2, 50, 256]: used by the exact-order MMDiT mode
from mimogpt.models.selftok.sd3.mmdit import TimestepEmbedder as DitTE  # noqa: E402
from selftoktokenizer_amd.schedule import FlowSchedule  # noqa: E402
fs = FlowSchedule(50, 1.0)
t = torch.from_numpy(fs.scheduled_t)
cond = DitTE.timestep_embedding(t * 1000.0, 256).float().numpy()
unc = DitTE.timestep_embedding(torch.floor(t * 1000).int().clamp(0, 999), 256).float().numpy()
out2 = os.path.join(ROOT, "selftoktokenizer_amd", "data", "flow50_t_sincos.npy")
np.save(out2, np.stack([cond, unc]))
print("wrote", out2, np.stack([cond, unc]).shape)
