"""tests/golden/silu_bf16_table.npy: torch-CPU's SiLU on every one of the 65536 bf16 bit patterns (uint16 -> uint16).
The reference applies `torch.nn.SiLU` to bf16 tensors (sd3_impls.py:247-251); ATen evaluates x / (1 + exp(-x)) in fp32 with Sleef's exp,
which is not correctly rounded: 17 inputs round differently than the exact function.  The function is position independent (checked
below on permuted / odd-length / multi-threaded tensors), so a table reproduces it bit for bit.  Run in the build container (CPU torch)."""
import os

import numpy as np
import torch

allb = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(torch.bfloat16)
t = torch.nn.functional.silu(allb.clone())
perm = torch.randperm(65536)
assert torch.equal(torch.nn.functional.silu(allb[perm].clone()).view(torch.int16), t.view(torch.int16)[perm])
assert torch.equal(torch.nn.SiLU(inplace=True)(allb[perm][:65521].reshape(1, 1, -1, 1).clone()).reshape(-1).view(torch.int16), t.view(torch.int16)[perm][:65521])
big = torch.nn.functional.silu(allb.repeat(64).reshape(4, 128, 64, 128)).reshape(64, 65536)
assert all(torch.equal(big[i].view(torch.int16), t.view(torch.int16)) for i in range(64))
x = allb.double()
exact = (x / (1 + torch.exp(-x))).bfloat16()
fin = torch.isfinite(allb.float())
diff = (exact.view(torch.int16) != t.view(torch.int16)) & fin
print("entries that differ from the correctly rounded function:", int(diff.sum()))
for i in torch.nonzero(diff)[:, 0].tolist():
    print(f"  in 0x{i:04x} ({float(allb[i]):+.6g}) -> torch 0x{int(t.view(torch.int16)[i]) & 0xffff:04x}, exact 0x{int(exact.view(torch.int16)[i]) & 0xffff:04x}")
out = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "silu_bf16_table.npy")
np.save(out, t.view(torch.int16).numpy().view(np.uint16))
print("wrote", out)
