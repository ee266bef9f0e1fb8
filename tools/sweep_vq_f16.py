import sys, os, json
sys.path.insert(0, "/root/repo")
import torch
from selftoktokenizer_amd import ops, synth, weights as W
cb = W._synth_tensor("encoder.quantizer._codebook.embed", (1, 32768, 16), "cpu")[0].contiguous().cuda()
pk = ops.vq_pack_codebook(cb)
n = 32768
z = synth.synthetic_vq_rows(n, device="cuda")
for rt in (2, 4):
  for sp in (4, 8, 16, 32, 64):
    f = lambda: ops.vq_encode(z, pk, packed=True, coarse=True, rt=rt, split=sp)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    print(rt, sp, round(s.elapsed_time(e) / 20, 4))
