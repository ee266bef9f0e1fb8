"""Launch the VQ argmax kernel a few times (for rocprofv3 --pmc passes).  N from argv (default 32768)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selftoktokenizer_amd import ops, synth, weights as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
cb = W._synth_tensor("encoder.quantizer._codebook.embed", (1, 32768, 16), "cpu")[0].contiguous().cuda()
pk = ops.vq_pack_codebook(cb)
z = synth.synthetic_vq_rows(n, device="cuda")
coarse = not (len(sys.argv) > 2 and sys.argv[2] == "fp32")      # default: f16 coarse pass + exact re-score; "fp32": round-1 kernel
for _ in range(5):
    ops.vq_encode(z, pk, packed=True, coarse=coarse)
torch.cuda.synchronize()
