"""VQ one-MFMA path on the reference run's encoder features (tests/golden/pipeline_b16.npz, tiled to N = 32768): main + finalize kernel
for several code-split counts (a flagged stream's exact re-scan walks tiles_per_split tiles).  Run under rocprofv3 --kernel-trace --stats
or read the HIP-event figures it prints (those include ~15 us of launch gap per kernel)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from selftoktokenizer_amd import ops, weights as W
cb = W._synth_tensor("encoder.quantizer._codebook.embed", (1, 32768, 16), "cpu")[0].contiguous().cuda()
pk = ops.vq_pack_codebook(cb)
zg = torch.from_numpy(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pipeline_b16.npz"))["z"]).reshape(-1, 16)
z = zg.repeat(4, 1).cuda()
for split in [int(a) for a in sys.argv[1:]] or [16, 32, 64]:
    for _ in range(3):
        ops.vq_encode(z, pk, packed=True, coarse=1, split=split)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.vq_encode(z, pk, packed=True, coarse=1, split=split)
    e.record(); torch.cuda.synchronize()
    print(json.dumps({"split": split, "both_launches_ms": round(s.elapsed_time(e) / 20, 4)}))
