"""Quick VQ kernel timing (HIP events) for both kernels at the BASELINE sizes."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selftoktokenizer_amd import ops, synth, weights as W

cb = W._synth_tensor("encoder.quantizer._codebook.embed", (1, 32768, 16), "cpu")[0].contiguous().cuda()
pk = ops.vq_pack_codebook(cb)
# features: synthetic rows, or (argv[1] == "enc") the Q-Former encoder's own features of synthetic images -- their top-1/top-2 gaps set how many
# candidates the wide window of the one-MFMA pass re-scores
import numpy as np
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pipeline_b16.npz")
real = len(sys.argv) > 1 and sys.argv[1] == "enc"
zg = torch.from_numpy(np.load(GOLD)["z"]).reshape(-1, 16) if real else None      # 8192 features of the reference's own encoder run
for n in ((32768,) if real else (512, 32768, 65536, 131072)):
    z = zg.repeat(n // zg.shape[0], 1).cuda() if real else synth.synthetic_vq_rows(n, device="cuda")
    for packed, name, coarse in ((False, "valu", None), (True, "mfma-fp32", False), (True, "f16-coarse(3 MFMAs)+exact", 3), (True, "f16-coarse(1 MFMA)+exact", 1)):
        c = pk if packed else cb
        for _ in range(3):
            ops.vq_encode(z, c, packed=packed, coarse=coarse)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        iters = 20
        for _ in range(iters):
            ops.vq_encode(z, c, packed=packed, coarse=coarse)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        fl = 2.0 * n * 32768 * 16
        rec = {"kernel": name, "N": n, "ms": round(ms, 4), "TFLOPs": round(fl / ms / 1e9, 1)}
        if packed:
            ids, lm, lf = ops.vq_encode_split_launch(z, pk, coarse=coarse)
            lm(); lf(); torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            tm = tf = 0.0
            for _ in range(iters):
                ev[0].record(); lm(); ev[1].record(); lf(); ev[2].record(); torch.cuda.synchronize()
                tm += ev[0].elapsed_time(ev[1]); tf += ev[1].elapsed_time(ev[2])
            rec.update(main_ms=round(tm / iters, 4), finalize_ms=round(tf / iters, 4), main_TFLOPs=round(fl / (tm / iters) / 1e9, 1))
        print(json.dumps(rec))
