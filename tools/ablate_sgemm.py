"""tools: timing-only ablations of csrc/gemm_fp32.hip's k-loop (tools/ablate_sgemm.sh builds the variants): which part of the loop costs the matrix pipe its idle
cycles.  Spawns itself once per variant (SELFTOK_HIP_LIB)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "product", 1: "no DMA (no issue, no vmcnt waits)", 2: "DMA issued, vmcnt waits removed", 3: "no DMA, no barriers", 4: "DMA + waits + barriers, no fragment reads"}
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, ROOT)
    from selftoktokenizer_amd import ops
    n = int(sys.argv[1])
    for (M, K, N) in ((16384, 1536, 4608), (16384, 6144, 1536)):
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        res = []
        for rep in range(2):                    # the first repetition warms the clocks
            for _ in range(3):
                ops.linear_f32(x, w, b, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.linear_f32(x, w, b, out=out)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            res.append(f"{ms:.3f} ms ({2.0 * M * N * K / ms / 1e9 / 157.3:.3f})")
        print(f"[{n}] {NAMES[n]:45s} [{M},{K}]x[{K},{N}]: " + "   ".join(res), flush=True)
else:
    for n in range(5):
        env = dict(os.environ)
        if n:
            env["SELFTOK_HIP_LIB"] = os.path.join(ROOT, "tools", "microbench", f"libselftok_sgabl{n}.so")
        subprocess.call([sys.executable, os.path.abspath(__file__), str(n)], env=env)
