"""GPU: where the cycles of one k-iteration of linear_f16x2_pre_kernel go (s_memtime stamps, ABL = 2048 build of
tools/microbench/libselftok_gemm_ablate.so): per wave of one mid-grid work-group, mean cycles per iteration of
[issue reads + DMA, reads landed] [vmcnt wait] [barrier 1] [24 MFMAs issued] [barrier 2]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SELFTOK_HIP_LIB"] = os.path.join(ROOT, "tools", "microbench", "libselftok_gemm_ablate.so")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from selftoktokenizer_amd import ops  # noqa: E402

M, N, K = 22912, 4608, 1536
a = torch.randn(M, K, device="cuda")
w = torch.randn(N, K, device="cuda") * 0.02
big = torch.zeros(N + 64, device="cuda")
b = big[:N]
packed = ops.linear_f16x2_pack(w)
xs = ops.split_f16x2(a)
os.environ["SELFTOK_GEMM_ABL"] = "2048"
ntiles = (M // 256 + (M % 256 > 0)) * (N // 128)
rec = torch.zeros(8 + ntiles * 8, dtype=torch.int32, device="cuda")
import time
for _ in range(3):
    ops.linear_f16x2_split(xs, packed, b, N, overflow=rec)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    ops.linear_f16x2_split(xs, packed, b, N, overflow=rec)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 10
print(f"instrumented kernel: {dt * 1e3:.3f} ms per launch, {M // 256 + (M % 256 > 0)} x {N // 128} tiles on 256 CUs")
st = big[N:].reshape(8, 8).cpu()
print("wave  issue+reads  vmcnt-wait  barrier1  mfma-issue  barrier2   sum | prologue  loop-total  epilogue")
for wv in range(8):
    r = st[wv].tolist()
    print(f"{wv:4d}  {r[0]:11.0f}  {r[1]:10.0f}  {r[2]:8.0f}  {r[3]:10.0f}  {r[4]:8.0f}  {sum(r[:5]):6.0f} | {r[5]:8.0f}  {r[6]:10.0f}  {r[7]:8.0f}")

# timeline of the last launch: per (XCC, HW_ID) = one CU, the work-groups it ran back to back
r = rec[8:].reshape(ntiles, 8).cpu().numpy().astype("int64") & 0xFFFFFFFF
start, end, cyc, hw, xcc = r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4] & 0xF
t0 = start.min()
dur = (end - start) % (1 << 32)
print(f"work-groups {ntiles}; kernel span {(((end - t0) % (1 << 32)).max()) / 100:.1f} us (100 MHz realtime counter)")
print(f"per work-group: {dur.mean() / 100:.2f} us mean, {cyc.mean():.0f} shader cycles mean -> {cyc.mean() / (dur.mean() / 100) / 1e3:.3f} GHz effective shader clock")
import collections
by_cu = collections.defaultdict(list)
cu_key = (xcc << 32) | (hw & 0xFFFFFF00 | 0)     # HW_ID bits: wave/simd in the low byte; CU / SH / SE above
for i in range(ntiles):
    by_cu[int(xcc[i]) << 40 | int(hw[i] >> 8 & 0xFFFFFF)].append(((start[i] - t0) % (1 << 32), (end[i] - t0) % (1 << 32)))
gaps, busy = [], []
for k, v in by_cu.items():
    v.sort()
    busy.append(sum(e - s_ for s_, e in v))
    gaps += [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
import numpy as np
gaps = np.array(gaps)
print(f"distinct CUs seen {len(by_cu)}; work-groups per CU {ntiles / len(by_cu):.2f}; gap between consecutive work-groups on a CU: mean {gaps.mean() / 100:.2f} us, "
      f"median {np.median(gaps) / 100:.2f}, max {gaps.max() / 100:.2f}; CU busy fraction {np.mean(busy) / ((end - t0) % (1 << 32)).max():.3f}")
