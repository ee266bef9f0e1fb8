"""tools: in-kernel timeline of csrc/gemm_fp32.hip (tools/stamp_sgemm.sh builds the stamped library): shader cycles a compute wave and a loader wave spend waiting at
the chunk barrier, issuing, and between barriers, for the first 40 chunks of two workgroups.   SELFTOK_HIP_LIB=tools/microbench/libselftok_sgstamp.so python tools/stamp_sgemm.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selftoktokenizer_amd import ops, _lib
M, K, N = 16384, 1536, 4608
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
ws = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
lib = _lib.load()
print("occupancy (workgroups per CU) free / MKL:", lib.selftok_sg_occupancy(0), lib.selftok_sg_occupancy(1))
for rep in range(12):          # back to back: the last launch's stamps are read (steady clocks)
    _lib.check(lib.selftok_linear_f32(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), None, 0, 0, None, 0, 0, out.data_ptr(), N, M, N, K, 0, ws.data_ptr(), 0,
                                      torch.cuda.current_stream().cuda_stream), "linear_f32")
    torch.cuda.synchronize()
st = ws[: 4 * 120 * 8].view(torch.int64).cpu().numpy().reshape(2, 2, 40, 3)
for wg in range(2):
    for wv, name in ((0, "compute wave 0"), (1, "loader wave 4 "),):
        t = st[wg, wv]
        t0 = st[wg, 0, 0, 0]
        wait = t[:, 1] - t[:, 0]
        work = t[:, 2] - t[:, 1]
        period = np.diff(t[:, 1])
        print(f"workgroup at list position {256 + 32 * wg}, {name}: first stamp at {int(t[0, 0] - t0)}; barrier wait per chunk (cycles) {wait[:24].tolist()}")
        print(f"    after-barrier work per chunk {work[:24].tolist()}")
        print(f"    barrier-to-barrier period {period[:24].tolist()}   mean of chunks 8..39: wait {wait[8:].mean():.0f} work {work[8:].mean():.0f} period {period[8:].mean():.0f}")
print("start offset between the two workgroups (cycles):", int(st[1, 0, 0, 0] - st[0, 0, 0, 0]))

# residency census: every workgroup's (entry, first barrier passed, k-loop end, exit) in us and its CU
nb = 8 * ((4608 + 7) // 8)
cen = ws[8192: 8192 + nb * 40].view(torch.int64).cpu().numpy().reshape(nb, 5)
cen = cen[cen[:, 1] > 0]
hw = cen[:, 2] & 0xFFFFFFFF
xcc = cen[:, 2] >> 32
cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5)
key = xcc * 1024 + cu
t0 = cen[:, 0].min()
print("workgroups recorded", len(cen), "distinct (xcc, cu) keys", len(set(key.tolist())), "kernel span (us)", (cen[:, 1].max() - t0) / 100.0)
us = lambda v: round((int(v) - int(t0)) / 100.0, 1)
for k0 in sorted(set(key.tolist()))[:2]:
    sel = cen[key == k0]
    sel = sel[np.argsort(sel[:, 0])]
    print(f"CU {k0}: (entry, loop start, loop end, exit) us:")
    for a, b, h, c, d in sel.tolist()[:12]:
        print(f"    entry {us(a):8.1f}  loop {us(c):8.1f} .. {us(d):8.1f} ({(d - c) / 100.0:6.1f})  exit {us(b):8.1f}   prologue {(c - a) / 100.0:6.1f}  epilogue {(b - d) / 100.0:6.1f}")
