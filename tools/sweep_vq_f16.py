"""GPU: launch-shape sweep of the f16 coarse VQ path at N = 32768 (rows per wave RT x code splits): main kernel and finalize
kernel timed separately with HIP events, ids checked against the default launch shape."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SELFTOK_HIP_LIB", os.path.join(ROOT, "tools", "microbench", "libselftok_tune.so"))     # tune build: clock stamps
sys.path.insert(0, ROOT)
import ctypes  # noqa: E402
import torch  # noqa: E402
from selftoktokenizer_amd import _lib, ops, synth, weights as W  # noqa: E402

cb = W._synth_tensor("encoder.quantizer._codebook.embed", (1, 32768, 16), "cpu")[0].contiguous().cuda()
pk = ops.vq_pack_codebook(cb)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
z = synth.synthetic_vq_rows(n, device="cuda")
lib = _lib.load()
ref = ops.vq_encode(z, pk, packed=True, coarse=False)
flops = 3 * 2.0 * n * 32768 * 16


def ev(fn, reps=50):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


ONE = bool(os.environ.get("SWEEP_COARSE1"))          # the one-MFMA pass (round 4's default) instead of the three-MFMA one
SHAPES = [(4, 16)] if os.environ.get("SWEEP_DEFAULT_ONLY") else [(rt, sp) for rt in (1, 2, 4) for sp in (4, 8, 16, 32)]
for rt, sp in SHAPES:
    if True:
        zz = z.contiguous()
        ids = torch.empty(n, dtype=torch.int64, device="cuda")
        ws = torch.empty(lib.selftok_vq_workspace_bytes(n, 32768), dtype=torch.uint8, device="cuda")
        flags = ops.VQ_F16COARSE | (ops.VQ_F16COARSE1 if ONE else 0) | (rt << 8) | (sp << 16)
        ns = ctypes.c_int(0)
        st = torch.cuda.current_stream().cuda_stream

        def main():
            _lib.check(lib.selftok_vq_argmax_partial_packed_f32(zz.data_ptr(), pk.data_ptr(), ws.data_ptr(), ctypes.addressof(ns), n, 32768, 16, flags, st), "main")

        def fin():
            _lib.check(lib.selftok_vq_finalize_packed(ws.data_ptr(), zz.data_ptr(), pk.data_ptr(), ids.data_ptr(), None, n, 32768, 16, ns.value, flags, st), "fin")
        main()
        fin()
        ok = bool(torch.equal(ids, ref.reshape(-1)))
        tm, tf = ev(main), ev(fin)
        clk = ""
        try:
            nwg = min(4096, ((n + 128 * rt - 1) // (128 * rt)) * sp)
            st2 = (ctypes.c_ulonglong * (2 + 3 * 4096))()
            lib.selftok_tune_vq_stamp.restype = ctypes.c_int
            torch.cuda.synchronize()
            if lib.selftok_tune_vq_stamp(st2, 2 + 3 * 4096) == 0 and st2[1] > 0:
                ghz = st2[0] / (st2[1] * 10.0)                   # shader cycles per 10 ns tick
                mfma_cyc = (32768 // 32 // sp) * rt * (1 if ONE else 3) * 32      # matrix-pipe cycles one wave of the workgroup issues (tiles x row blocks x 3 MFMAs x 32)
                import numpy as np
                a = np.array(st2[2:2 + 3 * nwg], dtype=np.uint64).reshape(nwg, 3)
                t0, t1 = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
                base = t0.min()
                span = (t1.max() - base) * 10e-3                  # us
                life = (t1 - t0) * 10e-3
                mid = base + (t1.max() - base) // 2
                alive_mid = int(((t0 <= mid) & (t1 > mid)).sum())
                started_1us = int((t0 <= base + 100).sum())
                hw = (a[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
                xcc = (a[:, 2] >> np.uint64(32)).astype(np.int64)
                cu = xcc * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 15) * 16      # (xcc, se, sh, cu)
                if os.environ.get("SWEEP_CENSUS") and rt == 4 and sp == 16:
                    import collections
                    per_cu = collections.Counter(cu.tolist())
                    print("    WGs per CU histogram:", sorted(collections.Counter(per_cu.values()).items()), " distinct CUs:", len(per_cu))
                    for x in range(8):
                        sel = xcc == x
                        print(f"    XCC {x}: {int(sel.sum())} WGs, lifetime median {np.median(life[sel]):.1f} us, min {life[sel].min():.1f}, max {life[sel].max():.1f}")
                    cnt = np.array([per_cu[c] for c in cu.tolist()])
                    for k in sorted(set(cnt.tolist())):
                        print(f"    WGs on a CU holding {k}: lifetime median {np.median(life[cnt == k]):.1f} us")
                    simd = (hw >> 4) & 3
                    print("    SIMD of each WG's wave 0:", sorted(collections.Counter(simd.tolist()).items()))
                clk = (f"  WG(0,0) {ghz:.2f} GHz, own MFMA issue {mfma_cyc / st2[0]:.2f} of its lifetime | {nwg} WGs: span {span:.1f} us, WG lifetime median {np.median(life):.1f} us "
                       f"(min {life.min():.1f} max {life.max():.1f}), alive at mid-kernel {alive_mid}, started within the first 1 us {started_1us}")
        except AttributeError:
            pass
        print(f"RT={rt} split={sp:2d}: main {tm * 1e3:7.1f} us ({flops / tm / 1e9 / 2500:.3f} of the f16 peak)  finalize {tf * 1e3:6.1f} us  total {1e3 * (tm + tf):7.1f} us  ids ok: {ok}{clk}", flush=True)
        assert ok
