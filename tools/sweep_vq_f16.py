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


for rt in (1, 2, 4):
    for sp in (4, 8, 16, 32):
        zz = z.contiguous()
        ids = torch.empty(n, dtype=torch.int64, device="cuda")
        ws = torch.empty(lib.selftok_vq_workspace_bytes(n, 32768), dtype=torch.uint8, device="cuda")
        flags = ops.VQ_F16COARSE | (rt << 8) | (sp << 16)
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
        if hasattr(lib._lib if hasattr(lib, "_lib") else lib, "selftok_tune_vq_stamp") or True:
            try:
                st2 = (ctypes.c_ulonglong * 2)()
                lib.selftok_tune_vq_stamp.restype = ctypes.c_int
                torch.cuda.synchronize()
                if lib.selftok_tune_vq_stamp(st2) == 0 and st2[1] > 0:
                    ghz = st2[0] / (st2[1] * 10.0) / 1.0        # cycles per 10 ns tick -> GHz
                    mfma_cyc = (32768 // 32 // sp) * rt * 3 * 32   # matrix-pipe cycles this wave's SIMD needs per wave (tiles x row blocks x 3 MFMAs x 32)
                    clk = f"  WG(0,0): {st2[0]} shader cycles in {st2[1] * 10} ns = {ghz:.2f} GHz; own MFMA issue {mfma_cyc} cycles = {mfma_cyc / st2[0]:.2f} of its lifetime"
            except AttributeError:
                pass
        print(f"RT={rt} split={sp:2d}: main {tm * 1e3:7.1f} us ({flops / tm / 1e9 / 2500:.3f} of the f16 peak)  finalize {tf * 1e3:6.1f} us  total {1e3 * (tm + tf):7.1f} us  ids ok: {ok}{clk}", flush=True)
        assert ok
