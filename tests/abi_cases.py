"""One table of C-ABI calls (include/selftok_hip.h) that can be issued against EITHER build of the interface:
the gfx950 library (selftoktokenizer_amd/libselftok_hip.so, device pointers) or its CPU twin (oracle/libselftok_cpu.so, host
pointers).  tests/test_cpu_twin.py pins the twin to independent references on the CPU;
tests/test_cpu_twin_gpu.py runs every case on both and compares -- bit for bit where `exact`, else within `tol`.

A case is a function  case(alloc) -> (symbol, ctypes args, {name: buffer})  where alloc(ndarray) copies the array to the side under
test and returns a buffer with `.ptr` and `.numpy()` (the runner keeps every buffer alive across the call).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from selftoktokenizer_amd import _lib

I32, PRENORMED, F16COARSE = 1, 2, 8
GELU = 1
ATTN_F16X2 = 1


class Host:
    def __init__(self, a):
        self.a = np.ascontiguousarray(a).copy()
        self.ptr = self.a.ctypes.data

    def numpy(self):
        return self.a


class Dev:
    def __init__(self, a):
        import torch
        self.t = torch.from_numpy(np.ascontiguousarray(a).copy()).cuda()
        self.ptr = self.t.data_ptr()

    def numpy(self):
        return self.t.cpu().numpy()


def bind(path):
    lib = C.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def rng(seed):
    return np.random.default_rng(seed)


def f32(a):
    return np.asarray(a, dtype=np.float32)


def to_bf16(a):
    """fp32 -> bf16 bit patterns (uint16), round to nearest even (torch's conversion: the same rounding, an order of magnitude faster
    than doing it in numpy integers)"""
    import torch
    return np.ascontiguousarray(torch.from_numpy(np.ascontiguousarray(f32(a))).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16))


def from_bf16(u):
    return (u.astype(np.uint32) << 16).view(np.float32)


def unit_rows(a):
    return f32(a / np.linalg.norm(a, axis=-1, keepdims=True))


CASES = {}


def case(name, exact=True, tol=0.0, compare=None):
    def deco(fn):
        CASES[name] = dict(fn=fn, exact=exact, tol=tol, compare=compare)
        return fn
    return deco


# ---- VQ ---------------------------------------------------------------------------------------------------------------------------
def _vq_inputs(seed, n=300, c=1024):
    r = rng(seed)
    z = f32(r.standard_normal((n, 16)))
    cb = unit_rows(r.standard_normal((c, 16)))
    z[7] = cb[33] * 2.5                       # an exact hit
    z[11] = 0.0                               # zero row: every score 0 -> first code
    return z, cb


def _vq_encode(flags):
    def fn(alloc):
        z, cb = _vq_inputs(1)
        if flags & PRENORMED:
            z = unit_rows(np.where(np.abs(z).sum(-1, keepdims=True) == 0, 1.0, z))
        n, c = z.shape[0], cb.shape[0]
        ids = alloc(np.zeros(n, np.int32 if flags & I32 else np.int64))
        best = alloc(np.zeros(n, np.float32))
        ws = alloc(np.zeros(128 * n, np.uint64))
        return "selftok_vq_encode_f32", [alloc(z).ptr, alloc(cb).ptr, ids.ptr, best.ptr, ws.ptr, n, c, 16, flags, None], dict(ids=ids, best=best)
    return fn


for _f in (0, I32, PRENORMED, I32 | PRENORMED):
    case(f"vq_encode_flags{_f}")(_vq_encode(_f))


@case("vq_encode_nan_rows")
def _(alloc):
    z, cb = _vq_inputs(2, n=64, c=256)
    z[3, 5] = np.nan
    cb = cb.copy()
    cb[100, 2] = np.nan                        # NaN code: every row's first NaN score is at code 100
    ids, best = alloc(np.zeros(64, np.int64)), alloc(np.zeros(64, np.float32))
    ws = alloc(np.zeros(128 * 64, np.uint64))
    return "selftok_vq_encode_f32", [alloc(z).ptr, alloc(cb).ptr, ids.ptr, best.ptr, ws.ptr, 64, 256, 16, 0, None], dict(ids=ids, best=best)


@case("vq_pack_codebook")
def _(alloc):
    _, cb = _vq_inputs(3)
    packed = alloc(np.zeros((cb.size + 64) + cb.size, np.float32))      # selftok_vq_packed_bytes / 4
    return "selftok_vq_pack_codebook", [alloc(cb).ptr, packed.ptr, cb.shape[0], 16, None], dict(packed=packed)


def _vq_packed(flags, seed=4, spoil=None):
    """pack + encode on the side under test; the packed image itself is pinned by vq_pack_codebook"""
    def fn(alloc):
        z, cb = _vq_inputs(seed)
        if spoil == "norm":
            cb = cb.copy(); cb[5] *= 1.7                                   # not unit-norm: the f16 coarse pass must fall back
        n, c = z.shape[0], cb.shape[0]
        packed = alloc(np.zeros((cb.size + 64) + cb.size, np.float32))
        ids = alloc(np.zeros(n, np.int32 if flags & I32 else np.int64))
        best = alloc(np.zeros(n, np.float32))
        ws = alloc(np.zeros(128 * n, np.uint64))
        pre = ("selftok_vq_pack_codebook", [alloc(cb).ptr, packed.ptr, c, 16, None])
        return ("selftok_vq_encode_packed_f32", [alloc(z).ptr, packed.ptr, ids.ptr, best.ptr, ws.ptr, n, c, 16, flags, None], dict(ids=ids, best=best), [pre])
    return fn


case("vq_encode_packed_fp32")(_vq_packed(0))
case("vq_encode_packed_i32")(_vq_packed(I32))
case("vq_encode_packed_f16coarse")(_vq_packed(F16COARSE))
case("vq_encode_packed_f16coarse_nonunit_codebook")(_vq_packed(F16COARSE, seed=5, spoil="norm"))


@case("code_gather_ln")
def _(alloc):
    r = rng(6)
    cb = unit_rows(r.standard_normal((512, 16)))
    ids = r.integers(0, 512, 200).astype(np.int64)
    ids[:3] = [-1, 511, 0]
    w, b = f32(1 + 0.1 * r.standard_normal(16)), f32(0.1 * r.standard_normal(16))
    out = alloc(np.zeros((200, 16), np.float32))
    return "selftok_code_gather_ln_f32", [alloc(ids).ptr, alloc(cb).ptr, alloc(w).ptr, alloc(b).ptr, out.ptr, 200, 512, 16, 1e-5, 0, None], dict(out=out)


@case("code_gather_plain_i32")
def _(alloc):
    r = rng(7)
    cb = f32(r.standard_normal((128, 16)))
    ids = r.integers(0, 128, 50).astype(np.int32)
    out = alloc(np.zeros((50, 16), np.float32))
    return "selftok_code_gather_ln_f32", [alloc(ids).ptr, alloc(cb).ptr, None, None, out.ptr, 50, 128, 16, 1e-5, I32, None], dict(out=out)


@case("vq_ema_accumulate", exact=False, tol=2e-6)          # fp32 atomics: the order of the adds is not defined on the GPU
def _(alloc):
    r = rng(8)
    z = f32(r.standard_normal((400, 16)))
    ids = r.integers(0, 64, 400).astype(np.int64)
    bins, esum = alloc(np.zeros(64, np.float32)), alloc(np.zeros((64, 16), np.float32))
    return "selftok_vq_ema_accumulate_f32", [alloc(z).ptr, alloc(ids).ptr, bins.ptr, esum.ptr, 400, 64, 16, 0, None], dict(bins=bins, esum=esum)


@case("vq_tpc_update")
def _(alloc):
    r = rng(9)
    tpc = alloc(f32(r.random((8, 64))))
    ids = r.integers(0, 64, (6, 8)).astype(np.int64)
    return "selftok_vq_tpc_update_f32", [tpc.ptr, alloc(ids).ptr, 6, 8, 64, 0.25, 0, None], dict(tpc=tpc)


@case("vq_tpc_update_heavy_weight")
def _(alloc):
    r = rng(10)
    tpc = alloc(f32(r.random((4, 32))))
    ids = r.integers(0, 32, (3, 4)).astype(np.int32)
    return "selftok_vq_tpc_update_f32", [tpc.ptr, alloc(ids).ptr, 3, 4, 32, 0.75, I32, None], dict(tpc=tpc)


def _softmax_inputs(seed, B, K, C):
    r = rng(seed)
    return f32(r.standard_normal((B, K, 16))), unit_rows(r.standard_normal((C, 16)))


def _softmax_stats(B, K, C, flags=0, colmean=True, seed=13):
    def fn(alloc):
        z, cb = _softmax_inputs(seed, B, K, C)
        if flags & PRENORMED:
            z = unit_rows(z)
        rows, cm = alloc(np.zeros((B * K, 2), np.float32)), alloc(np.zeros((K, C), np.float32))
        ws = alloc(np.zeros(8 * B * K * 36, np.float32))
        outs = dict(rowstats=rows)
        if colmean:
            outs["colmean"] = cm
        return "selftok_vq_softmax_stats_f32", [alloc(z).ptr, alloc(cb).ptr, rows.ptr, cm.ptr if colmean else None, ws.ptr, B, K, C, 16, 10.0, flags, None], outs
    return fn


def _softmax_backward(B, K, C, flags=0, seed=14):
    def fn(alloc):
        z, cb = _softmax_inputs(seed, B, K, C)
        if flags & PRENORMED:
            z = unit_rows(z)
        g = f32(rng(seed + 1).standard_normal((K, C)))
        rows, grad = alloc(np.zeros((B * K, 2), np.float32)), alloc(np.zeros((B, K, 16), np.float32))
        ws = alloc(np.zeros(8 * B * K * 36, np.float32))
        zz, cc = alloc(z), alloc(cb)
        pre = [("selftok_vq_softmax_stats_f32", [zz.ptr, cc.ptr, rows.ptr, None, ws.ptr, B, K, C, 16, 10.0, flags, None])]
        return ("selftok_vq_softmax_backward_f32", [zz.ptr, cc.ptr, rows.ptr, alloc(g).ptr, grad.ptr, ws.ptr, B, K, C, 16, 10.0, flags, None], dict(grad_z=grad), pre)
    return fn


case("vq_softmax_stats", exact=False, tol=2e-6)(_softmax_stats(5, 7, 300))
case("vq_softmax_stats_ragged_large", exact=False, tol=2e-6)(_softmax_stats(70, 3, 2051))
case("vq_softmax_stats_prenormed_rows_only", exact=False, tol=2e-6)(_softmax_stats(4, 9, 512, PRENORMED, colmean=False))
case("vq_softmax_backward", exact=False, tol=5e-6)(_softmax_backward(5, 7, 300))
case("vq_softmax_backward_ragged_large", exact=False, tol=5e-6)(_softmax_backward(70, 3, 2051))
case("vq_softmax_backward_prenormed", exact=False, tol=5e-6)(_softmax_backward(4, 9, 512, PRENORMED))


# ---- fused residual / LayerNorm / modulate ----------------------------------------------------------------------------------------
def _ln(H, B, T, resid, gated, mod, per_token, split=False, seed=11):
    def fn(alloc):
        r = rng(seed + H + 7 * per_token)
        x = f32(r.standard_normal((B, T, H)))
        y = f32(r.standard_normal((B, T, H))) if resid else None
        if per_token:      # tables [T, 3H]: batch stride 0, token stride 3H
            tab = f32(0.3 * r.standard_normal((T, 3 * H))); msb, mst = 0, 3 * H
        else:              # tables [B, 3H]
            tab = f32(0.3 * r.standard_normal((B, 3 * H))); msb, mst = 3 * H, 0
        t = alloc(tab)
        shift = t.ptr if mod else None
        scale = t.ptr + 4 * H if mod else None
        gate = t.ptr + 8 * H if (resid and gated) else None
        xo = alloc(np.zeros_like(x)) if resid else None
        outs = {}
        if resid:
            outs["x_out"] = xo
        if split:
            nb = alloc(np.zeros(((B * T + 15) // 16) * 16 * H * 2, np.uint16))
            ov = alloc(np.zeros(1, np.int32))
            outs.update(n_blk=nb, overflow=ov)
            args = [alloc(x).ptr, alloc(y).ptr if resid else None, gate, shift, scale, xo.ptr if resid else None, nb.ptr, ov.ptr, B, T, H, msb, mst, msb, mst, 1e-6, None]
            return "selftok_residual_ln_mod_split", args, outs
        n = alloc(np.zeros_like(x))
        outs["n_out"] = n
        args = [alloc(x).ptr, alloc(y).ptr if resid else None, gate, shift, scale, xo.ptr if resid else None, n.ptr, B, T, H, msb, mst, msb, mst, 1e-6, None]
        return "selftok_residual_ln_mod_f32", args, outs
    return fn


for _H in (64, 256, 512, 1024, 1536):
    case(f"ln_mod_H{_H}_resid_gate_mod_per_sample")(_ln(_H, 3, 21, True, True, True, False))
    case(f"ln_mod_H{_H}_resid_gate_mod_per_token")(_ln(_H, 5, 19, True, True, True, True))
    case(f"ln_mod_H{_H}_plain_ln")(_ln(_H, 2, 17, False, False, False, False))
case("ln_mod_H1536_resid_nogate")(_ln(1536, 2, 33, True, False, True, False))
case("ln_mod_H1536_walk_many_rows")(_ln(1536, 16, 96, True, True, True, True, seed=12))
for _H in (512, 1536):
    case(f"ln_mod_split_H{_H}_per_sample")(_ln(_H, 2, 32, True, True, True, False, split=True))
    case(f"ln_mod_split_H{_H}_per_token")(_ln(_H, 4, 16, False, False, True, True, split=True))


# ---- small element-wise kernels ---------------------------------------------------------------------------------------------------
@case("bias_gelu", exact=False, tol=1e-6)
def _(alloc):
    r = rng(20)
    h = alloc(f32(2 * r.standard_normal((37, 64))))
    return "selftok_bias_gelu_f32", [h.ptr, alloc(f32(r.standard_normal(64))).ptr, 37, 64, None], dict(h=h)


@case("silu", exact=False, tol=1e-6)
def _(alloc):
    x = f32(3 * rng(21).standard_normal(1000))
    out = alloc(np.zeros_like(x))
    return "selftok_silu_f32", [alloc(x).ptr, out.ptr, 1000, None], dict(out=out)


@case("add_rows")
def _(alloc):
    r = rng(22)
    x, t = f32(r.standard_normal((3, 10, 8))), f32(r.standard_normal((10, 8)))
    out = alloc(np.zeros_like(x))
    return "selftok_add_rows_f32", [alloc(x).ptr, alloc(t).ptr, out.ptr, 3, 80, None], dict(out=out)


@case("timestep_embed", exact=False, tol=2e-6)
def _(alloc):
    t = f32([0.0, 1.0, 333.0, 999.0, 0.62])
    freqs = f32(np.exp(-np.log(10000.0) * np.arange(128) / 128))
    out = alloc(np.zeros((5, 256), np.float32))
    return "selftok_timestep_embed_f32", [alloc(t).ptr, alloc(freqs).ptr, out.ptr, 5, 256, 1.0, None], dict(out=out)


@case("patchify")
def _(alloc):
    x = f32(rng(23).standard_normal((2, 16, 8, 12)))
    out = alloc(np.zeros((2, 24, 64), np.float32))
    return "selftok_patchify_f32", [alloc(x).ptr, out.ptr, 2, 16, 8, 12, None], dict(out=out)


def _unpatch(cfg):
    def fn(alloc):
        r = rng(24)
        yc, yu = f32(r.standard_normal((2, 24, 64))), f32(r.standard_normal((2, 24, 64)))
        x = f32(r.standard_normal((2, 16, 8, 12)))
        xo, vo = alloc(np.zeros_like(x)), alloc(np.zeros_like(x))
        return ("selftok_unpatchify_cfg_euler_f32", [alloc(yc).ptr, alloc(yu).ptr if cfg else None, alloc(x).ptr, xo.ptr, vo.ptr, 2, 16, 4, 6, 0.02, 3.5, None],
                dict(x_out=xo, v_out=vo))
    return fn


case("unpatchify_euler")(_unpatch(False))
case("unpatchify_cfg_euler")(_unpatch(True))


@case("rmsnorm", exact=False, tol=1e-6)
def _(alloc):
    r = rng(25)
    x, w = f32(r.standard_normal((40, 64))), f32(1 + 0.1 * r.standard_normal(64))
    out = alloc(np.zeros_like(x))
    return "selftok_rmsnorm_f32", [alloc(x).ptr, alloc(w).ptr, out.ptr, 40, 64, 1e-6, None], dict(out=out)


@case("rotary", exact=False, tol=2e-6)
def _(alloc):
    r = rng(26)
    t, f = f32(r.standard_normal((24, 32))), f32(3 * r.standard_normal((12, 32)))
    out = alloc(np.zeros_like(t))
    return "selftok_rotary_f32", [alloc(t).ptr, alloc(f).ptr, out.ptr, 24, 12, 32, 0.5, None], dict(out=out)


# ---- fp32-equivalent Linear on fp16 pairs -----------------------------------------------------------------------------------------
@case("split_f16x2")
def _(alloc):
    x = f32(rng(30).standard_normal((40, 96)) * 3)
    blk = alloc(np.zeros(48 * 96 * 2, np.uint16))
    ov = alloc(np.zeros(1, np.int32))
    return "selftok_split_f16x2_f32", [alloc(x).ptr, 96, blk.ptr, 40, 96, ov.ptr, None], dict(blk=blk, overflow=ov)


@case("split_f16x2_overflow")
def _(alloc):
    x = f32(rng(31).standard_normal((16, 32)))
    x[3, 4] = 1e6
    blk = alloc(np.zeros(16 * 32 * 2, np.uint16))
    ov = alloc(np.zeros(1, np.int32))
    return "selftok_split_f16x2_f32", [alloc(x).ptr, 32, blk.ptr, 16, 32, ov.ptr, None], dict(overflow=ov)


@case("linear_f16x2_pack_weight")
def _(alloc):
    w = f32(rng(32).standard_normal((256, 96)) * 0.1)
    packed = alloc(np.zeros(256 * 96 * 2, np.uint16))
    ov = alloc(np.zeros(1, np.int32))
    return "selftok_linear_f16x2_pack_weight", [alloc(w).ptr, packed.ptr, 256, 96, ov.ptr, None], dict(packed=packed, overflow=ov)


def _linear(kind, flags=0, M=70, N=256, K=160, seed=33, ksplit=0):
    """ksplit > 0: the small-M entry points (selftok_linear_f16x2_split_k / _split_residual_k) with that many work-groups per tile"""
    def fn(alloc):
        r = rng(seed)
        a, w, bias = f32(r.standard_normal((M, K))), f32(r.standard_normal((N, K)) / np.sqrt(K)), f32(r.standard_normal(N))
        packed = alloc(np.zeros(N * K * 2, np.uint16))
        ov = alloc(np.zeros(1, np.int32))
        pre = [("selftok_linear_f16x2_pack_weight", [alloc(w).ptr, packed.ptr, N, K, ov.ptr, None])]
        out = alloc(np.zeros((M, N), np.float32))
        outs = dict(out=out, overflow=ov)
        if kind == "f32":
            return "selftok_linear_f16x2_f32", [alloc(a).ptr, K, packed.ptr, alloc(bias).ptr, out.ptr, N, M, N, K, flags, ov.ptr, None], outs, pre
        ablk = alloc(np.zeros(((M + 15) // 16) * 16 * K * 2, np.uint16))
        pre.append(("selftok_split_f16x2_f32", [alloc(a).ptr, K, ablk.ptr, M, K, ov.ptr, None]))
        ws = alloc(np.zeros(max(ksplit, 1) * M * N, np.float32))
        ktail = [ksplit, ws.ptr] if ksplit else []
        sfx = "_k" if ksplit else ""
        if kind == "split":
            return "selftok_linear_f16x2_split" + sfx, [ablk.ptr, packed.ptr, alloc(bias).ptr, out.ptr, None, N, M, N, K, flags] + ktail + [ov.ptr, None], outs, pre
        if kind == "split_to_split":
            oblk = alloc(np.zeros(((M + 15) // 16) * 16 * N * 2, np.uint16))
            return ("selftok_linear_f16x2_split" + sfx, [ablk.ptr, packed.ptr, alloc(bias).ptr, None, oblk.ptr, N, M, N, K, flags] + ktail + [ov.ptr, None],
                    dict(out_blk=oblk, overflow=ov, _rows=M, _cols=N), pre)
        T = 35
        resid, gate = f32(r.standard_normal((M, N))), f32(r.standard_normal((M // T, N)))
        return ("selftok_linear_f16x2_split_residual" + sfx, [ablk.ptr, packed.ptr, alloc(bias).ptr, alloc(resid).ptr, N, alloc(gate).ptr if kind == "resid_gate" else None, N, 0, T,
                                                              out.ptr, N, M, N, K] + ktail + [ov.ptr, None], outs, pre)
    return fn


case("linear_f16x2_f32", exact=False, tol=1e-6)(_linear("f32"))
case("linear_f16x2_f32_gelu", exact=False, tol=1e-6)(_linear("f32", GELU))
case("linear_f16x2_split", exact=False, tol=1e-6)(_linear("split"))
case("linear_f16x2_split_to_split", exact=False, tol=1e-6)(_linear("split_to_split", GELU))
case("linear_f16x2_split_residual_gate", exact=False, tol=1e-6)(_linear("resid_gate"))
case("linear_f16x2_split_residual_nogate", exact=False, tol=1e-6)(_linear("resid"))
case("linear_f16x2_split_k2", exact=False, tol=1e-6)(_linear("split", K=192, ksplit=2))
case("linear_f16x2_split_k3_to_split_gelu", exact=False, tol=1e-6)(_linear("split_to_split", GELU, K=192, ksplit=3))
case("linear_f16x2_split_k6_nobias_rows", exact=False, tol=1e-6)(_linear("split", M=300, K=192, ksplit=6, seed=35))
case("linear_f16x2_split_residual_k2_gate", exact=False, tol=1e-6)(_linear("resid_gate", K=192, ksplit=2))
case("linear_f16x2_split_residual_k3_nogate", exact=False, tol=1e-6)(_linear("resid", K=192, ksplit=3))


# ---- attention --------------------------------------------------------------------------------------------------------------------
def _attn(B, H, Dh, n0, n1, kvis=None, sees=1, mode=0, blk=False, seed=40):
    def fn(alloc):
        r = rng(seed)
        W = H * Dh
        desc = _lib.AttnDesc()
        outs, keep = {}, [desc]
        for sg, n in ((0, n0), (1, n1)):
            s = desc.seg[sg]
            s.len = n
            if n == 0:
                continue
            qkv = alloc(f32(r.standard_normal((B, n, 3 * W))))           # fused [q | k | v] rows, like the block's qkv Linear output
            o = alloc(np.zeros((B, n, W), np.float32))
            s.q, s.k, s.v, s.o = qkv.ptr, qkv.ptr + 4 * W, qkv.ptr + 8 * W, o.ptr
            s.q_rs = s.k_rs = s.v_rs = 3 * W
            s.q_bs = s.k_bs = s.v_bs = n * 3 * W
            s.o_rs, s.o_bs = W, n * W
            outs[f"o{sg}"] = o
            keep.append(qkv)
            if blk:
                ob = alloc(np.zeros(((B * n + 15) // 16) * 16 * W * 2, np.uint16))
                desc.o_blk[sg] = ob.ptr
                s.o = None
                outs[f"o{sg}"] = ob
                outs[f"_blk{sg}"] = (B * n, W)
        desc.B, desc.H, desc.head_dim = B, H, Dh
        if kvis is not None:
            kv = alloc(np.asarray(kvis, np.int32))
            desc.kvis = kv.ptr
            keep.append(kv)
            outs["_kvis"] = np.asarray(kvis)
        desc.seg0_sees_seg1 = sees
        desc.scale = Dh ** -0.5
        desc.mode = mode
        ov = alloc(np.zeros(1, np.int32))
        desc.overflow = ov.ptr
        outs["overflow"] = ov
        outs["_keep"] = keep
        return "selftok_attn_f32", [C.addressof(desc), None], outs
    return fn


case("attn_two_segments_masked", exact=False, tol=2e-6)(_attn(2, 3, 64, 70, 40, kvis=[69, 12]))
case("attn_two_segments_full", exact=False, tol=2e-6)(_attn(1, 2, 64, 33, 64))
case("attn_context_blind_to_image", exact=False, tol=2e-6)(_attn(2, 2, 64, 48, 32, kvis=[47, 0], sees=0))
case("attn_single_segment_dim16", exact=False, tol=2e-6)(_attn(3, 4, 16, 0, 50))
case("attn_f16x2_mode", exact=False, tol=4e-6)(_attn(2, 2, 64, 40, 64, kvis=[39, 7], mode=ATTN_F16X2))
case("attn_f16x2_mode_split_out", exact=False, tol=4e-6)(_attn(2, 2, 64, 32, 64, kvis=[31, 15], mode=ATTN_F16X2, blk=True))


# ---- bf16 epilogues of the VAE ----------------------------------------------------------------------------------------------------
def _gn(silu):
    def fn(alloc):
        r = rng(50)
        x = to_bf16(r.standard_normal((2, 64, 8, 8)) * 2 + 0.3)
        w, b = to_bf16(1 + 0.2 * r.standard_normal(64)), to_bf16(0.2 * r.standard_normal(64))
        out = alloc(np.zeros_like(x))
        return "selftok_groupnorm_silu_bf16", [alloc(x).ptr, alloc(w).ptr, alloc(b).ptr, out.ptr, 2, 64, 64, 32, 1e-6, silu, None], dict(out_bf16=out)
    return fn


case("groupnorm_silu_bf16", exact=False, tol=0.0)(_gn(1))       # compared in bf16 ulps (see compare_outputs)
case("groupnorm_bf16", exact=False, tol=0.0)(_gn(0))


@case("latent_process_in")
def _(alloc):
    m = to_bf16(rng(51).standard_normal((2, 32, 64)) * 3)
    out = alloc(np.zeros((2, 16, 64), np.float32))
    return "selftok_latent_process_in", [alloc(m).ptr, out.ptr, 2, 32, 16, 64, 0.0609, 1.5305, None], dict(out=out)


@case("latent_process_out")
def _(alloc):
    z = f32(rng(52).standard_normal(2048) * 2)
    out = alloc(np.zeros(2048, np.uint16))
    return "selftok_latent_process_out", [alloc(z).ptr, out.ptr, 2048, 0.0609, 1.5305, None], dict(out=out)


@case("clamp01_bf16")
def _(alloc):
    x = to_bf16(rng(53).standard_normal(4096) * 1.5)
    x[5] = 0x7FC0
    img = alloc(x)
    return "selftok_clamp01_bf16", [img.ptr, 4096, None], dict(img=img)


# ---- channels-last bf16 convolution / GroupNorm (csrc/conv.hip) -------------------------------------------------------------------
def _conv(B, H, W, Cin, Cout, ks=3, stride=1, up=0, bn=128, resid=False, cin_store=None, cstore=None, seed=60):
    def fn(alloc):
        r = rng(seed + Cin + Cout)
        cin_s = cin_store or Cin
        x = to_bf16(r.standard_normal((B, H, W, cin_s)))
        w = to_bf16(r.standard_normal((Cout, Cin, ks, ks)) / np.sqrt(Cin * ks * ks))
        bias = to_bf16(r.standard_normal(Cout))
        cs = cstore or (Cout + 3) // 4 * 4
        Hi, Wi = H << up, W << up
        Ho, Wo = (Hi // 2, Wi // 2) if stride == 2 else (Hi, Wi)
        packed = alloc(np.zeros(((Cout + bn - 1) // bn) * ((Cin + 31) // 32) * ks * ks * bn * 32, np.uint16))
        out = alloc(np.zeros((B, Ho, Wo, cs), np.uint16))
        res = alloc(to_bf16(r.standard_normal((B, Ho, Wo, cs)))) if resid else None
        pre = [("selftok_conv2d_pack_weight_bf16", [alloc(w).ptr, packed.ptr, Cout, Cin, ks, bn, None])]
        return ("selftok_conv2d_nhwc_bf16", [alloc(x).ptr, packed.ptr, alloc(bias).ptr, res.ptr if resid else None, out.ptr, B, H, W, cin_s, Cout, cs, cs, ks, stride, up, bn, None],
                dict(out_bf16=out), pre)
    return fn


@case("conv2d_pack_weight")
def _(alloc):
    w = to_bf16(rng(61).standard_normal((40, 24, 3, 3)))
    packed = alloc(np.zeros(2 * 1 * 9 * 32 * 32, np.uint16))
    return "selftok_conv2d_pack_weight_bf16", [alloc(w).ptr, packed.ptr, 40, 24, 3, 32, None], dict(packed=packed)


case("conv2d_3x3_128_128", exact=False)(_conv(2, 16, 32, 128, 128))
case("conv2d_3x3_residual_ragged_tile", exact=False, tol=2.0 ** -5)(_conv(2, 10, 40, 64, 128, resid=True))       # tol: a 1-ulp flip of the convolution before x + h is absolute, not relative
case("conv2d_3x3_conv_in_3_channels", exact=False)(_conv(1, 16, 32, 3, 128, cin_store=8))
case("conv2d_3x3_latent_16_to_512", exact=False)(_conv(1, 8, 32, 16, 512))
case("conv2d_1x1_shortcut", exact=False)(_conv(2, 12, 32, 128, 256, ks=1))
case("conv2d_3x3_stride2_downsample", exact=False)(_conv(2, 16, 64, 128, 128, stride=2))
case("conv2d_3x3_upsample", exact=False)(_conv(1, 8, 16, 256, 256, up=1))
case("conv2d_3x3_narrow_out_32", exact=False)(_conv(2, 8, 32, 512, 32, bn=32))
case("conv2d_3x3_conv_out_3", exact=False)(_conv(1, 16, 32, 128, 3, bn=32))


def _gn_nhwc(C, silu, HW=24 * 24, B=2):
    def fn(alloc):
        r = rng(62 + C)
        x = to_bf16(r.standard_normal((B, HW, C)) * 2 + 0.3)
        w, b = to_bf16(1 + 0.2 * r.standard_normal(C)), to_bf16(0.2 * r.standard_normal(C))
        out = alloc(np.zeros_like(x))
        ws = alloc(np.zeros(B * 32 * (C // 4) * 2 * 8 + B * 64 * 8 + 256, np.uint8))          # selftok_groupnorm_nhwc_workspace_bytes
        return "selftok_groupnorm_silu_nhwc_bf16", [alloc(x).ptr, alloc(w).ptr, alloc(b).ptr, out.ptr, ws.ptr, B, HW, C, 32, 1e-6, silu, None], dict(out_bf16=out)
    return fn


for _C in (128, 256, 512):
    case(f"groupnorm_silu_nhwc_C{_C}", exact=False)(_gn_nhwc(_C, 1))
case("groupnorm_nhwc_C512_many_pixels", exact=False)(_gn_nhwc(512, 0, HW=70 * 70))


# ---- runner -----------------------------------------------------------------------------------------------------------------------
def split_planes(blk, rows, K):
    """split-activation buffer (uint16 halfs) -> (hi, lo) fp32 arrays [rows, K] of the live rows"""
    h = blk.view(np.float16).reshape(-1, K // 32, 2, 16, 32)            # [chunk, kt, plane, row%16, k%32]
    h = h.transpose(2, 0, 3, 1, 4).reshape(2, -1, K)[:, :rows]
    return h[0].astype(np.float32), h[1].astype(np.float32)


# ---- the exact-order Q-Former encoder entries (round 5): bit-exact on both sides ----------------------------------------------------------
@case("ex_linear_gelu_res_gate")
def _(alloc):
    r = rng(71)
    M, K, N, T = 192, 512, 96, 64
    x3 = f32(r.standard_normal((M, 2 * K)))                      # the input is a column slice of a wider tensor
    w = f32(r.standard_normal((N, K)) / np.sqrt(K)); b = f32(r.standard_normal(N) * 0.2)
    res = f32(r.standard_normal((M, N))); gate = f32(r.standard_normal((T, 3 * N)))
    out = alloc(np.zeros((M, N), np.float32))
    xd = alloc(x3)
    return "selftok_ex_linear_f32", [xd.ptr + 4 * K, 2 * K, alloc(w).ptr, alloc(b).ptr, alloc(res).ptr, N, 0, alloc(gate).ptr + 4 * N, 3 * N, T, out.ptr, N,
                                     M, N, K, 1, None], dict(out=out)


@case("ex_linear_k2048_n16")
def _(alloc):
    r = rng(72)
    M, K, N = 128, 2048, 16
    x = f32(r.standard_normal((M, K))); w = f32(r.standard_normal((N, K)) / np.sqrt(K)); b = f32(r.standard_normal(N))
    out = alloc(np.zeros((M, N), np.float32))
    return "selftok_ex_linear_f32", [alloc(x).ptr, K, alloc(w).ptr, alloc(b).ptr, None, 0, 0, None, 0, 0, out.ptr, N, M, N, K, 0, None], dict(out=out)


@case("ex_linear_bias_last_per_sample_gate")
def _(alloc):
    """the MMDiT's attention projection: bias added after the K-blocks (at::linear on a non-contiguous input), gate table indexed by sample"""
    r = rng(76)
    B, T, K, N = 3, 64, 1536, 64
    x = f32(r.standard_normal((B * T, K))); w = f32(r.standard_normal((N, K)) / np.sqrt(K)); b = f32(r.standard_normal(N))
    res = f32(r.standard_normal((B * T, N))); gate = f32(r.standard_normal((B, 6 * N)))
    out = alloc(np.zeros((B * T, N), np.float32))
    return "selftok_ex_linear_f32", [alloc(x).ptr, K, alloc(w).ptr, alloc(b).ptr, alloc(res).ptr, N, 0, alloc(gate).ptr + 4 * 2 * N, 6 * N, -T, out.ptr, N,
                                     B * T, N, K, 2, None], dict(out=out)


def _linear_f32(flags, exact_note):
    def fn(alloc):
        """csrc/gemm_fp32.hip through the C ABI: 300 rows (ragged tile), proj-like shape with the `x + gate * y` epilogue, a workspace for the tail split"""
        r = rng(78)
        M, K, N, T = 300, 1536, 256, 100
        x = f32(r.standard_normal((M, K))); w = f32(r.standard_normal((N, K)) / np.sqrt(K)); b = f32(r.standard_normal(N))
        res = f32(r.standard_normal((M, N))); gate = f32(r.standard_normal((T, 2 * N)))
        out = alloc(np.zeros((M, N), np.float32))
        ws = alloc(np.zeros(8 * 1 * 8 * 128 * 128 * 4, np.uint8))            # >= selftok_linear_f32_workspace_bytes(300, 256, 1536, .): 8 XCDs x 1 tail tile x (4 K-blocks | 8 units) planes
        return "selftok_linear_f32", [alloc(x).ptr, K, alloc(w).ptr, alloc(b).ptr, alloc(res).ptr, N, 0, alloc(gate).ptr + 4 * N, 2 * N, T, out.ptr, N,
                                      M, N, K, flags, ws.ptr, 8 * 8 * 128 * 128 * 4, None], dict(out=out)
    return fn


case("linear_f32_mkl_order_bias_last_gate")(_linear_f32(4 | 2, "MKL order"))                       # SELFTOK_LINEAR_MKL_ORDER | SELFTOK_LINEAR_BIAS_LAST: bit-exact on both builds
case("linear_f32_free_order_gate", exact=False, tol=4e-6)(_linear_f32(0, "free order"))            # free order: the tail tiles' chains are split on the GPU


@case("ex_layernorm_mod_per_sample")
def _(alloc):
    r = rng(77)
    B, T, N = 3, 32, 1536
    x = f32(r.standard_normal((B * T, N)) * 2 + 0.3); table = f32(r.standard_normal((B, 6 * N)) * 0.5)
    out = alloc(np.zeros((B * T, N), np.float32)); td = alloc(table)
    return "selftok_ex_layernorm_mod_f32", [alloc(x).ptr, N, out.ptr, N, td.ptr, td.ptr + 4 * N, 6 * N, -T, None, None, None, B * T, N, 1e-6, None], dict(out=out)


@case("ex_layernorm_mod")
def _(alloc):
    r = rng(73)
    rows, N, T = 80, 512, 16
    x = f32(r.standard_normal((rows, N)) * 2 + 0.3); table = f32(r.standard_normal((T, 6 * N)) * 0.5)
    out = alloc(np.zeros((rows, N), np.float32)); st = alloc(np.zeros((rows, 2), np.float32)); td = alloc(table)
    return "selftok_ex_layernorm_mod_f32", [alloc(x).ptr, N, out.ptr, N, td.ptr + 4 * 3 * N, td.ptr + 4 * 4 * N, 6 * N, T, None, None, st.ptr, rows, N, 1e-6, None], dict(out=out, stats=st)


def _ex_unary(mode):
    def fn(alloc):
        bits = np.concatenate([np.arange(0, 2 ** 32, 2 ** 14, dtype=np.uint64).astype(np.uint32), np.linspace(-12, 12, 1 << 14, dtype=np.float32).view(np.uint32)])
        x = np.ascontiguousarray(bits.view(np.float32))
        x = x[np.abs(x) < 80.0]                                    # finite results in every mode (NaN / inf: tests/test_encoder_exact_gpu.py)
        y = alloc(np.zeros_like(x))
        return "selftok_ex_unary_f32", [alloc(x).ptr, y.ptr, x.size, mode, None], dict(y=y)
    return fn


for _m in range(5):
    case(f"ex_unary_mode{_m}")(_ex_unary(_m))


@case("ex_attention_two_segments")
def _(alloc):
    r = rng(74)
    B, H, Tq, Tk1, Tk2, D = 1, 2, 64, 64, 512, 64
    HD = H * D
    qq = f32(r.standard_normal((B, Tk2, 3 * HD)) * 1.4); kvx = f32(r.standard_normal((B, Tk1, 2 * HD)) * 1.4)
    lib_ws = 4 * (B * H * Tq * (Tk1 + Tk2) + B * H * D * (Tk1 + Tk2) + 2 * B * H * Tq)
    out = alloc(np.zeros((B, Tq, HD), np.float32)); ws = alloc(np.zeros(lib_ws, np.uint8))
    qd, kd = alloc(qq), alloc(kvx)
    return "selftok_ex_attention_f32", [qd.ptr, 3 * HD, kd.ptr, kd.ptr + 4 * HD, 2 * HD, Tk1, Tk1, Tk1, qd.ptr + 4 * HD, qd.ptr + 8 * HD, 3 * HD, Tk2, out.ptr, ws.ptr,
                                        B, H, Tq, D, None], dict(out=out)


@case("ex_attention_prefix_mask")
def _(alloc):
    """the MMDiT's joint attention: 512 context key slots of which 300 are visible (held as 300 rows), then 64 image keys"""
    r = rng(75)
    B, H, Tq, slots, valid, Tk2, D = 1, 2, 64, 512, 300, 64, 64
    HD = H * D
    cq = f32(r.standard_normal((B, valid, 3 * HD)) * 1.4); xq = f32(r.standard_normal((B, Tk2, 3 * HD)) * 1.4)
    lib_ws = 4 * (B * H * Tq * (slots + Tk2) + B * H * D * (slots + Tk2) + 2 * B * H * Tq)
    out = alloc(np.zeros((B, Tq, HD), np.float32)); ws = alloc(np.zeros(lib_ws, np.uint8))
    cd, xd = alloc(cq), alloc(xq)
    return "selftok_ex_attention_f32", [xd.ptr, 3 * HD, cd.ptr + 4 * HD, cd.ptr + 8 * HD, 3 * HD, slots, valid, valid, xd.ptr + 4 * HD, xd.ptr + 8 * HD, 3 * HD, Tk2,
                                        out.ptr, ws.ptr, B, H, Tq, D, None], dict(out=out)


def run(lib, name, side):
    """issue case `name` on `lib` with buffers of class `side` (Host | Dev); returns {output name: ndarray} (names starting with
    '_' are metadata passed through)"""
    keep = []

    def alloc(a):
        keep.append(side(a))
        return keep[-1]
    built = CASES[name]["fn"](alloc)
    sym, args, outs = built[0], built[1], built[2]
    for psym, pargs in (built[3] if len(built) > 3 else []):
        rc = getattr(lib, psym)(*pargs)
        assert rc == 0, (psym, rc, lib.selftok_last_error())
    rc = getattr(lib, sym)(*args)
    assert rc == 0, (sym, rc, lib.selftok_last_error())
    res = {}
    for k, v in outs.items():
        res[k] = v if k.startswith("_") else v.numpy()
    return res
