"""not gpu: oracle/libselftok_cpu.so -- the C ABI of include/selftok_hip.h compiled for the CPU -- pinned to
references that do not share its code: the reference's golden vectors (tests/golden/), the scalar oracle (oracle/clib.py), torch-CPU
(the reference's own arithmetic for LayerNorm / SDPA / bf16 element-wise ops) and fp64 numpy.  tests/test_cpu_twin_gpu.py then holds the
gfx950 library to the twin, call by call."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import abi_cases as A
from oracle import clib
from selftoktokenizer_amd import _lib, synth, weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
EINVAL = -1


@pytest.fixture(scope="module")
def twin():
    path = os.path.join(ROOT, "oracle", "libselftok_cpu.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return A.bind(path)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def ptr(a):
    return a.ctypes.data


def test_exports_every_symbol_of_the_header(twin):
    import re
    hdr = open(os.path.join(ROOT, "include", "selftok_hip.h")).read()
    declared = set(re.findall(r"\b(selftok_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(twin, name), name
    assert twin.selftok_version() == 100
    assert twin.selftok_vq_workspace_bytes(1000, 32768) == 128 * 1000 * 8
    assert twin.selftok_vq_packed_bytes(32768, 16) == (32768 * 16 + 64) * 4 + 32768 * 16 * 2 * 2
    assert twin.selftok_linear_f16x2_packed_bytes(1536, 1536) == 4 * 1536 * 1536 and twin.selftok_linear_f16x2_packed_bytes(100, 32) == 0
    assert twin.selftok_split_f16x2_bytes(17, 64) == 32 * 64 * 4 and twin.selftok_split_f16x2_bytes(16, 48) == 0


def test_error_behaviour(twin):
    z = np.zeros((4, 8), np.float32)
    assert twin.selftok_vq_encode_f32(ptr(z), ptr(z), ptr(z), None, None, 4, 4, 8, 0, None) == EINVAL          # D != 16
    assert b"vq_encode" in twin.selftok_last_error()
    assert twin.selftok_vq_pack_codebook(ptr(z), ptr(z), 33, 16, None) == EINVAL                                  # C % 32
    assert twin.selftok_residual_ln_mod_f32(ptr(z), None, None, None, None, None, ptr(z), 1, 4, 8, 0, 0, 0, 0, 1e-6, None) == EINVAL   # H = 8
    assert b"hidden size" in twin.selftok_last_error()
    assert twin.selftok_bias_gelu_f32(ptr(z), None, 4, 6, None) == EINVAL
    d = _lib.AttnDesc()
    d.B, d.H, d.head_dim = 1, 1, 32
    assert twin.selftok_attn_f32(C.addressof(d), None) == EINVAL and b"head_dim" in twin.selftok_last_error()
    assert twin.selftok_groupnorm_silu_bf16(ptr(z), ptr(z), ptr(z), ptr(z), 1, 8, 4, 2, 1e-6, 1, None) == EINVAL   # HW % 8
    # empty inputs are not errors
    assert twin.selftok_vq_encode_f32(None, ptr(z), None, None, None, 0, 32, 16, 0, None) == 0
    assert twin.selftok_code_gather_ln_f32(None, None, None, None, None, 0, 32, 16, 1e-5, 0, None) == 0


@pytest.mark.parametrize("name", sorted(A.CASES))
def test_every_case_runs(twin, name):
    out = A.run(twin, name, A.Host)
    for k, v in out.items():
        if not k.startswith("_") and v.dtype == np.float32 and "nan" not in name:
            assert np.isfinite(v).all(), k


# ---- VQ: golden vectors of the reference + the scalar oracle --------------------------------------------------------------------------
@pytest.fixture(scope="module")
def codebook():
    return W._synth_tensor("encoder.quantizer._codebook.embed", (1, 32768, 16), "cpu")[0].contiguous().numpy()


def twin_vq(twin, z, cb, how, flags=0):
    n, c = z.shape[0], cb.shape[0]
    z, cb = np.ascontiguousarray(z, np.float32), np.ascontiguousarray(cb, np.float32)
    ids = np.zeros(n, np.int32 if flags & A.I32 else np.int64)
    best = np.zeros(n, np.float32)
    ws = np.zeros(128 * max(n, 1), np.uint64)
    if how == "raw":
        assert twin.selftok_vq_encode_f32(ptr(z), ptr(cb), ptr(ids), ptr(best), ptr(ws), n, c, 16, flags, None) == 0
        return ids, best
    packed = np.zeros(twin.selftok_vq_packed_bytes(c, 16) // 4, np.float32)
    assert twin.selftok_vq_pack_codebook(ptr(cb), ptr(packed), c, 16, None) == 0
    if how == "packed":
        assert twin.selftok_vq_encode_packed_f32(ptr(z), ptr(packed), ptr(ids), ptr(best), ptr(ws), n, c, 16, flags, None) == 0
    else:
        ns = C.c_int(0)
        assert twin.selftok_vq_argmax_partial_packed_f32(ptr(z), ptr(packed), ptr(ws), C.addressof(ns), n, c, 16, flags, None) == 0
        assert ns.value >= 1
        assert twin.selftok_vq_finalize_packed(ptr(ws), ptr(z), ptr(packed), ptr(ids), ptr(best), n, c, 16, ns.value, flags, None) == 0
    return ids, best


@pytest.mark.parametrize("how", ["raw", "packed", "two_call"])
def test_vq_matches_reference_golden(twin, codebook, how):
    """the reference's CosineSimCodebook.forward (eval) on 1024 rows incl. tie / zero / NaN / inf rows (vq_small.npz) and the ids of
    the reference encoder (encoder_b2.npz)"""
    g = np.load(os.path.join(GOLD, "vq_small.npz"))
    ids, best = twin_vq(twin, g["z"].reshape(-1, 16), codebook, how)
    np.testing.assert_array_equal(ids.reshape(2, 512), g["ids"])
    ref = g["best_bits"].reshape(-1).view(np.float32)
    nan = np.isnan(ref)
    np.testing.assert_array_equal(np.isnan(best), nan)
    np.testing.assert_array_equal(bits(best)[~nan], bits(ref)[~nan])
    ids2, _ = twin_vq(twin, g["z_proj"].reshape(-1, 16), codebook, how, flags=A.I32)
    assert ids2.dtype == np.int32
    np.testing.assert_array_equal(ids2.reshape(2, 512), g["ids_proj"])
    e = np.load(os.path.join(GOLD, "encoder_b2.npz"))
    ids3, _ = twin_vq(twin, e["z"].reshape(-1, 16), codebook, how)
    np.testing.assert_array_equal(ids3.reshape(2, 512), e["ids"])


def test_vq_matches_scalar_oracle(twin):
    z, cb = A._vq_inputs(77, n=500, c=2048)
    for flags, norm in ((0, True), (A.PRENORMED, False)):
        zz = A.unit_rows(np.where(np.abs(z).sum(-1, keepdims=True) == 0, 1.0, z)) if not norm else z
        ids_o, best_o = clib.vq_encode(zz, cb, normalize=norm)
        for how in ("raw", "packed", "two_call"):
            ids, best = twin_vq(twin, zz, cb, how, flags)
            np.testing.assert_array_equal(ids, ids_o)
            np.testing.assert_array_equal(bits(best), bits(best_o))


def test_packed_image_is_the_documented_fragment_order(twin):
    """include/selftok_hip.h: tile t = codes 32t .. 32t+31; element (i, k) at (k/8) 256 + ((k%2) 32 + i) 4 + (k/2)%4; second image = fp16
    hi / lo of 128 e in the A-operand order of v_mfma_f32_32x32x16_f16; metadata word: bit 0 non-finite, bit 1 fp16 range, bit 2 norm"""
    cb = A._vq_inputs(3, c=64)[1]
    packed = np.zeros(twin.selftok_vq_packed_bytes(64, 16) // 4, np.float32)
    assert twin.selftok_vq_pack_codebook(ptr(cb), ptr(packed), 64, 16, None) == 0
    for c, k in ((0, 0), (5, 3), (37, 15), (63, 8)):
        t, i = c >> 5, c & 31
        m, lane = k >> 1, (k & 1) * 32 + i
        assert packed[t * 512 + (m >> 2) * 256 + lane * 4 + (m & 3)] == cb[c, k]
        h = packed[64 * 16 + 64:].view(np.float16)[t * 1024:(t + 1) * 1024]
        hi, lo = h[((k >> 3) * 32 + i) * 8 + (k & 7)], h[512 + ((k >> 3) * 32 + i) * 8 + (k & 7)]
        assert hi == np.float16(np.float32(cb[c, k] * 128)) and abs(np.float32(hi) + np.float32(lo) - cb[c, k] * 128) < 128 * 2.0 ** -21
    assert packed[64 * 16:64 * 16 + 64].view(np.uint32)[0] == 0
    cb2 = cb.copy(); cb2[9] *= 2
    assert twin.selftok_vq_pack_codebook(ptr(cb2), ptr(packed), 64, 16, None) == 0
    assert packed[64 * 16:64 * 16 + 64].view(np.uint32)[0] == 4


def test_code_gather_ln_matches_torch(twin):
    r = A.rng(6)
    cb = A.unit_rows(r.standard_normal((512, 16)))
    ids = r.integers(0, 512, 200).astype(np.int64)
    w, b = A.f32(1 + 0.1 * r.standard_normal(16)), A.f32(0.1 * r.standard_normal(16))
    out = np.zeros((200, 16), np.float32)
    assert twin.selftok_code_gather_ln_f32(ptr(ids), ptr(cb), ptr(w), ptr(b), ptr(out), 200, 512, 16, 1e-5, 0, None) == 0
    ref = F.layer_norm(torch.from_numpy(cb[ids]), (16,), torch.from_numpy(w), torch.from_numpy(b), 1e-5).numpy()
    np.testing.assert_allclose(out, ref, rtol=2e-6, atol=2e-6)
    e = np.load(os.path.join(GOLD, "encoder_b2.npz"))
    book = W._synth_tensor("encoder.quantizer._codebook.embed", (1, 32768, 16), "cpu")[0].contiguous().numpy()
    q = np.zeros((1024, 16), np.float32)
    idr = np.ascontiguousarray(e["ids"].reshape(-1))
    assert twin.selftok_code_gather_ln_f32(ptr(idr), ptr(book), None, None, ptr(q), 1024, 32768, 16, 1e-5, 0, None) == 0
    np.testing.assert_array_equal(q, book[idr])


def test_vq_training_statistics_match_the_one_hot_form(twin):
    """vector_quantize_pytorch.py:568-611: bins = one_hot.sum(0), embed_sum = one_hot^T l2norm(z); tpc lerp (SelftokPipeline's token-per-slot EMA)"""
    r = A.rng(8)
    z = A.f32(r.standard_normal((400, 16)))
    ids = r.integers(0, 64, 400).astype(np.int64)
    ids[3] = -1                                                   # masked-out row
    bins, esum = np.zeros(64, np.float32), np.zeros((64, 16), np.float32)
    assert twin.selftok_vq_ema_accumulate_f32(ptr(z), ptr(ids), ptr(bins), ptr(esum), 400, 64, 16, 0, None) == 0
    ok = ids >= 0
    oh = F.one_hot(torch.from_numpy(ids[ok]), 64).float()
    np.testing.assert_array_equal(bins, oh.sum(0).numpy())
    np.testing.assert_allclose(esum, (oh.T @ F.normalize(torch.from_numpy(z[ok]), dim=-1)).numpy(), rtol=1e-5, atol=1e-5)
    for w in (0.25, 0.75):
        tpc0 = A.f32(r.random((8, 64)))
        tid = r.integers(0, 64, (6, 8)).astype(np.int64)
        tpc = tpc0.copy()
        assert twin.selftok_vq_tpc_update_f32(ptr(tpc), ptr(tid), 6, 8, 64, w, 0, None) == 0
        ref = torch.lerp(torch.from_numpy(tpc0), F.one_hot(torch.from_numpy(tid), 64).float().mean(0), w).numpy()
        np.testing.assert_allclose(tpc, ref, rtol=1e-6, atol=1e-7)


def test_entropy_regularisers_match_reference_golden(twin):
    """vq_entropy.npz = the reference's calc_entropy / calc_ema_entropy / get_group_perplexity on its own CosineSimCodebook's distances and
    torch autograd through them (tools/oracle/gen_golden.py vq_entropy).  Here: the two reductions from the C ABI, the small [K, C]
    epilogue in torch (autograd gives g = dF/d colmean), the backward from the C ABI."""
    g = np.load(os.path.join(GOLD, "vq_entropy.npz"))
    C, K, B = 2048, 128, 6
    embed0 = np.ascontiguousarray(F.normalize(synth.hash_normalish(int(g["embed0_seed"]), (C, 16)), dim=-1).numpy())
    ws = np.zeros(twin.selftok_vq_softmax_workspace_bytes(B * K) // 4, np.float32)
    for case in (0, 1):
        dw, ratio, r0, r1, w = g[f"args_{case}"]
        z = np.ascontiguousarray(synth.hash_normalish(int(g[f"seed_{case}"]), (B, K, 16)).numpy())
        rows, cm = np.zeros((B * K, 2), np.float32), np.zeros((K, C), np.float32)
        assert twin.selftok_vq_softmax_stats_f32(ptr(z), ptr(embed0), ptr(rows), ptr(cm), ptr(ws), B, K, C, 16, 10.0, 0, None) == 0
        np.testing.assert_allclose(rows[:, 1].mean(), g[f"entropy_to_min_{case}"], rtol=2e-6)
        apk = torch.from_numpy(cm).requires_grad_(True)
        ap = apk.mean(0)
        e_max = -(ap * ap.log()).sum()
        np.testing.assert_allclose(e_max.item(), g[f"entropy_to_max_{case}"], rtol=2e-6)
        tpc = torch.from_numpy(g[f"tpc_{case}"])
        ema = tpc * ratio + apk * (1 - ratio)
        c_ent = (-(ema * ema.log()).sum(-1)).mean()
        grp = torch.stack([t.mean(0) for t in ema.tensor_split(64, dim=0)])
        g_ent = (-(grp * grp.log()).sum(-1)).mean()
        np.testing.assert_allclose(c_ent.item(), g[f"codebook_entropy_{case}"], rtol=2e-6)
        np.testing.assert_allclose(g_ent.item(), g[f"group_entropy_{case}"], rtol=2e-6)
        loss = -dw * w * 0.5 * (c_ent + g_ent)
        np.testing.assert_allclose(loss.item(), g[f"diversity_loss_{case}"], rtol=3e-6)
        for F_, key in ((loss, f"grad_z_{case}"), (-dw * e_max, f"grad_z_entropy_to_max_{case}")):
            (gc,) = torch.autograd.grad(F_, apk, retain_graph=True)
            gc = np.ascontiguousarray(gc.float().numpy())
            gz = np.zeros((B, K, 16), np.float32)
            assert twin.selftok_vq_softmax_backward_f32(ptr(z), ptr(embed0), ptr(rows), ptr(gc), ptr(gz), ptr(ws), B, K, C, 16, 10.0, 0, None) == 0
            ref = g[key]
            assert np.abs(gz - ref).max() <= 2e-5 * np.abs(ref).max(), (key, np.abs(gz - ref).max(), np.abs(ref).max())


# ---- fused residual / LayerNorm / modulate against torch-CPU --------------------------------------------------------------------------
@pytest.mark.parametrize("H", [64, 256, 512, 1024, 1536])
@pytest.mark.parametrize("per_token", [False, True])
def test_residual_ln_mod_matches_torch(twin, H, per_token):
    B, T = 3, 10
    r = A.rng(H)
    x, y = A.f32(r.standard_normal((B, T, H))), A.f32(r.standard_normal((B, T, H)))
    tab = A.f32(0.3 * r.standard_normal(((T if per_token else B), 3 * H)))
    msb, mst = (0, 3 * H) if per_token else (3 * H, 0)
    xo, n = np.zeros_like(x), np.zeros_like(x)
    rc = twin.selftok_residual_ln_mod_f32(ptr(x), ptr(y), ptr(tab) + 8 * H, ptr(tab), ptr(tab) + 4 * H, ptr(xo), ptr(n), B, T, H, msb, mst, msb, mst, 1e-6, None)
    assert rc == 0, twin.selftok_last_error()
    t = torch.from_numpy(tab)
    sh, sc, g = (t[:, i * H:(i + 1) * H] for i in range(3))
    bc = (lambda a: a[None]) if per_token else (lambda a: a[:, None])
    xr = torch.from_numpy(x) + bc(g) * torch.from_numpy(y)
    np.testing.assert_array_equal(xo, xr.numpy())                                  # separate multiply and add: bit-exact
    ref = F.layer_norm(xr, (H,), None, None, 1e-6) * (1 + bc(sc)) + bc(sh)
    np.testing.assert_allclose(n, ref.numpy(), rtol=1e-5, atol=2e-6)
    # the split form stores the same values as fp16 pairs
    nb = np.zeros(((B * T + 15) // 16) * 16 * H * 2, np.uint16)
    ov = np.zeros(1, np.int32)
    if H % 32 == 0:
        rc = twin.selftok_residual_ln_mod_split(ptr(x), ptr(y), ptr(tab) + 8 * H, ptr(tab), ptr(tab) + 4 * H, ptr(xo), ptr(nb), ptr(ov), B, T, H, msb, mst, msb, mst, 1e-6, None)
        assert rc == 0 and ov[0] == 0
        hi, lo = A.split_planes(nb, B * T, H)
        np.testing.assert_array_equal(hi, n.reshape(-1, H).astype(np.float16).astype(np.float32))
        np.testing.assert_allclose(hi + lo / 2048, n.reshape(-1, H), rtol=2.0 ** -21, atol=2.0 ** -35)


# ---- f16x2 Linear: fp32-equivalent on the CPU too -----------------------------------------------------------------------------------------
def test_linear_f16x2_is_fp32_equivalent(twin):
    M, N, K = 48, 256, 1536
    r = A.rng(5)
    a, w, bias = A.f32(r.standard_normal((M, K))), A.f32(r.standard_normal((N, K)) / np.sqrt(K)), A.f32(r.standard_normal(N))
    packed = np.zeros(N * K * 2, np.uint16)
    ov = np.zeros(1, np.int32)
    assert twin.selftok_linear_f16x2_pack_weight(ptr(w), ptr(packed), N, K, ptr(ov), None) == 0
    out = np.zeros((M, N), np.float32)
    assert twin.selftok_linear_f16x2_f32(ptr(a), K, ptr(packed), ptr(bias), ptr(out), N, M, N, K, 0, ptr(ov), None) == 0 and ov[0] == 0
    exact = a.astype(np.float64) @ w.astype(np.float64).T + bias
    lib32 = (torch.from_numpy(a) @ torch.from_numpy(w).T + torch.from_numpy(bias)).numpy()       # the reference's arithmetic: MKL fp32
    e_split, e_lib = np.abs(out - exact).max(), np.abs(lib32 - exact).max()
    print(f"max abs error vs fp64: f16x2 split {e_split:.2e}, torch fp32 {e_lib:.2e}")
    assert e_split < 2e-6 and e_split < 4 * e_lib + 1e-7
    # split-activation entry points: bit-identical to the fp32-activation one; GELU and residual epilogues
    ablk = np.zeros(((M + 15) // 16) * 16 * K * 2, np.uint16)
    assert twin.selftok_split_f16x2_f32(ptr(a), K, ptr(ablk), M, K, ptr(ov), None) == 0
    out2 = np.zeros_like(out)
    assert twin.selftok_linear_f16x2_split(ptr(ablk), ptr(packed), ptr(bias), ptr(out2), None, N, M, N, K, 0, ptr(ov), None) == 0
    np.testing.assert_array_equal(out2, out)
    assert twin.selftok_linear_f16x2_split(ptr(ablk), ptr(packed), ptr(bias), ptr(out2), None, N, M, N, K, A.GELU, ptr(ov), None) == 0
    np.testing.assert_allclose(out2, F.gelu(torch.from_numpy(out), approximate="tanh").numpy(), rtol=2e-6, atol=2e-7)
    resid, gate = A.f32(r.standard_normal((M, N))), A.f32(r.standard_normal((2, N)))
    assert twin.selftok_linear_f16x2_split_residual(ptr(ablk), ptr(packed), ptr(bias), ptr(resid), N, ptr(gate), N, 0, 24, ptr(out2), N, M, N, K, ptr(ov), None) == 0
    np.testing.assert_array_equal(out2, resid + np.repeat(gate, 24, 0) * out)
    a[1, 2] = 7e4
    assert twin.selftok_linear_f16x2_f32(ptr(a), K, ptr(packed), ptr(bias), ptr(out), N, M, N, K, 0, ptr(ov), None) == 0 and ov[0] & 1


# ---- attention against SDPA with the reference's materialised mask ------------------------------------------------------------------------
@pytest.mark.parametrize("mode", [0, A.ATTN_F16X2])
@pytest.mark.parametrize("see", [1, 0])
def test_attention_matches_masked_sdpa(twin, mode, see):
    from oracle import model as OM
    B, H, n0, n1 = 2, 2, 40, 24
    kvis = [39, 7]
    out = A.CASES  # noqa: F841  (cases table is exercised by test_every_case_runs; here the inputs are rebuilt to own the reference)
    r = A.rng(40)
    W_ = H * 64
    ctx, img = A.f32(r.standard_normal((B, n0, 3 * W_))), A.f32(r.standard_normal((B, n1, 3 * W_)))
    o0, o1 = np.zeros((B, n0, W_), np.float32), np.zeros((B, n1, W_), np.float32)
    kv = np.asarray(kvis, np.int32)
    d = _lib.AttnDesc()
    for s, qkv, o, n in ((d.seg[0], ctx, o0, n0), (d.seg[1], img, o1, n1)):
        s.q, s.k, s.v, s.o, s.len = ptr(qkv), ptr(qkv) + 4 * W_, ptr(qkv) + 8 * W_, ptr(o), n
        s.q_rs = s.k_rs = s.v_rs = 3 * W_
        s.q_bs = s.k_bs = s.v_bs = n * 3 * W_
        s.o_rs, s.o_bs = W_, n * W_
    d.B, d.H, d.head_dim, d.kvis, d.seg0_sees_seg1, d.scale, d.mode = B, H, 64, ptr(kv), see, 0.125, mode
    assert twin.selftok_attn_f32(C.addressof(d), None) == 0
    qkv = torch.cat([torch.from_numpy(ctx), torch.from_numpy(img)], 1)
    q, k, v = qkv.reshape(B, n0 + n1, 3, H, 64).permute(2, 0, 3, 1, 4)
    mask = torch.arange(n0)[None] <= torch.tensor(kvis)[:, None]
    ref = F.scaled_dot_product_attention(q, k, v, attn_mask=OM.joint_mask(mask, n1, bool(see))).transpose(1, 2).reshape(B, n0 + n1, W_).numpy()
    np.testing.assert_allclose(o1, ref[:, n0:], rtol=1e-5, atol=2e-6)
    for b in range(B):                       # live context rows; dead rows (beyond kvis) are not written
        np.testing.assert_allclose(o0[b, :kvis[b] + 1], ref[b, :kvis[b] + 1], rtol=1e-5, atol=2e-6)
        assert not o0[b, kvis[b] + 1:].any()


# ---- small kernels against the reference's golden vectors / torch-CPU -------------------------------------------------------------------
def test_rmsnorm_rotary_match_reference_golden(twin):
    g = np.load(os.path.join(GOLD, "rmsnorm_rotary.npz"))
    U = lambda seed, shape, lo, hi: np.ascontiguousarray(synth.hash_uniform(seed, shape, lo, hi).numpy())
    x, w = U(14, (7, 24, 64), -2, 2), U(15, (64,), 0.9, 1.1)
    out = np.zeros_like(x)
    assert twin.selftok_rmsnorm_f32(ptr(x), ptr(w), ptr(out), 7 * 24, 64, 1e-6, None) == 0
    np.testing.assert_allclose(out, g["rms_affine"], rtol=1e-5, atol=1e-6)
    assert twin.selftok_rmsnorm_f32(ptr(x), None, ptr(out), 7 * 24, 64, 1e-6, None) == 0
    np.testing.assert_allclose(out, g["rms_plain"], rtol=1e-5, atol=1e-6)
    t, f = U(16, (2, 3, 10, 32), -2, 2), U(17, (10, 32), -3, 3)
    o = np.zeros_like(t)
    assert twin.selftok_rotary_f32(ptr(t), ptr(f), ptr(o), 60, 10, 32, 1.0, None) == 0
    np.testing.assert_allclose(o, g["rot_full"], rtol=1e-5, atol=1e-5)
    assert twin.selftok_rotary_f32(ptr(t), ptr(f), ptr(o), 60, 10, 32, 0.5, None) == 0
    np.testing.assert_allclose(o, g["rot_scaled"], rtol=1e-5, atol=1e-5)


def test_elementwise_match_torch(twin):
    r = A.rng(20)
    h, b = A.f32(2 * r.standard_normal((37, 64))), A.f32(r.standard_normal(64))
    ref = F.gelu(torch.from_numpy(h) + torch.from_numpy(b), approximate="tanh").numpy()
    assert twin.selftok_bias_gelu_f32(ptr(h), ptr(b), 37, 64, None) == 0
    np.testing.assert_allclose(h, ref, rtol=2e-6, atol=5e-7)              # 1 + tanh cancels in the negative tail
    x = A.f32(3 * r.standard_normal(1000)); o = np.zeros_like(x)
    assert twin.selftok_silu_f32(ptr(x), ptr(o), 1000, None) == 0
    np.testing.assert_allclose(o, F.silu(torch.from_numpy(x)).numpy(), rtol=2e-6, atol=1e-7)
    img = A.f32(r.standard_normal((2, 16, 8, 12)))
    p = np.zeros((2, 24, 64), np.float32)
    assert twin.selftok_patchify_f32(ptr(img), ptr(p), 2, 16, 8, 12, None) == 0
    ref = F.unfold(torch.from_numpy(img), 2, stride=2).transpose(1, 2).numpy()          # [B, L, C*4], feature = c*4 + p*2 + q
    np.testing.assert_array_equal(p, ref)
    yc, yu = A.f32(r.standard_normal((2, 24, 64))), A.f32(r.standard_normal((2, 24, 64)))
    xo, vo = np.zeros_like(img), np.zeros_like(img)
    assert twin.selftok_unpatchify_cfg_euler_f32(ptr(yc), ptr(yu), ptr(img), ptr(xo), ptr(vo), 2, 16, 4, 6, 0.02, 3.5, None) == 0
    un = lambda y: torch.einsum("nhwpqc->nchpwq", torch.from_numpy(y).reshape(2, 4, 6, 2, 2, 16)).reshape(2, 16, 8, 12)     # sd3/mmdit.py:898-916
    v = un(yu) + 3.5 * (un(yc) - un(yu))
    np.testing.assert_array_equal(vo, v.numpy())
    np.testing.assert_array_equal(xo, (torch.from_numpy(img) - np.float32(0.02) * v).numpy())


def test_bf16_epilogues_match_torch_cpu(twin):
    r = A.rng(50)
    x = torch.from_numpy(A.f32(r.standard_normal((2, 64, 8, 8)) * 2 + 0.3)).bfloat16()
    w, b = torch.from_numpy(A.f32(1 + 0.2 * r.standard_normal(64))).bfloat16(), torch.from_numpy(A.f32(0.2 * r.standard_normal(64))).bfloat16()
    as_u16 = lambda t: np.ascontiguousarray(t.view(torch.int16).numpy().view(np.uint16))
    xu, wu, bu = as_u16(x), as_u16(w), as_u16(b)
    out = np.zeros_like(xu)
    for silu in (0, 1):
        assert twin.selftok_groupnorm_silu_bf16(ptr(xu), ptr(wu), ptr(bu), ptr(out), 2, 64, 64, 32, 1e-6, silu, None) == 0
        ref = F.group_norm(x, 32, w, b, 1e-6)
        ref = F.silu(ref) if silu else ref
        d = np.abs(out.astype(np.int32) - as_u16(ref).astype(np.int32))
        assert d.max() <= 1 and (d != 0).mean() < 0.01, (d.max(), (d != 0).mean())          # statistics are summed in a different order
    m = torch.from_numpy(A.f32(r.standard_normal((2, 32, 64)) * 3)).bfloat16()
    o = np.zeros((2, 16, 64), np.float32)
    assert twin.selftok_latent_process_in(ptr(as_u16(m)), ptr(o), 2, 32, 16, 64, 0.0609, 1.5305, None) == 0
    np.testing.assert_array_equal(o, ((m[:, :16] - 0.0609) * 1.5305).float().numpy())            # sd3_impls.py:140-141 in bf16
    z = torch.from_numpy(A.f32(r.standard_normal(2048) * 2))
    ob = np.zeros(2048, np.uint16)
    assert twin.selftok_latent_process_out(ptr(z.numpy()), ptr(ob), 2048, 0.0609, 1.5305, None) == 0
    np.testing.assert_array_equal(ob, as_u16(((z / 1.5305) + 0.0609).bfloat16()))
    im = torch.from_numpy(A.f32(r.standard_normal(4096) * 1.5)).bfloat16()
    iu = as_u16(im).copy()
    assert twin.selftok_clamp01_bf16(ptr(iu), 4096, None) == 0
    ref = im.clone().clamp_(-1, 1).sub_(-1).div_(2)                                             # norm_ip: SelftokPipeline.py:135-137
    np.testing.assert_array_equal(iu, as_u16(ref))


def test_nhwc_conv_and_groupnorm_match_torch_cpu_bf16(twin):
    """the reference runs its VAE in bf16 on the CPU (oneDNN convolutions: fp32 accumulate incl. bias, one rounding): the channels-last entry
    points against torch-CPU's own bf16 conv2d / group_norm on the same values, for every geometry the VAE uses"""
    as_u16 = lambda t: np.ascontiguousarray(t.contiguous().view(torch.int16).numpy().view(np.uint16))
    r = A.rng(70)

    def check(name, got, ref_nchw, frac=0.005, ulps=1):
        ref = as_u16(ref_nchw.permute(0, 2, 3, 1))
        a, b = A.from_bf16(got), A.from_bf16(ref)                      # one bf16 ulp, or fp32-sum noise where the sum cancels to ~0
        assert (np.abs(a - b) <= ulps * 2.0 ** -7 * np.maximum(np.abs(b), 2.0 ** -6)).all() and (got != ref).mean() < frac, (name, float(np.abs(a - b).max()), float((got != ref).mean()))

    for name, (B, H, W, Cin, Cout, ks, stride, up, bn, resid) in dict(
            plain=(2, 8, 32, 64, 128, 3, 1, 0, 128, False), resid=(1, 6, 20, 32, 64, 3, 1, 0, 128, True), one=(1, 8, 32, 64, 128, 1, 1, 0, 128, False),
            down=(1, 8, 32, 32, 64, 3, 2, 0, 128, False), up=(1, 4, 16, 32, 64, 3, 1, 1, 128, False), narrow=(1, 8, 32, 64, 3, 3, 1, 0, 32, False)).items():
        x = torch.from_numpy(A.f32(r.standard_normal((B, Cin, H, W)))).bfloat16()
        w = torch.from_numpy(A.f32(r.standard_normal((Cout, Cin, ks, ks)) / np.sqrt(Cin * ks * ks))).bfloat16()
        bias = torch.from_numpy(A.f32(r.standard_normal(Cout))).bfloat16()
        xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
        if stride == 2:
            ref = F.conv2d(F.pad(xin, (0, 1, 0, 1)), w, bias, stride=2, padding=0)
        else:
            ref = F.conv2d(xin, w, bias, padding=ks // 2)
        cs = (Cout + 3) // 4 * 4
        res = torch.from_numpy(A.f32(r.standard_normal(ref.shape))).bfloat16() if resid else None
        if resid:
            ref = res + ref
        packed = np.zeros(twin.selftok_conv2d_packed_bytes(Cout, Cin, ks, bn) // 2, np.uint16)
        assert twin.selftok_conv2d_pack_weight_bf16(ptr(as_u16(w)), ptr(packed), Cout, Cin, ks, bn, None) == 0
        xn, bn_, out = as_u16(x.permute(0, 2, 3, 1)), as_u16(bias), np.zeros((B, ref.shape[2], ref.shape[3], cs), np.uint16)
        rn = None
        if resid:
            rn = np.zeros_like(out); rn[..., :Cout] = as_u16(res.permute(0, 2, 3, 1))
        rc = twin.selftok_conv2d_nhwc_bf16(ptr(xn), ptr(packed), ptr(bn_), ptr(rn) if resid else None, ptr(out), B, H, W, Cin, Cout, cs, cs, ks, stride, up, bn, None)
        assert rc == 0, twin.selftok_last_error()
        check(name, out[..., :Cout], ref)
        assert not out[..., Cout:].any()
    for C in (128, 512):
        x = torch.from_numpy(A.f32(r.standard_normal((2, C, 12, 12)) * 2 + 0.3)).bfloat16()
        w, b = torch.from_numpy(A.f32(1 + 0.2 * r.standard_normal(C))).bfloat16(), torch.from_numpy(A.f32(0.2 * r.standard_normal(C))).bfloat16()
        ws = np.zeros(twin.selftok_groupnorm_nhwc_workspace_bytes(2, 144, C), np.uint8)
        out = np.zeros((2, 144, C), np.uint16)
        assert twin.selftok_groupnorm_silu_nhwc_bf16(ptr(as_u16(x.permute(0, 2, 3, 1))), ptr(as_u16(w)), ptr(as_u16(b)), ptr(out), ptr(ws), 2, 144, C, 32, 1e-6, 1, None) == 0
        check(f"gn{C}", out.reshape(2, 12, 12, C), F.silu(F.group_norm(x, 32, w, b, 1e-6)), frac=0.01, ulps=2)     # a 1-ulp flip of the norm through SiLU's second rounding
