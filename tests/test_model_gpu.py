"""-m gpu: the MI355X product path (HIP kernels + hipBLASLt GEMMs) against the golden vectors captured from
the reference (tests/golden/, see tools/oracle/gen_golden.py) on the same hash-generated weights/inputs.

Tolerances (floating point, stated per BASELINE north_star): token ids bit-exact at the VQ kernel boundary
(tests/test_vq_gpu.py); end to end ids are reported as a match fraction because the upstream fp32 GEMMs of two
different BLAS libraries differ at ~1e-6 and a token whose top-1/top-2 gap is below that may flip -- every
mismatching token must have a golden gap below GAP_TOL.  Velocities/latents: 2e-3 absolute on O(4) values.
"""
import os

import numpy as np
import pytest
import torch

from selftoktokenizer_amd import ops, synth, weights as W

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
GAP_TOL = 2e-4


def gold(name):
    return np.load(os.path.join(GOLD, name))


@pytest.fixture(scope="module")
def sd():
    return W.synthetic_state_dict(W.expected_shapes(512), device="cuda")


@pytest.fixture(scope="module")
def encoder(sd):
    from selftoktokenizer_amd.encoder import QformerEncoderGPU
    return QformerEncoderGPU(sd, torch.device("cuda"), 512)


@pytest.fixture(scope="module", params=["fp32", "f16x2"])
def dit(sd, request):
    """both GEMM arithmetics of the block Linears: hipBLASLt fp32 and the f16x2 split kernel (csrc/gemm_split.hip)"""
    from selftoktokenizer_amd.mmdit import MMDiTGPU
    d = MMDiTGPU(sd, torch.device("cuda"), 512, gemm=request.param)
    assert d.gemm == request.param
    yield d
    assert int(d.overflow.item()) == 0


def test_encoder_features_and_ids(encoder):
    g = gold("encoder_b2.npz")
    x0 = synth.synthetic_latents(2, device="cuda")
    z = encoder.features(x0)
    err = float((z.cpu() - torch.from_numpy(g["z"])).abs().max())
    print("encoder z max abs err", err)
    assert err < 2e-4
    outs_q, ids = encoder(x0, d=None)
    assert ids.dtype == torch.int64 and tuple(ids.shape) == (2, 512)
    ids = ids.cpu().numpy()
    mism = ids != g["ids"]
    print("e2e token match", 1.0 - mism.mean(), "gaps of mismatches", g["gap"][mism])
    assert mism.sum() <= 1                      # measured: 1024 / 1024; one flip allowed, and only at a near-tie of the reference
    assert (g["gap"][mism] < GAP_TOL).all()
    # kernel boundary: feeding the reference's own features must give the reference's ids exactly
    ids_k = ops.vq_encode(torch.from_numpy(g["z"]).cuda(), encoder.codebook_packed, packed=True).cpu().numpy()
    np.testing.assert_array_equal(ids_k, g["ids"])
    torch.testing.assert_close(outs_q.cpu()[~torch.from_numpy(mism)], torch.from_numpy(g["outs_q"])[~torch.from_numpy(mism)], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_dit_forward_vs_reference(dit, encoder, case):
    g = gold("dit_forward_b1.npz")
    ids = torch.from_numpy(synth.synthetic_token_ids(1)).cuda()
    ehs = encoder.codes_ln(ids)
    torch.testing.assert_close(ehs.cpu(), torch.from_numpy(g["ehs"]), rtol=1e-5, atol=1e-5)
    x = synth.synthetic_noise(1, device="cuda")
    t = torch.full((1,), float(g[f"t_{case}"]), device="cuda")
    k = int(g[f"k_{case}"])
    mask = (torch.arange(512, device="cuda")[None] <= k)
    v, _ = dit(x, t, encoder_hidden_states=ehs, mask=mask, context_see_xt=True)
    ref = torch.from_numpy(g[f"v_{case}"])
    err = float((v.cpu() - ref).abs().max())
    print(f"dit case {case} (k={k}) max abs err {err:.3e} of absmax {float(ref.abs().max()):.2f}")
    assert err < 2e-3


def test_f16x2_gemm_gate_vs_reference(sd, encoder):
    """VERDICT r1 item 6 gate: with the block Linears on the f16x2 split kernel the velocity error against the REFERENCE
    (CPU, MKL fp32) must not exceed the error of the hipBLASLt fp32 path by more than fp32 noise."""
    from selftoktokenizer_amd.mmdit import MMDiTGPU
    g = gold("dit_forward_b1.npz")
    d = MMDiTGPU(sd, torch.device("cuda"), 512)
    ids = torch.from_numpy(synth.synthetic_token_ids(1)).cuda()
    ehs = encoder.codes_ln(ids)
    x = synth.synthetic_noise(1, device="cuda")
    errs = {}
    for mode in ("fp32", "f16x2"):
        assert d.set_gemm(mode) == mode
        for case in "abc":
            t = torch.full((1,), float(g[f"t_{case}"]), device="cuda")
            mask = (torch.arange(512, device="cuda")[None] <= int(g[f"k_{case}"]))
            v, _ = d(x, t, encoder_hidden_states=ehs, mask=mask, context_see_xt=True)
            errs[mode, case] = float((v.cpu() - torch.from_numpy(g[f"v_{case}"])).abs().max())
    print("velocity max abs err vs reference:", errs)
    assert int(d.overflow.item()) == 0
    for case in "abc":
        assert errs["f16x2", case] <= 1.25 * errs["fp32", case] + 2e-6
        assert errs["f16x2", case] < 5e-5


def test_f16x2_presplit_activations_bit_identical(sd, encoder):
    """f16x2 mode hands every block Linear its input already split by the producing kernel (LN-modulate, attention, fc1+GELU
    epilogues) and stages it by LDS-DMA; the split is the same function of the fp32 value as the in-GEMM split, so the whole
    forward must be bit-identical to the path that keeps fp32 activations between kernels."""
    from selftoktokenizer_amd.mmdit import MMDiTGPU
    d = MMDiTGPU(sd, torch.device("cuda"), 512, gemm="f16x2")
    d.SPLITK = False            # B = 2 is inside the small-M regime: compare the single-pass kernels (split-K changes the summation order)
    ids = torch.from_numpy(synth.synthetic_token_ids(2)).cuda()
    ehs = encoder.codes_ln(ids)
    x = synth.synthetic_noise(2, device="cuda")
    t = torch.tensor([620.0, 333.0], device="cuda")
    outs = {}
    for pre in (True, False):
        d.PRESPLIT = pre
        for name, mask in (("masked", torch.arange(512, device="cuda")[None] <= torch.tensor([[375], [100]], device="cuda")), ("full", None)):
            outs[pre, name], _ = d(x, t, encoder_hidden_states=ehs, mask=mask, context_see_xt=True)
    assert int(d.overflow.item()) == 0
    for name in ("masked", "full"):
        assert torch.equal(outs[True, name], outs[False, name]), name


def test_f16x2_small_m_split_k_forward(sd, encoder):
    """one image (M = 256 image rows, k + 1 context rows): the block Linears take the split-K entry points (ops.f16x2_ksplit); the
    velocity agrees with the single-pass kernels' to fp32 summation noise, run to run bit for bit, and B = 8 (2048 rows) is untouched"""
    from selftoktokenizer_amd.mmdit import MMDiTGPU
    d = MMDiTGPU(sd, torch.device("cuda"), 512, gemm="f16x2")
    ids = torch.from_numpy(synth.synthetic_token_ids(1)).cuda()
    ehs = encoder.codes_ln(ids)
    x = synth.synthetic_noise(1, device="cuda")
    t = torch.tensor([620.0], device="cuda")
    mask = torch.arange(512, device="cuda")[None] <= 300
    assert d.SPLITK and ops.f16x2_ksplit(256, 1536, 1536) > 1 and ops.f16x2_ksplit(8 * 256, 1536, 1536) == 1
    v1, _ = d(x, t, encoder_hidden_states=ehs, mask=mask, context_see_xt=True)
    v2, _ = d(x, t, encoder_hidden_states=ehs, mask=mask, context_see_xt=True)
    d.SPLITK = False
    v0, _ = d(x, t, encoder_hidden_states=ehs, mask=mask, context_see_xt=True)
    assert int(d.overflow.item()) == 0 and torch.equal(v1, v2)
    e = float((v1 - v0).abs().max())
    print(f"B=1 MMDiT.forward, split-K vs single-pass Linears: max abs diff {e:.3e} (|v| up to {float(v0.abs().max()):.2f})")
    assert e < 2e-5


def test_dit_truncated_context_equals_masked(dit, encoder):
    """sampler fast path (context truncated to k+1 tokens, no mask) == per-sample kvis path"""
    ids = torch.from_numpy(synth.synthetic_token_ids(2)).cuda()
    ehs = encoder.codes_ln(ids)
    x = synth.synthetic_noise(2, device="cuda")
    from selftoktokenizer_amd.encoder import sinusoid_host
    tf = sinusoid_host(torch.tensor([620.0, 620.0])).cuda()
    ctx0 = dit.embed_context(ehs)
    y_fast = dit.velocity_tokens(x, tf, ctx0, 376, True)
    kvis = torch.tensor([375, 375], dtype=torch.int32, device="cuda")
    y_mask = dit.core(dit.embed_image(x), dit.time_embed(tf), ctx0, True, kvis)
    torch.testing.assert_close(y_fast, y_mask, rtol=1e-4, atol=1e-4)


def test_renderer_vs_reference():
    g = gold("renderer_b1.npz")
    from selftoktokenizer_amd.encoder import QformerEncoderGPU
    from selftoktokenizer_amd.mmdit import MMDiTGPU
    sd = W.synthetic_state_dict(W.expected_shapes(512, renderer=True), device="cuda")
    enc = QformerEncoderGPU(sd, torch.device("cuda"), 512)
    dit = MMDiTGPU(sd, torch.device("cuda"), 512, renderer=True)
    ids = torch.from_numpy(g["ids"]).cuda()
    out, _ = dit(y=None, encoder_hidden_states=enc.codes_ln(ids))
    err = float((out.cpu() - torch.from_numpy(g["latent"])).abs().max())
    print("renderer max abs err", err)
    assert err < 2e-3


def test_vae_vs_mirror():
    """bf16 VAE against the reference's in-repo mirror run on the CPU (vae_b1.npz; the arithmetic of record, diffusers, is unavailable).
    A bf16 network is chaotic at the ulp level: fp32 summation-order differences flip 0.05 % of a convolution's
    output roundings and compound to ~85 % of the latent elements one ulp off at the end -- so the comparison is statistical: the rms
    deviation must stay below one bf16 ulp of the typical magnitude, the maximum within a few ulps, and the parity mode (own
    implicit-GEMM convolutions, csrc/conv.hip) must beat the rounds 1-2 arithmetic (`mode='fast'`: separate bf16 bias add, searched
    MIOpen solvers) on both; `mode='miopen'` is the same arithmetic through MIOpen's GEMM algorithm."""
    g = gold("vae_b1.npz")
    from selftoktokenizer_amd.vae import AutoencoderKLGPU
    img = synth.synthetic_images(1).to(torch.bfloat16)
    lat = synth.synthetic_latents(1).to(torch.bfloat16)
    ref, ref2 = torch.from_numpy(g["mean"]), torch.from_numpy(g["rec"])
    res = {}
    for mode in ("exact", "parity", "miopen", "fast"):
        vae = AutoencoderKLGPU(W.synthetic_vae_state_dict(), torch.device("cuda"), mode=mode)
        mean = vae.encode(img.cuda())[0].mode().float().cpu()
        rec = vae.decode(lat.cuda())[0].float().cpu()
        d1, d2 = mean - ref, rec - ref2
        res[mode] = (float(d1.abs().max()), float(d1.pow(2).mean().sqrt()), float(d2.abs().max()), float(d2.pow(2).mean().sqrt()))
        print(f"vae [{mode}] vs mirror: encoder mean max {res[mode][0]:.4f} rms {res[mode][1]:.5f} (|mean| max {float(ref.abs().max()):.2f}); "
              f"decoder max {res[mode][2]:.4f} rms {res[mode][3]:.5f} (|rec| max {float(ref2.abs().max()):.2f}), "
              f"decoder PSNR vs mirror (range 2) {10 * np.log10(4.0 / max(res[mode][3] ** 2, 1e-12)):.2f} dB")
        if mode != "fast":         # bit-stable: a second call returns the same bits
            assert torch.equal(vae.encode(img.cuda())[0].mode().float().cpu(), mean)
    assert res["exact"][0] == 0.0, "the exact-order encoder must reproduce the mirror's latent mean bit for bit"
    assert res["exact"][2:] == (0.0, 0.0), "round 5: the exact-order DECODER must reproduce the mirror's pixels bit for bit"
    e1, r1, e2, r2 = res["parity"]
    assert e1 <= 0.0313 and r1 < 0.0078            # encoder: <= 2 ulps at |x| in [2, 4) anywhere, rms below one ulp at |x| ~ 1
    assert e2 <= 0.0625 and r2 < 0.0117            # decoder output (|rec| up to 3.3: 4 ulps anywhere, rms below 3/4 ulp at |x| in [2, 4))
    assert r1 < res["fast"][1] and r2 < res["fast"][3]
    for i, bound in enumerate((0.0313, 0.0078, 0.0625, 0.0117)):                    # rounds 1-3 route to the same arithmetic (MIOpen GEMM algorithm)
        assert res["miopen"][i] <= bound
