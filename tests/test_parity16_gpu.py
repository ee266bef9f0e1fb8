"""-m gpu: end-to-end parity over 16 images against the REFERENCE's own pipeline run (tests/golden/pipeline_b16.npz, written by
tools/oracle/gen_golden.py pipeline16 from mimogpt.infer.SelftokPipeline on CPU with the synthetic weights): token ids from pixels
through the bf16 VAE, every flip characterised by the reference's top-1/top-2 gap and by the measured perturbation of the unit
feature that caused it, and reconstruction PSNR end to end and through the same decoder.

Where the numbers come from: the tokenizer / DiT path is fp32 and agrees with the reference to 1e-5 from identical latents; the
bf16 SD3-VAE is the one stage whose arithmetic no two implementations share bit for bit (the reference's CPU convolutions, the
oracle's CPU convolutions with Linear attention, MIOpen's GPU kernels): its latents differ by +-1 bf16 ulp (0.0156 .. 0.031 at
|x| = 2 .. 4) on a fraction of the elements.  That perturbation, pushed through the encoder, moves a unit feature by |dz|; a
token can flip only if its reference gap is below |dz| * |e_ref - e_new|.  The yardstick stored in the golden file and printed next
to the GPU's numbers is a SECOND CPU implementation of the same VAE -- oracle/ with the attention projections in diffusers' own
Linear formulation (the oracle's default, 1x1 convolutions like the mirror, is bit-identical to the reference and would read 0)."""
import os

import numpy as np
import pytest
import torch

from selftoktokenizer_amd import synth, weights as W
from selftoktokenizer_amd.config import default_config

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "pipeline_b16.npz")
B = 16


@pytest.fixture(scope="module")
def pipe():
    from mimogpt.infer.SelftokPipeline import SelftokPipeline
    sd = W.synthetic_state_dict(W.expected_shapes(512), device="cuda")
    p = SelftokPipeline(default_config(512), ckpt_path=None, sd3_path=None, device="cuda", state_dict=sd,
                        vae_state_dict=W.synthetic_vae_state_dict(device="cuda"))
    p.verbose = False
    return p


def _hist(vals, edges):
    h, _ = np.histogram(vals, bins=edges)
    return " ".join(f"[{edges[i]:.0e},{edges[i + 1]:.0e}):{int(h[i])}" for i in range(len(h)))


def test_ids_16_images_exact_mode_equals_the_reference_bit_for_bit(pipe):
    """round 4: the default VAE mode ('exact', csrc/vae_exact.hip) evaluates every reduction of the bf16 encoder in the reference's
    torch-CPU order: latents bit-equal to the reference pipeline's, hence ids from pixels 8192 / 8192 (the north star's "token ids
    bit-exact", end to end)"""
    g = np.load(GOLD)
    assert pipe.vae.mode == "exact"
    imgs = synth.synthetic_images(B, device="cuda")
    x0 = pipe.encode_latents(imgs).cpu()
    ref_x0 = torch.from_numpy(g["x0_bf16"]).view(torch.bfloat16).float()
    assert torch.equal(x0, ref_x0), f"{int((x0 != ref_x0).sum())} latent elements differ from the reference pipeline's"
    ids = pipe.encoding(imgs).cpu().numpy()
    flips = int((ids != g["tokens"].astype(np.int64)).sum())
    print(f"\ne2e ids vs the reference pipeline, {B} images, exact-order VAE encoder: {ids.size - flips} / {ids.size}")
    assert flips == 0
    # round 5: the Q-Former encoder runs in torch-CPU's orders too (csrc/encoder_exact.hip) -- the pre-quantizer FEATURES are the reference's bits
    z = pipe.model.encoder.features(pipe.encode_latents(imgs)).cpu().numpy()
    assert pipe.model.encoder.mode == "exact" and int((z.view(np.uint32) != g["z"].view(np.uint32)).sum()) == 0


def test_ids_64_images_one_batch_equal_the_reference_and_do_not_depend_on_the_batching(pipe):
    """BASELINE configs[1]'s encode leg: the reference's `encoding` on 64 images IN ONE BATCH (tests/golden/encode_b64.npz,
    SelftokPipeline.py:210-225): latents bit-equal, pre-quantizer features bit-equal, ids 32768 / 32768 -- including the 18 tokens whose two
    best codes are less than 1e-5 apart -- and IDENTICAL ids when the same 64 images are encoded as 4 x 16, 8 x 8 or 64 x 1"""
    g = np.load(os.path.join(os.path.dirname(GOLD), "encode_b64.npz"))
    imgs = synth.synthetic_images(64, device="cuda")
    x0 = pipe.encode_latents(imgs)
    ref_x0 = torch.from_numpy(g["x0_bf16"]).view(torch.bfloat16).float()
    assert torch.equal(x0.cpu(), ref_x0)
    z = pipe.model.encoder.features(x0).cpu().numpy()
    assert int((z.view(np.uint32) != g["z"].view(np.uint32)).sum()) == 0
    ids = pipe.encoding(imgs)
    ref = g["tokens"].astype(np.int64)
    mism = ids.cpu().numpy() != ref
    for b, k in np.argwhere(mism):
        print(f"flip: image {b} token {k}: reference {ref[b, k]} (gap {g['gap'][b, k]:.3e}, runner-up {g['id2'][b, k]}) -> {int(ids[b, k])}")
    print(f"\ne2e ids vs the reference's B = 64 run: {mism.size - int(mism.sum())} / {mism.size}; tokens with a reference gap below 1e-5: {int((g['gap'] < 1e-5).sum())}, smallest gap {g['gap'].min():.2e}")
    assert int(mism.sum()) == 0
    for gsz in (16, 8, 1):
        split = torch.cat([pipe.encoding(imgs[i:i + gsz]) for i in range(0, 64, gsz)])
        assert torch.equal(split, ids), f"ids depend on the batching ({64 // gsz} x {gsz}): {int((split != ids).sum())} differ"


def test_ids_16_images_vs_reference(pipe):
    """the `parity` VAE mode (bf16 matrix cores: the reference's PRECISION, its own summation order): what any implementation that
    does not reproduce oneDNN's order gets, characterised flip by flip"""
    from selftoktokenizer_amd.vae import AutoencoderKLGPU
    exact_vae = pipe.vae
    pipe.vae = AutoencoderKLGPU(W.synthetic_vae_state_dict(device="cuda"), pipe.device, mode="parity")
    try:
        _ids_parity_mode(pipe)
    finally:
        pipe.vae = exact_vae


def _ids_parity_mode(pipe):
    g = np.load(GOLD)
    ref = g["tokens"].astype(np.int64)
    imgs = synth.synthetic_images(B, device="cuda")
    ids = pipe.encoding(imgs).cpu().numpy()
    assert ids.shape == ref.shape
    mism = ids != ref
    nflip = int(mism.sum())
    gaps = g["gap"][mism]
    # measured perturbation of the unit feature of every token (GPU vs reference), and of the VAE latents that cause it
    x0 = pipe.encode_latents(imgs)
    z = pipe.model.encoder.features(x0).cpu()
    zn = torch.nn.functional.normalize(z.reshape(-1, 16), dim=-1)
    zr = torch.nn.functional.normalize(torch.from_numpy(g["z"]).reshape(-1, 16), dim=-1)
    dz = (zn - zr).norm(dim=-1).reshape(B, -1).numpy()
    dx = x0.cpu() - torch.from_numpy(g["x0_bf16"]).view(torch.bfloat16).float()
    mo = g["tokens_oracle"].astype(np.int64) != ref
    edges = [0, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3, 1e-2, 1.0]
    print(f"\ne2e ids vs the reference pipeline, {B} images: match {1 - mism.mean():.6f} ({nflip} flips of {mism.size}); "
          f"second CPU implementation (Linear attention projections) vs reference on the same images: {1 - mo.mean():.6f} ({int(mo.sum())} flips)")
    print("reference gap (top1 - top2) of the flipped tokens:", np.sort(gaps))
    print("  histogram of flip gaps      :", _hist(gaps, edges))
    print("  histogram of ALL token gaps :", _hist(g["gap"].reshape(-1), edges))
    print(f"VAE latent delta vs reference: max {float(dx.abs().max()):.4f} rms {float(dx.pow(2).mean().sqrt()):.5f} "
          f"(bf16 ulp at |x| in [2,4) = 0.0156; second CPU implementation vs reference rms {float((torch.from_numpy(g['x0_oracle_bf16']).view(torch.bfloat16).float() - torch.from_numpy(g['x0_bf16']).view(torch.bfloat16).float()).pow(2).mean().sqrt()):.5f})")
    print(f"unit-feature delta |dz| per token: median {np.median(dz):.2e} p99 {np.quantile(dz, 0.99):.2e} max {dz.max():.2e}")
    print(f"tokens whose reference gap is below 2 |dz| (could flip under this noise): {int((g['gap'] < 2 * dz).sum())}; flipped: {nflip}")
    # (1) every flip is explained by the measured upstream perturbation: for unit codes the score difference between the reference's
    #     winner and the new winner moves by at most |dz| * |e_ref - e_new| <= 2 |dz| -- a flip at a larger gap would be a kernel bug
    cb = pipe.model.encoder.codebook.cpu()
    if nflip:
        de = (cb[torch.from_numpy(ref[mism])] - cb[torch.from_numpy(ids[mism])]).norm(dim=-1).numpy()
        assert (gaps <= dz[mism] * de + 2e-6).all(), (gaps, dz[mism] * de)
    # (2) ... and that perturbation is small: flips sit only at near-ties of the reference.  Bound = 2 x the largest measured |dz| of
    #     this run, which itself must stay at the level bf16 VAE noise produces (measured: median 5e-4, max 1.7e-3 with the bias
    #     folded into the convolutions; 5e-3 with the separate bf16 bias add of rounds 1-2 -- profiles/r3_vae_modes_*.txt)
    bound = 2.0 * float(dz.max())
    assert bound < 5e-3, bound
    assert nflip == 0 or float(gaps.max()) < bound
    assert nflip == 0 or float(gaps.max()) < 1e-3          # measured: largest flip gap 2.6e-4
    # (3) the count stays at the measured level (this kernel set: 15, bit-stable; second CPU implementation: 7; rounds 1-2: 40) and
    #     every flip lands on the reference's runner-up code
    assert nflip <= 16, nflip
    if nflip:
        to_runner_up = int((ids[mism] == g["id2"].astype(np.int64)[mism]).sum())
        print("flips that went to the reference's runner-up:", to_runner_up, "of", nflip)
        assert to_runner_up == nflip


@pytest.mark.parametrize("gemm", ["fp32", "f16x2"])
def test_psnr_16_images_vs_reference(pipe, gemm):
    g = np.load(GOLD)
    ref = g["tokens"].astype(np.int64)
    orig = (synth.synthetic_images(B) + 1.0) / 2.0

    def psnr_each(px):
        mse = ((px.float().cpu() - orig) ** 2).reshape(B, -1).double().mean(dim=1)
        return (10.0 * torch.log10(1.0 / mse)).numpy()
    assert pipe.set_gemm(gemm) == gemm
    try:
        rec, lat = pipe.decoding(ref, noise=synth.synthetic_noise(B), return_latent=True)
    finally:
        pipe.set_gemm("fp32")
    lat_ref = torch.from_numpy(g["lat"]).cuda()
    lat_err = float((lat - lat_ref).abs().max())
    d_e2e = np.abs(psnr_each(rec) - g["psnr_ref"])
    both = pipe._to_pixels(torch.cat([lat_ref, lat]))            # ONE decoder call: only the latents differ
    d_same = np.abs(psnr_each(both[:B]) - psnr_each(both[B:]))
    d_or = np.abs(g["psnr_oracle"] - g["psnr_ref"])
    # the floor of this metric: the reference's OWN latents against themselves moved by 2 fp32 ulps, through the same decoder call.  The
    # decoder starts by rounding the latents to bf16 (SelftokPipeline.py:287): a 1e-6 perturbation flips that rounding for a handful
    # of the 16384 latent values of an image, and each flip moves the image's PSNR by ~1e-4 .. 1e-3 dB
    floor_px = pipe._to_pixels(torch.cat([lat_ref, lat_ref * (1.0 + 2.0 ** -22)]))
    d_floor = np.abs(psnr_each(floor_px[:B]) - psnr_each(floor_px[B:]))
    print(f"\n[{gemm}] final latents after 50 steps vs the reference: max abs diff {lat_err:.3e}")
    print(f"[{gemm}] reconstruction PSNR vs original, |ours - reference| over {B} images (reference mean {g['psnr_ref'].mean():.4f} dB):")
    print(f"   end to end (our latents, our MIOpen bf16 decoder)   : mean {d_e2e.mean():.2e} max {d_e2e.max():.2e} dB   each {np.round(d_e2e, 5)}")
    print(f"   same decoder (reference latents vs ours, one call)  : mean {d_same.mean():.2e} max {d_same.max():.2e} dB")
    print(f"   second CPU implementation's decoder on the reference's latents: mean {d_or.mean():.2e} max {d_or.max():.2e} dB   each {np.round(d_or, 5)}")
    print(f"   floor: reference latents vs themselves * (1 + 2^-22), same decoder call: mean {d_floor.mean():.2e} max {d_floor.max():.2e} dB")
    assert lat_err < 2e-5                                        # measured 2.9e-6 in both arithmetics
    # the north star's 1e-3 dB where only OUR path differs: met on average with a wide margin; the maximum over 16 images sits at the
    # metric's own floor (see d_floor: latents that differ by 3e-6 already reach ~1e-3 dB on single images)
    assert d_same.mean() < 5e-4 and d_same.max() < 1e-3, (d_same.mean(), d_same.max())      # the north star's 1e-3 dB on every image (measured max 6.3e-4)
    # end to end the delta is the bf16 decoder's implementation noise: it stays inside the spread between two CPU implementations of
    # the same decoder on the same latents (second CPU implementation vs reference: mean 3.9e-4, max 1.09e-3 dB; measured here: mean 2.9e-4, max 8.8e-4;
    # rounds 1-2, separate bias add + solver search: mean 6.1e-3)
    assert d_e2e.mean() < 5e-4 and d_e2e.max() < 1e-3, (d_e2e.mean(), d_e2e.max(), d_or.mean(), d_or.max())   # north star: 1e-3 dB, every image (measured mean 2.5e-4, max 8.4e-4)


def test_psnr_64_images_headline_modes_vs_reference(pipe):
    """the arithmetics the reported images/s are measured in (gemm 'fp32' = the headline, 'f16x2') pinned AT THE CONFIGURED BATCH of BASELINE
    configs[1]: the 64 ids + hash noise of the reference pipeline's own one-batch run (SelftokPipeline.py:227-294; tests/golden/encode_b64.npz, pipeline_b64.npz)
    through 50 steps.  The comparison latents are the exact mode's, proven to be the reference's by the golden crc32 of every image.  Gates = the north star:
    every image within 1e-3 dB of the reference's PSNR, mean below 5e-4 -- end to end with the mode's own decoder and through the same decoder.  `bench.py`
    reports the same object as `parity_64`."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    r = bench.parity_64(pipe)
    assert "error" not in r, r
    ex = r["exact"]
    assert ex["images_with_final_latents_bit_equal_to_the_reference"] == 64 and ex["images_with_pixels_bit_equal_to_the_reference"] == 64 and ex["psnr_identical_to_the_reference"]
    assert pipe.model.model.gemm == "fp32" and pipe.vae.decode_mode == "parity"          # the leg restores the pipeline's mode and decoder
    for mode in ("fp32", "f16x2"):
        m = r[mode]
        e2e, same = m["end_to_end"], m[f"same_decoder_{m['vae_decode']}"]
        print(f"\n[{mode}, decoder {m['vae_decode']}] 64 images vs the reference's one-batch run: final latents max |delta| {m['final_latent_max_abs_delta']:.3e}")
        print(f"   end to end  : mean {e2e['mean_dB']:.2e} max {e2e['max_dB']:.2e} dB (image {e2e['image_of_max']})")
        for dec in ("parity", "exact"):
            s = m[f"same_decoder_{dec}"]
            print(f"   same decoder ({dec}): mean {s['mean_dB']:.2e} max {s['max_dB']:.2e} dB (image {s['image_of_max']}; floor there {s['floor']['at_the_image_of_max_dB']:.2e}, "
                  f"floor mean {s['floor']['mean_dB']:.2e} max {s['floor']['max_dB']:.2e}); end to end with it: mean {s['end_to_end_with_this_decoder']['mean_dB']:.2e} "
                  f"max {s['end_to_end_with_this_decoder']['max_dB']:.2e}")
        assert m["final_latent_max_abs_delta"] < 2e-5
        assert e2e["mean_dB"] < 5e-4 and e2e["max_dB"] < 1e-3, e2e
        assert same["mean_dB"] < 5e-4 and same["max_dB"] < 1e-3, same
