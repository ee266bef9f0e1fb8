"""-m gpu: csrc/vae_exact.hip against its bit-for-bit CPU twin oracle/vae_exact.c (itself pinned on the CPU to torch's own bf16
ops and, end to end, to the latents of the REFERENCE pipeline run: tests/test_vae_exact_cpu.py).  Everything here is BIT-EXACT:
a single differing bf16 element fails.  Layer shapes are the real ones of the SD3-VAE encoder at 256 x 256."""
import os

import numpy as np
import pytest
import torch

from oracle import vae_exact as VX
from selftoktokenizer_amd import ops, synth, weights as W

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _bits(t):
    return VX.bf16_bits(t)


def _rand_bf16(seed, shape, scale=1.0, shift=0.0):
    return (synth.hash_normalish(seed, shape) * scale + shift).to(torch.bfloat16)


def _same(gpu: torch.Tensor, ref_bits: np.ndarray, what: str):
    g = _bits(gpu)
    bad = int((g != ref_bits).sum())
    assert bad == 0, f"{what}: {bad} of {g.size} bf16 elements differ from the oracle"


# (name, Cin, Cout, H, ksize, stride, order, B)  -- every convolution shape of the encoder
CONVS = [("128->128 3x3 @256", 128, 128, 256, 3, 1, 0, 1), ("Downsample 128 @256", 128, 128, 256, 3, 2, 3, 1), ("128->256 @128", 128, 256, 128, 3, 1, 0, 1),
         ("256->256 @128", 256, 256, 128, 3, 1, 0, 1), ("shortcut 128->256 1x1", 128, 256, 128, 1, 1, 0, 1), ("Downsample 256 @128", 256, 256, 128, 3, 2, 3, 2),
         ("256->512 @64", 256, 512, 64, 3, 1, 0, 1), ("512->512 @64", 512, 512, 64, 3, 1, 0, 1), ("Downsample 512 @64", 512, 512, 64, 3, 2, 0, 2),
         ("512->512 @32", 512, 512, 32, 3, 1, 0, 2), ("attention projection 1x1", 512, 512, 32, 1, 1, 0, 2), ("conv_out 512->32", 512, 32, 32, 3, 1, 0, 2)]


@pytest.mark.parametrize("name,cin,cout,H,k,stride,order,B", CONVS, ids=[c[0] for c in CONVS])
def test_conv_exact_order(name, cin, cout, H, k, stride, order, B):
    x = _rand_bf16(0xC0 + cin + H, (B, H, H, cin), 1.3, 0.1)
    w = _rand_bf16(0xC1 + cout, (cout, k, k, cin), (1.0 / (cin * k * k)) ** 0.5)
    b = _rand_bf16(0xC2, (cout,), 0.1)
    assert VX.conv_order(cin, k, stride, H) == order
    Ho = H // stride
    res = _rand_bf16(0xC3, (B, Ho, Ho, cout)) if (k == 3 and stride == 1 and cin == cout) else None
    ref = VX.conv2d(_bits(x), _bits(w), _bits(b), stride=stride, pad=1 if (k == 3 and stride == 1) else 0, residual=None if res is None else _bits(res))
    out = ops.vx_conv2d(x.cuda(), w.cuda(), b.cuda(), stride=stride, residual=None if res is None else res.cuda(), order=order)
    torch.cuda.synchronize()
    _same(out, ref, name)


def test_conv_in_exact_order():
    B, H = 2, 256
    x8 = torch.zeros(B, H, H, 8, dtype=torch.bfloat16)
    x8[..., :3] = synth.synthetic_images(B).permute(0, 2, 3, 1).to(torch.bfloat16)
    x8[..., 3:] = 7.0                                              # padding channels must be ignored
    w = _rand_bf16(0xD1, (128, 3, 3, 3), 0.2)
    b = _rand_bf16(0xD2, (128,), 0.1)
    ref = VX.conv2d(_bits(x8[..., :3].contiguous()), _bits(w), _bits(b))
    out = ops.vx_conv2d(x8.cuda(), w.cuda(), b.cuda(), order=2)
    _same(out, ref, "conv_in")


def test_silu_table_equals_torch_cpu():
    tab = ops.vx_silu_table("cuda").cpu().numpy().view(np.uint16)
    gold = np.load(os.path.join(GOLD, "silu_bf16_table.npy"))
    x = (np.arange(65536, dtype=np.uint32) << 16).view(np.float32)
    fin = np.isfinite(x)
    assert np.array_equal(tab[fin], gold[fin]), np.nonzero(tab[fin] != gold[fin])[0][:10]
    nan_out = np.isnan((gold.astype(np.uint32) << 16).view(np.float32))
    assert np.isnan((tab.astype(np.uint32) << 16).view(np.float32))[nan_out].all()


@pytest.mark.parametrize("B,C,H", [(1, 128, 256), (2, 128, 128), (1, 256, 128), (2, 256, 64), (1, 512, 64), (2, 512, 32)])
def test_groupnorm_exact_statistics_and_output(B, C, H):
    x = _rand_bf16(0xE0 + C + H, (B, H, H, C), 1.7, 0.3)
    g = (synth.hash_uniform(0xE1, (C,), 0.9, 1.1)).to(torch.bfloat16)
    b = _rand_bf16(0xE2, (C,), 0.1)
    tab = VX.silu_table()
    for act in (True, False):
        ref, st = VX.group_norm(_bits(x), _bits(g), _bits(b), silu=tab if act else None, want_stats=True)
        out, st_g = ops.vx_groupnorm(x.cuda(), g.cuda(), b.cuda(), silu_table=ops.vx_silu_table("cuda") if act else None, want_stats=True)
        assert np.array_equal(st_g.cpu().numpy().view(np.uint32), st.view(np.uint32)), "mean / rstd bits differ from ATen's"
        _same(out, ref, f"GroupNorm {'+ SiLU ' if act else ''}C={C} H={H}")


def test_glibc_expf_twin():
    g = torch.Generator().manual_seed(3)
    x = torch.cat([-torch.rand(200000, generator=g) * 30.0, -torch.rand(200000, generator=g) * 0.5, torch.rand(1000, generator=g) * 20.0,
                   torch.tensor([0.0, -0.0, -87.0, -88.5, -100.0, -103.9, -104.5, -1e30, float("-inf"), 88.0, 89.0])]).float()
    y = ops.vx_expf(x.cuda()).cpu()
    ref = torch.tensor([VX.expf(float(v)) for v in x[:5000]] + [VX.expf(float(v)) for v in x[-11:]])
    got = torch.cat([y[:5000], y[-11:]])
    assert torch.equal(got.view(torch.int32), ref.view(torch.int32))
    # and against libm itself (the build container's glibc, where the oracle was pinned to it on 4e7 inputs): torch.exp is another function
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.expf.restype = ctypes.c_float
    libm.expf.argtypes = [ctypes.c_float]
    ref2 = np.array([libm.expf(float(v)) for v in x[::97]], dtype=np.float32)
    assert np.array_equal(y[::97].numpy().view(np.uint32), ref2.view(np.uint32))


def test_attention_exact_order():
    B, T, C = 2, 1024, 512
    q, k, v = (_rand_bf16(0xA0 + i, (B, T, C), 1.5 if i < 2 else 1.0) for i in range(3))
    ref = VX.attention(_bits(q), _bits(k), _bits(v))
    out = ops.vx_attention(q.cuda(), k.cuda(), v.cuda())
    _same(out, ref, "attention")


# ---- other image sizes (round 5): the layer shapes of the VAE at 128 and 320 px -------------------------------------------------
# oneDNN's chunk order is a function of the layer's spatial size (tools/probe_cpu_bf16/, profiles/r5_cpu_bf16_resolution_orders.txt): a stride-2
# layer is channel-block major iff its input is >= 102 wide.  Row counts that are not a multiple of the kernel's 64 / 128-row tile (1600 = 40 x 40).
CONVS_RES = [("128 px: Downsample 256 @64 -> order 0", 256, 256, 64, 3, 2, 0, 1), ("128 px: Downsample 128 @128", 128, 128, 128, 3, 2, 3, 1),
             ("128 px: 512->512 @16", 512, 512, 16, 3, 1, 0, 1), ("128 px: conv_out 512->32 @16", 512, 32, 16, 3, 1, 0, 1),
             ("320 px: Downsample 256 @160", 256, 256, 160, 3, 2, 3, 1), ("320 px: Downsample 512 @80 -> order 0", 512, 512, 80, 3, 2, 0, 1),
             ("320 px: 512->512 @40, 1600 rows", 512, 512, 40, 3, 1, 0, 1), ("320 px: conv_out 512->32 @40, 1600 rows on 128-row tiles", 512, 32, 40, 3, 1, 0, 1),
             ("320 px: 1x1 @40, 3 images = 4800 rows", 512, 512, 40, 1, 1, 0, 3), ("320 px: 128->128 @320", 128, 128, 320, 3, 1, 0, 1)]


@pytest.mark.parametrize("name,cin,cout,H,k,stride,order,B", CONVS_RES, ids=[c[0] for c in CONVS_RES])
def test_conv_exact_order_other_sizes(name, cin, cout, H, k, stride, order, B):
    test_conv_exact_order(name, cin, cout, H, k, stride, order, B)


@pytest.mark.parametrize("cin,cout,H,up", [(256, 128, 64, False), (256, 256, 32, True), (512, 32, 16, False)])
def test_conv_order_1_channel_block_major_into_one_total(cin, cout, H, up):
    """order 1: what oneDNN does for a 3x3 layer whose bf16 activation reaches 2^31 bytes (64 images decoded in one call: the two layers around
    [64, 256, 256, 256]; tools/probe_cpu_bf16/check_conv_batch64.py) -- the kernel's third traversal against the C oracle at small sizes"""
    x = _rand_bf16(0xCA + cin, (2, H, H, cin), 1.3, 0.1)
    w = _rand_bf16(0xCB + cout, (cout, 3, 3, cin), (1.0 / (cin * 9)) ** 0.5)
    b = _rand_bf16(0xCC, (cout,), 0.1)
    ref = VX.conv2d(VX.upsample2x(_bits(x)) if up else _bits(x), _bits(w), _bits(b), order=1)
    out = ops.vx_conv2d(x.cuda(), w.cuda(), b.cuda(), order=1, upsample=up)
    _same(out, ref, f"order 1, {cin}->{cout}")
    assert int((_bits(out) != VX.conv2d(VX.upsample2x(_bits(x)) if up else _bits(x), _bits(w), _bits(b), order=0)).sum()) > 0      # and it is another order


def test_decoder_upsampling_conv_ragged_rows():
    """nearest-2x input addressing at 20 -> 40 (1600 output rows per image)"""
    x = _rand_bf16(0xC7, (1, 20, 20, 512), 1.3, 0.1)
    w = _rand_bf16(0xC8, (512, 3, 3, 512), (1.0 / (512 * 9)) ** 0.5)
    b = _rand_bf16(0xC9, (512,), 0.1)
    ref = VX.conv2d(VX.upsample2x(_bits(x)), _bits(w), _bits(b), order=0)
    out = ops.vx_conv2d(x.cuda(), w.cuda(), b.cuda(), order=0, upsample=True)
    _same(out, ref, "upsampling conv 20 -> 40")


# (B, C, H): the pass-1 routes of the exact GroupNorm -- aligned nodes of 16 / 4 / 2 chunks (H*W % 512 == 0; 25 ranges per channel at 320 / 160 px,
# 400 nodes per group at C = 512), raw half-moments per chunk (an odd number of chunks per channel: 80 x 80, 16 x 16), gathered chunks that straddle
# channel boundaries (40 x 40 = 6.25 chunks per channel)
GN_RES = [(1, 128, 320), (1, 256, 160), (1, 512, 160), (1, 256, 320), (2, 512, 80), (1, 256, 80), (2, 512, 40), (3, 512, 16), (1, 128, 64), (1, 256, 48), (2, 128, 12)]       # 12 x 12, 4 channels per group: 36 vectors = two chunks and a quarter


@pytest.mark.parametrize("B,C,H", GN_RES)
def test_groupnorm_exact_other_sizes(B, C, H):
    test_groupnorm_exact_statistics_and_output(B, C, H)


@pytest.mark.parametrize("B,T", [(2, 256), (1, 1600), (1, 576), (1, 32)])
def test_attention_exact_order_other_token_counts(B, T):
    """one kv block of 256 (128 px); three of 512 and one of 64 (320 px: the running maximum / sum / accumulator rescaled at every block); 512 + 64"""
    C = 512
    q, k, v = (_rand_bf16(0xA4 + i + T, (B, T, C), 1.5 if i < 2 else 1.0) for i in range(3))
    ref = VX.attention(_bits(q), _bits(k), _bits(v))
    out = ops.vx_attention(q.cuda(), k.cuda(), v.cuda())
    _same(out, ref, f"attention T={T}")


def test_encoder_latents_equal_the_reference_pipeline_bit_for_bit():
    """the whole point: VAE mean -> process_in for the 16 images of tests/golden/pipeline_b16.npz = the REFERENCE's own run, bit for bit,
    and with them the token ids from pixels: 8192 / 8192"""
    from selftoktokenizer_amd.config import default_config
    from mimogpt.infer.SelftokPipeline import SelftokPipeline
    g = np.load(os.path.join(GOLD, "pipeline_b16.npz"))
    sd = W.synthetic_state_dict(W.expected_shapes(512), device="cuda")
    pipe = SelftokPipeline(default_config(512), None, None, device="cuda", state_dict=sd, vae_state_dict=W.synthetic_vae_state_dict(device="cuda"),
                           verbose=False, vae_mode="exact")
    imgs = synth.synthetic_images(16, device="cuda")
    x0 = pipe.encode_latents(imgs)
    ref = torch.from_numpy(g["x0_bf16"]).view(torch.bfloat16).float()
    bad = int((x0.cpu() != ref).sum())
    assert bad == 0, f"{bad} of {ref.numel()} latent elements differ from the reference pipeline's"
    ids = pipe.encoding(imgs).cpu().numpy()
    flips = int((ids != g["tokens"].astype(np.int64)).sum())
    print("token ids from pixels vs the reference pipeline run:", ids.size - flips, "/", ids.size)
    assert flips == 0
    assert tuple(pipe.vae.encode_moments(imgs[:0]).shape) == (0, 32, 32, 32)       # empty batch through every exact-order entry
    # batch independence by construction: one image alone, and a batch of 64 whose first 16 are these
    assert torch.equal(pipe.encode_latents(imgs[3:4]), x0[3:4])
    big = pipe.encode_latents(synth.synthetic_images(64, device="cuda"))
    assert torch.equal(big[:16], x0)


# ---- round 5: the DECODER in the reference's orders ---------------------------------------------------------------------------------
# (name, Cin, Cout, H of the convolution's input view, ksize, upsample, B)
DEC_CONVS = [("conv_in 16(+16 zeros)->512 @32", 16, 512, 32, 3, False, 2), ("up 512->512 @32->64", 512, 512, 32, 3, True, 2), ("up 512->512 @64->128", 512, 512, 64, 3, True, 1),
             ("512->256 @128", 512, 256, 128, 3, False, 1), ("shortcut 512->256 1x1 @128", 512, 256, 128, 1, False, 1), ("up 256->256 @128->256", 256, 256, 128, 3, True, 1),
             ("256->128 @256", 256, 128, 256, 3, False, 1), ("shortcut 256->128 1x1 @256", 256, 128, 256, 1, False, 1), ("conv_out 128->3(+29 zeros) @256", 128, 3, 256, 3, False, 1)]


@pytest.mark.parametrize("name,cin,cout,H,k,up,B", DEC_CONVS, ids=[c[0] for c in DEC_CONVS])
def test_decoder_conv_exact_order(name, cin, cout, H, k, up, B):
    """every convolution shape the decoder adds to the encoder's: nearest-2x upsampling as input addressing, conv_in's 16-channel chunks and
    conv_out's 3 output channels through zero padding -- bit-equal to oracle/vae_exact.c, which is bit-equal to F.conv2d on the reference host"""
    x = _rand_bf16(0xD0 + cin + H, (B, H, H, cin), 1.3, 0.1)
    w = _rand_bf16(0xD1 + cout, (cout, k, k, cin), (1.0 / (cin * k * k)) ** 0.5)
    b = _rand_bf16(0xD2, (cout,), 0.1)
    ref = VX.conv2d(VX.upsample2x(_bits(x)) if up else _bits(x), _bits(w), _bits(b), pad=1 if k == 3 else 0, order=0)
    xg, wg, bg = x.cuda(), w.cuda(), b.cuda()
    if cin % 32:
        xg, wg = torch.nn.functional.pad(xg, (0, 32 - cin)).contiguous(), torch.nn.functional.pad(wg, (0, 32 - cin)).contiguous()
    if cout % 32:
        wg, bg = torch.nn.functional.pad(wg, (0, 0, 0, 0, 0, 0, 0, 32 - cout)).contiguous(), torch.nn.functional.pad(bg, (0, 32 - cout)).contiguous()
    out = ops.vx_conv2d(xg, wg, bg, order=0, upsample=up)[..., :cout].contiguous()
    torch.cuda.synchronize()
    _same(out, ref, name)


@pytest.mark.parametrize("B,C,H", [(1, 512, 128), (1, 256, 256)])
def test_groupnorm_exact_decoder_shapes(B, C, H):
    x = _rand_bf16(0xE8 + C + H, (B, H, H, C), 1.7, 0.3)
    g = (synth.hash_uniform(0xE1, (C,), 0.9, 1.1)).to(torch.bfloat16)
    b = _rand_bf16(0xE2, (C,), 0.1)
    ref, st = VX.group_norm(_bits(x), _bits(g), _bits(b), silu=VX.silu_table(), want_stats=True)
    out, st_g = ops.vx_groupnorm(x.cuda(), g.cuda(), b.cuda(), silu_table=ops.vx_silu_table("cuda"), want_stats=True)
    assert np.array_equal(st_g.cpu().numpy().view(np.uint32), st.view(np.uint32)), "mean / rstd bits differ from ATen's"
    _same(out, ref, f"GroupNorm + SiLU C={C} H={H}")


def test_decoder_pixels_equal_the_reference_bit_for_bit():
    """the REFERENCE's VAE decode (tests/golden/vae_b1.npz: one synthetic latent; decode_b16.npz: the final latents of its 16-image pipeline
    run, crc32 of every image's bf16 pixels after norm_ip): our `vae.decode` gives the same bits on all 17 images, at any batch size"""
    import zlib
    from selftoktokenizer_amd.vae import AutoencoderKLGPU
    from selftoktokenizer_amd import pipeline as P
    vae = AutoencoderKLGPU(W.synthetic_vae_state_dict(device="cuda"), torch.device("cuda", torch.cuda.current_device()), mode="exact")
    g1 = np.load(os.path.join(GOLD, "vae_b1.npz"))
    z1 = synth.synthetic_latents(1).to(torch.bfloat16).cuda()
    rec = vae.decode(z1)[0]
    ref = torch.from_numpy(g1["rec"]).to(torch.bfloat16)
    bad = int((rec.cpu().view(torch.int16) != ref.view(torch.int16)).sum())
    assert bad == 0, f"{bad} of {ref.numel()} pixels differ from the reference decoder's"
    g16 = np.load(os.path.join(GOLD, "decode_b16.npz"))
    lat = torch.from_numpy(np.load(os.path.join(GOLD, "pipeline_b16.npz"))["lat"]).cuda()
    z = ops.latent_process_out(lat, P.SD3_SHIFT, P.SD3_SCALE)
    px = P.norm_ip(vae.decode(z)[0].contiguous())
    bits = px.cpu().view(torch.int16).numpy().view(np.uint16)
    crc = np.array([zlib.crc32(np.ascontiguousarray(bits[i]).tobytes()) for i in range(16)], dtype=np.uint32)
    assert np.array_equal(bits.reshape(16, -1)[:, :256], g16["head"]), "first pixels differ"
    assert np.array_equal(crc, g16["crc"]), f"images whose pixels differ from the reference's: {np.nonzero(crc != g16['crc'])[0].tolist()}"
    # batch independence by construction
    assert torch.equal(P.norm_ip(vae.decode(z[5:6])[0].contiguous()), px[5:6])
