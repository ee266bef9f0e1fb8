"""not gpu: oracle/vae_exact.c -- the bit-for-bit CPU restatement of torch-CPU's bf16 SD3-VAE encoder arithmetic -- PINNED:
  * every convolution shape of the encoder against F.conv2d (oneDNN's AMX kernel) on random bf16 data: 0 differing elements;
  * GroupNorm against torch's group_norm (outputs AND the fp32 mean / rstd ATen returns), SiLU table against torch's silu on all
    65536 bf16 inputs, attention against F.scaled_dot_product_attention, expf against libm;
  * the whole encoder against the latents of the REFERENCE pipeline's own run (tests/golden/pipeline_b16.npz).
These comparisons only mean something on the machine class the reference ran on (this build container: AMX-bf16 Xeon, torch 2.10 CPU):
on a host whose oneDNN dispatches another kernel the torch-vs-oracle tests skip; the golden-vector tests run everywhere."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import abi_cases as A
from oracle import vae_exact as VX
from selftoktokenizer_amd import synth, weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _amx_host() -> bool:
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        return False
    return "amx_bf16" in flags and "avx512_bf16" in flags


needs_amx = pytest.mark.skipif(not _amx_host(), reason="torch-CPU's bf16 convolution dispatches oneDNN's AMX kernel only on an AMX-bf16 host "
                                                       "(the reference's run and the probes were made on one)")


def _rand(seed, shape, scale=1.0, shift=0.0):
    return (synth.hash_normalish(seed, shape) * scale + shift).to(torch.bfloat16)


def _nhwc_bits(t):          # [B,C,H,W] bf16 -> NHWC uint16
    return VX.bf16_bits(t.permute(0, 2, 3, 1))


# (name, Cin, Cout, H, ksize, stride) -- every convolution shape of the SD3-VAE encoder at 256 x 256
SHAPES = [("conv_in", 3, 128, 256, 3, 1), ("128@256", 128, 128, 256, 3, 1), ("down128", 128, 128, 256, 3, 2), ("128->256@128", 128, 256, 128, 3, 1),
          ("256@128", 256, 256, 128, 3, 1), ("sc128->256", 128, 256, 128, 1, 1), ("down256", 256, 256, 128, 3, 2), ("256->512@64", 256, 512, 64, 3, 1),
          ("512@64", 512, 512, 64, 3, 1), ("sc256->512", 256, 512, 64, 1, 1), ("down512", 512, 512, 64, 3, 2), ("512@32", 512, 512, 32, 3, 1),
          ("attn1x1", 512, 512, 32, 1, 1), ("conv_out", 512, 32, 32, 3, 1)]


@needs_amx
@pytest.mark.parametrize("name,cin,cout,H,k,stride", SHAPES, ids=[s[0] for s in SHAPES])
def test_conv_order_equals_onednn(name, cin, cout, H, k, stride):
    x = _rand(0x11 + cin + H, (1, cin, H, H), 1.2, 0.05)
    w = _rand(0x12 + cout, (cout, cin, k, k), (1.0 / (cin * k * k)) ** 0.5)
    b = _rand(0x13, (cout,), 0.1)
    with torch.no_grad():
        ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2) if stride == 2 else F.conv2d(x, w, b, padding=k // 2)
    mine = VX.conv2d(_nhwc_bits(x), VX.bf16_bits(w.permute(0, 2, 3, 1)), VX.bf16_bits(b), stride=stride, pad=(1 if k == 3 and stride == 1 else 0))
    bad = int((mine != _nhwc_bits(ref)).sum())
    assert bad == 0, f"{name}: {bad} of {mine.size} outputs differ from F.conv2d"


@needs_amx
@pytest.mark.parametrize("B,Cn,H", [(1, 128, 256), (2, 256, 64), (1, 512, 32)])
def test_groupnorm_equals_aten(B, Cn, H):
    x = _rand(0x21 + Cn, (B, Cn, H, H), 1.7, 0.3)
    g = synth.hash_uniform(0x22, (Cn,), 0.9, 1.1).to(torch.bfloat16)
    b = _rand(0x23, (Cn,), 0.1)
    ref = F.group_norm(x, 32, g, b, eps=1e-6)
    _, mean, rstd = torch.native_group_norm(x, g.float(), b.float(), B, Cn, H * H, 32, 1e-6)        # mixed-type call: ATen's fp32 statistics
    mine, st = VX.group_norm(_nhwc_bits(x), VX.bf16_bits(g), VX.bf16_bits(b), want_stats=True)
    assert np.array_equal(st[..., 0].view(np.uint32), mean.numpy().view(np.uint32)) and np.array_equal(st[..., 1].view(np.uint32), rstd.numpy().view(np.uint32))
    assert int((mine != _nhwc_bits(ref)).sum()) == 0
    act = VX.group_norm(_nhwc_bits(x), VX.bf16_bits(g), VX.bf16_bits(b), silu=VX.silu_table())
    assert int((act != _nhwc_bits(F.silu(ref))).sum()) == 0


# ---- other image sizes (round 5): oneDNN's order is a function of the layer's spatial size; ATen's GroupNorm / flash kernel at the new shapes ----
SHAPES_RES = [("128 px down128 (128 wide: order 3)", 128, 128, 128, 3, 2), ("128 px down256 (64 wide: order 0)", 256, 256, 64, 3, 2),
              ("width 100: order 0", 128, 128, 100, 3, 2), ("width 102: order 3", 128, 128, 102, 3, 2), ("128 px 512@16", 512, 512, 16, 3, 1),
              ("320 px down512 (80 wide: order 0)", 512, 512, 80, 3, 2), ("320 px 512@40", 512, 512, 40, 3, 1), ("320 px conv_out", 512, 32, 40, 3, 1)]


@needs_amx
@pytest.mark.parametrize("name,cin,cout,H,k,stride", SHAPES_RES, ids=[s[0] for s in SHAPES_RES])
def test_conv_order_equals_onednn_other_sizes(name, cin, cout, H, k, stride):
    test_conv_order_equals_onednn(name, cin, cout, H, k, stride)


@needs_amx
@pytest.mark.parametrize("B,Cn,H", [(1, 256, 80), (2, 512, 40), (2, 512, 16), (1, 128, 160)])
def test_groupnorm_equals_aten_other_sizes(B, Cn, H):
    test_groupnorm_equals_aten(B, Cn, H)


@needs_amx
@pytest.mark.parametrize("T", [256, 1600])
def test_attention_equals_aten_flash_kernel_other_token_counts(T):
    q, k, v = _rand(0x31 + T, (1, 1, T, 512), 1.5), _rand(0x32 + T, (1, 1, T, 512), 1.5), _rand(0x33 + T, (1, 1, T, 512))
    ref = F.scaled_dot_product_attention(q, k, v)
    mine = VX.attention(VX.bf16_bits(q[:, 0]), VX.bf16_bits(k[:, 0]), VX.bf16_bits(v[:, 0]))
    assert int((mine != VX.bf16_bits(ref[:, 0])).sum()) == 0


def test_silu_table_is_torch_cpu_silu_and_the_flush_rule():
    tab = VX.silu_table()
    allb = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(torch.bfloat16)
    assert np.array_equal(VX.bf16_bits(F.silu(allb.clone())), tab)
    # the rule the GPU table builder and the CPU twin implement: correctly rounded, except -0 where fp32 exp(-x) overflows
    twin = A.bind(os.path.join(ROOT, "oracle", "libselftok_cpu.so"))
    built = np.zeros(65536, dtype=np.uint16)
    assert twin.selftok_vx_silu_table_bf16(built.ctypes.data, None) == 0
    fin = np.isfinite((np.arange(65536, dtype=np.uint32) << 16).view(np.float32))
    assert np.array_equal(built[fin], tab[fin])


@needs_amx
def test_attention_equals_aten_flash_kernel():
    B, T, Cn = 1, 1024, 512
    q, k, v = _rand(0x31, (B, 1, T, Cn), 1.5), _rand(0x32, (B, 1, T, Cn), 1.5), _rand(0x33, (B, 1, T, Cn))
    ref = F.scaled_dot_product_attention(q, k, v)
    mine = VX.attention(VX.bf16_bits(q[:, 0]), VX.bf16_bits(k[:, 0]), VX.bf16_bits(v[:, 0]))
    assert int((mine != VX.bf16_bits(ref[:, 0])).sum()) == 0


def test_expf_restatement_equals_libm():
    libm = C.CDLL("libm.so.6")
    libm.expf.restype, libm.expf.argtypes = C.c_float, [C.c_float]
    g = torch.Generator().manual_seed(5)
    xs = torch.cat([-torch.rand(20000, generator=g) * 40.0, torch.rand(2000, generator=g) * 30.0, torch.tensor([0.0, -87.3, -100.0, -103.97, -104.1, 88.7, 89.0])]).float()
    for v in xs.tolist():
        a, b = np.float32(VX.expf(v)), np.float32(libm.expf(v))
        assert a.view(np.uint32) == b.view(np.uint32), (v, a, b)


def _encoder_latents(n, first=0):
    vsd = W.synthetic_vae_state_dict()
    pw = VX.pack_weights(vsd)
    img = synth.synthetic_images(n, first_index=first).to(torch.bfloat16)
    mom = VX.encode_moments(pw, VX.bf16_bits(img.permute(0, 2, 3, 1)))
    mean = VX.bits_to_torch(mom[..., :16]).permute(0, 3, 1, 2).contiguous()
    from oracle import model as OM
    return OM.process_in(mean)


def test_encoder_equals_the_reference_pipeline_run_image_0():
    """image 0 of the 16-image run of the REFERENCE pipeline (mimogpt.infer.SelftokPipeline on CPU): latents bit for bit.  Runs on any host
    (it compares against a committed vector, no torch arithmetic involved besides two element-wise bf16 ops)."""
    g = np.load(os.path.join(GOLD, "pipeline_b16.npz"))
    x0 = _encoder_latents(1)
    ref = torch.from_numpy(g["x0_bf16"][:1]).view(torch.bfloat16)
    assert int((x0.view(torch.int16) != ref.view(torch.int16)).sum()) == 0


@pytest.mark.slow
def test_encoder_equals_the_reference_pipeline_run_image_15():
    """one more of the 16 (all 16 were checked when the oracle was written: 0 of 262144 elements differ; the
    GPU suite checks all 16 through the HIP kernels); ~10 s per image on 8 cores"""
    g = np.load(os.path.join(GOLD, "pipeline_b16.npz"))
    x0 = _encoder_latents(1, first=15)
    ref = torch.from_numpy(g["x0_bf16"][15:16]).view(torch.bfloat16)
    assert int((x0.view(torch.int16) != ref.view(torch.int16)).sum()) == 0


def test_cpu_twin_exports_the_exact_entries_with_the_declared_signatures():
    """the C ABI entries of include/selftok_hip.h for the exact-order encoder, called through the same ctypes signatures the product
    uses (selftoktokenizer_amd/_lib.SIGNATURES), on the CPU twin"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    twin = A.bind(os.path.join(ROOT, "oracle", "libselftok_cpu.so"))
    x = VX.bf16_bits(_rand(1, (1, 32, 32, 32)))
    w = VX.bf16_bits(_rand(2, (32, 3, 3, 32), 0.1))
    b = VX.bf16_bits(_rand(3, (32,), 0.1))
    out = np.zeros((1, 32, 32, 32), dtype=np.uint16)
    assert twin.selftok_vx_conv2d_bf16(x.ctypes.data, w.ctypes.data, b.ctypes.data, None, out.ctypes.data, 1, 32, 32, 32, 32, 32, 3, 1, 0, None) == 0
    assert np.array_equal(out, VX.conv2d(x, w, b, order=0))
    assert twin.selftok_vx_conv2d_bf16(x.ctypes.data, w.ctypes.data, b.ctypes.data, None, out.ctypes.data, 1, 32, 32, 32, 32, 32, 5, 1, 0, None) == -1
    x128 = VX.bf16_bits(_rand(4, (1, 32, 32, 128)))
    g128, b128 = VX.bf16_bits(_rand(5, (128,), 0.1, 1.0)), VX.bf16_bits(_rand(6, (128,), 0.1))
    y = np.zeros_like(x128)
    st = np.zeros((1, 32, 2), dtype=np.float32)
    nbytes = twin.selftok_vx_groupnorm_workspace_bytes(1, 1024, 128)
    assert nbytes == 128 * 8 * 8 + 2 * 128 * 4
    ws = np.zeros(nbytes, dtype=np.uint8)
    assert twin.selftok_vx_groupnorm_bf16(x128.ctypes.data, g128.ctypes.data, b128.ctypes.data, y.ctypes.data, ws.ctypes.data, None, st.ctypes.data, 1, 1024, 128, 32, 1e-6, None) == 0
    ref, st_ref = VX.group_norm(x128, g128, b128, want_stats=True)
    assert np.array_equal(y, ref) and np.array_equal(st.view(np.uint32), st_ref.view(np.uint32))
    xs = np.array([-1.5, 0.0, 3.25], dtype=np.float32)
    ys = np.zeros_like(xs)
    assert twin.selftok_vx_expf_f32(xs.ctypes.data, ys.ctypes.data, 3, None) == 0
    assert np.array_equal(ys.view(np.uint32), np.array([VX.expf(float(v)) for v in xs], dtype=np.float32).view(np.uint32))
    # scores fp32 + probabilities bf16 + V transposed + (kv blocks + 1) x rows of rescale / row scale + 128 floats of slack behind them
    assert twin.selftok_vx_attention_workspace_bytes(2, 1024, 512) == 2 * 1024 * 1024 * 6 + 2 * 1024 * 512 * 2 + 3 * 2 * 1024 * 4 + 512


# ---- round 5: the decoder ---------------------------------------------------------------------------------------------------------------
DEC_SHAPES = [("conv_in 16->512 @32", 16, 512, 32, 3, False), ("up 512->512 @16->32 (the Upsample layers at a quarter of their smallest size: the real sizes are in the profile)", 512, 512, 16, 3, True), ("shortcut 512->256 1x1 @128", 512, 256, 128, 1, False),
              ("conv_out 128->3 @256", 128, 3, 256, 3, False)]


@needs_amx
@pytest.mark.parametrize("name,cin,cout,H,k,up", DEC_SHAPES, ids=[s[0] for s in DEC_SHAPES])
def test_decoder_conv_order_equals_onednn(name, cin, cout, H, k, up):
    """the decoder's own layer shapes (the big 3x3 ones: tools/probe_cpu_bf16/check_decoder_convs.py, profiles/r5_cpu_bf16_decoder_orders.txt):
    chunks in (kh, kw, channel-block) order everywhere, also behind a nearest-2x upsample, with 16 input channels, with 3 output channels"""
    x = _rand(0x70 + cin + H, (1, cin, H, H), 1.2, 0.05)
    w = _rand(0x71 + cout, (cout, cin, k, k), (1.0 / (cin * k * k)) ** 0.5)
    b = _rand(0x72, (cout,), 0.1)
    with torch.no_grad():
        xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
        ref = F.conv2d(xin, w, b, padding=k // 2)
    xb = _nhwc_bits(x)
    mine = VX.conv2d(VX.upsample2x(xb) if up else xb, VX.bf16_bits(w.permute(0, 2, 3, 1)), VX.bf16_bits(b), pad=k // 2, order=0)
    assert int((mine != _nhwc_bits(ref)).sum()) == 0


def test_whole_decoder_equals_the_reference_pixels():
    """oracle/vae_exact.py decode against the REFERENCE's decoder output (tests/golden/vae_b1.npz, the in-repo mirror the pipeline run executes):
    0 of 196608 bf16 pixels differ.  Host independent (plain C)."""
    vsd = W.synthetic_vae_state_dict()
    g = np.load(os.path.join(GOLD, "vae_b1.npz"))
    z = synth.synthetic_latents(1).to(torch.bfloat16)
    px = VX.decode(VX.pack_weights(vsd), VX.bf16_bits(z.permute(0, 2, 3, 1)))
    ref = VX.bf16_bits(torch.from_numpy(g["rec"]).to(torch.bfloat16).permute(0, 2, 3, 1))
    assert int((px != ref).sum()) == 0
