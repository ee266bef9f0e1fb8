"""-m gpu: csrc/gemm_fp32.hip (`selftok_linear_f32`, round 6) -- the LDS-DMA staged fp32-MFMA Linear.
MKL order: bit-equal to the CPU oracle (oracle/encoder_exact.c: torch-CPU's own F.linear bits) and to the round-5 kernel (`ex_linear(kernel='xe')`) at the MMDiT's full
shapes, with every epilogue, ragged row counts, and the tail round both planned and forced.  Free order: error against fp64 not above the fp32 library GEMM's."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import encoder_exact as EX
from selftoktokenizer_amd import ops, synth

pytestmark = pytest.mark.gpu

SHAPES = [("qkv", 4608, 1536), ("proj", 1536, 1536), ("fc1", 6144, 1536), ("fc2", 1536, 6144)]       # (N, K) of the joint blocks' Linears (sd3/mmdit.py:266-307, 413-419)


def _rand(seed, shape, scale=1.0, shift=0.0):
    return (synth.hash_normalish(seed, shape) * scale + shift).float().contiguous()


def _bits_equal(a: torch.Tensor, b: torch.Tensor, what: str):
    bad = a.view(torch.int32) != b.view(torch.int32)
    bad &= ~((a == 0) & (b == 0))
    n = int(bad.sum())
    assert n == 0, f"{what}: {n} of {a.numel()} fp32 elements differ"


@pytest.mark.parametrize("name,N,K", [("qkv", 4608, 1536), ("fc2", 1536, 6144), ("q_mlp.fc2 of the Q-Former (5 1/3 K-blocks)", 512, 2048), ("t_embedder (one K-block)", 512, 256)])
def test_mkl_order_equals_the_cpu_oracle(name, N, K):
    """against torch-CPU's own bits (oracle/encoder_exact.c), 300 rows: a ragged single row tile, the K-block fold incl. a short last block"""
    M = 300
    x = _rand(0x51 + K, (M, K), 1.2, 0.05)
    w = _rand(0x52 + N, (N, K), (1.0 / K) ** 0.5)
    b = _rand(0x53, (N,), 0.2)
    ref = torch.from_numpy(EX.linear(x.numpy(), w.numpy(), b.numpy()))
    nblk = (K // 32 + 11) // 12
    for split in (0, nblk):
        out = ops.linear_f32(x.cuda(), w.cuda(), b.cuda(), mkl_order=True, split=split).cpu()
        _bits_equal(out, ref, f"{name}, forced split {split}")


@pytest.mark.parametrize("name,N,K", SHAPES, ids=[s[0] for s in SHAPES])
def test_mkl_order_equals_the_round5_kernel_at_full_size(name, N, K):
    """22912 = 64 x 358 context rows (a ragged last row tile and a tail round) and 16384 = 64 x 256 image rows (whole rounds): `ex_linear(kernel='sg')` vs
    `kernel='xe'`, the epilogues the exact MMDiT uses (sd3/mmdit.py:485-496: x + gate * proj(attn) with bias last, per-token and per-sample gate tables; fc1 + GELU)"""
    dev = "cuda"
    for M, T in ((22912, 358), (16384, 256)):
        x = _rand(0x61 + M, (M, K), 1.1).to(dev)
        w = _rand(0x62 + N, (N, K), (1.0 / K) ** 0.5).to(dev)
        b = _rand(0x63, (N,), 0.2).to(dev)
        _bits_equal(ops.ex_linear(x, w, b, kernel="sg"), ops.ex_linear(x, w, b, kernel="xe"), f"{name} M={M} plain")
        if name == "fc1":
            _bits_equal(ops.ex_linear(x, w, b, gelu=True, kernel="sg"), ops.ex_linear(x, w, b, gelu=True, kernel="xe"), f"{name} M={M} GELU")
        if name in ("proj", "fc2"):
            res = _rand(0x64, (M, N)).to(dev)
            tab = _rand(0x65, (T, N), 0.7).to(dev)              # per-token table: row m % T
            per = _rand(0x66, (M // T, N), 0.7).to(dev)         # per-sample table: row m / T
            for kw in (dict(gate=tab, gate_mod=T), dict(gate=per, gate_mod=-T)):
                a = ops.ex_linear(x, w, b, res=res, bias_last=(name == "proj"), kernel="sg", **kw)
                c = ops.ex_linear(x, w, b, res=res, bias_last=(name == "proj"), kernel="xe", **kw)
                _bits_equal(a, c, f"{name} M={M} res + gate ({'per token' if kw['gate_mod'] > 0 else 'per sample'})")
            # in place (out aliases res), as the model's residual stream update could be issued
            r2 = res.clone()
            ops.ex_linear(x, w, b, res=r2, gate=tab, gate_mod=T, out=r2, kernel="sg")
            _bits_equal(r2, ops.ex_linear(x, w, b, res=res, gate=tab, gate_mod=T, kernel="xe"), f"{name} M={M} in place")


def test_auto_dispatch_and_refusals():
    x = _rand(0x71, (512, 1536)).cuda()
    w = _rand(0x72, (1536, 1536), 0.03).cuda()
    _bits_equal(ops.ex_linear(x, w), ops.ex_linear(x, w, kernel="xe"), "auto")              # 512 rows >= EX_LINEAR_SG_MIN_ROWS: the new kernel
    assert not ops.linear_f32_supported(192, 64) and not ops.linear_f32_supported(512, 512, mkl_order=True) and ops.linear_f32_supported(512, 512)
    from selftoktokenizer_amd._lib import SelftokHipError
    with pytest.raises(SelftokHipError):
        ops.linear_f32(x[:, :512].contiguous(), _rand(0x73, (512, 512)).cuda(), mkl_order=True)       # 384 < K < 768 in MKL order: two half blocks, ex_linear's kernel
    with pytest.raises(SelftokHipError):
        ops.linear_f32(x, _rand(0x74, (100, 1536)).cuda())                                              # N % 128
    assert ops.linear_f32(x[:0], w).shape == (0, 1536)


@pytest.mark.parametrize("name,N,K", SHAPES, ids=[s[0] for s in SHAPES])
def test_free_order_accuracy(name, N, K):
    """gemm='fp32' candidates: one k-ascending chain per output (tail tiles: S chains added in order).  Error vs fp64 <= the fp32 library GEMM's"""
    M = 3000
    x = _rand(0x81 + K, (M, K), 1.0).cuda()
    w = _rand(0x82 + N, (N, K), (1.0 / K) ** 0.5).cuda()
    b = _rand(0x83, (N,), 0.2).cuda()
    r64 = x.double() @ w.double().t() + b.double()
    e_lib = float((F.linear(x, w, b).double() - r64).pow(2).mean().sqrt())
    for split in (0, 2, 4, 8):
        e = float((ops.linear_f32(x, w, b, split=split).double() - r64).pow(2).mean().sqrt())
        print(f"{name}: free order, forced split {split}: rms error vs fp64 {e:.3e} (hipBLASLt {e_lib:.3e})")
        assert e <= 1.05 * e_lib + 1e-9
