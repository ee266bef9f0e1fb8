"""-m gpu: the f16x2-split Linear kernel (csrc/gemm_split.hip, through the C ABI) against an fp64 product.

Gate: the split GEMM may replace the fp32 library GEMM on the parity path only if it is at least
as accurate -- its error against the exact (fp64) product must not exceed the fp32 GEMM's own error on the same data.
"""
import pytest
import torch
import torch.nn.functional as F

from selftoktokenizer_amd import ops

pytestmark = pytest.mark.gpu


def _data(M, N, K, seed, act_scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn(M, K, device="cuda", generator=g) * (1.0 + 3.0 * torch.rand(1, K, device="cuda", generator=g)) * act_scale
    w = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1) * (3.0 / K) ** 0.5
    b = torch.randn(N, device="cuda", generator=g) * 0.1
    return a, w, b


def _errs(out, ref):
    e = out.double() - ref
    return float(e.abs().max()), float(e.pow(2).mean().sqrt())


@pytest.mark.parametrize("M,N,K", [(256, 128, 32), (1000, 1536, 1536), (2048 + 37, 4608, 1536), (777, 1536, 6144), (3, 256, 64)])
def test_split_linear_not_worse_than_fp32_gemm(M, N, K):
    a, w, b = _data(M, N, K, seed=M + N + K)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    packed = ops.linear_f16x2_pack(w, flag)
    out = ops.linear_f16x2(a, packed, b, N, overflow=flag)
    ref = a.double() @ w.double().t() + b.double()
    lib = F.linear(a, w, b)
    mx_s, rms_s = _errs(out, ref)
    mx_l, rms_l = _errs(lib, ref)
    print(f"M={M} N={N} K={K}: split max {mx_s:.3e} rms {rms_s:.3e} | fp32 library max {mx_l:.3e} rms {rms_l:.3e}")
    assert int(flag.item()) == 0
    assert rms_s <= rms_l * 1.05 + 1e-9 and mx_s <= mx_l * 2.0 + 1e-7
    # transpose-detecting: asymmetric data, exact row/col placement
    assert torch.allclose(out, lib, rtol=0, atol=8 * mx_l + 1e-6)


def test_split_linear_gelu_strided_rows_no_bias():
    M, N, K = 600, 6144, 1536
    a, w, b = _data(M, N, K, seed=5)
    big = torch.zeros(M, 3 * K, device="cuda")
    big[:, K:2 * K] = a
    view = big[:, K:2 * K]                                 # row stride 3K: a slice of a fused buffer
    packed = ops.linear_f16x2_pack(w)
    out = ops.linear_f16x2(view, packed, b, N, gelu=True)
    ref = F.gelu((a.double() @ w.double().t() + b.double()), approximate="tanh")
    lib = F.gelu(F.linear(a, w, b), approximate="tanh")
    assert _errs(out, ref)[1] <= _errs(lib, ref)[1] * 1.05
    out_nb = ops.linear_f16x2(a.reshape(2, 300, K), packed, None, N)
    assert out_nb.shape == (2, 300, N)
    assert _errs(out_nb.reshape(M, N), a.double() @ w.double().t())[1] <= _errs(F.linear(a, w), a.double() @ w.double().t())[1] * 1.05


def test_split_linear_tiny_and_large_magnitudes():
    """values far below the fp16 normal range ride in the scaled low part; values beyond 65504 raise the flag"""
    M, N, K = 512, 256, 1536
    for scale in (1e-3, 1e2):
        a, w, b = _data(M, N, K, seed=9, act_scale=scale)
        packed = ops.linear_f16x2_pack(w)
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        out = ops.linear_f16x2(a, packed, None, N, overflow=flag)
        ref = a.double() @ w.double().t()
        lib = F.linear(a, w)
        rs, rl = _errs(out, ref)[1], _errs(lib, ref)[1]
        print(f"scale {scale:g}: split rms {rs:.3e} fp32 library rms {rl:.3e}")
        assert int(flag.item()) == 0 and rs <= rl * 1.05 + 1e-12 * scale
    # a tensor that is tiny as a whole (1e-6): still no blow-up -- the absolute error per operand is bounded by the scaled low
    # part's subnormal spacing (2^-24 / 2^11 = 2.9e-11), i.e. ~16 significant bits at this magnitude (documented window:
    # full 22-bit operands for |x| in [1.2e-4, 65504))
    a, w, b = _data(M, N, K, seed=9, act_scale=1e-6)
    out = ops.linear_f16x2(a, ops.linear_f16x2_pack(w), None, N)
    ref = a.double() @ w.double().t()
    assert _errs(out, ref)[0] < 1e-9 and float(ref.abs().max()) > 1e-6
    a, w, b = _data(M, N, K, seed=9)
    a[17, 300] = 7.0e4
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.linear_f16x2(a, ops.linear_f16x2_pack(w), None, N, overflow=flag)
    assert int(flag.item()) & 1
    w[3, 3] = 1.0e5
    flag.zero_()
    ops.linear_f16x2_pack(w, flag)
    assert int(flag.item()) & 2


def test_split_linear_bad_arguments():
    from selftoktokenizer_amd._lib import SelftokHipError
    with pytest.raises(SelftokHipError):
        ops.linear_f16x2_pack(torch.zeros(100, 64, device="cuda"))          # N % 128
    with pytest.raises(SelftokHipError):
        ops.linear_f16x2_pack(torch.zeros(128, 48, device="cuda"))          # K % 32
    packed = ops.linear_f16x2_pack(torch.zeros(128, 64, device="cuda"))
    assert ops.linear_f16x2(torch.zeros(0, 64, device="cuda"), packed, None, 128).shape == (0, 128)


@pytest.mark.parametrize("M,N,K", [(256, 128, 32), (1000, 1536, 1536), (2048 + 37, 4608, 1536), (777, 1536, 6144), (3, 256, 64), (300, 128, 96)])
def test_presplit_activations_bit_identical(M, N, K):
    """The producer-side split (fp16 hi/lo planes, staged by LDS-DMA) is the same function of the fp32 value as the in-kernel
    split: selftok_linear_f16x2_split must reproduce selftok_linear_f16x2_f32 bit for bit, for every k-tile count and ragged M."""
    a, w, b = _data(M, N, K, seed=7 * M + N + K)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    packed = ops.linear_f16x2_pack(w, flag)
    xs = ops.split_f16x2(a, flag)
    assert xs.shape == (M, K) and xs.dtype == torch.float16 and xs.data.shape == ((M + 15) // 16, K // 32, 2, 16, 32)
    assert float((ops.split_to_f32(xs) - a).abs().max()) <= float(a.abs().max()) * 2.0 ** -21
    ref = ops.linear_f16x2(a, packed, b, N)
    out = ops.linear_f16x2_split(xs, packed, b, N, overflow=flag)
    assert torch.equal(out, ref)
    ref_g = ops.linear_f16x2(a, packed, b, N, gelu=True)
    out_g = ops.linear_f16x2_split(xs, packed, b, N, gelu=True, overflow=flag)
    assert torch.equal(out_g, ref_g)
    # split outputs: exactly the split of the fp32 output
    os_ = ops.linear_f16x2_split(xs, packed, b, N, gelu=True, overflow=flag, out_split=True)
    assert os_.shape == (M, N) and torch.equal(os_.planes(), ops.split_f16x2(ref_g).planes())
    assert int(flag.item()) == 0


def test_presplit_overflow_flag():
    a, w, b = _data(300, 128, 64, seed=3)
    packed = ops.linear_f16x2_pack(w)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    a[17, 5] = 7.0e4
    ops.split_f16x2(a, flag)
    assert int(flag.item()) & 1
    flag.zero_()
    big = a.clone()
    big[17, 5] = 1.0
    xs = ops.split_f16x2(big * 3.0e3)                      # inputs in range, outputs of the Linear beyond the fp16 range
    ops.linear_f16x2_split(xs, ops.linear_f16x2_pack(w * 50.0), None, 128, overflow=flag, out_split=True)
    assert int(flag.item()) & 1


def test_gelu_epilogue_accuracy():
    """the epilogue's GELU (x * sigmoid(2u) on v_exp_f32 / v_rcp_f32) against the fp64 tanh-GELU over the whole fp32-relevant range,
    next to torch's own fp32 GELU: identity weights make the Linear a pass-through, so the output is GELU(x) itself."""
    K = N = 128
    x = torch.cat([torch.linspace(-12, 12, 256 * K - 16, device="cuda"), torch.tensor([0.0, -0.0, 1e-20, -1e-20, 1e-6, -1e-6, 30.0, -30.0, 88.0, -88.0,
                                                                                       1e4, -1e4, 6e4, -6e4, 3.0, -3.0], device="cuda")]).reshape(256, K)
    xs = ops.split_f16x2(x)
    x_eff = ops.split_to_f32(xs)                          # what the kernel multiplies (22 significand bits)
    packed = ops.linear_f16x2_pack(torch.eye(N, device="cuda"))
    out = ops.linear_f16x2_split(xs, packed, None, N, gelu=True)
    ref = F.gelu(x_eff.double(), approximate="tanh")
    lib = F.gelu(x_eff, approximate="tanh")
    err, err_lib = (out.double() - ref).abs(), (lib.double() - ref).abs()
    # GELU is ill-conditioned for negative x (a 1-ulp error of u moves the result by several ulps: torch fp32 shows 4.4e-7 relative at
    # x = -2.4, this kernel 6.5e-7): 8 fp32 ulps of the value, plus the argument rounding of the exponential in the deep negative tail
    tol = 1.0e-6 * ref.abs() + 1e-9 * x_eff.abs().double().clamp(min=1.0)
    print(f"GELU epilogue: max |err| {float(err.max()):.3e} (torch fp32 GELU {float(err_lib.max()):.3e}); max err/tol {float((err / tol).max()):.2f} "
          f"(torch {float((err_lib / tol).max()):.2f})")
    worst = int((err / tol).argmax())
    assert bool((err <= tol).all()), (float(x_eff.flatten()[worst]), float(out.flatten()[worst]), float(ref.flatten()[worst]), float(lib.flatten()[worst]))
    assert float(err[x_eff.abs() <= 12].max()) <= float(err_lib[x_eff.abs() <= 12].max()) * 1.5 + 1e-9
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("B,T,N,K", [(2, 300, 1536, 1536), (3, 77, 256, 96), (1, 515, 1536, 6144)])   # N: a hidden size residual_ln_mod supports
def test_residual_epilogue_bit_identical(B, T, N, K):
    """out = resid + gate * Linear(x) in the Linear's epilogue == residual_ln_mod on the stored Linear output (per-token gate,
    per-sample gate, no gate; ragged row blocks)."""
    a, w, b = _data(B * T, N, K, seed=B + T + N + K)
    packed = ops.linear_f16x2_pack(w)
    xs = ops.split_f16x2(a.reshape(B, T, K))
    g = torch.Generator(device="cuda").manual_seed(1)
    resid = torch.randn(B, T, N, device="cuda", generator=g)
    tab_t = torch.randn(T, 3 * N, device="cuda", generator=g)
    tab_b = torch.randn(B, 3 * N, device="cuda", generator=g)
    y = ops.linear_f16x2_split(xs, packed, b, N)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    for gate, ps in ((tab_t[:, N:2 * N], False), (tab_b[:, 2 * N:], True), (None, False)):
        ref, _ = ops.residual_ln_mod(resid, y=y, gate=gate, gate_per_sample=ps, want_n=False)
        out = ops.linear_f16x2_split_residual(xs, packed, b, N, resid, gate=gate, gate_per_sample=ps, overflow=flag)
        assert torch.equal(out, ref)
    assert int(flag.item()) == 0


@pytest.mark.parametrize("M", [1, 256, 257, 513])
@pytest.mark.parametrize("N,K", [(4608, 1536), (1536, 1536), (6144, 1536), (1536, 6144)])
def test_split_k_small_m(M, N, K):
    """the small-M entry points (one image: 256 image rows, <= 513 context rows) at the model's four Linear shapes: `ksplit` work-groups
    per output tile + the reduction launch.  Deterministic (two runs bit-equal), every epilogue (plain, GELU + split output, residual
    with gate) a function of the SAME reduced sum, within fp32 rounding of the single-pass kernel, and not less accurate against fp64."""
    ks = max(ops.f16x2_ksplit(M, N, K), 2)                   # where the heuristic keeps the single-pass kernel, exercise 2 anyway
    assert (K // 32) % ks == 0 and ops.f16x2_ksplit(2048, N, K) == 1
    a, w, b = _data(M, N, K, seed=11 * M + N + K)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    packed = ops.linear_f16x2_pack(w, flag)
    xs = ops.split_f16x2(a.reshape(1, M, K), flag)
    one = ops.linear_f16x2_split(xs, packed, b, N, overflow=flag)
    out = ops.linear_f16x2_split(xs, packed, b, N, overflow=flag, ksplit=ks)
    again = ops.linear_f16x2_split(xs, packed, b, N, overflow=flag, ksplit=ks)
    assert torch.equal(out, again)
    ref = (a.double() @ w.double().T + b.double())
    e_one, e_k = float((one[0].double() - ref).abs().max()), float((out[0].double() - ref).abs().max())
    d = float((out - one).abs().max())
    scale = float(ref.abs().max())
    print(f"M={M} N={N} K={K} ksplit={ks}: |split-K - single pass| max {d:.3e} (outputs up to {scale:.2f}); vs fp64: single pass {e_one:.3e}, split-K {e_k:.3e}")
    assert d <= 1.5e-6 * scale and e_k <= 1.5 * e_one + 1e-7 * scale
    # GELU + split output == the split of the GELU'd fp32 output of the same launch pair
    g32 = ops.linear_f16x2_split(xs, packed, b, N, gelu=True, overflow=flag, ksplit=ks)
    gsp = ops.linear_f16x2_split(xs, packed, b, N, gelu=True, overflow=flag, ksplit=ks, out_split=True)
    assert torch.equal(gsp.planes(), ops.split_f16x2(g32).planes())
    assert float((g32 - ops.linear_f16x2_split(xs, packed, b, N, gelu=True)).abs().max()) <= 1.5e-6 * scale
    # no bias
    nb = ops.linear_f16x2_split(xs, packed, None, N, ksplit=ks)
    assert float((nb + b - out).abs().max()) <= 1e-6 * scale
    if N == 1536:                                           # residual epilogue (a hidden size residual_ln_mod supports)
        g = torch.Generator(device="cuda").manual_seed(2)
        resid = torch.randn(1, M, N, device="cuda", generator=g)
        tab = torch.randn(M, 2 * N, device="cuda", generator=g)
        for gate in (tab[:, N:], None):
            want, _ = ops.residual_ln_mod(resid, y=out, gate=gate, want_n=False)
            got = ops.linear_f16x2_split_residual(xs, packed, b, N, resid, gate=gate, overflow=flag, ksplit=ks)
            assert torch.equal(got, want)
    assert int(flag.item()) == 0
    from selftoktokenizer_amd._lib import SelftokHipError
    with pytest.raises(SelftokHipError):
        ops.linear_f16x2_split(xs, packed, b, N, ksplit=7)                  # not a divisor of K / 32


def test_split_k_overflow_flag_and_empty():
    a, w, b = _data(40, 128, 192, seed=4)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    xs = ops.split_f16x2(a * 3.0e3)
    ops.linear_f16x2_split(xs, ops.linear_f16x2_pack(w * 50.0), None, 128, overflow=flag, out_split=True, ksplit=3)
    assert int(flag.item()) & 1
    packed = ops.linear_f16x2_pack(w)
    assert ops.linear_f16x2_split(ops.split_f16x2(a[:0]), packed, b, 128, ksplit=2).shape == (0, 128)


def test_gemm_tune_installs_measured_kernels_and_keeps_the_numbers(capsys):
    """gemm_tune.autotune_linears: every family gets a report entry; whatever kernel it installs for a row count of the step, the Linear's
    result stays an fp32 GEMM of the same accuracy against fp64 as the default kernel's (a different summation order, nothing else).
    Hygiene: torch.cuda.tunable's flags are the caller's again afterwards; candidates of another
    hipBLASLt build are not touched; a candidate name the library does not know is dropped, not raised."""
    from selftoktokenizer_amd import gemm_tune as G
    tun = torch.cuda.tunable
    flags0 = (tun.is_enabled(), tun.tuning_is_enabled())
    rows, reps = G.step_row_counts(8, [511, 300, 77], 256)
    M = reps[1]
    a = torch.randn(M, 1536, device="cuda")
    w, b = torch.randn(4608, 1536, device="cuda") * 0.03, torch.randn(4608, device="cuda")
    before = torch.nn.functional.linear(a, w, b)
    # another library build: no-op, said loudly, nothing enabled
    assert G.autotune_linears([3, 5], torch.device("cuda"), found_with={"HIPBLASLT_VERSION": "some-other-build"}) is None
    assert "not tuning" in capsys.readouterr().out and (tun.is_enabled(), tun.tuning_is_enabled()) == flags0
    ok, why = G.validators_match()
    print("validators:", dict(tun.get_validators()), "match:", ok, why)
    if not ok:                                      # a box with another hipBLASLt than the candidates' build: the no-op path is the contract
        assert G.autotune_linears(rows, torch.device("cuda"), reps=reps) is None
        return
    # a solution name this library does not have is dropped (TunableOp raises inside the probing F.linear), the rest is measured
    rep_bad = G.autotune_linears([M + 64], torch.device("cuda"), families=((4608, 1536),), candidates=("Gemm_Hipblaslt_999999999",) + G.CANDIDATES[:2])
    assert rep_bad is not None and rep_bad[(4608, 1536)][0] in (None,) + G.CANDIDATES[:2]
    rep = G.autotune_linears(rows, torch.device("cuda"), reps=reps)
    assert rep is not None and set(rep) == set(G.FAMILIES)
    for fam, (best, t0, t1) in rep.items():
        assert best is None or (best in G.CANDIDATES and t1 <= t0), (fam, best, t0, t1)
    print({f"{n}x{k}": (b_ or "default", t0, t1) for (n, k), (b_, t0, t1) in rep.items()})
    assert (tun.is_enabled(), tun.tuning_is_enabled()) == flags0, "gemm_tune left torch.cuda.tunable's global flags changed"
    with G.enabled() as on:
        assert on and tun.is_enabled() and not tun.tuning_is_enabled()
        after = torch.nn.functional.linear(a, w, b)
    assert (tun.is_enabled(), tun.tuning_is_enabled()) == flags0
    ref = (a.double() @ w.double().t() + b.double())
    e0, e1 = float((before.double() - ref).abs().max()), float((after.double() - ref).abs().max())
    print(f"max abs err vs fp64: default kernel {e0:.2e}, installed kernel {e1:.2e}")
    assert e1 <= 2.0 * e0 + 1e-6
    assert G.autotune_linears(rows, torch.device("cuda"), reps=reps) is rep          # cached per (device, row counts)


@pytest.mark.gpu
def test_gemm_tune_leaves_small_row_counts_on_the_default_where_the_winner_loses():
    """round 6: the family winner is chosen at 16384 rows and the median context length; the last steps of a decode issue Linears of ~1300 rows, where such a kernel lost 3x
    to hipBLASLt's own small-M choice (profiles/r6_sweep_fp32_linear_vs_sg.txt: proj 175 vs 60 us).  Three probes at the low end of the row list now keep those rows on
    the default: inside an enabled() block the smallest row count of the step must not run much slower than outside"""
    from selftoktokenizer_amd import gemm_tune as G
    from selftoktokenizer_amd.config import default_config
    from selftoktokenizer_amd.schedule import DiTiCont
    if not G.validators_match()[0]:
        pytest.skip("another hipBLASLt build than the candidates': autotune is a no-op")
    p = default_config(512).tokenizer.params
    import numpy as np
    k_table = DiTiCont(1000, 512, p.stages, p.k_per_stage).to_indices(np.linspace(999, 0, 50).astype(np.int64))
    rows, reps = G.step_row_counts(64, k_table, 256)
    rep = G.autotune_linears(rows, torch.device("cuda"), reps=reps)
    assert rep is not None and set(G.LAST_CUTS) >= set(G.FAMILIES)
    M = rows[0]
    for (N, K), (best, _, _) in rep.items():
        assert 0 <= G.LAST_CUTS[(N, K)] < reps[0]
        a, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * 0.03, torch.randn(N, device="cuda")
        t_out = min(G._time(lambda: torch.nn.functional.linear(a, w, b)) for _ in range(3))
        with G.enabled():
            t_in = min(G._time(lambda: torch.nn.functional.linear(a, w, b)) for _ in range(3))
        print(f"({N}, {K}) winner {best or 'default'}, rows <= {G.LAST_CUTS[(N, K)]} keep the default; {M} rows: {1e3 * t_out:.1f} us default, {1e3 * t_in:.1f} us inside enabled()")
        assert t_in <= 2.0 * t_out + 0.02, ((N, K), best, t_in, t_out)          # the pathology was 3x (175 vs 60 us); generous against timing noise
