"""Thin torch-tensor front end of the C ABI (include/selftok_hip.h).

Every function takes CUDA(=HIP) tensors, passes `data_ptr()`s and the current torch stream to
libselftok_hip.so, and returns torch tensors.  PyTorch is only the allocator / stream owner here.
"""
from __future__ import annotations

import functools
from typing import Optional

import torch

from . import _lib

IDS_I32 = 1
PRENORMED = 2
VQ_F16COARSE = 8     # packed path: f16 coarse pass + exact fp32 re-score (same ids / score bits, ~4x faster than the fp32-MFMA kernel)
VQ_F16COARSE1 = 16   # with VQ_F16COARSE: ONE MFMA per 32 x 32 scores (hi x hi) and a wider exact re-score window (same ids / score bits)
VQ_DEFAULT_COARSE = True
VQ_COARSE_MFMAS = 1  # MFMAs per 32 x 32 coarse scores when `coarse` is True / None: 1 (hi x hi, window 17 x 2^-14; round 4: 0.115 -> 0.079 ms at
                     # N = 32768, profiles/r4_vq_bench_1mfma.txt) or 3 (hi*hi + hi*lo + lo*hi, window 2^-17)


def _coarse_flags(coarse) -> int:
    """coarse: None -> the defaults, False -> fp32-input MFMA kernel, True -> f16 coarse pass with VQ_COARSE_MFMAS, 3 / 1 -> that variant"""
    if coarse is None:
        coarse = VQ_DEFAULT_COARSE
    if coarse is False or coarse == 0:
        return 0
    n = VQ_COARSE_MFMAS if coarse is True else int(coarse)
    if n not in (1, 3):
        raise ValueError(f"coarse = {coarse!r}: expected None, False, True, 1 or 3")
    return VQ_F16COARSE | (VQ_F16COARSE1 if n == 1 else 0)
VQ_EVENTS = None     # bench.py sets this to a list: vq_encode(packed=True) then appends (start, after main kernel, after finalize) HIP events


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts):
    """every op launches on the CURRENT device's current stream: tensors on another GPU are refused (a kernel launched on
    GPU 0's stream against GPU 1 pointers would fault or race) -- SelftokPipeline's entry points select their device themselves
    (`with torch.cuda.device(...)`); direct callers of ops.* must do the same."""
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.SelftokHipError("selftok HIP ops need device tensors (there is no CPU fallback)")
        if cur is None:
            cur = torch.cuda.current_device()            # once per op, not once per tensor (host-bound small-batch decode loop)
        if t.device.index != cur:
            raise _lib.SelftokHipError(f"tensor on {t.device} but the current device is cuda:{cur}: "
                                       "call torch.cuda.set_device / use `with torch.cuda.device(...)` first")


def _p(t):
    return None if t is None else t.data_ptr()


def vq_pack_codebook(codebook: torch.Tensor) -> torch.Tensor:
    """[C,16] fp32 -> MFMA-fragment-ordered copy (one-time, the codebook is a constant)."""
    _need_cuda(codebook)
    cb = codebook.contiguous().float()
    C, D = cb.shape
    lib = _lib.load()
    packed = torch.zeros(lib.selftok_vq_packed_bytes(C, D) // 4, dtype=torch.float32, device=cb.device)
    _lib.check(lib.selftok_vq_pack_codebook(_p(cb), _p(packed), C, D, _stream()), "selftok_vq_pack_codebook")
    return packed


def packed_codes(packed: torch.Tensor, D: int = 16) -> int:
    """number of codes in a vq_pack_codebook image: C*D fp32 + 64 metadata floats + C*D fp16 hi/lo pairs (= C*D more floats)"""
    return (packed.numel() - 64) // (2 * D)


def vq_encode(z: torch.Tensor, codebook: torch.Tensor, *, packed: bool = False, return_best: bool = False,
              ids_dtype=torch.int64, prenormed: bool = False, rt: int = 0, split: int = 0, coarse=None):
    """z [...,16] fp32 (pre-norm) , codebook [C,16] (raw, or packed if packed=True) -> ids [...].
    rt / split: launch-shape overrides of the packed path (SELFTOK_VQ_RT / SELFTOK_VQ_SPLIT; 0 = automatic).
    coarse (packed path): True = f16 coarse pass + exact re-score (SELFTOK_VQ_F16COARSE) with VQ_COARSE_MFMAS MFMAs per 32 x 32 scores,
    3 / 1 = that variant explicitly (1 = SELFTOK_VQ_F16COARSE1), False = fp32-input MFMA kernel; None = VQ_DEFAULT_COARSE.  All return
    identical ids and score bits."""
    _need_cuda(z, codebook)
    lib = _lib.load()
    cflags = _coarse_flags(coarse) if packed else 0
    if packed and VQ_EVENTS is not None and not return_best and not prenormed and z.numel() > 0:
        # same two launches as selftok_vq_encode_packed_f32, with HIP events around the argmax kernel on its launch stream
        ids, launch_main, launch_fin = vq_encode_split_launch(z, codebook, ids_dtype, _flags=cflags)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        launch_main()
        e1.record()
        launch_fin()
        e2.record()
        VQ_EVENTS.append((e0, e1, e2))
        return ids
    zz = z.contiguous().float().reshape(-1, z.shape[-1])
    N, D = zz.shape
    C = packed_codes(codebook, D) if packed else codebook.shape[0]
    ids = torch.empty(N, dtype=ids_dtype, device=z.device)
    best = torch.empty(N, dtype=torch.float32, device=z.device) if return_best else None
    ws = torch.empty(lib.selftok_vq_workspace_bytes(N, C), dtype=torch.uint8, device=z.device)
    flags = ((IDS_I32 if ids_dtype == torch.int32 else 0) | (PRENORMED if prenormed else 0) | ((rt & 0xF) << 8) | ((split & 0xFF) << 16)
             | cflags)
    fn = lib.selftok_vq_encode_packed_f32 if packed else lib.selftok_vq_encode_f32
    _lib.check(fn(_p(zz), _p(codebook), _p(ids), _p(best), _p(ws), N, C, D, flags, _stream()),
               "selftok_vq_encode_packed_f32" if packed else "selftok_vq_encode_f32")
    ids = ids.reshape(z.shape[:-1])
    if return_best:
        return ids, best.reshape(z.shape[:-1])
    return ids


def vq_ema_accumulate(z: torch.Tensor, ids: torch.Tensor, C: int, prenormed: bool = False):
    """(bins [C], embed_sum [C,16]) of one batch: bins[c] = #rows with id c, embed_sum[c] = sum of their l2-normalised features
    (the reference's one-hot contractions, vector_quantize_pytorch.py:587-593)."""
    _need_cuda(z, ids)
    zz = z.contiguous().float().reshape(-1, z.shape[-1])
    flat = ids.contiguous().reshape(-1)
    if flat.dtype not in (torch.int32, torch.int64):
        flat = flat.to(torch.int64)
    assert flat.numel() == zz.shape[0]
    bins = torch.zeros(C, dtype=torch.float32, device=z.device)
    esum = torch.zeros(C, zz.shape[1], dtype=torch.float32, device=z.device)
    flags = (IDS_I32 if flat.dtype == torch.int32 else 0) | (PRENORMED if prenormed else 0)
    _lib.check(_lib.load().selftok_vq_ema_accumulate_f32(_p(zz), _p(flat), _p(bins), _p(esum), zz.shape[0], C, zz.shape[1], flags, _stream()),
               "selftok_vq_ema_accumulate_f32")
    return bins, esum


def vq_softmax_stats(z: torch.Tensor, codebook: torch.Tensor, scale: float = 10.0, colmean: bool = True, prenormed: bool = False):
    """The two reductions of p = softmax_c(scale * <l2norm(z[b,k]), codebook[c]>) that the reference's entropy regularisers read from
    its materialised [B, K, C] tensor (calc_entropy / calc_ema_entropy, vector_quantize_pytorch.py:89-118): z [B,K,16] ->
    (rowstats [B*K, 2] = (1 / sum_c exp, H(p[b,k,:])), colmean [K, C] = mean_b p[b,k,:] or None)."""
    _need_cuda(z, codebook)
    assert z.dim() == 3 and codebook.dim() == 2 and z.shape[-1] == codebook.shape[1]
    zz, cb = z.contiguous().float(), codebook.contiguous().float()
    B, K, Dm = zz.shape
    C = cb.shape[0]
    lib = _lib.load()
    rows = torch.empty(B * K, 2, dtype=torch.float32, device=z.device)
    cm = torch.empty(K, C, dtype=torch.float32, device=z.device) if colmean else None
    ws = torch.empty(lib.selftok_vq_softmax_workspace_bytes(B * K), dtype=torch.uint8, device=z.device)
    _lib.check(lib.selftok_vq_softmax_stats_f32(_p(zz), _p(cb), _p(rows), _p(cm), _p(ws), B, K, C, Dm, float(scale), PRENORMED if prenormed else 0, _stream()),
               "selftok_vq_softmax_stats_f32")
    return rows, cm


def vq_softmax_backward(z: torch.Tensor, codebook: torch.Tensor, rowstats: torch.Tensor, g_colmean: torch.Tensor, scale: float = 10.0,
                        prenormed: bool = False) -> torch.Tensor:
    """dF/dz [B,K,16] for a scalar F of vq_softmax_stats' colmean, given g_colmean = dF/d(colmean) [K, C] (code book detached)."""
    _need_cuda(z, codebook, rowstats, g_colmean)
    zz, cb, g = z.contiguous().float(), codebook.contiguous().float(), g_colmean.contiguous().float()
    B, K, Dm = zz.shape
    C = cb.shape[0]
    assert tuple(g.shape) == (K, C) and tuple(rowstats.shape) == (B * K, 2) and rowstats.is_contiguous() and rowstats.dtype == torch.float32
    lib = _lib.load()
    grad = torch.empty_like(zz)
    ws = torch.empty(lib.selftok_vq_softmax_workspace_bytes(B * K), dtype=torch.uint8, device=z.device)
    _lib.check(lib.selftok_vq_softmax_backward_f32(_p(zz), _p(cb), _p(rowstats), _p(g), _p(grad), _p(ws), B, K, C, Dm, float(scale),
                                                   PRENORMED if prenormed else 0, _stream()), "selftok_vq_softmax_backward_f32")
    return grad


def vq_tpc_update_(tpc: torch.Tensor, ids: torch.Tensor, weight: float) -> torch.Tensor:
    """in place: tpc [K,C] <- lerp(tpc, mean over samples of one_hot(ids [B,K]), weight)  (vector_quantize_pytorch.py:568-578)"""
    _need_cuda(tpc, ids)
    assert tpc.is_contiguous() and tpc.dtype == torch.float32 and ids.dim() == 2 and ids.shape[1] == tpc.shape[0]
    flat = ids.contiguous()
    if flat.dtype not in (torch.int32, torch.int64):
        flat = flat.to(torch.int64)
    K, C = tpc.shape
    _lib.check(_lib.load().selftok_vq_tpc_update_f32(_p(tpc), _p(flat), flat.shape[0], K, C, float(weight), IDS_I32 if flat.dtype == torch.int32 else 0,
                                                     _stream()), "selftok_vq_tpc_update_f32")
    return tpc


def code_gather_ln(ids: torch.Tensor, codebook: torch.Tensor, ln_w=None, ln_b=None, eps: float = 1e-6) -> torch.Tensor:
    """ids [...] (int64/int32) -> LayerNorm16(codebook[ids]) [...,16]"""
    _need_cuda(ids, codebook)
    if ids.is_floating_point() or ids.is_complex() or ids.dtype == torch.bool:
        raise TypeError(f"code_gather_ln: ids must be an integer tensor, got {ids.dtype}")
    if ids.dtype not in (torch.int32, torch.int64):
        ids = ids.to(torch.int64)                      # uint16 / int16 / uint8 wire formats: the kernel reads 4- or 8-byte ids only
    flat = ids.contiguous().reshape(-1)
    n = flat.numel()
    C, D = codebook.shape
    out = torch.empty(n, D, dtype=torch.float32, device=ids.device)
    flags = IDS_I32 if flat.dtype == torch.int32 else 0
    _lib.check(_lib.load().selftok_code_gather_ln_f32(_p(flat), _p(codebook), _p(ln_w), _p(ln_b), _p(out), n, C, D,
                                                      eps, flags, _stream()), "selftok_code_gather_ln_f32")
    return out.reshape(*ids.shape, D)


# ----------------------------------------------------------------------------------------------
# fused epilogues (csrc/elementwise.hip)
# ----------------------------------------------------------------------------------------------

def residual_ln_mod(x, *, y=None, gate=None, shift=None, scale=None, per_sample=False, gate_per_sample=None,
                    want_x=True, want_n=True, eps=1e-6, split=False, overflow=None):
    """x' = x + gate*y ; n = LN(x')*(1+scale)+shift.   x,y [B,T,H].  shift/scale/gate are 2-D views
    [T,H] (per token, default) or [B,H] (per_sample=True) -- typically column slices of a [*,6H] table.
    Returns (x', n) (either may be None).  split=True: n comes back as a SplitAct [B,T,H] for linear_f16x2_split; `overflow` bit 0
    is raised if |n| >= 65504."""
    _need_cuda(x)
    B, T, H = x.shape
    assert x.is_contiguous() and x.dtype == torch.float32

    def strides(t, ps):
        if t is None:
            return 0, 0
        assert t.dim() == 2 and t.stride(1) == 1 and t.shape[1] == H and t.shape[0] == (B if ps else T)
        return (t.stride(0), 0) if ps else (0, t.stride(0))

    if shift is not None:
        assert scale is not None and shift.stride(0) == scale.stride(0)
    msb, mst = strides(shift, per_sample)
    gsb, gst = strides(gate, per_sample if gate_per_sample is None else gate_per_sample)
    if y is not None:
        assert y.is_contiguous() and y.shape == x.shape
    x_out = torch.empty_like(x) if (y is not None and want_x) else None
    if split and want_n:
        n_s = SplitAct((B, T, H), x.device)
        _lib.check(_lib.load().selftok_residual_ln_mod_split(_p(x), _p(y), _p(gate), _p(shift), _p(scale), _p(x_out), _p(n_s.data),
                                                             _p(overflow), B, T, H, msb, mst, gsb, gst, eps, _stream()), "selftok_residual_ln_mod_split")
        return (x_out if y is not None else x), n_s
    n_out = torch.empty_like(x) if want_n else None
    _lib.check(_lib.load().selftok_residual_ln_mod_f32(_p(x), _p(y), _p(gate), _p(shift), _p(scale), _p(x_out), _p(n_out),
                                                       B, T, H, msb, mst, gsb, gst, eps, _stream()), "selftok_residual_ln_mod_f32")
    return (x_out if y is not None else x), n_out


def bias_gelu_(h, bias=None):
    _need_cuda(h)
    assert h.is_contiguous() and h.dtype == torch.float32
    cols = h.shape[-1]
    _lib.check(_lib.load().selftok_bias_gelu_f32(_p(h), _p(bias), h.numel() // cols, cols, _stream()), "selftok_bias_gelu_f32")
    return h


def linear_gelu(x, weight, bias):
    """gelu_tanh(x @ weight.T + bias) -- Mlp.fc1 + act (sd3/other_impls.py:86-88; timm Mlp modules.py:109,293).
    The bias add and the tanh-GELU ride in hipBLASLt's GEMM epilogue (one pass less over the [rows, 4H] hidden tensor:
    -8 % per MLP GEMM, measured equal to GEMM + selftok_bias_gelu_f32 to 5e-7); `bias_gelu_` remains the stand-alone kernel."""
    _need_cuda(x, weight)
    shp = x.shape
    out = torch._addmm_activation(bias, x.reshape(-1, shp[-1]), weight.t(), use_gelu=True)
    return out.reshape(*shp[:-1], weight.shape[0])


# ----------------------------------------------------------------------------------------------
# fp32-equivalent Linear on the f16 matrix cores (csrc/gemm_split.hip)
# ----------------------------------------------------------------------------------------------
LINEAR_GELU = 1


def linear_f16x2_supported(N: int, K: int) -> bool:
    return N % 128 == 0 and K % 32 == 0


def linear_f16x2_pack(weight: torch.Tensor, overflow: torch.Tensor = None) -> torch.Tensor:
    """nn.Linear weight [N,K] fp32 -> the kernel's split/tiled fp16 image (4*N*K bytes).  One-time, at load.
    `overflow` (int32 [1], device): bit 1 is set if a weight is outside the fp16 range."""
    _need_cuda(weight)
    w = weight.contiguous().float()
    N, K = w.shape
    lib = _lib.load()
    nbytes = lib.selftok_linear_f16x2_packed_bytes(N, K)
    if nbytes == 0:
        raise _lib.SelftokHipError(f"linear_f16x2: weight {N}x{K} needs N % 128 == 0 and K % 32 == 0")
    packed = torch.empty(nbytes // 2, dtype=torch.float16, device=w.device)
    _lib.check(lib.selftok_linear_f16x2_pack_weight(_p(w), _p(packed), N, K, _p(overflow), _stream()), "selftok_linear_f16x2_pack_weight")
    return packed


def linear_f16x2(x: torch.Tensor, packed: torch.Tensor, bias, N: int, gelu: bool = False, overflow: torch.Tensor = None) -> torch.Tensor:
    """act(x @ W.T + bias) with W given as linear_f16x2_pack(W).  x [..., K] fp32 (rows may be strided), out [..., N] fp32."""
    _need_cuda(x, packed)
    K = x.shape[-1]
    assert x.dtype == torch.float32 and packed.numel() * 2 == 4 * N * K, "packed weight does not match (N, K)"
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1 or (x2.shape[0] > 1 and (x2.stride(0) % 4 or x2.stride(0) < K)) or x2.data_ptr() % 16:
        x2 = x2.contiguous()                              # the kernel reads rows with 16-byte loads
    M = x2.shape[0]
    out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    lda = x2.stride(0) if M > 1 else K
    _lib.check(_lib.load().selftok_linear_f16x2_f32(_p(x2), lda, _p(packed), _p(bias), _p(out), N, M, N, K,
                                                    LINEAR_GELU if gelu else 0, _p(overflow), _stream()), "selftok_linear_f16x2_f32")
    return out.reshape(*x.shape[:-1], N)


class SplitAct:
    """A "split activation" (include/selftok_hip.h): the fp32 tensor of logical `shape` [..., K] held as fp16 hi / lo planes in
    1-KiB chunks of 16 rows x 32 k, `data` = fp16 [ceil(rows/16), K/32, 2, 16, 32].  What the fused producers (residual_ln_mod,
    attention, the f16x2 Linear's epilogue) write and linear_f16x2_split consumes; `dtype` is torch.float16 so that callers can
    tell it from an fp32 tensor the way they tell dtypes apart."""
    dtype = torch.float16

    def __init__(self, shape, device, zero: bool = False):
        self.shape = tuple(int(v) for v in shape)
        K = self.shape[-1]
        assert K % 32 == 0, "split activations need K % 32 == 0"
        self.rows = 1
        for v in self.shape[:-1]:
            self.rows *= v
        self.data = (torch.zeros if zero else torch.empty)((self.rows + 15) // 16, K // 32, 2, 16, 32, dtype=torch.float16, device=device)

    @property
    def device(self):
        return self.data.device

    def planes(self) -> torch.Tensor:
        """un-blocked copy fp16 [2, *shape] (hi, lo): tests / debugging"""
        K = self.shape[-1]
        p = self.data.permute(2, 0, 3, 1, 4).reshape(2, -1, K)[:, :self.rows]
        return p.reshape(2, *self.shape)


def split_f16x2(x: torch.Tensor, overflow: torch.Tensor = None) -> SplitAct:
    """fp32 [..., K] -> SplitAct: hi = fp16(x), lo = fp16((x - hi) * 2^11).  The stand-alone producer, for inputs no fused producer
    makes.  `overflow` bit 0 is raised for |x| >= 65504."""
    _need_cuda(x)
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1 or (x2.shape[0] > 1 and (x2.stride(0) % 4 or x2.stride(0) < K)) or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    out = SplitAct(x.shape, x.device)
    _lib.check(_lib.load().selftok_split_f16x2_f32(_p(x2), x2.stride(0) if out.rows > 1 else K, _p(out.data), out.rows, K,
                                                   _p(overflow), _stream()), "selftok_split_f16x2_f32")
    return out


def split_to_f32(xs: SplitAct) -> torch.Tensor:
    """the fp32 value a split activation stands for (tests / debugging)"""
    p = xs.planes()
    return p[0].float() + p[1].float() * (1.0 / 2048.0)


@functools.lru_cache(maxsize=None)
def _cu_count() -> int:
    return int(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count) if torch.cuda.is_available() else 256


SPLITK_MAX_ROWS = 1024      # rows up to which the block Linears of the f16x2 mode are worth splitting along K (one .. four images)


@functools.lru_cache(maxsize=4096)
def f16x2_ksplit(M: int, N: int, K: int) -> int:
    """how many work-groups should share an output tile of a [M, K] x [K, N] f16x2 Linear: 1 = the single-pass kernel.  Small M only
    (<= SPLITK_MAX_ROWS): the 256 x 128 tile grid then leaves most of the 256 CUs idle while each work-group walks all of K.  Among
    the divisors s of K / 32 that keep the grid within one work-group per CU (a second round of work-groups costs ~10 us:
    profiles/r4_splitk_sweep.txt) and >= 3 k-tiles per work-group (the DMA pipeline depth), the minimum of a two-term cost fitted to
    that sweep: 0.42 us per k-tile of the longest chain + 1.2 us per million fp32 partials written and re-read, + 4 us for the
    second launch.  NOTE: the split is chosen from M = batch rows, and a split-K sum rounds differently from the single pass -- in f16x2
    mode an image decoded at B <= 4 and at B > 4 differs at fp32 rounding level (1e-7 relative per Linear; the fp32 mode and every encode
    path are batch independent).  The cost constants are one MI355X sweep; the CU count comes from the device."""
    if M <= 0 or M > SPLITK_MAX_ROWS:
        return 1
    tiles = ((M + 255) // 256) * (N // 128)
    kt = K // 32
    cus = _cu_count()
    best, best_cost = 1, 0.42 * kt
    for s in range(2, 65):
        if kt % s or kt // s < 3 or tiles * s > cus:
            continue
        cost = 0.42 * (kt // s) + 1.2e-6 * s * M * N + 4.0
        if cost < best_cost:
            best, best_cost = s, cost
    return best


def _splitk_ws(M: int, N: int, ksplit: int, device) -> torch.Tensor:
    return torch.empty(ksplit * M * N, dtype=torch.float32, device=device)


def linear_f16x2_split(xs: SplitAct, packed: torch.Tensor, bias, N: int, gelu: bool = False, overflow: torch.Tensor = None,
                       out_split: bool = False, ksplit: int = 1):
    """linear_f16x2 on a split activation: both operands reach LDS by LDS-DMA.  Returns fp32 [..., N], or with out_split a SplitAct
    [..., N] for the next Linear.  Same results as linear_f16x2 on the fp32 tensor.  ksplit > 1 (small M, `f16x2_ksplit`): that many
    work-groups per output tile + a reduction launch; deterministic, equal to the single-pass result up to fp32 rounding of the
    partial sums (selftok_linear_f16x2_split_k)."""
    _need_cuda(xs.data, packed)
    K = xs.shape[-1]
    assert packed.numel() * 2 == 4 * N * K, "packed weight does not match (N, K)"
    M, lead = xs.rows, xs.shape[:-1]
    lib = _lib.load()
    flags = LINEAR_GELU if gelu else 0
    ws = _splitk_ws(M, N, ksplit, xs.device) if (ksplit > 1 and M > 0) else None
    if out_split:
        out = SplitAct((*lead, N), xs.device)
        _lib.check(lib.selftok_linear_f16x2_split_k(_p(xs.data), _p(packed), _p(bias), None, _p(out.data), N, M, N, K, flags, ksplit, _p(ws), _p(overflow), _stream()),
                   "selftok_linear_f16x2_split_k")
        return out
    out = torch.empty(M, N, dtype=torch.float32, device=xs.device)
    _lib.check(lib.selftok_linear_f16x2_split_k(_p(xs.data), _p(packed), _p(bias), _p(out), None, N, M, N, K, flags, ksplit, _p(ws), _p(overflow), _stream()),
               "selftok_linear_f16x2_split_k")
    return out.reshape(*lead, N)


def linear_f16x2_split_residual(xs: SplitAct, packed: torch.Tensor, bias, N: int, resid: torch.Tensor, gate=None,
                                gate_per_sample: bool = False, overflow: torch.Tensor = None, ksplit: int = 1) -> torch.Tensor:
    """resid + gate * (xs @ W.T + bias): linear_f16x2_split with the block's residual update fused into the epilogue.
    resid [B,T,N] fp32 contiguous; gate a 2-D view [T,N] (per token) or [B,N] (gate_per_sample) with unit inner stride, or None.
    Bit-identical to residual_ln_mod(resid, y=linear_f16x2_split(...), gate=gate)[0]."""
    _need_cuda(xs.data, packed, resid)
    K = xs.shape[-1]
    assert packed.numel() * 2 == 4 * N * K
    assert resid.dim() == 3 and resid.is_contiguous() and resid.dtype == torch.float32 and resid.shape[-1] == N
    B, T, _ = resid.shape
    M = B * T
    assert xs.rows == M
    gsb = gst = 0
    if gate is not None:
        assert gate.dim() == 2 and gate.stride(1) == 1 and gate.shape == ((B, N) if gate_per_sample else (T, N))
        gsb, gst = (gate.stride(0), 0) if gate_per_sample else (0, gate.stride(0))
    out = torch.empty_like(resid)
    ws = _splitk_ws(M, N, ksplit, resid.device) if (ksplit > 1 and M > 0) else None
    _lib.check(_lib.load().selftok_linear_f16x2_split_residual_k(_p(xs.data), _p(packed), _p(bias), _p(resid), N, _p(gate), gsb, gst, T,
                                                             _p(out), N, M, N, K, ksplit, _p(ws), _p(overflow), _stream()), "selftok_linear_f16x2_split_residual_k")
    return out


def silu(x):
    _need_cuda(x)
    x = x.contiguous()
    out = torch.empty_like(x)
    _lib.check(_lib.load().selftok_silu_f32(_p(x), _p(out), x.numel(), _stream()), "selftok_silu_f32")
    return out


def add_rows_(x, table):
    """x[b] += table for every b (in place). x [B,...], table [...] contiguous."""
    _need_cuda(x, table)
    assert x.is_contiguous() and table.is_contiguous() and x[0].numel() == table.numel()
    _lib.check(_lib.load().selftok_add_rows_f32(_p(x), _p(table), _p(x), x.shape[0], table.numel(), _stream()), "selftok_add_rows_f32")
    return x


def timestep_embed(t, freqs, t_scale=1.0):
    _need_cuda(t, freqs)
    t = t.contiguous().float()
    n, half = t.numel(), freqs.numel()
    out = torch.empty(n, 2 * half, dtype=torch.float32, device=t.device)
    _lib.check(_lib.load().selftok_timestep_embed_f32(_p(t), _p(freqs), _p(out), n, 2 * half, float(t_scale), _stream()), "selftok_timestep_embed_f32")
    return out


def patchify(x):
    _need_cuda(x)
    x = x.contiguous().float()
    B, C, H, W = x.shape
    out = torch.empty(B, (H // 2) * (W // 2), 4 * C, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().selftok_patchify_f32(_p(x), _p(out), B, C, H, W, _stream()), "selftok_patchify_f32")
    return out


def unpatchify_cfg_euler(y_cond, x=None, dt=0.0, y_uncond=None, cfg_scale=1.0, C=16, hp=16, wp=16, want_v=False):
    """returns (x_new or None, v or None)"""
    _need_cuda(y_cond)
    B = y_cond.shape[0]
    assert y_cond.is_contiguous() and y_cond.shape[1] == hp * wp and y_cond.shape[2] == 4 * C
    x_out = torch.empty(B, C, 2 * hp, 2 * wp, dtype=torch.float32, device=y_cond.device) if x is not None else None
    v_out = torch.empty(B, C, 2 * hp, 2 * wp, dtype=torch.float32, device=y_cond.device) if (want_v or x is None) else None
    if x is not None:
        assert x.is_contiguous()
    _lib.check(_lib.load().selftok_unpatchify_cfg_euler_f32(_p(y_cond), _p(y_uncond), _p(x), _p(x_out), _p(v_out), B, C, hp, wp,
                                                            float(dt), float(cfg_scale), _stream()), "selftok_unpatchify_cfg_euler_f32")
    return x_out, v_out


def rmsnorm(x, w=None, eps=1e-6):
    _need_cuda(x)
    x = x.contiguous().float()
    out = torch.empty_like(x)
    dim = x.shape[-1]
    _lib.check(_lib.load().selftok_rmsnorm_f32(_p(x), _p(w), _p(out), x.numel() // dim, dim, eps, _stream()), "selftok_rmsnorm_f32")
    return out


def rotary(t, freqs, start_index: int = 0, scale: float = 1.0):
    """apply_rotary_emb(freqs, t, start_index, scale) (utils/rotary_embedding_torch.py:37-53): t [..., seq, dim], freqs [seq, rot_dim];
    features start_index .. start_index + rot_dim are rotated, the rest passes through."""
    _need_cuda(t, freqs)
    t = t.contiguous().float()
    freqs = freqs.contiguous().float()
    seq, rot = freqs.shape
    dim = t.shape[-1]
    if rot > dim - start_index:
        raise ValueError(f"feature dimension {dim} is not of sufficient size to rotate in all the positions {rot}")
    mid = t if (start_index == 0 and rot == dim) else t[..., start_index:start_index + rot].contiguous()
    out = torch.empty_like(mid)
    _lib.check(_lib.load().selftok_rotary_f32(_p(mid), _p(freqs), _p(out), mid.numel() // rot, seq, rot, float(scale), _stream()), "selftok_rotary_f32")
    if mid is t:
        return out
    return torch.cat((t[..., :start_index], out, t[..., start_index + rot:]), dim=-1)


# ----------------------------------------------------------------------------------------------
# attention (csrc/attention.hip)
# ----------------------------------------------------------------------------------------------

def _seg(q, k, v, o):
    """q/k/v/o: 3-D views [B, L, H*Dh] with unit inner stride (slices of a fused qkv buffer are fine)."""
    s = _lib.AttnSeg()
    if k is None:
        return s
    if isinstance(o, SplitAct):      # split-activation output [B,L,H*Dh]: its pointer goes into the descriptor (attention())
        assert o.shape == (k.shape[0], q.shape[1], q.shape[2])
        o = None
    for name, t in (("q", q), ("k", k), ("v", v), ("o", o)):
        if t is None:
            continue
        assert t.dim() == 3 and t.stride(2) == 1 and t.dtype == torch.float32
        setattr(s, name, t.data_ptr())
        setattr(s, name + "_rs", t.stride(1))
        setattr(s, name + "_bs", t.stride(0))
    s.len = k.shape[1]
    return s


ATTN_F16X2 = 1


def attention(seg0, seg1, heads, head_dim, kvis=None, seg0_sees_seg1=True, scale=None, mode: int = 0, overflow=None):
    """seg = (q, k, v, o) tuples of [B,L,heads*head_dim] views (q and o None: keys/values only; seg None: empty).
    Writes into the `o` views.  kvis: int32 [B] or None.  mode = ATTN_F16X2: f16x2-split matrix products (head_dim 64),
    `overflow` (int32 [1] device tensor) gets bit 2 if an operand is outside the fp16 range."""
    lib = _lib.load()
    d = _lib.AttnDesc()
    ref = seg1 if seg1 is not None else seg0
    _need_cuda(ref[1])
    d.seg[0] = _seg(*seg0) if seg0 is not None else _lib.AttnSeg()
    d.seg[1] = _seg(*seg1) if seg1 is not None else _lib.AttnSeg()
    for i, sg in enumerate((seg0, seg1)):               # `o` given as a SplitAct [B,L,H*Dh]; f16x2 mode only
        if sg is not None and isinstance(sg[3], SplitAct):
            d.o_blk[i] = sg[3].data.data_ptr()
    d.B, d.H, d.head_dim = ref[1].shape[0], heads, head_dim
    if kvis is not None:
        assert kvis.dtype == torch.int32 and kvis.is_cuda and kvis.numel() == d.B
        d.kvis = kvis.data_ptr()
    d.seg0_sees_seg1 = 1 if seg0_sees_seg1 else 0
    d.scale = float(scale if scale is not None else head_dim ** -0.5)
    d.mode = int(mode) if head_dim == 64 else 0
    d.overflow = _p(overflow)
    import ctypes
    _lib.check(lib.selftok_attn_f32(ctypes.byref(d), _stream()), "selftok_attn_f32")


# ----------------------------------------------------------------------------------------------
# VAE epilogues (csrc/vae.hip)
# ----------------------------------------------------------------------------------------------

def groupnorm_silu(x, weight, bias, groups=32, eps=1e-6, silu_act=True):
    _need_cuda(x)
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    _lib.check(_lib.load().selftok_groupnorm_silu_bf16(_p(x), _p(weight), _p(bias), _p(out), B, C, H * W, groups, eps,
                                                       1 if silu_act else 0, _stream()), "selftok_groupnorm_silu_bf16")
    return out


class PackedConv:
    """one convolution's weights in the opaque image selftok_conv2d_nhwc_bf16 reads (csrc/conv.hip) + its bias and geometry"""
    __slots__ = ("packed", "bias", "cout", "cin", "ksize", "bn")

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor]):
        _need_cuda(weight, bias)
        assert weight.dtype == torch.bfloat16 and weight.dim() == 4 and weight.shape[2] == weight.shape[3] and weight.shape[2] in (1, 3)
        w = weight.contiguous()
        self.cout, self.cin, self.ksize = int(w.shape[0]), int(w.shape[1]), int(w.shape[2])
        self.bn = 128 if self.cout >= 64 else 32
        lib = _lib.load()
        self.packed = torch.empty(lib.selftok_conv2d_packed_bytes(self.cout, self.cin, self.ksize, self.bn), dtype=torch.uint8, device=w.device)
        _lib.check(lib.selftok_conv2d_pack_weight_bf16(_p(w), _p(self.packed), self.cout, self.cin, self.ksize, self.bn, _stream()), "selftok_conv2d_pack_weight_bf16")
        self.bias = None if bias is None else bias.to(torch.bfloat16).contiguous()


def conv2d_nhwc(x: torch.Tensor, pc: PackedConv, stride: int = 1, upsample: bool = False, residual: Optional[torch.Tensor] = None,
                cstore: Optional[int] = None) -> torch.Tensor:
    """x [B,H,W,Cin'] bf16 channels-last (Cin' = the layer's Cin rounded up to a multiple of 8, extra channels ignored) ->
    [B,Ho,Wo,cstore] bf16: the convolution with the reference's CPU arithmetic (fp32 accumulate incl. the bias, one rounding).
    stride 2 = Downsample (pad right/bottom + stride-2), upsample = nearest 2x applied to the input on the fly, residual [B,Ho,Wo,cstore]
    is added with its own bf16 rounding (ResnetBlock's `x + h`)."""
    _need_cuda(x, residual)
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and x.dim() == 4
    B, H, W, Cin = x.shape
    assert Cin % 8 == 0 and (Cin + 31) // 32 == (pc.cin + 31) // 32 and Cin >= pc.cin, (Cin, pc.cin)
    cs = (pc.cout + 3) // 4 * 4 if cstore is None else cstore
    Hi, Wi = (H * 2, W * 2) if upsample else (H, W)
    Ho, Wo = (Hi // 2, Wi // 2) if stride == 2 else (Hi, Wi)
    out = torch.empty(B, Ho, Wo, cs, dtype=torch.bfloat16, device=x.device)
    if residual is not None:
        assert residual.shape == out.shape and residual.is_contiguous() and residual.dtype == torch.bfloat16
    _lib.check(_lib.load().selftok_conv2d_nhwc_bf16(_p(x), _p(pc.packed), _p(pc.bias), _p(residual), _p(out), B, H, W, Cin, pc.cout, cs, cs, pc.ksize, stride,
                                                    1 if upsample else 0, pc.bn, _stream()), "selftok_conv2d_nhwc_bf16")
    return out


def groupnorm_silu_nhwc(x, weight, bias, groups=32, eps=1e-6, silu_act=True):
    """GroupNorm [+ SiLU] on a channels-last bf16 tensor [B, ..., C]"""
    _need_cuda(x, weight, bias)
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    lib = _lib.load()
    out = torch.empty_like(x)
    ws = torch.empty(lib.selftok_groupnorm_nhwc_workspace_bytes(B, HW, C), dtype=torch.uint8, device=x.device)
    _lib.check(lib.selftok_groupnorm_silu_nhwc_bf16(_p(x), _p(weight), _p(bias), _p(out), _p(ws), B, HW, C, groups, eps, 1 if silu_act else 0, _stream()),
               "selftok_groupnorm_silu_nhwc_bf16")
    return out


def latent_process_in(moments, c_keep=16, shift=0.0609, scale=1.5305):
    _need_cuda(moments)
    assert moments.dtype == torch.bfloat16 and moments.is_contiguous()
    B, Cin, H, W = moments.shape
    out = torch.empty(B, c_keep, H, W, dtype=torch.float32, device=moments.device)
    _lib.check(_lib.load().selftok_latent_process_in(_p(moments), _p(out), B, Cin, c_keep, H * W, shift, scale, _stream()), "selftok_latent_process_in")
    return out


def latent_process_out(z, shift=0.0609, scale=1.5305):
    _need_cuda(z)
    z = z.contiguous().float()
    out = torch.empty(z.shape, dtype=torch.bfloat16, device=z.device)
    _lib.check(_lib.load().selftok_latent_process_out(_p(z), _p(out), z.numel(), shift, scale, _stream()), "selftok_latent_process_out")
    return out


def clamp01_(img):
    _need_cuda(img)
    assert img.dtype == torch.bfloat16 and img.is_contiguous()
    _lib.check(_lib.load().selftok_clamp01_bf16(_p(img), img.numel(), _stream()), "selftok_clamp01_bf16")
    return img


def vq_encode_split_launch(z, packed_codebook, ids_dtype=torch.int64, coarse=None, _flags: Optional[int] = None):
    """Same result as vq_encode(packed=True) but returns (ids, launch_main, launch_finalize) closures so a
    benchmark can time the main argmax kernel alone (HIP events around launch_main on the current stream).
    `coarse`: None / False / True / 1 / 3 as in vq_encode; `_flags`: the raw SELFTOK_VQ_* flag word instead (sweeps)."""
    import ctypes
    lib = _lib.load()
    zz = z.contiguous().float().reshape(-1, z.shape[-1])
    N, D = zz.shape
    C = packed_codes(packed_codebook, D)
    ids = torch.empty(N, dtype=ids_dtype, device=z.device)
    ws = torch.empty(lib.selftok_vq_workspace_bytes(N, C), dtype=torch.uint8, device=z.device)
    flags = (IDS_I32 if ids_dtype == torch.int32 else 0) | (int(_flags) if _flags is not None else _coarse_flags(coarse))
    nsplit = ctypes.c_int(0)

    def launch_main():
        _lib.check(lib.selftok_vq_argmax_partial_packed_f32(_p(zz), _p(packed_codebook), _p(ws), ctypes.addressof(nsplit), N, C, D,
                                                            flags, _stream()), "selftok_vq_argmax_partial_packed_f32")

    def launch_finalize():
        _lib.check(lib.selftok_vq_finalize_packed(_p(ws), _p(zz), _p(packed_codebook), _p(ids), None, N, C, D, nsplit.value, flags,
                                                  _stream()), "selftok_vq_finalize_packed")

    return ids.reshape(z.shape[:-1]), launch_main, launch_finalize


# ---- the exact-order SD3-VAE encoder kernels (csrc/vae_exact.hip) -------------------------------------------------------------------
VX_UPSAMPLE2X = 8


def vx_conv2d(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, stride: int = 1, residual: Optional[torch.Tensor] = None, order: int = 0,
              cin: Optional[int] = None, upsample: bool = False) -> torch.Tensor:
    """x [B,H,W,ldx] bf16 channels-last, w [Cout,k,k,Cin] bf16 (the checkpoint's tensor permuted), bias [Cout] bf16 -> [B,Ho,Wo,Cout] bf16 with the
    summation order of the reference's CPU convolution (oneDNN AMX chunks; include/selftok_hip.h).  `order`: 0 / 3 / 1 / 2 (conv_in, cin = 3).
    `upsample`: the convolution reads the nearest-2x upsampled view of x (the decoder's Upsample layer; output [B,2H,2W,Cout])."""
    _need_cuda(x, w, bias, residual)
    assert x.dtype == w.dtype == bias.dtype == torch.bfloat16 and x.is_contiguous() and w.is_contiguous() and x.dim() == 4 and w.dim() == 4
    B, H, W, ldx = x.shape
    Cout, k, k2, Cin = w.shape
    assert k == k2 and (cin is None or cin == Cin)
    if upsample:
        assert stride == 1
        H, W, order = 2 * H, 2 * W, order | VX_UPSAMPLE2X
    Ho, Wo = (H // 2, W // 2) if stride == 2 else (H, W)
    out = torch.empty(B, Ho, Wo, Cout, dtype=torch.bfloat16, device=x.device)
    if residual is not None:
        assert residual.shape == out.shape and residual.is_contiguous() and residual.dtype == torch.bfloat16
    _lib.check(_lib.load().selftok_vx_conv2d_bf16(_p(x), _p(w), _p(bias), _p(residual), _p(out), B, H, W, ldx, Cin, Cout, k, stride, order, _stream()),
               "selftok_vx_conv2d_bf16")
    return out


def vx_silu_table(device) -> torch.Tensor:
    """torch-CPU's bf16 SiLU as a 65536-entry table (int16 view of bf16 bits), built on the device"""
    t = torch.empty(65536, dtype=torch.int16, device=device)
    with torch.cuda.device(t.device):
        _lib.check(_lib.load().selftok_vx_silu_table_bf16(_p(t), _stream()), "selftok_vx_silu_table_bf16")
    return t


def vx_groupnorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, silu_table: Optional[torch.Tensor] = None, groups: int = 32, eps: float = 1e-6,
                 want_stats: bool = False):
    """GroupNorm [+ SiLU via `silu_table`] on [B, ..., C] bf16 channels-last with ATen's CPU statistics order (include/selftok_hip.h)"""
    _need_cuda(x, gamma, beta, silu_table)
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    B, C = x.shape[0], x.shape[-1]
    if B == 0:
        return (torch.empty_like(x), torch.empty(0, groups, 2, dtype=torch.float32, device=x.device)) if want_stats else torch.empty_like(x)
    HW = x.numel() // (B * C)
    lib = _lib.load()
    nbytes = lib.selftok_vx_groupnorm_workspace_bytes(B, HW, C)
    if nbytes == 0:
        raise _lib.SelftokHipError(f"vx_groupnorm: unsupported shape B={B} HW={HW} C={C}")
    out = torch.empty_like(x)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    stats = torch.empty(B, groups, 2, dtype=torch.float32, device=x.device) if want_stats else None
    _lib.check(lib.selftok_vx_groupnorm_bf16(_p(x), _p(gamma), _p(beta), _p(out), _p(ws), _p(silu_table), _p(stats), B, HW, C, groups, float(eps), _stream()),
               "selftok_vx_groupnorm_bf16")
    return (out, stats) if want_stats else out


def vx_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """q, k, v [B,T,C] bf16 -> [B,T,C] bf16: one-head attention as ATen's CPU flash kernel evaluates it (T = 1024)"""
    _need_cuda(q, k, v)
    assert q.dtype == k.dtype == v.dtype == torch.bfloat16 and q.is_contiguous() and k.is_contiguous() and v.is_contiguous() and q.shape == k.shape == v.shape
    B, T, C = q.shape
    lib = _lib.load()
    out = torch.empty_like(q)
    if B == 0:
        return out
    ws = torch.empty(lib.selftok_vx_attention_workspace_bytes(B, T, C), dtype=torch.uint8, device=q.device)
    _lib.check(lib.selftok_vx_attention_bf16(_p(q), _p(k), _p(v), _p(out), _p(ws), B, T, C, _stream()), "selftok_vx_attention_bf16")
    return out


def vx_expf(x: torch.Tensor) -> torch.Tensor:
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    y = torch.empty_like(x)
    _lib.check(_lib.load().selftok_vx_expf_f32(_p(x), _p(y), x.numel(), _stream()), "selftok_vx_expf_f32")
    return y


# ----------------------------------------------------------------------------------------------
# the fp32 Q-Former encoder in the reference's exact summation orders (csrc/encoder_exact.hip)
# ----------------------------------------------------------------------------------------------

def _rows2d(t: torch.Tensor):
    """a [..., C] fp32 tensor whose rows are equally strided (contiguous, or a column slice of a contiguous fused projection) -> (rows, row stride)"""
    assert t.dtype == torch.float32 and t.stride(-1) == 1
    ld = t.stride(-2) if t.dim() > 1 else t.shape[-1]
    rows = t.numel() // t.shape[-1]
    for d in range(t.dim() - 2, 0, -1):                              # leading dims must step by whole row blocks
        assert t.stride(d - 1) == t.stride(d) * t.shape[d], "rows are not equally strided"
    return rows, ld


EX_LINEAR_GELU_ON_XE = True     # fc1 + GELU stays on xe_gemm128 with the GELU in its epilogue: in the model 620 / 252 ms per sampler step against 625 - 641 / 255 with
                                # csrc/gemm_fp32.hip + a GELU pass (profiles/r6_exact_mode_ab_v11_fused_ln.txt; the difference is inside the run-to-run noise, the pass is not)
EX_LINEAR_SG_MIN_ROWS = 256     # from this many rows on `ex_linear` runs on the LDS-DMA staged kernel (csrc/gemm_fp32.hip, bit-identical results) where the shape allows


def ex_linear(x: torch.Tensor, weight: torch.Tensor, bias=None, *, gelu: bool = False, res=None, res_mod: int = 0, gate=None, gate_mod: int = 0,
              out: Optional[torch.Tensor] = None, bias_last: bool = False, kernel: str = "auto") -> torch.Tensor:
    """F.linear in MKL sgemm's BLOCKED summation order -- bit-equal to torch-CPU's nn.Linear where MKL takes that path (probed: M >= 512 token rows, >= 16
    conditioning rows; below that MKL switches strategy and this kernel still computes the blocked order: the result for a row is the one it has inside a large
    batch, whatever M is here), optional exact GELU(tanh),
    optional `res + gate * y` epilogue (res / gate rows taken modulo res_mod / gate_mod when positive: per-token tables; divided by
    -mod when negative: per-sample tables).  bias_last: (sum of the K-blocks) + bias, what at::linear computes for a non-contiguous input.
    `kernel`: 'xe' = xe_gemm / xe_gemm128 (csrc/encoder_exact.hip, any shape), 'sg' = the LDS-DMA staged kernel of round 6 (csrc/gemm_fp32.hip: N % 128 == 0,
    K % 32 == 0, K outside (384, 768)), 'auto' = 'sg' from EX_LINEAR_SG_MIN_ROWS rows on where the shape allows.  Same bits either way (tests/test_gemm_fp32_gpu.py)."""
    N, K = weight.shape
    if kernel not in ("auto", "xe", "sg"):
        raise ValueError(f"ex_linear kernel {kernel!r}: expected 'auto', 'xe' or 'sg'")
    sg_ok = not (gelu and (res is not None or gate is not None or (out is not None and not out.is_contiguous())))      # the LDS-DMA kernel's GELU: contiguous out, no res / gate
    if kernel == "sg" or (kernel == "auto" and sg_ok and linear_f32_supported(N, K, mkl_order=True) and x.numel() // max(K, 1) >= EX_LINEAR_SG_MIN_ROWS
                          and not (gelu and EX_LINEAR_GELU_ON_XE)):
        # fc1 + GELU: selftok_linear_f32 runs the GELU as a second launch over `out` (in the GEMM's epilogue four waves per CU worked through ~80 VALU instructions
        # per output while the matrix pipe idled: +0.40 ms on a 2.28 ms Linear; as an element-wise pass +0.2 ms).  Same bits: GELU of the same fp32 value.
        return linear_f32(x, weight, bias, mkl_order=True, gelu=gelu, res=res, res_mod=res_mod, gate=gate, gate_mod=gate_mod, out=out, bias_last=bias_last)
    _need_cuda(x, weight, bias, res, gate)
    assert weight.dtype == torch.float32 and weight.is_contiguous() and x.shape[-1] == K
    M, ldx = _rows2d(x)
    if out is None:
        out = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
    assert out.is_contiguous() or out.stride(-1) == 1
    _, ldo = _rows2d(out)
    ldr = _rows2d(res)[1] if res is not None else 0
    ldg = _rows2d(gate)[1] if gate is not None else 0
    _lib.check(_lib.load().selftok_ex_linear_f32(_p(x), ldx, _p(weight), _p(bias), _p(res), ldr, int(res_mod), _p(gate), ldg, int(gate_mod), _p(out), ldo,
                                                 M, N, K, int(bool(gelu)) | (2 if bias_last else 0), _stream()), "selftok_ex_linear_f32")
    return out


LINEAR_BIAS_LAST, LINEAR_MKL_ORDER = 2, 4
_LINEAR_WS = {}          # (device index, stream handle) -> workspace tensors of the tail split, largest last: grown on demand, reused by the calls of THAT stream (they are
                         # ordered); two streams never share one (their tail rounds could overlap); outgrown tensors stay referenced (a captured hipGraph may replay on them)


def _linear_ws(device, nbytes: int):
    key = (device.index, int(torch.cuda.current_stream(device).cuda_stream))
    held = _LINEAR_WS.setdefault(key, [])
    if not held or held[-1].numel() < nbytes:
        held.append(torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device))
    return held[-1]


def linear_f32(x: torch.Tensor, weight: torch.Tensor, bias=None, *, mkl_order: bool = False, gelu: bool = False, res=None, res_mod: int = 0, gate=None,
               gate_mod: int = 0, out: Optional[torch.Tensor] = None, bias_last: bool = False, split: int = 0, use_workspace: bool = True, _flags: int = 0) -> torch.Tensor:
    """F.linear on the LDS-DMA staged fp32-MFMA kernel (csrc/gemm_fp32.hip): N % 128 == 0, K % 32 == 0.  mkl_order: MKL sgemm's K-blocking, bit-identical to
    `ex_linear` (gemm='exact'); otherwise one k-ascending chain per output (gemm='fp32').  Epilogue arguments as `ex_linear`.  `split` (tests / tools) forces the
    tail split; `use_workspace=False` runs the tail round unsplit."""
    _need_cuda(x, weight, bias, res, gate)
    N, K = weight.shape
    assert weight.dtype == torch.float32 and weight.is_contiguous() and x.shape[-1] == K
    M, ldx = _rows2d(x)
    if out is None:
        out = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
    assert out.is_contiguous() or out.stride(-1) == 1
    _, ldo = _rows2d(out)
    ldr = _rows2d(res)[1] if res is not None else 0
    ldg = _rows2d(gate)[1] if gate is not None else 0
    flags = (1 if gelu else 0) | (LINEAR_BIAS_LAST if bias_last else 0) | (LINEAR_MKL_ORDER if mkl_order else 0) | ((int(split) & 0xFF) << 8) | int(_flags)
    lib = _lib.load()
    ws, nws = None, 0
    if use_workspace:
        nws = int(lib.selftok_linear_f32_workspace_bytes(M, N, K, flags))
        ws = _linear_ws(x.device, nws) if nws else None
    _lib.check(lib.selftok_linear_f32(_p(x), ldx, _p(weight), _p(bias), _p(res), ldr, int(res_mod), _p(gate), ldg, int(gate_mod), _p(out), ldo, M, N, K, flags,
                                      _p(ws), (ws.numel() if ws is not None else 0), _stream()), "selftok_linear_f32")
    return out


def linear_f32_supported(N: int, K: int, mkl_order: bool = False) -> bool:
    return N % 128 == 0 and K % 32 == 0 and not (mkl_order and 384 < K < 768)


def ex_layernorm_mod(x: torch.Tensor, shift=None, scale=None, gamma=None, beta=None, eps: float = 1e-6, want_stats: bool = False, per_sample: bool = False):
    """nn.LayerNorm in ATen's arithmetic, then `* (1 + scale[tok]) + shift[tok]` (tok = row % T; shift / scale [T, N] views, equal row stride;
    per_sample: x [B, rows, N] with tables [B, N], tok = the sample)"""
    _need_cuda(x, shift, scale, gamma, beta)
    N = x.shape[-1]
    rows, ldx = _rows2d(x)
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    T, ldt = 0, 0
    if shift is not None:
        assert scale is not None and shift.dim() == 2 and shift.shape == scale.shape and shift.stride(0) == scale.stride(0) and shift.stride(1) == 1 == scale.stride(1)
        T, ldt = shift.shape[0], shift.stride(0)
        if per_sample:
            assert x.dim() == 3 and shift.shape[0] == x.shape[0]
            T = -x.shape[1]
    stats = torch.empty(rows, 2, dtype=torch.float32, device=x.device) if want_stats else None
    _lib.check(_lib.load().selftok_ex_layernorm_mod_f32(_p(x), ldx, _p(out), N, _p(shift), _p(scale), ldt, T, _p(gamma), _p(beta), _p(stats), rows, N, float(eps),
                                                        _stream()), "selftok_ex_layernorm_mod_f32")
    return (out, stats) if want_stats else out


def ex_res_layernorm_mod(x: torch.Tensor, lin: torch.Tensor, *, lin_bias=None, gate=None, gate_mod: int = 0, shift=None, scale=None, eps: float = 1e-6,
                         per_sample: bool = False, x_out: Optional[torch.Tensor] = None):
    """x' = x + gate[row] * (lin + lin_bias); n = LayerNorm(x') * (1 + scale[tok]) + shift[tok]  ->  (x', n).  The residual update of a DismantledBlock
    (sd3/mmdit.py:485-496) fused into the LayerNorm + modulate behind it: the same bits as `ex_linear(..., res=x, gate=gate)` + `ex_layernorm_mod`, with the
    Linear keeping its plain epilogue (round 6).  gate rows: m % gate_mod (> 0: per-token table), m / -gate_mod (< 0: per-sample), m (0).  x_out may be x."""
    _need_cuda(x, lin, lin_bias, gate, shift, scale)
    N = x.shape[-1]
    rows, ldx = _rows2d(x)
    rl, ldl = _rows2d(lin)
    assert rl == rows and lin.shape[-1] == N
    xo = torch.empty(x.shape, dtype=torch.float32, device=x.device) if x_out is None else x_out
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    ldg = _rows2d(gate)[1] if gate is not None else 0
    T, ldt = 0, 0
    if shift is not None:
        assert scale is not None and shift.dim() == 2 and shift.shape == scale.shape and shift.stride(0) == scale.stride(0) and shift.stride(1) == 1 == scale.stride(1)
        T, ldt = shift.shape[0], shift.stride(0)
        if per_sample:
            assert x.dim() == 3 and shift.shape[0] == x.shape[0]
            T = -x.shape[1]
    _lib.check(_lib.load().selftok_ex_res_layernorm_mod_f32(_p(x), ldx, _p(lin), ldl, _p(lin_bias), _p(gate), ldg, int(gate_mod), _p(xo), _rows2d(xo)[1], _p(out), N,
                                                            _p(shift), _p(scale), ldt, T, rows, N, float(eps), _stream()), "selftok_ex_res_layernorm_mod_f32")
    return xo, out


EX_UNARY = {"gelu_tanh": 0, "silu": 1, "sleef_expf": 2, "sleef_tanhf": 3, "exp_u20": 4}


def ex_unary(x: torch.Tensor, kind: str, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """element-wise: y[i] = f(x[i]); `out` may be x itself (in place)"""
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    y = torch.empty_like(x) if out is None else out
    assert y.dtype == torch.float32 and y.is_contiguous() and y.numel() == x.numel()
    _lib.check(_lib.load().selftok_ex_unary_f32(_p(x), _p(y), x.numel(), EX_UNARY[kind], _stream()), "selftok_ex_unary_f32")
    return y


EX_ATTENTION_WS_LIMIT = 6 << 30          # bytes of score workspace one call of the unfused exact attention may allocate; larger batches run in slices
EX_ATTENTION_DEFAULT = "auto"     # what kernel='auto' means in ex_attention (tools set 'unfused' for A/B runs)


def ex_attention(q: torch.Tensor, k1, v1, heads: int, k2=None, v2=None, slots1: Optional[int] = None, kernel: str = "auto") -> torch.Tensor:
    """F.scaled_dot_product_attention as ATen's fp32 CPU flash kernel evaluates it.  q [B,Tq,H*D], k1 / v1 [B,Tk1,H*D] and an
    optional second key / value segment that follows the first; all may be column slices of fused projections.  -> [B,Tq,H*D].
    `slots1`: the first segment occupies slots1 >= Tk1 key positions of which only the Tk1 given ones are visible (a prefix mask: the
    masked keys keep their place in the kv blocks, see include/selftok_hip.h); k1 = v1 = None with slots1: none of them is visible.
    `kernel`: 'fused' = one kernel, scores never leave the CU (round 6: head_dim 64, slot counts % 64 == 0), 'unfused' = scores GEMM -> row pass -> P V GEMM through a
    workspace (round 5, any shape), 'auto' = fused where it applies.  Same bits (tests/test_encoder_exact_gpu.py)."""
    if kernel == "auto":
        kernel = EX_ATTENTION_DEFAULT
    if kernel not in ("auto", "fused", "unfused"):
        raise ValueError(f"ex_attention kernel {kernel!r}: expected 'auto', 'fused' or 'unfused'")
    _need_cuda(q, k1, v1, k2, v2)
    B, Tq, HD = q.shape
    D = HD // heads
    _, qs = _rows2d(q)
    rows1, ks1 = 0, 0
    if k1 is not None:
        _, ks1 = _rows2d(k1)
        assert _rows2d(v1)[1] == ks1 and k1.shape == v1.shape
        rows1 = k1.shape[1]
    else:
        assert slots1 is not None and k2 is not None
    Tk1 = rows1 if slots1 is None else int(slots1)
    Tk2, ks2 = 0, 0
    if k2 is not None:
        Tk2, ks2 = k2.shape[1], _rows2d(k2)[1]
        assert _rows2d(v2)[1] == ks2 and k2.shape == v2.shape
    lib = _lib.load()
    out = torch.empty(B, Tq, HD, dtype=torch.float32, device=q.device)
    if B == 0:
        return out
    if kernel == "fused" or (kernel == "auto" and lib.selftok_ex_attention_fused_supported(Tk1, Tk2, D)):
        _lib.check(lib.selftok_ex_attention_fused_f32(_p(q), qs, _p(k1), _p(v1), ks1, Tk1, rows1, rows1, _p(k2), _p(v2), ks2, Tk2, _p(out), B, heads, Tq, D, _stream()),
                   "selftok_ex_attention_fused_f32")
        return out
    # the unfused route materialises the scores (B * heads * Tq * Tk * 4 bytes + V^T): bounded by running the batch in slices (rows are independent: same bits),
    # which also keeps B * heads inside gridDim.z
    per = int(lib.selftok_ex_attention_workspace_bytes(1, heads, Tq, Tk1 + Tk2, D))
    nb = max(1, min(B, EX_ATTENTION_WS_LIMIT // max(per, 1), 65535 // max(heads, 1)))
    ws = torch.empty(int(lib.selftok_ex_attention_workspace_bytes(nb, heads, Tq, Tk1 + Tk2, D)), dtype=torch.uint8, device=q.device)
    for b0 in range(0, B, nb):
        n = min(nb, B - b0)
        sl = lambda t: None if t is None else t[b0:b0 + n]
        _lib.check(lib.selftok_ex_attention_f32(_p(sl(q)), qs, _p(sl(k1)), _p(sl(v1)), ks1, Tk1, rows1, rows1, _p(sl(k2)), _p(sl(v2)), ks2, Tk2, _p(out[b0:b0 + n]), _p(ws),
                                                n, heads, Tq, D, _stream()), "selftok_ex_attention_f32")
    return out
