// bf16 convolutions and GroupNorm of the SD3 VAE, channels-last, on the bf16 matrix cores  (gfx950)
// ---------------------------------------------------------------------------------------------------------------------------------
// Reference: the VAE the pipeline calls at SelftokPipeline.py:215 (encode) and :288 (decode) -- ResnetBlock / Downsample / Upsample /
// conv_in / conv_out of sd3/sd3_impls.py:228-262, 286-318, 340-456 run in bf16 on the CPU: each convolution accumulates in fp32,
// adds the bias inside that accumulation and rounds ONCE to bf16 (oneDNN).  Rounds 1-3 reached that arithmetic through MIOpen's GEMM
// algorithm (im2col + GEMM per image, bias folded in as 8 extra input channels): exact, bit-stable, but 0.18 PFLOP/s; MIOpen's fast
// solvers (implicit GEMM, Winograd) are 7e-3 dB off on this network.  This file is the same arithmetic as ONE
// implicit-GEMM kernel:
//
//   out[b, oy, ox, co] = bf16( bias[co] + sum_{dy,dx,ci} x[b, oy*S - P + dy, ox*S - P + dx, ci] * w[co, ci, dy, dx] )      fp32 sum
//
//   * NHWC activations ([B, H, W, C], C % 8 == 0): the contraction index (ci) is contiguous for both operands, so a 16-byte LDS read
//     is one MFMA operand (8 bf16) with no transposition anywhere;
//   * a workgroup (8 waves; 4 for the narrow / stride-2 shapes) owns TH x 32 output pixels x BN output channels.  Per block of 32 input channels it stages the input halo
//     tile ((TH-1)S+3) x (31S+3) pixels x 64 B ONCE and runs all 9 taps from it (tap = an address offset into the tile): global/L2 reads
//     of the activations are 1/9 of the matrix-core operand reads; weights arrive as one contiguous 64 B x BN slab per (channel block,
//     tap) from a pre-packed image.  Out-of-image pixels are zero-filled while staging (padding), the optional nearest 2x upsample
//     of Upsample (sd3_impls.py:300-306) is an address shift while staging (the 4x larger tensor is never written);
//   * LDS rows are 64 B (32 channels) with the 16-byte chunk index XOR-ed by (row >> 2) & 3: every ds_read_b128 of 32 consecutive
//     output channels (weight slabs) is conflict-free; halo-tile rows are 80 B instead (64 B + 16 B pad): any 32 consecutive pixels hit
//     64 distinct banks, and a tap becomes a compile-time byte offset (stride-2 taps are 2-way);
//   * staging goes through registers with the loads issued one kernel row (or two slabs) ahead and NEVER followed by a select on the
//     loaded value (that would park the wave on the load in the slab that issued it): out-of-range chunks load from a clamped address
//     and are zeroed when written to LDS.  What moved the kernel was co-residency: 78 VGPRs and 41 KB of LDS (one halo tile, ONE weight
//     buffer refilled between two barriers per kernel row) let three 8-wave workgroups share a CU, so one's prologue, epilogue and
//     barriers hide behind the others' matrix work (conv3x3_rows_kernel<1>; the per-tap conv_nhwc_bf16_kernel serves 1x1, stride-2
//     and narrow-output layers);
//   * v_mfma_f32_32x32x16_bf16 with the WEIGHTS as the row operand: a lane then holds 4 consecutive output channels of one pixel per
//     accumulator quad, i.e. 8-byte NHWC stores straight from the accumulators;
//   * epilogue: + bias in fp32, one rounding to bf16; optional residual `x + h` (ResnetBlock.forward :262) with the reference's second
//     rounding: bf16(bf16(acc + bias) + res).
// GroupNorm(32 groups) + SiLU for the same layout: statistics in fp64 (sum and sum of squares of bf16 values are exact to 1e-16, so
// mean / variance are the correctly rounded fp32 values regardless of the summation order: deterministic, partials reduced in a fixed
// order), applied with the arithmetic of groupnorm_silu_bf16_kernel (vae.hip).
#include "common.h"
#include "selftok_hip.h"
#include <stdlib.h>
#include <type_traits>

namespace selftok {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
    const uint16_t* x;      // [B, H, W, Cin]
    const uint16_t* w;      // packed: [nblk][ncb][taps][BN][32]
    const uint16_t* bias;   // [Cout] bf16 or null
    const uint16_t* res;    // [B, Ho, Wo, ldo] or null
    uint16_t* out;          // [B, Ho, Wo, ldo]
    int B, H, W, Cin;       // stored input (before the optional upsample)
    int Ho, Wo, Cout, Cs, ldo;   // Cout: real output channels (bias length); Cs: channels stored (multiple of 4, <= ldo)
    int ncb, up, pad;
};

__device__ __forceinline__ float bf2f(uint16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ bf16x8 as_bf8(uint4 v)
{
    union { uint4 u; bf16x8 b; } c;
    c.u = v;
    return c.b;
}

// epilogue: the lane holds pixel (oy0 + mt, ox) of its two tile rows; rows of the MFMA tile = output channels n0 + nt 32 + 8 rg + 4 g + e
template <int NT>
__device__ __forceinline__ void conv_epilogue(const ConvParams& P, const f32x16 (&acc)[2][NT], int b, int oy0, int ox, int n0, int g)
{
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int oy = oy0 + mt;
        if (oy >= P.Ho || ox >= P.Wo) continue;
        const size_t pix = ((size_t)b * P.Ho + oy) * P.Wo + ox;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int co = n0 + nt * 32 + 8 * rg + 4 * g;
                if (co >= P.Cs) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[mt][nt][rg * 4 + e];
                    if (P.bias && co + e < P.Cout) v[e] += bf2f(P.bias[co + e]);
                    v[e] = rbf(v[e]);
                }
                if (P.res) {
                    const uint2 rr = *reinterpret_cast<const uint2*>(P.res + pix * P.ldo + co);
                    v[0] = rbf(v[0] + bf2f((uint16_t)(rr.x & 0xFFFF))); v[1] = rbf(v[1] + bf2f((uint16_t)(rr.x >> 16)));
                    v[2] = rbf(v[2] + bf2f((uint16_t)(rr.y & 0xFFFF))); v[3] = rbf(v[3] + bf2f((uint16_t)(rr.y >> 16)));
                }
                uint2 o;
                o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
                o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
                *reinterpret_cast<uint2*>(P.out + pix * P.ldo + co) = o;
            }
        }
    }
}

// two halo buffers when tile(s) + two weight slabs fit in 72 KB (two workgroups per CU), else one
constexpr int conv_abuf(int npix, int bn) { return (npix * 80 * 2 + bn * 64 * 2 <= 73728) ? 2 : 1; }
constexpr int conv_lds_bytes(int npix, int bn) { return conv_abuf(npix, bn) * npix * 80 + bn * 64 * 2; }

template <int WM, int WN, int NT, int TAPS, int STRIDE>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_nhwc_bf16_kernel(const ConvParams P)
{
    constexpr int NTH = WM * WN * 64;                          // 4 or 8 waves
    constexpr int KS = TAPS == 9 ? 3 : 1;
    constexpr int TH = 2 * WM, TW = 32;
    constexpr int HH = (TH - 1) * STRIDE + KS, HWD = (TW - 1) * STRIDE + KS;
    constexpr int NPIX = HH * HWD;
    constexpr int NA = (NPIX * 4 + NTH - 1) / NTH;
    constexpr int BN = WN * NT * 32;
    constexpr int NB = (BN * 4 + NTH - 1) / NTH;
    constexpr int ABUF = conv_abuf(NPIX, BN);
    static_assert(WM * WN == 4 || WM * WN == 8, "four or eight waves");
    // LDS (dynamic: up to 70 KB): halo tile(s) with 80-byte pixel rows (64 B of channels + 16 B pad: any 32 consecutive pixels hit 64
    // distinct banks with ds_read_b128, and a tap is a compile-time byte offset), then two weight slabs with 64-byte rows and the
    // 16-byte chunk index XOR-ed by (row >> 2) & 3
    extern __shared__ uint4 smem[];
    uint4* const s_a0 = smem;
    uint4* const s_b0 = smem + ABUF * NPIX * 5;

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wm = wv % WM, wn = wv / WM;
    const int l32 = lane & 31, g = lane >> 5;
    const int tiles_x = (P.Wo + TW - 1) / TW, tiles_y = (P.Ho + TH - 1) / TH;
    // raster order over (tile_x, tile_y, sample).  Giving each XCD a contiguous band of tiles (so that neighbouring tiles meet in one L2)
    // was measured and is SLOWER: 2.05 vs 1.68 ms on the 128-channel 256 x 256 layer (profiles/r3_conv_launch_shapes.txt) -- the eight
    // bands start at power-of-two strides and collide on HBM channels; halo rows are 1.6x of a small input stream anyway.
    int tile = blockIdx.x;
    const int tx0 = (tile % tiles_x) * TW; tile /= tiles_x;
    const int ty0 = (tile % tiles_y) * TH;
    const int b = tile / tiles_y;
    const int Hi = P.H << P.up, Wi = P.W << P.up;
    const uint16_t* __restrict__ xb = P.x + (size_t)b * P.H * P.W * P.Cin;
    const uint16_t* __restrict__ wb = P.w + (size_t)blockIdx.y * P.ncb * TAPS * BN * 32;

    const int Cin = P.Cin, Wsrc = P.W, up = P.up, ncb = P.ncb, nslab = P.ncb * TAPS;
    const int iy0 = ty0 * STRIDE - P.pad, ix0 = tx0 * STRIDE - P.pad;

    // ---- staging: chunk q = t + NTH i of the halo tile (pixel q / 4, 16-byte channel chunk q % 4); recomputed where used, not kept.
    // A load NEVER selects on its result (a select right after the load would make the wave wait for it in the slab that issued it and
    // collapse the prefetch distance): out-of-image / out-of-range chunks load from a clamped, valid address and are zeroed when the
    // value is written to LDS, slabs later.
    auto a_ok = [&](int cb, int i, int& off) -> bool {
        const int q = t + NTH * i;
        const int pp = q >> 2, c8 = (q & 3) * 8;
        const int hy = pp / HWD, hx = pp - hy * HWD;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool ok = (q < NPIX * 4) && iy >= 0 && iy < Hi && ix >= 0 && ix < Wi && cb * 32 + c8 < Cin;
        const int cy = min(max(iy, 0), Hi - 1) >> up, cx = min(max(ix, 0), Wi - 1) >> up;
        off = (cy * Wsrc + cx) * Cin + (cb * 32 + c8 < Cin ? cb * 32 + c8 : 0);
        return ok;
    };
    auto load_a1 = [&](int cb, int i) -> uint4 {
        int off;
        a_ok(cb, i, off);
        return *reinterpret_cast<const uint4*>(xb + off);
    };
    auto store_a1 = [&](uint4* __restrict__ dst, int cb, int i, uint4 v) {
        const int q = t + NTH * i;
        const int pp = q >> 2, c = q & 3;
        int off;
        if (!a_ok(cb, i, off)) v = make_uint4(0, 0, 0, 0);
        if (q < NPIX * 4) dst[pp * 5 + c] = v;
    };
    auto load_b1 = [&](int slab, int i) -> uint4 {
        const int q = min(t + NTH * i, BN * 4 - 1);
        return reinterpret_cast<const uint4*>(wb + (size_t)min(slab, nslab - 1) * BN * 32)[q];
    };
    auto store_b1 = [&](uint4* __restrict__ dst, int i, uint4 v) {
        const int q = t + NTH * i, row = q >> 2, c = q & 3;
        if (q < BN * 4) dst[row * 4 + (c ^ ((row >> 2) & 3))] = v;
    };

    // ---- operand addresses ----
    int w_off[NT];                                             // weights: row operand
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int r = (wn * NT + nt) * 32 + l32;
        w_off[nt] = r * 4 + (g ^ ((r >> 2) & 3));
    }
    int p_base[2];                                             // pixels: column operand, uint4 index of tap (0, 0), first k-group
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) p_base[mt] = (((wm * 2 + mt) * STRIDE) * HWD + l32 * STRIDE) * 5 + g;

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // Prefetch distances.  Weight slabs (one per (channel block, tap): 64 B x BN) are loaded TWO slabs ahead into registers and
    // written to the other LDS buffer ONE slab ahead; a slab is 8 - 16 MFMAs per wave (0.1 - 0.2 us), and with a single slab of
    // distance the first version spent 3000 cycles per slab for 256 cycles of matrix work (L2 latency exposed).  The halo tile of
    // the next channel block is fetched in two halves, at taps 0 and 4, and written three taps later (12 instead of 24 staging
    // registers).  Tap and slab parity are compile-time constants (integral_constant), so every staging register has a static name:
    // at slab s the set named by parity s & 1 holds slab s + 1 (written to LDS at the end), the other set receives slab s + 2.
    constexpr int NAH = (NA + 1) / 2;                           // chunks per half
    // 1 x 1 convolutions have a new halo tile every slab: with one chunk per thread it is pipelined like the weights (loaded two slabs
    // ahead into the set the slab parity names, written one slab ahead)
    constexpr bool PIPE1 = TAPS == 1 && ABUF == 2 && NA == 1;
    uint4 ra[NAH], ra2[NAH], rb0[NB], rb1[NB];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < NAH; ++i) if (h * NAH + i < NA) ra[i] = load_a1(0, h * NAH + i);
#pragma unroll
        for (int i = 0; i < NAH; ++i) if (h * NAH + i < NA) store_a1(s_a0, 0, h * NAH + i, ra[i]);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) rb0[i] = load_b1(0, i);
#pragma unroll
    for (int i = 0; i < NB; ++i) store_b1(s_b0, i, rb0[i]);
#pragma unroll
    for (int i = 0; i < NB; ++i) { rb0[i] = load_b1(1, i); rb1[i] = rb0[i]; }
#pragma unroll
    for (int i = 0; i < NAH; ++i) { ra2[i] = ra[i]; if (PIPE1) ra[i] = load_a1(1, i); }
    __syncthreads();

    int slab = 0;
    for (int cb = 0; cb < ncb; ++cb) {
        const bool more_cb = cb + 1 < ncb;
        uint4* __restrict__ sa_next = s_a0 + (ABUF == 2 ? ((cb + 1) & 1) : 0) * NPIX * 5;
        const uint4* __restrict__ sa = s_a0 + (ABUF == 2 ? (cb & 1) : 0) * NPIX * 5;
        auto one_tap = [&](auto tap_c, auto par_c) {
            constexpr int tap = decltype(tap_c)::value, par = decltype(par_c)::value;       // par = slab & 1
#pragma unroll
            for (int i = 0; i < NB; ++i) { if (par == 0) rb1[i] = load_b1(slab + 2, i); else rb0[i] = load_b1(slab + 2, i); }      // clamped past the end
            if (PIPE1) {
                if (par == 0) ra2[0] = load_a1(cb + 2, 0); else ra[0] = load_a1(cb + 2, 0);
            }
            if (ABUF == 2 && TAPS == 9 && (tap == 0 || tap == 4)) {          // unconditional (address clamped): no select on the result
#pragma unroll
                for (int i = 0; i < NAH; ++i) if ((tap / 4) * NAH + i < NA) ra[i] = load_a1(cb + 1, (tap / 4) * NAH + i);
            }
            const uint4* __restrict__ sb = s_b0 + par * BN * 4;
            constexpr int dy = tap / KS, dx = tap - dy * KS;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bf16x8 wf[NT], pf[2];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) wf[nt] = as_bf8(sb[w_off[nt] ^ (2 * j)]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) pf[mt] = as_bf8(sa[p_base[mt] + (dy * HWD + dx) * 5 + 2 * j]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], pf[mt], acc[mt][nt], 0, 0, 0);
            }
            if (ABUF == 2 && TAPS == 9 && (tap == 3 || tap == 7) && more_cb) {
#pragma unroll
                for (int i = 0; i < NAH; ++i) if ((tap / 4) * NAH + i < NA) store_a1(sa_next, cb + 1, (tap / 4) * NAH + i, ra[i]);
            }
            if (PIPE1 && more_cb) store_a1(sa_next, cb + 1, 0, par == 0 ? ra[0] : ra2[0]);
            if ((ABUF == 1 || TAPS == 1) && !PIPE1 && tap == TAPS - 1 && more_cb) {       // single halo buffer (stride 2) / wide 1x1 tiles: load and write here
                if (ABUF == 1) __syncthreads();
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int i = 0; i < NAH; ++i) if (h * NAH + i < NA) ra[i] = load_a1(cb + 1, h * NAH + i);
#pragma unroll
                    for (int i = 0; i < NAH; ++i) if (h * NAH + i < NA) store_a1(sa_next, cb + 1, h * NAH + i, ra[i]);
                }
            }
            if (slab + 1 < nslab) {
#pragma unroll
                for (int i = 0; i < NB; ++i) store_b1(s_b0 + (par ^ 1) * BN * 4, i, par == 0 ? rb0[i] : rb1[i]);
            }
            ++slab;
            __syncthreads();
        };
#define SELFTOK_TAP(T, CBPAR) one_tap(std::integral_constant<int, T>{}, std::integral_constant<int, ((CBPAR) + T) & 1>{})
        if constexpr (TAPS == 9) {                               // slab = 9 cb + tap: its parity is (cb + tap) & 1
            if ((cb & 1) == 0) { SELFTOK_TAP(0, 0); SELFTOK_TAP(1, 0); SELFTOK_TAP(2, 0); SELFTOK_TAP(3, 0); SELFTOK_TAP(4, 0); SELFTOK_TAP(5, 0); SELFTOK_TAP(6, 0); SELFTOK_TAP(7, 0); SELFTOK_TAP(8, 0); }
            else { SELFTOK_TAP(0, 1); SELFTOK_TAP(1, 1); SELFTOK_TAP(2, 1); SELFTOK_TAP(3, 1); SELFTOK_TAP(4, 1); SELFTOK_TAP(5, 1); SELFTOK_TAP(6, 1); SELFTOK_TAP(7, 1); SELFTOK_TAP(8, 1); }
        } else {
            if ((cb & 1) == 0) SELFTOK_TAP(0, 0); else SELFTOK_TAP(0, 1);
        }
#undef SELFTOK_TAP
    }

    conv_epilogue<NT>(P, acc, b, ty0 + wm * 2, tx0 + l32, blockIdx.y * BN + wn * NT * 32, g);
}

// 3 x 3, stride 1, 128 output channels per workgroup -- the shape of 95 % of the VAE's convolution work -- with ONE barrier per kernel
// ROW (3 taps, 12 MFMAs per wave) instead of one per tap: 8 waves, 4 tile rows x 32 pixels x 128 channels per workgroup (64 x 32 per
// wave).  LDS: one halo tile (16 KB, 80-byte pixel rows) + two weight groups of 3 slabs (2 x 24 KB) = 64 KB static, two workgroups per
// CU.  Weight group r + 1 sits in registers while group r is multiplied (loaded a whole group = 12 MFMAs x 4 waves per SIMD earlier) and is
// written to the other buffer right after the barrier; the next channel block's halo tile is loaded during kernel row 0 and written
// after the barrier that ends kernel row 2 (+1 barrier per channel block: 4 instead of 9).  Same products, same order per output as
// conv_nhwc_bf16_kernel: bit-identical results.
template <int NBUF>
__global__ __launch_bounds__(512, NBUF == 2 ? 2 : 6) void conv3x3_rows_kernel(const ConvParams P)
{
    constexpr int NTH = 512, WM = 2, TH = 4, TW = 32, HWD = 34, NPIX = 6 * 34, BN = 128;
    constexpr int NA = (NPIX * 4 + NTH - 1) / NTH;             // 2 halo chunks per thread
    constexpr int NBG = 3 * BN * 4 / NTH;                      // 3 weight chunks per thread and group
    __shared__ uint4 s_a[NPIX * 5];
    __shared__ uint4 s_b[NBUF][3 * BN * 4];

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, wm = wv % WM, wn = wv / WM;
    const int l32 = lane & 31, g = lane >> 5;
    const int tiles_x = (P.Wo + TW - 1) / TW, tiles_y = (P.Ho + TH - 1) / TH;
    int tile = blockIdx.x;
    const int tx0 = (tile % tiles_x) * TW; tile /= tiles_x;
    const int ty0 = (tile % tiles_y) * TH;
    const int b = tile / tiles_y;
    const int Hi = P.H << P.up, Wi = P.W << P.up;
    const uint16_t* __restrict__ xb = P.x + (size_t)b * P.H * P.W * P.Cin;
    const uint16_t* __restrict__ wb = P.w + (size_t)blockIdx.y * P.ncb * 9 * BN * 32;
    const int Cin = P.Cin, Wsrc = P.W, up = P.up, ncb = P.ncb, ngrp = P.ncb * 3;
    const int iy0 = ty0 - 1, ix0 = tx0 - 1;

    auto a_ok = [&](int cb, int i, int& off) -> bool {
        const int q = t + NTH * i;
        const int pp = q >> 2, c8 = (q & 3) * 8;
        const int hy = pp / HWD, hx = pp - hy * HWD;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool ok = (q < NPIX * 4) && iy >= 0 && iy < Hi && ix >= 0 && ix < Wi && cb * 32 + c8 < Cin;
        const int cy = min(max(iy, 0), Hi - 1) >> up, cx = min(max(ix, 0), Wi - 1) >> up;
        off = (cy * Wsrc + cx) * Cin + (cb * 32 + c8 < Cin ? cb * 32 + c8 : 0);
        return ok;
    };
    auto load_a1 = [&](int cb, int i) -> uint4 {               // never selects on its result (see conv_nhwc_bf16_kernel)
        int off;
        a_ok(cb, i, off);
        return *reinterpret_cast<const uint4*>(xb + off);
    };
    auto store_a1 = [&](int cb, int i, uint4 v) {
        const int q = t + NTH * i;
        const int pp = q >> 2, c = q & 3;
        int off;
        if (!a_ok(cb, i, off)) v = make_uint4(0, 0, 0, 0);
        if (q < NPIX * 4) s_a[pp * 5 + c] = v;
    };
    auto load_bg = [&](int grp, int i) -> uint4 {              // group = 3 consecutive slabs of the packed image, 1536 chunks
        return reinterpret_cast<const uint4*>(wb + (size_t)min(grp, ngrp - 1) * 3 * BN * 32)[t + NTH * i];
    };
    auto store_bg = [&](uint4* __restrict__ dst, int i, uint4 v) {
        const int q = t + NTH * i, row = q >> 2, c = q & 3;     // row = slab-in-group * 128 + channel
        dst[row * 4 + (c ^ ((row >> 2) & 3))] = v;
    };

    const int w_row = wn * 32 + l32;
    const int w_off = w_row * 4 + (g ^ ((w_row >> 2) & 3));
    int p_base[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) p_base[mt] = ((wm * 2 + mt) * HWD + l32) * 5 + g;

    f32x16 acc[2][1];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][0][r] = 0.f;

    uint4 ra[NA], rb[NBG];
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = load_a1(0, i);
#pragma unroll
    for (int i = 0; i < NBG; ++i) rb[i] = load_bg(0, i);
#pragma unroll
    for (int i = 0; i < NA; ++i) store_a1(0, i, ra[i]);
#pragma unroll
    for (int i = 0; i < NBG; ++i) store_bg(s_b[0], i, rb[i]);
#pragma unroll
    for (int i = 0; i < NBG; ++i) rb[i] = load_bg(1, i);        // rb <- group 1
    __syncthreads();

    int grp = 0;
    for (int cb = 0; cb < ncb; ++cb) {
        const bool more_cb = cb + 1 < ncb;
        auto one_row = [&](auto dy_c) {
            constexpr int dy = decltype(dy_c)::value;
            if (NBUF == 2) {
                // rb holds group grp + 1: write it to the buffer group grp - 1 was read from (everyone is past that barrier), then refill rb
                if (grp + 1 < ngrp) {
#pragma unroll
                    for (int i = 0; i < NBG; ++i) store_bg(s_b[(grp + 1) & 1], i, rb[i]);
                }
#pragma unroll
                for (int i = 0; i < NBG; ++i) rb[i] = load_bg(grp + 2, i);
            }
            if (dy == 0) {
#pragma unroll
                for (int i = 0; i < NA; ++i) ra[i] = load_a1(cb + 1, i);
            }
            const uint4* __restrict__ sb = s_b[NBUF == 2 ? (grp & 1) : 0];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bf16x8 wf = as_bf8(sb[(dx * BN * 4 + w_off) ^ (2 * j)]);
                    bf16x8 pf[2];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) pf[mt] = as_bf8(s_a[p_base[mt] + (dy * HWD + dx) * 5 + 2 * j]);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, pf[mt], acc[mt][0], 0, 0, 0);
                }
            }
            ++grp;
            __syncthreads();
            if (NBUF == 1) {
                // ONE weight buffer (41 KB of LDS per workgroup -> three workgroups per CU): everyone is done reading group grp - 1 and, at
                // dy == 2, the halo tile; refill both from registers, fetch the group after, and meet again
                if (grp < ngrp) {
#pragma unroll
                    for (int i = 0; i < NBG; ++i) store_bg(s_b[0], i, rb[i]);
                }
                if (dy == 2 && more_cb) {
#pragma unroll
                    for (int i = 0; i < NA; ++i) store_a1(cb + 1, i, ra[i]);
                }
#pragma unroll
                for (int i = 0; i < NBG; ++i) rb[i] = load_bg(grp + 1, i);
                __syncthreads();
            } else if (dy == 2 && more_cb) {                    // every wave is done with this channel block's halo tile
#pragma unroll
                for (int i = 0; i < NA; ++i) store_a1(cb + 1, i, ra[i]);
                __syncthreads();
            }
        };
        one_row(std::integral_constant<int, 0>{});
        one_row(std::integral_constant<int, 1>{});
        one_row(std::integral_constant<int, 2>{});
    }
    conv_epilogue<1>(P, acc, b, ty0 + wm * 2, tx0 + l32, blockIdx.y * BN + wn * 32, g);
}

// w [O, Cin, ks, ks] bf16 (the checkpoint's layout) -> [nblk][ncb][taps][BN][32], zero padded
__global__ __launch_bounds__(256) void conv_pack_weight_kernel(const uint16_t* __restrict__ w, uint16_t* __restrict__ packed, int O, int Cin, int ks, int bn, int ncb, long total)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int taps = ks * ks;
    const int ci = i & 31;
    long r = i >> 5;
    const int row = r % bn; r /= bn;
    const int tap = r % taps; r /= taps;
    const int cb = r % ncb;
    const int nb = r / ncb;
    const int co = nb * bn + row, c = cb * 32 + ci;
    packed[i] = (co < O && c < Cin) ? w[((size_t)co * Cin + c) * taps + tap] : (uint16_t)0;
}

// ---- GroupNorm (+ SiLU), NHWC ---------------------------------------------------------------------------------------------------
// pass 1: per (sample, pixel range) partial sums of x and x^2 per 4-channel quad, in fp64; a 16-byte chunk holds two quads.
// pixel ranges per sample: enough workgroups to fill the chip on the small feature maps (32 x 32: 4 per sample), at most 32
__host__ __device__ inline int gn_nblk(int HW) { const int n = HW / 256; return n < 1 ? 1 : (n > 32 ? 32 : n); }
__global__ __launch_bounds__(256) void gn_nhwc_partial_kernel(const uint16_t* __restrict__ x, double* __restrict__ part, int HW, int C, int nblk)
{
    const int GN_PIX_PER_BLOCK = (HW + nblk - 1) / nblk;
    __shared__ double red[256][4];
    const int cpp = C >> 3;                                // 16-byte chunks per pixel (C <= 2048)
    const int b = blockIdx.y, blk = blockIdx.x;
    const int p0 = blk * GN_PIX_PER_BLOCK, p1 = min(HW, p0 + GN_PIX_PER_BLOCK);
    const int chunk = threadIdx.x % cpp, slot = threadIdx.x / cpp, nslot = 256 / cpp;
    double s0 = 0, q0 = 0, s1 = 0, q1 = 0;
    if (slot < nslot) {
        const uint4* xp = reinterpret_cast<const uint4*>(x + (size_t)b * HW * C);
#pragma unroll 4
        for (int p = p0 + slot; p < p1; p += nslot) {
            const uint4 v = xp[(size_t)p * cpp + chunk];
            const float e[8] = {bf2f((uint16_t)(v.x & 0xFFFF)), bf2f((uint16_t)(v.x >> 16)), bf2f((uint16_t)(v.y & 0xFFFF)), bf2f((uint16_t)(v.y >> 16)),
                                bf2f((uint16_t)(v.z & 0xFFFF)), bf2f((uint16_t)(v.z >> 16)), bf2f((uint16_t)(v.w & 0xFFFF)), bf2f((uint16_t)(v.w >> 16))};
            // bf16 values: sums of 4 and of 4 squares are exact in fp32 only up to rounding of the adds -> go through fp64 per element
#pragma unroll
            for (int k = 0; k < 4; ++k) { const double d = e[k]; s0 += d; q0 = __builtin_fma(d, d, q0); }
#pragma unroll
            for (int k = 4; k < 8; ++k) { const double d = e[k]; s1 += d; q1 = __builtin_fma(d, d, q1); }
        }
    }
    red[threadIdx.x][0] = s0; red[threadIdx.x][1] = q0; red[threadIdx.x][2] = s1; red[threadIdx.x][3] = q1;
    __syncthreads();
    // quad qd (C/4 of them) = chunk qd/2, half qd&1: summed over the slots in a fixed order
    for (int qd = threadIdx.x; qd < (C >> 2); qd += 256) {
        double s = 0, q = 0;
        const int ch = qd >> 1, hf = (qd & 1) * 2;
        for (int sl = 0; sl < nslot; ++sl) { s += red[sl * cpp + ch][hf]; q += red[sl * cpp + ch][hf + 1]; }
        double* o = part + (((size_t)b * nblk + blk) * (C >> 2) + qd) * 2;
        o[0] = s; o[1] = q;
    }
}

// pass 2: stats[b][g] = (mean, rstd) in fp32 from the fp64 sums
__global__ void gn_nhwc_finalize_kernel(const double* __restrict__ part, float2* __restrict__ stats, int HW, int C, int groups, int nblk, float eps)
{
    const int b = blockIdx.x, gi = threadIdx.x;
    if (gi >= groups) return;
    const int cpg = C / groups, qpg = cpg >> 2;
    double s = 0, q = 0;
    for (int blk = 0; blk < nblk; ++blk)
        for (int k = 0; k < qpg; ++k) {
            const double* o = part + (((size_t)b * nblk + blk) * (C >> 2) + gi * qpg + k) * 2;
            s += o[0]; q += o[1];
        }
    const double n = (double)HW * cpg;
    const double mean = s / n;
    const double var = fmax(q / n - mean * mean, 0.0);
    const float meanf = (float)mean;
    // the reference (and groupnorm_silu_bf16_kernel) centre on the fp32 mean: var about meanf = var + (mean - meanf)^2
    const float varf = (float)(var + (mean - (double)meanf) * (mean - (double)meanf));
    stats[b * groups + gi] = make_float2(meanf, 1.0f / __builtin_sqrtf(varf + eps));
}

// pass 3: y = bf16((x - mean) * rstd * w + b) [SiLU on the bf16 value] -- the arithmetic of vae.hip.  256 % (C / 8) == 0, so a thread
// meets the same 8 channels in every iteration of its grid-stride loop: their statistics and affine parameters stay in registers.
__global__ __launch_bounds__(256) void gn_nhwc_apply_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, const uint16_t* __restrict__ bias,
                                                            const float2* __restrict__ stats, uint16_t* __restrict__ out, int C, int groups, int apply_silu, long chunks)
{
    const int cpp = C >> 3, cpg = C / groups;
    const int b = blockIdx.y, chunk = threadIdx.x % cpp;
    float mean[8], rstd[8], ww[8], bb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = chunk * 8 + e;
        const float2 st = stats[b * groups + ch / cpg];
        mean[e] = st.x; rstd[e] = st.y; ww[e] = bf2f(w[ch]); bb[e] = bf2f(bias[ch]);
    }
    const uint4* __restrict__ xp = reinterpret_cast<const uint4*>(x) + (size_t)b * chunks;
    uint4* __restrict__ op = reinterpret_cast<uint4*>(out) + (size_t)b * chunks;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (long)gridDim.x * 256) {
        const uint4 v = xp[i];
        const uint32_t in[4] = {v.x, v.y, v.z, v.w};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint16_t r2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 2 * k + h;
                float y = rbf((bf2f((uint16_t)(in[k] >> (16 * h))) - mean[e]) * rstd[e] * ww[e] + bb[e]);
                if (apply_silu) y = y / (1.0f + expf(-y));
                r2[h] = f2bf(y);
            }
            o[k] = (uint32_t)r2[0] | ((uint32_t)r2[1] << 16);
        }
        op[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

template <int WM, int WN, int NT, int TAPS, int STRIDE>
int launch_conv_one(const ConvParams& P, hipStream_t stream)
{
    constexpr int KS = TAPS == 9 ? 3 : 1, TH = 2 * WM, BN = WN * NT * 32;
    constexpr int NPIX = ((TH - 1) * STRIDE + KS) * (31 * STRIDE + KS);
    constexpr int LDS = conv_lds_bytes(NPIX, BN);
    static bool attr_set = false;
    if (LDS > 65536 && !attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_nhwc_bf16_kernel<WM, WN, NT, TAPS, STRIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return check_launch("conv_nhwc_bf16_kernel (dynamic LDS attribute)");
        attr_set = true;
    }
    const int tiles = ((P.Wo + 31) / 32) * ((P.Ho + TH - 1) / TH) * P.B;
    hipLaunchKernelGGL((conv_nhwc_bf16_kernel<WM, WN, NT, TAPS, STRIDE>), dim3(tiles, (P.Cs + BN - 1) / BN), dim3(WM * WN * 64), LDS, stream, P);
    return check_launch("conv_nhwc_bf16_kernel");
}

inline int conv_bn(int bn) { return bn == 32 ? 32 : 128; }

}  // namespace
}  // namespace selftok

using namespace selftok;

extern "C" {

size_t selftok_conv2d_packed_bytes(int Cout, int Cin, int ksize, int bn)
{
    if (Cout <= 0 || Cin <= 0 || (ksize != 1 && ksize != 3) || (bn != 32 && bn != 128)) return 0;
    return (size_t)((Cout + bn - 1) / bn) * ((Cin + 31) / 32) * ksize * ksize * bn * 32 * sizeof(uint16_t);
}

int selftok_conv2d_pack_weight_bf16(const void* w, void* packed, int Cout, int Cin, int ksize, int bn, hipStream_t stream)
{
    const size_t bytes = selftok_conv2d_packed_bytes(Cout, Cin, ksize, bn);
    if (!w || !packed || bytes == 0) { set_last_error("conv2d_pack_weight: bad argument (ksize 1|3, bn 32|128)"); return SELFTOK_EINVAL; }
    const long total = (long)(bytes / 2);
    hipLaunchKernelGGL(conv_pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const uint16_t*)w, (uint16_t*)packed, Cout, Cin, ksize, bn,
                       (Cin + 31) / 32, total);
    return check_launch("conv_pack_weight_kernel");
}

int selftok_conv2d_nhwc_bf16(const void* x, const void* packed, const void* bias, const void* residual, void* out, int B, int H, int W, int Cin, int Cout,
                             int Cstore, int ldo, int ksize, int stride, int upsample, int bn, hipStream_t stream)
{
    if (!x || !packed || !out || B < 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 7) || Cout <= 0 || Cstore < Cout || (Cstore & 3) || ldo < Cstore || (ldo & 3) ||
        (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2) || (ksize == 1 && stride != 1) || (upsample != 0 && upsample != 1) || (bn != 32 && bn != 128) ||
        (upsample && stride != 1) || (stride == 2 && bn != 128)) {
        set_last_error("conv2d_nhwc_bf16: bad argument (Cin % 8, Cstore % 4, ldo % 4, ksize 1|3, stride 1|2 (3x3 only), bn 32|128)");
        return SELFTOK_EINVAL;
    }
    if (B == 0) return SELFTOK_OK;
    ConvParams P;
    P.x = (const uint16_t*)x; P.w = (const uint16_t*)packed; P.bias = (const uint16_t*)bias; P.res = (const uint16_t*)residual; P.out = (uint16_t*)out;
    P.B = B; P.H = H; P.W = W; P.Cin = Cin; P.Cout = Cout; P.Cs = Cstore; P.ldo = ldo;
    P.ncb = (Cin + 31) / 32; P.up = upsample;
    const int Hi = H << upsample, Wi = W << upsample;
    if (stride == 2) { P.Ho = Hi / 2; P.Wo = Wi / 2; P.pad = 0; }            // Downsample: F.pad(x, (0,1,0,1)) + conv stride 2 padding 0 (sd3_impls.py:294-297)
    else { P.Ho = Hi; P.Wo = Wi; P.pad = ksize == 3 ? 1 : 0; }
    if ((size_t)H * W * Cin >= 0x7FFFFFFFull) { set_last_error("conv2d_nhwc_bf16: one image exceeds 2^31 elements"); return SELFTOK_EINVAL; }
    const int taps = ksize * ksize;
    if (conv_bn(bn) == 32) return taps == 9 ? launch_conv_one<4, 1, 1, 9, 1>(P, stream) : launch_conv_one<4, 1, 1, 1, 1>(P, stream);
    if (stride == 2) return launch_conv_one<2, 2, 2, 9, 2>(P, stream);           // its halo tile is 4x larger: 128 pixels per workgroup
#ifdef SELFTOK_TUNE
    if (const char* v = getenv("SELFTOK_CONV_VARIANT")) {      // tune build only (tools/build_tune.sh): launch-shape experiments
        if (v[0] == '0') return taps == 9 ? launch_conv_one<4, 2, 2, 9, 1>(P, stream) : launch_conv_one<4, 2, 2, 1, 1>(P, stream);
        if (v[0] == '1') return taps == 9 ? launch_conv_one<2, 2, 2, 9, 1>(P, stream) : launch_conv_one<2, 2, 2, 1, 1>(P, stream);
    }
#endif
    // 8 waves, 128 pixels x 128 channels per workgroup, 64 x 32 per wave: 120 VGPRs and 48 KB of LDS, so TWO workgroups share a CU and
    // one's prologue / epilogue / barriers hide behind the other's matrix work.  Measured against 256 x 128 tiles of 64 x 64 per
    // wave (one workgroup per CU, half the weight traffic per pixel) and 4-wave 128 x 128 tiles: +27 % on the 128-channel layers at
    // 256 x 256, +3 ... 6 % elsewhere (profiles/r3_conv_launch_shapes.txt; the three are bit-identical).
    if (taps == 9) {
        // 3 x 3, stride 1: one barrier pair per kernel row, ONE weight buffer -> 41 KB of LDS and 78 VGPRs, three workgroups per CU.
        // Measured against two weight buffers (two workgroups per CU) and against one barrier per tap: +2 ... 4 % and +3 ... 7 % on every
        // layer shape of the VAE (profiles/r3_conv_launch_shapes.txt); the three are bit-identical.
        const int tiles = ((P.Wo + 31) / 32) * ((P.Ho + 3) / 4) * P.B;
#ifdef SELFTOK_TUNE
        if (const char* v = getenv("SELFTOK_CONV_VARIANT")) {
            if (v[0] == '2') return launch_conv_one<2, 4, 1, 9, 1>(P, stream);
            if (v[0] == '9') { hipLaunchKernelGGL(conv3x3_rows_kernel<2>, dim3(tiles, (P.Cs + 127) / 128), dim3(512), 0, stream, P); return check_launch("conv3x3_rows_kernel<2>"); }
        }
#endif
        hipLaunchKernelGGL(conv3x3_rows_kernel<1>, dim3(tiles, (P.Cs + 127) / 128), dim3(512), 0, stream, P);
        return check_launch("conv3x3_rows_kernel");
    }
    return launch_conv_one<2, 4, 1, 1, 1>(P, stream);
}

size_t selftok_groupnorm_nhwc_workspace_bytes(int B, int HW, int C)
{
    if (B <= 0 || HW <= 0 || C <= 0) return 0;
    return (size_t)B * 32 * (C / 4) * 2 * sizeof(double) + (size_t)B * 64 * sizeof(float2) + 256;      // 32 = the most pixel ranges per sample
}

int selftok_groupnorm_silu_nhwc_bf16(const void* x, const void* weight, const void* bias, void* out, void* workspace, int B, int HW, int C, int groups,
                                     float eps, int apply_silu, hipStream_t stream)
{
    if (!x || !weight || !bias || !out || !workspace || B < 0 || HW <= 0 || groups <= 0 || groups > 64 || C % groups || ((C / groups) & 3) || (C & 7) || C > 2048 ||
        (256 % (C >> 3))) {
        set_last_error("groupnorm_silu_nhwc: need C % groups == 0, (C / groups) % 4 == 0, C / 8 a divisor of 256, groups <= 64");
        return SELFTOK_EINVAL;
    }
    if (B == 0) return SELFTOK_OK;
    const int nblk = gn_nblk(HW);
    double* part = (double*)workspace;
    float2* stats = (float2*)((char*)workspace + (((size_t)B * nblk * (C / 4) * 2 * sizeof(double) + 255) & ~(size_t)255));
    hipLaunchKernelGGL(gn_nhwc_partial_kernel, dim3(nblk, B), dim3(256), 0, stream, (const uint16_t*)x, part, HW, C, nblk);
    hipLaunchKernelGGL(gn_nhwc_finalize_kernel, dim3(B), dim3(64), 0, stream, (const double*)part, stats, HW, C, groups, nblk, eps);
    const long chunks = (long)HW * (C >> 3);                                  // per sample
    long blocks = (chunks + 255) / 256;
    const long cap = (256l * 16 + B - 1) / B;                                 // ~16 workgroups per CU over the whole batch
    if (blocks > cap) blocks = cap < 1 ? 1 : cap;
    hipLaunchKernelGGL(gn_nhwc_apply_kernel, dim3((unsigned)blocks, B), dim3(256), 0, stream, (const uint16_t*)x, (const uint16_t*)weight, (const uint16_t*)bias,
                       (const float2*)stats, (uint16_t*)out, C, groups, apply_silu, chunks);
    return check_launch("groupnorm_silu_nhwc");
}

}  // extern "C"
