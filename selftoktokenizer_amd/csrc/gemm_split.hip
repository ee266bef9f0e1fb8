// fp32-equivalent Linear layer on the gfx950 f16 matrix cores ("f16x2 split"):   out = act(A W^T + bias)
//
// Replaces the weight-bearing nn.Linear calls of the MMDiT decode loop (qkv / proj / fc1 / fc2 of both streams:
// sd3/mmdit.py:291-297, 485-496; sd3/other_impls.py:65-90) which the reference runs as fp32 GEMMs.  gfx950 has no
// TF32/xf32 path and its fp32-input MFMA runs at the fp32 VECTOR rate (157 TF), 1/16 of the f16 matrix rate, so an
// fp32 GEMM is the wall of the whole decode (87 % of a step in round 1).  Here every fp32 operand is split into two
// fp16 values,
//       x = x0 + x1 * 2^-11,      x0 = fp16(x),   x1 = fp16((x - x0) * 2^11)          (22 significand bits)
// and the product keeps the three terms of order <= 1:
//       a.w ~= a0 w0 + 2^-11 (a0 w1 + a1 w0)                                           (dropped: a1 w1 2^-22)
// Each term is an exact fp16 x fp16 product accumulated in fp32 by v_mfma_f32_32x32x16_f16; the high term and the two
// low terms have SEPARATE fp32 accumulators (the low ones live 2^11 larger, so they lose nothing against the large
// high sum) and are combined once in the epilogue.  Measured against an fp64 product (tools/probe_split_gemm2.py,
// tests/test_gemm_gpu.py): rms error 0.37x of hipBLASLt's fp32 GEMM at K=1536 and K=6144, i.e. this is MORE accurate
// than the fp32 library GEMM it replaces, at 3 matrix instructions of the 16x-rate pipe per fp32 one.
//
// Weights are split once at load time into the kernel's own tile order (selftok_linear_f16x2_pack_weight), so a weight
// tile reaches LDS by direct LDS-DMA (global_load_lds, 16 B per lane, no VGPR round trip) as one linear 16 KiB copy.
// Two kernels, bit-identical results:
//   linear_f16x2_kernel      fp32 activations, split in registers while staging to LDS (the general entry point);
//   linear_f16x2_pre_kernel  activations already split by the kernel that produced them ("split activation", common.h):
//                            both operands by LDS-DMA, ping-pong schedule, LDS-transposed epilogue with optional GELU,
//                            split-activation output and fused residual update -- what the decode step runs.
// The chip is power-limited under this instruction mix (1.37 GHz effective shader clock, matrix pipe busy 89 % of the
// k-loop; tools/stamp_gemm_pre.py): the second kernel runs at 0.9-1.05x the rate of the vendor's plain fp16 GEMM with
// the same number of MFMAs, while carrying operands of fp32 precision.
//
// Range: fp16 overflows at 65504.  An |activation| >= 65504 turns the output non-finite, which raises *overflow (device int,
// caller-owned, sticky) and the caller redoes the work with the fp32 library GEMM; weights are checked at pack time.  Values below the fp16 normal
// range are carried by the scaled low part (tests cover 1e-7..1e-3).
//
// Tiling (first kernel; the second differs in staging only, see its comment): workgroup = 8 waves = 256 (M) x 128 (N)
// outputs, K step 32, two activation + five weight LDS stages; each wave owns
// 64 x 64 = 2 x 2 MFMA blocks with hi+lo accumulators (128 VGPRs), 12 MFMAs per 8 ds_read_b128 per 16-deep k-step.
// LDS images are MFMA-fragment ordered [plane][k-group of 8][row][8 halfs]: every fragment read is 512 contiguous
// bytes per half wave (conflict-free); the activation image pads each k-group by 32 B so that the ds_write_b64 of the
// split pass are conflict-free too.  Work-groups are renumbered so that each XCD (own L2) owns a contiguous range of
// tiles, walked in 8-row-block groups, which keeps the A row panels and W column panels of concurrently running
// work-groups in one L2.
#include "common.h"
#include "selftok_hip.h"
#include <stdlib.h>

namespace selftok {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16v __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 128, BK = 32;
constexpr float LO_SCALE = 2048.0f, LO_INV = 1.0f / 2048.0f;
constexpr float F16_MAX = 65504.0f;

constexpr int A_G = BM * 16 + 32;          // bytes between k-groups of the activation image (padded)
constexpr int A_P = 4 * A_G;               // bytes between the hi and lo planes
constexpr int A_BYTES = 2 * A_P;           // 33024
constexpr int W_G = BN * 16;               // weight image: linear (filled by LDS-DMA)
constexpr int W_P = 4 * W_G;
constexpr int W_BYTES = 2 * W_P;           // 16384
constexpr int A_STAGES = 2;                // activation images: tile kt (being multiplied) and tile kt+1 (being written)
constexpr int W_STAGES = 5;                // weight images: tile kt .. kt+3 (three DMAs in flight) + the one freed last
constexpr int W_AHEAD = 3;                 // the weight DMA of tile kt+3 is issued in iteration kt
constexpr int W_BASE = A_STAGES * A_BYTES; // 66048
constexpr int LDS_BYTES = W_BASE + W_STAGES * W_BYTES;   // 147968 of the CU's 163840
constexpr int GROUP_M = 4;                 // 32 consecutive tiles (one XCD's resident set) = 4 row blocks x 8 column blocks

// GELU(tanh): 0.5 x (1 + tanh(u)) = x sigmoid(2u) = x / (1 + exp(-2u)), u = sqrt(2/pi) (x + 0.044715 x^3).  The sigmoid form needs one
// v_exp_f32 + one v_rcp_f32 (1 ulp each) instead of ocml's tanhf (~40 VALU ops: the GELU epilogue was 8 % of the fc1 kernel) and has no
// cancellation in 1 + tanh(u) for negative x; |error| vs the fp64 GELU stays below the tanhf form's (tests/test_gemm_gpu.py).
// -inf -> NaN and +inf -> +inf as the reference formula gives.
__device__ __forceinline__ float gelu_tanh_f(float x)
{
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u = k0 * (x + k1 * x * x * x);
    const float e = __builtin_amdgcn_exp2f(u * (-2.0f * 1.4426950408889634f));
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// four fp32 -> four (hi, lo) fp16 pairs, written with 2-wide vectors so that the residual and its 2^11 scaling are packed
// VALU ops (v_pk_add_f32 / v_pk_mul_f32): a wave64 VALU instruction occupies its SIMD for 8 cycles, and with two waves per SIMD
// the split would otherwise keep the VALU pipe as busy as the matrix pipe.  `mx` (optional) tracks max |x| for the range check.
template <typename V4>
__device__ __forceinline__ void split4(const V4& v, f16x4& hi, f16x4& lo)
{
    // __builtin_convertvector keeps the pair packed: v_cvt_pk_f16_f32, v_cvt_f32_f16 (+ one SDWA form for the upper half),
    // v_pk_add_f32, v_pk_mul_f32, v_cvt_pk_f16_f32 = 3 VALU ops per element (element-wise casts compile to 4.5)
    const f32x2v a = {v.x, v.y}, b = {v.z, v.w};
    const f16x2 ha = __builtin_convertvector(a, f16x2), hb = __builtin_convertvector(b, f16x2);
    const f32x2v ra = (a - __builtin_convertvector(ha, f32x2v)) * LO_SCALE;                        // exact residual, then 2^11
    const f32x2v rb = (b - __builtin_convertvector(hb, f32x2v)) * LO_SCALE;
    const f16x2 la = __builtin_convertvector(ra, f16x2), lb = __builtin_convertvector(rb, f16x2);
    hi[0] = ha[0]; hi[1] = ha[1]; hi[2] = hb[0]; hi[3] = hb[1];
    lo[0] = la[0]; lo[1] = la[1]; lo[2] = lb[0]; lo[3] = lb[1];
}
template <typename V4>
__device__ __forceinline__ void split4(const V4& v, f16x4& hi, f16x4& lo, float& mx)
{
    split4(v, hi, lo);
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
}

// W [N,K] fp32 row-major -> packed[(nb*KT + kt)][plane][g][n][8]  (halfs), nb = n/128, kt = k/32, g = (k%32)/8
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ W, _Float16* __restrict__ packed, int N, int K, int* __restrict__ overflow)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one (n, k-group of 8) per thread
    const int KG = K / 8;
    if (idx >= (long)N * KG) return;
    const int n = (int)(idx / KG), kg = (int)(idx % KG);
    const float4 v0 = *reinterpret_cast<const float4*>(W + (size_t)n * K + kg * 8);
    const float4 v1 = *reinterpret_cast<const float4*>(W + (size_t)n * K + kg * 8 + 4);
    f16x4 h0, l0, h1, l1;
    float mx = 0.f;
    split4(v0, h0, l0, mx);
    split4(v1, h1, l1, mx);
    const int nb = n / BN, nl = n % BN, kt = kg / 4, g = kg % 4, KT = K / BK;
    _Float16* tile = packed + ((size_t)nb * KT + kt) * (W_BYTES / 2);
    f16x8 hi = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    f16x8 lo = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    *reinterpret_cast<f16x8*>(tile + (size_t)g * (W_G / 2) + nl * 8) = hi;
    *reinterpret_cast<f16x8*>(tile + (W_P / 2) + (size_t)g * (W_G / 2) + nl * 8) = lo;
    if (!(mx < F16_MAX) && overflow) atomicOr(overflow, 2);
}

// ABL: ablation mask for tools/microbench (timing only, results wrong): 1 no split/ds_write, 2 no row loads, 4 no weight DMA,
// 8 no MFMA, 16 no fragment reads.  The product library instantiates ABL = 0 only.
//
// Pipeline (one barrier per 32-deep k-tile).  PMC showed the first version bound by memory latency, not by instructions:
// 30 % of the L2 requests miss (every A row panel is re-fetched from the Infinity Cache once per 32-tile round) and a miss
// costs more than one k-iteration, so everything is fetched several iterations ahead:
//   * activation rows (fp32) travel in registers, two sets: the set split+written in iteration kt (tile kt+1) was loaded in
//     iteration kt-2 and is re-issued at once for tile kt+3;
//   * weight tiles travel by LDS-DMA into a 5-deep ring, issued 3 iterations ahead; `s_waitcnt vmcnt(N)` is counted so that
//     only the DMA of tile kt+1 has to have landed at the barrier ending iteration kt (vmcnt retires in issue order).
template <int ACT, int ABL = 0>
__global__ __launch_bounds__(512, 2) void linear_f16x2_kernel(const float* __restrict__ A, long lda, const _Float16* __restrict__ Wp,
                                                              const float* __restrict__ bias, float* __restrict__ out, long ldo,
                                                              int M, int N, int K, int* __restrict__ overflow, int mblocks, int nblocks)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- tile id: XCD-contiguous renumbering (bijective), then grouped-M walk ----
    int mb, nb;
    {
        const int T = gridDim.x, orig = blockIdx.x;
        const int q8 = T >> 3, r8 = T & 7, xcd = orig & 7, idx = orig >> 3;
        const int w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
        const int per_group = GROUP_M * nblocks;
        const int group = w / per_group, first_m = group * GROUP_M;
        const int gsz = (mblocks - first_m) < GROUP_M ? (mblocks - first_m) : GROUP_M;
        const int in = w - group * per_group;
        mb = first_m + in % gsz;
        nb = in / gsz;
    }
    const int m0 = mb * BM, n0 = nb * BN;
    const int KT = K / BK, KL = KT - 1;

    // ---- staging maps ----
    const int a_q = tid & 7, a_r = tid >> 3;                       // float4 column (k = 4q) and row (+64 i)
    const float* a_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row = m0 + a_r + 64 * i;
        row = row < M ? row : M - 1;                               // ragged last row block: re-read the last row
        a_src[i] = A + (size_t)row * lda + 4 * a_q;
    }
    const int a_dst = (a_q >> 1) * A_G + a_r * 16 + (a_q & 1) * 8; // + i*64*16 (+ A_P for the lo plane)
    const _Float16* w_src = Wp + (size_t)nb * KT * (W_BYTES / 2) + (size_t)wave * 512 + lane * 8;   // chunk `wave`; +8 chunks for the second

    f32x4v preA[4], preB[4];                                       // the two register sets of activation rows in flight
    // The row loads are inline asm: hipcc waits vmcnt(0) at the first use of an ordinary load while LDS-DMAs are in flight
    // (it would drain the whole prefetch pipeline every iteration); hidden from it, they are waited for by the counted
    // `wait_rows` below, which names the registers so that no use can be scheduled above it.
    // Tile indices past the end are clamped to the last tile: the tail iterations then stage data nobody reads and the loop
    // body has no conditional code.
    auto load_a = [&](f32x4v (&pre)[4], int kt) {
        kt = kt < KL ? kt : KL;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* p = a_src[i] + (size_t)kt * BK;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pre[i]) : "v"(p) : "memory");
        }
    };
    // VM ops retire in issue order.  Per iteration (phase B): 2 weight DMAs, then 4 row loads.  The set consumed in phase A of
    // iteration kt was issued in iteration kt-2, so the 2 + 4 younger ops of iteration kt-1 may stay in flight.
    auto wait_rows = [&](f32x4v (&pre)[4]) {
        if (ABL & 6) asm volatile("s_waitcnt vmcnt(0)" : "+v"(pre[0]), "+v"(pre[1]), "+v"(pre[2]), "+v"(pre[3])::"memory");
        else asm volatile("s_waitcnt vmcnt(6)" : "+v"(pre[0]), "+v"(pre[1]), "+v"(pre[2]), "+v"(pre[3])::"memory");
    };
    auto dma_w = [&](int kt, int stage) {
        kt = kt < KL ? kt : KL;
        const _Float16* src = w_src + (size_t)kt * (W_BYTES / 2);
        unsigned char* dst = smem + W_BASE + stage * W_BYTES + wave * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 8 * 512),
                                         (__attribute__((address_space(3))) void*)(dst + 8 * 1024), 16, 0, 0);
    };
    auto write_a = [&](f32x4v (&pre)[4], int stage) {
        unsigned char* base = smem + stage * A_BYTES + a_dst;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f16x4 h, l;
            split4(pre[i], h, l);
            *reinterpret_cast<f16x4*>(base + i * 64 * 16) = h;
            *reinterpret_cast<f16x4*>(base + i * 64 * 16 + A_P) = l;
        }
    };

    f32x16v hi[2][2], lo[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            hi[i][j] = f32x16v{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            lo[i][j] = hi[i][j];
        }

    const int a_frag = lh * A_G + (wm * 64 + l31) * 16;            // + rb*32*16 + s*2*A_G (+ A_P)
    const int w_frag = W_BASE + lh * W_G + (wn * 64 + l31) * 16;   // + cb*32*16 + s*2*W_G (+ W_P)
    f16x8 af0[2][2], wf0[2][2], af1[2][2], wf1[2][2];              // fragments of the two 16-deep k-steps of a tile
    auto read_frags = [&](f16x8 (&af)[2][2], f16x8 (&wf)[2][2], int a_stage, int w_stage, int s) {
        const unsigned char* sa = smem + a_stage * A_BYTES + a_frag + s * 2 * A_G;
        const unsigned char* sw = smem + w_stage * W_BYTES + w_frag + s * 2 * W_G;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                af[b][p] = *reinterpret_cast<const f16x8*>(sa + b * 32 * 16 + p * A_P);
                wf[b][p] = *reinterpret_cast<const f16x8*>(sw + b * 32 * 16 + p * W_P);
            }
    };
    // 6 MFMAs: row block i of one 16-deep k-step of the wave's 64 x 64 tile
    auto mfma_row = [&](int i, f16x8 (&af)[2][2], f16x8 (&wf)[2][2]) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                hi[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], wf[j][0], hi[i][j], 0, 0, 0);
                lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], wf[j][1], lo[i][j], 0, 0, 0);
                lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][1], wf[j][0], lo[i][j], 0, 0, 0);
            }
    };

    // one k-iteration = two phases, each 12 MFMAs with the staging work of the OTHER phase's data interleaved by the
    // compiler (no fences inside a phase: an in-order wave can only hide VALU / LDS / VMEM issue in the 32-cycle shadow of
    // its own MFMAs, and the two waves of a SIMD run in lockstep on the one barrier per iteration):
    //   phase A: MFMA(tile kt, k-step 0)  ||  fragments of k-step 1 -> regs; rows of tile kt+1: wait, split, ds_write
    //   barrier (tile kt+1 complete in LDS; every wave is done reading the stage tile kt+2 will overwrite)
    //   phase B: MFMA(tile kt, k-step 1)  ||  weight DMA + row loads of tile kt+3; fragments of tile kt+1, k-step 0 -> regs
    auto iteration = [&](int kt, f32x4v (&pre)[4], int ws_cur, int ws_next, int ws_new) {
        const int as = kt & 1;
        if (!(ABL & 16)) read_frags(af1, wf1, as, ws_cur, 1);
        if (!(ABL & 8)) mfma_row(0, af0, wf0);                      // six MFMAs queued before the wave may block on its rows
        __builtin_amdgcn_sched_barrier(0);
        wait_rows(pre);                                             // also guarantees my (older) weight DMA of tile kt+1
        if (!(ABL & 1)) write_a(pre, as ^ 1);
        if (!(ABL & 8)) mfma_row(1, af0, wf0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 8)) mfma_row(0, af1, wf1);                      // queue matrix work first: hipcc waits lgkmcnt(0) before the
        __builtin_amdgcn_sched_barrier(0);                          // first MFMA after a ds_read, whatever that read feeds
        if (!(ABL & 16)) read_frags(af0, wf0, as ^ 1, ws_next, 0);
        if (!(ABL & 4)) dma_w(kt + W_AHEAD, ws_new);                // VM issue order per iteration: 2 DMAs, then 4 row loads
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 2)) load_a(pre, kt + 3);
        if (!(ABL & 8)) mfma_row(1, af1, wf1);
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: tile 0 staged, weight tiles 1 and 2 and the rows of tiles 1 (preB) and 2 (preA) in flight ----
    dma_w(0, 0);
    load_a(preA, 0);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(preA[0]), "+v"(preA[1]), "+v"(preA[2]), "+v"(preA[3])::"memory");
    __builtin_amdgcn_sched_barrier(0);
    write_a(preA, 0);
    __builtin_amdgcn_sched_barrier(0);
    dma_w(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    load_a(preB, 1);
    __builtin_amdgcn_sched_barrier(0);
    dma_w(2, 2);
    __builtin_amdgcn_sched_barrier(0);
    load_a(preA, 2);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(af0, wf0, 0, 0, 0);

    // weight ring position of tile kt is kt % 5; the loop is unrolled by two so that the register sets have static names
    auto ring = [](int x) { return x >= W_STAGES ? x - W_STAGES : x; };
    int wc = 0;                                                     // kt % W_STAGES
    int kt = 0;
    for (; kt + 1 < KT; kt += 2) {
        iteration(kt, preB, wc, ring(wc + 1), ring(wc + W_AHEAD));
        wc = ring(wc + 1);
        iteration(kt + 1, preA, wc, ring(wc + 1), ring(wc + W_AHEAD));
        wc = ring(wc + 1);
    }
    if (kt < KT) iteration(kt, preB, wc, ring(wc + 1), ring(wc + W_AHEAD));
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(preA[0]), "+v"(preA[1]), "+v"(preA[2]), "+v"(preA[3]), "+v"(preB[0]), "+v"(preB[1]), "+v"(preB[2]), "+v"(preB[3])::"memory");

    // ---- epilogue: D[i = (r&3) + 8 (r>>2) + 4 lh][j = l31] of each 32x32 block.  An activation beyond the fp16 range became inf
    // in its high part and surfaces here as a non-finite output (0 * inf = NaN in `chk`): flagged, the caller recomputes in fp32
    // (which also reproduces honestly whatever a non-finite INPUT gives) ----
    float chk = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rbase = m0 + wm * 64 + i * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                float v = hi[i][j][r] + lo[i][j][r] * LO_INV + bv;
                if (ACT == 1) v = gelu_tanh_f(v);
                if (row < M) { out[(size_t)row * ldo + col] = v; chk = __builtin_fmaf(v, 0.f, chk); }
            }
        }
    }
    if (overflow && chk != 0.f) atomicOr(overflow, 1);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Same product with the activations ALREADY split by the kernel that produced them (residual_ln_mod, the attention epilogue,
// or this kernel's own GELU epilogue): A arrives as a "split activation" (common.h: fp16 hi / lo planes in 1-KiB chunks of 16 rows x
// 32 k, the same 4 bytes per element as fp32; one chunk = one DMA piece = 8 full cache lines), so the activation tile reaches LDS by
// LDS-DMA exactly like the weight tile -- no register staging, no VALU split, no ds_write, no
// counted waits on registers.  The split is the same function of the fp32 value as `split4`, so the results are bit-identical
// to linear_f16x2_kernel on the un-split tensor.
//
// LDS image of an activation tile: [plane][row 0..255][4 slots of 16 B], slot = k-group ^ ((row >> 2) & 3).  A DMA instruction
// writes 1 KiB = 16 rows x 4 slots in lane order (that is all the hardware offers: M0 base + lane * 16), so the swizzle is applied
// on the SOURCE side: lane (row, slot) fetches k-group slot ^ ((row >> 2) & 3) of its row -- still inside the same contiguous chunk.
// A fragment read (32 rows x one k-group per half wave) then touches each of the 64 banks once per 16 lanes.
// Rings: 3 activation stages (32 KiB) + 4 weight stages (16 KiB) = the CU's whole 160 KiB; the DMAs of activation tile kt+2 and
// weight tile kt+3 are issued at the top of iteration kt into the stages iteration kt-1 released at its barrier.
constexpr int PA_ROW = 64;                       // bytes per row and plane of a 32-deep k-tile
constexpr int PA_P = BM * PA_ROW;                // 16384
constexpr int PA_BYTES = 2 * PA_P;               // 32768
constexpr int PA_STAGES = 3, PW_STAGES = 4;
constexpr int PW_BASE = PA_STAGES * PA_BYTES;    // 98304
constexpr int P_LDS_BYTES = PW_BASE + PW_STAGES * W_BYTES;   // 163840

// one LDS-DMA piece: 64 lanes x 16 B from (uniform base + per-lane 32-bit byte offset) to LDS [lds, lds + 1 KiB) in lane order.
// Inline asm because hipcc will not select the SGPR-base form for the builtin (it rebuilds a 64-bit VGPR address per piece).
__device__ __forceinline__ void lds_dma16(const void* base, unsigned voff, unsigned lds)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory", "m0");   // M0 declared as clobbered
}

// fused residual epilogue (RES): out = resid + gate * (A W^T + bias), gate element (row, col) at gate + (row / T) gsb + (row % T) gst + col
// (per-sample table: gst = 0; per-token table: gsb = 0; gate NULL: out = resid + y).  The multiply and the add are separate
// fp32 operations, exactly as residual_ln_mod_kernel performs them on the stored y: same bits, one tensor round trip less.
struct ResArgs { const float* resid; long ldr; const float* gate; long gsb, gst; int T; };

// SK = 1 (small M: a 256-row tile grid has 12 .. 96 work-groups for this model's Linears at one image): gridDim.y work-groups share an
// output tile, each over K / gridDim.y consecutive k-tiles, and write their raw partial sums (no bias) to plane blockIdx.y of a
// [gridDim.y][M][ldo] fp32 workspace (`out`); splitk_finish_kernel adds the planes in ascending order and runs the epilogue.
template <int ACT, int OSPLIT, int PP = 1, int ABL = 0, int RES = 0, int SK = 0>
__global__ __launch_bounds__(512, 2) void linear_f16x2_pre_kernel(const _Float16* __restrict__ Ablk,
                                                                  const _Float16* __restrict__ Wp, const float* __restrict__ bias,
                                                                  float* __restrict__ out, _Float16* __restrict__ oblk, long ldo,
                                                                  int M, int N, int K, int* __restrict__ overflow, int mblocks, int nblocks, ResArgs res)
{
    static_assert(!SK || (ACT == 0 && OSPLIT == 0 && RES == 0), "a split-K partial has no epilogue");
    __shared__ __attribute__((aligned(16))) unsigned char smem[P_LDS_BYTES];
    unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, rt0 = 0;
    if (ABL & 2048) { tk0 = __builtin_readcyclecounter(); rt0 = __builtin_amdgcn_s_memrealtime(); }

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: lives in an SGPR
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    int mb, nb;
    {
        const int T = gridDim.x, orig = blockIdx.x;
        const int q8 = T >> 3, r8 = T & 7, xcd = orig & 7, idx = orig >> 3;
        const int w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
        const int per_group = GROUP_M * nblocks;
        const int group = w / per_group, first_m = group * GROUP_M;
        const int gsz = (mblocks - first_m) < GROUP_M ? (mblocks - first_m) : GROUP_M;
        const int in = w - group * per_group;
        mb = first_m + in % gsz;
        nb = in / gsz;
    }
    const int m0 = mb * BM, n0 = nb * BN;
    const int KTA = K / BK;                                         // k-tiles of the operands (their strides)
    const int KT = SK ? KTA / (int)gridDim.y : KTA, KL = KT - 1;    // k-tiles of this work-group
    const int kt_off = SK ? (int)blockIdx.y * KT : 0;
    if (SK) out += (size_t)blockIdx.y * M * ldo;

    // ---- DMA maps: wave w moves rows 16w..16w+15 and 128+16w.. of both planes (4 instructions) and 2 KiB of the weight tile.
    // Every source address is (uniform 64-bit base in SGPRs) + (per-lane 32-bit offset that never changes) and every LDS
    // destination is scalar, so that issuing a piece costs scalar adds only: VALU issue slots are what the partner wave's
    // matrix stream leaves least of (MI355X_MICROARCH.md, "Two waves per SIMD") ----
    const int d_g = (lane & 3) ^ ((lane >> 4) & 3);                 // source k-group of LDS slot (lane & 3) in row (lane >> 2)
    const unsigned a_off = (unsigned)((lane >> 2) * 64 + d_g * 16);     // bytes inside a 1-KiB chunk [16 rows][32 halfs]
    const int rb_last = (M - 1) >> 4;                               // ragged M: chunks past the end re-read the last one (rows discarded)
    int rb0 = m0 / 16 + wave, rb1 = rb0 + 8;
    rb0 = rb0 < rb_last ? rb0 : rb_last;
    rb1 = rb1 < rb_last ? rb1 : rb_last;
    // chunk (row block rb, k-tile kt, plane p) starts at ((rb KT + kt) 2 + p) KiB
    const unsigned char* const a_base[2] = {(const unsigned char*)Ablk + ((size_t)rb0 * KTA + kt_off) * 2048, (const unsigned char*)Ablk + ((size_t)rb1 * KTA + kt_off) * 2048};
    const unsigned char* const w_base = (const unsigned char*)(Wp + ((size_t)nb * KTA + kt_off) * (W_BYTES / 2) + (size_t)wave * 512);
    const unsigned w_off = lane * 16;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto dma_a = [&](int kt, int stage) {
        kt = kt < KL ? kt : KL;                                     // past the end: stage the last tile again (nobody reads it)
        const unsigned dst = lds0 + stage * PA_BYTES + wave * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_dma16(a_base[j & 1] + (size_t)kt * 2048 + (j >> 1) * 1024, a_off, dst + (j & 1) * 8192 + (j >> 1) * PA_P);
    };
    auto dma_w = [&](int kt, int stage) {
        kt = kt < KL ? kt : KL;
        const unsigned char* src = w_base + (size_t)kt * W_BYTES;
        const unsigned dst = lds0 + PW_BASE + stage * W_BYTES + wave * 1024;
        lds_dma16(src, w_off, dst);
        lds_dma16(src + 8 * 1024, w_off, dst + 8 * 1024);
    };

    f32x16v hi[2][2], lo[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            hi[i][j] = f32x16v{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            lo[i][j] = hi[i][j];
        }

    const int a_frag = (wm * 64 + l31) * PA_ROW + ((lh ^ ((l31 >> 2) & 3)) * 16);   // k-step 1: ^ 32 (k-group + 2); + b*32*64 (+ PA_P)
    const int w_frag = PW_BASE + lh * W_G + (wn * 64 + l31) * 16;                   // + cb*32*16 + s*2*W_G (+ W_P)
    f16x8 af0[2][2], wf0[2][2], af1[2][2], wf1[2][2];
    auto read_frags = [&](f16x8 (&af)[2][2], f16x8 (&wf)[2][2], int a_stage, int w_stage, int s) {
        const unsigned char* sa = smem + a_stage * PA_BYTES + (a_frag ^ (s * 32));
        const unsigned char* sw = smem + w_stage * W_BYTES + w_frag + s * 2 * W_G;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                af[b][p] = *reinterpret_cast<const f16x8*>(sa + b * 32 * PA_ROW + p * PA_P);
                wf[b][p] = *reinterpret_cast<const f16x8*>(sw + b * 32 * 16 + p * W_P);
            }
    };
    auto mfma_row = [&](int i, f16x8 (&af)[2][2], f16x8 (&wf)[2][2]) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // operands swapped (weights as the A operand): the block comes out transposed, lane = output ROW, registers = 16
                // output columns in groups of 4 consecutive ones -> 16-byte (fp32) / 8-byte (fp16 plane) epilogue stores
                hi[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j][0], af[i][0], hi[i][j], 0, 0, 0);
                lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j][1], af[i][0], lo[i][j], 0, 0, 0);
                lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j][0], af[i][1], lo[i][j], 0, 0, 0);
            }
    };

    // The single-phase schedule (PP == 0; kept for tools/ablate_gemm_pre.py, the product runs the ping-pong loop below).
    // VM ops retire in issue order; per iteration a wave issues 4 activation DMAs (tile kt+2), then 2 weight DMAs (tile kt+3).
    // At the barrier that ends phase A of iteration kt, tile kt+1 must have landed: its activation DMAs were issued in iteration
    // kt-1 and may be followed by that iteration's 2 weight DMAs and this iteration's 6 -> vmcnt(8).
    auto iteration = [&](int kt, int a_cur, int a_next, int a_new, int w_cur, int w_next, int w_new) {
        dma_a(kt + 2, a_new);
        dma_w(kt + 3, w_new);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(af1, wf1, a_cur, w_cur, 1);
        mfma_row(0, af0, wf0);
        mfma_row(1, af0, wf0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(0, af1, wf1);                                      // matrix work queued before the reads: hipcc waits lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);                          // before the first MFMA that follows a ds_read
        read_frags(af0, wf0, a_next, w_next, 0);
        mfma_row(1, af1, wf1);
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: issue order A0 W0 W1 A1 W2 so that the loop's vmcnt(8) accounting holds from iteration 0 on ----
    dma_a(0, 0);
    dma_w(0, 0);
    dma_w(1, 1);
    dma_a(1, 1);
    dma_w(2, 2);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                // tile 0 (4 + 2 oldest) landed; W1, A1, W2 in flight
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    int ac = 0, wc = 0;                                             // kt % 3, kt % 4
    if (PP == 0) {
        read_frags(af0, wf0, 0, 0, 0);
        for (int kt = 0; kt < KT; ++kt) {
            const int a1 = ac == 2 ? 0 : ac + 1, a2 = a1 == 2 ? 0 : a1 + 1;
            iteration(kt, ac, a1, a2, wc, (wc + 1) & 3, (wc + 3) & 3);
            ac = a1;
            wc = (wc + 1) & 3;
        }
    } else {
        // "Ping-pong": the two waves of a SIMD (waves w and w + 4 of the work-group) run half an iteration apart.  While one of
        // them issues its 24 MFMAs of tile kt back to back (compute segment: nothing else in its stream, the matrix pipe never
        // waits for an operand), the other does ALL its memory work for its next tile (load segment: 16 fragment reads, 6 DMA
        // pieces, the counted wait) in the shadow of those MFMAs; at the barrier they swap roles.  A wave therefore never mixes
        // MFMAs with waits, and the SIMD's matrix pipe is fed by whichever wave is in its compute segment.
        //   segment:   0      1      2      3     ...
        //   waves 0-3  L(0)   C(0)   L(1)   C(1)
        //   waves 4-7   -     L(0)   C(0)   L(1)       (one extra barrier up front, one less at the end)
        // Tile kt is read in segments 2kt and 2kt+1; its activation stage is refilled with tile kt+3 by DMAs issued in the load
        // segments L(kt+1) (segments 2kt+2 / 2kt+3), its weight stage (ring of 4) with tile kt+4 likewise.  Each wave waits for its
        // own pieces of tile kt+1 at the end of L(kt) -- vmcnt(8) as above -- and the barrier that follows makes them visible.
        const int grp = wave >> 2;
        if (ABL & 2048) tk1 = __builtin_readcyclecounter();
        if (grp) __builtin_amdgcn_s_barrier();
        // static priority for the later-dispatched half: it loses every issue arbitration otherwise; per-segment priority flips
        // measured slower (profiles/r2_gemm_presplit_ablation.txt)
        if (!(ABL & 64) && grp) __builtin_amdgcn_s_setprio(1);
        // ABL & 2048 (tools/microbench only): s_memtime stamps at the segment boundaries, summed per wave and written behind the bias
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        unsigned acc_reads = 0, acc_vm = 0, acc_bar1 = 0, acc_mfma = 0, acc_bar2 = 0;
#define STAMP(t) do { if (ABL & 2048) { __builtin_amdgcn_sched_barrier(0); t = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } } while (0)
        for (int kt = 0; kt < KT; ++kt) {
            const int a1 = ac == 2 ? 0 : ac + 1, a2 = a1 == 2 ? 0 : a1 + 1;
            STAMP(t0);
            if ((ABL & 2048) && kt > 0) acc_bar2 += (unsigned)(t0 - t4);
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL & 16)) {
                read_frags(af0, wf0, ac, wc, 0);
                read_frags(af1, wf1, ac, wc, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL & 1)) dma_a(kt + 2, a2);
            if (!(ABL & 4)) dma_w(kt + 3, (wc + 3) & 3);
            __builtin_amdgcn_sched_barrier(0);
            if (ABL & 2048) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                STAMP(t1);                                          // all issued, fragment reads landed
            }
            if (ABL & 5) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            STAMP(t2);                                              // my pieces of the next tile landed
            if (!(ABL & 32)) __builtin_amdgcn_s_barrier();
            STAMP(t3);
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL & 8)) {                                       // per accumulator the same summation order as mfma_row
                mfma_row(0, af0, wf0);
                mfma_row(1, af0, wf0);
                mfma_row(0, af1, wf1);
                mfma_row(1, af1, wf1);
            }
            __builtin_amdgcn_sched_barrier(0);
            STAMP(t4);                                              // last MFMA issued
            if (ABL & 2048) { acc_reads += (unsigned)(t1 - t0); acc_vm += (unsigned)(t2 - t1); acc_bar1 += (unsigned)(t3 - t2); acc_mfma += (unsigned)(t4 - t3); }
            if (!(ABL & 32) && !(grp && kt == KL)) __builtin_amdgcn_s_barrier();
            ac = a1;
            wc = (wc + 1) & 3;
        }
#undef STAMP
        if (ABL & 2048) tk2 = __builtin_readcyclecounter();
        if ((ABL & 2048) && blockIdx.x == gridDim.x / 2 && lane == 0) {
            float* dbg = const_cast<float*>(bias) + N + wave * 8;
            dbg[0] = (float)acc_reads / KT; dbg[1] = (float)acc_vm / KT; dbg[2] = (float)acc_bar1 / KT;
            dbg[3] = (float)acc_mfma / KT; dbg[4] = (float)acc_bar2 / (KT - 1);
            dbg[5] = (float)(tk1 - tk0); dbg[6] = (float)(tk2 - tk1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // nothing of mine may still be writing LDS when the wave ends

    // ---- epilogue: D^T[n = (r&3) + 8 (r>>2) + 4 lh][m = l31] of each 32x32 block: a lane owns one output row and, per register
    // group g = r >> 2, four consecutive columns.  Storing those 8- or 16-byte pieces of 32 different rows per instruction
    // measured +10..20 % on the whole kernel (write transactions, not bytes), so the wave's 64 x 64 outputs are first transposed
    // through its own LDS slice (the rings are dead by now) and leave as 16 B per lane, 128 (fp16 planes) or 256 (fp32) contiguous
    // bytes per row.  OSPLIT: the output is written as the two fp16 planes the next Linear consumes ----
    constexpr int STG_ROW = OSPLIT ? 144 : 272, STG_PLANE = 64 * STG_ROW;   // 64 rows x (128 | 256 B + pad); 18 | 17 KiB per wave
    unsigned char* stg = smem + wave * (OSPLIT ? 2 * STG_PLANE : STG_PLANE);
    __syncthreads();                                                // every wave has drained its DMAs (vmcnt(0) above): LDS is free
    float chk = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cl = j * 32 + 8 * g + 4 * lh, col = n0 + wn * 64 + cl;
            const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rl = i * 32 + l31, row = m0 + wm * 64 + rl;
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[c] = hi[i][j][4 * g + c] + lo[i][j][4 * g + c] * LO_INV + (c == 0 ? bv.x : c == 1 ? bv.y : c == 2 ? bv.z : bv.w);
                    if (ACT == 1) v[c] = gelu_tanh_f(v[c]);
                }
                if (OSPLIT) {
                    f16x4 h, l;
                    split4(make_float4(opaque_f32(v[0]), opaque_f32(v[1]), opaque_f32(v[2]), opaque_f32(v[3])), h, l);   // common.h: why opaque
                    *reinterpret_cast<f16x4*>(stg + rl * STG_ROW + cl * 2) = h;
                    *reinterpret_cast<f16x4*>(stg + STG_PLANE + rl * STG_ROW + cl * 2) = l;
                    if (row < M) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) chk = __builtin_fmaf((float)h[c], 0.f, chk);   // |v| beyond fp16: h = inf -> flagged at the producer
                    }
                } else {
                    *reinterpret_cast<float4*>(stg + rl * STG_ROW + cl * 4) = make_float4(v[0], v[1], v[2], v[3]);
                    if (row < M) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) chk = __builtin_fmaf(v[c], 0.f, chk);
                    }
                }
            }
        }
    }
    if (OSPLIT) {                                                   // the slice is private to the wave: its LDS ops complete in order
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) {
                const int rl = pass * 8 + (lane >> 3), seg = lane & 7, row = m0 + wm * 64 + rl;
                const f16x8 val = *reinterpret_cast<const f16x8*>(stg + p * STG_PLANE + rl * STG_ROW + seg * 16);
                if (row < M) *reinterpret_cast<f16x8*>(oblk + split_blk_index(row, n0 + wn * 64 + seg * 8, p, N / 32)) = val;
            }
    } else {
#pragma unroll
        for (int pass = 0; pass < 16; ++pass) {
            const int rl = pass * 4 + (lane >> 4), seg = lane & 15, row = m0 + wm * 64 + rl;
            float4 val = *reinterpret_cast<const float4*>(stg + rl * STG_ROW + seg * 16);
            if (row < M) {
                const int col = n0 + wn * 64 + seg * 4;
                if (RES) {
                    const float4 r = *reinterpret_cast<const float4*>(res.resid + (size_t)row * res.ldr + col);
                    if (res.gate) {
                        const int b = row / res.T, t = row - b * res.T;
                        const float4 g = *reinterpret_cast<const float4*>(res.gate + b * res.gsb + t * res.gst + col);
                        val.x = r.x + g.x * val.x; val.y = r.y + g.y * val.y; val.z = r.z + g.z * val.z; val.w = r.w + g.w * val.w;
                    } else {
                        val.x += r.x; val.y += r.y; val.z += r.z; val.w += r.w;
                    }
                }
                *reinterpret_cast<float4*>(out + (size_t)row * ldo + col) = val;
            }
        }
    }
    if (overflow && chk != 0.f) atomicOr(overflow, 1);
    if ((ABL & 2048) && blockIdx.x == gridDim.x / 2 && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const_cast<float*>(bias)[N + wave * 8 + 7] = (float)(__builtin_readcyclecounter() - tk2);
    }
    if ((ABL & 2048) && overflow && tid == 0) {      // per work-group timeline record: [realtime start, end (100 MHz), cycles, HW_ID, XCC_ID]
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int* rec = overflow + 8 + (size_t)blockIdx.x * 8;
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime(), tk3 = __builtin_readcyclecounter();
        rec[0] = (int)(unsigned)rt0; rec[1] = (int)(unsigned)rt1; rec[2] = (int)(unsigned)(tk3 - tk0);
        rec[3] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 4); rec[4] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
}

// fp32 [rows, cols] (row stride ld) -> split activation (stand-alone producer: tests, and inputs that no fused producer writes)
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, long ld, _Float16* __restrict__ blk,
                                                         long rows, int cols, int* __restrict__ overflow)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = cols / 4;
    if (idx >= rows * c4) return;
    const long r = idx / c4;
    const int c = (int)(idx % c4) * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c);
    f16x4 h, l;
    float mx = 0.f;
    split4(v, h, l, mx);
    *reinterpret_cast<f16x4*>(blk + split_blk_index(r, c, 0, cols / 32)) = h;
    *reinterpret_cast<f16x4*>(blk + split_blk_index(r, c, 1, cols / 32)) = l;
    if (!(mx < F16_MAX) && overflow) atomicOr(overflow, 1);
}

// Second launch of a split-K Linear: v = sum_s partial[s] (planes in ascending order: deterministic) + bias, then the epilogue of
// linear_f16x2_pre_kernel, operation for operation -- GELU, the split-activation output for the next Linear, or the fused residual
// update.  One thread per (row, 8 consecutive columns): 16-byte loads per plane, one 16-byte store per fp16 plane or two float4.
template <int ACT, int OSPLIT, int RES>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ ws, int S, const float* __restrict__ bias,
                                                            float* __restrict__ out, _Float16* __restrict__ oblk, long ldo,
                                                            int M, int N, int* __restrict__ overflow, ResArgs res)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n8 = N >> 3;
    if (idx >= (long)M * n8) return;
    const int row = (int)(idx / n8), col = (int)(idx - (long)row * n8) << 3;
    const float* p = ws + (size_t)row * N + col;
    const size_t plane = (size_t)M * N;
    float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    for (int s_ = 1; s_ < S; ++s_) {
        const float4 c = *reinterpret_cast<const float4*>(p + s_ * plane), d = *reinterpret_cast<const float4*>(p + s_ * plane + 4);
        a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
        b.x += d.x; b.y += d.y; b.z += d.z; b.w += d.w;
    }
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    if (bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(bias + col), b1 = *reinterpret_cast<const float4*>(bias + col + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    }
    if (ACT == 1) {
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = gelu_tanh_f(v[c]);
    }
    float chk = 0.f;
    if (OSPLIT) {
        f16x4 h0, l0, h1, l1;
        split4(make_float4(opaque_f32(v[0]), opaque_f32(v[1]), opaque_f32(v[2]), opaque_f32(v[3])), h0, l0);
        split4(make_float4(opaque_f32(v[4]), opaque_f32(v[5]), opaque_f32(v[6]), opaque_f32(v[7])), h1, l1);
        const f16x8 h = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]}, l = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
#pragma unroll
        for (int c = 0; c < 8; ++c) chk = __builtin_fmaf((float)h[c], 0.f, chk);
        *reinterpret_cast<f16x8*>(oblk + split_blk_index(row, col, 0, N / 32)) = h;
        *reinterpret_cast<f16x8*>(oblk + split_blk_index(row, col, 1, N / 32)) = l;
    } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) chk = __builtin_fmaf(v[c], 0.f, chk);
        if (RES) {
            const float* rp = res.resid + (size_t)row * res.ldr + col;
            const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
            const float r[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            if (res.gate) {
                const int bb = row / res.T, t = row - bb * res.T;
                const float* gp = res.gate + bb * res.gsb + t * res.gst + col;
                const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 4);
                const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = r[c] + g[c] * v[c];
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] += r[c];
            }
        }
        float* op = out + (size_t)row * ldo + col;
        *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    if (overflow && chk != 0.f) atomicOr(overflow, 1);
}

// first launch of a split-K Linear: the partial planes
static int launch_splitk_partials(const void* a_blk, const void* packed, float* ws, int M, int N, int K, int ksplit, int* overflow, hipStream_t stream)
{
    const int mblocks = (M + BM - 1) / BM, nblocks = N / BN;
    hipLaunchKernelGGL((linear_f16x2_pre_kernel<0, 0, 1, 0, 0, 1>), dim3((unsigned)(mblocks * nblocks), (unsigned)ksplit), dim3(512), 0, stream,
                       (const _Float16*)a_blk, (const _Float16*)packed, (const float*)nullptr, ws, (_Float16*)nullptr, (long)N,
                       M, N, K, overflow, mblocks, nblocks, ResArgs{});
    return check_launch("linear_f16x2_pre_kernel(split-K partials)");
}

}  // namespace selftok

using namespace selftok;

extern "C" {

size_t selftok_linear_f16x2_packed_bytes(int N, int K)
{
    if (N <= 0 || K <= 0 || N % BN || K % BK) return 0;
    return (size_t)N * K * 4;          // two fp16 planes
}

int selftok_linear_f16x2_pack_weight(const float* W, void* packed, int N, int K, int* overflow, hipStream_t stream)
{
    if (!W || !packed || N <= 0 || K <= 0 || N % BN || K % BK) { set_last_error("linear_f16x2_pack_weight: need N % 128 == 0 and K % 32 == 0"); return SELFTOK_EINVAL; }
    const long n = (long)N * (K / 8);
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, W, (_Float16*)packed, N, K, overflow);
    return check_launch("pack_weight_kernel");
}

int selftok_linear_f16x2_f32(const float* A, long lda, const void* packed, const float* bias, float* out, long ldo,
                             int M, int N, int K, int flags, int* overflow, hipStream_t stream)
{
    if (M < 0 || N <= 0 || K <= 0 || N % BN || K % BK) { set_last_error("linear_f16x2: need N % 128 == 0 and K % 32 == 0"); return SELFTOK_EINVAL; }
    if (M == 0) return SELFTOK_OK;
    if (!A || !packed || !out || lda < K || ldo < N || (lda & 3)) { set_last_error("linear_f16x2: bad pointers/strides (lda % 4 == 0, lda >= K, ldo >= N)"); return SELFTOK_EINVAL; }
    const int mblocks = (M + BM - 1) / BM, nblocks = N / BN;
    const dim3 grid((unsigned)(mblocks * nblocks));
#ifdef SELFTOK_GEMM_ABLATE
    {
        const char* e = getenv("SELFTOK_GEMM_ABL");
        const int abl = e ? atoi(e) : 0;
#define ABL_CASE(v) if (abl == v) { hipLaunchKernelGGL((linear_f16x2_kernel<0, v>), grid, dim3(512), 0, stream, A, lda, (const _Float16*)packed, bias, out, ldo, M, N, K, overflow, mblocks, nblocks); return check_launch("linear_f16x2_kernel(ablated)"); }
        ABL_CASE(1) ABL_CASE(2) ABL_CASE(3) ABL_CASE(4) ABL_CASE(7) ABL_CASE(8) ABL_CASE(16) ABL_CASE(24) ABL_CASE(23) ABL_CASE(31)
#undef ABL_CASE
    }
#endif
    if (flags & SELFTOK_LINEAR_GELU)
        hipLaunchKernelGGL(linear_f16x2_kernel<1>, grid, dim3(512), 0, stream, A, lda, (const _Float16*)packed, bias, out, ldo, M, N, K, overflow, mblocks, nblocks);
    else
        hipLaunchKernelGGL(linear_f16x2_kernel<0>, grid, dim3(512), 0, stream, A, lda, (const _Float16*)packed, bias, out, ldo, M, N, K, overflow, mblocks, nblocks);
    return check_launch("linear_f16x2_kernel");
}

size_t selftok_split_f16x2_bytes(long rows, int cols)
{
    if (rows < 0 || cols <= 0 || cols % 32) return 0;
    return (size_t)((rows + 15) / 16) * 16 * cols * 4;      // two fp16 planes, rows padded to the 16-row chunk
}

int selftok_split_f16x2_f32(const float* x, long ld, void* blk, long rows, int cols, int* overflow, hipStream_t stream)
{
    if (rows < 0 || cols <= 0 || (cols & 31) || (ld & 3) || ld < cols) { set_last_error("split_f16x2: cols % 32 == 0, ld % 4 == 0, ld >= cols"); return SELFTOK_EINVAL; }
    if (rows == 0) return SELFTOK_OK;
    if (!x || !blk || ((size_t)blk & 15) || ((size_t)x & 15)) { set_last_error("split_f16x2: null / unaligned pointer"); return SELFTOK_EINVAL; }
    const long n = rows * (cols / 4);
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, ld, (_Float16*)blk, rows, cols, overflow);
    return check_launch("split_rows_kernel");
}

int selftok_linear_f16x2_split(const void* a_blk, const void* packed, const float* bias, float* out, void* out_blk, long ldo,
                               int M, int N, int K, int flags, int* overflow, hipStream_t stream)
{
    if (M < 0 || N <= 0 || K <= 0 || N % BN || K % BK) { set_last_error("linear_f16x2_split: need N % 128 == 0 and K % 32 == 0"); return SELFTOK_EINVAL; }
    if (M == 0) return SELFTOK_OK;
    const bool osplit = out_blk != nullptr;
    if (!a_blk || !packed || ((size_t)a_blk & 15) || ((size_t)out & 15) || ((size_t)out_blk & 15) || (bias && ((size_t)bias & 15))
        || (osplit ? out != nullptr : (!out || ldo < N || (ldo & 3)))) {
        set_last_error("linear_f16x2_split: bad pointers/strides (a_blk, out, out_blk and bias 16-byte aligned; either out with ldo % 4 == 0, ldo >= N, or out_blk)");
        return SELFTOK_EINVAL;
    }
    const int mblocks = (M + BM - 1) / BM, nblocks = N / BN;
    const dim3 grid((unsigned)(mblocks * nblocks));
    const _Float16* ab = (const _Float16*)a_blk;
    _Float16* ob = (_Float16*)out_blk;
#ifdef SELFTOK_GEMM_ABLATE
    {
        const char* e = getenv("SELFTOK_GEMM_ABL");
        const int abl = e ? atoi(e) : 0;
#define PRE_ABL(pp, v) if (abl == (pp ? v : 1000 + v)) { hipLaunchKernelGGL((linear_f16x2_pre_kernel<0, 0, pp, v>), grid, dim3(512), 0, stream, ab, (const _Float16*)packed, bias, out, ob, ldo, M, N, K, overflow, mblocks, nblocks, ResArgs{}); return check_launch("linear_f16x2_pre_kernel(ablated)"); }
        PRE_ABL(0, 0) PRE_ABL(1, 1) PRE_ABL(1, 4) PRE_ABL(1, 5) PRE_ABL(1, 8) PRE_ABL(1, 16) PRE_ABL(1, 21) PRE_ABL(1, 24) PRE_ABL(1, 32) PRE_ABL(1, 64) PRE_ABL(1, 13) PRE_ABL(1, 37) PRE_ABL(1, 2048)
#undef PRE_ABL
    }
#endif
#define PRE_LAUNCH(ACT, OS) hipLaunchKernelGGL((linear_f16x2_pre_kernel<ACT, OS>), grid, dim3(512), 0, stream, ab, (const _Float16*)packed, bias, out, ob, ldo, M, N, K, overflow, mblocks, nblocks, ResArgs{})
    if (flags & SELFTOK_LINEAR_GELU) { if (osplit) PRE_LAUNCH(1, 1); else PRE_LAUNCH(1, 0); }
    else { if (osplit) PRE_LAUNCH(0, 1); else PRE_LAUNCH(0, 0); }
#undef PRE_LAUNCH
    return check_launch("linear_f16x2_pre_kernel");
}

int selftok_linear_f16x2_split_residual(const void* a_blk, const void* packed, const float* bias,
                                        const float* resid, long ldr, const float* gate, long gate_stride_b, long gate_stride_t, int T,
                                        float* out, long ldo, int M, int N, int K, int* overflow, hipStream_t stream)
{
    if (M < 0 || N <= 0 || K <= 0 || N % BN || K % BK) { set_last_error("linear_f16x2_split_residual: need N % 128 == 0 and K % 32 == 0"); return SELFTOK_EINVAL; }
    if (M == 0) return SELFTOK_OK;
    if (!a_blk || !packed || !resid || !out || ldo < N || (ldo & 3) || ldr < N || (ldr & 3) || T <= 0
        || ((size_t)a_blk & 15) || ((size_t)out & 15) || ((size_t)resid & 15) || (bias && ((size_t)bias & 15))
        || (gate && (((size_t)gate & 15) || (gate_stride_b & 3) || (gate_stride_t & 3)))) {
        set_last_error("linear_f16x2_split_residual: bad pointers/strides (16-byte aligned, strides multiples of 4, T > 0)");
        return SELFTOK_EINVAL;
    }
    const int mblocks = (M + BM - 1) / BM, nblocks = N / BN;
    hipLaunchKernelGGL((linear_f16x2_pre_kernel<0, 0, 1, 0, 1>), dim3((unsigned)(mblocks * nblocks)), dim3(512), 0, stream,
                       (const _Float16*)a_blk, (const _Float16*)packed, bias, out, (_Float16*)nullptr, ldo,
                       M, N, K, overflow, mblocks, nblocks, ResArgs{resid, ldr, gate, gate_stride_b, gate_stride_t, T});
    return check_launch("linear_f16x2_pre_kernel(residual)");
}

size_t selftok_linear_f16x2_splitk_workspace_bytes(int M, int N, int ksplit)
{
    if (M <= 0 || N <= 0 || ksplit <= 1) return 0;
    return (size_t)ksplit * M * N * sizeof(float);
}

static bool splitk_ok(int K, int ksplit, const void* workspace)
{
    return ksplit >= 2 && ksplit <= 64 && (K / BK) % ksplit == 0 && workspace && !((size_t)workspace & 15);
}

int selftok_linear_f16x2_split_k(const void* a_blk, const void* packed, const float* bias, float* out, void* out_blk, long ldo,
                                 int M, int N, int K, int flags, int ksplit, void* workspace, int* overflow, hipStream_t stream)
{
    if (ksplit == 1) return selftok_linear_f16x2_split(a_blk, packed, bias, out, out_blk, ldo, M, N, K, flags, overflow, stream);
    if (M < 0 || N <= 0 || K <= 0 || N % BN || K % BK) { set_last_error("linear_f16x2_split_k: need N % 128 == 0 and K % 32 == 0"); return SELFTOK_EINVAL; }
    if (M == 0) return SELFTOK_OK;
    const bool osplit = out_blk != nullptr;
    if (!a_blk || !packed || ((size_t)a_blk & 15) || ((size_t)out & 15) || ((size_t)out_blk & 15) || (bias && ((size_t)bias & 15))
        || (osplit ? out != nullptr : (!out || ldo < N || (ldo & 3))) || !splitk_ok(K, ksplit, workspace)) {
        set_last_error("linear_f16x2_split_k: bad pointers/strides (as linear_f16x2_split), or ksplit not in 2..64 / not a divisor of K / 32, or no 16-byte aligned workspace");
        return SELFTOK_EINVAL;
    }
    float* ws = (float*)workspace;
    if (int rc = launch_splitk_partials(a_blk, packed, ws, M, N, K, ksplit, overflow, stream)) return rc;
    const long n = (long)M * (N / 8);
    const dim3 grid((unsigned)((n + 255) / 256));
    _Float16* ob = (_Float16*)out_blk;
#define FIN_LAUNCH(ACT, OS) hipLaunchKernelGGL((splitk_finish_kernel<ACT, OS, 0>), grid, dim3(256), 0, stream, (const float*)ws, ksplit, bias, out, ob, ldo, M, N, overflow, ResArgs{})
    if (flags & SELFTOK_LINEAR_GELU) { if (osplit) FIN_LAUNCH(1, 1); else FIN_LAUNCH(1, 0); }
    else { if (osplit) FIN_LAUNCH(0, 1); else FIN_LAUNCH(0, 0); }
#undef FIN_LAUNCH
    return check_launch("splitk_finish_kernel");
}

int selftok_linear_f16x2_split_residual_k(const void* a_blk, const void* packed, const float* bias,
                                          const float* resid, long ldr, const float* gate, long gate_stride_b, long gate_stride_t, int T,
                                          float* out, long ldo, int M, int N, int K, int ksplit, void* workspace, int* overflow, hipStream_t stream)
{
    if (ksplit == 1)
        return selftok_linear_f16x2_split_residual(a_blk, packed, bias, resid, ldr, gate, gate_stride_b, gate_stride_t, T, out, ldo, M, N, K, overflow, stream);
    if (M < 0 || N <= 0 || K <= 0 || N % BN || K % BK) { set_last_error("linear_f16x2_split_residual_k: need N % 128 == 0 and K % 32 == 0"); return SELFTOK_EINVAL; }
    if (M == 0) return SELFTOK_OK;
    if (!a_blk || !packed || !resid || !out || ldo < N || (ldo & 3) || ldr < N || (ldr & 3) || T <= 0
        || ((size_t)a_blk & 15) || ((size_t)out & 15) || ((size_t)resid & 15) || (bias && ((size_t)bias & 15))
        || (gate && (((size_t)gate & 15) || (gate_stride_b & 3) || (gate_stride_t & 3))) || !splitk_ok(K, ksplit, workspace)) {
        set_last_error("linear_f16x2_split_residual_k: bad pointers/strides (as linear_f16x2_split_residual), or ksplit not in 2..64 / not a divisor of K / 32, or no 16-byte aligned workspace");
        return SELFTOK_EINVAL;
    }
    float* ws = (float*)workspace;
    if (int rc = launch_splitk_partials(a_blk, packed, ws, M, N, K, ksplit, overflow, stream)) return rc;
    const long n = (long)M * (N / 8);
    hipLaunchKernelGGL((splitk_finish_kernel<0, 0, 1>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const float*)ws, ksplit, bias, out,
                       (_Float16*)nullptr, ldo, M, N, overflow, ResArgs{resid, ldr, gate, gate_stride_b, gate_stride_t, T});
    return check_launch("splitk_finish_kernel(residual)");
}

}  // extern "C"
