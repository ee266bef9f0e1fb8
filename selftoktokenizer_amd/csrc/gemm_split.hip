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
// Activations are split on the fly (fp32 in HBM, split in registers while staging to LDS); weights are split once at
// load time into the kernel's own tile order (selftok_linear_f16x2_pack_weight), so a weight tile reaches LDS by
// direct LDS-DMA (global_load_lds, 16 B per lane, no VGPR round trip) as one linear 16 KiB copy.
//
// Range: fp16 overflows at 65504.  |activation| >= 65504 raises *overflow (device int, caller-owned, sticky) and the
// caller redoes the work with the fp32 library GEMM; weights are checked at pack time.  Values below the fp16 normal
// range are carried by the scaled low part (tests cover 1e-7..1e-3).
//
// Tiling: workgroup = 8 waves = 256 (M) x 128 (N) outputs, K step 32, two LDS stages (2 x 48.25 KiB); each wave owns
// 64 x 64 = 2 x 2 MFMA blocks with hi+lo accumulators (128 VGPRs), 12 MFMAs per 8 ds_read_b128 per 16-deep k-step.
// LDS images are MFMA-fragment ordered [plane][k-group of 8][row][8 halfs]: every fragment read is 512 contiguous
// bytes per half wave (conflict-free); the activation image pads each k-group by 32 B so that the ds_write_b64 of the
// split pass are conflict-free too.  Work-groups are renumbered so that each XCD (own L2) owns a contiguous range of
// tiles, walked in 8-row-block groups, which keeps the A row panels and W column panels of concurrently running
// work-groups in one L2.
#include "common.h"
#include "selftok_hip.h"

namespace selftok {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16v __attribute__((ext_vector_type(16)));

constexpr int BM = 256, BN = 128, BK = 32;
constexpr float LO_SCALE = 2048.0f, LO_INV = 1.0f / 2048.0f;
constexpr float F16_MAX = 65504.0f;

constexpr int A_G = BM * 16 + 32;          // bytes between k-groups of the activation image (padded)
constexpr int A_P = 4 * A_G;               // bytes between the hi and lo planes
constexpr int A_BYTES = 2 * A_P;           // 33024
constexpr int W_G = BN * 16;               // weight image: linear (filled by LDS-DMA)
constexpr int W_P = 4 * W_G;
constexpr int W_BYTES = 2 * W_P;           // 16384
constexpr int STAGE = A_BYTES + W_BYTES;   // 49408
constexpr int GROUP_M = 8;
constexpr int NSTAGE = 3;                   // 3 x 48.25 KiB = 144.75 KiB of the CU's 160 KiB

__device__ __forceinline__ float gelu_tanh_f(float x)   // same formula as selftok_bias_gelu_f32 (elementwise.hip)
{
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float inner = k0 * (x + k1 * x * x * x);
    return 0.5f * x * (1.0f + tanhf(inner));
}

__device__ __forceinline__ void split4(const float4& v, f16x4& hi, f16x4& lo, float& mx)
{
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const _Float16 h = (_Float16)x[j];
        const float r = x[j] - (float)h;               // exact
        hi[j] = h;
        lo[j] = (_Float16)(r * LO_SCALE);
        mx = fmaxf(mx, fabsf(x[j]));
    }
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

template <int ACT>
__global__ __launch_bounds__(512, 2) void linear_f16x2_kernel(const float* __restrict__ A, long lda, const _Float16* __restrict__ Wp,
                                                              const float* __restrict__ bias, float* __restrict__ out, long ldo,
                                                              int M, int N, int K, int* __restrict__ overflow, int mblocks, int nblocks)
{
    // three-stage LDS ring: while tile t is multiplied, tile t+1 is complete (its first fragments are pre-read before
    // the barrier, so the matrix pipe restarts immediately after it) and tile t+2 is being filled.
    __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE];

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
    const int KT = K / BK;

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

    float4 pre[4];
    float mx = 0.f;
    auto load_a = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pre[i] = *reinterpret_cast<const float4*>(a_src[i] + (size_t)kt * BK);
    };
    auto dma_w = [&](int kt, int stage) {
        const _Float16* src = w_src + (size_t)kt * (W_BYTES / 2);
        unsigned char* dst = smem + stage * STAGE + A_BYTES + wave * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 8 * 512),
                                         (__attribute__((address_space(3))) void*)(dst + 8 * 1024), 16, 0, 0);
    };
    auto write_a = [&](int stage) {
        unsigned char* base = smem + stage * STAGE + a_dst;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f16x4 h, l;
            split4(pre[i], h, l, mx);
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
    const int w_frag = A_BYTES + lh * W_G + (wn * 64 + l31) * 16;  // + cb*32*16 + s*2*W_G (+ W_P)
    f16x8 a0[2][2], w0[2][2], a1[2][2], w1[2][2];                  // fragment sets of the two 16-deep k-steps of a tile
    auto read_frags = [&](int stage, int s, f16x8 (&af)[2][2], f16x8 (&wf)[2][2]) {
        const unsigned char* st = smem + stage * STAGE;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                af[b][p] = *reinterpret_cast<const f16x8*>(st + a_frag + b * 32 * 16 + s * 2 * A_G + p * A_P);
                wf[b][p] = *reinterpret_cast<const f16x8*>(st + w_frag + b * 32 * 16 + s * 2 * W_G + p * W_P);
            }
    };
    auto mfma_row = [&](int i, f16x8 (&af)[2][2], f16x8 (&wf)[2][2]) {      // 6 MFMAs: row block i x both column blocks
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            hi[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], wf[j][0], hi[i][j], 0, 0, 0);
            lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], wf[j][1], lo[i][j], 0, 0, 0);
            lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][1], wf[j][0], lo[i][j], 0, 0, 0);
        }
    };

    // ---- prologue: tiles 0 and 1 staged, tile 2's rows in flight.  Tile indices past the end are clamped to the last
    // tile everywhere: the tail iterations then stage data nobody reads, and the loop body has no conditional code
    // (a conditional prefetch makes hipcc wait for the loads at once to merge registers at the join) ----
    const int KL = KT - 1;
    load_a(0);
    dma_w(0, 0);
    write_a(0);
    load_a(1 < KL ? 1 : KL);
    dma_w(1 < KL ? 1 : KL, 1);
    write_a(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load_a(2 < KL ? 2 : KL);
    read_frags(0, 0, a0, w0);

    int cs = 0, ns = 1, ws = 2;                                     // ring positions of tiles kt, kt+1, kt+2
    for (int kt = 0; kt < KT; ++kt) {
        const int t2 = kt + 2 < KL ? kt + 2 : KL, t3 = kt + 3 < KL ? kt + 3 : KL;
        read_frags(cs, 1, a1, w1);
        mfma_row(0, a0, w0);
        __builtin_amdgcn_sched_barrier(0);
        write_a(ws);                                                // rows of tile kt+2 (loaded an iteration ago): split -> LDS
        dma_w(t2, ws);                                              // weights of tile kt+2 by LDS-DMA
        __builtin_amdgcn_sched_barrier(0);                          // vmcnt counts in issue order: the DMAs must stay OLDER than the loads
        load_a(t3);                                                 // rows of tile kt+3: a full iteration to arrive
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(1, a0, w0);
        mfma_row(0, a1, w1);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(ns, 0, a0, w0);                                  // tile kt+1 has been complete since the last barrier
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(1, a1, w1);
        __builtin_amdgcn_sched_barrier(0);
        // tile kt+2 must be in LDS (my DMAs landed, my ds_writes done) before anyone reads it after the NEXT barrier; the
        // four row loads of tile kt+3 (younger than the DMAs) stay in flight across the barrier
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int t = cs; cs = ns; ns = ws; ws = t;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (overflow && !(mx < F16_MAX)) atomicOr(overflow, 1);

    // ---- epilogue: D[i = (r&3) + 8 (r>>2) + 4 lh][j = l31] of each 32x32 block ----
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
                if (row < M) out[(size_t)row * ldo + col] = v;
            }
        }
    }
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
    if (flags & SELFTOK_LINEAR_GELU)
        hipLaunchKernelGGL(linear_f16x2_kernel<1>, grid, dim3(512), 0, stream, A, lda, (const _Float16*)packed, bias, out, ldo, M, N, K, overflow, mblocks, nblocks);
    else
        hipLaunchKernelGGL(linear_f16x2_kernel<0>, grid, dim3(512), 0, stream, A, lda, (const _Float16*)packed, bias, out, ldo, M, N, K, overflow, mblocks, nblocks);
    return check_launch("linear_f16x2_kernel");
}

}  // extern "C"
