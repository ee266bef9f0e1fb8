// The fp32 Q-Former ENCODER with the reference's exact summation orders and transcendental polynomials (round 5).  gfx950 only.
//
// The reference runs `Encoder.forward` (mimogpt/models/selftok/models_ours.py:204-257, 315-343; DualBlock / DualAttention,
// modules.py:165-327) in fp32 on the CPU; its token ids are the argmax of the resulting features, and a token at a near-tie of its two
// best codes (18 of the 32768 tokens of the reference's 64-image run have a gap below 1e-5) flips under ANY other rounding sequence.
// Round 4 made the VAE latents bit-equal; behind them the encoder ran on hipBLASLt GEMMs (another summation order per M) and the
// rounds 1-3 kernels (LayerNorm / softmax / GELU in their own arithmetic): ids matched on 16 images only because no near-tie was hit.
// This file evaluates every reduction as the same sequence of fp32 operations torch-CPU executes (oracle/encoder_exact.c documents
// how each was established and is the bit-for-bit CPU twin of every kernel here), so the pre-quantizer features -- and with the
// exact VQ kernel the token ids -- are the reference's bit for bit, at every batch size (each output row depends on its own row only).
//
//   xe_gemm_kernel      C[m][n] = sum_k A[m][k] B[n][k] in MKL sgemm's order: sequential fmaf chains from 0 per K-block (blocks of 384,
//                       two halves for 384 < K < 768), out = ((bias + c0) + c1) + ...  On gfx950 v_mfma_f32_32x32x2_f32 IS that chain:
//                       an fp32-input MFMA is bit-for-bit fma(a1, b1, fma(a0, b0, acc)) per output (csrc/vq.hip, round 1), so a
//                       K-block is K/2 chained MFMAs at the fp32 matrix rate and the block fold is 16 v_add_f32 per 32 x 32 tile.
//                       Operands are staged through LDS in 16-k chunks (coalesced 16-byte loads, double buffered, XOR swizzle) exactly
//                       as csrc/vae_exact.hip's xconv_kernel does.  Epilogues: Linear (bias first; optional exact GELU; optional
//                       `res + gate * y` with separately rounded multiply and add), attention scores (C * 1/sqrt(d)), and the P V
//                       product of ATen's flash kernel (C *= exp(old max - new max) at a kv-block boundary, out = C * (1 / sum)).
//   xe_ln_kernel        ATen LayerNormKernelImpl: RowwiseMoments over 8 fp32 lanes (Welford with FMAs, chunks of 16 vectors, binary
//                       cascade, scalar lane combination with GCC's FMA contractions), rstd = 1 / sqrtf(var + eps),
//                       y = fma((x - mean) * rstd, gamma, beta), then the reference's `x * (1 + scale) + shift` (modules.py:29-32).
//   xe_softmax_kernel   the row pass of ATen's cpu_flash_attention in fp32: kv blocks of 512, Vectorized<float>::exp_u20, 16-lane sums
//                       folded 8 / 4 / 2 / 1, glibc expf for the rescale, sum = fma(exp, old sum, block sum).
//   xe_gelu / xe_silu   ATen's GELU(tanh) / SiLU with Sleef's tanhf_u10 / expf_u10 restated operation for operation (double-float
//                       arithmetic in the FMA form; 0 mismatches against the library on all 2^32 inputs on the CPU side).
// This file is compiled with -ffp-contract=off: every FMA below is explicit.  Divisions and square roots are hipcc's correctly
// rounded defaults.
#include "common.h"
#include "exact_math.h"
#include "selftok_hip.h"

namespace selftok {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t xu32x4 __attribute__((ext_vector_type(4)));

// mode of the element-wise kernel: 0 GELU(tanh), 1 SiLU, 2 Sleef expf, 3 Sleef tanhf, 4 exp_u20 (2..4: test hooks of the building blocks)
__device__ __forceinline__ float xe_unary1(float v, int mode)
{
    if (mode == 0) return xe_gelu_tanh1(v);
    if (mode == 1) return xe_silu1(v);
    if (mode == 2) return xe_sleef_expf(v);
    if (mode == 3) return xe_sleef_tanhf(v);
    return xe_exp_u20(v);
}
// four consecutive elements per thread (16-byte loads / stores; round 6: fc1's GELU is its own pass over [rows, 6144] in the exact MMDiT, 3 % of the step with one
// element per thread).  x may alias y (same indices read and written by the same thread).
__global__ __launch_bounds__(256) void xe_unary_kernel(const float* x, float* y, long n, int mode)
{
    const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (i4 + 4 <= n && ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(y)) & 15) == 0) {
        const float4 v = *reinterpret_cast<const float4*>(x + i4);
        float4 r;
        r.x = xe_unary1(v.x, mode); r.y = xe_unary1(v.y, mode); r.z = xe_unary1(v.z, mode); r.w = xe_unary1(v.w, mode);
        *reinterpret_cast<float4*>(y + i4) = r;
    } else {
        for (long i = i4; i < n && i < i4 + 4; ++i) y[i] = xe_unary1(x[i], mode);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// GEMM in MKL's summation order
// ---------------------------------------------------------------------------------------------------------------------------------
#define XE_MAXBLK 16
struct XeGemmArgs {
    const float* a; long lda, a_bs, a_hs;      // A[z = b * H + h][m][k] at a + b a_bs + h a_hs + m lda + k
    const float* b; long ldb, b_bs, b_hs;      // B[z][n][k]
    float* c; long ldc, c_bs, c_hs;            // C[z][m][n]
    const float* bias;                          // mode 0: [N] or null
    const float* res; long ldr; int res_mod;    // mode 0: y = res[row(m, res_mod)][n] + (gate ? gate * y : y); row(m, d) = m % d (d > 0: per-token table), m / -d (d < 0: per-sample), m (0)
    const float* gate; long ldg; int gate_mod;  //         gate[row(m, gate_mod)][n]
    const float* rescale;                       // mode 2: [nres][Z * M]: C *= rescale[i][z M + m] before K-block j (bit j of rescale_mask; i = its rank)
    const float* rowscale;                      // mode 2: [Z * M]: out = C * rowscale
    float out_scale;                            // mode 1: out = C * out_scale
    int M, N, K, H;
    int mode, gelu;                             // gelu: bit 0 GELU(tanh) epilogue, bit 1 bias added after the K-blocks instead of before
    int nblk;
    int mt, nt;                                 // xe_gemm128_kernel: row / column tile counts (1-D XCD-aware grid)
    int blk_chunks;                             // xe_gemm128_kernel: uniform K-block length in 32-k chunks (the launcher checks that blk_end is uniform)
    int blk_end[XE_MAXBLK];                     // K-block ends (multiples of 4; K itself a multiple of 16)
    unsigned rescale_mask;
};

template <int NT, int WM, int WN>
__global__ __launch_bounds__(256) void xe_gemm_kernel(XeGemmArgs g)
{
    // workgroup tile: RA = 32 WM rows of A x RB = 32 NT WN rows of B; one staged chunk = 16 k = 64 bytes of every row
    constexpr int RA = 32 * WM, RB = 32 * NT * WN;
    constexpr int NPA = (RA * 4 + 255) / 256, NPB = (RB * 4 + 255) / 256;
    __shared__ xu32x4 lds[2][(RA + RB) * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int i = lane & 31, h = lane >> 5;
    const int z = blockIdx.z, zb = z / g.H, zh = z - zb * g.H;
    const int pwg = blockIdx.x * RA, nwg = blockIdx.y * RB;
    const int p0 = pwg + wm * 32, n0 = nwg + wn * (32 * NT);
    const float* A = g.a + (size_t)zb * g.a_bs + (size_t)zh * g.a_hs;
    const float* Bm = g.b + (size_t)zb * g.b_bs + (size_t)zh * g.b_hs;

    const float* arow[NPA]; int aslot[NPA]; bool aon[NPA];
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
        const int idx = tid + 256 * j, row = idx >> 2, q = idx & 3;
        aon[j] = idx < RA * 4;
        const int m = min(pwg + (aon[j] ? row : 0), g.M - 1);
        arow[j] = A + (size_t)m * g.lda + q * 4;
        aslot[j] = row * 4 + (q ^ ((row >> 2) & 3));
    }
    const float* brow[NPB]; int bslot[NPB]; bool bon[NPB];
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
        const int idx = tid + 256 * j, row = idx >> 2, q = idx & 3;
        bon[j] = idx < RB * 4;
        const int n = min(nwg + (bon[j] ? row : 0), g.N - 1);
        brow[j] = Bm + (size_t)n * g.ldb + q * 4;
        bslot[j] = (RA + row) * 4 + (q ^ ((row >> 2) & 3));
    }

    float C[NT][16];
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + t * 32 + i;
        const float b0 = (g.mode == 0 && g.bias != nullptr && n < g.N && !(g.gelu & 2)) ? g.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { C[t][r] = b0; acc[t][r] = 0.f; }
    }

    xu32x4 sa[NPA], sb[NPB];
    auto fetch = [&](int c) {
#pragma unroll
        for (int j = 0; j < NPA; ++j) sa[j] = *reinterpret_cast<const xu32x4*>(arow[j] + c * 16);
#pragma unroll
        for (int j = 0; j < NPB; ++j) sb[j] = *reinterpret_cast<const xu32x4*>(brow[j] + c * 16);
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NPA; ++j) if (aon[j]) lds[buf][aslot[j]] = sa[j];
#pragma unroll
        for (int j = 0; j < NPB; ++j) if (bon[j]) lds[buf][bslot[j]] = sb[j];
    };
    const int ra = wm * 32 + i, swa = (ra >> 2) & 3;
    int rb[NT], swb[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { rb[t] = RA + wn * (32 * NT) + t * 32 + i; swb[t] = ((rb[t] - RA) >> 2) & 3; }
    int bi = 0, ri = 0;
    auto fold = [&]() {                            // K-block done: C += chain, the next chain starts from 0; then the flash kernel's rescale, if one is due
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { C[t][r] = C[t][r] + acc[t][r]; acc[t][r] = 0.f; }
        ++bi;
        if (bi < g.nblk && ((g.rescale_mask >> bi) & 1u)) {   // `dst *= exp(old max - new max)` before the next kv block
            const float* rs = g.rescale + ((size_t)ri * gridDim.z + z) * g.M;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float f = rs[min(p0 + (r & 3) + 8 * (r >> 2) + 4 * h, g.M - 1)];
#pragma unroll
                for (int t = 0; t < NT; ++t) C[t][r] = C[t][r] * f;
            }
            ++ri;
        }
    };
    auto compute = [&](int buf, int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const xu32x4 va = lds[buf][ra * 4 + (q ^ swa)];
            xu32x4 vb[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) vb[t] = lds[buf][rb[t] * 4 + (q ^ swb[t])];
#pragma unroll
            for (int s = 0; s < 2; ++s) {          // one MFMA = k pair (4 q + 2 s, 4 q + 2 s + 1): lanes 0..31 feed the even k, lanes 32..63 the odd k
                const float fa = __uint_as_float(h ? va[2 * s + 1] : va[2 * s]);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float fb = __uint_as_float(h ? vb[t][2 * s + 1] : vb[t][2 * s]);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[t], 0, 0, 0);
                }
            }
            if (bi < g.nblk && k0 + 4 * q + 4 == g.blk_end[bi]) fold();      // block ends are multiples of 4 (uniform branch)
        }
    };

    const int nchunks = g.K >> 4;
    fetch(0);
    stage(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        if (more) fetch(c + 1);
        compute(c & 1, c * 16);
        if (more) stage((c + 1) & 1);
        __syncthreads();
    }

    // epilogue: lane (col = i, rows (r & 3) + 8 (r >> 2) + 4 h)
    float* Cout = g.c + (size_t)zb * g.c_bs + (size_t)zh * g.c_hs;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + t * 32 + i;
        if (n >= g.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = p0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= g.M) continue;
            float v = C[t][r];
            if (g.mode == 1) v = v * g.out_scale;
            else if (g.mode == 2) v = v * g.rowscale[(size_t)z * g.M + m];
            else {
                if ((g.gelu & 2) && g.bias != nullptr) v = v + g.bias[n];          // bias LAST: at::linear on a non-contiguous input = matmul, then add_(bias)
                if (g.gelu & 1) v = xe_gelu_tanh1(v);
                if (g.gate != nullptr) v = g.gate[(size_t)(g.gate_mod > 0 ? m % g.gate_mod : (g.gate_mod < 0 ? m / -g.gate_mod : m)) * g.ldg + n] * v;
                if (g.res != nullptr) v = g.res[(size_t)(g.res_mod > 0 ? m % g.res_mod : (g.res_mod < 0 ? m / -g.res_mod : m)) * g.ldr + n] + v;
            }
            Cout[(size_t)m * g.ldc + n] = v;
        }
    }
}

// The same GEMM for the big Linears (N >= 128, K a multiple of 32, K-block ends multiples of 32: every block Linear of the MMDiT, the wide ones of the Q-Former).
// Workgroup = 8 waves = 128 x 128 outputs, wave = 64 x 32 = two MFMA tiles (two independent chains); one staged chunk = 32 k.  The even / odd k of a row are
// de-interleaved while staging ([16 even | 16 odd] fp32 per row and chunk, as xconv_kernel's rows): a lane reads the operands of ITS half of every k pair (lanes 0..31
// the even k, 32..63 the odd k) as four ds_read_b128 per row tile -- no per-MFMA select, 32 MFMAs (2048 matrix cycles) per wave and barrier instead of 16, and
// ~110 VGPRs: two workgroups (16 waves) per CU.
template <int TN>      // column tiles per wave: 1 = 8 waves of 64 x 32 (the built default), 2 = 4 waves of 64 x 64
__global__ __launch_bounds__(512 / TN) void xe_gemm128_kernel(XeGemmArgs g)
{
    constexpr int RA = 128, RB = 128;
    __shared__ xu32x4 lds[2][(RA + RB) * 8];                      // 128 bytes per row and chunk: 64 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NTHR = 512 / TN, NPC = 1024 / NTHR;
    const int wm = wave & 1, wn = wave >> 1;
    const int i = lane & 31, h = lane >> 5;
    const int z = blockIdx.z, zb = z / g.H, zh = z - zb * g.H;
    // XCD-aware tile order (1-D grid of 8 * ceil(tiles / 8) workgroups): workgroups go round-robin to the 8 XCDs, each with its own L2.  All tiles are listed band by band
    // (a band = 8 row tiles; inside a band column tile by column tile) and XCD x takes the x-th eighth of that list, in order: the ~64 workgroups resident on an XCD then
    // cover 8 row tiles x 8 column tiles = 16 operand tiles (the plain row-fastest order: 64 row tiles x 1 column tile = 65 operand tiles, every A tile used once per L2)
    int tm_, tn_;
    {
        const int MT = g.mt, NT = g.nt, T = MT * NT, per = (T + 7) >> 3;
        const int L = blockIdx.x, sidx = (L & 7) * per + (L >> 3);
        if ((L >> 3) >= per || sidx >= T) return;
        const int full = MT >> 3, rem = MT & 7, cut = full * 8 * NT;
        if (sidx < cut) { const int band = sidx / (8 * NT), r = sidx - band * 8 * NT; tn_ = r >> 3; tm_ = band * 8 + (r & 7); }
        else { const int r = sidx - cut; tn_ = r / rem; tm_ = full * 8 + (r - tn_ * rem); }
    }
    const int pwg = tm_ * RA, nwg = tn_ * RB;
    const int p0 = pwg + wm * 64, n0 = nwg + wn * (32 * TN);
    const float* A = g.a + (size_t)zb * g.a_bs + (size_t)zh * g.a_hs;
    const float* Bm = g.b + (size_t)zb * g.b_bs + (size_t)zh * g.b_hs;

    // staging: piece (row, q) = k 8 q .. 8 q + 7 of the row's chunk (two 16-byte loads) -> even k (4 floats) to LDS piece q, odd k to piece 4 + q; 2 pieces per thread
    const float* src[NPC]; int slot[NPC], sw[NPC];
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
        const int idx = tid + NTHR * j, row = idx >> 2, q = idx & 3;            // rows 0..127 = A, 128..255 = B
        const bool isb = row >= RA;
        const int r = isb ? row - RA : row;
        const float* base = isb ? Bm + (size_t)min(nwg + r, g.N - 1) * g.ldb : A + (size_t)min(pwg + r, g.M - 1) * g.lda;
        src[j] = base + q * 8;
        sw[j] = (row >> 1) & 7;
        slot[j] = row * 8 + q;
    }
    float C[2][TN][16];
    f32x16 acc[2][TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + tn * 32 + i;
        const float b0 = (g.mode == 0 && g.bias != nullptr && n < g.N && !(g.gelu & 2)) ? g.bias[n] : 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) { C[tm][tn][r] = b0; acc[tm][tn][r] = 0.f; }
    }
    xu32x4 s0[NPC], s1[NPC];
    auto fetch = [&](int c) {
#pragma unroll
        for (int j = 0; j < NPC; ++j) {
            s0[j] = *reinterpret_cast<const xu32x4*>(src[j] + c * 32);
            s1[j] = *reinterpret_cast<const xu32x4*>(src[j] + c * 32 + 4);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NPC; ++j) {
            const xu32x4 ev = {s0[j][0], s0[j][2], s1[j][0], s1[j][2]}, od = {s0[j][1], s0[j][3], s1[j][1], s1[j][3]};
            const int base = slot[j] & ~7, q = slot[j] & 7;
            lds[buf][base + (q ^ sw[j])] = ev;
            lds[buf][base + ((4 + q) ^ sw[j])] = od;
        }
    };
    int ra[2], swa[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { ra[t] = wm * 64 + t * 32 + i; swa[t] = (ra[t] >> 1) & 7; }
    int rb[TN], swb[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) { rb[t] = RA + wn * (32 * TN) + t * 32 + i; swb[t] = (rb[t] >> 1) & 7; }
    auto fold = [&]() {                            // (no rescale here: the P V product of the attention has N = head_dim < 128 and takes xe_gemm_kernel)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) { C[tm][tn][r] = C[tm][tn][r] + acc[tm][tn][r]; acc[tm][tn][r] = 0.f; }
    };
    auto compute = [&](int buf) {
        // register double buffer over the four 8-k pieces of the chunk: piece q + 1 is read while piece q's 8 MFMAs issue; the scheduling barriers keep
        // the compiler from hoisting all twelve ds_read_b128 to the top (48 live registers: 177 VGPRs, one workgroup per CU instead of two)
        xu32x4 va[2], vb[TN];
#pragma unroll
        for (int t = 0; t < 2; ++t) va[t] = lds[buf][ra[t] * 8 + ((4 * h) ^ swa[t])];
#pragma unroll
        for (int t = 0; t < TN; ++t) vb[t] = lds[buf][rb[t] * 8 + ((4 * h) ^ swb[t])];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            xu32x4 na[2], nb[TN];
#pragma unroll
            for (int t = 0; t < 2; ++t) na[t] = q < 3 ? lds[buf][ra[t] * 8 + ((4 * h + q + 1) ^ swa[t])] : va[t];
#pragma unroll
            for (int t = 0; t < TN; ++t) nb[t] = q < 3 ? lds[buf][rb[t] * 8 + ((4 * h + q + 1) ^ swb[t])] : vb[t];
#pragma unroll
            for (int e = 0; e < 4; ++e)            // k pair 8 q + 2 e (+1): this lane's element is its half's e-th of the piece
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(va[tm][e]), __uint_as_float(vb[tn][e]), acc[tm][tn], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2; ++t) va[t] = na[t];
#pragma unroll
            for (int t = 0; t < TN; ++t) vb[t] = nb[t];
        }
    };
    const int nchunks = g.K >> 5;
    int cb = 0;
    fetch(0);
    stage(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        if (more) fetch(c + 1);
        compute(c & 1);
        if (++cb == g.blk_chunks || !more) { cb = 0; fold(); }      // uniform K-blocks of g.blk_chunks chunks (the last may be shorter): no table lookup in the loop
        if (more) stage((c + 1) & 1);
        __syncthreads();
    }
    float* Cout = g.c + (size_t)zb * g.c_bs + (size_t)zh * g.c_hs;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + tn * 32 + i;
        if (n >= g.N) continue;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = p0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= g.M) continue;
                float v = C[tm][tn][r];
                if (g.mode == 1) v = v * g.out_scale;
                else if (g.mode == 2) v = v * g.rowscale[(size_t)z * g.M + m];
                else {
                    if ((g.gelu & 2) && g.bias != nullptr) v = v + g.bias[n];
                    if (g.gelu & 1) v = xe_gelu_tanh1(v);
                    if (g.gate != nullptr) v = g.gate[(size_t)(g.gate_mod > 0 ? m % g.gate_mod : (g.gate_mod < 0 ? m / -g.gate_mod : m)) * g.ldg + n] * v;
                    if (g.res != nullptr) v = g.res[(size_t)(g.res_mod > 0 ? m % g.res_mod : (g.res_mod < 0 ? m / -g.res_mod : m)) * g.ldr + n] + v;
                }
                Cout[(size_t)m * g.ldc + n] = v;
                __builtin_amdgcn_sched_barrier(0);          // one output at a time: the unrolled epilogue must not set the kernel's register count
            }
    }
}

#ifndef XE_GEMM128_TN
#define XE_GEMM128_TN 1          // measured (profiles/r5_ex_gemm_wave_tile.txt): 64 x 32 wave tiles at 122 VGPRs / 4 waves per SIMD beat 64 x 64 at 211 / 2
#endif
static int launch_xe_gemm(const XeGemmArgs& g, int Z, hipStream_t stream)
{
    bool wide = g.rescale_mask == 0 && g.N >= 128 && g.M >= 128 && (g.K & 31) == 0 && (g.lda & 3) == 0 && (g.ldb & 3) == 0 && g.nblk > 0 && (g.blk_end[0] & 31) == 0;
    for (int j = 0; j < g.nblk && wide; ++j) wide = g.blk_end[j] == ((j + 1) * g.blk_end[0] < g.K ? (j + 1) * g.blk_end[0] : g.K);      // uniform blocks, last one shorter
    if (wide) {
        XeGemmArgs w = g;
        w.blk_chunks = g.blk_end[0] >> 5;
        w.mt = (g.M + 127) / 128; w.nt = (g.N + 127) / 128;
        dim3 grid((unsigned)(8 * ((w.mt * w.nt + 7) / 8)), 1, Z);
        hipLaunchKernelGGL((xe_gemm128_kernel<XE_GEMM128_TN>), grid, dim3(512 / XE_GEMM128_TN), 0, stream, w);
    } else if (g.N > 64) {
        dim3 grid((unsigned)((g.M + 63) / 64), (unsigned)((g.N + 127) / 128), Z);
        hipLaunchKernelGGL((xe_gemm_kernel<2, 2, 2>), grid, dim3(256), 0, stream, g);
    } else if (g.N > 32) {
        dim3 grid((unsigned)((g.M + 63) / 64), 1, Z);
        hipLaunchKernelGGL((xe_gemm_kernel<1, 2, 2>), grid, dim3(256), 0, stream, g);
    } else {
        dim3 grid((unsigned)((g.M + 127) / 128), 1, Z);
        hipLaunchKernelGGL((xe_gemm_kernel<1, 4, 1>), grid, dim3(256), 0, stream, g);
    }
    return check_launch("xe_gemm_kernel");
}

// MKL's K-blocking (probed: oracle/encoder_exact.c xe_mkl_kblock): K <= 384 one block; 384 < K < 768 two halves; else blocks of 384
static int mkl_blocks(int K, int k_base, int* ends, int n0)
{
    int n = n0;
    for (int k0 = 0; k0 < K;) {
        int kb;
        if (K <= 384) kb = K;
        else if (K < 768) kb = k0 == 0 ? (K + 1) / 2 : K - k0;
        else kb = K - k0 < 384 ? K - k0 : 384;
        k0 += kb;
        if (n >= XE_MAXBLK) return -1;
        ends[n++] = k_base + k0;
    }
    return n;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// LayerNorm (+ modulate): ATen RowwiseMoments<float>, 8 lanes
// ---------------------------------------------------------------------------------------------------------------------------------
struct XMom { float m1, m2; };
__device__ __forceinline__ void xe_add_moments_vec(int m0_add, XMom add, int& m0, XMom& acc)
{
    const int n = m0 + m0_add;
    const float c = n == 0 ? 0.f : (float)m0_add / (float)n;
    const float delta = add.m1 - acc.m1;
    const float m2_tmp = acc.m2 + add.m2;
    const float c_delta = c * delta;
    const float m0_delta = delta * (float)m0;
    acc.m1 = acc.m1 + c_delta;
    acc.m2 = fmaf(m0_delta, c_delta, m2_tmp);
    m0 = n;
}

// 8 threads per row: thread l owns fp32 lane l of ATen's 8-wide vectors (elements 8 j + l).  x [rows][N] (row stride ldx), N % 8 == 0,
// N <= 4096.  y = LN(x) [* (1 + scale[tok]) + shift[tok]], tok = row % T, tables with row stride ldt.
// FUSE (round 6): the row is first UPDATED, x' = x + gate[row(m)] * (lin [+ bias]) -- the residual update of a DismantledBlock (`x + gate * post_attention(attn)`,
// `x + gate * mlp(...)`, sd3/mmdit.py:485-496) that the Linear's epilogue carried in round 5: there 8 compute waves per CU waited for 64 operand loads each while
// the matrix pipe idled (proj at 0.72 of the peak against 0.87 with the plain epilogue); here it is two more streams of a bandwidth-bound pass.  The same fp32
// operations in the same order: v = lin + bias (only when the bias comes last), v = gate * v, x' = x + v.  x' is written to `xo` (may be x itself).
struct XeLnFuse {
    const float* lin; long ldl;                 // the Linear's output [rows][N]
    const float* gate; long ldg; int gate_mod;  // gate[row(m, gate_mod)][n] or null; row(m, d) = m % d (d > 0), m / -d (d < 0), m (0)
    const float* bias;                          // added to lin first, or null
    float* xo; long ldxo;
};

template <bool FUSE>
__global__ __launch_bounds__(256) void xe_ln_kernel(const float* x, long ldx, float* __restrict__ y, long ldy, const float* __restrict__ shift,
                                                    const float* __restrict__ scale, long ldt, int T, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, long rows, int N, float eps, float* __restrict__ stats, XeLnFuse f)
{
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long row = gid >> 3;
    const int l = (int)(gid & 7);
    if (row >= rows) return;                       // rows * 8 is padded to whole waves by the launcher's guard below (shuffles stay inside a row's 8 lanes)
    const float* xr = x + (size_t)row * ldx;
    const float* lr = FUSE ? f.lin + (size_t)row * f.ldl : nullptr;
    const float* gr = (FUSE && f.gate != nullptr) ? f.gate + (size_t)(f.gate_mod > 0 ? row % f.gate_mod : (f.gate_mod < 0 ? row / -f.gate_mod : row)) * f.ldg : nullptr;
    float* xor_ = FUSE ? f.xo + (size_t)row * f.ldxo : nullptr;
    const int n = N >> 3, m = (n + 15) >> 4;
    int depth = 0;
    while ((1 << depth) < m) ++depth;
    constexpr int MAXD = 6;                        // m <= 32 chunks
    XMom stk[MAXD]; int m0s[MAXD];
#pragma unroll
    for (int v = 0; v < MAXD; ++v) { stk[v] = XMom{0.f, 0.f}; m0s[v] = 0; }
    // whole chunks (N % 128 == 0: every shape of the models): the 16 elements of a chunk (and, fused, the 16 of lin / bias / gate) are loaded TOGETHER and one chunk AHEAD
    // of the Welford chain that consumes them -- one element at a time made every step of the chain wait for its own loads (round 6, fused: 284 -> 218 us at 22912 rows,
    // 200 -> 135 us at 16384: profiles/r6_ex_layernorm_batched_loads.txt).  Same operations in the same order.
    const bool whole = (n & 15) == 0;
    float xs[16], vs[16];
    auto fetch = [&](int ci) {                        // the two HBM streams; the bias row and the gate rows (one per sample) come from L1 / L2 at use
#pragma unroll
        for (int j = 0; j < 16; ++j) xs[j] = xr[(ci * 16 + j) * 8 + l];
        if (FUSE) {
#pragma unroll
            for (int j = 0; j < 16; ++j) vs[j] = lr[(ci * 16 + j) * 8 + l];
        }
    };
    if (whole) fetch(0);
    for (int ci = 0; ci < m; ++ci) {
        const int cnt = min(16, n - ci * 16);
        XMom a{0.f, 0.f};
        if (whole) {
            float cur[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) cur[j] = xs[j];
            if (FUSE) {
                float v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = vs[j];
                if (f.bias != nullptr) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) v[j] = v[j] + f.bias[(ci * 16 + j) * 8 + l];
                }
                if (gr != nullptr) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) v[j] = gr[(ci * 16 + j) * 8 + l] * v[j];
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) cur[j] = cur[j] + v[j];
            }
            if (ci + 1 < m) fetch(ci + 1);             // in flight while this chunk's chain runs (x_out may alias x: other columns than the stores below)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (FUSE) xor_[(ci * 16 + j) * 8 + l] = cur[j];          // read back below by the other lanes of this row's group (same wave): fenced
                const float cj = 1.0f / (float)(j + 1);
                const float d0 = cur[j] - a.m1;
                a.m1 = fmaf(d0, cj, a.m1);
                const float e0 = cur[j] - a.m1;
                a.m2 = fmaf(d0, e0, a.m2);
            }
        } else
        for (int j = 0; j < cnt; ++j) {
            const float cj = 1.0f / (float)(j + 1);
            const int col = (ci * 16 + j) * 8 + l;
            float x0 = xr[col];
            if (FUSE) {
                float v = lr[col];
                if (f.bias != nullptr) v = v + f.bias[col];
                if (gr != nullptr) v = gr[col] * v;
                x0 = x0 + v;
                xor_[col] = x0;                    // read back below by the other lanes of this row's group (same wave): fenced
            }
            const float d0 = x0 - a.m1;
            a.m1 = fmaf(d0, cj, a.m1);
            const float e0 = x0 - a.m1;
            a.m2 = fmaf(d0, e0, a.m2);
        }
        xe_add_moments_vec(cnt, a, m0s[0], stk[0]);
        int mask = ci + 1;
        bool go = true;
#pragma unroll
        for (int j = 1; j < MAXD; ++j) {
            go = go && j < depth && (mask & 1) == 0;
            if (go) {
                xe_add_moments_vec(m0s[j - 1], stk[j - 1], m0s[j], stk[j]);
                m0s[j - 1] = 0; stk[j - 1] = XMom{0.f, 0.f};
                mask >>= 1;
            }
        }
    }
#pragma unroll
    for (int j = 1; j < MAXD; ++j)
        if (j < depth) xe_add_moments_vec(m0s[j], stk[j], m0s[0], stk[0]);
    // scalar AddMoments over the 8 lanes (GCC contracts both updates into FMAs)
    float m1 = 0.f, m2 = 0.f;
    int m0 = 0;
    const int m0_add = m0s[0];
    for (int k = 0; k < 8; ++k) {
        const float a1 = __shfl(stk[0].m1, (threadIdx.x & 63 & ~7) + k, WAVE), a2 = __shfl(stk[0].m2, (threadIdx.x & 63 & ~7) + k, WAVE);
        const int nn = m0 + m0_add;
        const float c = nn == 0 ? 0.f : (float)m0_add / (float)nn;
        const float delta = a1 - m1;
        m1 = fmaf(c, delta, m1);
        m2 = m2 + fmaf(delta * delta * c, (float)m0, a2);
        m0 = nn;
    }
    const float var = m2 / (float)N;
    const float rstd = 1.0f / sqrtf(fmaxf(var, 0.f) + eps);
    if (stats != nullptr && l == 0) { stats[2 * row] = m1; stats[2 * row + 1] = rstd; }
    const float nmean = -m1;
    if (FUSE) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); xr = xor_; }
    float* yr = y + (size_t)row * ldy;
    const long tok = T > 0 ? row % T : row / -T;                       // T > 0: per-token tables (row % T); T < 0: per-sample (row / -T)
    const float* sh = shift != nullptr ? shift + (size_t)tok * ldt : nullptr;
    const float* sc = scale != nullptr ? scale + (size_t)tok * ldt : nullptr;
    for (int e0 = l * 4; e0 < N; e0 += 32) {       // 8 threads x 16 bytes = one 128-byte line per step
        const float4 v = *reinterpret_cast<const float4*>(xr + e0);
        float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = (o[e] + nmean) * rstd;
            float r = fmaf(t, gamma != nullptr ? gamma[e0 + e] : 1.0f, beta != nullptr ? beta[e0 + e] : 0.0f);
            if (sc != nullptr) r = r * (1.0f + sc[e0 + e]) + sh[e0 + e];
            o[e] = r;
        }
        *reinterpret_cast<float4*>(yr + e0) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// Wide rows (round 6): the same arithmetic with a row spread over G = chunks / 4 waves.  A chunk's Welford chain starts from zero, so the 12 chunks of a 1536-wide row
// are independent until ATen's cascade combines them: wave g of a workgroup computes the raw moments of chunks 4 g .. 4 g + 3 for the workgroup's 8 rows (8 lanes per
// row, as above) and leaves them in LDS; wave 0 then replays the cascade loop of xe_ln_kernel over the stored chunk moments -- the same calls with the same
// arguments in the same order -- and publishes mean / rstd; every wave applies 1 / G of the columns.  G x the waves in flight of the 8-threads-per-row kernel
// (whose grid is 8 waves per CU at 16384 rows).  N % 512 == 0 (whole chunks, a multiple of 4 of them), G <= 8.
template <bool FUSE>
__global__ __launch_bounds__(512) void xe_lnw_kernel(const float* x, long ldx, float* __restrict__ y, long ldy, const float* __restrict__ shift,
                                                     const float* __restrict__ scale, long ldt, int T, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, long rows, int N, float eps, float* __restrict__ stats, XeLnFuse f)
{
    __shared__ float s_m1[8][32][8], s_m2[8][32][8];            // [row of the workgroup][chunk][lane]
    __shared__ float s_mean[8], s_rstd[8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, rl = lane >> 3, l = lane & 7;
    const int G = blockDim.x >> 6;
    const long row = (long)blockIdx.x * 8 + rl;
    const bool live = row < rows;
    const long rowc = live ? row : rows - 1;
    const float* xr = x + (size_t)rowc * ldx;
    const float* lr = FUSE ? f.lin + (size_t)rowc * f.ldl : nullptr;
    const float* gr = (FUSE && f.gate != nullptr) ? f.gate + (size_t)(f.gate_mod > 0 ? rowc % f.gate_mod : (f.gate_mod < 0 ? rowc / -f.gate_mod : rowc)) * f.ldg : nullptr;
    float* xor_ = FUSE ? f.xo + (size_t)rowc * f.ldxo : nullptr;
    const int n = N >> 3, m = n >> 4;                          // m = 4 G whole chunks
    float xs[16], vs[16];
    auto fetch = [&](int ci) {
#pragma unroll
        for (int j = 0; j < 16; ++j) xs[j] = xr[(ci * 16 + j) * 8 + l];
        if (FUSE) {
#pragma unroll
            for (int j = 0; j < 16; ++j) vs[j] = lr[(ci * 16 + j) * 8 + l];
        }
    };
    fetch(4 * wave);
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const int ci = 4 * wave + cc;
        float cur[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) cur[j] = xs[j];
        if (FUSE) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = vs[j];
            if (f.bias != nullptr) {
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = v[j] + f.bias[(ci * 16 + j) * 8 + l];
            }
            if (gr != nullptr) {
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = gr[(ci * 16 + j) * 8 + l] * v[j];
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) cur[j] = cur[j] + v[j];
        }
        if (cc + 1 < 4) fetch(ci + 1);
        XMom a{0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (FUSE && live) xor_[(ci * 16 + j) * 8 + l] = cur[j];
            const float cj = 1.0f / (float)(j + 1);
            const float d0 = cur[j] - a.m1;
            a.m1 = fmaf(d0, cj, a.m1);
            const float e0 = cur[j] - a.m1;
            a.m2 = fmaf(d0, e0, a.m2);
        }
        s_m1[rl][ci][l] = a.m1; s_m2[rl][ci][l] = a.m2;
    }
    __syncthreads();
    if (wave == 0) {
        int depth = 0;
        while ((1 << depth) < m) ++depth;
        constexpr int MAXD = 6;
        XMom stk[MAXD]; int m0s[MAXD];
#pragma unroll
        for (int v = 0; v < MAXD; ++v) { stk[v] = XMom{0.f, 0.f}; m0s[v] = 0; }
        for (int ci = 0; ci < m; ++ci) {                        // xe_ln_kernel's loop with the chunk moments read instead of computed
            const XMom a{s_m1[rl][ci][l], s_m2[rl][ci][l]};
            xe_add_moments_vec(16, a, m0s[0], stk[0]);
            int mask = ci + 1;
            bool go = true;
#pragma unroll
            for (int j = 1; j < MAXD; ++j) {
                go = go && j < depth && (mask & 1) == 0;
                if (go) {
                    xe_add_moments_vec(m0s[j - 1], stk[j - 1], m0s[j], stk[j]);
                    m0s[j - 1] = 0; stk[j - 1] = XMom{0.f, 0.f};
                    mask >>= 1;
                }
            }
        }
#pragma unroll
        for (int j = 1; j < MAXD; ++j)
            if (j < depth) xe_add_moments_vec(m0s[j], stk[j], m0s[0], stk[0]);
        float m1 = 0.f, m2 = 0.f;
        int m0 = 0;
        const int m0_add = m0s[0];
        for (int k = 0; k < 8; ++k) {
            const float a1 = __shfl(stk[0].m1, (lane & ~7) + k, WAVE), a2 = __shfl(stk[0].m2, (lane & ~7) + k, WAVE);
            const int nn = m0 + m0_add;
            const float c = nn == 0 ? 0.f : (float)m0_add / (float)nn;
            const float delta = a1 - m1;
            m1 = fmaf(c, delta, m1);
            m2 = m2 + fmaf(delta * delta * c, (float)m0, a2);
            m0 = nn;
        }
        const float var = m2 / (float)N;
        const float rstd = 1.0f / sqrtf(fmaxf(var, 0.f) + eps);
        if (l == 0) {
            s_mean[rl] = m1; s_rstd[rl] = rstd;
            if (stats != nullptr && live) { stats[2 * row] = m1; stats[2 * row + 1] = rstd; }
        }
    }
    if (FUSE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (FUSE) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); xr = xor_; }
    if (!live) return;
    const float nmean = -s_mean[rl], rstd = s_rstd[rl];
    float* yr = y + (size_t)row * ldy;
    const long tok = T > 0 ? row % T : row / -T;
    const float* sh = shift != nullptr ? shift + (size_t)tok * ldt : nullptr;
    const float* sc = scale != nullptr ? scale + (size_t)tok * ldt : nullptr;
    const int per = N / G;                                       // this wave's columns
    for (int e0 = wave * per + l * 4; e0 < (wave + 1) * per; e0 += 32) {
        const float4 v = *reinterpret_cast<const float4*>(xr + e0);
        float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = (o[e] + nmean) * rstd;
            float r = fmaf(t, gamma != nullptr ? gamma[e0 + e] : 1.0f, beta != nullptr ? beta[e0 + e] : 0.0f);
            if (sc != nullptr) r = r * (1.0f + sc[e0 + e]) + sh[e0 + e];
            o[e] = r;
        }
        *reinterpret_cast<float4*>(yr + e0) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// attention row pass (fp32 flash kernel of ATen)
// ---------------------------------------------------------------------------------------------------------------------------------
// s [rows][Tk] scaled scores -> p (in place) un-normalised probabilities, each kv block of 512 relative to the running maximum after it;
// rescale [nb - 1][rows] = expf(max before block j - max after block j) for j >= 1; rowscale [rows] = 1 / sum.  16 threads per row:
// lane = key mod 16 sums its probabilities sequentially, then the 8 / 4 / 2 / 1 fold of vec_reduce_all.  Tk % 16 == 0.
// keys mlo .. mhi-1 are masked out (never computed, never read): score -inf, probability 0 written.
__global__ __launch_bounds__(256) void xe_softmax_kernel(float* __restrict__ s, float* __restrict__ rescale, float* __restrict__ rowscale, long rows, int Tk, int mlo, int mhi)
{
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long row = gid >> 4;
    const int l = (int)(gid & 15);
    if (row >= rows) return;
    float* sr = s + (size_t)row * Tk;
    float m_old = -__builtin_inff(), sum_old = 0.f;
    int jb = 0;
    for (int n0 = 0; n0 < Tk; n0 += 512, ++jb) {
        const int nb = min(512, Tk - n0), cnt = nb >> 4;
        float v[32];
        float bm = -__builtin_inff();
#pragma unroll
        for (int k = 0; k < 32; ++k) if (k < cnt) {
            const int key = n0 + 16 * k + l;
            v[k] = (key >= mlo && key < mhi) ? -__builtin_inff() : sr[key];
            bm = fmaxf(bm, v[k]);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) bm = fmaxf(bm, __shfl_xor(bm, o, WAVE));
        const float m_new = m_old > bm ? m_old : bm;
        if (m_new == -__builtin_inff()) {          // every key so far masked: ATen zero-fills the block's probabilities and leaves max / sum / accumulator alone
#pragma unroll
            for (int k = 0; k < 32; ++k) if (k < cnt) sr[n0 + 16 * k + l] = 0.f;
            if (jb > 0 && l == 0) rescale[(size_t)(jb - 1) * rows + row] = 1.0f;
            continue;
        }
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) if (k < cnt) { const float e = xe_exp_u20(v[k] - m_new); acc += e; sr[n0 + 16 * k + l] = e; }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc = acc + __shfl_xor(acc, o, WAVE);
        const float exp_tmp = xe_expf_glibc(m_old - m_new);
        sum_old = fmaf(exp_tmp, sum_old, acc);
        m_old = m_new;
        if (jb > 0 && l == 0) rescale[(size_t)(jb - 1) * rows + row] = exp_tmp;
    }
    if (l == 0) rowscale[row] = 1.0f / sum_old;
}

// v [B][T][*] (row stride vs, head h at column h D) -> vt [B][H][D][Tk] at key offset t_off: the P V product reads V as [d][key]
// T = key slots of this segment, `valid` of them present (rows valid .. T-1 are masked keys: zeros, never read), `rows` = rows per batch in v
__global__ void xe_transpose_v_kernel(const float* __restrict__ v, long vs, float* __restrict__ vt, int T, int valid, int rows, int H, int D, int Tk, int t_off)
{
    __shared__ float tile[32][33];
    const int z = blockIdx.z, b = z / H, h = z - b * H;
    const int t0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (t0 + r < T && d0 + tx < D) tile[r][tx] = (t0 + r < valid) ? v[((size_t)b * rows + t0 + r) * vs + h * D + d0 + tx] : 0.0f;
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (d0 + r < D && t0 + tx < T) vt[((size_t)z * D + d0 + r) * Tk + t_off + t0 + tx] = tile[tx][r];
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The same attention FUSED (round 6): one kernel per call, no score matrix in HBM (unfused: QK^T GEMM -> [B H, Tq, Tk] fp32 scores in a workspace -> row pass ->
// V transpose -> P V GEMM: four passes over 3.6 GB per joint block at B = 64, 19 % of the exact-order step).  head_dim 64, key slot counts multiples of 64.
//
// ATen's fp32 flash kernel takes the row maximum over a whole kv block of 512 keys before it exponentiates (p = exp_u20(s - max AFTER the block)), so a score is
// needed twice.  Keeping 128 rows x 512 scores per workgroup would take 256 KiB of LDS and 32 rows per workgroup would re-fetch K / V four times as often, so the
// scores are COMPUTED twice (a chain of 64 fp32 MFMAs gives the same bits both times): per kv block, sweep 1 walks the K tiles for the block maximum, sweep 2
// walks K and V tiles for probabilities, row sums and P V.  3 units of matrix work instead of 2, and nothing else leaves the CU.
//
//   workgroup = 4 waves = 128 query rows of one (sample, head); a wave owns 32 rows.  K / V tiles of 64 keys x 64 floats reach LDS by LDS-DMA (double buffered;
//   K rows XOR-swizzled on the source side for conflict-free ds_read_b128, V rows natural), 64 KiB per workgroup: two workgroups per CU.
//   S^T = K Q^T on v_mfma_f32_32x32x1_2b_f32, one instruction per d (the k-ascending chain from 0 that MKL's sgemm runs for K = 64), the two blocks = the two key
//   halves of the tile: lane (hh, i) ends with the scores of query i against keys 32 blk + (r & 3) + 8 (r >> 2) + 4 hh.  Scores * 1/sqrt(d), masked keys -inf.
//   Row pass in registers: exp_u20, the 16 key-class sums (class = key mod 16: a lane holds 8 of them, accumulated tile after tile in key order = the 16-lane
//   vector sum of ATen), folded 8 / 4 / 2 / 1 at the block end, sum = fma(expf(old max - new max), old sum, block sum) with glibc's expf.
//   O^T += V^T P^T on the same instruction, one per KEY in ascending order (the sequential chain over the keys), blocks = the two d halves; the probability of a key
//   sits in one lane half, v_permlane32_swap hands it to both.  Chains end every 256 keys (MKL's halves of a 512-key block): C += chain; C *= expf(old max - new
//   max) when a new kv block starts; out = C * (1 / sum).  Fully masked tiles are skipped: their probabilities are exact zeros and acc + 0 = acc.
// ---------------------------------------------------------------------------------------------------------------------------------
struct XfArgs {
    const float* q; long qs;
    const float *k1, *v1; long kvs1; int Tk1, valid1, rows1;
    const float *k2, *v2; long kvs2; int Tk2;
    float* out;
    int B, H, Tq, qtiles;
    float scale;
};

__device__ __forceinline__ void xf_dma16(const void* base, unsigned voff, unsigned lds)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory", "m0");
}

struct XfStep { int jb, sweep, t; };          // kv block, sweep (1: maximum, 2: probabilities + P V), staged key tile (64 slots) -- all uniform

__global__ __launch_bounds__(256, 2) void xe_fattn_kernel(XfArgs a)
{
    __shared__ __attribute__((aligned(1024))) float s_k[2][64 * 64];
    __shared__ __attribute__((aligned(1024))) float s_v[2][64 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, hh = lane >> 5;
    int qt, hd, b;
    {   // XCD-aware order: the q tiles of one (sample, head) share K / V -- they run on ONE XCD (consecutive entries of that XCD's eighth of the list)
        const int T = gridDim.x, orig = blockIdx.x;
        const int q8 = T >> 3, r8 = T & 7, xcd = orig & 7, idx = orig >> 3;
        const int w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
        qt = w % a.qtiles;
        hd = (w / a.qtiles) % a.H;
        b = w / (a.qtiles * a.H);
    }
    qt = __builtin_amdgcn_readfirstlane(qt); hd = __builtin_amdgcn_readfirstlane(hd); b = __builtin_amdgcn_readfirstlane(b);
    const int Tk = a.Tk1 + a.Tk2, nblk = (Tk + 511) >> 9;
    const int my_q = qt * 128 + wave * 32 + i;
    const bool row_ok = my_q < a.Tq;

    // this lane's half of its query row: d = 2 j + hh (v_mfma_f32_32x32x2_f32 takes the even k of a pair from lanes 0..31, the odd k from lanes 32..63 and adds
    // them in that order: fma(a1, b1, fma(a0, b0, acc)) -- a k-ascending chain, csrc/vq.hip)
    float qf[32];
    {
        const float* qp = a.q + ((size_t)b * a.Tq + (row_ok ? my_q : a.Tq - 1)) * a.qs + hd * 64 + hh;
#pragma unroll
        for (int j = 0; j < 32; ++j) qf[j] = qp[2 * j];
    }

    // ---- staging: a tile = 64 key rows x 256 bytes = 16 pieces of 4 rows; wave w issues pieces w, w + 4, w + 8, w + 12 of K (and of V in sweep 2) ----
    const unsigned lds_k = (unsigned)(size_t)(__attribute__((address_space(3))) float*)&s_k[0][0];
    const unsigned lds_v = (unsigned)(size_t)(__attribute__((address_space(3))) float*)&s_v[0][0];
    const int kr = lane >> 4, pos = lane & 15;                 // row inside a piece, 16-byte position inside the 256-byte row
    auto stage = [&](const XfStep& st, int buf) {
        const int slot0 = st.t * 64;
        const bool seg2 = slot0 >= a.Tk1;
        const float* kp = seg2 ? a.k2 : a.k1;
        const float* vp = seg2 ? a.v2 : a.v1;
        const long rs = seg2 ? a.kvs2 : a.kvs1;
        const int rows = seg2 ? a.Tk2 : a.rows1, nvis = seg2 ? a.Tk2 : a.valid1, key0 = seg2 ? slot0 - a.Tk1 : slot0;
        const char* kb = reinterpret_cast<const char*>(kp + (size_t)b * rows * rs + hd * 64);
        const char* vb = reinterpret_cast<const char*>(vp + (size_t)b * rows * rs + hd * 64);
        const unsigned rs4 = (unsigned)rs * 4u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = wave + 4 * u, row = 4 * p + kr;      // tile row 0..63
            const int src = min(key0 + row, nvis - 1);         // ragged tile: rows past the visible prefix fetch the last visible key (masked to -inf / probability 0 below)
            xf_dma16(kb, (unsigned)src * rs4 + (unsigned)((pos ^ (row & 15)) << 4), lds_k + buf * 16384 + p * 1024);
            if (st.sweep == 2) xf_dma16(vb, (unsigned)src * rs4 + (unsigned)(pos << 4), lds_v + buf * 16384 + p * 1024);
        }
    };
    auto visible = [&](int t) { const int s0 = t * 64; return s0 >= a.Tk1 || s0 < a.valid1; };
    auto advance = [&](XfStep st) {                             // successor of a step; jb == nblk: done
        for (;;) {
            ++st.t;
            const int tend = min((st.jb + 1) * 8, Tk >> 6);
            if (st.t >= tend) {
                if (st.sweep == 1) { st.sweep = 2; st.t = st.jb * 8 - 1; continue; }
                ++st.jb; st.sweep = 1; st.t = st.jb * 8 - 1;
                if (st.jb >= nblk) return st;
                continue;
            }
            if (visible(st.t)) return st;
        }
    };

    f32x16 o0, o1;                                              // the running chain: O^T[d = 32 dh + (r & 3) + 8 (r >> 2) + 4 hh][query i]
    float C0[16], C1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; C0[r] = 0.f; C1[r] = 0.f; }
    float m_old = -__builtin_inff(), sum_old = 0.f, m_new = -__builtin_inff(), lane_max = -__builtin_inff();
    float cls[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) cls[c] = 0.f;
    int chain = -1;
    bool block_live = false;                                     // sweep 2 of the current block has something to do (some key of it or before it is visible)

    // S^T of one 32-key half of the staged tile: rows (keys) 32 sub + i; this lane reads d = 4 c + hh and 4 c + 2 + hh of chunk c (one ds_read2_b32)
    const int ksw = i & 15;
    auto scores = [&](int buf, int sub, f32x16& sacc) {
        const float* sk = s_k[buf] + (32 * sub + i) * 64 + hh;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float* pc = sk + ((c ^ ksw) << 2);
            const float x0 = pc[0], x1 = pc[2];
            if (c == 0) { const f32x16 zero = {0}; sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, qf[0], zero, 0, 0, 0); }
            else sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, qf[2 * c], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, qf[2 * c + 1], sacc, 0, 0, 0);
        }
    };
    auto finish_scores = [&](int slot0, f32x16& sacc) {         // * 1/sqrt(d); keys past the visible prefix of a ragged half: -inf
        const int nvis = slot0 >= a.Tk1 ? 32 : a.valid1 - slot0;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = sacc[r] * a.scale;
        if (nvis < 32) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((r & 3) + 8 * (r >> 2) + 4 * hh >= nvis) sacc[r] = -__builtin_inff();
        }
    };
    auto fold_chain = [&]() {
#pragma unroll
        for (int r = 0; r < 16; ++r) { C0[r] = C0[r] + o0[r]; o0[r] = 0.f; C1[r] = C1[r] + o1[r]; o1[r] = 0.f; }
    };

    XfStep cur{0, 1, -1};
    cur = advance(cur);
    if (cur.jb >= nblk) return;                                  // (the launcher refuses calls without a visible key)
    stage(cur, 0);
    for (int s = 0; cur.jb < nblk; ++s) {
        const XfStep nxt = advance(cur);
        const int buf = s & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // my pieces of this step's tile have landed
        __syncthreads();                                         // everyone's have, and the other buffer (step s - 1) is free
        if (nxt.jb < nblk) stage(nxt, buf ^ 1);

        if (cur.sweep == 2 && block_live) {
            const int slot_in_blk = cur.t * 64 - cur.jb * 512;
            const int blen = min(512, Tk - cur.jb * 512);
            const int ch = 2 * cur.jb + ((blen == 512 && slot_in_blk >= 256) ? 1 : 0);
            if (ch != chain) { fold_chain(); chain = ch; }
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int slot0 = cur.t * 64 + 32 * sub;
            if (slot0 < a.Tk1 && slot0 >= a.valid1) continue;    // a fully masked half (uniform)
            if (cur.sweep == 2 && !block_live) continue;
            f32x16 sacc;
            scores(buf, sub, sacc);
            finish_scores(slot0, sacc);
            if (cur.sweep == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) lane_max = fmaxf(lane_max, sacc[r]);
                continue;
            }
            // probabilities, in place; the 8 key classes of this lane in key order: reg c (key < 16 of the half), then c + 8
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] = xe_exp_u20(sacc[r] - m_new);
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);       // four exponentials in flight, not sixteen
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) { cls[c] = cls[c] + sacc[c]; cls[c] = cls[c] + sacc[c + 8]; }
            // O^T += V^T P^T on key PAIRS in ascending order: the pair (k, k + 1) wants P[k] in lanes 0..31 and P[k + 1] in lanes 32..63; keys 8 g + {0..3} live in lane
            // half 0 (regs 4 g + {0..3}), 8 g + {4..7} in half 1 -- one v_permlane32_swap of regs (4 g + e, 4 g + e + 1) makes the operands of pairs (8 g + e, + 1) and (8 g + 4 + e, + 1)
            const float* sv = s_v[buf] + (32 * sub + hh) * 64 + i;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const auto w0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(sacc[4 * g]), __float_as_uint(sacc[4 * g + 1]), false, false);
                const auto w1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(sacc[4 * g + 2]), __float_as_uint(sacc[4 * g + 3]), false, false);
                const float px[4] = {__uint_as_float(w0[0]), __uint_as_float(w1[0]), __uint_as_float(w0[1]), __uint_as_float(w1[1])};   // pairs 8g+{0,1}, {2,3}, {4,5}, {6,7}
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = 8 * g + 2 * e;               // this lane's V row: key + hh
                    const float v0 = sv[key * 64], v1 = sv[key * 64 + 32];
                    o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, px[e], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, px[e], o1, 0, 0, 0);
                }
            }
        }
        if (cur.sweep == 1 && (nxt.jb != cur.jb || nxt.sweep != 1)) {        // the block's last tile of sweep 1: its maximum, the rescale of what was accumulated before it
            const float bm = fmaxf(lane_max, __shfl_xor(lane_max, 32, WAVE));
            m_new = m_old > bm ? m_old : bm;
            lane_max = -__builtin_inff();
            block_live = m_new != -__builtin_inff();             // every query of a call sees the same keys: uniform
            if (block_live && cur.jb > 0) {
                const float f = xe_expf_glibc(m_old - m_new);
#pragma unroll
                for (int r = 0; r < 16; ++r) { C0[r] = C0[r] * f; C1[r] = C1[r] * f; }
            }
        }
        if (cur.sweep == 2 && nxt.jb != cur.jb && block_live) {  // the kv block is done: its chains into C, its row sum into the running sum
            fold_chain();
            chain = -1;
            float s8[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) s8[j] = cls[j] + cls[j + 4];                         // lanes l, l ^ 8 of ATen's 16-lane vector
#pragma unroll
            for (int j = 0; j < 4; ++j) s8[j] = s8[j] + __shfl_xor(s8[j], 32, WAVE);        // l ^ 4: the other lane half
            const float u0 = s8[0] + s8[2], u1 = s8[1] + s8[3];                              // l ^ 2
            const float tot = u0 + u1;                                                       // l ^ 1
            sum_old = fmaf(xe_expf_glibc(m_old - m_new), sum_old, tot);
            m_old = m_new;
#pragma unroll
            for (int c = 0; c < 8; ++c) cls[c] = 0.f;
        }
        cur = nxt;
    }

    // out[b][q][hd * 64 + d] = C * (1 / sum): lane (hh, i) holds d = 32 dh + 8 g + 4 hh + j in reg 4 g + j of half dh
    if (row_ok) {
        const float inv = 1.0f / sum_old;
        float* op = a.out + ((size_t)b * a.Tq + my_q) * ((size_t)a.H * 64) + hd * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<float4*>(op + 8 * g + 4 * hh) = make_float4(C0[4 * g] * inv, C0[4 * g + 1] * inv, C0[4 * g + 2] * inv, C0[4 * g + 3] * inv);
            *reinterpret_cast<float4*>(op + 32 + 8 * g + 4 * hh) = make_float4(C1[4 * g] * inv, C1[4 * g + 1] * inv, C1[4 * g + 2] * inv, C1[4 * g + 3] * inv);
        }
    }
}

}  // namespace selftok

using namespace selftok;

extern "C" {

int selftok_ex_linear_f32(const float* x, long ldx, const float* w, const float* bias, const float* res, long ldr, int res_mod, const float* gate,
                          long ldg, int gate_mod, float* out, long ldo, long M, int N, int K, int gelu, hipStream_t stream)
{
    if (M == 0) return SELFTOK_OK;
    if (!x || !w || !out || M < 0 || M > 0x7fffffffL || N <= 0 || K <= 0 || K % 16 || ldx % 4 || ldx < K || ldo < N || (gate && !res)) {
        set_last_error("ex_linear: need K % 16 == 0, 16-byte aligned rows (ldx % 4 == 0), gate only with res"); return SELFTOK_EINVAL;
    }
    XeGemmArgs g{};
    g.a = x; g.lda = ldx; g.b = w; g.ldb = K; g.c = out; g.ldc = ldo; g.bias = bias;
    g.res = res; g.ldr = ldr; g.res_mod = res_mod; g.gate = gate; g.ldg = ldg; g.gate_mod = gate_mod;
    g.M = (int)M; g.N = N; g.K = K; g.H = 1; g.mode = 0; g.gelu = gelu;
    g.nblk = mkl_blocks(K, 0, g.blk_end, 0);
    if (g.nblk < 0) { set_last_error("ex_linear: K too large (more than 16 K-blocks)"); return SELFTOK_EINVAL; }
    for (int j = 0; j < g.nblk; ++j) if (g.blk_end[j] % 4) { set_last_error("ex_linear: a K-block boundary is not a multiple of 4"); return SELFTOK_EINVAL; }
    return launch_xe_gemm(g, 1, stream);
}

int selftok_ex_layernorm_mod_f32(const float* x, long ldx, float* out, long ldo, const float* shift, const float* scale, long ldt, int T, const float* gamma,
                                 const float* beta, float* stats, long rows, int N, float eps, hipStream_t stream)
{
    if (rows == 0) return SELFTOK_OK;
    if (!x || !out || rows < 0 || N <= 0 || N % 8 || N > 4096 || ldx % 4 || ldo % 4 || ((shift == nullptr) != (scale == nullptr)) || (scale && T == 0)) {
        set_last_error("ex_layernorm: need N % 8 == 0, N <= 4096, 16-byte aligned rows, shift and scale together"); return SELFTOK_EINVAL;
    }
    if (N % 512 == 0 && N >= 1024 && rows >= 2048) {           // wide rows of a large batch: N / 512 waves per row (the MMDiT's 1536)
        hipLaunchKernelGGL(xe_lnw_kernel<false>, dim3((unsigned)((rows + 7) / 8)), dim3(64 * (N / 512)), 0, stream, x, ldx, out, ldo, shift, scale, ldt, T != 0 ? T : 1, gamma, beta,
                           rows, N, eps, stats, XeLnFuse{});
        return check_launch("xe_lnw_kernel");
    }
    const long threads = rows * 8;
    hipLaunchKernelGGL(xe_ln_kernel<false>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, x, ldx, out, ldo, shift, scale, ldt, T != 0 ? T : 1, gamma, beta,
                       rows, N, eps, stats, XeLnFuse{});
    return check_launch("xe_ln_kernel");
}

int selftok_ex_res_layernorm_mod_f32(const float* x, long ldx, const float* lin, long ldl, const float* lin_bias, const float* gate, long ldg, int gate_mod,
                                     float* x_out, long ldxo, float* out, long ldo, const float* shift, const float* scale, long ldt, int T, long rows, int N, float eps,
                                     hipStream_t stream)
{
    if (rows == 0) return SELFTOK_OK;
    if (!x || !lin || !x_out || !out || rows < 0 || N <= 0 || N % 8 || N > 4096 || ldx % 4 || ldl % 4 || ldxo % 4 || ldo % 4 || ((shift == nullptr) != (scale == nullptr)) ||
        (scale && T == 0)) {
        set_last_error("ex_res_layernorm: need N % 8 == 0, N <= 4096, 16-byte aligned rows, shift and scale together"); return SELFTOK_EINVAL;
    }
    const long threads = rows * 8;
    XeLnFuse f{lin, ldl, gate, ldg, gate_mod, lin_bias, x_out, ldxo};
    if (N % 512 == 0 && N >= 1024 && rows >= 2048) {
        hipLaunchKernelGGL(xe_lnw_kernel<true>, dim3((unsigned)((rows + 7) / 8)), dim3(64 * (N / 512)), 0, stream, x, ldx, out, ldo, shift, scale, ldt, T != 0 ? T : 1,
                           (const float*)nullptr, (const float*)nullptr, rows, N, eps, (float*)nullptr, f);
        return check_launch("xe_lnw_kernel<fused residual update>");
    }
    hipLaunchKernelGGL(xe_ln_kernel<true>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, x, ldx, out, ldo, shift, scale, ldt, T != 0 ? T : 1,
                       (const float*)nullptr, (const float*)nullptr, rows, N, eps, (float*)nullptr, f);
    return check_launch("xe_ln_kernel<fused residual update>");
}

int selftok_ex_unary_f32(const float* x, float* y, long n, int mode, hipStream_t stream)
{
    if (n == 0) return SELFTOK_OK;
    if (!x || !y || n < 0 || mode < 0 || mode > 4) { set_last_error("ex_unary: bad argument"); return SELFTOK_EINVAL; }
    hipLaunchKernelGGL(xe_unary_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, stream, x, y, n, mode);
    return check_launch("xe_unary_kernel");
}

size_t selftok_ex_attention_workspace_bytes(int B, int H, int Tq, int Tk, int D)
{
    if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0 || D <= 0) return 0;
    const size_t rows = (size_t)B * H * Tq;
    const int nb = (Tk + 511) / 512;
    return rows * Tk * 4 + (size_t)B * H * D * Tk * 4 + rows * 4 * (size_t)(nb > 1 ? nb - 1 : 1) + rows * 4;
}

/* q [B][Tq][..] row stride qs; k1 / v1 [B][rows1][..] row stride kvs1 holding the first valid1 of the segment's Tk1 key slots (valid1 == rows1 == Tk1: no
 * mask); optional second key / value segment k2 / v2 [B][Tk2][..] row stride kvs2 (`torch.cat([k, query_k], dim=2)`); head h at column h D of every row;
 * out [B][Tq][H D] contiguous. */
int selftok_ex_attention_f32(const float* q, long qs, const float* k1, const float* v1, long kvs1, int Tk1, int valid1, int rows1, const float* k2, const float* v2,
                             long kvs2, int Tk2, float* out, void* workspace, int B, int H, int Tq, int D, hipStream_t stream)
{
    if (B == 0) return SELFTOK_OK;
    const int Tk = Tk1 + Tk2;
    if (!q || !out || !workspace || B < 0 || H <= 0 || Tq <= 0 || Tk1 <= 0 || Tk2 < 0 || valid1 < 0 || valid1 > Tk1 || rows1 < valid1 || (valid1 > 0 && (!k1 || !v1)) ||
        (Tk2 > 0 && (!k2 || !v2)) || D % 16 || D <= 0 || D > 128 || Tk1 % 16 || Tk2 % 16 || qs % 4 || kvs1 % 4 || kvs2 % 4 || (valid1 == 0 && Tk2 == 0)) {
        set_last_error("ex_attention: need head_dim % 16 == 0 (<= 128), key slot counts % 16 == 0, 16-byte aligned rows, 0 <= valid1 <= Tk1 <= ..., at least one visible key");
        return SELFTOK_EINVAL;
    }
    const int Z = B * H;
    const size_t rows = (size_t)Z * Tq;
    const int nb = (Tk + 511) / 512;
    float* s = (float*)workspace;
    float* vt = s + rows * Tk;
    float* rescale = vt + (size_t)Z * D * Tk;
    float* rowscale = rescale + rows * (size_t)(nb > 1 ? nb - 1 : 1);
    // scores, one launch per key segment: rows = queries, columns = keys, one chain over head_dim, * 1/sqrt(D); masked keys are not computed
    for (int seg = 0; seg < (Tk2 > 0 ? 2 : 1); ++seg) {
        if (seg == 0 && valid1 == 0) continue;
        XeGemmArgs g{};
        g.a = q; g.lda = qs; g.a_bs = (long)Tq * qs; g.a_hs = D;
        g.b = seg ? k2 : k1; g.ldb = seg ? kvs2 : kvs1; g.b_bs = (long)(seg ? Tk2 : rows1) * g.ldb; g.b_hs = D;
        g.c = s + (seg ? Tk1 : 0); g.ldc = Tk; g.c_bs = (long)H * Tq * Tk; g.c_hs = (long)Tq * Tk;
        g.M = Tq; g.N = seg ? Tk2 : valid1; g.K = D; g.H = H; g.mode = 1; g.out_scale = (float)(1.0 / sqrt((double)D));
        g.nblk = mkl_blocks(D, 0, g.blk_end, 0);
        int rc = launch_xe_gemm(g, Z, stream);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(xe_softmax_kernel, dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, stream, s, rescale, rowscale, (long)rows, Tk, valid1, Tk1);
    int rc = check_launch("xe_softmax_kernel");
    if (rc) return rc;
    for (int seg = 0; seg < (Tk2 > 0 ? 2 : 1); ++seg) {
        const int T = seg ? Tk2 : Tk1;
        hipLaunchKernelGGL(xe_transpose_v_kernel, dim3((T + 31) / 32, (D + 31) / 32, Z), dim3(256), 0, stream, seg ? v2 : v1, seg ? kvs2 : kvs1, vt, T, seg ? Tk2 : valid1,
                           seg ? Tk2 : rows1, H, D, Tk, seg ? Tk1 : 0);
        rc = check_launch("xe_transpose_v_kernel");
        if (rc) return rc;
    }
    // P V: reduction over the key SLOTS; every kv block of 512 is one MKL call (K = block length -> its own K-blocks), C *= rescale between them
    XeGemmArgs g{};
    g.a = s; g.lda = Tk; g.a_bs = (long)H * Tq * Tk; g.a_hs = (long)Tq * Tk;
    g.b = vt; g.ldb = Tk; g.b_bs = (long)H * D * Tk; g.b_hs = (long)D * Tk;
    g.c = out; g.ldc = (long)H * D; g.c_bs = (long)Tq * H * D; g.c_hs = D;
    g.M = Tq; g.N = D; g.K = Tk; g.H = H; g.mode = 2; g.rescale = rescale; g.rowscale = rowscale;
    int n = 0;
    for (int n0 = 0; n0 < Tk; n0 += 512) {
        if (n0 > 0) g.rescale_mask |= 1u << n;
        n = mkl_blocks(Tk - n0 < 512 ? Tk - n0 : 512, n0, g.blk_end, n);
        if (n < 0) { set_last_error("ex_attention: too many keys (more than 16 K-blocks)"); return SELFTOK_EINVAL; }
    }
    g.nblk = n;
    for (int j = 0; j < n; ++j) if (g.blk_end[j] % 4) { set_last_error("ex_attention: a K-block boundary is not a multiple of 4"); return SELFTOK_EINVAL; }
    return launch_xe_gemm(g, Z, stream);
}

/* The same attention in ONE kernel (xe_fattn_kernel above): no workspace, no score matrix in HBM; bit-identical to selftok_ex_attention_f32.  head_dim 64, key
 * slot counts multiples of 64, and a last kv block (Tk mod 512) of at most 384 keys (one MKL K-block) -- else SELFTOK_EINVAL: use the unfused entry. */
int selftok_ex_attention_fused_supported(int Tk1, int Tk2, int D)
{
    const int Tk = Tk1 + Tk2, last = Tk & 511;
    return D == 64 && Tk1 >= 0 && Tk2 >= 0 && Tk > 0 && (Tk1 & 63) == 0 && (Tk2 & 63) == 0 && (last == 0 || last <= 384);
}

int selftok_ex_attention_fused_f32(const float* q, long qs, const float* k1, const float* v1, long kvs1, int Tk1, int valid1, int rows1, const float* k2, const float* v2,
                                   long kvs2, int Tk2, float* out, int B, int H, int Tq, int D, hipStream_t stream)
{
    if (B == 0) return SELFTOK_OK;
    if (!q || !out || B < 0 || H <= 0 || Tq <= 0 || Tk1 < 0 || Tk2 < 0 || valid1 < 0 || valid1 > Tk1 || rows1 < valid1 || (valid1 > 0 && (!k1 || !v1)) || (Tk2 > 0 && (!k2 || !v2)) ||
        qs % 4 || kvs1 % 4 || kvs2 % 4 || (valid1 == 0 && Tk2 == 0) || !selftok_ex_attention_fused_supported(Tk1, Tk2, D) ||
        (size_t)(rows1 > Tk2 ? rows1 : Tk2) * (size_t)(kvs1 > kvs2 ? kvs1 : kvs2) * 4 > 0xffffffffull) {
        set_last_error("ex_attention_fused: need head_dim 64, key slot counts % 64 == 0, a last kv block of <= 384 keys, 16-byte aligned rows, at least one visible key");
        return SELFTOK_EINVAL;
    }
    XfArgs a{};
    a.q = q; a.qs = qs; a.k1 = k1; a.v1 = v1; a.kvs1 = kvs1; a.Tk1 = Tk1; a.valid1 = valid1; a.rows1 = rows1;
    a.k2 = k2; a.v2 = v2; a.kvs2 = kvs2; a.Tk2 = Tk2; a.out = out; a.B = B; a.H = H; a.Tq = Tq; a.qtiles = (Tq + 127) / 128;
    a.scale = (float)(1.0 / sqrt((double)D));
    hipLaunchKernelGGL(xe_fattn_kernel, dim3((unsigned)(B * H * a.qtiles)), dim3(256), 0, stream, a);
    return check_launch("xe_fattn_kernel");
}

}  // extern "C"
