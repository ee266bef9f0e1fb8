// Selftok VQ nearest-code lookup for gfx950 (MI355X).
//
// Replaces, on device, the eval path of the reference's
//   VectorQuantize.forward -> l2norm -> CosineSimCodebook.forward
//   (mimogpt/models/selftok/vector_quantize_pytorch.py:854, 561, 125-143)
// i.e. ids[n] = argmax_c <l2norm(z[n]), codebook[c]>, WITHOUT materialising the [N,C] score
// matrix or the two [N,C] one-hot tensors the reference builds (:561, :136, :969).
//
// Bit-exactness contract (checked against oracle/selftok_oracle.c, which is pinned to the
// reference on torch-CPU):
//   * l2norm: a_j = fma(z[j+8],z[j+8], z[j]*z[j]) (j<8), s = ((a0+a1)+...)+a7,
//             x = z / max(sqrt(s), 1e-12) with correctly rounded sqrt and divide;
//   * score : s = 0; for k in 0..15: s = fma(x[k], e[k], s)   (k-ordered fp32 FMA chain);
//   * argmax: first maximal index; a NaN score is the maximum and the first NaN wins.
// Two kernels implement the same arithmetic:
//   vq_valu_kernel : fp32 VALU.  Code tiles are staged coalesced HBM->LDS, each lane keeps 4
//                    whole codes in registers, the wave's rows are read as LDS broadcasts, every
//                    lane keeps a running (best,idx) per row and the 64 lanes are combined with a
//                    wave-shuffle reduction at the end.
//   vq_mfma_kernel : v_mfma_f32_32x32x2_f32.  On gfx950 the fp32-input MFMA is bit-for-bit the
//                    same k-ordered FMA chain at the fp32 vector rate, which frees the VALU for the
//                    argmax bookkeeping (~2x the VALU kernel).  Needs the codebook re-laid once
//                    into MFMA fragment order (vq_pack_kernel).
// Both write one 64-bit key per (code-split, row): (orderable(best) << 32) | ~idx, so that an
// unsigned max picks the larger score and, on ties, the lower index.  vq_finalize_kernel
// reduces the splits and emits ids (int64, the reference's dtype, or int32) and the top-1 score.
//
// This file is compiled with -ffp-contract=off: every FMA below is explicit.
#include "common.h"
#include "selftok_hip.h"   // the C ABI declared there must match the definitions below
#include <stdlib.h>

namespace selftok {

constexpr int D = 16;
constexpr uint32_t KEY_NAN = 0xFFFFFFFFu;

struct Best {
    float v;    // running best score; +inf once the best is a NaN
    int i;      // its code index
    bool nan;   // best is NaN (first NaN wins; never updated afterwards)
};

__device__ __forceinline__ void best_init(Best& b, int first_idx)
{
    b.v = -__builtin_inff();
    b.i = first_idx;
    b.nan = false;
}

// hot-path update: valid when the score is known to be non-NaN or the row state is already NaN
__device__ __forceinline__ void best_upd_fast(Best& b, float s, int idx)
{
    bool g = s > b.v;
    b.v = g ? s : b.v;
    b.i = g ? idx : b.i;
}

// exact update incl. NaN ("NaN is the maximum, first NaN wins" == torch.argmax on CPU)
__device__ __forceinline__ void best_upd_exact(Best& b, float s, int idx)
{
    if (!b.nan) {
        if (s != s) { b.nan = true; b.v = __builtin_inff(); b.i = idx; }
        else if (s > b.v) { b.v = s; b.i = idx; }
    }
}

__device__ __forceinline__ unsigned long long best_key(const Best& b)
{
    float v = b.v;
    if (v == 0.0f) v = 0.0f;  // -0 -> +0: torch treats them as equal, the lower index must win
    uint32_t hi = b.nan ? KEY_NAN : f32_orderable(v);
    return ((unsigned long long)hi << 32) | (uint32_t)(~(uint32_t)b.i);
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long k, int o)
{
    uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
    lo = __shfl_xor(lo, o, WAVE);
    hi = __shfl_xor(hi, o, WAVE);
    return ((unsigned long long)hi << 32) | lo;
}

// canonical l2norm of one 16-float row (see header)
__device__ __forceinline__ void l2norm16(const float (&z)[D], float (&x)[D])
{
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = __builtin_fmaf(z[j + 8], z[j + 8], z[j] * z[j]);
    float s = a[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) s = s + a[j];
    float nrm = __builtin_sqrtf(s);   // correctly rounded (refined v_sqrt); __fsqrt_rn is the raw 1-ulp v_sqrt_f32 on gfx950
    nrm = (nrm > 1e-12f) ? nrm : 1e-12f;
    if (s != s) nrm = s;
#pragma unroll
    for (int k = 0; k < D; ++k) x[k] = z[k] / nrm;   // IEEE divide (div_scale/div_fmas/div_fixup)
}

__device__ __forceinline__ void load_row16(const float* __restrict__ p, float (&z)[D])
{
    const float4* p4 = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 t = p4[q];
        z[4 * q + 0] = t.x; z[4 * q + 1] = t.y; z[4 * q + 2] = t.z; z[4 * q + 3] = t.w;
    }
}

// a value that can make a score non-finite: NaN/inf or absurdly large
__device__ __forceinline__ bool suspicious(float v) { return !(fabsf(v) < 1.0e18f); }

// ---------------------------------------------------------------------------------------
// VALU kernel
// ---------------------------------------------------------------------------------------
constexpr int V_ROWS = 16;             // rows per wave
constexpr int V_CL = 4;                // codes per lane per tile
constexpr int V_TILE = WAVE * V_CL;    // 256 codes per tile
constexpr int V_STRIDE = 20;           // padded LDS row stride (floats): 80 B keeps 16-B alignment, spreads banks

__global__ __launch_bounds__(256) void vq_valu_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                      unsigned long long* __restrict__ partial, int N, int C,
                                                      int tiles_per_split, int normalize)
{
    __shared__ __attribute__((aligned(16))) float s_code[2][V_TILE * V_STRIDE];   // 2 x 20 KB
    __shared__ __attribute__((aligned(16))) float s_x[4][V_ROWS * D];             // 4 KB

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = (blockIdx.x * 4 + wave) * V_ROWS;

    // ---- rows of this wave: lanes 0..15 normalise one row each into LDS ----
    bool xbad = false;
    if (lane < V_ROWS) {
        float zz[D], xx[D];
        int r = row0 + lane;
        if (r < N) load_row16(z + (size_t)r * D, zz);
        else {
#pragma unroll
            for (int k = 0; k < D; ++k) zz[k] = 0.f;
        }
        if (normalize) l2norm16(zz, xx);
        else {
#pragma unroll
            for (int k = 0; k < D; ++k) xx[k] = zz[k];
        }
#pragma unroll
        for (int k = 0; k < D; ++k) { s_x[wave][lane * D + k] = xx[k]; xbad |= suspicious(xx[k]); }
    }
    const bool wave_xbad = __any(xbad);

    Best best[V_ROWS];
    const int tile_first = blockIdx.y * tiles_per_split;
    const int ntiles_total = (C + V_TILE - 1) / V_TILE;
    int tile_last = tile_first + tiles_per_split;
    if (tile_last > ntiles_total) tile_last = ntiles_total;
#pragma unroll
    for (int r = 0; r < V_ROWS; ++r) best_init(best[r], tile_first * V_TILE);

    // stage one tile (256 codes x 16 floats = 1024 float4) coalesced: 4 float4 per thread
    auto stage = [&](int tile, int buf) {
        const float4* src = reinterpret_cast<const float4*>(cb) + (size_t)tile * V_TILE * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int f4 = q * 256 + tid;           // float4 index inside the tile
            int code = f4 >> 2, part = f4 & 3;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tile * V_TILE + code < C) v = src[f4];
            *reinterpret_cast<float4*>(&s_code[buf][code * V_STRIDE + part * 4]) = v;
        }
    };

    if (tile_first < tile_last) stage(tile_first, 0);
    __syncthreads();

    for (int tile = tile_first; tile < tile_last; ++tile) {
        const int buf = (tile - tile_first) & 1;
        if (tile + 1 < tile_last) stage(tile + 1, buf ^ 1);

        // my 4 codes -> registers
        float e[V_CL][D];
        bool ebad = false;
#pragma unroll
        for (int c = 0; c < V_CL; ++c) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 t = *reinterpret_cast<const float4*>(&s_code[buf][(lane * V_CL + c) * V_STRIDE + q * 4]);
                e[c][4 * q + 0] = t.x; e[c][4 * q + 1] = t.y; e[c][4 * q + 2] = t.z; e[c][4 * q + 3] = t.w;
            }
#pragma unroll
            for (int k = 0; k < D; ++k) ebad |= suspicious(e[c][k]);
        }
        const int code0 = tile * V_TILE + lane * V_CL;
        const bool slow = wave_xbad || __any(ebad);
        const bool tail = (tile + 1) * V_TILE > C;

#pragma unroll
        for (int r = 0; r < V_ROWS; ++r) {
            float x[D];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 t = *reinterpret_cast<const float4*>(&s_x[wave][r * D + q * 4]);   // LDS broadcast
                x[4 * q + 0] = t.x; x[4 * q + 1] = t.y; x[4 * q + 2] = t.z; x[4 * q + 3] = t.w;
            }
            float s[V_CL];
#pragma unroll
            for (int c = 0; c < V_CL; ++c) s[c] = 0.f;
#pragma unroll
            for (int k = 0; k < D; ++k)
#pragma unroll
                for (int c = 0; c < V_CL; ++c) s[c] = __builtin_fmaf(x[k], e[c][k], s[c]);
            if (!slow && !tail) {
#pragma unroll
                for (int c = 0; c < V_CL; ++c) best_upd_fast(best[r], s[c], code0 + c);
            } else {
#pragma unroll
                for (int c = 0; c < V_CL; ++c)
                    if (code0 + c < C) best_upd_exact(best[r], s[c], code0 + c);
            }
        }
        __syncthreads();
    }

    // ---- wave-shuffle reduction of the 64 per-lane candidates of every row ----
#pragma unroll
    for (int r = 0; r < V_ROWS; ++r) {
        unsigned long long k = best_key(best[r]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            unsigned long long other = shfl_xor_u64(k, o);
            k = other > k ? other : k;
        }
        if (lane == 0 && row0 + r < N) partial[(size_t)blockIdx.y * N + row0 + r] = k;
    }
}

// ---------------------------------------------------------------------------------------
// MFMA kernel
// ---------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

// coarse pass of the f16 path: both operands are pre-multiplied by 2^7 (exact) so that the fp16 residuals of ordinary components
// of unit vectors stay in the fp16 normal range; scores come out multiplied by 2^14.
constexpr float F16_PRESCALE = 128.0f;
constexpr float F16_SCORE_SCALE = F16_PRESCALE * F16_PRESCALE;
// |coarse - canonical| in ORIGINAL score units for unit-norm rows and codes (sum |x_k e_k| <= 1): the two operand representations
// and the dropped lo x lo term 3 x 2^-22 = 7.2e-7, the fp32 accumulation inside three chained MFMAs <= 7e-7 (a few ulp of a sum
// <= 1), the canonical chain's own 16 roundings <= 9.5e-7: < 2.4e-6 in the worst case.  The window is 3x that; measured maximum
// over 1.7e7 scores: 3.0e-7 (tests/test_vq_gpu.py::test_vq_coarse_pass_error_bound_and_adversarial_near_ties).
constexpr float F16_EPS = 7.62939453125e-06f;          // 2^-17
// ONE-MFMA coarse pass (round 4, SELFTOK_VQ_F16COARSE1): only hi x hi.  |x_k e_k - x0_k e0_k| <= |x_k| |e_k - e0_k| + |e0_k| |x_k - x0_k|
// <= |x_k e_k| (2^-11 + 2^-11 (1 + 2^-11)) (fp16 keeps 11 significand bits: relative rounding error <= 2^-11), summed with Cauchy-Schwarz
// over unit vectors (norm^2 <= 1.01): <= 9.86e-4; + the fp32 accumulation of one MFMA and the canonical chain's own roundings (< 2e-6)
// + fp16 subnormals of 2^7-scaled components below 5e-7 (< 1e-8).  The window constant is 17 x 2^-14 = 1.0376e-3, 5 % above that.
// Three times fewer MFMAs for a window 136x wider: 1.08 instead of 1.00 candidate streams per row and 0.4 % of the rows with a stream
// whose two best tiles are both inside the window (whole-stream exact re-scan) on the synthetic features -- a longer finalize.
constexpr float F16_EPS1 = 0.00103759765625f;
// the bound above holds for |x|^2, |e|^2 <= 1; up to 1.01 it grows by 1 % (the window has a 3x margin).  Rows / code books beyond
// it are flagged and take the exact scan (nothing enforced the unit-norm premise of the window).
constexpr float F16_NORM2_MAX = 1.01f;

// packed layout: tile t (32 codes) = 512 floats = [part 0..1][lane 0..63][4 floats]; lane l = (h = l>>5, i = l&31)
// owns e[t*32+i][2m+h] for m = 4*part + j -- exactly the A fragments of the 8 chained 32x32x2 MFMAs, stored so that
// one wave reads (or DMAs into LDS) a whole 1 KiB (tile, part) piece with 16 B per lane, conflict-free.
__device__ __forceinline__ int packed_offset(int i /*code in tile*/, int k /*element*/)
{
    const int m = k >> 1, lane = (k & 1) * 32 + i;
    return (m >> 2) * 256 + lane * 4 + (m & 3);
}
// The word after the last tile (packed[C*16], zeroed by the caller) becomes non-zero if any code element could make a
// score non-finite; the MFMA kernel then runs its exact NaN-aware scan instead of the fast one.
__global__ void vq_pack_kernel(const float* __restrict__ cb, float* __restrict__ packed, int C)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (code, k)
    if (g >= C * D) return;
    int c = g >> 4, k = g & 15;
    int t = c >> 5, i = c & 31;
    const float v = cb[g];
    packed[(size_t)t * 512 + packed_offset(i, k)] = v;
    // second image for the coarse pass (vq_f16_kernel): e * 2^7 split into fp16 hi + fp16 lo, in the A-operand order of
    // v_mfma_f32_32x32x16_f16: tile t = [plane][lane = (k>>3)*32 + i][k & 7]
    _Float16* p16 = reinterpret_cast<_Float16*>(packed + (size_t)C * D + 64) + (size_t)t * 1024;
    const float vs = v * F16_PRESCALE;
    const _Float16 hi = (_Float16)vs;
    p16[((k >> 3) * 32 + i) * 8 + (k & 7)] = hi;
    p16[512 + ((k >> 3) * 32 + i) * 8 + (k & 7)] = (_Float16)(vs - (float)hi);
    if (suspicious(v)) atomicOr(reinterpret_cast<unsigned int*>(packed) + (size_t)C * D, 1u);
    if (!(fabsf(vs) < 60000.f)) atomicOr(reinterpret_cast<unsigned int*>(packed) + (size_t)C * D, 2u);     // outside the fp16 range
    // F16_EPS is derived for unit-norm codes (sum |x_k e_k| <= |x| |e| <= 1): a code book that is not l2-normalised (the
    // reference's always is, vector_quantize_pytorch.py:452,605) makes the coarse error scale with |e| and could leave the
    // window -> flag it, every kernel of the coarse path then takes its exact scan (the fp32 kernels need no such bound)
    if (k == 0) {
        float n2 = 0.f;
#pragma unroll
        for (int j = 0; j < D; ++j) n2 = __builtin_fmaf(cb[(size_t)c * D + j], cb[(size_t)c * D + j], n2);
        if (!(n2 <= F16_NORM2_MAX)) atomicOr(reinterpret_cast<unsigned int*>(packed) + (size_t)C * D, 4u);
    }
}

// running best of one lane for one x-row, MFMA flavour: the index is kept as (tile, register slot) so that the
// per-score update is cmp + 2 cndmask with inline constants; code = tile*32 + (slot&3) + 8*(slot>>2) + 4*half.
struct BestT {
    float v;     // +inf once the best is a NaN
    int tile;
    int slot;
    bool nan;
};

// fast scan: only (max value, tile) are tracked -- 7 v_max3 + cmp + 2 cndmask per 16 scores instead of 48 ops.
// On gfx950 the fp32-input MFMA runs on the fp32 VALU lanes (measured: every VALU op costs ~3-4 MFMA-pipe cycles,
// tools/microbench/mfma_valu.hip), so scan instructions are not free: the slot inside the winning tile is
// recovered later by vq_finalize_packed_kernel, which recomputes the 16 candidate scores of that (tile, half).
__device__ __forceinline__ void scanmax(BestT& b, const f32x16& acc, int tile)
{
#ifdef SELFTOK_VQ_ABLATE_SCAN      // tools/ ablation builds only: keep the accumulators live, skip the scan
    asm volatile("" ::"v"(acc));
    b.tile = tile;
    return;
#endif
    // v_max3_f32 spelled out: `fmaxf` on MFMA results makes hipcc emit a canonicalising `v_max_f32 x,x` per chain
    // (3 of 13 scan ops).  Inputs here are finite by the fast-path precondition.  The asm reads an accumulator set whose
    // MFMAs were issued a full tile (>= 16 MFMAs) earlier -- the caller pins that order with sched_barrier -- so the
    // MFMA-write -> VALU-read wait states hipcc cannot see inside asm are satisfied by construction.
    auto max3 = [](float a, float bb, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(bb), "v"(c)); return r; };
    const float m0 = max3(acc[0], acc[1], acc[2]), m1 = max3(acc[3], acc[4], acc[5]), m2 = max3(acc[6], acc[7], acc[8]);
    const float m3 = max3(acc[9], acc[10], acc[11]), m4 = max3(acc[12], acc[13], acc[14]);
    const float m5 = max3(m0, m1, m2), m6 = max3(m3, m4, acc[15]);
    const float m = max3(m5, m6, b.v);   // includes the running best: g below is "strictly improved"
    const bool g = m > b.v;              // strict: the earliest tile holding the maximum wins
    b.v = m;
    b.tile = g ? tile : b.tile;
}

// same scan with compiler-visible `fmaxf` (hipcc inserts the MFMA->VALU wait states itself): used where the scan directly
// follows the MFMAs of the SAME accumulators (ragged last chunk), where the asm version above would be unsafe.
__device__ __forceinline__ void scanmax_safe(BestT& b, const f32x16& acc, int tile)
{
    float m0 = __builtin_fmaxf(__builtin_fmaxf(acc[0], acc[1]), acc[2]);
    float m1 = __builtin_fmaxf(__builtin_fmaxf(acc[3], acc[4]), acc[5]);
    float m2 = __builtin_fmaxf(__builtin_fmaxf(acc[6], acc[7]), acc[8]);
    float m3 = __builtin_fmaxf(__builtin_fmaxf(acc[9], acc[10]), acc[11]);
    float m4 = __builtin_fmaxf(__builtin_fmaxf(acc[12], acc[13]), acc[14]);
    float m5 = __builtin_fmaxf(__builtin_fmaxf(m0, m1), m2);
    float m6 = __builtin_fmaxf(__builtin_fmaxf(m3, m4), acc[15]);
    const float m = __builtin_fmaxf(m5, m6);
    const bool g = m > b.v;
    b.v = g ? m : b.v;
    b.tile = g ? tile : b.tile;
}

template <bool EXACT>
__device__ __forceinline__ void scan16(BestT& b, const f32x16& acc, int tile)
{
    const float before = b.v;
    const bool nan_before = b.nan;
    int slot = b.slot;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float s = acc[r];
        if (EXACT) {
            if (!b.nan) {
                if (s != s) { b.nan = true; b.v = __builtin_inff(); slot = r; }
                else if (s > b.v) { b.v = s; slot = r; }
            }
        } else {
            const bool g = s > b.v;
            b.v = g ? s : b.v;
            slot = g ? r : slot;
        }
    }
    // strict '>' means the value changed iff some slot of this tile won (or the first NaN appeared)
    const bool changed = EXACT ? ((b.v != before) || (b.nan != nan_before)) : (b.v != before);
    b.tile = changed ? tile : b.tile;
    b.slot = slot;
}

constexpr int M_CH = 8;                 // code tiles per LDS chunk: 8 x 2 KiB = 16 KiB, double buffered

template <int RT>
__global__ __launch_bounds__(256) void vq_mfma_kernel(const float* __restrict__ z, const float* __restrict__ packed,
                                                      unsigned long long* __restrict__ partial, int N, int C,
                                                      int tiles_per_split, int normalize)
{
    // code tiles are staged HBM/L2 -> LDS once per workgroup by direct LDS-DMA (global_load_lds, 16 B per lane, no VGPR
    // round trip) and shared by the 4 waves: without this every wave streams the whole code range through L1/L2
    // (measured: ~20 TB/s of L2 traffic, the kernel's real bound at 77 % MFMA utilisation).
    __shared__ __attribute__((aligned(16))) float s_frag[2][M_CH * 512];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int row0 = (blockIdx.x * 4 + wave) * 32 * RT;

    // B operands: B[k][j] = x[row j][k]; lane (half, col) holds k = 2m + half of row col
    float b[RT][8];
    bool xbad = false;
    const uint32_t hmask = half ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        float zz[D], xx[D];
        int r = row0 + t * 32 + col;
        r = r < N ? r : N - 1;                       // clamp: out-of-range lanes redo the last row, never stored
        load_row16(z + (size_t)r * D, zz);
        if (normalize) l2norm16(zz, xx);
        else {
#pragma unroll
            for (int k = 0; k < D; ++k) xx[k] = zz[k];
        }
#pragma unroll
        for (int k = 0; k < D; ++k) xbad |= suspicious(xx[k]);
#pragma unroll
        for (int m = 0; m < 8; ++m)   // bit-select (not an indexed load: that would push xx[] into scratch/LDS)
            b[t][m] = __uint_as_float((__float_as_uint(xx[2 * m + 1]) & hmask) | (__float_as_uint(xx[2 * m]) & ~hmask));
    }
    // exact (NaN-aware) scan only if this wave holds a non-finite row or the pack step flagged the codebook
    // metadata word: bit 0 = a code element is non-finite / absurd (bits 1, 2 concern the f16 coarse path only)
    const bool slow = __any(xbad) || ((reinterpret_cast<const uint32_t*>(packed)[(size_t)C * D] & 1u) != 0u);

    const int ntiles_total = C >> 5;
    const int tile_first = blockIdx.y * tiles_per_split;
    int tile_last = tile_first + tiles_per_split;
    if (tile_last > ntiles_total) tile_last = ntiles_total;
    const int nt = tile_last - tile_first;

    BestT best[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) { best[t].v = -__builtin_inff(); best[t].tile = tile_first; best[t].slot = 0; best[t].nan = false; }

    auto mfma_tile = [&](const float4& lo, const float4& hi, f32x16 (&acc)[RT]) {
        const float a[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < 8; ++m)       // k = 2m (lanes 0-31), 2m+1 (lanes 32-63): k-ordered chain per accumulator
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[t][m], acc[t], 0, 0, 0);
    };

    // this wave's share of a chunk: pieces wave, wave+4, ... of the 2*M_CH (tile, part) pieces, 1 KiB each
    auto stage = [&](int chunk, int buf) {
#pragma unroll
        for (int pi = wave; pi < 2 * M_CH; pi += 4) {
            int tl = chunk * M_CH + (pi >> 1);
            tl = tl < nt ? tl : nt - 1;                                   // ragged last chunk: re-read the last tile
            const float* src = packed + (size_t)(tile_first + tl) * 512 + (pi & 1) * 256 + lane * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(&s_frag[buf][pi * 256]), 16, 0, 0);
        }
    };
    auto frag_lo = [&](int buf, int j) { return *reinterpret_cast<const float4*>(&s_frag[buf][j * 512 + lane * 4]); };
    auto frag_hi = [&](int buf, int j) { return *reinterpret_cast<const float4*>(&s_frag[buf][j * 512 + 256 + lane * 4]); };

    if (nt > 0) {
        const int nchunks = (nt + M_CH - 1) / M_CH;
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            const int buf = c & 1;
            if (c + 1 < nchunks) stage(c + 1, buf ^ 1);                   // DMA the next chunk behind this chunk's MFMAs
            const int base = c * M_CH;                                    // first tile of the chunk (relative)
            const bool full = base + M_CH <= nt;
            f32x16 accA[RT], accB[RT];
            if (!slow && full) {
                // two accumulator sets ping-pong: the scan of tile j is issued after the MFMAs of tile j+1
                mfma_tile(frag_lo(buf, 0), frag_hi(buf, 0), accA);
#pragma unroll
                for (int j = 0; j < M_CH; j += 2) {
                    mfma_tile(frag_lo(buf, j + 1), frag_hi(buf, j + 1), accB);
                    __builtin_amdgcn_sched_barrier(0);        // the scan (inline asm) must stay behind accB's MFMAs
#pragma unroll
                    for (int t = 0; t < RT; ++t) scanmax(best[t], accA[t], tile_first + base + j);
                    __builtin_amdgcn_sched_barrier(0);
                    if (j + 2 < M_CH) mfma_tile(frag_lo(buf, j + 2), frag_hi(buf, j + 2), accA);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < RT; ++t) scanmax(best[t], accB[t], tile_first + base + j + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                for (int j = 0; j < M_CH && base + j < nt; ++j) {
                    mfma_tile(frag_lo(buf, j), frag_hi(buf, j), accA);
                    if (slow) {
#pragma unroll
                        for (int t = 0; t < RT; ++t) scan16<true>(best[t], accA[t], tile_first + base + j);
                    } else {
#pragma unroll
                        for (int t = 0; t < RT; ++t) scanmax_safe(best[t], accA[t], tile_first + base + j);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // my DMAs of chunk c+1 have landed
            __syncthreads();                                              // everyone's have, and chunk c's buffer is free
        }
    }

    // one entry per (split, half, row): hi = orderable(best) (NaN -> 0xFFFFFFFF); lo = winning tile (fast path) or
    // 0x80000000 | exact code index (NaN-aware path)
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        float v = best[t].v;
        if (v == 0.0f) v = 0.0f;
        const uint32_t hi = best[t].nan ? KEY_NAN : f32_orderable(v);
        uint32_t lo = (uint32_t)best[t].tile;
        if (slow) lo = 0x80000000u | (uint32_t)(best[t].tile * 32 + (best[t].slot & 3) + 8 * (best[t].slot >> 2) + 4 * half);
        int r = row0 + t * 32 + col;
        if (r < N) partial[((size_t)blockIdx.y * 2 + half) * N + r] = ((unsigned long long)hi << 32) | lo;
    }
}

// ---------------------------------------------------------------------------------------
// f16 coarse pass + exact re-score ("change the algorithm", VERDICT r1 item 5 / SURVEY section 7)
//
// The fp32-input MFMA runs at the fp32 vector rate; the f16 MFMA runs 16x faster.  vq_f16_kernel computes APPROXIMATE scores
// s' = (x0 e0 + x0 e1 + x1 e0) with three v_mfma_f32_32x32x16_f16 per (32 codes x 32 rows) -- D = 16 is exactly one k-step --
// where x = x0 + x1, e = e0 + e1 are fp16 hi/lo splits of the (2^7-scaled) operands, |s' - s| < F16_EPS for the canonical
// fp32 score s.  Per lane stream (one wave half of one code split) it keeps the best tile maximum m1 (its tile t1) and the
// runner-up tile maximum m2 (v_med3_f32 keeps m1 >= m2 in one op).  The TRUE argmax c* satisfies s'(c*) >= max s' - 2 eps, so:
//   * every stream with m1 >= M - 2 eps (M = max over streams) is a candidate; c* lies in one of them;
//   * inside a candidate stream c* is in tile t1 unless another tile also reaches M - 2 eps, which implies m2 >= m1 - 2 eps:
//     that stream is flagged and re-scanned exactly, all of it.
// vq_finalize_f16_kernel re-scores the 16 codes of every candidate (tile, half) -- or the whole flagged stream -- with the
// canonical k-ordered fp32 FMA chain and takes the maximum, lowest index on ties (all exact ties are inside the window too).
// Ids and top-1 scores are therefore bit-identical to the fp32 kernels and the CPU oracle; rows or code books with non-finite /
// out-of-range values are flagged wholesale and go through the exact NaN-aware scan.
// ---------------------------------------------------------------------------------------
typedef _Float16 vh8 __attribute__((ext_vector_type(8)));
typedef _Float16 vh2 __attribute__((ext_vector_type(2)));
typedef unsigned vu4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2)));

constexpr uint32_t F16_FLAG = 0x80000000u;       // entry.lo bit 31: re-scan the whole stream exactly
constexpr uint32_t F16_HAS2 = 0x40000000u;       // entry.lo bit 30 (one-MFMA pass): a second tile of the stream is inside the window, re-score it too
// 4-byte candidates of the one-MFMA pass: truncating the orderable key to its top 20 bits lowers a (2^14-scaled, < 2^15) score by
// less than C4_TRUNC scaled units; the finalize widens its window by that much (the candidate test `stored >= stored_max - win` compares
// two truncated values: the true maximum's stream has stored >= true - C4_TRUNC >= stored_max - win - C4_TRUNC)
constexpr uint32_t C4_KEY_MASK = 0xFFFFF000u, C4_HAS2 = 0x800u, C4_FLAG = 0x400u, C4_TILE_MASK = 0x3FFu;
constexpr float C4_KEYTRUNC = 2.0f;       // the scan keys drop 10 mantissa bits of a value < 2^15: < 2^(14 - 23 + 10) = 2
constexpr float C4_TRUNC = 16.0f;         // the stored 20-bit key drops 12: < 2^(14 - 23 + 12) = 8, doubled as margin for the [2^14, 2^15) binade

SELFTOK_STAMP_DECL(tune_stamp_vq_f16);

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int RT, int NM>       // NM = MFMAs per 32 x 32 scores: 3 (hi*hi + hi*lo + lo*hi, window 2^-17) or 1 (hi*hi, window F16_EPS1)
__global__ __launch_bounds__(256) void vq_f16_kernel(const float* __restrict__ z, const float* __restrict__ packed,
                                                     unsigned long long* __restrict__ partial, int N, int C,
                                                     int tiles_per_split, int normalize)
{
    // 8 tiles x (2 planes x 64 lanes x 16 B), double buffered, + 1 KiB of padding: 33 KiB instead of 32 so that FOUR workgroups fit a
    // CU's 160 KiB, not five.  The launch is sized at four workgroups per CU (1024 at N = 32768); with room for a fifth the
    // dispatcher packs some CUs with five and leaves others three, and the kernel lasts as long as its fullest CU (residency census
    // of tools/sweep_vq_f16.py: workgroup lifetimes 42 .. 93 us for identical work, profiles/r3_vq_f16_sweep.txt)
    __shared__ __attribute__((aligned(16))) unsigned char s_tile[2][M_CH * 2048 + 512];
    SELFTOK_STAMP_BEGIN();

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int row0 = (blockIdx.x * 4 + wave) * 32 * RT;
    const _Float16* packed16 = reinterpret_cast<const _Float16*>(packed + (size_t)C * D + 64);

    // B operands: lane (half, col) holds x[row col][k = 8 half + j], j = 0..7, as fp16 hi / lo pairs of 128 x
    vu4 x0[RT], x1[RT];
    bool xbad = false;
    const uint32_t hsel = half ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        float zz[D], xx[D];
        int r = row0 + t * 32 + col;
        r = r < N ? r : N - 1;
        load_row16(z + (size_t)r * D, zz);
        if (normalize) l2norm16(zz, xx);
        else {
#pragma unroll
            for (int k = 0; k < D; ++k) xx[k] = zz[k];
        }
#pragma unroll
        for (int k = 0; k < D; ++k) xbad |= !(fabsf(xx[k]) < 400.f);      // NaN / inf / beyond fp16 after the 2^7 scaling
        if (!normalize) {            // caller-normalised rows (SELFTOK_PRENORMED): the window needs |x| <= 1, check instead of trusting
            float n2 = 0.f;
#pragma unroll
            for (int k = 0; k < D; ++k) n2 = __builtin_fmaf(xx[k], xx[k], n2);
            xbad |= !(n2 <= F16_NORM2_MAX);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // bit-select between the two halves' elements (a `half ? a : b` on array elements becomes an indexed scratch access)
            const float va = __uint_as_float((__float_as_uint(xx[8 + 2 * j]) & hsel) | (__float_as_uint(xx[2 * j]) & ~hsel));
            const float vb = __uint_as_float((__float_as_uint(xx[9 + 2 * j]) & hsel) | (__float_as_uint(xx[2 * j + 1]) & ~hsel));
            const vf2 v = {va * F16_PRESCALE, vb * F16_PRESCALE};
            const vh2 h = __builtin_convertvector(v, vh2);
            const vh2 l = __builtin_convertvector(v - __builtin_convertvector(h, vf2), vh2);
            x0[t][j] = __builtin_bit_cast(unsigned, h);
            x1[t][j] = __builtin_bit_cast(unsigned, l);
        }
    }
    const bool bad = xbad || (reinterpret_cast<const uint32_t*>(packed)[(size_t)C * D] != 0u);

    const int ntiles_total = C >> 5;
    const int tile_first = blockIdx.y * tiles_per_split;
    int tile_last = tile_first + tiles_per_split;
    if (tile_last > ntiles_total) tile_last = ntiles_total;
    const int nt = tile_last - tile_first;

    float m1[RT], m2[RT];
    int t1[RT];
    // one-MFMA pass: the three largest tile maxima of the stream as sortable integer keys, (fp32 bits of max(tile maximum, 0)) with the low
    // 10 mantissa bits replaced by 1023 - (tile - tile_first): v_max_u32 + two v_med3_u32 per tile keep the top three (value, tile)
    // pairs, the earlier tile ahead on equal values.  A second tile inside the window is then RE-SCORED (its index goes to the finalize),
    // only a third one forces the whole-stream walk.  Clamping at 0 costs nothing (it rides in the last v_max3 of the tile maximum);
    // a stream whose best score is <= 0 is flagged for the exact walk.
    uint32_t k1[RT], k2[RT], k3[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) { m1[t] = -__builtin_inff(); m2[t] = -__builtin_inff(); t1[t] = tile_first; k1[t] = 0u; k2[t] = 0u; k3[t] = 0u; }

    auto stage = [&](int chunk, int buf) {      // this wave's share: pieces wave, wave+4, ... of the 2*M_CH 1-KiB (tile, plane) pieces
#pragma unroll
        for (int pi = wave; pi < 2 * M_CH; pi += 4) {
            if (NM == 1 && (pi & 1)) continue;                            // the lo plane is not read by the one-MFMA pass
            int tl = chunk * M_CH + (pi >> 1);
            tl = tl < nt ? tl : nt - 1;
            const _Float16* src = packed16 + (size_t)(tile_first + tl) * 1024 + (pi & 1) * 512 + lane * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(&s_tile[buf][pi * 1024]), 16, 0, 0);
        }
    };
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    if (nt > 0) {
        const int nchunks = (nt + M_CH - 1) / M_CH;
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            const int buf = c & 1;
            if (c + 1 < nchunks) stage(c + 1, buf ^ 1);
            const int base = c * M_CH;
#ifdef SELFTOK_VQ_PRIO
            // progress-based priority: a wave drops its issue priority as it gets through its code range, so the waves that share a
            // SIMD advance together instead of oldest-first (and finish together instead of leaving a one-wave-per-SIMD tail)
            { const int q4 = (4 * c) / nchunks, q4p = c > 0 ? (4 * (c - 1)) / nchunks : -1;
              if (q4 != q4p) { if (q4 == 0) __builtin_amdgcn_s_setprio(3); else if (q4 == 1) __builtin_amdgcn_s_setprio(2); else if (q4 == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); } }
#endif
            // (round 3: a straight-line full-chunk body with two accumulator sets ping-ponging -- the scan of tile j interleaved with
            // the MFMAs of tile j + 1 -- was built and measured: RT = 4 needs 274 VGPRs (one wave per SIMD) and runs 109 us, RT = 2
            // 148 VGPRs and 95 us, against 92 - 95 us for the loop below; the kernel is not issue-bound: its workgroups keep the matrix
            // pipe ~87 % busy in SHADER cycles while the chip runs them at 1.5 - 2.0 GHz under this load (tools/sweep_vq_f16.py clock
            // stamps, profiles/r3_vq_f16_sweep.txt).  Kept: this loop.)
            if (NM == 1 && base + M_CH <= nt) {
                // one-MFMA pass, full chunk: straight-line code -- the chunk's 8 code fragments are read from LDS up front, and the MFMA of
                // unit u + 1 (unit = tile x row block) is issued BEFORE the scan of unit u (two accumulator sets): the scan's 13 VALU ops
                // hide the next MFMA's latency inside the wave instead of leaving it to the other three waves of the SIMD (the branchy
                // per-tile loop below serialised read -> MFMA -> 12 wait states -> scan per unit: 98 cycles per unit for 32 + 52 of work)
                vh8 ev[M_CH];
#pragma unroll
                for (int j = 0; j < M_CH; ++j) ev[j] = *reinterpret_cast<const vh8*>(&s_tile[buf][j * 2048 + lane * 16]);
                uint32_t maskv;
                asm("v_mov_b32 %0, 0xfffffc00" : "=v"(maskv));          // in a VGPR: lets (bits & mask) | tile-code be ONE v_and_or_b32
                auto scan1 = [&](const f32x16& acc, int t, int rel) {
                    const float a0 = __builtin_fmaxf(__builtin_fmaxf(acc[0], acc[1]), acc[2]), a1 = __builtin_fmaxf(__builtin_fmaxf(acc[3], acc[4]), acc[5]);
                    const float a2 = __builtin_fmaxf(__builtin_fmaxf(acc[6], acc[7]), acc[8]), a3 = __builtin_fmaxf(__builtin_fmaxf(acc[9], acc[10]), acc[11]);
                    const float a4 = __builtin_fmaxf(__builtin_fmaxf(acc[12], acc[13]), acc[14]);
                    const float mt0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(a0, a1), a2), __builtin_fmaxf(__builtin_fmaxf(a3, a4), acc[15])), 0.f);
                    const uint32_t key = (__float_as_uint(mt0) & maskv) | (uint32_t)(1023 - rel);
                    k3[t] = umed3(k2[t], k3[t], key);
                    k2[t] = umed3(k1[t], k2[t], key);
                    k1[t] = key > k1[t] ? key : k1[t];
                };
                f32x16 accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(ev[0], __builtin_bit_cast(vh8, x0[0]), zero, 0, 0, 0), accB;
#pragma unroll
                for (int u = 0; u < M_CH * RT; u += 2) {
                    { const int un = u + 1; accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(ev[un / RT], __builtin_bit_cast(vh8, x0[un % RT]), zero, 0, 0, 0); }
                    scan1(accA, u % RT, base + u / RT);
                    if (u + 2 < M_CH * RT) { const int un = u + 2; accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(ev[un / RT], __builtin_bit_cast(vh8, x0[un % RT]), zero, 0, 0, 0); }
                    scan1(accB, (u + 1) % RT, base + (u + 1) / RT);
                }
            } else
#pragma unroll
            for (int j = 0; j < M_CH; ++j) {
                if (base + j < nt) {                                   // wave-uniform
                    const vh8 e0 = *reinterpret_cast<const vh8*>(&s_tile[buf][j * 2048 + lane * 16]);
                    vh8 e1 = e0;
                    if (NM == 3) e1 = *reinterpret_cast<const vh8*>(&s_tile[buf][j * 2048 + 1024 + lane * 16]);
                    const int tile = tile_first + base + j;
#pragma unroll
                    for (int t = 0; t < RT; ++t) {
                        f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, __builtin_bit_cast(vh8, x0[t]), zero, 0, 0, 0);
                        if (NM == 3) {
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, __builtin_bit_cast(vh8, x1[t]), acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, __builtin_bit_cast(vh8, x0[t]), acc, 0, 0, 0);
                        }
                        // tile maximum (8 x v_max3), then (m1, m2) <- the two largest of (m1, m2, mt): v_max + v_med3
                        const float a0 = __builtin_fmaxf(__builtin_fmaxf(acc[0], acc[1]), acc[2]), a1 = __builtin_fmaxf(__builtin_fmaxf(acc[3], acc[4]), acc[5]);
                        const float a2 = __builtin_fmaxf(__builtin_fmaxf(acc[6], acc[7]), acc[8]), a3 = __builtin_fmaxf(__builtin_fmaxf(acc[9], acc[10]), acc[11]);
                        const float a4 = __builtin_fmaxf(__builtin_fmaxf(acc[12], acc[13]), acc[14]);
                        if (NM == 1) {
                            const float mt0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(a0, a1), a2), __builtin_fmaxf(__builtin_fmaxf(a3, a4), acc[15])), 0.f);
                            const uint32_t key = (__float_as_uint(mt0) & ~C4_TILE_MASK) | (uint32_t)(1023 - (base + j));     // v_and_or_b32
                            k3[t] = umed3(k2[t], k3[t], key);
                            k2[t] = umed3(k1[t], k2[t], key);
                            k1[t] = key > k1[t] ? key : k1[t];
                        } else {
                            const float mt = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(a0, a1), a2), __builtin_fmaxf(__builtin_fmaxf(a3, a4), acc[15]));
                            m2[t] = __builtin_amdgcn_fmed3f(m1[t], m2[t], mt);
                            const bool g = mt > m1[t];                      // strict: the earliest tile holding the maximum is t1
                            m1[t] = __builtin_fmaxf(m1[t], mt);
                            t1[t] = g ? tile : t1[t];
                        }
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    SELFTOK_STAMP_END(tune_stamp_vq_f16);
    const float win = 2.0f * (NM == 1 ? F16_EPS1 : F16_EPS) * F16_SCORE_SCALE;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int r = row0 + t * 32 + col;
        if (NM == 1) {
            // values carry a truncation of < C4_KEYTRUNC (10 mantissa bits of a 2^14-scaled score < 2^15): "inside the window" is tested
            // conservatively.  4-byte candidate: [31:12] top 20 bits of the orderable key of m1, [11] a second tile is inside the window (its
            // index goes to the side array), [10] a third one is too / bad input / best <= 0: walk the whole stream, [9:0] best tile
            // relative to the split (tiles_per_split <= 1024)
            const float v1 = __uint_as_float(k1[t] & ~C4_TILE_MASK), v2 = __uint_as_float(k2[t] & ~C4_TILE_MASK), v3 = __uint_as_float(k3[t] & ~C4_TILE_MASK);
            const bool has2 = !(v2 < v1 - win - C4_KEYTRUNC) && (1023u - (k2[t] & C4_TILE_MASK)) < (uint32_t)nt;     // k2 == 0: no second tile seen
            const bool walk = bad || !(v3 < v1 - win - C4_KEYTRUNC) || !(v1 > 0.f);
            const uint32_t rel1 = 1023u - (k1[t] & C4_TILE_MASK), rel2 = 1023u - (k2[t] & C4_TILE_MASK);
            const uint32_t hi = bad ? KEY_NAN : f32_orderable(v1);
            const uint32_t e = (hi & C4_KEY_MASK) | (has2 ? C4_HAS2 : 0u) | (walk ? C4_FLAG : 0u) | rel1;
            if (r < N) {
                const size_t at = ((size_t)blockIdx.y * 2 + half) * N + r;
                reinterpret_cast<uint32_t*>(partial)[at] = e;
                if (has2 && !walk) reinterpret_cast<uint32_t*>(partial)[(size_t)128 * N + at] = rel2;      // second half of the workspace, sparse
            }
        } else {
            const bool flag = bad || !(m2[t] < m1[t] - win);                // also true when m1 is NaN
            float v = m1[t];
            if (v == 0.0f) v = 0.0f;
            // a row / code book with non-finite or out-of-range values: v_max3 skipped the NaNs, m1 means nothing -> every stream of
            // the row must be re-scanned exactly, which the NaN key (sorts highest, opens the window completely) forces
            const uint32_t hi = (bad || v != v) ? KEY_NAN : f32_orderable(v);
            const uint32_t lo = (flag ? F16_FLAG : 0u) | (uint32_t)t1[t];
            if (r < N) partial[((size_t)blockIdx.y * 2 + half) * N + r] = ((unsigned long long)hi << 32) | lo;
        }
    }
}

// exact resolution of the coarse pass: see the comment above vq_f16_kernel.  16 lanes per row.
template <typename IdT, bool C4>       // C4: 4-byte candidates of the one-MFMA pass
// 8 waves per SIMD (<= 64 VGPRs): the kernel is a chain of dependent loads per row; all 2048 workgroups of N = 32768 resident at once
__global__ __launch_bounds__(256, 8) void vq_finalize_f16_kernel(const unsigned long long* __restrict__ partial, const float* __restrict__ z,
                                                              const float* __restrict__ packed, IdT* __restrict__ ids, float* __restrict__ best,
                                                              int N, int C, int nsplit, int tiles_per_split, int normalize, float win)
{
    const int gl = threadIdx.x & 15;
    const int gsh = (threadIdx.x & 63) & ~15;                            // first lane of this row's 16-lane group inside the wave
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const bool live = r < N;
    const int rr = live ? r : N - 1;
    const int nentries = 2 * nsplit;
    float x[D];
    {
        float zz[D];
        load_row16(z + (size_t)rr * D, zz);
        if (normalize) l2norm16(zz, x);
        else {
#pragma unroll
            for (int k = 0; k < D; ++k) x[k] = zz[k];
        }
    }
    const uint32_t* partial4 = reinterpret_cast<const uint32_t*>(partial);
    // M = best coarse maximum over the streams (NaN keys sort highest and are flagged anyway)
    uint32_t gmax = 0;
    for (int s = gl; s < nentries; s += 16) {
        const uint32_t hi = C4 ? (partial4[(size_t)s * N + rr] & C4_KEY_MASK) : (uint32_t)(partial[(size_t)s * N + rr] >> 32);
        gmax = hi > gmax ? hi : gmax;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { const uint32_t other = __shfl_xor(gmax, o, 16); gmax = other > gmax ? other : gmax; }
    const bool gnan = C4 ? (gmax == C4_KEY_MASK) : (gmax == KEY_NAN);
    const float thresh = gnan ? -__builtin_inff() : f32_from_orderable(gmax) - win;       // win = 2 eps of the coarse pass, scaled (+ C4_TRUNC)

    Best b;
    best_init(b, 0);
    const int ntiles_total = C >> 5;
    for (int s0 = 0; s0 < nentries; s0 += 16) {
        const int s_mine = s0 + gl;
        unsigned long long e = 0ull;
        if (s_mine < nentries) {
            if (C4) {       // rebuild the 8-byte form: key in the high word; flag | has-second | absolute tile in the low word
                const uint32_t e4 = partial4[(size_t)s_mine * N + rr];
                const uint32_t tile = (uint32_t)((s_mine >> 1) * tiles_per_split) + (e4 & C4_TILE_MASK);
                e = ((unsigned long long)(e4 & C4_KEY_MASK) << 32) | ((e4 & C4_FLAG) ? F16_FLAG : 0u) | ((e4 & C4_HAS2) ? F16_HAS2 : 0u) | tile;
            } else e = partial[(size_t)s_mine * N + rr];
        }
        const uint32_t ehi = (uint32_t)(e >> 32);
        const bool cand = s_mine < nentries && (ehi == (C4 ? C4_KEY_MASK : KEY_NAN) || !(f32_from_orderable(ehi) < thresh));
        uint32_t mask = (uint32_t)((__ballot(cand) >> gsh) & 0xFFFFu);
        while (mask) {
            const int j = __ffs(mask) - 1;
            mask &= mask - 1;
            const uint32_t lo = (uint32_t)__shfl((uint32_t)e, j, 16);
            const int stream = s0 + j, half = stream & 1, split = stream >> 1;
            const int i = (gl & 3) + 8 * (gl >> 2) + 4 * half;           // this lane's code inside a tile of that half
            int tfirst, tlast;
            if (lo & F16_FLAG) { tfirst = split * tiles_per_split; tlast = tfirst + tiles_per_split; tlast = tlast < ntiles_total ? tlast : ntiles_total; }
            else { tfirst = (int)(lo & 0x3FFFFFFFu); tlast = tfirst + 1; }
            int tsecond = -1;                                            // one-MFMA pass: the stream's second tile inside the window
            if (C4 && (lo & F16_HAS2) && !(lo & F16_FLAG)) tsecond = split * tiles_per_split + (int)partial4[(size_t)128 * N + (size_t)stream * N + rr];
            auto score_tile = [&](int tile) {
                const float* pt = packed + (size_t)tile * 512;
                float sc = 0.f;
#pragma unroll
                for (int k = 0; k < D; ++k) sc = __builtin_fmaf(x[k], pt[packed_offset(i, k)], sc);
                return sc;
            };
            // candidates are NOT visited in code order (streams come in entry order): on an exact tie, and among NaNs, the lower
            // code index wins explicitly (torch.argmax: first maximum; a NaN counts as the maximum, first NaN wins)
            auto take = [&](float sc, int tile) {
                const int idx = tile * 32 + i;
                if (sc != sc) {
                    if (!b.nan || idx < b.i) { b.nan = true; b.v = __builtin_inff(); b.i = idx; }
                } else if (!b.nan && (sc > b.v || (sc == b.v && idx < b.i))) { b.v = sc; b.i = idx; }
            };
            if (tlast - tfirst == 1) {
                const float sa = score_tile(tfirst), sb2 = score_tile(tsecond >= 0 ? tsecond : tfirst);
                take(sa, tfirst);
                if (tsecond >= 0) take(sb2, tsecond);
            } else {
                // whole-stream re-scan (the stream's two best tiles are both inside the window; 0.3 % of the rows with the one-MFMA window
                // on the encoder's features): 64 tiles, each 16 gathered loads + a 16-FMA chain per lane, two tiles' loads in flight.
                // (Eight in flight cost 160 VGPRs = 3 waves per SIMD: the 2048 workgroups of N = 32768 then run in 2.7 rounds and the
                // kernel takes 2.7 x a workgroup's latency -- the walk is rare, the residency is not.)
                for (int tile = tfirst; tile < tlast; tile += 2) {
                    const float sa = score_tile(tile), sb2 = score_tile(tile + 1 < tlast ? tile + 1 : tile);
                    take(sa, tile);
                    if (tile + 1 < tlast) take(sb2, tile + 1);
                }
            }
        }
    }
    unsigned long long key = best_key(b);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        const uint32_t klo = __shfl_xor((uint32_t)key, o, 16), khi = __shfl_xor((uint32_t)(key >> 32), o, 16);
        const unsigned long long other = ((unsigned long long)khi << 32) | klo;
        key = other > key ? other : key;
    }
    if (live && gl == 0) {
        const uint32_t hi = (uint32_t)(key >> 32);
        ids[r] = (IdT)(~(uint32_t)key);
        if (best) best[r] = (hi == KEY_NAN) ? __uint_as_float(0x7FC00000u) : f32_from_orderable(hi);
    }
}

// Reduce the (split, half) candidates of every row and recover the exact code index: among the candidates that hold
// the row's maximum, fast-path entries only know the winning tile, so the 16 scores of that (tile, half) are recomputed
// with the same k-ordered FMA chain (bit-identical to the MFMA result) and the first slot equal to the maximum is taken.
template <typename IdT>
__global__ __launch_bounds__(256) void vq_finalize_packed_kernel(const unsigned long long* __restrict__ partial, const float* __restrict__ z,
                                                                 const float* __restrict__ packed, IdT* __restrict__ ids, float* __restrict__ best,
                                                                 int N, int nentries, int normalize)
{
    // 16 lanes per row: lane j reduces entries j, j+16, ... and later evaluates slot j of a candidate tile
    const int gl = threadIdx.x & 15;
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const bool live = r < N;
    const int rr = live ? r : N - 1;
    uint32_t gmax = 0;
    for (int s = gl; s < nentries; s += 16) {
        uint32_t hi = (uint32_t)(partial[(size_t)s * N + rr] >> 32);
        gmax = hi > gmax ? hi : gmax;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { uint32_t other = __shfl_xor(gmax, o, 16); gmax = other > gmax ? other : gmax; }
    const float vmax = f32_from_orderable(gmax);
    float x[D];
    {
        float zz[D];
        load_row16(z + (size_t)rr * D, zz);
        if (normalize) l2norm16(zz, x);
        else {
#pragma unroll
            for (int k = 0; k < D; ++k) x[k] = zz[k];
        }
    }
    uint32_t idx_best = 0xFFFFFFFFu;
    for (int s0 = 0; s0 < nentries; s0 += 16) {
        // each lane looks at one entry; candidates (== gmax) are then resolved one at a time by the whole group
        const int s_mine = s0 + gl;
        unsigned long long e = s_mine < nentries ? partial[(size_t)s_mine * N + rr] : 0ull;
        const bool cand = s_mine < nentries && (uint32_t)(e >> 32) == gmax;
        // group-local ballot (the 16 lanes of a row are contiguous inside the wave)
        unsigned long long bal = __ballot(cand);
        uint32_t mask = (uint32_t)((bal >> ((threadIdx.x & 63) & ~15)) & 0xFFFFu);
        while (mask) {
            const int j = __ffs(mask) - 1;
            mask &= mask - 1;
            const uint32_t lo = (uint32_t)__shfl(e, j, 16);
            uint32_t idx;
            if (lo & 0x80000000u) idx = lo & 0x7FFFFFFFu;
            else {
                const int tile = (int)lo, half = (s0 + j) & 1;
                const float* pt = packed + (size_t)tile * 512;
                const int slot = gl;
                const int i = (slot & 3) + 8 * (slot >> 2) + 4 * half;      // code inside the tile
                float sc = 0.f;
#pragma unroll
                for (int k = 0; k < D; ++k) sc = __builtin_fmaf(x[k], pt[packed_offset(i, k)], sc);
                unsigned long long beq = __ballot(sc == vmax);
                uint32_t meq = (uint32_t)((beq >> ((threadIdx.x & 63) & ~15)) & 0xFFFFu);
                const int win = meq ? (__ffs(meq) - 1) : 0;               // first slot equal to the maximum
                idx = (uint32_t)(tile * 32 + (win & 3) + 8 * (win >> 2) + 4 * half);
            }
            idx_best = idx < idx_best ? idx : idx_best;
        }
    }
    if (live && gl == 0) {
        ids[r] = (IdT)idx_best;
        if (best) best[r] = (gmax == KEY_NAN) ? __uint_as_float(0x7FC00000u) : vmax;
    }
}

template <typename IdT>
__global__ void vq_finalize_kernel(const unsigned long long* __restrict__ partial, IdT* __restrict__ ids,
                                   float* __restrict__ best, int N, int nsplit)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    unsigned long long k = partial[r];
    for (int s = 1; s < nsplit; ++s) {
        unsigned long long o = partial[(size_t)s * N + r];
        k = o > k ? o : k;
    }
    uint32_t hi = (uint32_t)(k >> 32);
    ids[r] = (IdT)(~(uint32_t)k);
    if (best) best[r] = (hi == KEY_NAN) ? __uint_as_float(0x7FC00000u) : f32_from_orderable(hi);
}

// codes = codebook[ids]  (reference get_codes_from_indices, vector_quantize_pytorch.py:787-794) fused with
// final_layer_norm3 = LayerNorm(16, eps=1e-6, affine) (models_ours.py:88,241; SelftokPipeline.py:239-240).
// One thread per token; ln_w == nullptr -> plain gather.
template <typename IdT>
__global__ void code_gather_ln_kernel(const IdT* __restrict__ ids, const float* __restrict__ cb, const float* __restrict__ ln_w,
                                      const float* __restrict__ ln_b, float* __restrict__ out, int n, int C, float eps)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    long long id = (long long)ids[r];
    if (id < 0) id += C;                       // torch indexing semantics for negative ids
    id = id < 0 ? 0 : (id >= C ? C - 1 : id);  // never read out of bounds
    float v[D];
    load_row16(cb + (size_t)id * D, v);
    if (ln_w) {
        float mean = 0.f;
#pragma unroll
        for (int k = 0; k < D; ++k) mean += v[k];
        mean *= (1.0f / D);
        float var = 0.f;
#pragma unroll
        for (int k = 0; k < D; ++k) { float d = v[k] - mean; var = __builtin_fmaf(d, d, var); }
        var *= (1.0f / D);
        float rstd = 1.0f / __builtin_sqrtf(var + eps);
#pragma unroll
        for (int k = 0; k < D; ++k) v[k] = (v[k] - mean) * rstd * ln_w[k] + ln_b[k];
    }
    float4* o4 = reinterpret_cast<float4*>(out + (size_t)r * D);
#pragma unroll
    for (int q = 0; q < 4; ++q) o4[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// ---------------------------------------------------------------------------------------
// training-side codebook maintenance (vector_quantize_pytorch.py:568-611): the reference builds a one-hot [N, C] tensor and
// contracts it with the features (a second N x C x D GEMM plus 4 B N C of one-hot traffic).  Here the ids of the argmax kernel
// are scattered directly:   bins[c] += 1,  embed_sum[c][:] += l2norm(z[row])   (fp32 L2 atomics; contention is low, N rows
// spread over C = 32768 codes).  The caller zeroes bins / embed_sum and all-reduces them across ranks.
// ---------------------------------------------------------------------------------------
template <typename IdT>
__global__ __launch_bounds__(256) void vq_ema_accumulate_kernel(const float* __restrict__ z, const IdT* __restrict__ ids, float* __restrict__ bins,
                                                                float* __restrict__ embed_sum, int N, int C, int normalize)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;   // (row, quarter of the 16-d code)
    const int row = g >> 2, part = g & 3;
    if (row >= N) return;
    long id = (long)ids[row];
    if (id < 0 || id >= C) return;                         // never produced by the argmax kernel; ignore foreign ids
    float zz[D], xx[D];
    load_row16(z + (size_t)row * D, zz);
    if (normalize) l2norm16(zz, xx);
    else {
#pragma unroll
        for (int k = 0; k < D; ++k) xx[k] = zz[k];
    }
    float* dst = embed_sum + (size_t)id * D + part * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = xx[0];
#pragma unroll
        for (int k = 1; k < D; ++k) v = (k == part * 4 + j) ? xx[k] : v;     // select without an indexed (scratch) access
        unsafeAtomicAdd(dst + j, v);
    }
    if (part == 0) unsafeAtomicAdd(bins + id, 1.0f);
}

// timestep_p_over_c [K, C] <- lerp(tpc, batch one-hot mean, w)  (vector_quantize_pytorch.py:568-578, ema_inplace :66-72) without the
// [B, K, C] one-hot: a dense decay pass (the lerp against 0) and a sparse pass adding w / B at (k, id) for every token.
__global__ __launch_bounds__(256) void vq_tpc_decay_kernel(float* __restrict__ tpc, long n4, float w)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 v = reinterpret_cast<float4*>(tpc)[i];
    if (w < 0.5f) {           // torch.lerp: self + w * (end - self), end = 0
        v.x = __builtin_fmaf(w, -v.x, v.x); v.y = __builtin_fmaf(w, -v.y, v.y); v.z = __builtin_fmaf(w, -v.z, v.z); v.w = __builtin_fmaf(w, -v.w, v.w);
    } else {                  //             end - (end - self) * (1 - w)
        const float k = 1.0f - w;
        v.x *= k; v.y *= k; v.z *= k; v.w *= k;
    }
    reinterpret_cast<float4*>(tpc)[i] = v;
}
template <typename IdT>
__global__ __launch_bounds__(256) void vq_tpc_scatter_kernel(float* __restrict__ tpc, const IdT* __restrict__ ids, int n, int K, int C, float add)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // token (b, k)
    if (i >= n) return;
    const long id = (long)ids[i];
    if (id < 0 || id >= C) return;
    unsafeAtomicAdd(tpc + (size_t)(i % K) * C + id, add);
}

}  // namespace selftok

using namespace selftok;

static int pick_split(int row_blocks, int ntiles, int max_split)
{
    // aim for >= ~1024 workgroups (4 per CU) while keeping >= 8 tiles per split
    int split = 1;
    while (row_blocks * split < 1024 && split * 2 <= max_split && ntiles / (split * 2) >= 8) split *= 2;
    return split;
}

// Code-split count such that row_blocks*split fills the chip in whole "rounds" of resident workgroups
// (slots = CUs x workgroups per CU from the occupancy query): a grid of 1.33 rounds runs at 67 % of a grid of 1.0.
static int pick_split_balanced(int row_blocks, int ntiles, int max_split, int slots)
{
    if (slots <= 0) return pick_split(row_blocks, ntiles, max_split);
    int best = 1;
    double best_eff = 0.0;
    for (int split = 1; split <= max_split && split <= ntiles; ++split) {
        int tps = (ntiles + split - 1) / split;
        if (tps < 4 && split > 1) break;
        int eff_split = (ntiles + tps - 1) / tps;
        long blocks = (long)row_blocks * eff_split;
        long rounds = (blocks + slots - 1) / slots;
        // time ~ rounds * tps (every block scans tps tiles); ideal = total tiles / slots
        double eff = ((double)row_blocks * ntiles / slots) / ((double)rounds * tps);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = eff_split; }
    }
    return best;
}

template <typename K>
static int resident_slots(K kernel)
{
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess) return 0;
    return cus * per_cu;
}


// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
extern "C" {

size_t selftok_vq_workspace_bytes(int N, int C)
{
    (void)C;
    return (size_t)128 * (size_t)(N > 0 ? N : 1) * sizeof(unsigned long long);   // up to 64 code splits x 2 wave halves
}

// fp32 fragment image + one metadata line (flags) + fp16 hi/lo image of the 2^7-scaled code book
size_t selftok_vq_packed_bytes(int C, int Dm) { return ((size_t)C * Dm + 64) * sizeof(float) + (size_t)C * Dm * 2 * sizeof(_Float16); }

int selftok_vq_pack_codebook(const float* codebook, float* packed, int C, int Dm, hipStream_t stream)
{
    if (!codebook || !packed || Dm != D || C <= 0 || (C & 31)) { set_last_error("vq_pack: need D==16 and C%32==0"); return SELFTOK_EINVAL; }
    int total = C * D;
    if (hipMemsetAsync(packed + (size_t)C * D, 0, 64 * sizeof(float), stream) != hipSuccess) return check_launch("vq_pack memset");
    hipLaunchKernelGGL(vq_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, codebook, packed, C);
    return check_launch("vq_pack_kernel");
}

// flags: bit0 = ids are int32 (default int64), bit1 = z is already unit-norm (skip l2norm)
int selftok_vq_encode_f32(const float* z, const float* codebook, void* ids, float* best, void* workspace,
                          int N, int C, int Dm, int flags, hipStream_t stream)
{
    if (N == 0 && Dm == D && C > 0) return SELFTOK_OK;     // empty batch: nothing to do (pointers may be null)
    if (Dm != D || C <= 0 || N < 0 || !z || !codebook || !ids || !workspace) { set_last_error("vq_encode: bad argument"); return SELFTOK_EINVAL; }
    unsigned long long* partial = (unsigned long long*)workspace;
    int ntiles = (C + V_TILE - 1) / V_TILE;
    int row_blocks = (N + 4 * V_ROWS - 1) / (4 * V_ROWS);
    int split = pick_split(row_blocks, ntiles, 64);
    int tps = (ntiles + split - 1) / split;
    split = (ntiles + tps - 1) / tps;
    hipLaunchKernelGGL(vq_valu_kernel, dim3(row_blocks, split), dim3(256), 0, stream, z, codebook, partial, N, C, tps, (flags & 2) ? 0 : 1);
    int rc = check_launch("vq_valu_kernel");
    if (rc) return rc;
    if (flags & 1) hipLaunchKernelGGL(vq_finalize_kernel<int32_t>, dim3((N + 255) / 256), dim3(256), 0, stream, partial, (int32_t*)ids, best, N, split);
    else hipLaunchKernelGGL(vq_finalize_kernel<long long>, dim3((N + 255) / 256), dim3(256), 0, stream, partial, (long long*)ids, best, N, split);
    return check_launch("vq_finalize_kernel");
}

// Main kernel only: per-(split,row) keys into `workspace`; *nsplit_out receives the number of code splits.
int selftok_vq_argmax_partial_packed_f32(const float* z, const float* packed, void* workspace, int* nsplit_out,
                                         int N, int C, int Dm, int flags, hipStream_t stream)
{
    if (nsplit_out) *nsplit_out = 0;
    if (N == 0 && Dm == D && C > 0 && !(C & 31) && nsplit_out) return SELFTOK_OK;     // empty batch
    if (Dm != D || C <= 0 || (C & 31) || N < 0 || !z || !packed || !workspace || !nsplit_out) { set_last_error("vq_argmax_partial_packed: bad argument"); return SELFTOK_EINVAL; }
    unsigned long long* partial = (unsigned long long*)workspace;
    const int ntiles = C >> 5;
    const int norm = (flags & 2) ? 0 : 1;
    if (flags & SELFTOK_VQ_F16COARSE) {
        // f16 coarse pass: the matrix work is 5x cheaper, so fewer, longer code splits (every split costs 16 B of candidates per row
        // and a finalize visit) and 4 row blocks per wave from 8192 rows on
        int rt = N >= 8192 ? 4 : (N >= 2048 ? 2 : 1);
        { const int f_rt = (flags >> 8) & 0xF; if (f_rt == 1 || f_rt == 2 || f_rt == 4) rt = f_rt; }
        const int row_blocks = (N + 128 * rt - 1) / (128 * rt);
        // code splits: 64 tiles (2048 codes) per split -- the exact re-scan of a flagged stream walks its tiles one after the other,
        // so stream length, not occupancy, sets the finalize time (measured at N = 32768: 8 / 16 / 32 splits -> 0.129 / 0.115 /
        // 0.112 ms in total, tools/sweep_vq_f16.py); small batches split further to fill the chip
        int split = ntiles / 64 > 0 ? ntiles / 64 : 1;
        while (row_blocks * split < 512 && split < 64 && ntiles / (split * 2) >= 8) split *= 2;
        if (split > 64) split = 64;
        { const int f_split = (flags >> 16) & 0xFF; if (f_split > 0 && f_split <= 64) split = f_split; }
        if ((flags & SELFTOK_VQ_F16COARSE1) && (ntiles + split - 1) / split > 1024) {
            // the 4-byte candidates carry a 10-bit tile field: at most 1024 tiles per split, and the workspace / side array hold 64 splits -> C <= 2^21
            // codes.  A larger code book takes the 3-MFMA variant (8-byte candidates, any stream length): same ids and score bits
            split = (ntiles + 1023) / 1024;
            if (split > 64) { set_last_error("vq_argmax_partial_packed: SELFTOK_VQ_F16COARSE1 supports at most 2^21 codes (use the 3-MFMA coarse pass)"); return SELFTOK_EINVAL; }
        }
        const int tps = (ntiles + split - 1) / split;
        split = (ntiles + tps - 1) / tps;
        *nsplit_out = split;
        dim3 grid(row_blocks, split), block(256);
        if (flags & SELFTOK_VQ_F16COARSE1) {
            if (rt == 4) hipLaunchKernelGGL((vq_f16_kernel<4, 1>), grid, block, 0, stream, z, packed, partial, N, C, tps, norm);
            else if (rt == 2) hipLaunchKernelGGL((vq_f16_kernel<2, 1>), grid, block, 0, stream, z, packed, partial, N, C, tps, norm);
            else hipLaunchKernelGGL((vq_f16_kernel<1, 1>), grid, block, 0, stream, z, packed, partial, N, C, tps, norm);
        } else {
            if (rt == 4) hipLaunchKernelGGL((vq_f16_kernel<4, 3>), grid, block, 0, stream, z, packed, partial, N, C, tps, norm);
            else if (rt == 2) hipLaunchKernelGGL((vq_f16_kernel<2, 3>), grid, block, 0, stream, z, packed, partial, N, C, tps, norm);
            else hipLaunchKernelGGL((vq_f16_kernel<1, 3>), grid, block, 0, stream, z, packed, partial, N, C, tps, norm);
        }
        return check_launch("vq_f16_kernel");
    }
    int rt = N >= 32768 ? 4 : (N >= 8192 ? 2 : 1);     // measured: 130 / 126 / 118 TF at N=32768 for RT = 4 / 2 / 1
    {   // explicit tuning override in `flags` (tests sweep it; results never depend on it): SELFTOK_VQ_RT(1|2|4)
        const int f_rt = (flags >> 8) & 0xF;
        if (f_rt == 1 || f_rt == 2 || f_rt == 4) rt = f_rt;
    }
    int row_blocks = (N + 128 * rt - 1) / (128 * rt);
    // resident workgroups per kernel variant: a pure function of (device, kernel), memoised per device -- no call-to-call state
    static int slots[16][3] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (slots[dev][0] == 0) {
        slots[dev][2] = resident_slots(vq_mfma_kernel<4>); slots[dev][1] = resident_slots(vq_mfma_kernel<2>);
        slots[dev][0] = resident_slots(vq_mfma_kernel<1>);
    }
    int split = pick_split_balanced(row_blocks, ntiles, 64, rt == 4 ? slots[dev][2] : (rt == 2 ? slots[dev][1] : slots[dev][0]));
    {   // SELFTOK_VQ_SPLIT(n), 1 <= n <= 64
        const int f_split = (flags >> 16) & 0xFF;
        if (f_split > 0 && f_split <= 64) split = f_split;
    }
    int tps = (ntiles + split - 1) / split;
    split = (ntiles + tps - 1) / tps;
    *nsplit_out = split;
    dim3 grid(row_blocks, split), block(256);
    if (rt == 4) hipLaunchKernelGGL(vq_mfma_kernel<4>, grid, block, 0, stream, z, packed, partial, N, C, tps, norm);
    else if (rt == 2) hipLaunchKernelGGL(vq_mfma_kernel<2>, grid, block, 0, stream, z, packed, partial, N, C, tps, norm);
    else hipLaunchKernelGGL(vq_mfma_kernel<1>, grid, block, 0, stream, z, packed, partial, N, C, tps, norm);
    return check_launch("vq_mfma_kernel");
}

// Reduce the candidates of a partial pass: ids (int64, or int32 with SELFTOK_IDS_I32) and optional top-1 score.
// Needs z and the packed codebook again to pin down the slot inside the winning tile (see the kernel).
int selftok_vq_finalize_packed(const void* workspace, const float* z, const float* packed, void* ids, float* best,
                               int N, int C, int Dm, int nsplit, int flags, hipStream_t stream)
{
    if (N == 0) return SELFTOK_OK;
    if (!workspace || !z || !packed || !ids || N < 0 || nsplit <= 0 || Dm != D || (C & 31)) { set_last_error("vq_finalize_packed: bad argument"); return SELFTOK_EINVAL; }
    const unsigned long long* partial = (const unsigned long long*)workspace;
    const int norm = (flags & 2) ? 0 : 1;
    if (flags & SELFTOK_VQ_F16COARSE) {
        const int ntiles = C >> 5, tps = (ntiles + nsplit - 1) / nsplit;
        if (flags & SELFTOK_VQ_F16COARSE1) {
            const float win = 2.0f * F16_EPS1 * F16_SCORE_SCALE + C4_TRUNC;      // the main kernel's window + the truncation of the 4-byte keys
            if (flags & 1) hipLaunchKernelGGL((vq_finalize_f16_kernel<int32_t, true>), dim3((N + 15) / 16), dim3(256), 0, stream, partial, z, packed, (int32_t*)ids, best, N, C, nsplit, tps, norm, win);
            else hipLaunchKernelGGL((vq_finalize_f16_kernel<long long, true>), dim3((N + 15) / 16), dim3(256), 0, stream, partial, z, packed, (long long*)ids, best, N, C, nsplit, tps, norm, win);
        } else {
            const float win = 2.0f * F16_EPS * F16_SCORE_SCALE;                  // must match the main kernel's
            if (flags & 1) hipLaunchKernelGGL((vq_finalize_f16_kernel<int32_t, false>), dim3((N + 15) / 16), dim3(256), 0, stream, partial, z, packed, (int32_t*)ids, best, N, C, nsplit, tps, norm, win);
            else hipLaunchKernelGGL((vq_finalize_f16_kernel<long long, false>), dim3((N + 15) / 16), dim3(256), 0, stream, partial, z, packed, (long long*)ids, best, N, C, nsplit, tps, norm, win);
        }
        return check_launch("vq_finalize_f16_kernel");
    }
    if (flags & 1) hipLaunchKernelGGL(vq_finalize_packed_kernel<int32_t>, dim3((N + 15) / 16), dim3(256), 0, stream, partial, z, packed, (int32_t*)ids, best, N, 2 * nsplit, norm);
    else hipLaunchKernelGGL(vq_finalize_packed_kernel<long long>, dim3((N + 15) / 16), dim3(256), 0, stream, partial, z, packed, (long long*)ids, best, N, 2 * nsplit, norm);
    return check_launch("vq_finalize_packed_kernel");
}

int selftok_vq_encode_packed_f32(const float* z, const float* packed, void* ids, float* best, void* workspace,
                                 int N, int C, int Dm, int flags, hipStream_t stream)
{
    if (!ids && N != 0) { set_last_error("vq_encode_packed: bad argument"); return SELFTOK_EINVAL; }
    int split = 0;
    int rc = selftok_vq_argmax_partial_packed_f32(z, packed, workspace, &split, N, C, Dm, flags, stream);
    if (rc || N == 0) return rc;
    return selftok_vq_finalize_packed(workspace, z, packed, ids, best, N, C, Dm, split, flags, stream);
}

#ifdef SELFTOK_TUNE
// tools/ builds only: (shader cycles, 100 MHz ticks) of workgroup (0, 0) of the last vq_f16_kernel launch
int selftok_tune_vq_stamp(unsigned long long* out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(tune_stamp_vq_f16), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? SELFTOK_OK : SELFTOK_EHIP;
}
#endif

// flags bit0: ids are int32 (default int64).  ln_w/ln_b may be NULL (plain gather).
int selftok_code_gather_ln_f32(const void* ids, const float* codebook, const float* ln_w, const float* ln_b, float* out,
                               int n, int C, int Dm, float eps, int flags, hipStream_t stream)
{
    if (n == 0) return SELFTOK_OK;
    if (Dm != D || n < 0 || !ids || !codebook || !out || ((ln_w == nullptr) != (ln_b == nullptr))) { set_last_error("code_gather_ln: bad argument"); return SELFTOK_EINVAL; }
    if (flags & 1) hipLaunchKernelGGL(code_gather_ln_kernel<int32_t>, dim3((n + 255) / 256), dim3(256), 0, stream, (const int32_t*)ids, codebook, ln_w, ln_b, out, n, C, eps);
    else hipLaunchKernelGGL(code_gather_ln_kernel<long long>, dim3((n + 255) / 256), dim3(256), 0, stream, (const long long*)ids, codebook, ln_w, ln_b, out, n, C, eps);
    return check_launch("code_gather_ln_kernel");
}

// bins [C] and embed_sum [C,16] (both zeroed by the caller) += the batch's one-hot statistics (see the kernel comment)
int selftok_vq_ema_accumulate_f32(const float* z, const void* ids, float* bins, float* embed_sum, int N, int C, int Dm, int flags, hipStream_t stream)
{
    if (N == 0) return SELFTOK_OK;
    if (Dm != D || N < 0 || C <= 0 || !z || !ids || !bins || !embed_sum) { set_last_error("vq_ema_accumulate: bad argument"); return SELFTOK_EINVAL; }
    const int norm = (flags & 2) ? 0 : 1;
    const unsigned grid = (unsigned)(((long)N * 4 + 255) / 256);
    if (flags & 1) hipLaunchKernelGGL(vq_ema_accumulate_kernel<int32_t>, dim3(grid), dim3(256), 0, stream, z, (const int32_t*)ids, bins, embed_sum, N, C, norm);
    else hipLaunchKernelGGL(vq_ema_accumulate_kernel<long long>, dim3(grid), dim3(256), 0, stream, z, (const long long*)ids, bins, embed_sum, N, C, norm);
    return check_launch("vq_ema_accumulate_kernel");
}

// tpc [K,C] <- lerp(tpc, mean over the B samples of one_hot(ids [B,K]), weight)
int selftok_vq_tpc_update_f32(float* tpc, const void* ids, int B, int K, int C, float weight, int flags, hipStream_t stream)
{
    if (!tpc || K <= 0 || C <= 0 || (C & 3) || B < 0 || (B > 0 && !ids)) { set_last_error("vq_tpc_update: bad argument (C % 4 == 0)"); return SELFTOK_EINVAL; }
    const long n4 = (long)K * C / 4;
    hipLaunchKernelGGL(vq_tpc_decay_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, tpc, n4, weight);
    int rc = check_launch("vq_tpc_decay_kernel");
    if (rc || B == 0) return rc;
    const int n = B * K;
    const float add = weight / (float)B;
    if (flags & 1) hipLaunchKernelGGL(vq_tpc_scatter_kernel<int32_t>, dim3((n + 255) / 256), dim3(256), 0, stream, tpc, (const int32_t*)ids, n, K, C, add);
    else hipLaunchKernelGGL(vq_tpc_scatter_kernel<long long>, dim3((n + 255) / 256), dim3(256), 0, stream, tpc, (const long long*)ids, n, K, C, add);
    return check_launch("vq_tpc_scatter_kernel");
}

}  // extern "C"
