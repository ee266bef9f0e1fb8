// fp32 Linear on the fp32-input matrix cores with BOTH operands staged by LDS-DMA (round 6).  gfx950 only.
//
//   out[m][n] = epilogue( sum_k x[m][k] w[n][k] )          x [M][K] (row stride ldx), w [N][K] contiguous -- nn.Linear / F.linear
//
// replaces (a) in gemm='exact' the wide Linears of the MMDiT joint blocks and of the Q-Former (mimogpt/models/selftok/sd3/mmdit.py:266-307, 413-419;
// modules.py:186-199, 293) that xe_gemm128_kernel (csrc/encoder_exact.hip) carried at 0.76 - 0.84 of the fp32 matrix peak -- same bits: MKL sgemm's order, one
// sequential fmaf chain per K-block of 384, out = ((bias + c0) + c1) + ... -- and (b) in gemm='fp32' the Linears for which hipBLASLt's kernels run at 0.79 - 0.90
// of that peak (qkv, proj): there the order is free and the whole K is ONE chain per output.
//
// Design (why it is not xe_gemm128 with other constants):
//   * v_mfma_f32_32x32x1_2b_f32: one k per instruction and TWO 32 x 32 blocks.  A lane supplies A[row 32 h + i][k] and B[col i][k] (h = lane / 32, i = lane % 32):
//     the two blocks are the two row halves of a 64-row wave tile.  An fp32-input MFMA is fma(a, b, acc) per output, bit for bit (csrc/vq.hip, round 1), so a
//     k-ascending instruction stream IS the sequential chain -- with the operands in their NATURAL row-major order.  The 32x32x2 form xe_gemm128 uses wants the even
//     and the odd k of a row in different lane halves: a de-interleave while staging = global -> VGPR -> 8 v_mov -> ds_write_b128, on the VALU port the fp32 MFMAs
//     share.  With the natural order a 16-byte piece of a row is what a lane needs as it lies in memory, so
//   * both operand tiles reach LDS by `global_load_lds_dwordx4` (1 KiB = 8 rows x 128 bytes per wave instruction; uniform SGPR base + one constant per-lane offset
//     register per piece; the 16-byte chunks of a row XOR-swizzled on the SOURCE side so that every ds_read_b128 lane group covers all 64 banks once): no staging
//     registers, no ds_write, no VALU in the k-loop besides the K-block fold.
//   * workgroup = 8 waves = 256 x 128 outputs (wave: 64 x 64 = two 2-block accumulators, 64 VGPRs), 32-k chunks, THREE stages of 48 KiB (144 of the CU's 160 KiB):
//     chunk c + 2 is in flight while chunk c computes; one barrier per chunk = per 64 MFMAs (4096 matrix cycles) per wave.
//   * XCD-aware 1-D tile order (bands of 4 row tiles: the 32 workgroups resident on an XCD cover 4 x 8 tiles = 4 A + 8 B operand tiles per chunk round).
//   * the TAIL ROUND is split along K: the tiles left over after the last full round of 256 workgroups (proj: 4.2 rounds -> 5 at 0.84 efficiency, the figure
//     hipBLASLt's proj sits at too) are computed as S units of 1 / S of the K range each, raw partial sums to a workspace, and a small second kernel adds
//     the planes IN ORDER and runs the epilogue.  In MKL order a unit is a whole number of K-blocks and writes one plane per K-block, so the result is still
//     ((bias + c0) + c1) + ... bit for bit; in free order a unit writes one plane.
//
// Epilogue (both kernels): [+ bias last] -> [GELU(tanh), Sleef arithmetic] -> [gate * y] -> [res + y], each separately rounded (-ffp-contract=off), the
// operation sequence of csrc/encoder_exact.hip's epilogue.
#include "common.h"
#include "exact_math.h"
#include "selftok_hip.h"

#include <type_traits>

namespace selftok {

typedef float sg_f32x32 __attribute__((ext_vector_type(32)));
typedef float sg_f32x4 __attribute__((ext_vector_type(4)));

constexpr int SG_BM = 256, SG_BN = 128, SG_BK = 32;
constexpr int SG_A_BYTES = SG_BM * SG_BK * 4;                 // 32 KiB
constexpr int SG_STAGE = (SG_BM + SG_BN) * SG_BK * 4;         // 48 KiB
constexpr int SG_STAGES = 3;
constexpr int SG_PLANE = SG_BM * SG_BN;                       // floats per workspace plane

struct SgArgs {
    const float* a; long lda;                   // x [M][K], row stride lda (floats, multiple of 4)
    const float* b;                             // w [N][K] contiguous
    float* c; long ldc;
    const float* bias;
    const float* res; long ldr; int res_mod;    // y = res[row(m, res_mod)][n] + (gate ? gate * y : y); row(m, d) = m % d (d > 0), m / -d (d < 0), m (0)
    const float* gate; long ldg; int gate_mod;
    float* ws;                                  // tail planes: [xcd][tail tile][plane][256][128]
    int M, N, K;
    int gelu, bias_last;
    int mt, nt, tiles, per;                     // tile grid; per = ceil(tiles / 8) list entries per XCD
    int full_pos, tail_cnt, split, planes;      // entries < full_pos of every XCD's list: full tiles; the next tail_cnt: `split` units each; planes per tail tile
    int blk_chunks;                             // MKL order: K-block length in chunks (12 = 384 / 32); free order: K / 32 (one block)
    int nchunks;                                // K / 32
};

// list entry `sidx` (0 .. tiles-1) -> (row tile, column tile): bands of 4 row tiles, inside a band column tile by column tile
__host__ __device__ inline void sg_tile_of(int sidx, int MT, int NT, int& tm, int& tn)
{
    const int full = MT >> 2, rem = MT & 3, cut = full * 4 * NT;
    if (sidx < cut) { const int band = sidx / (4 * NT), r = sidx - band * 4 * NT; tn = r >> 2; tm = band * 4 + (r & 3); }
    else { const int r = sidx - cut; tn = r / rem; tm = full * 4 + (r - tn * rem); }
}

__device__ __forceinline__ void sg_dma16(const void* base, unsigned voff, unsigned lds)
{
    // M0 = LDS byte address of the 1-KiB piece (declared clobbered: no compiler version may keep a value of its own in M0 across this)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory", "m0");
}

__device__ __forceinline__ float sg_epilogue(const SgArgs& g, float v, int m, int n)
{
    if (g.bias_last && g.bias != nullptr) v = v + g.bias[n];
    if (g.gelu) v = xe_gelu_tanh1(v);
    if (g.gate != nullptr) v = g.gate[(size_t)(g.gate_mod > 0 ? m % g.gate_mod : (g.gate_mod < 0 ? m / -g.gate_mod : m)) * g.ldg + n] * v;
    if (g.res != nullptr) v = g.res[(size_t)(g.res_mod > 0 ? m % g.res_mod : (g.res_mod < 0 ? m / -g.res_mod : m)) * g.ldr + n] + v;
    return v;
}

template <bool MKL>
__global__ __launch_bounds__(512) void sg_gemm_kernel(SgArgs g)
{
    __shared__ __attribute__((aligned(1024))) char lds[SG_STAGES * SG_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int i = lane & 31, h = lane >> 5;

    // ---- which tile, which part of K ----
    int tm, tn, c0 = 0, c1 = g.nchunks, unit = -1, te = 0;
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
    {
        int e = pos;
        if (pos >= g.full_pos) {
            const int u = pos - g.full_pos;
            te = u / g.split;
            unit = u - te * g.split;
            e = g.full_pos + te;
            const int per_unit = ((g.nchunks + g.blk_chunks - 1) / g.blk_chunks + g.split - 1) / g.split * g.blk_chunks;      // whole K-blocks per unit (free order: blk_chunks = chunks per unit)
            c0 = unit * per_unit;
            c1 = min(c0 + per_unit, g.nchunks);
        }
        const int sidx = xcd * g.per + e;
        if (e >= g.per || sidx >= g.tiles || c0 >= c1) return;
        sg_tile_of(sidx, g.mt, g.nt, tm, tn);
    }
    // integer divisions by run-time values are VALU sequences: their (uniform) results come back in VGPRs, and an inline-asm "s" operand is not legalised
    tm = __builtin_amdgcn_readfirstlane(tm); tn = __builtin_amdgcn_readfirstlane(tn);
    c0 = __builtin_amdgcn_readfirstlane(c0); c1 = __builtin_amdgcn_readfirstlane(c1);
    unit = __builtin_amdgcn_readfirstlane(unit); te = __builtin_amdgcn_readfirstlane(te);
    const int row0 = tm * SG_BM, col0 = tn * SG_BN;

    // ---- staging: wave w issues pieces w, w + 8, ..., w + 40 of a chunk (pieces 0..31 = A rows 8 p .. 8 p + 7, pieces 32..47 = B rows) ----
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)&lds[0];
    const int pr = lane >> 3, pc = lane & 7;                                  // row inside a piece, 16-byte position inside the 128-byte row
    const int psw = ((wave & 1) << 2) | (pr >> 1);                            // (tile row >> 1) & 7 of this lane's row, the same for all six pieces (p = w mod 8)
    unsigned voa[4], vob[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave + 8 * j) * 8 + pr;
        const int rg = min(row0 + r, g.M - 1) - row0;                         // ragged last row tile: fetch the matrix's last row instead (never stored)
        voa[j] = (unsigned)rg * (unsigned)g.lda * 4u + (unsigned)((pc ^ psw) << 4);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) vob[j] = (unsigned)((wave + 8 * j) * 8 + pr) * (unsigned)g.K * 4u + (unsigned)((pc ^ psw) << 4);
    const char* abase = reinterpret_cast<const char*>(g.a + (size_t)row0 * g.lda);
    const char* bbase = reinterpret_cast<const char*>(g.b + (size_t)col0 * g.K);
    auto issue = [&](int chunk, int stage) {
        const char* ab = abase + (size_t)chunk * (SG_BK * 4);
        const char* bb = bbase + (size_t)chunk * (SG_BK * 4);
        const unsigned l = lds0 + stage * SG_STAGE + wave * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) sg_dma16(ab, voa[j], l + j * 8192);
#pragma unroll
        for (int j = 0; j < 2; ++j) sg_dma16(bb, vob[j], l + SG_A_BYTES + j * 8192);
    };

    // ---- fragment addresses: A row 64 wm + lane, B rows 64 wn + 32 t + i; chunk q of a row sits at position q ^ ((lane >> 1) & 7) ----
    const int sw16 = ((lane >> 1) & 7) << 4;
    const int a_row = (64 * wm + lane) * 128 + sw16;
    const int b_row = SG_A_BYTES + (64 * wn + i) * 128 + sw16;

    sg_f32x32 acc[2];
    float C[MKL ? 2 : 1][MKL ? 32 : 1];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 32; ++r) acc[t][r] = 0.f;
    if (MKL) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float b0 = (g.bias != nullptr && !g.bias_last && unit < 0) ? g.bias[col0 + 64 * wn + 32 * t + i] : 0.f;
#pragma unroll
            for (int r = 0; r < 32; ++r) C[t][r] = b0;
        }
    }

    // a unit of a tail tile writes its raw K-block sums (MKL order: one plane per K-block; free order: one plane) to the workspace
    float* wsp = g.ws + ((size_t)(xcd * g.tail_cnt + te) * g.planes) * SG_PLANE;
    auto write_plane = [&](int plane) {
        float* p = wsp + (size_t)plane * SG_PLANE;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    p[(64 * wm + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h) * SG_BN + 64 * wn + 32 * t + i] = acc[t][16 * b + r];
    };

    auto compute = [&](auto STAGE) {
        constexpr int st = decltype(STAGE)::value * SG_STAGE;
        const char* la = lds + st;
        sg_f32x4 va = *reinterpret_cast<const sg_f32x4*>(la + (a_row ^ 0));
        sg_f32x4 vb0 = *reinterpret_cast<const sg_f32x4*>(la + (b_row ^ 0));
        sg_f32x4 vb1 = *reinterpret_cast<const sg_f32x4*>(la + (b_row ^ 0) + 4096);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            sg_f32x4 na = va, nb0 = vb0, nb1 = vb1;
            if (q < 7) {                               // register double buffer over the eight 4-k pieces: piece q + 1 is read while piece q's 8 MFMAs issue
                na = *reinterpret_cast<const sg_f32x4*>(la + (a_row ^ ((q + 1) << 4)));
                nb0 = *reinterpret_cast<const sg_f32x4*>(la + (b_row ^ ((q + 1) << 4)));
                nb1 = *reinterpret_cast<const sg_f32x4*>(la + (b_row ^ ((q + 1) << 4)) + 4096);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x1f32(va[e], vb0[e], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x1f32(va[e], vb1[e], acc[1], 0, 0, 0);
            }
            va = na; vb0 = nb0; vb1 = nb1;
        }
    };

    const int nch = c1 - c0;
    int cb = 0, plane = (unit < 0 || !MKL) ? 0 : c0 / g.blk_chunks;
    issue(c0, 0);
    if (nch > 1) issue(c0 + 1, 1);
    auto body = [&](int kc, auto STAGE) {
        constexpr int s = decltype(STAGE)::value;
        if (kc + 1 < nch) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");       // my six pieces of chunk kc have landed (chunk kc + 1's may still fly)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                          // everyone's have; and the stage chunk kc + 2 goes to (read in iteration kc - 1) is free
        if (kc + 2 < nch) issue(c0 + kc + 2, (s + 2) % SG_STAGES);
        compute(STAGE);
        if (MKL && (++cb == g.blk_chunks || kc + 1 == nch)) {                     // K-block done: C += chain, the next chain starts from 0
            cb = 0;
            if (unit >= 0) { write_plane(plane); ++plane; }
            else {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 32; ++r) C[t][r] = C[t][r] + acc[t][r];
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 32; ++r) acc[t][r] = 0.f;
        }
    };
    for (int kc = 0; kc < nch; kc += 3) {
        body(kc, std::integral_constant<int, 0>{});
        if (kc + 1 < nch) body(kc + 1, std::integral_constant<int, 1>{});
        if (kc + 2 < nch) body(kc + 2, std::integral_constant<int, 2>{});
    }
    if (unit >= 0) {
        if (!MKL) write_plane(unit);
        return;
    }

    // ---- epilogue: lane (col = 64 wn + 32 t + i, rows 64 wm + 32 b + (r & 3) + 8 (r >> 2) + 4 h) ----
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = col0 + 64 * wn + 32 * t + i;
        const float bfree = (!MKL && g.bias != nullptr && !g.bias_last) ? g.bias[n] : 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = row0 + 64 * wm + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= g.M) continue;
                float v = MKL ? C[t][16 * b + r] : acc[t][16 * b + r] + bfree;
                g.c[(size_t)m * g.ldc + n] = sg_epilogue(g, v, m, n);
                __builtin_amdgcn_sched_barrier(0);          // one output at a time: the unrolled epilogue must not set the kernel's register count
            }
    }
}

// tail tiles: out = epilogue(((bias + plane 0) + plane 1) + ...), planes in K order.  One thread = 4 consecutive columns of one row.
template <bool MKL>
__global__ __launch_bounds__(256) void sg_tail_finish_kernel(SgArgs g)
{
    const int tt = blockIdx.y;                          // tail tile: xcd * tail_cnt + te
    const int xcd = tt / g.tail_cnt, te = tt - xcd * g.tail_cnt;
    const int e = g.full_pos + te, sidx = xcd * g.per + e;
    if (e >= g.per || sidx >= g.tiles) return;
    int tm, tn;
    sg_tile_of(sidx, g.mt, g.nt, tm, tn);
    const int idx = blockIdx.x * 256 + threadIdx.x;     // 0 .. 256 * 32 - 1
    const int r = idx >> 5, c4 = (idx & 31) << 2;
    const int m = tm * SG_BM + r;
    if (m >= g.M) return;
    const float* p = g.ws + (size_t)tt * g.planes * SG_PLANE + (size_t)r * SG_BN + c4;
    const int n = tn * SG_BN + c4;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (MKL && g.bias != nullptr && !g.bias_last) ? g.bias[n + j] : 0.f;
    for (int pl = 0; pl < g.planes; ++pl) {
        const sg_f32x4 x = *reinterpret_cast<const sg_f32x4*>(p + (size_t)pl * SG_PLANE);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] + x[j];
    }
    sg_f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (!MKL && g.bias != nullptr && !g.bias_last) v[j] = v[j] + g.bias[n + j];
        o[j] = sg_epilogue(g, v[j], m, n + j);
    }
    *reinterpret_cast<sg_f32x4*>(g.c + (size_t)m * g.ldc + n) = o;
}

// The launch plan (host): tile list split over the 8 XCDs, full rounds of 32 workgroups per XCD, the tail round split along K.
struct SgPlan { int mt, nt, tiles, per, full_pos, tail_cnt, split, planes, blk_chunks, nchunks; size_t ws_bytes; };

static SgPlan sg_plan(long M, int N, int K, bool mkl, size_t ws_avail, int force_split)
{
    SgPlan p{};
    p.mt = (int)((M + SG_BM - 1) / SG_BM); p.nt = N / SG_BN; p.tiles = p.mt * p.nt; p.per = (p.tiles + 7) / 8;
    p.nchunks = K / SG_BK;
    const int nblk = mkl ? (p.nchunks + 11) / 12 : 1;
    p.blk_chunks = mkl ? 12 : p.nchunks;
    p.full_pos = p.per / 32 * 32; p.tail_cnt = p.per - p.full_pos; p.split = 1; p.planes = 1;
    if (p.tail_cnt == 0) return p;
    // cost of the tail in tile times: ceil(tail_cnt * S / 32) rounds of 1 / S each (+ ~4 % per unit for its pipeline fill, plane write and the finish pass)
    int best = 1; double bc = 1.0;
    const int smax = mkl ? nblk : (p.nchunks < 8 ? p.nchunks : 8);
    for (int S = 2; S <= smax; ++S) {
        if (mkl ? (nblk % S != 0) : (p.nchunks % S != 0)) continue;
        const size_t ws = (size_t)8 * p.tail_cnt * (mkl ? nblk : S) * SG_PLANE * 4;
        if (ws > ws_avail) continue;
        const double c = (double)((p.tail_cnt * S + 31) / 32) / S + 0.04;
        if (c < bc - 0.02) { bc = c; best = S; }
    }
    if (force_split > 0) best = force_split;
    if (best > 1) {
        p.split = best; p.planes = mkl ? nblk : best;
        p.ws_bytes = (size_t)8 * p.tail_cnt * p.planes * SG_PLANE * 4;
    } else { p.full_pos = p.per; p.tail_cnt = 0; }
    return p;
}

}  // namespace selftok

using namespace selftok;

extern "C" {

size_t selftok_linear_f32_workspace_bytes(long M, int N, int K, int flags)
{
    if (M <= 0 || N <= 0 || K <= 0 || N % SG_BN || K % SG_BK) return 0;
    // the largest plan: every tail tile of every XCD split into all of its K-blocks (MKL order) or 8 units (free order)
    const bool mkl = (flags & SELFTOK_LINEAR_MKL_ORDER) != 0;
    const long mt = (M + SG_BM - 1) / SG_BM, tiles = mt * (N / SG_BN), per = (tiles + 7) / 8, tail = per % 32;
    const int planes = mkl ? (K / SG_BK + 11) / 12 : 8;
    return (size_t)8 * tail * planes * SG_PLANE * 4;
}

int selftok_linear_f32(const float* x, long ldx, const float* w, const float* bias, const float* res, long ldr, int res_mod, const float* gate, long ldg,
                       int gate_mod, float* out, long ldo, long M, int N, int K, int flags, void* workspace, size_t workspace_bytes, hipStream_t stream)
{
    if (M == 0) return SELFTOK_OK;
    if (!x || !w || !out || M < 0 || M > 0x7fffffffL || N <= 0 || K <= 0 || N % SG_BN || K % SG_BK || ldx % 4 || ldx < K || ldo < N || ldo % 4 || (gate && !res) ||
        (size_t)SG_BM * (size_t)ldx * 4 > 0xffffffffull) {
        set_last_error("linear_f32: need N % 128 == 0, K % 32 == 0, 16-byte aligned rows (ldx % 4 == 0, ldo % 4 == 0), gate only with res");
        return SELFTOK_EINVAL;
    }
    const bool mkl = (flags & SELFTOK_LINEAR_MKL_ORDER) != 0;
    if (mkl && K > 384 && K < 768) { set_last_error("linear_f32: MKL order for 384 < K < 768 (two half blocks) is served by selftok_ex_linear_f32"); return SELFTOK_EINVAL; }
    const int force = (flags >> 8) & 0xff;                       // tools / tests: SELFTOK_LINEAR_SPLIT(n) forces the tail split
    SgPlan p = sg_plan(M, N, K, mkl, workspace ? workspace_bytes : 0, 0);
    if (force > 0) {
        p = sg_plan(M, N, K, mkl, (size_t)-1, force);
        if (p.ws_bytes > (workspace ? workspace_bytes : 0)) { set_last_error("linear_f32: forced tail split needs a larger workspace"); return SELFTOK_EINVAL; }
        if (p.tail_cnt > 0 && (mkl ? ((p.nchunks + 11) / 12) % force : p.nchunks % force)) { set_last_error("linear_f32: forced split does not divide K"); return SELFTOK_EINVAL; }
    }
    SgArgs g{};
    g.a = x; g.lda = ldx; g.b = w; g.c = out; g.ldc = ldo; g.bias = bias;
    g.res = res; g.ldr = ldr; g.res_mod = res_mod; g.gate = gate; g.ldg = ldg; g.gate_mod = gate_mod;
    g.ws = (float*)workspace;
    g.M = (int)M; g.N = N; g.K = K; g.gelu = (flags & SELFTOK_LINEAR_GELU) ? 1 : 0; g.bias_last = (flags & SELFTOK_LINEAR_BIAS_LAST) ? 1 : 0;
    g.mt = p.mt; g.nt = p.nt; g.tiles = p.tiles; g.per = p.per; g.full_pos = p.full_pos; g.tail_cnt = p.tail_cnt; g.split = p.split; g.planes = p.planes;
    g.blk_chunks = (!mkl && p.split > 1) ? p.nchunks / p.split : p.blk_chunks; g.nchunks = p.nchunks;
    const unsigned grid = 8u * (unsigned)(p.full_pos + p.tail_cnt * p.split);
    if (mkl) hipLaunchKernelGGL((sg_gemm_kernel<true>), dim3(grid), dim3(512), 0, stream, g);
    else hipLaunchKernelGGL((sg_gemm_kernel<false>), dim3(grid), dim3(512), 0, stream, g);
    int rc = check_launch("sg_gemm_kernel");
    if (rc || p.tail_cnt == 0) return rc;
    if (mkl) hipLaunchKernelGGL((sg_tail_finish_kernel<true>), dim3(32, 8 * p.tail_cnt), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((sg_tail_finish_kernel<false>), dim3(32, 8 * p.tail_cnt), dim3(256), 0, stream, g);
    return check_launch("sg_tail_finish_kernel");
}

}  // extern "C"
