// fp32 Linear on the fp32-input matrix cores with BOTH operands staged by LDS-DMA (round 6).  gfx950 only.
//
//   out[m][n] = epilogue( sum_k x[m][k] w[n][k] )          x [M][K] (row stride ldx), w [N][K] contiguous -- nn.Linear / F.linear
//
// replaces (a) in gemm='exact' the wide Linears of the MMDiT joint blocks and of the Q-Former (mimogpt/models/selftok/sd3/mmdit.py:266-307, 413-419;
// modules.py:186-199, 293) that xe_gemm128_kernel (csrc/encoder_exact.hip) carried at 0.76 - 0.84 of the fp32 matrix peak -- same bits: MKL sgemm's order, one
// sequential fmaf chain per K-block of 384, out = ((bias + c0) + c1) + ... -- and offers (b) a FREE-order form (flags = 0: the whole K is ONE chain per output), built as a candidate for
// gemm='fp32' and not wired in: over the 204 Linear shapes of a B = 64 step the tuned library kernels lose to it at 7 (0.06 % of the time; profiles/r6_sweep_fp32_linear_vs_sg.txt).
//
// Design:
//   * v_mfma_f32_32x32x1_2b_f32: one k per instruction and TWO 32 x 32 blocks.  A lane supplies A[row 32 h + i][k] and B[col i][k] (h = lane / 32, i = lane % 32):
//     the two blocks are the two row halves of a 64-row wave tile.  An fp32-input MFMA is fma(a, b, acc) per output, bit for bit (csrc/vq.hip, round 1), so a
//     k-ascending instruction stream IS the sequential chain -- with the operands in their NATURAL row-major order.  The 32x32x2 form xe_gemm128 uses wants the even
//     and the odd k of a row in different lane halves: a de-interleave while staging = global -> VGPR -> 8 v_mov -> ds_write_b128, on the VALU port the fp32 MFMAs
//     share.  With the natural order a 16-byte piece of a row is what a lane needs as it lies in memory, so
//   * both operand tiles reach LDS by `global_load_lds_dwordx4` (1 KiB = 8 rows x 128 bytes per wave instruction; uniform SGPR base + one constant per-lane offset
//     register per piece; the 16-byte chunks of a row XOR-swizzled on the SOURCE side so that every ds_read_b128 lane group covers all 64 banks once: 0 bank
//     conflicts in the PMC pass): no staging registers, no ds_write, no VALU in the k-loop besides the K-block fold.
//   * workgroup = EIGHT compute waves = 256 x 128 outputs (wave: 64 x 64 = two 2-block accumulators, 64 VGPRs; two compute waves per SIMD) + TWO LOADER waves
//     (one issues the A tile's 32 DMA pieces of a chunk, the other the B tile's 16), 32-k chunks, THREE stages of 48 KiB (144 of the CU's 160 KiB: one workgroup
//     per CU); one barrier per chunk.  How the measurements led here (profiles/r6_sgemm_v*.txt, r6_mfma_f32_forms*.txt; everything below is pipe CYCLES -- the
//     bare MFMA loops hold 2.39 GHz, the finished kernel does not: see the note on the clock at the end of this list):
//       - a DMA piece costs the ISSUING wave 60 - 180 cycles during which its MFMA stream pauses: with the pieces issued by the compute waves the kernel lost 6 - 8 %
//         wherever they were placed ("no DMA" 0.88 / 0.94, "DMA issued, waits removed" = the product) -> loader waves, which never touch the matrix pipe;
//       - ONE compute wave per SIMD pays ~6 pipe cycles for every instruction between two MFMAs (fragment reads, waits): 69.5 cycles per MFMA in the k-loop, 0.865
//         (in-kernel stamps; a bare loop of that shape: 0.66 with one wave per SIMD, 0.80 - 0.93 with two) -> two compute waves per SIMD: the partner's MFMA takes
//         the slot (8791 cycles per chunk for the 128 MFMAs of a SIMD = 0.93 in the loop);
//       - two INDEPENDENT workgroups per CU do not interleave: issue arbitration is by age, the older workgroup's waves issue back to back and the younger one's
//         prologue / first MFMAs wait (census: loops of the two workgroups of a CU never overlapped) -- and a 65-KiB workgroup is admitted once per CU whatever the
//         occupancy query says.  So: one workgroup per CU, its stages deep enough (3) that the DMA round trip (~4300 cycles under load, longer than a chunk's
//         MFMAs) is off the critical path.
//     Prologue (first DMA round trip, 1.9 us) and epilogue (1.7 us plain) of a 176-us tile are what is left outside the loop.
//       - THE CLOCK: per cycle this kernel equals the vendor's (PMC: matrix pipe busy 0.926 - 0.929 of GUI-active, hipBLASLt 0.925), in time it is 0.85 - 0.90 against
//         0.92 - 0.96 -- a probe kernel beside it reads 2.14 - 2.27 GHz (beside hipBLASLt: 2.39; tools/clock_probe.py).  The K = 1 MFMA form moves its 32 accumulator
//         registers per instruction (4x the register traffic per FLOP of the vendor's 16x16x4) and with the GEMM's DMA / L2 traffic crosses what the chip sustains at
//         2.4 GHz; every K >= 2 form holds the clock but needs lane halves to hold ALTERNATE k, which row-major operands give only at a cost (ds_read2_b32, lane
//         swaps, k-interleaved operands: 0.86 - 0.91 in tools/microbench/mfma_power_forms.hip against this form's 0.89 - 0.93 there).  DESIGN.md section 16.4.
//   * XCD-aware 1-D tile order (bands of 8 row tiles: the 64 workgroups resident on an XCD cover 8 x 8 tiles = 8 A + 8 B operand tiles per chunk round).
//   * the TAIL ROUND is split along K: the tiles left over after the last full round of 256 tiles (one per CU) are computed as S units of 1 / S of the K range each, raw
//     partial sums to a workspace, and a small second kernel adds the planes IN ORDER and runs the epilogue.  In MKL order a unit is ONE K-block (S = the number of
//     K-blocks), its plane the block's chain, so the result is still ((bias + c0) + c1) + ... bit for bit; in free order a unit is 1 / S of the chunks.
//
// Epilogue (both kernels): [+ bias last] -> [gate * y] -> [res + y], each separately rounded (-ffp-contract=off), the operation sequence of csrc/encoder_exact.hip's
// epilogue; GELU(tanh) (fc1 -> act: no res / gate there) is a second launch of the element-wise kernel over `out`, issued by the C entry.
#include "common.h"
#include "exact_math.h"
#include "selftok_hip.h"

#include <type_traits>

#pragma clang diagnostic ignored "-Winline-asm"      // M0 on the clobber list of the LDS-DMA statement is deliberate (sg_dma16)

// tools builds only (-DSG_ABL=n, tools/ablate_sgemm.sh): timing-only ablations of the k-loop -- 1: no DMA (no issue, no vmcnt waits), 2: DMA issued, vmcnt waits
// removed, 3: no DMA and no barriers, 4: DMA + waits + barriers but no fragment reads (operands stay what the first piece held).  Results are garbage.
#ifndef SG_ABL
#define SG_ABL 0
#endif
#ifndef SG_ONE_WG
#define SG_ONE_WG 0
#endif

namespace selftok {

typedef float sg_f32x32 __attribute__((ext_vector_type(32)));
typedef float sg_f32x4 __attribute__((ext_vector_type(4)));

constexpr int SG_BM = 256, SG_BN = 128, SG_BK = 32;
constexpr int SG_A_BYTES = SG_BM * SG_BK * 4;                 // 32 KiB
constexpr int SG_STAGE = (SG_BM + SG_BN) * SG_BK * 4;         // 48 KiB
constexpr int SG_STAGES = 3;
constexpr int SG_PLANE = SG_BM * SG_BN;                       // floats per workspace plane
constexpr int SG_CW = 8;                                      // compute waves: 4 (rows) x 2 (columns) wave tiles of 64 x 64
constexpr int SG_THREADS = 64 * (SG_CW + 2);                  // + 2 loader waves
constexpr int SG_ROUND = 32;                                  // one tile per CU and tile time: the two workgroups of a CU run one BEHIND the other (issue priority by age), the
                                                              // second one only fills the first one's prologue / epilogue / stalls (profiles/r6_sgemm_v4_stamps_census.txt)

struct SgArgs {
    const float* a; long lda;                   // x [M][K], row stride lda (floats, multiple of 4)
    const float* b;                             // w [N][K] contiguous
    float* c; long ldc;
    const float* bias;
    const float* res; long ldr; int res_mod;    // y = res[row(m, res_mod)][n] + (gate ? gate * y : y); row(m, d) = m % d (d > 0), m / -d (d < 0), m (0)
    const float* gate; long ldg; int gate_mod;
    float* ws;                                  // tail planes: [xcd][tail tile][plane][128][128]
    int M, N, K;
    int gelu, bias_last;
    int mt, nt, tiles, per;                     // tile grid; per = ceil(tiles / 8) list entries per XCD
    int full_pos, tail_cnt, split, planes;      // entries < full_pos of every XCD's list: full tiles; the next tail_cnt: `split` units each; planes per tail tile
    int blk_chunks;                             // MKL order: K-block length in chunks (12 = 384 / 32); free order: chunks per unit (K / 32 when nothing is split)
    int nchunks;                                // K / 32
    int flat_prio;                              // tools: keep the raised priority through the k-loop
};

// list entry `sidx` (0 .. tiles-1) -> (row tile, column tile): bands of 8 row tiles, inside a band column tile by column tile
__host__ __device__ inline void sg_tile_of(int sidx, int MT, int NT, int& tm, int& tn)
{
    const int full = MT >> 3, rem = MT & 7, cut = full * 8 * NT;
    if (sidx < cut) { const int band = sidx / (8 * NT), r = sidx - band * 8 * NT; tn = r >> 3; tm = band * 8 + (r & 7); }
    else { const int r = sidx - cut; tn = r / rem; tm = full * 8 + (r - tn * rem); }
}

__device__ __forceinline__ void sg_dma16(const void* base, unsigned voff, unsigned lds)
{
    // M0 = LDS byte address of the 1-KiB piece (declared clobbered: no compiler version may keep a value of its own in M0 across this)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory", "m0");
}

__device__ __forceinline__ float sg_epilogue(const SgArgs& g, float v, int m, int n)
{
    if (g.bias_last && g.bias != nullptr) v = v + g.bias[n];
    if (g.gate != nullptr) v = g.gate[(size_t)(g.gate_mod > 0 ? m % g.gate_mod : (g.gate_mod < 0 ? m / -g.gate_mod : m)) * g.ldg + n] * v;
    if (g.res != nullptr) v = g.res[(size_t)(g.res_mod > 0 ? m % g.res_mod : (g.res_mod < 0 ? m / -g.res_mod : m)) * g.ldr + n] + v;
    return v;
}

template <bool MKL>
__global__ __launch_bounds__(SG_THREADS, 3) void sg_gemm_kernel(SgArgs g)
{
    __shared__ __attribute__((aligned(1024))) char lds[SG_STAGES * SG_STAGE];
    // Issue arbitration on a SIMD is by priority, then AGE.  The prologue and the loader waves run at raised priority (short, and the DMA round trips they start are
    // what the loop waits for), the k-loop at priority 0, the epilogue raised again (the next workgroup's prologue overlaps it).
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = (wave >> 2) & 1;
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
            c0 = unit * g.blk_chunks;                  // a unit = ONE K-block (MKL order: 12 chunks; free order: blk_chunks = nchunks / split)
            c1 = min(c0 + g.blk_chunks, g.nchunks);
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
    const int nch = c1 - c0;
#ifdef SG_STAMP
    // tools: s_memtime at [before barrier, after barrier, after the chunk's MFMAs were issued] of the first 40 chunks, waves 0 (compute) and 4 (loader) of the
    // workgroups at list positions 256 and 288 of XCD 0 (two workgroups of the fifth round: every CU holds two by then) -> g.ws as u64 [wg 2][wave 2][chunk 40][3]
    unsigned long long* stamp = nullptr;
    if (xcd == 0 && (pos == 128 || pos == 160) && (wave == 0 || wave == SG_CW) && lane == 0)
        stamp = reinterpret_cast<unsigned long long*>(g.ws) + ((pos == 160 ? 1 : 0) * 2 + (wave ? 1 : 0)) * 120;
#define SG_TS(kc, j) do { if (stamp && (kc) < 40) stamp[(kc) * 3 + (j)] = __builtin_readcyclecounter(); } while (0)
    // residency census: every workgroup's (start, end) in 100 MHz ticks and where it ran -> u64 [blockIdx][3] behind the first 8 KiB
    struct SgCensus {
        unsigned long long* p; unsigned long long t0, t1 = 0, t2 = 0;
        __device__ SgCensus(unsigned long long* q) : p(q), t0(__builtin_amdgcn_s_memrealtime()) {}
        __device__ ~SgCensus() {
            if (!p) return;
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            p[0] = t0; p[1] = __builtin_amdgcn_s_memrealtime(); p[2] = ((unsigned long long)(xcc & 0xF) << 32) | hw; p[3] = t1; p[4] = t2;
        }
    } census((tid == 0 && blockIdx.x < 16384) ? reinterpret_cast<unsigned long long*>(g.ws) + 1024 + 5 * blockIdx.x : nullptr);
#define SG_CENSUS_LOOP_START() do { if (census.p && census.t1 == 0) census.t1 = __builtin_amdgcn_s_memrealtime(); } while (0)
#define SG_CENSUS_LOOP_END() do { if (census.p) census.t2 = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SG_TS(kc, j) do { } while (0)
#define SG_CENSUS_LOOP_START() do { } while (0)
#define SG_CENSUS_LOOP_END() do { } while (0)
#endif
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)&lds[0];

    if (wave >= SG_CW) {
        // ---- the two LOADER waves: wave 8 stages the A tile (32 pieces of 8 rows x 128 bytes per chunk), wave 9 the B tile (16).  They never touch the matrix
        // pipe; the compute waves never issue a VMEM instruction.  Per chunk: [chunk kc landed] -> barrier kc -> issue chunk kc + 1 into the other stage (every
        // compute wave finished reading chunk kc - 1 before it arrived at barrier kc). ----
        const bool isA = wave == SG_CW;
        const char* base = isA ? reinterpret_cast<const char*>(g.a + (size_t)row0 * g.lda) : reinterpret_cast<const char*>(g.b + (size_t)col0 * g.K);
        const unsigned rs = (unsigned)(isA ? g.lda : (long)g.K) * 4u;
        const int rlast = isA ? g.M - 1 - row0 : SG_BN - 1;                      // ragged last row tile: fetch the matrix's last row instead (never stored)
        const int pr = lane >> 3, pc = lane & 7;                                  // row inside a piece, 16-byte position inside the 128-byte row
        const unsigned swz0 = (unsigned)((pc ^ (pr >> 1)) << 4), swz1 = (unsigned)((pc ^ (4 | (pr >> 1))) << 4);      // chunk (pc ^ ((tile row >> 1) & 7)): even / odd pieces
        auto issue = [&](int chunk, int stage) {
            if (SG_ABL == 1 || SG_ABL == 3) return;
            const char* cb = base + (size_t)chunk * (SG_BK * 4);
            const unsigned l = lds0 + stage * SG_STAGE + (isA ? 0 : SG_A_BYTES);
            if (isA) {
#pragma unroll
                for (int p = 0; p < SG_BM / 8; ++p) sg_dma16(cb, (unsigned)min(8 * p + pr, rlast) * rs + ((p & 1) ? swz1 : swz0), l + p * 1024);
            } else {
#pragma unroll
                for (int p = 0; p < SG_BN / 8; ++p) sg_dma16(cb, (unsigned)(8 * p + pr) * rs + ((p & 1) ? swz1 : swz0), l + p * 1024);
            }
        };
        issue(c0, 0);
        if (nch > 1) issue(c0 + 1, 1);
        for (int kc = 0; kc < nch; ++kc) {
            // chunk kc has landed: at most the pieces of chunk kc + 1 (32 of the A tile / 16 of the B tile) may still fly
            if (SG_ABL != 1 && SG_ABL != 2 && SG_ABL != 3) {
                if (kc + 1 >= nch) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (isA) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            }
            SG_TS(kc, 0);
            if (SG_ABL != 3) __syncthreads();
            SG_TS(kc, 1);
            if (kc + 2 < nch) issue(c0 + kc + 2, (kc + 2) % SG_STAGES);      // its stage held chunk kc - 1: every compute wave finished it before barrier kc
            SG_TS(kc, 2);
        }
        return;
    }

    // ---- the eight COMPUTE waves.  Fragment addresses: A row 64 wm + lane, B rows 64 wn + 32 t + i; chunk q of a row sits at position q ^ ((lane >> 1) & 7) ----
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
            // a tail unit's plane is its raw chain: C starts at -0 there (-0 + x = x for every x, signed zeros included), so that after the loop only C is live
            const float b0 = unit >= 0 ? -0.f : ((g.bias != nullptr && !g.bias_last) ? g.bias[col0 + 64 * wn + 32 * t + i] : 0.f);
#pragma unroll
            for (int r = 0; r < 32; ++r) C[t][r] = b0;
        }
    }

    // ---- the k-loop: per chunk barrier (the chunk has landed, says the loaders' vmcnt(0) in front of it) -> 8 x (3 fragment reads of the next 4 k, 8 MFMAs) ----
    int cb = 0;
    sg_f32x4 va, vb0, vb1;
    if (!g.flat_prio) __builtin_amdgcn_s_setprio(0);
    auto ldq = [&](const char* la, int q, sg_f32x4& a, sg_f32x4& b0, sg_f32x4& b1) {
        if (SG_ABL == 4 && q > 0) return;
        a = *reinterpret_cast<const sg_f32x4*>(la + (a_row ^ (q << 4)));
        b0 = *reinterpret_cast<const sg_f32x4*>(la + (b_row ^ (q << 4)));
        b1 = *reinterpret_cast<const sg_f32x4*>(la + (b_row ^ (q << 4)) + 4096);
    };
    auto body = [&](int kc, auto STAGE) {
        constexpr int s = decltype(STAGE)::value;
        const char* la = lds + s * SG_STAGE;
        SG_TS(kc, 0);
        if (SG_ABL != 3) __syncthreads();
        SG_TS(kc, 1);
        SG_CENSUS_LOOP_START();
        ldq(la, 0, va, vb0, vb1);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            sg_f32x4 na = va, nb0 = vb0, nb1 = vb1;
            if (q < 7) ldq(la, q + 1, na, nb0, nb1);         // register double buffer over the 4-k pieces: piece q + 1 is read while piece q's 8 MFMAs issue
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x1f32(va[e], vb0[e], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x1f32(va[e], vb1[e], acc[1], 0, 0, 0);
            }
            va = na; vb0 = nb0; vb1 = nb1;
        }
        SG_TS(kc, 2);
        if (MKL && (++cb == g.blk_chunks || kc + 1 == nch)) {                     // K-block done: C += chain, the next chain starts from 0 (a tail unit IS one K-block)
            cb = 0;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 32; ++r) { C[t][r] = C[t][r] + acc[t][r]; acc[t][r] = 0.f; }
        }
    };
    for (int kc = 0; kc < nch; kc += 3) {
        body(kc, std::integral_constant<int, 0>{});
        if (kc + 1 < nch) body(kc + 1, std::integral_constant<int, 1>{});
        if (kc + 2 < nch) body(kc + 2, std::integral_constant<int, 2>{});
    }
    SG_CENSUS_LOOP_END();
    if (unit >= 0) {
        // a unit of a tail tile: its raw sum (MKL order: K-block `unit`; free order: chunks c0 .. c1) goes to plane `unit` of the tile's workspace slot
        float* p = g.ws + ((size_t)(xcd * g.tail_cnt + te) * g.planes + unit) * SG_PLANE;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    p[(64 * wm + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h) * SG_BN + 64 * wn + 32 * t + i] = MKL ? C[t][16 * b + r] : acc[t][16 * b + r];
                    __builtin_amdgcn_sched_barrier(0);
                }
        return;
    }

    // ---- epilogue: lane (col = 64 wn + 32 t + i, rows 64 wm + 32 b + (r & 3) + 8 (r >> 2) + 4 h).  The epilogue's VALU work shares the SIMD with the other
    // workgroup's MFMAs (they do not overlap: same port), so it is kept short: uniform decisions hoisted, row addresses scalar (row base in SGPRs + one
    // per-lane offset), the table rows of gate / res looked up in LDS. ----
    __builtin_amdgcn_s_setprio(3);
    const bool plain = g.gate == nullptr && g.res == nullptr && !(g.bias_last && g.bias != nullptr);
    const bool whole = row0 + SG_BM <= g.M;
    const int ncol = col0 + 64 * wn + i;
    if (plain && whole) {
        const size_t loff = (size_t)(4 * h) * g.ldc + ncol;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float bfree = (!MKL && g.bias != nullptr) ? g.bias[ncol + 32 * t] : 0.f;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* rowp = g.c + (size_t)(row0 + 64 * wm + 32 * b + (r & 3) + 8 * (r >> 2)) * g.ldc;       // uniform: SGPR pair
                    rowp[loff + 32 * t] = MKL ? C[t][16 * b + r] : acc[t][16 * b + r] + bfree;
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        return;
    }
    const bool has_gate = g.gate != nullptr, has_res = g.res != nullptr, bias_l = g.bias_last && g.bias != nullptr;
    // the row of the gate / res table for every row of the tile: one integer division per ROW (not per output), kept in the first KiB of the stage memory -- free now:
    // every chunk was consumed.  (A separate 1-KiB array made the workgroup 65 KiB and the hardware then admitted ONE workgroup per CU, whatever the occupancy query
    // said: profiles/r6_sgemm_v8_stamps.txt.)  The loaders have left; the barrier counts the eight compute waves.
    int (*rowidx)[SG_BM] = reinterpret_cast<int (*)[SG_BM]>(lds);
    __syncthreads();                           // nobody still reads the last chunk's fragments
    if (tid < SG_BM && (has_gate || has_res)) {
        const int m = min(row0 + tid, g.M - 1);
        rowidx[0][tid] = g.gate_mod > 0 ? m % g.gate_mod : (g.gate_mod < 0 ? m / -g.gate_mod : m);
        rowidx[1][tid] = g.res_mod > 0 ? m % g.res_mod : (g.res_mod < 0 ? m / -g.res_mod : m);
    }
    __syncthreads();
    // 16 outputs (one 32 x 32 block's rows of this lane) at a time: their gate / res operands are loaded TOGETHER, then combined and stored -- one output at a
    // time made every output wait for its own two loads (64 serial round trips per lane: proj ran at 0.65 of the peak, profiles/r6_sgemm_v6b_model_epilogues.txt)
    // addresses as (uniform 64-bit base) + (32-bit per-lane byte offset): the launcher guarantees that the tables and the output stay below 4 GiB
    const char* gbase = reinterpret_cast<const char*>(g.gate);
    const char* rbase = reinterpret_cast<const char*>(g.res);
    char* cbase = reinterpret_cast<char*>(g.c + (size_t)row0 * g.ldc);
    const unsigned ldg4 = (unsigned)g.ldg * 4u, ldr4 = (unsigned)g.ldr * 4u, ldc4 = (unsigned)g.ldc * 4u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = ncol + 32 * t;
        const unsigned n4 = (unsigned)n * 4u;
        const float bfree = (!MKL && g.bias != nullptr && !g.bias_last) ? g.bias[n] : 0.f;
        const float blast = bias_l ? g.bias[n] : 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            // EB outputs at a time (16 = one 32 x 32 block's rows of this lane)
            constexpr int EB = 16;
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += EB) {
                float gv[EB], rv[EB];
#pragma unroll
                for (int r = 0; r < EB; ++r) { gv[r] = 1.f; rv[r] = 0.f; }
                if (has_gate) {                             // the uniform decision OUTSIDE the loads: inside, every load sat in its own branch with its own wait
#pragma unroll
                    for (int r = 0; r < EB; ++r)
                        gv[r] = *reinterpret_cast<const float*>(gbase + ((unsigned)rowidx[0][64 * wm + 32 * b + ((r0 + r) & 3) + 8 * ((r0 + r) >> 2) + 4 * h] * ldg4 + n4));
                }
                if (has_res) {
#pragma unroll
                    for (int r = 0; r < EB; ++r)
                        rv[r] = *reinterpret_cast<const float*>(rbase + ((unsigned)rowidx[1][64 * wm + 32 * b + ((r0 + r) & 3) + 8 * ((r0 + r) >> 2) + 4 * h] * ldr4 + n4));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < EB; ++r) {
                    const int ml = 64 * wm + 32 * b + ((r0 + r) & 3) + 8 * ((r0 + r) >> 2) + 4 * h;
                    float v = MKL ? C[t][16 * b + r0 + r] : acc[t][16 * b + r0 + r] + bfree;
                    if (bias_l) v = v + blast;
                    if (has_gate) v = gv[r] * v;
                    if (has_res) v = rv[r] + v;
                    if (whole || row0 + ml < g.M) *reinterpret_cast<float*>(cbase + ((unsigned)ml * ldc4 + n4)) = v;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
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
    const int idx = blockIdx.x * 256 + threadIdx.x;     // 0 .. SG_BM * 32 - 1
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

// The launch plan (host): tile list split over the 8 XCDs, full rounds of 32 tiles per XCD (one per CU and tile time), the tail round split along K.
struct SgPlan { int mt, nt, tiles, per, full_pos, tail_cnt, split, planes, blk_chunks, nchunks; size_t ws_bytes; };

static SgPlan sg_plan(long M, int N, int K, bool mkl, size_t ws_avail, int force_split)
{
    SgPlan p{};
    p.mt = (int)((M + SG_BM - 1) / SG_BM); p.nt = N / SG_BN; p.tiles = p.mt * p.nt; p.per = (p.tiles + 7) / 8;
    p.nchunks = K / SG_BK;
    const int nblk = mkl ? (p.nchunks + 11) / 12 : 1;
    p.blk_chunks = mkl ? 12 : p.nchunks;
    p.full_pos = p.per / SG_ROUND * SG_ROUND; p.tail_cnt = p.per - p.full_pos; p.split = 1; p.planes = 1;
    if (p.tail_cnt == 0) return p;
    // cost of the tail in tile times: ceil(tail_cnt * S / 32) rounds of 1 / S each (+ ~8 % of a tile per unit round for pipeline fill, plane write and the finish pass).
    // MKL order: a unit is ONE K-block (S = the number of K-blocks: the planes are the blocks, added in order by the finish kernel)
    int best = 1; double bc = 1.0;
    for (int S = 2; S <= (mkl ? nblk : 8); ++S) {
        if (mkl ? (S != nblk) : (p.nchunks % S != 0)) continue;
        const size_t ws = (size_t)8 * p.tail_cnt * S * SG_PLANE * 4;
        if (ws > ws_avail) continue;
        const int rounds = (p.tail_cnt * S + SG_ROUND - 1) / SG_ROUND;
        const double c = (double)rounds / S + 0.08 * rounds;
        if (c < bc - 0.05) { bc = c; best = S; }
    }
    if (force_split > 0) best = force_split;
    if (best > 1) {
        p.split = best; p.planes = best;
        p.ws_bytes = (size_t)8 * p.tail_cnt * p.planes * SG_PLANE * 4;
    } else { p.full_pos = p.per; p.tail_cnt = 0; }
    return p;
}

}  // namespace selftok

using namespace selftok;

extern "C" {

#ifdef SG_STAMP
int selftok_sg_occupancy(int mkl)       // tools: workgroups per CU the runtime admits for the kernel
{
    int n = -1;
    if (mkl) hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sg_gemm_kernel<true>, SG_THREADS, 0);
    else hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sg_gemm_kernel<false>, SG_THREADS, 0);
    return n;
}
#endif

size_t selftok_linear_f32_workspace_bytes(long M, int N, int K, int flags)
{
    if (M <= 0 || N <= 0 || K <= 0 || N % SG_BN || K % SG_BK) return 0;
    // the largest plan: every tail tile of every XCD split into all of its K-blocks (MKL order) or 8 units (free order)
    const bool mkl = (flags & SELFTOK_LINEAR_MKL_ORDER) != 0;
    const long mt = (M + SG_BM - 1) / SG_BM, tiles = mt * (N / SG_BN), per = (tiles + 7) / 8, tail = per % SG_ROUND;
    const int planes = mkl ? (K / SG_BK + 11) / 12 : 8;
    return (size_t)8 * tail * planes * SG_PLANE * 4;
}

int selftok_linear_f32(const float* x, long ldx, const float* w, const float* bias, const float* res, long ldr, int res_mod, const float* gate, long ldg,
                       int gate_mod, float* out, long ldo, long M, int N, int K, int flags, void* workspace, size_t workspace_bytes, hipStream_t stream)
{
    if (M == 0) return SELFTOK_OK;
    if (!x || !w || !out || M < 0 || M > 0x7fffffffL || N <= 0 || K <= 0 || N % SG_BN || K % SG_BK || ldx % 4 || ldx < K || ldo < N || ldo % 4 || (gate && !res) ||
        (size_t)SG_BM * (size_t)ldx * 4 > 0xffffffffull || (size_t)SG_BM * (size_t)ldo * 4 > 0xffffffffull ||
        (res && (size_t)M * (size_t)ldr * 4 > 0xffffffffull) || (gate && (size_t)M * (size_t)ldg * 4 > 0xffffffffull)) {
        set_last_error("linear_f32: need N % 128 == 0, K % 32 == 0, 16-byte aligned rows (ldx % 4 == 0, ldo % 4 == 0), gate only with res, res / gate tables below 4 GiB");
        return SELFTOK_EINVAL;
    }
    const bool mkl = (flags & SELFTOK_LINEAR_MKL_ORDER) != 0;
    // GELU (fc1 -> act, no residual in the reference) is a second launch over `out`: ~80 VALU instructions per output inside the GEMM's epilogue kept four waves per
    // CU busy while the matrix pipe idled (+0.40 ms on a 2.28 ms Linear) and set the kernel's register count; as an element-wise pass over every SIMD it is +0.2 ms
    const bool gelu = (flags & SELFTOK_LINEAR_GELU) != 0;
    if (gelu && (res || gate || ldo != N)) { set_last_error("linear_f32: SELFTOK_LINEAR_GELU needs a contiguous out and no res / gate (the Mlp's fc1 -> act)"); return SELFTOK_EINVAL; }
    if (mkl && K > 384 && K < 768) { set_last_error("linear_f32: MKL order for 384 < K < 768 (two half blocks) is served by selftok_ex_linear_f32"); return SELFTOK_EINVAL; }
    const int force = (flags >> 8) & 0xff;                       // tools / tests: SELFTOK_LINEAR_SPLIT(n) forces the tail split
    SgPlan p = sg_plan(M, N, K, mkl, workspace ? workspace_bytes : 0, 0);
    if (force > 0) {
        p = sg_plan(M, N, K, mkl, (size_t)-1, force);
        if (p.ws_bytes > (workspace ? workspace_bytes : 0)) { set_last_error("linear_f32: forced tail split needs a larger workspace"); return SELFTOK_EINVAL; }
        if (p.tail_cnt > 0 && (mkl ? (p.nchunks + 11) / 12 != force : p.nchunks % force != 0)) { set_last_error("linear_f32: forced split must divide K / 32 (MKL order: equal the number of K-blocks)"); return SELFTOK_EINVAL; }
    }
    SgArgs g{};
    g.a = x; g.lda = ldx; g.b = w; g.c = out; g.ldc = ldo; g.bias = bias;
    g.res = res; g.ldr = ldr; g.res_mod = res_mod; g.gate = gate; g.ldg = ldg; g.gate_mod = gate_mod;
    g.ws = (float*)workspace;
    g.flat_prio = (flags >> 16) & 1;
    g.M = (int)M; g.N = N; g.K = K; g.gelu = 0; g.bias_last = (flags & SELFTOK_LINEAR_BIAS_LAST) ? 1 : 0;
    g.mt = p.mt; g.nt = p.nt; g.tiles = p.tiles; g.per = p.per; g.full_pos = p.full_pos; g.tail_cnt = p.tail_cnt; g.split = p.split; g.planes = p.planes;
    g.blk_chunks = (!mkl && p.split > 1) ? p.nchunks / p.split : p.blk_chunks; g.nchunks = p.nchunks;
    const unsigned grid = 8u * (unsigned)(p.full_pos + p.tail_cnt * p.split);
    if (mkl) hipLaunchKernelGGL((sg_gemm_kernel<true>), dim3(grid), dim3(SG_THREADS), 0, stream, g);
    else hipLaunchKernelGGL((sg_gemm_kernel<false>), dim3(grid), dim3(SG_THREADS), 0, stream, g);
    int rc = check_launch("sg_gemm_kernel");
    if (rc) return rc;
    if (p.tail_cnt > 0) {
        if (mkl) hipLaunchKernelGGL((sg_tail_finish_kernel<true>), dim3(SG_PLANE / 1024, 8 * p.tail_cnt), dim3(256), 0, stream, g);
        else hipLaunchKernelGGL((sg_tail_finish_kernel<false>), dim3(SG_PLANE / 1024, 8 * p.tail_cnt), dim3(256), 0, stream, g);
        if ((rc = check_launch("sg_tail_finish_kernel")) != 0) return rc;
    }
    return gelu ? selftok_ex_unary_f32(out, out, M * (long)N, 0, stream) : SELFTOK_OK;
}

}  // extern "C"
