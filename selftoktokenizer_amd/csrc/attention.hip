// Two-segment flash attention with an implicit prefix-visibility mask, fp32 in / fp32 out, on the
// gfx950 fp32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate).
//
// Replaces the reference's  attention(q, k, v, heads, mask)  = F.scaled_dot_product_attention with a
// materialised bool mask [B,1,S,S]  (sd3/other_impls.py:37-45, called from block_mixing sd3/mmdit.py:529-530,
// mask built at sd3/mmdit.py:1041-1094) and the two SDPA calls of DualAttention's uni branch
// (modules.py:235-238 latent stream, :263-266 query stream over cat(to_query_kv(x), query_kv)).
//
// Keys/values (and optionally queries) live in up to two segments that are never concatenated in memory:
//   segment 0 = context / latent-kv stream,  segment 1 = image / query stream.
// Visibility (exactly what the reference's mask encodes):
//   * a segment-0 key j is visible to every row iff j <= kvis[b]            (kvis == NULL: all visible)
//   * a segment-1 key is visible to segment-1 rows always, and to segment-0 rows iff seg0_sees_seg1
// Segment-0 rows beyond kvis[b] are dead in the reference (nobody can attend to them and the model only
// returns the image stream), so they are skipped here and their outputs are left untouched.
//
// Work decomposition: one workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32
// rows.  Per 32-key tile a wave computes S^T = K Q^T with 32 chained 32x32x2 MFMAs (the "swapped" product
// puts all scores of one query in one lane pair, so the online-softmax max/sum are per-lane plus one
// cross-half shuffle), exponentiates in registers, and feeds P^T straight back as the B operand of the
// O^T += V^T P^T MFMAs -- the accumulator register index IS the k index, no LDS round trip for P.
// K tiles are staged [key][68] (b128 reads, conflict-free), V tiles [key][64] (b32 reads, conflict-free).
#include "common.h"
#include "selftok_hip.h"
#include <stdlib.h>
#include <type_traits>

namespace selftok {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct AttnSeg {
    const float* q;   // may be NULL: segment contributes keys/values only
    const float* k;
    const float* v;
    float* o;
    _Float16* o_blk;  // f16x2 kernel: split-activation output (or NULL): the segment's [B * len, H * 64] matrix
    int len;          // rows in this segment
    long q_rs, k_rs, v_rs, o_rs;   // row strides (floats)
    long q_bs, k_bs, v_bs, o_bs;   // batch strides (floats)
};

struct AttnParams {
    AttnSeg seg[2];
    int B, H;
    const int* kvis;        // [B] or NULL
    int seg0_sees_seg1;
    float scale;
    int qtiles;             // 128-row query tiles per (sample, head), both segments
    int xcd_remap;
    int prio;               // attn64_dma_kernel: raise the wave's issue priority inside its MFMA clusters (s_setprio)
};

constexpr int KT = 32;           // keys per tile
constexpr int KSTR = 68;         // padded K row stride in LDS (floats)
constexpr int QROWS = 128;       // query rows per workgroup

__global__ __launch_bounds__(256) void attn64_kernel(AttnParams P)
{
    // K and V tiles, double buffered: tile t+1 is written while tile t is being consumed -> one barrier per tile
    __shared__ __attribute__((aligned(16))) float s_kb[2][KT * KSTR];
    __shared__ __attribute__((aligned(16))) float s_vb[2][KT * 64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    // XCD-aware work mapping: workgroup `orig` runs on XCD orig % 8 (observed dispatch order; speed only, never
    // correctness).  Give every XCD a contiguous range of work items so that the q-tiles of one (sample, head), which
    // re-read the same K/V, share one L2 instead of pulling K/V through the fabric once per XCD.
    int qt, h, b;
    {
        const int T = gridDim.x, orig = blockIdx.x;
        const int q8 = T >> 3, r8 = T & 7, xcd = orig & 7, idx = orig >> 3;
        const int w = P.xcd_remap ? (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx : orig;
        qt = w % P.qtiles;
        h = (w / P.qtiles) % P.H;
        b = w / (P.qtiles * P.H);
    }

    int n0 = P.seg[0].len;
    if (P.kvis) { int kv = P.kvis[b] + 1; n0 = kv < n0 ? (kv < 0 ? 0 : kv) : n0; }
    const int rows0 = P.seg[0].q ? n0 : 0;                  // live query rows of segment 0
    int s, r0;
    {
        const int t0 = P.seg[0].q ? (P.seg[0].len + QROWS - 1) / QROWS : 0;   // grid is sized on len, not on kvis
        if (qt < t0) { s = 0; r0 = qt * QROWS; }
        else { s = 1; r0 = (qt - t0) * QROWS; }
    }
    const int rows_live = (s == 0) ? rows0 : (P.seg[1].q ? P.seg[1].len : 0);
    if (r0 >= rows_live) return;                            // dead context rows / empty tile

    const AttnSeg& qs = P.seg[s];
    const int n1 = (s == 1 || P.seg0_sees_seg1) ? P.seg[1].len : 0;

    // ---- Q fragments: lane (half, col) holds Q[row][dd = m + 32*half], m = 0..31 ----
    const int my_row = r0 + wave * 32 + col;
    const bool row_ok = my_row < rows_live;
    float qf[32];
    {
        const float* qp = qs.q + (size_t)b * qs.q_bs + (size_t)(row_ok ? my_row : (rows_live - 1)) * qs.q_rs + h * 64 + 32 * half;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float4 t = *reinterpret_cast<const float4*>(qp + 4 * j);
            qf[4 * j] = t.x; qf[4 * j + 1] = t.y; qf[4 * j + 2] = t.z; qf[4 * j + 3] = t.w;
        }
    }

    f32x16 o0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 o1 = o0;
    const float c = P.scale * 1.4426950408889634f;   // scores are tracked in the log2 domain
    float m_run = -__builtin_inff(), l_run = 0.f;

    // staging map: thread -> 2 float4 of K and 2 of V per tile (coalesced: 16 threads cover one 256-B row)
    const int st_key = tid >> 4, st_part = tid & 15;   // + 16 keys for the second float4
    float4 rk[2], rv[2];

    // running per-thread source pointers (advanced by one tile per iteration; recomputed only at the segment switch):
    // keeps the 64-bit address arithmetic out of the loop -- every VALU op next to fp32 MFMAs costs matrix-pipe time
    const float* kp = nullptr;
    const float* vp = nullptr;
    long k_step = 0, v_step = 0, k_half = 0, v_half = 0;
    auto set_segment = [&](int seg) {
        const AttnSeg& ks = P.seg[seg];
        kp = ks.k + (size_t)b * ks.k_bs + (size_t)st_key * ks.k_rs + h * 64 + st_part * 4;
        vp = ks.v + (size_t)b * ks.v_bs + (size_t)st_key * ks.v_rs + h * 64 + st_part * 4;
        k_step = (long)KT * ks.k_rs; v_step = (long)KT * ks.v_rs;
        k_half = 16 * ks.k_rs; v_half = 16 * ks.v_rs;
    };
    auto issue_loads = [&](int key0, int nkeys) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int key = key0 + st_key + 16 * u;
            if (key < nkeys) {
                rk[u] = *reinterpret_cast<const float4*>(kp + u * k_half);
                rv[u] = *reinterpret_cast<const float4*>(vp + u * v_half);
            } else {
                rk[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                rv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        kp += k_step; vp += v_step;
    };

    // flattened tile list: segment 0 keys [0,n0) then segment 1 keys [0,n1)
    const int nt0 = (n0 + KT - 1) / KT, nt1 = (n1 + KT - 1) / KT;
    const int ntiles = nt0 + nt1;
    if (ntiles == 0) return;
    set_segment(nt0 > 0 ? 0 : 1);
    issue_loads(0, nt0 > 0 ? n0 : n1);
    auto write_tile = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int key = st_key + 16 * u;
            *reinterpret_cast<float4*>(&s_kb[buf][key * KSTR + st_part * 4]) = rk[u];
            *reinterpret_cast<float4*>(&s_vb[buf][key * 64 + st_part * 4]) = rv[u];
        }
    };
    write_tile(0);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int seg = t < nt0 ? 0 : 1;
        const int key0 = (seg == 0 ? t : t - nt0) * KT;
        const int nkeys = seg == 0 ? n0 : n1;
        const float* s_k = s_kb[t & 1];
        const float* s_v = s_vb[t & 1];
        if (t + 1 < ntiles) {                                // global loads of the next tile fly behind this tile's MFMAs
            const int seg_n = (t + 1) < nt0 ? 0 : 1;
            if (t + 1 == nt0) set_segment(1);                // first tile of segment 1
            issue_loads((seg_n == 0 ? t + 1 : t + 1 - nt0) * KT, seg_n == 0 ? n0 : n1);
        }

        // ---- S^T[key][q] = sum_dd K[key][dd] Q[q][dd] ----
        f32x16 sc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float4 kf = *reinterpret_cast<const float4*>(&s_k[col * KSTR + 32 * half + 4 * j]);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[4 * j + 0], sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[4 * j + 1], sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[4 * j + 2], sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[4 * j + 3], sc, 0, 0, 0);
        }
        // sc[r] = S[q = col][key = key0 + (r&3) + 8*(r>>2) + 4*half]
        if (key0 + KT > nkeys) {                             // ragged last tile: mask the padding keys
            // the empty volatile asm keeps this block a real (wave-uniform) branch: hipcc otherwise if-converts it into 16 x
            // (v_subrev, v_cmp, v_cndmask) executed on EVERY tile -- 48 VALU ops beside the fp32 MFMAs for one tile per segment
            asm volatile("; ragged tile");
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (key0 + (r & 3) + 8 * (r >> 2) + 4 * half >= nkeys) sc[r] = -__builtin_inff();
        }
        float mx;
        {   // 16 -> 1 with v_max3_f32 (7 ops instead of 15)
            float m0 = fmaxf(fmaxf(sc[0], sc[1]), sc[2]), m1 = fmaxf(fmaxf(sc[3], sc[4]), sc[5]);
            float m2 = fmaxf(fmaxf(sc[6], sc[7]), sc[8]), m3 = fmaxf(fmaxf(sc[9], sc[10]), sc[11]);
            float m4 = fmaxf(fmaxf(sc[12], sc[13]), sc[14]);
            mx = fmaxf(fmaxf(fmaxf(m0, m1), m2), fmaxf(fmaxf(m3, m4), sc[15]));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, WAVE));
        const float m_new = fmaxf(m_run, mx * c);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], c, -m_new));
            sc[r] = p;
            psum += p;
        }
        // the running maximum of a row stops moving after the first few tiles: when it did not move for ANY row of this
        // wave the rescale factor is exactly 1 and the 32 accumulator multiplies are skipped (exact, not a threshold)
        if (__any(m_new != m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            m_run = m_new;
        }
        l_run += psum;

        // ---- O^T[d][q] += sum_key V[key][d] P[q][key];  MFMA m carries keys (m&3)+8*(m>>2) (+4 for half 1) ----
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int key = (m & 3) + 8 * (m >> 2) + 4 * half;
            float v0 = s_v[key * 64 + col];
            float v1 = s_v[key * 64 + 32 + col];
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, sc[m], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, sc[m], o1, 0, 0, 0);
        }
        if (t + 1 < ntiles) write_tile((t + 1) & 1);        // that buffer was last read in iteration t-1 (barrier below)
        __syncthreads();
    }

    // ---- epilogue: O[q][d] = O^T / l ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, WAVE);
    const float inv = 1.0f / l_tot;
    if (row_ok) {
        float* op = qs.o + (size_t)b * qs.o_bs + (size_t)my_row * qs.o_rs + h * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = 8 * g + 4 * half;
            *reinterpret_cast<float4*>(op + d0) = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            *reinterpret_cast<float4*>(op + 32 + d0) = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        }
    }
}

// ---------------------------------------------------------------------------------------
// attn64_dma_kernel: attn64_kernel with the K / V tiles staged by LDS-DMA (round 3).
// The arithmetic is untouched -- the same chained v_mfma_f32_32x32x2_f32, the same online softmax, bit-identical outputs -- only
// the way a 32-key tile reaches LDS changes: `global_load_lds_dwordx4` (uniform SGPR base + one constant per-lane offset register,
// 16 B per lane, 1 KiB = 4 keys per instruction) instead of global_load -> VGPR -> ds_write_b128.  That removes 16 staging VGPRs,
// the 64-bit address arithmetic, the zero-fill selects and 4 ds_write_b128 per wave and tile from the VALU / LDS issue stream that
// shares the SIMD with the fp32 MFMAs, and brings the kernel from 140 to <= 128 VGPRs (3 -> 4 waves per SIMD).
// A DMA writes lane-contiguously, so the K rows cannot be padded (the [key][68] stride of attn64_kernel): the 16-byte chunks of
// a K row are XOR-swizzled instead -- chunk c of key k sits at position c ^ (k & 15) of its 256-byte row -- applied on the SOURCE
// side (lane (row, position) fetches chunk position ^ (row & 15)); the fragment reads `ds_read_b128` of a lane group then hit 16
// different positions = all 64 banks once.  V rows stay natural ([key][64], b32 reads of consecutive lanes).
// Ragged last tile of a segment: rows past the end are fetched from the segment's LAST key (clamped per-lane offsets, never past
// the tensor): their scores are masked to -inf, their probabilities are exactly 0, and 0 x (a finite V row) adds nothing.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void attn_lds_dma16(const void* base, unsigned voff, unsigned lds)
{
    // M0 = LDS byte address of the 1-KiB piece; declared as clobbered so that no compiler version keeps a value of its own in M0 across this
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory", "m0");
}

__global__ __launch_bounds__(256) void attn64_dma_kernel(AttnParams P)
{
    __shared__ __attribute__((aligned(1024))) float s_kb[2][KT * 64];     // swizzled rows, see above
    __shared__ __attribute__((aligned(1024))) float s_vb[2][KT * 64];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col = lane & 31;
    const bool prio = P.prio != 0;
    int qt, h, b;
    {
        const int T = gridDim.x, orig = blockIdx.x;
        const int q8 = T >> 3, r8 = T & 7, xcd = orig & 7, idx = orig >> 3;
        const int w = P.xcd_remap ? (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx : orig;
        qt = w % P.qtiles;
        h = (w / P.qtiles) % P.H;
        b = w / (P.qtiles * P.H);
    }
    int n0 = P.seg[0].len;
    if (P.kvis) { int kv = P.kvis[b] + 1; n0 = kv < n0 ? (kv < 0 ? 0 : kv) : n0; }
    const int rows0 = P.seg[0].q ? n0 : 0;
    int s, r0;
    {
        const int t0 = P.seg[0].q ? (P.seg[0].len + QROWS - 1) / QROWS : 0;
        if (qt < t0) { s = 0; r0 = qt * QROWS; }
        else { s = 1; r0 = (qt - t0) * QROWS; }
    }
    const int rows_live = (s == 0) ? rows0 : (P.seg[1].q ? P.seg[1].len : 0);
    if (r0 >= rows_live) return;
    const AttnSeg& qs = P.seg[s];
    const int n1 = (s == 1 || P.seg0_sees_seg1) ? P.seg[1].len : 0;

    const int my_row = r0 + wave * 32 + col;
    const bool row_ok = my_row < rows_live;
    float qf[32];
    {
        const float* qp = qs.q + (size_t)b * qs.q_bs + (size_t)(row_ok ? my_row : (rows_live - 1)) * qs.q_rs + h * 64 + 32 * half;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float4 t = *reinterpret_cast<const float4*>(qp + 4 * j);
            qf[4 * j] = t.x; qf[4 * j + 1] = t.y; qf[4 * j + 2] = t.z; qf[4 * j + 3] = t.w;
        }
    }
    f32x16 o0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 o1 = o0;
    const float c = P.scale * 1.4426950408889634f;
    float m_run = -__builtin_inff(), l_run = 0.f;

    // ---- staging: wave w issues K pieces w, w + 4 and V pieces w, w + 4 of a tile (piece i = keys 4 i .. 4 i + 3 = 1 KiB) ----
    const int kr = lane >> 4, pos = lane & 15;                 // row inside a piece, 16-byte position inside the row
    const int ksw = pos ^ ((4 * wave + kr) & 15);              // source chunk of that position (pieces w and w + 4: same rows mod 16)
    const unsigned lds_k = (unsigned)(size_t)(__attribute__((address_space(3))) float*)&s_kb[0][0];
    const unsigned lds_v = (unsigned)(size_t)(__attribute__((address_space(3))) float*)&s_vb[0][0];
    auto stage = [&](int seg, int key0, int nkeys, int buf) {
        const AttnSeg& ks = P.seg[seg];
        const char* kb = reinterpret_cast<const char*>(ks.k + (size_t)b * ks.k_bs + (size_t)key0 * ks.k_rs + h * 64);
        const char* vb = reinterpret_cast<const char*>(ks.v + (size_t)b * ks.v_bs + (size_t)key0 * ks.v_rs + h * 64);
        const unsigned krs = (unsigned)ks.k_rs * 4u, vrs = (unsigned)ks.v_rs * 4u;
        if (key0 + KT <= nkeys) {                              // full tile: uniform bases, constant per-lane offsets
            const unsigned ko = kr * krs + ksw * 16, vo = kr * vrs + pos * 16;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = wave + 4 * u;
                attn_lds_dma16(kb + (size_t)(4 * i) * krs, ko, lds_k + buf * (KT * 256) + i * 1024);
                attn_lds_dma16(vb + (size_t)(4 * i) * vrs, vo, lds_v + buf * (KT * 256) + i * 1024);
            }
        } else {                                               // ragged last tile of the segment: clamp the source row per lane
            const int last = nkeys - 1 - key0;                 // >= 0
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = wave + 4 * u;
                const int row = 4 * i + kr, src = row < last ? row : last;
                attn_lds_dma16(kb, (unsigned)src * krs + ksw * 16, lds_k + buf * (KT * 256) + i * 1024);
                attn_lds_dma16(vb, (unsigned)src * vrs + pos * 16, lds_v + buf * (KT * 256) + i * 1024);
            }
        }
    };
    // fragment-read addresses of the K tile (floats): row col, chunk (8 half + j) at position chunk ^ (col & 15)
    int kaddr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) kaddr[j] = col * 64 + (((8 * half + j) ^ (col & 15)) << 2);

    const int nt0 = (n0 + KT - 1) / KT, nt1 = (n1 + KT - 1) / KT;
    const int ntiles = nt0 + nt1;
    if (ntiles == 0) return;
    stage(nt0 > 0 ? 0 : 1, 0, nt0 > 0 ? n0 : n1, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // one 32-key tile; BUF (the LDS buffer it sits in) is a compile-time constant so that every fragment read is
    // (tile-invariant address register) + (immediate offset): with a run-time buffer index hipcc spends a v_or + v_add per read
    auto tile = [&](int t, auto BUF) {
        constexpr int buf = decltype(BUF)::value;
        const int seg = t < nt0 ? 0 : 1;
        const int key0 = (seg == 0 ? t : t - nt0) * KT;
        const int nkeys = seg == 0 ? n0 : n1;
        const float* s_k = s_kb[buf];
        const float* s_v = s_vb[buf];
        if (t + 1 < ntiles) {                                // the next tile's DMAs fly behind this tile's MFMAs
            const int seg_n = (t + 1) < nt0 ? 0 : 1;
            stage(seg_n, (seg_n == 0 ? t + 1 : t + 1 - nt0) * KT, seg_n == 0 ? n0 : n1, buf ^ 1);
        }

        f32x16 sc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (prio) __builtin_amdgcn_s_setprio(1);             // MFMA cluster: ahead of the co-resident waves' softmax VALU work
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float4 kf = *reinterpret_cast<const float4*>(&s_k[kaddr[j]]);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[4 * j + 0], sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[4 * j + 1], sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[4 * j + 2], sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[4 * j + 3], sc, 0, 0, 0);
        }
        if (prio) __builtin_amdgcn_s_setprio(0);
        if (key0 + KT > nkeys) {
            asm volatile("; ragged tile");                   // a real branch, not 48 if-converted VALU ops per tile (see attn64_kernel)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (key0 + (r & 3) + 8 * (r >> 2) + 4 * half >= nkeys) sc[r] = -__builtin_inff();
        }
        float mx;
        {
            float m0 = fmaxf(fmaxf(sc[0], sc[1]), sc[2]), m1 = fmaxf(fmaxf(sc[3], sc[4]), sc[5]);
            float m2 = fmaxf(fmaxf(sc[6], sc[7]), sc[8]), m3 = fmaxf(fmaxf(sc[9], sc[10]), sc[11]);
            float m4 = fmaxf(fmaxf(sc[12], sc[13]), sc[14]);
            mx = fmaxf(fmaxf(fmaxf(m0, m1), m2), fmaxf(fmaxf(m3, m4), sc[15]));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, WAVE));
        const float m_new = fmaxf(m_run, mx * c);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], c, -m_new));
            sc[r] = p;
            psum += p;
        }
        if (__any(m_new != m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            m_run = m_new;
        }
        l_run += psum;
        if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int key = (m & 3) + 8 * (m >> 2) + 4 * half;
            float v0 = s_v[key * 64 + col];
            float v1 = s_v[key * 64 + 32 + col];
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, sc[m], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, sc[m], o1, 0, 0, 0);
        }
        if (prio) __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my pieces of tile t + 1 have landed
        __syncthreads();                                     // everyone's have, and tile t's buffer is free
    };
    for (int t = 0; t < ntiles; t += 2) {
        tile(t, std::integral_constant<int, 0>{});
        if (t + 1 < ntiles) tile(t + 1, std::integral_constant<int, 1>{});
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, WAVE);
    const float inv = 1.0f / l_tot;
    if (row_ok) {
        float* op = qs.o + (size_t)b * qs.o_bs + (size_t)my_row * qs.o_rs + h * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = 8 * g + 4 * half;
            *reinterpret_cast<float4*>(op + d0) = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            *reinterpret_cast<float4*>(op + 32 + d0) = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        }
    }
}

// ---------------------------------------------------------------------------------------
// attn64_f16x2_kernel: the same attention with both contractions on the f16 matrix cores ("f16x2 split", see
// gemm_split.hip): every fp32 operand x is split into x0 = fp16(x) and x1 = fp16(x - x0) and a product keeps the three
// terms x0 y0 + x0 y1 + x1 y0 (exact fp16 products, fp32 accumulate).  S^T = K Q^T and O^T = V^T P^T then cost 12 + 12
// v_mfma_f32_32x32x16_f16 per 32-key tile (768 matrix-pipe cycles) instead of 64 fp32-input MFMAs (4096 cycles that also
// occupy the fp32 VALU lanes), and the softmax VALU work runs beside them on its own pipe.  The residual parts are NOT
// rescaled here (one accumulator per product): what a residual below the fp16 normal range loses is bounded in ABSOLUTE terms
// by the subnormal spacing 6e-8, against rows whose large entries are O(1), and probabilities live in [0, 1].
// |q|, |k|, |v| >= 65504 cannot be represented: the output turns non-finite, *overflow gets bit 2 and the caller recomputes
// with attn64_kernel.  Measured against fp64 softmax attention this kernel is slightly MORE accurate than the fp32-MFMA one
// (tests/test_kernels_gpu.py::test_attention_f16x2_accuracy_gate_and_range_flag).
//
// LDS images are MFMA-fragment ordered, 16 B (8 halfs) per lane and plane:
//   K  : [plane][d-group g = d/8 (8)][key (32)]           A operand of S^T = K Q^T   (rows = keys,  k = d)
//   V^T: [plane][key-group (4)][d' (64)]                  A operand of O^T = V^T P^T (rows = d',    k = keys)
// The key order inside a V^T entry is the order in which a lane holds its 16 probabilities (accumulator register r <->
// key (r&3) + 8 (r>>2) + 4 half), so P^T goes from the softmax registers straight into the B operand; d' = (d&3)*16 + d/4
// makes the transposing ds_write_b64 of the staging pass bank-conflict free and still lets the epilogue store float4s.
// ---------------------------------------------------------------------------------------
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int KG_STRIDE = 32 * 16 + 16;          // bytes between d-groups of the K image (padded: conflict-free b128 staging writes)
constexpr int K_PLANE = 8 * KG_STRIDE;           // 4224
constexpr int V_PLANE = 4 * 64 * 16;             // 4096
constexpr int KV_BUF = 2 * K_PLANE + 2 * V_PLANE;   // 16640 bytes per staged tile

// (a, b) -> packed fp16 pair hi = rne(a, b) and the packed residual lo = rne(a - hi.x, b - hi.y).  Plain C++ on purpose: an
// inline-asm version around v_fma_mix_f32 (4 ops per pair instead of 6) measured 12 % SLOWER -- every asm statement costs
// boundary s_nops and v_movs to gather its scalar outputs into the 128-bit MFMA operands.
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct HiLo { unsigned hi, lo; };
__device__ __forceinline__ HiLo split_pair(float a, float b)
{
    const f32x2 x = {a, b};
    const h16x2 h = __builtin_convertvector(x, h16x2);                                             // v_cvt_pk_f16_f32
    const h16x2 l = __builtin_convertvector(x - __builtin_convertvector(h, f32x2), h16x2);         // 2 cvt + v_pk_add + v_cvt_pk
    return HiLo{__builtin_bit_cast(unsigned, h), __builtin_bit_cast(unsigned, l)};
}
// the Linear kernels' form of the split (gemm_split.hip `split4`): the residual is carried scaled by 2^11
__device__ __forceinline__ HiLo split_pair_scaled(float a, float b)
{
    const f32x2 x = {a, b};
    const h16x2 h = __builtin_convertvector(x, h16x2);
    const h16x2 l = __builtin_convertvector((x - __builtin_convertvector(h, f32x2)) * 2048.0f, h16x2);
    return HiLo{__builtin_bit_cast(unsigned, h), __builtin_bit_cast(unsigned, l)};
}
__device__ __forceinline__ h16x8 as_h8(const u32x4& v) { return __builtin_bit_cast(h16x8, v); }
// tools/ builds only: what the staging pass would cost if K / V arrived already split (one conversion per pair stands in for the
// v_perm of a 16-bit transpose; results are hi-only, i.e. wrong in the low bits -- a timing probe, tools/bench_attn.py)
#if defined(SELFTOK_TUNE) && defined(SELFTOK_ATTN_PRESPLIT_PROBE)
__device__ __forceinline__ HiLo split_pair_kv(float a, float b)
{
    const f32x2 x = {a, b};
    return HiLo{__builtin_bit_cast(unsigned, __builtin_convertvector(x, h16x2)), 0u};
}
#else
__device__ __forceinline__ HiLo split_pair_kv(float a, float b) { return split_pair(a, b); }
#endif

__global__ __launch_bounds__(256, 2) void attn64_f16x2_kernel(AttnParams P, int* __restrict__ overflow)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_kv[2 * KV_BUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    int qt, h, b;
    {
        const int T = gridDim.x, orig = blockIdx.x;
        const int q8 = T >> 3, r8 = T & 7, xcd = orig & 7, idx = orig >> 3;
        const int w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
        qt = w % P.qtiles;
        h = (w / P.qtiles) % P.H;
        b = w / (P.qtiles * P.H);
    }
    int n0 = P.seg[0].len;
    if (P.kvis) { int kv = P.kvis[b] + 1; n0 = kv < n0 ? (kv < 0 ? 0 : kv) : n0; }
    const int rows0 = P.seg[0].q ? n0 : 0;
    int s, r0;
    {
        const int t0 = P.seg[0].q ? (P.seg[0].len + QROWS - 1) / QROWS : 0;
        if (qt < t0) { s = 0; r0 = qt * QROWS; }
        else { s = 1; r0 = (qt - t0) * QROWS; }
    }
    const int rows_live = (s == 0) ? rows0 : (P.seg[1].q ? P.seg[1].len : 0);
    if (r0 >= rows_live) return;
    const AttnSeg& qs = P.seg[s];
    const int n1 = (s == 1 || P.seg0_sees_seg1) ? P.seg[1].len : 0;

    // ---- Q fragments (B operand of S^T = K Q^T): lane (half, col) holds Q[row col][d = 16 ks + 8 half + j], pre-multiplied
    // by scale * log2(e) so that the scores come out of the matrix pipe in the log2 domain ----
    const float c = P.scale * 1.4426950408889634f;
    const int my_row = r0 + wave * 32 + col;
    const bool row_ok = my_row < rows_live;
    u32x4 q0[4], q1[4];
    {
        const float* qp = qs.q + (size_t)b * qs.q_bs + (size_t)(row_ok ? my_row : (rows_live - 1)) * qs.q_rs + h * 64 + 8 * half;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float4 a = *reinterpret_cast<const float4*>(qp + 16 * ks);
            const float4 c4 = *reinterpret_cast<const float4*>(qp + 16 * ks + 4);
            const float v[8] = {a.x * c, a.y * c, a.z * c, a.w * c, c4.x * c, c4.y * c, c4.z * c, c4.w * c};
#pragma unroll
            for (int j = 0; j < 4; ++j) { const HiLo t_ = split_pair(v[2 * j], v[2 * j + 1]); q0[ks][j] = t_.hi; q1[ks][j] = t_.lo; }
        }
    }

    f32x16 o0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 o1 = o0;
    float m_run = -__builtin_inff(), l_run = 0.f;

    // ---- staging: threads 0..127 own the K tile (key = t>>2, 16 d each), threads 128..255 the V tile (4 keys x 4 d each) ----
    const bool is_k = tid < 128;
    const int u = tid & 127;
    const int st_key = is_k ? (u >> 2) : 4 * (u >> 4);       // first key this thread loads
    const int st_d = is_k ? 16 * (u & 3) : 4 * (u & 15);     // first d
    // Keys past the end of a segment are staged as zeros (their scores are masked to -inf, their probabilities are exactly 0).
    // Measured: clamping the row index instead (no v_cndmask) costs 12 % -- the extra address registers and the branch hurt more.
    float4 rg[4];
    const float* sp = nullptr;         // running pointer: row key0 + st_key of the next tile to load
    long s_rs = 0;
    auto set_segment = [&](int seg) {
        const AttnSeg& ks = P.seg[seg];
        if (is_k) { sp = ks.k + (size_t)b * ks.k_bs + (size_t)st_key * ks.k_rs + h * 64 + st_d; s_rs = ks.k_rs; }
        else { sp = ks.v + (size_t)b * ks.v_bs + (size_t)st_key * ks.v_rs + h * 64 + st_d; s_rs = ks.v_rs; }
    };
    auto issue_loads = [&](int key0, int nkeys) {
        if (is_k) {
            const bool ok = key0 + st_key < nkeys;
#pragma unroll
            for (int i = 0; i < 4; ++i) rg[i] = ok ? *reinterpret_cast<const float4*>(sp + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rg[i] = (key0 + st_key + i < nkeys) ? *reinterpret_cast<const float4*>(sp + i * s_rs) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        sp += (long)KT * s_rs;
    };
    auto write_tile = [&](int buf) {
        unsigned char* base = s_kv + buf * KV_BUF;
        if (is_k) {   // 16 consecutive d of one key -> two 8-half entries per plane
            const float v[16] = {rg[0].x, rg[0].y, rg[0].z, rg[0].w, rg[1].x, rg[1].y, rg[1].z, rg[1].w,
                                 rg[2].x, rg[2].y, rg[2].z, rg[2].w, rg[3].x, rg[3].y, rg[3].z, rg[3].w};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                u32x4 hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) { const HiLo t_ = split_pair_kv(v[8 * e + 2 * j], v[8 * e + 2 * j + 1]); hi[j] = t_.hi; lo[j] = t_.lo; }
                const int g = (st_d >> 3) + e;
                *reinterpret_cast<u32x4*>(base + g * KG_STRIDE + st_key * 16) = hi;
                *reinterpret_cast<u32x4*>(base + K_PLANE + g * KG_STRIDE + st_key * 16) = lo;
            }
        } else {      // 4 keys x 4 d block, transposed: per d one run of 4 consecutive keys
            const float v[4][4] = {{rg[0].x, rg[0].y, rg[0].z, rg[0].w}, {rg[1].x, rg[1].y, rg[1].z, rg[1].w},
                                   {rg[2].x, rg[2].y, rg[2].z, rg[2].w}, {rg[3].x, rg[3].y, rg[3].z, rg[3].w}};
            // key run k0..k0+3 (k0 = st_key, multiple of 4) sits in key-group (k0>>4)*2 + ((k0>>2)&1), half-entry (k0>>3)&1
            const int kg = ((st_key >> 4) << 1) + ((st_key >> 2) & 1), hb = (st_key >> 3) & 1;
            unsigned char* vb = base + 2 * K_PLANE + kg * (64 * 16) + hb * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {              // d = st_d + i  ->  d' = (d&3)*16 + d/4 = i*16 + st_d/4
                u32x2 hi, lo;
                { const HiLo t_ = split_pair_kv(v[0][i], v[1][i]); hi[0] = t_.hi; lo[0] = t_.lo; }
                { const HiLo t_ = split_pair_kv(v[2][i], v[3][i]); hi[1] = t_.hi; lo[1] = t_.lo; }
                const int dp = i * 16 + (st_d >> 2);
                *reinterpret_cast<u32x2*>(vb + dp * 16) = hi;
                *reinterpret_cast<u32x2*>(vb + V_PLANE + dp * 16) = lo;
            }
        }
    };

    const int nt0 = (n0 + KT - 1) / KT, nt1 = (n1 + KT - 1) / KT;
    const int ntiles = nt0 + nt1;
    if (ntiles == 0) return;
    set_segment(nt0 > 0 ? 0 : 1);
    issue_loads(0, nt0 > 0 ? n0 : n1);
    write_tile(0);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int seg = t < nt0 ? 0 : 1;
        const int key0 = (seg == 0 ? t : t - nt0) * KT;
        const int nkeys = seg == 0 ? n0 : n1;
        const unsigned char* sb = s_kv + (t & 1) * KV_BUF;
        if (t + 1 < ntiles) {
            const int seg_n = (t + 1) < nt0 ? 0 : 1;
            if (t + 1 == nt0) set_segment(1);
            issue_loads((seg_n == 0 ? t + 1 : t + 1 - nt0) * KT, seg_n == 0 ? n0 : n1);
        }

        // ---- S^T[key][q] = sum_d K[key][d] Q'[q][d]: 4 k-steps of 16 d, three f16 MFMAs each ----
        f32x16 sc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const h16x8 k0 = *reinterpret_cast<const h16x8*>(sb + (2 * ks + half) * KG_STRIDE + col * 16);
            const h16x8 k1 = *reinterpret_cast<const h16x8*>(sb + K_PLANE + (2 * ks + half) * KG_STRIDE + col * 16);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, as_h8(q0[ks]), sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, as_h8(q1[ks]), sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, as_h8(q0[ks]), sc, 0, 0, 0);
        }
        // sc[r] = log2(e) * scale * S[q = col][key = key0 + (r&3) + 8*(r>>2) + 4*half]
        if (key0 + KT > nkeys) {
            asm volatile("; ragged tile");                   // a real branch, not 48 if-converted VALU ops per tile (see attn64_kernel)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (key0 + (r & 3) + 8 * (r >> 2) + 4 * half >= nkeys) sc[r] = -__builtin_inff();
        }
        float mx;
        {
            float m0 = fmaxf(fmaxf(sc[0], sc[1]), sc[2]), m1 = fmaxf(fmaxf(sc[3], sc[4]), sc[5]);
            float m2 = fmaxf(fmaxf(sc[6], sc[7]), sc[8]), m3 = fmaxf(fmaxf(sc[9], sc[10]), sc[11]);
            float m4 = fmaxf(fmaxf(sc[12], sc[13]), sc[14]);
            mx = fmaxf(fmaxf(fmaxf(m0, m1), m2), fmaxf(fmaxf(m3, m4), sc[15]));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, WAVE));
        const float m_new = fmaxf(m_run, mx);
        const f32x2 mm = {m_new, m_new};
        f32x2 psum = {0.f, 0.f};
        u32x4 p0[2], p1[2];                                     // P^T fragments: k-step ks2 holds registers 8 ks2 .. 8 ks2 + 7
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const f32x2 a = f32x2{sc[r], sc[r + 1]} - mm;        // v_pk_add_f32
            const f32x2 p = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
            psum += p;                                           // v_pk_add_f32
            { const HiLo t_ = split_pair(p[0], p[1]); p0[r >> 3][(r & 7) >> 1] = t_.hi; p1[r >> 3][(r & 7) >> 1] = t_.lo; }
        }
        if (__any(m_new != m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            m_run = m_new;
        }
        l_run += psum[0] + psum[1];

        // ---- O^T[d'][q] += sum_key V[key][d'] P[q][key]: 2 k-steps of 16 keys x 2 blocks of 32 d' ----
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            const unsigned char* vb = sb + 2 * K_PLANE + (2 * ks2 + half) * (64 * 16) + col * 16;
            const h16x8 va0 = *reinterpret_cast<const h16x8*>(vb), va1 = *reinterpret_cast<const h16x8*>(vb + V_PLANE);
            const h16x8 vb0 = *reinterpret_cast<const h16x8*>(vb + 32 * 16), vb1 = *reinterpret_cast<const h16x8*>(vb + V_PLANE + 32 * 16);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(va0, as_h8(p0[ks2]), o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vb0, as_h8(p0[ks2]), o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(va0, as_h8(p1[ks2]), o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vb0, as_h8(p1[ks2]), o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(va1, as_h8(p0[ks2]), o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vb1, as_h8(p0[ks2]), o1, 0, 0, 0);
        }
        if (t + 1 < ntiles) write_tile((t + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: o_db[r] = O[q = col][d' = 32 db + (r&3) + 8 (r>>2) + 4 half],  d = 4 (d' & 15) + (d' >> 4) ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, WAVE);
    const float inv = 1.0f / l_tot;
    // an operand beyond the fp16 range became inf in its high part: it shows up as a non-finite output (0 * inf = NaN below).
    // Non-finite INPUTS also land here; the caller's fp32 recomputation then reproduces their NaN/inf honestly.
    float chk = l_tot * 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { chk = __builtin_fmaf(o0[r], 0.f, chk); chk = __builtin_fmaf(o1[r], 0.f, chk); }
    if (overflow && row_ok && chk != 0.f) atomicOr(overflow, 4);
    if (row_ok) {
        const size_t off = (size_t)b * qs.o_bs + (size_t)my_row * qs.o_rs + h * 64;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                // registers a + 4 bb (+8): d' = a + 8 bb + 4 half (+16)  ->  d = 4 (a + 8 bb + 4 half) + {0, 1} + 2 db
                const int r = a + 4 * bb;
                const int d = 4 * (a + 8 * bb + 4 * half);
                if (qs.o_blk) {      // split activation for the proj Linear (gemm_split.hip); |o| <= max |v| < 65504 here
                    const HiLo ab = split_pair_scaled(opaque_f32(o0[r] * inv), opaque_f32(o0[r + 8] * inv));
                    const HiLo cd = split_pair_scaled(opaque_f32(o1[r] * inv), opaque_f32(o1[r + 8] * inv));
                    const long row_g = (long)b * qs.len + my_row;
                    *reinterpret_cast<uint2*>(qs.o_blk + split_blk_index(row_g, h * 64 + d, 0, P.H * 2)) = make_uint2(ab.hi, cd.hi);
                    *reinterpret_cast<uint2*>(qs.o_blk + split_blk_index(row_g, h * 64 + d, 1, P.H * 2)) = make_uint2(ab.lo, cd.lo);
                } else {
                    *reinterpret_cast<float4*>(qs.o + off + d) = make_float4(o0[r] * inv, o0[r + 8] * inv, o1[r] * inv, o1[r + 8] * inv);
                }
            }
    }
}

// ---------------------------------------------------------------------------------------
// head_dim 16 self-attention of the encoder's latent stream (modules.py:235-238): 4 heads x 256 tokens,
// 4 MFLOP per (sample, head) -- far too small for matrix cores to matter.  One workgroup per
// (batch, head): K and V of the head sit in LDS, every thread owns one query row and runs the online
// softmax over broadcast LDS reads.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn16_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                     float* __restrict__ o, int L, long q_rs, long k_rs, long v_rs, long o_rs,
                                                     long q_bs, long k_bs, long v_bs, long o_bs, float scale)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];   // K [L][16] then V [L][16]
    float* s_k = smem;
    float* s_v = smem + (size_t)L * 16;
    const int b = blockIdx.y, h = blockIdx.x;
    for (int i = threadIdx.x; i < L * 4; i += blockDim.x) {
        int key = i >> 2, part = i & 3;
        *reinterpret_cast<float4*>(&s_k[key * 16 + part * 4]) = *reinterpret_cast<const float4*>(k + (size_t)b * k_bs + (size_t)key * k_rs + h * 16 + part * 4);
        *reinterpret_cast<float4*>(&s_v[key * 16 + part * 4]) = *reinterpret_cast<const float4*>(v + (size_t)b * v_bs + (size_t)key * v_rs + h * 16 + part * 4);
    }
    __syncthreads();
    const float c = scale * 1.4426950408889634f;
    for (int row = threadIdx.x; row < L; row += blockDim.x) {
        float qq[16], acc[16];
        const float* qp = q + (size_t)b * q_bs + (size_t)row * q_rs + h * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 t = *reinterpret_cast<const float4*>(qp + 4 * j);
            qq[4 * j] = t.x; qq[4 * j + 1] = t.y; qq[4 * j + 2] = t.z; qq[4 * j + 3] = t.w;
        }
#pragma unroll
        for (int d = 0; d < 16; ++d) acc[d] = 0.f;
        float m_run = -__builtin_inff(), l_run = 0.f;
        for (int key = 0; key < L; ++key) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 16; ++d) s = __builtin_fmaf(qq[d], s_k[key * 16 + d], s);
            float sl = s * c;
            float m_new = fmaxf(m_run, sl);
            float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float p = __builtin_amdgcn_exp2f(sl - m_new);
            l_run = l_run * alpha + p;
#pragma unroll
            for (int d = 0; d < 16; ++d) acc[d] = __builtin_fmaf(p, s_v[key * 16 + d], acc[d] * alpha);
            m_run = m_new;
        }
        const float inv = 1.0f / l_run;
        float* op = o + (size_t)b * o_bs + (size_t)row * o_rs + h * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<float4*>(op + 4 * j) = make_float4(acc[4 * j] * inv, acc[4 * j + 1] * inv, acc[4 * j + 2] * inv, acc[4 * j + 3] * inv);
    }
}

}  // namespace selftok

using namespace selftok;

extern "C" {

int selftok_attn_f32(const selftok_attn_desc* d, hipStream_t stream)
{
    if (!d || d->B < 0 || d->H <= 0) { set_last_error("attn: bad descriptor"); return SELFTOK_EINVAL; }
    if (d->B == 0) return SELFTOK_OK;
    if (d->head_dim == 64) {
        AttnParams P;
        for (int s = 0; s < 2; ++s) {
            const selftok_attn_seg& a = d->seg[s];
            const bool osplit = d->mode == SELFTOK_ATTN_F16X2 && d->o_blk[s] != nullptr;
            if (d->o_blk[s] && (d->mode != SELFTOK_ATTN_F16X2 || ((size_t)d->o_blk[s] & 15))) { set_last_error("attn: split outputs need the f16x2 mode and 16-byte alignment"); return SELFTOK_EINVAL; }
            if (a.len < 0 || (a.len > 0 && (!a.k || !a.v)) || (a.q && !a.o && !osplit)) { set_last_error("attn: bad segment"); return SELFTOK_EINVAL; }
            if (((a.q_rs | a.k_rs | a.v_rs | a.o_rs | a.q_bs | a.k_bs | a.v_bs | a.o_bs) & 3) != 0) { set_last_error("attn: strides must be multiples of 4 floats"); return SELFTOK_EINVAL; }
            P.seg[s] = AttnSeg{a.len > 0 ? a.q : nullptr, a.k, a.v, a.o, (_Float16*)d->o_blk[s], a.len, a.q_rs, a.k_rs, a.v_rs, a.o_rs, a.q_bs, a.k_bs, a.v_bs, a.o_bs};
        }
        P.B = d->B; P.H = d->H; P.kvis = d->kvis; P.seg0_sees_seg1 = d->seg0_sees_seg1; P.scale = d->scale;
        int t0 = P.seg[0].q ? (P.seg[0].len + QROWS - 1) / QROWS : 0;
        int t1 = P.seg[1].q ? (P.seg[1].len + QROWS - 1) / QROWS : 0;
        if (t0 + t1 == 0) return SELFTOK_OK;
        P.qtiles = t0 + t1;
        P.xcd_remap = 1;
        P.prio = 0;
#ifdef SELFTOK_TUNE
        { const char* e = getenv("SELFTOK_ATTN_PRIO"); if (e) P.prio = atoi(e); }
#endif
        if (d->mode == SELFTOK_ATTN_F16X2) {
            hipLaunchKernelGGL(attn64_f16x2_kernel, dim3((t0 + t1) * d->H * d->B), dim3(256), 0, stream, P, d->overflow);
            return check_launch("attn64_f16x2_kernel");
        }
        if (d->mode != 0) { set_last_error("attn: unknown mode"); return SELFTOK_EINVAL; }
        // the LDS-DMA staged kernel needs 16-byte aligned K / V rows (strides are multiples of 4 floats by the check above)
        bool dma = true;
        for (int s = 0; s < 2; ++s)
            if (P.seg[s].len > 0 && ((((size_t)P.seg[s].k | (size_t)P.seg[s].v) & 15) != 0 || P.seg[s].k_rs >= (1l << 24) || P.seg[s].v_rs >= (1l << 24))) dma = false;   // 32-bit per-lane byte offsets
#ifdef SELFTOK_TUNE
        { const char* e = getenv("SELFTOK_ATTN_VARIANT"); if (e && atoi(e) == 0) dma = false; }      // 0: register-staged kernel of rounds 1-2
#endif
        if (dma) {
            hipLaunchKernelGGL(attn64_dma_kernel, dim3((t0 + t1) * d->H * d->B), dim3(256), 0, stream, P);
            return check_launch("attn64_dma_kernel");
        }
        hipLaunchKernelGGL(attn64_kernel, dim3((t0 + t1) * d->H * d->B), dim3(256), 0, stream, P);
        return check_launch("attn64_kernel");
    }
    if (d->head_dim == 16) {
        // single-segment, unmasked self-attention (segment 1 only)
        const selftok_attn_seg& a = d->seg[1];
        if (d->seg[0].len != 0 || d->kvis || !a.q || !a.k || !a.v || !a.o || a.len <= 0 || a.len > 1024) { set_last_error("attn(head_dim 16): single unmasked segment of <= 1024 rows only"); return SELFTOK_EINVAL; }
        size_t lds = (size_t)a.len * 16 * 2 * sizeof(float);
        hipLaunchKernelGGL(attn16_kernel, dim3(d->H, d->B), dim3(256), lds, stream, a.q, a.k, a.v, a.o, a.len,
                           a.q_rs, a.k_rs, a.v_rs, a.o_rs, a.q_bs, a.k_bs, a.v_bs, a.o_bs, d->scale);
        return check_launch("attn16_kernel");
    }
    set_last_error("attn: head_dim must be 64 or 16");
    return SELFTOK_EINVAL;
}

}  // extern "C"
