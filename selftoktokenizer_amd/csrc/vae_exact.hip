// The bf16 SD3-VAE ENCODER with the reference's exact summation orders (round 4).  gfx950 only.
//
// The reference runs `self.vae.encode(images)[0].mode()` (mimogpt/infer/SelftokPipeline.py:215; topology of the mirror
// mimogpt/models/selftok/sd3/sd3_impls.py:221-377) in bf16 on the CPU, and its token ids depend on the exact bf16 rounding of every
// layer.  csrc/conv.hip reproduces the PRECISION of that arithmetic (fp32 accumulation, one rounding) on the bf16 matrix cores and
// lands within 1 bf16 ulp of it -- 15 of 8192 tokens then flip at reference near-ties.  This file reproduces the ORDER: every
// reduction is evaluated as the same sequence of fp32 operations torch-CPU executes (oracle/vae_exact.c documents how each order was
// established and is the bit-for-bit CPU twin of every kernel here), so the latents -- and with them the token ids from pixels --
// are the reference's bit for bit.
//
//   xconv_kernel   convolution / GEMM in oneDNN's AMX order.  One TDPBF16PS takes 32 input channels: even elements are summed
//                  sequentially in one fp32 accumulator, odd elements in another, chunk = even + odd, C += chunk.  Here a chunk is 16
//                  v_mfma_f32_32x32x1_2b_f32: a ONE-k-step MFMA is an fma per output (the products of two bf16 values are exact, so
//                  fma = round(acc + product), what the AMX unit does), block 0 of the instruction carries the even chain and block 1
//                  the odd chain -- lanes 0..31 feed the low halves of the 16 packed bf16 pairs, lanes 32..63 the high halves --
//                  and the first instruction of a chunk starts from the inline constant 0.  The fold C += (even + odd) is 16 + 16
//                  v_add_f32 per 32 x 32 tile.  fp32 MFMA rate (157 TF/s): 16x slower than conv.hip's bf16 MFMAs, which is the
//                  price of a prescribed order (a bf16 MFMA sums its 16 k-steps in an order of its own).
//                  Staging: the workgroup (4 waves: 64 pixels x 128 channels) copies the chunk's 64 bytes of each of its 192 rows to
//                  LDS with coalesced 16-byte loads (one full row per 4 threads), double buffered, one barrier per chunk, the next
//                  chunk in flight during this chunk's 32 MFMAs per wave; lanes read their row's four 16-byte pieces back with
//                  ds_read_b128 (XOR-swizzled: rows are 16 banks apart).  A first version had every lane read its own row from
//                  global memory: 8x the cache-line accesses, 0.595 of the fp32 matrix peak (profiles/r4_vae_exact_bench_first.txt).
//                  Epilogues: bf16(C + bias) [+ residual with its own rounding]; fp32 C * scale (attention scores);
//                  a per-row rescale of C at a chunk boundary and bf16(C * rowscale) (the P V product of the flash kernel).
//   xconv_in       conv_in: 3 input channels = ONE chunk of 27 elements in (kw, kh, ic) order; fp32 VALU FMAs, weights in LDS.
//   xgn_*          ATen's GroupNorm: Welford over 16-element vectors in 8 fp32 lanes, chunks of 16 vectors, binary cascade,
//                  scalar lane combination, rstd through fp64, y = bf16(fma(scale, x, bias)), then SiLU by table.
//   xsilu_table    torch's bf16 SiLU is a function of the input alone: correctly rounded x / (1 + exp(-x)), except that
//                  fp32 exp(-x) overflows for x < -88.72 and the quotient becomes -0.
//   xattn_softmax  the row pass of ATen's cpu_flash_attention between the two GEMMs (kv blocks of 512, fexp_u20, 16-lane sums,
//                  glibc expf for the rescale).
#include "common.h"
#include "selftok_hip.h"

namespace selftok {

typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short us8v __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float xbf2f(unsigned short u) { return __uint_as_float(((uint32_t)u) << 16); }
__device__ __forceinline__ unsigned short xf2bf(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (unsigned short)0x7fc0;
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// convolution / GEMM in AMX chunk order
// ---------------------------------------------------------------------------------------------------------------------------------
struct XConvArgs {
    const unsigned short* x;      // [B][H][W][IC] bf16                      (+ z * x_bs)
    const unsigned short* w;      // [OC][KH*KW][IC] bf16                    (+ z * w_bs)
    const unsigned short* bias;   // [OC] bf16 or null
    const unsigned short* res;    // [P][OC] bf16 or null                    (+ z * y_bs)
    void* y;                      // [P][OC] bf16 (mode 0, 2) / fp32 (mode 1) (+ z * y_bs)
    const float* rescale;         // [nblk][rs_stride] or null: C *= rescale[j][p] before chunk j * split, j >= 1   (+ z * v_bs)
    const float* rowscale;        // [P]: mode 2, y = bf16(C * rowscale[p])              (+ z * v_bs)
    float out_scale;              // mode 1: y = C * out_scale
    int H, W, IC, OC, KH, KW, stride, pad, OH, OW;
    int split, mode;
    int up;                       // 1: the convolution reads a nearest-2x upsampled view of x ([B][H/2][W/2][IC] in memory; H, W = the upsampled size)
    long x_bs, w_bs, y_bs, v_bs;
    long P;                       // rows (output pixels) per z: the last workgroup's rows beyond P are neither read nor written
    long rs_stride;
};

typedef float xf32x2 __attribute__((ext_vector_type(2)));

// ORD: 0 = chunks in (kh, kw, channel-block) order; 3 = channel-block major, every block's taps summed privately (S) and then added to the total;
// 1 = channel-block major into the ONE running total (what oneDNN does for the two decoder layers whose activation reaches 2^31 bytes: 64 images)
template <int NT, int WM, int WN, int ORD>
__global__ __launch_bounds__(64 * WM * WN) void xconv_kernel(XConvArgs a)
{
    constexpr bool PARTIAL = ORD == 3, ICB_MAJOR = ORD != 0;
    // workgroup tile: RA = 32 WM pixel rows x RB = 32 NT WN output channels; one chunk = 32 bf16 channels of every row, staged as fp32:
    // round 5 -- the bf16 -> fp32 expansion happens ONCE per staged element (a shift / a mask on the packed pair, by the thread that copies it to
    // LDS) instead of once per reading lane (v_perm per MFMA operand: 24 per 32 x 32 tile and chunk), and a row's chunk is laid out as
    // [16 even elements | 16 odd elements] so that a lane reads the 16 operands of ITS chain (block 0: even, block 1: odd) as four ds_read_b128.
    // fp32-input MFMAs and VALU instructions share the issue slot (tools/microbench/mfma_valu.hip: 0 / 4 / 9 VALU per MFMA -> 151 / 125 / 93 TF),
    // so the kernel's rate is set by the VALU instructions beside its MFMAs: 3.5 per MFMA before, 1.75 now (with the packed fold below).
    constexpr int RA = 32 * WM, RB = 32 * NT * WN, NTHR = 64 * WM * WN;
    constexpr int NPA = (RA * 4 + NTHR - 1) / NTHR, NPB = (RB * 4 + NTHR - 1) / NTHR;          // 16-byte (8 x bf16) global pieces per thread
    __shared__ u32x4 lds[2][(RA + RB) * 8];                                        // 128 bytes per row: 8 pieces of 4 fp32
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int i = lane & 31, h = lane >> 5;
    const int z = blockIdx.z;
    const int H = a.H, W = a.W, IC = a.IC, OC = a.OC, KH = a.KH, KW = a.KW;
    const int KT = KH * KW, nicb = IC >> 5;
    const long pwg = (long)blockIdx.x * RA;
    const int nwg = blockIdx.y * RB;
    const long p0 = pwg + wm * 32;
    const int n0 = nwg + wn * (32 * NT);
    const unsigned short* x = a.x + (size_t)z * a.x_bs;
    const unsigned short* w = a.w + (size_t)z * a.w_bs;

    // what this thread stages: global piece (row, q) = elements 8 q .. 8 q + 7 of the row's chunk -> even elements 4 q .. 4 q + 3 of the even half
    // (LDS piece q) and of the odd half (LDS piece 4 + q); piece p of row r lives at lds[r * 8 + (p ^ ((r >> 1) & 7))]: the XOR spreads the 16 rows of a
    // ds_read_b128 phase over all 64 banks (rows are 128 bytes = 32 banks apart)
    const unsigned short* arow[NPA];
    int aiy0[NPA], aix0[NPA], aslot[NPA], asw[NPA];
    bool aon[NPA];
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
        const int idx = tid + NTHR * j, row = idx >> 2, q = idx & 3;
        aon[j] = idx < RA * 4 && pwg + row < a.P;
        const long pa = pwg + (aon[j] ? row : 0);
        const int ohw = a.OH * a.OW;
        const int b = (int)(pa / ohw);
        const int rem = (int)(pa - (long)b * ohw);
        const int oy = rem / a.OW, ox = rem - oy * a.OW;
        aiy0[j] = oy * a.stride - a.pad; aix0[j] = ox * a.stride - a.pad;
        arow[j] = x + (size_t)b * (H >> a.up) * (W >> a.up) * IC + q * 8;
        asw[j] = (row >> 1) & 7;
        aslot[j] = row * 8 + q;
    }
    const unsigned short* brow[NPB];
    int bslot[NPB], bsw[NPB];
    bool bon[NPB];
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
        const int idx = tid + NTHR * j, row = idx >> 2, q = idx & 3;
        bon[j] = idx < RB * 4;
        brow[j] = w + (size_t)(nwg + (bon[j] ? row : 0)) * KT * IC + q * 8;
        bsw[j] = ((RA + row) >> 1) & 7;
        bslot[j] = (RA + row) * 8 + q;
    }

    float C[NT][16], S[PARTIAL ? NT : 1][16];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { C[t][r] = 0.f; if (PARTIAL) S[t][r] = 0.f; }

    int kh = 0, kw = 0, icb = 0;
    u32x4 sa[NPA], sb[NPB];
    // the input pixel of a staged row moves only when the TAP (kh, kw) does: its element offset and bounds flag are kept per piece and recomputed
    // on a tap change -- every chunk in channel-block-major order, every IC / 32 chunks otherwise (15 VALU instructions per piece and chunk
    // less beside the fp32 MFMAs, which share their issue slot)
    int aoff[NPA];
    bool aok[NPA];
    auto retap = [&]() {
#pragma unroll
        for (int j = 0; j < NPA; ++j) {
            const int iy = aiy0[j] + kh, ix = aix0[j] + kw;
            aok[j] = aon[j] && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            aoff[j] = aok[j] ? (((iy >> a.up) * (W >> a.up) + (ix >> a.up)) * IC) : 0;
        }
    };
    retap();
    auto fetch = [&]() {                 // global -> registers for the chunk (kh, kw, icb), then step to the next chunk
#pragma unroll
        for (int j = 0; j < NPA; ++j) {
            const u32x4 zero = {0u, 0u, 0u, 0u};
            const u32x4 v = *reinterpret_cast<const u32x4*>(arow[j] + (size_t)(unsigned)aoff[j] + icb * 32);
            sa[j] = aok[j] ? v : zero;
        }
        const size_t woff = (size_t)(kh * KW + kw) * IC + icb * 32;
#pragma unroll
        for (int j = 0; j < NPB; ++j) sb[j] = *reinterpret_cast<const u32x4*>(brow[j] + woff);
        if (!ICB_MAJOR) { if (++icb == nicb) { icb = 0; if (++kw == KW) { kw = 0; ++kh; } retap(); } }
        else { if (++kw == KW) { kw = 0; if (++kh == KH) { kh = 0; ++icb; } } retap(); }
    };
    auto expand = [&](const u32x4& v, u32x4& ev, u32x4& od) {      // 4 packed bf16 pairs -> their low halves and their high halves as fp32 bit patterns
#pragma unroll
        for (int e = 0; e < 4; ++e) { ev[e] = v[e] << 16; od[e] = v[e] & 0xFFFF0000u; }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NPA; ++j) if (aon[j]) {
            u32x4 ev, od; expand(sa[j], ev, od);
            const int base = aslot[j] & ~7, q = aslot[j] & 7;
            lds[buf][base + (q ^ asw[j])] = ev;
            lds[buf][base + ((4 + q) ^ asw[j])] = od;
        }
#pragma unroll
        for (int j = 0; j < NPB; ++j) if (bon[j]) {
            u32x4 ev, od; expand(sb[j], ev, od);
            const int base = bslot[j] & ~7, q = bslot[j] & 7;
            lds[buf][base + (q ^ bsw[j])] = ev;
            lds[buf][base + ((4 + q) ^ bsw[j])] = od;
        }
    };
    const int ra = wm * 32 + i, swa = (ra >> 1) & 7;
    int rb[NT], swb[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { rb[t] = RA + wn * (32 * NT) + t * 32 + i; swb[t] = (rb[t] >> 1) & 7; }
    auto compute = [&](int buf) {
        f32x32 acc[NT];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32x4 va = lds[buf][ra * 8 + ((4 * h + q) ^ swa)];          // 4 consecutive operands of this lane's chain (h = 0: even, 1: odd)
            u32x4 vb[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) vb[t] = lds[buf][rb[t] * 8 + ((4 * h + q) ^ swb[t])];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float fa = __uint_as_float(va[e]);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float fb = __uint_as_float(vb[t][e]);
                    if (q == 0 && e == 0) { const f32x32 zero = {0}; acc[t] = __builtin_amdgcn_mfma_f32_32x32x1f32(fa, fb, zero, 0, 0, 0); }
                    else acc[t] = __builtin_amdgcn_mfma_f32_32x32x1f32(fa, fb, acc[t], 0, 0, 0);
                }
            }
        }
        // chunk = even chain + odd chain, C += chunk: two values per v_pk_add_f32 (each component rounds like v_add_f32)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const xf32x2 ev = {acc[t][r], acc[t][r + 1]}, od = {acc[t][16 + r], acc[t][17 + r]};
                const xf32x2 c = ev + od;
                if (PARTIAL) { xf32x2 s2 = {S[t][r], S[t][r + 1]}; s2 = s2 + c; S[t][r] = s2[0]; S[t][r + 1] = s2[1]; }
                else { xf32x2 c2 = {C[t][r], C[t][r + 1]}; c2 = c2 + c; C[t][r] = c2[0]; C[t][r + 1] = c2[1]; }
            }
    };

    const int nchunks = KT * nicb;
    fetch();
    stage(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        if (more) fetch();                                       // the next chunk's 12 KB are in flight during this chunk's MFMAs
        if (a.rescale != nullptr && c > 0 && c % a.split == 0) {  // the flash kernel's `dst *= exp(old max - new max)` when a new kv block starts
            // rows beyond P of a ragged last tile read past their image's entries: the next image's, the next block's or the row-scale array's
            // (the workspace carries 128 floats of slack behind it) -- their products are never stored.  No clamp: a loop-invariant min() per
            // accumulator row is hoisted out of the chunk loop into 30 live registers (152 -> 182 VGPRs, one wave per SIMD less).
            const float* rs = a.rescale + (size_t)(c / a.split) * a.rs_stride + (size_t)z * a.v_bs + p0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float f = rs[(r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
                for (int t = 0; t < NT; ++t) C[t][r] = C[t][r] * f;
            }
        }
        compute(c & 1);
        if (PARTIAL && (c + 1) % KT == 0) {                      // order 3: an ic-block's private sum joins the total when its taps are done
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) { C[t][r] = C[t][r] + S[t][r]; S[t][r] = 0.f; }
        }
        if (more) stage((c + 1) & 1);
        __syncthreads();
    }

    // epilogue: lane (col = i, rows (r & 3) + 8 (r >> 2) + 4 h)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int oc = n0 + t * 32 + i;
        const float bias = (a.mode == 0 && a.bias != nullptr) ? xbf2f(a.bias[oc]) : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long p = p0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (p >= a.P) continue;
            const size_t off = (size_t)z * a.y_bs + (size_t)p * OC + oc;
            if (a.mode == 1) {
                reinterpret_cast<float*>(a.y)[off] = C[t][r] * a.out_scale;
            } else if (a.mode == 2) {
                reinterpret_cast<unsigned short*>(a.y)[off] = xf2bf(C[t][r] * a.rowscale[(size_t)z * a.v_bs + p]);
            } else {
                unsigned short hb = xf2bf(C[t][r] + bias);
                if (a.res != nullptr) hb = xf2bf(xbf2f(a.res[off]) + xbf2f(hb));
                reinterpret_cast<unsigned short*>(a.y)[off] = hb;
            }
        }
    }
}

// conv_in: x [B][H][W][ldx] bf16 (channels 0..2 used), w [OC][3][3][3] bf16 ([oc][kh][kw][ic]), y [B][H][W][OC] bf16; OC == 128.
// One thread per output pixel; the 27 products in (kw, kh, ic) order: even positions in one chain, odd in the other.
__global__ __launch_bounds__(256) void xconv_in_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
                                                       const unsigned short* __restrict__ bias, unsigned short* __restrict__ y, int B, int H, int W, int ldx, int OC)
{
    extern __shared__ float wl[];            // [27][OC] in chain order, then bias [OC]
    for (int e = threadIdx.x; e < 27 * OC; e += blockDim.x) {
        const int pos = e / OC, oc = e - pos * OC;
        const int kw = pos / 9, kh = (pos / 3) % 3, ic = pos % 3;
        wl[e] = xbf2f(w[((size_t)oc * 9 + kh * 3 + kw) * 3 + ic]);
    }
    for (int e = threadIdx.x; e < OC; e += blockDim.x) wl[27 * OC + e] = xbf2f(bias[e]);
    __syncthreads();
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long)B * H * W) return;
    const int b = (int)(p / ((long)H * W));
    const int rem = (int)(p - (long)b * H * W);
    const int oy = rem / W, ox = rem - oy * W;
    float xv[27];
#pragma unroll
    for (int pos = 0; pos < 27; ++pos) {
        const int kw = pos / 9, kh = (pos / 3) % 3, ic = pos % 3;
        const int iy = oy - 1 + kh, ix = ox - 1 + kw;
        const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        xv[pos] = ok ? xbf2f(x[((size_t)(b * H + iy) * W + ix) * ldx + ic]) : 0.f;
    }
    unsigned short* yp = y + (size_t)p * OC;
    for (int o0 = 0; o0 < OC; o0 += 8) {
        us8v out;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int oc = o0 + e;
            float te = 0.f, to = 0.f;
#pragma unroll
            for (int pos = 0; pos < 27; ++pos) {
                const float wv = wl[pos * OC + oc];
                if (pos & 1) to = fmaf(xv[pos], wv, to); else te = fmaf(xv[pos], wv, te);      // exact product: fma = add
            }
            float c = 0.f + (te + to);
            out[e] = xf2bf(c + wl[27 * OC + oc]);
        }
        *reinterpret_cast<us8v*>(yp + o0) = out;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// GroupNorm: ATen RowwiseMomentsImpl<BFloat16> (AVX2 build), restated
// ---------------------------------------------------------------------------------------------------------------------------------
struct Mom { float m1, m2; };

// AddMomentsVec (one lane): (m0_add, add) joins (m0, acc)
__device__ __forceinline__ void add_moments_vec(int m0_add, Mom add, int& m0, Mom& acc)
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

// Pass 1.  Workgroup = 128 threads = (lane l = tid / 16 of the 8 fp32 lanes, cq = tid % 16 -> channels cblk*128 + cq*8 .. +7) of one
// image and one range of RP consecutive pixels.  A "chunk" is 256 consecutive pixels of a channel = 16 vectors of 16 bf16; lane l of
// a vector's first half is pixel 16 j + l, of its second half pixel 16 j + 8 + l.  Every thread runs ATen's chunk loop + binary cascade
// over the RP / 256 chunks of its range for its 8 channels; the range's node (level log2(RP / 256)) goes to the workspace.
// node layout: [B][C][R][8 lanes] of Mom.
template <int NCH>       // chunks per range: 16 (RP = 4096), 4 (RP = 1024) or 2 (RP = 512)
__global__ __launch_bounds__(128) void xgn_partial_kernel(const unsigned short* __restrict__ x, Mom* __restrict__ nodes, int HW, int C, int R)
{
    constexpr int LV = NCH == 16 ? 5 : (NCH == 4 ? 3 : 2);          // levels 0 .. log2(NCH)
    const int l = threadIdx.x >> 4, cq = threadIdx.x & 15;
    const int r = blockIdx.x, cblk = blockIdx.y, b = blockIdx.z;
    const int c0 = cblk * 128 + cq * 8;
    const unsigned short* xp = x + ((size_t)b * HW + (size_t)r * NCH * 256) * C + c0;
    Mom stk[LV][8];
    int m0s[LV];
#pragma unroll
    for (int v = 0; v < LV; ++v) { m0s[v] = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) stk[v][e] = Mom{0.f, 0.f}; }
#pragma unroll 1
    for (int u = 0; u < NCH; ++u) {
        Mom a[8], bb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] = Mom{0.f, 0.f}; bb[e] = Mom{0.f, 0.f}; }
#pragma unroll 4
        for (int j = 0; j < 16; ++j) {
            const float cj = 1.0f / (float)(j + 1);
            const us8v va = *reinterpret_cast<const us8v*>(xp + (size_t)(u * 256 + j * 16 + l) * C);
            const us8v vb = *reinterpret_cast<const us8v*>(xp + (size_t)(u * 256 + j * 16 + 8 + l) * C);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x0 = xbf2f(va[e]), x1 = xbf2f(vb[e]);
                const float d0 = x0 - a[e].m1, d1 = x1 - bb[e].m1;
                a[e].m1 = fmaf(d0, cj, a[e].m1); bb[e].m1 = fmaf(d1, cj, bb[e].m1);
                const float e0 = x0 - a[e].m1, e1 = x1 - bb[e].m1;
                a[e].m2 = fmaf(d0, e0, a[e].m2); bb[e].m2 = fmaf(d1, e1, bb[e].m2);
            }
        }
        {
            int m0 = m0s[0];
#pragma unroll
            for (int e = 0; e < 8; ++e) { int t0 = m0s[0]; add_moments_vec(16, a[e], t0, stk[0][e]); add_moments_vec(16, bb[e], t0, stk[0][e]); m0 = t0; }
            m0s[0] = m0;
        }
        int mask = u + 1;
#pragma unroll
        for (int j = 1; j < LV; ++j) {
            if ((mask & 1) != 0) break;
            int m0 = m0s[j];
#pragma unroll
            for (int e = 0; e < 8; ++e) { int t0 = m0s[j]; add_moments_vec(m0s[j - 1], stk[j - 1][e], t0, stk[j][e]); stk[j - 1][e] = Mom{0.f, 0.f}; m0 = t0; }
            m0s[j] = m0; m0s[j - 1] = 0;
            mask >>= 1;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) nodes[(((size_t)b * C + c0 + e) * R + r) * 8 + l] = stk[LV - 1][e];
}

// Pass 2.  One thread = one fp32 lane of one (image, group): continues the cascade over the group's D * R nodes (channel-major), then
// thread 0 of the 8 combines the lanes with the scalar AddMoments (GCC's FMA contractions), rstd through fp64, and writes per-channel
// scale = rstd * gamma, bias = fma(-scale, mean, beta).  node_count = elements per lane in one node.
__global__ void xgn_finish_kernel(const Mom* __restrict__ nodes, const unsigned short* __restrict__ gamma, const unsigned short* __restrict__ beta,
                                  float* __restrict__ scale, float* __restrict__ bias, float* __restrict__ stats, int BG, int C, int G, int R, int HW,
                                  int node_count, double eps)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int grp = t >> 3, l = t & 7;
    if (grp >= BG) return;
    const int b = grp / G, g = grp % G, D = C / G;
    const int Q = D * R;
    int depth = 0;
    while ((1 << depth) < Q) ++depth;
    constexpr int MAXD = 10;                    // Q <= 512 nodes; every stack index below is a compile-time constant (registers, no scratch)
    Mom stk[MAXD];
    int m0s[MAXD];
#pragma unroll
    for (int v = 0; v < MAXD; ++v) { stk[v] = Mom{0.f, 0.f}; m0s[v] = 0; }
    for (int q = 0; q < Q; ++q) {
        const int d = q / R, r = q - d * R;
        const Mom nd = nodes[(((size_t)b * C + g * D + d) * R + r) * 8 + l];
        add_moments_vec(node_count, nd, m0s[0], stk[0]);
        int mask = q + 1;
        bool go = true;
#pragma unroll
        for (int j = 1; j < MAXD; ++j) {
            go = go && j < depth && (mask & 1) == 0;
            if (go) {
                add_moments_vec(m0s[j - 1], stk[j - 1], m0s[j], stk[j]);
                m0s[j - 1] = 0; stk[j - 1] = Mom{0.f, 0.f};
                mask >>= 1;
            }
        }
    }
#pragma unroll
    for (int j = 1; j < MAXD; ++j)
        if (j < depth) add_moments_vec(m0s[j], stk[j], m0s[0], stk[0]);
    // lane combination on lane 0 of the 8 (the values of lanes 1..7 through shuffles)
    float m1 = 0.f, m2 = 0.f;
    int m0 = 0;
    const int m0_add = m0s[0];
    for (int k = 0; k < 8; ++k) {
        const float a1 = __shfl(stk[0].m1, (threadIdx.x & ~7) + k, WAVE), a2 = __shfl(stk[0].m2, (threadIdx.x & ~7) + k, WAVE);
        const int n = m0 + m0_add;
        const float c = n == 0 ? 0.f : (float)m0_add / (float)n;
        const float delta = a1 - m1;
        m1 = fmaf(c, delta, m1);
        m2 = m2 + fmaf(delta * delta * c, (float)m0, a2);
        m0 = n;
    }
    if (l != 0) return;
    const float N = (float)((long)D * HW);
    const float var = m2 / N;
    const float rstd = (float)(1.0 / sqrt((double)fmaxf(var, 0.f) + eps));
    if (stats != nullptr) { stats[2 * grp] = m1; stats[2 * grp + 1] = rstd; }
    for (int d = 0; d < D; ++d) {
        const int c = g * D + d;
        const float sc = rstd * xbf2f(gamma[c]);
        scale[(size_t)b * C + c] = sc;
        bias[(size_t)b * C + c] = fmaf(-sc, m1, xbf2f(beta[c]));
    }
}

// Shapes whose channels hold an ODD number of chunks (H*W = 6400, 256: the 80 x 80 and 16 x 16 maps of a 320 / 128 px image) or whose chunks
// straddle channel boundaries (H*W % 256 != 0: 40 x 40) have no aligned pair of chunks to pre-combine: level 0 of ATen's cascade takes the two
// halves (a, b) of chunk i and then of chunk i + 1 one after the other, so pass 1 stores every chunk's two half-moments RAW and pass 2 replays
// the whole loop.  raw layout: [B][G][m chunks of the group][2 halves][8 lanes] of Mom, m = ceil(D * HW / 256).
// Pass 1r, H*W % 256 == 0: the chunk grid is the same for every channel -> the vectorised body of xgn_partial_kernel, one chunk per thread.
__global__ __launch_bounds__(128) void xgn_chunk_kernel(const unsigned short* __restrict__ x, Mom* __restrict__ raw, int HW, int C, int NC)
{
    const int l = threadIdx.x >> 4, cq = threadIdx.x & 15;
    const int ic = blockIdx.x, cblk = blockIdx.y, b = blockIdx.z;
    const int c0 = cblk * 128 + cq * 8;
    const unsigned short* xp = x + ((size_t)b * HW + (size_t)ic * 256) * C + c0;
    Mom a[8], bb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = Mom{0.f, 0.f}; bb[e] = Mom{0.f, 0.f}; }
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
        const float cj = 1.0f / (float)(j + 1);
        const us8v va = *reinterpret_cast<const us8v*>(xp + (size_t)(j * 16 + l) * C);
        const us8v vb = *reinterpret_cast<const us8v*>(xp + (size_t)(j * 16 + 8 + l) * C);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x0 = xbf2f(va[e]), x1 = xbf2f(vb[e]);
            const float d0 = x0 - a[e].m1, d1 = x1 - bb[e].m1;
            a[e].m1 = fmaf(d0, cj, a[e].m1); bb[e].m1 = fmaf(d1, cj, bb[e].m1);
            const float e0 = x0 - a[e].m1, e1 = x1 - bb[e].m1;
            a[e].m2 = fmaf(d0, e0, a[e].m2); bb[e].m2 = fmaf(d1, e1, bb[e].m2);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        Mom* o = raw + (((size_t)b * C + c0 + e) * NC + ic) * 16;          // (b C + g D + d) NC + ic = (b G + g) m + (d NC + ic)
        o[l] = a[e]; o[8 + l] = bb[e];
    }
}

// Pass 1r, any H*W % 16 == 0: element e of the group's NCHW sequence is channel e / HW, pixel e % HW.  One thread = (chunk, half, lane).
__global__ __launch_bounds__(256) void xgn_gather_kernel(const unsigned short* __restrict__ x, Mom* __restrict__ raw, int HW, int C, int G, int m, long nvec, long total)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int l = (int)(t & 7), half = (int)((t >> 3) & 1);
    const long ch = t >> 4;                                   // (b G + g) m + chunk
    const int gi = (int)(ch % m);
    const long bg = ch / m;
    const int g = (int)(bg % G), D = C / G;
    const long b = bg / G;
    const long left = nvec - 16L * gi;
    const int m0 = left < 16 ? (int)left : 16;
    const unsigned short* xp = x + (size_t)b * HW * C + g * D;
    Mom a = Mom{0.f, 0.f};
    for (int j = 0; j < m0; ++j) {
        const long e = 256L * gi + 16 * j + 8 * half + l;
        const int d = (int)(e / HW), px = (int)(e - (long)d * HW);
        const float cj = 1.0f / (float)(j + 1);
        const float x0 = xbf2f(xp[(size_t)px * C + d]);
        const float d0 = x0 - a.m1;
        a.m1 = fmaf(d0, cj, a.m1);
        a.m2 = fmaf(d0, x0 - a.m1, a.m2);
    }
    raw[t] = a;
}

// Pass 2r.  One thread = one fp32 lane of one (image, group): ATen's chunk loop on the stored half-moments, then as xgn_finish_kernel.
__global__ void xgn_finish_raw_kernel(const Mom* __restrict__ raw, const unsigned short* __restrict__ gamma, const unsigned short* __restrict__ beta,
                                      float* __restrict__ scale, float* __restrict__ bias, float* __restrict__ stats, int BG, int C, int G, int m, long nvec, int HW, double eps)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int grp = t >> 3, l = t & 7;
    if (grp >= BG) return;
    const int b = grp / G, g = grp % G, D = C / G;
    int depth = 0;
    while ((1 << depth) < m) ++depth;
    constexpr int MAXD = 10;                    // m <= 512 chunks
    Mom stk[MAXD];
    int m0s[MAXD];
#pragma unroll
    for (int v = 0; v < MAXD; ++v) { stk[v] = Mom{0.f, 0.f}; m0s[v] = 0; }
    const Mom* rp = raw + (size_t)grp * m * 16 + l;
    for (int q = 0; q < m; ++q) {
        const long left = nvec - 16L * q;
        const int m0 = left < 16 ? (int)left : 16;
        add_moments_vec(m0, rp[(size_t)q * 16], m0s[0], stk[0]);
        add_moments_vec(m0, rp[(size_t)q * 16 + 8], m0s[0], stk[0]);
        int mask = q + 1;
        bool go = true;
#pragma unroll
        for (int j = 1; j < MAXD; ++j) {
            go = go && j < depth && (mask & 1) == 0;
            if (go) {
                add_moments_vec(m0s[j - 1], stk[j - 1], m0s[j], stk[j]);
                m0s[j - 1] = 0; stk[j - 1] = Mom{0.f, 0.f};
                mask >>= 1;
            }
        }
    }
#pragma unroll
    for (int j = 1; j < MAXD; ++j)
        if (j < depth) add_moments_vec(m0s[j], stk[j], m0s[0], stk[0]);
    float m1 = 0.f, m2 = 0.f;
    int m0 = 0;
    const int m0_add = m0s[0];
    for (int k = 0; k < 8; ++k) {
        const float a1 = __shfl(stk[0].m1, (threadIdx.x & ~7) + k, WAVE), a2 = __shfl(stk[0].m2, (threadIdx.x & ~7) + k, WAVE);
        const int n = m0 + m0_add;
        const float c = n == 0 ? 0.f : (float)m0_add / (float)n;
        const float delta = a1 - m1;
        m1 = fmaf(c, delta, m1);
        m2 = m2 + fmaf(delta * delta * c, (float)m0, a2);
        m0 = n;
    }
    if (l != 0) return;
    const float N = (float)((long)D * HW);
    const float var = m2 / N;
    const float rstd = (float)(1.0 / sqrt((double)fmaxf(var, 0.f) + eps));
    if (stats != nullptr) { stats[2 * grp] = m1; stats[2 * grp + 1] = rstd; }
    for (int d = 0; d < D; ++d) {
        const int c = g * D + d;
        const float sc = rstd * xbf2f(gamma[c]);
        scale[(size_t)b * C + c] = sc;
        bias[(size_t)b * C + c] = fmaf(-sc, m1, xbf2f(beta[c]));
    }
}

// Pass 3.  y = table[bf16(fma(scale, x, bias))] (table == null: no activation), 8 channels per thread.
__global__ __launch_bounds__(256) void xgn_apply_kernel(const unsigned short* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ bias,
                                                        const unsigned short* __restrict__ table, unsigned short* __restrict__ y, long n8, int C, long HWC)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n8) return;
    const long e0 = idx * 8;
    const int b = (int)(e0 / HWC), c0 = (int)(e0 % C);
    const us8v v = *reinterpret_cast<const us8v*>(x + e0);
    const float4 s0 = *reinterpret_cast<const float4*>(scale + (size_t)b * C + c0), s1 = *reinterpret_cast<const float4*>(scale + (size_t)b * C + c0 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(bias + (size_t)b * C + c0), b1 = *reinterpret_cast<const float4*>(bias + (size_t)b * C + c0 + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, bi[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    us8v out;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const unsigned short hb = xf2bf(fmaf(sc[e], xbf2f(v[e]), bi[e]));
        out[e] = table != nullptr ? table[hb] : hb;
    }
    *reinterpret_cast<us8v*>(y + e0) = out;
}

// torch-CPU's SiLU on every bf16 bit pattern: x / (1 + exp(-x)) evaluated in fp32 with an exp that is exact to the last bf16 bit
// everywhere -- except that exp(-x) overflows fp32 for -x > 88.7228 and the quotient becomes a signed zero.
__global__ void xsilu_table_kernel(unsigned short* __restrict__ table)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 65536) return;
    const float x = xbf2f((unsigned short)i);
    unsigned short out;
    if (x != x) out = (unsigned short)(i | 0x40);
    else if (-x > 88.72284f) out = xf2bf(x / __builtin_inff());        // -0 for finite x, NaN for -inf (inf / inf)
    else {
        // fp64 quotient -> fp32 -> bf16: the two roundings torch's own fp32 evaluation + bf16 store amount to (checked against
        // torch-CPU on all 65536 inputs: tests/golden/silu_bf16_table.npy)
        const double xd = (double)x;
        out = xf2bf((float)(xd / (1.0 + exp(-xd))));
    }
    table[i] = out;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// attention row pass
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float xfexp_u20(float x)       // Vectorized<float>::fexp_u20 (ATen/cpu/vec/vec512/vec512_float.h)
{
    const float c0 = 0.00010703434948458272f, c1 = 0.30354260500649682f, c2 = -0.22433836478672356f, c3 = -0.079204240219773236f;
    const float log2e = __uint_as_float(0x3fb8aa3bu), a = 8388608.0f, b = 8388608.0f * 127.f;
    float src = x * log2e;
    const float fr = src - floorf(src);
    float res = fmaf(fr, c3, c2);
    res = fmaf(fr, res, c1);
    res = fmaf(fr, res, c0);
    src = src - res;
    const float tmp = fmaf(a, src, b);
    int ci = (int)tmp;                     // truncation, as cvttps2dq
    if (x < __uint_as_float(0xc2aeac50u)) ci = 0;
    if (x > __uint_as_float(0x42b17218u)) ci = 0x7F800000;
    return __int_as_float(ci);
}

__constant__ unsigned long long XEXP2F_T[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

// glibc's expf (sysdeps/ieee754/flt-32/e_expf.c, 32-entry table, fp64 arithmetic): what `std::exp(float)` evaluates in the flash kernel
__device__ __forceinline__ float xexpf_glibc(float x)
{
    if (x != x) return x;
    if (x > 0x1.62e42ep6f) return __builtin_inff();
    if (x < -0x1.9fe368p6f) return 0.f;
    const double InvLn2N = 0x1.71547652b82fep+0 * 32, Shift = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
    const double z = InvLn2N * (double)x;
    double kd = z + Shift;
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= Shift;
    const double r = z - kd;
    const unsigned long long t = XEXP2F_T[ki % 32] + (ki << 47);
    const double s = __longlong_as_double((long long)t);
    const double zz = fma(C0, r, C1), r2 = r * r;
    double y = fma(C2, r, 1.0);
    y = fma(zz, r2, y);
    return (float)(y * s);
}

__global__ void xexpf_kernel(const float* __restrict__ x, float* __restrict__ y, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = xexpf_glibc(x[i]);
}

// s [B*T][T] fp32 scaled scores -> p [B*T][T] bf16 un-normalised probabilities (every kv block of 512 keys relative to the running maximum
// after that block), rescale [nblk][B*T] = expf(old max - new max) at each block (block 0: 0), rowscale [B*T] = 1 / sum.  T % 16 == 0, T <= 512 nblk.
// 16 lanes per row: lane = key mod 16 sums its <= 32 probabilities of a block sequentially, then the 8 / 4 / 2 / 1 fold of vec_reduce_all;
// sum = fma(exp_tmp, old sum, block sum) (ATen cpu_flash_attention, kvSplitSize 512; oracle/vae_exact.c vx_attention).
__global__ __launch_bounds__(256) void xattn_softmax_kernel(const float* __restrict__ s, unsigned short* __restrict__ p, float* __restrict__ rescale,
                                                            float* __restrict__ rowscale, long rows, int T)
{
    const int l = threadIdx.x & 15;
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= rows) return;
    float m_old = -__builtin_inff(), sum_old = 0.f;
    for (int n0 = 0, blk = 0; n0 < T; n0 += 512, ++blk) {
        const int cnt = (T - n0 < 512 ? T - n0 : 512) >> 4;
        const float* sr = s + (size_t)row * T + n0;
        unsigned short* pr = p + (size_t)row * T + n0;
        float v[32];
        float bm = -__builtin_inff();
#pragma unroll
        for (int k = 0; k < 32; ++k) { v[k] = k < cnt ? sr[16 * k + l] : -__builtin_inff(); bm = fmaxf(bm, v[k]); }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) bm = fmaxf(bm, __shfl_xor(bm, o, WAVE));
        const float m_new = fmaxf(m_old, bm);
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) if (k < cnt) { const float e = xfexp_u20(v[k] - m_new); acc += e; pr[16 * k + l] = xf2bf(e); }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc = acc + __shfl_xor(acc, o, WAVE);
        const float exp_tmp = xexpf_glibc(m_old - m_new);          // block 0: expf(-inf) = 0
        sum_old = fmaf(exp_tmp, sum_old, acc);
        m_old = m_new;
        if (l == 0) rescale[(size_t)blk * rows + row] = exp_tmp;
    }
    if (l == 0) rowscale[row] = 1.0f / sum_old;
}

// v [B][T][C] -> vt [B][C][T] (bf16): the P V product reads V as [output channel][key]
__global__ void xtranspose_kernel(const unsigned short* __restrict__ v, unsigned short* __restrict__ vt, int T, int C)
{
    __shared__ unsigned short tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) tile[r][tx] = v[((size_t)b * T + t0 + r) * C + c0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8) vt[((size_t)b * C + c0 + r) * T + t0 + tx] = tile[tx][r];
}

static int launch_xconv(XConvArgs a, long P, int nz, int order, hipStream_t stream)
{
    a.P = P;
#ifndef XCONV_WM
#define XCONV_WM 2
#endif
    if (a.OC % 128 == 0) {
        dim3 grid((unsigned)((P + 32 * XCONV_WM - 1) / (32 * XCONV_WM)), a.OC / 128, nz);
        if (order == 3) hipLaunchKernelGGL((xconv_kernel<2, XCONV_WM, 2, 3>), grid, dim3(128 * XCONV_WM), 0, stream, a);
        else if (order == 1) hipLaunchKernelGGL((xconv_kernel<2, XCONV_WM, 2, 1>), grid, dim3(128 * XCONV_WM), 0, stream, a);
        else hipLaunchKernelGGL((xconv_kernel<2, XCONV_WM, 2, 0>), grid, dim3(128 * XCONV_WM), 0, stream, a);
    } else if (a.OC % 64 == 0 && order == 0) {                 // the scores of a 1600-token attention: 1600 keys = 25 x 64
        dim3 grid((unsigned)((P + 63) / 64), a.OC / 64, nz);
        hipLaunchKernelGGL((xconv_kernel<1, 2, 2, 0>), grid, dim3(256), 0, stream, a);
    } else {
        dim3 grid((unsigned)((P + 127) / 128), a.OC / 32, nz);
        if (order == 3) hipLaunchKernelGGL((xconv_kernel<1, 4, 1, 3>), grid, dim3(256), 0, stream, a);
        else if (order == 1) hipLaunchKernelGGL((xconv_kernel<1, 4, 1, 1>), grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((xconv_kernel<1, 4, 1, 0>), grid, dim3(256), 0, stream, a);
    }
    return check_launch("xconv_kernel");
}

}  // namespace selftok

using namespace selftok;

extern "C" {

int selftok_vx_conv2d_bf16(const void* x, const void* w, const void* bias, const void* residual, void* out, int B, int H, int W, int ldx, int Cin, int Cout,
                           int ksize, int stride, int order, hipStream_t stream)
{
    if (B == 0) return SELFTOK_OK;                       // empty batch: nothing to do (the pointers of empty tensors may be null)
    if (!x || !w || !bias || !out || B < 0 || (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2) || (stride == 2 && ksize != 3)) {
        set_last_error("vx_conv2d: bad argument"); return SELFTOK_EINVAL;
    }
    const int OH = stride == 2 ? H / 2 : H, OW = stride == 2 ? W / 2 : W;
    const long P = (long)B * OH * OW;
    if (order == 2) {
        if (Cin != 3 || ksize != 3 || stride != 1 || Cout % 8 || residual) { set_last_error("vx_conv2d: order 2 is conv_in (3 channels, 3x3)"); return SELFTOK_EINVAL; }
        hipLaunchKernelGGL(xconv_in_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), (size_t)(28 * Cout) * sizeof(float), stream, (const unsigned short*)x,
                           (const unsigned short*)w, (const unsigned short*)bias, (unsigned short*)out, B, H, W, ldx, Cout);
        return check_launch("xconv_in_kernel");
    }
    const int up = (order & SELFTOK_VX_UPSAMPLE2X) ? 1 : 0;
    order &= ~SELFTOK_VX_UPSAMPLE2X;
    if (up && (stride != 1 || ((H | W) & 1))) { set_last_error("vx_conv2d: SELFTOK_VX_UPSAMPLE2X needs stride 1 and even H, W (the upsampled size)"); return SELFTOK_EINVAL; }
    if ((order != 0 && order != 1 && order != 3) || Cin % 32 || ldx != Cin || Cout % 32 || (stride == 2 && ((H | W) & 1))) {
        set_last_error("vx_conv2d: need Cin % 32 == 0, Cout % 32 == 0, order 0 / 1 / 2 / 3"); return SELFTOK_EINVAL;
    }
    XConvArgs a{};
    a.x = (const unsigned short*)x; a.w = (const unsigned short*)w; a.bias = (const unsigned short*)bias; a.res = (const unsigned short*)residual; a.y = out;
    a.H = H; a.W = W; a.IC = Cin; a.OC = Cout; a.KH = a.KW = ksize; a.stride = stride; a.pad = (ksize == 3 && stride == 1) ? 1 : 0; a.OH = OH; a.OW = OW;
    a.split = -1; a.mode = 0; a.up = up;
    return launch_xconv(a, P, 1, order, stream);
}

// pass-1 plan of the exact GroupNorm: chunks of 256 elements per channel (nc); NCH = 16 / 4 / 2 aligned chunks pre-combined per thread when nc is
// a multiple (H*W % 512 == 0), else 0 = the raw route
static int xgn_plan(int HW) { if (HW % 256) return 0; const int nc = HW / 256; return nc % 16 == 0 ? 16 : (nc % 4 == 0 ? 4 : (nc % 2 == 0 ? 2 : 0)); }

size_t selftok_vx_groupnorm_workspace_bytes(int B, int HW, int C)
{
    if (B <= 0 || HW <= 0 || HW % 16 || C % 128) return 0;
    const int nch = xgn_plan(HW);
    const size_t moms = nch ? (size_t)B * C * (HW / (256 * nch)) * 8 : ((size_t)B * C * HW / 256 + (size_t)B * C) * 16;        // raw: m <= D HW / 256 + 1 chunks per group
    return moms * sizeof(Mom) + (size_t)2 * B * C * sizeof(float);
}

int selftok_vx_groupnorm_bf16(const void* x, const void* gamma, const void* beta, void* out, void* workspace, const void* silu_table, float* stats, int B, int HW,
                              int C, int groups, double eps, hipStream_t stream)
{
    if (B == 0) return SELFTOK_OK;
    if (!x || !gamma || !beta || !out || !workspace || B < 0 || groups <= 0 || C % groups || C % 128 || HW <= 0 || HW % 16 || (C / groups) & ((C / groups) - 1)) {
        set_last_error("vx_groupnorm: need C % 128 == 0, H*W % 16 == 0, power-of-two channels per group"); return SELFTOK_EINVAL;
    }
    const int nch = xgn_plan(HW), D = C / groups, BG = B * groups;
    const size_t moms = nch ? (size_t)B * C * (HW / (256 * nch)) * 8 : ((size_t)B * C * HW / 256 + (size_t)B * C) * 16;
    Mom* nodes = (Mom*)workspace;
    float* scale = (float*)((char*)workspace + moms * sizeof(Mom));
    float* bias = scale + (size_t)B * C;
    int rc;
    if (nch) {
        const int R = HW / (256 * nch);
        if ((long)D * R > 512) { set_last_error("vx_groupnorm: more than 512 nodes per group"); return SELFTOK_EINVAL; }
        dim3 grid(R, C / 128, B);
        if (nch == 16) hipLaunchKernelGGL((xgn_partial_kernel<16>), grid, dim3(128), 0, stream, (const unsigned short*)x, nodes, HW, C, R);
        else if (nch == 4) hipLaunchKernelGGL((xgn_partial_kernel<4>), grid, dim3(128), 0, stream, (const unsigned short*)x, nodes, HW, C, R);
        else hipLaunchKernelGGL((xgn_partial_kernel<2>), grid, dim3(128), 0, stream, (const unsigned short*)x, nodes, HW, C, R);
        rc = check_launch("xgn_partial_kernel");
        if (rc) return rc;
        hipLaunchKernelGGL(xgn_finish_kernel, dim3((BG * 8 + 63) / 64), dim3(64), 0, stream, nodes, (const unsigned short*)gamma, (const unsigned short*)beta, scale, bias, stats,
                           BG, C, groups, R, HW, 32 * nch, eps);
        rc = check_launch("xgn_finish_kernel");
    } else {
        const long nvec = (long)D * HW / 16;
        const int m = (int)((nvec + 15) / 16);
        if (m > 512) { set_last_error("vx_groupnorm: more than 512 chunks per group on the raw route"); return SELFTOK_EINVAL; }
        if (HW % 256 == 0) {
            hipLaunchKernelGGL(xgn_chunk_kernel, dim3(HW / 256, C / 128, B), dim3(128), 0, stream, (const unsigned short*)x, nodes, HW, C, HW / 256);
            rc = check_launch("xgn_chunk_kernel");
        } else {
            const long total = (long)BG * m * 16;
            hipLaunchKernelGGL(xgn_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const unsigned short*)x, nodes, HW, C, groups, m, nvec, total);
            rc = check_launch("xgn_gather_kernel");
        }
        if (rc) return rc;
        hipLaunchKernelGGL(xgn_finish_raw_kernel, dim3((BG * 8 + 63) / 64), dim3(64), 0, stream, nodes, (const unsigned short*)gamma, (const unsigned short*)beta, scale, bias, stats,
                           BG, C, groups, m, nvec, HW, eps);
        rc = check_launch("xgn_finish_raw_kernel");
    }
    if (rc) return rc;
    const long n8 = (long)B * HW * C / 8;
    hipLaunchKernelGGL(xgn_apply_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, stream, (const unsigned short*)x, scale, bias, (const unsigned short*)silu_table,
                       (unsigned short*)out, n8, C, (long)HW * C);
    return check_launch("xgn_apply_kernel");
}

int selftok_vx_silu_table_bf16(void* table, hipStream_t stream)
{
    if (!table) { set_last_error("vx_silu_table: null"); return SELFTOK_EINVAL; }
    hipLaunchKernelGGL(xsilu_table_kernel, dim3(256), dim3(256), 0, stream, (unsigned short*)table);
    return check_launch("xsilu_table_kernel");
}

size_t selftok_vx_attention_workspace_bytes(int B, int T, int C)
{
    if (B <= 0 || T <= 0) return 0;
    const size_t nblk = (size_t)(T + 511) / 512;
    return (size_t)B * T * T * 4 + (size_t)B * T * T * 2 + (size_t)B * T * C * 2 + (nblk + 1) * B * T * 4 + 512;          // + 128 floats of slack: see the rescale read of xconv_kernel
}

int selftok_vx_attention_bf16(const void* q, const void* k, const void* v, void* out, void* workspace, int B, int T, int C, hipStream_t stream)
{
    if (B == 0) return SELFTOK_OK;
    if (!q || !k || !v || !out || !workspace || B < 0 || T <= 0 || T % 32 || C % 128 || C > 4096) {
        set_last_error("vx_attention: one head, T % 32 == 0 (kv blocks of 512 keys, the last one shorter; 256 / 1024 / 1600 tokens = the SD3 VAE at 128 / 256 / 320 px), C % 128 == 0");
        return SELFTOK_EINVAL;
    }
    const int nblk = (T + 511) / 512;
    float* s = (float*)workspace;
    unsigned short* p = (unsigned short*)((char*)workspace + (size_t)B * T * T * 4);
    unsigned short* vt = p + (size_t)B * T * T;
    float* rescale = (float*)(vt + (size_t)B * T * C);               // [nblk][B * T]
    float* rowscale = rescale + (size_t)nblk * B * T;
    XConvArgs a{};
    // scores: rows = queries (per image), "output channels" = keys; fp32 C * 1/sqrt(C)
    a.x = (const unsigned short*)q; a.w = (const unsigned short*)k; a.y = s; a.H = 1; a.W = T; a.IC = C; a.OC = T; a.KH = a.KW = 1; a.stride = 1; a.pad = 0; a.OH = 1; a.OW = T;
    a.split = -1; a.mode = 1; a.out_scale = (float)(1.0 / sqrt((double)C));
    a.x_bs = (long)T * C; a.w_bs = (long)T * C; a.y_bs = (long)T * T; a.v_bs = T;
    int rc = launch_xconv(a, T, B, 0, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(xattn_softmax_kernel, dim3((unsigned)(((long)B * T + 15) / 16)), dim3(256), 0, stream, s, p, rescale, rowscale, (long)B * T, T);
    rc = check_launch("xattn_softmax_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(xtranspose_kernel, dim3(T / 32, C / 32, B), dim3(256), 0, stream, (const unsigned short*)v, vt, T, C);
    rc = check_launch("xtranspose_kernel");
    if (rc) return rc;
    // P V: rows = queries, reduction over the keys in chunks of 32; C *= rescale[block] before chunk 16 * block; bf16(C * 1/sum)
    XConvArgs g{};
    g.x = p; g.w = vt; g.y = out; g.H = 1; g.W = T; g.IC = T; g.OC = C; g.KH = g.KW = 1; g.stride = 1; g.pad = 0; g.OH = 1; g.OW = T;
    g.rescale = nblk > 1 ? rescale : nullptr; g.rs_stride = (long)B * T; g.rowscale = rowscale; g.split = 16; g.mode = 2;
    g.x_bs = (long)T * T; g.w_bs = (long)C * T; g.y_bs = (long)T * C; g.v_bs = T;
    return launch_xconv(g, T, B, 0, stream);
}

int selftok_vx_expf_f32(const float* x, float* y, long n, hipStream_t stream)
{
    if (!x || !y || n < 0) { set_last_error("vx_expf: bad argument"); return SELFTOK_EINVAL; }
    if (n == 0) return SELFTOK_OK;
    hipLaunchKernelGGL(xexpf_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, y, n);
    return check_launch("xexpf_kernel");
}

}  // extern "C"
