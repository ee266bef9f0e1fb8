/* oracle/vae_exact.c -- TEST INFRASTRUCTURE (only tests/, __graft_entry__.smoke() and bench.py's checker legs may load it).
 *
 * A bit-for-bit CPU restatement of the arithmetic torch 2.10 (CPU, this image: oneDNN 3.7.1 on an AMX-bf16 Xeon, ATen AVX2/AVX-512
 * kernels, glibc 2.35) executes for the bf16 SD3-VAE ENCODER of the reference (`self.vae.encode(images)[0].mode()`,
 * mimogpt/infer/SelftokPipeline.py:215; topology of the in-repo mirror mimogpt/models/selftok/sd3/sd3_impls.py:221-377).
 * The reference's token ids are a function of the exact bf16 rounding of every layer (a bf16 network is chaotic at the ulp level:
 * DESIGN.md section 12), so "token ids bit-exact from pixels" needs the summation ORDER of every reduction, not just its precision.
 * How each order below was established (round 4, tools/probe_cpu_bf16/, profiles/r4_cpu_bf16_orders.txt):
 *
 *  convolution  oneDNN `brg_conv_fwd:avx10_1_512_amx` / `brgconv_1x1:avx10_1_512_amx`.  One TDPBF16PS consumes 32 input channels
 *               ("chunk"): the products of the EVEN elements of the chunk are accumulated sequentially in one fp32 accumulator and the
 *               ODD elements in a second one (both from 0, round-to-nearest-even after every add; bf16 x bf16 products are exact in
 *               fp32), the chunk's value is fl(even + odd), and the tile accumulator takes C = fl(C + chunk).  Found with an
 *               FPRev-style probe (a +2^60 / -2^60 pair among unit summands reveals the summation tree) and confirmed with an exact
 *               fp32 read-out of a single tile product (a second chunk cancels the first 16 bits of the first).  Chunks follow in
 *               (kh, kw, ic-block) order; the two stride-2 layers with 128 / 256 channels instead run ic-block-major with a private
 *               partial sum per ic-block that is added to the total when its 9 taps are done; conv_in (3 channels) is ONE chunk of
 *               27 elements in (kw, kh, ic) order.  The bias is added to the fp32 total, then ONE rounding to bf16.
 *               Checked against F.conv2d on every layer shape of the encoder, B = 1, 2: 0 mismatches in 1e8 outputs.
 *               (Not modelled: the AMX unit flushes fp32 denormals, DAZ = FTZ = 1; partial sums below 1.2e-38 do not occur here.)
 *  GroupNorm    ATen GroupNormKernelImpl (contiguous NCHW path) runs the AVX2 build of RowwiseMoments (moments_utils.h): 16-element
 *               bf16 vectors split in two 8-lane fp32 halves, Welford over chunks of 16 vectors with FMAs, a binary cascade of
 *               AddMomentsVec, the 8 lanes combined by scalar AddMoments (whose two updates GCC contracts into FMAs),
 *               rstd = float(1 / sqrt(double(var) + eps)), scale = rstd * gamma, bias = fma(-scale, mean, beta),
 *               y = bf16(fma(scale, x, bias)).  0 mismatches in 4e7 outputs, mean / rstd equal to the fp32 values ATen returns.
 *  SiLU         a function of the bf16 input alone (Sleef exp + division in fp32): 65536-entry table, generated from torch itself
 *               (tests/golden/silu_bf16_table.npy); 17 entries differ from the correctly rounded x / (1 + exp(-x)).
 *  attention    ATen cpu_flash_attention, kv blocks of 512: scores by MKL's bf16 GEMM (same chunk structure as above over the 512
 *               channels), * 1/sqrt(512) in fp32, running maximum, probabilities by Vectorized<float>::fexp_u20 summed in 16 lanes
 *               (lane = key mod 16) then folded 8 / 4 / 2 / 1, rounded to bf16 for the P V product, which CONTINUES the fp32
 *               accumulator (scaled by glibc's expf(old max - new max)) chunk after chunk; sum = fma(exp, old sum, block sum);
 *               out = bf16(acc * (1 / sum)).  0 mismatches in 4e6 outputs against F.scaled_dot_product_attention.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline uint16_t f2bf(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ------------------------------------------------------------------------------------------------------------------------------
 * convolution.  x [B][H][W][IC], w [OC][KH][KW][IC], bias [OC], residual (may be NULL) and y [B][OH][OW][OC]: bf16 bit patterns.
 * Out-of-range taps read zero (pad = top/left padding; the Downsample's bottom/right zero row is the bounds check).
 * order 0: chunks in (kh, kw, icb) order; 1: (icb, kh, kw) [probe only]; 2: one flattened (kw, kh, ic) sequence cut in chunks of 32;
 * 3: icb-major with a private partial sum per icb.  residual: y = bf16(float(bf16(conv)) + float(residual)) (ResnetBlock's x + h).
 * ------------------------------------------------------------------------------------------------------------------------------ */
int vx_conv2d_nhwc(const uint16_t* xb, const uint16_t* wb, const uint16_t* bb, const uint16_t* rb, uint16_t* y, int B, int H, int W, int IC,
                   int OC, int KH, int KW, int stride, int pad, int OH, int OW, int order) {
    size_t nx = (size_t)B * H * W * IC, nw = (size_t)OC * KH * KW * IC;
    float* x = (float*)malloc(nx * sizeof(float));
    float* wt = (float*)malloc(nw * sizeof(float));          /* [KH][KW][IC][OC] */
    float* bias = (float*)malloc((size_t)OC * sizeof(float));
    if (!x || !wt || !bias) { free(x); free(wt); free(bias); return -1; }
    for (size_t i = 0; i < nx; i++) x[i] = bf2f(xb[i]);
    for (int o = 0; o < OC; o++) {
        bias[o] = bf2f(bb[o]);
        for (int t = 0; t < KH * KW; t++)
            for (int c = 0; c < IC; c++) wt[((size_t)t * IC + c) * OC + o] = bf2f(wb[((size_t)o * KH * KW + t) * IC + c]);
    }
    int nicb = (IC + 31) / 32;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int oy = 0; oy < OH; oy++) {
            float* C = (float*)malloc(sizeof(float) * OC * 4);
            float *te = C + OC, *to = C + 2 * OC, *S = C + 3 * OC;
            for (int ox = 0; ox < OW; ox++) {
                for (int o = 0; o < OC; o++) C[o] = 0.f;
                if (order == 2) {
                    int pos = 0;
                    for (int o = 0; o < OC; o++) { te[o] = 0.f; to[o] = 0.f; }
                    for (int kw = 0; kw < KW; kw++)
                        for (int kh = 0; kh < KH; kh++)
                            for (int c = 0; c < IC; c++) {
                                int iy = oy * stride - pad + kh, ix = ox * stride - pad + kw;
                                float xv = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[((size_t)(b * H + iy) * W + ix) * IC + c] : 0.f;
                                const float* wr = wt + ((size_t)(kh * KW + kw) * IC + c) * OC;
                                float* t = (pos & 1) ? to : te;
                                for (int o = 0; o < OC; o++) t[o] = t[o] + xv * wr[o];
                                if (++pos == 32) {
                                    for (int o = 0; o < OC; o++) { C[o] = C[o] + (te[o] + to[o]); te[o] = 0.f; to[o] = 0.f; }
                                    pos = 0;
                                }
                            }
                    if (pos) for (int o = 0; o < OC; o++) C[o] = C[o] + (te[o] + to[o]);
                } else {
                    int n1 = order == 0 ? KH * KW : nicb, n2 = order == 0 ? nicb : KH * KW;
                    for (int a = 0; a < n1; a++) {
                        if (order == 3) for (int o = 0; o < OC; o++) S[o] = 0.f;
                        for (int bb2 = 0; bb2 < n2; bb2++) {
                            int tap = order == 0 ? a : bb2, icb = order == 0 ? bb2 : a;
                            int kh = tap / KW, kw = tap % KW;
                            int iy = oy * stride - pad + kh, ix = ox * stride - pad + kw;
                            for (int o = 0; o < OC; o++) { te[o] = 0.f; to[o] = 0.f; }
                            if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                                const float* xr = x + ((size_t)(b * H + iy) * W + ix) * IC + icb * 32;
                                const float* wr = wt + ((size_t)(kh * KW + kw) * IC + icb * 32) * OC;
                                int kc = IC - icb * 32 < 32 ? IC - icb * 32 : 32;
                                for (int k = 0; k < kc; k += 2) {
                                    float x0 = xr[k], x1 = (k + 1 < kc) ? xr[k + 1] : 0.f;
                                    const float* w0 = wr + (size_t)k * OC;
                                    const float* w1 = wr + (size_t)(k + 1 < kc ? k + 1 : k) * OC;
                                    for (int o = 0; o < OC; o++) { te[o] = te[o] + x0 * w0[o]; to[o] = to[o] + x1 * w1[o]; }
                                }
                            }
                            if (order == 3) for (int o = 0; o < OC; o++) S[o] = S[o] + (te[o] + to[o]);
                            else for (int o = 0; o < OC; o++) C[o] = C[o] + (te[o] + to[o]);
                        }
                        if (order == 3) for (int o = 0; o < OC; o++) C[o] = C[o] + S[o];
                    }
                }
                size_t off = ((size_t)(b * OH + oy) * OW + ox) * OC;
                for (int o = 0; o < OC; o++) {
                    uint16_t h = f2bf(C[o] + bias[o]);
                    y[off + o] = rb ? f2bf(bf2f(rb[off + o]) + bf2f(h)) : h;
                }
            }
            free(C);
        }
    free(x); free(wt); free(bias);
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------------------
 * GroupNorm.  ATen's RowwiseMomentsImpl<BFloat16> as the AVX2 build executes it (moments_utils.h), on the group's elements in NCHW
 * order (channel-major, then pixels) read from an NHWC tensor.
 * ------------------------------------------------------------------------------------------------------------------------------ */
#define GL 8
typedef struct { float v[GL]; } gvec;

static void add_moments_vec(int64_t m0_add, const gvec* m1_add, const gvec* m2_add, int64_t* m0, gvec* m1, gvec* m2) {
    int64_t n = *m0 + m0_add;
    float c = n == 0 ? 0.f : (float)m0_add / (float)n;
    float m0f = (float)*m0;
    for (int l = 0; l < GL; l++) {
        float delta = m1_add->v[l] - m1->v[l];
        float m2_tmp = m2->v[l] + m2_add->v[l];
        float c_delta = c * delta;
        float m0_delta = delta * m0f;
        m1->v[l] = m1->v[l] + c_delta;
        m2->v[l] = fmaf(m0_delta, c_delta, m2_tmp);
    }
    *m0 = n;
}

static void add_moments(int64_t m0_add, float m1_add, float m2_add, int64_t* m0, float* m1, float* m2) {
    int64_t n = *m0 + m0_add;
    float c = n == 0 ? 0.f : (float)m0_add / (float)n;
    float delta = m1_add - *m1;
    *m1 = fmaf(c, delta, *m1);                                       /* GCC contracts `m1 += c * delta` */
    *m2 = *m2 + fmaf(delta * delta * c, (float)(*m0), m2_add);        /* ... and `m2_add + delta * delta * c * m0` */
    *m0 = n;
}

/* element e (NCHW order within the group) of group g of image b, read from the NHWC tensor */
static inline float gn_elem(const uint16_t* X, int64_t HW, int C, int D, int g, int64_t e) {
    int64_t d = e / HW, p = e % HW;
    return bf2f(X[p * C + (int64_t)g * D + d]);
}

static void rowwise_moments(const uint16_t* X, int64_t HW, int C, int D, int g, float* mean, float* var) {
    const int kVec = 16, kChunk = 16;
    int64_t N = (int64_t)D * HW, n = N / kVec, m = (n + kChunk - 1) / kChunk;
    int depth = 0;
    while (((int64_t)1 << depth) < m) depth++;
    int64_t m0_stk[64];
    gvec m1_stk[64], m2_stk[64];
    memset(m0_stk, 0, sizeof m0_stk); memset(m1_stk, 0, sizeof m1_stk); memset(m2_stk, 0, sizeof m2_stk);
    for (int64_t i = 0; i < m; i++) {
        int64_t base = i * kChunk * kVec;
        int64_t m0 = n - i * kChunk < kChunk ? n - i * kChunk : kChunk;
        gvec a1, b1, a2, b2;
        memset(&a1, 0, sizeof a1); memset(&b1, 0, sizeof b1); memset(&a2, 0, sizeof a2); memset(&b2, 0, sizeof b2);
        for (int64_t j = 0; j < m0; j++) {
            float cj = 1.0f / (float)(j + 1);
            for (int l = 0; l < GL; l++) {
                float x0 = gn_elem(X, HW, C, D, g, base + j * kVec + l), x1 = gn_elem(X, HW, C, D, g, base + j * kVec + 8 + l);
                float d0 = x0 - a1.v[l], d1 = x1 - b1.v[l];
                a1.v[l] = fmaf(d0, cj, a1.v[l]); b1.v[l] = fmaf(d1, cj, b1.v[l]);
                float e0 = x0 - a1.v[l], e1 = x1 - b1.v[l];
                a2.v[l] = fmaf(d0, e0, a2.v[l]); b2.v[l] = fmaf(d1, e1, b2.v[l]);
            }
        }
        add_moments_vec(m0, &a1, &a2, &m0_stk[0], &m1_stk[0], &m2_stk[0]);
        add_moments_vec(m0, &b1, &b2, &m0_stk[0], &m1_stk[0], &m2_stk[0]);
        int64_t mask = i + 1;
        for (int j = 1; j < depth && (mask & 1) == 0; ++j) {
            add_moments_vec(m0_stk[j - 1], &m1_stk[j - 1], &m2_stk[j - 1], &m0_stk[j], &m1_stk[j], &m2_stk[j]);
            m0_stk[j - 1] = 0; memset(&m1_stk[j - 1], 0, sizeof(gvec)); memset(&m2_stk[j - 1], 0, sizeof(gvec));
            mask >>= 1;
        }
    }
    for (int i = 1; i < depth; i++) add_moments_vec(m0_stk[i], &m1_stk[i], &m2_stk[i], &m0_stk[0], &m1_stk[0], &m2_stk[0]);
    int64_t m0 = 0;
    float m1 = 0.f, m2 = 0.f;
    for (int64_t i = n * kVec; i < N; i++) {           /* scalar tail (never taken at the VAE's shapes) */
        float x = gn_elem(X, HW, C, D, g, i), delta = x - m1;
        ++m0; m1 += delta / (float)m0; m2 += delta * (x - m1);
    }
    int64_t m0_add = n * kVec / GL;
    for (int l = 0; l < GL; l++) add_moments(m0_add, m1_stk[0].v[l], m2_stk[0].v[l], &m0, &m1, &m2);
    *mean = m1; *var = m2 / (float)N;
}

/* x, y [B][HW][C] bf16 (NHWC); gamma, beta [C] bf16; stats (may be NULL) [B][G][2] = mean, rstd; silu (may be NULL): 65536-entry table */
int vx_group_norm_nhwc(const uint16_t* x, const uint16_t* gamma, const uint16_t* beta, uint16_t* y, int B, int64_t HW, int C, int G, double eps,
                       const uint16_t* silu, float* stats) {
    int D = C / G;
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < B * G; i++) {
        int b = i / G, g = i % G;
        const uint16_t* X = x + (size_t)b * HW * C;
        uint16_t* Y = y + (size_t)b * HW * C;
        float mean, var;
        rowwise_moments(X, HW, C, D, g, &mean, &var);
        float rstd = (float)(1.0 / sqrt((double)fmaxf(var, 0.f) + eps));
        if (stats) { stats[2 * i] = mean; stats[2 * i + 1] = rstd; }
        for (int j = 0; j < D; j++) {
            int c = g * D + j;
            float scale = rstd * bf2f(gamma[c]);
            float bias = fmaf(-scale, mean, bf2f(beta[c]));
            for (int64_t p = 0; p < HW; p++) {
                uint16_t h = f2bf(fmaf(scale, bf2f(X[p * C + c]), bias));
                Y[p * C + c] = silu ? silu[h] : h;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------------------
 * attention (one head): q, k, v, out [B][T][Cd] bf16.  ATen cpu_flash_attention<BFloat16> with kvSplitSize = 512.
 * ------------------------------------------------------------------------------------------------------------------------------ */
static float fexp_u20(float x) {           /* Vectorized<float>::fexp_u20 (vec512_float.h) */
    const float c0 = 0.00010703434948458272f, c1 = 0.30354260500649682f, c2 = -0.22433836478672356f, c3 = -0.079204240219773236f;
    const float log2e = u2f(0x3fb8aa3b), a = 8388608.0f, b = 8388608.0f * 127.f;
    float src = x * log2e;
    float fr = src - floorf(src);
    float res = fmaf(fr, c3, c2);
    res = fmaf(fr, res, c1);
    res = fmaf(fr, res, c0);
    src = src - res;
    float tmp = fmaf(a, src, b);
    int32_t ci = (int32_t)tmp;               /* cvttps2dq */
    if (x < u2f(0xc2aeac50)) ci = 0;
    if (x > u2f(0x42b17218)) ci = 0x7F800000;
    float r; memcpy(&r, &ci, 4);
    return r;
}

int vx_attention(const uint16_t* qb, const uint16_t* kb, const uint16_t* vb, uint16_t* ob, int B, int T, int Cd) {
    const int kvsplit = 512, lanes = 16;
    float scale = (float)(1.0 / sqrt((double)Cd));
    if (Cd % 32 || T % 32) return -1;
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int b = 0; b < B; b++)
        for (int i = 0; i < T; i++) {
            float* q = (float*)malloc(sizeof(float) * (Cd * 2 + kvsplit * 2));
            float *dst = q + Cd, *s = dst + Cd, *p = s + kvsplit;
            for (int d = 0; d < Cd; d++) { q[d] = bf2f(qb[((size_t)b * T + i) * Cd + d]); dst[d] = 0.f; }
            float m_old = -INFINITY, sum_old = 0.f;
            for (int n0 = 0; n0 < T; n0 += kvsplit) {
                int nb = T - n0 < kvsplit ? T - n0 : kvsplit;
                float bm = -INFINITY;
                for (int j = 0; j < nb; j++) {
                    const uint16_t* kr = kb + ((size_t)b * T + n0 + j) * Cd;
                    float C = 0.f;
                    for (int k0 = 0; k0 < Cd; k0 += 32) {
                        float te = 0.f, to = 0.f;
                        for (int k = k0; k < k0 + 32; k += 2) { te = te + q[k] * bf2f(kr[k]); to = to + q[k + 1] * bf2f(kr[k + 1]); }
                        C = C + (te + to);
                    }
                    s[j] = C * scale;
                    if (s[j] > bm) bm = s[j];
                }
                float m_new = m_old > bm ? m_old : bm;
                float lane[16];
                for (int l = 0; l < lanes; l++) lane[l] = 0.f;
                int nv = nb / lanes * lanes;
                for (int j = 0; j < nv; j++) { float e = fexp_u20(s[j] - m_new); lane[j % lanes] += e; p[j] = bf2f(f2bf(e)); }
                for (int st = lanes / 2; st >= 1; st /= 2) for (int l = 0; l < st; l++) lane[l] = lane[l] + lane[l + st];
                float tsum = lane[0];
                for (int j = nv; j < nb; j++) { float e = expf(s[j] - m_new); tsum += e; p[j] = bf2f(f2bf(e)); }
                float exp_tmp = expf(m_old - m_new);
                sum_old = fmaf(exp_tmp, sum_old, tsum);
                m_old = m_new;
                if (n0 > 0) for (int d = 0; d < Cd; d++) dst[d] = dst[d] * exp_tmp;
                for (int d = 0; d < Cd; d++) {
                    float C = n0 > 0 ? dst[d] : 0.f;
                    for (int j0 = 0; j0 < nb; j0 += 32) {
                        float te = 0.f, to = 0.f;
                        for (int j = j0; j < j0 + 32; j += 2) {
                            te = te + p[j] * bf2f(vb[((size_t)b * T + n0 + j) * Cd + d]);
                            to = to + p[j + 1] * bf2f(vb[((size_t)b * T + n0 + j + 1) * Cd + d]);
                        }
                        C = C + (te + to);
                    }
                    dst[d] = C;
                }
            }
            float rs = 1.0f / sum_old;
            for (int d = 0; d < Cd; d++) ob[((size_t)b * T + i) * Cd + d] = f2bf(dst[d] * rs);
            free(q);
        }
    return 0;
}

/* glibc 2.35 expf (sysdeps/ieee754/flt-32/e_expf.c, EXP2F_TABLE_BITS = 5) restated: what `std::exp(float)` evaluates in the flash kernel.
 * Exposed so that the GPU twin of this routine can be compared with it AND with libm's expf on the same inputs. */
static const uint64_t EXP2F_T[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b, 0x3fef54873168b9aa,
    0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
    0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74,
    0x3feea11473eb0187, 0x3feea589994cce13, 0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
    0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
    0x3fefa4afa2a490da, 0x3fefd0765b6e4540};

float vx_expf(float x) {
    const double InvLn2N = 0x1.71547652b82fep+0 * 32, Shift = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
    if (x != x) return x;
    if (x > 0x1.62e42ep6f) return INFINITY;
    if (x < -0x1.9fe368p6f) return 0.f;
    double xd = (double)x, z = InvLn2N * xd, kd = z + Shift;
    uint64_t ki; memcpy(&ki, &kd, 8);
    kd -= Shift;
    double r = z - kd;
    uint64_t t = EXP2F_T[ki % 32] + (ki << 47);
    double s; memcpy(&s, &t, 8);
    double zz = fma(C0, r, C1), r2 = r * r, y = fma(C2, r, 1.0);
    y = fma(zz, r2, y);
    return (float)(y * s);
}
