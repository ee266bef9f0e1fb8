/* libselftok_cpu.so -- the C ABI of include/selftok_hip.h compiled for the CPU ("same symbols compiled for CPU =
 * the CPU restatement").  TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (selftoktokenizer_amd/, mimogpt/) never does and has no CPU fallback.
 *
 * Every entry point takes HOST pointers, ignores `stream`, and computes what the gfx950 kernel of the same name computes
 * (the .hip files under selftoktokenizer_amd/csrc, each of which cites the reference call site it replaces):
 *   - bit for bit where the GPU arithmetic is order-defined on both sides: the VQ nearest-code lookup (canonical l2norm, k-ordered FMA
 *     chain, first maximum, NaN-as-maximum: vector_quantize_pytorch.py:854,561,125-143), code gather + LayerNorm16, the fused
 *     residual / LayerNorm / modulate pass INCLUDING its wave-shuffle reduction order, the split-activation producers, latent format
 *     in/out, norm_ip, patchify, unpatchify + CFG + Euler, add_rows, RMSNorm's sum order;
 *   - to fp32 rounding where the GPU uses hardware transcendentals or matrix-core summation orders (v_exp_f32, v_rsq_f32, MFMA
 *     accumulation): attention, the f16x2-split Linear (same three-term arithmetic a0 w0 + 2^-11 (a0 w1 + a1 w0), k-ordered fp32
 *     accumulation), GELU, SiLU, sin/cos tables, GroupNorm statistics.
 * Opaque buffers (the packed code book, the packed f16x2 weight, the VQ workspace) have the GPU library's SIZES; the packed code book and
 * the split-activation layout are also the GPU's byte for byte (interchangeable), the packed weight image is the GPU's tile order.
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off -shared -fPIC -I../include selftok_cpu.c -o libselftok_cpu.so -lm   (oracle/Makefile)
 */
#include "selftok_hip.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static _Thread_local char g_err[256] = "";
static int fail(const char* m) { strncpy(g_err, m, sizeof(g_err) - 1); return SELFTOK_EINVAL; }
const char* selftok_last_error(void) { return g_err; }
int selftok_version(void) { return 100; }

/* ---- scalar helpers ----------------------------------------------------------------------------------------------------------- */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t orderable(float f) { uint32_t u = f2u(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
static inline float from_orderable(uint32_t k) { return u2f((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }
#define KEY_NAN 0xFFFFFFFFu

/* IEEE binary16 <-> binary32, round to nearest even (what v_cvt_f16_f32 / v_cvt_pk_f16_f32 do) */
static inline uint16_t f32_to_f16(float f)
{
    uint32_t x = f2u(f), sign = (x >> 16) & 0x8000u, ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | ((ax > 0x7F800000u) ? 0x200u | ((ax >> 13) & 0x3FFu) : 0));   /* inf / NaN */
    if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                       /* rounds to >= 65520 -> inf */
    if (ax < 0x38800000u) {                                                         /* subnormal half or zero */
        if (ax < 0x33000000u) return (uint16_t)sign;                                /* < 2^-25 -> 0 */
        uint32_t e = ax >> 23, m = (ax & 0x7FFFFFu) | 0x800000u;
        int shift = 126 - (int)e;                                                   /* 14 .. 24 */
        uint32_t r = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1))) r++;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ax - 0x38000000u, rem = r & 0x1FFFu;                               /* rebias exponent 127 -> 15 */
    r >>= 13;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) r++;
    return (uint16_t)(sign | r);
}
static inline float f16_to_f32(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1F, m = h & 0x3FFu;
    if (e == 0) {
        if (m == 0) return u2f(sign);
        float v = (float)m * 5.9604644775390625e-08f;                               /* m * 2^-24 */
        return (sign ? -v : v);
    }
    if (e == 31) return u2f(sign | 0x7F800000u | (m << 13));
    return u2f(sign | ((e + 112) << 23) | (m << 13));
}
static inline float bf2f(uint16_t u) { return u2f(((uint32_t)u) << 16); }
static inline uint16_t f2bf(float f)
{
    uint32_t u = f2u(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float rbf(float f) { return bf2f(f2bf(f)); }

/* ================================================================================================================================
 * VQ nearest-code lookup (csrc/vq.hip)
 * ============================================================================================================================== */
#define VD 16
static void l2norm16(const float* z, float* x)
{
    float a[8], s;
    for (int j = 0; j < 8; ++j) a[j] = fmaf(z[j + 8], z[j + 8], z[j] * z[j]);
    s = a[0];
    for (int j = 1; j < 8; ++j) s = s + a[j];
    float nrm = sqrtf(s);
    nrm = (nrm > 1e-12f) ? nrm : 1e-12f;
    if (s != s) nrm = s;
    for (int k = 0; k < VD; ++k) x[k] = z[k] / nrm;
}
static inline int packed_offset(int i, int k) { const int m = k >> 1, lane = (k & 1) * 32 + i; return (m >> 2) * 256 + lane * 4 + (m & 3); }

/* code c of a raw [C,16] book or of the fragment-ordered packed image */
static inline void load_code(const float* book, int packed, int c, float* e)
{
    if (!packed) { memcpy(e, book + (size_t)c * VD, VD * sizeof(float)); return; }
    const float* t = book + (size_t)(c >> 5) * 512;
    for (int k = 0; k < VD; ++k) e[k] = t[packed_offset(c & 31, k)];
}
static void vq_rows(const float* z, const float* book, int packed, void* ids, float* best, int N, int C, int flags)
{
    const int norm = (flags & SELFTOK_PRENORMED) ? 0 : 1;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < N; ++r) {
        float x[VD], e[VD];
        if (norm) l2norm16(z + (size_t)r * VD, x); else memcpy(x, z + (size_t)r * VD, sizeof(x));
        float bv = -INFINITY; int bi = 0, nan = 0;
        for (int c = 0; c < C && !nan; ++c) {
            load_code(book, packed, c, e);
            float s = 0.f;
            for (int k = 0; k < VD; ++k) s = fmaf(x[k], e[k], s);
            if (s != s) { nan = 1; bi = c; }
            else if (s > bv) { bv = s; bi = c; }
        }
        if (flags & SELFTOK_IDS_I32) ((int32_t*)ids)[r] = bi; else ((long long*)ids)[r] = bi;
        if (best) best[r] = nan ? u2f(0x7FC00000u) : bv;
    }
}
size_t selftok_vq_workspace_bytes(int N, int C) { (void)C; return (size_t)128 * (size_t)(N > 0 ? N : 1) * sizeof(unsigned long long); }
size_t selftok_vq_packed_bytes(int C, int D) { return ((size_t)C * D + 64) * sizeof(float) + (size_t)C * D * 2 * sizeof(uint16_t); }
int selftok_vq_encode_f32(const float* z, const float* codebook, void* ids, float* best, void* workspace, int N, int C, int D, int flags, hipStream_t s)
{
    (void)s; (void)workspace;
    if (N == 0 && D == VD && C > 0) return SELFTOK_OK;
    if (D != VD || C <= 0 || N < 0 || !z || !codebook || !ids) return fail("vq_encode: bad argument");
    vq_rows(z, codebook, 0, ids, best, N, C, flags);
    return SELFTOK_OK;
}
int selftok_vq_pack_codebook(const float* cb, float* packed, int C, int D, hipStream_t s)
{
    (void)s;
    if (!cb || !packed || D != VD || C <= 0 || (C & 31)) return fail("vq_pack: need D==16 and C%32==0");
    uint32_t flag = 0;
    uint16_t* p16 = (uint16_t*)(packed + (size_t)C * VD + 64);
    memset(packed + (size_t)C * VD, 0, 64 * sizeof(float));
    for (int c = 0; c < C; ++c) {
        const int t = c >> 5, i = c & 31;
        float n2 = 0.f;
        for (int k = 0; k < VD; ++k) {
            const float v = cb[(size_t)c * VD + k];
            packed[(size_t)t * 512 + packed_offset(i, k)] = v;
            const float vs = v * 128.0f;
            const uint16_t hi = f32_to_f16(vs);
            uint16_t* tile = p16 + (size_t)t * 1024;
            tile[((k >> 3) * 32 + i) * 8 + (k & 7)] = hi;
            tile[512 + ((k >> 3) * 32 + i) * 8 + (k & 7)] = f32_to_f16(vs - f16_to_f32(hi));
            if (!(fabsf(v) < 1.0e18f)) flag |= 1u;
            if (!(fabsf(vs) < 60000.f)) flag |= 2u;
            n2 = fmaf(v, v, n2);
        }
        if (!(n2 <= 1.01f)) flag |= 4u;
    }
    memcpy(packed + (size_t)C * VD, &flag, 4);
    return SELFTOK_OK;
}
int selftok_vq_argmax_partial_packed_f32(const float* z, const float* packed, void* workspace, int* nsplit_out, int N, int C, int D, int flags, hipStream_t s)
{
    (void)s;
    if (nsplit_out) *nsplit_out = 0;
    if (N == 0 && D == VD && C > 0 && !(C & 31) && nsplit_out) return SELFTOK_OK;
    if (D != VD || C <= 0 || (C & 31) || N < 0 || !z || !packed || !workspace || !nsplit_out) return fail("vq_argmax_partial_packed: bad argument");
    /* one exact candidate per row: (orderable(best) << 32) | ~idx -- the workspace format is private to the library */
    long long* idx = (long long*)malloc((size_t)(N > 0 ? N : 1) * sizeof(long long));
    float* bst = (float*)malloc((size_t)(N > 0 ? N : 1) * sizeof(float));
    if (!idx || !bst) { free(idx); free(bst); return fail("out of memory"); }
    vq_rows(z, packed, 1, idx, bst, N, C, flags & ~SELFTOK_IDS_I32);
    unsigned long long* ws = (unsigned long long*)workspace;
    for (int r = 0; r < N; ++r) {
        float v = bst[r]; if (v == 0.0f) v = 0.0f;
        uint32_t hi = (v != v) ? KEY_NAN : orderable(v);
        ws[r] = ((unsigned long long)hi << 32) | (uint32_t)(~(uint32_t)idx[r]);
    }
    free(idx); free(bst);
    *nsplit_out = 1;
    return SELFTOK_OK;
}
int selftok_vq_finalize_packed(const void* workspace, const float* z, const float* packed, void* ids, float* best, int N, int C, int D, int nsplit, int flags, hipStream_t s)
{
    (void)s; (void)z; (void)packed;
    if (N == 0) return SELFTOK_OK;
    if (!workspace || !ids || N < 0 || nsplit <= 0 || D != VD || (C & 31)) return fail("vq_finalize_packed: bad argument");
    const unsigned long long* ws = (const unsigned long long*)workspace;
    for (int r = 0; r < N; ++r) {
        const uint32_t hi = (uint32_t)(ws[r] >> 32), id = ~(uint32_t)ws[r];
        if (flags & SELFTOK_IDS_I32) ((int32_t*)ids)[r] = (int32_t)id; else ((long long*)ids)[r] = id;
        if (best) best[r] = (hi == KEY_NAN) ? u2f(0x7FC00000u) : from_orderable(hi);
    }
    return SELFTOK_OK;
}
int selftok_vq_encode_packed_f32(const float* z, const float* packed, void* ids, float* best, void* workspace, int N, int C, int D, int flags, hipStream_t s)
{
    if (!ids && N != 0) return fail("vq_encode_packed: bad argument");
    int split = 0;
    int rc = selftok_vq_argmax_partial_packed_f32(z, packed, workspace, &split, N, C, D, flags, s);
    if (rc || N == 0) return rc;
    return selftok_vq_finalize_packed(workspace, z, packed, ids, best, N, C, D, split, flags, s);
}
int selftok_code_gather_ln_f32(const void* ids, const float* cb, const float* ln_w, const float* ln_b, float* out, int n, int C, int D, float eps, int flags, hipStream_t s)
{
    (void)s;
    if (n == 0) return SELFTOK_OK;
    if (D != VD || n < 0 || !ids || !cb || !out || ((ln_w == NULL) != (ln_b == NULL))) return fail("code_gather_ln: bad argument");
    for (int r = 0; r < n; ++r) {
        long long id = (flags & SELFTOK_IDS_I32) ? ((const int32_t*)ids)[r] : ((const long long*)ids)[r];
        if (id < 0) id += C;
        id = id < 0 ? 0 : (id >= C ? C - 1 : id);
        float v[VD];
        memcpy(v, cb + (size_t)id * VD, sizeof(v));
        if (ln_w) {
            float mean = 0.f, var = 0.f;
            for (int k = 0; k < VD; ++k) mean += v[k];
            mean *= (1.0f / VD);
            for (int k = 0; k < VD; ++k) { float d = v[k] - mean; var = fmaf(d, d, var); }
            var *= (1.0f / VD);
            const float rstd = 1.0f / sqrtf(var + eps);
            for (int k = 0; k < VD; ++k) v[k] = (v[k] - mean) * rstd * ln_w[k] + ln_b[k];
        }
        memcpy(out + (size_t)r * VD, v, sizeof(v));
    }
    return SELFTOK_OK;
}
int selftok_vq_ema_accumulate_f32(const float* z, const void* ids, float* bins, float* esum, int N, int C, int D, int flags, hipStream_t s)
{
    (void)s;
    if (N == 0) return SELFTOK_OK;
    if (D != VD || N < 0 || C <= 0 || !z || !ids || !bins || !esum) return fail("vq_ema_accumulate: bad argument");
    for (int r = 0; r < N; ++r) {
        long id = (flags & SELFTOK_IDS_I32) ? ((const int32_t*)ids)[r] : (long)((const long long*)ids)[r];
        if (id < 0 || id >= C) continue;
        float x[VD];
        if (flags & SELFTOK_PRENORMED) memcpy(x, z + (size_t)r * VD, sizeof(x)); else l2norm16(z + (size_t)r * VD, x);
        for (int k = 0; k < VD; ++k) esum[(size_t)id * VD + k] += x[k];
        bins[id] += 1.0f;
    }
    return SELFTOK_OK;
}
int selftok_vq_tpc_update_f32(float* tpc, const void* ids, int B, int K, int C, float w, int flags, hipStream_t s)
{
    (void)s;
    if (!tpc || K <= 0 || C <= 0 || (C & 3) || B < 0 || (B > 0 && !ids)) return fail("vq_tpc_update: bad argument (C % 4 == 0)");
    const long n = (long)K * C;
    if (w < 0.5f) { for (long i = 0; i < n; ++i) tpc[i] = fmaf(w, -tpc[i], tpc[i]); }
    else { const float k = 1.0f - w; for (long i = 0; i < n; ++i) tpc[i] *= k; }
    const float add = B > 0 ? w / (float)B : 0.f;
    for (long i = 0; i < (long)B * K; ++i) {
        long id = (flags & SELFTOK_IDS_I32) ? ((const int32_t*)ids)[i] : (long)((const long long*)ids)[i];
        if (id < 0 || id >= C) continue;
        tpc[(size_t)(i % K) * C + id] += add;
    }
    return SELFTOK_OK;
}

/* ---- entropy regularisers without the [B, K, C] tensor (csrc/vq_entropy.hip): same reductions, scalar loops in double ------------------ */
size_t selftok_vq_softmax_workspace_bytes(int N) { return (size_t)8 * (size_t)(N > 0 ? N : 1) * 36 * sizeof(float); }
static void unit_row(const float* z, float* x, int norm, float* nrm_out)
{
    if (norm) {
        l2norm16(z, x);
        float a[8], s;
        for (int j = 0; j < 8; ++j) a[j] = fmaf(z[j + 8], z[j + 8], z[j] * z[j]);
        s = a[0];
        for (int j = 1; j < 8; ++j) s = s + a[j];
        float n = sqrtf(s);
        *nrm_out = (n > 1e-12f) ? n : 1e-12f;
    } else { memcpy(x, z, VD * sizeof(float)); *nrm_out = 1.0f; }
}
static inline float score16(const float* x, const float* e) { float s = 0.f; for (int k = 0; k < VD; ++k) s = fmaf(x[k], e[k], s); return s; }
int selftok_vq_softmax_stats_f32(const float* z, const float* cb, float* rowstats, float* colmean, void* workspace, int B, int K, int C, int D,
                                 float scale, int flags, hipStream_t s)
{
    (void)s;
    if (D != VD || B < 0 || K <= 0 || C <= 0 || K > 65535) return fail("vq_softmax_stats: bad argument (D == 16, K <= 65535)");
    if (B == 0) return SELFTOK_OK;
    if (!z || !cb || !rowstats || !workspace) return fail("vq_softmax_stats: null pointer");
    const int norm = (flags & SELFTOK_PRENORMED) ? 0 : 1;
    const long N = (long)B * K;
#pragma omp parallel for schedule(static)
    for (long n = 0; n < N; ++n) {
        float x[VD], nr;
        unit_row(z + n * VD, x, norm, &nr);
        double S = 0.0, T = 0.0;
        for (int c = 0; c < C; ++c) { const double l = (double)(score16(x, cb + (size_t)c * VD) * scale); const double p = exp(l); S += p; T += p * l; }
        rowstats[2 * n] = (float)(1.0 / S);
        rowstats[2 * n + 1] = (float)(log(S) - T / S);
    }
    if (!colmean) return SELFTOK_OK;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; ++k) {
        double* acc = (double*)calloc((size_t)C, sizeof(double));
        for (int b = 0; b < B; ++b) {
            const long n = (long)b * K + k;
            float x[VD], nr;
            unit_row(z + n * VD, x, norm, &nr);
            const double inv = rowstats[2 * n];
            for (int c = 0; c < C; ++c) acc[c] += exp((double)(score16(x, cb + (size_t)c * VD) * scale)) * inv;
        }
        for (int c = 0; c < C; ++c) colmean[(size_t)k * C + c] = (float)(acc[c] / B);
        free(acc);
    }
    return SELFTOK_OK;
}
int selftok_vq_softmax_backward_f32(const float* z, const float* cb, const float* rowstats, const float* g, float* grad_z, void* workspace,
                                    int B, int K, int C, int D, float scale, int flags, hipStream_t s)
{
    (void)s;
    if (D != VD || B < 0 || K <= 0 || C <= 0 || K > 65535) return fail("vq_softmax_backward: bad argument (D == 16, K <= 65535)");
    if (B == 0) return SELFTOK_OK;
    if (!z || !cb || !rowstats || !g || !grad_z || !workspace) return fail("vq_softmax_backward: null pointer");
    const int norm = (flags & SELFTOK_PRENORMED) ? 0 : 1;
    const long N = (long)B * K;
#pragma omp parallel for schedule(static)
    for (long n = 0; n < N; ++n) {
        const int k = (int)(n % K);
        float x[VD], nr;
        unit_row(z + n * VD, x, norm, &nr);
        double A[VD] = {0}, m[VD] = {0}, t = 0.0;
        const double inv = rowstats[2 * n];
        for (int c = 0; c < C; ++c) {
            const float* e = cb + (size_t)c * VD;
            const double p = exp((double)(score16(x, e) * scale)) * inv, pg = p * g[(size_t)k * C + c];
            t += pg;
            for (int q = 0; q < VD; ++q) { m[q] += p * e[q]; A[q] += pg * e[q]; }
        }
        double gx[VD], xd = 0.0;
        for (int q = 0; q < VD; ++q) { gx[q] = (double)scale / B * (A[q] - t * m[q]); xd += x[q] * gx[q]; }
        for (int q = 0; q < VD; ++q) grad_z[n * VD + q] = (float)(norm ? (gx[q] - x[q] * xd) / nr : gx[q]);
    }
    return SELFTOK_OK;
}

/* ================================================================================================================================
 * fused element-wise passes (csrc/elementwise.hip)
 * ============================================================================================================================== */
static inline size_t split_blk_index(long row, int k, int plane, int KT)
{
    return ((((size_t)(row >> 4) * KT + (k >> 5)) * 2 + plane) << 9) + ((row & 15) << 5) + (k & 31);
}
/* the kernels' group_sum<G>: butterfly of xor-shuffles, every lane ends with the same value */
static float butterfly(float* v, int G)
{
    float t[64];
    for (int o = G / 2; o > 0; o >>= 1) {
        for (int l = 0; l < G; ++l) t[l] = v[l] + v[l ^ o];
        memcpy(v, t, (size_t)G * sizeof(float));
    }
    return v[0];
}
static int ln_mod(const float* x, const float* y, const float* gate, const float* shift, const float* scale, float* x_out, float* n_out, uint16_t* n_blk,
                  int* overflow, int B, int T, int H, long msb, long mst, long gsb, long gst, float eps)
{
    if (!x || B < 0 || T < 0 || ((shift == NULL) != (scale == NULL)) || (n_blk && (H % 32)) || (!n_out && !n_blk && !(y && x_out))) return fail("residual_ln_mod: bad argument");
    int G, VPL;
    switch (H) { case 64: G = 16; VPL = 1; break; case 256: G = 64; VPL = 1; break; case 512: G = 64; VPL = 2; break; case 1024: G = 64; VPL = 4; break;
                 case 1536: G = 64; VPL = 6; break; default: return fail("residual_ln_mod: unsupported hidden size (64/256/512/1024/1536)"); }
    const long rows = (long)B * T;
    int ovf = 0;
#pragma omp parallel for schedule(static) reduction(| : ovf)
    for (long gid = 0; gid < rows; ++gid) {
        const long b = gid / T, t = gid - b * T;
        float v[1536], part[64];
        const float* xr = x + (size_t)gid * H;
        memcpy(v, xr, (size_t)H * sizeof(float));
        if (y) {
            const float* yr = y + (size_t)gid * H;
            const float* gr = gate ? gate + b * gsb + t * gst : NULL;
            for (int c = 0; c < H; ++c) v[c] = gr ? v[c] + gr[c] * yr[c] : v[c] + yr[c];
            if (x_out) memcpy(x_out + (size_t)gid * H, v, (size_t)H * sizeof(float));
        }
        if (!n_out && !n_blk) continue;
        /* lane l owns float4 chunks (i * G + l), i < VPL: per-lane partial in chunk order, then the butterfly */
        for (int l = 0; l < G; ++l) {
            float s = 0.f;
            for (int i = 0; i < VPL; ++i) { const float* q = v + (size_t)(i * G + l) * 4; s += (q[0] + q[1]) + (q[2] + q[3]); }
            part[l] = s;
        }
        const float mean = butterfly(part, G) * (1.0f / H);
        for (int l = 0; l < G; ++l) {
            float q2 = 0.f;
            for (int i = 0; i < VPL; ++i) {
                const float* q = v + (size_t)(i * G + l) * 4;
                const float a = q[0] - mean, bq = q[1] - mean, c = q[2] - mean, d = q[3] - mean;
                q2 += (a * a + bq * bq) + (c * c + d * d);
            }
            part[l] = q2;
        }
        const float var = butterfly(part, G) * (1.0f / H);
        const float rstd = 1.0f / sqrtf(var + eps);
        const float* sh = shift ? shift + b * msb + t * mst : NULL;
        const float* sc = scale ? scale + b * msb + t * mst : NULL;
        float mx = 0.f;
        for (int c = 0; c < H; ++c) {
            float o = (v[c] - mean) * rstd;
            if (sc) o = o * (1.0f + sc[c]) + sh[c];
            if (n_out) n_out[(size_t)gid * H + c] = o;
            if (n_blk) {
                const uint16_t hh = f32_to_f16(o);
                n_blk[split_blk_index(gid, c, 0, H / 32)] = hh;
                n_blk[split_blk_index(gid, c, 1, H / 32)] = f32_to_f16((o - f16_to_f32(hh)) * 2048.0f);
                mx = fmaxf(mx, fabsf(o));
            }
        }
        if (n_blk && !(mx < 65504.0f)) ovf |= 1;
    }
    if (ovf && overflow) *overflow |= ovf;
    return SELFTOK_OK;
}
int selftok_residual_ln_mod_f32(const float* x, const float* y, const float* gate, const float* shift, const float* scale, float* x_out, float* n_out,
                                int B, int T, int H, long msb, long mst, long gsb, long gst, float eps, hipStream_t s)
{
    (void)s;
    return ln_mod(x, y, gate, shift, scale, x_out, n_out, NULL, NULL, B, T, H, msb, mst, gsb, gst, eps);
}
int selftok_residual_ln_mod_split(const float* x, const float* y, const float* gate, const float* shift, const float* scale, float* x_out, void* n_blk, int* overflow,
                                  int B, int T, int H, long msb, long mst, long gsb, long gst, float eps, hipStream_t s)
{
    (void)s;
    if (!n_blk) return fail("residual_ln_mod_split: null output");
    return ln_mod(x, y, gate, shift, scale, x_out, NULL, (uint16_t*)n_blk, overflow, B, T, H, msb, mst, gsb, gst, eps);
}
static inline float gelu_tanh(float x) { const float k0 = 0.7978845608028654f, k1 = 0.044715f; return 0.5f * x * (1.0f + tanhf(k0 * (x + k1 * x * x * x))); }
int selftok_bias_gelu_f32(float* h, const float* bias, long rows, int cols, hipStream_t s)
{
    (void)s;
    if (!h || rows < 0 || cols <= 0 || (cols & 3)) return fail("bias_gelu: cols must be a multiple of 4");
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) { float v = h[r * cols + c]; if (bias) v += bias[c]; h[r * cols + c] = gelu_tanh(v); }
    return SELFTOK_OK;
}
int selftok_silu_f32(const float* in, float* out, long n, hipStream_t s)
{
    (void)s;
    if (!in || !out || n < 0) return fail("silu: bad argument");
    for (long i = 0; i < n; ++i) out[i] = in[i] / (1.0f + expf(-in[i]));
    return SELFTOK_OK;
}
int selftok_add_rows_f32(const float* in, const float* table, float* out, int B, long per_sample, hipStream_t s)
{
    (void)s;
    if (!in || !table || !out || B < 0 || per_sample <= 0 || (per_sample & 3)) return fail("add_rows: bad argument");
    for (long i = 0; i < (long)B * per_sample; ++i) out[i] = in[i] + table[i % per_sample];
    return SELFTOK_OK;
}
int selftok_timestep_embed_f32(const float* t, const float* freqs, float* out, int n, int dim, float t_scale, hipStream_t s)
{
    (void)s;
    if (!t || !freqs || !out || n < 0 || dim <= 0 || (dim & 1)) return fail("timestep_embed: bad argument");
    const int half = dim / 2;
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < half; ++c) { const float a = (t[r] * t_scale) * freqs[c]; out[(size_t)r * dim + c] = cosf(a); out[(size_t)r * dim + half + c] = sinf(a); }
    return SELFTOK_OK;
}
int selftok_patchify_f32(const float* x, float* out, int B, int C, int Hh, int Ww, hipStream_t s)
{
    (void)s;
    if (!x || !out || B < 0 || C <= 0 || (Hh & 1) || (Ww & 1)) return fail("patchify: bad argument");
    const int hp = Hh / 2, wp = Ww / 2;
    for (int b = 0; b < B; ++b) for (int h = 0; h < hp; ++h) for (int w = 0; w < wp; ++w) for (int c = 0; c < C; ++c) {
        const float* src = x + (((size_t)b * C + c) * Hh + 2 * h) * Ww + 2 * w;
        float* o = out + (((size_t)b * hp + h) * wp + w) * (C * 4) + c * 4;
        o[0] = src[0]; o[1] = src[1]; o[2] = src[Ww]; o[3] = src[Ww + 1];
    }
    return SELFTOK_OK;
}
int selftok_unpatchify_cfg_euler_f32(const float* yc, const float* yu, const float* x, float* x_out, float* v_out, int B, int C, int hp, int wp, float dt, float cfg, hipStream_t s)
{
    (void)s;
    if (!yc || B < 0 || (x_out && !x) || (!x_out && !v_out)) return fail("unpatchify_cfg_euler: bad argument");
    const int Hh = 2 * hp, Ww = 2 * wp;
    for (int b = 0; b < B; ++b) for (int c = 0; c < C; ++c) for (int row = 0; row < Hh; ++row) for (int col = 0; col < Ww; ++col) {
        const int h = row >> 1, p = row & 1, w = col >> 1, q = col & 1;
        const size_t tok = ((size_t)b * hp + h) * wp + w, f = (size_t)(p * 2 + q) * C + c;
        float v = yc[tok * (4 * C) + f];
        if (yu) { const float u = yu[tok * (4 * C) + f]; v = u + cfg * (v - u); }
        const size_t o = (((size_t)b * C + c) * Hh + row) * Ww + col;
        if (v_out) v_out[o] = v;
        if (x_out) x_out[o] = x[o] - dt * v;
    }
    return SELFTOK_OK;
}
int selftok_rmsnorm_f32(const float* x, const float* w, float* out, long rows, int dim, float eps, hipStream_t s)
{
    (void)s;
    if (!x || !out || rows < 0 || dim <= 0) return fail("rmsnorm: bad argument");
    for (long r = 0; r < rows; ++r) {
        float part[64];
        for (int l = 0; l < 16; ++l) { float a = 0.f; for (int c = l; c < dim; c += 16) a += x[r * dim + c] * x[r * dim + c]; part[l] = a; }
        const float ss = butterfly(part, 16);
        const float rr = 1.0f / sqrtf(ss / dim + eps);          /* the GPU uses v_rsq_f32 (1 ulp) */
        for (int c = 0; c < dim; ++c) out[r * dim + c] = x[r * dim + c] * rr * (w ? w[c] : 1.0f);
    }
    return SELFTOK_OK;
}
int selftok_rotary_f32(const float* t, const float* freqs, float* out, long rows, int seq, int dim, float scale, hipStream_t s)
{
    (void)s;
    if (!t || !freqs || !out || rows < 0 || seq <= 0 || dim <= 0 || (dim & 1)) return fail("rotary: bad argument");
    for (long r = 0; r < rows; ++r) {
        const float* f = freqs + (size_t)(r % seq) * dim;
        for (int pr = 0; pr < dim / 2; ++pr) {
            const float x1 = t[r * dim + 2 * pr], x2 = t[r * dim + 2 * pr + 1], f1 = f[2 * pr], f2 = f[2 * pr + 1];
            out[r * dim + 2 * pr] = x1 * cosf(f1) * scale + (-x2) * sinf(f1) * scale;
            out[r * dim + 2 * pr + 1] = x2 * cosf(f2) * scale + x1 * sinf(f2) * scale;
        }
    }
    return SELFTOK_OK;
}

/* ================================================================================================================================
 * fp32-equivalent Linear on fp16 pairs (csrc/gemm_split.hip):  x = x0 + x1 2^-11,  a.w ~ a0 w0 + 2^-11 (a0 w1 + a1 w0)
 * ============================================================================================================================== */
#define BN 128
#define BK 32
#define W_TILE_HALFS 8192          /* one (n-block, k-tile): [plane 2][g 4][n 128][8] */
size_t selftok_linear_f16x2_packed_bytes(int N, int K) { return (N > 0 && K > 0 && N % BN == 0 && K % BK == 0) ? (size_t)4 * N * K : 0; }
size_t selftok_split_f16x2_bytes(long rows, int cols) { return (rows >= 0 && cols > 0 && cols % 32 == 0) ? (size_t)((rows + 15) / 16) * 16 * cols * 4 : 0; }
static inline void split1(float v, uint16_t* hi, uint16_t* lo) { *hi = f32_to_f16(v); *lo = f32_to_f16((v - f16_to_f32(*hi)) * 2048.0f); }
static inline size_t wp_index(int n, int k, int plane, int KT)
{
    const int nb = n / BN, nl = n % BN, kt = k / BK, g = (k % BK) / 8;
    return ((size_t)nb * KT + kt) * W_TILE_HALFS + (size_t)plane * 4096 + (size_t)g * 1024 + (size_t)nl * 8 + (k & 7);
}
int selftok_linear_f16x2_pack_weight(const float* W, void* packed, int N, int K, int* overflow, hipStream_t s)
{
    (void)s;
    if (!W || !packed || selftok_linear_f16x2_packed_bytes(N, K) == 0) return fail("linear_f16x2_pack_weight: need N % 128 == 0 and K % 32 == 0");
    uint16_t* p = (uint16_t*)packed;
    int ovf = 0;
    for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) {
        const float v = W[(size_t)n * K + k];
        split1(v, &p[wp_index(n, k, 0, K / BK)], &p[wp_index(n, k, 1, K / BK)]);
        if (!(fabsf(v) < 65504.0f)) ovf = 2;
    }
    if (ovf && overflow) *overflow |= ovf;
    return SELFTOK_OK;
}
int selftok_split_f16x2_f32(const float* x, long ld, void* blk, long rows, int cols, int* overflow, hipStream_t s)
{
    (void)s;
    if (!x || !blk || rows < 0 || cols <= 0 || (cols % 32)) return fail("split_f16x2: bad argument");
    uint16_t* p = (uint16_t*)blk;
    int ovf = 0;
    for (long r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) {
        const float v = x[r * ld + c];
        split1(v, &p[split_blk_index(r, c, 0, cols / 32)], &p[split_blk_index(r, c, 1, cols / 32)]);
        if (!(fabsf(v) < 65504.0f)) ovf = 1;
    }
    if (ovf && overflow) *overflow |= ovf;
    return SELFTOK_OK;
}
/* x sigmoid(2u): the GEMM epilogue's form of the tanh-GELU (gemm_split.hip gelu_tanh_f) */
static inline float gelu_sig(float x) { const float k0 = 0.7978845608028654f, k1 = 0.044715f; const float u = k0 * (x + k1 * x * x * x); return x * (1.0f / (1.0f + exp2f(u * (-2.0f * 1.4426950408889634f)))); }
/* row m of (a0, a1) against weight row n: separate fp32 accumulators for the high and the two low terms, combined once.  One
 * v_mfma_f32_32x32x16_f16 adds 16 exact fp16 x fp16 products to its fp32 accumulator with ONE rounding: modelled as a double sum of the
 * 16 products (exact: 22-bit products, 16 of them) rounded to fp32 together with the accumulator, k-blocks in ascending order. */
static float dot_split_range(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* wp, int n, int K, int kbeg, int kend)
{
    float hi = 0.f, lo = 0.f;
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        double sh = 0.0, s1 = 0.0, s2 = 0.0;
        for (int k = k0; k < k0 + 16; ++k) {
            const double a0 = f16_to_f32(a_hi[k]), a1 = f16_to_f32(a_lo[k]);
            const double w0 = f16_to_f32(wp[wp_index(n, k, 0, K / BK)]), w1 = f16_to_f32(wp[wp_index(n, k, 1, K / BK)]);
            sh += a0 * w0; s1 += a0 * w1; s2 += a1 * w0;
        }
        hi = (float)((double)hi + sh);
        lo = (float)((double)lo + s1);
        lo = (float)((double)lo + s2);
    }
    return hi + lo * (1.0f / 2048.0f);
}
/* ksplit == 1: the single-pass kernel.  ksplit > 1 (selftok_linear_f16x2_split_k): each of the ksplit work-groups of a tile produces
 * hi + lo 2^-11 over its K / ksplit consecutive k (+ 0 bias), the finish kernel adds those fp32 partials in ascending order */
static float dot_split(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* wp, int n, int K, int ksplit)
{
    const int span = K / ksplit;
    float v = dot_split_range(a_hi, a_lo, wp, n, K, 0, span);
    if (ksplit > 1) v = v + 0.f;                      /* the partial kernel's epilogue adds a zero bias: -0 becomes +0 */
    for (int s = 1; s < ksplit; ++s) v = v + (dot_split_range(a_hi, a_lo, wp, n, K, s * span, (s + 1) * span) + 0.f);
    return v;
}
static int linear_core_k(const float* A, long lda, const uint16_t* a_blk, const void* packed, const float* bias, float* out, uint16_t* out_blk, long ldo,
                       const float* resid, long ldr, const float* gate, long gsb, long gst, int T, int M, int N, int K, int flags, int ksplit, int* overflow)
{
    if (!packed || selftok_linear_f16x2_packed_bytes(N, K) == 0 || M < 0) return fail("linear_f16x2: bad argument (N % 128 == 0, K % 32 == 0)");
    const uint16_t* wp = (const uint16_t*)packed;
    int ovf = 0;
#pragma omp parallel for schedule(static) reduction(| : ovf)
    for (int m = 0; m < M; ++m) {
        uint16_t ah[6144], al[6144];
        for (int k = 0; k < K; ++k) {
            if (A) { const float v = A[(size_t)m * lda + k]; split1(v, &ah[k], &al[k]); if (!(fabsf(v) < 65504.0f)) ovf |= 1; }
            else { ah[k] = a_blk[split_blk_index(m, k, 0, K / 32)]; al[k] = a_blk[split_blk_index(m, k, 1, K / 32)]; }
        }
        for (int n = 0; n < N; ++n) {
            float v = dot_split(ah, al, wp, n, K, ksplit);
            if (bias || ksplit == 1) v = v + (bias ? bias[n] : 0.f);
            if (flags & SELFTOK_LINEAR_GELU) v = gelu_sig(v);
            if (resid) {
                const float g = gate ? gate[(size_t)(m / T) * gsb + (size_t)(m % T) * gst + n] : 1.0f;
                v = gate ? resid[(size_t)m * ldr + n] + g * v : resid[(size_t)m * ldr + n] + v;
            }
            if (!(fabsf(v) < INFINITY)) ovf |= 1;
            if (out) out[(size_t)m * ldo + n] = v;
            if (out_blk) { split1(v, &out_blk[split_blk_index(m, n, 0, N / 32)], &out_blk[split_blk_index(m, n, 1, N / 32)]); if (!(fabsf(v) < 65504.0f)) ovf |= 1; }
        }
    }
    if (ovf && overflow) *overflow |= ovf;
    return SELFTOK_OK;
}
static int linear_core(const float* A, long lda, const uint16_t* a_blk, const void* packed, const float* bias, float* out, uint16_t* out_blk, long ldo,
                       const float* resid, long ldr, const float* gate, long gsb, long gst, int T, int M, int N, int K, int flags, int* overflow)
{
    return linear_core_k(A, lda, a_blk, packed, bias, out, out_blk, ldo, resid, ldr, gate, gsb, gst, T, M, N, K, flags, 1, overflow);
}
int selftok_linear_f16x2_f32(const float* A, long lda, const void* packed, const float* bias, float* out, long ldo, int M, int N, int K, int flags, int* overflow, hipStream_t s)
{
    (void)s;
    if (!A || !out || K > 6144) return fail("linear_f16x2: bad argument");
    return linear_core(A, lda, NULL, packed, bias, out, NULL, ldo, NULL, 0, NULL, 0, 0, 1, M, N, K, flags, overflow);
}
int selftok_linear_f16x2_split(const void* a_blk, const void* packed, const float* bias, float* out, void* out_blk, long ldo, int M, int N, int K, int flags, int* overflow, hipStream_t s)
{
    (void)s;
    if (!a_blk || (!out && !out_blk) || K > 6144) return fail("linear_f16x2_split: bad argument");
    return linear_core(NULL, 0, (const uint16_t*)a_blk, packed, bias, out, (uint16_t*)out_blk, ldo, NULL, 0, NULL, 0, 0, 1, M, N, K, flags, overflow);
}
int selftok_linear_f16x2_split_residual(const void* a_blk, const void* packed, const float* bias, const float* resid, long ldr, const float* gate, long gsb, long gst, int T,
                                        float* out, long ldo, int M, int N, int K, int* overflow, hipStream_t s)
{
    (void)s;
    if (!a_blk || !resid || !out || T <= 0 || K > 6144) return fail("linear_f16x2_split_residual: bad argument");
    return linear_core(NULL, 0, (const uint16_t*)a_blk, packed, bias, out, NULL, ldo, resid, ldr, gate, gsb, gst, T, M, N, K, 0, overflow);
}

size_t selftok_linear_f16x2_splitk_workspace_bytes(int M, int N, int ksplit) { return (M > 0 && N > 0 && ksplit > 1) ? (size_t)ksplit * M * N * 4 : 0; }
static int splitk_ok(int K, int ksplit, const void* workspace) { return ksplit >= 2 && ksplit <= 64 && (K / BK) % ksplit == 0 && workspace && !((size_t)workspace & 15); }
int selftok_linear_f16x2_split_k(const void* a_blk, const void* packed, const float* bias, float* out, void* out_blk, long ldo, int M, int N, int K, int flags,
                                 int ksplit, void* workspace, int* overflow, hipStream_t s)
{
    if (ksplit == 1) return selftok_linear_f16x2_split(a_blk, packed, bias, out, out_blk, ldo, M, N, K, flags, overflow, s);
    if (M == 0 && N > 0 && K > 0 && N % BN == 0 && K % BK == 0) return SELFTOK_OK;
    if (!a_blk || (!out && !out_blk) || K > 6144 || K <= 0 || K % BK || !splitk_ok(K, ksplit, workspace)) return fail("linear_f16x2_split_k: bad argument");
    return linear_core_k(NULL, 0, (const uint16_t*)a_blk, packed, bias, out, (uint16_t*)out_blk, ldo, NULL, 0, NULL, 0, 0, 1, M, N, K, flags, ksplit, overflow);
}
int selftok_linear_f16x2_split_residual_k(const void* a_blk, const void* packed, const float* bias, const float* resid, long ldr, const float* gate, long gsb, long gst, int T,
                                          float* out, long ldo, int M, int N, int K, int ksplit, void* workspace, int* overflow, hipStream_t s)
{
    if (ksplit == 1) return selftok_linear_f16x2_split_residual(a_blk, packed, bias, resid, ldr, gate, gsb, gst, T, out, ldo, M, N, K, overflow, s);
    if (M == 0 && N > 0 && K > 0 && N % BN == 0 && K % BK == 0) return SELFTOK_OK;
    if (!a_blk || !resid || !out || T <= 0 || K > 6144 || K <= 0 || K % BK || !splitk_ok(K, ksplit, workspace)) return fail("linear_f16x2_split_residual_k: bad argument");
    return linear_core_k(NULL, 0, (const uint16_t*)a_blk, packed, bias, out, NULL, ldo, resid, ldr, gate, gsb, gst, T, M, N, K, 0, ksplit, overflow);
}

/* ================================================================================================================================
 * two-segment attention with the implicit prefix-visibility mask (csrc/attention.hip)
 * ============================================================================================================================== */
int selftok_attn_f32(const selftok_attn_desc* d, hipStream_t s)
{
    (void)s;
    if (!d || d->B < 0 || d->H <= 0) return fail("attn: bad descriptor");
    if (d->head_dim != 64 && d->head_dim != 16) return fail("attn: head_dim must be 64 or 16");
    const int Dh = d->head_dim;
    if (Dh == 16 && (d->seg[0].len != 0 || d->kvis || !d->seg[1].q)) return fail("attn(head_dim 16): single unmasked segment only");
    if (d->mode != 0 && d->mode != SELFTOK_ATTN_F16X2) return fail("attn: unknown mode");
    int ovf = 0;
#pragma omp parallel for collapse(2) schedule(dynamic) reduction(| : ovf)
    for (int b = 0; b < d->B; ++b) for (int h = 0; h < d->H; ++h) {
        int n0 = d->seg[0].len;
        if (d->kvis) { int kv = d->kvis[b] + 1; n0 = kv < n0 ? (kv < 0 ? 0 : kv) : n0; }
        for (int sg = 0; sg < 2; ++sg) {
            const selftok_attn_seg* qs = &d->seg[sg];
            if (!qs->q || qs->len <= 0) continue;
            const int rows = sg == 0 ? n0 : qs->len;
            const int n1 = (sg == 1 || d->seg0_sees_seg1) ? d->seg[1].len : 0;
            const int nk = n0 + n1;
            float* sc = (float*)malloc((size_t)(nk > 0 ? nk : 1) * sizeof(float));
            for (int r = 0; r < rows; ++r) {
                const float* q = qs->q + (size_t)b * qs->q_bs + (size_t)r * qs->q_rs + h * Dh;
                float mx = -INFINITY;
                for (int j = 0; j < nk; ++j) {
                    const selftok_attn_seg* ks = &d->seg[j < n0 ? 0 : 1];
                    const int kj = j < n0 ? j : j - n0;
                    const float* k = ks->k + (size_t)b * ks->k_bs + (size_t)kj * ks->k_rs + h * Dh;
                    float a = 0.f;
                    for (int e = 0; e < Dh; ++e) { a = fmaf(q[e], k[e], a); if (d->mode && (!(fabsf(q[e]) < 65504.0f) || !(fabsf(k[e]) < 65504.0f))) ovf |= 4; }
                    sc[j] = a * d->scale;
                    mx = fmaxf(mx, sc[j]);
                }
                float l = 0.f, o[64];
                for (int e = 0; e < Dh; ++e) o[e] = 0.f;
                for (int j = 0; j < nk; ++j) {
                    const selftok_attn_seg* ks = &d->seg[j < n0 ? 0 : 1];
                    const int kj = j < n0 ? j : j - n0;
                    const float* v = ks->v + (size_t)b * ks->v_bs + (size_t)kj * ks->v_rs + h * Dh;
                    const float p = expf(sc[j] - mx);
                    l += p;
                    for (int e = 0; e < Dh; ++e) { o[e] = fmaf(p, v[e], o[e]); if (d->mode && !(fabsf(v[e]) < 65504.0f)) ovf |= 4; }
                }
                const float inv = 1.0f / l;
                const long row_g = (long)b * qs->len + r;
                for (int e = 0; e < Dh; ++e) {
                    const float val = o[e] * inv;
                    if (d->mode == SELFTOK_ATTN_F16X2 && d->o_blk[sg]) {
                        uint16_t* ob = (uint16_t*)d->o_blk[sg];
                        split1(val, &ob[split_blk_index(row_g, h * Dh + e, 0, d->H * Dh / 32)], &ob[split_blk_index(row_g, h * Dh + e, 1, d->H * Dh / 32)]);
                    } else if (qs->o) {
                        qs->o[(size_t)b * qs->o_bs + (size_t)r * qs->o_rs + h * Dh + e] = val;
                    }
                }
            }
            free(sc);
        }
    }
    if (ovf && d->overflow) *d->overflow |= ovf;
    return SELFTOK_OK;
}

/* ================================================================================================================================
 * bf16 epilogues around the VAE convolutions (csrc/vae.hip)
 * ============================================================================================================================== */
int selftok_groupnorm_silu_bf16(const void* xv, const void* wv, const void* bv, void* outv, int B, int C, int HW, int groups, float eps, int apply_silu, hipStream_t s)
{
    (void)s;
    if (!xv || !wv || !bv || !outv || B < 0 || groups <= 0 || C % groups || (HW & 7)) return fail("groupnorm_silu: need C%groups==0 and H*W%8==0");
    const uint16_t *x = (const uint16_t*)xv, *w = (const uint16_t*)wv, *bb = (const uint16_t*)bv;
    uint16_t* out = (uint16_t*)outv;
    const int cpg = C / groups;
    const long n = (long)cpg * HW;
#pragma omp parallel for schedule(static)
    for (long bg = 0; bg < (long)B * groups; ++bg) {
        const int g = (int)(bg % groups);
        const uint16_t* xs = x + bg * n;
        double s1 = 0.0, s2 = 0.0;
        for (long i = 0; i < n; ++i) s1 += bf2f(xs[i]);
        const float mean = (float)(s1 / (double)n);
        for (long i = 0; i < n; ++i) { const double dd = (double)bf2f(xs[i]) - (double)mean; s2 += dd * dd; }
        const float var = (float)(s2 / (double)n);
        const float rstd = 1.0f / sqrtf(var + eps);
        for (long i = 0; i < n; ++i) {
            const int ch = g * cpg + (int)(i / HW);
            float yv = rbf((bf2f(xs[i]) - mean) * rstd * bf2f(w[ch]) + bf2f(bb[ch]));
            if (apply_silu) yv = yv / (1.0f + expf(-yv));
            out[bg * n + i] = f2bf(yv);
        }
    }
    return SELFTOK_OK;
}
int selftok_latent_process_in(const void* mv, float* out, int B, int c_in, int c_keep, int HW, float shift, float scale, hipStream_t s)
{
    (void)s;
    if (!mv || !out || B < 0 || c_keep > c_in) return fail("latent_process_in: bad argument");
    const uint16_t* m = (const uint16_t*)mv;
    for (int b = 0; b < B; ++b) for (int c = 0; c < c_keep; ++c) for (int p = 0; p < HW; ++p) {
        const float z = bf2f(m[((size_t)b * c_in + c) * HW + p]);
        out[((size_t)b * c_keep + c) * HW + p] = rbf(rbf(z - rbf(shift)) * scale);
    }
    return SELFTOK_OK;
}
int selftok_latent_process_out(const float* z, void* outv, long n, float shift, float scale, hipStream_t s)
{
    (void)s;
    if (!z || !outv || n < 0) return fail("latent_process_out: bad argument");
    uint16_t* out = (uint16_t*)outv;
    for (long i = 0; i < n; ++i) out[i] = f2bf((z[i] / scale) + shift);
    return SELFTOK_OK;
}
int selftok_clamp01_bf16(void* imgv, long n, hipStream_t s)
{
    (void)s;
    if (!imgv || n < 0) return fail("clamp01: bad argument");
    uint16_t* img = (uint16_t*)imgv;
    for (long i = 0; i < n; ++i) {
        const float x0 = bf2f(img[i]);
        float v = (x0 != x0) ? x0 : fminf(fmaxf(x0, -1.0f), 1.0f);
        v = rbf(v - (-1.0f));
        img[i] = f2bf(v / 2.0f);
    }
    return SELFTOK_OK;
}

/* ================================================================================================================================
 * channels-last bf16 convolution and GroupNorm (csrc/conv.hip): plain loops, double accumulation (the GPU sums fp32 in matrix-core
 * order; both round once to bf16, so they agree except at rounding ties of the sum)
 * ============================================================================================================================== */
size_t selftok_conv2d_packed_bytes(int Cout, int Cin, int ksize, int bn)
{
    if (Cout <= 0 || Cin <= 0 || (ksize != 1 && ksize != 3) || (bn != 32 && bn != 128)) return 0;
    return (size_t)((Cout + bn - 1) / bn) * ((Cin + 31) / 32) * ksize * ksize * bn * 32 * sizeof(uint16_t);
}
static inline size_t conv_pk_index(int co, int c, int tap, int taps, int bn, int ncb)
{
    return ((((size_t)(co / bn) * ncb + c / 32) * taps + tap) * bn + co % bn) * 32 + c % 32;
}
int selftok_conv2d_pack_weight_bf16(const void* wv, void* pv, int Cout, int Cin, int ksize, int bn, hipStream_t s)
{
    (void)s;
    const size_t bytes = selftok_conv2d_packed_bytes(Cout, Cin, ksize, bn);
    if (!wv || !pv || bytes == 0) return fail("conv2d_pack_weight: bad argument (ksize 1|3, bn 32|128)");
    const uint16_t* w = (const uint16_t*)wv;
    uint16_t* p = (uint16_t*)pv;
    memset(p, 0, bytes);
    const int taps = ksize * ksize, ncb = (Cin + 31) / 32;
    for (int co = 0; co < Cout; ++co) for (int c = 0; c < Cin; ++c) for (int t = 0; t < taps; ++t)
        p[conv_pk_index(co, c, t, taps, bn, ncb)] = w[((size_t)co * Cin + c) * taps + t];
    return SELFTOK_OK;
}
int selftok_conv2d_nhwc_bf16(const void* xv, const void* pv, const void* bv, const void* rv, void* ov, int B, int H, int W, int Cin, int Cout,
                             int Cstore, int ldo, int ksize, int stride, int upsample, int bn, hipStream_t s)
{
    (void)s;
    if (!xv || !pv || !ov || B < 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 7) || Cout <= 0 || Cstore < Cout || (Cstore & 3) || ldo < Cstore || (ldo & 3) ||
        (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2) || (ksize == 1 && stride != 1) || (upsample != 0 && upsample != 1) || (bn != 32 && bn != 128) ||
        (upsample && stride != 1) || (stride == 2 && bn != 128))
        return fail("conv2d_nhwc_bf16: bad argument (Cin % 8, Cstore % 4, ldo % 4, ksize 1|3, stride 1|2 (3x3 only), bn 32|128)");
    const uint16_t *x = (const uint16_t*)xv, *pk = (const uint16_t*)pv, *bias = (const uint16_t*)bv, *res = (const uint16_t*)rv;
    uint16_t* out = (uint16_t*)ov;
    const int taps = ksize * ksize, ncb = (Cin + 31) / 32, Hi = H << upsample, Wi = W << upsample;
    const int Ho = stride == 2 ? Hi / 2 : Hi, Wo = stride == 2 ? Wi / 2 : Wi, pad = (stride == 1 && ksize == 3) ? 1 : 0;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) for (int oy = 0; oy < Ho; ++oy) for (int ox = 0; ox < Wo; ++ox) {
        const size_t pix = ((size_t)b * Ho + oy) * Wo + ox;
        for (int co = 0; co < Cstore; ++co) {
            double acc = 0.0;
            if (co < Cout) {
                for (int t = 0; t < taps; ++t) {
                    const int iy = oy * stride - pad + t / ksize, ix = ox * stride - pad + t % ksize;
                    if (iy < 0 || iy >= Hi || ix < 0 || ix >= Wi) continue;
                    const uint16_t* xp = x + (((size_t)b * H + (iy >> upsample)) * W + (ix >> upsample)) * Cin;
                    for (int c = 0; c < Cin; ++c) acc += (double)bf2f(xp[c]) * (double)bf2f(pk[conv_pk_index(co, c, t, taps, bn, ncb)]);
                }
                if (bias) acc += (double)bf2f(bias[co]);
            }
            float v = rbf((float)acc);
            if (res) v = rbf(v + bf2f(res[pix * ldo + co]));
            out[pix * ldo + co] = f2bf(v);
        }
    }
    return SELFTOK_OK;
}
size_t selftok_groupnorm_nhwc_workspace_bytes(int B, int HW, int C)
{
    if (B <= 0 || HW <= 0 || C <= 0) return 0;
    return (size_t)B * 32 * (C / 4) * 2 * sizeof(double) + (size_t)B * 64 * 2 * sizeof(float) + 256;
}
int selftok_groupnorm_silu_nhwc_bf16(const void* xv, const void* wv, const void* bv, void* ov, void* workspace, int B, int HW, int C, int groups,
                                     float eps, int apply_silu, hipStream_t s)
{
    (void)s;
    if (!xv || !wv || !bv || !ov || !workspace || B < 0 || HW <= 0 || groups <= 0 || groups > 64 || C % groups || ((C / groups) & 3) || (C & 7) || C > 2048 || (256 % (C >> 3)))
        return fail("groupnorm_silu_nhwc: need C % groups == 0, (C / groups) % 4 == 0, C / 8 a divisor of 256, groups <= 64");
    const uint16_t *x = (const uint16_t*)xv, *w = (const uint16_t*)wv, *bb = (const uint16_t*)bv;
    uint16_t* out = (uint16_t*)ov;
    const int cpg = C / groups;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        double s1[64] = {0}, s2[64] = {0};
        float meanf[64], rstd[64];
        for (int p = 0; p < HW; ++p) {                         /* one pass over the sample, pixel-major (the layout's order) */
            const uint16_t* xp = x + ((size_t)b * HW + p) * C;
            for (int c = 0; c < C; ++c) { const double d = bf2f(xp[c]); s1[c / cpg] += d; s2[c / cpg] += d * d; }
        }
        for (int g = 0; g < groups; ++g) {
            const double n = (double)HW * cpg, mean = s1[g] / n;
            double var = s2[g] / n - mean * mean;
            if (var < 0) var = 0;
            meanf[g] = (float)mean;
            const float varf = (float)(var + (mean - (double)meanf[g]) * (mean - (double)meanf[g]));
            rstd[g] = 1.0f / sqrtf(varf + eps);
        }
        for (int p = 0; p < HW; ++p) for (int c = 0; c < C; ++c) {
            const size_t i = ((size_t)b * HW + p) * C + c;
            float y = rbf((bf2f(x[i]) - meanf[c / cpg]) * rstd[c / cpg] * bf2f(w[c]) + bf2f(bb[c]));
            if (apply_silu) y = y / (1.0f + expf(-y));
            out[i] = f2bf(y);
        }
    }
    return SELFTOK_OK;
}

/* ---- the exact-order VAE encoder entries (include/selftok_hip.h, round 4): the CPU twin IS oracle/vae_exact.c -------------------- */
int vx_conv2d_nhwc(const uint16_t* xb, const uint16_t* wb, const uint16_t* bb, const uint16_t* rb, uint16_t* y, int B, int H, int W, int IC, int OC, int KH,
                   int KW, int stride, int pad, int OH, int OW, int order);
int vx_group_norm_nhwc(const uint16_t* x, const uint16_t* gamma, const uint16_t* beta, uint16_t* y, int B, int64_t HW, int C, int G, double eps,
                       const uint16_t* silu, float* stats);
int vx_attention(const uint16_t* qb, const uint16_t* kb, const uint16_t* vb, uint16_t* ob, int B, int T, int Cd);
float vx_expf(float x);

int selftok_vx_conv2d_bf16(const void* x, const void* w, const void* bias, const void* residual, void* out, int B, int H, int W, int ldx, int Cin, int Cout,
                           int ksize, int stride, int order, hipStream_t s)
{
    (void)s;
    if (B == 0) return SELFTOK_OK;
    if (!x || !w || !bias || !out || B < 0 || (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2) || (stride == 2 && ksize != 3)) return fail("vx_conv2d: bad argument");
    const int up = (order & SELFTOK_VX_UPSAMPLE2X) ? 1 : 0;
    order &= ~SELFTOK_VX_UPSAMPLE2X;
    if (up && (stride != 1 || ((H | W) & 1) || ldx != Cin)) return fail("vx_conv2d: SELFTOK_VX_UPSAMPLE2X needs stride 1 and even H, W");
    if (order == 2 ? (Cin != 3 || ksize != 3 || stride != 1 || residual) : ((order != 0 && order != 1 && order != 3) || Cin % 32 || ldx != Cin || Cout % 32))
        return fail("vx_conv2d: need Cin % 32 == 0, Cout % 32 == 0, order 0 / 1 / 2 / 3");
    const int OH = stride == 2 ? H / 2 : H, OW = stride == 2 ? W / 2 : W;
    const uint16_t* xs = (const uint16_t*)x;
    uint16_t* tmp = NULL;
    if (ldx != Cin) {                                   /* conv_in reads the first 3 of ldx channels */
        tmp = (uint16_t*)malloc((size_t)B * H * W * Cin * 2);
        for (size_t p = 0; p < (size_t)B * H * W; ++p) for (int c = 0; c < Cin; ++c) tmp[p * Cin + c] = xs[p * ldx + c];
        xs = tmp;
    }
    if (up) {                                           /* materialise the nearest-2x view (a copy: F.interpolate(mode="nearest")) */
        tmp = (uint16_t*)malloc((size_t)B * H * W * Cin * 2);
        if (!tmp) return fail("vx_conv2d: out of memory");
        for (int b = 0; b < B; ++b) for (int y = 0; y < H; ++y) for (int xx = 0; xx < W; ++xx)
            memcpy(tmp + (((size_t)b * H + y) * W + xx) * Cin, xs + (((size_t)b * (H / 2) + y / 2) * (W / 2) + xx / 2) * Cin, (size_t)Cin * 2);
        xs = tmp;
    }
    const int rc = vx_conv2d_nhwc(xs, (const uint16_t*)w, (const uint16_t*)bias, (const uint16_t*)residual, (uint16_t*)out, B, H, W, Cin, Cout, ksize, ksize, stride,
                                  (ksize == 3 && stride == 1) ? 1 : 0, OH, OW, order);
    free(tmp);
    return rc ? fail("vx_conv2d: out of memory") : SELFTOK_OK;
}
size_t selftok_vx_groupnorm_workspace_bytes(int B, int HW, int C)
{
    if (B <= 0 || HW <= 0 || HW % 16 || C % 128) return 0;
    /* unused here; the HIP build's figure (csrc/vae_exact.hip xgn_plan): nodes of 16 / 4 / 2 aligned chunks, else raw half-moments per chunk */
    int nch = 0;
    if (HW % 256 == 0) { const int nc = HW / 256; nch = nc % 16 == 0 ? 16 : (nc % 4 == 0 ? 4 : (nc % 2 == 0 ? 2 : 0)); }
    const size_t moms = nch ? (size_t)B * C * (HW / (256 * nch)) * 8 : ((size_t)B * C * HW / 256 + (size_t)B * C) * 16;
    return moms * 8 + (size_t)2 * B * C * sizeof(float);
}
int selftok_vx_groupnorm_bf16(const void* x, const void* gamma, const void* beta, void* out, void* workspace, const void* silu_table, float* stats, int B, int HW,
                              int C, int groups, double eps, hipStream_t s)
{
    (void)s; (void)workspace;
    if (B == 0) return SELFTOK_OK;
    if (!x || !gamma || !beta || !out || B < 0 || groups <= 0 || C % groups || C % 128 || HW <= 0 || HW % 16 || ((C / groups) & (C / groups - 1)))
        return fail("vx_groupnorm: need C % 128 == 0, H*W % 16 == 0, power-of-two channels per group");
    vx_group_norm_nhwc((const uint16_t*)x, (const uint16_t*)gamma, (const uint16_t*)beta, (uint16_t*)out, B, HW, C, groups, eps, (const uint16_t*)silu_table, stats);
    return SELFTOK_OK;
}
int selftok_vx_silu_table_bf16(void* table, hipStream_t s)
{
    (void)s;
    if (!table) return fail("vx_silu_table: null");
    uint16_t* t = (uint16_t*)table;
    for (int i = 0; i < 65536; ++i) {
        const float x = bf2f((uint16_t)i);
        if (x != x) t[i] = (uint16_t)(i | 0x40);
        else if (-x > 88.72284f) t[i] = f2bf(x / INFINITY);
        else { const double xd = (double)x; t[i] = f2bf((float)(xd / (1.0 + exp(-xd)))); }
    }
    return SELFTOK_OK;
}
size_t selftok_vx_attention_workspace_bytes(int B, int T, int C)
{
    if (B <= 0 || T <= 0) return 0;
    return (size_t)B * T * T * 4 + (size_t)B * T * T * 2 + (size_t)B * T * C * 2 + ((size_t)(T + 511) / 512 + 1) * B * T * 4 + 512;
}
int selftok_vx_attention_bf16(const void* q, const void* k, const void* v, void* out, void* workspace, int B, int T, int C, hipStream_t s)
{
    (void)s; (void)workspace;
    if (B == 0) return SELFTOK_OK;
    if (!q || !k || !v || !out || B < 0 || T <= 0 || T % 32 || C % 128 || C > 4096) return fail("vx_attention: one head, T % 32 == 0, C % 128 == 0");
    return vx_attention((const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, (uint16_t*)out, B, T, C) ? fail("vx_attention") : SELFTOK_OK;
}
int selftok_vx_expf_f32(const float* x, float* y, long n, hipStream_t s)
{
    (void)s;
    if (!x || !y || n < 0) return fail("vx_expf: bad argument");
    for (long i = 0; i < n; ++i) y[i] = vx_expf(x[i]);
    return SELFTOK_OK;
}

/* ---- the exact-order Q-Former encoder entries (include/selftok_hip.h, round 5): the CPU twin IS oracle/encoder_exact.c ------------ */
void xe_linear(const float* x, const float* w, const float* bias, float* out, long M, int N, int K);
void xe_layernorm(const float* X, float* Y, const float* gamma, const float* beta, long rows, int N, float eps, float* stats);
void xe_attention_masked(const float* Q, long qs, const float* K1, const float* V1, long kvs1, int Tk1, int valid1, int rows1, const float* K2, const float* V2, long kvs2,
                         int Tk2, float* O, int B, int H, int Tq, int D);
float xe_gelu_tanh1(float v);
float xe_silu1(float v);
float xe_sleef_expf(float v);
float xe_sleef_tanhf(float v);
float xe_exp_u20(float v);

int selftok_ex_linear_f32(const float* x, long ldx, const float* w, const float* bias, const float* res, long ldr, int res_mod, const float* gate, long ldg,
                          int gate_mod, float* out, long ldo, long M, int N, int K, int gelu, hipStream_t s)
{
    (void)s;
    if (M == 0) return SELFTOK_OK;
    if (!x || !w || !out || M < 0 || N <= 0 || K <= 0 || K % 16 || ldx % 4 || ldx < K || ldo < N || (gate && !res)) return fail("ex_linear: bad argument");
    float* xc = (float*)malloc((size_t)M * K * sizeof(float));
    float* yc = (float*)malloc((size_t)M * N * sizeof(float));
    if (!xc || !yc) { free(xc); free(yc); return fail("ex_linear: out of memory"); }
    for (long m = 0; m < M; ++m) memcpy(xc + (size_t)m * K, x + (size_t)m * ldx, (size_t)K * sizeof(float));
    xe_linear(xc, w, (gelu & 2) ? NULL : bias, yc, M, N, K);
    for (long m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float v = yc[(size_t)m * N + n];
            if ((gelu & 2) && bias) v = v + bias[n];
            if (gelu & 1) v = xe_gelu_tanh1(v);
            if (gate) v = gate[(size_t)(gate_mod > 0 ? m % gate_mod : (gate_mod < 0 ? m / -gate_mod : m)) * ldg + n] * v;
            if (res) v = res[(size_t)(res_mod > 0 ? m % res_mod : (res_mod < 0 ? m / -res_mod : m)) * ldr + n] + v;
            out[(size_t)m * ldo + n] = v;
        }
    free(xc); free(yc);
    return SELFTOK_OK;
}

/* csrc/gemm_fp32.hip on the CPU.  MKL order: the arithmetic of selftok_ex_linear_f32 (K <= 384 or K >= 768).  Free order: one k-ascending fmaf chain per output
 * over the whole K, bias added last -- what the GPU kernel computes for every tile of a full round; its tail tiles are S shorter chains added in order (S is a
 * property of the launch plan on a 256-CU chip), so the two builds agree to accumulation-order noise there, not bit for bit. */
size_t selftok_linear_f32_workspace_bytes(long M, int N, int K, int flags)
{
    (void)M; (void)N; (void)K; (void)flags;
    return 0;
}

int selftok_linear_f32(const float* x, long ldx, const float* w, const float* bias, const float* res, long ldr, int res_mod, const float* gate, long ldg,
                       int gate_mod, float* out, long ldo, long M, int N, int K, int flags, void* workspace, size_t workspace_bytes, hipStream_t s)
{
    (void)workspace; (void)workspace_bytes;
    if (M == 0) return SELFTOK_OK;
    if (!x || !w || !out || M < 0 || N <= 0 || K <= 0 || N % 128 || K % 32 || ldx % 4 || ldx < K || ldo < N || ldo % 4 || (gate && !res)) return fail("linear_f32: bad argument");
    if ((flags & SELFTOK_LINEAR_GELU) && (res || gate || ldo != N)) return fail("linear_f32: SELFTOK_LINEAR_GELU needs a contiguous out and no res / gate");
    if (flags & SELFTOK_LINEAR_MKL_ORDER) {
        if (K > 384 && K < 768) return fail("linear_f32: MKL order for 384 < K < 768 is served by selftok_ex_linear_f32");
        return selftok_ex_linear_f32(x, ldx, w, bias, res, ldr, res_mod, gate, ldg, gate_mod, out, ldo, M, N, K,
                                     ((flags & SELFTOK_LINEAR_GELU) ? 1 : 0) | ((flags & SELFTOK_LINEAR_BIAS_LAST) ? 2 : 0), s);
    }
#pragma omp parallel for schedule(static)
    for (long m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            const float *xr = x + (size_t)m * ldx, *wr = w + (size_t)n * K;
            float v = 0.0f;
            for (int k = 0; k < K; ++k) v = fmaf(xr[k], wr[k], v);
            if (bias) v = v + bias[n];
            if (flags & SELFTOK_LINEAR_GELU) v = xe_gelu_tanh1(v);
            if (gate) v = gate[(size_t)(gate_mod > 0 ? m % gate_mod : (gate_mod < 0 ? m / -gate_mod : m)) * ldg + n] * v;
            if (res) v = res[(size_t)(res_mod > 0 ? m % res_mod : (res_mod < 0 ? m / -res_mod : m)) * ldr + n] + v;
            out[(size_t)m * ldo + n] = v;
        }
    return SELFTOK_OK;
}

int selftok_ex_layernorm_mod_f32(const float* x, long ldx, float* out, long ldo, const float* shift, const float* scale, long ldt, int T, const float* gamma,
                                 const float* beta, float* stats, long rows, int N, float eps, hipStream_t s)
{
    (void)s;
    if (rows == 0) return SELFTOK_OK;
    if (!x || !out || rows < 0 || N <= 0 || N % 8 || N > 4096 || ldx % 4 || ldo % 4 || ((shift == NULL) != (scale == NULL)) || (scale && T == 0))
        return fail("ex_layernorm: bad argument");
    for (long r = 0; r < rows; ++r) {
        xe_layernorm(x + (size_t)r * ldx, out + (size_t)r * ldo, gamma, beta, 1, N, eps, stats ? stats + 2 * r : NULL);
        if (scale) {
            const long tok = T > 0 ? r % T : r / -T;                  /* T < 0: per-sample tables, -T rows per sample */
            const float *sc = scale + (size_t)tok * ldt, *sh = shift + (size_t)tok * ldt;
            float* o = out + (size_t)r * ldo;
            for (int j = 0; j < N; ++j) o[j] = o[j] * (1.0f + sc[j]) + sh[j];
        }
    }
    return SELFTOK_OK;
}

int selftok_ex_res_layernorm_mod_f32(const float* x, long ldx, const float* lin, long ldl, const float* lin_bias, const float* gate, long ldg, int gate_mod,
                                     float* x_out, long ldxo, float* out, long ldo, const float* shift, const float* scale, long ldt, int T, long rows, int N, float eps,
                                     hipStream_t s)
{
    if (rows == 0) return SELFTOK_OK;
    if (!x || !lin || !x_out || !out || rows < 0 || N <= 0 || N % 8 || N > 4096 || ldx % 4 || ldl % 4 || ldxo % 4 || ldo % 4 || ((shift == NULL) != (scale == NULL)) ||
        (scale && T == 0)) return fail("ex_res_layernorm: bad argument");
    for (long r = 0; r < rows; ++r) {
        const float* g = gate ? gate + (size_t)(gate_mod > 0 ? r % gate_mod : (gate_mod < 0 ? r / -gate_mod : r)) * ldg : NULL;
        for (int n = 0; n < N; ++n) {
            float v = lin[(size_t)r * ldl + n];
            if (lin_bias) v = v + lin_bias[n];
            if (g) v = g[n] * v;
            x_out[(size_t)r * ldxo + n] = x[(size_t)r * ldx + n] + v;
        }
    }
    return selftok_ex_layernorm_mod_f32(x_out, ldxo, out, ldo, shift, scale, ldt, T, NULL, NULL, NULL, rows, N, eps, s);
}

int selftok_ex_unary_f32(const float* x, float* y, long n, int mode, hipStream_t s)
{
    (void)s;
    if (n == 0) return SELFTOK_OK;
    if (!x || !y || n < 0 || mode < 0 || mode > 4) return fail("ex_unary: bad argument");
    for (long i = 0; i < n; ++i)
        y[i] = mode == 0 ? xe_gelu_tanh1(x[i]) : mode == 1 ? xe_silu1(x[i]) : mode == 2 ? xe_sleef_expf(x[i]) : mode == 3 ? xe_sleef_tanhf(x[i]) : xe_exp_u20(x[i]);
    return SELFTOK_OK;
}

size_t selftok_ex_attention_workspace_bytes(int B, int H, int Tq, int Tk, int D)
{
    if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0 || D <= 0) return 0;
    const size_t rows = (size_t)B * H * Tq;
    const int nb = (Tk + 511) / 512;
    return rows * Tk * 4 + (size_t)B * H * D * Tk * 4 + rows * 4 * (size_t)(nb > 1 ? nb - 1 : 1) + rows * 4;      /* the GPU library's size */
}

int selftok_ex_attention_f32(const float* q, long qs, const float* k1, const float* v1, long kvs1, int Tk1, int valid1, int rows1, const float* k2, const float* v2,
                             long kvs2, int Tk2, float* out, void* workspace, int B, int H, int Tq, int D, hipStream_t s)
{
    (void)s; (void)workspace;
    if (B == 0) return SELFTOK_OK;
    if (!q || !out || B < 0 || H <= 0 || Tq <= 0 || Tk1 <= 0 || Tk2 < 0 || valid1 < 0 || valid1 > Tk1 || rows1 < valid1 || (valid1 > 0 && (!k1 || !v1)) ||
        (Tk2 > 0 && (!k2 || !v2)) || D % 16 || D <= 0 || D > 128 || Tk1 % 16 || Tk2 % 16 || qs % 4 || kvs1 % 4 || kvs2 % 4 || (valid1 == 0 && Tk2 == 0))
        return fail("ex_attention: bad argument");
    xe_attention_masked(q, qs, k1, v1, kvs1, Tk1, valid1, rows1, k2, v2, kvs2, Tk2, out, B, H, Tq, D);
    return SELFTOK_OK;
}

/* the fused entry of round 6: the same arithmetic (the CPU twin has one implementation), the GPU entry's shape restrictions */
int selftok_ex_attention_fused_supported(int Tk1, int Tk2, int D)
{
    const int Tk = Tk1 + Tk2, last = Tk & 511;
    return D == 64 && Tk1 >= 0 && Tk2 >= 0 && Tk > 0 && (Tk1 & 63) == 0 && (Tk2 & 63) == 0 && (last == 0 || last <= 384);
}

int selftok_ex_attention_fused_f32(const float* q, long qs, const float* k1, const float* v1, long kvs1, int Tk1, int valid1, int rows1, const float* k2, const float* v2,
                                   long kvs2, int Tk2, float* out, int B, int H, int Tq, int D, hipStream_t s)
{
    if (B == 0) return SELFTOK_OK;
    if (!selftok_ex_attention_fused_supported(Tk1, Tk2, D) || Tk1 <= 0) return fail("ex_attention_fused: unsupported shape");
    return selftok_ex_attention_f32(q, qs, k1, v1, kvs1, Tk1, valid1, rows1, k2, v2, kvs2, Tk2, out, NULL, B, H, Tq, D, s);
}
