// HBM-bound fused epilogue kernels of the Selftok encoder / MMDiT decode loop (gfx950).
//
// All activations are fp32 [rows, H] (rows = B*T, H in {64, 512, 1536}); every kernel makes exactly
// one pass over its tensors with 16-byte accesses, one wave (or a 16-lane group for H=64) per row,
// the row held in registers between the statistics and the normalise step.
//
// Modulation operands (shift / scale / gate) are addressed as  ptr + b*stride_b + t*stride_t + col :
//   per-token table  [T, 6H]  (context stream 'pos_emb', encoder queries):  stride_b = 0,  stride_t = 6H
//   per-sample table [B, 6H]  (image stream 't_emb'):                        stride_b = 6H, stride_t = 0
// so the reference's [B,T,H] broadcasts (modules.py:29-37, sd3/mmdit.py:78-83) are never materialised.
#include "common.h"
#include "selftok_hip.h"   // the C ABI declared there must match the definitions below
#include <stdlib.h>

namespace selftok {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

template <int G>
__device__ __forceinline__ float group_sum(float v)
{
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// ---------------------------------------------------------------------------------------
// fused  x' = x + gate*y ;  n = LN(x') * (1+scale) + shift
//   y == nullptr      -> no residual update (x' = x, x_out not written)
//   n_out == nullptr  -> residual only
//   shift == nullptr  -> plain LayerNorm (no affine, eps)           [modules.py:104-106]
//   gate == nullptr   -> x' = x + y                                  [modules.py:322-323]
// reference: DualBlock.forward modules.py:321-326 ; DismantledBlock.pre/post_attention sd3/mmdit.py:472-495 ;
//            FinalLayer.forward sd3/mmdit.py:641-645.
// G lanes cooperate on one row, each holding VPL float4.  H == G*VPL*4.
// ---------------------------------------------------------------------------------------
template <int G, int VPL>
__global__ __launch_bounds__(256) void residual_ln_mod_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gate,
    const float* __restrict__ shift, const float* __restrict__ scale,
    float* __restrict__ x_out, float* __restrict__ n_out,
    int rows, int T, long mod_stride_b, long mod_stride_t, long gate_stride_b, long gate_stride_t, float eps,
    _Float16* __restrict__ n_blk = nullptr, int* __restrict__ overflow = nullptr)
{
    constexpr int H = G * VPL * 4;
    const int gid = (blockIdx.x * blockDim.x + threadIdx.x) / G;   // row
    const int gl = threadIdx.x % G;
    if (gid >= rows) return;
    const int b = gid / T, t = gid - b * T;
    const float* xr = x + (size_t)gid * H;
    float4 v[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) v[i] = ld4(xr + (i * G + gl) * 4);
    if (y) {
        const float* yr = y + (size_t)gid * H;
        const float* gr = gate ? gate + b * gate_stride_b + t * gate_stride_t : nullptr;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            float4 yy = ld4(yr + (i * G + gl) * 4);
            if (gr) {
                float4 g = ld4(gr + (i * G + gl) * 4);
                v[i].x = v[i].x + g.x * yy.x; v[i].y = v[i].y + g.y * yy.y;
                v[i].z = v[i].z + g.z * yy.z; v[i].w = v[i].w + g.w * yy.w;
            } else {
                v[i].x += yy.x; v[i].y += yy.y; v[i].z += yy.z; v[i].w += yy.w;
            }
        }
        if (x_out) {
            float* xo = x_out + (size_t)gid * H;
#pragma unroll
            for (int i = 0; i < VPL; ++i) st4(xo + (i * G + gl) * 4, v[i]);
        }
    }
    if (!n_out && !n_blk) return;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = group_sum<G>(s) * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        float a = v[i].x - mean, bq = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + bq * bq) + (c * c + d * d);
    }
    const float var = group_sum<G>(q) * (1.0f / H);
    const float rstd = 1.0f / __builtin_sqrtf(var + eps);
    float* nr = n_out ? n_out + (size_t)gid * H : nullptr;
    float mx = 0.f;
    const float* sh = shift ? shift + b * mod_stride_b + t * mod_stride_t : nullptr;
    const float* sc = scale ? scale + b * mod_stride_b + t * mod_stride_t : nullptr;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        float4 o;
        o.x = (v[i].x - mean) * rstd; o.y = (v[i].y - mean) * rstd;
        o.z = (v[i].z - mean) * rstd; o.w = (v[i].w - mean) * rstd;
        if (sc) {
            float4 c4 = ld4(sc + (i * G + gl) * 4), s4 = ld4(sh + (i * G + gl) * 4);
            o.x = o.x * (1.0f + c4.x) + s4.x; o.y = o.y * (1.0f + c4.y) + s4.y;
            o.z = o.z * (1.0f + c4.z) + s4.z; o.w = o.w * (1.0f + c4.w) + s4.w;
        }
        if (nr) st4(nr + (i * G + gl) * 4, o);
        if (n_blk) {     // "split activation" for the f16x2 Linear that consumes n (gemm_split.hip): hi = fp16(n), lo = fp16((n - hi) 2^11)
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 ov = {opaque_f32(o.x), opaque_f32(o.y), opaque_f32(o.z), opaque_f32(o.w)};   // see common.h
            const h4 hh = __builtin_convertvector(ov, h4);
            const h4 ll = __builtin_convertvector((ov - __builtin_convertvector(hh, f4)) * 2048.0f, h4);
            *reinterpret_cast<h4*>(n_blk + split_blk_index(gid, (i * G + gl) * 4, 0, H / 32)) = hh;
            *reinterpret_cast<h4*>(n_blk + split_blk_index(gid, (i * G + gl) * 4, 1, H / 32)) = ll;
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
        }
    }
    if (n_blk && overflow && !(mx < 65504.0f)) atomicOr(overflow, 1);
}

// ---------------------------------------------------------------------------------------
// The same fused pass for the wide rows of the MMDiT / query streams (H = 64 * VPL * 4: one wave per row), re-shaped around what
// bounds it -- HBM -- (measurements: profiles/r3_ln_variants.txt, r3_hbm_sweep.txt):
//   * a wave WALKS R rows along the direction in which its modulation operands do not change: per-token tables [T, 6H] (context
//     stream, encoder queries) -> the same token of R consecutive samples (row stride T); per-sample tables [B, 6H] (image stream)
//     -> R consecutive tokens of one sample.  shift / scale (HM) and gate (HG), when they are constant along the walk, are read
//     ONCE per wave and stay in registers: in the one-row-per-wave kernel above they are 18 KB of L2 / Infinity-Cache reads next to
//     24 KB of row data (+14 ... 20 % at R = 4);
//   * NT: non-temporal stores for x_out / n_out / the split planes.  Both outputs are consumed by a GEMM that starts after this
//     kernel has finished; written with the default policy they evict the rows still to be read from L2 / Infinity Cache
//     (2-read + 2-write stream of 4 x 141 MB: 5.4 TB/s with plain stores, 7.8 TB/s non-temporal, tools/microbench/hbm_sweep.hip).
// A register-prefetched variant (two row sets per wave, the next row's loads issued before the reduction) was measured and dropped:
// 256 + 86 registers at H = 1536 leave one wave per SIMD and it is 10 - 25 % slower than letting 2 - 4 waves per SIMD overlap.
// Arithmetic and its order are exactly those of residual_ln_mod_kernel: results are bit-identical.
// ---------------------------------------------------------------------------------------
template <bool NT, typename V>
__device__ __forceinline__ void stx(V* p, V v)
{
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

template <int VPL, bool NT, bool HM, bool HG>
__global__ __launch_bounds__(256) void residual_ln_mod_walk_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gate,
    const float* __restrict__ shift, const float* __restrict__ scale,
    float* __restrict__ x_out, float* __restrict__ n_out,
    int B, int T, long mod_stride_b, long mod_stride_t, long gate_stride_b, long gate_stride_t, float eps,
    _Float16* __restrict__ n_blk, int* __restrict__ overflow,
    int walk_tokens /* 0: R samples of one token (row stride T), 1: R tokens of one sample (row stride 1) */, int R)
{
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    constexpr int H = 64 * VPL * 4;
    const int lane = threadIdx.x & 63;
    const long wave_g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    int b0, t0, cnt;
    long step;
    if (!walk_tokens) {
        const long c = wave_g / T;
        t0 = (int)(wave_g - c * T);
        b0 = (int)c * R;
        if (b0 >= B) return;
        cnt = B - b0 < R ? B - b0 : R;
        step = T;
    } else {
        const int nch = (T + R - 1) / R;
        b0 = (int)(wave_g / nch);
        if (b0 >= B) return;
        t0 = (int)(wave_g - (long)b0 * nch) * R;
        cnt = T - t0 < R ? T - t0 : R;
        step = 1;
    }
    const long row0 = (long)b0 * T + t0;
    float4 hs[HM ? VPL : 1], hc[HM ? VPL : 1], hg[HG ? VPL : 1];
    if (HM) {
        const float* sh = shift + b0 * mod_stride_b + t0 * mod_stride_t;
        const float* sc = scale + b0 * mod_stride_b + t0 * mod_stride_t;
#pragma unroll
        for (int i = 0; i < VPL; ++i) { hs[i] = ld4(sh + (i * 64 + lane) * 4); hc[i] = ld4(sc + (i * 64 + lane) * 4); }
    }
    if (HG) {
        const float* gr = gate + b0 * gate_stride_b + t0 * gate_stride_t;
#pragma unroll
        for (int i = 0; i < VPL; ++i) hg[i] = ld4(gr + (i * 64 + lane) * 4);
    }
    float mx = 0.f;
#pragma unroll 1
    for (int j = 0; j < cnt; ++j) {
        const long row = row0 + j * step;
        const int b = walk_tokens ? b0 : b0 + j, t = walk_tokens ? t0 + j : t0;
        float4 v[VPL];
        const float* xr = x + (size_t)row * H;
#pragma unroll
        for (int i = 0; i < VPL; ++i) v[i] = ld4(xr + (i * 64 + lane) * 4);
        if (y) {
            const float* yr = y + (size_t)row * H;
            const float* gr = (gate && !HG) ? gate + b * gate_stride_b + t * gate_stride_t : nullptr;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const float4 yy = ld4(yr + (i * 64 + lane) * 4);
                if (HG || gr) {
                    const float4 g = HG ? hg[i] : ld4(gr + (i * 64 + lane) * 4);
                    v[i].x = v[i].x + g.x * yy.x; v[i].y = v[i].y + g.y * yy.y;
                    v[i].z = v[i].z + g.z * yy.z; v[i].w = v[i].w + g.w * yy.w;
                } else {
                    v[i].x += yy.x; v[i].y += yy.y; v[i].z += yy.z; v[i].w += yy.w;
                }
            }
            if (x_out) {
                float* xo = x_out + (size_t)row * H;
#pragma unroll
                for (int i = 0; i < VPL; ++i) stx<NT>(reinterpret_cast<f4v*>(xo + (i * 64 + lane) * 4), f4v{v[i].x, v[i].y, v[i].z, v[i].w});
            }
        }
        if (!n_out && !n_blk) continue;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        const float mean = group_sum<64>(s) * (1.0f / H);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            float a = v[i].x - mean, bq = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + bq * bq) + (c * c + d * d);
        }
        const float var = group_sum<64>(q) * (1.0f / H);
        const float rstd = 1.0f / __builtin_sqrtf(var + eps);
        float* nr = n_out ? n_out + (size_t)row * H : nullptr;
        const float* sh = (scale && !HM) ? shift + b * mod_stride_b + t * mod_stride_t : nullptr;
        const float* sc = (scale && !HM) ? scale + b * mod_stride_b + t * mod_stride_t : nullptr;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            float4 o;
            o.x = (v[i].x - mean) * rstd; o.y = (v[i].y - mean) * rstd;
            o.z = (v[i].z - mean) * rstd; o.w = (v[i].w - mean) * rstd;
            if (HM || sc) {
                const float4 c4 = HM ? hc[i] : ld4(sc + (i * 64 + lane) * 4), s4 = HM ? hs[i] : ld4(sh + (i * 64 + lane) * 4);
                o.x = o.x * (1.0f + c4.x) + s4.x; o.y = o.y * (1.0f + c4.y) + s4.y;
                o.z = o.z * (1.0f + c4.z) + s4.z; o.w = o.w * (1.0f + c4.w) + s4.w;
            }
            if (nr) stx<NT>(reinterpret_cast<f4v*>(nr + (i * 64 + lane) * 4), f4v{o.x, o.y, o.z, o.w});
            if (n_blk) {
                const f4v ov = {opaque_f32(o.x), opaque_f32(o.y), opaque_f32(o.z), opaque_f32(o.w)};   // see common.h
                const h4 hh = __builtin_convertvector(ov, h4);
                const h4 ll = __builtin_convertvector((ov - __builtin_convertvector(hh, f4v)) * 2048.0f, h4);
                stx<NT>(reinterpret_cast<h4*>(n_blk + split_blk_index(row, (i * 64 + lane) * 4, 0, H / 32)), hh);
                stx<NT>(reinterpret_cast<h4*>(n_blk + split_blk_index(row, (i * 64 + lane) * 4, 1, H / 32)), ll);
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
            }
        }
    }
    if (n_blk && overflow && !(mx < 65504.0f)) atomicOr(overflow, 1);
}

// in-place  h = gelu_tanh(h + bias)   (timm / sd3 Mlp act: modules.py:109,293 ; sd3/other_impls.py:82-90)
// torch: 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715 x^3)))
__device__ __forceinline__ float gelu_tanh(float x)
{
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float inner = k0 * (x + k1 * x * x * x);
    return 0.5f * x * (1.0f + tanhf(inner));
}

__global__ __launch_bounds__(256) void bias_gelu_kernel(float* __restrict__ h, const float* __restrict__ bias, long n4, int cols4)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        float4 v = ld4(h + i * 4);
        if (bias) {
            float4 bb = ld4(bias + (i % cols4) * 4);
            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
        }
        v.x = gelu_tanh(v.x); v.y = gelu_tanh(v.y); v.z = gelu_tanh(v.z); v.w = gelu_tanh(v.w);
        st4(h + i * 4, v);
    }
}

// in-place  h = silu(h)  (adaLN_modulation[0], TimestepEmbedder.mlp[1]: modules.py:297 ; sd3/mmdit.py:428,151)
__global__ __launch_bounds__(256) void silu_kernel(const float* __restrict__ in, float* __restrict__ out, long n)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float v = in[i];
        out[i] = v / (1.0f + expf(-v));
    }
}

// out[b,t,:] = in[b,t,:] + table[t,:]   (patch-embed bias + cropped pos-embed, context pos-embed:
// models_ours.py:211-214 ; sd3/mmdit.py:1000,1026).  In-place allowed.
__global__ __launch_bounds__(256) void add_rows_kernel(const float* __restrict__ in, const float* __restrict__ table,
                                                       float* __restrict__ out, long n4, long per_sample4)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        float4 v = ld4(in + i * 4), tt = ld4(table + (i % per_sample4) * 4);
        v.x += tt.x; v.y += tt.y; v.z += tt.z; v.w += tt.w;
        st4(out + i * 4, v);
    }
}

// sinusoidal timestep embedding  out[n, 2*half] = [cos(t f_i), sin(t f_i)]
// (TimestepEmbedder.timestep_embedding: models.py:56-74 ; sd3/mmdit.py:156-175).  freqs[half] come from
// the host (computed exactly as torch-CPU does) so that only cos/sin are evaluated on device.
__global__ void timestep_embed_kernel(const float* __restrict__ t, const float* __restrict__ freqs,
                                      float* __restrict__ out, int n, int half, float t_scale)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * half) return;
    int r = i / half, c = i - r * half;
    float a = (t[r] * t_scale) * freqs[c];
    out[(size_t)r * 2 * half + c] = cosf(a);
    out[(size_t)r * 2 * half + half + c] = sinf(a);
}

// patchify for the k=2,s=2 PatchEmbed conv (sd3/mmdit.py:66-75) so that it becomes one GEMM:
// x [B,C,Hh,Ww] -> patches [B, (Hh/2)*(Ww/2), C*4], feature index = c*4 + p*2 + q  (== conv weight.view(O,-1))
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                       int B, int C, int Hh, int Ww)
{
    const int hp = Hh / 2, wp = Ww / 2;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (b, token, c): writes 4 floats
    long total = (long)B * hp * wp * C;
    if (i >= total) return;
    int c = i % C;
    long r = i / C;
    int w = r % wp; r /= wp;
    int h = r % hp;
    int b = r / hp;
    const float* src = x + (((size_t)b * C + c) * Hh + 2 * h) * Ww + 2 * w;
    float2 r0 = *reinterpret_cast<const float2*>(src);
    float2 r1 = *reinterpret_cast<const float2*>(src + Ww);
    st4(out + (((size_t)b * hp + h) * wp + w) * (C * 4) + c * 4, make_float4(r0.x, r0.y, r1.x, r1.y));
}

// fused unpatchify ('nhwpqc->nchpwq', sd3/mmdit.py:898-916) + CFG mix (rectified_flow.py:289) +
// Euler step x_prev = x - (a_t - a_prev) * v (rectified_flow.py:301-304).
// y_cond/y_uncond [B, hp*wp, 4*C] (FinalLayer output, feature = (p*2+q)*C + c); x, x_out [B,C,2hp,2wp].
// y_uncond == nullptr -> v = y_cond.  dt = a_t - a_prev is computed by the caller in fp32.
__global__ __launch_bounds__(256) void unpatchify_euler_kernel(const float* __restrict__ y_cond, const float* __restrict__ y_uncond,
                                                               const float* __restrict__ x, float* __restrict__ x_out, float* __restrict__ v_out,
                                                               int B, int C, int hp, int wp, float dt, float cfg_scale)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per output pixel pair (q = 0,1)
    const int Hh = 2 * hp, Ww = 2 * wp;
    long total = (long)B * C * Hh * wp;
    if (i >= total) return;
    int w = i % wp;
    long r = i / wp;
    int row = r % Hh; r /= Hh;
    int c = r % C;
    int b = r / C;
    int h = row >> 1, p = row & 1;
    size_t tok = ((size_t)b * hp + h) * wp + w;
    const float* yc = y_cond + tok * (4 * C) + (p * 2) * C + c;
    float v0 = yc[0], v1 = yc[C];
    if (y_uncond) {
        const float* yu = y_uncond + tok * (4 * C) + (p * 2) * C + c;
        float u0 = yu[0], u1 = yu[C];
        v0 = u0 + cfg_scale * (v0 - u0);
        v1 = u1 + cfg_scale * (v1 - u1);
    }
    size_t o = (((size_t)b * C + c) * Hh + row) * Ww + 2 * w;
    if (v_out) *reinterpret_cast<float2*>(v_out + o) = make_float2(v0, v1);
    if (x_out) {
        float2 xx = *reinterpret_cast<const float2*>(x + o);
        *reinterpret_cast<float2*>(x_out + o) = make_float2(xx.x - dt * v0, xx.y - dt * v1);
    }
}

// RMSNorm over the last dim (modules.py:73-95): x * rsqrt(mean(x^2)+eps) * w.  Only active with
// qk_norm='rms' (no shipped config); one 16-lane group per row of `dim` <= 256 floats.
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      float* __restrict__ out, long rows, int dim, float eps)
{
    const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) / 16;
    const int gl = threadIdx.x % 16;
    if (row >= rows) return;
    const float* xr = x + row * dim;
    float s = 0.f;
    for (int c = gl; c < dim; c += 16) s += xr[c] * xr[c];
    s = group_sum<16>(s);
    const float r = rsqrtf(s / dim + eps);
    for (int c = gl; c < dim; c += 16) out[row * dim + c] = xr[c] * r * (w ? w[c] : 1.0f);
}

// rotary embedding (utils/rotary_embedding_torch.py:37-53): out = (t*cos(f))*scale + (rotate_half(t)*sin(f))*scale,
// rotate_half on interleaved pairs (x1,x2) -> (-x2,x1).  Off the executed path (no call site in the reference).
__global__ void rotary_kernel(const float* __restrict__ t, const float* __restrict__ freqs, float* __restrict__ out,
                              long rows, int seq, int dim, float scale)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per pair
    long total = rows * (dim / 2);
    if (i >= total) return;
    long row = i / (dim / 2);
    int pr = i % (dim / 2);
    int pos = row % seq;
    const float* f = freqs + (size_t)pos * dim;
    float x1 = t[row * dim + 2 * pr], x2 = t[row * dim + 2 * pr + 1];
    float f1 = f[2 * pr], f2 = f[2 * pr + 1];
    out[row * dim + 2 * pr] = x1 * cosf(f1) * scale + (-x2) * sinf(f1) * scale;
    out[row * dim + 2 * pr + 1] = x2 * cosf(f2) * scale + x1 * sinf(f2) * scale;
}

}  // namespace selftok

using namespace selftok;

static inline int grid_for(long n, int block = 256, int cap = 256 * 16)
{
    long g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" {

static int residual_ln_mod_launch(const float* x, const float* y, const float* gate, const float* shift, const float* scale,
                                  float* x_out, float* n_out, _Float16* n_blk, int* overflow, int B, int T, int H,
                                  long mod_stride_b, long mod_stride_t, long gate_stride_b, long gate_stride_t,
                                  float eps, hipStream_t stream)
{
    if (!x || B < 0 || T < 0 || (shift == nullptr) != (scale == nullptr) || (n_blk && (H % 32 || ((size_t)n_blk & 15)))
        || (!n_out && !n_blk && !(y && x_out))) {
        set_last_error("residual_ln_mod: bad argument");
        return SELFTOK_EINVAL;
    }
    const int rows = B * T;
    if (rows == 0) return SELFTOK_OK;
    // wide rows: walk kernel (see its comment).  Direction: along the samples when the modulation is per token (or absent), along
    // the tokens when it is per sample; R rows per wave, fewer when that would leave the chip with < ~6 waves per CU
    int vpl = 0;
    switch (H) { case 256: vpl = 1; break; case 512: vpl = 2; break; case 1024: vpl = 4; break; case 1536: vpl = 6; break; default: break; }
    int variant = 1, R = 8, nt = 1;      // measured: profiles/r3_ln_variants.txt
#ifdef SELFTOK_TUNE
    { const char* e = getenv("SELFTOK_LN_VARIANT"); if (e) variant = atoi(e);      // 0: one row per wave (round-1 kernel), 1: walk
      e = getenv("SELFTOK_LN_R"); if (e) R = atoi(e);
      e = getenv("SELFTOK_LN_NT"); if (e) nt = atoi(e); }
#endif
    if (vpl && variant >= 1) {
        const bool per_sample_mod = shift ? (mod_stride_t == 0 && mod_stride_b != 0) : (gate && gate_stride_t == 0 && gate_stride_b != 0);
        const int walk_tokens = per_sample_mod ? 1 : 0;
        const int extent = walk_tokens ? T : B;
        if (R > extent) R = extent;
        while (R > 1 && (long)((extent + R - 1) / R) * (walk_tokens ? B : T) < 6 * 256) R = (R + 1) / 2;
        const long waves = walk_tokens ? (long)B * ((T + R - 1) / R) : (long)((B + R - 1) / R) * T;
        const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
        // operands that do not change along the walk are hoisted into registers (template flags: no dead register sets)
        bool hm = scale && (walk_tokens ? mod_stride_t == 0 : mod_stride_b == 0);
        bool hg = y && gate && (walk_tokens ? gate_stride_t == 0 : gate_stride_b == 0);
#ifdef SELFTOK_TUNE
        { const char* e = getenv("SELFTOK_LN_HOIST"); if (e) { const int m = atoi(e); hm = hm && (m & 1); hg = hg && (m & 2); } }
#endif
#define WALK(V, N, M, G)                                                                                                     \
        hipLaunchKernelGGL((residual_ln_mod_walk_kernel<V, N, M, G>), grid, block, 0, stream, x, y, gate, shift, scale, x_out, n_out, B, T, \
                           mod_stride_b, mod_stride_t, gate_stride_b, gate_stride_t, eps, n_blk, overflow, walk_tokens, R)
#define WALK_H(V, N) do { if (hm) { if (hg) WALK(V, N, true, true); else WALK(V, N, true, false); } else { if (hg) WALK(V, N, false, true); else WALK(V, N, false, false); } } while (0)
#ifdef SELFTOK_TUNE
#define WALK_NT(V) do { if (nt) WALK_H(V, true); else WALK_H(V, false); } while (0)
#else
#define WALK_NT(V) do { (void)nt; WALK_H(V, true); } while (0)
#endif
        switch (vpl) { case 1: WALK_NT(1); break; case 2: WALK_NT(2); break; case 4: WALK_NT(4); break; default: WALK_NT(6); break; }
#undef WALK_NT
#undef WALK_H
#undef WALK
        return check_launch("residual_ln_mod_walk_kernel");
    }
#define LAUNCH(G, VPL)                                                                                                  \
    hipLaunchKernelGGL((residual_ln_mod_kernel<G, VPL>), dim3((rows + (256 / G) - 1) / (256 / G)), dim3(256), 0, stream, \
                       x, y, gate, shift, scale, x_out, n_out, rows, T, mod_stride_b, mod_stride_t, gate_stride_b,      \
                       gate_stride_t, eps, n_blk, overflow)
    switch (H) {
        case 64: LAUNCH(16, 1); break;
        case 512: LAUNCH(64, 2); break;
        case 1536: LAUNCH(64, 6); break;
        case 1024: LAUNCH(64, 4); break;
        case 256: LAUNCH(64, 1); break;
        default: set_last_error("residual_ln_mod: unsupported hidden size (64/256/512/1024/1536)"); return SELFTOK_EINVAL;
    }
#undef LAUNCH
    return check_launch("residual_ln_mod_kernel");
}

int selftok_residual_ln_mod_f32(const float* x, const float* y, const float* gate, const float* shift, const float* scale,
                                float* x_out, float* n_out, int B, int T, int H,
                                long mod_stride_b, long mod_stride_t, long gate_stride_b, long gate_stride_t,
                                float eps, hipStream_t stream)
{
    return residual_ln_mod_launch(x, y, gate, shift, scale, x_out, n_out, nullptr, nullptr, B, T, H,
                                  mod_stride_b, mod_stride_t, gate_stride_b, gate_stride_t, eps, stream);
}

int selftok_residual_ln_mod_split(const float* x, const float* y, const float* gate, const float* shift, const float* scale,
                                  float* x_out, void* n_blk, int* overflow, int B, int T, int H,
                                  long mod_stride_b, long mod_stride_t, long gate_stride_b, long gate_stride_t,
                                  float eps, hipStream_t stream)
{
    if (!n_blk) { set_last_error("residual_ln_mod_split: null output"); return SELFTOK_EINVAL; }
    return residual_ln_mod_launch(x, y, gate, shift, scale, x_out, nullptr, (_Float16*)n_blk, overflow, B, T, H,
                                  mod_stride_b, mod_stride_t, gate_stride_b, gate_stride_t, eps, stream);
}

int selftok_bias_gelu_f32(float* h, const float* bias, long rows, int cols, hipStream_t stream)
{
    if (!h || rows < 0 || cols <= 0 || (cols & 3)) { set_last_error("bias_gelu: cols must be a multiple of 4"); return SELFTOK_EINVAL; }
    long n4 = rows * cols / 4;
    if (n4 == 0) return SELFTOK_OK;
    hipLaunchKernelGGL(bias_gelu_kernel, dim3(grid_for(n4)), dim3(256), 0, stream, h, bias, n4, cols / 4);
    return check_launch("bias_gelu_kernel");
}

int selftok_silu_f32(const float* in, float* out, long n, hipStream_t stream)
{
    if (!in || !out || n < 0) { set_last_error("silu: bad argument"); return SELFTOK_EINVAL; }
    if (n == 0) return SELFTOK_OK;
    hipLaunchKernelGGL(silu_kernel, dim3(grid_for(n)), dim3(256), 0, stream, in, out, n);
    return check_launch("silu_kernel");
}

int selftok_add_rows_f32(const float* in, const float* table, float* out, int B, long per_sample, hipStream_t stream)
{
    if (!in || !table || !out || B < 0 || per_sample <= 0 || (per_sample & 3)) { set_last_error("add_rows: bad argument"); return SELFTOK_EINVAL; }
    long n4 = (long)B * per_sample / 4;
    if (n4 == 0) return SELFTOK_OK;
    hipLaunchKernelGGL(add_rows_kernel, dim3(grid_for(n4)), dim3(256), 0, stream, in, table, out, n4, per_sample / 4);
    return check_launch("add_rows_kernel");
}

int selftok_timestep_embed_f32(const float* t, const float* freqs, float* out, int n, int dim, float t_scale, hipStream_t stream)
{
    if (!t || !freqs || !out || n < 0 || dim <= 0 || (dim & 1)) { set_last_error("timestep_embed: bad argument"); return SELFTOK_EINVAL; }
    if (n == 0) return SELFTOK_OK;
    int half = dim / 2;
    hipLaunchKernelGGL(timestep_embed_kernel, dim3((n * half + 255) / 256), dim3(256), 0, stream, t, freqs, out, n, half, t_scale);
    return check_launch("timestep_embed_kernel");
}

int selftok_patchify_f32(const float* x, float* out, int B, int C, int Hh, int Ww, hipStream_t stream)
{
    if (!x || !out || B < 0 || C <= 0 || (Hh & 1) || (Ww & 1)) { set_last_error("patchify: bad argument"); return SELFTOK_EINVAL; }
    long total = (long)B * (Hh / 2) * (Ww / 2) * C;
    if (total == 0) return SELFTOK_OK;
    hipLaunchKernelGGL(patchify_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, x, out, B, C, Hh, Ww);
    return check_launch("patchify_kernel");
}

int selftok_unpatchify_cfg_euler_f32(const float* y_cond, const float* y_uncond, const float* x, float* x_out, float* v_out,
                                     int B, int C, int hp, int wp, float dt, float cfg_scale, hipStream_t stream)
{
    if (!y_cond || B < 0 || (x_out && !x) || (!x_out && !v_out)) { set_last_error("unpatchify_cfg_euler: bad argument"); return SELFTOK_EINVAL; }
    long total = (long)B * C * (2 * hp) * wp;
    if (total == 0) return SELFTOK_OK;
    hipLaunchKernelGGL(unpatchify_euler_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, y_cond, y_uncond, x, x_out, v_out,
                       B, C, hp, wp, dt, cfg_scale);
    return check_launch("unpatchify_euler_kernel");
}

int selftok_rmsnorm_f32(const float* x, const float* w, float* out, long rows, int dim, float eps, hipStream_t stream)
{
    if (!x || !out || rows < 0 || dim <= 0) { set_last_error("rmsnorm: bad argument"); return SELFTOK_EINVAL; }
    if (rows == 0) return SELFTOK_OK;
    hipLaunchKernelGGL(rmsnorm_kernel, dim3((rows * 16 + 255) / 256), dim3(256), 0, stream, x, w, out, rows, dim, eps);
    return check_launch("rmsnorm_kernel");
}

int selftok_rotary_f32(const float* t, const float* freqs, float* out, long rows, int seq, int dim, float scale, hipStream_t stream)
{
    if (!t || !freqs || !out || rows < 0 || seq <= 0 || dim <= 0 || (dim & 1)) { set_last_error("rotary: bad argument"); return SELFTOK_EINVAL; }
    long total = rows * (dim / 2);
    if (total == 0) return SELFTOK_OK;
    hipLaunchKernelGGL(rotary_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, t, freqs, out, rows, seq, dim, scale);
    return check_launch("rotary_kernel");
}

}  // extern "C"
