// Entropy regularisers of the training-side code book WITHOUT the [N, C] probability matrix  (gfx950)
// ---------------------------------------------------------------------------------------------------------------------------------
// Reference (mimogpt/models/selftok/vector_quantize_pytorch.py): VectorQuantize.forward in training keeps `distances` [1, B, K, C]
// (cosine scores of every row against every code, :561), scales them by 10 (:1007) and takes
//     calc_entropy      (:89-100)   p = softmax_c ;  H(mean_n p)  "entropy_to_max" ,  mean_n H(p_n)  "entropy_to_min"
//     calc_ema_entropy  (:109-118)  ap_k = mean_b p[b, k, :] ; ema_p = tpc (1 - r) + ap_k r ; entropy per token position and per
//                                   group of positions
// i.e. softmax, mean and log over an [N, C] fp32 tensor: 4.3 GB per materialised copy at N = C = 32768, several copies, plus their
// autograd twins.  Everything those functions need from the matrix is two reductions of it:
//     per row     S_n = sum_c exp(a s_nc),  T_n = sum_c exp(a s_nc) a s_nc      ->  H(p_n) = log S_n - T_n / S_n
//     per (k, c)  ap_k[c] = 1/B sum_b exp(a s_(b,k),c) / S_(b,k)                ->  [K, C]  (64 MiB at K = 512)
// (a = 10; |a s| <= 10 because both sides are unit vectors, so exp needs no running maximum), and the gradient of any F(ap_k) with
// respect to the rows needs three more reductions of the same matrix (below).  Each kernel re-computes the D = 16 scores it needs
// in registers: 2 N C D flops per pass against 4 N C bytes per materialised tensor.
//
//   vq_softmax_rowstats_kernel   thread = row, code tiles broadcast from LDS             -> partial (S, T) per code split
//   vq_softmax_rowstats_reduce   sums the splits, emits (1 / S_n, H(p_n))
//   vq_softmax_colmean_kernel    thread = code (in registers), block = one token position k, rows b broadcast from LDS -> ap_k[c]
//   vq_softmax_backward_kernel   lanes = samples b of ONE token position k, codes and g[k, c] arrive by scalar loads (wave-uniform):
//                                A = sum_c p g e_c , m = sum_c p e_c , t = sum_c p g    per row and code split
//   vq_softmax_backward_reduce   dF/dx = a/B (A - t m) ;  through the l2norm:  dF/dz = (dF/dx - x (x . dF/dx)) / |z|
// with g = dF/d(ap_k) supplied by the caller (autograd over the small [K, C] epilogue in torch, vq_train.py).
// Scores use the canonical k-ascending FMA chain of vq.hip (same bits as the argmax's scores); exp is v_exp_f32 (__expf): relative
// error <= 1e-6 at |a s| <= 10, against the reference's fp32 softmax that is rounding-level.
// Compiled with -ffp-contract=off.
#include "common.h"
#include "selftok_hip.h"

namespace selftok {
namespace {

constexpr int D = 16;
constexpr int SPLITS = 8;          // code splits of the row-oriented passes (N / 64 waves alone would not fill 1024 SIMDs)
constexpr int TILE = 256;          // codes per LDS tile / rows per LDS tile
constexpr int BW_STRIDE = 36;      // floats per (split, row) of the backward partials: A[16] m[16] t, padded to 16-byte multiples

__device__ __forceinline__ void load16(const float* __restrict__ p, float (&v)[D])
{
    const float4* p4 = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int q = 0; q < 4; ++q) { float4 t = p4[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
}

// canonical l2norm of vq.hip; also returns the norm used
__device__ __forceinline__ float unit16(const float (&z)[D], float (&x)[D], int normalize)
{
    if (!normalize) {
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = z[k];
        return 1.0f;
    }
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = __builtin_fmaf(z[j + 8], z[j + 8], z[j] * z[j]);
    float s = a[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) s = s + a[j];
    float nrm = __builtin_sqrtf(s);
    nrm = (nrm > 1e-12f) ? nrm : 1e-12f;
    if (s != s) nrm = s;
#pragma unroll
    for (int k = 0; k < D; ++k) x[k] = z[k] / nrm;
    return nrm;
}

__device__ __forceinline__ float dot16(const float (&x)[D], const float (&e)[D])
{
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < D; ++k) s = __builtin_fmaf(x[k], e[k], s);
    return s;
}

__global__ __launch_bounds__(256) void vq_softmax_rowstats_kernel(const float* __restrict__ z, const float* __restrict__ cb, float2* __restrict__ part,
                                                                  int N, int C, int csplit, float scale, int normalize)
{
    __shared__ float4 s_e[TILE * 4];
    const int row = blockIdx.x * 256 + threadIdx.x;
    const int c0 = blockIdx.y * csplit, c1 = min(C, c0 + csplit);
    float zz[D], x[D];
#pragma unroll
    for (int k = 0; k < D; ++k) zz[k] = 0.f;
    if (row < N) load16(z + (size_t)row * D, zz);
    unit16(zz, x, normalize);
    float S = 0.f, T = 0.f;
    for (int t0 = c0; t0 < c1; t0 += TILE) {
        __syncthreads();
        const int c = t0 + threadIdx.x;
        if (c < c1) {
            const float4* p = reinterpret_cast<const float4*>(cb + (size_t)c * D);
#pragma unroll
            for (int q = 0; q < 4; ++q) s_e[q * TILE + threadIdx.x] = p[q];          // [quarter][code]: conflict-free stores, broadcast reads
        }
        __syncthreads();
        const int n = min(TILE, c1 - t0);
        float St = 0.f, Tt = 0.f;                                                   // per-tile partial sums: two-level accumulation
#pragma unroll 4
        for (int j = 0; j < n; ++j) {
            float e[D];
#pragma unroll
            for (int q = 0; q < 4; ++q) { float4 t = s_e[q * TILE + j]; e[4 * q] = t.x; e[4 * q + 1] = t.y; e[4 * q + 2] = t.z; e[4 * q + 3] = t.w; }
            const float l = dot16(x, e) * scale;
            const float p = __expf(l);
            St += p;
            Tt = __builtin_fmaf(p, l, Tt);
        }
        S += St;
        T += Tt;
    }
    if (row < N) part[(size_t)blockIdx.y * N + row] = make_float2(S, T);
}

__global__ __launch_bounds__(256) void vq_softmax_rowstats_reduce(const float2* __restrict__ part, float2* __restrict__ rowstats, int N, int nsplit)
{
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= N) return;
    float S = 0.f, T = 0.f;
    for (int s = 0; s < nsplit; ++s) { float2 v = part[(size_t)s * N + row]; S += v.x; T += v.y; }
    rowstats[row] = make_float2(1.0f / S, logf(S) - T / S);
}

__global__ __launch_bounds__(256) void vq_softmax_colmean_kernel(const float* __restrict__ z, const float* __restrict__ cb, const float2* __restrict__ rowstats,
                                                                 float* __restrict__ out, int B, int K, int C, float scale, int normalize)
{
    __shared__ float4 s_x[TILE * 4];
    __shared__ float s_inv[TILE];
    const int k = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    float e[D];
#pragma unroll
    for (int q = 0; q < D; ++q) e[q] = 0.f;
    if (c < C) load16(cb + (size_t)c * D, e);
    float acc = 0.f;
    for (int b0 = 0; b0 < B; b0 += TILE) {
        __syncthreads();
        const int b = b0 + threadIdx.x;
        if (b < B) {
            const size_t row = (size_t)b * K + k;
            float zz[D], x[D];
            load16(z + row * D, zz);
            unit16(zz, x, normalize);
#pragma unroll
            for (int q = 0; q < 4; ++q) s_x[q * TILE + threadIdx.x] = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
            s_inv[threadIdx.x] = rowstats[row].x;
        }
        __syncthreads();
        const int n = min(TILE, B - b0);
        float at = 0.f;
#pragma unroll 4
        for (int j = 0; j < n; ++j) {
            float x[D];
#pragma unroll
            for (int q = 0; q < 4; ++q) { float4 t = s_x[q * TILE + j]; x[4 * q] = t.x; x[4 * q + 1] = t.y; x[4 * q + 2] = t.z; x[4 * q + 3] = t.w; }
            at = __builtin_fmaf(__expf(dot16(x, e) * scale), s_inv[j], at);
        }
        acc += at;
    }
    if (c < C) out[(size_t)k * C + c] = acc / (float)B;
}

__global__ __launch_bounds__(64) void vq_softmax_backward_kernel(const float* __restrict__ z, const float* __restrict__ cb, const float2* __restrict__ rowstats,
                                                                 const float* __restrict__ g, float* __restrict__ part, int B, int K, int C, int csplit,
                                                                 float scale, int normalize)
{
    const int k = blockIdx.y, split = blockIdx.z;
    const int b = blockIdx.x * 64 + threadIdx.x;
    const int c0 = split * csplit, c1 = min(C, c0 + csplit);
    const int N = B * K;
    const bool live = b < B;
    const size_t row = live ? (size_t)b * K + k : (size_t)k;
    float zz[D], x[D];
    load16(z + row * D, zz);
    unit16(zz, x, normalize);
    const float inv = rowstats[row].x;
    float A[D], m[D], t = 0.f;
#pragma unroll
    for (int q = 0; q < D; ++q) { A[q] = 0.f; m[q] = 0.f; }
    const float* __restrict__ gk = g + (size_t)k * C;
#pragma unroll 2
    for (int c = c0; c < c1; ++c) {                       // c, e_c and g[k, c] are wave-uniform: scalar loads, SGPR operands
        float e[D];
        load16(cb + (size_t)c * D, e);
        const float p = __expf(dot16(x, e) * scale) * inv;
        const float pg = p * gk[c];
        t += pg;
#pragma unroll
        for (int q = 0; q < D; ++q) { m[q] = __builtin_fmaf(p, e[q], m[q]); A[q] = __builtin_fmaf(pg, e[q], A[q]); }
    }
    if (!live) return;
    float4* o = reinterpret_cast<float4*>(part + ((size_t)split * N + row) * BW_STRIDE);
#pragma unroll
    for (int q = 0; q < 4; ++q) { o[q] = make_float4(A[4 * q], A[4 * q + 1], A[4 * q + 2], A[4 * q + 3]); o[4 + q] = make_float4(m[4 * q], m[4 * q + 1], m[4 * q + 2], m[4 * q + 3]); }
    o[8] = make_float4(t, 0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(256) void vq_softmax_backward_reduce(const float* __restrict__ z, const float* __restrict__ part, float* __restrict__ grad_z,
                                                                  int N, int nsplit, float coef, int normalize)
{
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= N) return;
    float A[D], m[D], t = 0.f;
#pragma unroll
    for (int q = 0; q < D; ++q) { A[q] = 0.f; m[q] = 0.f; }
    for (int s = 0; s < nsplit; ++s) {
        const float4* o = reinterpret_cast<const float4*>(part + ((size_t)s * N + row) * BW_STRIDE);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 a = o[q], mm = o[4 + q];
            A[4 * q] += a.x; A[4 * q + 1] += a.y; A[4 * q + 2] += a.z; A[4 * q + 3] += a.w;
            m[4 * q] += mm.x; m[4 * q + 1] += mm.y; m[4 * q + 2] += mm.z; m[4 * q + 3] += mm.w;
        }
        t += o[8].x;
    }
    float gx[D];
#pragma unroll
    for (int q = 0; q < D; ++q) gx[q] = coef * (A[q] - t * m[q]);                  // softmax backward contracted with the codes
    if (normalize) {                                                               // x = z / |z|:  dz = (dx - x (x . dx)) / |z|
        float zz[D], x[D];
        load16(z + (size_t)row * D, zz);
        const float nrm = unit16(zz, x, 1);
        const float xd = dot16(x, gx);
#pragma unroll
        for (int q = 0; q < D; ++q) gx[q] = (gx[q] - x[q] * xd) / nrm;
    }
    float4* o = reinterpret_cast<float4*>(grad_z + (size_t)row * D);
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = make_float4(gx[4 * q], gx[4 * q + 1], gx[4 * q + 2], gx[4 * q + 3]);
}

}  // namespace
}  // namespace selftok

using namespace selftok;

extern "C" {

size_t selftok_vq_softmax_workspace_bytes(int N) { return (size_t)SPLITS * (size_t)(N > 0 ? N : 1) * BW_STRIDE * sizeof(float); }

int selftok_vq_softmax_stats_f32(const float* z, const float* codebook, float* rowstats, float* colmean, void* workspace, int B, int K, int C, int Dm,
                                 float scale, int flags, hipStream_t stream)
{
    if (Dm != D || B < 0 || K <= 0 || C <= 0 || K > 65535 || (long)B * K > 0x7FFFFFFFl) { set_last_error("vq_softmax_stats: bad argument (D == 16, K <= 65535)"); return SELFTOK_EINVAL; }
    if (B == 0) return SELFTOK_OK;
    if (!z || !codebook || !rowstats || !workspace) { set_last_error("vq_softmax_stats: null pointer"); return SELFTOK_EINVAL; }
    const int N = B * K, norm = (flags & SELFTOK_PRENORMED) ? 0 : 1;
    const int csplit = (C + SPLITS - 1) / SPLITS, nsplit = (C + csplit - 1) / csplit;
    hipLaunchKernelGGL(vq_softmax_rowstats_kernel, dim3((N + 255) / 256, nsplit), dim3(256), 0, stream, z, codebook, (float2*)workspace, N, C, csplit, scale, norm);
    hipLaunchKernelGGL(vq_softmax_rowstats_reduce, dim3((N + 255) / 256), dim3(256), 0, stream, (const float2*)workspace, (float2*)rowstats, N, nsplit);
    if (colmean)
        hipLaunchKernelGGL(vq_softmax_colmean_kernel, dim3((C + 255) / 256, K), dim3(256), 0, stream, z, codebook, (const float2*)rowstats, colmean, B, K, C, scale, norm);
    return check_launch("vq_softmax_stats");
}

int selftok_vq_softmax_backward_f32(const float* z, const float* codebook, const float* rowstats, const float* g_colmean, float* grad_z, void* workspace,
                                    int B, int K, int C, int Dm, float scale, int flags, hipStream_t stream)
{
    if (Dm != D || B < 0 || K <= 0 || C <= 0 || K > 65535 || (long)B * K > 0x7FFFFFFFl) { set_last_error("vq_softmax_backward: bad argument (D == 16, K <= 65535)"); return SELFTOK_EINVAL; }
    if (B == 0) return SELFTOK_OK;
    if (!z || !codebook || !rowstats || !g_colmean || !grad_z || !workspace) { set_last_error("vq_softmax_backward: null pointer"); return SELFTOK_EINVAL; }
    const int N = B * K, norm = (flags & SELFTOK_PRENORMED) ? 0 : 1;
    const int csplit = (C + SPLITS - 1) / SPLITS, nsplit = (C + csplit - 1) / csplit;
    hipLaunchKernelGGL(vq_softmax_backward_kernel, dim3((B + 63) / 64, K, nsplit), dim3(64), 0, stream, z, codebook, (const float2*)rowstats, g_colmean,
                       (float*)workspace, B, K, C, csplit, scale, norm);
    hipLaunchKernelGGL(vq_softmax_backward_reduce, dim3((N + 255) / 256), dim3(256), 0, stream, z, (const float*)workspace, grad_z, N, nsplit, scale / (float)B, norm);
    return check_launch("vq_softmax_backward");
}

}  // extern "C"
