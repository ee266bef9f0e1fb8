// bf16 epilogue kernels around the SD3 VAE convolutions (the convs themselves run through
// PyTorch-ROCm / MIOpen).  gfx950, NCHW contiguous bf16.
//
// Reference arithmetic: torch ops on bf16 tensors compute in fp32 and round the result of EVERY op to
// bf16 (sd3_impls.py:215-254 GroupNorm -> SiLU; SelftokPipeline.py:135-137,216-218,285-290 for the
// latent-format and norm_ip element-wise chains), so the fused kernels below round at the same points.
#include "common.h"
#include "selftok_hip.h"   // the C ABI declared there must match the definitions below
#include <hip/hip_bf16.h>

namespace selftok {

__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float(((uint32_t)u) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f)
{   // round-to-nearest-even, NaN preserved (matches torch's c10::BFloat16 conversion)
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

typedef unsigned short us8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float block_sum(float v, float* red)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    return t;
}

// GroupNorm(32 groups, eps, affine) fused with SiLU over NCHW bf16 (ResnetBlock / norm_out epilogue:
// sd3_impls.py:244-253, 373-375, 440-442).  One workgroup per (sample, group): the group's
// (C/32)*H*W elements are contiguous.  Two passes over the group (the second one is L2-resident).
__global__ __launch_bounds__(512) void groupnorm_silu_bf16_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
                                                                  const unsigned short* __restrict__ bias, unsigned short* __restrict__ out,
                                                                  int C, int HW, int groups, float eps, int apply_silu)
{
    __shared__ float red[8];
    const int cpg = C / groups;
    const long n = (long)cpg * HW;
    const int g = blockIdx.x % groups;
    const long base = (long)blockIdx.x * n;       // (b*groups + g) * n
    const us8* x8 = reinterpret_cast<const us8*>(x + base);
    const long n8 = n >> 3;
    float s = 0.f;
    for (long i = threadIdx.x; i < n8; i += blockDim.x) {
        us8 v = x8[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) s += bf2f(v[e]);
    }
    const float mean = block_sum(s, red) / (float)n;
    float q = 0.f;
    for (long i = threadIdx.x; i < n8; i += blockDim.x) {
        us8 v = x8[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) { float d = bf2f(v[e]) - mean; q += d * d; }
    }
    const float var = block_sum(q, red) / (float)n;
    const float rstd = 1.0f / __builtin_sqrtf(var + eps);
    us8* o8 = reinterpret_cast<us8*>(out + base);
    for (long i = threadIdx.x; i < n8; i += blockDim.x) {
        const int ch = g * cpg + (int)((i << 3) / HW);     // HW % 8 == 0: the 8 elements share a channel
        const float ww = bf2f(w[ch]), bb = bf2f(bias[ch]);
        us8 v = x8[i], r;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = rbf((bf2f(v[e]) - mean) * rstd * ww + bb);     // GroupNorm output rounded to bf16
            if (apply_silu) y = y / (1.0f + expf(-y));                // SiLU on the bf16 value
            r[e] = f2bf(y);
        }
        o8[i] = r;
    }
}

// SD3LatentFormat.process_in on the bf16 VAE mean, then .to(fp32) (SelftokPipeline.py:216-218; sd3_impls.py:140-141):
//   out = float( bf16( bf16(z - shift) * scale ) ).   mean = first `c_keep` of `c_in` channels (.mode()).
__global__ void latent_process_in_kernel(const unsigned short* __restrict__ moments, float* __restrict__ out, int B, int c_in, int c_keep,
                                         int HW, float shift, float scale)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)B * c_keep * HW;
    if (i >= total) return;
    int p = i % HW;
    long r = i / HW;
    int c = r % c_keep;
    int b = r / c_keep;
    float z = bf2f(moments[((long)b * c_in + c) * HW + p]);
    out[i] = rbf(rbf(z - rbf(shift)) * scale);   // torch-CPU: sub's python scalar is cast to bf16, mul's stays fp32 (probed); each op rounds to bf16
}

// SD3LatentFormat.process_out in fp32, then .to(bf16) (SelftokPipeline.py:285-287; sd3_impls.py:143-144)
__global__ void latent_process_out_kernel(const float* __restrict__ z, unsigned short* __restrict__ out, long n, float shift, float scale)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = f2bf((z[i] / scale) + shift);
}

// norm_ip(recons, -1, 1) in place on bf16 (SelftokPipeline.py:135-137): clamp_, sub_(low), div_(high-low)
__global__ void clamp01_bf16_kernel(unsigned short* __restrict__ img, long n)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x0 = bf2f(img[i]);
    float v = (x0 != x0) ? x0 : fminf(fmaxf(x0, -1.0f), 1.0f);   // clamp_ keeps NaN
    v = rbf(v - (-1.0f));
    img[i] = f2bf(v / 2.0f);
}

}  // namespace selftok

using namespace selftok;

extern "C" {

int selftok_groupnorm_silu_bf16(const void* x, const void* weight, const void* bias, void* out, int B, int C, int HW, int groups,
                                float eps, int apply_silu, hipStream_t stream)
{
    if (!x || !weight || !bias || !out || B < 0 || groups <= 0 || C % groups || (HW & 7)) { set_last_error("groupnorm_silu: need C%groups==0 and H*W%8==0"); return SELFTOK_EINVAL; }
    if (B == 0) return SELFTOK_OK;
    hipLaunchKernelGGL(groupnorm_silu_bf16_kernel, dim3(B * groups), dim3(512), 0, stream, (const unsigned short*)x, (const unsigned short*)weight,
                       (const unsigned short*)bias, (unsigned short*)out, C, HW, groups, eps, apply_silu);
    return check_launch("groupnorm_silu_bf16_kernel");
}

int selftok_latent_process_in(const void* moments_bf16, float* out, int B, int c_in, int c_keep, int HW, float shift, float scale, hipStream_t stream)
{
    if (!moments_bf16 || !out || B < 0 || c_keep > c_in) { set_last_error("latent_process_in: bad argument"); return SELFTOK_EINVAL; }
    long total = (long)B * c_keep * HW;
    if (total == 0) return SELFTOK_OK;
    hipLaunchKernelGGL(latent_process_in_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, (const unsigned short*)moments_bf16, out, B, c_in, c_keep, HW, shift, scale);
    return check_launch("latent_process_in_kernel");
}

int selftok_latent_process_out(const float* z, void* out_bf16, long n, float shift, float scale, hipStream_t stream)
{
    if (!z || !out_bf16 || n < 0) { set_last_error("latent_process_out: bad argument"); return SELFTOK_EINVAL; }
    if (n == 0) return SELFTOK_OK;
    hipLaunchKernelGGL(latent_process_out_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, z, (unsigned short*)out_bf16, n, shift, scale);
    return check_launch("latent_process_out_kernel");
}

int selftok_clamp01_bf16(void* img, long n, hipStream_t stream)
{
    if (!img || n < 0) { set_last_error("clamp01: bad argument"); return SELFTOK_EINVAL; }
    if (n == 0) return SELFTOK_OK;
    hipLaunchKernelGGL(clamp01_bf16_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (unsigned short*)img, n);
    return check_launch("clamp01_bf16_kernel");
}

}  // extern "C"
