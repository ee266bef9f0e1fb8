// Measured ceilings on the box we run on ("measure, don't assume"): fp32-input MFMA rate, fp32 VALU FMA
// rate, HBM copy bandwidth.  Build: hipcc --offload-arch=gfx950 -O3 -o peaks peaks.hip ; run: ./peaks
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_f32_loop(float* out, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x16{0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void valu_fma_loop(float* out, int iters, float a0)
{
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = a0 + i + threadIdx.x;
    float m = 1.0000001f, c = 1e-7f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], m, c);
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void copy4(const float4* __restrict__ in, float4* __restrict__ out, size_t n4)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) out[i] = in[i];
}

template <typename F> float time_ms(F f, int reps)
{
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    f(); hipDeviceSynchronize();
    hipEventRecord(s); for (int i = 0; i < reps; ++i) f(); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); return ms / reps;
}

int main()
{
    float* out; hipMalloc(&out, 256 * 8192 * sizeof(float));
    const int iters = 2000;
    for (int wpe = 1; wpe <= 4; ++wpe) {      // waves per SIMD via blocks per CU
        int blocks = 256 * wpe;
        float ms1 = time_ms([&] { hipLaunchKernelGGL(mfma_f32_loop<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 5);
        float ms2 = time_ms([&] { hipLaunchKernelGGL(mfma_f32_loop<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 5);
        float ms4 = time_ms([&] { hipLaunchKernelGGL(mfma_f32_loop<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 5);
        double fl = 2.0 * 32 * 32 * 2 * 8.0 * iters * blocks * 4;     // per accumulator
        printf("mfma_f32_32x32x2 waves/SIMD=%d : 1acc %.1f TF  2acc %.1f TF  4acc %.1f TF\n", wpe, fl / ms1 / 1e9, 2 * fl / ms2 / 1e9, 4 * fl / ms4 / 1e9);
    }
    {
        int blocks = 256 * 8;
        float ms = time_ms([&] { hipLaunchKernelGGL(valu_fma_loop, dim3(blocks), dim3(256), 0, 0, out, 4000, 1.f); }, 5);
        printf("v_fma_f32 (16 chains/lane, 8 waves/SIMD): %.1f TF\n", 2.0 * 16 * 4000 * blocks * 256 / ms / 1e9);
    }
    {
        size_t n = (size_t)1 << 30;   // 1 GiB each way
        float4 *a, *b; hipMalloc(&a, n); hipMalloc(&b, n); hipMemset(a, 1, n);
        float ms = time_ms([&] { hipLaunchKernelGGL(copy4, dim3(256 * 16), dim3(256), 0, 0, a, b, n / 16); }, 5);
        printf("HBM copy float4 1 GiB: %.2f TB/s (read+write)\n", 2.0 * n / ms / 1e9);
    }
    return 0;
}
