// Block-structured mix: 16 MFMAs (2 chains of 8) then a max3 scan of the PREVIOUS block's 32 results (ping-pong),
// no memory traffic.  Tells what the VQ hot loop can reach when only the MFMA/VALU mix matters.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float max16(const f32x16& a)
{
    float m0 = __builtin_fmaxf(__builtin_fmaxf(a[0], a[1]), a[2]), m1 = __builtin_fmaxf(__builtin_fmaxf(a[3], a[4]), a[5]);
    float m2 = __builtin_fmaxf(__builtin_fmaxf(a[6], a[7]), a[8]), m3 = __builtin_fmaxf(__builtin_fmaxf(a[9], a[10]), a[11]);
    float m4 = __builtin_fmaxf(__builtin_fmaxf(a[12], a[13]), a[14]);
    return __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(m0, m1), m2), __builtin_fmaxf(__builtin_fmaxf(m3, m4), a[15]));
}

template <int SCAN>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0)
{
    float a[8], b0v[8], b1v[8];
    for (int i = 0; i < 8; ++i) { a[i] = a0 + i * 0.01f + threadIdx.x * 1e-6f; b0v[i] = b0 + i; b1v[i] = b0 - i; }
    float bv0 = -1e30f, bv1 = -1e30f; int bt0 = 0, bt1 = 0;
    f32x16 z = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
    f32x16 A0 = z, A1 = z, B0 = z, B1 = z;
    for (int it = 0; it < iters; ++it) {
        B0 = z; B1 = z;
#pragma unroll
        for (int m = 0; m < 8; ++m) { B0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b0v[m], B0, 0, 0, 0); B1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b1v[m], B1, 0, 0, 0); }
        if (SCAN) { float m = max16(A0); bool g = m > bv0; bv0 = g ? m : bv0; bt0 = g ? it : bt0; m = max16(A1); g = m > bv1; bv1 = g ? m : bv1; bt1 = g ? it : bt1; }
        else asm volatile("" ::"v"(A0), "v"(A1));
        A0 = z; A1 = z;
#pragma unroll
        for (int m = 0; m < 8; ++m) { A0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b0v[m], A0, 0, 0, 0); A1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b1v[m], A1, 0, 0, 0); }
        if (SCAN) { float m = max16(B0); bool g = m > bv0; bv0 = g ? m : bv0; bt0 = g ? it : bt0; m = max16(B1); g = m > bv1; bv1 = g ? m : bv1; bt1 = g ? it : bt1; }
        else asm volatile("" ::"v"(B0), "v"(B1));
        a[it & 7] += 1e-7f;     // keep the operands loop-variant
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = bv0 + bv1 + bt0 + bt1 + A0[0] + A1[0];
}

template <typename F> float time_ms(F f, int reps)
{
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(s); for (int i = 0; i < reps; ++i) f(); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e); return ms / reps;
}
int main()
{
    float* out; (void)hipMalloc(&out, 256 * 8192 * sizeof(float));
    const int iters = 400;
    for (int wpe = 1; wpe <= 4; ++wpe) {
        int blocks = 256 * wpe;
        double fl = 2.0 * 32 * 32 * 2 * 32.0 * iters * blocks * 4;
        float m0 = time_ms([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 3);
        float m1 = time_ms([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 3);
        printf("waves/SIMD=%d : no scan %.1f TF   max3 scan %.1f TF\n", wpe, fl / m0 / 1e9, fl / m1 / 1e9);
    }
    return 0;
}
