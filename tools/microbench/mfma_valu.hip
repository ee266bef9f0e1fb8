// How many VALU ops hide in the shadow of one v_mfma_f32_32x32x2_f32?  (design input for the VQ scan)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0)
{
    f32x16 acc0 = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}, acc1 = acc0;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    float bv = -1e30f; int bi = 0;
    float s[16];
    for (int i = 0; i < 16; ++i) s[i] = a0 * i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (r & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (KIND == 0) {            // cmp + 2 cndmask (the argmax scan)
                    if (v % 3 == 0) { bool g = s[(r + v) & 15] > bv; bv = g ? s[(r + v) & 15] : bv; bi = g ? (r + v) : bi; }
                } else {                    // independent fma
                    s[(r * NV + v) & 15] = __builtin_fmaf(s[(r * NV + v) & 15], 1.0000001f, 1e-7f);
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, KIND == 0 ? NV : NV, 0);
        }
        s[it & 15] += bv;
    }
    float t = bv + bi;
    for (int r = 0; r < 16; ++r) t += acc0[r] + acc1[r] + s[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <typename F> float time_ms(F f, int reps)
{
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(s); for (int i = 0; i < reps; ++i) f(); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e); return ms / reps;
}
#define RUN(NV, KIND) { float ms = time_ms([&] { hipLaunchKernelGGL((k<NV, KIND>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 3); \
    printf("  kind=%d valu/mfma=%2d : %.1f TF mfma\n", KIND, NV, fl / ms / 1e9); }
int main()
{
    float* out; (void)hipMalloc(&out, 256 * 8192 * sizeof(float));
    const int iters = 500;
    for (int wpe = 1; wpe <= 4; wpe *= 2) {
        int blocks = 256 * wpe;
        double fl = 2.0 * 32 * 32 * 2 * 16.0 * iters * blocks * 4;
        printf("waves/SIMD=%d\n", wpe);
        RUN(0, 1) RUN(4, 1) RUN(8, 1) RUN(12, 1) RUN(16, 1) RUN(24, 1)
        RUN(3, 0) RUN(6, 0) RUN(9, 0) RUN(12, 0)
    }
    return 0;
}
