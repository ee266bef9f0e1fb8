// Do VALU instructions run in the shadow of a v_mfma_f32_32x32x16_f16 (8 passes = 32 cycles) on gfx950, or do they share its issue time?
// The VQ coarse kernel (csrc/vq.hip, vq_f16_kernel<RT, 1>) executes per 32 x 32 scores ONE such MFMA and 12 VALU instructions (8 v_max3_f32, v_and_or_b32,
// 2 v_med3_u32, v_max_u32 = 48 issue cycles): overlapped the tile costs max(32, 48) = 48 cycles, serialised 80.
// KIND 0: VALU on registers that do not depend on the MFMA (pure issue question); KIND 1: the VALU of iteration i reduce the accumulator the MFMA of
// iteration i - 1 wrote (the kernel's pattern: two accumulator sets, MFMA of unit u + 1 issued before the scan of unit u).
// Output: ns per MFMA per SIMD (4 waves per SIMD share it) and the ratio to the MFMA-only loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 vh8 __attribute__((ext_vector_type(8)));

template <int NV, int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0)
{
    const f32x16 zero = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
    vh8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(a0 + 0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.5f + 0.002f * i); }
    float s[16];
    for (int i = 0; i < 16; ++i) s[i] = a0 * i + threadIdx.x;
    f32x16 accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, zero, 0, 0, 0), accB = accA;
    (void)zero;
    float m = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            f32x16& cur = (r & 1) ? accB : accA;          // written one iteration ago
            f32x16& nxt = (r & 1) ? accA : accB;
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(nxt) : "v"(a), "v"(b));          // volatile: identical MFMAs must not be merged
            if (KIND == 1) {
                // the kernel's scan, NV <= 12: 8 v_max3 over the 16 accumulator values (+ clamp), then NV - 8 dependent integer-ish ops
#define MAX3(d, x, y, z) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z))
                float a0_, a1_, a2_, a3_, a4_, t0, t1, t;
                MAX3(a0_, cur[0], cur[1], cur[2]); MAX3(a1_, cur[3], cur[4], cur[5]); MAX3(a2_, cur[6], cur[7], cur[8]); MAX3(a3_, cur[9], cur[10], cur[11]);
                MAX3(a4_, cur[12], cur[13], cur[14]); MAX3(t0, a0_, a1_, a2_); MAX3(t1, a3_, a4_, cur[15]); MAX3(t, t0, t1, m);
#pragma unroll
                for (int v = 8; v < NV; ++v) asm volatile("v_fma_f32 %0, %1, 1.0, %2" : "=v"(t) : "v"(t), "v"(m));
                m = t;
            } else {
#pragma unroll
                for (int v = 0; v < NV; ++v) { const int i = (r * NV + v) & 15; asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(s[i]) : "v"(s[i]), "v"(s[(i + 5) & 15]), "v"(s[(i + 11) & 15])); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float t = m;
    for (int r = 0; r < 16; ++r) t += accA[r] + accB[r] + s[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <typename F> float time_ms(F f, int reps)
{
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(s); for (int i = 0; i < reps; ++i) f(); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e); return ms / reps;
}
static double base_ns = 0;
#define RUN(NV, KIND) { float ms = time_ms([&] { hipLaunchKernelGGL((k<NV, KIND>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); }, 3); \
    const double ns = ms * 1e6 / (8.0 * iters * wpe); if (NV == 0) base_ns = ns; \
    printf("  kind=%d valu/mfma=%2d : %7.2f ns per MFMA and SIMD = %.2f x the MFMA-only loop (overlapped: %.2f, serialised: %.2f)\n", KIND, NV, ns, ns / base_ns, \
           (4.0 * NV > 32 ? 4.0 * NV : 32.0) / 32.0, (32.0 + 4.0 * NV) / 32.0); }
int main()
{
    float* out; (void)hipMalloc(&out, 256 * 8192 * sizeof(float));
    const int iters = 2000;
    for (int wpe = 1; wpe <= 4; wpe *= 2) {
        int blocks = 256 * wpe;          // 256 CUs x wpe workgroups of 4 waves = wpe waves per SIMD
        printf("waves/SIMD=%d\n", wpe);
        RUN(0, 0) RUN(4, 0) RUN(8, 0) RUN(12, 0) RUN(16, 0)
        RUN(8, 1) RUN(12, 1)
    }
    return 0;
}
