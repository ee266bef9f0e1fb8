// What bounds an fp32-input MFMA GEMM loop on this chip?  (round 6: every fp32-MFMA kernel of this repo -- xe_gemm128, xconv, the first three versions of
// csrc/gemm_fp32.hip's loop -- sits at 0.79 - 0.83 of the 157 TF peak, the vendor's GEMM at 0.93 - 0.96.)  One 512-thread workgroup per CU (two waves per
// SIMD, as csrc/gemm_fp32.hip), each wave a stream of v_mfma into two accumulators; variants add, one at a time, what the real loop has beside the MFMAs:
//   form      : 32x32x2 (VGPR acc) | 32x32x1_2b (VGPR acc) | 32x32x1_2b with the accumulators pinned to AGPRs (inline asm)
//   operands  : constants | fragments read from LDS (3 ds_read_b128 per 8 MFMAs, random data: the toggling a real GEMM has)
//   barrier   : none | one s_barrier per 64 MFMAs
// Reports TFLOP/s and the effective shader clock (s_memtime cycles / s_memrealtime 100 MHz ticks of workgroup 0).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f32_forms mfma_f32_forms.hip ; run: ./mfma_f32_forms
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned long long g_stamp[2];

template <int FORM, bool LDSOP, bool BAR>
__global__ __launch_bounds__(512) void loop_kernel(float* out, const float* seed, int iters, int prefetch)
{
    __shared__ __attribute__((aligned(1024))) char lds[144 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 144 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = seed[i];
    __syncthreads();
    unsigned long long t0 = 0, r0 = 0;
    if (tid == 0) { t0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
    const int a_row = (64 * (wave & 3) + lane) * 128 + (((lane >> 1) & 7) << 4);
    const int b_row = 32768 + (64 * (wave >> 2) + (lane & 31)) * 128 + (((lane >> 1) & 7) << 4);
    f32x32 acc0 = {0}, acc1 = {0};
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    f32x4 va = {1.f + lane * 1e-6f, 1.1f, 1.2f, 1.3f}, vb0 = {2.f, 2.1f, 2.2f, 2.3f}, vb1 = {3.f, 3.1f, 3.2f, 3.3f};
    for (int it = 0; it < iters; ++it) {
        const char* la = lds + (it % 3) * 49152;
        if (BAR) __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            f32x4 na = va, nb0 = vb0, nb1 = vb1;
            if (LDSOP) {
                const int qq = prefetch ? ((q + 1) & 7) : q;        // prefetch: piece q + 1 is read while piece q's MFMAs issue (the product loop's register double buffer)
                na = *reinterpret_cast<const f32x4*>(la + (a_row ^ (qq << 4)));
                nb0 = *reinterpret_cast<const f32x4*>(la + (b_row ^ (qq << 4)));
                nb1 = *reinterpret_cast<const f32x4*>(la + (b_row ^ (qq << 4)) + 4096);
                if (!prefetch) { va = na; vb0 = nb0; vb1 = nb1; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (FORM == 0) {          // 32x32x2: four 16-register accumulators = the same 64 x 64 wave tile, k pair per instruction -> half the instructions per k
                    if (e & 1) continue;
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[e], vb0[e], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[e], vb1[e], c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[e + 1], vb0[e + 1], c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[e + 1], vb1[e + 1], c3, 0, 0, 0);
                } else if (FORM == 1) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x1f32(va[e], vb0[e], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x1f32(va[e], vb1[e], acc1, 0, 0, 0);
                } else {
                    asm volatile("v_mfma_f32_32x32x1_2b_f32 %0, %1, %2, %0" : "+a"(acc0) : "v"(va[e]), "v"(vb0[e]));
                    asm volatile("v_mfma_f32_32x32x1_2b_f32 %0, %1, %2, %0" : "+a"(acc1) : "v"(va[e]), "v"(vb1[e]));
                }
            }
            if (LDSOP && prefetch) { va = na; vb0 = nb0; vb1 = nb1; }
        }
    }
    float s = 0;
    for (int r = 0; r < 32; ++r) s += acc0[r] + acc1[r];
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (tid == 0 && blockIdx.x == 0) { g_stamp[0] = __builtin_readcyclecounter() - t0; g_stamp[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

static int g_threads = 512, g_prefetch = 0;
template <typename K> void run(const char* name, K kern, float* out, const float* seed, int iters)
{
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(kern, dim3(256), dim3(g_threads), 0, 0, out, seed, iters, g_prefetch); hipDeviceSynchronize();
    hipEventRecord(s);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(g_threads), 0, 0, out, seed, iters, g_prefetch);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); ms /= 3;
    unsigned long long st[2]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof(st));
    const double fl = 2.0 * 64 * 64 * 32 * (double)iters * (g_threads / 64) * 256;       // a 64 x 64 x 32 wave tile per iteration
    printf("[%d waves/SIMD%s] %-60s %7.1f TF  (%.3f of 157.3)  clock %.2f GHz\n", g_threads / 256, g_prefetch ? ", prefetch" : "", name, fl / ms / 1e9, fl / ms / 1e9 / 157.3, (double)st[0] / ((double)st[1] * 10.0) );
}

int main()
{
    float *out, *seed; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&seed, 144 * 1024);
    float* h = (float*)malloc(144 * 1024);
    srand(1);
    for (int i = 0; i < 144 * 1024 / 4; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(seed, h, 144 * 1024, hipMemcpyHostToDevice);
    const int it = 3000;
    for (int threads = 512; threads >= 256; threads -= 256) {
        g_threads = threads;
        for (g_prefetch = 0; g_prefetch < 2; ++g_prefetch) {
            if (!g_prefetch) {
                run("32x32x2     VGPR acc, constant operands", loop_kernel<0, false, false>, out, seed, it);
                run("32x32x1_2b  VGPR acc, constant operands", loop_kernel<1, false, false>, out, seed, it);
                run("32x32x1_2b  AGPR acc, constant operands", loop_kernel<2, false, false>, out, seed, it);
            }
            run("32x32x2     VGPR acc, operands from LDS (random)", loop_kernel<0, true, false>, out, seed, it);
            run("32x32x1_2b  VGPR acc, operands from LDS (random)", loop_kernel<1, true, false>, out, seed, it);
            run("32x32x1_2b  AGPR acc, operands from LDS (random)", loop_kernel<2, true, false>, out, seed, it);
            run("32x32x2     VGPR acc, operands from LDS, barrier per 64 MFMA-k", loop_kernel<0, true, true>, out, seed, it);
            run("32x32x1_2b  VGPR acc, operands from LDS, barrier per 64 MFMAs", loop_kernel<1, true, true>, out, seed, it);
        }
    }
    return 0;
}
