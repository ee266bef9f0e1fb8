// Which fp32-MFMA loop shapes does this chip run at its full shader clock?  (round 6: tools/clock_probe.py shows the vendor's fp32 GEMM holding 2.39 GHz at
// 0.925 of the 157.3 TF peak while csrc/gemm_fp32.hip -- 0.92 in cycles -- is throttled to 2.20 - 2.27 GHz.)  The skeleton of csrc/gemm_fp32.hip with the data path kept
// (3 x 48 KiB LDS stages filled by two LDS-DMA loader waves from an L2-resident buffer, one barrier per 32-k chunk, fragments read with ds_read_b128 from random data)
// and the MFMA form / wave tile / accumulator file switchable:
//   form 0: 32x32x1 (2 blocks), 8 compute waves of 64 x 64, accumulators in VGPRs   (csrc/gemm_fp32.hip v11)
//   form 1: same, accumulators pinned to AGPRs
//   form 2: 32x32x2, 8 compute waves of 64 x 64 (4 accumulators of 16), VGPRs
//   form 3: 32x32x2, 4 compute waves of 64 x 128 (8 accumulators of 16), VGPRs
//   form 4: same, AGPRs
//   form 5: 16x16x4, 4 compute waves of 64 x 128 (32 accumulators of 4), VGPRs      (the vendor kernel's shape; k order inside a 16-k group is permuted: free order only)
//   form 6: same, AGPRs
//   form 15 / 16: forms 7 / 8 on a layout that makes ds_read2_b32 conflict free (DMA pieces staggered by 4 bytes, swizzle by row & 7)
//   form 11 / 12: forms 2 / 3 fed k-ascending from ALIGNED b128 reads: lane (r, h) reads k = 8 t + 4 h .. + 3, two v_permlane32_swap pair them up
//   form 13 / 14: forms 2 / 3 fed k-ascending by ds_read_b64 from operands stored k-interleaved (k0 k2 k1 k3 per aligned four) in memory
//   form 9 / 10: forms 2 / 3 with lanes 32..63 reading their ds_read_b128 at +4 bytes (elements [0], [2] = k + h, k + 2 + h): k ascending at b128 cost?
//   form 7 / 8: forms 2 / 3 with each fragment's operands read by ds_read2_b32 (elements h and 2 + h of a 4-k group, h = lane / 32): the feed a k-ASCENDING chain needs
// Reports TFLOP/s of the MFMAs issued, and the shader clock (s_memtime cycles per 100 MHz s_memrealtime tick, workgroup 0).   Results are garbage numbers.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_power_forms mfma_power_forms.hip ; run: ./mfma_power_forms [iters] [dma 0|1]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#pragma clang diagnostic ignored "-Winline-asm"
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // a ds_read_b128 at a 4-byte aligned address (gfx950 runs LDS in unaligned mode)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

constexpr int STAGE = 48 * 1024, STAGES = 3;
__device__ unsigned long long g_stamp[2];

__device__ __forceinline__ void dma16(const void* base, unsigned voff, unsigned lds)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory", "m0");
}

#define MFMA_A(op, acc, a, b) asm volatile(op " %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

template <int FORM>
__global__ __launch_bounds__((FORM <= 2 || FORM == 7 || FORM == 9 || FORM == 11 || FORM == 13 || FORM == 15) ? 640 : 384) void loop_kernel(float* out, const float* src, int iters, int dma)
{
    constexpr int CW = (FORM <= 2 || FORM == 7 || FORM == 9 || FORM == 11 || FORM == 13 || FORM == 15) ? 8 : 4;
    __shared__ __attribute__((aligned(1024))) char lds[STAGES * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < STAGES * STAGE / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = src[(i * 7 + blockIdx.x * 13) & 0xFFFFF];
    __syncthreads();
    unsigned long long t0 = 0, r0 = 0;
    if (tid == 0) { t0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
    if (wave >= CW) {
        // loader waves: wave CW 32 pieces (the A tile's 32 KiB), wave CW + 1 16 pieces (B), per chunk, from a 12 MiB window every workgroup walks in step (L2 / MALL hits,
        // as a GEMM's tiles are shared between the workgroups of a round)
        const bool isA = wave == CW;
        const unsigned lbase = (unsigned)(size_t)lds + (isA ? 0 : 32768);
        if (wave > CW + 1) { for (int it = 0; it < iters; ++it) __syncthreads(); return; }
        for (int it = 0; it < iters; ++it) {
            if (dma) {
                const char* cb = reinterpret_cast<const char*>(src) + ((size_t)(it % 250) * STAGE) + (isA ? 0 : 32768);
                const unsigned l = lbase + ((it + 2) % STAGES) * STAGE;
                if (isA) {
#pragma unroll
                    for (int p = 0; p < 32; ++p) dma16(cb, p * 1024 + lane * 16, l + p * 1024);
                    asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                } else {
#pragma unroll
                    for (int p = 0; p < 16; ++p) dma16(cb, p * 1024 + lane * 16, l + p * 1024);
                    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                }
            }
            __syncthreads();
        }
        return;
    }
    float s = 0;
    if (FORM <= 1) {
        const int a_row = (64 * (wave & 3) + lane) * 128 + (((lane >> 1) & 7) << 4);
        const int b_row = 32768 + (64 * (wave >> 2) + (lane & 31)) * 128 + (((lane >> 1) & 7) << 4);
        f32x32 acc0 = {0}, acc1 = {0};
        for (int it = 0; it < iters; ++it) {
            const char* la = lds + (it % STAGES) * STAGE;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 va = *reinterpret_cast<const f32x4*>(la + (a_row ^ (q << 4)));
                const f32x4 vb0 = *reinterpret_cast<const f32x4*>(la + (b_row ^ (q << 4)));
                const f32x4 vb1 = *reinterpret_cast<const f32x4*>(la + (b_row ^ (q << 4)) + 4096);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (FORM == 0) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x1f32(va[e], vb0[e], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x1f32(va[e], vb1[e], acc1, 0, 0, 0);
                    } else {
                        MFMA_A("v_mfma_f32_32x32x1_2b_f32", acc0, va[e], vb0[e]);
                        MFMA_A("v_mfma_f32_32x32x1_2b_f32", acc1, va[e], vb1[e]);
                    }
                }
            }
        }
        for (int r = 0; r < 32; ++r) s += acc0[r] + acc1[r];
    } else if (FORM == 2) {
        // 8 waves of 64 x 64 on 32x32x2: 2 A fragments x 2 B fragments; a b128 read (4 consecutive k) feeds two MFMA steps (elements [0], [2] of a read that starts at k + lane/32)
        const int a_row = (64 * (wave & 3) + (lane & 31)) * 128 + (((lane >> 1) & 7) << 4);
        const int b_row = 32768 + (64 * (wave >> 2) + (lane & 31)) * 128 + (((lane >> 1) & 7) << 4);
        f32x16 c[4] = {{0}, {0}, {0}, {0}};
        for (int it = 0; it < iters; ++it) {
            const char* la = lds + (it % STAGES) * STAGE;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                f32x4 va[2], vb[2];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    va[f] = *reinterpret_cast<const f32x4*>(la + (a_row ^ (q << 4)) + f * 4096);
                    vb[f] = *reinterpret_cast<const f32x4*>(la + (b_row ^ (q << 4)) + f * 4096);
                }
#pragma unroll
                for (int e = 0; e < 4; e += 2)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) c[2 * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[i][e], vb[j][e], c[2 * i + j], 0, 0, 0);
            }
        }
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += c[k][r];
    } else if (FORM == 11 || FORM == 12) {
        // k ascending from aligned b128 reads: lane (r, h) reads k = 8 t + 4 h .. + 3; v_permlane32_swap of registers (0, 1) and (2, 3) leaves
        // [lanes 0..31: k | lanes 32..63: k + 1] pairs for the four MFMA steps of the 8-k group -- one b128 + two swaps per fragment per 8 k
        constexpr int NA = 2, NB = (FORM == 11) ? 2 : 4;
        const int hsw = (lane >> 5) << 4;
        const int a_row = (64 * (FORM == 11 ? (wave & 3) : wave) + (lane & 31)) * 128;
        const int b_row = 32768 + ((FORM == 11 ? 64 * (wave >> 2) : 0) + (lane & 31)) * 128;
        const int sw = ((lane >> 1) & 7) << 4;
        f32x16 c[NA * NB];
#pragma unroll
        for (int k = 0; k < NA * NB; ++k) c[k] = f32x16{0};
        for (int it = 0; it < iters; ++it) {
            const char* la = lds + (it % STAGES) * STAGE;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float va[NA][4], vb[NB][4];
#pragma unroll
                for (int f = 0; f < NA + NB; ++f) {
                    const int row = f < NA ? a_row + f * 4096 : b_row + (f - NA) * 4096;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(la + row + (((q << 5) | hsw) ^ sw));
                    const auto p01 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]), false, false);
                    const auto p23 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3]), false, false);
                    float* d = f < NA ? va[f] : vb[f - NA];
                    d[0] = __builtin_bit_cast(float, p01[0]); d[2] = __builtin_bit_cast(float, p01[1]);
                    d[1] = __builtin_bit_cast(float, p23[0]); d[3] = __builtin_bit_cast(float, p23[1]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < NA; ++i)
#pragma unroll
                        for (int j = 0; j < NB; ++j) c[NB * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[i][e], vb[j][e], c[NB * i + j], 0, 0, 0);
            }
        }
        for (int k = 0; k < NA * NB; ++k) for (int r = 0; r < 16; ++r) s += c[k][r];
    } else if (FORM == 13 || FORM == 14) {
        // k ascending from ds_read_b64, given operands stored k-INTERLEAVED in memory (each aligned group of four k kept as k0 k2 k1 k3: the producer of X swaps two
        // elements of every float4 it writes, W is re-laid once): lane (r, h) reads the 8 bytes at 4 t + 2 h = (k + h, k + 2 + h)
        constexpr int NA = 2, NB = (FORM == 13) ? 2 : 4;
        const int a_row = (64 * (FORM == 13 ? (wave & 3) : wave) + (lane & 31)) * 128 + (lane >> 5) * 8;
        const int b_row = 32768 + ((FORM == 13 ? 64 * (wave >> 2) : 0) + (lane & 31)) * 128 + (lane >> 5) * 8;
        const int sw = ((lane >> 1) & 7) << 4;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x16 c[NA * NB];
#pragma unroll
        for (int k = 0; k < NA * NB; ++k) c[k] = f32x16{0};
        for (int it = 0; it < iters; ++it) {
            const char* la = lds + (it % STAGES) * STAGE;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                f32x2 va[NA], vb[NB];
#pragma unroll
                for (int f = 0; f < NA; ++f) va[f] = *reinterpret_cast<const f32x2*>(la + a_row + f * 4096 + ((q << 4) ^ sw));
#pragma unroll
                for (int f = 0; f < NB; ++f) vb[f] = *reinterpret_cast<const f32x2*>(la + b_row + f * 4096 + ((q << 4) ^ sw));
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int i = 0; i < NA; ++i)
#pragma unroll
                        for (int j = 0; j < NB; ++j) c[NB * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[i][e], vb[j][e], c[NB * i + j], 0, 0, 0);
            }
        }
        for (int k = 0; k < NA * NB; ++k) for (int r = 0; r < 16; ++r) s += c[k][r];
    } else if (FORM == 9) {
        // form 2 with lanes 32..63 reading their b128 4 bytes further on: elements [0], [2] are k = 4 t + h and 4 t + 2 + h (k ascending), [1], [3] unused
        const int a_row = (64 * (wave & 3) + (lane & 31)) * 128 + (((lane >> 1) & 7) << 4);
        const int b_row = 32768 + (64 * (wave >> 2) + (lane & 31)) * 128 + (((lane >> 1) & 7) << 4);
        const int hh = (lane >> 5) * 4;
        f32x16 c[4] = {{0}, {0}, {0}, {0}};
        for (int it = 0; it < iters; ++it) {
            const char* la = lds + (it % STAGES) * STAGE;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                f32x4 va[2], vb[2];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    va[f] = *reinterpret_cast<const f32x4u*>(la + ((a_row ^ (q << 4)) + hh) + f * 4096);
                    vb[f] = *reinterpret_cast<const f32x4u*>(la + ((b_row ^ (q << 4)) + hh) + f * 4096);
                }
#pragma unroll
                for (int e = 0; e < 4; e += 2)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) c[2 * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[i][e], vb[j][e], c[2 * i + j], 0, 0, 0);
            }
        }
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += c[k][r];
    } else if (FORM == 10) {
        const int a_row = (64 * wave + (lane & 31)) * 128 + (((lane >> 1) & 7) << 4);
        const int b_row = 32768 + (lane & 31) * 128 + (((lane >> 1) & 7) << 4);
        const int hh = (lane >> 5) * 4;
        f32x16 c[8] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}};
        for (int it = 0; it < iters; ++it) {
            const char* la = lds + (it % STAGES) * STAGE;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                f32x4 va[2], vb[4];
#pragma unroll
                for (int f = 0; f < 2; ++f) va[f] = *reinterpret_cast<const f32x4u*>(la + ((a_row ^ (q << 4)) + hh) + f * 4096);
#pragma unroll
                for (int f = 0; f < 4; ++f) vb[f] = *reinterpret_cast<const f32x4u*>(la + ((b_row ^ (q << 4)) + hh) + f * 4096);
#pragma unroll
                for (int e = 0; e < 4; e += 2)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) c[4 * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[i][e], vb[j][e], c[4 * i + j], 0, 0, 0);
            }
        }
        for (int k = 0; k < 8; ++k) for (int r = 0; r < 16; ++r) s += c[k][r];
    } else if (FORM == 15 || FORM == 16) {
        // forms 7 / 8 with a layout in which ds_read2_b32 is conflict free: the 1-KiB DMA piece that holds rows 8 p .. 8 p + 7 lands 4 (p % 4) bytes further on (M0 is a
        // byte address) and a row's 16-byte granules are swizzled by row & 7 -- the 32 rows of a fragment then cover 8 granule positions x 4 dword residues = all 32 banks
        constexpr int NA = 2, NB = (FORM == 15) ? 2 : 4;
        const int r = lane & 31;
        const int lane_off = r * 128 + ((r >> 3) & 3) * 4 + (lane >> 5) * 4;
        const int a_row = (64 * (FORM == 15 ? (wave & 3) : wave)) * 128 + lane_off;
        const int b_row = 32768 + (FORM == 15 ? 64 * (wave >> 2) : 0) * 128 + lane_off;
        const int sw = (r & 7) << 4;
        f32x16 c[NA * NB];
#pragma unroll
        for (int k = 0; k < NA * NB; ++k) c[k] = f32x16{0};
        for (int it = 0; it < iters; ++it) {
            const char* la = lds + (it % STAGES) * STAGE;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float va[NA][2], vb[NB][2];
#pragma unroll
                for (int f = 0; f < NA; ++f) { const float* pa = reinterpret_cast<const float*>(la + a_row + f * 4096 + ((q << 4) ^ sw)); va[f][0] = pa[0]; va[f][1] = pa[2]; }
#pragma unroll
                for (int f = 0; f < NB; ++f) { const float* pb = reinterpret_cast<const float*>(la + b_row + f * 4096 + ((q << 4) ^ sw)); vb[f][0] = pb[0]; vb[f][1] = pb[2]; }
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int i = 0; i < NA; ++i)
#pragma unroll
                        for (int j = 0; j < NB; ++j) c[NB * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[i][e], vb[j][e], c[NB * i + j], 0, 0, 0);
            }
        }
        for (int k = 0; k < NA * NB; ++k) for (int rr = 0; rr < 16; ++rr) s += c[k][rr];
    } else if (FORM == 7) {
        // form 2 with the operands read as ds_read2_b32 (elements h and 2 + h of the 4-k group): what a k-ascending chain needs from 32x32x2
        const int a_row = (64 * (wave & 3) + (lane & 31)) * 128 + (((lane >> 1) & 7) << 4) + (lane >> 5) * 4;
        const int b_row = 32768 + (64 * (wave >> 2) + (lane & 31)) * 128 + (((lane >> 1) & 7) << 4) + (lane >> 5) * 4;
        f32x16 c[4] = {{0}, {0}, {0}, {0}};
        for (int it = 0; it < iters; ++it) {
            const char* la = lds + (it % STAGES) * STAGE;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float va[2][2], vb[2][2];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const float* pa = reinterpret_cast<const float*>(la + (a_row ^ (q << 4)) + f * 4096);
                    const float* pb = reinterpret_cast<const float*>(la + (b_row ^ (q << 4)) + f * 4096);
                    va[f][0] = pa[0]; va[f][1] = pa[2]; vb[f][0] = pb[0]; vb[f][1] = pb[2];
                }
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) c[2 * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[i][e], vb[j][e], c[2 * i + j], 0, 0, 0);
            }
        }
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += c[k][r];
    } else if (FORM == 8) {
        const int a_row = (64 * wave + (lane & 31)) * 128 + (((lane >> 1) & 7) << 4) + (lane >> 5) * 4;
        const int b_row = 32768 + (lane & 31) * 128 + (((lane >> 1) & 7) << 4) + (lane >> 5) * 4;
        f32x16 c[8] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}};
        for (int it = 0; it < iters; ++it) {
            const char* la = lds + (it % STAGES) * STAGE;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float va[2][2], vb[4][2];
#pragma unroll
                for (int f = 0; f < 2; ++f) { const float* pa = reinterpret_cast<const float*>(la + (a_row ^ (q << 4)) + f * 4096); va[f][0] = pa[0]; va[f][1] = pa[2]; }
#pragma unroll
                for (int f = 0; f < 4; ++f) { const float* pb = reinterpret_cast<const float*>(la + (b_row ^ (q << 4)) + f * 4096); vb[f][0] = pb[0]; vb[f][1] = pb[2]; }
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) c[4 * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[i][e], vb[j][e], c[4 * i + j], 0, 0, 0);
            }
        }
        for (int k = 0; k < 8; ++k) for (int r = 0; r < 16; ++r) s += c[k][r];
    } else if (FORM == 3 || FORM == 4) {
        // 4 waves of 64 x 128 on 32x32x2: 2 A x 4 B fragments, 8 accumulators of 16
        const int a_row = (64 * wave + (lane & 31)) * 128 + (((lane >> 1) & 7) << 4);
        const int b_row = 32768 + (lane & 31) * 128 + (((lane >> 1) & 7) << 4);
        f32x16 c[8] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}};
        for (int it = 0; it < iters; ++it) {
            const char* la = lds + (it % STAGES) * STAGE;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                f32x4 va[2], vb[4];
#pragma unroll
                for (int f = 0; f < 2; ++f) va[f] = *reinterpret_cast<const f32x4*>(la + (a_row ^ (q << 4)) + f * 4096);
#pragma unroll
                for (int f = 0; f < 4; ++f) vb[f] = *reinterpret_cast<const f32x4*>(la + (b_row ^ (q << 4)) + f * 4096);
#pragma unroll
                for (int e = 0; e < 4; e += 2)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (FORM == 3) c[4 * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[i][e], vb[j][e], c[4 * i + j], 0, 0, 0);
                            else MFMA_A("v_mfma_f32_32x32x2_f32", c[4 * i + j], va[i][e], vb[j][e]);
                        }
            }
        }
        for (int k = 0; k < 8; ++k) for (int r = 0; r < 16; ++r) s += c[k][r];
    } else {
        // 4 waves of 64 x 128 on 16x16x4: 4 A x 8 B fragments, 32 accumulators of 4; a b128 read feeds four k4 steps (16 k)
        const int a_row = (64 * wave + (lane & 15)) * 128 + (lane >> 4) * 16;
        const int b_row = 32768 + (lane & 15) * 128 + (lane >> 4) * 16;
        f32x4 c[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) c[k] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
            const char* la = lds + (it % STAGES) * STAGE;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f32x4 va[4], vb[8];
#pragma unroll
                for (int f = 0; f < 4; ++f) va[f] = *reinterpret_cast<const f32x4*>(la + ((a_row + f * 2048) ^ (((lane & 7)) << 4)) + q * 64);
#pragma unroll
                for (int f = 0; f < 8; ++f) vb[f] = *reinterpret_cast<const f32x4*>(la + ((b_row + f * 2048) ^ (((lane & 7)) << 4)) + q * 64);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            if (FORM == 5) c[8 * i + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[i][e], vb[j][e], c[8 * i + j], 0, 0, 0);
                            else MFMA_A("v_mfma_f32_16x16x4_f32", c[8 * i + j], va[i][e], vb[j][e]);
                        }
            }
        }
        for (int k = 0; k < 32; ++k) for (int r = 0; r < 4; ++r) s += c[k][r];
    }
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) { g_stamp[0] = __builtin_readcyclecounter() - t0; g_stamp[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

template <typename K> void run(const char* name, K kern, float* out, const float* src, int iters, int dma, int threads = 384)
{
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, src, iters, dma);
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, src, iters, dma);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); ms /= 3;
    unsigned long long st[2]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof(st));
    const double fl = 2.0 * 256 * 128 * 32 * (double)iters * 256;       // a 256 x 128 x 32 chunk per iteration per workgroup
    const double ghz = (double)st[0] / ((double)st[1] * 10.0);
    printf("%-78s %6.2f ms %6.1f TF  (%.3f of 157.3)  clock %.3f GHz  -> %.3f of the peak at that clock\n", name, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3, ghz,
           fl / ms / 1e9 / (157.3 * ghz / 2.4));
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 6000;
    float *out, *src; hipMalloc(&out, 256 * 640 * 4); hipMalloc(&src, 16 << 20);
    float* h = (float*)malloc(16 << 20);
    srand(1);
    for (int i = 0; i < (16 << 20) / 4; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(src, h, 16 << 20, hipMemcpyHostToDevice);
    for (int dma = 1; dma >= 0; --dma) {
        printf("--- LDS-DMA loader waves %s\n", dma ? "ON (48 KiB per chunk per CU)" : "off (barriers only)");
        run("0: 32x32x1_2b, 8 waves of 64x64, VGPR acc (gemm_fp32.hip v11)", loop_kernel<0>, out, src, iters, dma, 640);
        run("1: 32x32x1_2b, 8 waves of 64x64, AGPR acc", loop_kernel<1>, out, src, iters, dma, 640);
        run("2: 32x32x2,    8 waves of 64x64, VGPR acc", loop_kernel<2>, out, src, iters, dma, 640);
        run("3: 32x32x2,    4 waves of 64x128, VGPR acc", loop_kernel<3>, out, src, iters, dma);
        run("4: 32x32x2,    4 waves of 64x128, AGPR acc", loop_kernel<4>, out, src, iters, dma);
        run("5: 16x16x4,    4 waves of 64x128, VGPR acc (k permuted inside 16)", loop_kernel<5>, out, src, iters, dma);
        run("6: 16x16x4,    4 waves of 64x128, AGPR acc (the vendor kernel's shape)", loop_kernel<6>, out, src, iters, dma);
        run("7: 32x32x2,    8 waves of 64x64, VGPR acc, ds_read2_b32 feed (k ascending)", loop_kernel<7>, out, src, iters, dma, 640);
        run("8: 32x32x2,    4 waves of 64x128, VGPR acc, ds_read2_b32 feed (k ascending)", loop_kernel<8>, out, src, iters, dma);
        run("15: 32x32x2,   8 waves of 64x64, VGPR acc, ds_read2_b32 from a dword-STAGGERED layout (k ascending)", loop_kernel<15>, out, src, iters, dma, 640);
        run("16: 32x32x2,   4 waves of 64x128, VGPR acc, ds_read2_b32 from a dword-STAGGERED layout (k ascending)", loop_kernel<16>, out, src, iters, dma);
        run("11: 32x32x2,   8 waves of 64x64, VGPR acc, b128 per 8 k + 2 permlane32_swap (k ascending)", loop_kernel<11>, out, src, iters, dma, 640);
        run("12: 32x32x2,   4 waves of 64x128, VGPR acc, b128 per 8 k + 2 permlane32_swap (k ascending)", loop_kernel<12>, out, src, iters, dma);
        run("13: 32x32x2,   8 waves of 64x64, VGPR acc, ds_read_b64 of k-interleaved operands (k ascending)", loop_kernel<13>, out, src, iters, dma, 640);
        run("14: 32x32x2,   4 waves of 64x128, VGPR acc, ds_read_b64 of k-interleaved operands (k ascending)", loop_kernel<14>, out, src, iters, dma);
        run("9: 32x32x2,    8 waves of 64x64, VGPR acc, b128 at +4 bytes for lanes 32..63 (k ascending)", loop_kernel<9>, out, src, iters, dma, 640);
        run("10: 32x32x2,   4 waves of 64x128, VGPR acc, b128 at +4 bytes for lanes 32..63 (k ascending)", loop_kernel<10>, out, src, iters, dma);
    }
    return 0;
}
