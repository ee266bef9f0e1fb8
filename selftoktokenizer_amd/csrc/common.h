// Shared helpers for the Selftok gfx950 kernels.  CDNA4 only: wave64, no portability shims.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SELFTOK_OK 0
#define SELFTOK_EINVAL (-1)
#define SELFTOK_EHIP (-2)

#define WAVE 64

namespace selftok {

void set_last_error(const char* msg);
int check_launch(const char* what);

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

// monotone map float -> uint32 for every non-NaN value (after -0 -> +0 canonicalisation by the caller)
__device__ __forceinline__ uint32_t f32_orderable(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_orderable(uint32_t k)
{
    uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
    return v;
}

// "Split activation" layout (include/selftok_hip.h): a [rows, K] fp32 tensor as fp16 hi / lo planes, stored in 1-KiB chunks of
// 16 rows x 32 k per plane, [rows/16][K/32][plane][16][32] -- one chunk is exactly one LDS-DMA piece of the consuming GEMM (8 full
// cache lines; with row-major planes a piece was 16 half lines, each fetched twice).  Index in halfs of element (row, k), plane p:
__device__ __forceinline__ size_t split_blk_index(long row, int k, int plane, int KT)
{
    return ((((size_t)(row >> 4) * KT + (k >> 5)) * 2 + plane) << 9) + ((row & 15) << 5) + (k & 31);
}

// The fp32 value `v`, made opaque to the optimiser.  hipcc folds fptrunc(fmul/fadd) into one v_fma_mixlo_f16, i.e. it rounds
// the EXACT product or sum to fp16 once instead of rounding the fp32 result -- a different fp16 value whenever the fp32 result
// sits on an fp16 rounding tie.  Every producer of a "split activation" passes its fp32 result through here first, so that the
// planes are a function of the fp32 value alone (bit-identical to splitting the stored fp32 tensor).
__device__ __forceinline__ float opaque_f32(float v)
{
    asm("" : "+v"(v));
    return v;
}

// tools/ builds only (-DSELFTOK_TUNE): shader-clock (s_memtime) and 100 MHz wall-clock (s_memrealtime) stamps of the first workgroup
// of a kernel -> effective shader clock while that kernel runs (the chip clocks to its power budget).  The product build has none.
#ifdef SELFTOK_TUNE
// sym[0..1]: workgroup (0,0)'s shader cycles and 100 MHz ticks; sym[2 + 3 w ..]: (start tick, end tick, XCC id << 8 | CU-ish hw id) of
// workgroup w (first 4096 workgroups) -- a residency census: how many workgroups are alive at once, and where
#define SELFTOK_STAMP_DECL(sym) __device__ unsigned long long sym[2 + 3 * 4096]
#define SELFTOK_STAMP_BEGIN()                                                                                    \
    unsigned long long stamp_tk0_ = 0, stamp_rt0_ = 0;                                                          \
    const bool stamp_on_ = threadIdx.x == 0;                                                                     \
    if (stamp_on_) { stamp_tk0_ = __builtin_readcyclecounter(); stamp_rt0_ = __builtin_amdgcn_s_memrealtime(); }
#define SELFTOK_STAMP_END(sym)                                                                                   \
    if (stamp_on_) {                                                                                             \
        const unsigned long long rt1_ = __builtin_amdgcn_s_memrealtime();                                        \
        const unsigned wg_ = blockIdx.x + blockIdx.y * gridDim.x;                                                \
        if (wg_ == 0) { sym[0] = __builtin_readcyclecounter() - stamp_tk0_; sym[1] = rt1_ - stamp_rt0_; }        \
        if (wg_ < 4096) { unsigned hw_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));         \
            unsigned xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                   \
            sym[2 + 3 * wg_] = stamp_rt0_; sym[3 + 3 * wg_] = rt1_; sym[4 + 3 * wg_] = ((unsigned long long)(xcc_ & 0xF) << 32) | hw_; }  \
    }
#else
#define SELFTOK_STAMP_DECL(sym)
#define SELFTOK_STAMP_BEGIN()
#define SELFTOK_STAMP_END(sym)
#endif

}  // namespace selftok
