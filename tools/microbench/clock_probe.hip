// One wave that samples (s_memrealtime [100 MHz], s_memtime [shader clock]) every `gap` s_sleep units while other kernels run beside it on another stream:
// the shader clock the chip actually holds under a given kernel (round 6: does it throttle under this repo's fp32-MFMA GEMMs and not under the vendor's?).
// Built and driven by tools/clock_probe.py.
#include <hip/hip_runtime.h>
extern "C" __global__ void clock_probe_kernel(unsigned long long* buf, int n, int gap)
{
    if (threadIdx.x != 0) return;
    for (int i = 0; i < n; ++i) {
        buf[2 * i] = __builtin_amdgcn_s_memrealtime();
        buf[2 * i + 1] = __builtin_readcyclecounter();
        for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(127);
    }
}
extern "C" int clock_probe_launch(unsigned long long* buf, int n, int gap, void* stream)
{
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, buf, n, gap);
    return (int)hipGetLastError();
}
