// What a streaming kernel can reach on THIS box (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy; round 1 measured 4.75 TB/s
// with one configuration): read+write bandwidth of a float4 copy and of a 2-read / 2-write "LN-shaped" stream over grid sizes,
// loads in flight per thread, and store policy.  Build: hipcc --offload-arch=gfx950 -O3 -o hbm_sweep hbm_sweep.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_k(const f4v* __restrict__ in, f4v* __restrict__ out, long n4)
{
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4v v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], &out[i + u * stride]); else out[i + u * stride] = v[u]; }
    }
    for (; i < n4; i += stride) out[i] = in[i];
}

// two inputs, two outputs (x, y -> x + y, (x + y) * 0.5): the traffic mix of residual_ln_mod
template <int U, bool NT>
__global__ __launch_bounds__(256) void rw22_k(const f4v* __restrict__ a, const f4v* __restrict__ b, f4v* __restrict__ o1, f4v* __restrict__ o2, long n4)
{
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4v x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { x[u] = a[i + u * stride]; y[u] = b[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f4v s = x[u] + y[u], h = s * 0.5f;
            if (NT) { __builtin_nontemporal_store(s, &o1[i + u * stride]); __builtin_nontemporal_store(h, &o2[i + u * stride]); }
            else { o1[i + u * stride] = s; o2[i + u * stride] = h; }
        }
    }
}

template <typename F>
static float time_ms(F f, int n = 10)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); f();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < n; ++i) f();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / n;
}

int main()
{
    const long BYTES = 141l << 20;           // one [64, 358, 1536] fp32 tensor ~ 141 MB; the LN stream touches four of them
    const long n4 = BYTES / 16;
    f4v *a, *b, *c, *d;
    CK(hipMalloc(&a, BYTES)); CK(hipMalloc(&b, BYTES)); CK(hipMalloc(&c, BYTES)); CK(hipMalloc(&d, BYTES));
    CK(hipMemset(a, 1, BYTES)); CK(hipMemset(b, 2, BYTES));
    const int grids[] = {1024, 2048, 4096, 8192, 16384, 36096};
    printf("float4 copy, %ld MB in + %ld MB out (TB/s read+write)\n", BYTES >> 20, BYTES >> 20);
    for (int g : grids) {
        float t1 = time_ms([&] { hipLaunchKernelGGL((copy_k<1, false>), dim3(g), dim3(256), 0, 0, a, c, n4); });
        float t2 = time_ms([&] { hipLaunchKernelGGL((copy_k<4, false>), dim3(g), dim3(256), 0, 0, a, c, n4); });
        float t3 = time_ms([&] { hipLaunchKernelGGL((copy_k<4, true>), dim3(g), dim3(256), 0, 0, a, c, n4); });
        float t4 = time_ms([&] { hipLaunchKernelGGL((copy_k<8, true>), dim3(g), dim3(256), 0, 0, a, c, n4); });
        printf("  grid %6d: U1 %.2f  U4 %.2f  U4+nt %.2f  U8+nt %.2f\n", g, 2.0 * BYTES / t1 / 1e9, 2.0 * BYTES / t2 / 1e9, 2.0 * BYTES / t3 / 1e9, 2.0 * BYTES / t4 / 1e9);
    }
    printf("2 reads + 2 writes of %ld MB each (TB/s)\n", BYTES >> 20);
    for (int g : grids) {
        float t1 = time_ms([&] { hipLaunchKernelGGL((rw22_k<1, false>), dim3(g), dim3(256), 0, 0, a, b, c, d, n4); });
        float t2 = time_ms([&] { hipLaunchKernelGGL((rw22_k<2, false>), dim3(g), dim3(256), 0, 0, a, b, c, d, n4); });
        float t3 = time_ms([&] { hipLaunchKernelGGL((rw22_k<2, true>), dim3(g), dim3(256), 0, 0, a, b, c, d, n4); });
        float t4 = time_ms([&] { hipLaunchKernelGGL((rw22_k<4, true>), dim3(g), dim3(256), 0, 0, a, b, c, d, n4); });
        printf("  grid %6d: U1 %.2f  U2 %.2f  U2+nt %.2f  U4+nt %.2f\n", g, 4.0 * BYTES / t1 / 1e9, 4.0 * BYTES / t2 / 1e9, 4.0 * BYTES / t3 / 1e9, 4.0 * BYTES / t4 / 1e9);
    }
    return 0;
}
