// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the VQ kernels (VERDICT r2 item 1:
// "calibrate the FETCH correction on the finalize kernel's access pattern before doubling it").  MI355X_MICROARCH.md: FETCH_SIZE
// reports 1/2 of the bytes of a wide (16 B/lane) coalesced streaming read; other widths and WRITE_SIZE are uncalibrated.
// Every kernel below touches a KNOWN number of distinct bytes exactly once, from a buffer no earlier launch has touched in that
// pass order (the host re-fills a 512 MiB scratch between launches to evict L2 / Infinity Cache), in one of the patterns of
// csrc/vq.hip:
//   calib_read_stream16 : 16 B per lane, coalesced                     (vq_f16_kernel: z rows, LDS-DMA of the code tiles)
//   calib_read_cand8    : 8 B per lane, entry s of row r at (s*N + r)*8, 16 lanes per row  (vq_finalize_f16_kernel: candidates)
//   calib_read_tile4    : 4 B per lane, the 16 codes of a (tile, half) in fragment order   (vq_finalize_f16_kernel: exact re-score)
//   calib_write_cand8   : 8 B per lane, 32 consecutive rows per wave half                  (vq_f16_kernel: candidate store)
//   calib_write_stream16: 16 B per lane, coalesced
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip ; run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   and   rocprofv3 --pmc WRITE_SIZE --kernel-trace ...
// and divide the known byte counts (printed) by the counters: tools/pmc_vq_traffic.sh does that.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void calib_read_stream16(const float4* __restrict__ in, float* __restrict__ out, long n4)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    float s = 0.f;
    for (; i < n4; i += stride) { float4 v = in[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) out[0] = s;
}

// entries [S][N] of 8 bytes; 16 lanes per row, lane gl reads entries gl, gl+16, ... of its row (vq_finalize_f16_kernel's first loop)
__global__ __launch_bounds__(256) void calib_read_cand8(const unsigned long long* __restrict__ partial, float* __restrict__ out, int N, int S)
{
    const int gl = threadIdx.x & 15;
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (r >= N) return;
    unsigned long long acc = 0;
    for (int s = gl; s < S; s += 16) acc ^= partial[(size_t)s * N + r];
    if (acc == 0x123456789ull) out[0] = 1.f;
}

__device__ __forceinline__ int packed_offset(int i, int k)
{
    const int m = k >> 1, lane = (k & 1) * 32 + i;
    return (m >> 2) * 256 + lane * 4 + (m & 3);
}
// tiles of 512 floats; a 16-lane group reads the 16 codes of (tile, half) with 16 scalar loads per lane, group g -> (tile g/2, half g&1)
__global__ __launch_bounds__(256) void calib_read_tile4(const float* __restrict__ packed, float* __restrict__ out, int ntiles)
{
    const int gl = threadIdx.x & 15;
    const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int tile = g >> 1, half = g & 1;
    if (tile >= ntiles) return;
    const int i = (gl & 3) + 8 * (gl >> 2) + 4 * half;
    const float* pt = packed + (size_t)tile * 512;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += pt[packed_offset(i, k)];
    if (s == 123.456f) out[0] = s;
}

// vq_f16_kernel's store: lane (half, col) of a wave writes entry (stream = 2*split + half) of row row0 + col
__global__ __launch_bounds__(256) void calib_write_cand8(unsigned long long* __restrict__ partial, int N, int S)
{
    const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
    const int wave_g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int rows_w = N / 32;                                  // row groups of 32
    const int split = wave_g / rows_w, rg = wave_g - split * rows_w;
    if (split * 2 >= S) return;
    partial[((size_t)split * 2 + half) * N + rg * 32 + col] = ((unsigned long long)wave_g << 32) | lane;
}

__global__ __launch_bounds__(256) void calib_write_stream16(float4* __restrict__ out, long n4)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) out[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}

__global__ __launch_bounds__(256) void calib_evict(float4* __restrict__ scratch, long n4, float v)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) scratch[i] = make_float4(v, v, v, v);
}

int main()
{
    const long STREAM_BYTES = 64l << 20;          // 64 MiB
    const int N = 32768, S = 32;                  // 8 MiB of candidates, as at B = 64 x K = 512 with 16 splits x 2 halves
    const int NT = 32768;                         // 32768 tiles x 2 KiB = 64 MiB image (each tile read once)
    const long SCRATCH = 512l << 20;
    float4 *stream, *scratch, *wstream;
    unsigned long long *cand, *wcand;
    float *img, *out;
    CK(hipMalloc(&stream, STREAM_BYTES)); CK(hipMalloc(&wstream, STREAM_BYTES)); CK(hipMalloc(&scratch, SCRATCH));
    CK(hipMalloc(&cand, (size_t)N * S * 8)); CK(hipMalloc(&wcand, (size_t)N * S * 8));
    CK(hipMalloc(&img, (size_t)NT * 2048)); CK(hipMalloc(&out, 256));
    CK(hipMemset(stream, 1, STREAM_BYTES)); CK(hipMemset(cand, 1, (size_t)N * S * 8)); CK(hipMemset(img, 0, (size_t)NT * 2048));
    CK(hipMemset(wstream, 0, STREAM_BYTES)); CK(hipMemset(wcand, 0, (size_t)N * S * 8));
    auto evict = [&](float v) { hipLaunchKernelGGL(calib_evict, dim3(4096), dim3(256), 0, 0, scratch, SCRATCH / 16, v); };
    for (int rep = 0; rep < 3; ++rep) {
        evict(1.f + rep);
        hipLaunchKernelGGL(calib_read_stream16, dim3(2048), dim3(256), 0, 0, stream, out, STREAM_BYTES / 16);
        evict(2.f + rep);
        hipLaunchKernelGGL(calib_read_cand8, dim3(N * 16 / 256), dim3(256), 0, 0, cand, out, N, S);
        evict(3.f + rep);
        hipLaunchKernelGGL(calib_read_tile4, dim3(NT * 2 * 16 / 256), dim3(256), 0, 0, img, out, NT);
        evict(4.f + rep);
        hipLaunchKernelGGL(calib_write_cand8, dim3((N / 32) * (S / 2) * 64 / 256), dim3(256), 0, 0, wcand, N, S);
        evict(5.f + rep);
        hipLaunchKernelGGL(calib_write_stream16, dim3(2048), dim3(256), 0, 0, wstream, STREAM_BYTES / 16);
    }
    CK(hipDeviceSynchronize());
    printf("{\"calib_read_stream16\": %ld, \"calib_read_cand8\": %ld, \"calib_read_tile4\": %ld, \"calib_write_cand8\": %ld, \"calib_write_stream16\": %ld}\n",
           STREAM_BYTES, (long)N * S * 8, (long)NT * 2048, (long)N * S * 8, STREAM_BYTES);
    return 0;
}
