// What does a 32-byte gather cost in HBM traffic on gfx950, and what does rocprofv3's FETCH_SIZE say about it?
//   hipcc --offload-arch=gfx950 -O3 -o gather32 gather32.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./gather32        (and once with TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum)
// Three kernels over a 4 GiB array (far beyond the 256 MiB Infinity Cache), N = 2^26 accesses each:
//   k_stream   every lane reads 16 B, consecutive lanes consecutive addresses       (the guide's calibration case: FETCH_SIZE = bytes / 2)
//   k_gather32 every lane reads one random, 32-byte-aligned 32-byte record (2 x 16 B) -- the density octet fetch of the walks
//   k_gather16 every lane reads one random 16-byte record
// Printed: bytes the lanes asked for, kernel time, the rate that implies.  The profile's FETCH_SIZE per kernel then says how many bytes
// the memory side moved per record: 32, 64 or 128.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void k_stream(const float4* __restrict__ a, float* out, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0;
    for (; i < n16; i += (size_t)gridDim.x * blockDim.x) { float4 v = a[i]; s += v.x + v.w; }
    if (s == 123.456f) out[0] = s;
}
__global__ void k_gather32(const float4* __restrict__ a, float* out, uint32_t mask32, uint32_t per_thread) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0;
    for (uint32_t k = 0; k < per_thread; ++k) {
        const size_t rec = hash32(t * per_thread + k) & mask32;   // record index (32-byte records)
        const float4 lo = a[rec * 2], hi = a[rec * 2 + 1];
        s += lo.x + hi.w;
    }
    if (s == 123.456f) out[0] = s;
}
__global__ void k_gather16(const float4* __restrict__ a, float* out, uint32_t mask16, uint32_t per_thread) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0;
    for (uint32_t k = 0; k < per_thread; ++k) {
        const size_t rec = hash32(t * per_thread + k) & mask16;
        const float4 v = a[rec];
        s += v.x + v.w;
    }
    if (s == 123.456f) out[0] = s;
}
int main() {
    const size_t bytes = (size_t)4 << 30;
    float4* a; float* out;
    hipMalloc(&a, bytes); hipMalloc(&out, 4);
    hipMemset(a, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t threads = 256, blocks = 256 * 8 * 4, per_thread = 32;   // 2^21 threads x 32 = 2^26 accesses
    const double N = (double)threads * blocks * per_thread;
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(threads), 0, 0, a, out, bytes / 16); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("k_stream   %.0f MiB asked, %.3f ms, %.0f GB/s\n", bytes / 1048576.0, ms, bytes / ms * 1e-6);
        hipEventRecord(e0); hipLaunchKernelGGL(k_gather32, dim3(blocks), dim3(threads), 0, 0, a, out, (uint32_t)(bytes / 32 - 1), per_thread); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("k_gather32 %.0f MiB asked (%.0f records of 32 B), %.3f ms, %.1f G records/s, %.0f GB/s asked\n", N * 32 / 1048576.0, N, ms, N / ms * 1e-6, N * 32 / ms * 1e-6);
        hipEventRecord(e0); hipLaunchKernelGGL(k_gather16, dim3(blocks), dim3(threads), 0, 0, a, out, (uint32_t)(bytes / 16 - 1), per_thread); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("k_gather16 %.0f MiB asked (%.0f records of 16 B), %.3f ms, %.1f G records/s, %.0f GB/s asked\n", N * 16 / 1048576.0, N, ms, N / ms * 1e-6, N * 16 / ms * 1e-6);
    }
    return 0;
}
