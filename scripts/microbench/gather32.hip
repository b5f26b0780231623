// What does a 32-byte gather cost in HBM traffic on gfx950, and what does rocprofv3's FETCH_SIZE say about it?
//   hipcc --offload-arch=gfx950 -O3 -o gather32 gather32.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./gather32        (and once with TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum)
// Three kernels over a 4 GiB array (far beyond the 256 MiB Infinity Cache), N = 2^26 accesses each:
//   k_stream   every lane reads 16 B, consecutive lanes consecutive addresses       (the guide's calibration case: FETCH_SIZE = bytes / 2)
//   k_gather32 every lane reads one random, 32-byte-aligned 32-byte record (2 x 16 B) -- the density octet fetch of the walks
//   k_gather16 every lane reads one random 16-byte record
//   k_gather128    every lane reads one random, 128-byte-aligned 128-byte line in full (8 x 16 B)
//   k_gather2x32   every lane reads 32 B at offset 0 and 32 B at offset 64 of one random 128-byte line
// The last two settle the GRANULARITY question: if the memory side moves 128-byte lines, they run at k_gather32's record rate and
// show the same FETCH_SIZE per record; if it moved 64-byte sectors, they would need two requests per record (half the rate, twice
// the FETCH_SIZE).
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
__global__ void k_gather128(const float4* __restrict__ a, float* out, uint32_t mask128, uint32_t per_thread) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0;
    for (uint32_t k = 0; k < per_thread; ++k) {
        const size_t rec = hash32(t * per_thread + k) & mask128;   // line index (128-byte lines)
        const float4* p = a + rec * 8;
        const float4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3], v4 = p[4], v5 = p[5], v6 = p[6], v7 = p[7];
        s += v0.x + v1.y + v2.z + v3.w + v4.x + v5.y + v6.z + v7.w;
    }
    if (s == 123.456f) out[0] = s;
}
__global__ void k_gather2x32(const float4* __restrict__ a, float* out, uint32_t mask128, uint32_t per_thread) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0;
    for (uint32_t k = 0; k < per_thread; ++k) {
        const size_t rec = hash32(t * per_thread + k) & mask128;
        const float4* p = a + rec * 8;
        const float4 v0 = p[0], v1 = p[1], v4 = p[4], v5 = p[5];   // bytes 0-31 and 64-95 of the line
        s += v0.x + v1.y + v4.z + v5.w;
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
        hipEventRecord(e0); hipLaunchKernelGGL(k_gather128, dim3(blocks), dim3(threads), 0, 0, a, out, (uint32_t)(bytes / 128 - 1), per_thread); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("k_gather128 %.0f MiB asked (%.0f lines of 128 B), %.3f ms, %.1f G lines/s, %.0f GB/s asked\n", N * 128 / 1048576.0, N, ms, N / ms * 1e-6, N * 128 / ms * 1e-6);
        hipEventRecord(e0); hipLaunchKernelGGL(k_gather2x32, dim3(blocks), dim3(threads), 0, 0, a, out, (uint32_t)(bytes / 128 - 1), per_thread); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("k_gather2x32 %.0f MiB asked (%.0f lines, 32 B of each half), %.3f ms, %.1f G lines/s\n", N * 64 / 1048576.0, N, ms, N / ms * 1e-6);
    }
    return 0;
}
