// What does a random 32-byte record cost at the memory side of gfx950's L2?  (round 4; settles the reading of
// profiles/r03_microbench_gather32.txt the round-3 review questioned: there a lane read a whole 128-B line with EIGHT loads,
// so its line rate was bound by request issue, not by the memory side, and could not be compared with the 32-B gather's.)
//   hipcc --offload-arch=gfx950 -O3 -o gather_lines gather_lines.hip
//   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d out -- ./gather_lines
// Every kernel issues ONE 16-byte load per lane per step; what differs is how many consecutive lanes share a random block:
//   k_coal<128>  8 lanes read the eight 16-B pieces of one random 128-B line   (a wave-instruction touches 8 lines, whole)
//   k_coal<64>   4 lanes read one random 64-B half line                        (16 half lines)
//   k_coal<32>   2 lanes read one random 32-B record                           (32 records: the density octet, coalesced)
//   k_coal<16>   every lane its own random 16-B piece                          (64 pieces)
//   k_rec32      every lane reads its own random 32-B record with TWO loads    (the walk kernels' octet fetch as it is)
//   k_stream     consecutive lanes, consecutive addresses                      (calibration)
// 4 GiB array (far past the 256 MiB Infinity Cache), 2^26 blocks per kernel.  If the memory side moves 128-B lines whatever is
// asked, all k_coal<B> run at the same BLOCK rate (the line rate of the stream) and the counters show one 128-B request per block;
// if it moves 32- or 64-B sectors, smaller blocks run at a higher block rate and show smaller requests.
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
template <int B>   // bytes per block: B / 16 consecutive lanes share a block
__global__ void k_coal(const float4* __restrict__ a, float* out, uint32_t mask, uint32_t steps) {
    constexpr uint32_t L = B / 16;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t group = t / L, piece = t % L;
    float s = 0;
    for (uint32_t k = 0; k < steps; ++k) {
        const size_t blk = hash32(group * steps + k) & mask;
        const float4 v = a[blk * L + piece];
        s += v.x + v.w;
    }
    if (s == 123.456f) out[0] = s;
}
__global__ void k_rec32(const float4* __restrict__ a, float* out, uint32_t mask32, uint32_t steps) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0;
    for (uint32_t k = 0; k < steps; ++k) {
        const size_t rec = hash32(t * steps + k) & mask32;
        const float4 lo = a[rec * 2], hi = a[rec * 2 + 1];
        s += lo.x + hi.w;
    }
    if (s == 123.456f) out[0] = s;
}
template <class F>
static float timed(F&& launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}
int main() {
    const size_t bytes = (size_t)4 << 30;
    float4* a; float* out;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(a, 0, bytes);
    hipDeviceSynchronize();
    const double NB = 67108864.0;   // 2^26 blocks per kernel
    for (int rep = 0; rep < 2; ++rep) {
        float ms = timed([&] { hipLaunchKernelGGL(k_stream, dim3(8192), dim3(256), 0, 0, a, out, bytes / 16); });
        printf("k_stream     %.3f ms, %.0f GB/s = %.1f G lines/s\n", ms, bytes / ms * 1e-6, bytes / 128.0 / ms * 1e-6);
#define RUN(B)                                                                                                               \
        {                                                                                                                    \
            const uint32_t lanes = B / 16, steps = 32;                                                                       \
            const uint32_t threads = (uint32_t)(NB / steps) * lanes;                                                         \
            ms = timed([&] { hipLaunchKernelGGL(k_coal<B>, dim3(threads / 256), dim3(256), 0, 0, a, out, (uint32_t)(bytes / B - 1), steps); }); \
            printf("k_coal<%3d>  %.3f ms, %.1f G blocks/s, %.0f GB/s asked\n", B, ms, NB / ms * 1e-6, NB * B / ms * 1e-6);   \
        }
        RUN(128) RUN(64) RUN(32) RUN(16)
        ms = timed([&] { hipLaunchKernelGGL(k_rec32, dim3((uint32_t)(NB / 32) / 256), dim3(256), 0, 0, a, out, (uint32_t)(bytes / 32 - 1), 32u); });
        printf("k_rec32      %.3f ms, %.1f G records/s, %.0f GB/s asked\n", ms, NB / ms * 1e-6, NB * 32 / ms * 1e-6);
    }
    return 0;
}
