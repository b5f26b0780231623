// Does a cache policy on the load change what a random 32-byte record costs at the memory side of gfx950's L2?  (round 5)
// gather_lines.hip showed: a default global load of 16 / 32 / 64 / 128 B of a random line issues ONE 128-byte request
// (TCC_EA0_RDREQ_128B) at the stream's line rate, ~51 G lines/s.  Here the walk kernels' octet fetch (every lane its own random
// 32-byte record, two 16-byte loads) with the load's cache-policy bits set: nt (non-temporal), sc1, sc0 sc1 (system scope).  If a
// policy made the L2 ask for 32- or 64-byte sectors, its record rate would exceed the line rate and the RDREQ_32B / _64B counters
// would show it.
//   hipcc --offload-arch=gfx950 -O3 -o gather_policy gather_policy.hip
//   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d out -- ./gather_policy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
typedef float v4f __attribute__((ext_vector_type(4)));
template <int POLICY> __device__ __forceinline__ v4f ld(const v4f* p) {
    v4f v;
    if constexpr (POLICY == 0) asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POLICY == 1) asm volatile("global_load_dwordx4 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POLICY == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// (each step waits for its load -- the same for every policy; 8 steps in flight per lane come from 8 independent records per iteration)
template <int POLICY>
__global__ void k_rec32(const v4f* __restrict__ a, float* out, uint32_t mask32, uint32_t steps) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0;
    for (uint32_t k = 0; k < steps; ++k) {
        const size_t rec = hash32(t * steps + k) & mask32;
        const v4f lo = ld<POLICY>(a + rec * 2), hi = ld<POLICY>(a + rec * 2 + 1);
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
    v4f* a; float* out;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(a, 0, bytes);
    hipDeviceSynchronize();
    const double NB = 67108864.0;   // 2^26 records per kernel
    const uint32_t steps = 32, threads = (uint32_t)(NB / steps), mask = (uint32_t)(bytes / 32 - 1);
    for (int rep = 0; rep < 2; ++rep) {
#define RUN(P, NAME)                                                                                                          \
        {                                                                                                                    \
            const float ms = timed([&] { hipLaunchKernelGGL(k_rec32<P>, dim3(threads / 256), dim3(256), 0, 0, a, out, mask, steps); }); \
            printf("k_rec32<%d> %-12s %.3f ms, %.1f G records/s\n", P, NAME, ms, NB / ms * 1e-6);                             \
        }
        RUN(0, "default") RUN(1, "nt") RUN(2, "sc1") RUN(3, "sc0 sc1") RUN(4, "sc0 sc1 nt")
    }
    return 0;
}
