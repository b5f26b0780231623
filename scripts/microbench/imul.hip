// Issue-rate microbenchmark: 32-bit / 64-bit integer multiplies vs f32 fma vs f64 fma on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 4096
template <int MODE>
__global__ void k(uint64_t* out, uint32_t seed) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 ^ 0x1234567;
    uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3;
    float f0 = a0, f1 = a1, f2 = a2, f3 = a3;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    for (int i = 0; i < N; ++i) {
        if (MODE == 0) { a0 = a0 * 0x4c957f2du + 1; a1 = a1 * 0x4c957f2du + 3; a2 = a2 * 0x4c957f2du + 5; a3 = a3 * 0x4c957f2du + 7; }
        if (MODE == 1) { q0 = q0 * 0x5851f42d4c957f2dull + 1; q1 = q1 * 0x5851f42d4c957f2dull + 3; q2 = q2 * 0x5851f42d4c957f2dull + 5; q3 = q3 * 0x5851f42d4c957f2dull + 7; }
        if (MODE == 2) { f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f1 = __builtin_fmaf(f1, 1.0001f, 0.5f); f2 = __builtin_fmaf(f2, 1.0001f, 0.5f); f3 = __builtin_fmaf(f3, 1.0001f, 0.5f); }
        if (MODE == 3) { d0 = __builtin_fma(d0, 1.0001, 0.5); d1 = __builtin_fma(d1, 1.0001, 0.5); d2 = __builtin_fma(d2, 1.0001, 0.5); d3 = __builtin_fma(d3, 1.0001, 0.5); }
        if (MODE == 4) { a0 = __umulhi(a0, 0x4c957f2du) + 1; a1 = __umulhi(a1, 0x4c957f2du) + 3; a2 = __umulhi(a2, 0x4c957f2du) + 5; a3 = __umulhi(a3, 0x4c957f2du) + 7; }
        if (MODE == 5) { f0 = f0 / (f1 + 1.5f); f1 = f1 / (f2 + 1.5f); f2 = f2 / (f3 + 1.5f); f3 = f3 / (f0 + 1.5f); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + q0 + q1 + q2 + q3 + (uint64_t)(f0 + f1 + f2 + f3) + (uint64_t)(d0 + d1 + d2 + d3);
}
template <int MODE> void run(const char* name, uint64_t* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8, threads = 256;  // 8 waves per SIMD: issue-bound
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 2u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * (threads / 64) * N * 4;  // wave-level ops
    double cyc_per_op = ms * 1e-3 * 2.4e9 * 1024 / ops;    // SIMD-cycles per wave op at 2.4 GHz nominal
    printf("%-22s %.3f ms  ~%.1f SIMD-cycles per wave-op (at 2.4 GHz)\n", name, ms, cyc_per_op);
}
int main() {
    uint64_t* d; hipMalloc(&d, 256 * 8 * 256 * 8);
    run<2>("f32 fma", d); run<0>("u32 mul+add", d); run<4>("u32 mulhi+add", d); run<1>("u64 mul+add (PCG step)", d); run<3>("f64 fma", d); run<5>("f32 IEEE div (+add)", d);
    return 0;
}
