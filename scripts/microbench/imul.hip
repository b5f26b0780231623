// Issue-rate microbenchmark (gfx950): how many SIMD cycles one wave-instruction of each class costs at 1, 2, 4 and 8
// resident waves per SIMD.  Output feeds DESIGN.md 4.1 (the "is the path kernel issue-bound" question); run by
// scripts/gpu_microbench.sh, result kept under profiles/.
//   hipcc --offload-arch=gfx950 -O3 -o imul imul.hip && ./imul
// Every class runs four independent dependency chains per lane, so a single wave can issue back to back.
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
        if (MODE == 6) { f0 = __builtin_sqrtf(f0 + 1.5f); f1 = __builtin_sqrtf(f1 + 1.5f); f2 = __builtin_sqrtf(f2 + 1.5f); f3 = __builtin_sqrtf(f3 + 1.5f); }
        if (MODE == 7) { f0 = __builtin_amdgcn_rcpf(f0 + 1.5f); f1 = __builtin_amdgcn_rcpf(f1 + 1.5f); f2 = __builtin_amdgcn_rcpf(f2 + 1.5f); f3 = __builtin_amdgcn_rcpf(f3 + 1.5f); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + q0 + q1 + q2 + q3 + (uint64_t)(f0 + f1 + f2 + f3) + (uint64_t)(d0 + d1 + d2 + d3);
}
// ops_per_iter: wave-level source operations per loop iteration (4 chains); insts_per_op: machine instructions per
// operation as hipcc -O3 emits them for gfx950 (llvm-objdump of this file: u32 mul+add = v_mad_u32_u24-free
// v_mul_lo_u32 + v_add = 2, u64 mul+add = v_mad_u64_u32 x2 + v_mul_lo_u32 ... reported as measured cycles per OPERATION)
template <int MODE> void run(const char* name, uint64_t* d, int cus, int wps, double clk_ghz) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = cus * wps, threads = 256;  // one 256-thread block = one wave per SIMD of a CU
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 2u + r);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 4;
    const double ops_per_simd = (double)wps * N * 4;              // wave-level operations each SIMD executed
    const double cyc_per_op = ms * 1e-3 * clk_ghz * 1e9 / ops_per_simd;
    printf("%-26s waves/SIMD %d  %.3f ms  %.2f SIMD-cycles per wave-operation (at %.2f GHz)\n", name, wps, ms, cyc_per_op, clk_ghz);
}
int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e-6;  // kHz -> GHz (nominal; the chip may hold less under load)
    printf("device %s, %d CUs, nominal clock %.2f GHz\n", prop.name, cus, clk);
    uint64_t* d; hipMalloc(&d, (size_t)cus * 8 * 256 * 8);
    const int wpss[4] = {1, 2, 4, 8};
    for (int w = 0; w < 4; ++w) {
        const int wps = wpss[w];
        run<2>("f32 fma", d, cus, wps, clk); run<0>("u32 mul+add", d, cus, wps, clk); run<4>("u32 mulhi+add", d, cus, wps, clk);
        run<1>("u64 mul+add (PCG step)", d, cus, wps, clk); run<3>("f64 fma", d, cus, wps, clk);
        run<5>("f32 IEEE div (+add)", d, cus, wps, clk); run<6>("f32 IEEE sqrt (+add)", d, cus, wps, clk); run<7>("v_rcp_f32 (+add)", d, cus, wps, clk);
    }
    return 0;
}
