// Issue-rate microbenchmark for gfx950, round 3 (replaces imul.hip's method).
//   hipcc --offload-arch=gfx950 -O3 -o issue issue.hip && ./issue
// What changed against imul.hip:
//   * cycles are read IN the kernel: every wave stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its
//     loop, so a figure is "shader cycles per wave-instruction" with no assumption about the clock, and the clock the
//     chip actually held (d memtime / d memrealtime x 100 MHz) is printed beside it (MI355X_MICROARCH.md, constants (6));
//   * the instruction streams are inline assembly (the compiler cannot reassociate, fold or re-pair them); the IEEE
//     division / square root rows use the compiler's own expansion on data-dependent operands, and their machine
//     instruction count per operation is printed from the loop's measured SQ-independent count (see kDivInsts);
//   * a launch lasts >= 10 ms (N = 1 << 16 iterations of 32 wave-instructions), launched back to back for > 2 s before
//     the measured launch so the clock is the sustained one.
// One 256-thread block = one wave per SIMD of a CU; `wps` blocks per CU give 1, 2, 4, 8 waves per SIMD (the kernel uses
// < 32 VGPRs, so 8 fit).  All waves of a SIMD run the same stream; the per-wave figure is
//   cycles_per_wave_inst_per_SIMD = d memtime x 1 / (insts per wave x waves per SIMD)
// i.e. how many cycles of the SIMD one wave-instruction occupies when `wps` waves share it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define N (1 << 15)
enum { M_FMA = 0, M_ADD, M_MUL, M_MOV, M_MULLO, M_MULHI, M_MAD64, M_FMA64, M_RCP, M_RSQ, M_SQRT, M_EXP, M_LDEXP, M_CNDMASK, M_CMP, M_DIVSCALE,
       M_DIVFMAS, M_DIVFIXUP, M_ADDU32, M_IEEE_DIV, M_IEEE_SQRT, M_MIX, M_COUNT };

struct Stamp {
    unsigned long long cyc, real;
};

#define REP8(x) x x x x x x x x
// 4 independent chains x 8 = 32 instructions per iteration
#define CHAIN4(op, tail)                                               \
    asm volatile(REP8(op " %0, %0" tail "\n\t" op " %1, %1" tail "\n\t" op " %2, %2" tail "\n\t" op " %3, %3" tail "\n\t") \
                 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)                 \
                 : "v"(c0), "v"(c1))

template <int MODE>
__global__ __launch_bounds__(256) void k(Stamp *out, float *sink, uint32_t seed) {
    float f0 = 1.0f + (float)(threadIdx.x & 7) * 0.125f + seed * 1e-9f, f1 = f0 + 0.25f, f2 = f0 + 0.5f, f3 = f0 + 0.75f;
    const float c0 = 0.999999f, c1 = 1e-7f;
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 ^ 0x1234567;
    uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3;
    double d0 = f0, d1 = f1, d2 = f2, d3 = f3;
    const double dc0 = 0.999999, dc1 = 1e-7;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int i = 0; i < N; ++i) {
        if (MODE == M_FMA) CHAIN4("v_fma_f32", ", %4, %5");
        if (MODE == M_ADD) CHAIN4("v_add_f32", ", %5");
        if (MODE == M_MUL) CHAIN4("v_mul_f32", ", %4");
        if (MODE == M_MOV) asm volatile(REP8("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %0\n\t") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
        if (MODE == M_MULLO) asm volatile(REP8("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(0x4c957f2du));
        if (MODE == M_MULHI) asm volatile(REP8("v_mul_hi_u32 %0, %0, %4\n\tv_mul_hi_u32 %1, %1, %4\n\tv_mul_hi_u32 %2, %2, %4\n\tv_mul_hi_u32 %3, %3, %4\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(0xfc957f2du));
        if (MODE == M_ADDU32) asm volatile(REP8("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(0x4c957f2du));
        if (MODE == M_MAD64) asm volatile(REP8("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3\n\t") : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(a0), "v"(a1) : "vcc");
        if (MODE == M_FMA64) asm volatile(REP8("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5\n\t") : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dc0), "v"(dc1));
        if (MODE == M_RCP) CHAIN4("v_rcp_f32", "");
        if (MODE == M_RSQ) CHAIN4("v_rsq_f32", "");
        if (MODE == M_SQRT) CHAIN4("v_sqrt_f32", "");
        if (MODE == M_EXP) CHAIN4("v_exp_f32", "");
        if (MODE == M_LDEXP) asm volatile(REP8("v_ldexp_f32 %0, %0, %4\n\tv_ldexp_f32 %1, %1, %4\n\tv_ldexp_f32 %2, %2, %4\n\tv_ldexp_f32 %3, %3, %4\n\t") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(0));
        if (MODE == M_CNDMASK) asm volatile(REP8("v_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc\n\tv_cndmask_b32 %2, %2, %3, vcc\n\tv_cndmask_b32 %3, %3, %0, vcc\n\t") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : : "vcc");
        if (MODE == M_CMP) asm volatile(REP8("v_cmp_lt_f32 vcc, %0, %1\n\tv_cmp_lt_f32 vcc, %1, %2\n\tv_cmp_lt_f32 vcc, %2, %3\n\tv_cmp_lt_f32 vcc, %3, %0\n\t") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : : "vcc");
        if (MODE == M_DIVSCALE) asm volatile(REP8("v_div_scale_f32 %0, vcc, %0, %4, %0\n\tv_div_scale_f32 %1, vcc, %1, %4, %1\n\tv_div_scale_f32 %2, vcc, %2, %4, %2\n\tv_div_scale_f32 %3, vcc, %3, %4, %3\n\t") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(c0) : "vcc");
        if (MODE == M_DIVFMAS) asm volatile(REP8("v_div_fmas_f32 %0, %0, %4, %5\n\tv_div_fmas_f32 %1, %1, %4, %5\n\tv_div_fmas_f32 %2, %2, %4, %5\n\tv_div_fmas_f32 %3, %3, %4, %5\n\t") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(c0), "v"(c1) : "vcc");
        if (MODE == M_DIVFIXUP) asm volatile(REP8("v_div_fixup_f32 %0, %0, %4, %5\n\tv_div_fixup_f32 %1, %1, %4, %5\n\tv_div_fixup_f32 %2, %2, %4, %5\n\tv_div_fixup_f32 %3, %3, %4, %5\n\t") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(c0), "v"(c1));
        if (MODE == M_IEEE_DIV) {  // 8 divisions per iteration, 4 independent chains, operands data dependent
#pragma unroll
            for (int j = 0; j < 2; ++j) { f0 = (f0 + 1.5f) / (f1 + 2.5f); f1 = (f1 + 1.5f) / (f2 + 2.5f); f2 = (f2 + 1.5f) / (f3 + 2.5f); f3 = (f3 + 1.5f) / (f0 + 2.5f); }
        }
        if (MODE == M_IEEE_SQRT) {
#pragma unroll
            for (int j = 0; j < 2; ++j) { f0 = __builtin_sqrtf(f0 + 1.5f); f1 = __builtin_sqrtf(f1 + 1.5f); f2 = __builtin_sqrtf(f2 + 1.5f); f3 = __builtin_sqrtf(f3 + 1.5f); }
        }
        if (MODE == M_MIX) {  // the path kernels' rough mix: 16 fp32 fma/add/mul, 6 int, 4 cndmask/cmp, 2 mad64, 2 f64, 1 rcp, 1 ldexp
            asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n\tv_mul_f32 %1, %1, %8\n\tv_add_f32 %2, %2, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                "v_add_u32 %4, %4, %5\n\tv_mul_f32 %0, %0, %8\n\tv_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %2, %2, %3, vcc\n\t"
                "v_mad_u64_u32 %6, vcc, %4, %5, %6\n\tv_fma_f32 %1, %1, %8, %9\n\tv_add_f32 %3, %3, %9\n\tv_mul_lo_u32 %5, %5, %4\n\t"
                "v_fma_f64 %7, %7, %10, %11\n\tv_fma_f32 %0, %0, %8, %9\n\tv_mul_f32 %2, %2, %8\n\tv_xor_b32 %4, %4, %5\n\t"
                "v_rcp_f32 %3, %3\n\tv_fma_f32 %1, %1, %8, %9\n\tv_add_f32 %0, %0, %9\n\tv_lshrrev_b32 %5, 3, %5\n\t"
                "v_mad_u64_u32 %6, vcc, %5, %4, %6\n\tv_mul_f32 %2, %2, %8\n\tv_cmp_gt_f32 vcc, %2, %0\n\tv_cndmask_b32 %1, %1, %0, vcc\n\t"
                "v_fma_f64 %7, %7, %10, %11\n\tv_fma_f32 %3, %3, %8, %9\n\tv_ldexp_f32 %0, %0, %12\n\tv_add_u32 %4, %4, %5\n\t"
                "v_fma_f32 %2, %2, %8, %9\n\tv_add_f32 %1, %1, %9\n\tv_and_b32 %5, %5, %4\n\tv_mul_f32 %3, %3, %8\n\t"
                : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(a0), "+v"(a1), "+v"(q0), "+v"(d0)
                : "v"(c0), "v"(c1), "v"(dc0), "v"(dc1), "v"(0)
                : "vcc");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t1 - t0, r1 - r0};
    sink[blockIdx.x * blockDim.x + threadIdx.x] = f0 + f1 + f2 + f3 + (float)(a0 + a1 + a2 + a3) + (float)(q0 + q1 + q2 + q3) + (float)(d0 + d1 + d2 + d3);
}

struct Row {
    const char *name;
    int insts_per_iter;  // wave-instructions per loop iteration as written (asm rows); source operations for the IEEE rows
};
static const Row kRows[M_COUNT] = {
    {"v_fma_f32", 32}, {"v_add_f32", 32}, {"v_mul_f32", 32}, {"v_mov_b32", 32}, {"v_mul_lo_u32", 32}, {"v_mul_hi_u32", 32},
    {"v_mad_u64_u32 (PCG step piece)", 32}, {"v_fma_f64", 32}, {"v_rcp_f32", 32}, {"v_rsq_f32", 32}, {"v_sqrt_f32", 32}, {"v_exp_f32", 32},
    {"v_ldexp_f32", 32}, {"v_cndmask_b32", 32}, {"v_cmp_lt_f32", 32}, {"v_div_scale_f32", 32}, {"v_div_fmas_f32", 32}, {"v_div_fixup_f32", 32},
    {"v_add_u32", 32}, {"IEEE f32 division (per division, +2 adds)", 8}, {"IEEE f32 sqrt (per sqrt, +1 add)", 8}, {"path-kernel-like mix (32 insts)", 32}};

template <int MODE>
void run(Stamp *d, float *sink, int cus, int wps, bool warm) {
    const int blocks = cus * wps;
    if (warm)  // hold the chip busy for a while first: the figure is for the sustained clock
        for (int r = 0; r < 6; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, sink, 1u + r);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, sink, 17u);
    hipDeviceSynchronize();
    std::vector<Stamp> h(blocks * 4);
    hipMemcpy(h.data(), d, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (auto &s : h) {
        cyc.push_back((double)s.cyc);
        clk.push_back((double)s.cyc / (double)s.real * 0.1);  // GHz
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    const double med = cyc[cyc.size() / 2], ghz = clk[clk.size() / 2];
    const double per = med / ((double)N * kRows[MODE].insts_per_iter * wps);
    printf("%-44s waves/SIMD %d  %8.3f cycles of the SIMD per wave-%s  (one wave: %7.2f)  clock %.3f GHz  loop %.1f ms\n", kRows[MODE].name, wps, per,
           kRows[MODE].insts_per_iter == 32 ? "instruction" : "operation", med / ((double)N * kRows[MODE].insts_per_iter), ghz, med / ghz * 1e-6);
}

template <int MODE>
void sweep(Stamp *d, float *sink, int cus) {
    const int wpss[4] = {1, 2, 4, 8};
    for (int w = 0; w < 4; ++w) run<MODE>(d, sink, cus, wpss[w], w == 0);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, nominal clock %.2f GHz; cycles are s_memtime ticks (shader clock), clock = d memtime / d memrealtime x 100 MHz\n",
           prop.gcnArchName, cus, prop.clockRate * 1e-6);
    Stamp *d;
    float *sink;
    hipMalloc(&d, (size_t)cus * 8 * 4 * sizeof(Stamp));
    hipMalloc(&sink, (size_t)cus * 8 * 256 * sizeof(float));
    sweep<M_FMA>(d, sink, cus); sweep<M_ADD>(d, sink, cus); sweep<M_MUL>(d, sink, cus); sweep<M_MOV>(d, sink, cus);
    sweep<M_ADDU32>(d, sink, cus); sweep<M_MULLO>(d, sink, cus); sweep<M_MULHI>(d, sink, cus); sweep<M_MAD64>(d, sink, cus);
    sweep<M_FMA64>(d, sink, cus); sweep<M_RCP>(d, sink, cus); sweep<M_RSQ>(d, sink, cus); sweep<M_SQRT>(d, sink, cus); sweep<M_EXP>(d, sink, cus);
    sweep<M_LDEXP>(d, sink, cus); sweep<M_CNDMASK>(d, sink, cus); sweep<M_CMP>(d, sink, cus);
    sweep<M_DIVSCALE>(d, sink, cus); sweep<M_DIVFMAS>(d, sink, cus); sweep<M_DIVFIXUP>(d, sink, cus);
    sweep<M_IEEE_DIV>(d, sink, cus); sweep<M_IEEE_SQRT>(d, sink, cus); sweep<M_MIX>(d, sink, cus);
    return 0;
}
