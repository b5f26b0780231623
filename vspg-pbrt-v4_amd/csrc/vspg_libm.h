// vspg_libm.h -- float logf / sinf / cosf that reproduce the HOST libm bit for bit on device.
//
// Why: the reference calls std::log(float), std::sin(float), std::cos(float) (glibc).  A single
// last-ulp difference in a direction or a free-flight distance re-seeds the shadow-ray RNG
// (RNG(Hash(lightRay.o), Hash(lightRay.d)), guidedvolpathvspgintegrator.cpp:1193) and turns
// into an O(1) change of that path's radiance.  Seed parity therefore needs the same float
// functions as the CPU reference run, not merely accurate ones.
//
// What: the published algorithms glibc >= 2.28 uses for these three functions (Szabolcs Nagy's
// optimized-routines: 16-entry-table logf, sincosf.h polynomial sinf/cosf), in the operation
// order of the x86-64 `*_fma` ifunc variants (every a*b+c fused), with the coefficient tables
// of glibc 2.35.  All arithmetic is IEEE double with explicit fma, identical on gfx950 and x86.
// tests/test_libm_model.py builds this header for the host and checks it against the running
// libm on >10^7 arguments; tests/test_gpu_parity.py checks the device build the same way.
//
// Scope: the argument ranges the path produces (logf: positive normal floats; sinf/cosf:
// |x| < 120).  Outside them the host build falls back to a double evaluation rounded once; the
// DEVICE build returns NaN (the path cannot produce such arguments, and the fallbacks would pull
// ocml's Payne-Hanek double sin/cos and double log into every call site: ~10x the code size
// and register pressure of the functions themselves).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define VSPG_HD __host__ __device__ __forceinline__
#if defined(__HIP_DEVICE_COMPILE__)
#define VSPG_LIBM_OUT_OF_SCOPE(expr) __builtin_nanf("")
#endif
#else
#include <math.h>
#include <string.h>
#define VSPG_HD static inline
#endif

#ifndef VSPG_LIBM_OUT_OF_SCOPE
#define VSPG_LIBM_OUT_OF_SCOPE(expr) (expr)
#endif

namespace vspg_libm {

VSPG_HD uint32_t asuint(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return u;
}
VSPG_HD float asfloat(uint32_t u) {
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

// ---- logf ---------------------------------------------------------------------------------
// __logf_data (glibc 2.35 sysdeps/ieee754/flt-32/e_logf_data.c): {invc, logc} x 16
VSPG_HD double logf_tab_entry(int j) {  // flat view: entry i = {T[2i], T[2i+1]}
    static const double T[32] = {
        0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2, 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2,
        0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2,  0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3,
        0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3, 0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3,
        0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4, 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4,
        0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5, 0x1p+0, 0x0p+0,
        0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5,  0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4,
        0x1.b2036576afce6p-1, 0x1.526e57720db08p-3,  0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3,
        0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2,  0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2};
    return T[j];
}
#if defined(__HIPCC__)
// Device: the table is looked up per lane, and from its home in constant memory that is a vector load
// with a full cache round trip (hundreds of cycles, waited for at once: every free-flight step calls
// logf).  Kernels copy it into LDS first (stage_logf_tab_lds() + __syncthreads()); the lookup is then
// one ds_read_b128.
__shared__ double s_logf_tab[32];
__device__ __forceinline__ void stage_logf_tab_lds() {
    if (threadIdx.x < 32) s_logf_tab[threadIdx.x] = logf_tab_entry((int)threadIdx.x);
}
#endif
VSPG_HD void logf_tab(int i, double *invc, double *logc) {
#if defined(__HIP_DEVICE_COMPILE__)
    *invc = s_logf_tab[2 * i];
    *logc = s_logf_tab[2 * i + 1];
#else
    *invc = logf_tab_entry(2 * i);
    *logc = logf_tab_entry(2 * i + 1);
#endif
}

VSPG_HD float logf_host_exact(float x) {
    const double Ln2 = 0x1.62e42fefa39efp-1;
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    uint32_t ix = asuint(x);
    if (ix == 0x3f800000u) return 0.f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) return VSPG_LIBM_OUT_OF_SCOPE((float)log((double)x));  // subnormal, <=0, inf, nan
    uint32_t tmp = ix - 0x3f330000u;
    int i = (int)((tmp >> 19) & 15u);
    int k = (int32_t)tmp >> 23;
    uint32_t iz = ix - (tmp & 0xff800000u);
    double invc, logc;
    logf_tab(i, &invc, &logc);
    double z = (double)asfloat(iz);
    double r = __builtin_fma(z, invc, -1.0);
    double y0 = __builtin_fma((double)k, Ln2, logc);
    double r2 = r * r;
    double y = __builtin_fma(A1, r, A2);
    y = __builtin_fma(A0, r2, y);
    double t = r + y0;
    y = __builtin_fma(r2, y, t);
    return (float)y;
}

// ---- sinf / cosf --------------------------------------------------------------------------
// __sincosf_table (glibc 2.35 sysdeps/ieee754/flt-32/s_sincosf_data.c); table 1 = table 0 with the
// cosine coefficients negated (used when bit 1 of the quadrant is set)
VSPG_HD float sincosf_poly(double x, double x2, bool neg, int n) {
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    const double C0 = 0x1p+0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
                 C4 = 0x1.99343027bf8c3p-16;
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = __builtin_fma(x2, S3, S2);
        double x7 = x3 * x2;
        double s = __builtin_fma(x3, S1, x);
        return (float)__builtin_fma(s1, x7, s);
    } else {
        const double sg = neg ? -1.0 : 1.0;  // exact sign flips of the table-1 coefficients
        double x4 = x2 * x2;
        double c1 = __builtin_fma(x2, sg * C1, sg * C0);
        double c2 = __builtin_fma(x2, sg * C4, sg * C3);
        double x6 = x4 * x2;
        double c = __builtin_fma(x4, sg * C2, c1);
        return (float)__builtin_fma(c2, x6, c);
    }
}
VSPG_HD double sincosf_reduce_fast(double x, int *np) {
    const double hpi_inv = 0x1.45f306dc9c883p+23;  // 2/pi * 2^24
    const double hpi = 0x1.921fb54442d18p+0;
    double r = x * hpi_inv;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return __builtin_fma(-(double)n, hpi, x);
}
VSPG_HD double sincosf_sign(int q) { return (q == 1 || q == 2) ? -1.0 : 1.0; }  // sign[4] = {1,-1,-1,1}

// glibc's sinf/cosf take a short cut for |y| < pi/4 (no reduction) and return y / 1.0f for
// |y| < 2^-12.  Both are special cases of the general path: for |y| < pi/4 the reduction yields n = 0
// and x unchanged, and for |y| < 2^-12 the polynomial's correction is below half an ulp of the float
// result, so one branch-free path reproduces all three (verified exhaustively over every float in
// [0, 8] and its negative by tests/test_libm_model.py::test_sincos_exhaustive).
VSPG_HD float sinf_host_exact(float y) {
    uint32_t top = (asuint(y) >> 20) & 0x7ffu;
    if (top > 0x42eu) return VSPG_LIBM_OUT_OF_SCOPE((float)sin((double)y));  // |y| >= 120, inf, nan
    int n;
    double x = sincosf_reduce_fast((double)y, &n);
    double s = sincosf_sign(n & 3);
    float r = sincosf_poly(x * s, x * x, (n & 2) != 0, n);
    return y == 0.f ? y : r;  // sin(-0) = -0: the only input whose sign the polynomial loses
}
VSPG_HD float cosf_host_exact(float y) {
    uint32_t top = (asuint(y) >> 20) & 0x7ffu;
    if (top > 0x42eu) return VSPG_LIBM_OUT_OF_SCOPE((float)cos((double)y));
    int n;
    double x = sincosf_reduce_fast((double)y, &n);
    double s = sincosf_sign(n & 3);
    return sincosf_poly(x * s, x * x, (n & 2) != 0, n ^ 1);
}

}  // namespace vspg_libm
