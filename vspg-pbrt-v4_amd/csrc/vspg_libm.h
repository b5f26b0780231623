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

#if defined(__HIPCC__)
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

// ---- log (double) ---------------------------------------------------------------------------
// The reference's free-flight sampling in optical-depth space calls std::log(1.0 - x) in DOUBLE
// (media_sampleTMaj.h:379-404).  glibc 2.35's log (sysdeps/ieee754/dbl-64/e_log.c, Szabolcs Nagy's
// optimized-routines): 128-entry {invc, logc} table + degree-5 polynomial, and a separate degree-11
// polynomial with a hi/lo split for arguments near 1; operation order and fusing of the x86-64
// `__log_fma` ifunc variant.  Scope: positive normal doubles (the path passes 1 - u, u a float in
// [0, 1)); outside it the host build calls libm, the device build returns NaN.
VSPG_HD double log_tab_entry(int j) {  // entry i = {invc = T[2i], logc = T[2i+1]}
    static const double T[256] = {
        0x1.734f0c3e0de9fp+0, -0x1.7cc7f79e69000p-2, 0x1.713786a2ce91fp+0, -0x1.76feec20d0000p-2,
        0x1.6f26008fab5a0p+0, -0x1.713e31351e000p-2, 0x1.6d1a61f138c7dp+0, -0x1.6b85b38287800p-2,
        0x1.6b1490bc5b4d1p+0, -0x1.65d5590807800p-2, 0x1.69147332f0cbap+0, -0x1.602d076180000p-2,
        0x1.6719f18224223p+0, -0x1.5a8ca86909000p-2, 0x1.6524f99a51ed9p+0, -0x1.54f4356035000p-2,
        0x1.63356aa8f24c4p+0, -0x1.4f637c36b4000p-2, 0x1.614b36b9ddc14p+0, -0x1.49da7fda85000p-2,
        0x1.5f66452c65c4cp+0, -0x1.445923989a800p-2, 0x1.5d867b5912c4fp+0, -0x1.3edf439b0b800p-2,
        0x1.5babccb5b90dep+0, -0x1.396ce448f7000p-2, 0x1.59d61f2d91a78p+0, -0x1.3401e17bda000p-2,
        0x1.5805612465687p+0, -0x1.2e9e2ef468000p-2, 0x1.56397cee76bd3p+0, -0x1.2941b3830e000p-2,
        0x1.54725e2a77f93p+0, -0x1.23ec58cda8800p-2, 0x1.52aff42064583p+0, -0x1.1e9e129279000p-2,
        0x1.50f22dbb2bddfp+0, -0x1.1956d2b48f800p-2, 0x1.4f38f4734ded7p+0, -0x1.141679ab9f800p-2,
        0x1.4d843cfde2840p+0, -0x1.0edd094ef9800p-2, 0x1.4bd3ec078a3c8p+0, -0x1.09aa518db1000p-2,
        0x1.4a27fc3e0258ap+0, -0x1.047e65263b800p-2, 0x1.4880524d48434p+0, -0x1.feb224586f000p-3,
        0x1.46dce1b192d0bp+0, -0x1.f474a7517b000p-3, 0x1.453d9d3391854p+0, -0x1.ea4443d103000p-3,
        0x1.43a2744b4845ap+0, -0x1.e020d44e9b000p-3, 0x1.420b54115f8fbp+0, -0x1.d60a22977f000p-3,
        0x1.40782da3ef4b1p+0, -0x1.cc00104959000p-3, 0x1.3ee8f5d57fe8fp+0, -0x1.c202956891000p-3,
        0x1.3d5d9a00b4ce9p+0, -0x1.b81178d811000p-3, 0x1.3bd60c010c12bp+0, -0x1.ae2c9ccd3d000p-3,
        0x1.3a5242b75dab8p+0, -0x1.a45402e129000p-3, 0x1.38d22cd9fd002p+0, -0x1.9a877681df000p-3,
        0x1.3755bc5847a1cp+0, -0x1.90c6d69483000p-3, 0x1.35dce49ad36e2p+0, -0x1.87120a645c000p-3,
        0x1.34679984dd440p+0, -0x1.7d68fb4143000p-3, 0x1.32f5cceffcb24p+0, -0x1.73cb83c627000p-3,
        0x1.3187775a10d49p+0, -0x1.6a39a9b376000p-3, 0x1.301c8373e3990p+0, -0x1.60b3154b7a000p-3,
        0x1.2eb4ebb95f841p+0, -0x1.5737d76243000p-3, 0x1.2d50a0219a9d1p+0, -0x1.4dc7b8fc23000p-3,
        0x1.2bef9a8b7fd2ap+0, -0x1.4462c51d20000p-3, 0x1.2a91c7a0c1babp+0, -0x1.3b08abc830000p-3,
        0x1.293726014b530p+0, -0x1.31b996b490000p-3, 0x1.27dfa5757a1f5p+0, -0x1.2875490a44000p-3,
        0x1.268b39b1d3bbfp+0, -0x1.1f3b9f879a000p-3, 0x1.2539d838ff5bdp+0, -0x1.160c8252ca000p-3,
        0x1.23eb7aac9083bp+0, -0x1.0ce7f57f72000p-3, 0x1.22a012ba940b6p+0, -0x1.03cdc49fea000p-3,
        0x1.2157996cc4132p+0, -0x1.f57bdbc4b8000p-4, 0x1.201201dd2fc9bp+0, -0x1.e370896404000p-4,
        0x1.1ecf4494d480bp+0, -0x1.d17983ef94000p-4, 0x1.1d8f5528f6569p+0, -0x1.bf9674ed8a000p-4,
        0x1.1c52311577e7cp+0, -0x1.adc79202f6000p-4, 0x1.1b17c74cb26e9p+0, -0x1.9c0c3e7288000p-4,
        0x1.19e010c2c1ab6p+0, -0x1.8a646b372c000p-4, 0x1.18ab07bb670bdp+0, -0x1.78d01b3ac0000p-4,
        0x1.1778a25efbcb6p+0, -0x1.674f145380000p-4, 0x1.1648d354c31dap+0, -0x1.55e0e6d878000p-4,
        0x1.151b990275fddp+0, -0x1.4485cdea1e000p-4, 0x1.13f0ea432d24cp+0, -0x1.333d94d6aa000p-4,
        0x1.12c8b7210f9dap+0, -0x1.22079f8c56000p-4, 0x1.11a3028ecb531p+0, -0x1.10e4698622000p-4,
        0x1.107fbda8434afp+0, -0x1.ffa6c6ad20000p-5, 0x1.0f5ee0f4e6bb3p+0, -0x1.dda8d4a774000p-5,
        0x1.0e4065d2a9fcep+0, -0x1.bbcece4850000p-5, 0x1.0d244632ca521p+0, -0x1.9a1894012c000p-5,
        0x1.0c0a77ce2981ap+0, -0x1.788583302c000p-5, 0x1.0af2f83c636d1p+0, -0x1.5715e67d68000p-5,
        0x1.09ddb98a01339p+0, -0x1.35c8a49658000p-5, 0x1.08cabaf52e7dfp+0, -0x1.149e364154000p-5,
        0x1.07b9f2f4e28fbp+0, -0x1.e72c082eb8000p-6, 0x1.06ab58c358f19p+0, -0x1.a55f152528000p-6,
        0x1.059eea5ecf92cp+0, -0x1.63d62cf818000p-6, 0x1.04949cdd12c90p+0, -0x1.228fb8caa0000p-6,
        0x1.038c6c6f0ada9p+0, -0x1.c317b20f90000p-7, 0x1.02865137932a9p+0, -0x1.419355daa0000p-7,
        0x1.0182427ea7348p+0, -0x1.81203c2ec0000p-8, 0x1.008040614b195p+0, -0x1.0040979240000p-9,
        0x1.fe01ff726fa1ap-1, 0x1.feff384900000p-9, 0x1.fa11cc261ea74p-1, 0x1.7dc41353d0000p-7,
        0x1.f6310b081992ep-1, 0x1.3cea3c4c28000p-6, 0x1.f25f63ceeadcdp-1, 0x1.b9fc114890000p-6,
        0x1.ee9c8039113e7p-1, 0x1.1b0d8ce110000p-5, 0x1.eae8078cbb1abp-1, 0x1.58a5bd001c000p-5,
        0x1.e741aa29d0c9bp-1, 0x1.95c8340d88000p-5, 0x1.e3a91830a99b5p-1, 0x1.d276aef578000p-5,
        0x1.e01e009609a56p-1, 0x1.07598e598c000p-4, 0x1.dca01e577bb98p-1, 0x1.253f5e30d2000p-4,
        0x1.d92f20b7c9103p-1, 0x1.42edd8b380000p-4, 0x1.d5cac66fb5ccep-1, 0x1.606598757c000p-4,
        0x1.d272caa5ede9dp-1, 0x1.7da76356a0000p-4, 0x1.cf26e3e6b2ccdp-1, 0x1.9ab434e1c6000p-4,
        0x1.cbe6da2a77902p-1, 0x1.b78c7bb0d6000p-4, 0x1.c8b266d37086dp-1, 0x1.d431332e72000p-4,
        0x1.c5894bd5d5804p-1, 0x1.f0a3171de6000p-4, 0x1.c26b533bb9f8cp-1, 0x1.067152b914000p-3,
        0x1.bf583eeece73fp-1, 0x1.147858292b000p-3, 0x1.bc4fd75db96c1p-1, 0x1.2266ecdca3000p-3,
        0x1.b951e0c864a28p-1, 0x1.303d7a6c55000p-3, 0x1.b65e2c5ef3e2cp-1, 0x1.3dfc33c331000p-3,
        0x1.b374867c9888bp-1, 0x1.4ba366b7a8000p-3, 0x1.b094b211d304ap-1, 0x1.5933928d1f000p-3,
        0x1.adbe885f2ef7ep-1, 0x1.66acd2418f000p-3, 0x1.aaf1d31603da2p-1, 0x1.740f8ec669000p-3,
        0x1.a82e63fd358a7p-1, 0x1.815c0f51af000p-3, 0x1.a5740ef09738bp-1, 0x1.8e92954f68000p-3,
        0x1.a2c2a90ab4b27p-1, 0x1.9bb3602f84000p-3, 0x1.a01a01393f2d1p-1, 0x1.a8bed1c2c0000p-3,
        0x1.9d79f24db3c1bp-1, 0x1.b5b515c01d000p-3, 0x1.9ae2505c7b190p-1, 0x1.c2967ccbcc000p-3,
        0x1.9852ef297ce2fp-1, 0x1.cf635d5486000p-3, 0x1.95cbaeea44b75p-1, 0x1.dc1bd3446c000p-3,
        0x1.934c69de74838p-1, 0x1.e8c01b8cfe000p-3, 0x1.90d4f2f6752e6p-1, 0x1.f5509c0179000p-3,
        0x1.8e6528effd79dp-1, 0x1.00e6c121fb800p-2, 0x1.8bfce9fcc007cp-1, 0x1.071b80e93d000p-2,
        0x1.899c0dabec30ep-1, 0x1.0d46b9e867000p-2, 0x1.87427aa2317fbp-1, 0x1.13687334bd000p-2,
        0x1.84f00acb39a08p-1, 0x1.1980d67234800p-2, 0x1.82a49e8653e55p-1, 0x1.1f8ffe0cc8000p-2,
        0x1.8060195f40260p-1, 0x1.2595fd7636800p-2, 0x1.7e22563e0a329p-1, 0x1.2b9300914a800p-2,
        0x1.7beb377dcb5adp-1, 0x1.3187210436000p-2, 0x1.79baa679725c2p-1, 0x1.377266dec1800p-2,
        0x1.77907f2170657p-1, 0x1.3d54ffbaf3000p-2, 0x1.756cadbd6130cp-1, 0x1.432eee32fe000p-2,
    };
    return T[j];
}
#if defined(__HIPCC__)
__shared__ double s_log_tab[256];  // LDS copy, filled by stage_log_tab_lds() (see s_logf_tab)
__device__ __forceinline__ void stage_log_tab_lds() {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_log_tab[i] = log_tab_entry(i);
}
#endif
VSPG_HD uint64_t asuint64(double f) {
    uint64_t u;
    __builtin_memcpy(&u, &f, 8);
    return u;
}
VSPG_HD double asdouble(uint64_t u) {
    double f;
    __builtin_memcpy(&f, &u, 8);
    return f;
}
VSPG_HD double log_host_exact(double x) {
    const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
    const double A0 = -0x1.0000000000001p-1, A1 = 0x1.555555551305bp-2, A2 = -0x1.fffffffeb4590p-3, A3 = 0x1.999b324f10111p-3, A4 = -0x1.55575e506c89fp-3;
    const double B0 = -0x1.0000000000000p-1, B1 = 0x1.5555555555577p-2, B2 = -0x1.ffffffffffdcbp-3, B3 = 0x1.999999995dd0cp-3, B4 = -0x1.55555556745a7p-3, B5 = 0x1.24924a344de30p-3,
                 B6 = -0x1.fffffa4423d65p-4, B7 = 0x1.c7184282ad6cap-4, B8 = -0x1.999eb43b068ffp-4, B9 = 0x1.78182f7afd085p-4, B10 = -0x1.5521375d145cdp-4;
    const uint64_t ix = asuint64(x);
    if (ix - 0x3fee000000000000ull <= 0x308ffffffffffull) {  // 1 - 2^-4 <= x < 1 + 0x1.09p-4
        if (ix == 0x3ff0000000000000ull) return 0.0;
        const double r = x - 1.0;
        const double r2 = r * r;
        const double r3 = r * r2;
        const double s1 = __builtin_fma(r2, B3, __builtin_fma(r, B2, B1));
        const double s2 = __builtin_fma(r2, B6, __builtin_fma(r, B5, B4));
        double s3 = __builtin_fma(r2, B9, __builtin_fma(r, B8, B7));
        s3 = __builtin_fma(r3, B10, s3);
        const double t2 = __builtin_fma(s3, r3, s2);
        const double t1 = __builtin_fma(t2, r3, s1);
        const double rw = __builtin_fma(r, 0x1p27, r);       // r + w, w = r * 2^27
        const double rhi = __builtin_fma(-0x1p27, r, rw);    // r + w - w
        const double rlo = r - rhi;
        const double hh = rhi * rhi;
        const double hi = __builtin_fma(hh, B0, r);          // r + rhi*rhi*B[0]
        double lo = __builtin_fma(hh, B0, r - hi);           // r - hi + w
        lo = __builtin_fma(B0 * rlo, r + rhi, lo);
        const double y = __builtin_fma(t1, r3, lo);
        return hi + y;
    }
    if ((ix >> 48) - 0x0010u >= 0x7ff0u - 0x0010u) return VSPG_LIBM_OUT_OF_SCOPE(log(x));  // subnormal, <= 0, inf, nan
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const int i = (int)((tmp >> 45) & 127u);
    const int k = (int)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & 0xfff0000000000000ull);
    double invc, logc;
#if defined(__HIP_DEVICE_COMPILE__)
    invc = s_log_tab[2 * i];
    logc = s_log_tab[2 * i + 1];
#else
    invc = log_tab_entry(2 * i);
    logc = log_tab_entry(2 * i + 1);
#endif
    const double z = asdouble(iz);
    const double r = __builtin_fma(z, invc, -1.0);
    const double kd = (double)k;
    const double w = __builtin_fma(kd, Ln2hi, logc);
    const double p1 = __builtin_fma(r, A2, A1);
    const double hi = r + w;
    const double r2 = r * r;
    double lo = (w - hi) + r;
    lo = __builtin_fma(kd, Ln2lo, lo);
    const double r3 = r * r2;
    const double p2 = __builtin_fma(r, A4, A3);
    const double q = __builtin_fma(r2, A0, lo);
    const double p = __builtin_fma(p2, r2, p1);
    return __builtin_fma(r3, p, q) + hi;
}

// ---- powf ---------------------------------------------------------------------------------
// The NDS+ collision-probability bias calls std::pow(float, float) (guidedvolpathvspgintegrator.cpp:937).
// glibc 2.35's powf (sysdeps/ieee754/flt-32/e_powf.c, optimized-routines): log2 through a 16-entry
// {invc, logc} table + degree-5 polynomial in double, exp2 through a 32-entry table + degree-3
// polynomial, one rounding to float at the end; operation order and fusing of the x86-64 `__powf_fma`
// ifunc variant, round-to-nearest.  Complete over all float arguments (special cases included).
VSPG_HD double powf_log2_tab_entry(int j) {  // __powf_log2_data.tab: entry i = {T[2i] invc, T[2i+1] logc}
    static const double T[32] = {
        0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2, 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2,
        0x1.49539f0f010bp+0,  -0x1.7418b0a1fb77bp-2, 0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2,
        0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2, 0x1.25e227b0b8eap+0,  -0x1.97c1d1b3b7afp-3,
        0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3, 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4,
        0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5, 0x1p+0,               0x0p+0,
        0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4,  0x1.ca4b31f026aap-1,  0x1.476a9543891bap-3,
        0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3,  0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2,
        0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2,  0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2,
    };
    return T[j];
}
VSPG_HD uint64_t exp2f_tab_entry(int j) {  // __exp2f_data.tab
    static const uint64_t T[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
        0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
        0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
        0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
        0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
        0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
        0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
    };
    return T[j];
}
// checkint (e_powf.c): 0 = y is not an integer, 1 = odd integer, 2 = even integer
VSPG_HD int powf_checkint(uint32_t iy) {
    const int e = (int)(iy >> 23 & 0xff);
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1u)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}
VSPG_HD bool powf_zeroinfnan(uint32_t ix) { return 2u * ix - 1u >= 2u * 0x7f800000u - 1u; }
VSPG_HD float powf_host_exact(float x, float y) {
    uint32_t sign_bias = 0;
    uint32_t ix = asuint(x);
    const uint32_t iy = asuint(y);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || powf_zeroinfnan(iy)) {
        // x < 0x1p-126 or inf or nan, or y is 0 or inf or nan
        if (powf_zeroinfnan(iy)) {
            if (2u * iy == 0) return 1.0f;                 // (signalling NaNs are not distinguished)
            if (ix == 0x3f800000u) return 1.0f;
            if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y;
            if (2u * ix == 2u * 0x3f800000u) return 1.0f;
            if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;  // |x|<1 && y==inf or |x|>1 && y==-inf
            return y * y;
        }
        if (powf_zeroinfnan(ix)) {
            float x2 = x * x;
            if ((ix & 0x80000000u) && powf_checkint(iy) == 1) {
                x2 = -x2;
                sign_bias = 1;
            }
            if (2u * ix == 0 && (iy & 0x80000000u)) return sign_bias ? -__builtin_inff() : __builtin_inff();  // __math_divzerof
            return (iy & 0x80000000u) ? 1 / x2 : x2;
        }
        // x and y are non-zero finite
        if (ix & 0x80000000u) {
            const int yint = powf_checkint(iy);
            if (yint == 0) return __builtin_nanf("");      // __math_invalidf
            if (yint == 1) sign_bias = 1u << 16;           // SIGN_BIAS = 1 << (EXP2F_TABLE_BITS + 11)
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {  // normalise a subnormal x
            ix = asuint(asfloat(ix) * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    // log2_inline
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 23;
    const double invc = powf_log2_tab_entry(2 * i), logc = powf_log2_tab_entry(2 * i + 1);
    const double z = (double)asfloat(iz);
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2,
                 A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp+0;
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double ya = __builtin_fma(A0, r, A1);
    const double pp = __builtin_fma(A2, r, A3);
    const double r2 = r * r;
    double q = __builtin_fma(A4, r, y0);
    const double r4 = r2 * r2;
    q = __builtin_fma(pp, r2, q);
    const double logx = __builtin_fma(ya, r4, q);
    const double ylogx = (double)y * logx;
    if ((asuint64(ylogx) >> 47 & 0xffffu) >= (0x405f800000000000ull >> 47)) {  // |y log2 x| >= 126
        if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -__builtin_inff() : __builtin_inff();  // __math_oflowf
        if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;                                       // __math_uflowf
        if (ylogx < -149.0) return sign_bias ? -0x1p-149f : 0x1p-149f;  // __math_may_uflowf: (0x1.4p-75f)^2 rounded
    }
    // exp2_inline
    const double SHIFT = 0x1.8p47;  // 0x1.8p52 / 32
    const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
    double kd = ylogx + SHIFT;
    const uint64_t ki = asuint64(kd);
    kd -= SHIFT;
    const double rr = ylogx - kd;
    uint64_t t = exp2f_tab_entry((int)(ki & 31u));
    const uint64_t ski = ki + sign_bias;
    t += ski << 47;
    const double sc = asdouble(t);
    const double zz = __builtin_fma(C0, rr, C1);
    const double rr2 = rr * rr;
    double yy = __builtin_fma(C2, rr, 1.0);
    yy = __builtin_fma(zz, rr2, yy);
    return (float)(yy * sc);
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

// ---- atanhf (std::atanh(float) of SampleVisibleWavelengths, util/sampling.h:169-171: the wavelengths a temperature grid's
// blackbody emission is evaluated at).  glibc 2.35: sysdeps/ieee754/flt-32/e_atanhf.c (0.5 * log1pf(...)) over s_log1pf.c, the
// fdlibm float code; x86-64 has no multiarch variant of either, so the build is the baseline one: every operation a separately
// rounded FLOAT operation, no fused multiply-add.  tests/test_libm_model.py checks the pair against the running libm. ----
VSPG_HD float log1pf_host_exact(float x) {
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    const float Lp1 = 6.6666668653e-01f, Lp2 = 4.0000000596e-01f, Lp3 = 2.8571429849e-01f, Lp4 = 2.2222198546e-01f,
                Lp5 = 1.8183572590e-01f, Lp6 = 1.5313838422e-01f, Lp7 = 1.4798198640e-01f;
    float hfsq, f = 0, c = 0, s, z, R, u;
    int32_t k, hx, hu = 0, ax;
    hx = (int32_t)asuint(x);
    ax = hx & 0x7fffffff;
    k = 1;
    if (hx < 0x3ed413d7) {  // x < 0.41422
        if (ax >= 0x3f800000) {  // x <= -1
            if (x == -1.0f) return -__builtin_inff();
            return __builtin_nanf("");
        }
        if (ax < 0x31000000) {  // |x| < 2^-29
            if (ax < 0x24800000) return x;  // |x| < 2^-54
            return x - x * x * 0.5f;
        }
        if (hx > 0 || hx <= (int32_t)0xbe95f61f) {  // -0.2929 < x < 0.41422
            k = 0;
            f = x;
            hu = 1;
        }
    }
    if (hx >= 0x7f800000) return x + x;
    if (k != 0) {
        if (hx < 0x5a000000) {
            u = 1.0f + x;
            hu = (int32_t)asuint(u);
            k = (hu >> 23) - 127;
            c = (k > 0) ? 1.0f - (u - x) : x - (u - 1.0f);  // correction term
            c /= u;
        } else {
            u = x;
            hu = (int32_t)asuint(u);
            k = (hu >> 23) - 127;
            c = 0;
        }
        hu &= 0x007fffff;
        if (hu < 0x3504f7) {
            u = asfloat((uint32_t)hu | 0x3f800000u);  // normalize u
        } else {
            k += 1;
            u = asfloat((uint32_t)hu | 0x3f000000u);  // normalize u/2
            hu = (0x00800000 - hu) >> 2;
        }
        f = u - 1.0f;
    }
    hfsq = 0.5f * f * f;
    if (hu == 0) {  // |f| < 2^-20
        if (f == 0.0f) {
            if (k == 0) return 0.0f;
            c += k * ln2_lo;
            return k * ln2_hi + c;
        }
        R = hfsq * (1.0f - 0.66666666666666666f * f);
        if (k == 0) return f - R;
        return k * ln2_hi - ((R - (k * ln2_lo + c)) - f);
    }
    s = f / (2.0f + f);
    z = s * s;
    R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + c))) - f);
}
VSPG_HD float atanhf_host_exact(float x) {
    const float xa = __builtin_fabsf(x);
    float t;
    if (xa < 0.5f) {
        if (xa < 0x1.0p-28f) return x;
        t = xa + xa;
        t = 0.5f * log1pf_host_exact(t + t * xa / (1.0f - xa));
    } else if (xa < 1.0f) {
        t = 0.5f * log1pf_host_exact((xa + xa) / (1.0f - xa));
    } else {
        if (xa > 1.0f) return __builtin_nanf("");
        return x > 0 ? __builtin_inff() : -__builtin_inff();
    }
    return __builtin_copysignf(t, x);
}

}  // namespace vspg_libm
