// vspg_device.h -- device-side building blocks of the MI355X-native VSPG hot path (gfx950).
//
// Written from the reference's text (file:line cited per function, paths relative to the
// reference root), independently of oracle/.  Float evaluation order follows the reference
// expression by expression so that identical RNG seeds give identical paths; the TU is built
// with -ffp-contract=off, explicit FMA() in the reference == __builtin_fmaf here.
//
// libm policy: the reference calls glibc's logf/sinf/cosf and a double-precision std::log.  The
// float functions are reproduced bit for bit on device (vspg_libm.h); double log is ocml's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vspg.h"
#include "vspg_arith.h"
#include "vspg_libm.h"

#define VDEV __device__ __forceinline__
#ifdef VSPG_WF_DEBUG  // diagnostic build only: out-of-range accesses are clamped and flagged instead of faulting
__device__ unsigned int g_dbg_err[8];
#define VSPG_DBG_CHECK(cond, bit) do { if (!(cond)) atomicOr(&g_dbg_err[0], 1u << (bit)); } while (0)
#else
#define VSPG_DBG_CHECK(cond, bit)
#endif
// leaf functions with small interfaces.  Measured on MI355X (round 1): as real calls
// (-DVSPG_NOINLINE_LEAVES) k_render_wave drops from 248 to 220 VGPRs but runs 8 % slower at the same
// 2 waves/SIMD, and the 168-VGPR build (3 waves/SIMD) still spills 101 registers and is 17 % slower,
// so they are inlined.
#ifdef VSPG_NOINLINE_LEAVES
#define VLEAF __device__ __attribute__((noinline))
#else
#define VLEAF __device__ __forceinline__
#endif

VSPG_NS_BEGIN

// ---------------------------------------------------------------------------------------
// diagnostic build only (-DVSPG_PROFILE, csrc/Makefile target `prof`): per-section wave time,
// active-lane count and execution count.  The shipped library contains none of this.
// ---------------------------------------------------------------------------------------
#ifdef VSPG_PROFILE
enum { PS_INTERSECT = 0, PS_HASHRNG, PS_DIST_GUIDED, PS_DIST_PLAIN, PS_SURF_PRE, PS_NEE, PS_NEE_TR, PS_VOL_SAMPLE,
       PS_SURF_SAMPLE, PS_FINISH, PS_START, PS_REFILL, PS_SEGMENT, PS_WG_R, PS_WG_A, PS_WG_B, PS_WG_BAR_R, PS_WG_BAR_A,
       PS_WG_BAR_B, PS_WG_VERTEX, PS_WG_TOTAL, PS_COUNT };
__device__ unsigned long long g_prof[PS_COUNT][3];
struct ProfScope {
    int sec;
    unsigned long long t0;
    __device__ __forceinline__ ProfScope(int s) : sec(s) { t0 = __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ ~ProfScope() {
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        unsigned long long m = __ballot(1);
        if ((int)(threadIdx.x & 63) == (int)(__ffsll((long long)m) - 1)) {
            atomicAdd(&g_prof[sec][0], t1 - t0);
            atomicAdd(&g_prof[sec][1], (unsigned long long)__popcll(m));
            atomicAdd(&g_prof[sec][2], 1ull);
        }
    }
};
#define VSPG_PROF(sec) ProfScope prof_scope_##sec(sec)
// the same for a section a wavefront enters very often (the idle polls of k_render_wave_wg3): time summed in registers, one flush
struct ProfAccum {
    int sec;
    unsigned long long t, n, t0;
    __device__ __forceinline__ ProfAccum(int s) : sec(s), t(0), n(0), t0(0) {}
    __device__ __forceinline__ void begin() { t0 = __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void end() { t += __builtin_amdgcn_s_memtime() - t0; n++; }
    __device__ __forceinline__ ~ProfAccum() {
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&g_prof[sec][0], t);
            atomicAdd(&g_prof[sec][1], 64ull * n);
            atomicAdd(&g_prof[sec][2], n);
        }
    }
};
#define VSPG_PROF_ACC(name, sec) ProfAccum name(sec)
#define VSPG_PROF_ACC_BEGIN(name) name.begin()
#define VSPG_PROF_ACC_END(name) name.end()
#else
#define VSPG_PROF(sec)
#define VSPG_PROF_ACC(name, sec)
#define VSPG_PROF_ACC_BEGIN(name)
#define VSPG_PROF_ACC_END(name)
#endif

// ---------------------------------------------------------------------------------------
// constants (src/pbrt/util/math.h:30-60, util/float.h:27)
// ---------------------------------------------------------------------------------------
constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.31830988618379067154f;
constexpr float kInv4Pi = 0.07957747154594766788f;
constexpr float kPiOver2 = 1.57079632679489661923f;
constexpr float kPiOver4 = 0.78539816339744830961f;
constexpr float kOneMinusEps = 0x1.fffffep-1f;
constexpr float kMachineEps = 0x1p-24f;
constexpr float kShadowEps = 0.0001f;
constexpr float kFltMax = 3.402823466e+38f;
constexpr float kInf = __builtin_huge_valf();

// ---------------------------------------------------------------------------------------
// vectors / RGB spectra
// ---------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
VDEV V3 mk(float x, float y, float z) { return V3{x, y, z}; }
VDEV V3 ld3(const float *p) { return V3{p[0], p[1], p[2]}; }
VDEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
VDEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
VDEV V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
VDEV V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
VDEV V3 operator*(float s, V3 a) { return V3{a.x * s, a.y * s, a.z * s}; }
VDEV V3 vabs(V3 a) { return V3{__builtin_fabsf(a.x), __builtin_fabsf(a.y), __builtin_fabsf(a.z)}; }
VDEV float sqr(float x) { return x * x; }
VDEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // vecmath.h:964
VDEV float absdot(V3 a, V3 b) { return __builtin_fabsf(dot(a, b)); }
VDEV float len2(V3 a) { return sqr(a.x) + sqr(a.y) + sqr(a.z); }           // vecmath.h:948
VDEV float len(V3 a) { return __builtin_sqrtf(len2(a)); }
VDEV V3 normalize(V3 a) {  // vecmath.h:959: v / Length(v), one division per component
    float l = len(a);
    return V3{a.x / l, a.y / l, a.z / l};
}
VDEV float diff_of_products(float a, float b, float c, float d) {  // math.h:569-574
    float cd = c * d;
    float dop = __builtin_fmaf(a, b, -cd);
    float err = __builtin_fmaf(-c, d, cd);
    return dop + err;
}
VDEV V3 cross(V3 v, V3 w) {  // vecmath.h:999-1004
    return V3{diff_of_products(v.y, w.z, v.z, w.y), diff_of_products(v.z, w.x, v.x, w.z),
              diff_of_products(v.x, w.y, v.y, w.x)};
}
VDEV V3 lerp(float t, V3 a, V3 b) { return a * (1 - t) + b * t; }  // math.h:210 on tuples
VDEV float fmax_(float a, float b) { return a < b ? b : a; }       // std::max semantics
VDEV float fmin_(float a, float b) { return b < a ? b : a; }       // std::min semantics
VDEV float safe_sqrt(float x) { return __builtin_sqrtf(fmax_(0.f, x)); }
VDEV float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
VDEV bool isinf_(float x) { return __builtin_isinf(x); }
VDEV bool isnan_(float x) { return x != x; }

struct Spec {
    float r, g, b;
};
VDEV Spec sp(float v) { return Spec{v, v, v}; }
VDEV Spec lds(const float *p) { return Spec{p[0], p[1], p[2]}; }
// runtime channel select without runtime-indexed arrays (they would go to scratch)
VDEV float ch_of(Spec s, int ch) { return ch == 0 ? s.r : (ch == 1 ? s.g : s.b); }
VDEV Spec operator+(Spec a, Spec b) { return Spec{a.r + b.r, a.g + b.g, a.b + b.b}; }
VDEV Spec operator-(Spec a, Spec b) { return Spec{a.r - b.r, a.g - b.g, a.b - b.b}; }
VDEV Spec operator*(Spec a, Spec b) { return Spec{a.r * b.r, a.g * b.g, a.b * b.b}; }
// IEEE division is a ~10-instruction sequence on gfx950 and the path divides spectra all the time.
// When every lane of the wavefront holds a GREY spectrum (r == g == b bitwise-equal values, the
// usual case for fog / white walls: beta, r_u, r_l, T_maj, sigma_*), one channel is computed and
// copied: identical results, a third of the work.  The test is wave-uniform (__all), so it is a
// scalar branch, never divergence.
// (bitwise equality: a spectrum built as {x, x, x} from ONE value is grey at compile time -- see
// HomogeneousMediumT<true> -- and (0, -0, 0), whose channels would differ in sign after a division, is not.)
VDEV bool grey(Spec a) { return __float_as_uint(a.r) == __float_as_uint(a.g) && __float_as_uint(a.g) == __float_as_uint(a.b); }
VDEV Spec operator/(Spec a, Spec b) {
    if (__all(grey(a) && grey(b))) {
        float q = a.r / b.r;
        return Spec{q, q, q};
    }
    return Spec{a.r / b.r, a.g / b.g, a.b / b.b};
}
VDEV Spec operator*(Spec a, float f) { return Spec{a.r * f, a.g * f, a.b * f}; }
VDEV Spec operator/(Spec a, float f) {
    if (__all(grey(a))) {
        float q = a.r / f;
        return Spec{q, q, q};
    }
    return Spec{a.r / f, a.g / f, a.b / f};
}
// ---- WEIGHT quotients (vspg_arith.h) -------------------------------------------------------------------------------------
// wdiv(a, b) is a / b at a site whose result only scales a path's contribution (throughput, MIS weights, pdfs, reservoir and
// roulette probabilities): never a distance, a direction or a point.  kArith == 0: the IEEE division the reference performs, bit
// for bit.  kArith >= 1: a * v_rcp_f32(b) (1 ulp reciprocal; the same special cases: x / 0 = inf, 0 / 0 = nan, x / inf = 0).
constexpr int kArith = VSPG_ARITH;
VDEV float wrcp(float b) {
    if constexpr (kArith >= 1) return __builtin_amdgcn_rcpf(b);
    else return 1 / b;
}
VDEV float wdiv(float a, float b) {
    if constexpr (kArith >= 1) return a * __builtin_amdgcn_rcpf(b);
    else return a / b;
}
VDEV Spec wdiv(Spec a, float f) {
    if constexpr (kArith >= 1) {
        const float r = __builtin_amdgcn_rcpf(f);
        return Spec{a.r * r, a.g * r, a.b * r};
    } else {
        return a / f;
    }
}
VDEV Spec wdiv(Spec a, Spec b) {
    if constexpr (kArith >= 1) return Spec{a.r * __builtin_amdgcn_rcpf(b.r), a.g * __builtin_amdgcn_rcpf(b.g), a.b * __builtin_amdgcn_rcpf(b.b)};
    else return a / b;
}
VDEV bool nonzero(Spec a) { return a.r != 0 || a.g != 0 || a.b != 0; }  // spectrum.h:263-268
VDEV float avg(Spec a) {  // spectrum.h:288-294 (its callers weigh and decide with it: MIS weights, Russian roulette)
    float s = a.r;
    s += a.g;
    s += a.b;
    if constexpr (kArith >= 1) return s * 0.333333343f;
    else return s / 3;
}
VDEV float maxc(Spec a) { return fmax_(fmax_(a.r, a.g), a.b); }
VDEV Spec clamp_zero(Spec a) { return Spec{fmax_(0.f, a.r), fmax_(0.f, a.g), fmax_(0.f, a.b)}; }
VDEV bool has_nan(Spec a) { return isnan_(a.r) || isnan_(a.g) || isnan_(a.b); }
VDEV bool has_inf(Spec a) { return isinf_(a.r) || isinf_(a.g) || isinf_(a.b); }

// ---------------------------------------------------------------------------------------
// float <-> bits, NextFloatUp/Down (util/float.h:164-193)
// ---------------------------------------------------------------------------------------
VDEV uint32_t f2b(float f) { return __float_as_uint(f); }
VDEV float b2f(uint32_t u) { return __uint_as_float(u); }
// branch-free integer forms: identical results for every non-NaN input (the reference's branches
// on +-inf and -0 become selects)
VDEV float next_float_up(float v) {
    uint32_t ui = f2b(v);
    ui = (ui == 0x80000000u) ? 0u : ui;                          // -0 -> +0
    uint32_t r = ((int32_t)ui >= 0) ? ui + 1u : ui - 1u;         // v >= 0 ? ++ui : --ui
    return b2f(ui == 0x7f800000u ? ui : r);                      // +inf stays
}
VDEV float next_float_down(float v) {
    uint32_t ui = f2b(v);
    ui = (ui == 0u) ? 0x80000000u : ui;                          // +0 -> -0
    uint32_t r = ((int32_t)ui > 0) ? ui - 1u : ui + 1u;          // v > 0 ? --ui : ++ui
    return b2f(ui == 0xff800000u ? ui : r);                      // -inf stays
}

// ---------------------------------------------------------------------------------------
// a2: MurmurHash64A specialised to the key sizes the path uses (util/hash.h:19-63, 96-103)
// ---------------------------------------------------------------------------------------
constexpr uint64_t kMurmurM = 0xc6a4a7935bd1e995ull;
VDEV uint64_t murmur_finish(uint64_t h) {
    h ^= h >> 47;
    h *= kMurmurM;
    h ^= h >> 47;
    return h;
}
// Hash(float): len 4 -> no 8-byte block, tail case 4..1 then h *= m
VDEV uint64_t hash_u32(uint32_t w) {
    uint64_t h = 0 ^ (4ull * kMurmurM);
    h ^= (uint64_t)w;
    h *= kMurmurM;
    return murmur_finish(h);
}
VDEV uint64_t hash_float(float f) { return hash_u32(f2b(f)); }
// 12-byte keys: Hash(Point3f), Hash(Point2i, int): one 8-byte block + 4-byte tail
VDEV uint64_t hash_3u32(uint32_t w0, uint32_t w1, uint32_t w2) {
    uint64_t h = 0 ^ (12ull * kMurmurM);
    uint64_t k = (uint64_t)w0 | ((uint64_t)w1 << 32);
    k *= kMurmurM;
    k ^= k >> 47;
    k *= kMurmurM;
    h ^= k;
    h *= kMurmurM;
    h ^= (uint64_t)w2;
    h *= kMurmurM;
    return murmur_finish(h);
}
VDEV uint64_t hash_v3(V3 p) { return hash_3u32(f2b(p.x), f2b(p.y), f2b(p.z)); }
VDEV uint64_t mix_bits(uint64_t v) {  // hash.h:70-77
    v ^= (v >> 31);
    v *= 0x7fb5d329728ea185ull;
    v ^= (v >> 27);
    v *= 0x81dadef4bc2dd44dull;
    v ^= (v >> 33);
    return v;
}

// ---------------------------------------------------------------------------------------
// a1: PCG32 (util/rng.h:30-150)
// ---------------------------------------------------------------------------------------
constexpr uint64_t kPcgMult = 0x5851f42d4c957f2dULL;
struct Rng {
    uint64_t state, inc;
    VDEV uint32_t u32() {  // rng.h:82-88
        uint64_t old = state;
        state = old * kPcgMult + inc;
        uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
    }
    VDEV float uniform() {  // rng.h:128-130
        float f = (float)u32() * 0x1p-32f;
        return fmin_(kOneMinusEps, f);
    }
    VDEV void set_sequence(uint64_t seq, uint64_t seed) {  // rng.h:119-125
        state = 0u;
        inc = (seq << 1u) | 1u;
        u32();
        state += seed;
        u32();
    }
    VDEV void set_sequence(uint64_t seq) { set_sequence(seq, mix_bits(seq)); }  // rng.h:43-45
    VDEV void advance(uint64_t delta) {  // rng.h:137-150
        uint64_t curMult = kPcgMult, curPlus = inc, accMult = 1u, accPlus = 0u;
        while (delta > 0) {
            if (delta & 1) {
                accMult *= curMult;
                accPlus = accPlus * curMult + curPlus;
            }
            curPlus = (curMult + 1) * curPlus;
            curMult *= curMult;
            delta /= 2;
        }
        state = accMult * state + accPlus;
    }
};

// RNG::Advance(delta) maps state -> accMult*state + accPlus, and accPlus is linear in `inc`
// (every curPlus in rng.h:137-150 is a multiple of inc), so accPlus == inc * G(delta) mod 2^64 with
// G(delta) the same recurrence started from curPlus = 1.  (accMult, G) depend on delta only: the
// host computes them once per launch and every lane applies them with two 64-bit multiplies instead
// of running the ~24-iteration skip-ahead loop per path.  Exact (integer arithmetic mod 2^64).
struct PcgJump {
    uint64_t mult, g;
};
// IndependentSampler (src/pbrt/samplers.h:442-476)
struct Sampler {
    Rng rng;
    VDEV void start_pixel_sample(int px, int py, int seed, int sampleIndex) {
        rng.set_sequence(hash_3u32((uint32_t)px, (uint32_t)py, (uint32_t)seed));
        rng.advance((uint64_t)sampleIndex * 65536ull);
    }
    VDEV void start_pixel_sample(int px, int py, int seed, PcgJump j) {  // j = jump for sampleIndex*65536
        rng.set_sequence(hash_3u32((uint32_t)px, (uint32_t)py, (uint32_t)seed));
        rng.state = j.mult * rng.state + rng.inc * j.g;
    }
    VDEV float get1d() { return rng.uniform(); }
};

// ---------------------------------------------------------------------------------------
// libm: float functions reproduce the host glibc bit for bit (vspg_libm.h); the reference's
// `std::log(1.0 - x)` in media_sampleTMaj.h is DOUBLE precision -> the double log of vspg_libm.h, rounded once
// ---------------------------------------------------------------------------------------
// (kArith == 2, vspg_arith.h: the hardware's v_log_f32 / v_sin_f32 / v_cos_f32 -- 1 ulp log2, sin / cos of an argument in turns)
VLEAF float logf_(float x) {
    if constexpr (kArith >= 2) return __builtin_amdgcn_logf(x) * 0.693147182f;
    else return vspg_libm::logf_host_exact(x);
}
VLEAF float sinf_(float x) {
    if constexpr (kArith >= 2) return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x * 0.159154937f));
    else return vspg_libm::sinf_host_exact(x);
}
VLEAF float cosf_(float x) {
    if constexpr (kArith >= 2) return __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(x * 0.159154937f));
    else return vspg_libm::cosf_host_exact(x);
}
VLEAF float neg_log1m_d(float x) {  // -std::log(1.0 - x)
    if constexpr (kArith >= 2) return -(__builtin_amdgcn_logf(1 - x) * 0.693147182f);
    else return (float)(-vspg_libm::log_host_exact(1.0 - (double)x));
}

// ---------------------------------------------------------------------------------------
// a3: FastExp / SampleExponential / SampleDiscrete
// ---------------------------------------------------------------------------------------
VDEV float fast_exp(float x) {  // util/math.h:450-474 (CPU branch; NOT __expf)
    float xp = x * 1.442695041f;
    float fxp = __builtin_floorf(xp), f = xp - fxp;
    int i = (int)fxp;
    float twoToF = __builtin_fmaf(f, __builtin_fmaf(f, __builtin_fmaf(f, 0.0781455737f, 0.226173572f), 0.695556856f), 1.f);
    // f is in [0, 1] and the polynomial maps it into [1, 2): Exponent(twoToF) == 0, so the reference's exponent
    // surgery (exponent = Exponent(twoToF) + i, two range checks, bit splice) is an exact scaling by 2^i --
    // one v_ldexp_f32 (overflow gives +inf like the reference's second early return) plus the underflow select.
    float r = __builtin_ldexpf(twoToF, i);
    return i < -126 ? 0.f : r;
}
VDEV Spec fast_exp(Spec a) {
    if (__all(grey(a))) {  // wave-uniform grey fast path, see operator/
        float e = fast_exp(a.r);
        return Spec{e, e, e};
    }
    return Spec{fast_exp(a.r), fast_exp(a.g), fast_exp(a.b)};
}
VDEV float sample_exponential(float u, float a) { return -logf_(1 - u) / a; }  // sampling.h:222
// ---- blackbody emission of temperature grids in the RGB build ----
// SampledWavelengths::SampleVisible (util/spectrum.h:369-386): lambda_i = SampleVisibleWavelengths(up_i) (util/sampling.h:169-171),
// std::atanh(float) = the host's atanhf (vspg_libm.h)
VDEV float sample_visible_wavelength(float u, int i) {
    float up = u + (float)i / 3;
    if (up > 1) up -= 1;
    return 538 - 138.888889f * vspg_libm::atanhf_host_exact(0.85691062f - 1.82750197f * up);
}
VDEV float blackbody(float lambda, float T) {  // Blackbody() (util/spectrum.h:83-94); Pow<5>(l) = ((l l)(l l)) l (util/math.h:294-309)
    if (T <= 0) return 0;
    const float c = 299792458.f;
    const float h = 6.62606957e-34f;
    const float kb = 1.3806488e-23f;
    float l = lambda * 1e-9f;
    float l2 = l * l;
    float l5 = l2 * l2 * l;
    return (2 * h * c * c) / (l5 * (fast_exp((h * c) / (l * kb * T)) - 1));
}
VDEV Spec blackbody_sample(float T, float lu) {  // BlackbodySpectrum(T).Sample(lambda) (util/spectrum.h:568-588)
    float lambdaMax = 2.8977721e-3f / T;
    float normalizationFactor = 1 / blackbody(lambdaMax * 1e9f, T);
    return Spec{blackbody(sample_visible_wavelength(lu, 0), T) * normalizationFactor, blackbody(sample_visible_wavelength(lu, 1), T) * normalizationFactor,
                blackbody(sample_visible_wavelength(lu, 2), T) * normalizationFactor};
}
VDEV int sample_discrete2(float w0, float w1, float u) {  // sampling.h:79-113, two weights
    float sumWeights = 0;
    sumWeights += w0;
    sumWeights += w1;
    float up = u * sumWeights;
    if (up == sumWeights) up = next_float_down(up);
    // while (sum + weights[offset] <= up) sum += weights[offset++];
    if (!(0.f + w0 <= up)) return 0;
    return 1;  // the reference would index past two weights only if w0+w1 <= up, which
               // cannot happen after the up == sumWeights fix-up (DCHECK_LT in the reference)
}

// ---------------------------------------------------------------------------------------
// a4: Henyey-Greenstein, frames, direction sampling
// ---------------------------------------------------------------------------------------
VDEV float henyey_greenstein(float cosTheta, float g) {  // scattering.h:50-59
    // isotropic media (g == 0, the benchmark's fog): denom = 1 + 0 + 2 * 0 * cosTheta is exactly 1 for every finite cosTheta,
    // so the expression below is kInv4Pi * 1 / (1 * 1) = kInv4Pi bit for bit -- without its square root and division
    if (g == 0.f && __builtin_fabsf(cosTheta) < kInf) return kInv4Pi;
    g = clampf(g, (float)-.99, (float).99);
    float denom = 1 + sqr(g) + 2 * g * cosTheta;
    return wdiv(kInv4Pi * (1 - sqr(g)), denom * safe_sqrt(denom));
}
VDEV void coordinate_system(V3 v1, V3 *v2, V3 *v3) {  // vecmath.h:1007-1013
    float sign = __builtin_copysignf(1.f, v1.z);
    float a = -1 / (sign + v1.z);
    float b = v1.x * v1.y * a;
    *v2 = V3{1 + sign * sqr(v1.x) * a, sign * b, -sign * v1.x};
    *v3 = V3{b, sign + sqr(v1.y) * a, -v1.y};
}
struct Frame {
    V3 x, y, z;
    VDEV V3 from_local(V3 v) const { return v.x * x + v.y * y + v.z * z; }  // vecmath.h:1914
    VDEV V3 to_local(V3 v) const { return V3{dot(v, x), dot(v, y), dot(v, z)}; }
};
// SampleHenyeyGreenstein (sampling.cpp:348-374) split around its sin/cos so that the kernel can
// evaluate sinf/cosf once for phase-function and BSDF lanes together (same operations, same order
// per lane): hg_pre returns phi, hg_post finishes with sin(phi), cos(phi).
VDEV float hg_pre(float g, float u0, float u1, float *sinTheta, float *cosTheta) {
    g = clampf(g, (float)-.99, (float).99);
    float ct;
    if (__builtin_fabsf(g) < 1e-3f)
        ct = 1 - 2 * u0;
    else
        ct = -1 / (2 * g) * (1 + sqr(g) - sqr((1 - sqr(g)) / (1 + g - 2 * g * u0)));
    *cosTheta = ct;
    *sinTheta = safe_sqrt(1 - sqr(ct));
    return 2 * kPi * u1;
}
VDEV V3 hg_post(V3 wo, float g, float sinTheta, float cosTheta, float sinPhi, float cosPhi, float *pdf) {
    g = clampf(g, (float)-.99, (float).99);
    Frame f;
    f.z = wo;
    coordinate_system(wo, &f.x, &f.y);
    // SphericalDirection (vecmath.h:1666-1672)
    V3 local = V3{clampf(sinTheta, -1, 1) * cosPhi, clampf(sinTheta, -1, 1) * sinPhi, clampf(cosTheta, -1, 1)};
    *pdf = henyey_greenstein(cosTheta, g);
    return f.from_local(local);
}
VDEV V3 sample_henyey_greenstein(V3 wo, float g, float u0, float u1, float *pdf) {
    float st, ct;
    float phi = hg_pre(g, u0, u1, &st, &ct);
    return hg_post(wo, g, st, ct, sinf_(phi), cosf_(phi), pdf);
}
// SampleCosineHemisphere / SampleUniformDiskConcentric (sampling.h:325-341, 409-413), split the same way
VDEV float cos_hemi_pre(float u0, float u1, float *r, bool *degenerate) {
    float ox = 2 * u0 - 1, oy = 2 * u1 - 1;
    *degenerate = (ox == 0 && oy == 0);
    // both branches of the concentric map as one select (a single division)
    const bool xmajor = __builtin_fabsf(ox) > __builtin_fabsf(oy);
    const float q = (xmajor ? oy : ox) / (xmajor ? ox : oy);
    *r = xmajor ? ox : oy;
    const float t = kPiOver4 * q;
    return xmajor ? t : kPiOver2 - t;
}
VDEV V3 cos_hemi_post(float r, bool degenerate, float sinT, float cosT) {
    float dx = degenerate ? 0.f : r * cosT;
    float dy = degenerate ? 0.f : r * sinT;
    float z = safe_sqrt(1 - sqr(dx) - sqr(dy));
    return V3{dx, dy, z};
}
VDEV V3 sample_cosine_hemisphere(float u0, float u1) {
    float r;
    bool deg;
    float theta = cos_hemi_pre(u0, u1, &r, &deg);
    return cos_hemi_post(r, deg, sinf_(theta), cosf_(theta));
}

// ---------------------------------------------------------------------------------------
// Point3fi (vecmath.h:737-760; Interval math.h:815-862) and ray spawning (ray.h:75-108)
// ---------------------------------------------------------------------------------------
struct P3i {
    V3 lo, hi;
    VDEV V3 mid() const { return V3{(lo.x + hi.x) / 2, (lo.y + hi.y) / 2, (lo.z + hi.z) / 2}; }
    VDEV V3 err() const { return V3{(hi.x - lo.x) / 2, (hi.y - lo.y) / 2, (hi.z - lo.z) / 2}; }
};
VDEV P3i p3i_exact(V3 p) { return P3i{p, p}; }
VDEV void interval_ve(float v, float e, float *lo, float *hi) {
    float l = next_float_down(v - e), h = next_float_up(v + e);
    *lo = e == 0 ? v : l;
    *hi = e == 0 ? v : h;
}
VDEV P3i p3i_from_err(V3 p, V3 e) {
    P3i r;
    interval_ve(p.x, e.x, &r.lo.x, &r.hi.x);
    interval_ve(p.y, e.y, &r.lo.y, &r.hi.y);
    interval_ve(p.z, e.z, &r.lo.z, &r.hi.z);
    return r;
}
VDEV float offset_axis(float po, float off) {
    float up = next_float_up(po), dn = next_float_down(po);
    return off > 0 ? up : (off < 0 ? dn : po);
}
VDEV V3 offset_ray_origin(P3i pi, V3 n, V3 w) {
    float d = dot(vabs(n), pi.err());
    V3 offset = n * d;
    if (dot(w, n) < 0) offset = -offset;
    V3 po = pi.mid() + offset;
    return V3{offset_axis(po.x, offset.x), offset_axis(po.y, offset.y), offset_axis(po.z, offset.z)};
}

// ---------------------------------------------------------------------------------------
// device-resident scene (uploaded once; uniform accesses become scalar loads)
// ---------------------------------------------------------------------------------------
#define VSPG_GLOBAL_AS __attribute__((address_space(1)))
struct DQuad {
    float p00[3], p10[3], p01[3], p11[3], e1[3], e2[3];
    float n[3], dpdu_n[3], perr[3];
    float inv_l1, inv_l2, area;
    float Kd[3], Le[3];
    int32_t two_sided, is_light, has_lobes;
    int32_t flags;  // SURF_* (medium boundaries)
};
// Medium boundaries (round 4; include/vspg.h VSPG_MATERIAL_* / VSPG_IFACE_*): what a surface hands a hit besides its BSDF --
// bit 0: Material "interface" (no BSDF: Li skips the hit, :399-404); bits 1-2: the VSPG_IFACE_* bits of a medium TRANSITION
// (MediumInterface with inside != outside), 0 for every other surface (the ray's medium carries on, interaction.h:218-229)
enum { SURF_INTERFACE = 1, SURF_IFACE_SHIFT = 1 };
// Packed per-rectangle intersection record: ONE 64-byte scalar load per test (s_load_dwordx16) with
// no dependent address arithmetic.  kind 0: generic {n, p00, e1, e2, inv_l1, inv_l2}; kind 1: axis
// aligned {nsign, pa = p00[axis], pu = p00[uaxis], pv = p00[vaxis], l1, l2, inv_l1, inv_l2}.
struct IsectRec {
    float f[14];
    int32_t kind;   // 0 generic, 1 axis-aligned
    int32_t axes;   // axis | uaxis << 2 | vaxis << 4
};
// guiding field in HBM (see vspg_guiding.h): [0] surface, [1] volume; nodes == nullptr -> untrained
struct DField {
    int32_t n_nodes, n_regions;
    const VspgKdNode *nodes;
    const VspgFieldRegion *regions;
    // per region 2 x VSPG_FIELD_LOBES floats: b_k = weight[k] * vmf_norm(kappa_clamp(kappa[k])) then kappa_clamp(kappa[k]) -- functions of
    // the stored lobes only, evaluated once per field update by the same device code the queries would run
    // (k_field_aux) instead of once per lobe per mixture evaluation (an IEEE division and a FastExp each)
    const float *aux;
    // per region 1 + 2 x VSPG_FIELD_LOBES float4: the same region as an array of lobes, for the workgroup kernel's guided vertex
    // (vspg_guided_wg.h: region_lobes): {pivot, n_lobes}, then per lobe {mu, distance} {weight, b, kappa (clamped), vsp}
    const float4 *lobes;
};
// Light samplers beyond the uniform pick (vspg_lightsampler.h): the BVH light sampler's nodes as CompactLightBounds' accessors
// return them, the light -> bit-trail map of its PMF, the power sampler's alias table.  Built on the host with the renderer.
constexpr int kMaxLights = VSPG_MAX_QUADS + VSPG_MAX_INFINITE_LIGHTS;
struct DLightNode {
    float bmin[3], bmax[3], w[3];
    float phi, cosTheta_o, cosTheta_e;
    int32_t twoSided;
    uint32_t child_or_light;  // interior: second child (the first is node + 1); leaf: index into the light list
    int32_t is_leaf;
};
struct DLightSampler {
    int32_t mode;             // VSPG_LIGHTSAMPLER_*; scenes with at most one light are given UNIFORM (the same pick, pmf 1)
    int32_t n_nodes, n_inf, n_alias;
    DLightNode nodes[2 * kMaxLights];
    int32_t inf_light[VSPG_MAX_INFINITE_LIGHTS];
    uint32_t bit_trail[kMaxLights];
    float alias_p[kMaxLights], alias_q[kMaxLights];
    int32_t alias_i[kMaxLights];
    int32_t light_of_quad[VSPG_MAX_QUADS];  // rectangle -> index into the light list (-1: not a light)
};
// Triangle geometry.  DTri: the three vertices plus what Triangle::InteractionFromIntersection
// (shapes.h:883-1010) derives from them alone -- n = Normalize(Cross(p0 - p2, p1 - p2)) and the normalised dpdu of the
// default (u,v) parameterisation -- computed once on the host with the same float operations.  80 bytes = five 16-byte loads.
struct DTri {
    float p0[3], nx;
    float p1[3], ny;
    float p2[3], nz;
    float dpdu_n[3];
    int32_t id;      // index in the caller's soup (the tie-break of equal hit distances)
    float Kd[3];
    int32_t flags;   // SURF_* (nx, ny, nz already carry the VSPG_TRI_FLIP_NORMAL flip)
};
// Shape "sphere" (shapes.h:107-330; full spheres): rows 0..2 of renderFromObject's m and mInv, the constants the Sphere
// constructor derives (computed on the host with the same libm calls), DiffuseMaterial reflectance or interface flags
struct DSphere {
    float m[12], mi[12];
    float radius, thetaZMin, thetaZMax, phiMax;
    float Kd[3];
    int32_t flip, has_lobes, flags;
};
// Four-wide BVH (round 5; own builder in vspg_capi.hip: binned SAH over triangles, then every inner node absorbs the children of
// its largest children until it has four).  A node is the BOXES OF ITS CHILDREN, structure-of-arrays -- one 128-byte fetch tests
// four boxes, and a ray makes half as many dependent fetches on its way down as through the binary tree of rounds 1-4 -- plus
// what each child is: >= 0 an inner node's index, < 0 a leaf -(first triangle * 8 + count) - 1 (count 1..7), kBvhAbsent: no child.
// The traversal keeps the hit children it has not visited on a small per-lane stack (cpu/aggregates.cpp:529-590 does the same
// with `nodesToVisit[64]`); the builder keeps the tree shallow enough for it (kBvhStack).
struct DBvh4Node {
    float lox[4], loy[4], loz[4], hix[4], hiy[4], hiz[4];
    int32_t child[4];
    int32_t pad[4];
};
static_assert(sizeof(DBvh4Node) == 128, "one node = eight 16-byte loads");
constexpr int32_t kBvhAbsent = (int32_t)0x80000000;
constexpr int kBvhStack = 48;   // entries; a node leaves at most three behind per level, the builder bounds the depth at 15
struct DScene {
    int32_t n_quads, n_lights;
    int32_t n_tris, n_bvh_nodes;
    // medium boundaries: has_boundaries = some surface is an interface material or a medium transition, or the camera sits
    // outside the medium (0: the medium fills the scene -- every boundary branch below is skipped, wave-uniformly);
    // camera_in_medium = the camera ray's medium (1 = the scene's medium)
    int32_t n_spheres, has_boundaries, camera_in_medium, pad_b;
    DSphere spheres[VSPG_MAX_SPHERES];
    const DTri *tris;          // in BVH leaf order
    const DBvh4Node *bvh;
    // infinite lights (light order: the emissive rectangles, then these); scene_radius: Bounds3::BoundingSphere of the scene bounds
    int32_t n_inf;
    int32_t inf_type[VSPG_MAX_INFINITE_LIGHTS];
    float inf_L[VSPG_MAX_INFINITE_LIGHTS][3], inf_w[VSPG_MAX_INFINITE_LIGHTS][3];
    float scene_radius;
    DField field[2];
    IsectRec irec[VSPG_MAX_QUADS];
    int32_t light_quads[VSPG_MAX_QUADS];
    DQuad quads[VSPG_MAX_QUADS];
    VspgCamera cam;
    // medium
    int32_t medium_type;
    float sigma_a[3], sigma_s[3], Le[3], g;
    // homogeneous medium: sigma_s + sigma_a and (sigma_t - sigma_a) - sigma_s, formed on the host with the
    // same IEEE float operations the kernels would use.  They are wave-uniform; the vector ALU is the
    // only float adder on this chip, so computing them in the kernel parks the results in VGPRs for
    // the whole persistent loop -- as scalar loads they stay in SGPRs.
    float sigma_t[3], sigma_n_raw[3];
    // grid medium (GridMedium, media.h:284-390): density samples nx*ny*nz (x fastest) and the 16^3
    // majorant grid (media.cpp:252-269), both in HBM
    int32_t nx, ny, nz;
    float bounds_min[3], bounds_max[3];
    const float *density;       // raw samples: only the brick build reads them (freed afterwards)
    const float *majorant;
    // Device layout of the density grid ("octet bricks behind a coarse index"): the trilinear filter of a query needs the
    // 2x2x2 voxels around it; as 8 scattered 4-byte loads from the raw array they cost ~4 cache lines (512 B of HBM /
    // Infinity-Cache traffic per query -- the walk kernels ran at ~6 TB/s of it).  Here every base voxel (ix, iy, iz) in
    // [-1, n-1]^3 owns an OCTET {v000, v100, v010, v110, v001, v101, v011, v111} (voxels outside the grid = 0, the
    // reference's bounds checks baked in): a query is ONE aligned 32-byte fetch.  Octets are stored per 8x8x8 BRICK and
    // only for bricks that hold a non-zero value; brick_index[bz][by][bx] is the brick's slot or -1 (all zero: no fetch).
    // 8x the voxel bytes for the occupied part -- this part has 288 GB.  Same floats, same interpolation: same bits.
    const int32_t *brick_index;
    const float4 *octets;
    int32_t bnx, bny, bnz;
    // renderFromMedium of grid media: rows 0..2 of mInv (row-major, affine), has_xform = 0: identity
    int32_t has_xform;
    float minv[12];
    // NanoVDB-semantics dense medium: index bbox min, 1 / voxel_size, world position of index (0,0,0), density offset
    int32_t index_min[3];
    float inv_voxel[3], grid_origin[3], density_offset;
    // emissive GridMedium (media.h:326-342): the LeScale grid, null = not emissive
    const float *le_scale;
    int32_t le_nx, le_ny, le_nz;
    // temperature grid of an emissive GridMedium / NanoVDBMedium (media.h:333-341, :724-735): raw samples in the density grid's
    // layout (only the delta-tracking callback's emission reads it: eight plain loads, no bricks), null = none
    const float *temperature;
    float temperature_offset, temperature_scale, nvdb_le_scale;
    // integrator parameters
    VspgIntegratorParams prm;
    // render config
    int32_t xres, yres, seed, shard_index, shard_count;
    // TrBuffer (cpu/trbuffer.h): running mean of the primary rays' ratio-tracking transmittance, RGB per pixel;
    // tr_calc = calculateTrBuffer (the resampling routine records), tr_load = trBufferLoad (NDS+ reads)
    float *tr_rgb;
    int32_t *tr_spp;
    int32_t tr_calc, tr_load;
    // guided Russian roulette: the image-space contribution estimate (one float per pixel, 0 = none; own stand-in for
    // ImageSpaceGuidingBuffer::GetContributionEstimate), refreshed with the VSP buffer
    const float *contrib;
    int32_t contrib_ready;
    // LightSampler::Create(prm.lightsampler, lights): lsamp.mode is UNIFORM wherever the pick is trivial
    DLightSampler lsamp;
};
// DScene::has_boundaries as the path code reads it.  (-DVSPG_NO_BOUNDARIES: a diagnostic build without the boundary code, to price
// its presence in the kernels that serve scenes without boundaries -- same results on those scenes.)
// BND: what the instantiation knows at compile time -- 0: no boundaries (the code for them is not compiled in: the wavefront
// pipeline's instantiations for scenes where the medium fills everything keep their round-3 shape), 1: boundaries, -1: decided at
// run time (the per-lane kernels, which serve everything).  Medium types carry it as kBnd.
#ifdef VSPG_NO_BOUNDARIES
template <int BND> VDEV bool has_bnd(const DScene &) { return false; }
#else
template <int BND> VDEV bool has_bnd(const DScene &S) {
    if constexpr (BND >= 0) return BND != 0;
    else return S.has_boundaries != 0;
}
#endif
// vsp_ready as handed to the path functions: bit 0 = the VSP buffer holds estimates; bit 1 = a debug path trace
// (vspg_trace_paths), which does not feed the per-pixel buffers
enum { VSP_READY = 1, VSP_NO_FEED = 2 };

// Rectangle records that are indexed PER LANE (the rectangle a lane hit, the light it sampled) are
// read from an LDS copy: through the DScene in HBM each such access is a vector load with a full
// L2 round trip (~1 us, and the light lookup is two dependent ones) that two waves per SIMD cannot
// hide; measured, those loads were the largest single wait of the path kernels.  Wave-uniform
// accesses (S.irec[i], camera, medium, parameters) stay scalar loads from the constant cache.
// Every kernel that runs path code calls stage_scene_lds() + __syncthreads() first (it also stages
// the logf table of vspg_libm.h).
__shared__ DQuad s_scene_quads[VSPG_MAX_QUADS];
__shared__ int32_t s_scene_light_quads[VSPG_MAX_QUADS];
// Homogeneous-medium constants {sigma_a[3], sigma_s[3], Le[3], g, sigma_t[3], sigma_n_raw[3]}: as one
// 16-dword scalar load they stayed live over the whole persistent loop, and the register allocator
// spilled and restored the 16-SGPR tuple as a unit (16 v_readlane + hazard nops at each of 32 use
// sites -- 7 % of the kernel's vector instructions).  From LDS each site reads the 1-3 floats it needs.
enum { MED_SIGMA_A = 0, MED_SIGMA_S = 3, MED_LE = 6, MED_G = 9, MED_SIGMA_T = 10, MED_SIGMA_N_RAW = 13, MED_COUNT = 16 };
__shared__ float s_scene_medium[MED_COUNT];
VDEV void stage_scene_lds(const DScene &S) {
    vspg_libm::stage_logf_tab_lds();
    vspg_libm::stage_log_tab_lds();
    const uint32_t *src = reinterpret_cast<const uint32_t *>(S.quads);
    uint32_t *dst = reinterpret_cast<uint32_t *>(s_scene_quads);
    const int n = S.n_quads * (int)(sizeof(DQuad) / 4);
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    if (threadIdx.x < VSPG_MAX_QUADS) s_scene_light_quads[threadIdx.x] = S.light_quads[threadIdx.x];
    if (threadIdx.x < MED_COUNT) {
        const int k = threadIdx.x;
        s_scene_medium[k] = k < 3 ? S.sigma_a[k] : k < 6 ? S.sigma_s[k - 3] : k < 9 ? S.Le[k - 6] : k == 9 ? S.g
                            : k < 13 ? S.sigma_t[k - 10] : S.sigma_n_raw[k - 13];
    }
}
VDEV const DQuad &quad_at(int i) { return s_scene_quads[i]; }
VDEV const DQuad &light_quad_at(int lightIndex) { return s_scene_quads[s_scene_light_quads[lightIndex]]; }

// A hit is (prim, three floats): `quad` = rectangle index >= 0, -2 - (triangle position in S.tris), or kSpherePrim + sphere; `p` = the
// re-projected point on a rectangle, the BARYCENTRICS (b0, b1, b2) of a triangle hit, the object-space point of a sphere hit.
// The interaction point with its error bounds (surf_pi), the normal and the shading frame are pure functions of the two,
// recomputed where they are read -- nothing else crosses a phase or kernel boundary (round 4: the per-hit error bound used to
// travel beside the point, three more floats per vertex and per previous-vertex context).
struct Isect {
    bool hit;
    float t;
    int quad;
    V3 p, n;
};
VDEV bool is_tri(int prim) { return prim <= -2; }
VDEV int tri_of(int prim) { return -2 - prim; }
// spheres: prim = kSpherePrim + index.  A sphere hit carries its OBJECT-SPACE hit point in `p` (Isect, Vertex, PrevCtx): the
// interaction point with its error bounds, the normal and the shading frame are pure functions of it (sphere_interaction),
// recomputed where they are read -- three floats name the hit, as (p, rectangle) does for a rectangle.
constexpr int kSpherePrim = 32;
static_assert(VSPG_MAX_QUADS <= kSpherePrim, "rectangle indices lie below the first sphere primitive");
VDEV bool is_sphere(int prim) { return prim >= kSpherePrim; }
VDEV int sphere_of(int prim) { return prim - kSpherePrim; }

// ray / rectangle (own geometry stand-in, contract in DESIGN.md): plane hit t = n.(p00-o)/n.d,
// accepted when 0 < t < tMax and the parametric (u,v) of o+t*d lie in [0,1]^2; the reported point
// is re-projected onto the rectangle, p00 + (u*e1 + v*e2).
// The sign pre-test only skips work whose outcome is already decided: t = num/denom can be > 0
// only when num and denom are non-zero with equal signs.
// Conservative range pre-test: with num, denom of equal sign, |num| > tMax*|denom|*(1+2^-20) implies
// the correctly rounded quotient num/denom is >= tMax, so the exact test `t < tMax` below would
// fail anyway; the IEEE division (~10 instructions) and the (u,v) work are skipped.  The margin
// covers the rounding of the product and of the division (each <= 2^-24 relative).
VDEV bool beyond(float num, float denom, float tMax) {
    return __builtin_fabsf(num) > tMax * __builtin_fabsf(denom) * 1.000001f;
}
VDEV float comp(V3 v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }  // axis is wave-uniform
VDEV bool rect_hit_uv(const IsectRec &r, V3 o, V3 d, float tMax, float *tHit, float *uHit, float *vHit) {
    float num, denom, u, v, t;
    if (r.kind == 1) {
        // axis-aligned rectangle: n.d == nsign*d[a] and n.(p00-o) == nsign*(p00[a]-o[a]) exactly (the other
        // products are +-0), so t == (p00[a]-o[a])/d[a] exactly; likewise (p-p00).e1 == rel[uaxis]*l1.
        const int a = r.axes & 3, ua = (r.axes >> 2) & 3, va = (r.axes >> 4) & 3;
        denom = r.f[0] * comp(d, a);
        num = r.f[0] * (r.f[1] - comp(o, a));
        bool cand = (num > 0 && denom > 0) || (num < 0 && denom < 0);
        if (!cand) return false;
        if (beyond(num, denom, tMax)) return false;
        t = num / denom;
        if (!(t > 0) || !(t < tMax)) return false;
        float pu = comp(o, ua) + comp(d, ua) * t, pv = comp(o, va) + comp(d, va) * t;
        u = ((pu - r.f[2]) * r.f[4]) * r.f[6];
        v = ((pv - r.f[3]) * r.f[5]) * r.f[7];
    } else {
        V3 n = V3{r.f[0], r.f[1], r.f[2]}, p00 = V3{r.f[3], r.f[4], r.f[5]};
        denom = dot(n, d);
        num = dot(n, p00 - o);
        bool cand = (num > 0 && denom > 0) || (num < 0 && denom < 0);
        if (!cand) return false;
        if (beyond(num, denom, tMax)) return false;
        t = num / denom;
        if (!(t > 0) || !(t < tMax)) return false;
        V3 p = o + d * t;
        V3 rel = p - p00;
        u = dot(rel, V3{r.f[6], r.f[7], r.f[8]}) * r.f[12];
        v = dot(rel, V3{r.f[9], r.f[10], r.f[11]}) * r.f[13];
    }
    if (u < 0 || u > 1 || v < 0 || v > 1) return false;
    *tHit = t;
    *uHit = u;
    *vHit = v;
    return true;
}
VDEV bool quad_hit_uv(const DQuad &q, V3 o, V3 d, float tMax, float *tHit, float *uHit, float *vHit) {
    V3 n = ld3(q.n), p00 = ld3(q.p00);
    float denom = dot(n, d);
    float num = dot(n, p00 - o);
    bool cand = (num > 0 && denom > 0) || (num < 0 && denom < 0);
    if (!cand) return false;
    float t = num / denom;
    if (!(t > 0) || !(t < tMax)) return false;
    V3 p = o + d * t;
    V3 rel = p - p00;
    float u = dot(rel, ld3(q.e1)) * q.inv_l1;
    float v = dot(rel, ld3(q.e2)) * q.inv_l2;
    if (u < 0 || u > 1 || v < 0 || v > 1) return false;
    *tHit = t;
    *uHit = u;
    *vHit = v;
    return true;
}
VDEV V3 quad_point(const DQuad &q, float u, float v) { return ld3(q.p00) + (ld3(q.e1) * u + ld3(q.e2) * v); }
VDEV bool quad_intersect(const DQuad &q, V3 o, V3 d, float tMax, float *tHit, V3 *pHit) {
    float u, v;
    if (!quad_hit_uv(q, o, d, tMax, tHit, &u, &v)) return false;
    *pHit = quad_point(q, u, v);
    return true;
}
// ---------------------------------------------------------------------------------------
// f1: triangles.  IntersectTriangle (shapes.cpp:168-262) restated; the BVH only decides WHICH triangles are tested.
// ---------------------------------------------------------------------------------------
struct TriHit { float t, b0, b1, b2; };
VDEV V3 permute3(V3 v, int kx, int ky, int kz) { return V3{comp(v, kx), comp(v, ky), comp(v, kz)}; }
VDEV float max3abs(float a, float b, float c) { return fmax_(fmax_(__builtin_fabsf(a), __builtin_fabsf(b)), __builtin_fabsf(c)); }
// the part of IntersectTriangle that depends on the ray only (shapes.cpp:180-197: the permutation that makes |d.z| the largest
// component and the shear that aligns d with +z): computed ONCE per ray by the traversals instead of once per triangle test --
// three IEEE divisions (~150 SIMD-cycles) per test otherwise; the same operations on the same operands, so the same bits
struct TriRay {
    int kx, ky, kz;
    float Sx, Sy, Sz;
};
VDEV TriRay tri_ray(V3 d) {
    TriRay R;
    const V3 ad = vabs(d);
    R.kz = (ad.x > ad.y) ? ((ad.x > ad.z) ? 0 : 2) : ((ad.y > ad.z) ? 1 : 2);  // MaxComponentIndex (vecmath.h:454-456)
    R.kx = R.kz + 1; if (R.kx == 3) R.kx = 0;
    R.ky = R.kx + 1; if (R.ky == 3) R.ky = 0;
    const V3 dp = permute3(d, R.kx, R.ky, R.kz);
    R.Sx = -dp.x / dp.z; R.Sy = -dp.y / dp.z; R.Sz = 1 / dp.z;
    return R;
}
// (the degenerate-triangle test of :172-173 runs when the soup is uploaded -- derive_triangle, same expression: such triangles
//  never reach the device array)
VDEV bool tri_intersect(V3 o, const TriRay &R, float tMax, V3 p0, V3 p1, V3 p2, TriHit *hit) {
    V3 p0t = p0 - o, p1t = p1 - o, p2t = p2 - o;
    const int kx = R.kx, ky = R.ky, kz = R.kz;
    p0t = permute3(p0t, kx, ky, kz); p1t = permute3(p1t, kx, ky, kz); p2t = permute3(p2t, kx, ky, kz);
    const float Sx = R.Sx, Sy = R.Sy, Sz = R.Sz;
    p0t.x += Sx * p0t.z; p0t.y += Sy * p0t.z;
    p1t.x += Sx * p1t.z; p1t.y += Sy * p1t.z;
    p2t.x += Sx * p2t.z; p2t.y += Sy * p2t.z;
    float e0 = diff_of_products(p1t.x, p2t.y, p1t.y, p2t.x);
    float e1 = diff_of_products(p2t.x, p0t.y, p2t.y, p0t.x);
    float e2 = diff_of_products(p0t.x, p1t.y, p0t.y, p1t.x);
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {  // double-precision test at triangle edges
        const double p2txp1ty = (double)p2t.x * (double)p1t.y, p2typ1tx = (double)p2t.y * (double)p1t.x;
        e0 = (float)(p2typ1tx - p2txp1ty);
        const double p0txp2ty = (double)p0t.x * (double)p2t.y, p0typ2tx = (double)p0t.y * (double)p2t.x;
        e1 = (float)(p0typ2tx - p0txp2ty);
        const double p1txp0ty = (double)p1t.x * (double)p0t.y, p1typ0tx = (double)p1t.y * (double)p0t.x;
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
    const float det = e0 + e1 + e2;
    if (det == 0) return false;
    p0t.z *= Sz; p1t.z *= Sz; p2t.z *= Sz;
    const float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0 && (tScaled >= 0 || tScaled < tMax * det)) return false;
    else if (det > 0 && (tScaled <= 0 || tScaled > tMax * det)) return false;
    const float invDet = 1 / det;
    const float b0 = e0 * invDet, b1 = e1 * invDet, b2 = e2 * invDet;
    const float t = tScaled * invDet;
    // ensure that the computed t is conservatively greater than zero (shapes.cpp:237-256)
    constexpr float g2 = (2 * kMachineEps) / (1 - 2 * kMachineEps), g3 = (3 * kMachineEps) / (1 - 3 * kMachineEps),
                    g5 = (5 * kMachineEps) / (1 - 5 * kMachineEps);
    const float maxZt = max3abs(p0t.z, p1t.z, p2t.z);
    const float deltaZ = g3 * maxZt;
    const float maxXt = max3abs(p0t.x, p1t.x, p2t.x), maxYt = max3abs(p0t.y, p1t.y, p2t.y);
    const float deltaX = g5 * (maxXt + maxZt), deltaY = g5 * (maxYt + maxZt);
    const float deltaE = 2 * (g2 * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
    const float maxE = max3abs(e0, e1, e2);
    const float deltaT = 3 * (g3 * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * __builtin_fabsf(invDet);
    if (t <= deltaT) return false;
    hit->t = t; hit->b0 = b0; hit->b1 = b1; hit->b2 = b2;
    return true;
}
// slab test for the traversal only: a conservative SUPERSET test (every triangle whose hit distance is <= tLimit lies in a
// box this accepts), so which boxes are visited never changes a result.  *t_near: where the ray enters the box.
VDEV bool bvh_box_hit(float lox, float loy, float loz, float hix, float hiy, float hiz, V3 o, V3 inv, float tLimit, float *t_near) {
    float t0 = 0.f, t1 = tLimit;
    float a = (lox - o.x) * inv.x, b = (hix - o.x) * inv.x;
    t0 = fmax_(t0, fmin_(a, b)); t1 = fmin_(t1, fmax_(a, b) * 1.00001f);
    a = (loy - o.y) * inv.y; b = (hiy - o.y) * inv.y;
    t0 = fmax_(t0, fmin_(a, b)); t1 = fmin_(t1, fmax_(a, b) * 1.00001f);
    a = (loz - o.z) * inv.z; b = (hiz - o.z) * inv.z;
    t0 = fmax_(t0, fmin_(a, b)); t1 = fmin_(t1, fmax_(a, b) * 1.00001f);
    *t_near = t0;
    return !(t0 > t1);  // NaN (0 * inf on a slab boundary) keeps the node
}
// One traversal for both queries.  ANY = false: the closest triangle hit with distance < tMax -- every triangle is tested against
// the RAY's tMax (never against the running closest distance: the test's acceptance would then depend on the visiting order),
// candidates compare by (t, id), boxes beyond the closest hit so far are culled with slack.  ANY = true: whether any triangle is
// hit before tMax.  A single loop: a lane is at an inner node (four box tests, the nearest hit child next, the others stacked) or
// at a leaf (its triangles) -- the "while-while" shape (every lane walks to its next leaf, then the triangle tests run together)
// measured 8-18 % slower on the binary tree of round 2.
template <bool ANY>
VDEV bool bvh_traverse(const DScene &S, V3 o, V3 d, float tMax, int *tri_pos, TriHit *best) {
    const V3 inv = V3{1 / d.x, 1 / d.y, 1 / d.z};
    const TriRay R = tri_ray(d);
    int st_ref[kBvhStack];
    float st_t[ANY ? 1 : kBvhStack];
    int sp = 0;
    bool found = false;
    int best_id = 0x7fffffff;
    float limit = tMax;
    int ref = 0;  // the root
    while (true) {
        int next = kBvhAbsent;
        if (ref >= 0) {
            const float4 *nd = reinterpret_cast<const float4 *>(S.bvh + ref);
            const float4 lox = nd[0], loy = nd[1], loz = nd[2], hix = nd[3], hiy = nd[4], hiz = nd[5];
            const int4 ch = *reinterpret_cast<const int4 *>(nd + 6);
            const float lim = !ANY && found ? limit * 1.00001f : limit;  // cull with slack: the box may touch the triangle exactly where it is hit
            float next_t = kInf;
            const auto child = [&](float bx0, float by0, float bz0, float bx1, float by1, float bz1, int c) {
                float tn;
                if (c != kBvhAbsent && bvh_box_hit(bx0, by0, bz0, bx1, by1, bz1, o, inv, lim, &tn)) {
                    int push = c;
                    float push_t = tn;
                    if (tn < next_t || next == kBvhAbsent) {  // the nearest hit child is visited next, the others wait
                        push = next; push_t = next_t;
                        next = c; next_t = tn;
                    }
                    if (push != kBvhAbsent) {
                        st_ref[sp] = push;
                        if constexpr (!ANY) st_t[sp] = push_t;
                        sp++;
                    }
                }
            };
            child(lox.x, loy.x, loz.x, hix.x, hiy.x, hiz.x, ch.x);
            child(lox.y, loy.y, loz.y, hix.y, hiy.y, hiz.y, ch.y);
            child(lox.z, loy.z, loz.z, hix.z, hiy.z, hiz.z, ch.z);
            child(lox.w, loy.w, loz.w, hix.w, hiy.w, hiz.w, ch.w);
        } else {
            const int leaf = -ref - 1, first = leaf >> 3, cnt = leaf & 7;
            for (int k = 0; k < cnt; ++k) {
                const DTri &T = S.tris[first + k];
                TriHit h;
                if (tri_intersect(o, R, tMax, ld3(T.p0), ld3(T.p1), ld3(T.p2), &h) && h.t < tMax) {
                    if constexpr (ANY) return true;
                    if (!found || h.t < best->t || (h.t == best->t && T.id < best_id)) {
                        found = true;
                        *best = h;
                        best_id = T.id;
                        *tri_pos = first + k;
                        limit = h.t;
                    }
                }
            }
        }
        // what waits: the most recent first; entries beyond the closest hit so far are dropped unvisited
        while (next == kBvhAbsent) {
            if (sp == 0) return found;
            --sp;
            if (ANY || !(st_t[sp] > (found ? limit * 1.00001f : limit))) next = st_ref[sp];
        }
        ref = next;
    }
}
VDEV bool bvh_closest(const DScene &S, V3 o, V3 d, float tMax, int *tri_pos, TriHit *best) { return bvh_traverse<false>(S, o, d, tMax, tri_pos, best); }
VDEV bool bvh_any(const DScene &S, V3 o, V3 d, float tMax) {
    int tp;
    TriHit h;
    return bvh_traverse<true>(S, o, d, tMax, &tp, &h);
}

// ---------------------------------------------------------------------------------------
// Shape "sphere" (round 4).  Interval arithmetic (util/math.h:818-1010 with the CPU branches of the rounding helpers,
// util/float.h:199-297: NextFloatUp / NextFloatDown of the rounded-to-nearest result), Sphere::BasicIntersect
// (shapes.h:147-229), InteractionFromIntersection (:237-284) carried to render space by
// Transform::operator()(SurfaceInteraction) (transform.cpp:229-261).  Same operations as oracle/vspg_oracle.c.
// ---------------------------------------------------------------------------------------
struct Ivl { float lo, hi; };
VDEV Ivl ivl_x(float v) { return Ivl{v, v}; }
VDEV Ivl ivl_mk(float a, float b) { return Ivl{fmin_(a, b), fmax_(a, b)}; }  // Interval(low, high): std::min / std::max
VDEV float ivl_mid(Ivl a) { return (a.lo + a.hi) / 2; }
VDEV Ivl ivl_from_err(float v, float e) { Ivl r; interval_ve(v, e, &r.lo, &r.hi); return r; }
VDEV Ivl ivl_add(Ivl a, Ivl b) { return ivl_mk(next_float_down(a.lo + b.lo), next_float_up(a.hi + b.hi)); }
VDEV Ivl ivl_sub(Ivl a, Ivl b) { return ivl_mk(next_float_down(a.lo + -b.hi), next_float_up(a.hi + -b.lo)); }
VDEV Ivl ivl_mul(Ivl a, Ivl b) {
    const float p0 = a.lo * b.lo, p1 = a.hi * b.lo, p2 = a.lo * b.hi, p3 = a.hi * b.hi;
    const float lo = fmin_(fmin_(fmin_(next_float_down(p0), next_float_down(p1)), next_float_down(p2)), next_float_down(p3));
    const float hi = fmax_(fmax_(fmax_(next_float_up(p0), next_float_up(p1)), next_float_up(p2)), next_float_up(p3));
    return ivl_mk(lo, hi);
}
VDEV bool ivl_has(Ivl i, float v) { return v >= i.lo && v <= i.hi; }
VDEV Ivl ivl_div(Ivl a, Ivl b) {
    if (ivl_has(b, 0.f)) return ivl_mk(-kInf, kInf);
    const float q0 = a.lo / b.lo, q1 = a.hi / b.lo, q2 = a.lo / b.hi, q3 = a.hi / b.hi;
    const float lo = fmin_(fmin_(fmin_(next_float_down(q0), next_float_down(q1)), next_float_down(q2)), next_float_down(q3));
    const float hi = fmax_(fmax_(fmax_(next_float_up(q0), next_float_up(q1)), next_float_up(q2)), next_float_up(q3));
    return ivl_mk(lo, hi);
}
VDEV Ivl ivl_sqr(Ivl i) {
    float alow = __builtin_fabsf(i.lo), ahigh = __builtin_fabsf(i.hi);
    if (alow > ahigh) { const float t = alow; alow = ahigh; ahigh = t; }
    if (ivl_has(i, 0.f)) return ivl_mk(0.f, next_float_up(ahigh * ahigh));
    return ivl_mk(next_float_down(alow * alow), next_float_up(ahigh * ahigh));
}
VDEV Ivl ivl_fmul(float f, Ivl i) {  // operator*(Float, Interval)
    if (f > 0) return ivl_mk(next_float_down(f * i.lo), next_float_up(f * i.hi));
    return ivl_mk(next_float_down(f * i.hi), next_float_up(f * i.lo));
}
VDEV Ivl ivl_sqrt(Ivl i) { return ivl_mk(fmax_(0.f, next_float_down(__builtin_sqrtf(i.lo))), next_float_up(__builtin_sqrtf(i.hi))); }

// Sphere::BasicIntersect for zMin = -radius, zMax = radius, phiMax = fl(2 pi): the clip tests of :182-203 never fire
VDEV bool sphere_intersect(const DSphere &S, V3 ro, V3 rd, float tMax, float *tHit, V3 *pObj) {
    constexpr float g3 = (3 * kMachineEps) / (1 - 3 * kMachineEps);
    Ivl oi[3], di[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // (*objectFromRender)(Point3fi(r.o)), (*objectFromRender)(Vector3fi(r.d)): exact inputs
        const float *row = S.mi + 4 * k;
        const float v = (row[0] * ro.x + row[1] * ro.y) + (row[2] * ro.z + row[3]);
        const float e = g3 * (__builtin_fabsf(row[0] * ro.x) + __builtin_fabsf(row[1] * ro.y) + __builtin_fabsf(row[2] * ro.z) + __builtin_fabsf(row[3]));
        oi[k] = ivl_from_err(v, e);
        const float ev = g3 * (__builtin_fabsf(row[0] * rd.x) + __builtin_fabsf(row[1] * rd.y) + __builtin_fabsf(row[2] * rd.z));
        const float vv = row[0] * rd.x + row[1] * rd.y + row[2] * rd.z;
        di[k] = ivl_from_err(vv, ev);
    }
    const Ivl a = ivl_add(ivl_add(ivl_sqr(di[0]), ivl_sqr(di[1])), ivl_sqr(di[2]));
    const Ivl b = ivl_fmul(2.f, ivl_add(ivl_add(ivl_mul(di[0], oi[0]), ivl_mul(di[1], oi[1])), ivl_mul(di[2], oi[2])));
    const Ivl c = ivl_sub(ivl_add(ivl_add(ivl_sqr(oi[0]), ivl_sqr(oi[1])), ivl_sqr(oi[2])), ivl_sqr(ivl_x(S.radius)));
    const Ivl f = ivl_div(b, ivl_fmul(2.f, a));
    const Ivl v0 = ivl_sub(oi[0], ivl_mul(f, di[0])), v1 = ivl_sub(oi[1], ivl_mul(f, di[1])), v2 = ivl_sub(oi[2], ivl_mul(f, di[2]));
    const Ivl length = ivl_sqrt(ivl_add(ivl_add(ivl_sqr(v0), ivl_sqr(v1)), ivl_sqr(v2)));
    const Ivl discrim = ivl_mul(ivl_mul(ivl_fmul(4.f, a), ivl_add(ivl_x(S.radius), length)), ivl_sub(ivl_x(S.radius), length));
    if (discrim.lo < 0) return false;
    const Ivl rootDiscrim = ivl_sqrt(discrim);
    Ivl q;
    if (ivl_mid(b) < 0) q = ivl_fmul(-.5f, ivl_sub(b, rootDiscrim));
    else q = ivl_fmul(-.5f, ivl_add(b, rootDiscrim));
    Ivl t0 = ivl_div(q, a), t1 = ivl_div(c, q);
    if (t0.lo > t1.lo) { const Ivl t = t0; t0 = t1; t1 = t; }
    if (t0.hi > tMax || t1.lo <= 0) return false;
    Ivl tShapeHit = t0;
    if (tShapeHit.lo <= 0) {
        tShapeHit = t1;
        if (tShapeHit.hi > tMax) return false;
    }
    const float ts = ivl_mid(tShapeHit);
    V3 pHit = V3{ivl_mid(oi[0]) + ts * ivl_mid(di[0]), ivl_mid(oi[1]) + ts * ivl_mid(di[1]), ivl_mid(oi[2]) + ts * ivl_mid(di[2])};
    const float sc = S.radius / len(pHit);  // pHit *= radius / Distance(pHit, Point3f(0, 0, 0))
    pHit = V3{pHit.x * sc, pHit.y * sc, pHit.z * sc};
    if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * S.radius;
    *tHit = ts;
    *pObj = pHit;
    return true;
}
// pi (render space, with error bounds), the geometric normal and Normalize(shading.dpdu) of the hit at object-space pHit
struct SphereSurf {
    P3i pi;
    V3 n, dpdu_n;
};
template <bool WITH_DPDU = true>
VDEV SphereSurf sphere_interaction(const DSphere &S, V3 pHit) {
    SphereSurf r;
    const float cosTheta = pHit.z / S.radius;
    const float zRadius = __builtin_sqrtf(sqr(pHit.x) + sqr(pHit.y));
    const float cosPhi = pHit.x / zRadius, sinPhi = pHit.y / zRadius;
    const V3 dpdu = V3{-S.phiMax * pHit.y, S.phiMax * pHit.x, 0.f};
    const float sinTheta = safe_sqrt(1 - sqr(cosTheta));
    const float dth = S.thetaZMax - S.thetaZMin;
    const V3 dpdv = V3{dth * (pHit.z * cosPhi), dth * (pHit.z * sinPhi), dth * (-S.radius * sinTheta)};
    V3 no = normalize(cross(dpdu, dpdv));
    if (S.flip) no = -no;
    constexpr float g5 = (5 * kMachineEps) / (1 - 5 * kMachineEps), g3 = (3 * kMachineEps) / (1 - 3 * kMachineEps);
    const Ivl px = ivl_from_err(pHit.x, g5 * __builtin_fabsf(pHit.x)), py = ivl_from_err(pHit.y, g5 * __builtin_fabsf(pHit.y)),
              pz = ivl_from_err(pHit.z, g5 * __builtin_fabsf(pHit.z));
    const float x = ivl_mid(px), y = ivl_mid(py), z = ivl_mid(pz);
    const float ex = (px.hi - px.lo) / 2, ey = (py.hi - py.lo) / 2, ez = (pz.hi - pz.lo) / 2;
    const bool exact = (px.hi - px.lo) == 0 && (py.hi - py.lo) == 0 && (pz.hi - pz.lo) == 0;
    float pv[3], pe[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // Transform::operator()(Point3fi) (transform.h:133-175)
        const float *row = S.m + 4 * k;
        pv[k] = (row[0] * x + row[1] * y) + (row[2] * z + row[3]);
        const float ea = g3 * (__builtin_fabsf(row[0] * x) + __builtin_fabsf(row[1] * y) + __builtin_fabsf(row[2] * z) + __builtin_fabsf(row[3]));
        pe[k] = exact ? ea : (g3 + 1) * (__builtin_fabsf(row[0]) * ex + __builtin_fabsf(row[1]) * ey + __builtin_fabsf(row[2]) * ez) + ea;
    }
    r.pi = p3i_from_err(V3{pv[0], pv[1], pv[2]}, V3{pe[0], pe[1], pe[2]});
    const float *mi = S.mi;  // normals: the inverse transpose (transform.h:329-334), then Normalize
    r.n = normalize(V3{mi[0] * no.x + mi[4] * no.y + mi[8] * no.z, mi[1] * no.x + mi[5] * no.y + mi[9] * no.z, mi[2] * no.x + mi[6] * no.y + mi[10] * no.z});
    if (WITH_DPDU) {
        const float *m = S.m;
        r.dpdu_n = normalize(V3{m[0] * dpdu.x + m[1] * dpdu.y + m[2] * dpdu.z, m[4] * dpdu.x + m[5] * dpdu.y + m[6] * dpdu.z, m[8] * dpdu.x + m[9] * dpdu.y + m[10] * dpdu.z});
    } else {
        r.dpdu_n = mk(0, 0, 0);
    }
    return r;
}

// FULL = false: the scene is known to hold rectangles only (the workgroup kernel's instantiations, see HomogeneousMediumT::kSimpleScene)
template <bool FULL = true>
VLEAF Isect scene_intersect(const DScene &S, V3 o, V3 d, float tMax) {
    Isect best;
    best.hit = false;
    best.t = tMax;
    best.quad = 0;
    float bu = 0, bv = 0;
    for (int i = 0; i < S.n_quads; ++i) {
        float t, u, v;
        if (rect_hit_uv(S.irec[i], o, d, best.t, &t, &u, &v)) {
            best.hit = true;
            best.t = t;
            best.quad = i;
            bu = u;
            bv = v;
        }
    }
    const DQuad &q = quad_at(best.quad);  // per-lane index: LDS copy
    best.p = quad_point(q, bu, bv);
    best.n = ld3(q.n);
    if (FULL && S.n_tris > 0) {  // wave-uniform: rectangles win ties (they are tested first, strictly closer)
        int tp = 0;
        TriHit h;
        if (bvh_closest(S, o, d, best.t, &tp, &h)) {
            const DTri &T = S.tris[tp];
            best.hit = true;
            best.t = h.t;
            best.quad = -2 - tp;
            best.p = V3{h.b0, h.b1, h.b2};  // the barycentrics name the hit (tri_pi)
            best.n = V3{T.nx, T.ny, T.nz};
        }
    }
    if (FULL && S.n_spheres > 0) {  // wave-uniform.  Every sphere against the RAY's tMax (BasicIntersect rejects on interval bounds
        // of t against the tMax it is given), the closest by t; a sphere hit's `p` is the object-space point (kSpherePrim)
        for (int i = 0; i < S.n_spheres; ++i) {
            float t;
            V3 pObj;
            if (sphere_intersect(S.spheres[i], o, d, tMax, &t, &pObj) && t < best.t) {
                best.hit = true;
                best.t = t;
                best.quad = kSpherePrim + i;
                best.p = pObj;
                best.n = sphere_interaction<false>(S.spheres[i], pObj).n;
            }
        }
    }
    return best;
}
template <bool FULL = true>
VLEAF bool scene_intersect_any(const DScene &S, V3 o, V3 d, float tMax) {
    bool any = false;
    for (int i = 0; i < S.n_quads; ++i) {
        float t, u, v;
        any = any || rect_hit_uv(S.irec[i], o, d, tMax, &t, &u, &v);
    }
    if (FULL && S.n_tris > 0 && !any) any = bvh_any(S, o, d, tMax);
    if (FULL && S.n_spheres > 0 && !any) {
        for (int i = 0; i < S.n_spheres; ++i) {
            float t;
            V3 pObj;
            any = any || sphere_intersect(S.spheres[i], o, d, tMax, &t, &pObj);
        }
    }
    return any;
}
// ---- medium boundaries -----------------------------------------------------------------------------------------------
// SURF_* flags of the surface a hit names (rectangle: LDS copy; triangle / sphere: the scene arrays)
VDEV int surf_flags(const DScene &S, int prim) {
    if (is_tri(prim)) return S.tris[tri_of(prim)].flags;
    if (is_sphere(prim)) return S.spheres[sphere_of(prim)].flags;
    return quad_at(prim).flags;
}
// Triangle::InteractionFromIntersection (shapes.h:922-930): pHit = b0 p0 + b1 p1 + b2 p2, pError = gamma(7) (|b0 p0| + |b1 p1| + |b2 p2|)
VDEV P3i tri_pi(const DTri &T, V3 b) {
    const V3 p0 = ld3(T.p0), p1 = ld3(T.p1), p2 = ld3(T.p2);
    const V3 p = p0 * b.x + p1 * b.y + p2 * b.z;
    const V3 s = vabs(p0 * b.x) + vabs(p1 * b.y) + vabs(p2 * b.z);
    constexpr float g7 = (7 * kMachineEps) / (1 - 7 * kMachineEps);
    return p3i_from_err(p, V3{g7 * s.x, g7 * s.y, g7 * s.z});
}
// SurfaceInteraction::pi of the hit (prim, p)
template <bool FULL = true>
VDEV P3i surf_pi(const DScene &S, int prim, V3 p) {
    if (FULL && is_sphere(prim)) return sphere_interaction<false>(S.spheres[sphere_of(prim)], p).pi;
    if (FULL && is_tri(prim)) return tri_pi(S.tris[tri_of(prim)], p);
    return p3i_from_err(p, ld3(quad_at(prim).perr));
}
// Dot(Vector3f, Normal3f) (vecmath.h:1064-1068): FMA(n.x, v.x, SumOfProducts(n.y, v.y, n.z, v.z)) (math.h:577-583)
VDEV float dot_vn(V3 v, V3 n) {
    const float cd = n.z * v.z;
    const float sop = __builtin_fmaf(n.y, v.y, cd);
    const float err = __builtin_fmaf(n.z, v.z, -cd);
    return __builtin_fmaf(n.x, v.x, sop + err);
}
// Interaction::GetMedium(w) (interaction.h:117-121) with one medium: true = the scene's medium.  `medium`: the
// interaction's own member (the medium of the ray that hit a non-transition surface / of a medium interaction).
VDEV bool get_medium(int flags, V3 n, V3 w, bool medium) {
    const int iface = flags >> SURF_IFACE_SHIFT;
    if (iface) return dot_vn(w, n) > 0 ? (iface & VSPG_IFACE_OUTSIDE) != 0 : (iface & VSPG_IFACE_INSIDE) != 0;
    return medium;
}

// ---------------------------------------------------------------------------------------
// a5: media.  MediumProperties (media.h:77-82) and majorant iterators.
// ---------------------------------------------------------------------------------------
struct MediumProps {
    Spec sigma_a, sigma_s, Le;
    float g;
    Spec sigma_t;  // sigma_s + sigma_a
    // temperature grids (media.h:333-341, :724-735): Le = bb_scale * BlackbodySpectrum(bb_temp).Sample(lambda), finished by
    // medium_Le() where the path reads mp.Le (the delta-tracking callback is the only reader); bb_temp = 0: Le above is the value
    float bb_temp = 0.f, bb_scale = 0.f;
};
struct MajSeg {
    float tMin, tMax;
    Spec sigma_maj;
};
// HomogeneousMedium (media.h:221-283): one segment [0,tMax], sigma_maj = sigma_a + sigma_s
// GREY = true is chosen by the host when sigma_a, sigma_s and Le are bitwise grey (fog): every medium
// spectrum is then built as {x, x, x} from ONE loaded value, so the compiler sees the three channels of
// sigma_maj, T_maj, sigma_n, r_u, r_l ... as the same value and all per-channel arithmetic on them -- IEEE
// divisions and FastExp included -- collapses to one channel by common-subexpression elimination, without a
// different spectrum type.  Same operations on the same bits per channel: results unchanged.
// GREY = 2 additionally promises that every rectangle's Kd is bitwise grey: the BSDF's reflectance is built
// the same way and the path throughput beta (only ever multiplied by grey medium terms, grey BSDF weights and
// scalars) is grey by construction too.
// NULLZERO promises (the host checked the uploaded floats) that the null-collision coefficient
// ClampZero((sigma_t - sigma_a) - sigma_s) is exactly 0 in every channel -- the usual outcome of the float
// arithmetic for a homogeneous medium: the ratio-tracking estimate of a shadow ray is then 0 at its first
// tentative collision (sample_Ld), a fact the compiler cannot see in the runtime constants.
// SIMPLE promises (the host routes accordingly) a scene of rectangles and area lights only -- no triangle soup, no infinite
// lights: the code for those (BVH traversal, per-hit error bounds, the escaped-ray light loop, DistantLight sampling) is
// compiled out of the instantiation, which keeps the benchmark kernel at its round-1 register shape.
template <int GREY, bool NULLZERO = false, bool SIMPLE = false>
struct HomogeneousMediumT {
    static constexpr int kGrey = GREY;   // 0 none, 1 medium spectra, 2 medium spectra + surface reflectances
    static constexpr bool kNullZero = NULLZERO;
    static constexpr bool kSimpleScene = SIMPLE;
    static constexpr int kBnd = -1;  // medium boundaries: decided at run time (has_bnd; simple-scene kernels never ask)
    // constants live in the workgroup's LDS copy (s_scene_medium, staged by stage_scene_lds)
    static VDEV Spec m3(int o) {
        if constexpr (GREY) {
            const float x = s_scene_medium[o];
            return Spec{x, x, x};
        }
        return Spec{s_scene_medium[o], s_scene_medium[o + 1], s_scene_medium[o + 2]};
    }
    struct Iter {  // HomogeneousMajorantIterator (media.h:84-106)
        MajSeg seg;
        bool called;
        VDEV bool next(MajSeg *s) {
            if (called) return false;
            called = true;
            *s = seg;
            return true;
        }
    };
    VDEV Iter sample_ray(V3, V3, float tMax) const { return Iter{MajSeg{0, tMax, m3(MED_SIGMA_T)}, false}; }  // sigma_a + sigma_s
    VDEV MediumProps sample_point(V3) const {
        return MediumProps{m3(MED_SIGMA_A), m3(MED_SIGMA_S), m3(MED_LE), s_scene_medium[MED_G], m3(MED_SIGMA_T)};
    }
    // null-collision coefficient ClampZero(sigma_maj - sigma_a - sigma_s) for this medium's own majorant
    VDEV Spec sigma_n(const MediumProps &, Spec) const { return clamp_zero(m3(MED_SIGMA_N_RAW)); }
    VDEV bool is_homogeneous() const { return true; }
    static constexpr bool kSingleSegment = true;
    // sigma_t = sigma_s + sigma_a at the collision and sigma_maj = sigma_a + sigma_s of the segment are
    // the same float sum, so pScatter = sigma_t/sigma_maj == 1 and pNull == 0 exactly:
    // SampleDiscrete({1, 0}, u) returns 0 for every u in [0, 1) -- the first collision of the
    // distance-sampling walk is always a real one (no null collisions in a homogeneous medium).
    static constexpr bool kAlwaysRealCollision = true;
    static constexpr bool kEmit = false;  // (no temperature grid)
};
using HomogeneousMedium = HomogeneousMediumT<0>;
using HomogeneousMediumSimple = HomogeneousMediumT<0, false, true>;
using HomogeneousMediumGrey = HomogeneousMediumT<1, false, true>;
using HomogeneousMediumGreyScene = HomogeneousMediumT<2, false, true>;
using HomogeneousMediumGreySceneNullZero = HomogeneousMediumT<2, true, true>;

// ---------------------------------------------------------------------------------------
// a6: GridMedium (media.h:284-390) with the 3-D DDA majorant iterator (media.h:140-218),
// identity renderFromMedium.  Per-axis DDA state is kept in named registers and selected with
// compares (runtime-indexed private arrays would live in scratch).
// ---------------------------------------------------------------------------------------
constexpr int kMajRes = 16;      // GridMedium majorant grid (media.cpp:252)
constexpr int kMajResNvdb = 64;  // NanoVDBMedium majorant grid (media.cpp:574)
VDEV float sel3(float a0, float a1, float a2, int axis) { return axis == 0 ? a0 : (axis == 1 ? a1 : a2); }
VDEV int sel3i(int a0, int a1, int a2, int axis) { return axis == 0 ? a0 : (axis == 1 ? a1 : a2); }
// NVDB = false: GridMedium ("uniformgrid").  NVDB = true: NanoVDBMedium semantics over a dense copy of the
// grid (media.h:686-719): index-space trilinear sampling a + w (b - a) with background 0 outside the index
// bounding box, "densityoffset", 64^3 majorants read from HBM / L2 (1 MB: too large for LDS).  The sparse
// NanoVDB tree and its sampler are absent from the reference tree: parity unpinned for the fetch itself.
#ifndef VSPG_NVDB_MAJ_GLOBAL
#define VSPG_NVDB_MAJ_GLOBAL 1   // (0: the 64^3 majorants through the generic pointer as well, for A/B builds)
#endif
// MAJLDS: `majorant` points into the block's LDS (the pipeline kernels' copy of a 16^3 grid, wf_block_medium): the iterator reads it
// with LDS instructions.  Through the generic pointer the member is, the compiler emits a flat load per majorant cell -- issued to
// the LDS and the vector-memory path alike, waited for on both counters (seen in the ISA of the walk kernels, round 5).
template <bool NVDB, bool GREY = false, int BND = -1, bool EMIT = false, bool MAJLDS = false>
struct GridMediumT {
    static constexpr bool kMajLds = MAJLDS;
    static constexpr int kBnd = BND;  // medium boundaries known at compile time (0 / 1) or decided at run time (-1): has_bnd
    // blackbody emission of a temperature grid compiled in: the per-lane kernels (BND = -1: whatever the scene holds) and the NDS
    // pipeline's emissive instantiations.  Elsewhere the code is left out: in the NDS kernel it costs 16-176 B of scratch per lane
    // whether or not the scene has a temperature grid (the resampling kernels never read Le and are the same either way).
    static constexpr bool kEmit = EMIT || BND < 0;
    static constexpr int kRes = NVDB ? kMajResNvdb : kMajRes;
    // wavefront walk kernels: majorant-cell advances tried per tracking step before the collision code runs (a 64^3
    // majorant grid has 4x as many cell crossings per tentative collision as the 16^3 one)
    static constexpr int kAdvanceRounds = NVDB ? 6 : 3;
    static constexpr int kGrey = GREY ? 1 : 0;  // sigma_a, sigma_s built from one value each (see HomogeneousMediumT)
    static constexpr bool kSimpleScene = false;
    Spec sigma_a, sigma_s;
    float g;
    int nx, ny, nz;
    V3 bmin, bmax;
    const int32_t *brick_index;  // octet bricks (DScene)
    const float4 *octets;
    int bnx, bny;
    const float *majorant;  // HBM or the block's LDS copy
    int imx, imy, imz;      // NVDB: index bbox min
    V3 inv_voxel, origin;   // NVDB: worldToIndexF(p) = (p - origin) * inv_voxel
    float density_offset;   // NVDB
    const float *le_scale;  // emissive GridMedium: LeScale grid (null = not emissive), its size, Le_spec
    int lnx, lny, lnz;
    Spec Le;
    const float *minv;      // rows 0..2 of mInv when renderFromMedium is not the identity, else null (wave-uniform)
    const float *temperature;  // temperature grid (raw, nx ny nz; null = none): blackbody emission
    float t_offset, t_scale, t_lescale;  // t_lescale: NanoVDBMedium's LeScale (a GridMedium scales by its LeScale grid)

    // Transform::ApplyInverse(Point3<T>) (util/transform.h:387-398), affine matrix
    VDEV V3 to_medium(V3 p) const {
        if (!minv) return p;
        return V3{(minv[0] * p.x + minv[1] * p.y) + (minv[2] * p.z + minv[3]), (minv[4] * p.x + minv[5] * p.y) + (minv[6] * p.z + minv[7]),
                  (minv[8] * p.x + minv[9] * p.y) + (minv[10] * p.z + minv[11])};
    }

    struct Iter {  // DDAMajorantIterator
        Spec sigma_t;
        float tMin, tMax;
        const float *maj;
        float ncx, ncy, ncz;  // nextCrossingT
        float dtx, dty, dtz;  // deltaT
        int vx, vy, vz;       // voxel
        int neg;              // bit a set: step[a] == -1 (voxelLimit -1), else +1 (voxelLimit res)
        VDEV bool next(MajSeg *s) {  // media.h:178-207
            // `!(tMin < tMax)` == the reference's `tMin >= tMax` for every ordered pair; a NaN ray (the reference would walk
            // its majorant grid out of bounds) ends the iteration instead
            if (!(tMin < tMax)) return false;
            int bits = ((ncx < ncy) << 2) + ((ncx < ncz) << 1) + ((ncy < ncz));
            int stepAxis = (0xA66 >> (2 * bits)) & 3;  // cmpToAxis[8] = {2,1,2,1,2,2,0,0}
            float nc = sel3(ncx, ncy, ncz, stepAxis);
            float tVoxelExit = fmin_(tMax, nc);
            VSPG_DBG_CHECK((unsigned)vx < (unsigned)kRes && (unsigned)vy < (unsigned)kRes && (unsigned)vz < (unsigned)kRes, 2);
#ifdef VSPG_WF_DEBUG
            if (!((unsigned)vx < (unsigned)kRes && (unsigned)vy < (unsigned)kRes && (unsigned)vz < (unsigned)kRes)) {
                g_dbg_err[2] = (unsigned)vx; g_dbg_err[3] = (unsigned)vy; g_dbg_err[4] = (unsigned)vz; g_dbg_err[5] = (unsigned)neg;
                g_dbg_err[6] = __float_as_uint(tMin); g_dbg_err[7] = __float_as_uint(tMax);
                tMin = tMax;
                return false;
            }
#endif
            float md;
            if constexpr (MAJLDS) md = ((const float __attribute__((address_space(3))) *)maj)[vx + kRes * (vy + kRes * vz)];
            else if constexpr (kRes != kMajRes && VSPG_NVDB_MAJ_GLOBAL) md = *((const float VSPG_GLOBAL_AS *)maj + (vx + kRes * (vy + kRes * vz)));  // 64^3: never staged in LDS
            else md = maj[vx + kRes * (vy + kRes * vz)];
            s->tMin = tMin;
            s->tMax = tVoxelExit;
            s->sigma_maj = sigma_t * md;
            tMin = tVoxelExit;
            if (nc > tMax) tMin = tMax;
            int st = ((neg >> stepAxis) & 1) ? -1 : 1;
            int lim = ((neg >> stepAxis) & 1) ? -1 : kRes;
            int v = sel3i(vx, vy, vz, stepAxis) + st;
            if (v == lim) tMin = tMax;
            float ncn = nc + sel3(dtx, dty, dtz, stepAxis);
            if (stepAxis == 0) { vx = v; ncx = ncn; }
            else if (stepAxis == 1) { vy = v; ncy = ncn; }
            else { vz = v; ncz = ncn; }
            return true;
        }
    };

    VDEV V3 offset(V3 p) const {  // Bounds3::Offset (vecmath.h:1323-1332)
        V3 o = p - bmin;
        if (bmax.x > bmin.x) o.x /= bmax.x - bmin.x;
        if (bmax.y > bmin.y) o.y /= bmax.y - bmin.y;
        if (bmax.z > bmin.z) o.z /= bmax.z - bmin.z;
        return o;
    }
    // the octet of base voxel (ix, iy, iz): its 2x2x2 corner values (zeros outside the grid / in an empty brick)
    struct Octet { float v000, v100, v010, v110, v001, v101, v011, v111; };
    VDEV Octet octet(int ix, int iy, int iz) const {
        const int ox = ix + 1, oy = iy + 1, oz = iz + 1;
        Octet o{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if ((unsigned)ox <= (unsigned)nx && (unsigned)oy <= (unsigned)ny && (unsigned)oz <= (unsigned)nz) {
            // brick_index == nullptr: every brick is stored, in grid order (round 3, "dense bricks") -- the slot is arithmetic and
            // a density query is ONE memory round trip instead of two dependent ones (index, then octet).  Both arrays are read
            // through global-address-space pointers (flat loads wait on two counters and are issued where they stand).
            const int cell = ((oz >> 3) * bny + (oy >> 3)) * bnx + (ox >> 3);
            int b = cell;
            if (brick_index) b = *(const int VSPG_GLOBAL_AS *)(brick_index + cell);
            VSPG_DBG_CHECK(b < 1 << 20 || !brick_index, 1);
#ifdef VSPG_WF_DEBUG
            if (brick_index && b >= 1 << 20) return o;
#endif
            if (b >= 0) {
                typedef float v4f_ __attribute__((ext_vector_type(4)));
                const v4f_ VSPG_GLOBAL_AS *q = (const v4f_ VSPG_GLOBAL_AS *)(octets + ((size_t)b * 512u + (size_t)((ox & 7) + 8 * ((oy & 7) + 8 * (oz & 7)))) * 2u);
                const v4f_ lo = q[0], hi = q[1];
                o = Octet{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            }
        }
        return o;
    }
    VDEV float lookup(V3 p) const {  // SampledGrid::Lookup(Point3f) (containers.h:804-819) over the octet of floor(p * n - .5)
        float sx = p.x * nx - .5f, sy = p.y * ny - .5f, sz = p.z * nz - .5f;
        float fx = __builtin_floorf(sx), fy = __builtin_floorf(sy), fz = __builtin_floorf(sz);
        int ix = (int)fx, iy = (int)fy, iz = (int)fz;
        float dx = sx - (float)ix, dy = sy - (float)iy, dz = sz - (float)iz;
        const Octet o = octet(ix, iy, iz);
        float d00 = (1 - dx) * o.v000 + dx * o.v100;
        float d10 = (1 - dx) * o.v010 + dx * o.v110;
        float d01 = (1 - dx) * o.v001 + dx * o.v101;
        float d11 = (1 - dx) * o.v011 + dx * o.v111;
        float a = (1 - dy) * d00 + dy * d10, b = (1 - dy) * d01 + dy * d11;
        return (1 - dz) * a + dz * b;
    }
    static VDEV float at(const float *data, int nx, int ny, int nz, int x, int y, int z) {  // SampledGrid::Lookup(Point3i) (containers.h:830-835)
        if (x < 0 || y < 0 || z < 0 || x >= nx || y >= ny || z >= nz) return 0.f;
        return data[((size_t)z * ny + y) * nx + x];
    }
    static VDEV float lookup(const float *data, int nx, int ny, int nz, V3 p) {  // SampledGrid::Lookup(Point3f) (containers.h:804-819)
        auto at = [&](int x, int y, int z) { return GridMediumT::at(data, nx, ny, nz, x, y, z); };
        float sx = p.x * nx - .5f, sy = p.y * ny - .5f, sz = p.z * nz - .5f;
        float fx = __builtin_floorf(sx), fy = __builtin_floorf(sy), fz = __builtin_floorf(sz);
        int ix = (int)fx, iy = (int)fy, iz = (int)fz;
        float dx = sx - (float)ix, dy = sy - (float)iy, dz = sz - (float)iz;
        // issue the 8 voxel loads together, then interpolate
        float v000 = at(ix, iy, iz), v100 = at(ix + 1, iy, iz), v010 = at(ix, iy + 1, iz), v110 = at(ix + 1, iy + 1, iz);
        float v001 = at(ix, iy, iz + 1), v101 = at(ix + 1, iy, iz + 1), v011 = at(ix, iy + 1, iz + 1), v111 = at(ix + 1, iy + 1, iz + 1);
        float d00 = (1 - dx) * v000 + dx * v100;
        float d10 = (1 - dx) * v010 + dx * v110;
        float d01 = (1 - dx) * v001 + dx * v101;
        float d11 = (1 - dx) * v011 + dx * v111;
        float a = (1 - dy) * d00 + dy * d10, b = (1 - dy) * d01 + dy * d11;
        return (1 - dz) * a + dz * b;
    }
    VDEV Iter empty_iter() const {  // default-constructed iterator: next() yields nothing
        Iter it;
        it.sigma_t = sigma_a + sigma_s;
        it.tMin = kInf;
        it.tMax = -kInf;
        it.maj = majorant;
        it.ncx = it.ncy = it.ncz = 0;
        it.dtx = it.dty = it.dtz = 0;
        it.vx = it.vy = it.vz = 0;
        it.neg = 0;
        return it;
    }
    VDEV Iter sample_ray(V3 o, V3 d, float raytMax) const {  // media.h:347-362
        Iter it = empty_iter();
        // ray = renderFromMedium.ApplyInverse(ray, &tMax) (transform.h:416-429, transform.cpp:263-303): the origin carries the
        // error bound gamma(3) * sum |mInv[i][k] * o[k]| and is pushed along d by dt; from here on (o, d)
        // is the medium-space ray -- its t parametrises the render-space ray too (affine map)
        const float g3 = (3 * kMachineEps) / (1 - 3 * kMachineEps);
        V3 oerr;
        if (minv) {
            const float x = o.x, y = o.y, z = o.z;
            oerr = V3{g3 * (__builtin_fabsf(minv[0] * x) + __builtin_fabsf(minv[1] * y) + __builtin_fabsf(minv[2] * z)),
                      g3 * (__builtin_fabsf(minv[4] * x) + __builtin_fabsf(minv[5] * y) + __builtin_fabsf(minv[6] * z)),
                      g3 * (__builtin_fabsf(minv[8] * x) + __builtin_fabsf(minv[9] * y) + __builtin_fabsf(minv[10] * z))};
            o = to_medium(o);
            d = V3{minv[0] * d.x + minv[1] * d.y + minv[2] * d.z, minv[4] * d.x + minv[5] * d.y + minv[6] * d.z,
                   minv[8] * d.x + minv[9] * d.y + minv[10] * d.z};
        } else {
            oerr = V3{g3 * (__builtin_fabsf(o.x) + 0.f + 0.f), g3 * (0.f + __builtin_fabsf(o.y) + 0.f), g3 * (0.f + 0.f + __builtin_fabsf(o.z))};
        }
        P3i oi = p3i_from_err(o, oerr);
        float lengthSquared = len2(d);
        if (lengthSquared > 0) {
            V3 oe = oi.err();
            float dt = dot(vabs(d), oe) / lengthSquared;
            V3 sh = d * dt;
            oi.lo = V3{next_float_down(oi.lo.x + sh.x), next_float_down(oi.lo.y + sh.y), next_float_down(oi.lo.z + sh.z)};
            oi.hi = V3{next_float_up(oi.hi.x + sh.x), next_float_up(oi.hi.y + sh.y), next_float_up(oi.hi.z + sh.z)};
            raytMax -= dt;
        }
        V3 ro = oi.mid();
        // Bounds3::IntersectP (vecmath.h:1547-1571)
        float t0 = 0, t1 = raytMax;
        {
            const float k = 1 + 2 * g3;
            float inv = 1 / d.x, tn = (bmin.x - ro.x) * inv, tf = (bmax.x - ro.x) * inv;
            if (tn > tf) { float t = tn; tn = tf; tf = t; }
            tf *= k; t0 = tn > t0 ? tn : t0; t1 = tf < t1 ? tf : t1;
            if (t0 > t1) return it;
            inv = 1 / d.y; tn = (bmin.y - ro.y) * inv; tf = (bmax.y - ro.y) * inv;
            if (tn > tf) { float t = tn; tn = tf; tf = t; }
            tf *= k; t0 = tn > t0 ? tn : t0; t1 = tf < t1 ? tf : t1;
            if (t0 > t1) return it;
            inv = 1 / d.z; tn = (bmin.z - ro.z) * inv; tf = (bmax.z - ro.z) * inv;
            if (tn > tf) { float t = tn; tn = tf; tf = t; }
            tf *= k; t0 = tn > t0 ? tn : t0; t1 = tf < t1 ? tf : t1;
            if (t0 > t1) return it;
        }
        // DDAMajorantIterator ctor (media.h:145-176)
        it.tMin = t0;
        it.tMax = t1;
        V3 diag = bmax - bmin;
        V3 go = offset(ro);
        V3 gd = V3{d.x / diag.x, d.y / diag.y, d.z / diag.z};
        V3 gi = go + gd * t0;
        auto axis_setup = [&](float gia, float gda, int bit, int *voxel, float *deltaT, float *nextT) {
            float v = gia * kRes;
            *voxel = (int)(v < 0 ? 0.f : (v > (float)(kRes - 1) ? (float)(kRes - 1) : v));
            *deltaT = 1 / (__builtin_fabsf(gda) * kRes);
            if (gda == -0.f) gda = 0.f;
            if (gda >= 0) {
                float nextVoxelPos = (float)(*voxel + 1) / kRes;
                *nextT = t0 + (nextVoxelPos - gia) / gda;
            } else {
                float nextVoxelPos = (float)(*voxel) / kRes;
                *nextT = t0 + (nextVoxelPos - gia) / gda;
                it.neg |= bit;
            }
        };
        axis_setup(gi.x, gd.x, 1, &it.vx, &it.dtx, &it.ncx);
        axis_setup(gi.y, gd.y, 2, &it.vy, &it.dty, &it.ncy);
        axis_setup(gi.z, gd.z, 4, &it.vz, &it.dtz, &it.ncz);
        return it;
    }
    // nanovdb::SampleFromVoxels<Tree, 1, false> over the dense copy: ijk = floor(x), uvw = x - ijk, corner values
    // with background 0, lerp(a, b, w) = a + w (b - a) along z, then y, then x
    VDEV float lookup_index(V3 x) const {
        const float fx = __builtin_floorf(x.x), fy = __builtin_floorf(x.y), fz = __builtin_floorf(x.z);
        // a query far outside the grid must not wrap the int conversion into the valid range: clamp (any such octet is zero)
        const float cx = fmax_(fmin_(fx, 1e9f), -1e9f), cy = fmax_(fmin_(fy, 1e9f), -1e9f), cz = fmax_(fmin_(fz, 1e9f), -1e9f);
        const int i = (int)cx, j = (int)cy, k = (int)cz;
        const float u = x.x - fx, v = x.y - fy, w = x.z - fz;
        const Octet o = octet(i - imx, j - imy, k - imz);  // accessor.getValue: background 0 outside the index bounding box
        const float v000 = o.v000, v001 = o.v001, v010 = o.v010, v011 = o.v011, v100 = o.v100, v101 = o.v101, v110 = o.v110, v111 = o.v111;
        const float a00 = v000 + w * (v001 - v000), a01 = v010 + w * (v011 - v010);
        const float a10 = v100 + w * (v101 - v100), a11 = v110 + w * (v111 - v110);
        const float b0 = a00 + v * (a01 - a00), b1 = a10 + v * (a11 - a10);
        return b0 + u * (b1 - b0);
    }
    // the temperature grid through nanovdb's sampler (see lookup_index; same index bounding box and index-to-world map as the density grid)
    VDEV float lookup_index_raw(const float *data, V3 x) const {
        const float fx = __builtin_floorf(x.x), fy = __builtin_floorf(x.y), fz = __builtin_floorf(x.z);
        const float cx = fmax_(fmin_(fx, 1e9f), -1e9f), cy = fmax_(fmin_(fy, 1e9f), -1e9f), cz = fmax_(fmin_(fz, 1e9f), -1e9f);
        const int i = (int)cx - imx, j = (int)cy - imy, k = (int)cz - imz;
        const float u = x.x - fx, v = x.y - fy, w = x.z - fz;
        auto at = [&](int a, int b, int c) { return GridMediumT::at(data, nx, ny, nz, a, b, c); };
        const float v000 = at(i, j, k), v001 = at(i, j, k + 1), v010 = at(i, j + 1, k), v011 = at(i, j + 1, k + 1);
        const float v100 = at(i + 1, j, k), v101 = at(i + 1, j, k + 1), v110 = at(i + 1, j + 1, k), v111 = at(i + 1, j + 1, k + 1);
        const float a00 = v000 + w * (v001 - v000), a01 = v010 + w * (v011 - v010);
        const float a10 = v100 + w * (v101 - v100), a11 = v110 + w * (v111 - v110);
        const float b0 = a00 + v * (a01 - a00), b1 = a10 + v * (a11 - a10);
        return b0 + u * (b1 - b0);
    }
    VDEV MediumProps sample_point(V3 p) const {  // media.h:316-345 / :686-703
        float d;
        Spec le = sp(0.f);
        float bb_temp = 0.f, bb_scale = 0.f;
        p = to_medium(p);  // renderFromMedium.ApplyInverse(p) (media.h:322 / :693)
        if constexpr (NVDB) {
            const V3 xi = V3{(p.x - origin.x) * inv_voxel.x, (p.y - origin.y) * inv_voxel.y, (p.z - origin.z) * inv_voxel.z};
            d = lookup_index(xi);
            d += density_offset;
            if (kEmit && temperature) {  // NanoVDBMedium::Le (media.h:724-735); wave-uniform
                float temp = lookup_index_raw(temperature, xi);
                temp = (temp - t_offset) * t_scale;
                if (temp > 100.f) { bb_temp = temp; bb_scale = t_lescale; }
            }
        } else {
            const V3 po = offset(p);
            d = lookup(po);
            if (le_scale) {  // isEmissive (wave-uniform): Le = scale * Le_spec where the LeScale grid is positive (:326-342)
                const float scale = lookup(le_scale, lnx, lny, lnz, po);
                if (scale > 0) {
                    if (kEmit && temperature) {  // media.h:333-341
                        float temp = lookup(temperature, nx, ny, nz, po);
                        temp = (temp - t_offset) * t_scale;
                        if (temp > 100.f) { bb_temp = temp; bb_scale = scale; }
                    } else
                        le = Le * scale;
                }
            }
        }
        Spec sa = sigma_a * d, ss = sigma_s * d;
        return MediumProps{sa, ss, le, g, ss + sa, bb_temp, bb_scale};
    }
    VDEV Spec sigma_n(const MediumProps &mp, Spec sigma_maj) const { return clamp_zero(sigma_maj - mp.sigma_a - mp.sigma_s); }
    VDEV bool is_homogeneous() const { return false; }
    static constexpr bool kSingleSegment = false;
    static constexpr bool kAlwaysRealCollision = false;
    static constexpr bool kNullZero = false;
};
using GridMedium = GridMediumT<false>;
using GridMediumGrey = GridMediumT<false, true>;
using NanoDenseMedium = GridMediumT<true>;
using NanoDenseMediumGrey = GridMediumT<true, true>;
template <bool NVDB, bool GREY, int BND = -1, bool EMIT = false, bool MAJLDS = false>
VDEV GridMediumT<NVDB, GREY, BND, EMIT, MAJLDS> make_grid(const DScene &S, const float *majorant) {
    return GridMediumT<NVDB, GREY, BND, EMIT, MAJLDS>{GREY ? sp(S.sigma_a[0]) : lds(S.sigma_a), GREY ? sp(S.sigma_s[0]) : lds(S.sigma_s), S.g, S.nx, S.ny, S.nz, ld3(S.bounds_min), ld3(S.bounds_max), S.brick_index, S.octets, S.bnx, S.bny,
                             majorant, S.index_min[0], S.index_min[1], S.index_min[2], ld3(S.inv_voxel), ld3(S.grid_origin),
                             S.density_offset, S.le_scale, S.le_nx, S.le_ny, S.le_nz, lds(S.Le), S.has_xform ? S.minv : nullptr,
                             S.temperature, S.temperature_offset, S.temperature_scale, S.nvdb_le_scale};
}
template <class M> struct MediumMaker;
template <int GREY, bool NZ, bool SIMPLE> struct MediumMaker<HomogeneousMediumT<GREY, NZ, SIMPLE>> {
    static VDEV HomogeneousMediumT<GREY, NZ, SIMPLE> make(const DScene &, const float *) { return HomogeneousMediumT<GREY, NZ, SIMPLE>{}; }
};
template <bool NVDB, bool GREY, int BND, bool EMIT, bool MAJLDS> struct MediumMaker<GridMediumT<NVDB, GREY, BND, EMIT, MAJLDS>> {
    static VDEV GridMediumT<NVDB, GREY, BND, EMIT, MAJLDS> make(const DScene &S, const float *majorant) {
        return make_grid<NVDB, GREY, BND, EMIT, MAJLDS>(S, majorant ? majorant : S.majorant);
    }
};

// ---------------------------------------------------------------------------------------
// a7: SampleT_maj (src/pbrt/media_sampleTMaj.h:49-117)
//   callback: bool cb(V3 p, const MediumProps&, Spec sigma_maj, Spec T_maj, bool activateNDS)
// ---------------------------------------------------------------------------------------
template <class Medium, class F>
VDEV Spec sample_T_maj(const Medium &medium, V3 ro, V3 rd, float tMax, float u, Rng &rng, int ch, F &&cb) {
    tMax *= len(rd);
    rd = normalize(rd);
    auto iter = medium.sample_ray(ro, rd, tMax);
    Spec T_maj = sp(1.f);
    bool done = false;
    while (!done) {
        MajSeg seg;
        if (!iter.next(&seg)) return T_maj;
        float smaj = ch_of(seg.sigma_maj, ch);
        if (smaj == 0) {
            float dt = seg.tMax - seg.tMin;
            if (isinf_(dt)) dt = kFltMax;
            T_maj = T_maj * fast_exp(seg.sigma_maj * -dt);
            continue;
        }
        float tMin = seg.tMin;
        while (true) {
            float t = tMin + sample_exponential(u, smaj);
            u = rng.uniform();
            if (t < seg.tMax) {
                T_maj = T_maj * fast_exp(seg.sigma_maj * -(t - tMin));
                V3 p = ro + rd * t;
                MediumProps mp = medium.sample_point(p);
                if (!cb(p, mp, seg.sigma_maj, T_maj, false)) {
                    done = true;
                    break;
                }
                T_maj = sp(1.f);
                tMin = t;
            } else {
                float dt = seg.tMax - tMin;
                if (isinf_(dt)) dt = kFltMax;
                T_maj = T_maj * fast_exp(seg.sigma_maj * -dt);
                break;
            }
        }
    }
    return sp(1.f);
}

// The pre-pass of the two guided traversals on its own: sum of sigma_maj * dt over the ray's majorant segments = the
// expected number of tentative collisions of a tracking walk along it.  The workgroup kernel sorts rays by it.
template <class Medium>
VDEV float majorant_optical_depth(const Medium &medium, V3 ro, V3 rd, float tMax, int ch) {
    tMax *= len(rd);
    rd = normalize(rd);
    auto pre = medium.sample_ray(ro, rd, tMax);
    float total = 0.f;
    while (true) {
        MajSeg seg;
        if (!pre.next(&seg)) break;
        float smaj = ch_of(seg.sigma_maj, ch);
        if (smaj == 0) continue;
        total += smaj * (seg.tMax - seg.tMin);
    }
    return total;
}

// ---------------------------------------------------------------------------------------
// a9: SampleT_maj_Resampling (media_sampleTMaj.h:136-248)
// ---------------------------------------------------------------------------------------
template <class Medium, class F>
VDEV Spec sample_T_maj_resampling(const Medium &medium, V3 ro, V3 rd, float tMax, float u, Rng &rng, int ch,
                                  bool guide, float vsp, float *vrc, float *majorantScale, F &&cb) {
    tMax *= len(rd);
    rd = normalize(rd);
    auto iter = medium.sample_ray(ro, rd, tMax);
    auto pre = iter;
    float totalLength = 0.f;
    while (true) {
        MajSeg seg;
        if (!pre.next(&seg)) break;
        float smaj = ch_of(seg.sigma_maj, ch);
        if (smaj == 0) continue;
        totalLength += smaj * (seg.tMax - seg.tMin);
    }
    if (totalLength == 0.f) return sp(1.f);
    *majorantScale = 1.0f;
    *vrc = vsp;
    if (guide) {
        float minTotalLength = -logf_(1 - vsp);
        if (minTotalLength > totalLength) {
            *majorantScale = minTotalLength / totalLength;
            totalLength = minTotalLength;
        }
        float expNegTotalLength = fast_exp(-totalLength);
        *vrc = wdiv(vsp, 1 - expNegTotalLength);
    }
    Spec T_maj = sp(1.f);
    bool done = false;
    int count = 0;
    while (!done) {
        MajSeg seg;
        if (!iter.next(&seg)) return T_maj;
        seg.sigma_maj = seg.sigma_maj * *majorantScale;
        float smaj = ch_of(seg.sigma_maj, ch);
        if (smaj == 0) {
            float dt = seg.tMax - seg.tMin;
            if (isinf_(dt)) dt = kFltMax;
            T_maj = T_maj * fast_exp(seg.sigma_maj * -dt);
            continue;
        }
        float tMin = seg.tMin;
        while (true) {
            count++;
            float t = tMin + sample_exponential(u, smaj);
            u = rng.uniform();
            if (t < seg.tMax) {
                if (count > 10000) break;
                T_maj = T_maj * fast_exp(seg.sigma_maj * -(t - tMin));
                V3 p = ro + rd * t;
                MediumProps mp = medium.sample_point(p);
                if (!cb(p, mp, seg.sigma_maj, T_maj, false)) {
                    done = true;
                    break;
                }
                T_maj = sp(1.f);
                tMin = t;
            } else {
                float dt = seg.tMax - tMin;
                if (isinf_(dt)) dt = kFltMax;
                T_maj = T_maj * fast_exp(seg.sigma_maj * -dt);
                break;
            }
        }
    }
    return sp(1.f);
}

// ---------------------------------------------------------------------------------------
// a8: SampleT_maj_OpticalDepthSpace (media_sampleTMaj.h:269-491)
// ---------------------------------------------------------------------------------------
VDEV Spec ruf_from(float alpha, Spec tp) {  // SampledSpectrum(a)/tp + SampledSpectrum(1-a)
    return wdiv(sp(alpha), tp) + sp(1 - alpha);
}
template <class Medium, class F>
VDEV Spec sample_T_maj_ods(const Medium &medium, V3 ro, V3 rd, float tMax, float u, Rng &rng, int ch,
                           bool guide, float vsp, float alpha, bool NDS, Spec *r_u_factor, F &&cb) {
    // The reference falls back to SampleT_maj at three places (:279 unguided, :307 infinite segment,
    // :327 NDS with vsp below the volume's own scatter probability); they are folded into ONE call
    // site here (one inlined copy of the tracking loop + callback instead of three).  The first
    // fallback passes the ray as given, the other two the normalised ray and scaled tMax -- as there.
    bool fallback = !guide || alpha == 0.f;
    float t_v = 0.f, t_n = -1.f, t_n_current = -1.f;
    decltype(medium.sample_ray(ro, rd, tMax)) iter;
    if (!fallback) {
        tMax *= len(rd);
        rd = normalize(rd);
        iter = medium.sample_ray(ro, rd, tMax);
        auto pre = iter;
        bool inf_seg = false;
        while (true) {
            MajSeg seg;
            if (!pre.next(&seg)) break;
            if (isinf_(seg.tMax)) {
                inf_seg = true;
                break;
            }
            float smaj = ch_of(seg.sigma_maj, ch);
            if (smaj == 0) continue;
            t_v += smaj * (seg.tMax - seg.tMin);
        }
        if (inf_seg) {
            fallback = true;
        } else {
            if (t_v == 0.f) return sp(1.f);
            if (NDS) {
                float OneMinusENegTv = 1.f - fast_exp(-t_v);
                if (vsp < 1 - fast_exp(-t_v)) {
                    fallback = true;
                } else {
                    t_n = neg_log1m_d(OneMinusENegTv / vsp);
                    t_n_current = t_n;
                }
            }
        }
    }
    if (fallback) return sample_T_maj(medium, ro, rd, tMax, u, rng, ch, cb);

    Spec T_maj = sp(1.f), tpScale = sp(1.f);
    float t_v_current = t_v;
    float remainingDist = 0;
    bool deltaTracking = false;
    if (u > alpha) {
        deltaTracking = true;
        u = (u - alpha) / (1 - alpha);
    } else {
        u /= alpha;
    }
    bool done = false, overTheEnd = false;
    int count = 0;
    const float ScatterEpsilon = 1e-5;
    while (!done) {
        MajSeg seg;
        if (!iter.next(&seg)) return T_maj;
        float smaj = ch_of(seg.sigma_maj, ch);
        if (smaj == 0 || overTheEnd) {
            float dt = seg.tMax - seg.tMin;
            if (isinf_(dt)) dt = kFltMax;
            T_maj = T_maj * fast_exp(seg.sigma_maj * -dt);
            continue;
        }
        float tMin = seg.tMin;
        // (grey medium: this is x / x == 1; selecting the constant instead of dividing measured 0.7 % SLOWER on the headline
        //  kernel -- the select keeps both operands alive across the block -- so the division stays)
        Spec nMaj = seg.sigma_maj / smaj;
        // a medium with ONE majorant segment never carries a remaining optical distance into a next
        // segment: the block below is dead for it and the loop body runs once
        if (!Medium::kSingleSegment && remainingDist > 0) {
            tMin += remainingDist / smaj;
            if (tMin > seg.tMax + ScatterEpsilon) {
                float dist = (seg.tMax - seg.tMin) * smaj;
                t_v_current -= dist;
                t_n_current -= dist;
                remainingDist -= dist;
                T_maj = T_maj * fast_exp(seg.sigma_maj * -(seg.tMax - seg.tMin));
                continue;
            }
            t_v_current -= remainingDist;
            t_n_current -= remainingDist;
            remainingDist = 0;
            T_maj = T_maj * fast_exp(seg.sigma_maj * -(tMin - seg.tMin));
            V3 p = ro + rd * tMin;
            MediumProps mp = medium.sample_point(p);
            *r_u_factor = ruf_from(alpha, tpScale);
            if (!cb(p, mp, seg.sigma_maj, T_maj, true)) break;
            T_maj = sp(1.f);
        }
        while (true) {
            count++;
            float dist = kFltMax;
            Spec tpStep;
            // (1 - FastExp(-t * normalizedMaj)) [/ vsp]: Float * SampledSpectrum products, then the
            // grey-aware fast_exp / division
            if (NDS)
                tpStep = sp(1.0f) - fast_exp(nMaj * -t_n_current);
            else
                tpStep = (sp(1.0f) - fast_exp(nMaj * -t_v_current)) / vsp;
            // one -std::log(1.0 - x) per step: x = u (delta tracking) or u * tpStep[ch] (guided)
            bool want_log = deltaTracking || NDS || u < vsp;
            if (want_log) dist = neg_log1m_d(deltaTracking ? u : u * ch_of(tpStep, ch));

            bool passThrough = (t_v_current - dist < ScatterEpsilon) || dist == 0;
            if (NDS || !passThrough) tpScale = tpScale * tpStep;
            if (passThrough) {
                if (NDS) {
                    tpScale = wdiv(tpScale, 1.0f - fast_exp(-t_n + t_v));
                } else {
                    tpScale = tpScale * wdiv(fast_exp(nMaj * -t_v_current), 1 - vsp);
                }
                *r_u_factor = ruf_from(alpha, tpScale);
                overTheEnd = true;
                T_maj = T_maj * fast_exp(seg.sigma_maj * -(seg.tMax - tMin));
                break;
            }
            float t = tMin + dist / smaj;
            u = rng.uniform();
            if (t <= seg.tMax + ScatterEpsilon) {
                if (count > 10000) break;
                t_v_current -= dist;
                t_n_current -= dist;
                remainingDist = 0;
                T_maj = T_maj * fast_exp(seg.sigma_maj * -(t - tMin));
                V3 p = ro + rd * t;
                MediumProps mp = medium.sample_point(p);
                *r_u_factor = ruf_from(alpha, tpScale);
                if (!cb(p, mp, seg.sigma_maj, T_maj, true)) {
                    done = true;
                    break;
                }
                T_maj = sp(1.f);
                tMin = t;
            } else {
                float dt = seg.tMax - tMin;
                if (isinf_(dt)) dt = kFltMax;
                T_maj = T_maj * fast_exp(seg.sigma_maj * -dt);
                float distWithinThisSeg = dt * smaj;
                remainingDist = dist - distWithinThisSeg;
                t_v_current -= distWithinThisSeg;
                t_n_current -= distWithinThisSeg;
                break;
            }
        }
        if (Medium::kSingleSegment) return done ? sp(1.f) : T_maj;  // the next iter.next() would fail (:350)
    }
    return sp(1.f);
}

VSPG_NS_END  // namespace vspg
