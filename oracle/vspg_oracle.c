/*
 * vspg_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See vspg_oracle.h.
 *
 * Plain-C restatement of the GuidedVolPathVSPG hot path of kehanxuuu/vspg-pbrt-v4.
 * Reference paths are relative to the reference root.  Build with
 *   gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math   (see oracle/Makefile)
 * so that float arithmetic follows the reference's default x86-64 build (no FMA contraction;
 * explicit FMA() in the reference == fmaf() here).
 *
 * Float-evaluation notes that matter for seed parity:
 *   - `std::log(1.0 - x)` in media_sampleTMaj.h is DOUBLE precision (1.0 is a double literal);
 *     `std::log(1 - u)` in SampleExponential is FLOAT (logf).
 *   - FastExp is the reference's own 2^x polynomial, not libm.
 */
#define _GNU_SOURCE
#include "vspg_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------ */
/* small vector / spectrum types (RGB mode: SampledSpectrum == 3 floats,                 */
/* src/pbrt/util/spectrum.h:40-44)                                                        */
/* ------------------------------------------------------------------------------------ */
typedef struct { float x, y, z; } v3;
typedef struct { float c[3]; } spec;

#define PI_F 3.14159265358979323846f
#define INV_PI_F 0.31830988618379067154f
#define INV_4PI_F 0.07957747154594766788f
#define PI_OVER_2_F 1.57079632679489661923f
#define PI_OVER_4_F 0.78539816339744830961f
#define ONE_MINUS_EPS 0x1.fffffep-1f   /* util/float.h:27 */
#define MACHINE_EPS 0x1p-24f           /* numeric_limits<float>::epsilon()*0.5 */
#define SHADOW_EPS 0.0001f             /* util/math.h:42 */

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_from(const float *p) { return V3(p[0], p[1], p[2]); }
static inline v3 v_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v_scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 v_neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline v3 v_abs(v3 a) { return V3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
static inline float sqr(float x) { return x * x; }
/* util/vecmath.h:964 */
static inline float v_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float v_absdot(v3 a, v3 b) { return fabsf(v_dot(a, b)); }
/* util/vecmath.h:948-961: Normalize divides every component by the length */
static inline float v_len2(v3 a) { return sqr(a.x) + sqr(a.y) + sqr(a.z); }
static inline float v_len(v3 a) { return sqrtf(v_len2(a)); }
static inline v3 v_normalize(v3 a) { float l = v_len(a); return V3(a.x / l, a.y / l, a.z / l); }
/* util/math.h:569-574 */
static inline float diff_of_products(float a, float b, float c, float d) {
    float cd = c * d;
    float dop = fmaf(a, b, -cd);
    float err = fmaf(-c, d, cd);
    return dop + err;
}
/* util/vecmath.h:999-1004 */
static inline v3 v_cross(v3 v, v3 w) {
    return V3(diff_of_products(v.y, w.z, v.z, w.y), diff_of_products(v.z, w.x, v.x, w.z),
              diff_of_products(v.x, w.y, v.y, w.x));
}
/* Lerp(t, p0, p1) = (1-t)*p0 + t*p1 (util/math.h:210) */
static inline v3 v_lerp(float t, v3 a, v3 b) { return v_add(v_scale(a, 1 - t), v_scale(b, t)); }
static inline float safe_sqrt(float x) { return sqrtf(fmaxf(0.f, x)); }
static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

static inline spec S1(float v) { spec s = {{v, v, v}}; return s; }
static inline spec s_from(const float *p) { spec s = {{p[0], p[1], p[2]}}; return s; }
#define S_BINOP(name, op)                                                  \
    static inline spec name(spec a, spec b) {                              \
        spec r;                                                            \
        for (int i = 0; i < 3; ++i) r.c[i] = a.c[i] op b.c[i];             \
        return r;                                                          \
    }
S_BINOP(s_add, +)
S_BINOP(s_sub, -)
S_BINOP(s_mul, *)
S_BINOP(s_div, /)
static inline spec s_scale(spec a, float f) { spec r; for (int i = 0; i < 3; ++i) r.c[i] = a.c[i] * f; return r; }
static inline spec s_divf(spec a, float f) { spec r; for (int i = 0; i < 3; ++i) r.c[i] = a.c[i] / f; return r; }
static inline int s_nonzero(spec a) { return a.c[0] != 0 || a.c[1] != 0 || a.c[2] != 0; }
/* util/spectrum.h:288-294 */
static inline float s_avg(spec a) { float sum = a.c[0]; sum += a.c[1]; sum += a.c[2]; return sum / 3; }
static inline float s_max(spec a) { float m = a.c[0]; m = fmaxf(m, a.c[1]); m = fmaxf(m, a.c[2]); return m; }
static inline spec s_clamp_zero(spec a) { spec r; for (int i = 0; i < 3; ++i) r.c[i] = fmaxf(0.f, a.c[i]); return r; }
static inline int s_has_nan(spec a) { return isnan(a.c[0]) || isnan(a.c[1]) || isnan(a.c[2]); }
static inline int s_has_inf(spec a) { return isinf(a.c[0]) || isinf(a.c[1]) || isinf(a.c[2]); }

/* ------------------------------------------------------------------------------------ */
/* a2: hashing (src/pbrt/util/hash.h:19-107)                                              */
/* ------------------------------------------------------------------------------------ */
uint64_t oracle_murmur64a(const unsigned char *key, size_t len, uint64_t seed) {
    const uint64_t m = 0xc6a4a7935bd1e995ull;
    const int r = 47;
    uint64_t h = seed ^ (len * m);
    const unsigned char *end = key + 8 * (len / 8);
    while (key != end) {
        uint64_t k;
        memcpy(&k, key, 8);
        key += 8;
        k *= m;
        k ^= k >> r;
        k *= m;
        h ^= k;
        h *= m;
    }
    switch (len & 7) {
    case 7: h ^= (uint64_t)key[6] << 48; /* fallthrough */
    case 6: h ^= (uint64_t)key[5] << 40; /* fallthrough */
    case 5: h ^= (uint64_t)key[4] << 32; /* fallthrough */
    case 4: h ^= (uint64_t)key[3] << 24; /* fallthrough */
    case 3: h ^= (uint64_t)key[2] << 16; /* fallthrough */
    case 2: h ^= (uint64_t)key[1] << 8;  /* fallthrough */
    case 1: h ^= (uint64_t)key[0]; h *= m;
    }
    h ^= h >> r;
    h *= m;
    h ^= h >> r;
    return h;
}
/* hash.h:70-77 */
uint64_t oracle_mix_bits(uint64_t v) {
    v ^= (v >> 31);
    v *= 0x7fb5d329728ea185ull;
    v ^= (v >> 27);
    v *= 0x81dadef4bc2dd44dull;
    v ^= (v >> 33);
    return v;
}
/* Hash(args...) (hash.h:96-103): the arguments are memcpy'd back to back, seed 0 */
uint64_t oracle_hash_float(float f) { return oracle_murmur64a((const unsigned char *)&f, 4, 0); }
uint64_t oracle_hash_pixel_seed(int32_t x, int32_t y, int32_t seed) {
    int32_t buf[3] = {x, y, seed};
    return oracle_murmur64a((const unsigned char *)buf, 12, 0);
}
uint64_t oracle_hash_point3(float x, float y, float z) {
    float buf[3] = {x, y, z};
    return oracle_murmur64a((const unsigned char *)buf, 12, 0);
}

/* ------------------------------------------------------------------------------------ */
/* a1: PCG32 RNG (src/pbrt/util/rng.h:30-150)                                             */
/* ------------------------------------------------------------------------------------ */
#define PCG32_MULT 0x5851f42d4c957f2dULL
typedef struct { uint64_t state, inc; } rng_t;

static inline uint32_t rng_u32(rng_t *r) { /* rng.h:82-88 */
    uint64_t oldstate = r->state;
    r->state = oldstate * PCG32_MULT + r->inc;
    uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
    uint32_t rot = (uint32_t)(oldstate >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}
static inline void rng_set_sequence2(rng_t *r, uint64_t seq, uint64_t seed) { /* rng.h:119-125 */
    r->state = 0u;
    r->inc = (seq << 1u) | 1u;
    rng_u32(r);
    r->state += seed;
    rng_u32(r);
}
static inline void rng_set_sequence1(rng_t *r, uint64_t seq) { /* rng.h:43-45 */
    rng_set_sequence2(r, seq, oracle_mix_bits(seq));
}
static inline float rng_float(rng_t *r) { /* rng.h:128-130 */
    float f = (float)rng_u32(r) * 0x1p-32f;
    return f < ONE_MINUS_EPS ? f : ONE_MINUS_EPS;
}
static inline void rng_advance(rng_t *r, int64_t idelta) { /* rng.h:137-150 */
    uint64_t curMult = PCG32_MULT, curPlus = r->inc, accMult = 1u;
    uint64_t accPlus = 0u, delta = (uint64_t)idelta;
    while (delta > 0) {
        if (delta & 1) {
            accMult *= curMult;
            accPlus = accPlus * curMult + curPlus;
        }
        curPlus = (curMult + 1) * curPlus;
        curMult *= curMult;
        delta /= 2;
    }
    r->state = accMult * r->state + accPlus;
}
void oracle_rng_seq(uint64_t seq, uint64_t seed, int has_seed, int64_t advance, int n,
                    uint32_t *out_u32, float *out_f) {
    rng_t a, b;
    if (has_seed) rng_set_sequence2(&a, seq, seed); else rng_set_sequence1(&a, seq);
    rng_advance(&a, advance);
    b = a;
    for (int i = 0; i < n; ++i) {
        if (out_u32) out_u32[i] = rng_u32(&a);
        if (out_f) out_f[i] = rng_float(&b);
    }
}

/* IndependentSampler (src/pbrt/samplers.h:442-476) */
typedef struct { rng_t rng; } sampler_t;
static inline void sampler_start_pixel_sample(sampler_t *s, int32_t px, int32_t py, int32_t seed,
                                              int sampleIndex, int dimension) {
    rng_set_sequence1(&s->rng, oracle_hash_pixel_seed(px, py, seed));
    rng_advance(&s->rng, (int64_t)((uint64_t)sampleIndex * 65536ull + (uint64_t)dimension));
}
static inline float sampler_get1d(sampler_t *s) { return rng_float(&s->rng); }
void oracle_independent_sampler(int32_t px, int32_t py, int32_t seed, int32_t sampleIndex, int n,
                                float *out) {
    sampler_t s;
    sampler_start_pixel_sample(&s, px, py, seed, sampleIndex, 0);
    for (int i = 0; i < n; ++i) out[i] = sampler_get1d(&s);
}

/* ------------------------------------------------------------------------------------ */
/* a3: FastExp / SampleExponential / SampleDiscrete                                       */
/* ------------------------------------------------------------------------------------ */
static inline uint32_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float b2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
/* util/math.h:450-474 (CPU branch) */
float oracle_fast_exp(float x) {
    float xp = x * 1.442695041f;
    float fxp = floorf(xp), f = xp - fxp;
    int i = (int)fxp;
    /* EvaluatePolynomial(f, 1, c1, c2, c3) = FMA(f, FMA(f, FMA(f, c3, c2), c1), 1) (math.h:334-337) */
    float twoToF = fmaf(f, fmaf(f, fmaf(f, 0.0781455737f, 0.226173572f), 0.695556856f), 1.f);
    int exponent = (int)((f2b(twoToF) >> 23) & 0xff) - 127 + i; /* Exponent(), util/float.h */
    if (exponent < -126) return 0;
    if (exponent > 127) return INFINITY;
    uint32_t bits = f2b(twoToF);
    bits &= 0x807fffffu;
    bits |= (uint32_t)(exponent + 127) << 23;
    return b2f(bits);
}
static inline spec s_fast_exp(spec a) { spec r; for (int i = 0; i < 3; ++i) r.c[i] = oracle_fast_exp(a.c[i]); return r; }

/* SampledWavelengths::SampleVisible in the RGB build (util/spectrum.h:369-386): three wavelengths from the sample lu that also
 * picks the hero channel, lambda_i = SampleVisibleWavelengths(up_i) = 538 - 138.888889 atanh(0.85691062 - 1.82750197 up_i)
 * (util/sampling.h:169-171; std::atanh(float) is the host's atanhf).  Only volume emission of a temperature grid reads them. */
static void sample_visible_wavelengths(float u, float lambda[3]) {
    for (int i = 0; i < 3; ++i) {
        float up = u + (float)i / 3;
        if (up > 1) up -= 1;
        lambda[i] = 538 - 138.888889f * atanhf(0.85691062f - 1.82750197f * up);
    }
}
/* Blackbody(lambda, T) (util/spectrum.h:83-94): Planck's law with FastExp and Pow<5> (util/math.h:294-309: ((l l)(l l)) l) */
static float blackbody(float lambda, float T) {
    if (T <= 0) return 0;
    const float c = 299792458.f;
    const float h = 6.62606957e-34f;
    const float kb = 1.3806488e-23f;
    float l = lambda * 1e-9f;
    float l2 = l * l * 1.f;
    float l5 = l2 * l2 * l;
    float Le = (2 * h * c * c) / (l5 * (oracle_fast_exp((h * c) / (l * kb * T)) - 1));
    return Le;
}
/* BlackbodySpectrum(T).Sample(lambda) (util/spectrum.h:568-588): normalised by the value at Wien's peak; in the RGB build the three
 * sampled wavelengths' values ARE the R, G, B channels */
static spec blackbody_sample(float T, const float lambda[3]) {
    float lambdaMax = 2.8977721e-3f / T;
    float normalizationFactor = 1 / blackbody(lambdaMax * 1e9f, T);
    spec s;
    for (int i = 0; i < 3; ++i) s.c[i] = blackbody(lambda[i], T) * normalizationFactor;
    return s;
}
/* test API: [lambda0..2, Le0..2] of SampleVisible(u) and BlackbodySpectrum(T).Sample (golden "blackbody") */
void oracle_blackbody(float u, float T, float *out6) {
    float lambda[3];
    sample_visible_wavelengths(u, lambda);
    spec s = blackbody_sample(T, lambda);
    for (int i = 0; i < 3; ++i) { out6[i] = lambda[i]; out6[3 + i] = s.c[i]; }
}
/* test API: Blackbody(lambda [nm], T) itself, for the reference's own known answers (util/spectrum_test.cpp:19-48) */
float oracle_blackbody_radiance(float lambda_nm, float T) { return blackbody(lambda_nm, T); }
/* util/sampling.h:222-225 -- float log */
float oracle_sample_exponential(float u, float a) { return -logf(1 - u) / a; }
/* util/float.h:164-193 */
static inline float next_float_up(float v) {
    if (isinf(v) && v > 0.f) return v;
    if (v == -0.f) v = 0.f;
    uint32_t ui = f2b(v);
    if (v >= 0) ++ui; else --ui;
    return b2f(ui);
}
static inline float next_float_down(float v) {
    if (isinf(v) && v < 0.f) return v;
    if (v == 0.f) v = -0.f;
    uint32_t ui = f2b(v);
    if (v > 0) --ui; else ++ui;
    return b2f(ui);
}
/* util/sampling.h:79-113 for two weights */
int oracle_sample_discrete2(float w0, float w1, float u) {
    float weights[2] = {w0, w1};
    float sumWeights = 0;
    sumWeights += w0;
    sumWeights += w1;
    float up = u * sumWeights;
    if (up == sumWeights) up = next_float_down(up);
    int offset = 0;
    float sum = 0;
    while (sum + weights[offset] <= up) sum += weights[offset++];
    return offset;
}

/* ------------------------------------------------------------------------------------ */
/* a4: Henyey-Greenstein + direction sampling                                             */
/* ------------------------------------------------------------------------------------ */
/* util/scattering.h:50-59 */
float oracle_henyey_greenstein(float cosTheta, float g) {
    g = clampf(g, (float)-.99, (float).99);
    float denom = 1 + sqr(g) + 2 * g * cosTheta;
    return INV_4PI_F * (1 - sqr(g)) / (denom * safe_sqrt(denom));
}
/* util/vecmath.h:1007-1013 */
static inline void coordinate_system(v3 v1, v3 *v2, v3 *v3o) {
    float sign = copysignf(1.f, v1.z);
    float a = -1 / (sign + v1.z);
    float b = v1.x * v1.y * a;
    *v2 = V3(1 + sign * sqr(v1.x) * a, sign * b, -sign * v1.x);
    *v3o = V3(b, sign + sqr(v1.y) * a, -v1.y);
}
void oracle_coordinate_system(const float v[3], float v2[3], float v3o[3]) {
    v3 a, b;
    coordinate_system(v3_from(v), &a, &b);
    v2[0] = a.x; v2[1] = a.y; v2[2] = a.z;
    v3o[0] = b.x; v3o[1] = b.y; v3o[2] = b.z;
}
typedef struct { v3 x, y, z; } frame_t;
/* Frame::FromLocal (vecmath.h:1914): v.x*x + v.y*y + v.z*z */
static inline v3 frame_from_local(const frame_t *f, v3 v) {
    return v_add(v_add(v_scale(f->x, v.x), v_scale(f->y, v.y)), v_scale(f->z, v.z));
}
static inline v3 frame_to_local(const frame_t *f, v3 v) { return V3(v_dot(v, f->x), v_dot(v, f->y), v_dot(v, f->z)); }
/* exported for the golden-vector test: Frame::FromXZ(x, z) = Frame(x, Cross(z, x), z) (vecmath.h:1862), ToLocal, FromLocal */
void oracle_frame_xz(const float x[3], const float z[3], const float v[3], float out_y[3], float out_local[3], float out_world[3]) {
    frame_t f;
    f.x = v3_from(x);
    f.z = v3_from(z);
    f.y = v_cross(f.z, f.x);
    v3 l = frame_to_local(&f, v3_from(v)), w = frame_from_local(&f, v3_from(v));
    out_y[0] = f.y.x; out_y[1] = f.y.y; out_y[2] = f.y.z;
    out_local[0] = l.x; out_local[1] = l.y; out_local[2] = l.z;
    out_world[0] = w.x; out_world[1] = w.y; out_world[2] = w.z;
}
/* util/vecmath.h:1666-1672 */
static inline v3 spherical_direction(float sinTheta, float cosTheta, float phi) {
    return V3(clampf(sinTheta, -1, 1) * cosf(phi), clampf(sinTheta, -1, 1) * sinf(phi),
              clampf(cosTheta, -1, 1));
}
/* util/sampling.cpp:348-374 */
static v3 sample_henyey_greenstein(v3 wo, float g, float u0, float u1, float *pdf) {
    g = clampf(g, (float)-.99, (float).99);
    float cosTheta;
    if (fabsf(g) < 1e-3f)
        cosTheta = 1 - 2 * u0;
    else
        cosTheta = -1 / (2 * g) * (1 + sqr(g) - sqr((1 - sqr(g)) / (1 + g - 2 * g * u0)));
    float sinTheta = safe_sqrt(1 - sqr(cosTheta));
    float phi = 2 * PI_F * u1;
    frame_t wFrame;
    wFrame.z = wo;
    coordinate_system(wo, &wFrame.x, &wFrame.y); /* Frame::FromZ (vecmath.h:1867-1871) */
    v3 wi = frame_from_local(&wFrame, spherical_direction(sinTheta, cosTheta, phi));
    if (pdf) *pdf = oracle_henyey_greenstein(cosTheta, g);
    return wi;
}
void oracle_sample_henyey_greenstein(const float wo[3], float g, float u0, float u1, float wi[3],
                                     float *pdf) {
    v3 r = sample_henyey_greenstein(v3_from(wo), g, u0, u1, pdf);
    wi[0] = r.x; wi[1] = r.y; wi[2] = r.z;
}
/* util/sampling.h:391-396 */
void oracle_sample_uniform_sphere(float u0, float u1, float out[3]) {
    float z = 1 - 2 * u0;
    float r = safe_sqrt(1 - sqr(z));
    float phi = 2 * PI_F * u1;
    out[0] = r * cosf(phi); out[1] = r * sinf(phi); out[2] = z;
}
/* util/sampling.h:325-341, 409-413 */
static v3 sample_cosine_hemisphere(float u0, float u1) {
    float ox = 2 * u0 - 1, oy = 2 * u1 - 1;
    float dx, dy;
    if (ox == 0 && oy == 0) {
        dx = 0; dy = 0;
    } else {
        float theta, r;
        if (fabsf(ox) > fabsf(oy)) {
            r = ox;
            theta = PI_OVER_4_F * (oy / ox);
        } else {
            r = oy;
            theta = PI_OVER_2_F - PI_OVER_4_F * (ox / oy);
        }
        dx = r * cosf(theta); dy = r * sinf(theta);
    }
    float z = safe_sqrt(1 - sqr(dx) - sqr(dy));
    return V3(dx, dy, z);
}
void oracle_sample_cosine_hemisphere(float u0, float u1, float out[3]) {
    v3 r = sample_cosine_hemisphere(u0, u1);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* ------------------------------------------------------------------------------------ */
/* ray spawning (src/pbrt/ray.h:75-108)                                                   */
/* ------------------------------------------------------------------------------------ */
/* Point3fi (util/vecmath.h:737-760) over Interval (util/math.h:815-862): a point with
 * error bounds is stored as [low, high] per axis; p() is the midpoint, Error() half the width. */
typedef struct { v3 lo, hi; } p3i;
static inline p3i p3i_exact(v3 p) { p3i r = {p, p}; return r; }
static inline void interval_from_value_and_error(float v, float err, float *lo, float *hi) {
    if (err == 0) { *lo = *hi = v; }
    else { *lo = next_float_down(v - err); *hi = next_float_up(v + err); } /* SubRoundDown / AddRoundUp */
}
static inline p3i p3i_from_err(v3 p, v3 e) {
    p3i r;
    interval_from_value_and_error(p.x, e.x, &r.lo.x, &r.hi.x);
    interval_from_value_and_error(p.y, e.y, &r.lo.y, &r.hi.y);
    interval_from_value_and_error(p.z, e.z, &r.lo.z, &r.hi.z);
    return r;
}
static inline v3 p3i_mid(p3i a) { return V3((a.lo.x + a.hi.x) / 2, (a.lo.y + a.hi.y) / 2, (a.lo.z + a.hi.z) / 2); }
static inline v3 p3i_err(p3i a) { return V3((a.hi.x - a.lo.x) / 2, (a.hi.y - a.lo.y) / 2, (a.hi.z - a.lo.z) / 2); }

static v3 offset_ray_origin(p3i pi, v3 n, v3 w) {
    float d = v_dot(v_abs(n), p3i_err(pi));
    v3 offset = v_scale(n, d);
    if (v_dot(w, n) < 0) offset = v_neg(offset);
    v3 po = v_add(p3i_mid(pi), offset);
    float *o = &offset.x, *q = &po.x;
    for (int i = 0; i < 3; ++i) {
        if (o[i] > 0) q[i] = next_float_up(q[i]);
        else if (o[i] < 0) q[i] = next_float_down(q[i]);
    }
    return po;
}
/* SpawnRayTo(Point3fi pFrom, Normal3f nFrom, Float time, Point3fi pTo, Normal3f nTo) (ray.h:103-108) */
static void spawn_ray_to(p3i pFrom, v3 nFrom, p3i pTo, v3 nTo, v3 *o, v3 *d) {
    v3 pf = offset_ray_origin(pFrom, nFrom, v_sub(p3i_mid(pTo), p3i_mid(pFrom)));
    v3 pt = offset_ray_origin(pTo, nTo, v_sub(pf, p3i_mid(pTo)));
    *o = pf;
    *d = v_sub(pt, pf);
}
void oracle_spawn_ray_to(const float pf[3], const float pferr[3], const float nf[3], const float pt[3], const float pterr[3],
                         const float nt[3], float out_o[3], float out_d[3]) {
    v3 o, d;
    spawn_ray_to(p3i_from_err(v3_from(pf), v3_from(pferr)), v3_from(nf), p3i_from_err(v3_from(pt), v3_from(pterr)), v3_from(nt), &o, &d);
    out_o[0] = o.x; out_o[1] = o.y; out_o[2] = o.z;
    out_d[0] = d.x; out_d[1] = d.y; out_d[2] = d.z;
}
void oracle_offset_ray_origin(const float p[3], const float perr[3], const float n[3],
                              const float w[3], float out[3]) {
    v3 r = offset_ray_origin(p3i_from_err(v3_from(p), v3_from(perr)), v3_from(n), v3_from(w));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* ------------------------------------------------------------------------------------ */
/* scene: rectangles standing in for "bilinearmesh" quads (out-of-scope geometry restated */
/* minimally; the formulas follow the cited pbrt code)            */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    v3 p00, p10, p01, p11, e1, e2;
    v3 n;          /* Normalize(Cross(e1,e2)), negated if reverse_orientation */
    v3 dpdu_n;     /* Normalize(e1): BSDF shading frame x (bsdf.h:25-26) */
    v3 perr;       /* gamma(6) * (|p00|+|p01|+|p10|+|p11|) (shapes.cpp:1207-1208) */
    float inv_l1, inv_l2; /* 1/|e1|^2, 1/|e2|^2 */
    float area;
    spec Kd, Le;
    int two_sided, is_light, has_bsdf_lobes;
    int has_material; /* 0: Material "interface" (GetBSDF returns no BSDF, :399-404) */
    int iface;        /* VSPG_IFACE_* bits when the surface is a medium TRANSITION, else 0 (interaction.h:218-229) */
} rquad_t;

typedef struct {
    int hit;
    float t;
    int quad;   /* rectangle index, or -1 when `tri` names a triangle */
    int tri;    /* triangle (index into r->tris), -1 for a rectangle hit */
    int sphere; /* sphere index, -1 otherwise */
    v3 p, n;
    v3 perr;    /* error bound of p */
    p3i pi;     /* sphere hits: the interaction point as Transform::operator()(Point3fi) returned it (p = its midpoint) */
    v3 dpdu_n;  /* sphere hits: Normalize(shading.dpdu) in render space (the BSDF frame's x) */
    int has_material, iface; /* of the surface hit (see rquad_t) */
} isect_t;
/* SurfaceInteraction::pi of a hit */
static inline p3i isect_pi(const isect_t *si) { return si->sphere >= 0 ? si->pi : p3i_from_err(si->p, si->perr); }

/* f1: a triangle of the soup with what Triangle::InteractionFromIntersection derives from its vertices (shapes.h:888-938) */
typedef struct {
    v3 p0, p1, p2;
    v3 n;       /* Normalize(Cross(p0 - p2, p1 - p2)) */
    v3 dpdu_n;  /* Normalize(dpdu), default (u,v) parameterisation */
    spec Kd;
    int id;
    int has_bsdf_lobes;
    int has_material, iface;
} rtri_t;

/* MediumInterface::IsMediumTransition (base/medium.h:124): inside != outside; anything else leaves the ray's medium alone */
static int iface_transition(int bits) {
    bits &= VSPG_IFACE_INSIDE | VSPG_IFACE_OUTSIDE;
    return (bits == VSPG_IFACE_INSIDE || bits == VSPG_IFACE_OUTSIDE) ? bits : 0;
}
/* Dot(Vector3f, Normal3f) (vecmath.h:1064-1068): FMA(n.x, v.x, SumOfProducts(n.y, v.y, n.z, v.z)), math.h:577-583 */
static inline float dot_vn(v3 v, v3 n) {
    float cd = n.z * v.z;
    float sop = fmaf(n.y, v.y, cd);
    float err = fmaf(n.z, v.z, -cd);
    return fmaf(n.x, v.x, sop + err);
}
/* Interaction::GetMedium(w) (interaction.h:117-121) with one medium: 1 = the scene's medium, 0 = none.  `medium` is the
 * interaction's own `medium` member: the medium of the ray that hit a non-transition surface / of a medium interaction. */
static inline int get_medium(int iface, v3 n, v3 w, int medium) {
    if (iface) return dot_vn(w, n) > 0 ? ((iface & VSPG_IFACE_OUTSIDE) ? 1 : 0) : ((iface & VSPG_IFACE_INSIDE) ? 1 : 0);
    return medium;
}
static void quad_init(rquad_t *q, const VspgQuad *in) {
    q->p00 = v3_from(in->p00);
    q->e1 = v3_from(in->e1);
    q->e2 = v3_from(in->e2);
    q->p10 = v_add(q->p00, q->e1);
    q->p01 = v_add(q->p00, q->e2);
    q->p11 = v_add(q->p10, q->e2);
    v3 c = v_cross(q->e1, q->e2);
    q->area = v_len(c);
    q->n = v_normalize(c);
    if (in->reverse_orientation) q->n = v_neg(q->n);
    q->dpdu_n = v_normalize(q->e1);
    float g6 = (6 * MACHINE_EPS) / (1 - 6 * MACHINE_EPS); /* gamma(6), util/float.h:195 */
    v3 s = v_add(v_add(v_abs(q->p00), v_abs(q->p01)), v_add(v_abs(q->p10), v_abs(q->p11)));
    q->perr = v_scale(s, g6);
    q->inv_l1 = 1.f / v_len2(q->e1);
    q->inv_l2 = 1.f / v_len2(q->e2);
    q->Kd = s_from(in->Kd);
    for (int i = 0; i < 3; ++i) q->Kd.c[i] = clampf(q->Kd.c[i], 0, 1); /* DiffuseMaterial clamps R */
    q->Le = s_from(in->Le);
    q->two_sided = in->two_sided;
    q->is_light = s_nonzero(q->Le);
    q->has_bsdf_lobes = s_nonzero(q->Kd); /* DiffuseBxDF::Flags (bxdfs.h:82-84) */
    q->has_material = in->material != VSPG_MATERIAL_INTERFACE;
    q->iface = iface_transition(in->medium_interface);
}

/* ray / rectangle: plane hit, then parametric (u,v) test; the reported point is re-projected
 * onto the rectangle (p00 + u e1 + v e2) as pbrt does for bilinear patches
 * (shapes.h InteractionFromIntersection). */
static int quad_intersect(const rquad_t *q, v3 o, v3 d, float tMax, float *tHit, v3 *pHit) {
    float denom = v_dot(q->n, d);
    if (denom == 0) return 0;
    float t = v_dot(q->n, v_sub(q->p00, o)) / denom;
    if (!(t > 0) || !(t < tMax)) return 0;
    v3 p = v_add(o, v_scale(d, t));
    v3 rel = v_sub(p, q->p00);
    float u = v_dot(rel, q->e1) * q->inv_l1;
    float v = v_dot(rel, q->e2) * q->inv_l2;
    if (u < 0 || u > 1 || v < 0 || v > 1) return 0;
    *tHit = t;
    *pHit = v_add(q->p00, v_add(v_scale(q->e1, u), v_scale(q->e2, v)));
    return 1;
}

/* ------------------------------------------------------------------------------------ */
/* Shape "sphere" (round 4; shapes.h:107-330): Interval arithmetic (util/math.h:818-1010,  */
/* CPU branches of the rounding helpers util/float.h:199-297), Sphere::BasicIntersect       */
/* (shapes.h:147-229), InteractionFromIntersection (:237-284),                              */
/* Transform::operator()(SurfaceInteraction) (transform.cpp:229-261).  Full spheres only.   */
/* ------------------------------------------------------------------------------------ */
typedef struct { float lo, hi; } ivl;
static inline float min_f(float a, float b) { return b < a ? b : a; } /* std::min */
static inline float max_f(float a, float b) { return a < b ? b : a; } /* std::max */
static inline ivl ivl_x(float v) { ivl r = {v, v}; return r; }
static inline ivl ivl_mk(float a, float b) { ivl r = {min_f(a, b), max_f(a, b)}; return r; } /* Interval(low, high) */
static inline float ivl_mid(ivl a) { return (a.lo + a.hi) / 2; }
static inline ivl ivl_from_err(float v, float e) { ivl r; interval_from_value_and_error(v, e, &r.lo, &r.hi); return r; }
static inline ivl ivl_add(ivl a, ivl b) { return ivl_mk(next_float_down(a.lo + b.lo), next_float_up(a.hi + b.hi)); }
static inline ivl ivl_sub(ivl a, ivl b) { return ivl_mk(next_float_down(a.lo + -b.hi), next_float_up(a.hi + -b.lo)); }
static inline float min4(const float *v) { float m = v[0]; for (int i = 1; i < 4; ++i) m = min_f(m, v[i]); return m; }
static inline float max4(const float *v) { float m = v[0]; for (int i = 1; i < 4; ++i) m = max_f(m, v[i]); return m; }
static inline ivl ivl_mul(ivl a, ivl b) {
    float lp[4] = {next_float_down(a.lo * b.lo), next_float_down(a.hi * b.lo), next_float_down(a.lo * b.hi), next_float_down(a.hi * b.hi)};
    float hp[4] = {next_float_up(a.lo * b.lo), next_float_up(a.hi * b.lo), next_float_up(a.lo * b.hi), next_float_up(a.hi * b.hi)};
    return ivl_mk(min4(lp), max4(hp));
}
static inline int ivl_has(ivl i, float v) { return v >= i.lo && v <= i.hi; } /* InRange(v, i) */
static inline ivl ivl_div(ivl a, ivl b) {
    if (ivl_has(b, 0)) return ivl_mk(-INFINITY, INFINITY);
    float lq[4] = {next_float_down(a.lo / b.lo), next_float_down(a.hi / b.lo), next_float_down(a.lo / b.hi), next_float_down(a.hi / b.hi)};
    float hq[4] = {next_float_up(a.lo / b.lo), next_float_up(a.hi / b.lo), next_float_up(a.lo / b.hi), next_float_up(a.hi / b.hi)};
    return ivl_mk(min4(lq), max4(hq));
}
static inline ivl ivl_sqr(ivl i) {
    float alow = fabsf(i.lo), ahigh = fabsf(i.hi);
    if (alow > ahigh) { float t = alow; alow = ahigh; ahigh = t; }
    if (ivl_has(i, 0)) return ivl_mk(0, next_float_up(ahigh * ahigh));
    return ivl_mk(next_float_down(alow * alow), next_float_up(ahigh * ahigh));
}
static inline ivl ivl_fmul(float f, ivl i) { /* operator*(Float, Interval) */
    if (f > 0) return ivl_mk(next_float_down(f * i.lo), next_float_up(f * i.hi));
    return ivl_mk(next_float_down(f * i.hi), next_float_up(f * i.lo));
}
static inline ivl ivl_sqrt(ivl i) { return ivl_mk(max_f(0.f, next_float_down(sqrtf(i.lo))), next_float_up(sqrtf(i.hi))); }

typedef struct {
    float m[16], mi[16]; /* renderFromObject: m and mInv, row-major */
    float radius;
    int flip;            /* reverseOrientation ^ transformSwapsHandedness */
    float thetaZMin, thetaZMax, phiMax;
    spec Kd;
    int has_bsdf_lobes, has_material, iface;
} rsphere_t;

/* Transform::SwapsHandedness (transform.cpp:145-152) with Determinant(SquareMatrix<3>) (math.h:1419-1425) */
static int swaps_handedness(const float *m) {
    float minor12 = diff_of_products(m[5], m[10], m[6], m[9]);
    float minor02 = diff_of_products(m[4], m[10], m[6], m[8]);
    float minor01 = diff_of_products(m[4], m[9], m[5], m[8]);
    return fmaf(m[2], minor01, diff_of_products(m[0], minor12, m[1], minor02)) < 0;
}
static int sphere_init(rsphere_t *S, const VspgSphere *in) {
    memcpy(S->m, in->render_from_object, sizeof S->m);
    memcpy(S->mi, in->object_from_render, sizeof S->mi);
    if (!(in->radius > 0)) return VSPG_EINVAL;
    for (int k = 0; k < 3; ++k) if (S->m[12 + k] != 0 || S->mi[12 + k] != 0) return VSPG_EINVAL; /* affine only */
    if (S->m[15] != 1 || S->mi[15] != 1) return VSPG_EINVAL;
    S->radius = in->radius;
    S->flip = (in->reverse_orientation ? 1 : 0) ^ swaps_handedness(S->m);
    /* zMin = -radius, zMax = radius, phiMax = 360 (the defaults, shapes.cpp:231-237) */
    S->thetaZMin = acosf(clampf(min_f(-S->radius, S->radius) / S->radius, -1, 1));
    S->thetaZMax = acosf(clampf(max_f(-S->radius, S->radius) / S->radius, -1, 1));
    S->phiMax = (PI_F / 180) * clampf(360.f, 0, 360); /* Radians() (math.h:261-263); == fl(2 pi): the phi clip of :184, :200 never fires */
    S->Kd = s_from(in->Kd);
    for (int i = 0; i < 3; ++i) S->Kd.c[i] = clampf(S->Kd.c[i], 0, 1);
    S->has_bsdf_lobes = s_nonzero(S->Kd);
    S->has_material = in->material != VSPG_MATERIAL_INTERFACE;
    S->iface = iface_transition(in->medium_interface);
    return 0;
}
/* Sphere::BasicIntersect (shapes.h:147-229) for zMin = -radius, zMax = radius, phiMax = 2 pi (no clipping) */
static int sphere_intersect(const rsphere_t *S, v3 ro, v3 rd, float tMax, float *tHit, v3 *pObj) {
    const float *mi = S->mi;
    const float g3 = (3 * MACHINE_EPS) / (1 - 3 * MACHINE_EPS);
    /* oi = (*objectFromRender)(Point3fi(r.o)): exact input (transform.h:133-175) */
    ivl oi[3], di[3];
    {
        float x = ro.x, y = ro.y, z = ro.z;
        for (int k = 0; k < 3; ++k) {
            const float *row = mi + 4 * k;
            float v = (row[0] * x + row[1] * y) + (row[2] * z + row[3]);
            float e = g3 * (fabsf(row[0] * x) + fabsf(row[1] * y) + fabsf(row[2] * z) + fabsf(row[3]));
            oi[k] = ivl_from_err(v, e);
        }
    }
    { /* di = (*objectFromRender)(Vector3fi(r.d)) (transform.h:272-305) */
        float x = rd.x, y = rd.y, z = rd.z;
        for (int k = 0; k < 3; ++k) {
            const float *row = mi + 4 * k;
            float e = g3 * (fabsf(row[0] * x) + fabsf(row[1] * y) + fabsf(row[2] * z));
            float v = row[0] * x + row[1] * y + row[2] * z;
            di[k] = ivl_from_err(v, e);
        }
    }
    ivl a = ivl_add(ivl_add(ivl_sqr(di[0]), ivl_sqr(di[1])), ivl_sqr(di[2]));
    ivl b = ivl_fmul(2.f, ivl_add(ivl_add(ivl_mul(di[0], oi[0]), ivl_mul(di[1], oi[1])), ivl_mul(di[2], oi[2])));
    ivl c = ivl_sub(ivl_add(ivl_add(ivl_sqr(oi[0]), ivl_sqr(oi[1])), ivl_sqr(oi[2])), ivl_sqr(ivl_x(S->radius)));
    /* v = oi - b / (2 * a) * di; discrim = 4 a (radius + |v|)(radius - |v|) */
    ivl f = ivl_div(b, ivl_fmul(2.f, a));
    ivl v[3];
    for (int k = 0; k < 3; ++k) v[k] = ivl_sub(oi[k], ivl_mul(f, di[k]));
    ivl length = ivl_sqrt(ivl_add(ivl_add(ivl_sqr(v[0]), ivl_sqr(v[1])), ivl_sqr(v[2])));
    ivl discrim = ivl_mul(ivl_mul(ivl_fmul(4.f, a), ivl_add(ivl_x(S->radius), length)), ivl_sub(ivl_x(S->radius), length));
    if (discrim.lo < 0) return 0;
    ivl rootDiscrim = ivl_sqrt(discrim);
    ivl q;
    if (ivl_mid(b) < 0) q = ivl_fmul(-.5f, ivl_sub(b, rootDiscrim));
    else q = ivl_fmul(-.5f, ivl_add(b, rootDiscrim));
    ivl t0 = ivl_div(q, a), t1 = ivl_div(c, q);
    if (t0.lo > t1.lo) { ivl t = t0; t0 = t1; t1 = t; }
    if (t0.hi > tMax || t1.lo <= 0) return 0;
    ivl tShapeHit = t0;
    if (tShapeHit.lo <= 0) {
        tShapeHit = t1;
        if (tShapeHit.hi > tMax) return 0;
    }
    /* pHit = Point3f(oi) + (Float)tShapeHit * Vector3f(di); refined onto the sphere */
    float ts = ivl_mid(tShapeHit);
    v3 pHit = V3(ivl_mid(oi[0]) + ts * ivl_mid(di[0]), ivl_mid(oi[1]) + ts * ivl_mid(di[1]), ivl_mid(oi[2]) + ts * ivl_mid(di[2]));
    float sc = S->radius / v_len(pHit); /* Distance(pHit, Point3f(0, 0, 0)) */
    pHit = V3(pHit.x * sc, pHit.y * sc, pHit.z * sc);
    if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * S->radius;
    /* (phi = atan2(y, x) feeds only the clip against phiMax -- never true for phiMax = fl(2 pi) -- and the (u, v)
     *  parameterisation, which a diffuse / interface surface does not read) */
    *tHit = ts;
    *pObj = pHit;
    return 1;
}
/* InteractionFromIntersection (shapes.h:237-284) carried to render space by Transform::operator()(SurfaceInteraction)
 * (transform.cpp:229-261): pi, the geometric normal and Normalize(shading.dpdu) */
static void sphere_interaction(const rsphere_t *S, v3 pHit, p3i *pi, v3 *n, v3 *dpdu_n) {
    const float *m = S->m, *mi = S->mi;
    float cosTheta = pHit.z / S->radius;
    float zRadius = sqrtf(sqr(pHit.x) + sqr(pHit.y));
    float cosPhi = pHit.x / zRadius, sinPhi = pHit.y / zRadius;
    v3 dpdu = V3(-S->phiMax * pHit.y, S->phiMax * pHit.x, 0);
    float sinTheta = safe_sqrt(1 - sqr(cosTheta));
    float dth = S->thetaZMax - S->thetaZMin;
    v3 dpdv = V3(dth * (pHit.z * cosPhi), dth * (pHit.z * sinPhi), dth * (-S->radius * sinTheta));
    v3 no = v_normalize(v_cross(dpdu, dpdv));
    if (S->flip) no = v_neg(no);
    const float g5 = (5 * MACHINE_EPS) / (1 - 5 * MACHINE_EPS), g3 = (3 * MACHINE_EPS) / (1 - 3 * MACHINE_EPS);
    v3 pErr = V3(g5 * fabsf(pHit.x), g5 * fabsf(pHit.y), g5 * fabsf(pHit.z));
    /* t(si.pi): Point3fi with error bounds (transform.h:133-175, the approximate branch; exact when pError == 0) */
    ivl pin[3] = {ivl_from_err(pHit.x, pErr.x), ivl_from_err(pHit.y, pErr.y), ivl_from_err(pHit.z, pErr.z)};
    float x = ivl_mid(pin[0]), y = ivl_mid(pin[1]), z = ivl_mid(pin[2]);
    float ex = (pin[0].hi - pin[0].lo) / 2, ey = (pin[1].hi - pin[1].lo) / 2, ez = (pin[2].hi - pin[2].lo) / 2;
    int exact = (pin[0].hi - pin[0].lo) == 0 && (pin[1].hi - pin[1].lo) == 0 && (pin[2].hi - pin[2].lo) == 0;
    float pv[3], pe[3];
    for (int k = 0; k < 3; ++k) {
        const float *row = m + 4 * k;
        pv[k] = (row[0] * x + row[1] * y) + (row[2] * z + row[3]);
        if (exact)
            pe[k] = g3 * (fabsf(row[0] * x) + fabsf(row[1] * y) + fabsf(row[2] * z) + fabsf(row[3]));
        else
            pe[k] = (g3 + 1) * (fabsf(row[0]) * ex + fabsf(row[1]) * ey + fabsf(row[2]) * ez) +
                    g3 * (fabsf(row[0] * x) + fabsf(row[1] * y) + fabsf(row[2] * z) + fabsf(row[3]));
    }
    *pi = p3i_from_err(V3(pv[0], pv[1], pv[2]), V3(pe[0], pe[1], pe[2]));
    /* Normalize(t(si.n)): normals transform by the inverse transpose (transform.h:329-334) */
    v3 nr = V3(mi[0] * no.x + mi[4] * no.y + mi[8] * no.z, mi[1] * no.x + mi[5] * no.y + mi[9] * no.z, mi[2] * no.x + mi[6] * no.y + mi[10] * no.z);
    *n = v_normalize(nr);
    /* shading.dpdu = t(dpdu) (Vector3 transform, transform.h:322-327); the BSDF frame takes Normalize() of it (bsdf.h:25-26) */
    v3 du = V3(m[0] * dpdu.x + m[1] * dpdu.y + m[2] * dpdu.z, m[4] * dpdu.x + m[5] * dpdu.y + m[6] * dpdu.z, m[8] * dpdu.x + m[9] * dpdu.y + m[10] * dpdu.z);
    *dpdu_n = v_normalize(du);
}
/* test entry points: one Interval operation (0 add, 1 sub, 2 mul, 3 div, 4 Sqr(a), 5 Sqrt(Abs(a)), 6 f * a) */
void oracle_interval_op(int op, float alo, float ahi, float blo, float bhi, float f, float out[2]) {
    ivl a = {alo, ahi}, b = {blo, bhi}, r;
    switch (op) {
    case 0: r = ivl_add(a, b); break;
    case 1: r = ivl_sub(a, b); break;
    case 2: r = ivl_mul(a, b); break;
    case 3: r = ivl_div(a, b); break;
    case 4: r = ivl_sqr(a); break;
    case 5: { /* Abs(Interval) (math.h:994-1004) */
        ivl aa = a.lo >= 0 ? a : (a.hi <= 0 ? ivl_mk(-a.hi, -a.lo) : ivl_mk(0, max_f(-a.lo, a.hi)));
        r = ivl_sqrt(aa);
        break;
    }
    default: r = ivl_fmul(f, a); break;
    }
    out[0] = r.lo; out[1] = r.hi;
}
/* ... and what Li does at an interface hit on a sphere: SkipIntersection's new origin and GetMedium's side test (:399-404) */
void oracle_sphere_skip(const VspgSphere *sp, const float o[3], const float d[3], float tMax, int *hit, float p_obj[3], float out_o[3], int *outside) {
    rsphere_t S;
    *hit = 0;
    if (sphere_init(&S, sp)) return;
    v3 pObj;
    float tHit;
    if (!sphere_intersect(&S, v3_from(o), v3_from(d), tMax, &tHit, &pObj)) return;
    p3i pi; v3 nn, du;
    sphere_interaction(&S, pObj, &pi, &nn, &du);
    *hit = 1;
    p_obj[0] = pObj.x; p_obj[1] = pObj.y; p_obj[2] = pObj.z;
    v3 so = offset_ray_origin(pi, nn, v3_from(d));
    out_o[0] = so.x; out_o[1] = so.y; out_o[2] = so.z;
    *outside = dot_vn(v3_from(d), nn) > 0;
}
void oracle_sphere_intersect(const VspgSphere *sp, const float o[3], const float d[3], float tMax, int *hit, float *tHit, float p_lo[3], float p_hi[3],
                             float n[3], float dpdu_n[3]) {
    rsphere_t S;
    *hit = 0;
    if (sphere_init(&S, sp)) return;
    v3 pObj;
    if (!sphere_intersect(&S, v3_from(o), v3_from(d), tMax, tHit, &pObj)) return;
    p3i pi; v3 nn, du;
    sphere_interaction(&S, pObj, &pi, &nn, &du);
    *hit = 1;
    p_lo[0] = pi.lo.x; p_lo[1] = pi.lo.y; p_lo[2] = pi.lo.z; p_hi[0] = pi.hi.x; p_hi[1] = pi.hi.y; p_hi[2] = pi.hi.z;
    n[0] = nn.x; n[1] = nn.y; n[2] = nn.z; dpdu_n[0] = du.x; dpdu_n[1] = du.y; dpdu_n[2] = du.z;
}

/* ------------------------------------------------------------------------------------ */
/* renderer state                                                                         */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    spec sigma_a, sigma_s, Le;
    float g; /* HG phase */
    /* temperature grids: MediumProperties::Le is BlackbodySpectrum(temp).Sample(lambda) times a scale (media.h:333-341, :724-735).
     * SamplePoint here has no wavelengths at hand, so it records the two factors and medium_Le() below finishes the product where
     * the path reads mp.Le (the delta-tracking callback, the only reader: guidedvolpathvspgintegrator.cpp:895-906).
     * bb_temp = 0: Le above is the value. */
    float bb_temp, bb_scale;
} medium_props_t; /* MediumProperties, media.h:77-82 */

typedef struct lightsampler lightsampler_t; /* "light samplers" section below */
struct OracleRenderer {
    lightsampler_t *lsamp; /* LightSampler::Create(prm.lightsampler, lights) (lightsamplers.cpp:49-64), built with the renderer */
    int light_of_quad[VSPG_MAX_QUADS]; /* rectangle -> index into the light list, -1: not a light */
    VspgScene scene;
    VspgIntegratorParams prm;
    VspgRenderConfig cfg;
    int n_quads;
    rquad_t quads[VSPG_MAX_QUADS];
    rtri_t *tris; /* non-degenerate triangles of the soup, caller's order */
    int n_tris;
    int n_spheres;
    rsphere_t spheres[VSPG_MAX_SPHERES];
    /* medium boundaries (round 4): has_medium = the scene holds a medium at all; has_boundaries = some surface is an
     * interface material or a medium transition, or the camera sits outside the medium (else: the medium fills the scene) */
    int has_medium, has_boundaries, camera_in_medium;
    /* infinite lights (lights.h:207-250 DistantLight, :554-601 UniformInfiniteLight); light order = emissive rectangles, then these */
    int n_inf;
    float scene_radius; /* Bounds3::BoundingSphere of the scene bounds (vecmath.h:1335-1338), light.Preprocess (integrators.h:74-81) */
    int n_lights;
    int light_quads[VSPG_MAX_QUADS];
    /* guiding fields (copies): [0] surface, [1] volume; nodes == NULL -> untrained */
    struct { int n_nodes, n_regions; VspgKdNode *nodes; VspgFieldRegion *regions; } field[2];
    /* GridMedium: density samples (copied) and the 16^3 majorant grid (media.cpp:252-269) */
    float *density;
    float *majorant;
    float *le_scale; /* emissive GridMedium: the LeScale grid (copied), NULL = not emissive */
    float *temperature; /* temperature grid of an emissive GridMedium / NanoVDBMedium (copied), NULL = none */
    int le_dim[3];
    /* film: RGBFilm::Pixel (film.h:314-318) */
    double *film; /* W*H*4 */
    /* image-space VSP buffer (own design; OpenPGL absent) */
    float *isg_stats; /* W*H*VSPG_ISG_STATS */
    float *vsp;       /* W*H */
    int vsp_ready;
    int vsp_loaded; /* ImageSpaceGuidingBuffer(fileName): never updated (:151-159, :251-256) */
    int wave_counter, buffer_wave;
    VspgCounters counters;
    /* a18: guiding-cache training (see "training" section below) */
    int training, field_iteration, field_uploaded;
    VspgTrainSample *samples;
    size_t n_samples, cap_samples;
    uint64_t n_zero_samples;
    struct region_stats *rstats[2];
    int cap_nodes[2], cap_regions[2];
    /* TrBuffer (cpu/trbuffer.h:17-104): per-pixel running mean of the primary ray's ratio-tracking
     * transmittance estimate; tr_calc = calculateTrBuffer, tr_load = trBufferLoad (:180-193) */
    float *trbuf; /* W*H*3 */
    int *tr_spp;  /* W*H */
    int tr_calc, tr_load;
    /* image-space contribution estimate for guided RR (own stand-in for ImageSpaceGuidingBuffer::GetContributionEstimate):
     * 5x5 box-filtered film mean, refreshed with the VSP buffer at waves 1, 2, 4, ... */
    float *contrib; /* W*H*3 */
    int contrib_ready;
    int in_wave;  /* render_wave is running (debug path traces do not feed the buffers) */
};

typedef struct { /* LightSampleContext (base/light.h) */
    p3i pi;
    v3 n, ns;
} lsctx_t;

typedef struct {
    uint64_t segments, volume_scatters, surface_hits, density_queries, shadow_rays, shadow_queries;
} path_counters_t;

typedef struct {
    int valid;        /* an event was recorded */
    int surface_event;
    float vsp_used;   /* clamped primary VSP used on the first segment, -1 if unguided */
} isg_sample_t;

static _Thread_local uint32_t g_dbg_flags;
#define DBG(f) (g_dbg_flags |= (f))

/* IntersectTriangle (shapes.cpp:168-262) */
static float max3abs(float a, float b, float c) {
    float m = fabsf(a);
    m = m < fabsf(b) ? fabsf(b) : m;
    return m < fabsf(c) ? fabsf(c) : m;
}
static float v_comp(v3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }
static int tri_intersect(v3 o, v3 d, float tMax, v3 p0, v3 p1, v3 p2, float *t_out, float b_out[3]) {
    if (v_len2(v_cross(v_sub(p2, p0), v_sub(p1, p0))) == 0) return 0;
    v3 p0t = v_sub(p0, o), p1t = v_sub(p1, o), p2t = v_sub(p2, o);
    v3 ad = v_abs(d);
    int kz = (ad.x > ad.y) ? ((ad.x > ad.z) ? 0 : 2) : ((ad.y > ad.z) ? 1 : 2);
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    v3 dp = V3(v_comp(d, kx), v_comp(d, ky), v_comp(d, kz));
    p0t = V3(v_comp(p0t, kx), v_comp(p0t, ky), v_comp(p0t, kz));
    p1t = V3(v_comp(p1t, kx), v_comp(p1t, ky), v_comp(p1t, kz));
    p2t = V3(v_comp(p2t, kx), v_comp(p2t, ky), v_comp(p2t, kz));
    float Sx = -dp.x / dp.z, Sy = -dp.y / dp.z, Sz = 1 / dp.z;
    p0t.x += Sx * p0t.z; p0t.y += Sy * p0t.z;
    p1t.x += Sx * p1t.z; p1t.y += Sy * p1t.z;
    p2t.x += Sx * p2t.z; p2t.y += Sy * p2t.z;
    float e0 = diff_of_products(p1t.x, p2t.y, p1t.y, p2t.x);
    float e1 = diff_of_products(p2t.x, p0t.y, p2t.y, p0t.x);
    float e2 = diff_of_products(p0t.x, p1t.y, p0t.y, p1t.x);
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        double p2txp1ty = (double)p2t.x * (double)p1t.y, p2typ1tx = (double)p2t.y * (double)p1t.x;
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)p0t.x * (double)p2t.y, p0typ2tx = (double)p0t.y * (double)p2t.x;
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)p1t.x * (double)p0t.y, p1typ0tx = (double)p1t.y * (double)p0t.x;
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return 0;
    float det = e0 + e1 + e2;
    if (det == 0) return 0;
    p0t.z *= Sz; p1t.z *= Sz; p2t.z *= Sz;
    float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0 && (tScaled >= 0 || tScaled < tMax * det)) return 0;
    else if (det > 0 && (tScaled <= 0 || tScaled > tMax * det)) return 0;
    float invDet = 1 / det;
    float b0 = e0 * invDet, b1 = e1 * invDet, b2 = e2 * invDet;
    float t = tScaled * invDet;
    const float g2 = (2 * MACHINE_EPS) / (1 - 2 * MACHINE_EPS), g3 = (3 * MACHINE_EPS) / (1 - 3 * MACHINE_EPS),
                g5 = (5 * MACHINE_EPS) / (1 - 5 * MACHINE_EPS);
    float maxZt = max3abs(p0t.z, p1t.z, p2t.z);
    float deltaZ = g3 * maxZt;
    float maxXt = max3abs(p0t.x, p1t.x, p2t.x), maxYt = max3abs(p0t.y, p1t.y, p2t.y);
    float deltaX = g5 * (maxXt + maxZt), deltaY = g5 * (maxYt + maxZt);
    float deltaE = 2 * (g2 * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
    float maxE = max3abs(e0, e1, e2);
    float deltaT = 3 * (g3 * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * fabsf(invDet);
    if (t <= deltaT) return 0;
    *t_out = t; b_out[0] = b0; b_out[1] = b1; b_out[2] = b2;
    return 1;
}
/* closest hit (stands in for Integrator::Intersect -> BVHAggregate::Intersect, integrators.cpp:341-349, cpu/aggregates.cpp:529-590):
 * rectangles first, then EVERY triangle against the rectangles' closest distance (brute force: no BVH on this side);
 * equal distances go to the smaller triangle index -- the device's BVH traversal applies the same rule, so the visiting
 * order never matters.  (The reference's traversal lets the LAST visited of equal distances win: ties are measure-zero.) */
static isect_t scene_intersect(const OracleRenderer *r, v3 o, v3 d, float tMax) {
    isect_t best;
    memset(&best, 0, sizeof best);
    best.t = tMax;
    best.tri = -1;
    best.sphere = -1;
    for (int i = 0; i < r->n_quads; ++i) {
        float t;
        v3 p;
        if (quad_intersect(&r->quads[i], o, d, best.t, &t, &p)) {
            best.hit = 1; best.t = t; best.quad = i; best.p = p; best.n = r->quads[i].n;
        }
    }
    best.perr = r->quads[best.quad].perr;
    best.has_material = r->quads[best.quad].has_material;
    best.iface = r->quads[best.quad].iface;
    if (r->n_tris > 0) {
        const float tRect = best.t;
        int found = 0, bi = -1;
        float bt = 0, bb[3] = {0, 0, 0};
        for (int i = 0; i < r->n_tris; ++i) {
            float t, b[3];
            const rtri_t *T = &r->tris[i];
            if (tri_intersect(o, d, tRect, T->p0, T->p1, T->p2, &t, b) && t < tRect &&
                (!found || t < bt || (t == bt && T->id < r->tris[bi].id))) {
                found = 1; bi = i; bt = t; bb[0] = b[0]; bb[1] = b[1]; bb[2] = b[2];
            }
        }
        if (found) {
            const rtri_t *T = &r->tris[bi];
            best.hit = 1; best.t = bt; best.quad = -1; best.tri = bi;
            best.p = v_add(v_add(v_scale(T->p0, bb[0]), v_scale(T->p1, bb[1])), v_scale(T->p2, bb[2])); /* shapes.h:922 */
            v3 sum = v_add(v_add(v_abs(v_scale(T->p0, bb[0])), v_abs(v_scale(T->p1, bb[1]))), v_abs(v_scale(T->p2, bb[2])));
            const float g7 = (7 * MACHINE_EPS) / (1 - 7 * MACHINE_EPS);
            best.perr = V3(g7 * sum.x, g7 * sum.y, g7 * sum.z); /* shapes.h:929-930 */
            best.n = T->n;
            best.has_material = T->has_material;
            best.iface = T->iface;
        }
    }
    /* spheres: every one against the RAY's tMax (BasicIntersect rejects on interval bounds of t against the tMax it is
     * given; with the running closest distance the outcome would depend on the visiting order), closest by t */
    for (int i = 0; i < r->n_spheres; ++i) {
        float t;
        v3 pObj;
        if (sphere_intersect(&r->spheres[i], o, d, tMax, &t, &pObj) && t < best.t) {
            best.hit = 1; best.t = t; best.quad = -1; best.tri = -1; best.sphere = i;
            sphere_interaction(&r->spheres[i], pObj, &best.pi, &best.n, &best.dpdu_n);
            best.p = p3i_mid(best.pi);
            best.perr = p3i_err(best.pi);
            best.has_material = r->spheres[i].has_material;
            best.iface = r->spheres[i].iface;
        }
    }
    return best;
}
static int scene_intersect_any(const OracleRenderer *r, v3 o, v3 d, float tMax) {
    for (int i = 0; i < r->n_quads; ++i) {
        float t;
        v3 p;
        if (quad_intersect(&r->quads[i], o, d, tMax, &t, &p)) return 1;
    }
    for (int i = 0; i < r->n_tris; ++i) {
        float t, b[3];
        const rtri_t *T = &r->tris[i];
        if (tri_intersect(o, d, tMax, T->p0, T->p1, T->p2, &t, b) && t < tMax) return 1;
    }
    for (int i = 0; i < r->n_spheres; ++i) {
        float t;
        v3 pObj;
        if (sphere_intersect(&r->spheres[i], o, d, tMax, &t, &pObj)) return 1;
    }
    return 0;
}

/* The reference's shape tests replayed (shapes_test.cpp): Intersect(ray), then Interaction::SpawnRay(w) (interaction.h:99-101:
 * OffsetRayOrigin(pi, n, w), direction w) or SpawnRayTo(p2) (interaction.h:104-108 over ray.h:93-98: d = p2 - Point3f(pi), origin
 * OffsetRayOrigin(pi, n, d)) and Intersect / IntersectP of that ray. */
int oracle_ray_batch(OracleRenderer *r, int n, const VspgRayQuery *q, VspgRayResult *out) {
    for (int i = 0; i < n; ++i) {
        VspgRayResult *o = &out[i];
        memset(o, 0, sizeof *o);
        const isect_t si = scene_intersect(r, v3_from(q[i].o), v3_from(q[i].d), q[i].tMax);
        if (!si.hit) continue;
        const p3i pi = isect_pi(&si);
        const v3 pm = p3i_mid(pi);
        o->hit = 1;
        o->prim = si.sphere >= 0 ? 2000000 + si.sphere : (si.tri >= 0 ? 1000000 + r->tris[si.tri].id : si.quad);
        o->t = si.t;
        o->p[0] = pm.x; o->p[1] = pm.y; o->p[2] = pm.z;
        o->n[0] = si.n.x; o->n[1] = si.n.y; o->n[2] = si.n.z;
        if (q[i].mode == 0) continue;
        const v3 w = v3_from(q[i].w);
        const v3 d2 = q[i].mode == 1 ? w : v_sub(w, pm);
        const v3 o2 = offset_ray_origin(pi, si.n, d2);
        o->o2[0] = o2.x; o->o2[1] = o2.y; o->o2[2] = o2.z;
        o->d2[0] = d2.x; o->d2[1] = d2.y; o->d2[2] = d2.z;
        const isect_t s2 = scene_intersect(r, o2, d2, q[i].tMax2);
        o->hit2 = s2.hit;
        o->t2 = s2.hit ? s2.t : 0.f;
        o->any2 = scene_intersect_any(r, o2, d2, q[i].tMax2);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* a5/a6: media -- majorant iterators                                                     */
/* ------------------------------------------------------------------------------------ */
typedef struct { float tMin, tMax; spec sigma_maj; } majseg_t; /* RayMajorantSegment */
typedef struct {
    int type;
    int called;   /* HomogeneousMajorantIterator (media.h:84-106) */
    majseg_t seg;
    /* DDAMajorantIterator (media.h:140-218) */
    spec sigma_t;
    float tMin, tMax;
    const float *maj; /* MajorantGrid voxels, res 16^3 */
    float nextCrossingT[3], deltaT[3];
    int step[3], voxelLimit[3], voxel[3];
} majiter_t;

#define MAJ_RES 16      /* GridMedium majorant grid resolution (media.cpp:252) */
#define MAJ_RES_NVDB 64 /* NanoVDBMedium majorant grid resolution (media.cpp:574) */
static int medium_is_grid_like(int type) { return type == VSPG_MEDIUM_GRID || type == VSPG_MEDIUM_NANOVDB; }
static int medium_maj_res(int type) { return type == VSPG_MEDIUM_NANOVDB ? MAJ_RES_NVDB : MAJ_RES; }

static int majiter_next(majiter_t *it, majseg_t *seg) {
    if (it->type == VSPG_MEDIUM_HOMOGENEOUS) {
        if (it->called) return 0;
        it->called = 1;
        *seg = it->seg;
        return 1;
    }
    if (medium_is_grid_like(it->type)) { /* DDAMajorantIterator::Next (media.h:178-207) */
        const int MR = medium_maj_res(it->type);
        if (!(it->tMin < it->tMax)) return 0; /* == the reference's `tMin >= tMax` for ordered values; a NaN ray ends here instead of leaving the grid */
        int bits = ((it->nextCrossingT[0] < it->nextCrossingT[1]) << 2) +
                   ((it->nextCrossingT[0] < it->nextCrossingT[2]) << 1) +
                   ((it->nextCrossingT[1] < it->nextCrossingT[2]));
        const int cmpToAxis[8] = {2, 1, 2, 1, 2, 2, 0, 0};
        int stepAxis = cmpToAxis[bits];
        float tVoxelExit = it->nextCrossingT[stepAxis] < it->tMax ? it->nextCrossingT[stepAxis] : it->tMax; /* std::min(tMax, next) */
        float md = it->maj[it->voxel[0] + MR * (it->voxel[1] + MR * it->voxel[2])];
        seg->tMin = it->tMin;
        seg->tMax = tVoxelExit;
        seg->sigma_maj = s_scale(it->sigma_t, md);
        it->tMin = tVoxelExit;
        if (it->nextCrossingT[stepAxis] > it->tMax) it->tMin = it->tMax;
        it->voxel[stepAxis] += it->step[stepAxis];
        if (it->voxel[stepAxis] == it->voxelLimit[stepAxis]) it->tMin = it->tMax;
        it->nextCrossingT[stepAxis] += it->deltaT[stepAxis];
        return 1;
    }
    return 0;
}
/* Bounds3::Offset (vecmath.h:1323-1332) */
static v3 bounds_offset(const VspgMedium *m, v3 p) {
    v3 o = V3(p.x - m->bounds_min[0], p.y - m->bounds_min[1], p.z - m->bounds_min[2]);
    if (m->bounds_max[0] > m->bounds_min[0]) o.x /= m->bounds_max[0] - m->bounds_min[0];
    if (m->bounds_max[1] > m->bounds_min[1]) o.y /= m->bounds_max[1] - m->bounds_min[1];
    if (m->bounds_max[2] > m->bounds_min[2]) o.z /= m->bounds_max[2] - m->bounds_min[2];
    return o;
}
/* the same two functions for an arbitrary SampledGrid (the LeScale grid of an emissive GridMedium) */
static float sgrid_at(const float *data, int nx, int ny, int nz, int x, int y, int z) {
    if (x < 0 || y < 0 || z < 0 || x >= nx || y >= ny || z >= nz) return 0.f;
    return data[((size_t)z * ny + y) * nx + x];
}
static float sgrid_lookup(const float *data, int nx, int ny, int nz, v3 p) {
    float sx = p.x * nx - .5f, sy = p.y * ny - .5f, sz = p.z * nz - .5f;
    int ix = (int)floorf(sx), iy = (int)floorf(sy), iz = (int)floorf(sz);
    float dx = sx - (float)ix, dy = sy - (float)iy, dz = sz - (float)iz;
#define LERPF(t, a, b) ((1 - (t)) * (a) + (t) * (b))
#define G(x, y, z) sgrid_at(data, nx, ny, nz, x, y, z)
    float d00 = LERPF(dx, G(ix, iy, iz), G(ix + 1, iy, iz));
    float d10 = LERPF(dx, G(ix, iy + 1, iz), G(ix + 1, iy + 1, iz));
    float d01 = LERPF(dx, G(ix, iy, iz + 1), G(ix + 1, iy, iz + 1));
    float d11 = LERPF(dx, G(ix, iy + 1, iz + 1), G(ix + 1, iy + 1, iz + 1));
    float a = LERPF(dy, d00, d10), b = LERPF(dy, d01, d11);
    return LERPF(dz, a, b);
#undef G
#undef LERPF
}
/* Bounds3::IntersectP(o, d, tMax, &t0, &t1) (vecmath.h:1547-1571) */
static int bounds_intersect_p(const float bmin[3], const float bmax[3], v3 o, v3 d, float tMax, float *hit0, float *hit1) {
    const float g3 = (3 * MACHINE_EPS) / (1 - 3 * MACHINE_EPS);
    float t0 = 0, t1 = tMax;
    const float *oo = &o.x, *dd = &d.x;
    for (int i = 0; i < 3; ++i) {
        float invRayDir = 1 / dd[i];
        float tNear = (bmin[i] - oo[i]) * invRayDir;
        float tFar = (bmax[i] - oo[i]) * invRayDir;
        if (tNear > tFar) { float t = tNear; tNear = tFar; tFar = t; }
        tFar *= 1 + 2 * g3;
        t0 = tNear > t0 ? tNear : t0;
        t1 = tFar < t1 ? tFar : t1;
        if (t0 > t1) return 0;
    }
    *hit0 = t0;
    *hit1 = t1;
    return 1;
}
/* Transform::ApplyInverse(const Ray &, Float *tMax) with an IDENTITY matrix (transform.h:416-429, transform.cpp:263-303):
 * the origin picks up the conservative error bound gamma(3)*|o| and is pushed along d by dt, tMax shrinks by dt
 * (SURVEY.md App. C #15).  Returns the new origin. */
static v3 apply_inverse_identity(v3 o, v3 d, float *tMax) {
    const float g3 = (3 * MACHINE_EPS) / (1 - 3 * MACHINE_EPS);
    v3 oerr = V3(g3 * (fabsf(o.x) + 0.f + 0.f), g3 * (0.f + fabsf(o.y) + 0.f), g3 * (0.f + 0.f + fabsf(o.z)));
    p3i oi = p3i_from_err(o, oerr);
    float lengthSquared = v_len2(d);
    if (lengthSquared > 0) {
        v3 oe = p3i_err(oi);
        float dt = v_dot(v_abs(d), oe) / lengthSquared;
        v3 sh = v_scale(d, dt);
        /* Interval + Float: {AddRoundDown(low, f), AddRoundUp(high, f)} (math.h:873-875) */
        oi.lo = V3(next_float_down(oi.lo.x + sh.x), next_float_down(oi.lo.y + sh.y), next_float_down(oi.lo.z + sh.z));
        oi.hi = V3(next_float_up(oi.hi.x + sh.x), next_float_up(oi.hi.y + sh.y), next_float_up(oi.hi.z + sh.z));
        *tMax -= dt;
    }
    return p3i_mid(oi);
}
void oracle_apply_inverse_identity(const float o[3], const float d[3], float tMax, float out_o[3], float *out_tMax) {
    v3 r = apply_inverse_identity(v3_from(o), v3_from(d), &tMax);
    out_o[0] = r.x; out_o[1] = r.y; out_o[2] = r.z;
    *out_tMax = tMax;
}
/* Transform::ApplyInverse(const Point3fi &) for an EXACT point (transform.cpp:263-303) then ApplyInverse(const Ray &, Float *tMax)
 * (transform.h:416-429), general affine matrix mInv (row-major).  With the identity matrix every product is exact and
 * the result equals apply_inverse_identity. */
static v3 apply_inverse_ray(const float mi[16], v3 o, v3 d, float *tMax, v3 *d_out) {
    const float g3 = (3 * MACHINE_EPS) / (1 - 3 * MACHINE_EPS);
    float x = o.x, y = o.y, z = o.z;
    float xp = (mi[0] * x + mi[1] * y) + (mi[2] * z + mi[3]);
    float yp = (mi[4] * x + mi[5] * y) + (mi[6] * z + mi[7]);
    float zp = (mi[8] * x + mi[9] * y) + (mi[10] * z + mi[11]);
    float wp = (mi[12] * x + mi[13] * y) + (mi[14] * z + mi[15]);
    v3 oerr = V3(g3 * (fabsf(mi[0] * x) + fabsf(mi[1] * y) + fabsf(mi[2] * z)),
                 g3 * (fabsf(mi[4] * x) + fabsf(mi[5] * y) + fabsf(mi[6] * z)),
                 g3 * (fabsf(mi[8] * x) + fabsf(mi[9] * y) + fabsf(mi[10] * z)));
    p3i oi = p3i_from_err(V3(xp, yp, zp), oerr);
    (void)wp; /* affine matrices only: wp == 1, no division (vspg.h) */
    v3 dd = V3(mi[0] * d.x + mi[1] * d.y + mi[2] * d.z, mi[4] * d.x + mi[5] * d.y + mi[6] * d.z, mi[8] * d.x + mi[9] * d.y + mi[10] * d.z);
    float lengthSquared = v_len2(dd);
    if (lengthSquared > 0) {
        v3 oe = p3i_err(oi);
        float dt = v_dot(v_abs(dd), oe) / lengthSquared;
        v3 sh = v_scale(dd, dt);
        oi.lo = V3(next_float_down(oi.lo.x + sh.x), next_float_down(oi.lo.y + sh.y), next_float_down(oi.lo.z + sh.z));
        oi.hi = V3(next_float_up(oi.hi.x + sh.x), next_float_up(oi.hi.y + sh.y), next_float_up(oi.hi.z + sh.z));
        *tMax -= dt;
    }
    *d_out = dd;
    return p3i_mid(oi);
}
/* Transform::ApplyInverse(Point3<T>) (transform.h:387-398), affine */
static v3 apply_inverse_point(const float mi[16], v3 p) {
    float x = p.x, y = p.y, z = p.z;
    return V3((mi[0] * x + mi[1] * y) + (mi[2] * z + mi[3]), (mi[4] * x + mi[5] * y) + (mi[6] * z + mi[7]),
              (mi[8] * x + mi[9] * y) + (mi[10] * z + mi[11]));
}
void oracle_apply_inverse_ray(const float minv[16], const float o[3], const float d[3], float tMax, float out_o[3], float out_d[3],
                              float *out_tMax) {
    v3 dd;
    v3 r = apply_inverse_ray(minv, v3_from(o), v3_from(d), &tMax, &dd);
    out_o[0] = r.x; out_o[1] = r.y; out_o[2] = r.z;
    out_d[0] = dd.x; out_d[1] = dd.y; out_d[2] = dd.z;
    *out_tMax = tMax;
}
void oracle_apply_inverse_point(const float minv[16], const float p[3], float out[3]) {
    v3 r = apply_inverse_point(minv, v3_from(p));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
/* exported for the golden-vector test of Bounds3::Offset / IntersectP (tests/golden/primitives.json "bounds3") */
int oracle_bounds3(const float bmin[3], const float bmax[3], const float o[3], const float d[3], float tMax, float t01[2],
                   float offset[3]) {
    VspgMedium m;
    memset(&m, 0, sizeof m);
    for (int i = 0; i < 3; ++i) { m.bounds_min[i] = bmin[i]; m.bounds_max[i] = bmax[i]; }
    v3 off = bounds_offset(&m, v3_from(o));
    offset[0] = off.x; offset[1] = off.y; offset[2] = off.z;
    t01[0] = t01[1] = 0.f;
    return bounds_intersect_p(bmin, bmax, v3_from(o), v3_from(d), tMax, &t01[0], &t01[1]);
}
/* SampledGrid<Float>::Lookup(Point3i) (containers.h:830-835) */
static float grid_at(const OracleRenderer *r, int x, int y, int z) {
    const VspgMedium *m = &r->scene.medium;
    if (x < 0 || y < 0 || z < 0 || x >= m->nx || y >= m->ny || z >= m->nz) return 0.f;
    return r->density[((size_t)z * m->ny + y) * m->nx + x];
}
/* SampledGrid<Float>::Lookup(Point3f) trilinear (containers.h:804-819) */
static float grid_lookup(const OracleRenderer *r, v3 p) {
    const VspgMedium *m = &r->scene.medium;
    float sx = p.x * m->nx - .5f, sy = p.y * m->ny - .5f, sz = p.z * m->nz - .5f;
    int ix = (int)floorf(sx), iy = (int)floorf(sy), iz = (int)floorf(sz);
    float dx = sx - (float)ix, dy = sy - (float)iy, dz = sz - (float)iz;
#define LERPF(t, a, b) ((1 - (t)) * (a) + (t) * (b))
    float d00 = LERPF(dx, grid_at(r, ix, iy, iz), grid_at(r, ix + 1, iy, iz));
    float d10 = LERPF(dx, grid_at(r, ix, iy + 1, iz), grid_at(r, ix + 1, iy + 1, iz));
    float d01 = LERPF(dx, grid_at(r, ix, iy, iz + 1), grid_at(r, ix + 1, iy, iz + 1));
    float d11 = LERPF(dx, grid_at(r, ix, iy + 1, iz + 1), grid_at(r, ix + 1, iy + 1, iz + 1));
    float a = LERPF(dy, d00, d10), b = LERPF(dy, d01, d11);
    return LERPF(dz, a, b);
#undef LERPF
}
/* majorant grid construction: SampledGrid::MaxValue over each voxel's bounds
 * (media.cpp:262-269, containers.h:838-854) */
static void build_majorant_grid(OracleRenderer *r) {
    const VspgMedium *m = &r->scene.medium;
    for (int z = 0; z < MAJ_RES; ++z)
        for (int y = 0; y < MAJ_RES; ++y)
            for (int x = 0; x < MAJ_RES; ++x) {
                float b0[3] = {(float)x / MAJ_RES, (float)y / MAJ_RES, (float)z / MAJ_RES};
                float b1[3] = {(float)(x + 1) / MAJ_RES, (float)(y + 1) / MAJ_RES, (float)(z + 1) / MAJ_RES};
                int n[3] = {m->nx, m->ny, m->nz}, lo[3], hi[3];
                for (int k = 0; k < 3; ++k) {
                    int a = (int)floorf(b0[k] * n[k] - .5f);
                    int b = (int)floorf(b1[k] * n[k] - .5f) + 1;
                    lo[k] = a > 0 ? a : 0;
                    hi[k] = b < n[k] - 1 ? b : n[k] - 1;
                }
                float mx = grid_at(r, lo[0], lo[1], lo[2]);
                for (int zz = lo[2]; zz <= hi[2]; ++zz)
                    for (int yy = lo[1]; yy <= hi[1]; ++yy)
                        for (int xx = lo[0]; xx <= hi[0]; ++xx) {
                            float v = grid_at(r, xx, yy, zz);
                            mx = mx < v ? v : mx; /* std::max(maxValue, v) */
                        }
                r->majorant[x + MAJ_RES * (y + MAJ_RES * z)] = mx;
            }
}

/* Medium::SampleRay: HomogeneousMedium (media.h:263-269), GridMedium (media.h:347-362) */
static majiter_t medium_sample_ray(const OracleRenderer *r, v3 o, v3 d, float tMax) {
    majiter_t it;
    memset(&it, 0, sizeof it);
    const VspgMedium *m = &r->scene.medium;
    it.type = m->type;
    if (m->type == VSPG_MEDIUM_HOMOGENEOUS) {
        it.called = 0;
        it.seg.tMin = 0;
        it.seg.tMax = tMax;
        it.seg.sigma_maj = s_add(s_from(m->sigma_a), s_from(m->sigma_s));
    } else if (medium_is_grid_like(m->type)) { /* GridMedium::SampleRay (media.h:347-362) == NanoVDBMedium::SampleRay (:705-719) */
        const int MR = medium_maj_res(m->type);
        it.tMin = INFINITY; it.tMax = -INFINITY; /* default-constructed iterator: Next() returns nothing */
        /* ray = renderFromMedium.ApplyInverse(ray, &raytMax) with an identity transform
         * (transform.h:416-429, transform.cpp:263-303): the origin picks up the conservative
         * error bound gamma(3)*|o| and is pushed along d by dt */
        v3 ro;
        if (m->has_transform) ro = apply_inverse_ray(m->medium_from_render, o, d, &tMax, &d); /* d: the medium-space direction from here on */
        else ro = apply_inverse_identity(o, d, &tMax);
        float t0, t1;
        if (!bounds_intersect_p(m->bounds_min, m->bounds_max, ro, d, tMax, &t0, &t1)) return it;
        /* DDAMajorantIterator ctor (media.h:145-176) */
        it.tMin = t0; it.tMax = t1;
        it.maj = r->majorant;
        it.sigma_t = s_add(s_from(m->sigma_a), s_from(m->sigma_s));
        v3 diag = V3(m->bounds_max[0] - m->bounds_min[0], m->bounds_max[1] - m->bounds_min[1], m->bounds_max[2] - m->bounds_min[2]);
        v3 go = bounds_offset(m, ro);
        float gd[3] = {d.x / diag.x, d.y / diag.y, d.z / diag.z};
        float gi[3] = {go.x + gd[0] * t0, go.y + gd[1] * t0, go.z + gd[2] * t0};
        for (int axis = 0; axis < 3; ++axis) {
            float v = gi[axis] * MR; /* Clamp(float, 0, res-1) -> float, then converted to int */
            it.voxel[axis] = (int)(v < 0 ? 0.f : (v > (float)(MR - 1) ? (float)(MR - 1) : v));
            it.deltaT[axis] = 1 / (fabsf(gd[axis]) * MR);
            if (gd[axis] == -0.f) gd[axis] = 0.f;
            if (gd[axis] >= 0) {
                float nextVoxelPos = (float)(it.voxel[axis] + 1) / MR;
                it.nextCrossingT[axis] = t0 + (nextVoxelPos - gi[axis]) / gd[axis];
                it.step[axis] = 1;
                it.voxelLimit[axis] = MR;
            } else {
                float nextVoxelPos = (float)(it.voxel[axis]) / MR;
                it.nextCrossingT[axis] = t0 + (nextVoxelPos - gi[axis]) / gd[axis];
                it.step[axis] = -1;
                it.voxelLimit[axis] = -1;
            }
        }
    } else {
        it.called = 1;
    }
    return it;
}
/* Medium::SamplePoint: HomogeneousMedium (media.h:256-261), GridMedium (media.h:316-345) */
/* NanoVDBMedium over a dense copy of the grid.  worldToIndexF(p) = (p - grid_origin) * (1 / voxel_size);
 * nanovdb::SampleFromVoxels<Tree, 1, false>: ijk = floor(x), uvw = x - ijk, the 8 corner values (background 0
 * outside the index bounding box), lerp(a, b, w) = a + w (b - a) along z, then y, then x.  NanoVDB itself is
 * absent from the reference tree (submodule not vendored): PARITY UNPINNED for this fetch. */
static float nvdb_value_of(const OracleRenderer *r, const float *values, int i, int j, int k) {
    const VspgMedium *m = &r->scene.medium;
    int x = i - m->index_min[0], y = j - m->index_min[1], z = k - m->index_min[2];
    if (x < 0 || y < 0 || z < 0 || x >= m->nx || y >= m->ny || z >= m->nz) return 0.f;
    return values[((size_t)z * m->ny + y) * m->nx + x];
}
static float nvdb_value(const OracleRenderer *r, int i, int j, int k) { return nvdb_value_of(r, r->density, i, j, k); }
/* values: the density grid, or the temperature grid (same index bounding box and index-to-world map: include/vspg.h) */
static float nvdb_sample_of(const OracleRenderer *r, const float *values, v3 p) {
    const VspgMedium *m = &r->scene.medium;
    float inv[3] = {1.0f / m->voxel_size[0], 1.0f / m->voxel_size[1], 1.0f / m->voxel_size[2]};
    float x = (p.x - m->grid_origin[0]) * inv[0], y = (p.y - m->grid_origin[1]) * inv[1], z = (p.z - m->grid_origin[2]) * inv[2];
    float fx = floorf(x), fy = floorf(y), fz = floorf(z);
    int i = (int)fx, j = (int)fy, k = (int)fz;
    float u = x - fx, v = y - fy, w = z - fz;
#define NLERP(a, b, t) ((a) + (t) * ((b) - (a)))
#define NV(a, b, c) nvdb_value_of(r, values, a, b, c)
    float a00 = NLERP(NV(i, j, k), NV(i, j, k + 1), w);
    float a01 = NLERP(NV(i, j + 1, k), NV(i, j + 1, k + 1), w);
    float a10 = NLERP(NV(i + 1, j, k), NV(i + 1, j, k + 1), w);
    float a11 = NLERP(NV(i + 1, j + 1, k), NV(i + 1, j + 1, k + 1), w);
    float b0 = NLERP(a00, a01, v), b1 = NLERP(a10, a11, v);
    return NLERP(b0, b1, u);
#undef NV
#undef NLERP
}
static float nvdb_sample(const OracleRenderer *r, v3 p) { return nvdb_sample_of(r, r->density, p); }
/* NanoVDBMedium ctor, "Initialize majorantGrid" (media.cpp:600-671) */
static void build_majorant_grid_nvdb(OracleRenderer *r) {
    const VspgMedium *m = &r->scene.medium;
    const int R = MAJ_RES_NVDB;
    int imin[3] = {m->index_min[0], m->index_min[1], m->index_min[2]};
    int imax[3] = {m->index_min[0] + m->nx - 1, m->index_min[1] + m->ny - 1, m->index_min[2] + m->nz - 1};
    for (int z = 0; z < R; ++z)
        for (int y = 0; y < R; ++y)
            for (int x = 0; x < R; ++x) {
                int c[3] = {x, y, z}, lo[3], hi[3];
                for (int k = 0; k < 3; ++k) {
                    float t0 = (float)c[k] / R, t1 = (float)(c[k] + 1) / R;
                    float w0 = (1 - t0) * m->bounds_min[k] + t0 * m->bounds_max[k]; /* bounds.Lerp */
                    float w1 = (1 - t1) * m->bounds_min[k] + t1 * m->bounds_max[k];
                    double i0 = ((double)w0 - (double)m->grid_origin[k]) / (double)m->voxel_size[k]; /* worldToIndexF(Vec3R) */
                    double i1 = ((double)w1 - (double)m->grid_origin[k]) / (double)m->voxel_size[k];
                    float delta = 1.f; /* filter slop */
                    int a = (int)(i0 - delta), b = (int)(i1 + delta);
                    lo[k] = a > imin[k] ? a : imin[k];
                    hi[k] = b < imax[k] ? b : imax[k];
                }
                float mx = 0;
                for (int kk = lo[2]; kk <= hi[2]; ++kk)
                    for (int jj = lo[1]; jj <= hi[1]; ++jj)
                        for (int ii = lo[0]; ii <= hi[0]; ++ii) {
                            float v = nvdb_value(r, ii, jj, kk);
                            mx = mx < v ? v : mx;
                        }
                r->majorant[x + R * (y + R * z)] = (mx + m->density_offset) * m->majorant_scale;
            }
}
static medium_props_t medium_sample_point(const OracleRenderer *r, v3 p) {
    medium_props_t mp;
    const VspgMedium *m = &r->scene.medium;
    mp.sigma_a = s_from(m->sigma_a);
    mp.sigma_s = s_from(m->sigma_s);
    mp.Le = s_from(m->Le);
    mp.g = m->g;
    mp.bb_temp = 0.f; mp.bb_scale = 0.f;
    if (medium_is_grid_like(m->type)) {
        /* p = renderFromMedium.ApplyInverse(p) (media.h:322 / :693); identity: p unchanged */
        if (m->has_transform) p = apply_inverse_point(m->medium_from_render, p);
        float d;
        spec Le = S1(0.f);
        if (m->type == VSPG_MEDIUM_NANOVDB) { /* media.h:686-703 */
            d = nvdb_sample(r, p);
            d += m->density_offset;
            if (r->temperature) { /* NanoVDBMedium::Le (media.h:724-735), whatever IsEmissive() says (SamplePoint does not ask) */
                float temp = nvdb_sample_of(r, r->temperature, p);
                temp = (temp - m->temperature_offset) * m->temperature_scale;
                if (temp > 100.f) { mp.bb_temp = temp; mp.bb_scale = m->nvdb_le_scale; }
            }
        } else {
            v3 po = bounds_offset(m, p);
            d = grid_lookup(r, po);
            if (r->le_scale) { /* isEmissive (media.h:326-342): Le = scale * Le_spec.Sample(lambda) where the scale is positive */
                float scale = sgrid_lookup(r->le_scale, r->le_dim[0], r->le_dim[1], r->le_dim[2], po);
                if (scale > 0) {
                    if (r->temperature) { /* media.h:333-341 */
                        float temp = sgrid_lookup(r->temperature, m->nx, m->ny, m->nz, po);
                        temp = (temp - m->temperature_offset) * m->temperature_scale;
                        if (temp > 100.f) { mp.bb_temp = temp; mp.bb_scale = scale; }
                    } else
                        Le = s_scale(s_from(m->Le), scale); /* SampledSpectrum operator*(Float, s): s * a */
                }
            }
        }
        mp.sigma_a = s_scale(mp.sigma_a, d);
        mp.sigma_s = s_scale(mp.sigma_s, d);
        mp.Le = Le;
    }
    return mp;
}
/* mp.Le as the reference's SamplePoint(p, lambda) would have returned it (see medium_props_t) */
static spec medium_Le(const medium_props_t *mp, const float lambda[3]) {
    if (mp->bb_temp > 0) return s_scale(blackbody_sample(mp->bb_temp, lambda), mp->bb_scale);
    return mp->Le;
}
static int medium_is_homogeneous(const OracleRenderer *r) { return r->scene.medium.type == VSPG_MEDIUM_HOMOGENEOUS; }

/* ------------------------------------------------------------------------------------ */
/* a7: SampleT_maj (src/pbrt/media_sampleTMaj.h:49-117)                                   */
/* ------------------------------------------------------------------------------------ */
typedef int (*tmaj_cb_t)(void *ctx, v3 p, const medium_props_t *mp, spec sigma_maj, spec T_maj,
                         int activateNDS);

static spec sample_T_maj(const OracleRenderer *r, v3 ro, v3 rd, float tMax, float u, rng_t *rng,
                         int ch, tmaj_cb_t cb, void *cbctx) {
    tMax *= v_len(rd);
    rd = v_normalize(rd);
    majiter_t iter = medium_sample_ray(r, ro, rd, tMax);
    spec T_maj = S1(1.f);
    int done = 0;
    while (!done) {
        majseg_t seg;
        if (!majiter_next(&iter, &seg)) return T_maj;
        if (seg.sigma_maj.c[ch] == 0) {
            float dt = seg.tMax - seg.tMin;
            if (isinf(dt)) dt = FLT_MAX;
            T_maj = s_mul(T_maj, s_fast_exp(s_scale(seg.sigma_maj, -dt)));
            continue;
        }
        float tMin = seg.tMin;
        while (1) {
            float t = tMin + oracle_sample_exponential(u, seg.sigma_maj.c[ch]);
            u = rng_float(rng);
            if (t < seg.tMax) {
                T_maj = s_mul(T_maj, s_fast_exp(s_scale(seg.sigma_maj, -(t - tMin))));
                v3 p = v_add(ro, v_scale(rd, t));
                medium_props_t mp = medium_sample_point(r, p);
                if (!cb(cbctx, p, &mp, seg.sigma_maj, T_maj, 0)) {
                    done = 1;
                    break;
                }
                T_maj = S1(1.f);
                tMin = t;
            } else {
                float dt = seg.tMax - tMin;
                if (isinf(dt)) dt = FLT_MAX;
                T_maj = s_mul(T_maj, s_fast_exp(s_scale(seg.sigma_maj, -dt)));
                break;
            }
        }
    }
    return S1(1.f);
}

/* ------------------------------------------------------------------------------------ */
/* a9: SampleT_maj_Resampling (media_sampleTMaj.h:136-248)                                */
/* ------------------------------------------------------------------------------------ */
static spec sample_T_maj_resampling(const OracleRenderer *r, v3 ro, v3 rd, float tMax, float u,
                                    rng_t *rng, int ch, int guide, float vsp, float *vrc,
                                    float *majorantScale, tmaj_cb_t cb, void *cbctx) {
    tMax *= v_len(rd);
    rd = v_normalize(rd);
    majiter_t iter = medium_sample_ray(r, ro, rd, tMax);
    majiter_t pre = iter;
    float totalLength = 0.f;
    while (1) {
        majseg_t seg;
        if (!majiter_next(&pre, &seg)) break;
        if (seg.sigma_maj.c[ch] == 0) continue;
        totalLength += seg.sigma_maj.c[ch] * (seg.tMax - seg.tMin);
    }
    if (totalLength == 0.f) return S1(1.f);
    *majorantScale = 1.0f;
    *vrc = vsp;
    if (guide) {
        float minTotalLength = -logf(1 - vsp);
        if (minTotalLength > totalLength) {
            *majorantScale = minTotalLength / totalLength;
            totalLength = minTotalLength;
        }
        float expNegTotalLength = oracle_fast_exp(-totalLength);
        *vrc = vsp / (1 - expNegTotalLength);
    }
    spec T_maj = S1(1.f);
    int done = 0, count = 0;
    while (!done) {
        majseg_t seg;
        if (!majiter_next(&iter, &seg)) return T_maj;
        seg.sigma_maj = s_scale(seg.sigma_maj, *majorantScale);
        if (seg.sigma_maj.c[ch] == 0) {
            float dt = seg.tMax - seg.tMin;
            if (isinf(dt)) dt = FLT_MAX;
            T_maj = s_mul(T_maj, s_fast_exp(s_scale(seg.sigma_maj, -dt)));
            continue;
        }
        float tMin = seg.tMin;
        while (1) {
            count++;
            float t = tMin + oracle_sample_exponential(u, seg.sigma_maj.c[ch]);
            u = rng_float(rng);
            if (t < seg.tMax) {
                if (count > 10000) break;
                T_maj = s_mul(T_maj, s_fast_exp(s_scale(seg.sigma_maj, -(t - tMin))));
                v3 p = v_add(ro, v_scale(rd, t));
                medium_props_t mp = medium_sample_point(r, p);
                if (!cb(cbctx, p, &mp, seg.sigma_maj, T_maj, 0)) {
                    done = 1;
                    break;
                }
                T_maj = S1(1.f);
                tMin = t;
            } else {
                float dt = seg.tMax - tMin;
                if (isinf(dt)) dt = FLT_MAX;
                T_maj = s_mul(T_maj, s_fast_exp(s_scale(seg.sigma_maj, -dt)));
                break;
            }
        }
    }
    return S1(1.f);
}

/* ------------------------------------------------------------------------------------ */
/* a8: SampleT_maj_OpticalDepthSpace (media_sampleTMaj.h:269-491)                         */
/* ------------------------------------------------------------------------------------ */
static spec sample_T_maj_ods(const OracleRenderer *r, v3 ro, v3 rd, float tMax, float u, rng_t *rng,
                             int ch, int guide, float vsp, float vspMISRatio, int NDS,
                             spec *beta_factor, spec *r_u_factor, tmaj_cb_t cb, void *cbctx) {
    (void)beta_factor; /* never written by the reference */
    if (!guide || vspMISRatio == 0.f) return sample_T_maj(r, ro, rd, tMax, u, rng, ch, cb, cbctx);

    tMax *= v_len(rd);
    rd = v_normalize(rd);
    majiter_t iter = medium_sample_ray(r, ro, rd, tMax);
    majiter_t pre = iter;
    float t_v = 0.f;
    while (1) {
        majseg_t seg;
        if (!majiter_next(&pre, &seg)) break;
        if (isinf(seg.tMax)) return sample_T_maj(r, ro, rd, tMax, u, rng, ch, cb, cbctx); /* normalised ray, scaled tMax (:307) */
        if (seg.sigma_maj.c[ch] == 0) continue;
        t_v += seg.sigma_maj.c[ch] * (seg.tMax - seg.tMin);
    }
    if (t_v == 0.f) return S1(1.f);

    float OneMinusENegTv = 1.f - oracle_fast_exp(-t_v);
    float t_n = -1.f, t_n_current = -1.f;
    if (NDS) {
        if (vsp < 1 - oracle_fast_exp(-t_v))
            return sample_T_maj(r, ro, rd, tMax, u, rng, ch, cb, cbctx); /* (:327) */
        else {
            t_n = (float)(-log(1.0 - (double)(OneMinusENegTv / vsp)));
            t_n_current = t_n;
        }
    }
    spec T_maj = S1(1.f), tpScaleFactor = S1(1.f);
    float t_v_current = t_v;
    float remainingDist = 0;
    int deltaTracking = 0;
    if (u > vspMISRatio) {
        deltaTracking = 1;
        u = (u - vspMISRatio) / (1 - vspMISRatio);
    } else {
        u /= vspMISRatio;
    }
    int done = 0, count = 0, overTheEnd = 0;
    const float ScatterEpsilon = 1e-5;
    while (!done) {
        majseg_t seg;
        if (!majiter_next(&iter, &seg)) return T_maj;
        if (seg.sigma_maj.c[ch] == 0 || overTheEnd) {
            float dt = seg.tMax - seg.tMin;
            if (isinf(dt)) dt = FLT_MAX;
            T_maj = s_mul(T_maj, s_fast_exp(s_scale(seg.sigma_maj, -dt)));
            continue;
        }
        float tMin = seg.tMin;
        spec normalizedMaj = s_divf(seg.sigma_maj, seg.sigma_maj.c[ch]);

        if (remainingDist > 0) {
            tMin += remainingDist / seg.sigma_maj.c[ch];
            if (tMin > seg.tMax + ScatterEpsilon) {
                float dist = (seg.tMax - seg.tMin) * seg.sigma_maj.c[ch];
                t_v_current -= dist;
                t_n_current -= dist;
                remainingDist -= dist;
                T_maj = s_mul(T_maj, s_fast_exp(s_scale(seg.sigma_maj, -(seg.tMax - seg.tMin))));
                continue;
            }
            t_v_current -= remainingDist;
            t_n_current -= remainingDist;
            remainingDist = 0;
            T_maj = s_mul(T_maj, s_fast_exp(s_scale(seg.sigma_maj, -(tMin - seg.tMin))));
            v3 p = v_add(ro, v_scale(rd, tMin));
            medium_props_t mp = medium_sample_point(r, p);
            for (int i = 0; i < 3; ++i)
                r_u_factor->c[i] = vspMISRatio / tpScaleFactor.c[i] + (1 - vspMISRatio);
            if (!cb(cbctx, p, &mp, seg.sigma_maj, T_maj, 1)) break;
            T_maj = S1(1.f);
        }

        while (1) {
            count++;
            float dist = FLT_MAX;
            spec tpStep;
            if (NDS) {
                for (int i = 0; i < 3; ++i)
                    tpStep.c[i] = 1.0f - oracle_fast_exp(-t_n_current * normalizedMaj.c[i]);
                if (!deltaTracking) dist = (float)(-log(1.0 - (double)(u * tpStep.c[ch])));
            } else {
                for (int i = 0; i < 3; ++i)
                    tpStep.c[i] = (1.0f - oracle_fast_exp(-t_v_current * normalizedMaj.c[i])) / vsp;
                if (!deltaTracking) {
                    if (u < vsp) dist = (float)(-log(1.0 - (double)(u * tpStep.c[ch])));
                }
            }
            if (deltaTracking) dist = (float)(-log(1.0 - (double)u));

            int passThrough = (t_v_current - dist < ScatterEpsilon) || dist == 0;
            if (NDS || !passThrough) tpScaleFactor = s_mul(tpScaleFactor, tpStep);

            if (passThrough) {
                if (NDS) {
                    tpScaleFactor = s_divf(tpScaleFactor, 1.0f - oracle_fast_exp(-t_n + t_v));
                } else {
                    spec e;
                    for (int i = 0; i < 3; ++i)
                        e.c[i] = oracle_fast_exp(-t_v_current * normalizedMaj.c[i]) / (1 - vsp);
                    tpScaleFactor = s_mul(tpScaleFactor, e);
                }
                for (int i = 0; i < 3; ++i)
                    r_u_factor->c[i] = vspMISRatio / tpScaleFactor.c[i] + (1 - vspMISRatio);
                overTheEnd = 1;
                T_maj = s_mul(T_maj, s_fast_exp(s_scale(seg.sigma_maj, -(seg.tMax - tMin))));
                break;
            }

            float t = tMin + dist / seg.sigma_maj.c[ch];
            u = rng_float(rng);
            if (t <= seg.tMax + ScatterEpsilon) {
                if (count > 10000) break;
                t_v_current -= dist;
                t_n_current -= dist;
                remainingDist = 0;
                T_maj = s_mul(T_maj, s_fast_exp(s_scale(seg.sigma_maj, -(t - tMin))));
                v3 p = v_add(ro, v_scale(rd, t));
                medium_props_t mp = medium_sample_point(r, p);
                for (int i = 0; i < 3; ++i)
                    r_u_factor->c[i] = vspMISRatio / tpScaleFactor.c[i] + (1 - vspMISRatio);
                if (!cb(cbctx, p, &mp, seg.sigma_maj, T_maj, 1)) {
                    done = 1;
                    break;
                }
                T_maj = S1(1.f);
                tMin = t;
            } else {
                float dt = seg.tMax - tMin;
                if (isinf(dt)) dt = FLT_MAX;
                T_maj = s_mul(T_maj, s_fast_exp(s_scale(seg.sigma_maj, -dt)));
                float distWithinThisSeg = dt * seg.sigma_maj.c[ch];
                remainingDist = dist - distWithinThisSeg;
                t_v_current -= distWithinThisSeg;
                t_n_current -= distWithinThisSeg;
                break;
            }
        }
    }
    return S1(1.f);
}

/* ------------------------------------------------------------------------------------ */
/* surface scattering: DiffuseBxDF behind BSDF (bxdfs.h:31-80, bsdf.h:20-88)              */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    frame_t frame; /* Frame::FromXZ(Normalize(dpdus), ns) (vecmath.h:1862) */
    spec R;
    int has_lobes;
} bsdf_t;

static bsdf_t bsdf_make(const rquad_t *q) {
    bsdf_t b;
    b.frame.x = q->dpdu_n;
    b.frame.z = q->n;
    b.frame.y = v_cross(q->n, q->dpdu_n);
    b.R = q->Kd;
    b.has_lobes = q->has_bsdf_lobes;
    return b;
}
static bsdf_t bsdf_make_tri(const rtri_t *T) {
    bsdf_t b;
    b.frame.x = T->dpdu_n;
    b.frame.z = T->n;
    b.frame.y = v_cross(T->n, T->dpdu_n);
    b.R = T->Kd;
    b.has_lobes = T->has_bsdf_lobes;
    return b;
}
static bsdf_t bsdf_make_sphere(const rsphere_t *S, const isect_t *si) {
    bsdf_t b;
    b.frame.x = si->dpdu_n;
    b.frame.z = si->n;
    b.frame.y = v_cross(si->n, si->dpdu_n);
    b.R = S->Kd;
    b.has_lobes = S->has_bsdf_lobes;
    return b;
}
static spec bsdf_f(const bsdf_t *b, v3 woR, v3 wiR) {
    v3 wi = frame_to_local(&b->frame, wiR), wo = frame_to_local(&b->frame, woR);
    if (wo.z == 0) return S1(0.f);
    if (!(wi.z * wo.z > 0)) return S1(0.f); /* SameHemisphere */
    return s_scale(b->R, INV_PI_F);
}
static float bsdf_pdf(const bsdf_t *b, v3 woR, v3 wiR) {
    v3 wo = frame_to_local(&b->frame, woR), wi = frame_to_local(&b->frame, wiR);
    if (wo.z == 0) return 0;
    if (!b->has_lobes) return 0;
    if (!(wi.z * wo.z > 0)) return 0;
    return fabsf(wi.z) * INV_PI_F; /* CosineHemispherePDF(AbsCosTheta) */
}
/* returns 0 if no sample */
static int bsdf_sample_f(const bsdf_t *b, v3 woR, float uc, float u0, float u1, spec *f, v3 *wiR,
                         float *pdf) {
    (void)uc;
    v3 wo = frame_to_local(&b->frame, woR);
    if (wo.z == 0 || !b->has_lobes) return 0;
    v3 wi = sample_cosine_hemisphere(u0, u1);
    if (wo.z < 0) wi.z *= -1;
    *pdf = fabsf(wi.z) * INV_PI_F;
    *f = s_scale(b->R, INV_PI_F);
    if (!s_nonzero(*f) || *pdf == 0 || wi.z == 0) return 0;
    *wiR = frame_from_local(&b->frame, wi);
    return 1;
}

/* ------------------------------------------------------------------------------------ */
/* lights: DiffuseAreaLight on a rectangle, sampled uniformly by area                     */
/* (lights.cpp:796-820; shapes.cpp:1155-1212 + area->solid-angle branch :1267-1283)       */
/* ------------------------------------------------------------------------------------ */
static spec light_L(const rquad_t *q, v3 n, v3 w) { /* lights.h:492-511 */
    if (!q->two_sided && v_dot(n, w) < 0) return S1(0.f);
    return q->Le;
}
typedef struct { spec L; v3 wi; float pdf; p3i pLight; v3 nLight; } lightli_t;

static int light_sample_li(const rquad_t *q, v3 ctxp, float u0, float u1, lightli_t *ls) {
    /* BilinearPatch::Sample(u), rectangle branch: uv = u */
    v3 pu0 = v_lerp(u1, q->p00, q->p01), pu1 = v_lerp(u1, q->p10, q->p11);
    v3 p = v_lerp(u0, pu0, pu1);
    v3 dpdu = v_sub(pu1, pu0);
    v3 dpdv = v_sub(v_lerp(u0, q->p01, q->p11), v_lerp(u0, q->p00, q->p10));
    if (v_len2(dpdu) == 0 || v_len2(dpdv) == 0) return 0;
    v3 c = v_cross(dpdu, dpdv);
    v3 n = v_normalize(c);
    if (v_dot(n, q->n) < 0) n = v_neg(n); /* reverseOrientation flip */
    float pdf = 1 / v_len(c);
    p3i pint = p3i_from_err(p, q->perr); /* Interaction(Point3fi(p, pError), n, st) */
    p = p3i_mid(pint);                   /* ss->intr.p() */
    /* area -> solid angle (shapes.cpp:1271-1281) */
    v3 wi = v_sub(p, ctxp);
    if (v_len2(wi) == 0) return 0;
    wi = v_normalize(wi);
    v3 d = v_sub(ctxp, p);
    pdf /= v_absdot(n, v_neg(wi)) / v_len2(d);
    if (isinf(pdf)) return 0;
    /* DiffuseAreaLight::SampleLi */
    if (pdf == 0) return 0;
    spec Le = light_L(q, n, v_neg(wi));
    if (!s_nonzero(Le)) return 0;
    ls->L = Le; ls->wi = wi; ls->pdf = pdf; ls->pLight = pint; ls->nLight = n;
    return 1;
}
/* DiffuseAreaLight::PDF_Li -> BilinearPatch::PDF(ctx, wi), area branch (shapes.cpp:1329-1352) */
static float light_pdf_li(const rquad_t *q, const lsctx_t *ctx, v3 wi) {
    v3 o = offset_ray_origin(ctx->pi, ctx->n, wi);
    float t;
    v3 p;
    if (!quad_intersect(q, o, wi, INFINITY, &t, &p)) return 0;
    p = p3i_mid(p3i_from_err(p, q->perr));
    v3 d = v_sub(p3i_mid(ctx->pi), p);
    float pdf = (1 / q->area) * (v_len2(d) / v_absdot(q->n, v_neg(wi)));
    return isinf(pdf) ? 0 : pdf;
}


/* ------------------------------------------------------------------------------------ */
/* a14: guiding-cache query.  The wrapper logic (GuidedBSDF / GuidedPhaseFunction: MIS and */
/* RIS mixing, PDF, VolumeScatterProbability) restates src/pbrt/cpu/guiding.h:57-638.  What */
/* sits behind the OpenPGL calls it makes (Init, ApplyCosineProduct,                       */
/* ApplySingleLobeHenyeyGreensteinProduct, PDF, SamplePDF, IncomingRadiancePDF,            */
/* VolumeScatterProbability) is OpenPGL code absent from the reference tree: own design,   */
/* PARITY UNPINNED -- kd-tree leaf lookup, parallax-aware vMF mixture. */
/* Only +,-,*,/,sqrt, FastExp and the libm float functions are used, in one fixed order,   */
/* so the device build can reproduce it bit for bit.                                       */
/* ------------------------------------------------------------------------------------ */
#define GK VSPG_FIELD_LOBES
#define TWO_PI_F 6.28318530717958647692f
/* Round 3: the arithmetic behind the OpenPGL calls was re-laid for the device (own design, still PARITY UNPINNED; the wrapper
 * logic above it -- guiding.h's MIS / RIS flows -- is untouched).  Same model as before -- parallax-re-aimed vMF lobes, closed-form
 * product with one vMF, responsibility-weighted VSP -- with the evaluation organised so that a lobe costs 3 IEEE divisions
 * and 2 square roots at Init (was 8 and 2) and keeps 7 floats of state (was 9):
 *   raw_k      the region's lobe k re-aimed at the query point (normalised with ONE reciprocal),
 *   b_k        weight_k * vmf_norm(kappa_k)                  (per region, cached per field update: DField::aux on the device)
 *   product with (m2, k2):  s = raw_k kappa_k + m2 k2,  kp = |s|,  kc = clamp(kp),
 *   a_k        (b_k * vmf_norm(k2)) * FastExp((kp - kappa_k) - k2)        the product lobe's density weight, un-normalised
 *   wo_k       a_k * ((2 pi (1 - FastExp(-2 kc))) / kc)                     its mass (= a_k / vmf_norm(kc)),  sum = SUM wo_k
 *   c1_k, c2_k kappa_k / kp, k2 / kp:   mu_k . w == c1_k (raw_k . w) + c2_k (m2 . w)   -- the product's mean is never formed
 *   PDF(w)     = (SUM_k a_k FastExp(kc_k (mu_k . w - 1))) * (1 / sum)
 *   IncomingRadiancePDF(w) = SUM_k b_k FastExp(kappa_k (raw_k . w - 1));   VolumeScatterProbability(w) = SUM e_k vsp_k / SUM e_k
 * with e_k the terms of the IncomingRadiancePDF sum (so its denominator IS that PDF).  Without a product lobe (isotropic
 * phase function) a_k = b_k, wo_k = weight_k, c1 = 1, c2 = 0. */
typedef struct {
    int ok;            /* Init() succeeded */
    int field, region; /* which field / leaf */
    v3 p;              /* query position (parallax) */
    int n;             /* lobes */
    v3 m2;             /* the product lobe's axis (unused when c2 == 0) */
    float isum;        /* 1 / SUM wo */
    v3 raw[GK];
    float a[GK], kc[GK], c1[GK], c2[GK], wn[GK]; /* wn = wo * isum: the normalised masses (lobe selection) */
} gdist_t;

static float vmf_norm(float kappa) { /* kappa / (2 pi (1 - e^{-2 kappa})) */
    return kappa / (TWO_PI_F * (1 - oracle_fast_exp(-2 * kappa)));
}
static float vmf_eval(v3 mu, float kappa, v3 w) { return vmf_norm(kappa) * oracle_fast_exp(kappa * (v_dot(mu, w) - 1)); }
static float kappa_clamp(float k) { return k < 1e-2f ? 1e-2f : (k > 1e4f ? 1e4f : k); }

/* kd-tree descent to the leaf containing p */
static int field_lookup(const OracleRenderer *r, int f, v3 p) {
    if (!r->field[f].nodes || r->field[f].n_nodes <= 0) return -1;
    uint32_t node = 0;
    for (int depth = 0; depth < 64; ++depth) {
        VspgKdNode nd = r->field[f].nodes[node];
        uint32_t axis = nd.packed & 3u, idx = nd.packed >> 2;
        if (axis == 3u) return (int)idx < r->field[f].n_regions ? (int)idx : -1;
        float c = axis == 0 ? p.x : (axis == 1 ? p.y : p.z);
        node = idx + (c < nd.split ? 0u : 1u);
        if ((int)node >= r->field[f].n_nodes) return -1;
    }
    return -1;
}
/* parallax-shifted incident-radiance lobe k of a region seen from p */
static v3 lobe_dir(const VspgFieldRegion *R, int k, v3 p) {
    v3 mu = V3(R->mu[0][k], R->mu[1][k], R->mu[2][k]);
    float d = R->distance[k];
    if (!(d > 0) || isinf(d)) return mu;
    v3 t = v_add(v_sub(v3_from(R->pivot), p), v_scale(mu, d)); /* (pivot - p) + mu d: the lobe's source seen from p */
    float l2 = v_len2(t);
    if (!(l2 > 0)) return mu;
    float inv = 1.0f / sqrtf(l2);
    return v_scale(t, inv);
}
static float lobe_b(const VspgFieldRegion *R, int k) { return R->weight[k] * vmf_norm(kappa_clamp(R->kappa[k])); }
/* {Surface,Volume}SamplingDistribution::Init + Apply*Product */
static gdist_t gdist_init(const OracleRenderer *r, int f, v3 p, int have_product, v3 m2, float k2) {
    gdist_t d;
    memset(&d, 0, sizeof d);
    d.field = f; d.p = p; d.m2 = m2;
    d.region = field_lookup(r, f, p);
    if (d.region < 0) return d;
    const VspgFieldRegion *R = &r->field[f].regions[d.region];
    if (R->n_lobes <= 0) return d;
    d.ok = 1;
    d.n = R->n_lobes < GK ? R->n_lobes : GK;
    const float nk2 = have_product ? vmf_norm(k2) : 0.f;
    float wo[GK];
    float sum = 0;
    for (int k = 0; k < d.n; ++k) {
        d.raw[k] = lobe_dir(R, k, p);
        const float kr = kappa_clamp(R->kappa[k]), b = lobe_b(R, k);
        if (have_product) {
            v3 s = v_add(v_scale(d.raw[k], kr), v_scale(m2, k2));
            float kp = sqrtf(v_len2(s));
            float kpinv = kp > 1e-6f ? 1.0f / kp : 0.f; /* lobes cancel: c1 = c2 = 0, i.e. mu . w := 0 under kc = 0.01 -- nearly uniform */
            float kc = kappa_clamp(kp);
            float E = oracle_fast_exp(-2 * kc);
            d.a[k] = (b * nk2) * oracle_fast_exp((kp - kr) - k2);
            wo[k] = d.a[k] * ((TWO_PI_F * (1 - E)) / kc);
            d.kc[k] = kc; d.c1[k] = kr * kpinv; d.c2[k] = k2 * kpinv;
        } else {
            d.a[k] = b; wo[k] = R->weight[k]; d.kc[k] = kr; d.c1[k] = 1.f; d.c2[k] = 0.f;
        }
        sum += wo[k];
    }
    if (sum > 0 && !isinf(sum)) {
        d.isum = 1.0f / sum;
    } else { /* degenerate product: fall back to the incident-radiance mixture */
        d.isum = 1.f;
        for (int k = 0; k < d.n; ++k) { d.a[k] = lobe_b(R, k); wo[k] = R->weight[k]; d.kc[k] = kappa_clamp(R->kappa[k]); d.c1[k] = 1.f; d.c2[k] = 0.f; }
    }
    for (int k = 0; k < d.n; ++k) d.wn[k] = wo[k] * d.isum;
    return d;
}
#define COSINE_LOBE_KAPPA 2.18853f /* vMF fit of the clamped cosine (App. E.2) */
static gdist_t gdist_init_surface(const OracleRenderer *r, v3 p, v3 n) { return gdist_init(r, 0, p, 1, n, COSINE_LOBE_KAPPA); }
static gdist_t gdist_init_volume(const OracleRenderer *r, v3 p, v3 dir, float g) {
    /* gphase.init(&phase, p, ray.d, v) hands the ray's PROPAGATION direction to
     * ApplySingleLobeHenyeyGreensteinProduct(dir, meanCosine) (guiding.h:389-391): vMF with mean
     * cosine |g| around the HG peak, +dir for forward scattering (g > 0); isotropic -> no product */
    float ag = fabsf(g);
    if (ag < 1e-3f) return gdist_init(r, 1, p, 0, V3(0, 0, 1), 0);
    if (ag > 0.99f) ag = 0.99f;
    float kg = ag * (3 - ag * ag) / (1 - ag * ag);
    v3 axis = g > 0 ? dir : v_neg(dir);
    return gdist_init(r, 1, p, 1, v_normalize(axis), kg);
}
static float gdist_pdf(const gdist_t *d, v3 w) {
    const float x2 = v_dot(d->m2, w);
    float s = 0;
    for (int k = 0; k < d->n; ++k) {
        const float dp = d->c1[k] * v_dot(d->raw[k], w) + d->c2[k] * x2;
        s += d->a[k] * oracle_fast_exp(d->kc[k] * (dp - 1));
    }
    return s * d->isum;
}
/* incident-radiance mixture (before the product) */
static float gdist_incoming_pdf(const OracleRenderer *r, const gdist_t *d, v3 w) {
    const VspgFieldRegion *R = &r->field[d->field].regions[d->region];
    float s = 0;
    for (int k = 0; k < d->n; ++k) s += lobe_b(R, k) * oracle_fast_exp(kappa_clamp(R->kappa[k]) * (v_dot(d->raw[k], w) - 1));
    return s;
}
/* VolumeScatterProbability(wi): responsibility-weighted per-lobe estimate */
static float gdist_vsp(const OracleRenderer *r, int f, int region, v3 p, v3 w) {
    const VspgFieldRegion *R = &r->field[f].regions[region];
    int n = R->n_lobes < GK ? R->n_lobes : GK;
    float num = 0, den = 0;
    for (int k = 0; k < n; ++k) {
        float e = lobe_b(R, k) * oracle_fast_exp(kappa_clamp(R->kappa[k]) * (v_dot(lobe_dir(R, k, p), w) - 1));
        num += e * R->vsp[k];
        den += e;
    }
    if (!(den > 0)) return -1.f;
    return num / den;
}
/* SamplePDF(u2, &wi) */
static float gdist_sample(const gdist_t *d, float u0, float u1, v3 *wi) {
    /* lobe selection by CDF walk over the normalised masses, u0 rescaled inside the chosen lobe */
    int k = 0;
    float acc = 0;
    for (; k < d->n - 1; ++k) {
        if (u0 < acc + d->wn[k]) break;
        acc += d->wn[k];
    }
    float uw = d->wn[k] > 0 ? (u0 - acc) / d->wn[k] : 0.f;
    uw = uw < 0 ? 0 : (uw > ONE_MINUS_EPS ? ONE_MINUS_EPS : uw);
    float kap = d->kc[k];
    float W = 1 + logf(uw + (1 - uw) * oracle_fast_exp(-2 * kap)) / kap;
    W = clampf(W, -1, 1);
    float sinT = safe_sqrt(1 - W * W);
    float phi = TWO_PI_F * u1;
    frame_t fr;
    /* the product lobe's mean, (raw kappa + m2 k2) / kp; the bare lobe where there is no product or the two cancel */
    fr.z = d->c1[k] == 0.f && d->c2[k] == 0.f ? d->raw[k] : v_add(v_scale(d->raw[k], d->c1[k]), v_scale(d->m2, d->c2[k]));
    coordinate_system(fr.z, &fr.x, &fr.y);
    *wi = frame_from_local(&fr, V3(sinT * cosf(phi), sinT * sinf(phi), W));
    return gdist_pdf(d, *wi);
}

static void free_field(OracleRenderer *r, int f) {
    free(r->field[f].nodes); free(r->field[f].regions); free(r->rstats[f]);
    r->rstats[f] = NULL;
    memset(&r->field[f], 0, sizeof r->field[f]);
}
static int copy_field(OracleRenderer *r, int f, const VspgField *src) {
    free_field(r, f);
    if (!src || src->n_nodes <= 0 || src->n_regions <= 0) return 0;
    for (int i = 0; i < src->n_nodes; ++i) {
        uint32_t axis = src->nodes[i].packed & 3u, idx = src->nodes[i].packed >> 2;
        if (axis == 3u ? (int)idx >= src->n_regions : (int)idx + 1 >= src->n_nodes || (int)idx <= i) return VSPG_EINVAL;
    }
    r->field[f].n_nodes = src->n_nodes; r->field[f].n_regions = src->n_regions;
    r->field[f].nodes = (VspgKdNode *)malloc(sizeof(VspgKdNode) * src->n_nodes);
    r->field[f].regions = (VspgFieldRegion *)malloc(sizeof(VspgFieldRegion) * src->n_regions);
    memcpy(r->field[f].nodes, src->nodes, sizeof(VspgKdNode) * src->n_nodes);
    memcpy(r->field[f].regions, src->regions, sizeof(VspgFieldRegion) * src->n_regions);
    return 0;
}
int oracle_renderer_set_guiding_field(OracleRenderer *r, const VspgField *surface_field, const VspgField *volume_field) {
    r->training = 0; /* a loaded cache is not trained further (:117-122) */
    r->field_uploaded = 1;
    int rc = copy_field(r, 0, surface_field);
    if (rc) return rc;
    return copy_field(r, 1, volume_field);
}
int oracle_guiding_query_batch(OracleRenderer *r, int is_volume, float g, int n, const float *p, const float *n_or_wo,
                               const float *wi, const float *u, int32_t *out_ok, float *out_pdf, float *out_incoming_pdf,
                               float *out_vsp, float *out_ws, float *out_pdf_s) {
    for (int i = 0; i < n; ++i) {
        v3 pp = v3_from(p + 3 * i), a = v3_from(n_or_wo + 3 * i), w = v3_from(wi + 3 * i);
        gdist_t d = is_volume ? gdist_init_volume(r, pp, a, g) : gdist_init_surface(r, pp, a);
        out_ok[i] = d.ok;
        out_pdf[i] = out_incoming_pdf[i] = out_pdf_s[i] = 0; out_vsp[i] = -1;
        out_ws[3 * i] = out_ws[3 * i + 1] = out_ws[3 * i + 2] = 0;
        if (!d.ok) continue;
        out_pdf[i] = gdist_pdf(&d, w);
        out_incoming_pdf[i] = gdist_incoming_pdf(r, &d, w);
        out_vsp[i] = gdist_vsp(r, d.field, d.region, d.p, w);
        v3 ws;
        out_pdf_s[i] = gdist_sample(&d, u[2 * i], u[2 * i + 1], &ws);
        out_ws[3 * i] = ws.x; out_ws[3 * i + 1] = ws.y; out_ws[3 * i + 2] = ws.z;
    }
    return 0;
}


/* GuidedBSDF / GuidedPhaseFunction state that outlives a vertex (guiding.h:345-357, 625-637) */
typedef struct {
    gdist_t d;
    int useGuiding, useScatterGuiding;
} gwrap_t;
#define GUIDING_PROBABILITY 0.5f /* guiding.h:348, 628 */

/* ------------------------------------------------------------------------------------ */
/* Light samplers (round 3): PowerLightSampler (lightsamplers.h:63-98, lightsamplers.cpp:76-99) and BVHLightSampler    */
/* (lightsamplers.h:100-430, lightsamplers.cpp:108-262) -- "bvh" is the reference's default (:1318).  Built once per      */
/* renderer from the light list {emissive rectangles in rectangle order, then the infinite lights}; the node records keep  */
/* what CompactLightBounds' accessors return (the quantised box / cosines / octahedral axis, de-quantised once).          */
/* ------------------------------------------------------------------------------------ */
typedef struct { v3 bmin, bmax; float phi; v3 w; float cosTheta_o, cosTheta_e; int twoSided; } lightbounds_t; /* LightBounds (lights.h:104-135) */
typedef struct {
    v3 bmin, bmax, w;              /* CompactLightBounds::Bounds(allb), Vector3f(w) */
    float phi, cosTheta_o, cosTheta_e;
    int32_t twoSided;
    uint32_t child_or_light;       /* interior: second child (the first is node + 1); leaf: index into the light list */
    int32_t is_leaf;
} lbvh_node_t;
#define LBVH_MAX_LIGHTS (VSPG_MAX_QUADS + VSPG_MAX_INFINITE_LIGHTS)
struct lightsampler {
    int n_nodes, n_inf;                       /* nodes.size(), infiniteLights.size() */
    lbvh_node_t nodes[2 * LBVH_MAX_LIGHTS];
    int inf_light[VSPG_MAX_INFINITE_LIGHTS];  /* infiniteLights[k] as an index into the light list */
    uint32_t bit_trail[LBVH_MAX_LIGHTS];      /* lightToBitTrail; 0xffffffff: not in the tree (infinite, or phi == 0) */
    /* PowerLightSampler's AliasTable (util/sampling.cpp:563-646) */
    int n_alias;
    float alias_p[LBVH_MAX_LIGHTS], alias_q[LBVH_MAX_LIGHTS];
    int alias_i[LBVH_MAX_LIGHTS];
};

static float safe_asin_f(float x) { return asinf(clampf(x, -1, 1)); }
static float safe_acos_f(float x) { return acosf(clampf(x, -1, 1)); }
static float angle_between(v3 a, v3 b) { /* vecmath.h:972-977 */
    if (v_dot(a, b) < 0) return PI_F - 2 * safe_asin_f(v_len(v_add(a, b)) / 2);
    return 2 * safe_asin_f(v_len(v_sub(b, a)) / 2);
}
typedef struct { v3 w; float cosTheta; } dircone_t; /* DirectionCone (vecmath.h:1785-1808); cosTheta == INFINITY: empty */
static dircone_t dircone(v3 w, float c) { dircone_t d; d.w = v_normalize(w); d.cosTheta = c; return d; }
static dircone_t dircone_union(dircone_t a, dircone_t b) { /* vecmath.cpp:56-83 */
    if (isinf(a.cosTheta)) return b;
    if (isinf(b.cosTheta)) return a;
    float theta_a = safe_acos_f(a.cosTheta), theta_b = safe_acos_f(b.cosTheta);
    float theta_d = angle_between(a.w, b.w);
    if (fminf(theta_d + theta_b, PI_F) <= theta_a) return a;
    if (fminf(theta_d + theta_a, PI_F) <= theta_b) return b;
    float theta_o = (theta_a + theta_d + theta_b) / 2;
    if (theta_o >= PI_F) return dircone(V3(0, 0, 1), -1);
    float theta_r = theta_o - theta_a;
    v3 wr = v_cross(a.w, b.w);
    if (v_len2(wr) == 0) return dircone(V3(0, 0, 1), -1);
    /* Rotate(Degrees(theta_r), wr)(a.w) (transform.h:220-247): sin / cos of Radians(Degrees(theta_r)) */
    float deg = (180 / PI_F) * theta_r, rad = (PI_F / 180) * deg;
    float sinT = sinf(rad), cosT = cosf(rad);
    v3 ax = v_normalize(wr);
    float m[3][3];
    m[0][0] = ax.x * ax.x + (1 - ax.x * ax.x) * cosT; m[0][1] = ax.x * ax.y * (1 - cosT) - ax.z * sinT; m[0][2] = ax.x * ax.z * (1 - cosT) + ax.y * sinT;
    m[1][0] = ax.x * ax.y * (1 - cosT) + ax.z * sinT; m[1][1] = ax.y * ax.y + (1 - ax.y * ax.y) * cosT; m[1][2] = ax.y * ax.z * (1 - cosT) - ax.x * sinT;
    m[2][0] = ax.x * ax.z * (1 - cosT) - ax.y * sinT; m[2][1] = ax.y * ax.z * (1 - cosT) + ax.x * sinT; m[2][2] = ax.z * ax.z + (1 - ax.z * ax.z) * cosT;
    v3 w = V3(m[0][0] * a.w.x + m[0][1] * a.w.y + m[0][2] * a.w.z, m[1][0] * a.w.x + m[1][1] * a.w.y + m[1][2] * a.w.z,
              m[2][0] * a.w.x + m[2][1] * a.w.y + m[2][2] * a.w.z); /* Transform::operator()(Vector3f), transform.h:351-356 */
    return dircone(w, cosf(theta_o));
}
static lightbounds_t lb_empty(void) { lightbounds_t b; memset(&b, 0, sizeof b); b.bmin = V3(INFINITY, INFINITY, INFINITY); b.bmax = V3(-INFINITY, -INFINITY, -INFINITY); return b; }
static v3 v_min3(v3 a, v3 b) { return V3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
static v3 v_max3(v3 a, v3 b) { return V3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }
static lightbounds_t lb_union(lightbounds_t a, lightbounds_t b) { /* lights.h:137-153 */
    if (a.phi == 0) return b;
    if (b.phi == 0) return a;
    dircone_t ca, cb; ca.w = a.w; ca.cosTheta = a.cosTheta_o; cb.w = b.w; cb.cosTheta = b.cosTheta_o;
    ca.w = v_normalize(ca.w); cb.w = v_normalize(cb.w); /* DirectionCone(w, cosTheta) normalises */
    dircone_t cone = dircone_union(ca, cb);
    lightbounds_t r;
    r.bmin = v_min3(a.bmin, b.bmin); r.bmax = v_max3(a.bmax, b.bmax);
    r.w = v_normalize(cone.w);
    r.phi = a.phi + b.phi;
    r.cosTheta_o = cone.cosTheta;
    r.cosTheta_e = fminf(a.cosTheta_e, b.cosTheta_e);
    r.twoSided = a.twoSided | b.twoSided;
    return r;
}
static v3 lb_centroid(const lightbounds_t *b) { return v_scale(v_add(b->bmin, b->bmax), 0.5f); } /* (pMin + pMax) / 2 */
static float vcomp(v3 v, int d) { return d == 0 ? v.x : (d == 1 ? v.y : v.z); }
/* CompactLightBounds(lb, allb) and back through its accessors (lightsamplers.h:100-142, 212-242; vecmath.h:1733-1782) */
static float quantize_bounds(float c, float mn, float mx) { return mn == mx ? 0.f : 65535.f * clampf((c - mn) / (mx - mn), 0, 1); }
static float lerpf(float t, float a, float b) { return (1 - t) * a + t * b; }
static uint16_t oct_encode(float f) { return (uint16_t)roundf(clampf((f + 1) / 2, 0, 1) * 65535.f); }
static void compact_node(lbvh_node_t *nd, const lightbounds_t *lb, v3 amin, v3 amax) {
    /* OctahedralVector(Normalize(lb.w)) -> Vector3f */
    v3 v = v_normalize(lb->w);
    float l1 = fabsf(v.x) + fabsf(v.y) + fabsf(v.z);
    v = V3(v.x / l1, v.y / l1, v.z / l1);
    uint16_t ox, oy;
    if (v.z >= 0) { ox = oct_encode(v.x); oy = oct_encode(v.y); }
    else { ox = oct_encode((1 - fabsf(v.y)) * copysignf(1.f, v.x)); oy = oct_encode((1 - fabsf(v.x)) * copysignf(1.f, v.y)); }
    v3 d;
    d.x = -1 + 2 * (ox / 65535.f);
    d.y = -1 + 2 * (oy / 65535.f);
    d.z = 1 - (fabsf(d.x) + fabsf(d.y));
    if (d.z < 0) { float xo = d.x; d.x = (1 - fabsf(d.y)) * copysignf(1.f, xo); d.y = (1 - fabsf(xo)) * copysignf(1.f, d.y); }
    nd->w = v_normalize(d);
    nd->phi = lb->phi;
    unsigned qo = (unsigned)floorf(32767.f * ((lb->cosTheta_o + 1) / 2)), qe = (unsigned)floorf(32767.f * ((lb->cosTheta_e + 1) / 2));
    nd->cosTheta_o = 2 * (qo / 32767.f) - 1;
    nd->cosTheta_e = 2 * (qe / 32767.f) - 1;
    nd->twoSided = lb->twoSided;
    float lo[3], hi[3];
    for (int c = 0; c < 3; ++c) {
        uint16_t q0 = (uint16_t)floorf(quantize_bounds(vcomp(lb->bmin, c), vcomp(amin, c), vcomp(amax, c)));
        uint16_t q1 = (uint16_t)ceilf(quantize_bounds(vcomp(lb->bmax, c), vcomp(amin, c), vcomp(amax, c)));
        lo[c] = lerpf(q0 / 65535.f, vcomp(amin, c), vcomp(amax, c));
        hi[c] = lerpf(q1 / 65535.f, vcomp(amin, c), vcomp(amax, c));
    }
    nd->bmin = V3(lo[0], lo[1], lo[2]); nd->bmax = V3(hi[0], hi[1], hi[2]);
}
static float lb_evaluate_cost(const lightbounds_t *b, v3 bdiag, int dim) { /* lightsamplers.h:398-411 */
    float theta_o = acosf(b->cosTheta_o), theta_e = acosf(b->cosTheta_e);
    float theta_w = fminf(theta_o + theta_e, PI_F);
    float sinTheta_o = safe_sqrt(1 - sqr(b->cosTheta_o));
    float M_omega = 2 * PI_F * (1 - b->cosTheta_o) +
                    PI_F / 2 * (2 * theta_w * sinTheta_o - cosf(theta_o - 2 * theta_w) - 2 * theta_o * sinTheta_o + b->cosTheta_o);
    float Kr = fmaxf(bdiag.x, fmaxf(bdiag.y, bdiag.z)) / vcomp(bdiag, dim);
    v3 d = v_sub(b->bmax, b->bmin);
    float area = 2 * (d.x * d.y + d.x * d.z + d.y * d.z); /* Bounds3::SurfaceArea */
    return b->phi * M_omega * Kr * area;
}
typedef struct { int light; lightbounds_t lb; } lbvh_item_t;
static int lbvh_bucket(const lightbounds_t *lb, v3 cmin, v3 cmax, int dim) { /* nBuckets * centroidBounds.Offset(pc)[dim], 12 buckets */
    float pc = vcomp(lb_centroid(lb), dim), mn = vcomp(cmin, dim), mx = vcomp(cmax, dim);
    float o = pc - mn;
    if (mx > mn) o /= mx - mn;
    int b = (int)(12 * o);
    return b == 12 ? 11 : b;
}
static int lbvh_build(lightsampler_t *ls, lbvh_item_t *it, int start, int end, uint32_t bitTrail, int depth, v3 amin, v3 amax, lightbounds_t *out) {
    if (end - start == 1) {
        int nodeIndex = ls->n_nodes++;
        compact_node(&ls->nodes[nodeIndex], &it[start].lb, amin, amax);
        ls->nodes[nodeIndex].child_or_light = (uint32_t)it[start].light;
        ls->nodes[nodeIndex].is_leaf = 1;
        ls->bit_trail[it[start].light] = bitTrail;
        *out = it[start].lb;
        return nodeIndex;
    }
    v3 bmin = V3(INFINITY, INFINITY, INFINITY), bmax = V3(-INFINITY, -INFINITY, -INFINITY), cmin = bmin, cmax = bmax;
    for (int i = start; i < end; ++i) {
        bmin = v_min3(bmin, it[i].lb.bmin); bmax = v_max3(bmax, it[i].lb.bmax);
        v3 c = lb_centroid(&it[i].lb);
        cmin = v_min3(cmin, c); cmax = v_max3(cmax, c);
    }
    float minCost = INFINITY;
    int minCostSplitBucket = -1, minCostSplitDim = -1;
    const v3 bdiag = v_sub(bmax, bmin);
    for (int dim = 0; dim < 3; ++dim) {
        if (vcomp(cmax, dim) == vcomp(cmin, dim)) continue;
        lightbounds_t bucket[12];
        for (int b = 0; b < 12; ++b) bucket[b] = lb_empty();
        for (int i = start; i < end; ++i) {
            int b = lbvh_bucket(&it[i].lb, cmin, cmax, dim);
            bucket[b] = lb_union(bucket[b], it[i].lb);
        }
        float cost[11];
        for (int i = 0; i < 11; ++i) {
            lightbounds_t b0 = lb_empty(), b1 = lb_empty();
            for (int j = 0; j <= i; ++j) b0 = lb_union(b0, bucket[j]);
            for (int j = i + 1; j < 12; ++j) b1 = lb_union(b1, bucket[j]);
            cost[i] = lb_evaluate_cost(&b0, bdiag, dim) + lb_evaluate_cost(&b1, bdiag, dim);
        }
        for (int i = 1; i < 11; ++i)
            if (cost[i] > 0 && cost[i] < minCost) { minCost = cost[i]; minCostSplitBucket = i; minCostSplitDim = dim; }
    }
    int mid;
    if (minCostSplitDim == -1) mid = (start + end) / 2;
    else {
        /* std::partition (libstdc++, forward-iterator form is not used for pointers: the bidirectional algorithm) */
        int first = start, last = end;
        while (1) {
            while (1) {
                if (first == last) goto done;
                if (lbvh_bucket(&it[first].lb, cmin, cmax, minCostSplitDim) <= minCostSplitBucket) ++first; else break;
            }
            --last;
            while (1) {
                if (first == last) goto done;
                if (!(lbvh_bucket(&it[last].lb, cmin, cmax, minCostSplitDim) <= minCostSplitBucket)) --last; else break;
            }
            lbvh_item_t t = it[first]; it[first] = it[last]; it[last] = t;
            ++first;
        }
    done:
        mid = first;
        if (mid == start || mid == end) mid = (start + end) / 2;
    }
    int nodeIndex = ls->n_nodes++;
    lightbounds_t l0, l1;
    lbvh_build(ls, it, start, mid, bitTrail, depth + 1, amin, amax, &l0);
    int child1 = lbvh_build(ls, it, mid, end, bitTrail | (1u << depth), depth + 1, amin, amax, &l1);
    lightbounds_t lb = lb_union(l0, l1);
    compact_node(&ls->nodes[nodeIndex], &lb, amin, amax);
    ls->nodes[nodeIndex].child_or_light = (uint32_t)child1;
    ls->nodes[nodeIndex].is_leaf = 0;
    *out = lb;
    return nodeIndex;
}
/* DiffuseAreaLight::Bounds on a rectangle (lights.cpp:845-864; BilinearPatch::Bounds / NormalBounds, shapes.cpp:1070-1126) */
static lightbounds_t quad_light_bounds(const rquad_t *q, int reverse_orientation) {
    lightbounds_t lb;
    lb.bmin = v_min3(v_min3(q->p00, q->p01), v_min3(q->p10, q->p11));
    lb.bmax = v_max3(v_max3(q->p00, q->p01), v_max3(q->p10, q->p11));
    v3 n00 = v_normalize(v_cross(v_sub(q->p10, q->p00), v_sub(q->p01, q->p00)));
    v3 n10 = v_normalize(v_cross(v_sub(q->p11, q->p10), v_sub(q->p00, q->p10)));
    v3 n01 = v_normalize(v_cross(v_sub(q->p00, q->p01), v_sub(q->p11, q->p01)));
    v3 n11 = v_normalize(v_cross(v_sub(q->p01, q->p11), v_sub(q->p10, q->p11)));
    if (reverse_orientation) { n00 = v_neg(n00); n10 = v_neg(n10); n01 = v_neg(n01); n11 = v_neg(n11); }
    v3 n = v_normalize(v_add(v_add(n00, n10), v_add(n01, n11)));
    float cosTheta = fminf(fminf(v_dot(n, n00), v_dot(n, n01)), fminf(v_dot(n, n10), v_dot(n, n11)));
    dircone_t nb = dircone(n, clampf(cosTheta, -1, 1));
    float phi = fmaxf(q->Le.c[0], fmaxf(q->Le.c[1], q->Le.c[2])); /* Lemit->MaxValue(), RGB rendering mode (spectrum.h:772-779) */
    phi *= 1.f * q->area * PI_F;                                     /* scale * area * Pi */
    lb.w = v_normalize(nb.w);
    lb.phi = phi;
    lb.cosTheta_o = nb.cosTheta;
    lb.cosTheta_e = cosf(PI_F / 2);
    lb.twoSided = q->two_sided;
    return lb;
}
static void lightsampler_build(OracleRenderer *r, lightsampler_t *ls) {
    memset(ls, 0, sizeof *ls);
    const int n_all = r->n_lights + r->n_inf;
    for (int i = 0; i < LBVH_MAX_LIGHTS; ++i) ls->bit_trail[i] = 0xffffffffu;
    /* BVHLightSampler (lightsamplers.cpp:108-136) */
    lbvh_item_t items[LBVH_MAX_LIGHTS];
    int n_items = 0;
    v3 amin = V3(INFINITY, INFINITY, INFINITY), amax = V3(-INFINITY, -INFINITY, -INFINITY);
    for (int i = 0; i < n_all; ++i) {
        if (i >= r->n_lights) { ls->inf_light[ls->n_inf++] = i; continue; } /* Bounds() == {} */
        const int qi = r->light_quads[i];
        lightbounds_t lb = quad_light_bounds(&r->quads[qi], r->scene.quads[qi].reverse_orientation);
        if (lb.phi > 0) {
            items[n_items].light = i; items[n_items].lb = lb; n_items++;
            amin = v_min3(amin, lb.bmin); amax = v_max3(amax, lb.bmax);
        }
    }
    if (n_items > 0) { lightbounds_t root; lbvh_build(ls, items, 0, n_items, 0, 0, amin, amax, &root); }
    /* PowerLightSampler (lightsamplers.cpp:76-99): phi = SafeDiv(light.Phi(lambda), lambda.PDF()).Average() for
     * lambda = SampledWavelengths::SampleVisible(0.5f) (spectrum.h:367-386, sampling.h SampleVisibleWavelengths / VisibleWavelengthsPDF) */
    if (n_all > 0) {
        float pdf[3], power[LBVH_MAX_LIGHTS], acc0 = 0.f;
        for (int i = 0; i < 3; ++i) {
            float up = 0.5f + (float)i / 3;
            if (up > 1) up -= 1;
            float lambda = 538 - 138.888889f * atanhf(0.85691062f - 1.82750197f * up);
            pdf[i] = lambda < 360 || lambda > 830 ? 0.f : 0.0039398042f / sqr(coshf(0.0072f * (lambda - 538)));
        }
        for (int i = 0; i < n_all; ++i) {
            spec L; float k;
            if (i < r->n_lights) {
                const rquad_t *q = &r->quads[r->light_quads[i]];
                L = q->Le; k = PI_F * (q->two_sided ? 2 : 1) * q->area; /* DiffuseAreaLight::Phi (lights.cpp:826-843) */
            } else {
                const VspgInfiniteLight *il = &r->scene.infinite_lights[i - r->n_lights];
                L = s_from(il->L);
                k = il->type == VSPG_LIGHT_DISTANT ? PI_F * sqr(r->scene_radius)              /* lights.cpp:251-253 */
                                                    : 4 * PI_F * PI_F * sqr(r->scene_radius);  /* lights.cpp:1038-1040 */
            }
            float s = 0.f;
            for (int c = 0; c < 3; ++c) s += pdf[c] != 0 ? (k * L.c[c]) / pdf[c] : 0.f;
            power[i] = s / 3;
            acc0 += power[i];
        }
        if (acc0 == 0.f) for (int i = 0; i < n_all; ++i) power[i] = 1.f;
        /* AliasTable (util/sampling.cpp:563-618) */
        double sum = 0.;
        for (int i = 0; i < n_all; ++i) sum += power[i];
        const float fsum = (float)sum;
        ls->n_alias = n_all;
        struct { float pHat; int index; } under[LBVH_MAX_LIGHTS], over[LBVH_MAX_LIGHTS];
        int nu = 0, no = 0;
        for (int i = 0; i < n_all; ++i) {
            ls->alias_p[i] = power[i] / fsum;
            float pHat = ls->alias_p[i] * n_all;
            if (pHat < 1) { under[nu].pHat = pHat; under[nu].index = i; nu++; } else { over[no].pHat = pHat; over[no].index = i; no++; }
        }
        while (nu > 0 && no > 0) {
            float upH = under[nu - 1].pHat, ovH = over[no - 1].pHat;
            int ui = under[nu - 1].index, oi = over[no - 1].index;
            nu--; no--;
            ls->alias_q[ui] = upH; ls->alias_i[ui] = oi;
            float pExcess = upH + ovH - 1;
            if (pExcess < 1) { under[nu].pHat = pExcess; under[nu].index = oi; nu++; } else { over[no].pHat = pExcess; over[no].index = oi; no++; }
        }
        while (no > 0) { no--; ls->alias_q[over[no].index] = 1; ls->alias_i[over[no].index] = -1; }
        while (nu > 0) { nu--; ls->alias_q[under[nu].index] = 1; ls->alias_i[under[nu].index] = -1; }
    }
}
/* CompactLightBounds::Importance (lightsamplers.h:144-206) on the de-quantised node */
static float lbvh_importance(const lbvh_node_t *nd, v3 p, v3 n) {
    v3 pc = v_scale(v_add(nd->bmin, nd->bmax), 0.5f);
    v3 dd = v_sub(p, pc);
    float d2 = v_len2(dd);
    d2 = fmaxf(d2, v_len(v_sub(nd->bmax, nd->bmin)) / 2);
    v3 wi = v_normalize(v_sub(p, pc));
    float cosTheta_w = v_dot(nd->w, wi);
    if (nd->twoSided) cosTheta_w = fabsf(cosTheta_w);
    float sinTheta_w = safe_sqrt(1 - sqr(cosTheta_w));
    /* BoundSubtendedDirections(bounds, p).cosTheta (vecmath.h:1815-1828) */
    float cosTheta_b;
    {
        int inside = pc.x >= nd->bmin.x && pc.x <= nd->bmax.x && pc.y >= nd->bmin.y && pc.y <= nd->bmax.y && pc.z >= nd->bmin.z && pc.z <= nd->bmax.z;
        float radius = inside ? v_len(v_sub(pc, nd->bmax)) : 0.f;
        if (v_len2(v_sub(p, pc)) < sqr(radius)) cosTheta_b = -1.f;
        else {
            float sin2ThetaMax = sqr(radius) / v_len2(v_sub(pc, p));
            cosTheta_b = safe_sqrt(1 - sin2ThetaMax);
        }
    }
    float sinTheta_b = safe_sqrt(1 - sqr(cosTheta_b));
    float cosTheta_o = nd->cosTheta_o, cosTheta_e = nd->cosTheta_e;
    float sinTheta_o = safe_sqrt(1 - sqr(cosTheta_o));
    /* cosSubClamped / sinSubClamped */
    float cosTheta_x = cosTheta_w > cosTheta_o ? 1.f : cosTheta_w * cosTheta_o + sinTheta_w * sinTheta_o;
    float sinTheta_x = cosTheta_w > cosTheta_o ? 0.f : sinTheta_w * cosTheta_o - cosTheta_w * sinTheta_o;
    float cosThetap = cosTheta_x > cosTheta_b ? 1.f : cosTheta_x * cosTheta_b + sinTheta_x * sinTheta_b;
    if (cosThetap <= cosTheta_e) return 0;
    float importance = nd->phi * cosThetap / d2;
    if (!(n.x == 0 && n.y == 0 && n.z == 0)) {
        float cosTheta_i = v_absdot(wi, n);
        float sinTheta_i = safe_sqrt(1 - sqr(cosTheta_i));
        float cosThetap_i = cosTheta_i > cosTheta_b ? 1.f : cosTheta_i * cosTheta_b + sinTheta_i * sinTheta_b;
        importance *= cosThetap_i;
    }
    return fmaxf(importance, 0);
}
/* lightSampler.Sample(ctx, u): light index into the light list and its pmf; 0 = no light */
static int lightsampler_sample(const OracleRenderer *r, const lightsampler_t *ls, v3 p, v3 n, float u, int *lightIndex, float *pmf) {
    const int n_all = r->n_lights + r->n_inf;
    if (r->prm.lightsampler == VSPG_LIGHTSAMPLER_UNIFORM) { /* lightsamplers.h:33-38 */
        if (n_all == 0) return 0;
        int li = (int)(u * (float)n_all);
        *lightIndex = li < n_all - 1 ? li : n_all - 1;
        *pmf = 1.f / (float)n_all;
        return 1;
    }
    if (r->prm.lightsampler == VSPG_LIGHTSAMPLER_POWER) { /* AliasTable::Sample (sampling.cpp:620-646) */
        if (!ls->n_alias) return 0;
        int offset = (int)(u * (float)ls->n_alias);
        if (offset > ls->n_alias - 1) offset = ls->n_alias - 1;
        float up = fminf(u * (float)ls->n_alias - (float)offset, ONE_MINUS_EPS);
        if (up < ls->alias_q[offset]) { *lightIndex = offset; *pmf = ls->alias_p[offset]; }
        else { *lightIndex = ls->alias_i[offset]; *pmf = ls->alias_p[ls->alias_i[offset]]; }
        return 1;
    }
    /* BVHLightSampler::Sample (lightsamplers.h:283-342) */
    const float pInfinite = (float)ls->n_inf / (float)(ls->n_inf + (ls->n_nodes == 0 ? 0 : 1));
    if (u < pInfinite) {
        u /= pInfinite;
        int index = (int)(u * (float)ls->n_inf);
        if (index > ls->n_inf - 1) index = ls->n_inf - 1;
        *pmf = pInfinite / (float)ls->n_inf;
        *lightIndex = ls->inf_light[index];
        return 1;
    }
    if (ls->n_nodes == 0) return 0;
    u = fminf((u - pInfinite) / (1 - pInfinite), ONE_MINUS_EPS);
    int nodeIndex = 0;
    float pm = 1 - pInfinite;
    while (1) {
        const lbvh_node_t *node = &ls->nodes[nodeIndex];
        if (!node->is_leaf) {
            float ci[2] = {lbvh_importance(&ls->nodes[nodeIndex + 1], p, n), lbvh_importance(&ls->nodes[node->child_or_light], p, n)};
            if (ci[0] == 0 && ci[1] == 0) return 0;
            /* SampleDiscrete(ci, u, &nodePMF, &u) (sampling.h:79-113) */
            float sumW = 0; sumW += ci[0]; sumW += ci[1];
            float up = u * sumW;
            if (up == sumW) up = next_float_down(up);
            int offset = 0; float sum = 0;
            while (sum + ci[offset] <= up) sum += ci[offset++];
            float nodePMF = ci[offset] / sumW;
            u = fminf((up - sum) / ci[offset], ONE_MINUS_EPS);
            pm *= nodePMF;
            nodeIndex = offset == 0 ? nodeIndex + 1 : (int)node->child_or_light;
        } else {
            if (nodeIndex > 0 || lbvh_importance(node, p, n) > 0) { *lightIndex = (int)node->child_or_light; *pmf = pm; return 1; }
            return 0;
        }
    }
}
/* lightSampler.PMF(ctx, light) */
static float lightsampler_pmf(const OracleRenderer *r, const lightsampler_t *ls, v3 p, v3 n, int lightIndex) {
    const int n_all = r->n_lights + r->n_inf;
    if (r->prm.lightsampler == VSPG_LIGHTSAMPLER_UNIFORM) return n_all ? 1.f / (float)n_all : 0.f;
    if (r->prm.lightsampler == VSPG_LIGHTSAMPLER_POWER) return ls->n_alias ? ls->alias_p[lightIndex] : 0.f;
    /* BVHLightSampler::PMF (lightsamplers.h:344-381) */
    if (ls->bit_trail[lightIndex] == 0xffffffffu) return 1.f / (float)(ls->n_inf + (ls->n_nodes == 0 ? 0 : 1));
    uint32_t bitTrail = ls->bit_trail[lightIndex];
    const float pInfinite = (float)ls->n_inf / (float)(ls->n_inf + (ls->n_nodes == 0 ? 0 : 1));
    float pm = 1 - pInfinite;
    int nodeIndex = 0;
    while (1) {
        const lbvh_node_t *node = &ls->nodes[nodeIndex];
        if (node->is_leaf) return pm;
        float ci[2] = {lbvh_importance(&ls->nodes[nodeIndex + 1], p, n), lbvh_importance(&ls->nodes[node->child_or_light], p, n)};
        pm *= ci[bitTrail & 1] / (ci[0] + ci[1]);
        nodeIndex = (bitTrail & 1) ? (int)node->child_or_light : nodeIndex + 1;
        bitTrail >>= 1;
    }
}

/* lightSampler.Sample(ctx, u) / PMF(ctx, light) for n contexts {p, ns} (tests/test_oracle_lightsamplers.py) */
int oracle_light_sample_batch(OracleRenderer *r, int n, const float *p, const float *ns, const float *u, int32_t *out_light, float *out_pmf) {
    for (int i = 0; i < n; ++i) {
        int li = -1; float pm = 0;
        if (!lightsampler_sample(r, r->lsamp, v3_from(p + 3 * i), v3_from(ns + 3 * i), u[i], &li, &pm)) { li = -1; pm = 0; }
        out_light[i] = li; out_pmf[i] = pm;
    }
    return 0;
}
int oracle_light_pmf_batch(OracleRenderer *r, int n, const float *p, const float *ns, const int32_t *light, float *out_pmf) {
    for (int i = 0; i < n; ++i) out_pmf[i] = lightsampler_pmf(r, r->lsamp, v3_from(p + 3 * i), v3_from(ns + 3 * i), light[i]);
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* a15: SampleLd (guidedvolpathvspgintegrator.cpp:1136-1252), guiding distributions        */
/* inactive (field untrained => GuidedBSDF/GuidedPhaseFunction fall through to the plain   */
/* BSDF / phase function, guiding.h:115-118,271-289,400-402,542-558)                       */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    const OracleRenderer *r;
    int ch;
    spec T_ray, r_l, r_u;
    rng_t *rng;
    path_counters_t *pc;
} shadow_cb_ctx_t;

static int shadow_cb(void *vctx, v3 p, const medium_props_t *mp, spec sigma_maj, spec T_maj, int nds) {
    (void)p; (void)nds;
    shadow_cb_ctx_t *c = (shadow_cb_ctx_t *)vctx;
    if (c->pc && medium_is_grid_like(c->r->scene.medium.type)) c->pc->shadow_queries++; /* VspgCounters.shadow_density_queries */
    spec sigma_n = s_clamp_zero(s_sub(s_sub(sigma_maj, mp->sigma_a), mp->sigma_s));
    float pdf = T_maj.c[c->ch] * sigma_maj.c[c->ch];
    c->T_ray = s_mul(c->T_ray, s_divf(s_mul(T_maj, sigma_n), pdf));
    c->r_l = s_mul(c->r_l, s_divf(s_mul(T_maj, sigma_maj), pdf));
    c->r_u = s_mul(c->r_u, s_divf(s_mul(T_maj, sigma_n), pdf));
    spec Tr = s_divf(c->T_ray, s_avg(s_add(c->r_l, c->r_u)));
    if (s_max(Tr) < 0.05f) {
        float q = 0.75f;
        if (rng_float(c->rng) < q)
            c->T_ray = S1(0.f);
        else
            c->T_ray = s_divf(c->T_ray, 1 - q);
    }
    if (!s_nonzero(c->T_ray)) return 0;
    return 1;
}

typedef struct {
    int is_surface;
    p3i pi;        /* interaction point with error bounds (exact for medium interactions) */
    v3 n;          /* geometric normal (0 for medium) */
    v3 wo;
    const bsdf_t *bsdf; /* surface */
    float g;            /* medium: HG */
    int iface;          /* surface: its medium transition (0: none) */
    int medium;         /* Interaction::medium: of the ray that hit a non-transition surface / of the medium interaction */
} intr_t;

static spec sample_Ld(const OracleRenderer *r, const intr_t *intr, const gwrap_t *gw, int ch, sampler_t *sampler,
                      spec r_p, path_counters_t *pc) {
    /* LightSampleContext */
    v3 ctxp = p3i_mid(intr->pi);
    if (intr->is_surface) {
        /* reflective-only BSDF: ctx.pi = intr.OffsetRayOrigin(intr.wo) (:1147-1149) */
        if (intr->bsdf->has_lobes) ctxp = offset_ray_origin(intr->pi, intr->n, intr->wo);
    }
    float u = sampler_get1d(sampler);
    /* lightSampler.Sample(ctx, u) (:1157): uniform / power / bvh; ctx.p() is the (offset) point, ctx.ns the surface normal (0 in a medium) */
    int lightIndex = 0;
    float lightPmf = 0;
    const int have_light = lightsampler_sample(r, r->lsamp, ctxp, intr->is_surface ? intr->n : V3(0, 0, 0), u, &lightIndex, &lightPmf);
    float ul0 = sampler_get1d(sampler), ul1 = sampler_get1d(sampler);
    if (!have_light) return S1(0.f);
    lightli_t ls;
    int delta_light = 0;
    if (lightIndex < r->n_lights) {
        const rquad_t *lq = &r->quads[r->light_quads[lightIndex]];
        if (!light_sample_li(lq, ctxp, ul0, ul1, &ls)) return S1(0.f);
    } else {
        const VspgInfiniteLight *il = &r->scene.infinite_lights[lightIndex - r->n_lights];
        /* UniformInfiniteLight::SampleLi(ctx, u, lambda, allowIncompletePDF = true) returns {} (lights.cpp:1019-1023) */
        if (il->type != VSPG_LIGHT_DISTANT) return S1(0.f);
        /* DistantLight::SampleLi (lights.h:320-327): wi = the light's direction, pdf 1, pLight = ctx.p() + wi * (2 * sceneRadius) */
        ls.L = s_from(il->L);
        if (!s_nonzero(ls.L)) return S1(0.f);
        ls.wi = v3_from(il->w_light);
        ls.pdf = 1;
        ls.pLight = p3i_exact(v_add(ctxp, v_scale(ls.wi, 2 * r->scene_radius)));
        ls.nLight = V3(0, 0, 0);
        delta_light = 1;
    }
    float p_l = lightPmf * ls.pdf;

    float scatterPDF;
    spec f_hat;
    v3 wo = intr->wo, wi = ls.wi;
    /* GuidedBSDF::PDF / GuidedPhaseFunction::PDF (guiding.h:271-289, 542-558): MIS and RIS types
     * both give 0.5*scatterPDF + 0.5*guidedPDF for guidingProbability 0.5 */
    if (intr->is_surface) {
        f_hat = s_scale(bsdf_f(intr->bsdf, wo, wi), v_absdot(wi, intr->n));
        float bsdfPDF = bsdf_pdf(intr->bsdf, wo, wi);
        if (gw && gw->useGuiding) bsdfPDF = ((1.0f - GUIDING_PROBABILITY) * bsdfPDF) + (GUIDING_PROBABILITY * gdist_pdf(&gw->d, wi));
        scatterPDF = 1.0f * bsdfPDF;
    } else {
        f_hat = S1(oracle_henyey_greenstein(v_dot(wo, wi), intr->g));
        float phasePDF = oracle_henyey_greenstein(v_dot(wo, wi), intr->g);
        if (gw && gw->useGuiding) phasePDF = ((1.0f - GUIDING_PROBABILITY) * phasePDF) + (GUIDING_PROBABILITY * gdist_pdf(&gw->d, wi));
        scatterPDF = 1.0f * phasePDF;
    }
    if (!s_nonzero(f_hat)) return S1(0.f);

    /* lightRay = intr.SpawnRayTo(ls->pLight) (interaction.h:111-115, ray.h:103-108) */
    v3 lo, ld;
    spawn_ray_to(intr->pi, intr->n, ls.pLight, ls.nLight, &lo, &ld);
    spec T_ray = S1(1.f), r_l = S1(1.f), r_u = S1(1.f);
    rng_t rng;
    rng_set_sequence2(&rng, oracle_hash_point3(lo.x, lo.y, lo.z), oracle_hash_point3(ld.x, ld.y, ld.z));
    if (pc) pc->shadow_rays++;
    int lmed = r->has_medium ? get_medium(intr->iface, intr->n, ld, intr->medium) : 0; /* r.medium = GetMedium(r.d) (interaction.h:111-115) */
    while (!(ld.x == 0 && ld.y == 0 && ld.z == 0)) { /* :1195-1244 */
        /* Intersect(lightRay, 1 - ShadowEpsilon): a hit with a material blocks (:1197-1200); an interface hit ends the segment.
         * Without boundaries every surface has a material: any hit decides, the closest need not be found. */
        isect_t si;
        memset(&si, 0, sizeof si);
        if (r->has_boundaries) {
            si = scene_intersect(r, lo, ld, 1 - SHADOW_EPS);
            if (si.hit && si.has_material) return S1(0.f);
        } else if (scene_intersect_any(r, lo, ld, 1 - SHADOW_EPS)) {
            return S1(0.f);
        }
        if (lmed) {
            float tMax = si.hit ? si.t : 1 - SHADOW_EPS;
            float us = rng_float(&rng);
            shadow_cb_ctx_t c;
            c.r = r; c.ch = ch; c.T_ray = T_ray; c.r_l = r_l; c.r_u = r_u; c.rng = &rng; c.pc = pc;
            spec T_maj = sample_T_maj(r, lo, ld, tMax, us, &rng, ch, shadow_cb, &c);
            T_ray = c.T_ray; r_l = c.r_l; r_u = c.r_u;
            T_ray = s_mul(T_ray, s_divf(T_maj, T_maj.c[ch]));
            r_l = s_mul(r_l, s_divf(T_maj, T_maj.c[ch]));
            r_u = s_mul(r_u, s_divf(T_maj, T_maj.c[ch]));
        }
        if (!s_nonzero(T_ray)) return S1(0.f);
        if (!si.hit) break;
        /* lightRay = si->intr.SpawnRayTo(ls->pLight) (:1243): from the interface, in the medium its far side holds */
        spawn_ray_to(isect_pi(&si), si.n, ls.pLight, ls.nLight, &lo, &ld);
        lmed = get_medium(si.iface, si.n, ld, lmed);
    }
    r_l = s_mul(r_l, s_scale(r_p, p_l));
    r_u = s_mul(r_u, s_scale(r_p, scatterPDF));
    DBG(intr->is_surface ? 32 : 64);
    if (delta_light) return s_divf(s_mul(s_mul(f_hat, T_ray), ls.L), s_avg(r_l)); /* IsDeltaLight (:1248-1249) */
    return s_divf(s_mul(s_mul(f_hat, T_ray), ls.L), s_avg(s_add(r_l, r_u)));
}

/* StandardThroughputBasedRussianRoulette lives in OpenPGL (absent; SURVEY.md App. A.5
 * "[unverified]"): restated as pbrt's own rule q = max(0, 1 - maxComponent)
 * (cf. the commented-out rrBeta code at guidedvolpathvspgintegrator.cpp:594-596). UNPINNED. */
/* openpgl::cpp::util::GuidedRussianRoulette(throughput, adjointEstimate, pixelContributionEstimate, minSurvival) is
 * OpenPGL code (absent from the reference tree).  Own definition in the spirit of adjoint-driven RR: the survival
 * probability is the largest per-channel ratio of the path's expected contribution (throughput x adjoint) to the pixel's
 * contribution estimate, clamped to [minSurvival, 1]; channels without an estimate do not vote, none -> 1.  Unpinned. */
static float guided_russian_roulette(spec throughput, spec adjoint, spec reference, float minSurvival) {
    float s = 0.f;
    int any = 0;
    for (int k = 0; k < 3; ++k)
        if (reference.c[k] > 0) {
            float q = throughput.c[k] * adjoint.c[k] / reference.c[k];
            s = q > s ? q : s;
            any = 1;
        }
    if (!any) return 1.f;
    return fminf(1.f, fmaxf(minSurvival, s));
}
static float standard_throughput_rr(spec w) {
    float m = s_max(w);
    return fminf(1.f, fmaxf(0.f, m));
}

/* ------------------------------------------------------------------------------------ */
/* a10-a13: SampleDistance (guidedvolpathvspgintegrator.cpp:637-1096)                     */
/* ------------------------------------------------------------------------------------ */
/* ------------------------------------------------------------------------------------ */
/* a18: path-segment recording -- the guiding_* hooks of src/pbrt/cpu/guiding.h:682-832.   */
/* A segment is openpgl::cpp::PathSegment reduced to what PropagateSamples reads.          */
/* ------------------------------------------------------------------------------------ */
#define TRAIN_MAX_SEG 64 /* records a path keeps: min(2 * maxdepth, 64); beyond it NextSegment() == nullptr (vspg_train.h) */
typedef struct {
    v3 p, wi;
    int has_wi, volume, is_delta;
    float pdf, mi_weight, rr;
    spec scattering_weight, transmittance_weight, direct, scattered;
} pathseg_t;
typedef struct {
    int n, cur; /* cur: the reference's pathSegmentData pointer, -1 = nullptr */
    int cap;    /* pss->Reserve(maxDepth >= 1 ? maxDepth * 2 : 30) (:135-138): NextSegment() returns nullptr beyond it */
    pathseg_t seg[TRAIN_MAX_SEG];
} pathrec_t;
static int rec_capacity(int maxdepth) {
    const int c = maxdepth >= 1 ? maxdepth * 2 : 30;
    return c < TRAIN_MAX_SEG ? c : TRAIN_MAX_SEG;
}
static spec s_max0(spec a) { return s_clamp_zero(a); } /* std::max(0.f, .) per component */
/* guiding_newSurfacePathSegment / guiding_newVolumePathSegment (:682-732) */
static void rec_new_segment(pathrec_t *rec, v3 p, int volume) {
    if (!rec) return;
    if (rec->n >= rec->cap) { rec->cur = -1; return; } /* NextSegment() == nullptr */
    pathseg_t *g = &rec->seg[rec->n];
    memset(g, 0, sizeof *g);
    g->p = p; g->volume = volume;
    g->scattered = S1(0.f); g->direct = S1(0.f); g->transmittance_weight = S1(1.f);
    g->scattering_weight = S1(0.f); g->mi_weight = 1.f; g->rr = 1.f;
    rec->cur = rec->n++;
}
static void rec_add_transmittance_weight(pathrec_t *rec, spec T) { /* :754-764 */
    if (rec && rec->cur >= 0) rec->seg[rec->cur].transmittance_weight = s_max0(T);
}
static void rec_add_surface_emission(pathrec_t *rec, spec Le, float w) { /* :744-752 */
    if (rec && rec->cur >= 0) { rec->seg[rec->cur].direct = s_max0(Le); rec->seg[rec->cur].mi_weight = w; }
}
static void rec_add_scattered_direct_light(pathrec_t *rec, spec Ld) { /* :734-742 */
    if (rec && rec->cur >= 0) rec->seg[rec->cur].scattered = s_add(rec->seg[rec->cur].scattered, s_max0(Ld));
}
/* guiding_addInfiniteLightEmission (guiding.h:759-784): a NEW segment at ray.o + guidingInfiniteLightDistance * ray.d
 * (integrators.h:608: 1e6f) carrying the light's emission and its MIS weight -- one per infinite light */
#define GUIDING_INFINITE_LIGHT_DISTANCE 1e6f
static void rec_new_segment(pathrec_t *rec, v3 p, int volume);
static void rec_add_infinite_light_emission(pathrec_t *rec, v3 o, v3 d, spec Le, float misWeight) {
    if (!rec) return;
    rec_new_segment(rec, v_add(o, v_scale(d, GUIDING_INFINITE_LIGHT_DISTANCE)), 0);
    if (rec->cur >= 0) { rec->seg[rec->cur].direct = s_max0(Le); rec->seg[rec->cur].mi_weight = misWeight; }
}
/* guiding_addSurfaceData / guiding_addVolumeData (:791-832) */
static void rec_add_scatter_data(pathrec_t *rec, int volume, spec weight, v3 wi, float pdf, float roughness, float survivalProb) {
    if (!rec || rec->cur < 0) return;
    pathseg_t *g = &rec->seg[rec->cur];
    g->transmittance_weight = S1(1.f);
    g->volume = volume;
    g->wi = wi; g->has_wi = 1;
    g->pdf = pdf;
    g->scattering_weight = s_max0(weight);
    g->is_delta = roughness < 0.001f;
    g->rr = survivalProb;
}

typedef struct {
    const OracleRenderer *r;
    int ch;
    const float *lambda;         /* the path's SampledWavelengths (temperature-grid emission only) */
    int px, py;                  /* pPixel */
    int guideRR;                 /* :276-285 */
    spec pixelContributionEstimate;
    sampler_t *sampler;
    rng_t *rng;
    pathrec_t *rec;              /* a18 recorder, NULL when not training */
    spec *transmittanceWeight;   /* :317, reset per path-loop iteration */
    /* path state by reference */
    v3 *ray_o, *ray_d;
    int *depth;
    spec *L, *beta, *r_u, *r_l;
    int *specularBounce, *anyNonSpecularBounces, *lastVertexVolume;
    lsctx_t *prevIntrCtx;
    int scattered, terminated;
    float rr_correction; /* by value */
    spec beta_factor, r_u_factor;
    isg_sample_t *isg;
    path_counters_t *pc;
    gwrap_t *gbsdf, *gphase; /* persist across the path loop like the GuidedBSDF / GuidedPhaseFunction objects (:287-288) */
    int in_medium; /* ray.medium != nullptr (always 1 where SampleDistance runs, :318) */
} sd_ctx_t;

/* volume-scatter tail shared by both branches (:804-875 == :988-1058) */
static void scatter_tail(sd_ctx_t *c, v3 p, const medium_props_t *mp) {
    const OracleRenderer *r = c->r;
    if (s_nonzero(*c->beta) && s_nonzero(*c->r_u)) {
        intr_t intr;
        memset(&intr, 0, sizeof intr);
        intr.is_surface = 0;
        intr.pi = p3i_exact(p);
        intr.wo = v_neg(*c->ray_d);
        intr.g = mp->g;
        intr.medium = c->in_medium; /* MediumInteraction(p, wo, time, ray.medium, phase) (:806 / :990) */
        float v = sampler_get1d(c->sampler);
        (void)v;
        /* gphase.init(&intr.phase, p, ray.d, v) (guiding.h:383-398) */
        c->gphase->d = gdist_init_volume(r, p, *c->ray_d, mp->g);
        c->gphase->useGuiding = r->prm.volumeguiding ? c->gphase->d.ok : 0;
        c->gphase->useScatterGuiding = r->prm.vspsecondaryguiding ? c->gphase->d.ok : 0;
        float survivalProb = 1.0f;
        if (*c->depth > r->prm.minrrdepth) { /* :817-830 */
            if (c->guideRR) {
                if (r->prm.volumerrguiding)
                    survivalProb = *c->specularBounce ? 0.95f : guided_russian_roulette(*c->beta, S1(1.f), c->pixelContributionEstimate, 0.1f);
                else
                    survivalProb = 1.f;
            } else {
                spec rrw = s_scale(s_divf(*c->beta, s_avg(*c->r_u)), c->rr_correction);
                survivalProb = *c->specularBounce ? 0.95f : standard_throughput_rr(rrw);
            }
        }
        if (r->prm.usenee) {
            spec Ld = sample_Ld(r, &intr, c->gphase, c->ch, c->sampler, *c->r_u, c->pc);
            *c->L = s_add(*c->L, s_mul(*c->beta, Ld));
            rec_add_scattered_direct_light(c->rec, Ld); /* :838 */
        }
        if (survivalProb < 1 && *c->depth > r->prm.minrrdepth) {
            float q = fmaxf(0.f, 1 - survivalProb);
            if (sampler_get1d(c->sampler) < q) {
                DBG(16);
                c->terminated = 1;
                return;
            }
            *c->beta = s_divf(*c->beta, 1 - q);
        }
        float u0 = sampler_get1d(c->sampler), u1 = sampler_get1d(c->sampler);
        /* gphase.Sample_p(-ray.d, u) (guiding.h:404-540) */
        v3 wo = v_neg(*c->ray_d);
        float ps_p = 0, ps_pdf = 0;
        v3 wi = V3(0, 0, 0);
        int have = 0;
        const gwrap_t *gw = c->gphase;
        if (!gw->useGuiding) {
            wi = sample_henyey_greenstein(wo, mp->g, u0, u1, &ps_pdf);
            ps_p = ps_pdf;
            have = 1;
        } else if (r->prm.volumeguidingtype == VSPG_GUIDE_MIS) { /* Sample_p_MIS :404-445 */
            int samplePhase = 1;
            if (GUIDING_PROBABILITY > u0) { u0 /= GUIDING_PROBABILITY; samplePhase = 0; }
            else { u0 -= GUIDING_PROBABILITY; u0 /= (1.0f - GUIDING_PROBABILITY); }
            if (samplePhase) {
                wi = sample_henyey_greenstein(wo, mp->g, u0, u1, &ps_pdf);
                ps_p = ps_pdf;
                float guidedPDF = gdist_pdf(&gw->d, wi);
                ps_pdf = ((1.0f - GUIDING_PROBABILITY) * ps_pdf) + (GUIDING_PROBABILITY * guidedPDF);
                have = 1;
            } else {
                float guidedPDF = gdist_sample(&gw->d, u0, u1, &wi);
                float pp = oracle_henyey_greenstein(v_dot(wo, wi), mp->g);
                float phasePDF = pp;
                if (phasePDF > 0.f) {
                    ps_p = pp;
                    ps_pdf = ((1.0f - GUIDING_PROBABILITY) * phasePDF) + (GUIDING_PROBABILITY * guidedPDF);
                    have = 1;
                }
            }
        } else { /* Sample_p_RIS :447-530 */
            const float uniformIncomingRadiancePDF = (float)(1.0f / (4.0f * M_PI));
            float rp[2] = {0, 0}, rphase[2] = {0, 0}, rguid[2] = {0, 0}, rmis[2] = {0, 0}, rinc[2] = {0, 0}, rw[2] = {0, 0};
            v3 rwi[2];
            rwi[0] = sample_henyey_greenstein(wo, mp->g, u0, u1, &rphase[0]);
            rp[0] = rphase[0];
            rguid[0] = gdist_pdf(&gw->d, rwi[0]);
            rinc[0] = gdist_incoming_pdf(r, &gw->d, rwi[0]);
            rmis[0] = 0.5f * (rphase[0] + rguid[0]);
            float s0 = sampler_get1d(c->sampler), s1 = sampler_get1d(c->sampler);
            rguid[1] = gdist_sample(&gw->d, s0, s1, &rwi[1]);
            rinc[1] = gdist_incoming_pdf(r, &gw->d, rwi[1]);
            rp[1] = oracle_henyey_greenstein(v_dot(wo, rwi[1]), mp->g);
            rphase[1] = rp[1];
            rmis[1] = 0.5f * (rphase[1] + rguid[1]);
            float sumW = 0.f;
            int nS = 0;
            for (int i = 0; i < 2; ++i)
                if (rphase[i] > 0.f) {
                    rw[i] = (rphase[i] * ((1.0f - GUIDING_PROBABILITY) * uniformIncomingRadiancePDF + GUIDING_PROBABILITY * rinc[i]));
                    rw[i] /= rmis[i];
                    sumW += rw[i];
                    nS++;
                }
            if (!(nS == 0 || sumW <= 0.f)) {
                int idx = 0;
                float sample1D = sumW * sampler_get1d(c->sampler);
                float sumR = 0.f;
                for (int i = 0; i < 2; ++i) {
                    sumR += rw[i];
                    if (sample1D <= sumR) { idx = i; break; }
                }
                ps_pdf = (rw[idx] * rmis[idx]) * ((float)2 / sumW);
                ps_p = rp[idx];
                wi = rwi[idx];
                have = 1;
            }
        }
        if (!have || ps_pdf == 0) {
            c->terminated = 1;
        } else {
            float phaseFunctionWeight = ps_p / ps_pdf;
            *c->beta = s_scale(*c->beta, phaseFunctionWeight);
            *c->r_l = s_divf(*c->r_u, ps_pdf);
            memset(c->prevIntrCtx, 0, sizeof *c->prevIntrCtx);
            c->prevIntrCtx->pi = p3i_exact(p);
            c->scattered = 1;
            *c->ray_o = p;
            *c->ray_d = wi;
            *c->specularBounce = 0;
            *c->anyNonSpecularBounces = 1;
            /* guiding_addVolumeData(..., phaseFunctionWeight, ps->wi, ps->pdf, ps->meanCosine, survivalProb) (:871) */
            rec_add_scatter_data(c->rec, 1, S1(phaseFunctionWeight), wi, ps_pdf, 1.0f - fabsf(mp->g), survivalProb);
            *c->lastVertexVolume = 1;
        }
    }
}

/* delta-tracking callback (:885-1078) */
static int delta_cb(void *vctx, v3 p, const medium_props_t *mp, spec sigma_maj, spec T_maj, int activateNDS) {
    sd_ctx_t *c = (sd_ctx_t *)vctx;
    const OracleRenderer *r = c->r;
    int ch = c->ch;
    c->pc->density_queries++;
    if (!s_nonzero(*c->beta)) {
        c->terminated = 1;
        return 0;
    }
    const spec mpLe = medium_Le(mp, c->lambda);
    if (*c->depth < r->prm.maxdepth && s_nonzero(mpLe)) {
        float pdf = sigma_maj.c[ch] * T_maj.c[ch];
        spec betap = s_divf(s_mul(*c->beta, T_maj), pdf);
        spec r_e = s_divf(s_mul(s_mul(*c->r_u, sigma_maj), T_maj), pdf);
        if (s_nonzero(r_e))
            *c->L = s_add(*c->L, s_divf(s_mul(s_mul(betap, mp->sigma_a), mpLe), s_avg(r_e)));
    }
    spec sigma_t = s_add(mp->sigma_s, mp->sigma_a);
    float pScatter = sigma_t.c[ch] / sigma_maj.c[ch];
    int NDS_plus = 0;
    if (*c->depth == 0 && r->prm.vspsamplingmethod == VSPG_VSP_NDS && r->prm.collisionProbabilityBias && r->tr_load && activateNDS) {
        /* NDS+: adjust the real/null-collision probability by the cached transmittance (:930-938) */
        NDS_plus = 1;
        float trEstCache = r->trbuf[((size_t)c->py * r->cfg.xres + c->px) * 3 + ch];
        float gamma = 1 + trEstCache;
        pScatter = powf(pScatter, 1 / gamma);
    }
    float pNull = fmaxf(0.f, 1 - pScatter);
    float um = rng_float(c->rng);
    int mode = oracle_sample_discrete2(pScatter, pNull, um);
    if (mode == 0) {
        if (*c->depth == 0) {
            c->isg->valid = 1;
            c->isg->surface_event = 0;
        }
        if ((*c->depth)++ >= r->prm.maxdepth) {
            c->terminated = 1;
            return 0;
        }
        c->pc->volume_scatters++;
        float pdf = T_maj.c[ch] * sigma_t.c[ch];
        *c->beta = s_mul(*c->beta, s_divf(s_mul(T_maj, mp->sigma_s), pdf));
        *c->r_u = s_mul(*c->r_u, s_divf(s_mul(T_maj, sigma_t), pdf));
        if (NDS_plus) *c->r_u = s_mul(*c->r_u, s_div(s_scale(sigma_maj, pScatter), sigma_t)); /* :975-976 */
        *c->beta = s_mul(*c->beta, c->beta_factor);
        *c->r_u = s_mul(*c->r_u, c->r_u_factor);
        if (c->rec) { /* :978-986 */
            spec tw = s_mul(*c->transmittanceWeight, s_divf(s_mul(T_maj, mp->sigma_s), pdf));
            tw = s_mul(tw, s_divf(c->beta_factor, c->r_u_factor.c[ch]));
            rec_add_transmittance_weight(c->rec, tw);
            rec_new_segment(c->rec, p, 1);
            *c->transmittanceWeight = S1(1.f);
        }
        scatter_tail(c, p, mp);
        return 0;
    } else {
        spec sigma_n = s_clamp_zero(s_sub(s_sub(sigma_maj, mp->sigma_a), mp->sigma_s));
        float pdf = T_maj.c[ch] * sigma_n.c[ch];
        *c->beta = s_mul(*c->beta, s_divf(s_mul(T_maj, sigma_n), pdf));
        if (c->rec) *c->transmittanceWeight = s_mul(*c->transmittanceWeight, s_divf(s_mul(T_maj, sigma_n), pdf)); /* :1067 */
        if (pdf == 0) {
            *c->beta = S1(0.f);
            if (c->rec) *c->transmittanceWeight = S1(0.f);
        }
        *c->r_u = s_mul(*c->r_u, s_divf(s_mul(T_maj, sigma_n), pdf));
        if (NDS_plus) *c->r_u = s_mul(*c->r_u, s_div(s_scale(sigma_maj, 1 - pScatter), sigma_n)); /* :1072-1073 */
        *c->r_l = s_mul(*c->r_l, s_divf(s_mul(T_maj, sigma_maj), pdf));
        return s_nonzero(*c->beta) && s_nonzero(*c->r_u);
    }
}

/* resampling callback state (:684-719) */
typedef struct {
    sd_ctx_t *sd;
    float weightSum;
    spec trRatioEst, beta_rs, r_u_rs;
    /* CandidateData (integrators.h:526-543) */
    int have_sel;
    v3 sel_p;
    medium_props_t sel_mp;
    float sel_wi, sel_sigmaTTr;
    spec sel_num, sel_den;
} rs_ctx_t;

static int resampling_cb(void *vctx, v3 p, const medium_props_t *mp, spec sigma_maj, spec T_maj, int nds) {
    (void)nds;
    rs_ctx_t *c = (rs_ctx_t *)vctx;
    int ch = c->sd->ch;
    c->sd->pc->density_queries++;
    spec sigma_t = s_add(mp->sigma_s, mp->sigma_a);
    spec sigma_n = s_clamp_zero(s_sub(sigma_maj, sigma_t));
    float wi = s_mul(s_div(sigma_t, sigma_maj), c->trRatioEst).c[ch];
    float sigmaTTrEstScalar = wi;
    if (wi > 0) {
        c->weightSum += wi;
        if (sampler_get1d(c->sd->sampler) < wi / c->weightSum) {
            float pdf = T_maj.c[ch] * sigma_t.c[ch];
            c->sel_num = s_divf(s_mul(s_mul(c->beta_rs, T_maj), mp->sigma_s), pdf);
            c->sel_den = s_divf(s_mul(s_mul(c->r_u_rs, T_maj), sigma_t), pdf);
            c->sel_p = p; c->sel_mp = *mp; c->sel_wi = wi; c->sel_sigmaTTr = sigmaTTrEstScalar;
            c->have_sel = 1;
        }
    }
    float pdf = T_maj.c[ch] * sigma_n.c[ch];
    c->beta_rs = s_mul(c->beta_rs, s_divf(s_mul(T_maj, sigma_n), pdf));
    c->r_u_rs = s_mul(c->r_u_rs, s_divf(s_mul(T_maj, sigma_n), pdf));
    c->trRatioEst = s_mul(c->trRatioEst, s_div(sigma_n, sigma_maj));
    return 1;
}

/* VSP fetch (:654-671, :1098-1134).  Primary rays: the image-space buffer; secondary rays: the previous vertex's
 * GuidedBSDF / GuidedPhaseFunction::VolumeScatterProbability(ray.d) (guiding.h:295-305, 564-574), -1 where the cache
 * has no trained distribution. */
static float fetch_vsp(const OracleRenderer *r, int px, int py, int depth, int lastVertexVolume, const gwrap_t *gbsdf,
                       const gwrap_t *gphase, v3 rayd, int *guide) {
    float vsp = -1.f;
    *guide = 0;
    if (depth == 0) {
        if (r->prm.vspguiding && r->prm.vspprimaryguiding) {
            if (r->vsp_ready) vsp = r->vsp[(size_t)py * r->cfg.xres + px];
            else vsp = 0.5f;
            *guide = !(isnan(vsp) || vsp < 0.f || vsp > 1.f);
        }
    } else {
        if (r->prm.vspguiding && r->prm.vspsecondaryguiding) {
            /* g{phase,bsdf}.VolumeScatterProbability(ray.d) (:661-668, guiding.h:295-305, 564-574) */
            const gwrap_t *gw = lastVertexVolume ? gphase : gbsdf;
            vsp = gw->useScatterGuiding ? gdist_vsp(r, gw->d.field, gw->d.region, gw->d.p, rayd) : -1.f;
            *guide = !(isnan(vsp) || vsp < 0.f || vsp > 1.f);
        }
    }
    if (*guide) vsp = fmaxf(fminf(vsp, 0.999f), 0.001f);
    return vsp;
}

static void sample_distance(sd_ctx_t *c, int px, int py, float tMax) {
    const OracleRenderer *r = c->r;
    int ch = c->ch;
    int guide;
    float vsp = fetch_vsp(r, px, py, *c->depth, *c->lastVertexVolume, c->gbsdf, c->gphase, *c->ray_d, &guide);
    if (*c->depth == 0) c->isg->vsp_used = guide ? vsp : -1.f;

    int use_resampling = r->prm.vspsamplingmethod == VSPG_VSP_RESAMPLING && !medium_is_homogeneous(r);
    if (use_resampling) {
        rs_ctx_t rc;
        memset(&rc, 0, sizeof rc);
        rc.sd = c;
        rc.weightSum = 0;
        rc.trRatioEst = S1(1.f); rc.beta_rs = S1(1.f); rc.r_u_rs = S1(1.f);
        float vrc = 0, majorantScale = 1;
        float u = sampler_get1d(c->sampler);
        spec T_maj = sample_T_maj_resampling(r, *c->ray_o, *c->ray_d, tMax, u, c->rng, ch, guide, vsp,
                                             &vrc, &majorantScale, resampling_cb, &rc);
        rc.beta_rs = s_mul(rc.beta_rs, s_divf(T_maj, T_maj.c[ch]));
        rc.r_u_rs = s_mul(rc.r_u_rs, s_divf(T_maj, T_maj.c[ch]));
        if (*c->depth == 0 && r->tr_calc && r->in_wave) { /* trBuffer->AddSample (:727-728, trbuffer.h:40-45) */
            size_t pix = (size_t)py * r->cfg.xres + px;
            r->tr_spp[pix] += 1;
            float alpha = 1.f / r->tr_spp[pix];
            for (int k = 0; k < 3; ++k) r->trbuf[pix * 3 + k] = (1.f - alpha) * r->trbuf[pix * 3 + k] + alpha * rc.trRatioEst.c[k];
        }
        float trScalar = rc.trRatioEst.c[ch];
        /* surfaceCandidate */
        float surf_wi = trScalar, surf_sigmaTTr = trScalar;
        spec surf_num = rc.beta_rs, surf_den = rc.r_u_rs;
        if (guide && trScalar < 1 && trScalar > 0 && rc.weightSum > 0) {
            float trEstForScale = trScalar;
            float volRatio = vrc * r->prm.vspmisratio + (1 - trEstForScale) * (1 - r->prm.vspmisratio);
            float surfRatio = 1 - volRatio;
            surf_wi = surfRatio / volRatio * rc.weightSum;
        }
        rc.weightSum += surf_wi;
        int selectSurface = 0;
        if (rc.weightSum == 0) {
            return;
        } else if (sampler_get1d(c->sampler) < surf_wi / rc.weightSum) {
            rc.sel_wi = surf_wi; rc.sel_sigmaTTr = surf_sigmaTTr; rc.sel_num = surf_num; rc.sel_den = surf_den;
            selectSurface = 1;
        }
        float resamplingFactorScalar = rc.weightSum * rc.sel_sigmaTTr / rc.sel_wi;
        if (selectSurface) {
            *c->beta = s_mul(*c->beta, s_scale(rc.sel_num, resamplingFactorScalar));
            *c->r_u = s_mul(*c->r_u, rc.sel_den);
            /* SUBSTITUTION (:766, and :792 below): the reference tests IsInf(beta.y(lambda)) / IsInf(r_u.y(lambda)) -- the CIE-Y weighted
             * mean of the three channels, weights Y(lambda_i) / pdf(lambda_i) at the SAMPLED wavelengths (spectrum.cpp:252-256; they
             * need atanh / cosh of the wavelength sample and the tabulated CIE curve).  This restatement tests "some channel is
             * infinite".  The two agree unless a weight is exactly 0 (Y(lambda) vanishes at the ends of the visible range: 0 * inf is
             * NaN, not Inf), infinities of both signs meet (NaN), or the finite weighted sum itself overflows: in those corners the
             * reference carries the path on with a non-finite throughput that the radiance clean-up (integrators.cpp:308-318)
             * blacks out anyway, this side ends the path here -- same pixel value, zero.  Noted in DESIGN.md section 6. */
            if (s_has_nan(*c->beta) || s_has_nan(*c->r_u) || s_has_inf(*c->beta) || s_has_inf(*c->r_u)) {
                c->terminated = 1;
                return;
            }
        } else {
            v3 p = rc.sel_p;
            medium_props_t mp = rc.sel_mp;
            if (*c->depth == 0) {
                c->isg->valid = 1;
                c->isg->surface_event = 0;
            }
            if ((*c->depth)++ >= r->prm.maxdepth) {
                c->terminated = 1;
                return;
            }
            c->pc->volume_scatters++;
            *c->beta = s_mul(*c->beta, s_scale(rc.sel_num, resamplingFactorScalar));
            *c->r_u = s_mul(*c->r_u, rc.sel_den);
            if (s_has_nan(*c->beta) || s_has_nan(*c->r_u) || s_has_inf(*c->beta) || s_has_inf(*c->r_u)) {
                c->terminated = 1;
                return;
            }
            if (c->rec) { /* :798-802 */
                spec tw = s_mul(*c->transmittanceWeight, s_div(s_scale(rc.sel_num, resamplingFactorScalar), rc.sel_den));
                rec_add_transmittance_weight(c->rec, tw);
                rec_new_segment(c->rec, p, 1);
                *c->transmittanceWeight = S1(1.f);
            }
            scatter_tail(c, p, &mp);
        }
    } else {
        c->beta_factor = S1(1.f);
        c->r_u_factor = S1(1.f);
        float u = sampler_get1d(c->sampler);
        spec T_maj = sample_T_maj_ods(r, *c->ray_o, *c->ray_d, tMax, u, c->rng, ch, guide, vsp,
                                      r->prm.vspmisratio, r->prm.vspsamplingmethod == VSPG_VSP_NDS,
                                      &c->beta_factor, &c->r_u_factor, delta_cb, c);
        int multiply_T_maj = !(c->scattered || c->terminated || !s_nonzero(*c->beta) || !s_nonzero(*c->r_u));
        if (multiply_T_maj) {
            *c->beta = s_mul(*c->beta, s_divf(T_maj, T_maj.c[ch]));
            *c->r_u = s_mul(*c->r_u, s_divf(T_maj, T_maj.c[ch]));
            *c->r_l = s_mul(*c->r_l, s_divf(T_maj, T_maj.c[ch]));
            *c->beta = s_mul(*c->beta, c->beta_factor);
            *c->r_u = s_mul(*c->r_u, c->r_u_factor);
            *c->r_l = s_mul(*c->r_l, c->r_u_factor);
            if (c->rec) { /* :1085, :1090 */
                spec tw = s_mul(*c->transmittanceWeight, s_divf(T_maj, T_maj.c[ch]));
                *c->transmittanceWeight = s_mul(tw, s_divf(c->beta_factor, c->r_u_factor.c[ch]));
            }
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* a16: Li (guidedvolpathvspgintegrator.cpp:262-635)                                      */
/* ------------------------------------------------------------------------------------ */
static spec Li(const OracleRenderer *r, int px, int py, v3 ro, v3 rd, int ch, const float lambda[3], sampler_t *sampler,
               isg_sample_t *isg, path_counters_t *pc, pathrec_t *rec) {
    float rr_correction = 1.0f;
    spec L = S1(0.f), beta = S1(1.f), r_u = S1(1.f), r_l = S1(1.f);
    int specularBounce = 0, anyNonSpecularBounces = 0;
    int depth = 0;
    float etaScale = 1;
    int lastVertexVolume = 0;
    lsctx_t prevIntrCtx;
    memset(&prevIntrCtx, 0, sizeof prevIntrCtx);
    isg->valid = 0; isg->surface_event = 0; isg->vsp_used = -1.f;
    gwrap_t gbsdf, gphase; /* GuidedBSDF gbsdf / GuidedPhaseFunction gphase (:287-288) */
    memset(&gbsdf, 0, sizeof gbsdf);
    memset(&gphase, 0, sizeof gphase);
    /* :274-285: guided RR needs the pixel's contribution estimate; until the image-space buffer is ready the path
     * uses the standard rule */
    int guideRR = 0;
    spec pixelContributionEstimate = S1(1.f);
    if (r->prm.rrguiding && r->contrib_ready) {
        const float *ce = &r->contrib[((size_t)py * r->cfg.xres + px) * 3];
        pixelContributionEstimate.c[0] = ce[0]; pixelContributionEstimate.c[1] = ce[1]; pixelContributionEstimate.c[2] = ce[2];
        guideRR = 1;
    }

    /* ray.medium: the camera ray starts in the camera's medium (cameras.h CameraBase::medium, set from the graphics state's
     * outside medium at the Camera directive, scene.cpp:153-155); every spawned ray takes GetMedium(direction) */
    int in_medium = r->camera_in_medium;
    while (1) {
        pc->segments++;
        isect_t si = scene_intersect(r, ro, rd, INFINITY);
        float tMax = si.hit ? si.t : INFINITY;
        spec transmittanceWeight = S1(1.0f); /* :317 */
        if (in_medium && !isinf(tMax)) { /* :318 -- a ray that escapes the scene is not sampled (rounds 1-3 sampled it: wrong) */
            rng_t rng;
            uint64_t hash0 = oracle_hash_float(sampler_get1d(sampler));
            uint64_t hash1 = oracle_hash_float(sampler_get1d(sampler));
            rng_set_sequence2(&rng, hash0, hash1);
            sd_ctx_t c;
            memset(&c, 0, sizeof c);
            c.r = r; c.ch = ch; c.lambda = lambda; c.px = px; c.py = py; c.sampler = sampler; c.rng = &rng;
            c.guideRR = guideRR; c.pixelContributionEstimate = pixelContributionEstimate;
            c.ray_o = &ro; c.ray_d = &rd; c.depth = &depth;
            c.L = &L; c.beta = &beta; c.r_u = &r_u; c.r_l = &r_l;
            c.specularBounce = &specularBounce; c.anyNonSpecularBounces = &anyNonSpecularBounces;
            c.lastVertexVolume = &lastVertexVolume; c.prevIntrCtx = &prevIntrCtx;
            c.rr_correction = rr_correction;
            c.isg = isg; c.pc = pc;
            c.gbsdf = &gbsdf; c.gphase = &gphase;
            c.rec = rec; c.transmittanceWeight = &transmittanceWeight;
            c.in_medium = in_medium;
            sample_distance(&c, px, py, tMax);
            if (c.terminated || !s_nonzero(beta) || !s_nonzero(r_u)) break;
            if (c.scattered) continue;
        }
        rec_add_transmittance_weight(rec, transmittanceWeight); /* :350 */
        if (!si.hit) { /* :353-374: contributions from infinite light sources (this fork lists DeltaDirection lights there too, integrators.h:79) */
            for (int k = 0; k < r->n_inf; ++k) {
                const VspgInfiniteLight *il = &r->scene.infinite_lights[k];
                spec Le = s_from(il->L); /* UniformInfiniteLight::Le / DistantLight::Le (lights.cpp:1014-1017, lights.h:291-293) */
                if (il->type == VSPG_LIGHT_DISTANT && depth != 0) Le = S1(0.f);
                if (depth == 0 || specularBounce) {
                    L = s_add(L, s_divf(s_mul(beta, Le), s_avg(r_u)));
                    rec_add_infinite_light_emission(rec, ro, rd, Le, 1.0f); /* :361 */
                } else {
                    /* lightSampler.PMF * light.PDF_Li(prevIntrContext, ray.d, true): both light types return 0 for the incomplete PDF */
                    float lightPDF = lightsampler_pmf(r, r->lsamp, p3i_mid(prevIntrCtx.pi), prevIntrCtx.ns, r->n_lights + k) * 0.f;
                    r_l = s_scale(r_l, lightPDF);
                    float w_b = r->prm.usenee ? 1.0f / s_avg(s_add(r_u, r_l)) : 1.f;
                    L = s_add(L, s_mul(s_scale(beta, w_b), Le));
                    rec_add_infinite_light_emission(rec, ro, rd, Le, w_b); /* :369 */
                }
            }
            break;
        }

        const rquad_t *q = &r->quads[si.quad < 0 ? 0 : si.quad];
        /* isect.Le(-ray.d) (:377-397); triangles and spheres carry no area light */
        spec Le = si.quad >= 0 && q->is_light ? light_L(q, si.n, v_neg(rd)) : S1(0.f);
        int add_direct_contribution = 0;
        float w_direct = 0.f;
        if (s_nonzero(Le)) {
            add_direct_contribution = 1;
            if (depth == 0 || specularBounce) {
                DBG(1);
                L = s_add(L, s_divf(s_mul(beta, Le), s_avg(r_u)));
                w_direct = 1.0f;
            } else {
                DBG(lastVertexVolume ? 2 : 4);
                float lightPDF = lightsampler_pmf(r, r->lsamp, p3i_mid(prevIntrCtx.pi), prevIntrCtx.ns, r->light_of_quad[si.quad]) * light_pdf_li(q, &prevIntrCtx, rd);
                r_l = s_scale(r_l, lightPDF);
                float w_l = r->prm.usenee ? 1.0f / s_avg(s_add(r_u, r_l)) : 1.0f;
                L = s_add(L, s_mul(s_scale(beta, w_l), Le));
                w_direct = w_l;
            }
        }
        p3i pi = isect_pi(&si); /* SurfaceInteraction pi (shapes.h InteractionFromIntersection) */
        if (!si.has_material) { /* :399-404: no BSDF -- a medium boundary.  isect.SkipIntersection(&ray, si->tHit) (interaction.cpp:91-97) */
            ro = offset_ray_origin(pi, si.n, rd); /* SpawnRay(ray->d) (interaction.h:99-101) */
            in_medium = r->has_medium ? get_medium(si.iface, si.n, rd, in_medium) : 0;
            continue;
        }
        bsdf_t bsdf = si.tri >= 0 ? bsdf_make_tri(&r->tris[si.tri]) : (si.sphere >= 0 ? bsdf_make_sphere(&r->spheres[si.sphere], &si) : bsdf_make(q));
        rec_new_segment(rec, v_add(ro, v_scale(rd, si.t)), 0); /* guiding_newSurfacePathSegment (:406) */
        if (add_direct_contribution) rec_add_surface_emission(rec, Le, w_direct); /* :409-412 */
        if (depth == 0) {
            isg->valid = 1;
            isg->surface_event = 1;
        }
        if (depth++ >= r->prm.maxdepth) break;
        pc->surface_hits++;

        float v = sampler_get1d(sampler);
        (void)v;
        float survivalProb = 1.f;
        if (guideRR && depth > r->prm.minrrdepth) { /* :465-472 (adjointEstimate stays 1: OPENPGL_RADIANCE_CACHES is not defined) */
            if (r->prm.surfacerrguiding)
                survivalProb = specularBounce ? 0.95f : guided_russian_roulette(beta, S1(1.f), pixelContributionEstimate, 0.1f);
            else
                survivalProb = 1.f;
        }
        intr_t intr;
        memset(&intr, 0, sizeof intr);
        intr.is_surface = 1;
        intr.pi = pi; intr.n = si.n;
        intr.iface = si.iface; intr.medium = in_medium;
        intr.wo = v_normalize(v_neg(rd)); /* Interaction ctor normalises wo (interaction.h:31-32) */
        intr.bsdf = &bsdf;
        /* gbsdf.init(&bsdf, ray, si, v) (guiding.h:83-109): p = ray.o + tHit*ray.d, cosine product
         * with the shading normal flipped towards -ray.d; only for non-specular BSDFs */
        memset(&gbsdf.d, 0, sizeof gbsdf.d);
        if (bsdf.has_lobes) {
            v3 pg = v_add(ro, v_scale(rd, si.t));
            v3 ng = si.n;
            if (v_dot(v_neg(rd), si.n) < 0.f) ng = v_neg(ng);
            gbsdf.d = gdist_init_surface(r, pg, ng);
        }
        gbsdf.useGuiding = r->prm.surfaceguiding ? gbsdf.d.ok : 0;
        gbsdf.useScatterGuiding = r->prm.vspsecondaryguiding ? gbsdf.d.ok : 0;
        if (r->prm.usenee && bsdf.has_lobes) { /* IsNonSpecular(bsdf.Flags()) */
            spec Ld = sample_Ld(r, &intr, &gbsdf, ch, sampler, r_u, pc);
            L = s_add(L, s_mul(beta, Ld));
            rec_add_scattered_direct_light(rec, Ld); /* :485 */
        }
        prevIntrCtx.pi = pi; prevIntrCtx.n = si.n; prevIntrCtx.ns = si.n;

        v3 wo = v_neg(rd);
        float u = sampler_get1d(sampler);
        float u20 = sampler_get1d(sampler), u21 = sampler_get1d(sampler);
        /* gbsdf.Sample_f(wo, u, u2) (guiding.h:120-269) */
        spec f = S1(0.f);
        v3 wi = V3(0, 0, 0);
        float pdf = 0, bsdfPdf = 0, misPdf = 0;
        int have = 0;
        if (!gbsdf.useGuiding) {
            have = bsdf_sample_f(&bsdf, wo, u, u20, u21, &f, &wi, &pdf);
            bsdfPdf = misPdf = pdf;
        } else if (r->prm.surfaceguidingtype == VSPG_GUIDE_MIS) { /* Sample_f_MIS :120-167 */
            int sampleBSDF = 1;
            if (GUIDING_PROBABILITY > u) { u /= GUIDING_PROBABILITY; sampleBSDF = 0; }
            else { u -= GUIDING_PROBABILITY; u /= (1.0f - GUIDING_PROBABILITY); }
            if (sampleBSDF) {
                have = bsdf_sample_f(&bsdf, wo, u, u20, u21, &f, &wi, &pdf);
                if (have) {
                    float guidedPDF = gdist_pdf(&gbsdf.d, wi);
                    bsdfPdf = pdf;
                    pdf = ((1.0f - GUIDING_PROBABILITY) * pdf) + (GUIDING_PROBABILITY * guidedPDF);
                    misPdf = pdf;
                }
            } else {
                float guidedPDF = gdist_sample(&gbsdf.d, u20, u21, &wi);
                f = bsdf_f(&bsdf, wo, wi);
                float bPDF = bsdf_pdf(&bsdf, wo, wi);
                if (bPDF > 0.f) {
                    pdf = ((1.0f - GUIDING_PROBABILITY) * bPDF) + (GUIDING_PROBABILITY * guidedPDF);
                    bsdfPdf = bPDF;
                    misPdf = pdf;
                    have = 1;
                }
            }
        } else { /* Sample_f_RIS :169-257 */
            const float uniformIncomingRadiancePDF = (float)(1.0f / (4.0f * M_PI));
            spec rf[2] = {S1(0.f), S1(0.f)};
            float rb[2] = {0, 0}, rguid[2] = {0, 0}, rmis[2] = {0, 0}, rinc[2] = {0, 0}, rw[2] = {0, 0};
            v3 rwi[2] = {V3(0, 0, 0), V3(0, 0, 0)};
            float p0;
            if (bsdf_sample_f(&bsdf, wo, u, u20, u21, &rf[0], &rwi[0], &p0)) {
                rb[0] = p0;
                rguid[0] = gdist_pdf(&gbsdf.d, rwi[0]);
                rinc[0] = gdist_incoming_pdf(r, &gbsdf.d, rwi[0]);
                rmis[0] = 0.5f * (rb[0] + rguid[0]);
            }
            float s0 = sampler_get1d(sampler), s1 = sampler_get1d(sampler);
            rguid[1] = gdist_sample(&gbsdf.d, s0, s1, &rwi[1]);
            rinc[1] = gdist_incoming_pdf(r, &gbsdf.d, rwi[1]);
            rf[1] = bsdf_f(&bsdf, wo, rwi[1]);
            rb[1] = bsdf_pdf(&bsdf, wo, rwi[1]);
            rmis[1] = 0.5f * (rb[1] + rguid[1]);
            float sumW = 0.f;
            int nS = 0;
            for (int i = 0; i < 2; ++i)
                if (rb[i] > 0.f) {
                    rw[i] = (rb[i] * ((1.0f - GUIDING_PROBABILITY) * uniformIncomingRadiancePDF + GUIDING_PROBABILITY * rinc[i]));
                    rw[i] /= rmis[i];
                    sumW += rw[i];
                    nS++;
                }
            if (!(nS == 0 || sumW <= 0.f)) {
                int idx = 0;
                float sample1D = sumW * sampler_get1d(sampler);
                float sumR = 0.f;
                for (int i = 0; i < 2; ++i) {
                    sumR += rw[i];
                    if (sample1D <= sumR) { idx = i; break; }
                }
                pdf = (rw[idx] * rmis[idx]) * ((float)2 / sumW);
                misPdf = rmis[idx];
                bsdfPdf = rb[idx];
                f = rf[idx];
                wi = rwi[idx];
                have = 1;
            }
        }
        if (!have) break;
        lastVertexVolume = 0;
        rr_correction *= pdf / bsdfPdf;
        spec bsdfWeight = s_divf(s_scale(f, v_absdot(wi, si.n)), pdf);
        beta = s_mul(beta, bsdfWeight);
        r_l = s_divf(r_u, misPdf);
        specularBounce = 0;
        anyNonSpecularBounces = 1;
        ro = offset_ray_origin(pi, si.n, wi); /* SpawnRay (interaction.h:99-101) */
        rd = wi;
        in_medium = r->has_medium ? get_medium(si.iface, si.n, wi, in_medium) : 0; /* ... with GetMedium(wi) */

        if (!s_nonzero(beta)) break;
        if (!guideRR && depth > r->prm.minrrdepth) { /* :597-600 */
            spec rrw = s_scale(s_scale(s_divf(beta, s_avg(r_u)), rr_correction), etaScale);
            survivalProb = specularBounce ? 0.95f : standard_throughput_rr(rrw);
        }
        if (survivalProb < 1 && depth > r->prm.minrrdepth) {
            float qq = fmaxf(0.f, 1 - survivalProb);
            if (sampler_get1d(sampler) < qq) { DBG(8); break; }
            beta = s_divf(beta, 1 - qq);
        }
        /* guiding_addSurfaceData(..., bsdfWeight, bs->wi, bs->eta, bs->sampledRoughness, bs->pdf, survivalProb) (:608);
         * DiffuseBxDF: sampledRoughness 1 */
        rec_add_scatter_data(rec, 0, bsdfWeight, wi, pdf, 1.0f, survivalProb);
    }
    return L;
}

/* ------------------------------------------------------------------------------------ */
/* a18: training.  PathSegmentStorage::PropagateSamples and Field::Update are OpenPGL code   */
/* (absent from the reference tree): OWN DESIGN, PARITY UNPINNED.  The product (HIP) follows  */
/* the same definitions; radiance samples are compared bit for bit, the fitted field within   */
/* a tolerance (the device sums with float atomics, this file with doubles).                 */
/* ------------------------------------------------------------------------------------ */
/* Incident-radiance samples from one finished path, walked from the last vertex to the first:
 *   Lin(i)  = T(i) * Lout(i+1)                     radiance arriving at vertex i along wi(i)
 *   Lout(i) = mi(i)*direct(i) + scattered(i) + scatteringWeight(i) * Lin(i) / rr(i)
 * (direct = emission seen at the vertex with its MIS weight, scattered = NEE estimate, T = the
 * transmittance weight of the segment leaving i).  Vertex i yields a sample when it has a sampled,
 * non-delta direction and a next vertex; weight = average(Lin)/pdf; zero-valued samples are counted
 * and dropped. */
static void propagate_samples(OracleRenderer *r, const pathrec_t *rec) {
    VspgTrainSample out[TRAIN_MAX_SEG];
    int n_out = 0, n_zero = 0;
    spec Lout_next = S1(0.f);
    v3 p_next = V3(0, 0, 0);
    int have_next = 0, next_volume = 0;
    for (int i = rec->n - 1; i >= 0; --i) {
        const pathseg_t *g = &rec->seg[i];
        spec Lin = have_next ? s_mul(g->transmittance_weight, Lout_next) : S1(0.f);
        if (g->has_wi && !g->is_delta && have_next && g->pdf > 0) {
            float w = s_avg(Lin) / g->pdf;
            if (w > 0 && !isinf(w)) {
                VspgTrainSample *o = &out[n_out++];
                o->p[0] = g->p.x; o->p[1] = g->p.y; o->p[2] = g->p.z;
                o->dir[0] = g->wi.x; o->dir[1] = g->wi.y; o->dir[2] = g->wi.z;
                o->weight = w; o->pdf = g->pdf;
                o->distance = v_len(v_sub(p_next, g->p));
                o->flags = (g->volume ? VSPG_SAMPLE_VOLUME : 0u) | (next_volume ? VSPG_SAMPLE_NEXT_VOLUME : 0u);
            } else {
                n_zero++;
            }
        }
        spec Lout = s_add(s_scale(g->direct, g->mi_weight), g->scattered);
        if (g->has_wi) Lout = s_add(Lout, s_divf(s_mul(g->scattering_weight, Lin), g->rr));
        Lout_next = Lout; p_next = g->p; have_next = 1; next_volume = g->volume;
    }
    if (n_out == 0 && n_zero == 0) return;
#pragma omp critical(vspg_samples)
    {
        if (r->n_samples + (size_t)n_out > r->cap_samples) {
            size_t nc = r->cap_samples ? r->cap_samples * 2 : 65536;
            while (nc < r->n_samples + (size_t)n_out) nc *= 2;
            r->samples = (VspgTrainSample *)realloc(r->samples, nc * sizeof(VspgTrainSample));
            r->cap_samples = nc;
        }
        if (n_out > 0) memcpy(r->samples + r->n_samples, out, (size_t)n_out * sizeof(VspgTrainSample));
        r->n_samples += (size_t)n_out;
        r->n_zero_samples += (uint64_t)n_zero;
    }
}

/* ---- Field::Update ------------------------------------------------------------------------
 * field = kd-tree over positions, leaf -> region = parallax-aware vMF mixture (VspgFieldRegion) plus
 * decayed sufficient statistics (region_stats).  One update = decay, spatial refinement (a leaf that
 * has seen more than TRAIN_SPLIT_COUNT samples splits at the mean position along its axis of largest
 * variance; both children inherit the mixture and half the statistics), then ONE weighted EM step of
 * every region over the samples that fall into it:
 *   E: gamma_k ~ weight_k vMF(omega' | mu_k, kappa_k), omega' = the sample direction re-aimed at the
 *      region's pivot through the sample's hit distance (parallax compensation);
 *   statistics per lobe: S += w gamma, R += w gamma omega', D += w gamma / distance,
 *      V += w gamma [next event is a volume scatter], Qv / Qs += gamma w^2 [volume / surface next event];
 *   M: weight = S / sum S (floored), mu = R/|R|, kappa from the mean resultant length
 *      rbar(3 - rbar^2)/(1 - rbar^2), distance = S / D, vsp = V/S (contribution criterion) or
 *      sqrt(Qv)/(sqrt(Qv)+sqrt(Qs)) (variance criterion).
 * w is the sample weight clamped to TRAIN_WEIGHT_CLAMP times the batch mean weight. */
#define TRAIN_SPLIT_COUNT 4096.0f
#define TRAIN_DECAY 0.75f
#define TRAIN_MAX_DEPTH 24
#define TRAIN_WEIGHT_CLAMP 32.0f
#define TRAIN_KAPPA_INIT 2.0f
#define TRAIN_MIN_UPDATE_SAMPLES 128 /* guidedvolpathvspgintegrator.cpp:238 */
struct region_stats {
    float n;             /* decayed sample count */
    float sum_p[3], sum_p2[3];
    float S[VSPG_FIELD_LOBES], R[3][VSPG_FIELD_LOBES], D[VSPG_FIELD_LOBES], V[VSPG_FIELD_LOBES],
        Qv[VSPG_FIELD_LOBES], Qs[VSPG_FIELD_LOBES];
    int32_t depth;
};
#define TRAIN_STAT_FLOATS (7 + 8 * VSPG_FIELD_LOBES) /* every float member of region_stats, in order */
typedef struct { /* double-precision accumulators of one update (the device uses float atomics) */
    double n, sum_p[3], sum_p2[3];
    double S[VSPG_FIELD_LOBES], R[3][VSPG_FIELD_LOBES], D[VSPG_FIELD_LOBES], V[VSPG_FIELD_LOBES],
        Qv[VSPG_FIELD_LOBES], Qs[VSPG_FIELD_LOBES];
} region_acc_t;

static void region_init_lobes(VspgFieldRegion *R) {
    /* 8 lobes on the cube diagonals, kappa 2, equal weights, no parallax, vsp 0.5 */
    const float c = 0.57735026918962576451f;
    R->n_lobes = VSPG_FIELD_LOBES;
    for (int k = 0; k < VSPG_FIELD_LOBES; ++k) {
        R->weight[k] = 1.0f / VSPG_FIELD_LOBES;
        R->kappa[k] = TRAIN_KAPPA_INIT;
        R->mu[0][k] = (k & 1) ? -c : c;
        R->mu[1][k] = (k & 2) ? -c : c;
        R->mu[2][k] = (k & 4) ? -c : c;
        R->distance[k] = INFINITY;
        R->vsp[k] = 0.5f;
    }
}
static int field_alloc(OracleRenderer *r, int f) {
    if (r->field[f].nodes) return 0;
    r->cap_nodes[f] = 8192; r->cap_regions[f] = 4097;
    r->field[f].nodes = (VspgKdNode *)calloc((size_t)r->cap_nodes[f], sizeof(VspgKdNode));
    r->field[f].regions = (VspgFieldRegion *)calloc((size_t)r->cap_regions[f], sizeof(VspgFieldRegion));
    r->rstats[f] = (struct region_stats *)calloc((size_t)r->cap_regions[f], sizeof(struct region_stats));
    r->field[f].n_nodes = 1; r->field[f].n_regions = 1;
    r->field[f].nodes[0].split = 0; r->field[f].nodes[0].packed = 3u; /* leaf -> region 0, untrained */
    return 0;
}
static v3 reaim(const VspgFieldRegion *R, v3 p, v3 w, float dist) {
    if (!(dist > 0) || isinf(dist)) return w;
    v3 t = v_add(v_sub(p, v3_from(R->pivot)), v_scale(w, dist));
    float l2 = v_len2(t);
    if (!(l2 > 0)) return w;
    return v_normalize(t);
}
static void field_update_one(OracleRenderer *r, int f, float mean_w) {
    int want_vol = f == 1;
    VspgKdNode *nodes = r->field[f].nodes;
    VspgFieldRegion *regs = r->field[f].regions;
    struct region_stats *st = r->rstats[f];
    /* 1. decay */
    for (int i = 0; i < r->field[f].n_regions; ++i) {
        float *v = &st[i].n;
        for (int k = 0; k < TRAIN_STAT_FLOATS; ++k) v[k] *= TRAIN_DECAY;
    }
    /* 2. position statistics of the new samples */
    region_acc_t *acc = (region_acc_t *)calloc((size_t)r->cap_regions[f], sizeof(region_acc_t));
    for (size_t i = 0; i < r->n_samples; ++i) {
        const VspgTrainSample *sm = &r->samples[i];
        if (((sm->flags & VSPG_SAMPLE_VOLUME) != 0) != want_vol) continue;
        int reg = field_lookup(r, f, v3_from(sm->p));
        if (reg < 0) continue;
        acc[reg].n += 1;
        for (int a = 0; a < 3; ++a) { acc[reg].sum_p[a] += sm->p[a]; acc[reg].sum_p2[a] += (double)sm->p[a] * sm->p[a]; }
    }
    int n_reg0 = r->field[f].n_regions;
    for (int i = 0; i < n_reg0; ++i) {
        st[i].n += (float)acc[i].n;
        for (int a = 0; a < 3; ++a) { st[i].sum_p[a] += (float)acc[i].sum_p[a]; st[i].sum_p2[a] += (float)acc[i].sum_p2[a]; }
    }
    /* 3. spatial refinement: one split level per update */
    int n_nodes0 = r->field[f].n_nodes;
    for (int nd = 0; nd < n_nodes0; ++nd) {
        if ((nodes[nd].packed & 3u) != 3u) continue;
        int reg = (int)(nodes[nd].packed >> 2);
        struct region_stats *s0 = &st[reg];
        if (!(s0->n > TRAIN_SPLIT_COUNT) || s0->depth >= TRAIN_MAX_DEPTH) continue;
        if (r->field[f].n_nodes + 2 > r->cap_nodes[f] || r->field[f].n_regions + 1 > r->cap_regions[f]) continue;
        float mean[3], var[3];
        for (int a = 0; a < 3; ++a) {
            mean[a] = s0->sum_p[a] / s0->n;
            var[a] = s0->sum_p2[a] / s0->n - mean[a] * mean[a];
        }
        int axis = var[0] >= var[1] ? (var[0] >= var[2] ? 0 : 2) : (var[1] >= var[2] ? 1 : 2);
        if (!(var[axis] > 0)) continue;
        int left = r->field[f].n_nodes, newreg = r->field[f].n_regions;
        r->field[f].n_nodes += 2; r->field[f].n_regions += 1;
        /* halve the statistics, give both children the parent's mixture */
        float *v = &s0->n;
        for (int k = 0; k < TRAIN_STAT_FLOATS; ++k) v[k] *= 0.5f;
        s0->depth += 1;
        st[newreg] = *s0;
        regs[newreg] = regs[reg];
        nodes[left].split = 0; nodes[left].packed = ((uint32_t)reg << 2) | 3u;
        nodes[left + 1].split = 0; nodes[left + 1].packed = ((uint32_t)newreg << 2) | 3u;
        nodes[nd].split = mean[axis];
        nodes[nd].packed = ((uint32_t)left << 2) | (uint32_t)axis;
    }
    /* 4. pivots / default lobes of regions that see their first samples, then the E step */
    memset(acc, 0, (size_t)r->cap_regions[f] * sizeof(region_acc_t));
    for (size_t i = 0; i < r->n_samples; ++i) {
        const VspgTrainSample *sm = &r->samples[i];
        if (((sm->flags & VSPG_SAMPLE_VOLUME) != 0) != want_vol) continue;
        int reg = field_lookup(r, f, v3_from(sm->p));
        if (reg < 0) continue;
        acc[reg].n += 1;
        for (int a = 0; a < 3; ++a) acc[reg].sum_p[a] += sm->p[a];
    }
    for (int i = 0; i < r->field[f].n_regions; ++i)
        if (regs[i].n_lobes == 0 && acc[i].n > 0) {
            for (int a = 0; a < 3; ++a) regs[i].pivot[a] = (float)(acc[i].sum_p[a] / acc[i].n);
            region_init_lobes(&regs[i]);
        }
    memset(acc, 0, (size_t)r->cap_regions[f] * sizeof(region_acc_t));
    const float wmax = TRAIN_WEIGHT_CLAMP * mean_w;
    for (size_t i = 0; i < r->n_samples; ++i) {
        const VspgTrainSample *sm = &r->samples[i];
        if (((sm->flags & VSPG_SAMPLE_VOLUME) != 0) != want_vol) continue;
        v3 p = v3_from(sm->p);
        int reg = field_lookup(r, f, p);
        if (reg < 0) continue;
        const VspgFieldRegion *R = &regs[reg];
        if (R->n_lobes <= 0) continue;
        float w = sm->weight < wmax ? sm->weight : wmax;
        v3 om = reaim(R, p, v3_from(sm->dir), sm->distance);
        float g[VSPG_FIELD_LOBES], gs = 0;
        for (int k = 0; k < R->n_lobes; ++k) {
            g[k] = R->weight[k] * vmf_eval(V3(R->mu[0][k], R->mu[1][k], R->mu[2][k]), kappa_clamp(R->kappa[k]), om);
            gs += g[k];
        }
        if (!(gs > 0) || isinf(gs)) continue;
        int nextvol = (sm->flags & VSPG_SAMPLE_NEXT_VOLUME) != 0;
        for (int k = 0; k < R->n_lobes; ++k) {
            float wg = w * (g[k] / gs);
            acc[reg].S[k] += wg;
            acc[reg].R[0][k] += wg * om.x; acc[reg].R[1][k] += wg * om.y; acc[reg].R[2][k] += wg * om.z;
            if (sm->distance > 0 && !isinf(sm->distance)) acc[reg].D[k] += wg / sm->distance;
            if (nextvol) { acc[reg].V[k] += wg; acc[reg].Qv[k] += wg * w; } else acc[reg].Qs[k] += wg * w;
        }
    }
    /* 5. M step */
    for (int i = 0; i < r->field[f].n_regions; ++i) {
        VspgFieldRegion *R = &regs[i];
        if (R->n_lobes <= 0) continue;
        struct region_stats *s1 = &st[i];
        float Stot = 0;
        for (int k = 0; k < R->n_lobes; ++k) {
            s1->S[k] += (float)acc[i].S[k];
            for (int a = 0; a < 3; ++a) s1->R[a][k] += (float)acc[i].R[a][k];
            s1->D[k] += (float)acc[i].D[k]; s1->V[k] += (float)acc[i].V[k];
            s1->Qv[k] += (float)acc[i].Qv[k]; s1->Qs[k] += (float)acc[i].Qs[k];
            Stot += s1->S[k];
        }
        if (!(Stot > 0)) continue;
        const float floorw = 1e-3f / VSPG_FIELD_LOBES;
        float wsum = 0;
        for (int k = 0; k < R->n_lobes; ++k) {
            float wk = s1->S[k] / Stot;
            wk = wk < floorw ? floorw : wk;
            R->weight[k] = wk;
            wsum += wk;
            float rl = sqrtf(s1->R[0][k] * s1->R[0][k] + s1->R[1][k] * s1->R[1][k] + s1->R[2][k] * s1->R[2][k]);
            if (s1->S[k] > 0 && rl > 0) {
                R->mu[0][k] = s1->R[0][k] / rl; R->mu[1][k] = s1->R[1][k] / rl; R->mu[2][k] = s1->R[2][k] / rl;
                float rbar = rl / s1->S[k];
                rbar = rbar > 0.9999f ? 0.9999f : rbar;
                R->kappa[k] = kappa_clamp(rbar * (3 - rbar * rbar) / (1 - rbar * rbar));
                R->distance[k] = s1->D[k] > 0 ? s1->S[k] / s1->D[k] : INFINITY;
                if (r->prm.vspcriterion == VSPG_VSP_VARIANCE) {
                    float qv = sqrtf(s1->Qv[k]), qs = sqrtf(s1->Qs[k]);
                    R->vsp[k] = qv + qs > 0 ? qv / (qv + qs) : 0.5f;
                } else {
                    R->vsp[k] = s1->V[k] / s1->S[k];
                }
            }
        }
        for (int k = 0; k < R->n_lobes; ++k) R->weight[k] = R->weight[k] / wsum;
    }
    free(acc);
}
/* PostProcessWave's training branch (:234-248) */
static void field_update(OracleRenderer *r) {
    if (r->n_samples > TRAIN_MIN_UPDATE_SAMPLES) {
        double sw = 0;
        for (size_t i = 0; i < r->n_samples; ++i) sw += r->samples[i].weight;
        float mean_w = (float)(sw / (double)r->n_samples);
        field_update_one(r, 0, mean_w);
        field_update_one(r, 1, mean_w);
        r->field_iteration++;
        if (r->field_iteration >= r->prm.guide_num_training_waves) r->training = 0;
    }
    r->n_samples = 0;
    r->n_zero_samples = 0;
}
int oracle_renderer_training_stats(OracleRenderer *r, VspgTrainStats *out) {
    if (!r || !out) return VSPG_EINVAL;
    memset(out, 0, sizeof *out);
    out->training = r->training; out->iteration = r->field_iteration;
    out->n_samples = r->n_samples; out->n_zero = r->n_zero_samples;
    for (int f = 0; f < 2; ++f) { out->n_nodes[f] = r->field[f].n_nodes; out->n_regions[f] = r->field[f].n_regions; }
    return 0;
}
int oracle_train_samples_read(OracleRenderer *r, VspgTrainSample *out, size_t max_samples, size_t *n_out) {
    if (!r || !n_out) return VSPG_EINVAL;
    size_t n = r->n_samples < max_samples ? r->n_samples : max_samples;
    if (out && n) memcpy(out, r->samples, n * sizeof(VspgTrainSample));
    *n_out = r->n_samples;
    return 0;
}
int oracle_renderer_get_guiding_field(OracleRenderer *r, int volume_field, VspgKdNode *nodes, VspgFieldRegion *regions,
                                      int32_t *n_nodes, int32_t *n_regions) {
    if (!r || !n_nodes || !n_regions) return VSPG_EINVAL;
    int f = volume_field ? 1 : 0;
    *n_nodes = r->field[f].n_nodes; *n_regions = r->field[f].n_regions;
    if (nodes && r->field[f].nodes) memcpy(nodes, r->field[f].nodes, sizeof(VspgKdNode) * (size_t)r->field[f].n_nodes);
    if (regions && r->field[f].regions) memcpy(regions, r->field[f].regions, sizeof(VspgFieldRegion) * (size_t)r->field[f].n_regions);
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* EvaluatePixelSample (src/pbrt/cpu/integrators.cpp:272-334) + film + ISG sample          */
/* ------------------------------------------------------------------------------------ */
static void camera_ray(const VspgCamera *cam, float fx, float fy, v3 *o, v3 *d) {
    v3 pc = V3(cam->sx * fx + cam->ox, cam->sy * fy + cam->oy, 1.f);
    v3 dir = v_normalize(pc);
    frame_t f;
    f.x = v3_from(cam->right); f.y = v3_from(cam->up); f.z = v3_from(cam->fwd);
    *o = v3_from(cam->origin);
    *d = frame_from_local(&f, dir);
}

static spec evaluate_pixel_sample(const OracleRenderer *r, int px, int py, int sampleIndex,
                                  isg_sample_t *isg, path_counters_t *pc, float *filterWeight, pathrec_t *rec) {
    sampler_t sampler;
    sampler_start_pixel_sample(&sampler, px, py, r->cfg.seed, sampleIndex, 0);
    float lu = sampler_get1d(&sampler);
    int ch = (int)floorf(lu * 3); /* SampledWavelengths::SampleVisible, spectrum.h:380-384 */
    if (ch > 2) ch = 2;
    float lambda[3];
    sample_visible_wavelengths(lu, lambda);
    /* GetCameraSample (samplers.h:796-815) with BoxFilter radius 0.5 (filters.h:67-69) */
    float f0 = sampler_get1d(&sampler), f1 = sampler_get1d(&sampler);
    float fpx = (1 - f0) * -0.5f + f0 * 0.5f;
    float fpy = (1 - f1) * -0.5f + f1 * 0.5f;
    float pfx = ((float)px + fpx) + 0.5f, pfy = ((float)py + fpy) + 0.5f;
    (void)sampler_get1d(&sampler); /* time */
    (void)sampler_get1d(&sampler); (void)sampler_get1d(&sampler); /* pLens */
    *filterWeight = 1.f;
    v3 o, d;
    camera_ray(&r->scene.camera, pfx, pfy, &o, &d);
    if (rec) { rec->n = 0; rec->cur = -1; rec->cap = rec_capacity(r->prm.maxdepth); }
    spec L = Li(r, px, py, o, d, ch, lambda, &sampler, isg, pc, rec);
    /* L = cameraRay->weight * L with weight 1; NaN / Inf -> black (:308-318) */
    if (s_has_nan(L)) L = S1(0.f);
    else if (s_has_inf(L)) L = S1(0.f);
    return L;
}

/* image-space VSP statistics: own design standing in for ImageSpaceGuidingBuffer::AddSample
 * (guidedvolpathvspgintegrator.cpp:613-622).  PARITY UNPINNED (OpenPGL absent).
 * stats[0]=n  [1]=sum c*[vol]  [2]=sum c*[surf]  [3]=sum c^2*q*[vol]  [4]=sum c^2*(1-q)*[surf]
 * [5]=n_vol  [6],[7] reserved; c = average of the RGB contribution, q = VSP used (0.5 if unguided) */
static void isg_add_sample(float *st, spec L, const isg_sample_t *isg) {
    if (!isg->valid) return;
    float c = s_avg(L);
    float q = isg->vsp_used >= 0.f ? isg->vsp_used : 0.5f;
    st[0] += 1.f;
    if (isg->surface_event) {
        st[2] += c;
        st[4] += c * c * (1 - q);
    } else {
        st[1] += c;
        st[3] += c * c * q;
        st[5] += 1.f;
    }
}

static void film_add_sample(double *px, spec L, float weight) { /* film.h:251-267 */
    /* sensor->ToSensorRGB: imagingRatio 1; maxComponentValue = Infinity */
    for (int c = 0; c < 3; ++c) px[c] += (double)(weight * L.c[c]);
    px[3] += (double)weight;
}

static void counters_merge(VspgCounters *dst, const path_counters_t *pc, uint64_t paths) {
    dst->paths += paths;
    dst->segments += pc->segments;
    dst->volume_scatters += pc->volume_scatters;
    dst->surface_hits += pc->surface_hits;
    dst->density_queries += pc->density_queries;
    dst->shadow_rays += pc->shadow_rays;
    dst->shadow_density_queries += pc->shadow_queries;
}

int oracle_render_window(OracleRenderer *r, int x0, int y0, int x1, int y1, int wave_start,
                         int wave_end, int nthreads) {
    if (!r || wave_end < wave_start) return VSPG_EINVAL;
    int W = r->cfg.xres;
    int ntx = (x1 - x0 + 15) / 16, nty = (y1 - y0 + 15) / 16;
    int ntiles = ntx * nty;
    VspgCounters total;
    memset(&total, 0, sizeof total);
    r->in_wave = 1;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
    {
        path_counters_t pc;
        memset(&pc, 0, sizeof pc);
        uint64_t paths = 0;
        pathrec_t rec;
        rec.n = 0; rec.cur = -1;
#pragma omp for schedule(dynamic, 1)
        for (int tile = 0; tile < ntiles; ++tile) {
            int tx = tile % ntx, ty = tile / ntx;
            int xa = x0 + tx * 16, ya = y0 + ty * 16;
            int xb = xa + 16 < x1 ? xa + 16 : x1, yb = ya + 16 < y1 ? ya + 16 : y1;
            for (int py = ya; py < yb; ++py)
                for (int px = xa; px < xb; ++px)
                    for (int s = wave_start; s < wave_end; ++s) {
                        if (r->cfg.shard_count > 1 && (s % r->cfg.shard_count) != r->cfg.shard_index) continue;
                        isg_sample_t isg;
                        float w;
                        spec L = evaluate_pixel_sample(r, px, py, s, &isg, &pc, &w, r->training ? &rec : NULL);
                        if (r->training) propagate_samples(r, &rec);
                        size_t idx = (size_t)py * W + px;
                        film_add_sample(&r->film[idx * 4], L, w);
                        isg_add_sample(&r->isg_stats[idx * VSPG_ISG_STATS], L, &isg);
                        paths++;
                    }
        }
#pragma omp critical
        counters_merge(&total, &pc, paths);
    }
    counters_merge(&r->counters, &(path_counters_t){total.segments, total.volume_scatters, total.surface_hits,
                                                    total.density_queries, total.shadow_rays, total.shadow_density_queries},
                   total.paths);
    r->in_wave = 0;
    return 0;
}

int oracle_render_wave(OracleRenderer *r, int wave_start, int wave_end, int nthreads) {
    return oracle_render_window(r, 0, 0, r->cfg.xres, r->cfg.yres, wave_start, wave_end, nthreads);
}

/* PostProcessWave (guidedvolpathvspgintegrator.cpp:230-260): the guiding-field update (its
 * training branch: field_update above); VSP buffer update at waveCounter == 2^bufferWave.
 * Update() = 5x5 box filter over the sufficient statistics, then the criterion (own design). */
#define ISG_FILTER_RADIUS 2
static int isg_update_due(const OracleRenderer *r, int n_waves) {
    return (double)(r->wave_counter + n_waves) >= pow(2.0, (double)r->buffer_wave);
}
int oracle_isg_update_due(OracleRenderer *r, int n_waves) {
    const int do_vsp = r->prm.vspguiding && r->prm.vspprimaryguiding && !r->vsp_loaded;
    return (do_vsp || r->prm.rrguiding) && isg_update_due(r, n_waves);
}
/* PostProcessWave after a step of n_waves sample indices (a sharded render: n_waves = ranks); stats_sum = the statistics
 * summed over all ranks, or NULL for the renderer's own.  n_waves = 1, NULL: the reference's schedule (:251). */
int oracle_post_process_step(OracleRenderer *r, int n_waves, const float *stats_sum) {
    const int due = isg_update_due(r, n_waves);
    const float *stats = stats_sum ? stats_sum : r->isg_stats;
    r->wave_counter += n_waves;
    if (r->training) field_update(r); /* :234-246 */
    r->n_samples = 0;                 /* guiding_sampleStorage->Clear() (:248) */
    r->n_zero_samples = 0;
    if (due) {
        int W = r->cfg.xres, H = r->cfg.yres;
        const int do_vsp = r->prm.vspguiding && r->prm.vspprimaryguiding && !r->vsp_loaded;
        const int do_contrib = r->prm.rrguiding != 0; /* cfg.EnableContributionEstimate(guideRR) (:164-168) */
        if (do_vsp || do_contrib) {
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    float a[5] = {0, 0, 0, 0, 0};
                    for (int dy = -ISG_FILTER_RADIUS; dy <= ISG_FILTER_RADIUS; ++dy)
                        for (int dx = -ISG_FILTER_RADIUS; dx <= ISG_FILTER_RADIUS; ++dx) {
                            int xx = x + dx, yy = y + dy;
                            if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                            const float *st = &stats[((size_t)yy * W + xx) * VSPG_ISG_STATS];
                            for (int k = 0; k < 5; ++k) a[k] += st[k];
                        }
                    if (do_vsp) {
                        float vsp = -1.f;
                        if (a[0] > 0) {
                            float v, s;
                            if (r->prm.vspcriterion == VSPG_VSP_VARIANCE) {
                                v = sqrtf(a[3] / a[0]);
                                s = sqrtf(a[4] / a[0]);
                            } else {
                                v = a[1] / a[0];
                                s = a[2] / a[0];
                            }
                            if (v + s > 0) vsp = v / (v + s);
                        }
                        r->vsp[(size_t)y * W + x] = vsp;
                    }
                    if (do_contrib) {
                        /* contribution estimate (own stand-in): filtered mean of the samples' average radiance, the same value
                         * in the three channels; 0 = no estimate */
                        float ce = a[0] > 0 ? (a[1] + a[2]) / a[0] : 0.f;
                        float *dst = &r->contrib[((size_t)y * W + x) * 3];
                        dst[0] = dst[1] = dst[2] = ce;
                    }
                }
            if (do_vsp) r->vsp_ready = 1;
            if (do_contrib) r->contrib_ready = 1;
        }
        while (pow(2.0, (double)r->buffer_wave) <= (double)r->wave_counter) r->buffer_wave++;
    }
    return 0;
}
int oracle_post_process_wave(OracleRenderer *r) { return oracle_post_process_step(r, 1, NULL); }


/* ------------------------------------------------------------------------------------ */
/* life cycle + accessors                                                                 */
/* ------------------------------------------------------------------------------------ */
static int validate_params(const VspgScene *scene, const VspgIntegratorParams *p, const VspgRenderConfig *cfg) {
    if (!scene || !p || !cfg) return VSPG_EINVAL;
    if (cfg->xres <= 0 || cfg->yres <= 0) return VSPG_EINVAL;
    if (scene->n_quads < 0 || scene->n_quads > VSPG_MAX_QUADS) return VSPG_EINVAL;
    if (scene->n_spheres < 0 || scene->n_spheres > VSPG_MAX_SPHERES) return VSPG_EINVAL;
    if (medium_is_grid_like(scene->medium.type)) {
        const VspgMedium *m = &scene->medium;
        if (m->nx <= 0 || m->ny <= 0 || m->nz <= 0 || !m->density) return VSPG_EINVAL;
        if (m->type == VSPG_MEDIUM_NANOVDB && !(m->voxel_size[0] > 0 && m->voxel_size[1] > 0 && m->voxel_size[2] > 0 && m->majorant_scale > 0))
            return VSPG_EINVAL;
        if (m->Le[0] != 0 || m->Le[1] != 0 || m->Le[2] != 0) {
            if (m->type == VSPG_MEDIUM_NANOVDB) return VSPG_EINVAL; /* NanoVDBMedium emits through its temperature grid only */
            if (m->temperature) return VSPG_EINVAL; /* "Both \"Le\" and \"temperature\" values were provided." (media.cpp:307-308) */
        }
        if (m->type == VSPG_MEDIUM_GRID && (m->temperature || m->Le[0] != 0 || m->Le[1] != 0 || m->Le[2] != 0))
            if (m->le_scale && (m->le_nx <= 0 || m->le_ny <= 0 || m->le_nz <= 0)) return VSPG_EINVAL;
        /* temperature grids (media.h:333-341, :724-735): volume emission is sampled by the delta-tracking routine only (:895-906) --
         * under "resampling" a heterogeneous medium never evaluates it */
    }
    return 0;
}

int oracle_renderer_create(const VspgScene *scene, const VspgIntegratorParams *params,
                           const VspgRenderConfig *cfg, OracleRenderer **out) {
    int rc = validate_params(scene, params, cfg);
    if (rc) return rc;
    OracleRenderer *r = (OracleRenderer *)calloc(1, sizeof *r);
    r->scene = *scene;
    r->prm = *params;
    r->cfg = *cfg;
    if (r->cfg.shard_count < 1) { r->cfg.shard_count = 1; r->cfg.shard_index = 0; }
    r->n_quads = scene->n_quads;
    for (int i = 0; i < r->n_quads; ++i) {
        quad_init(&r->quads[i], &scene->quads[i]);
        r->light_of_quad[i] = -1;
        if (r->quads[i].is_light) { r->light_of_quad[i] = r->n_lights; r->light_quads[r->n_lights++] = i; }
    }
    if (scene->n_triangles > 0 && scene->tri_p) { /* f1: what InteractionFromIntersection derives per triangle (shapes.h:888-938) */
        r->tris = (rtri_t *)calloc((size_t)scene->n_triangles, sizeof(rtri_t));
        for (int i = 0; i < scene->n_triangles; ++i) {
            const float *q = scene->tri_p + 9 * (size_t)i;
            v3 p0 = V3(q[0], q[1], q[2]), p1 = V3(q[3], q[4], q[5]), p2 = V3(q[6], q[7], q[8]);
            if (v_len2(v_cross(v_sub(p2, p0), v_sub(p1, p0))) == 0) continue; /* IntersectTriangle never reports it */
            v3 dp02 = v_sub(p0, p2), dp12 = v_sub(p1, p2);
            const float duv02[2] = {0.f - 1.f, 0.f - 1.f}, duv12[2] = {1.f - 1.f, 0.f - 1.f}; /* default uv (0,0), (1,0), (1,1) */
            float determinant = diff_of_products(duv02[0], duv12[1], duv02[1], duv12[0]);
            if (fabsf(determinant) < 1e-9f) continue;
            float invdet = 1 / determinant;
            v3 dpdu = V3(diff_of_products(duv12[1], dp02.x, duv02[1], dp12.x) * invdet, diff_of_products(duv12[1], dp02.y, duv02[1], dp12.y) * invdet,
                         diff_of_products(duv12[1], dp02.z, duv02[1], dp12.z) * invdet);
            v3 dpdv = V3(diff_of_products(duv02[0], dp12.x, duv12[0], dp02.x) * invdet, diff_of_products(duv02[0], dp12.y, duv12[0], dp02.y) * invdet,
                         diff_of_products(duv02[0], dp12.z, duv12[0], dp02.z) * invdet);
            if (v_len2(v_cross(dpdu, dpdv)) == 0) continue; /* (the reference's CoordinateSystem fallback: such slivers are dropped on both sides) */
            rtri_t *T = &r->tris[r->n_tris++];
            T->p0 = p0; T->p1 = p1; T->p2 = p2;
            T->n = v_normalize(v_cross(dp02, dp12));
            T->dpdu_n = v_normalize(dpdu);
            T->id = i;
            for (int k = 0; k < 3; ++k) {
                float v = scene->tri_kd ? scene->tri_kd[3 * (size_t)i + k] : 0.5f;
                T->Kd.c[k] = v < 0 ? 0 : (v > 1 ? 1 : v);
            }
            T->has_bsdf_lobes = T->Kd.c[0] != 0 || T->Kd.c[1] != 0 || T->Kd.c[2] != 0;
            const int fl = scene->tri_flags ? scene->tri_flags[i] : 0;
            T->has_material = !(fl & VSPG_TRI_INTERFACE);
            T->iface = iface_transition(fl >> VSPG_TRI_IFACE_SHIFT);
            if (fl & VSPG_TRI_FLIP_NORMAL) T->n = v_neg(T->n); /* reverseOrientation ^ transformSwapsHandedness (shapes.h:934-936) */
        }
        r->scene.tri_p = NULL; r->scene.tri_kd = NULL; r->scene.tri_flags = NULL;
    }
    r->n_spheres = scene->n_spheres;
    for (int i = 0; i < r->n_spheres; ++i) {
        rc = sphere_init(&r->spheres[i], &scene->spheres[i]);
        if (rc) { free(r->tris); free(r); return rc; }
    }
    r->has_medium = scene->medium.type != VSPG_MEDIUM_NONE;
    r->camera_in_medium = r->has_medium && !scene->camera_outside_medium;
    r->has_boundaries = scene->camera_outside_medium != 0;
    for (int i = 0; i < r->n_quads; ++i) r->has_boundaries |= !r->quads[i].has_material || r->quads[i].iface;
    for (int i = 0; i < r->n_tris; ++i) r->has_boundaries |= !r->tris[i].has_material || r->tris[i].iface;
    for (int i = 0; i < r->n_spheres; ++i) r->has_boundaries |= !r->spheres[i].has_material || r->spheres[i].iface;
    r->n_inf = scene->n_infinite_lights;
    { /* scene bounds = union of every primitive's bounds (the rectangles' corners, the triangles' vertices), BoundingSphere */
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = 0; i < r->n_quads; ++i) {
            const v3 c[4] = {r->quads[i].p00, r->quads[i].p10, r->quads[i].p01, r->quads[i].p11};
            for (int j = 0; j < 4; ++j) {
                const float q[3] = {c[j].x, c[j].y, c[j].z};
                for (int k = 0; k < 3; ++k) { lo[k] = q[k] < lo[k] ? q[k] : lo[k]; hi[k] = q[k] > hi[k] ? q[k] : hi[k]; }
            }
        }
        for (int i = 0; i < r->n_tris; ++i) {
            const v3 c[3] = {r->tris[i].p0, r->tris[i].p1, r->tris[i].p2};
            for (int j = 0; j < 3; ++j) {
                const float q[3] = {c[j].x, c[j].y, c[j].z};
                for (int k = 0; k < 3; ++k) { lo[k] = q[k] < lo[k] ? q[k] : lo[k]; hi[k] = q[k] > hi[k] ? q[k] : hi[k]; }
            }
        }
        for (int i = 0; i < r->n_spheres; ++i) { /* Sphere::Bounds (shapes.cpp:33-36): the eight corners of the object-space box, transformed */
            const rsphere_t *S = &r->spheres[i];
            for (int c = 0; c < 8; ++c) {
                const float x = (c & 1) ? S->radius : -S->radius, y = (c & 2) ? S->radius : -S->radius, z = (c & 4) ? S->radius : -S->radius;
                for (int k = 0; k < 3; ++k) {
                    const float *row = S->m + 4 * k;
                    const float q = row[0] * x + row[1] * y + row[2] * z + row[3];
                    lo[k] = q < lo[k] ? q : lo[k]; hi[k] = q > hi[k] ? q : hi[k];
                }
            }
        }
        r->scene_radius = 0.f;
        if (lo[0] <= hi[0]) {
            const float cx = (lo[0] + hi[0]) / 2, cy = (lo[1] + hi[1]) / 2, cz = (lo[2] + hi[2]) / 2;
            const float dx = cx - hi[0], dy = cy - hi[1], dz = cz - hi[2];
            r->scene_radius = sqrtf(dx * dx + dy * dy + dz * dz); /* Distance(center, pMax) */
        }
    }
    r->lsamp = (lightsampler_t *)calloc(1, sizeof(lightsampler_t));
    lightsampler_build(r, r->lsamp);
    if (medium_is_grid_like(scene->medium.type)) {
        const VspgMedium *m = &scene->medium;
        size_t n = (size_t)m->nx * m->ny * m->nz;
        r->density = (float *)malloc(n * sizeof(float));
        memcpy(r->density, m->density, n * sizeof(float));
        r->scene.medium.density = r->density;
        const int MR = medium_maj_res(m->type);
        r->majorant = (float *)calloc((size_t)MR * MR * MR, sizeof(float));
        if (m->type == VSPG_MEDIUM_NANOVDB) build_majorant_grid_nvdb(r);
        else build_majorant_grid(r);
        if (m->temperature) { /* same number of samples as the density grid (media.cpp:283-288; the .nvdb reader checks the bounding boxes) */
            r->temperature = (float *)malloc(n * sizeof(float));
            memcpy(r->temperature, m->temperature, n * sizeof(float));
            r->scene.medium.temperature = r->temperature;
        }
        /* isEmissive = temperatureGrid ? true : Le_spec.MaxValue() > 0 (media.cpp:261) */
        if (m->type == VSPG_MEDIUM_GRID && (m->temperature || m->Le[0] != 0 || m->Le[1] != 0 || m->Le[2] != 0)) {
            if (m->le_scale) {
                size_t ln = (size_t)m->le_nx * m->le_ny * m->le_nz;
                r->le_scale = (float *)malloc(ln * sizeof(float));
                memcpy(r->le_scale, m->le_scale, ln * sizeof(float));
                r->le_dim[0] = m->le_nx; r->le_dim[1] = m->le_ny; r->le_dim[2] = m->le_nz;
            } else { /* "Lescale" absent: SampledGrid({1}, 1, 1, 1) (media.cpp:319-320, normalisation applied by the caller) */
                r->le_scale = (float *)malloc(sizeof(float));
                r->le_scale[0] = 1.f;
                r->le_dim[0] = r->le_dim[1] = r->le_dim[2] = 1;
            }
            r->scene.medium.le_scale = r->le_scale;
        }
    }
    size_t npix = (size_t)cfg->xres * cfg->yres;
    r->film = (double *)calloc(npix * 4, sizeof(double));
    r->isg_stats = (float *)calloc(npix * VSPG_ISG_STATS, sizeof(float));
    r->vsp = (float *)calloc(npix, sizeof(float));
    if (params->rrguiding) { /* :195-197 */
        r->prm.minrrdepth = 1;
        r->contrib = (float *)calloc(npix * 3, sizeof(float));
    }
    /* calculateTrBuffer (:190-193); a buffer handed over through oracle_tr_buffer_write is trBufferLoad (:180-188) */
    r->tr_calc = params->storeTrBuffer || (params->vspguiding && params->vspprimaryguiding &&
                                           params->vspsamplingmethod == VSPG_VSP_NDS && params->collisionProbabilityBias);
    if (r->tr_calc) {
        r->trbuf = (float *)calloc(npix * 3, sizeof(float));
        r->tr_spp = (int *)calloc(npix, sizeof(int));
    }
    /* guideTraining (:109).  The reference also trains when only the guided-RR flags are set (they default to
     * true); this build trains iff the field will be queried. */
    r->training = params->surfaceguiding || params->volumeguiding || params->vspsecondaryguiding;
    if (r->training) {
        field_alloc(r, 0);
        field_alloc(r, 1);
    }
    *out = r;
    return 0;
}
void oracle_renderer_destroy(OracleRenderer *r) {
    if (!r) return;
    free_field(r, 0); free_field(r, 1);
    free(r->samples);
    free(r->trbuf); free(r->tr_spp); free(r->le_scale); free(r->temperature); free(r->contrib);
    free(r->film); free(r->isg_stats); free(r->vsp); free(r->density); free(r->majorant); free(r->tris); free(r->lsamp); free(r);
}
void oracle_film_read(OracleRenderer *r, float *rgbw) {
    size_t n = (size_t)r->cfg.xres * r->cfg.yres * 4;
    for (size_t i = 0; i < n; ++i) rgbw[i] = (float)r->film[i];
}
void oracle_film_read_f64(OracleRenderer *r, double *rgbw) {
    memcpy(rgbw, r->film, (size_t)r->cfg.xres * r->cfg.yres * 4 * sizeof(double));
}
void oracle_film_clear(OracleRenderer *r) { memset(r->film, 0, (size_t)r->cfg.xres * r->cfg.yres * 4 * sizeof(double)); }
void oracle_vsp_buffer_read(OracleRenderer *r, float *vsp, int *is_ready) {
    memcpy(vsp, r->vsp, (size_t)r->cfg.xres * r->cfg.yres * sizeof(float));
    if (is_ready) *is_ready = r->vsp_ready;
}
void oracle_vsp_buffer_write(OracleRenderer *r, const float *vsp, int is_ready) {
    memcpy(r->vsp, vsp, (size_t)r->cfg.xres * r->cfg.yres * sizeof(float));
    r->vsp_ready = is_ready;
}
void oracle_vsp_buffer_load(OracleRenderer *r, const float *vsp) {
    memcpy(r->vsp, vsp, (size_t)r->cfg.xres * r->cfg.yres * sizeof(float));
    r->vsp_ready = 1;
    r->vsp_loaded = 1;
}
int oracle_tr_buffer_read(OracleRenderer *r, float *rgb) {
    if (!r->trbuf) return VSPG_EINVAL;
    memcpy(rgb, r->trbuf, (size_t)r->cfg.xres * r->cfg.yres * 3 * sizeof(float));
    return 0;
}
int oracle_tr_buffer_write(OracleRenderer *r, const float *rgb) { /* TrBuffer(fileName) (:180-185) */
    size_t npix = (size_t)r->cfg.xres * r->cfg.yres;
    if (!r->trbuf) r->trbuf = (float *)calloc(npix * 3, sizeof(float));
    memcpy(r->trbuf, rgb, npix * 3 * sizeof(float));
    r->tr_load = 1;
    r->tr_calc = 0;
    return 0;
}
void oracle_isg_stats_read(OracleRenderer *r, float *stats) {
    memcpy(stats, r->isg_stats, (size_t)r->cfg.xres * r->cfg.yres * VSPG_ISG_STATS * sizeof(float));
}
void oracle_get_counters(OracleRenderer *r, VspgCounters *out) { *out = r->counters; }
void oracle_reset_counters(OracleRenderer *r) { memset(&r->counters, 0, sizeof r->counters); }

int oracle_trace_paths(OracleRenderer *r, int n, const int32_t *pixel_xy, const int32_t *sample_index,
                       float *out_L, int32_t *out_segments) {
    for (int i = 0; i < n; ++i) {
        g_dbg_flags = 0;
        path_counters_t pc;
        memset(&pc, 0, sizeof pc);
        isg_sample_t isg;
        float w;
        spec L = evaluate_pixel_sample(r, pixel_xy[2 * i], pixel_xy[2 * i + 1], sample_index[i], &isg, &pc, &w, NULL);
        out_L[3 * i] = L.c[0]; out_L[3 * i + 1] = L.c[1]; out_L[3 * i + 2] = L.c[2];
        if (out_segments) out_segments[i] = (int32_t)pc.segments;
    }
    return 0;
}

/* recording callback for the free-flight batch driver */
typedef struct {
    int ch, stop_after;
    VspgTmajResult *res;
    v3 ro, rd_n;
} rec_ctx_t;
static int rec_cb(void *vctx, v3 p, const medium_props_t *mp, spec sigma_maj, spec T_maj, int nds) {
    (void)T_maj; (void)nds;
    rec_ctx_t *c = (rec_ctx_t *)vctx;
    VspgTmajResult *o = c->res;
    o->n_callbacks++;
    o->last_p[0] = p.x; o->last_p[1] = p.y; o->last_p[2] = p.z;
    o->last_t = v_dot(v_sub(p, c->ro), c->rd_n);
    spec sigma_t = s_add(mp->sigma_s, mp->sigma_a);
    o->sum_sigt_over_maj += sigma_t.c[c->ch] / sigma_maj.c[c->ch];
    if (c->stop_after > 0 && o->n_callbacks >= c->stop_after) return 0;
    return 1;
}
int oracle_sample_tmaj_batch(OracleRenderer *r, int variant, int n, const VspgTmajQuery *q,
                             VspgTmajResult *out) {
    for (int i = 0; i < n; ++i) {
        VspgTmajResult *o = &out[i];
        memset(o, 0, sizeof *o);
        o->last_t = -1.f;
        for (int k = 0; k < 3; ++k) o->r_u_factor[k] = 1.f;
        o->majorant_scale = 1.f;
        rng_t rng;
        rng_set_sequence2(&rng, oracle_hash_float(q[i].rng_a), oracle_hash_float(q[i].rng_b));
        rec_ctx_t c;
        c.ch = q[i].channel; c.stop_after = q[i].stop_after; c.res = o;
        c.ro = v3_from(q[i].o); c.rd_n = v_normalize(v3_from(q[i].d));
        int guide = q[i].vsp >= 0.f;
        float vsp = guide ? fmaxf(fminf(q[i].vsp, 0.999f), 0.001f) : q[i].vsp;
        spec T, bf = S1(1.f), rf = S1(1.f);
        if (variant == VSPG_TMAJ_PLAIN) {
            T = sample_T_maj(r, c.ro, v3_from(q[i].d), q[i].tMax, q[i].u, &rng, c.ch, rec_cb, &c);
        } else if (variant == VSPG_TMAJ_OPTICAL_DEPTH) {
            T = sample_T_maj_ods(r, c.ro, v3_from(q[i].d), q[i].tMax, q[i].u, &rng, c.ch, guide, vsp,
                                 r->prm.vspmisratio, r->prm.vspsamplingmethod == VSPG_VSP_NDS, &bf, &rf, rec_cb, &c);
        } else if (variant == VSPG_TMAJ_RESAMPLING) {
            float vrc = vsp, ms = 1.f;
            T = sample_T_maj_resampling(r, c.ro, v3_from(q[i].d), q[i].tMax, q[i].u, &rng, c.ch, guide, vsp,
                                        &vrc, &ms, rec_cb, &c);
            o->vrc = vrc; o->majorant_scale = ms;
        } else {
            return VSPG_EINVAL;
        }
        for (int k = 0; k < 3; ++k) { o->T_maj[k] = T.c[k]; o->r_u_factor[k] = rf.c[k]; }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* scene helpers                                                                          */
/* ------------------------------------------------------------------------------------ */
void oracle_integrator_params_default(VspgIntegratorParams *p) {
    /* guidedvolpathvspgintegrator.cpp:1263-1319 */
    memset(p, 0, sizeof *p);
    p->maxdepth = 5; p->minrrdepth = 1; p->usenee = 1;
    p->surfaceguiding = 1; p->volumeguiding = 1;
    p->surfaceguidingtype = VSPG_GUIDE_RIS; p->volumeguidingtype = VSPG_GUIDE_MIS;
    p->vspguiding = 1; p->vspprimaryguiding = 1; p->vspsecondaryguiding = 1;
    p->vspmisratio = 0.5f;
    p->vspcriterion = VSPG_VSP_VARIANCE;
    p->vspsamplingmethod = VSPG_VSP_RESAMPLING;
    p->collisionProbabilityBias = 0; p->rrguiding = 0;
    p->lightsampler = VSPG_LIGHTSAMPLER_BVH; p->regularize = 0;
    p->guide_num_training_waves = 128;
    p->surfacerrguiding = 1; p->volumerrguiding = 1;
}

int oracle_camera_look_at(VspgCamera *cam, const float eye[3], const float look[3], const float up[3],
                          float fov_degrees, int xres, int yres) {
    double e[3] = {eye[0], eye[1], eye[2]}, l[3] = {look[0], look[1], look[2]}, u[3] = {up[0], up[1], up[2]};
    double f[3] = {l[0] - e[0], l[1] - e[1], l[2] - e[2]};
    double fl = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    if (fl == 0) return VSPG_EINVAL;
    for (int i = 0; i < 3; ++i) f[i] /= fl;
    /* pbrt LookAt (util/transform.cpp): right = Normalize(Cross(Normalize(up), dir)) */
    double ul = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    for (int i = 0; i < 3; ++i) u[i] /= ul;
    double rt[3] = {u[1] * f[2] - u[2] * f[1], u[2] * f[0] - u[0] * f[2], u[0] * f[1] - u[1] * f[0]};
    double rl = sqrt(rt[0] * rt[0] + rt[1] * rt[1] + rt[2] * rt[2]);
    if (rl == 0) return VSPG_EINVAL;
    for (int i = 0; i < 3; ++i) rt[i] /= rl;
    double nu[3] = {f[1] * rt[2] - f[2] * rt[1], f[2] * rt[0] - f[0] * rt[2], f[0] * rt[1] - f[1] * rt[0]};
    for (int i = 0; i < 3; ++i) {
        cam->origin[i] = (float)e[i]; cam->right[i] = (float)rt[i]; cam->up[i] = (float)nu[i]; cam->fwd[i] = (float)f[i];
    }
    /* screen window [-a,a]x[-1,1] (a = aspect if >1) scaled by tan(fov/2) on the shorter axis
     * (cameras.cpp:474-489); raster y grows downwards (cameras.h:268-274) */
    double aspect = (double)xres / (double)yres;
    double sxw = aspect > 1 ? aspect : 1.0, syw = aspect > 1 ? 1.0 : 1.0 / aspect;
    double th = tan(fov_degrees * 3.14159265358979323846 / 360.0);
    cam->sx = (float)(2.0 * sxw * th / xres);
    cam->ox = (float)(-sxw * th);
    cam->sy = (float)(-2.0 * syw * th / yres);
    cam->oy = (float)(syw * th);
    return 0;
}

static void set_quad(VspgQuad *q, float px, float py, float pz, float ax, float ay, float az, float bx,
                     float by, float bz, float kd, float ler, float leg, float leb) {
    memset(q, 0, sizeof *q);
    q->p00[0] = px; q->p00[1] = py; q->p00[2] = pz;
    q->e1[0] = ax; q->e1[1] = ay; q->e1[2] = az;
    q->e2[0] = bx; q->e2[1] = by; q->e2[2] = bz;
    q->Kd[0] = q->Kd[1] = q->Kd[2] = kd;
    q->Le[0] = ler; q->Le[1] = leg; q->Le[2] = leb;
}

/* SURVEY.md App. F.  Quad normals (e1 x e2) point into the box. */
int oracle_scene_fog_box(VspgScene *s, int xres, int yres) {
    memset(s, 0, sizeof *s);
    const float k = 0.73f;
    s->n_quads = 7;
    set_quad(&s->quads[0], -1, -1, -1, 0, 0, 2, 2, 0, 0, k, 0, 0, 0);  /* floor   y=-1, n=+y */
    set_quad(&s->quads[1], -1, 1, -1, 2, 0, 0, 0, 0, 2, k, 0, 0, 0);   /* ceiling y=+1, n=-y */
    set_quad(&s->quads[2], -1, -1, 1, 0, 2, 0, 2, 0, 0, k, 0, 0, 0);   /* back    z=+1, n=-z */
    set_quad(&s->quads[3], -1, -1, -1, 2, 0, 0, 0, 2, 0, k, 0, 0, 0);  /* front   z=-1, n=+z */
    set_quad(&s->quads[4], -1, -1, -1, 0, 2, 0, 0, 0, 2, k, 0, 0, 0);  /* left    x=-1, n=+x */
    set_quad(&s->quads[5], 1, -1, -1, 0, 0, 2, 0, 2, 0, k, 0, 0, 0);   /* right   x=+1, n=-x */
    set_quad(&s->quads[6], -0.25f, 0.999f, -0.25f, 0.5f, 0, 0, 0, 0, 0.5f, 0, 17, 12, 4); /* light, n=-y */
    float eye[3] = {0, 0, -0.95f}, look[3] = {0, 0, 0}, up[3] = {0, 1, 0};
    int rc = oracle_camera_look_at(&s->camera, eye, look, up, 60.f, xres, yres);
    if (rc) return rc;
    s->medium.type = VSPG_MEDIUM_HOMOGENEOUS;
    for (int i = 0; i < 3; ++i) { s->medium.sigma_a[i] = 0.05f; s->medium.sigma_s[i] = 0.45f; s->medium.Le[i] = 0; }
    s->medium.g = 0.f;
    return 0;
}
