// vspg_guiding.h -- device-side guiding-cache query.
//
// The wrapper logic (GuidedBSDF / GuidedPhaseFunction: init, MIS / RIS sampling, PDF,
// VolumeScatterProbability) follows src/pbrt/cpu/guiding.h:57-638 and lives in vspg_path.h.  This
// file is what sits behind the OpenPGL calls those wrappers make -- OpenPGL is absent from the
// reference tree, so it is this build's own design (parity unpinned):
//   Init(field, p, rand)                  kd-tree descent to the leaf region containing p
//   parallax-aware vMF mixture            lobe k re-aimed at pivot + distance_k * mu_k
//   ApplyCosineProduct(n)                 closed-form product with a kappa = 2.18853 vMF around n
//   ApplySingleLobeHenyeyGreensteinProduct closed-form product with a vMF of mean cosine |g|
//   PDF / SamplePDF / IncomingRadiancePDF / VolumeScatterProbability
// Layout: kd nodes (8 B) and regions (240 B) in HBM, read through L2; the per-lane mixture (8 floats x 8 lobes) lives in LDS for
// the lifetime of a vertex in the per-lane kernels, strided by the workgroup size so that lanes hit distinct banks, and in
// registers in the workgroup kernel.  Arithmetic uses +,-,*,/,sqrt, FastExp and the host-exact logf/sinf/cosf only, in the
// same order as the CPU checker, so results are bit-identical.
#pragma once
#include "vspg_device.h"

VSPG_NS_BEGIN

constexpr int GK = VSPG_FIELD_LOBES;
constexpr float kTwoPi = 6.28318530717958647692f;
constexpr float kCosineLobeKappa = 2.18853f;
constexpr float kGuidingProbability = 0.5f;  // guiding.h:348, 628
constexpr float kUniformIncomingRadiancePDF = (float)(1.0f / (4.0f * 3.14159265358979323846));  // guiding.h:179

#define VSPG_AS1 __attribute__((address_space(1)))
typedef float vspg_v4f __attribute__((ext_vector_type(4)));
typedef unsigned int vspg_v2u __attribute__((ext_vector_type(2)));
// 16-byte / 8-byte loads through a global-address-space pointer (builtin vector types: a class type cannot be copied out of
// another address space)
VDEV float4 ld4_global(const void *p) {
    const vspg_v4f v = *(const vspg_v4f VSPG_AS1 *)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
VDEV float vmf_norm(float kappa) { return kappa / (kTwoPi * (1 - fast_exp(-2 * kappa))); }
VDEV float vmf_eval(V3 mu, float kappa, V3 w) { return vmf_norm(kappa) * fast_exp(kappa * (dot(mu, w) - 1)); }
VDEV float kappa_clamp(float k) { return k < 1e-2f ? 1e-2f : (k > 1e4f ? 1e4f : k); }

// lds / n_lds: the first n_lds nodes of the tree staged in LDS (the workgroup kernel, guided builds).  Node numbers grow with
// creation time, so the low-numbered nodes are the upper levels: a descent takes its first ~10 steps at LDS latency (~64
// cycles each) instead of L2 latency (~500 each, dependent) and finishes in HBM/L2 for the last one or two.
VDEV int field_lookup(const DField &F, V3 p, const VspgKdNode *lds = nullptr, int n_lds = 0) {
    if (!F.nodes || F.n_nodes <= 0) return -1;
    uint32_t node = 0;
    for (int depth = 0; depth < 64; ++depth) {
        VspgKdNode nd;
        if ((int)node < n_lds) nd = lds[node];
        else {
            const vspg_v2u v = *(const vspg_v2u VSPG_AS1 *)(F.nodes + node);
            nd.split = __builtin_bit_cast(float, v.x);
            nd.packed = v.y;
        }
        uint32_t axis = nd.packed & 3u, idx = nd.packed >> 2;
        if (axis == 3u) return (int)idx < F.n_regions ? (int)idx : -1;
        float c = axis == 0 ? p.x : (axis == 1 ? p.y : p.z);
        node = idx + (c < nd.split ? 0u : 1u);
        if ((int)node >= F.n_nodes) return -1;
    }
    return -1;
}
// Region records are read four lobes at a time with 16-byte loads (the SoA rows of VspgFieldRegion and of DField::aux
// are 32 bytes, 16-byte aligned): the lanes of a wavefront sit in unrelated regions, so every load instruction
// costs one cache-line lookup per lane whatever its width -- the guided kernels were bound by exactly that.
static_assert(sizeof(VspgFieldRegion) % 16 == 0 && offsetof(VspgFieldRegion, weight) % 16 == 0 && GK % 4 == 0, "16-byte rows");
// (field data is read through GLOBAL-address-space pointers: through the generic pointers a kernel argument's members are, the
//  compiler emits flat loads and waits for each where it stands)
VDEV float4 ld4(const float *p) { return ld4_global(p); }
VDEV float ld1g(const float *p) { return *(const float VSPG_AS1 *)p; }
VDEV int ld1g(const int32_t *p) { return *(const int32_t VSPG_AS1 *)p; }
VDEV V3 ld3g(const float *p) { return V3{ld1g(p), ld1g(p + 1), ld1g(p + 2)}; }
VDEV float c4(float4 v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }  // j is a compile-time constant after unrolling
// the region's lobe re-aimed at the query point; normalised with ONE reciprocal (round 3: was three IEEE divisions)
// pp = pivot - p (one subtraction per vertex, not per lobe)
VDEV V3 lobe_dir(V3 pp, V3 mu, float d) {
    if (!(d > 0) || isinf_(d)) return mu;
    V3 t = pp + mu * d;
    float l2 = len2(t);
    if (!(l2 > 0)) return mu;
    const float inv = 1.0f / __builtin_sqrtf(l2);
    return t * inv;
}

// ---- the mixture a vertex works with (round 3 layout; arithmetic: oracle/vspg_oracle.c "a14", the comment above gdist_t) ----
// Per lobe k, 8 floats:  e 0-2 raw_k (the region's lobe re-aimed at the query point),  e 3 a_k (density weight of the product
// lobe, un-normalised),  e 4 kc_k (its concentration),  e 5-6 c1_k, c2_k (mu_k . w == c1 (raw_k . w) + c2 (m2 . w): the product's
// mean direction is never formed),  e 7 wn_k (normalised mass: lobe selection).  Per vertex: m2 and isum = 1 / SUM wo_k.
// DField::aux holds, per region, b_k = weight_k * vmf_norm(kappa_k) and the clamped kappa_k (k_field_aux, once per field update).
// Two homes for the per-lobe floats, same contents:
//   GStoreLds  per-lane columns of s_gmix, element e of lobe k at lds[(e * GK + k) * stride] (the per-lane kernels: a path lives in
//              a lane's registers for its whole life and has none to spare); a lane's column stays valid until its next gdist_init;
//   GStoreReg  registers (every loop over lobes fully unrolled with `k < n` predicates: constant indices).
constexpr int kGElems = 8;
constexpr int kGFloats = kGElems * GK;
constexpr int kGuideBlock = 256;  // threads per block of every kernel that keeps the guided scratch in LDS
__shared__ float s_gmix[kGuideBlock * kGFloats];
VDEV float *guide_lds() { return s_gmix + threadIdx.x; }
VDEV const float *region_aux(const DField &F, int region) { return F.aux + (size_t)region * (2 * GK); }
struct GStoreLds {
    float *lds;
    int stride;
    VDEV float get(int e, int k) const { return lds[(e * GK + k) * stride]; }
    VDEV void set(int e, int k, float v) { lds[(e * GK + k) * stride] = v; }
};
struct GStoreReg {
    float a[kGFloats];
    VDEV float get(int e, int k) const { return a[e * GK + k]; }
    VDEV void set(int e, int k, float v) { a[e * GK + k] = v; }
};
template <class ST>
struct GDistT {
    bool ok;
    int field, region, n;
    V3 p, m2;
    float isum;
    ST st;
    VDEV V3 raw(int k) const { return V3{st.get(0, k), st.get(1, k), st.get(2, k)}; }
    VDEV void set_raw(int k, V3 m) { st.set(0, k, m.x); st.set(1, k, m.y); st.set(2, k, m.z); }
    VDEV float a(int k) const { return st.get(3, k); }
    VDEV float kc(int k) const { return st.get(4, k); }
    VDEV float c1(int k) const { return st.get(5, k); }
    VDEV float c2(int k) const { return st.get(6, k); }
    VDEV float wn(int k) const { return st.get(7, k); }
    VDEV void set_lobe(int k, float a_, float kc_, float c1_, float c2_, float wo_) {
        st.set(3, k, a_); st.set(4, k, kc_); st.set(5, k, c1_); st.set(6, k, c2_); st.set(7, k, wo_);
    }
    VDEV void set_wn(int k, float v) { st.set(7, k, v); }
};
using GDist = GDistT<GStoreLds>;
using GDistReg = GDistT<GStoreReg>;
VDEV GStoreLds gstore_lds(float *lds, int stride) { return GStoreLds{lds, stride}; }

// the product of one re-aimed lobe with the vMF (m2, k2): density weight a, mass wo, concentration kc, and the two
// coefficients that stand for its mean direction (b = weight * vmf_norm(kr), nk2 = vmf_norm(k2))
VDEV void lobe_product(V3 raw, float kr, float b, V3 m2, float k2, float nk2, float *a, float *wo, float *kc, float *c1, float *c2) {
    const V3 s = raw * kr + m2 * k2;
    const float kp = __builtin_sqrtf(len2(s));
    const float kpinv = kp > 1e-6f ? 1.0f / kp : 0.f;  // the lobes cancel: mu . w := 0 under kc = 0.01, nearly uniform
    *kc = kappa_clamp(kp);
    const float E = fast_exp(-2 * *kc);
    *a = (b * nk2) * fast_exp((kp - kr) - k2);
    *wo = *a * ((kTwoPi * (1 - E)) / *kc);
    *c1 = kr * kpinv;
    *c2 = k2 * kpinv;
}

// fills `d` in place (d.st is set by the caller for the LDS home; the register home is never copied: a by-value copy of the
// array would pin it in scratch memory)
#ifndef VSPG_KD_LDS
#define VSPG_KD_LDS 256
#endif
constexpr int kKdLdsNodes = VSPG_KD_LDS;  // nodes per field staged in LDS by the workgroup kernel (8 B each)
template <class ST>
VDEV void gdist_init(GDistT<ST> &d, const DField *fields, int f, V3 p, bool have_product, V3 m2, float k2,
                     const VspgKdNode *kd_lds = nullptr) {
    d.ok = false;
    d.field = f;
    d.p = p;
    d.m2 = m2;
    d.isum = 1.f;
    d.n = 0;
    // f differs per lane when a wavefront holds volume and surface vertices: select between the two (scalar-loaded)
    // field records instead of indexing the array per lane
    DField F = fields[0];
    if (f) F = fields[1];
    if (kd_lds) {
        const int nl = F.n_nodes < kKdLdsNodes ? F.n_nodes : kKdLdsNodes;
        d.region = field_lookup(F, p, kd_lds + (f ? kKdLdsNodes : 0), nl);
    } else {
        d.region = field_lookup(F, p);
    }
    if (d.region < 0) return;
    const VspgFieldRegion &R = F.regions[d.region];
    const int n_lobes = ld1g(&R.n_lobes);
    if (n_lobes <= 0) return;
    d.ok = true;
    d.n = n_lobes < GK ? n_lobes : GK;
    const float *ax = region_aux(F, d.region);  // [k] b_k = weight_k * vmf_norm(kappa_k), [GK + k] kappa_k (clamped)
    const float nk2 = have_product ? vmf_norm(k2) : 0.f;
    const V3 pp = ld3g(R.pivot) - p;
    float sum = 0;
#pragma unroll
    for (int h = 0; h < GK; h += 4) {
        if (h < d.n) {
            const float4 w4 = ld4(R.weight + h), mx4 = ld4(R.mu[0] + h), my4 = ld4(R.mu[1] + h), mz4 = ld4(R.mu[2] + h),
                         d4 = ld4(R.distance + h), b4 = ld4(ax + h), k4 = ld4(ax + GK + h);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = h + j;
                if (k < d.n) {
                    const V3 raw = lobe_dir(pp, V3{c4(mx4, j), c4(my4, j), c4(mz4, j)}, c4(d4, j));
                    d.set_raw(k, raw);
                    float a = c4(b4, j), wo = c4(w4, j), kc = c4(k4, j), c1 = 1.f, c2 = 0.f;
                    if (have_product) lobe_product(raw, c4(k4, j), c4(b4, j), m2, k2, nk2, &a, &wo, &kc, &c1, &c2);
                    d.set_lobe(k, a, kc, c1, c2, wo);
                    sum += wo;
                }
            }
        }
    }
    if (sum > 0 && !isinf_(sum)) {
        d.isum = 1.0f / sum;
    } else {  // degenerate product: fall back to the incident-radiance mixture
        d.isum = 1.f;
#pragma unroll
        for (int k = 0; k < GK; ++k)
            if (k < d.n) d.set_lobe(k, ld1g(ax + k), ld1g(ax + GK + k), 1.f, 0.f, ld1g(R.weight + k));
    }
#pragma unroll
    for (int k = 0; k < GK; ++k)
        if (k < d.n) d.set_wn(k, d.wn(k) * d.isum);
}
// The product lobe of GuidedPhaseFunction::init as data (so that one gdist_init call serves both vertex kinds)
VDEV void volume_product_lobe(V3 dir, float g, bool *have_product, V3 *m2, float *k2) {
    float ag = __builtin_fabsf(g);
    if (ag < 1e-3f) {
        *have_product = false;
        *m2 = mk(0, 0, 1);
        *k2 = 0;
        return;
    }
    if (ag > 0.99f) ag = 0.99f;
    float kg = ag * (3 - ag * ag) / (1 - ag * ag);
    V3 axis = g > 0 ? dir : -dir;
    *have_product = true;
    *m2 = normalize(axis);
    *k2 = kg;
}
VDEV GDist gdist_init_surface(const DField *fields, V3 p, V3 n, float *lds, int stride) {
    GDist d;
    d.st = gstore_lds(lds, stride);
    gdist_init(d, fields, 0, p, true, n, kCosineLobeKappa);
    return d;
}
VDEV GDist gdist_init_volume(const DField *fields, V3 p, V3 dir, float g, float *lds, int stride) {
    GDist d;
    d.st = gstore_lds(lds, stride);
    bool hp;
    V3 m2;
    float k2;
    volume_product_lobe(dir, g, &hp, &m2, &k2);
    gdist_init(d, fields, 1, p, hp, m2, k2);
    return d;
}
template <class ST>
VDEV float gdist_pdf(const GDistT<ST> &d, V3 w) {
    const float x2 = dot(d.m2, w);
    float s = 0;
#pragma unroll
    for (int k = 0; k < GK; ++k)
        if (k < d.n) {
            const float dp = d.c1(k) * dot(d.raw(k), w) + d.c2(k) * x2;
            s += d.a(k) * fast_exp(d.kc(k) * (dp - 1));
        }
    return s * d.isum;
}
template <class ST>
VDEV float gdist_incoming_pdf(const DField *fields, const GDistT<ST> &d, V3 w) {
    DField F = fields[0];
    if (d.field) F = fields[1];
    const float *ax = region_aux(F, d.region);
    float s = 0;
#pragma unroll
    for (int h = 0; h < GK; h += 4) {
        if (h < d.n) {
            const float4 b4 = ld4(ax + h), k4 = ld4(ax + GK + h);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = h + j;
                if (k < d.n) s += c4(b4, j) * fast_exp(c4(k4, j) * (dot(d.raw(k), w) - 1));
            }
        }
    }
    return s;
}
// IncomingRadiancePDF of two directions in one pass over the region record (RIS: both candidates); per direction the
// same operations in the same order as gdist_incoming_pdf
template <class ST>
VDEV void gdist_incoming_pdf2(const DField *fields, const GDistT<ST> &d, bool want0, V3 w0, V3 w1, float *inc0, float *inc1) {
    DField F = fields[0];
    if (d.field) F = fields[1];
    const float *ax = region_aux(F, d.region);
    float s0 = 0, s1 = 0;
#pragma unroll
    for (int h = 0; h < GK; h += 4) {
        if (h < d.n) {
            const float4 b4 = ld4(ax + h), k4 = ld4(ax + GK + h);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = h + j;
                if (k < d.n) {
                    const V3 dir = d.raw(k);
                    if (want0) s0 += c4(b4, j) * fast_exp(c4(k4, j) * (dot(dir, w0) - 1));
                    s1 += c4(b4, j) * fast_exp(c4(k4, j) * (dot(dir, w1) - 1));
                }
            }
        }
    }
    *inc0 = s0;
    *inc1 = s1;
}
// VolumeScatterProbability(w) of the region the scratch was initialised for (field f, `region`, at its point)
template <class ST>
VDEV float gdist_vsp(const DField *fields, int f, int region, const GDistT<ST> &d, V3 w) {
    DField F = fields[0];
    if (f) F = fields[1];
    const VspgFieldRegion &R = F.regions[region];
    const float *ax = region_aux(F, region);
    const int n_lobes = ld1g(&R.n_lobes);
    int n = n_lobes < GK ? n_lobes : GK;
    float num = 0, den = 0;
#pragma unroll
    for (int h = 0; h < GK; h += 4) {
        if (h < n) {
            const float4 b4 = ld4(ax + h), k4 = ld4(ax + GK + h), v4 = ld4(R.vsp + h);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = h + j;
                if (k < n) {
                    float e = c4(b4, j) * fast_exp(c4(k4, j) * (dot(d.raw(k), w) - 1));
                    num += e * c4(v4, j);
                    den += e;
                }
            }
        }
    }
    if (!(den > 0)) return -1.f;
    return num / den;
}
// the direction SamplePDF draws once the lobe is chosen: acc = the masses before it, wk its own
VDEV V3 vmf_sample_dir(float u0, float u1, float acc, float wk, float kap, V3 mz) {
    float uw = wk > 0 ? (u0 - acc) / wk : 0.f;
    uw = uw < 0 ? 0 : (uw > kOneMinusEps ? kOneMinusEps : uw);
    float W = 1 + logf_(uw + (1 - uw) * fast_exp(-2 * kap)) / kap;
    W = clampf(W, -1, 1);
    float sinT = safe_sqrt(1 - W * W);
    float phi = kTwoPi * u1;
    Frame fr;
    fr.z = mz;
    coordinate_system(fr.z, &fr.x, &fr.y);
    return fr.from_local(V3{sinT * cosf_(phi), sinT * sinf_(phi), W});
}
template <class ST>
VDEV float gdist_sample(const GDistT<ST> &d, float u0, float u1, V3 *wi) {
    // the lobe: `for (k = 0; k < n - 1; ++k) { if (u0 < acc + wn_k) break; acc += wn_k; }` as a predicated scan (constant indices)
    bool scanning = true;
    int ks = 0;
    float acc = 0;
#pragma unroll
    for (int j = 0; j < GK - 1; ++j) {
        if (scanning && j < d.n - 1) {
            const float wj = d.wn(j);
            if (u0 < acc + wj) {
                scanning = false;
            } else {
                acc += wj;
                ks = j + 1;
            }
        }
    }
    float wk = 0, kap = 0, c1 = 0, c2 = 0;
    V3 raw = mk(0, 0, 1);
#pragma unroll
    for (int j = 0; j < GK; ++j)
        if (j == ks) {
            wk = d.wn(j);
            kap = d.kc(j);
            c1 = d.c1(j);
            c2 = d.c2(j);
            raw = d.raw(j);
        }
    // the product lobe's mean, (raw kappa + m2 k2) / kp; the bare lobe where there is no product or the two cancel
    const V3 mz = c1 == 0.f && c2 == 0.f ? raw : raw * c1 + d.m2 * c2;
    *wi = vmf_sample_dir(u0, u1, acc, wk, kap, mz);
    return gdist_pdf(d, *wi);
}

// GuidedBSDF / GuidedPhaseFunction state that outlives the vertex: the NEXT segment asks the vertex's distribution for
// VolumeScatterProbability(ray.d) (guiding.h:295-305, 564-574).  The new direction is known when the vertex code ends,
// so the value is evaluated there (no random numbers involved: same inputs, same result) and ONE float crosses to the
// next segment instead of {field, region, point} plus the lane's scratch; -1 = no estimate (scatter guiding off,
// region untrained), exactly what the deferred query would return.
struct GuideState {
    float vsp_next;
};

VSPG_NS_END  // namespace vspg
