// vspg_path.h -- device-side path logic: SampleLd, SampleDistance, Li, EvaluatePixelSample.
// (reference: src/pbrt/cpu/guidedvolpathvspgintegrator.cpp; citations per function)
#pragma once
#include <type_traits>

#include "vspg_device.h"
#include "vspg_guiding.h"
#include "vspg_lightsampler.h"
#include "vspg_train.h"

VSPG_NS_BEGIN

// ---------------------------------------------------------------------------------------
// DiffuseBxDF behind BSDF (src/pbrt/bxdfs.h:31-80, src/pbrt/bsdf.h:20-88)
// ---------------------------------------------------------------------------------------
struct Bsdf {
    Frame frame;  // Frame::FromXZ(Normalize(dpdus), ns) (vecmath.h:1862)
    Spec R;
    bool has_lobes;
};
template <bool GREY_KD = false>
VDEV Bsdf bsdf_make(const DQuad &q) {
    Bsdf b;
    b.frame.x = ld3(q.dpdu_n);
    b.frame.z = ld3(q.n);
    b.frame.y = cross(b.frame.z, b.frame.x);
    b.R = GREY_KD ? sp(q.Kd[0]) : lds(q.Kd);
    b.has_lobes = q.has_lobes != 0;
    return b;
}
VDEV Bsdf bsdf_make_tri(const DTri &T) {  // DiffuseMaterial on a triangle: BSDF(ns, dpdus) with ns = n, dpdus = dpdu (shapes.h:933-938)
    Bsdf b;
    b.frame.x = ld3(T.dpdu_n);
    b.frame.z = V3{T.nx, T.ny, T.nz};
    b.frame.y = cross(b.frame.z, b.frame.x);
    b.R = lds(T.Kd);
    b.has_lobes = T.Kd[0] != 0 || T.Kd[1] != 0 || T.Kd[2] != 0;
    return b;
}
VDEV Bsdf bsdf_make_sphere(const DSphere &sp_, const SphereSurf &ss) {  // BSDF(ns, dpdus) of the transformed interaction (bsdf.h:25-26)
    Bsdf b;
    b.frame.x = ss.dpdu_n;
    b.frame.z = ss.n;
    b.frame.y = cross(b.frame.z, b.frame.x);
    b.R = lds(sp_.Kd);
    b.has_lobes = sp_.has_lobes != 0;
    return b;
}
VDEV Spec bsdf_f(const Bsdf &b, V3 woR, V3 wiR) {
    V3 wi = b.frame.to_local(wiR), wo = b.frame.to_local(woR);
    if (wo.z == 0) return sp(0.f);
    if (!(wi.z * wo.z > 0)) return sp(0.f);  // SameHemisphere
    return b.R * kInvPi;
}
VDEV float bsdf_pdf(const Bsdf &b, V3 woR, V3 wiR) {
    V3 wo = b.frame.to_local(woR), wi = b.frame.to_local(wiR);
    if (wo.z == 0) return 0;
    if (!b.has_lobes) return 0;
    if (!(wi.z * wo.z > 0)) return 0;
    return __builtin_fabsf(wi.z) * kInvPi;  // CosineHemispherePDF(AbsCosTheta(wi))
}
VDEV bool bsdf_sample_f(const Bsdf &b, V3 woR, float u0, float u1, Spec *f, V3 *wiR, float *pdf) {
    V3 wo = b.frame.to_local(woR);
    if (wo.z == 0 || !b.has_lobes) return false;
    V3 wi = sample_cosine_hemisphere(u0, u1);
    if (wo.z < 0) wi.z *= -1;
    *pdf = __builtin_fabsf(wi.z) * kInvPi;
    *f = b.R * kInvPi;
    if (!nonzero(*f) || *pdf == 0 || wi.z == 0) return false;
    *wiR = b.frame.from_local(wi);
    return true;
}

// ---------------------------------------------------------------------------------------
// DiffuseAreaLight on a rectangle, uniform-area sampling
// (src/pbrt/lights.cpp:796-820, lights.h:492-511, src/pbrt/shapes.cpp:1155-1212, 1267-1283)
// ---------------------------------------------------------------------------------------
VDEV Spec light_L(const DQuad &q, V3 n, V3 w) {
    if (!q.two_sided && dot(n, w) < 0) return sp(0.f);
    return lds(q.Le);
}
struct LightLi {
    Spec L;
    V3 wi;
    float pdf;
    P3i pLight;
    V3 nLight;
};
VDEV bool light_sample_li(const DQuad &q, V3 ctxp, float u0, float u1, LightLi *ls) {
    V3 p00 = ld3(q.p00), p10 = ld3(q.p10), p01 = ld3(q.p01), p11 = ld3(q.p11);
    V3 pu0 = lerp(u1, p00, p01), pu1 = lerp(u1, p10, p11);
    V3 p = lerp(u0, pu0, pu1);
    V3 dpdu = pu1 - pu0;
    V3 dpdv = lerp(u0, p01, p11) - lerp(u0, p00, p10);
    if (len2(dpdu) == 0 || len2(dpdv) == 0) return false;
    V3 c = cross(dpdu, dpdv);
    V3 n = normalize(c);
    if (dot(n, ld3(q.n)) < 0) n = -n;  // reverseOrientation
    float pdf = wrcp(len(c));
    P3i pint = p3i_from_err(p, ld3(q.perr));
    p = pint.mid();
    V3 wi = p - ctxp;
    if (len2(wi) == 0) return false;
    wi = normalize(wi);
    V3 d = ctxp - p;
    pdf = wdiv(pdf, wdiv(absdot(n, -wi), len2(d)));
    if (isinf_(pdf)) return false;
    if (pdf == 0) return false;
    Spec Le = light_L(q, n, -wi);
    if (!nonzero(Le)) return false;
    ls->L = Le;
    ls->wi = wi;
    ls->pdf = pdf;
    ls->pLight = pint;
    ls->nLight = n;
    return true;
}
// light.SampleLi(ctx, uLight, lambda, allowIncompletePDF = true) for light `lightIndex` of the scene's light list
template <bool FULL = true>
VDEV bool sample_light(const DScene &S, int lightIndex, V3 ctxp, float u0, float u1, LightLi *ls, bool *delta_light) {
    *delta_light = false;
    if (!FULL || lightIndex < S.n_lights) return light_sample_li(light_quad_at(lightIndex), ctxp, u0, u1, ls);
    const int k = lightIndex - S.n_lights;
    // UniformInfiniteLight::SampleLi returns {} for the incomplete PDF (lights.cpp:1019-1023): the sky is reached by escaping rays only
    if (S.inf_type[k] != VSPG_LIGHT_DISTANT) return false;
    // DistantLight::SampleLi (lights.h:320-327)
    ls->L = lds(S.inf_L[k]);
    if (!nonzero(ls->L)) return false;
    ls->wi = ld3(S.inf_w[k]);
    ls->pdf = 1;
    ls->pLight = p3i_exact(ctxp + ls->wi * (2 * S.scene_radius));
    ls->nLight = mk(0, 0, 0);
    *delta_light = true;
    return true;
}
struct LsCtx {  // LightSampleContext
    P3i pi;
    V3 n;
};
// The previous vertex's LightSampleContext, carried across a path segment in compressed form: a
// surface vertex is {re-projected hit point, rectangle} and LightSampleContext(isect) =
// {Point3fi(p, pError), n} is a pure function of the two; a medium vertex (quad < 0) is the exact
// point with n = 0.  Expanded only where it is read (emitter hit by a non-specular path).
struct PrevCtx {
    V3 p;      // as Isect::p: point on a rectangle / barycentrics of a triangle hit / object-space point on a sphere / the medium vertex
    int quad;
    template <bool FULL = true>
    VDEV LsCtx expand(const DScene &S) const {
        LsCtx c;
        if (FULL && is_sphere(quad)) {  // p = the object-space hit point (kSpherePrim)
            const SphereSurf ss = sphere_interaction<false>(S.spheres[sphere_of(quad)], p);
            c.pi = ss.pi;
            c.n = ss.n;
        } else if (quad >= 0) {
            const DQuad &q = quad_at(quad);
            c.pi = p3i_from_err(p, ld3(q.perr));
            c.n = ld3(q.n);
        } else if (FULL && is_tri(quad)) {
            const DTri &T = S.tris[tri_of(quad)];
            c.pi = tri_pi(T, p);
            c.n = V3{T.nx, T.ny, T.nz};
        } else {
            c.pi = p3i_exact(p);
            c.n = mk(0, 0, 0);
        }
        return c;
    }
};
VDEV float light_pdf_li(const DQuad &q, const LsCtx &ctx, V3 wi) {  // shapes.cpp:1329-1352, area branch
    V3 o = offset_ray_origin(ctx.pi, ctx.n, wi);
    float t;
    V3 p;
    if (!quad_intersect(q, o, wi, kInf, &t, &p)) return 0;
    p = p3i_from_err(p, ld3(q.perr)).mid();
    V3 d = ctx.pi.mid() - p;
    float pdf = wrcp(q.area) * wdiv(len2(d), absdot(ld3(q.n), -wi));
    return isinf_(pdf) ? 0 : pdf;
}

// ---------------------------------------------------------------------------------------
// per-lane counters (reduced per workgroup at kernel end)
// ---------------------------------------------------------------------------------------
// Two interchangeable counter sinks (the path functions take either, `class PC`):
//   PathCounters  one set per lane in registers (per-lane kernels; k_trace_paths reports segments per path);
//   WaveCounters  one set per workgroup in LDS: the active lanes of a wavefront are counted with a
//                 ballot and one lane adds the total -- no per-lane registers, ~10 instructions per count.
enum { CNT_PATHS = 0, CNT_SEGMENTS, CNT_VOLUME_SCATTERS, CNT_SURFACE_HITS, CNT_DENSITY_QUERIES, CNT_SHADOW_RAYS, CNT_SHADOW_QUERIES, CNT_COUNT };
// Either sink also carries the a18 recorder (`rec`): NullRecorder everywhere except the training
// instantiations of the per-lane kernel (vspg_train.h).
template <class REC>
struct PathCountersT {
    uint32_t segments, volume_scatters, surface_hits, density_queries, shadow_rays, shadow_queries;
    REC rec;
    VDEV void segment() { segments++; }
    VDEV void volume_scatter() { volume_scatters++; }
    VDEV void surface_hit() { surface_hits++; }
    VDEV void density_query() { density_queries++; }
    VDEV void shadow_ray() { shadow_rays++; }
    VDEV void shadow_query() { shadow_queries++; }
};
using PathCounters = PathCountersT<NullRecorder>;
template <class REC>
struct WaveCountersT {
    unsigned int *c;  // LDS, CNT_COUNT entries
    REC rec;
    VDEV void add(int k) const {
        const unsigned long long m = __ballot(1);
        if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(c + k, (unsigned int)__popcll(m));
    }
    VDEV void path() const { add(CNT_PATHS); }
    VDEV void segment() const { add(CNT_SEGMENTS); }
    VDEV void volume_scatter() const { add(CNT_VOLUME_SCATTERS); }
    VDEV void surface_hit() const { add(CNT_SURFACE_HITS); }
    VDEV void density_query() const { add(CNT_DENSITY_QUERIES); }
    VDEV void shadow_ray() const { add(CNT_SHADOW_RAYS); }
    VDEV void shadow_query() const { add(CNT_SHADOW_QUERIES); }
};
using WaveCounters = WaveCountersT<NullRecorder>;

struct IsgSample {
    bool valid, surface_event;
    float vsp_used;
};

// GuidedRussianRoulette(throughput, adjointEstimate, pixelContributionEstimate, minSurvival) is OpenPGL code (absent):
// own definition (oracle/vspg_oracle.c:guided_russian_roulette), adjoint-driven in spirit -- the largest per-channel ratio of
// throughput x adjoint to the pixel's contribution estimate, clamped to [minSurvival, 1]; no estimate -> 1.  Unpinned.
// (adjointEstimate is 1 in the reference: OPENPGL_RADIANCE_CACHES is not defined; the estimate here is one value for
// the three channels.)
VDEV float guided_russian_roulette(Spec throughput, float reference, float minSurvival) {
    if (!(reference > 0)) return 1.f;
    float s = 0.f;
    float q = throughput.r * 1.f / reference;
    s = q > s ? q : s;
    q = throughput.g * 1.f / reference;
    s = q > s ? q : s;
    q = throughput.b * 1.f / reference;
    s = q > s ? q : s;
    return fmin_(1.f, fmax_(minSurvival, s));
}
// StandardThroughputBasedRussianRoulette is OpenPGL code (absent from the reference tree):
// pbrt's own rule, survival = clamp(maxComponent, 0, 1).  Unpinned, see DESIGN.md.
VDEV float standard_throughput_rr(Spec w) { return fmin_(1.f, fmax_(0.f, maxc(w))); }

// ---------------------------------------------------------------------------------------
// a15: SampleLd (guidedvolpathvspgintegrator.cpp:1136-1252); guiding distributions inactive
// ---------------------------------------------------------------------------------------
struct Intr {
    bool is_surface;
    P3i pi;
    V3 n;
    V3 wo;
    float g;
    // medium boundaries (read only where has_bnd(S)): Interaction::medium -- the medium of the ray that reached a
    // non-transition surface, of the medium interaction itself -- and the surface's SURF_* flags (GetMedium, interaction.h:117-121)
    bool medium;
    int sflags;
};
// v * (T_maj / T_maj[ch]) (:1083-1084, :1236-1238, :723-724).  In a grey medium every channel of T_maj IS T_maj[ch] (built from
// one value, HomogeneousMediumT / GridMediumT): the quotient is x / x -- exactly 1 for every finite non-zero x -- and v * 1 is v.
// The division only runs where that is not so (a FastExp underflow to 0: 0 / 0, as in the reference).
template <int GREY>
VDEV Spec mul_tmaj_ratio(Spec v, Spec T_maj, int ch) {
    const float tm = ch_of(T_maj, ch);
    if (GREY >= 1 && tm > 0.f && tm < kInf) return v;
    return v * wdiv(T_maj, tm);
}
// sample_Ld from the shadow ray on (:1190-1251): lightRay = intr.SpawnRayTo(ls->pLight), the ratio-tracked transmittance,
// the estimate.  (Its own function since round 3: the workgroup kernel's guided vertex runs the two halves of sample_Ld at
// different times, vspg_guided_wg.h.)
template <class Medium, class PC>
VDEV Spec sample_Ld_shadow(const DScene &S, const Medium &medium, const Intr &intr, int ch, const LightLi &ls, Spec f_hat, float p_l,
                           float scatterPDF, Spec r_p, PC &pc, bool delta_light = false) {
    constexpr bool kFull = !Medium::kSimpleScene;
    // lightRay = intr.SpawnRayTo(ls->pLight) (interaction.h:111-115, ray.h:103-108)
    V3 pf = offset_ray_origin(intr.pi, intr.n, ls.pLight.mid() - intr.pi.mid());
    V3 pt = offset_ray_origin(ls.pLight, ls.nLight, pf - ls.pLight.mid());
    V3 lo = pf, ld = pt - pf;
    Spec T_ray = sp(1.f), r_l = sp(1.f), r_u = sp(1.f);
    Rng rng;
    rng.set_sequence(hash_v3(lo), hash_v3(ld));  // :1193
    pc.shadow_ray();
    // ratio tracking (:1207-1232)
    auto ratio_cb = [&](V3, const MediumProps &mp, Spec sigma_maj, Spec T_maj, bool) {
        if constexpr (!Medium::kSingleSegment) pc.shadow_query();  // heterogeneous media: a density fetch, as in the distance walk
        float pdf = ch_of(T_maj, ch) * ch_of(sigma_maj, ch);
        if constexpr (Medium::kNullZero) {
            // sigma_n == 0: T_ray *= T_maj * 0 / pdf is an exact 0 for a positive finite pdf (T_maj is
            // finite), whatever the Russian roulette below draws; the callback's `!T_ray -> return false`
            // follows.  (pdf == 0 -- a FastExp underflow over ~175 mean free paths -- keeps the generic code.)
            if (pdf > 0 && !isinf_(pdf)) {
                T_ray = sp(0.f);
                return false;
            }
        }
        Spec sigma_n = medium.sigma_n(mp, sigma_maj);
        T_ray = T_ray * wdiv(T_maj * sigma_n, pdf);
        r_l = r_l * wdiv(T_maj * sigma_maj, pdf);
        r_u = r_u * wdiv(T_maj * sigma_n, pdf);
        Spec Tr = wdiv(T_ray, avg(r_l + r_u));
        if (maxc(Tr) < 0.05f) {
            float q = 0.75f;
            if (rng.uniform() < q)
                T_ray = sp(0.f);
            else
                T_ray = wdiv(T_ray, 1 - q);
        }
        if (!nonzero(T_ray)) return false;
        return true;
    };
    if (kFull && has_bnd<Medium::kBnd>(S)) {
        // The loop of :1195-1244 over the segments of the light ray: Intersect(lightRay, 1 - ShadowEpsilon) finds the CLOSEST hit;
        // one with a material blocks, an interface ends the segment and the ray goes on from it, in the medium its far side
        // holds (si->intr.SpawnRayTo(ls->pLight), :1243).
        bool lmed = S.medium_type != VSPG_MEDIUM_NONE && get_medium(intr.sflags, intr.n, ld, intr.medium);  // r.medium = GetMedium(r.d)
        while (!(ld.x == 0 && ld.y == 0 && ld.z == 0)) {
            const Isect si = scene_intersect<true>(S, lo, ld, 1 - kShadowEps);
            int sfl = 0;
            if (si.hit) {
                sfl = surf_flags(S, si.quad);
                if (!(sfl & SURF_INTERFACE)) return sp(0.f);
            }
            if (lmed) {
                VSPG_PROF(PS_NEE_TR);
                const float tMax = si.hit ? si.t : 1 - kShadowEps;
                const float us = rng.uniform();
                const Spec T_maj = sample_T_maj(medium, lo, ld, tMax, us, rng, ch, ratio_cb);
                T_ray = mul_tmaj_ratio<Medium::kGrey>(T_ray, T_maj, ch);
                r_l = mul_tmaj_ratio<Medium::kGrey>(r_l, T_maj, ch);
                r_u = mul_tmaj_ratio<Medium::kGrey>(r_u, T_maj, ch);
            }
            if (!nonzero(T_ray)) return sp(0.f);
            if (!si.hit) break;
            const P3i spi = surf_pi(S, si.quad, si.p);
            const V3 nf = offset_ray_origin(spi, si.n, ls.pLight.mid() - spi.mid());
            const V3 nt = offset_ray_origin(ls.pLight, ls.nLight, nf - ls.pLight.mid());
            lo = nf;
            ld = nt - nf;
            lmed = S.medium_type != VSPG_MEDIUM_NONE && get_medium(sfl, si.n, ld, lmed);  // (no medium: IFACE bits on a surface name nothing)
        }
    } else if (!(ld.x == 0 && ld.y == 0 && ld.z == 0)) {
        // every surface here carries a material: any hit is an opaque blocker (:1197-1200)
#ifndef VSPG_EXP_NOANYHIT
        if (scene_intersect_any<kFull>(S, lo, ld, 1 - kShadowEps)) return sp(0.f);
#endif
        if (S.medium_type != VSPG_MEDIUM_NONE) {
            VSPG_PROF(PS_NEE_TR);
            float tMax = 1 - kShadowEps;
            float us = rng.uniform();
            Spec T_maj = sample_T_maj(medium, lo, ld, tMax, us, rng, ch, ratio_cb);
            T_ray = mul_tmaj_ratio<Medium::kGrey>(T_ray, T_maj, ch);
            r_l = mul_tmaj_ratio<Medium::kGrey>(r_l, T_maj, ch);
            r_u = mul_tmaj_ratio<Medium::kGrey>(r_u, T_maj, ch);
        }
        if (!nonzero(T_ray)) return sp(0.f);
    }
    r_l = r_l * (r_p * p_l);
    r_u = r_u * (r_p * scatterPDF);
    if (kFull && delta_light) return wdiv(f_hat * T_ray * ls.L, avg(r_l));  // IsDeltaLight (:1248-1249)
    return wdiv(f_hat * T_ray * ls.L, avg(r_l + r_u));
}

template <class Medium, class PC, class GD = GDist>
VDEV Spec sample_Ld(const DScene &S, const Medium &medium, const Intr &intr, const Bsdf *bsdf, int ch,
                    Sampler &sampler, Spec r_p, PC &pc, const GD *gd = nullptr, bool use_gd = false) {
    // (use_gd separate from the pointer: a `cond ? &gd : nullptr` argument keeps a register-resident distribution in scratch)
    V3 ctxp = intr.pi.mid();
    if (intr.is_surface && bsdf->has_lobes) ctxp = offset_ray_origin(intr.pi, intr.n, intr.wo);  // :1147-1149
    float u = sampler.get1d();
    // lightSampler.Sample(ctx, u) (:1157) over the emissive rectangles followed by the infinite lights: UniformLightSampler
    // (lightsamplers.h:33-38) inline; the power / BVH samplers of multi-light scenes (vspg_lightsampler.h) in the full-scene kernels
    constexpr bool kFull = !Medium::kSimpleScene;
    const int n_all = kFull ? S.n_lights + S.n_inf : S.n_lights;
    bool have_light = n_all > 0;
    int lightIndex = 0;
    float lightPmf = 0;
    if (kFull && S.lsamp.mode != VSPG_LIGHTSAMPLER_UNIFORM) {
        have_light = light_sampler_sample(S, ctxp, intr.is_surface ? intr.n : mk(0, 0, 0), u, &lightIndex, &lightPmf);
    } else if (have_light) {
        int li = (int)(u * (float)n_all);
        lightIndex = li < n_all - 1 ? li : n_all - 1;
        lightPmf = wrcp((float)n_all);
    }
    float ul0 = sampler.get1d(), ul1 = sampler.get1d();
    if (!have_light) return sp(0.f);
    LightLi ls;
    bool delta_light = false;
    if (!sample_light<kFull>(S, lightIndex, ctxp, ul0, ul1, &ls, &delta_light)) return sp(0.f);
    float p_l = lightPmf * ls.pdf;

    float scatterPDF;
    Spec f_hat;
    V3 wo = intr.wo, wi = ls.wi;
    // GuidedBSDF::PDF / GuidedPhaseFunction::PDF (guiding.h:271-289, 542-558); gd != nullptr iff useGuiding
    if (intr.is_surface) {
        f_hat = bsdf_f(*bsdf, wo, wi) * absdot(wi, intr.n);
        float bsdfPDF = bsdf_pdf(*bsdf, wo, wi);
        if (use_gd) bsdfPDF = ((1.0f - kGuidingProbability) * bsdfPDF) + (kGuidingProbability * gdist_pdf(*gd, wi));
        scatterPDF = 1.0f * bsdfPDF;
    } else {
        float p = henyey_greenstein(dot(wo, wi), intr.g);
        f_hat = sp(p);
        float phasePDF = p;
        if (use_gd) phasePDF = ((1.0f - kGuidingProbability) * phasePDF) + (kGuidingProbability * gdist_pdf(*gd, wi));
        scatterPDF = 1.0f * phasePDF;
    }
    if (!nonzero(f_hat)) return sp(0.f);

    return sample_Ld_shadow<Medium>(S, medium, intr, ch, ls, f_hat, p_l, scatterPDF, r_p, pc, delta_light);
}

// ---------------------------------------------------------------------------------------
// path state carried across the Li loop (:294-307)
// ---------------------------------------------------------------------------------------
struct PathState {
    V3 ro, rd;
    Spec L, beta, r_u, r_l;
    PrevCtx prevCtx;
    int depth;
    bool specularBounce, anyNonSpecularBounces, lastVertexVolume;
    float rr_correction, etaScale;
    float vsp0;  // primary-ray VSP of this pixel, loaded when the path starts (hides the HBM latency)
    float pce;   // guided builds, rrguiding: the pixel's contribution estimate (0 = none); guideRR = rrguiding && buffer ready
    bool guideRR;
    float lu;        // the wavelength sample (its lambdas: blackbody emission of a temperature grid only; dead code elsewhere)
    bool in_medium;  // ray.medium != nullptr.  Constant without medium boundaries (has_bnd(S) == 0); simple-scene kernels never read it
    GuideState gs;  // guided builds only: the previous vertex's distribution for the secondary-ray VSP
};

// VSP fetch (:654-671, :1098-1134).  Secondary-ray VSP comes from the guiding cache, which is
// untrained in the configurations this build accepts -> VolumeScatterProbability() == -1
// (guiding.h:295-298, 564-567).
template <bool GUIDED>
VDEV float fetch_vsp(const DScene &S, const PathState &st, bool *guide) {
    float vsp = -1.f;
    *guide = false;
    if (st.depth == 0) {
        if (S.prm.vspguiding && S.prm.vspprimaryguiding) {
            vsp = st.vsp0;  // imageSpaceGuidingBuffer estimate, or 0.5 before the first update (:1101-1105)
            *guide = !(isnan_(vsp) || vsp < 0.f || vsp > 1.f);
        }
    } else if constexpr (GUIDED) {
        if (S.prm.vspguiding && S.prm.vspsecondaryguiding) {
            // g{phase,bsdf}.VolumeScatterProbability(ray.d) of the PREVIOUS vertex (:661-668)
            vsp = st.gs.vsp_next;  // evaluated by the previous vertex (li_vertex_guided)
            *guide = !(isnan_(vsp) || vsp < 0.f || vsp > 1.f);
        }
    }
    if (*guide) vsp = fmax_(fmin_(vsp, 0.999f), 0.001f);
    return vsp;
}

// ---------------------------------------------------------------------------------------
// a10-a13: SampleDistance (:637-1096), restructured around an EVENT: the traversal callbacks only
// do the cheap bookkeeping the reference does at a collision; when a real scattering event is
// chosen the callback records where (p, phase g) and stops the traversal, and the volume-scatter
// tail (:804-875 == :988-1058: cache init, NEE, RR, phase sampling) runs ONCE afterwards in
// li_segment, merged with the surface-vertex code.  The reference runs the tail inside the
// callback and then returns false, so nothing else happens in between: same operations, same
// random-number order, one copy of the expensive code instead of one per callback site.
// ---------------------------------------------------------------------------------------
enum { EV_PASS = 0, EV_SCATTER = 1, EV_TERMINATE = 2 };
struct DistEvent {
    int kind;
    V3 p;     // scattering position
    float g;  // HG asymmetry of the phase function at p
};

// SEG tells the compiler which segment of a path the caller is at, when the caller knows (the
// workgroup-level kernel runs primary and later segments in different phases):
//   SEG_PRIMARY    depth == 0: the camera segment, the only one whose VSP comes from the image-space buffer;
//   SEG_SECONDARY  depth  > 0: without the guiding cache no VSP exists there, SampleT_maj_OpticalDepthSpace
//                  returns SampleT_maj(...) at its first line (media_sampleTMaj.h:279) -- the call is made
//                  directly and the optical-depth-space code (the register-hungriest part of the path
//                  loop) is not instantiated in that phase;
//   SEG_ANY        decided at run time (per-lane kernels).
enum { SEG_ANY = 0, SEG_PRIMARY = 1, SEG_SECONDARY = 2 };
template <class Medium, bool GUIDED, int SEG = SEG_ANY, class PC>
VDEV DistEvent sample_distance(const DScene &S, const Medium &medium, const float *vsp_buf, int vsp_ready, int px, int py,
                               PathState &st, float tMax, int ch, Sampler &sampler, Rng &rng, IsgSample &isg,
                               PC &pc, Spec &tw /* transmittanceWeight (:317), training builds only */) {
    DistEvent ev;
    ev.kind = EV_PASS;
    ev.p = mk(0, 0, 0);
    ev.g = 0;
#ifdef VSPG_EXP_NOODS  // timing experiment only (primary rays lose their VSP guiding)
    constexpr bool kPlainOnly = !GUIDED;
#else
    constexpr bool kRec = std::remove_reference<decltype(pc.rec)>::type::kActive;
    constexpr bool kPlainOnly = SEG == SEG_SECONDARY && !GUIDED;
#endif
    if constexpr (SEG == SEG_PRIMARY) __builtin_assume(st.depth == 0);
    if constexpr (SEG == SEG_SECONDARY) __builtin_assume(st.depth > 0);
    bool guide = false;
    float vsp = -1.f;
    if constexpr (!kPlainOnly) vsp = fetch_vsp<GUIDED>(S, st, &guide);
    if (st.depth == 0) isg.vsp_used = guide ? vsp : -1.f;

    bool use_resampling = S.prm.vspsamplingmethod == VSPG_VSP_RESAMPLING && !medium.is_homogeneous();
    if (use_resampling) {
        float weightSum = 0;
        Spec trRatioEst = sp(1.f), beta_rs = sp(1.f), r_u_rs = sp(1.f);
        // CandidateData (integrators.h:526-543)
        V3 sel_p = mk(0, 0, 0);
        float sel_g = 0;
        float sel_wi = 0, sel_sTTr = 0;
        Spec sel_num = sp(0), sel_den = sp(0);
        float vrc = 0, majorantScale = 1;
        float u = sampler.get1d();
        Spec T_maj = sample_T_maj_resampling(
            medium, st.ro, st.rd, tMax, u, rng, ch, guide, vsp, &vrc, &majorantScale,
            [&](V3 p, const MediumProps &mp, Spec sigma_maj, Spec T_maj, bool) {
                pc.density_query();
                Spec sigma_t = mp.sigma_t;
                Spec sigma_n = clamp_zero(sigma_maj - sigma_t);
                float wi = ch_of(wdiv(sigma_t, sigma_maj) * trRatioEst, ch);
                if (wi > 0) {
                    weightSum += wi;
                    if (sampler.get1d() < wdiv(wi, weightSum)) {
                        float pdf = ch_of(T_maj, ch) * ch_of(sigma_t, ch);
                        sel_num = wdiv(beta_rs * T_maj * mp.sigma_s, pdf);
                        sel_den = wdiv(r_u_rs * T_maj * sigma_t, pdf);
                        sel_p = p;
                        sel_g = mp.g;
                        sel_wi = wi;
                        sel_sTTr = wi;
                    }
                }
                float pdf = ch_of(T_maj, ch) * ch_of(sigma_n, ch);
                beta_rs = beta_rs * wdiv(T_maj * sigma_n, pdf);
                r_u_rs = r_u_rs * wdiv(T_maj * sigma_n, pdf);
                trRatioEst = trRatioEst * wdiv(sigma_n, sigma_maj);
                return true;
            });
        beta_rs = mul_tmaj_ratio<Medium::kGrey>(beta_rs, T_maj, ch);
        r_u_rs = mul_tmaj_ratio<Medium::kGrey>(r_u_rs, T_maj, ch);
        if (st.depth == 0 && S.tr_calc && !(vsp_ready & VSP_NO_FEED)) {  // trBuffer->AddSample (:727-728, trbuffer.h:40-45)
            // one lane owns the pixel for the whole launch and runs its samples in order (k_render_wave)
            const size_t pix = (size_t)py * S.xres + px;
            const int n = S.tr_spp[pix] + 1;
            S.tr_spp[pix] = n;
            const float alpha = wrcp((float)n);
            float *t = S.tr_rgb + pix * 3;
            t[0] = (1.f - alpha) * t[0] + alpha * trRatioEst.r;
            t[1] = (1.f - alpha) * t[1] + alpha * trRatioEst.g;
            t[2] = (1.f - alpha) * t[2] + alpha * trRatioEst.b;
        }
        float trScalar = ch_of(trRatioEst, ch);
        float surf_wi = trScalar;
        if (guide && trScalar < 1 && trScalar > 0 && weightSum > 0) {
            float volRatio = vrc * S.prm.vspmisratio + (1 - trScalar) * (1 - S.prm.vspmisratio);
            float surfRatio = 1 - volRatio;
            surf_wi = wdiv(surfRatio, volRatio) * weightSum;
        }
        weightSum += surf_wi;
        bool selectSurface = false;
        if (weightSum == 0) return ev;
        if (sampler.get1d() < wdiv(surf_wi, weightSum)) {
            sel_wi = surf_wi;
            sel_sTTr = trScalar;
            sel_num = beta_rs;
            sel_den = r_u_rs;
            selectSurface = true;
        }
        float factor = wdiv(weightSum * sel_sTTr, sel_wi);
        if (!selectSurface) {
            if (st.depth == 0) {
                isg.valid = true;
                isg.surface_event = false;
            }
            if (st.depth++ >= S.prm.maxdepth) {
                ev.kind = EV_TERMINATE;
                return ev;
            }
            pc.volume_scatter();
        }
        st.beta = st.beta * (sel_num * factor);
        st.r_u = st.r_u * sel_den;
        if (has_nan(st.beta) || has_nan(st.r_u) || has_inf(st.beta) || has_inf(st.r_u)) {
            ev.kind = EV_TERMINATE;
            return ev;
        }
        if (!selectSurface) {
            if constexpr (kRec) {  // :798-802
                tw = tw * ((sel_num * factor) / sel_den);
                pc.rec.add_transmittance_weight(tw);
                pc.rec.new_segment(sel_p, true);
                tw = sp(1.f);
            }
            ev.kind = EV_SCATTER;
            ev.p = sel_p;
            ev.g = sel_g;
        }
        return ev;
    }

    // the delta-tracking routine (:878-1094)
#ifdef VSPG_PROFILE
    ProfScope prof_dist(guide ? PS_DIST_GUIDED : PS_DIST_PLAIN);
#endif
    Spec r_u_factor = sp(1.f);  // beta_factor is never written by the reference (always 1)
    float u = sampler.get1d();
    auto on_collision = [&](V3 p, const MediumProps &mp, Spec sigma_maj, Spec T_maj, bool activateNDS) {
            pc.density_query();
            if (!nonzero(st.beta)) {
                ev.kind = EV_TERMINATE;
                return false;
            }
            Spec mpLe = mp.Le;  // MediumProperties::Le as SamplePoint(p, lambda) returns it (see MediumProps)
            if constexpr (Medium::kEmit)
                if (mp.bb_temp > 0.f) mpLe = blackbody_sample(mp.bb_temp, st.lu) * mp.bb_scale;
            if (st.depth < S.prm.maxdepth && nonzero(mpLe)) {  // :895-906
                float pdf = ch_of(sigma_maj, ch) * ch_of(T_maj, ch);
                Spec betap = wdiv(st.beta * T_maj, pdf);
                Spec r_e = wdiv(st.r_u * sigma_maj * T_maj, pdf);
                if (nonzero(r_e)) st.L = st.L + wdiv(betap * mp.sigma_a * mpLe, avg(r_e));
            }
            Spec sigma_t = mp.sigma_t;
            float pScatter = wdiv(ch_of(sigma_t, ch), ch_of(sigma_maj, ch));
            // NDS+ (:929-938): bias the real/null-collision probability of the PRIMARY ray by the cached transmittance.
            // With a homogeneous medium pScatter == 1 and every NDS+ factor is exactly 1: compiled out.
            bool NDS_plus = false;
            if constexpr (!Medium::kAlwaysRealCollision && SEG != SEG_SECONDARY) {
                if (st.depth == 0 && S.prm.vspsamplingmethod == VSPG_VSP_NDS && S.prm.collisionProbabilityBias && S.tr_load && activateNDS) {
                    NDS_plus = true;
                    const float trEstCache = S.tr_rgb[((size_t)py * S.xres + px) * 3 + ch];
                    const float gamma = 1 + trEstCache;
                    pScatter = vspg_libm::powf_host_exact(pScatter, 1 / gamma);
                }
            }
            float pNull = fmax_(0.f, 1 - pScatter);
            float um = rng.uniform();
            // Medium::kAlwaysRealCollision: the draw cannot change the outcome (see HomogeneousMedium), and
            // with it the walk's private RNG -- seeded by two 64-bit hashes per segment -- is dead code
            const int mode = Medium::kAlwaysRealCollision ? 0 : sample_discrete2(pScatter, pNull, um);
            if (mode == 0) {
                if (st.depth == 0) {
                    isg.valid = true;
                    isg.surface_event = false;
                }
                if (st.depth++ >= S.prm.maxdepth) {
                    ev.kind = EV_TERMINATE;
                    return false;
                }
                pc.volume_scatter();
                float pdf = ch_of(T_maj, ch) * ch_of(sigma_t, ch);
                st.beta = st.beta * wdiv(T_maj * mp.sigma_s, pdf);
                // grey medium: T_maj * sigma_t is, channel by channel, the very product pdf is -- x / x == 1 (see mul_tmaj_ratio)
                if (!(Medium::kGrey >= 1 && pdf > 0.f && pdf < kInf)) st.r_u = st.r_u * wdiv(T_maj * sigma_t, pdf);
                if (NDS_plus) st.r_u = st.r_u * (sigma_maj * pScatter / sigma_t);  // :975-976
                st.r_u = st.r_u * r_u_factor;
                if constexpr (kRec) {  // :978-986 (beta_factor is 1)
                    tw = tw * ((T_maj * mp.sigma_s) / pdf);
                    tw = tw * (sp(1.f) / ch_of(r_u_factor, ch));
                    pc.rec.add_transmittance_weight(tw);
                    pc.rec.new_segment(p, true);
                    tw = sp(1.f);
                }
                ev.kind = EV_SCATTER;
                ev.p = p;
                ev.g = mp.g;
                return false;
            } else {
                Spec sigma_n = medium.sigma_n(mp, sigma_maj);
                float pdf = ch_of(T_maj, ch) * ch_of(sigma_n, ch);
                st.beta = st.beta * wdiv(T_maj * sigma_n, pdf);
                if constexpr (kRec) tw = tw * (T_maj * sigma_n / pdf);  // :1067
                if (pdf == 0) {
                    st.beta = sp(0.f);
                    if constexpr (kRec) tw = sp(0.f);
                }
                st.r_u = st.r_u * wdiv(T_maj * sigma_n, pdf);
                if (NDS_plus) st.r_u = st.r_u * (sigma_maj * (1 - pScatter) / sigma_n);  // :1072-1073
                st.r_l = st.r_l * wdiv(T_maj * sigma_maj, pdf);
                return nonzero(st.beta) && nonzero(st.r_u);
            }
        };
    Spec T_maj;
    if constexpr (kPlainOnly)
        T_maj = sample_T_maj(medium, st.ro, st.rd, tMax, u, rng, ch, on_collision);
    else
        T_maj = sample_T_maj_ods(medium, st.ro, st.rd, tMax, u, rng, ch, guide, vsp, S.prm.vspmisratio,
                                 S.prm.vspsamplingmethod == VSPG_VSP_NDS, &r_u_factor, on_collision);
    // :1080-1091 -- after a scatter event the reference never multiplies T_maj in (scattered or
    // terminated is set by the tail, or beta / r_u is zero)
    bool multiply_T_maj = ev.kind == EV_PASS && nonzero(st.beta) && nonzero(st.r_u);
    if (multiply_T_maj) {
        st.beta = mul_tmaj_ratio<Medium::kGrey>(st.beta, T_maj, ch);
        st.r_u = mul_tmaj_ratio<Medium::kGrey>(st.r_u, T_maj, ch);
        st.r_l = mul_tmaj_ratio<Medium::kGrey>(st.r_l, T_maj, ch);
        st.r_u = st.r_u * r_u_factor;
        st.r_l = st.r_l * r_u_factor;
        if constexpr (kRec) {  // :1085, :1090
            tw = mul_tmaj_ratio<Medium::kGrey>(tw, T_maj, ch);
            tw = tw * (sp(1.f) / ch_of(r_u_factor, ch));
        }
    }
    return ev;
}

// ---------------------------------------------------------------------------------------
// one iteration of the Li path loop (:309-609): returns false when the path ends.
// Surface vertices (:376-608) and volume vertices (the scatter tail) share the NEE code.
// ---------------------------------------------------------------------------------------
// The path-loop iteration is split at the vertex: li_segment_a runs up to the point where the path
// either ended or stands at a vertex (volume: scatter position + phase g; surface: rectangle + hit
// point), li_segment_b processes that vertex (NEE, RR, new direction).  The workgroup-level wavefront
// kernel runs the two halves in separate phases with the path state parked in LDS in between; the
// per-lane kernels simply call one after the other.  Same operations, same order per path.
struct Vertex {
    bool volume;
    V3 p;      // volume: scatter position; surface: Isect::p (point on a rectangle / barycentrics / object-space point on a sphere)
    float g;   // volume: HG asymmetry
    int quad;  // surface: rectangle index, -2 - triangle, kSpherePrim + sphere (Isect::quad)
    float t;   // surface: tHit (only the guided build needs it, for p = ray.o + tHit * ray.d)
};

// the part of the path-loop iteration between distance sampling and the vertex, for a path that reached
// the surface (:350-412): emission with MIS, ISG bookkeeping, depth test
// Returns LI_END (the path is over), LI_VERTEX (it stands at a surface vertex) or -- full scenes with medium boundaries --
// LI_SKIP: the hit has no BSDF (Material "interface"), SkipIntersection moved the ray past it (:399-404) and the path loop
// goes round again without a vertex.  LI_END is 0: callers that cannot meet LI_SKIP read the result as a bool.
enum { LI_END = 0, LI_VERTEX = 1, LI_SKIP = 2 };
template <bool FULL = true, int BND = -1, class PC>
VDEV int li_surface_pre(const DScene &S, PathState &st, IsgSample &isg, PC &pc, const Isect &si, Spec tw) {
    constexpr bool kRec = std::remove_reference<decltype(pc.rec)>::type::kActive;
    VSPG_PROF(PS_SURF_PRE);
    if constexpr (kRec) pc.rec.add_transmittance_weight(tw);  // :350
    if (!si.hit) {  // :353-374: infinite light sources (this fork lists DeltaDirection lights among them, integrators.h:79)
        const int n_all = S.n_lights + S.n_inf;
        for (int k = 0; FULL && k < S.n_inf; ++k) {
            Spec Le = lds(S.inf_L[k]);  // UniformInfiniteLight::Le / DistantLight::Le (lights.cpp:1014-1017, lights.h:291-293)
            if (S.inf_type[k] == VSPG_LIGHT_DISTANT && st.depth != 0) Le = sp(0.f);
            if (st.depth == 0 || st.specularBounce) {
                st.L = st.L + wdiv(st.beta * Le, avg(st.r_u));
                if constexpr (kRec) pc.rec.add_infinite_light_emission(st.ro + st.rd * kGuidingInfiniteLightDistance, Le, 1.0f);  // :361
            } else {
                // lightSampler.PMF * light.PDF_Li(prevIntrContext, ray.d, true): both light types return 0 for the incomplete PDF
                const float pmf = S.lsamp.mode != VSPG_LIGHTSAMPLER_UNIFORM ? light_sampler_pmf(S, st.prevCtx.template expand<FULL>(S).pi.mid(), st.prevCtx.template expand<FULL>(S).n, S.n_lights + k)
                                                                            : wrcp((float)n_all);
                const float lightPDF = pmf * 0.f;
                st.r_l = st.r_l * lightPDF;
                const float w_b = S.prm.usenee ? wrcp(avg(st.r_u + st.r_l)) : 1.f;
                st.L = st.L + st.beta * w_b * Le;
                if constexpr (kRec) pc.rec.add_infinite_light_emission(st.ro + st.rd * kGuidingInfiniteLightDistance, Le, w_b);  // :369
            }
        }
        return LI_END;
    }
    const bool tri_hit = FULL && (is_tri(si.quad) || is_sphere(si.quad));
    const DQuad &q = quad_at(tri_hit ? 0 : si.quad);
    Spec Le = !tri_hit && q.is_light ? light_L(q, si.n, -st.rd) : sp(0.f);  // :377 (triangles and spheres carry no area light)
    float w_direct = 0.f;
    if (nonzero(Le)) {
        if (st.depth == 0 || st.specularBounce) {
            st.L = st.L + wdiv(st.beta * Le, avg(st.r_u));
            w_direct = 1.0f;
        } else {
            const LsCtx pctx = st.prevCtx.template expand<FULL>(S);
            const float pmf = FULL && S.lsamp.mode != VSPG_LIGHTSAMPLER_UNIFORM ? light_sampler_pmf(S, pctx.pi.mid(), pctx.n, S.lsamp.light_of_quad[si.quad])
                                                                                 : wrcp((float)(FULL ? S.n_lights + S.n_inf : S.n_lights));
            float lightPDF = pmf * light_pdf_li(q, pctx, st.rd);
            st.r_l = st.r_l * lightPDF;
            float w_l = S.prm.usenee ? wrcp(avg(st.r_u + st.r_l)) : 1.0f;
            st.L = st.L + st.beta * w_l * Le;
            w_direct = w_l;
        }
    }
    if (FULL && has_bnd<BND>(S)) {  // :399-404: GetBSDF of an interface material returns no BSDF -- a medium boundary
        const int sfl = surf_flags(S, si.quad);
        if (sfl & SURF_INTERFACE) {
            // isect.SkipIntersection(&ray, si->tHit) (interaction.cpp:91-97): ray = SpawnRay(ray.d), medium = GetMedium(ray.d)
            st.ro = offset_ray_origin(surf_pi(S, si.quad, si.p), si.n, st.rd);
            st.in_medium = S.medium_type != VSPG_MEDIUM_NONE && get_medium(sfl, si.n, st.rd, st.in_medium);
            return LI_SKIP;
        }
    }
    if constexpr (kRec) {
        pc.rec.new_segment(st.ro + st.rd * si.t, false);                // guiding_newSurfacePathSegment (:406)
        if (nonzero(Le)) pc.rec.add_surface_emission(Le, w_direct);      // :409-412
    }
    if (st.depth == 0) {
        isg.valid = true;
        isg.surface_event = true;
    }
    if (st.depth++ >= S.prm.maxdepth) return LI_END;
    pc.surface_hit();
    return LI_VERTEX;
}

template <class Medium, bool GUIDED = false, int SEG = SEG_ANY, class PC>
VDEV int li_segment_a(const DScene &S, const Medium &medium, const float *vsp_buf, int vsp_ready, int px, int py,
                      PathState &st, int ch, Sampler &sampler, IsgSample &isg, PC &pc, Vertex &vx) {
    VSPG_PROF(PS_SEGMENT);
    pc.segment();
    Isect si;
    {
        VSPG_PROF(PS_INTERSECT);
        si = scene_intersect<!Medium::kSimpleScene>(S, st.ro, st.rd, kInf);
    }
    float tMax = si.hit ? si.t : kInf;
    vx.volume = false;
    vx.p = si.p;
    vx.g = 0;
    vx.quad = si.quad;
    vx.t = si.t;
    constexpr bool kRec = std::remove_reference<decltype(pc.rec)>::type::kActive;
    Spec tw = sp(1.f);  // transmittanceWeight (:317)
    // :318 `if (ray.medium && !std::isinf(tMax))`: a ray that escapes the scene is not sampled (rounds 1-3 did: wrong).  Without
    // medium boundaries every ray is in the scene's medium.
    const bool in_medium = Medium::kSimpleScene || !has_bnd<Medium::kBnd>(S) ? S.medium_type != VSPG_MEDIUM_NONE : st.in_medium;
    if (in_medium && si.hit) {
        Rng rng;
        {
            VSPG_PROF(PS_HASHRNG);
            uint64_t hash0 = hash_float(sampler.get1d());
            uint64_t hash1 = hash_float(sampler.get1d());
            rng.set_sequence(hash0, hash1);
        }
        DistEvent ev = sample_distance<Medium, GUIDED, SEG>(S, medium, vsp_buf, vsp_ready, px, py, st, tMax, ch, sampler, rng, isg, pc, tw);
        if (ev.kind == EV_TERMINATE || !nonzero(st.beta) || !nonzero(st.r_u)) return LI_END;  // :343-344
        if (ev.kind == EV_SCATTER) {
            vx.volume = true;
            vx.p = ev.p;
            vx.g = ev.g;
            return LI_VERTEX;
        }
    }
    return li_surface_pre<!Medium::kSimpleScene, Medium::kBnd>(S, st, isg, pc, si, tw);
}

// GREG: the guiding scratch lives in registers (GStoreReg, the workgroup kernel) instead of the lane's LDS column (glds, gstride)
template <class Medium, bool GUIDED, bool GREG, class PC>
VDEV bool li_vertex_guided(const DScene &S, const Medium &medium, PathState &st, int ch, Sampler &sampler, PC &pc,
                           bool volume_vertex, V3 vp, float vg, const Isect &si, Intr &intr, Bsdf &bsdf, float *glds, int gstride);
// the same with the NEE as a hook: nee(gd, useGuiding) -> Ld.  SPLIT (the wavefront pipeline): the hook only SETS UP the shadow ray
// (sample_Ld_begin); its result is added to L by the next kernel, with the throughput the hook saw
template <class Medium, bool GREG, bool SPLIT, class PC, class NeeFn>
VDEV bool li_vertex_guided_impl(const DScene &S, const Medium &medium, PathState &st, int ch, Sampler &sampler, PC &pc,
                                bool volume_vertex, V3 vp, float vg, const Isect &si, Intr &intr, Bsdf &bsdf, float *glds, int gstride,
                                NeeFn &&nee);

// li_segment_b in three pieces, so that a scheduler can park a path between them (the per-lane state
// machine for grid media runs the NEE transmittance walk step by step in between):
//   vertex_setup   Interaction / BSDF of the vertex (a pure function of the path state and the vertex)
//   vertex_pre     the cache-init sampler dimension and the volume vertex's survival probability
//   vertex_tail    Russian roulette + new direction
struct VertexCtx {
    Isect si;
    Intr intr;
    Bsdf bsdf;
};
template <bool GREY_KD = false, bool FULL = true, int BND = -1>
VDEV void vertex_setup(const DScene &S, const PathState &st, const Vertex &vx, VertexCtx &c) {
    Isect &si = c.si;
    Intr &intr = c.intr;
    Bsdf &bsdf = c.bsdf;
    const bool volume_vertex = vx.volume;
    const V3 vp = vx.p;
    const float vg = vx.g;
    si.hit = !vx.volume;
    si.t = vx.t;
    si.quad = vx.quad;
    si.p = vx.p;
    si.n = mk(0, 0, 0);
    if (volume_vertex) {
        // MediumInteraction intr(p, -ray.d, ...) (:806 / :990)
        intr.is_surface = false;
        intr.pi = p3i_exact(vp);
        intr.n = mk(0, 0, 0);
        intr.wo = -st.rd;
        intr.g = vg;
        bsdf.has_lobes = false;
        intr.medium = true;  // MediumInteraction(p, wo, time, ray.medium, phase): a scatter happens inside the medium
        intr.sflags = 0;
    } else {
        intr.medium = FULL ? st.in_medium : true;
        intr.sflags = FULL && has_bnd<BND>(S) ? surf_flags(S, vx.quad) : 0;
        if (FULL && is_sphere(vx.quad)) {  // vx.p = the object-space hit point (kSpherePrim)
            const DSphere &sph = S.spheres[sphere_of(vx.quad)];
            const SphereSurf ss = sphere_interaction(sph, vx.p);
            si.n = ss.n;
            bsdf = bsdf_make_sphere(sph, ss);
            intr.pi = ss.pi;
        } else if (FULL && is_tri(vx.quad)) {
            const DTri &T = S.tris[tri_of(vx.quad)];
            si.n = V3{T.nx, T.ny, T.nz};
            bsdf = bsdf_make_tri(T);
            intr.pi = tri_pi(T, vx.p);
        } else {
            const DQuad &q = quad_at(vx.quad);
            si.n = ld3(q.n);
            bsdf = bsdf_make<GREY_KD>(q);
            intr.pi = p3i_from_err(si.p, ld3(q.perr));
        }
        intr.is_surface = true;
        intr.n = si.n;
        intr.wo = normalize(-st.rd);  // Interaction ctor normalises wo (interaction.h:31-32)
        intr.g = 0;
    }
}
VDEV float vertex_pre(const DScene &S, const PathState &st, Sampler &sampler, const Vertex &vx) {
    float survivalProb = 1.f;
    (void)sampler.get1d();  // v: gbsdf.init / gphase.init with an untrained field (:457-458, :809-810)
    if (vx.volume && st.depth > S.prm.minrrdepth) {  // :817-830: survival probability BEFORE the NEE
        Spec rrw = wdiv(st.beta, avg(st.r_u)) * st.rr_correction;
        survivalProb = st.specularBounce ? 0.95f : standard_throughput_rr(rrw);
    }
    return survivalProb;
}
template <bool FULL = true, int BND = -1>
VDEV bool vertex_tail(const DScene &S, PathState &st, Sampler &sampler, const Vertex &vx, const VertexCtx &c, float survivalProb) {
    const bool volume_vertex = vx.volume;
    const V3 vp = vx.p;
    const float vg = vx.g;
    const Isect &si = c.si;
    const Intr &intr = c.intr;
    const Bsdf &bsdf = c.bsdf;
    // ---- new direction.  Phase-function lanes (HG, :842-874) and BSDF lanes (cosine hemisphere,
    // :487-519) both need sinf/cosf of one angle: each side prepares its angle, the sin/cos are
    // evaluated once for all lanes, each side finishes.  Per lane the operations and the sampler
    // order are the reference's.
    float ang = 0, a0 = 0, a1 = 0;
    bool degenerate = false;
    if (volume_vertex) {
        VSPG_PROF(PS_VOL_SAMPLE);
        if (survivalProb < 1 && st.depth > S.prm.minrrdepth) {  // :842-849
            float q = fmax_(0.f, 1 - survivalProb);
            if (sampler.get1d() < q) return false;
            st.beta = wdiv(st.beta, 1 - q);
        }
        float u0 = sampler.get1d(), u1 = sampler.get1d();
        ang = hg_pre(vg, u0, u1, &a0, &a1);  // a0 = sinTheta, a1 = cosTheta
    } else {
        VSPG_PROF(PS_SURF_SAMPLE);
        st.prevCtx.p = si.p;  // :487 LightSampleContext(isect)
        st.prevCtx.quad = vx.quad;
        (void)sampler.get1d();  // u (unused by DiffuseBxDF)
        float u20 = sampler.get1d(), u21 = sampler.get1d();
        // BSDF::Sample_f / DiffuseBxDF::Sample_f (bsdf.h:58-78, bxdfs.h:47-58)
        a1 = bsdf.frame.to_local(-st.rd).z;  // wo.z in the shading frame
        if (a1 == 0 || !bsdf.has_lobes) return false;
        ang = cos_hemi_pre(u20, u21, &a0, &degenerate);  // a0 = r
    }
    const float sinA = sinf_(ang), cosA = cosf_(ang);

    if (volume_vertex) {
        VSPG_PROF(PS_VOL_SAMPLE);
        float pdf;
        V3 wi = hg_post(-st.rd, vg, a0, a1, sinA, cosA, &pdf);  // gphase.Sample_p(-ray.d, u)
        if (pdf == 0) return false;
        // ps->p / ps->pdf: the phase function's value IS its pdf, so the weight is x / x -- exactly 1 for every finite non-zero
        // x (pdf == 0 returned above), and beta * 1 is beta; the division only runs where it is not (an infinite or NaN pdf)
        if (!(pdf < kInf)) st.beta = st.beta * (pdf / pdf);
        st.r_l = wdiv(st.r_u, pdf);
        st.prevCtx.p = vp;
        st.prevCtx.quad = -1;
        st.ro = vp;
        st.rd = wi;
        st.specularBounce = false;
        st.anyNonSpecularBounces = true;
        st.lastVertexVolume = true;
        return true;
    }

    VSPG_PROF(PS_SURF_SAMPLE);
    V3 wl = cos_hemi_post(a0, degenerate, sinA, cosA);
    if (a1 < 0) wl.z *= -1;
    float pdf = __builtin_fabsf(wl.z) * kInvPi;
    Spec f = bsdf.R * kInvPi;
    if (!nonzero(f) || pdf == 0 || wl.z == 0) return false;
    V3 wi = bsdf.frame.from_local(wl);
    st.lastVertexVolume = false;
    if (!(pdf < kInf)) st.rr_correction *= pdf / pdf;  // bs->pdf / bs->bsdfPdf == x / x == 1 for finite non-zero x (pdf == 0 returned above)
    Spec bsdfWeight = wdiv(f * absdot(wi, si.n), pdf);
    st.beta = st.beta * bsdfWeight;
    st.r_l = wdiv(st.r_u, pdf);  // misPdf == pdf without guiding
    st.specularBounce = false;
    st.anyNonSpecularBounces = true;
    st.ro = offset_ray_origin(intr.pi, si.n, wi);  // SpawnRay (interaction.h:99-101) ...
    st.rd = wi;
    if (FULL && has_bnd<BND>(S)) st.in_medium = S.medium_type != VSPG_MEDIUM_NONE && get_medium(intr.sflags, si.n, wi, st.in_medium);  // ... with GetMedium(wi)

    if (!nonzero(st.beta)) return false;
    if (st.depth > S.prm.minrrdepth) {
        Spec rrw = wdiv(st.beta, avg(st.r_u)) * st.rr_correction * st.etaScale;
        survivalProb = st.specularBounce ? 0.95f : standard_throughput_rr(rrw);
    }
    if (survivalProb < 1 && st.depth > S.prm.minrrdepth) {
        float qq = fmax_(0.f, 1 - survivalProb);
        if (sampler.get1d() < qq) return false;
        st.beta = wdiv(st.beta, 1 - qq);
    }
    return true;
}

template <class Medium, bool GUIDED = false, bool GREG = false, class PC>
VDEV bool li_segment_b(const DScene &S, const Medium &medium, PathState &st, int ch, Sampler &sampler, PC &pc,
                       const Vertex &vx, float *glds = nullptr, int gstride = 0) {
    VertexCtx c;
    vertex_setup<(Medium::kGrey >= 2), !Medium::kSimpleScene, Medium::kBnd>(S, st, vx, c);
    if constexpr (GUIDED)
        return li_vertex_guided<Medium, GUIDED, GREG>(S, medium, st, ch, sampler, pc, vx.volume, vx.p, vx.g, c.si, c.intr, c.bsdf, glds, gstride);
    const float survivalProb = vertex_pre(S, st, sampler, vx);
    if (S.prm.usenee && (vx.volume || c.bsdf.has_lobes)) {  // :479 IsNonSpecular(bsdf.Flags()) / :833
        VSPG_PROF(PS_NEE);
        Spec Ld = sample_Ld(S, medium, c.intr, &c.bsdf, ch, sampler, st.r_u, pc);
        st.L = st.L + st.beta * Ld;
    }
    return vertex_tail<!Medium::kSimpleScene, Medium::kBnd>(S, st, sampler, vx, c, survivalProb);
}

template <class Medium, bool GUIDED = false, class PC>
VDEV bool li_segment(const DScene &S, const Medium &medium, const float *vsp_buf, int vsp_ready, int px, int py,
                     PathState &st, int ch, Sampler &sampler, IsgSample &isg, PC &pc, float *glds = nullptr,
                     int gstride = 0) {
    Vertex vx;
    const int a = li_segment_a<Medium, GUIDED>(S, medium, vsp_buf, vsp_ready, px, py, st, ch, sampler, isg, pc, vx);
    if (a != LI_VERTEX) return a != LI_END;  // LI_SKIP: a medium boundary was crossed, the loop goes round again
    return li_segment_b<Medium, GUIDED, false>(S, medium, st, ch, sampler, pc, vx, glds, gstride);
}

// ---------------------------------------------------------------------------------------
// vertex processing with the guiding cache active (guided kernel instantiations only):
// gbsdf.init / gphase.init, NEE with the guided PDF, MIS / RIS sampling (guiding.h:83-109, 120-269,
// 383-398, 404-540) -- same flow as the unguided tail of li_segment, written straight (no sin/cos
// fusion) because the sampling branches differ per guiding type.
// ---------------------------------------------------------------------------------------
template <class Medium, bool GUIDED, bool GREG, class PC>
VDEV bool li_vertex_guided(const DScene &S, const Medium &medium, PathState &st, int ch, Sampler &sampler, PC &pc,
                           bool volume_vertex, V3 vp, float vg, const Isect &si, Intr &intr, Bsdf &bsdf, float *glds, int gstride) {
    return li_vertex_guided_impl<Medium, GREG, false>(S, medium, st, ch, sampler, pc, volume_vertex, vp, vg, si, intr, bsdf, glds, gstride,
                                                      [&](const auto &gd, bool use_gd) {
                                                          return sample_Ld(S, medium, intr, &bsdf, ch, sampler, st.r_u, pc, &gd, use_gd);
                                                      });
}
template <class Medium, bool GREG, bool SPLIT, class PC, class NeeFn>
VDEV bool li_vertex_guided_impl(const DScene &S, const Medium &medium, PathState &st, int ch, Sampler &sampler, PC &pc,
                                bool volume_vertex, V3 vp, float vg, const Isect &si, Intr &intr, Bsdf &bsdf, float *glds, int gstride,
                                NeeFn &&nee) {
    (void)sampler.get1d();  // v (the stochastic-lookup sample of Init)
    using Store = typename std::conditional<GREG, GStoreReg, GStoreLds>::type;
    using GD = GDistT<Store>;
    GD gd;
    if constexpr (!GREG) gd.st = gstore_lds(glds, gstride);
    bool useGuiding;
    float survivalProb = 1.f;
    // gphase.init(&phase, p, ray.d, v) / gbsdf.init(&bsdf, ray, si, v): ONE Init call site for both vertex kinds -- a
    // wavefront usually holds both, and two inlined copies of the most expensive guiding routine would run one after
    // the other at half occupancy each.  The kinds differ in data only: field, point, product lobe.
    int gfield;
    V3 gpoint, gm2;
    float gk2;
    bool gprod, ginit;
    if (volume_vertex) {
        gfield = 1;
        gpoint = vp;
        ginit = true;
        volume_product_lobe(st.rd, vg, &gprod, &gm2, &gk2);
    } else {
        gfield = 0;
        gpoint = st.ro + st.rd * si.t;  // p = ray.o + si->tHit * ray.d (guiding.h:85)
        ginit = bsdf.has_lobes;
        gprod = true;
        gm2 = si.n;
        if (dot(-st.rd, si.n) < 0.f) gm2 = -gm2;
        gk2 = kCosineLobeKappa;
    }
    gd.ok = false;
    gd.region = -1;
    // (register home: `glds` carries the workgroup's LDS copy of the upper kd levels instead of a scratch column)
    if (ginit) gdist_init(gd, S.field, gfield, gpoint, gprod, gm2, gk2, GREG ? reinterpret_cast<const VspgKdNode *>(glds) : nullptr);
    const bool useScatterGuiding = S.prm.vspsecondaryguiding ? gd.ok : false;
    st.gs.vsp_next = -1.f;
    if (volume_vertex) {
        useGuiding = S.prm.volumeguiding ? gd.ok : false;
        if (st.depth > S.prm.minrrdepth) {  // :817-830
            if (st.guideRR) {
                survivalProb = S.prm.volumerrguiding ? (st.specularBounce ? 0.95f : guided_russian_roulette(st.beta, st.pce, 0.1f)) : 1.f;
            } else {
                Spec rrw = (st.beta / avg(st.r_u)) * st.rr_correction;
                survivalProb = st.specularBounce ? 0.95f : standard_throughput_rr(rrw);
            }
        }
    } else {
        useGuiding = S.prm.surfaceguiding ? gd.ok : false;
        if (st.guideRR && st.depth > S.prm.minrrdepth)  // :465-472, BEFORE the NEE with the pre-bounce throughput
            survivalProb = S.prm.surfacerrguiding ? (st.specularBounce ? 0.95f : guided_russian_roulette(st.beta, st.pce, 0.1f)) : 1.f;
    }
    if (S.prm.usenee && (volume_vertex || bsdf.has_lobes)) {
        if constexpr (SPLIT) {
            (void)nee(gd, useGuiding);
        } else {
            Spec Ld = nee(gd, useGuiding);
            st.L = st.L + st.beta * Ld;
            pc.rec.add_scattered_direct_light(Ld);  // :485 / :838
        }
    }

    // ---- new direction: Sample_p / Sample_f, MIS or RIS flavour (guiding.h:120-257, 404-530) ----------------------
    // One flow for volume and surface vertices and for both flavours: the expensive mixture routines (gdist_pdf,
    // gdist_sample, the two IncomingRadiancePDF evaluations) have ONE call site each, entered by every lane that needs
    // them, whatever its vertex kind -- a wavefront holds both kinds, and the reference's defaults even mix the flavours
    // (surfaces RIS, volumes MIS).  Per lane the operations, their order and the sampler dimensions are those of the four
    // separate routines; only the cheap scattering-function pieces (Henyey-Greenstein / diffuse BSDF) branch on the kind.
    if (volume_vertex) {  // Russian roulette comes BEFORE the direction at a volume vertex (:848-853)
        if (survivalProb < 1 && st.depth > S.prm.minrrdepth) {
            float q = fmax_(0.f, 1 - survivalProb);
            if (sampler.get1d() < q) return false;
            st.beta = st.beta / (1 - q);
        }
    } else {
        st.prevCtx.p = si.p;
        st.prevCtx.quad = si.quad;
    }
    const V3 wo = -st.rd;
    float usel = 0;  // surface: the MIS selector u; volume: u0 doubles as selector
    if (!volume_vertex) usel = sampler.get1d();
    float ua = sampler.get1d(), ub = sampler.get1d();  // u0,u1 (volume) / u2 (surface)
    const bool ris = useGuiding && (volume_vertex ? S.prm.volumeguidingtype : S.prm.surfaceguidingtype) != VSPG_GUIDE_MIS;
    bool sampleSF = true;  // MIS: take the direction from the scattering function (else from the mixture)
    if (useGuiding && !ris) {
        if (volume_vertex) {
            if (kGuidingProbability > ua) {
                ua /= kGuidingProbability;
                sampleSF = false;
            } else {
                ua -= kGuidingProbability;
                ua /= (1.0f - kGuidingProbability);
            }
        } else if (kGuidingProbability > usel) {
            sampleSF = false;
        }
    }
    // candidate 0: the scattering function's own sample (unguided lanes, RIS, MIS when it was chosen)
    V3 w0 = mk(0, 0, 0);
    float sf0 = 0;        // its pdf (== its value for the phase function)
    Spec f0 = sp(0.f);    // surface: BSDF value
    bool ok0 = false;
    if (!useGuiding || ris || sampleSF) {
        // SampleHenyeyGreenstein and DiffuseBxDF::Sample_f both need sin/cos of one angle: one evaluation for both kinds
        // (as in vertex_tail)
        float ang, a0 = 0, a1 = 0;
        bool degenerate = false, okf = true;
        if (volume_vertex) {
            ang = hg_pre(vg, ua, ub, &a0, &a1);  // a0 = sinTheta, a1 = cosTheta
        } else {
            a1 = bsdf.frame.to_local(wo).z;     // BSDF::Sample_f / DiffuseBxDF::Sample_f (bsdf.h:58-78, bxdfs.h:47-58)
            okf = !(a1 == 0 || !bsdf.has_lobes);
            ang = cos_hemi_pre(ua, ub, &a0, &degenerate);  // a0 = r
        }
        const float sinA = sinf_(ang), cosA = cosf_(ang);
        if (volume_vertex) {
            w0 = hg_post(wo, vg, a0, a1, sinA, cosA, &sf0);
            ok0 = true;
        } else if (okf) {
            V3 wl = cos_hemi_post(a0, degenerate, sinA, cosA);
            if (a1 < 0) wl.z *= -1;
            sf0 = __builtin_fabsf(wl.z) * kInvPi;
            f0 = bsdf.R * kInvPi;
            if (!(!nonzero(f0) || sf0 == 0 || wl.z == 0)) {
                w0 = bsdf.frame.from_local(wl);
                ok0 = true;
            }
        }
    }
    float g0 = 0, inc0 = 0, g1 = 0, inc1 = 0;
    V3 w1 = mk(0, 0, 0);
    if (useGuiding && ok0 && (ris || sampleSF)) g0 = gdist_pdf(gd, w0);
    float s0 = ua, s1 = ub;  // MIS samples the mixture with the vertex's own 2D sample, RIS with a fresh one
    if (ris) {
        s0 = sampler.get1d();
        s1 = sampler.get1d();
    }
    const bool cand1 = useGuiding && (ris || !sampleSF);
    if (cand1) g1 = gdist_sample(gd, s0, s1, &w1);
    if (ris) gdist_incoming_pdf2(S.field, gd, ok0, w0, w1, &inc0, &inc1);
    // scattering function at the mixture's direction
    float sf1 = 0;
    Spec f1 = sp(0.f);
    if (cand1) {
        if (volume_vertex) {
            sf1 = henyey_greenstein(dot(wo, w1), vg);
        } else {
            f1 = bsdf_f(bsdf, wo, w1);
            sf1 = bsdf_pdf(bsdf, wo, w1);
        }
    }
    // combine
    V3 wi = mk(0, 0, 0);
    Spec f = sp(0.f);
    float pdf = 0, sfPdf = 0, misPdf = 0;
    bool have = false;
    if (!useGuiding) {
        have = ok0;
        wi = w0; f = f0;
        pdf = sfPdf = misPdf = sf0;
    } else if (!ris) {
        if (sampleSF) {
            if (ok0) {
                wi = w0; f = f0;
                sfPdf = sf0;
                pdf = ((1.0f - kGuidingProbability) * sf0) + (kGuidingProbability * g0);
                misPdf = pdf;
                have = true;
            }
        } else if (sf1 > 0.f) {
            wi = w1; f = f1;
            sfPdf = sf1;
            pdf = ((1.0f - kGuidingProbability) * sf1) + (kGuidingProbability * g1);
            misPdf = pdf;
            have = true;
        }
    } else {
        const float rb0 = ok0 ? sf0 : 0.f;
        const float rmis0 = ok0 ? 0.5f * (rb0 + g0) : 0.f;
        const float rmis1 = 0.5f * (sf1 + g1);
        float sumW = 0.f, rw0 = 0.f, rw1 = 0.f;
        int nS = 0;
        if (rb0 > 0.f) {
            rw0 = (rb0 * ((1.0f - kGuidingProbability) * kUniformIncomingRadiancePDF + kGuidingProbability * inc0));
            rw0 /= rmis0;
            sumW += rw0;
            nS++;
        }
        if (sf1 > 0.f) {
            rw1 = (sf1 * ((1.0f - kGuidingProbability) * kUniformIncomingRadiancePDF + kGuidingProbability * inc1));
            rw1 /= rmis1;
            sumW += rw1;
            nS++;
        }
        if (!(nS == 0 || sumW <= 0.f)) {
            float sample1D = sumW * sampler.get1d();
            // for (i < 2) { sumRis += w_i; if (sample1D <= sumRis) { idx = i; break; } }  (idx defaults to 0)
            float sumR = 0.f + rw0;
            int idx = 0;
            if (!(sample1D <= sumR)) {
                sumR += rw1;
                if (sample1D <= sumR) idx = 1;
            }
            const float rw = idx ? rw1 : rw0;
            misPdf = idx ? rmis1 : rmis0;
            pdf = (rw * misPdf) * ((float)2 / sumW);
            sfPdf = idx ? sf1 : rb0;
            f = idx ? f1 : f0;
            wi = idx ? w1 : w0;
            have = true;
        }
    }

    bool cont = false;  // the path goes on with (st.ro, st.rd)
    if (volume_vertex) {
        if (have && pdf != 0) {
            float w = sfPdf / pdf;  // ps->p / ps->pdf
            st.beta = st.beta * w;
            st.r_l = st.r_u / pdf;
            st.prevCtx.p = vp;
            st.prevCtx.quad = -1;
            st.ro = vp;
            st.rd = wi;
            st.specularBounce = false;
            st.anyNonSpecularBounces = true;
            // guiding_addVolumeData(..., phaseFunctionWeight, ps->wi, ps->pdf, ps->meanCosine, survivalProb) (:871)
            pc.rec.add_scatter_data(true, sp(w), wi, pdf, 1.0f - __builtin_fabsf(vg), survivalProb);
            st.lastVertexVolume = true;
            cont = true;
        }
    } else if (have) {
        st.lastVertexVolume = false;
        st.rr_correction *= pdf / sfPdf;
        Spec bsdfWeight = f * absdot(wi, si.n) / pdf;
        st.beta = st.beta * bsdfWeight;
        st.r_l = st.r_u / misPdf;
        st.specularBounce = false;
        st.anyNonSpecularBounces = true;
        st.ro = offset_ray_origin(intr.pi, si.n, wi);
        st.rd = wi;
        if (!Medium::kSimpleScene && has_bnd<Medium::kBnd>(S)) st.in_medium = S.medium_type != VSPG_MEDIUM_NONE && get_medium(intr.sflags, si.n, wi, st.in_medium);
        if (nonzero(st.beta)) {
            if (!st.guideRR && st.depth > S.prm.minrrdepth) {  // :597-600
                Spec rrw = (st.beta / avg(st.r_u)) * st.rr_correction * st.etaScale;
                survivalProb = st.specularBounce ? 0.95f : standard_throughput_rr(rrw);
            }
            cont = true;
            if (survivalProb < 1 && st.depth > S.prm.minrrdepth) {
                float qq = fmax_(0.f, 1 - survivalProb);
                if (sampler.get1d() < qq) cont = false;
                else st.beta = st.beta / (1 - qq);
            }
            // guiding_addSurfaceData(..., bsdfWeight, bs->wi, bs->eta, bs->sampledRoughness, bs->pdf, survivalProb) (:608);
            // DiffuseBxDF: sampledRoughness 1
            if (cont) pc.rec.add_scatter_data(false, bsdfWeight, wi, pdf, 1.0f, survivalProb);
        }
    }
    // the next segment's VolumeScatterProbability(ray.d) of this vertex's distribution (one call site for both kinds)
    if (cont && useScatterGuiding) st.gs.vsp_next = gdist_vsp(S.field, gfield, gd.region, gd, wi);
    return cont;
}

// EvaluatePixelSample up to the camera ray (src/pbrt/cpu/integrators.cpp:272-304)
VDEV void start_path_common(const DScene &S, int px, int py, Sampler &sampler, PathState &st, int *ch, IsgSample &isg);
VDEV void start_path(const DScene &S, const float *vsp_buf, int vsp_ready, int px, int py, int sampleIndex, Sampler &sampler,
                     PathState &st, int *ch, IsgSample &isg) {
    st.vsp0 = (vsp_ready & VSP_READY) ? vsp_buf[(size_t)py * S.xres + px] : 0.5f;
    sampler.start_pixel_sample(px, py, S.seed, sampleIndex);
    start_path_common(S, px, py, sampler, st, ch, isg);
}
VDEV void start_path(const DScene &S, const float *vsp_buf, int vsp_ready, int px, int py, PcgJump jump, Sampler &sampler,
                     PathState &st, int *ch, IsgSample &isg) {
    st.vsp0 = (vsp_ready & VSP_READY) ? vsp_buf[(size_t)py * S.xres + px] : 0.5f;
    sampler.start_pixel_sample(px, py, S.seed, jump);
    start_path_common(S, px, py, sampler, st, ch, isg);
}
VDEV void start_path_common(const DScene &S, int px, int py, Sampler &sampler, PathState &st, int *ch, IsgSample &isg) {
    float lu = sampler.get1d();
    st.lu = lu;
    int c = (int)__builtin_floorf(lu * 3);  // SampledWavelengths::SampleVisible (spectrum.h:380-384)
    *ch = c > 2 ? 2 : c;
    // GetCameraSample (samplers.h:796-815), BoxFilter radius .5 (filters.h:67-69)
    float f0 = sampler.get1d(), f1 = sampler.get1d();
    float fpx = (1 - f0) * -0.5f + f0 * 0.5f;
    float fpy = (1 - f1) * -0.5f + f1 * 0.5f;
    float pfx = ((float)px + fpx) + 0.5f, pfy = ((float)py + fpy) + 0.5f;
    (void)sampler.get1d();  // time
    (void)sampler.get1d();  // pLens
    (void)sampler.get1d();
    V3 pc_ = mk(S.cam.sx * pfx + S.cam.ox, S.cam.sy * pfy + S.cam.oy, 1.f);
    V3 dir = normalize(pc_);
    Frame f{ld3(S.cam.right), ld3(S.cam.up), ld3(S.cam.fwd)};
    st.ro = ld3(S.cam.origin);
    st.rd = f.from_local(dir);
    st.L = sp(0.f);
    st.beta = sp(1.f);
    st.r_u = sp(1.f);
    st.r_l = sp(1.f);
    st.prevCtx.p = mk(0, 0, 0);
    st.prevCtx.quad = -1;
    st.depth = 0;
    st.specularBounce = false;
    st.anyNonSpecularBounces = false;
    st.lastVertexVolume = false;
    st.rr_correction = 1.0f;
    st.etaScale = 1;
    st.gs.vsp_next = -1.f;
    st.pce = 0.f;
    st.guideRR = false;
    st.in_medium = S.camera_in_medium != 0;  // CameraBase::medium (scene.cpp:153-155)
    isg.valid = false;
    isg.surface_event = false;
    isg.vsp_used = -1.f;
}

// NaN / Inf radiance -> black (integrators.cpp:308-318)
VDEV Spec finish_radiance(Spec L) {
    if (has_nan(L)) return sp(0.f);
    if (has_inf(L)) return sp(0.f);
    return L;
}

// image-space VSP statistics (own design standing in for ImageSpaceGuidingBuffer::AddSample,
// guidedvolpathvspgintegrator.cpp:613-622; OpenPGL absent -> unpinned)
//   st[0]=n [1]=sum c[vol] [2]=sum c[surf] [3]=sum c^2 q [vol] [4]=sum c^2 (1-q) [surf] [5]=n_vol
// Fire-and-forget accumulation: the lane that owns the pixel is the only writer during a launch, so
// a no-return float atomic add gives the same single IEEE addition as load + add + store, but the
// wave never waits for the HBM round trip (a finished path would otherwise stall its whole
// wavefront for ~1-2 us every loop iteration).
#ifdef VSPG_EXP_NOFILM  // timing experiment only: results are wrong
VDEV void add_noret(float *p, float v) { (void)p; (void)v; }
#else
VDEV void add_noret(float *p, float v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
VDEV void film_add_sample(float4 *film_px, Spec L) {  // RGBFilm::AddSample (film.h:251-267): weight 1, no clamp
    float *f = reinterpret_cast<float *>(film_px);
    add_noret(f + 0, L.r);
    add_noret(f + 1, L.g);
    add_noret(f + 2, L.b);
    add_noret(f + 3, 1.f);
}
VDEV void isg_add_sample_atomic(float *st, Spec L, const IsgSample &isg) {
    if (!isg.valid) return;
    float c = avg(L);
    float q = isg.vsp_used >= 0.f ? isg.vsp_used : 0.5f;
    add_noret(st + 0, 1.f);
    if (isg.surface_event) {
        add_noret(st + 2, c);
        add_noret(st + 4, c * c * (1 - q));
    } else {
        add_noret(st + 1, c);
        add_noret(st + 3, c * c * q);
        add_noret(st + 5, 1.f);
    }
}
// Read-modify-write forms for launches in which a pixel receives exactly ONE sample (the benchmark's wave of
// 1 spp): no other path touches the pixel, so a plain 16-byte load / add / store is the same single IEEE
// addition per channel.  Ten no-return float atomics per finished path are L2-atomic-unit bound at 1080p
// (measured: 0.28 ms of a 0.42 ms maxdepth-0 wave, 0.12 ms of the 0.94 ms benchmark wave); the RMW moves
// 96 B per path through the ordinary load/store path instead.
VDEV void film_add_sample_rmw(float4 *film_px, Spec L) {
    float4 v = *film_px;
    v.x += L.r;
    v.y += L.g;
    v.z += L.b;
    v.w += 1.f;
    *film_px = v;
}
VDEV void isg_add_sample_rmw(float *st, Spec L, const IsgSample &isg) {
    if (!isg.valid) return;
    float c = avg(L);
    float q = isg.vsp_used >= 0.f ? isg.vsp_used : 0.5f;
    float4 *p4 = reinterpret_cast<float4 *>(st);   // st[0..3]
    float2 *p2 = reinterpret_cast<float2 *>(st + 4);  // st[4..5]
    float4 a = *p4;
    float2 b = *p2;
    a.x += 1.f;
    if (isg.surface_event) {
        a.z += c;
        b.x += c * c * (1 - q);
    } else {
        a.y += c;
        a.w += c * c * q;
        b.y += 1.f;
    }
    *p4 = a;
    *p2 = b;
}
VDEV void isg_add_sample(float *st, Spec L, const IsgSample &isg) {
    if (!isg.valid) return;
    float c = avg(L);
    float q = isg.vsp_used >= 0.f ? isg.vsp_used : 0.5f;
    st[0] += 1.f;
    if (isg.surface_event) {
        st[2] += c;
        st[4] += c * c * (1 - q);
    } else {
        st[1] += c;
        st[3] += c * c * q;
        st[5] += 1.f;
    }
}

VSPG_NS_END  // namespace vspg
