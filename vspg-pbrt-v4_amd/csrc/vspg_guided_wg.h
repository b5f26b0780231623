// vspg_guided_wg.h -- the guided vertex of the workgroup kernel (k_render_wave_wg2<..., GUIDED>), round 3.
//
// What it computes is li_vertex_guided_impl (vspg_path.h: gbsdf.init / gphase.init, NEE with the guided PDF, Russian
// roulette, MIS / RIS direction sampling, the next segment's VolumeScatterProbability -- guiding.h:83-109, 120-269, 383-398,
// 404-540, 295-305, 564-574) for one path parked in the workgroup's LDS pool: per path the same operations on the same
// inputs in the same order, the same sampler dimensions -- bit-identical to the per-lane kernel and the CPU checker
// (tests: wg == lane == oracle).  What differs is WHEN things are live.  The per-lane flow keeps the path state, the NEE's
// intermediate results and the 8-lobe mixture alive together (238 VGPRs, two waves per SIMD).  Here:
//   1. nothing of the path is in registers unless a step needs it -- the pool record is read field by field, late;
//   2. every direction the mixture will be asked about is known BEFORE the mixture exists: the NEE's direction needs only
//      the light sample, the scattering function's own candidate only sampler dimensions.  So the lobes are built once and
//      every sum over them that does not depend on the mixture's own sample is accumulated in that same pass
//      (PDF at the NEE direction, PDF / IncomingRadiancePDF / VolumeScatterProbability at candidate 0);
//   3. a second pass over the lobe registers serves the mixture's own sample (PDF / IncomingRadiancePDF / VSP at candidate 1);
//   4. the NEE's shadow ray -- the register-hungriest piece next to the mixture -- runs AFTER the mixture is dead; only
//      {light, two sampler values, one PDF} cross it, and the light sample is re-derived from them (same inputs, same bits).
// Region records are read through DField::lobes, an array-of-lobes copy built once per field update (k_field_aux): two
// 16-byte loads per lobe, consumed at once, instead of seven 16-byte rows per four lobes held until used.
#pragma once
#include "vspg_wg_kernel.h"

VSPG_NS_BEGIN

// DField::lobes, per region 1 + 2 GK float4:  [0] {pivot, n_lobes (int bits)}   [1 + 2k] {mu_k, distance_k}   [2 + 2k] {weight_k, b_k, kappa_k (clamped), vsp_k}
constexpr int kRegionLobeQuads = 1 + 2 * GK;
// The records are read through a GLOBAL-address-space pointer: through the generic pointer a kernel argument's member is,
// the compiler emits flat loads and waits for each one where it stands (seen in the ISA: sixteen exposed L2 round trips per
// vertex); global loads can be issued ahead -- the loops below fetch lobe k + 1 while lobe k is being worked on.
struct GlobalQuads {
    const float4 *p;
    VDEV float4 operator[](int i) const { return ld4_global(p + i); }
};
VDEV GlobalQuads region_lobes(const DField &F, int region) { return GlobalQuads{F.lobes + (size_t)region * kRegionLobeQuads}; }

// light.SampleLi for the light the NEE picked: everything sample_Ld derives from (lightIndex, ul0, ul1) at this vertex
struct NeePick {
    bool ok;         // a light sample with non-zero f_hat exists (sample_Ld has not returned yet)
    int lightIndex;
    float ul0, ul1;
    V3 wi;           // ls.wi
    float sfPDF;     // the scattering function's own PDF at wi (bsdfPDF / phasePDF before the guided mix)
};

template <class Medium, class PC>
VDEV bool li_vertex_guided_wg(const DScene &S, const Medium &medium, const Pool &P, int slot, uint32_t fl, PC &pc,
                              const VspgKdNode *kd_lds, Spec *L_out, IsgSample *isg_out) {
    constexpr bool kFull = !Medium::kSimpleScene;
    static_assert(!kFull, "the workgroup kernel's homogeneous instantiations are rectangle scenes with area lights");
    constexpr int G = Medium::kGrey;
    using LY = PoolLayout<true, G>;
    const bool volume_vertex = (fl & FL_VX_VOLUME) != 0;
    const int depth = (int)(fl & FL_DEPTH_MASK);
    const int ch = (int)((fl >> FL_CH_SHIFT) & 3u);
    const bool specularBounce = (fl & FL_SPECULAR) != 0;
    Sampler sampler;
    pool_load_rng<LY>(P, slot, sampler);
    const Vertex vx = pool_load_vertex<true, G>(P, slot, fl);
    PathState st;  // only ro / rd are read below (vertex_setup, the query point)
    st.ro = P.v3(LY::RO, slot);
    st.rd = P.v3(LY::RD, slot);
    VertexCtx c;
    vertex_setup<(G >= 2), false>(S, st, vx, c);
    const float vg = vx.g;

    (void)sampler.get1d();  // v (the stochastic-lookup sample of Init)
    // ---- gphase.init / gbsdf.init: which field, where, which product lobe (li_vertex_guided_impl) ------------------------
    int gfield;
    V3 gpoint, gm2;
    float gk2;
    bool gprod, ginit;
    if (volume_vertex) {
        gfield = 1;
        gpoint = vx.p;
        ginit = true;
        volume_product_lobe(st.rd, vg, &gprod, &gm2, &gk2);
    } else {
        gfield = 0;
        gpoint = st.ro + st.rd * vx.t;  // p = ray.o + si->tHit * ray.d (guiding.h:85)
        ginit = c.bsdf.has_lobes;
        gprod = true;
        gm2 = c.si.n;
        if (dot(-st.rd, c.si.n) < 0.f) gm2 = -gm2;
        gk2 = kCosineLobeKappa;
    }
    DField F = S.field[0];
    if (gfield) F = S.field[1];
    int region = -1, n = 0;
    GlobalQuads RL = region_lobes(F, 0);
    float4 q0n = make_float4(0, 0, 0, 0), q1n = q0n;  // the next lobe's two quads, in flight
    V3 pp = mk(0, 0, 0);  // pivot - query point
    if (ginit) {
        const int nl = F.n_nodes < kKdLdsNodes ? F.n_nodes : kKdLdsNodes;
        region = field_lookup(F, gpoint, kd_lds + (gfield ? kKdLdsNodes : 0), nl);
        if (region >= 0) {
            RL = region_lobes(F, region);
            const float4 h = RL[0];
            q0n = RL[1];  // lobe 0, issued with the header (every region record holds GK lobes, set or not)
            q1n = RL[2];
            pp = V3{h.x, h.y, h.z} - gpoint;
            n = __builtin_bit_cast(int, h.w);
            n = n < GK ? n : GK;
        }
    }
    const bool gok = region >= 0 && n > 0;
    const bool useScatterGuiding = S.prm.vspsecondaryguiding ? gok : false;
    const bool useGuiding = (volume_vertex ? S.prm.volumeguiding : S.prm.surfaceguiding) ? gok : false;

    // ---- survival probability of a volume vertex, BEFORE the NEE (:817-830) ---------------------------------------------
    float survivalProb = 1.f;
    if (volume_vertex && depth > S.prm.minrrdepth) {
        const Spec rrw = (pool_load_beta<LY, G>(P, slot) / avg(pool_load_ru<LY, G>(P, slot))) * P.f(LY::RRC, slot);
        survivalProb = specularBounce ? 0.95f : standard_throughput_rr(rrw);
    }

    // ---- NEE, first half: the light and the direction (sample_Ld up to the shadow ray, :1136-1190) -----------------------
    NeePick nee;
    nee.ok = false;
    nee.lightIndex = 0;
    nee.ul0 = nee.ul1 = nee.sfPDF = 0.f;
    nee.wi = mk(0, 0, 0);
    const bool do_nee = S.prm.usenee && (volume_vertex || c.bsdf.has_lobes);
    if (do_nee) {
        V3 ctxp = c.intr.pi.mid();
        if (c.intr.is_surface && c.bsdf.has_lobes) ctxp = offset_ray_origin(c.intr.pi, c.intr.n, c.intr.wo);  // :1147-1149
        const float u = sampler.get1d();
        const int n_all = S.n_lights;
        nee.ul0 = sampler.get1d();
        nee.ul1 = sampler.get1d();
        if (n_all > 0) {
            const int li = (int)(u * (float)n_all);
            nee.lightIndex = li < n_all - 1 ? li : n_all - 1;
            LightLi ls;
            if (light_sample_li(light_quad_at(nee.lightIndex), ctxp, nee.ul0, nee.ul1, &ls)) {
                nee.wi = ls.wi;
                bool nz;
                if (c.intr.is_surface) {
                    nz = nonzero(bsdf_f(c.bsdf, c.intr.wo, ls.wi) * absdot(ls.wi, c.intr.n));
                    nee.sfPDF = bsdf_pdf(c.bsdf, c.intr.wo, ls.wi);
                } else {
                    const float p = henyey_greenstein(dot(c.intr.wo, ls.wi), c.intr.g);
                    nz = p != 0;
                    nee.sfPDF = p;
                }
                nee.ok = nz;
            }
        }
    }

    // ---- Russian roulette of a volume vertex (:842-849): the sampler dimension comes here; the NEE still counts --------
    bool dead = false;
    float rr_q = 0.f;
    if (volume_vertex && survivalProb < 1 && depth > S.prm.minrrdepth) {
        rr_q = fmax_(0.f, 1 - survivalProb);
        if (sampler.get1d() < rr_q) dead = true;
    }

    // ---- the vertex's own sampler dimensions and candidate 0, the scattering function's sample --------------------------
    float usel = 0, ua = 0, ub = 0, s0 = 0, s1 = 0;
    bool ris = false, sampleSF = true, ok0 = false;
    V3 w0 = mk(0, 0, 0);
    float sf0 = 0;
    if (!dead) {
        if (!volume_vertex) usel = sampler.get1d();
        ua = sampler.get1d();
        ub = sampler.get1d();
        ris = useGuiding && (volume_vertex ? S.prm.volumeguidingtype : S.prm.surfaceguidingtype) != VSPG_GUIDE_MIS;
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
        if (!useGuiding || ris || sampleSF) {
            float ang, a0 = 0, a1 = 0;
            bool degenerate = false, okf = true;
            if (volume_vertex) {
                ang = hg_pre(vg, ua, ub, &a0, &a1);
            } else {
                a1 = c.bsdf.frame.to_local(-st.rd).z;
                okf = !(a1 == 0 || !c.bsdf.has_lobes);
                ang = cos_hemi_pre(ua, ub, &a0, &degenerate);
            }
            const float sinA = sinf_(ang), cosA = cosf_(ang);
            if (volume_vertex) {
                w0 = hg_post(-st.rd, vg, a0, a1, sinA, cosA, &sf0);
                ok0 = true;
            } else if (okf) {
                V3 wl = cos_hemi_post(a0, degenerate, sinA, cosA);
                if (a1 < 0) wl.z *= -1;
                sf0 = __builtin_fabsf(wl.z) * kInvPi;
                const Spec f0 = c.bsdf.R * kInvPi;
                if (!(!nonzero(f0) || sf0 == 0 || wl.z == 0)) {
                    w0 = c.bsdf.frame.from_local(wl);
                    ok0 = true;
                }
            }
        }
        s0 = ua;  // MIS samples the mixture with the vertex's own 2D sample, RIS with a fresh one
        s1 = ub;
        if (ris) {
            s0 = sampler.get1d();
            s1 = sampler.get1d();
        }
    }
    // the sampler and the NEE's pick wait in the pool while the lobes hold the registers: the record's RNG field (the state is
    // written back at the end anyway) and the previous-vertex context (dead during a vertex: the tail below rewrites it)
    pool_store_rng<LY>(P, slot, sampler);
    P.i(LY::PCQ, slot) = nee.lightIndex;
    P.f(LY::PCP + 0, slot) = nee.ul0;
    P.f(LY::PCP + 1, slot) = nee.ul1;
    P.f(LY::PCP + 2, slot) = nee.sfPDF;
    const bool cand1 = !dead && useGuiding && (ris || !sampleSF);
    const bool want_g_nee = nee.ok && useGuiding;                       // gdist_pdf at the NEE direction
    const bool want_g0 = !dead && useGuiding && ok0 && (ris || sampleSF);  // gdist_pdf at candidate 0
    const bool want_e0 = !dead && ok0 && (ris || useScatterGuiding);       // IncomingRadiancePDF / VSP at candidate 0

    // ---- the mixture: pass 1 builds the lobes and takes every sum that does not depend on the mixture's own sample -------
    float g_nee = 0, g0 = 0, g1 = 0, den0 = 0, num0 = 0, den1 = 0, num1 = 0;
    V3 w1 = mk(0, 0, 0);
    if (gok) {
        float lr[GK][3], la[GK], lkc[GK], lc1[GK], lc2[GK], lwo[GK];  // registers: every index below is a compile-time constant
        const float nk2 = gprod ? vmf_norm(gk2) : 0.f;
        const float x2n = dot(gm2, nee.wi), x20 = dot(gm2, w0);
        float sum = 0, pn = 0, p0 = 0;
#pragma unroll
        for (int k = 0; k < GK; ++k) {
            lr[k][0] = lr[k][1] = lr[k][2] = la[k] = lkc[k] = lc1[k] = lc2[k] = lwo[k] = 0.f;
            if (k < n) {
                const float4 q0 = q0n, q1 = q1n;  // {mu, distance}, {weight, b, kappa, vsp}
                if (k + 1 < GK) {
                    q0n = RL[3 + 2 * k];
                    q1n = RL[4 + 2 * k];
                }
                const V3 raw = lobe_dir(pp, V3{q0.x, q0.y, q0.z}, q0.w);
                float a = q1.y, wo_k = q1.x, kc = q1.z, c1 = 1.f, c2 = 0.f;
                if (useGuiding && gprod) lobe_product(raw, q1.z, q1.y, gm2, gk2, nk2, &a, &wo_k, &kc, &c1, &c2);
                lr[k][0] = raw.x; lr[k][1] = raw.y; lr[k][2] = raw.z;
                la[k] = a; lkc[k] = kc; lc1[k] = c1; lc2[k] = c2; lwo[k] = wo_k;
                sum += wo_k;
                if (want_g_nee) pn += a * fast_exp(kc * ((c1 * dot(raw, nee.wi) + c2 * x2n) - 1));
                const float x0 = dot(raw, w0);
                if (want_g0) p0 += a * fast_exp(kc * ((c1 * x0 + c2 * x20) - 1));
                if (want_e0) {
                    const float e = q1.y * fast_exp(q1.z * (x0 - 1));
                    num0 += e * q1.w;
                    den0 += e;
                }
            }
        }
        float isum = 1.f;
        if (useGuiding) {
            if (sum > 0 && !isinf_(sum)) {
                isum = 1.0f / sum;
            } else {  // degenerate product: the incident-radiance mixture stands in (gdist_init's fallback); rare
                pn = p0 = 0;
#pragma unroll
                for (int k = 0; k < GK; ++k)
                    if (k < n) {
                        const float4 q1 = RL[2 + 2 * k];
                        const V3 raw = V3{lr[k][0], lr[k][1], lr[k][2]};
                        la[k] = q1.y; lkc[k] = q1.z; lc1[k] = 1.f; lc2[k] = 0.f; lwo[k] = q1.x;
                        if (want_g_nee) pn += q1.y * fast_exp(q1.z * ((1.f * dot(raw, nee.wi) + 0.f * x2n) - 1));
                        if (want_g0) p0 += q1.y * fast_exp(q1.z * ((1.f * dot(raw, w0) + 0.f * x20) - 1));
                    }
            }
        }
        g_nee = pn * isum;
        g0 = p0 * isum;
        // ---- candidate 1: the mixture's own sample (SamplePDF), then pass 2 over the lobe registers ------------------
        if (cand1) {
            bool scanning = true;
            int ks = 0;
            float acc = 0;
#pragma unroll
            for (int j = 0; j < GK - 1; ++j) {
                if (scanning && j < n - 1) {
                    const float wj = lwo[j] * isum;
                    if (s0 < acc + wj) {
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
                    wk = lwo[j] * isum;
                    kap = lkc[j];
                    c1 = lc1[j];
                    c2 = lc2[j];
                    raw = V3{lr[j][0], lr[j][1], lr[j][2]};
                }
            const V3 mz = c1 == 0.f && c2 == 0.f ? raw : raw * c1 + gm2 * c2;
            const bool want_e1 = ris || useScatterGuiding;
            float4 q1nn = q1n;
            if (want_e1) {  // pass 2's first two region quads travel while the direction is being computed
                q1n = RL[2];
                q1nn = RL[4];  // (unconditional: a region record holds GK lobes, set or not -- no control flow around a load in flight)
                __builtin_amdgcn_sched_barrier(0);
            }
            w1 = vmf_sample_dir(s0, s1, acc, wk, kap, mz);
            const float x21 = dot(gm2, w1);
            float p1 = 0;
#pragma unroll
            for (int k = 0; k < GK; ++k)
                if (k < n) {
                    const float4 q1 = q1n;
                    q1n = q1nn;
                    if (want_e1) {  // two lobes ahead: a lobe of this pass is ~50 instructions, an L2 round trip several hundred cycles
                        if (k + 2 < GK) q1nn = RL[6 + 2 * k];
                        __builtin_amdgcn_sched_barrier(0);  // (the load stays HERE, ahead of this lobe's arithmetic)
                    }
                    const float x1 = dot(V3{lr[k][0], lr[k][1], lr[k][2]}, w1);
                    p1 += la[k] * fast_exp(lkc[k] * ((lc1[k] * x1 + lc2[k] * x21) - 1));
                    if (want_e1) {
                        const float e = q1.y * fast_exp(q1.z * (x1 - 1));
                        num1 += e * q1.w;
                        den1 += e;
                    }
                }
            g1 = p1 * isum;
        }
    }

    // ---- the vertex again: Interaction / BSDF are rebuilt from the pool record rather than carried across the lobes (the same
    // function of the same inputs: same bits; `slot_b` is an opaque copy of the slot index so that the compiler does not
    // merge the two evaluations into ~40 registers that live through both passes over the lobes)
    int slot_b = slot;
    asm volatile("" : "+v"(slot_b));
    pool_load_rng<LY>(P, slot_b, sampler);
    nee.lightIndex = P.i(LY::PCQ, slot_b);
    nee.ul0 = P.f(LY::PCP + 0, slot_b);
    nee.ul1 = P.f(LY::PCP + 1, slot_b);
    nee.sfPDF = P.f(LY::PCP + 2, slot_b);
    st.ro = P.v3(LY::RO, slot_b);
    st.rd = P.v3(LY::RD, slot_b);
    const Vertex vxb = pool_load_vertex<true, G>(P, slot_b, P.u(LY::FLAGS, slot_b));
    vertex_setup<(G >= 2), false>(S, st, vxb, c);
    const V3 wo = -st.rd;
    V3 ctxp = c.intr.pi.mid();
    if (c.intr.is_surface && c.bsdf.has_lobes) ctxp = offset_ray_origin(c.intr.pi, c.intr.n, c.intr.wo);

    // ---- NEE, second half: shadow ray, transmittance, the estimate (sample_Ld :1190-1251); L += beta * Ld ----------------
    Spec L = P.sp3(LY::L, slot);
    if (nee.ok) {
        LightLi ls;
        (void)light_sample_li(light_quad_at(nee.lightIndex), ctxp, nee.ul0, nee.ul1, &ls);  // the same sample again: same bits
        const float p_l = (1.f / (float)S.n_lights) * ls.pdf;
        Spec f_hat;
        float sfp = nee.sfPDF;
        if (c.intr.is_surface) f_hat = bsdf_f(c.bsdf, c.intr.wo, ls.wi) * absdot(ls.wi, c.intr.n);
        else f_hat = sp(nee.sfPDF);
        if (useGuiding) sfp = ((1.0f - kGuidingProbability) * sfp) + (kGuidingProbability * g_nee);
        const float scatterPDF = 1.0f * sfp;
        const Spec r_p = pool_load_ru<LY, G>(P, slot);
        const Spec Ld = sample_Ld_shadow<Medium>(S, medium, c.intr, ch, ls, f_hat, p_l, scatterPDF, r_p, pc);
        L = L + pool_load_beta<LY, G>(P, slot) * Ld;
        pc.rec.add_scattered_direct_light(Ld);  // :485 / :838
    } else if (do_nee) {
        pc.rec.add_scattered_direct_light(sp(0.f));  // SampleLd ran and returned nothing: the hook is still called
    }
    if (dead) {
        *L_out = L;
        return false;
    }

    // ---- combine: Sample_p / Sample_f, MIS or RIS flavour (guiding.h:120-257, 404-530), as in li_vertex_guided_impl ------
    float inc0 = den0, inc1 = den1;
    float sf1 = 0;
    Spec f0 = sp(0.f), f1 = sp(0.f);
    if (!volume_vertex && ok0) f0 = c.bsdf.R * kInvPi;
    if (cand1) {
        if (volume_vertex) {
            sf1 = henyey_greenstein(dot(wo, w1), vg);
        } else {
            f1 = bsdf_f(c.bsdf, wo, w1);
            sf1 = bsdf_pdf(c.bsdf, wo, w1);
        }
    }
    V3 wi = mk(0, 0, 0);
    Spec f = sp(0.f);
    float pdf = 0, sfPdf = 0, misPdf = 0;
    bool have = false;
    int pick = 0;  // which candidate became wi (for the VSP of the next segment)
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
            pick = 1;
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
            const float sample1D = sumW * sampler.get1d();
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
            pick = idx;
        }
    }

    // ---- the path moves on (li_vertex_guided_impl's tail) ------------------------------------------------------------------
    Spec beta = pool_load_beta<LY, G>(P, slot);
    if (volume_vertex && rr_q > 0.f) beta = beta / (1 - rr_q);  // (:849; survivalProb < 1 <=> rr_q > 0)
    const Spec r_u = pool_load_ru<LY, G>(P, slot);
    bool cont = false;
    float rr_correction = P.f(LY::RRC, slot);
    V3 ro = mk(0, 0, 0);
    Spec r_l = sp(0.f);
    if (volume_vertex) {
        if (have && pdf != 0) {
            const float w = sfPdf / pdf;  // ps->p / ps->pdf
            beta = beta * w;
            r_l = r_u / pdf;
            ro = vx.p;
            // guiding_addVolumeData(..., phaseFunctionWeight, ps->wi, ps->pdf, ps->meanCosine, survivalProb) (:871)
            pc.rec.add_scatter_data(true, sp(w), wi, pdf, 1.0f - __builtin_fabsf(vg), survivalProb);
            cont = true;
        }
    } else if (have) {
        rr_correction *= pdf / sfPdf;
        const Spec bsdfWeight = f * absdot(wi, c.si.n) / pdf;
        beta = beta * bsdfWeight;
        r_l = r_u / misPdf;
        ro = offset_ray_origin(c.intr.pi, c.si.n, wi);
        if (nonzero(beta)) {
            if (depth > S.prm.minrrdepth) {  // :597-600 (etaScale is 1: no transmissive surfaces here)
                const Spec rrw = (beta / avg(r_u)) * rr_correction * 1.f;
                survivalProb = standard_throughput_rr(rrw);  // (specularBounce was just cleared)
            }
            cont = true;
            if (survivalProb < 1 && depth > S.prm.minrrdepth) {
                const float qq = fmax_(0.f, 1 - survivalProb);
                if (sampler.get1d() < qq) cont = false;
                else beta = beta / (1 - qq);
            }
            // guiding_addSurfaceData(..., bsdfWeight, bs->wi, bs->eta, bs->sampledRoughness, bs->pdf, survivalProb) (:608)
            if (cont) pc.rec.add_scatter_data(false, bsdfWeight, wi, pdf, 1.0f, survivalProb);
        }
    }
    if (!cont) {
        *L_out = L;
        return false;
    }
    // the next segment's VolumeScatterProbability(ray.d) of this vertex's distribution
    float vsp_next = -1.f;
    if (useScatterGuiding) {
        const float num = pick ? num1 : num0, den = pick ? den1 : den0;
        vsp_next = den > 0 ? num / den : -1.f;
    }
    // ---- what the vertex changed goes back to the pool (pool_store_full's fields, minus the ones it left alone) ----------
    P.set3(LY::RO, slot, ro);
    P.set3(LY::RD, slot, wi);
    P.sets(LY::L, slot, L);
    pool_store_beta<LY, G>(P, slot, beta);
    pool_store_rl<LY, G>(P, slot, r_l);
    if (volume_vertex) {
        P.set3(LY::PCP, slot, vx.p);
        P.i(LY::PCQ, slot) = -1;
    } else {
        P.set3(LY::PCP, slot, c.si.p);
        P.i(LY::PCQ, slot) = vx.quad;
    }
    pool_store_rng<LY>(P, slot, sampler);
    P.f(LY::GS, slot) = vsp_next;
    // depth / channel / ISG bits are unchanged by a vertex; specularBounce is cleared, lastVertexVolume set to the vertex kind
    uint32_t nfl = (fl & (FL_DEPTH_MASK | (3u << FL_CH_SHIFT) | FL_ISG_VALID | FL_ISG_SURF)) | FL_LIVE;
    if (volume_vertex) nfl |= FL_LASTVOL;
    P.u(LY::FLAGS, slot) = nfl;
    P.f(LY::RRC, slot) = rr_correction;
    (void)isg_out;
    return true;
}

VSPG_NS_END  // namespace vspg
