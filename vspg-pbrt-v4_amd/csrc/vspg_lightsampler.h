// vspg_lightsampler.h -- PowerLightSampler / BVHLightSampler (src/pbrt/lightsamplers.h:63-98, 100-430; round 3).
//
// "bvh" is the reference's default light sampler (guidedvolpathvspgintegrator.cpp:1318).  The tree is built once per renderer on
// the host (vspg_capi.hip: lsb::build, the reference's buildBVH restated) over the light list {emissive rectangles in rectangle
// order, then the infinite lights}; a node record keeps what CompactLightBounds' accessors return -- the box, the cosines and the
// axis after their round trip through the 16-bit / 15-bit / octahedral quantisation -- so the device does the float arithmetic
// of CompactLightBounds::Importance on them and nothing else.  Scenes with a single light (every sampler picks it with pmf 1) and
// the uniform sampler keep the two-line pick of rounds 1-2, and only the full-scene kernels (per-lane, wavefront pipeline)
// carry this code: the workgroup kernels serve the configurations whose pick is trivial.
#pragma once
#include "vspg_device.h"

VSPG_NS_BEGIN

// lightSampler.Sample(ctx, u) where the pick is not uniform: p = ctx.p(), n = ctx.ns (0 at a medium vertex)
VDEV float light_importance(const DLightNode &nd, V3 p, V3 n) {  // CompactLightBounds::Importance (lightsamplers.h:144-206)
    const V3 bmin = ld3(nd.bmin), bmax = ld3(nd.bmax);
    const V3 pc = (bmin + bmax) * 0.5f;
    float d2 = len2(p - pc);
    d2 = fmax_(d2, len(bmax - bmin) / 2);
    const V3 wi = normalize(p - pc);
    float cosTheta_w = dot(ld3(nd.w), wi);
    if (nd.twoSided) cosTheta_w = __builtin_fabsf(cosTheta_w);
    const float sinTheta_w = safe_sqrt(1 - cosTheta_w * cosTheta_w);
    float cosTheta_b;  // BoundSubtendedDirections(bounds, p).cosTheta (vecmath.h:1815-1828)
    {
        const bool inside = pc.x >= bmin.x && pc.x <= bmax.x && pc.y >= bmin.y && pc.y <= bmax.y && pc.z >= bmin.z && pc.z <= bmax.z;
        const float radius = inside ? len(pc - bmax) : 0.f;
        if (len2(p - pc) < radius * radius) cosTheta_b = -1.f;
        else {
            const float sin2ThetaMax = (radius * radius) / len2(pc - p);
            cosTheta_b = safe_sqrt(1 - sin2ThetaMax);
        }
    }
    const float sinTheta_b = safe_sqrt(1 - cosTheta_b * cosTheta_b);
    const float cosTheta_o = nd.cosTheta_o, cosTheta_e = nd.cosTheta_e;
    const float sinTheta_o = safe_sqrt(1 - cosTheta_o * cosTheta_o);
    const float cosTheta_x = cosTheta_w > cosTheta_o ? 1.f : cosTheta_w * cosTheta_o + sinTheta_w * sinTheta_o;  // cosSubClamped
    const float sinTheta_x = cosTheta_w > cosTheta_o ? 0.f : sinTheta_w * cosTheta_o - cosTheta_w * sinTheta_o;  // sinSubClamped
    const float cosThetap = cosTheta_x > cosTheta_b ? 1.f : cosTheta_x * cosTheta_b + sinTheta_x * sinTheta_b;
    if (cosThetap <= cosTheta_e) return 0;
    float importance = nd.phi * cosThetap / d2;
    if (!(n.x == 0 && n.y == 0 && n.z == 0)) {
        const float cosTheta_i = absdot(wi, n);
        const float sinTheta_i = safe_sqrt(1 - cosTheta_i * cosTheta_i);
        const float cosThetap_i = cosTheta_i > cosTheta_b ? 1.f : cosTheta_i * cosTheta_b + sinTheta_i * sinTheta_b;
        importance *= cosThetap_i;
    }
    return fmax_(importance, 0.f);
}
// returns false for "no light"
VDEV bool light_sampler_sample(const DScene &S, V3 p, V3 n, float u, int *lightIndex, float *pmf) {
    const DLightSampler &ls = S.lsamp;
    const int n_all = S.n_lights + S.n_inf;
    if (ls.mode == VSPG_LIGHTSAMPLER_UNIFORM) {  // lightsamplers.h:33-38
        if (n_all == 0) return false;
        const int li = (int)(u * (float)n_all);
        *lightIndex = li < n_all - 1 ? li : n_all - 1;
        *pmf = 1.f / (float)n_all;
        return true;
    }
    if (ls.mode == VSPG_LIGHTSAMPLER_POWER) {  // AliasTable::Sample (util/sampling.cpp:620-646)
        if (!ls.n_alias) return false;
        int offset = (int)(u * (float)ls.n_alias);
        if (offset > ls.n_alias - 1) offset = ls.n_alias - 1;
        const float up = fmin_(u * (float)ls.n_alias - (float)offset, kOneMinusEps);
        if (up < ls.alias_q[offset]) {
            *lightIndex = offset;
            *pmf = ls.alias_p[offset];
        } else {
            *lightIndex = ls.alias_i[offset];
            *pmf = ls.alias_p[ls.alias_i[offset]];
        }
        return true;
    }
    // BVHLightSampler::Sample (lightsamplers.h:283-342)
    const float pInfinite = (float)ls.n_inf / (float)(ls.n_inf + (ls.n_nodes == 0 ? 0 : 1));
    if (u < pInfinite) {
        u /= pInfinite;
        int index = (int)(u * (float)ls.n_inf);
        if (index > ls.n_inf - 1) index = ls.n_inf - 1;
        *pmf = pInfinite / (float)ls.n_inf;
        *lightIndex = ls.inf_light[index];
        return true;
    }
    if (ls.n_nodes == 0) return false;
    u = fmin_((u - pInfinite) / (1 - pInfinite), kOneMinusEps);
    int nodeIndex = 0;
    float pm = 1 - pInfinite;
    for (int guard = 0; guard < 64; ++guard) {
        const DLightNode &node = ls.nodes[nodeIndex];
        if (!node.is_leaf) {
            const float c0 = light_importance(ls.nodes[nodeIndex + 1], p, n), c1 = light_importance(ls.nodes[node.child_or_light], p, n);
            if (c0 == 0 && c1 == 0) return false;
            // SampleDiscrete({c0, c1}, u, &nodePMF, &u) (util/sampling.h:79-113)
            float sumW = 0;
            sumW += c0;
            sumW += c1;
            float up = u * sumW;
            if (up == sumW) up = next_float_down(up);
            int offset = 0;
            float sum = 0;
            if (sum + c0 <= up) {
                sum += c0;
                offset = 1;
            }
            const float cw = offset ? c1 : c0;
            pm *= cw / sumW;
            u = fmin_((up - sum) / cw, kOneMinusEps);
            nodeIndex = offset == 0 ? nodeIndex + 1 : (int)node.child_or_light;
        } else {
            if (nodeIndex > 0 || light_importance(node, p, n) > 0) {
                *lightIndex = (int)node.child_or_light;
                *pmf = pm;
                return true;
            }
            return false;
        }
    }
    return false;
}
// lightSampler.PMF(ctx, light)
VDEV float light_sampler_pmf(const DScene &S, V3 p, V3 n, int lightIndex) {
    const DLightSampler &ls = S.lsamp;
    const int n_all = S.n_lights + S.n_inf;
    if (ls.mode == VSPG_LIGHTSAMPLER_UNIFORM) return n_all ? 1.f / (float)n_all : 0.f;
    if (ls.mode == VSPG_LIGHTSAMPLER_POWER) return ls.n_alias ? ls.alias_p[lightIndex] : 0.f;
    // BVHLightSampler::PMF (lightsamplers.h:344-381)
    uint32_t bitTrail = ls.bit_trail[lightIndex];
    if (bitTrail == 0xffffffffu) return 1.f / (float)(ls.n_inf + (ls.n_nodes == 0 ? 0 : 1));
    const float pInfinite = (float)ls.n_inf / (float)(ls.n_inf + (ls.n_nodes == 0 ? 0 : 1));
    float pm = 1 - pInfinite;
    int nodeIndex = 0;
    for (int guard = 0; guard < 64; ++guard) {
        const DLightNode &node = ls.nodes[nodeIndex];
        if (node.is_leaf) return pm;
        const float c0 = light_importance(ls.nodes[nodeIndex + 1], p, n), c1 = light_importance(ls.nodes[node.child_or_light], p, n);
        pm *= ((bitTrail & 1u) ? c1 : c0) / (c0 + c1);
        nodeIndex = (bitTrail & 1u) ? (int)node.child_or_light : nodeIndex + 1;
        bitTrail >>= 1;
    }
    return pm;
}

VSPG_NS_END  // namespace vspg
