// vspg_wg_kernel.h -- workgroup-level wavefront scheduler for the path loop (gfx950).
//
// The per-lane persistent kernel (k_render_wave in vspg_capi.hip) keeps a path in one lane's
// registers from birth to death; measured on MI355X only ~44 % of its VALU lanes do useful work,
// because lanes of one wavefront sit at different kinds of vertices (volume / surface / primary
// segment with VSP-guided sampling / secondary segment / finished) and SIMT serialises the kinds.
//
// Here a 256-thread workgroup owns a POOL of NP paths whose state lives in LDS (SoA, 36 dwords per
// path), and advances all of them phase by phase.  Between phases the live paths are compacted into
// index lists with wave ballots + one LDS atomic per wavefront, so every phase runs over a dense
// list of paths that all need the SAME code:
//     new paths         camera ray + primary segment: intersect + VSP-guided distance sampling (coherent)
//     continuing paths  one secondary segment: intersect + plain delta tracking
//     volume vertices   NEE, Russian roulette, phase-function sampling
//     surface vertices  NEE, BSDF sampling, Russian roulette
// A path's registers exist only inside a phase; the SoA record of SURVEY 8d is what crosses phase
// boundaries -- through LDS instead of HBM.  Per path the operations and their order are exactly
// those of li_segment_a / li_segment_b, so results are bit-identical to the per-lane kernel.
#pragma once
#include "vspg_path.h"

namespace vspg {

// ---- LDS pool record (field-major SoA: field f of slot s at base[f * NP + s]) ----------------
enum {
    PF_RO = 0,        // 3
    PF_RD = 3,        // 3
    PF_L = 6,         // 3
    PF_BETA = 9,      // 3
    PF_RU = 12,       // 3
    PF_RL = 15,       // 3
    PF_PCP = 18,      // 3  previous light-sample context: raw point
    PF_PCQ = 21,      // 1  ... its rectangle (int), -1 = medium vertex (exact point, n = 0)
    PF_RNG = 22,      // 4  sampler PCG state / inc (2 x u64)
    PF_FLAGS = 26,    // 1  packed: depth, ch, bools, life cycle
    PF_RRC = 27,      // 1  rr_correction
    PF_PIXEL = 28,    // 1  pixel index (int)
    PF_SAMPLE = 29,   // 1  sample index (int)
    PF_VSP = 30,      // 1  isg.vsp_used (the primary VSP itself never leaves the primary phase)
    PF_VXG = 31,      // 1  vertex: volume: g; surface: rectangle index (int)
    PF_VXT = 32,      // 1  vertex: surface tHit
    PF_GS = 33,       // 4  guided builds: gs.vsp_next (1 used)
    PF_COUNT = 37,
    // The vertex position lives only from the segment phase to the vertex phase, the ray origin only
    // from the vertex phase to the next segment phase (the unguided vertex code never reads the old
    // origin): they share three dwords.  480 -> 512 paths fit the 80 KB a workgroup may use.
    PF_VXP = PF_RO,
    // guided builds: gbsdf.init queries the cache at ray.o + tHit * ray.d (guiding.h:85), so the origin must survive
    PF_VXP_GUIDED = 37,  // 3
    PF_COUNT_GUIDED = 40
};
template <bool GUIDED> struct PoolVx { static constexpr int kField = GUIDED ? (int)PF_VXP_GUIDED : (int)PF_VXP; };
enum {
    FL_DEPTH_MASK = 0xff,
    FL_CH_SHIFT = 8,            // 2 bits
    FL_SPECULAR = 1 << 10,
    FL_LASTVOL = 1 << 11,
    FL_ISG_VALID = 1 << 12,
    FL_ISG_SURF = 1 << 13,
    FL_GS_SCATTER = 1 << 14,
    FL_GS_FIELD = 1 << 15,
    FL_LIVE = 1 << 16,          // slot holds a path
    FL_RESTART = 1 << 17,       // slot's pixel has another sample to start
    FL_VX_VOLUME = 1 << 18,
    FL_DONE = 1 << 19,          // finished path parked for the film flush: PF_L = final radiance, PF_VSP = isg.vsp_used
};

struct Pool {
    float *base;
    int np;
    VDEV float &f(int field, int slot) const { return base[field * np + slot]; }
    VDEV int &i(int field, int slot) const { return reinterpret_cast<int *>(base)[field * np + slot]; }
    VDEV uint32_t &u(int field, int slot) const { return reinterpret_cast<uint32_t *>(base)[field * np + slot]; }
    VDEV V3 v3(int field, int slot) const { return V3{f(field, slot), f(field + 1, slot), f(field + 2, slot)}; }
    VDEV void set3(int field, int slot, V3 v) const { f(field, slot) = v.x; f(field + 1, slot) = v.y; f(field + 2, slot) = v.z; }
    VDEV Spec sp3(int field, int slot) const { return Spec{f(field, slot), f(field + 1, slot), f(field + 2, slot)}; }
    VDEV void sets(int field, int slot, Spec v) const { f(field, slot) = v.r; f(field + 1, slot) = v.g; f(field + 2, slot) = v.b; }
};

VDEV uint32_t pool_pack_flags(const PathState &st, int ch, const IsgSample &isg, uint32_t keep_flags) {
    uint32_t fl = keep_flags | (uint32_t)(st.depth & FL_DEPTH_MASK) | ((uint32_t)ch << FL_CH_SHIFT);
    if (st.specularBounce) fl |= FL_SPECULAR;
    if (st.lastVertexVolume) fl |= FL_LASTVOL;
    if (isg.valid) fl |= FL_ISG_VALID;
    if (isg.surface_event) fl |= FL_ISG_SURF;
    return fl;
}
VDEV void pool_store_rng(const Pool &P, int slot, const Sampler &sampler) {
    P.u(PF_RNG + 0, slot) = (uint32_t)sampler.rng.state;
    P.u(PF_RNG + 1, slot) = (uint32_t)(sampler.rng.state >> 32);
    P.u(PF_RNG + 2, slot) = (uint32_t)sampler.rng.inc;
    P.u(PF_RNG + 3, slot) = (uint32_t)(sampler.rng.inc >> 32);
}

// everything a path carries into its next segment (after the primary segment and after a vertex)
// GREY (grey medium, see HomogeneousMediumT): r_u and r_l are grey by construction -- one channel is parked
// and broadcast on load, which also tells the compiler that the three channels are one value
template <int GREY>
VDEV void pool_store_beta(const Pool &P, int slot, const PathState &st) {
    if constexpr (GREY >= 2) P.f(PF_BETA, slot) = st.beta.r;
    else P.sets(PF_BETA, slot, st.beta);
}
template <int GREY>
VDEV void pool_store_ru_rl(const Pool &P, int slot, const PathState &st) {
    if constexpr (GREY >= 1) {
        P.f(PF_RU, slot) = st.r_u.r;
        P.f(PF_RL, slot) = st.r_l.r;
    } else {
        P.sets(PF_RU, slot, st.r_u);
        P.sets(PF_RL, slot, st.r_l);
    }
}
template <bool GUIDED, int GREY = 0>
VDEV void pool_store_full(const Pool &P, int slot, const PathState &st, const Sampler &sampler, int ch, const IsgSample &isg,
                          uint32_t keep_flags) {
    P.set3(PF_RO, slot, st.ro);
    P.set3(PF_RD, slot, st.rd);
    P.sets(PF_L, slot, st.L);
    pool_store_beta<GREY>(P, slot, st);
    pool_store_ru_rl<GREY>(P, slot, st);
    P.set3(PF_PCP, slot, st.prevCtx.p);
    P.i(PF_PCQ, slot) = st.prevCtx.quad;
    pool_store_rng(P, slot, sampler);
    uint32_t fl = pool_pack_flags(st, ch, isg, keep_flags);
    if constexpr (GUIDED) {
        P.f(PF_GS, slot) = st.gs.vsp_next;
    }
    P.u(PF_FLAGS, slot) = fl;
    P.f(PF_RRC, slot) = st.rr_correction;
    P.f(PF_VSP, slot) = st.depth == 0 ? st.vsp0 : isg.vsp_used;
}

template <bool GUIDED = false>
VDEV void pool_store_vertex(const Pool &P, int slot, const Vertex &vx) {
    P.set3(PoolVx<GUIDED>::kField, slot, vx.p);
    if (vx.volume) P.f(PF_VXG, slot) = vx.g; else P.i(PF_VXG, slot) = vx.quad;
    P.f(PF_VXT, slot) = vx.t;
}
// after li_segment_a: only what that half changes (L, beta, r_u, r_l, sampler, depth / ISG flags) plus
// the vertex it stopped at; ray, previous context, rr_correction and guiding state are untouched
template <int GREY = 0, bool GUIDED = false>
VDEV void pool_store_a(const Pool &P, int slot, const PathState &st, const Sampler &sampler, int ch, const IsgSample &isg,
                       const Vertex &vx, uint32_t keep_flags) {
    P.sets(PF_L, slot, st.L);
    pool_store_beta<GREY>(P, slot, st);
    pool_store_ru_rl<GREY>(P, slot, st);
    pool_store_rng(P, slot, sampler);
    uint32_t fl = pool_pack_flags(st, ch, isg, keep_flags);
    if (vx.volume) fl |= FL_VX_VOLUME;
    P.u(PF_FLAGS, slot) = fl;
    P.f(PF_VSP, slot) = st.depth == 0 ? st.vsp0 : isg.vsp_used;
    pool_store_vertex<GUIDED>(P, slot, vx);
}
template <bool GUIDED = false>
VDEV Vertex pool_load_vertex(const Pool &P, int slot, uint32_t fl) {
    Vertex vx;
    vx.volume = (fl & FL_VX_VOLUME) != 0;
    vx.p = P.v3(PoolVx<GUIDED>::kField, slot);
    vx.g = vx.volume ? P.f(PF_VXG, slot) : 0.f;
    vx.quad = vx.volume ? -1 : P.i(PF_VXG, slot);
    __builtin_assume(vx.quad >= -1);
    vx.t = P.f(PF_VXT, slot);
    vx.perr = vx.volume ? mk(0, 0, 0) : ld3(quad_at(vx.quad).perr);
    return vx;
}

template <bool GUIDED, int GREY = 0>
VDEV uint32_t pool_load(const Pool &P, int slot, const DScene &S, PathState &st, Sampler &sampler, int *ch, IsgSample &isg) {
    st.ro = P.v3(PF_RO, slot);
    st.rd = P.v3(PF_RD, slot);
    st.L = P.sp3(PF_L, slot);
    st.beta = GREY >= 2 ? sp(P.f(PF_BETA, slot)) : P.sp3(PF_BETA, slot);
    if constexpr (GREY >= 1) {
        st.r_u = sp(P.f(PF_RU, slot));
        st.r_l = sp(P.f(PF_RL, slot));
    } else {
        st.r_u = P.sp3(PF_RU, slot);
        st.r_l = P.sp3(PF_RL, slot);
    }
    st.prevCtx.p = P.v3(PF_PCP, slot);
    st.prevCtx.quad = P.i(PF_PCQ, slot);
    st.prevCtx.perr = mk(0, 0, 0);  // (triangle scenes run the per-lane / wavefront kernels)
    __builtin_assume(st.prevCtx.quad >= -1);
    sampler.rng.state = (uint64_t)P.u(PF_RNG + 0, slot) | ((uint64_t)P.u(PF_RNG + 1, slot) << 32);
    sampler.rng.inc = (uint64_t)P.u(PF_RNG + 2, slot) | ((uint64_t)P.u(PF_RNG + 3, slot) << 32);
    const uint32_t fl = P.u(PF_FLAGS, slot);
    st.depth = (int)(fl & FL_DEPTH_MASK);
    *ch = (int)((fl >> FL_CH_SHIFT) & 3u);
    st.specularBounce = (fl & FL_SPECULAR) != 0;
    st.anyNonSpecularBounces = false;
    st.lastVertexVolume = (fl & FL_LASTVOL) != 0;
    isg.valid = (fl & FL_ISG_VALID) != 0;
    isg.surface_event = (fl & FL_ISG_SURF) != 0;
    st.rr_correction = P.f(PF_RRC, slot);
    st.etaScale = 1;
    const float v = P.f(PF_VSP, slot);
    st.vsp0 = v;
    isg.vsp_used = st.depth == 0 ? -1.f : v;
    if constexpr (GUIDED) {
        st.gs.vsp_next = P.f(PF_GS, slot);
        st.pce = 0.f;        // (guided RR is served by the per-lane kernels)
        st.guideRR = false;
    } else {
        st.gs.vsp_next = -1.f;
        st.pce = 0.f;
        st.guideRR = false;
    }
    return fl;
}

// append `slot` to an LDS list: one LDS atomic per wavefront (ballot + prefix count)
VDEV void list_push(bool pred, int slot, unsigned short *list, unsigned int *count) {
    const unsigned long long m = __ballot(pred);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    unsigned int base = 0;
    if (lane == leader) base = atomicAdd(count, (unsigned int)__popcll(m));
    base = __shfl(base, leader);
    if (pred) list[base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)slot;
}

// same, growing downwards from `last` (the two-ended lists keep both kinds dense in one array)
VDEV void list_push_back(bool pred, int slot, unsigned short *last, unsigned int *count) {
    const unsigned long long m = __ballot(pred);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    unsigned int base = 0;
    if (lane == leader) base = atomicAdd(count, (unsigned int)__popcll(m));
    base = __shfl(base, leader);
    if (pred) *(last - (int)(base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull)))) = (unsigned short)slot;
}

}  // namespace vspg
