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

VSPG_NS_BEGIN

// ---- LDS pool record (field-major SoA: field f of slot s at base[f * NP + s]) ----------------
// The record is laid out per kernel instantiation (round 3): a grey medium parks ONE channel of r_u / r_l (GREY >= 1) and, with
// grey surfaces, of beta (GREY >= 2) -- see HomogeneousMediumT -- and the fields those instantiations do not store take no
// room, so more paths fit the LDS a workgroup may use (the pool's size is what the phases' list lengths, i.e. the wavefronts'
// occupancy with work, hang on: 320 -> 384 paths took the guided kernel from 2.14 to 1.86 ms per wave).
//   unguided: 31 dwords generic, 27 with a grey medium, 25 with grey surfaces too;  guided: 37 / 33 / 31.
template <bool GUIDED, int GREY, bool TRAIN = false, bool FULL = false>
struct PoolLayout {
    static constexpr int RO = 0;                                  // 3
    static constexpr int RD = 3;                                  // 3
    static constexpr int L = 6;                                   // 3
    static constexpr int BETA = 9;                                // 3 (1 with grey surfaces)
    static constexpr int RU = BETA + (GREY >= 2 ? 1 : 3);         // 3 (1 in a grey medium)
    static constexpr int RL = RU + (GREY >= 1 ? 1 : 3);           // 3 (1)
    static constexpr int PCP = RL + (GREY >= 1 ? 1 : 3);          // 3  previous light-sample context: raw point
    static constexpr int PCQ = PCP + 3;                           // 1  ... its rectangle (int), -1 = medium vertex (exact point, n = 0)
    static constexpr int RNG = PCQ + 1;                           // 4  sampler PCG state / inc (2 x u64)
    static constexpr int FLAGS = RNG + 4;                         // 1  packed: depth, ch, bools, life cycle
    // (round 5: two fields only guided builds read take no room elsewhere -- k_render_wave_wg3's chunks stay full by the pool's
    //  slack over the workgroup's lanes, and at 25 dwords the headline instantiation holds 704 paths instead of 640)
    static constexpr int RRC = FLAGS + 1;                         // 1  rr_correction: guided builds only (without the guiding cache it is
                                                                  //    1 for the path's whole life: bs->pdf / bs->bsdfPdf of vertex_tail is x / x)
    static constexpr int PIXEL = RRC + (GUIDED ? 1 : 0);          // 1  pixel (x | y << 16)
    static constexpr int SAMPLE = PIXEL + 1;                      // 1  sample index (int)
    static constexpr int VSP = SAMPLE + 1;                        // 1  isg.vsp_used (the primary VSP itself never leaves the primary phase)
    static constexpr int VXG = VSP + 1;                           // 1  vertex: volume: g; surface: rectangle index (int)
    static constexpr int VXT = VXG + 1;                           // 1  vertex: surface tHit: guided builds only (gbsdf.init's p = ray.o + tHit * ray.d)
    static constexpr int GS = VXT + (GUIDED ? 1 : 0);             // 1  guided builds: gs.vsp_next
    // The vertex position lives only from the segment phase to the vertex phase, the ray origin only from the vertex phase to
    // the next segment phase (the unguided vertex code never reads the old origin): they share three dwords.  Guided builds:
    // gbsdf.init queries the cache at ray.o + tHit * ray.d (guiding.h:85), so the origin must survive.
    static constexpr int VXP = GUIDED ? GS + 1 : RO;              // 3
    // (appended LAST: every other field sits at the same offset with and without it, so the helpers below need not know)
    // training launches (a18): the recorder's state between phases, packed (pool_store_rec); the records themselves go straight to HBM
    static constexpr int REC = GUIDED ? VXP + 3 : GS;             // 1 (pool_store_rec)
    // full scenes (round 4: triangles, spheres, infinite lights, medium boundaries on the workgroup kernel): a path may cross medium
    // boundaries at depth 0, so isg.vsp_used has to be kept BESIDE the pixel's primary VSP (the VSP slot), not instead of it
    static constexpr int VSPU = REC + (TRAIN ? 1 : 0);            // 1 (FULL only)
    static constexpr int COUNT = VSPU + (FULL ? 1 : 0);
};
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
    FL_INMED = 1 << 24,         // full scenes: ray.medium != nullptr (same bit as the pipeline's WFL_INMED)
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

// The recorder's state between phases in ONE dword: records written (<= 64), whether a current record exists (it is always the
// last one: new_segment sets cur = n++, and NextSegment() == nullptr leaves none), its six flag bits.  The accumulated scattered
// direct light is not parked: a record sees at most one NEE, inside one vertex phase, and new_segment zeroes the sum.
template <class LY>
VDEV void pool_store_rec(const Pool &P, int slot, const PathRecorder &rec) {
    P.u(LY::REC, slot) = (uint32_t)rec.n | (rec.cur >= 0 ? 128u : 0u) | (rec.cur_flags << 8);
}
template <class LY>
VDEV void pool_load_rec(const Pool &P, int slot, PathRecorder &rec) {
    const uint32_t w = P.u(LY::REC, slot);
    rec.n = (int)(w & 127u);
    rec.cur = (w & 128u) ? rec.n - 1 : -1;
    rec.cur_flags = w >> 8;
    rec.scat_r = rec.scat_g = rec.scat_b = 0.f;
}
VDEV void pool_store_rec(const Pool &, int, const NullRecorder &) {}
VDEV void pool_load_rec(const Pool &, int, NullRecorder &) {}
VDEV bool fl_in_medium(uint32_t fl) { return (fl & FL_INMED) != 0; }
VDEV uint32_t pool_pack_flags(const PathState &st, int ch, const IsgSample &isg, uint32_t keep_flags) {
    uint32_t fl = keep_flags | (uint32_t)(st.depth & FL_DEPTH_MASK) | ((uint32_t)ch << FL_CH_SHIFT);
    if (st.specularBounce) fl |= FL_SPECULAR;
    if (st.lastVertexVolume) fl |= FL_LASTVOL;
    if (isg.valid) fl |= FL_ISG_VALID;
    if (isg.surface_event) fl |= FL_ISG_SURF;
    return fl;
}
template <class LY>
VDEV void pool_store_rng(const Pool &P, int slot, const Sampler &sampler) {
    P.u(LY::RNG + 0, slot) = (uint32_t)sampler.rng.state;
    P.u(LY::RNG + 1, slot) = (uint32_t)(sampler.rng.state >> 32);
    P.u(LY::RNG + 2, slot) = (uint32_t)sampler.rng.inc;
    P.u(LY::RNG + 3, slot) = (uint32_t)(sampler.rng.inc >> 32);
}
template <class LY>
VDEV void pool_load_rng(const Pool &P, int slot, Sampler &sampler) {
    sampler.rng.state = (uint64_t)P.u(LY::RNG + 0, slot) | ((uint64_t)P.u(LY::RNG + 1, slot) << 32);
    sampler.rng.inc = (uint64_t)P.u(LY::RNG + 2, slot) | ((uint64_t)P.u(LY::RNG + 3, slot) << 32);
}

// everything a path carries into its next segment (after the primary segment and after a vertex)
// GREY (grey medium, see HomogeneousMediumT): r_u and r_l are grey by construction -- one channel is parked
// and broadcast on load, which also tells the compiler that the three channels are one value
template <class LY, int GREY>
VDEV void pool_store_beta(const Pool &P, int slot, Spec beta) {
    if constexpr (GREY >= 2) P.f(LY::BETA, slot) = beta.r;
    else P.sets(LY::BETA, slot, beta);
}
template <class LY, int GREY>
VDEV Spec pool_load_beta(const Pool &P, int slot) { return GREY >= 2 ? sp(P.f(LY::BETA, slot)) : P.sp3(LY::BETA, slot); }
template <class LY, int GREY>
VDEV Spec pool_load_ru(const Pool &P, int slot) { return GREY >= 1 ? sp(P.f(LY::RU, slot)) : P.sp3(LY::RU, slot); }
template <class LY, int GREY>
VDEV void pool_store_rl(const Pool &P, int slot, Spec r_l) {
    if constexpr (GREY >= 1) P.f(LY::RL, slot) = r_l.r;
    else P.sets(LY::RL, slot, r_l);
}
template <class LY, int GREY>
VDEV void pool_store_ru_rl(const Pool &P, int slot, const PathState &st) {
    if constexpr (GREY >= 1) {
        P.f(LY::RU, slot) = st.r_u.r;
        P.f(LY::RL, slot) = st.r_l.r;
    } else {
        P.sets(LY::RU, slot, st.r_u);
        P.sets(LY::RL, slot, st.r_l);
    }
}
template <bool GUIDED, int GREY = 0, bool FULL = false>
VDEV void pool_store_full(const Pool &P, int slot, const PathState &st, const Sampler &sampler, int ch, const IsgSample &isg,
                          uint32_t keep_flags) {
    using LY = PoolLayout<GUIDED, GREY>;
    if constexpr (FULL) {
        keep_flags = (keep_flags & ~(uint32_t)FL_INMED) | (st.in_medium ? (uint32_t)FL_INMED : 0u);
        P.f(PoolLayout<GUIDED, GREY, false, true>::VSPU, slot) = isg.vsp_used;
    }
    P.set3(LY::RO, slot, st.ro);
    P.set3(LY::RD, slot, st.rd);
    P.sets(LY::L, slot, st.L);
    pool_store_beta<LY, GREY>(P, slot, st.beta);
    pool_store_ru_rl<LY, GREY>(P, slot, st);
    P.set3(LY::PCP, slot, st.prevCtx.p);
    P.i(LY::PCQ, slot) = st.prevCtx.quad;
    pool_store_rng<LY>(P, slot, sampler);
    uint32_t fl = pool_pack_flags(st, ch, isg, keep_flags);
    if constexpr (GUIDED) {
        P.f(LY::GS, slot) = st.gs.vsp_next;
    }
    P.u(LY::FLAGS, slot) = fl;
    if constexpr (GUIDED) P.f(LY::RRC, slot) = st.rr_correction;
    P.f(LY::VSP, slot) = st.depth == 0 ? st.vsp0 : isg.vsp_used;
}

template <bool GUIDED = false, int GREY = 0>
VDEV void pool_store_vertex(const Pool &P, int slot, const Vertex &vx) {
    using LY = PoolLayout<GUIDED, GREY>;
    P.set3(LY::VXP, slot, vx.p);
    if (vx.volume) P.f(LY::VXG, slot) = vx.g; else P.i(LY::VXG, slot) = vx.quad;
    if constexpr (GUIDED) P.f(LY::VXT, slot) = vx.t;
}
// after li_segment_a: only what that half changes (L, beta, r_u, r_l, sampler, depth / ISG flags) plus
// the vertex it stopped at; ray, previous context, rr_correction and guiding state are untouched
template <int GREY = 0, bool GUIDED = false, bool FULL = false>
VDEV void pool_store_a(const Pool &P, int slot, const PathState &st, const Sampler &sampler, int ch, const IsgSample &isg,
                       const Vertex &vx, uint32_t keep_flags) {
    using LY = PoolLayout<GUIDED, GREY>;
    if constexpr (FULL) P.f(PoolLayout<GUIDED, GREY, false, true>::VSPU, slot) = isg.vsp_used;  // (in_medium does not change inside a segment that reaches a vertex)
    P.sets(LY::L, slot, st.L);
    pool_store_beta<LY, GREY>(P, slot, st.beta);
    pool_store_ru_rl<LY, GREY>(P, slot, st);
    pool_store_rng<LY>(P, slot, sampler);
    uint32_t fl = pool_pack_flags(st, ch, isg, keep_flags);
    if (vx.volume) fl |= FL_VX_VOLUME;
    P.u(LY::FLAGS, slot) = fl;
    P.f(LY::VSP, slot) = st.depth == 0 ? st.vsp0 : isg.vsp_used;
    pool_store_vertex<GUIDED, GREY>(P, slot, vx);
}
template <bool GUIDED = false, int GREY = 0, bool FULL = false>
VDEV Vertex pool_load_vertex(const Pool &P, int slot, uint32_t fl) {
    using LY = PoolLayout<GUIDED, GREY>;
    Vertex vx;
    vx.volume = (fl & FL_VX_VOLUME) != 0;
    vx.p = P.v3(LY::VXP, slot);
    vx.g = vx.volume ? P.f(LY::VXG, slot) : 0.f;
    vx.quad = vx.volume ? -1 : P.i(LY::VXG, slot);
    if constexpr (!FULL) __builtin_assume(vx.quad >= -1);
    vx.t = GUIDED ? P.f(LY::VXT, slot) : 0.f;  // (unguided vertex code never reads the hit distance)
    return vx;
}

template <bool GUIDED, int GREY = 0, bool FULL = false>
VDEV uint32_t pool_load(const Pool &P, int slot, const DScene &S, PathState &st, Sampler &sampler, int *ch, IsgSample &isg) {
    using LY = PoolLayout<GUIDED, GREY>;
    st.ro = P.v3(LY::RO, slot);
    st.rd = P.v3(LY::RD, slot);
    st.L = P.sp3(LY::L, slot);
    st.beta = GREY >= 2 ? sp(P.f(LY::BETA, slot)) : P.sp3(LY::BETA, slot);
    if constexpr (GREY >= 1) {
        st.r_u = sp(P.f(LY::RU, slot));
        st.r_l = sp(P.f(LY::RL, slot));
    } else {
        st.r_u = P.sp3(LY::RU, slot);
        st.r_l = P.sp3(LY::RL, slot);
    }
    st.prevCtx.p = P.v3(LY::PCP, slot);
    st.prevCtx.quad = P.i(LY::PCQ, slot);
    if constexpr (!FULL) __builtin_assume(st.prevCtx.quad >= -1);
    sampler.rng.state = (uint64_t)P.u(LY::RNG + 0, slot) | ((uint64_t)P.u(LY::RNG + 1, slot) << 32);
    sampler.rng.inc = (uint64_t)P.u(LY::RNG + 2, slot) | ((uint64_t)P.u(LY::RNG + 3, slot) << 32);
    const uint32_t fl = P.u(LY::FLAGS, slot);
    st.depth = (int)(fl & FL_DEPTH_MASK);
    *ch = (int)((fl >> FL_CH_SHIFT) & 3u);
    st.specularBounce = (fl & FL_SPECULAR) != 0;
    st.anyNonSpecularBounces = false;
    st.lastVertexVolume = (fl & FL_LASTVOL) != 0;
    isg.valid = (fl & FL_ISG_VALID) != 0;
    isg.surface_event = (fl & FL_ISG_SURF) != 0;
    st.rr_correction = GUIDED ? P.f(LY::RRC, slot) : 1.f;
    st.etaScale = 1;
    const float v = P.f(LY::VSP, slot);
    st.vsp0 = v;
    isg.vsp_used = st.depth == 0 ? -1.f : v;
    if constexpr (FULL) {  // (PoolLayout::VSPU: appended behind every other field, whatever TRAIN says -- unguided full-scene pools do not train)
        isg.vsp_used = P.f(PoolLayout<GUIDED, GREY, false, true>::VSPU, slot);
        st.in_medium = (fl_in_medium(P.u(LY::FLAGS, slot)));
    }
    if constexpr (GUIDED) {
        st.gs.vsp_next = P.f(LY::GS, slot);
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

VSPG_NS_END  // namespace vspg
