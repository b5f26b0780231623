// vspg_wavefront.h -- multi-kernel wavefront pipeline for heterogeneous media (gfx950).
//
// A tracking walk through a grid medium visits every tentative collision of its ray, and the rays of a frame have wildly
// different collision counts (0 for a ray that misses the cloud, tens through its core).  Inside one kernel that keeps a
// path in a lane (k_render_wave) or a segment in a lane (k_render_wave_wg) a wavefront lasts as long as its longest
// walk: 18 % of the vector lanes did useful work on the 256^3 cloud (round 1, profiles/r01_pmc_wave_state_cloud.txt).
//
// Here the frame's paths live in HBM as a field-major SoA (one column per pixel of the wave: this part has 288 GB, the
// ~600 B per path of a 4K wave are 5 GB) and every path-loop iteration (:309-609) is cut at its two walks:
//
//   k_wf_start       dense   camera rays + the first segment's begin (once per pass)
//   k_wf_dist_walk   WALK    SampleT_maj_Resampling's traversal + the reservoir callback (:691-719) -- one JOB per path
//   k_wf_vertex      dense   [the previous vertex's NEE result, if a shadow walk was out for it] candidate selection
//                            (:721-771), surface emission / depth test (:350-412), the vertex -- NEE light sample and
//                            shadow-ray set-up (:1136-1200), Russian roulette, new direction (:842-874 / :487-606) -- and the
//                            NEXT segment's begin (scene intersection, RNG seeding :323-325, the resampling routine's majorant
//                            pre-pass, media_sampleTMaj.h:150-176); survivors are compacted into the next list with a wave
//                            ballot + prefix count + one atomic per workgroup per four rounds
//   k_wf_shadow_walk WALK    the shadow ray's ratio-tracking transmittance (:1205-1232) -- one JOB per unoccluded NEE
//   (guided pipelines: the next segment's begin is its own dense kernel, k_wf_begin -- the guided vertex fills the register file)
//
// Nothing at a vertex after the NEE's set-up depends on the shadow ray's transmittance -- it only scales what is added to L --
// so the vertex does not wait for it (round 3): k_wf_vertex goes on to the next segment, the shadow walk of iteration i runs
// CONCURRENTLY with the distance walk of iteration i + 1 (own stream, own job fields), and its result is added first thing in
// the next k_wf_vertex, before anything else touches L (the additions to L keep the reference's order).  A path that ends at
// the vertex with a shadow walk still out stays on the list as WFL_DEAD for that one addition.  An NEE that needs no walk is
// added on the spot.  Per iteration the path record crosses HBM once (it used to be twice: k_wf_seg_end, then k_wf_advance).
//
// The WALK kernels are persistent: a lane runs one job at a time and a wavefront refills its idle lanes from the job
// list once a quarter of them are idle (one returning atomic per 64 jobs), so the lanes of a wavefront are at DIFFERENT
// jobs but always in the SAME loop -- one tracking step per iteration: majorant-cell advance, then the tentative-collision
// draw, then the collision itself (8-voxel fetch + callback) for the lanes that drew one.  Path state crosses a kernel
// boundary once per walk, as 16-byte-per-lane group accesses.
//
// Per path the operations and their order are those of li_segment_a / li_segment_b / sample_distance / sample_Ld /
// sample_T_maj*: the film is bit-identical to the per-lane kernel's (and to the oracle's paths).
#pragma once
#include "vspg_path.h"
#include "vspg_wg_kernel.h"  // flag bits, list_push

VSPG_NS_BEGIN

// ---- path record: groups of kWfGroup floats, group-major: field f of slot s at base[((f / kWfGroup) * n + s) * kWfGroup + f % kWfGroup] --
// A 3-vector / spectrum / RNG state sits inside ONE 16-byte quad, so a lane moves it with one dwordx3 / dwordx4 access.  With
// groups of one quad a wavefront over consecutive slots touches whole cache lines; wider groups keep a lane's quads of a group in
// one line however thin the lists get, but coalesce consecutive slots worse.  Measured (same box, ms per wave, groups of
// 4 / 8 / 16 / 32 floats): cloud 9.9 / 10.1 / 11.0 / 11.4; NanoVDB semantics 14.4 / 14.5 / 15.1 / 15.4; guided cloud 11.65 / 11.1 /
// 11.1 / 11.4.  Quads stay (a width chosen at run time per pipeline costs far more in address arithmetic than it returns:
// 12.6 / 16.6 ms -- the field offsets must fold at compile time).
#ifndef VSPG_WF_GROUP
#define VSPG_WF_GROUP 4
#endif
constexpr int kWfGroup = VSPG_WF_GROUP;
enum {
    WF_RO = 0,        // 3  ray origin            | +3: WF_VSP
    WF_VSP = 3,       // 1  isg.vsp_used
    WF_RD = 4,        // 3  ray direction         | +3: WF_RRC
    WF_RRC = 7,       // 1  rr_correction
    WF_L = 8,         // 3                        | +3: WF_FLAGS
    WF_FLAGS = 11,    // 1  packed like the LDS pool's (vspg_wg_kernel.h) + WFL_* bits
    WF_BETA = 12,     // 3                        | +3: WF_SURV
    WF_SURV = 15,     // 1  (was: survivalProb of the vertex) now WF_VRC: vsp / (1 - exp(-tau)) of the segment in flight (media_sampleTMaj.h:172)
    WF_RU = 16,       // 3                        | +3: WF_PL
    WF_PL = 19,       // 1  NEE: p_l
    WF_RL = 20,       // 3                        | +3: WF_SPDF
    WF_SPDF = 23,     // 1  NEE: scatterPDF
    WF_PCP = 24,      // 3  previous light-sample context: raw point | +3: WF_PCQ
    WF_PCQ = 27,      // 1  ... its rectangle, -1 = medium vertex
    WF_RNG = 28,      // 4  sampler PCG state / inc
    WF_VXP = 32,      // 3  vertex position (surface: re-projected hit point; volume: selected candidate) | +3: WF_VXG
    WF_VXG = 35,      // 1  surface: rectangle (int); volume: g
    // distance-walk job
    WF_IT = 36,       // 9 (+3 below)  DDA iterator at the start of the ray: tMin, tMax, nextCrossingT[3], deltaT[3], packed voxel | step signs
    WF_VXT = 45,      // 1  surface tHit
    WF_WU = 46,       // 1  first uniform of the walk
    WF_MSCALE = 47,   // 1  majorantScale
    WF_RDN = 48,      // 3  normalised direction  | +3: WF_VRC
    WF_VRC = 15,      // 1  in the throughput's quad, which the vertex kernel reads anyway (the walk has no use for it)
    WF_WRNG = 52,     // 4  the walk's private PCG (:323-325 / :1193)
    // distance-walk result
    WF_TMAJ = 56,     // 3                        | +3: WF_WSUM
    WF_WSUM = 59,     // 1  weightSum
    WF_TRR = 60,      // 3  trRatioEst            | +3: WF_SELW
    WF_SELW = 63,     // 1  selected candidate's weight (0: none selected)
    WF_BRS = 64,      // 3  beta_rs               | +3: WF_VSPG
    WF_VSPG = 67,     // (unused: nothing downstream of the segment's begin reads the VSP itself, only WF_VRC)
    WF_RURS = 68,     // 3  r_u_rs
    WF_SELP = 72,     // 3  selected candidate: position
    WF_SELNUM = 76,   // 3
    WF_SELDEN = 80,   // 3
    // shadow job / result
    WF_SLO = 84,      // 3  shadow-ray origin
    WF_FHAT = 88,     // 3  f_hat
    WF_LSL = 92,      // 3  ls.L
    WF_TRAY = 96,     // 3  shadow result: T_ray
    WF_SRL = 100,     // 3  ... r_l
    WF_SRU = 104,     // 3  ... r_u
    WF_STMAJ = 108,   // 3  ... residual T_maj
    // (112..119: round 3 kept the per-hit error bounds of triangle hits here; a hit is (prim, three floats) now, Isect)
    // guided pipeline: the whole vertex (cache init, NEE set-up, Russian roulette, new direction) runs in k_wf_seg_end
    WF_BNEE = 120,    // 3  the throughput the NEE saw (the vertex code has since moved st.beta on) | +3: WF_GSVSP
    WF_GSVSP = 123,   // 1  VolumeScatterProbability(ray.d) of the vertex for the next segment (gs.vsp_next)
    // training passes: the path recorder's state (vspg_train.h: PathRecorder) between kernels
    WF_RECN = 124,    // 1  records written so far
    WF_RECCUR = 125,  // 1  current record (-1: none)
    WF_RECFL = 126,   // 1  its flags word
    // shadow-walk job (own fields: the shadow walk of iteration i runs beside the distance walk of iteration i + 1)
    WF_SIT = 128,     // 8  DDA iterator of the shadow ray (floats; its packed word: WF_SITP, in the direction's quad)
    WF_SWU = 137,     // (unused: the shadow walk's first uniform sits in the origin's quad, WF_SWU2)
    WF_SRDN = 140,    // 3  normalised shadow-ray direction
    WF_SWRNG = 144,   // 4  the shadow ray's private PCG (:1193)
    WF_COUNT = 148
};
constexpr int kWfPoolFloats = (WF_COUNT + kWfGroup - 1) / kWfGroup * kWfGroup;  // floats per path in the pool (whole groups)
enum {  // bits above the LDS pool's FL_* (which end at 1 << 19)
    WFL_NODIST = 1 << 20,    // no SampleDistance this segment (no medium / the ray escapes)
    WFL_NOWALK = 1 << 21,    // the resampling routine returned before its traversal (tau == 0)
    WFL_GUIDE = 1 << 22,     // the segment is VSP-guided
    WFL_SHADOW_WALK = 1 << 23,   // NEE of the previous vertex: a ratio-tracking walk is out; its result is added by the next k_wf_vertex
    WFL_INMED = 1 << 24,     // ray.medium != nullptr (only read in scenes with medium boundaries, DScene::has_boundaries)
    WFL_HIT = 1 << 25,       // the segment's ray hit a surface
    WFL_DELTA = 1 << 27,     // that NEE sampled a delta light (DistantLight)
    WFL_DEAD = 1 << 28,      // the path ended at the previous vertex; only that NEE result is still to be added
};

struct WfPool {
    float *base;
    size_t n;
    VDEV size_t at(int field, unsigned slot) const {
#ifdef VSPG_WF_DEBUG
        if (slot >= n) { atomicOr(&g_dbg_err[0], 1u); atomicMax(&g_dbg_err[1], slot); slot = 0; }
#endif
        return ((size_t)(field / kWfGroup) * n + slot) * (size_t)kWfGroup + (size_t)(field % kWfGroup);
    }
    VDEV float &f(int field, unsigned slot) const { return base[at(field, slot)]; }
    VDEV int &i(int field, unsigned slot) const { return reinterpret_cast<int *>(base)[at(field, slot)]; }
    VDEV uint32_t &u(int field, unsigned slot) const { return reinterpret_cast<uint32_t *>(base)[at(field, slot)]; }
    VDEV V3 v3(int field, unsigned slot) const { return V3{f(field, slot), f(field + 1, slot), f(field + 2, slot)}; }
    VDEV void set3(int field, unsigned slot, V3 v) const { f(field, slot) = v.x; f(field + 1, slot) = v.y; f(field + 2, slot) = v.z; }
    VDEV Spec sp3(int field, unsigned slot) const { return Spec{f(field, slot), f(field + 1, slot), f(field + 2, slot)}; }
    VDEV void sets(int field, unsigned slot, Spec v) const { f(field, slot) = v.r; f(field + 1, slot) = v.g; f(field + 2, slot) = v.b; }
    VDEV void load_rng(int field, unsigned slot, Rng &r) const {
        r.state = (uint64_t)u(field, slot) | ((uint64_t)u(field + 1, slot) << 32);
        r.inc = (uint64_t)u(field + 2, slot) | ((uint64_t)u(field + 3, slot) << 32);
    }
    VDEV void store_rng(int field, unsigned slot, const Rng &r) const {
        u(field, slot) = (uint32_t)r.state; u(field + 1, slot) = (uint32_t)(r.state >> 32);
        u(field + 2, slot) = (uint32_t)r.inc; u(field + 3, slot) = (uint32_t)(r.inc >> 32);
    }
};

// ---- what the walk kernels hand back, and where -----------------------------------------------------------------------------
// A grey medium's walk (the walk kernels' kGrey instantiations) works on spectra with three equal channels: its results travel as
// ONE float apiece, packed into two quads (+ the selected position's) instead of seven, and the shadow walk's four spectra into
// one -- every quad a lane touches costs a line once the lists have thinned out (11 of the ~60 quad accesses of a path-loop
// iteration).  `compact` (WfArgs::compact_results) is what both sides agree on: the host sets it from the walk medium's kGrey.
struct WfWalkResult {
    Spec T_maj, trRatioEst, beta_rs, r_u_rs, sel_num, sel_den;
    float weightSum, sel_wi;
    V3 sel_p;
};
VDEV void wf_store_walk_result(const WfPool &P, unsigned slot, bool compact, const WfWalkResult &r) {
    if (compact) {
        P.f(WF_TMAJ + 0, slot) = r.T_maj.r; P.f(WF_TMAJ + 1, slot) = r.trRatioEst.r; P.f(WF_TMAJ + 2, slot) = r.beta_rs.r; P.f(WF_WSUM, slot) = r.weightSum;
        P.f(WF_TRR + 0, slot) = r.r_u_rs.r; P.f(WF_TRR + 1, slot) = r.sel_num.r; P.f(WF_TRR + 2, slot) = r.sel_den.r; P.f(WF_SELW, slot) = r.sel_wi;
    } else {
        P.sets(WF_TMAJ, slot, r.T_maj);
        P.f(WF_WSUM, slot) = r.weightSum;
        P.sets(WF_TRR, slot, r.trRatioEst);
        P.sets(WF_BRS, slot, r.beta_rs);
        P.sets(WF_RURS, slot, r.r_u_rs);
        P.f(WF_SELW, slot) = r.sel_wi;
        P.sets(WF_SELNUM, slot, r.sel_num);
        P.sets(WF_SELDEN, slot, r.sel_den);
    }
    P.set3(WF_SELP, slot, r.sel_p);
}
VDEV WfWalkResult wf_load_walk_result(const WfPool &P, unsigned slot, bool compact) {
    WfWalkResult r;
    if (compact) {
        r.T_maj = sp(P.f(WF_TMAJ + 0, slot)); r.trRatioEst = sp(P.f(WF_TMAJ + 1, slot)); r.beta_rs = sp(P.f(WF_TMAJ + 2, slot));
        r.r_u_rs = sp(P.f(WF_TRR + 0, slot)); r.sel_num = sp(P.f(WF_TRR + 1, slot)); r.sel_den = sp(P.f(WF_TRR + 2, slot));
    } else {
        r.T_maj = P.sp3(WF_TMAJ, slot);
        r.trRatioEst = P.sp3(WF_TRR, slot);
        r.beta_rs = P.sp3(WF_BRS, slot);
        r.r_u_rs = P.sp3(WF_RURS, slot);
        r.sel_num = P.sp3(WF_SELNUM, slot);
        r.sel_den = P.sp3(WF_SELDEN, slot);
    }
    r.weightSum = P.f(WF_WSUM, slot);
    r.sel_wi = P.f(WF_SELW, slot);
    r.sel_p = P.v3(WF_SELP, slot);
    return r;
}
struct WfShadowResult { Spec T_ray, r_l, r_u, T_res; };
VDEV void wf_store_shadow_result(const WfPool &P, unsigned slot, bool compact, const WfShadowResult &r) {
    if (compact) {
        P.f(WF_TRAY + 0, slot) = r.T_ray.r; P.f(WF_TRAY + 1, slot) = r.r_l.r; P.f(WF_TRAY + 2, slot) = r.r_u.r; P.f(WF_TRAY + 3, slot) = r.T_res.r;
    } else {
        P.sets(WF_TRAY, slot, r.T_ray); P.sets(WF_SRL, slot, r.r_l); P.sets(WF_SRU, slot, r.r_u); P.sets(WF_STMAJ, slot, r.T_res);
    }
}
VDEV WfShadowResult wf_load_shadow_result(const WfPool &P, unsigned slot, bool compact) {
    if (compact) return WfShadowResult{sp(P.f(WF_TRAY + 0, slot)), sp(P.f(WF_TRAY + 1, slot)), sp(P.f(WF_TRAY + 2, slot)), sp(P.f(WF_TRAY + 3, slot))};
    return WfShadowResult{P.sp3(WF_TRAY, slot), P.sp3(WF_SRL, slot), P.sp3(WF_SRU, slot), P.sp3(WF_STMAJ, slot)};
}

// per-iteration control block (one per path-loop iteration, zeroed once per launch of the pipeline)
struct WfIter {
    // (every word below that kernels hammer with atomics sits on a 128-byte line of its own: the three list counters are bumped by
    // every workgroup of a dense kernel every few rounds, the chunk cursor once per round)
    unsigned int n_active;       // list entries of this iteration: paths with a segment in flight (+ WFL_DEAD ones awaiting their NEE)
    unsigned int pad0[31];
    unsigned int dense_head;     // chunk cursor of this iteration's vertex kernel (VSPG_WF_DENSE_CLAIM)
    unsigned int pad1[31];
    unsigned int n_walk;         // distance-walk jobs
    unsigned int pad2[31];
    unsigned int n_shadow;       // shadow-walk jobs
    unsigned int pad3[31];
    // Job cursors of the walk kernels: up to kWfSegs per list, each on a 128-byte line of its own (wf_claim_refill).
    // [0] distance walk  [1] shadow walk  [2], [3] k_wf_walk (launched with / without distance jobs)
    unsigned int seg_head[4][8][32];
};
constexpr int kWfSegs = 8;
static_assert(sizeof(WfIter) == 4 * 128 + 4 * 8 * 128, "WfIter: four header lines + the cursor lines");
struct WfArgs {
    const DScene *scene;
    WfPool P;
    float4 *film;
    float *isg_stats;
    const float *vsp_buf;
    int vsp_ready;
    int sample;                  // the sample index this pass renders
    PcgJump jump;                // PCG skip-ahead for sample * 65536
    unsigned int n_items;        // tiles * 64
    unsigned int tilesX;
    unsigned int *list_active;   // slots with a segment in flight, even iterations
    unsigned int *list_active2;  // ... odd iterations (k_wf_vertex reads one and writes the other)
    unsigned int *list_walk;
    unsigned int *list_shadow;
    WfIter *iters;               // [maxdepth + 2]
    unsigned long long *counters;
    int walk_rounds;             // majorant-cell advances tried per tracking step before the collision code runs
    int walk_refill;             // a walk wavefront refills its idle lanes once this many are idle
    TrainArgs train;             // training passes (a18): the wave's segment-record buffer; k_propagate runs after the pass
    int rec_cap;                 // records a path may keep (train_rec_capacity)
    int compact_results;         // the walk kernels run a grey medium: their results travel packed (wf_store_walk_result)
};
// the recorder of a path between two kernels of a training pass: records go straight to the path's column of the wave's
// buffer; n / cur / flags travel in the path record.  (scat_*: a record sees at most one NEE.)
VDEV void wf_rec_load(const WfArgs &a, unsigned slot, PathRecorder &rec) {
    rec.base = a.train.segbuf + slot;
    rec.stride = (int)a.train.n_items;
    rec.max_seg = a.rec_cap;
    rec.n = a.P.i(WF_RECN, slot);
    rec.cur = a.P.i(WF_RECCUR, slot);
    rec.cur_flags = a.P.u(WF_RECFL, slot);
    rec.scat_r = rec.scat_g = rec.scat_b = 0.f;
}
VDEV void wf_rec_store(const WfArgs &a, unsigned slot, const PathRecorder &rec) {
    a.P.i(WF_RECN, slot) = rec.n;
    a.P.i(WF_RECCUR, slot) = rec.cur;
    a.P.u(WF_RECFL, slot) = rec.cur_flags;
}
VDEV void wf_rec_load(const WfArgs &, unsigned, NullRecorder &) {}
VDEV void wf_rec_store(const WfArgs &, unsigned, const NullRecorder &) {}
VDEV void wf_rec_finish(const WfArgs &a, unsigned slot, const PathRecorder &rec) { a.train.seg_count[slot] = rec.n; }  // PropagateSamples (:627) follows in k_propagate
VDEV void wf_rec_finish(const WfArgs &, unsigned, const NullRecorder &) {}

VDEV void wf_pixel_of(unsigned slot, unsigned tilesX, int *px, int *py) {
    const unsigned tile = slot >> 6, l = slot & 63u;
    const unsigned ty = tile / tilesX, tx = tile - ty * tilesX;
    *px = (int)(tx * 8u + (l & 7u));
    *py = (int)(ty * 8u + (l >> 3));
}

// Appending to a global list.  One returning atomic per WAVEFRONT on the list's counter is too many: a single hot
// word serves ~88 returning atomics per microsecond on this chip, and a dense kernel over a 1080p wave would issue
// 32 k of them per list (measured: 0.6-0.75 ms per dense kernel, all of it that).  So a workgroup stages the slots of
// kWfStageRounds rounds in LDS (wave ballot + prefix count + one LDS atomic per wavefront) and flushes them with ONE
// global atomic.  List order is irrelevant: every path's result depends on its own state only.
constexpr int kWfStageRounds = 4;
#ifndef VSPG_WF_CHAIN_MAX
#define VSPG_WF_CHAIN_MAX 4
#endif
constexpr int kWfChainMax = VSPG_WF_CHAIN_MAX;  // no-walk segments a lane runs back to back in one k_wf_vertex pass (boundary scenes) before the path waits for the next iteration
// The dense kernels' list walk.  Static: workgroup b takes chunks b, b + G, ... (grid stride).  Claimed (VSPG_WF_DENSE_CLAIM, the
// vertex kernel): chunks of one workgroup's width from a cursor, one returning atomic and one barrier per chunk -- a chunk's cost
// follows what its paths do (ended, NEE or not, hit or escape), so static shares end on their slowest chunks.
#ifndef VSPG_WF_DENSE_CLAIM
#define VSPG_WF_DENSE_CLAIM 1
#endif
#ifndef VSPG_WF_REGROUP
#define VSPG_WF_REGROUP 1   // boundary scenes: links of a no-walk chain regrouped over a workgroup's lanes (k_wf_vertex); 0: run by the lane that began them
#endif
struct WfStage {
    unsigned int *buf;   // LDS, kWfStageRounds * block entries
    unsigned int *cnt;   // LDS
    VDEV void push(bool pred, unsigned slot) const {
        const unsigned long long m = __ballot(pred);
        if (m == 0ull) return;
        const int lane = threadIdx.x & 63;
        const int leader = __ffsll((long long)m) - 1;
        unsigned int base = 0;
        if (lane == leader) base = atomicAdd(cnt, (unsigned int)__popcll(m));
        base = __shfl(base, leader);
        if (pred) buf[base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = slot;
    }
    // every thread of the workgroup calls this (three barriers).  The counter is reset BETWEEN the second and the third
    // barrier, when every thread has read it and nobody can push yet: resetting it after the last barrier raced with the
    // next round's pushes of faster wavefronts (lost and duplicated list entries).
    VDEV void flush(unsigned int *list, unsigned int *gcount, unsigned int *s_base) const {
        __syncthreads();
        const unsigned int c = *cnt;
        if (threadIdx.x == 0 && c) *s_base = atomicAdd(gcount, c);
        __syncthreads();
        const unsigned int b = *s_base;
        if (threadIdx.x == 0) *cnt = 0;
        for (unsigned int i = threadIdx.x; i < c; i += blockDim.x) list[b + i] = buf[i];
        __syncthreads();
    }
};

// Two or three stages flushed together: their global atomics travel side by side (threads 0..N-1 ask, one each) and the workgroup
// pays three barriers, not three per list.  Same protocol as WfStage::flush.
template <int N>
VDEV void wf_flush_stages(const WfStage (&st)[N], unsigned int *const (&list)[N], unsigned int *const (&gcount)[N], unsigned int *s_base) {
    __syncthreads();
    unsigned int c[N];
#pragma unroll
    for (int k = 0; k < N; ++k) c[k] = *st[k].cnt;
    if (threadIdx.x < (unsigned)N) {
#pragma unroll
        for (int k = 0; k < N; ++k)
            if (threadIdx.x == (unsigned)k && c[k]) s_base[k] = atomicAdd(gcount[k], c[k]);
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)N) {
#pragma unroll
        for (int k = 0; k < N; ++k)
            if (threadIdx.x == (unsigned)k) *st[k].cnt = 0;
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const unsigned int b = s_base[k];
        for (unsigned int i = threadIdx.x; i < c[k]; i += blockDim.x) list[k][b + i] = st[k].buf[i];
    }
    __syncthreads();
}

// With a grey medium r_u and r_l are one float each: they share ONE quad with the NEE's two pdfs -- {r_u, r_l, scatterPDF, p_l}
// -- instead of two (every quad a lane touches is a line of its own once the lists have thinned out).
template <int GREY> constexpr int wf_rl_field() { return GREY >= 1 ? WF_RU + 1 : WF_RL; }
template <int GREY> constexpr int wf_spdf_field() { return GREY >= 1 ? WF_RU + 2 : WF_SPDF; }
template <int GREY>
VDEV void wf_store_path(const WfPool &P, unsigned slot, const PathState &st, const Sampler &sampler, int ch, const IsgSample &isg,
                        uint32_t extra_flags) {
    P.set3(WF_RO, slot, st.ro);
    P.set3(WF_RD, slot, st.rd);
    P.sets(WF_L, slot, st.L);
    P.sets(WF_BETA, slot, st.beta);
    if constexpr (GREY >= 1) {
        P.f(WF_RU, slot) = st.r_u.r;
        P.f(wf_rl_field<GREY>(), slot) = st.r_l.r;
    } else {
        P.sets(WF_RU, slot, st.r_u);
        P.sets(WF_RL, slot, st.r_l);
    }
    P.set3(WF_PCP, slot, st.prevCtx.p);
    P.i(WF_PCQ, slot) = st.prevCtx.quad;
    P.store_rng(WF_RNG, slot, sampler.rng);
    P.u(WF_FLAGS, slot) = pool_pack_flags(st, ch, isg, extra_flags | (st.in_medium ? (uint32_t)WFL_INMED : 0u));
    P.f(WF_RRC, slot) = st.rr_correction;
    P.f(WF_VSP, slot) = isg.vsp_used;  // (the pixel's primary VSP itself is consumed by the first segment's begin)
}
template <int GREY>
VDEV uint32_t wf_load_path(const WfPool &P, unsigned slot, PathState &st, Sampler &sampler, int *ch, IsgSample &isg) {
    st.ro = P.v3(WF_RO, slot);
    st.rd = P.v3(WF_RD, slot);
    st.L = P.sp3(WF_L, slot);
    st.beta = P.sp3(WF_BETA, slot);
    if constexpr (GREY >= 1) {
        st.r_u = sp(P.f(WF_RU, slot));
        st.r_l = sp(P.f(wf_rl_field<GREY>(), slot));
    } else {
        st.r_u = P.sp3(WF_RU, slot);
        st.r_l = P.sp3(WF_RL, slot);
    }
    st.prevCtx.p = P.v3(WF_PCP, slot);
    st.prevCtx.quad = P.i(WF_PCQ, slot);
    P.load_rng(WF_RNG, slot, sampler.rng);
    const uint32_t fl = P.u(WF_FLAGS, slot);
    st.depth = (int)(fl & FL_DEPTH_MASK);
    *ch = (int)((fl >> FL_CH_SHIFT) & 3u);
    st.specularBounce = (fl & FL_SPECULAR) != 0;
    st.anyNonSpecularBounces = false;
    st.lastVertexVolume = (fl & FL_LASTVOL) != 0;
    isg.valid = (fl & FL_ISG_VALID) != 0;
    isg.surface_event = (fl & FL_ISG_SURF) != 0;
    st.rr_correction = P.f(WF_RRC, slot);
    st.etaScale = 1;
    isg.vsp_used = P.f(WF_VSP, slot);
    st.vsp0 = isg.vsp_used;
    st.gs.vsp_next = -1.f;
    st.pce = 0.f;
    st.guideRR = false;
    st.in_medium = (fl & WFL_INMED) != 0;
    st.lu = 0.f;  // only k_wf_segment_vertex reads it (a temperature grid's emission) and recomputes it there
    return fl;
}

// DDA iterator <-> 8 floats at F.. + one packed word at FP: voxel (3 x 7 bits), step signs (3 bits), and the path's hero channel
// (2 bits: the walk kernels read it here instead of fetching the path's flags quad for it).  Distance walk: F = WF_IT, FP = WF_IT + 8
// (the quad it shares with tHit, the first uniform and the majorant scale); shadow walk: F = WF_SIT, FP = WF_SRDN + 3 (the
// direction's quad), its first uniform in the origin's quad (WF_SLO + 3) -- eight quads for a shadow job instead of nine.
constexpr int WF_SITP = WF_SRDN + 3, WF_SWU2 = WF_SLO + 3;
template <class Iter>
VDEV void wf_store_iter(const WfPool &P, unsigned slot, const Iter &it, int ch, int F = WF_IT, int FP = WF_IT + 8) {
    P.f(F + 0, slot) = it.tMin; P.f(F + 1, slot) = it.tMax;
    P.f(F + 2, slot) = it.ncx; P.f(F + 3, slot) = it.ncy; P.f(F + 4, slot) = it.ncz;
    P.f(F + 5, slot) = it.dtx; P.f(F + 6, slot) = it.dty; P.f(F + 7, slot) = it.dtz;
    P.u(FP, slot) = (uint32_t)it.vx | ((uint32_t)it.vy << 7) | ((uint32_t)it.vz << 14) | ((uint32_t)it.neg << 21) | ((uint32_t)ch << 24);
}
template <class Medium>
VDEV typename Medium::Iter wf_load_iter(const WfPool &P, unsigned slot, const Medium &medium, int *ch, int F = WF_IT, int FP = WF_IT + 8) {
    typename Medium::Iter it = medium.empty_iter();
    it.tMin = P.f(F + 0, slot); it.tMax = P.f(F + 1, slot);
    it.ncx = P.f(F + 2, slot); it.ncy = P.f(F + 3, slot); it.ncz = P.f(F + 4, slot);
    it.dtx = P.f(F + 5, slot); it.dty = P.f(F + 6, slot); it.dtz = P.f(F + 7, slot);
    const uint32_t pk = P.u(FP, slot);
    it.vx = (int)(pk & 127u); it.vy = (int)((pk >> 7) & 127u); it.vz = (int)((pk >> 14) & 127u); it.neg = (int)((pk >> 21) & 7u);
    *ch = (int)((pk >> 24) & 3u);
    return it;
}

// a finished path: RGBFilm::AddSample + the image-space statistics.  The pipeline renders ONE sample per pixel per
// pass, so nobody else touches the pixel: plain read-modify-write (film_add_sample_rmw).
VDEV void wf_finish_path(const WfArgs &a, unsigned slot, const PathState &st, const IsgSample &isg) {
    int px, py;
    wf_pixel_of(slot, a.tilesX, &px, &py);
    const size_t pidx = (size_t)py * a.scene->xres + px;
    const Spec L = finish_radiance(st.L);
    film_add_sample_rmw(a.film + pidx, L);
    isg_add_sample_rmw(a.isg_stats + pidx * VSPG_ISG_STATS, L, isg);
}

// ---- a15 cut at the shadow walk: SampleLd up to the transmittance estimate (:1136-1204) ... ------------------------
struct ShadowSetup {
    int status;  // 0: no contribution; 1: ratio-tracking walk needed; 2: contributes with T_ray = 1 (no medium on the way);
                 // 3 (medium boundaries): the light ray crosses the medium more than once -- estimated in the lane, Ld below
    float tMax;  // status 1: where the walk's segment ends, in units of ld (1 - ShadowEpsilon, or the interface the ray leaves the medium through)
    Spec Ld;
    Spec f_hat, L;
    float p_l, scatterPDF;
    V3 lo, ld;
    Rng rng;
    float us;
    bool delta_light;
};
template <class Medium, class PC, class GD = GDist>
VDEV ShadowSetup sample_Ld_begin(const DScene &S, const Medium &medium, int ch, Spec r_p, const Intr &intr, const Bsdf *bsdf, Sampler &sampler, PC &pc,
                                 const GD *gd = nullptr, bool use_gd = false) {
    ShadowSetup r;
    r.status = 0;
    r.tMax = 1 - kShadowEps;
    r.Ld = sp(0.f);
    r.f_hat = r.L = sp(0.f);
    r.p_l = r.scatterPDF = r.us = 0.f;
    r.lo = r.ld = mk(0, 0, 0);
    r.rng.state = r.rng.inc = 0;
    r.delta_light = false;
    V3 ctxp = intr.pi.mid();
    if (intr.is_surface && bsdf->has_lobes) ctxp = offset_ray_origin(intr.pi, intr.n, intr.wo);  // :1147-1149
    float u = sampler.get1d();
    const int n_all = S.n_lights + S.n_inf;
    bool have_light = n_all > 0;
    int lightIndex = 0;
    float lightPmf = 0;
    if (S.lsamp.mode != VSPG_LIGHTSAMPLER_UNIFORM) {  // power / BVH sampler of a multi-light scene (vspg_lightsampler.h)
        have_light = light_sampler_sample(S, ctxp, intr.is_surface ? intr.n : mk(0, 0, 0), u, &lightIndex, &lightPmf);
    } else if (have_light) {  // UniformLightSampler::Sample (lightsamplers.h:33-38)
        int li = (int)(u * (float)n_all);
        lightIndex = li < n_all - 1 ? li : n_all - 1;
        lightPmf = wrcp((float)n_all);
    }
    float ul0 = sampler.get1d(), ul1 = sampler.get1d();
    if (!have_light) return r;
    LightLi ls;
    if (!sample_light(S, lightIndex, ctxp, ul0, ul1, &ls, &r.delta_light)) return r;
    r.p_l = lightPmf * ls.pdf;
    V3 wo = intr.wo, wi = ls.wi;
    // GuidedBSDF::PDF / GuidedPhaseFunction::PDF (guiding.h:271-289, 542-558) when the vertex is guided (sample_Ld, vspg_path.h)
    if (intr.is_surface) {
        r.f_hat = bsdf_f(*bsdf, wo, wi) * absdot(wi, intr.n);
        float bsdfPDF = bsdf_pdf(*bsdf, wo, wi);
        if (use_gd) bsdfPDF = ((1.0f - kGuidingProbability) * bsdfPDF) + (kGuidingProbability * gdist_pdf(*gd, wi));
        r.scatterPDF = 1.0f * bsdfPDF;
    } else {
        float p = henyey_greenstein(dot(wo, wi), intr.g);
        r.f_hat = sp(p);
        float phasePDF = p;
        if (use_gd) phasePDF = ((1.0f - kGuidingProbability) * phasePDF) + (kGuidingProbability * gdist_pdf(*gd, wi));
        r.scatterPDF = 1.0f * phasePDF;
    }
    if (!nonzero(r.f_hat)) return r;
    r.L = ls.L;
    // lightRay = intr.SpawnRayTo(ls->pLight) (interaction.h:111-115, ray.h:103-108)
    V3 pf = offset_ray_origin(intr.pi, intr.n, ls.pLight.mid() - intr.pi.mid());
    V3 pt = offset_ray_origin(ls.pLight, ls.nLight, pf - ls.pLight.mid());
    r.lo = pf;
    r.ld = pt - pf;
    r.rng.set_sequence(hash_v3(r.lo), hash_v3(r.ld));  // :1193
    if (has_bnd<Medium::kBnd>(S)) {
        // Medium boundaries: the light ray is a chain of segments (:1195-1244), each ending at the closest hit -- a surface with a
        // material blocks, an interface hands the ray on in the medium of its far side.  Which segments there are is geometry
        // (no random number decides it), so the chain is walked HERE; what is left for the walk kernel is the ratio tracking
        // inside the medium.  A bounding shape is usually convex: at most ONE segment lies in the medium, and it becomes the
        // shadow-walk job, with its own origin, direction and end.  (The private RNG draws `u` at the start of a medium segment
        // and nowhere else outside the tracking: the job's first uniform is the same number.)  A ray that crosses the medium more
        // than once is estimated in the lane (sample_Ld_shadow, the per-lane kernels' code).
        V3 lo = r.lo, ld = r.ld;
        bool lmed = S.medium_type != VSPG_MEDIUM_NONE && get_medium(intr.sflags, intr.n, ld, intr.medium);
        int n_med = 0;
        V3 jlo = lo, jld = ld;
        float jt = 1 - kShadowEps;
        bool blocked = false;
        while (!(ld.x == 0 && ld.y == 0 && ld.z == 0)) {
            const Isect si = scene_intersect(S, lo, ld, 1 - kShadowEps);
            int sfl = 0;
            if (si.hit) {
                sfl = surf_flags(S, si.quad);
                if (!(sfl & SURF_INTERFACE)) { blocked = true; break; }
            }
            if (lmed) {
                if (n_med == 0) { jlo = lo; jld = ld; jt = si.hit ? si.t : 1 - kShadowEps; }
                n_med++;
            }
            if (!si.hit) break;
            const P3i spi = surf_pi(S, si.quad, si.p);
            const V3 nf = offset_ray_origin(spi, si.n, ls.pLight.mid() - spi.mid());
            const V3 nt = offset_ray_origin(ls.pLight, ls.nLight, nf - ls.pLight.mid());
            lo = nf;
            ld = nt - nf;
            lmed = S.medium_type != VSPG_MEDIUM_NONE && get_medium(sfl, si.n, ld, lmed);
        }
        if (blocked) {
            pc.shadow_ray();
            return r;  // status 0.  (The reference may have returned 0 earlier, at a medium segment whose estimate died: 0 either way.)
        }
        if (n_med >= 2) {
            r.Ld = sample_Ld_shadow<Medium>(S, medium, intr, ch, ls, r.f_hat, r.p_l, r.scatterPDF, r_p, pc, r.delta_light);  // (counts the shadow ray)
            r.status = 3;
            return r;
        }
        pc.shadow_ray();
        r.status = 2;
        if (n_med == 1) {
            r.lo = jlo;
            r.ld = jld;
            r.tMax = jt;
            r.us = r.rng.uniform();
            r.status = 1;
        }
        return r;
    }
    pc.shadow_ray();
    r.status = 2;
    if (!(r.ld.x == 0 && r.ld.y == 0 && r.ld.z == 0)) {
        if (scene_intersect_any(S, r.lo, r.ld, 1 - kShadowEps)) {
            r.status = 0;
            return r;
        }
        if (S.medium_type != VSPG_MEDIUM_NONE) {
            r.us = r.rng.uniform();
            r.status = 1;
        }
    }
    return r;
}
// ... and from the estimate on (:1233-1251).  walked: the medium block ran (T_maj is the walk's residual majorant transmittance)
template <int GREY = 0>
VDEV Spec sample_Ld_end(bool walked, bool delta_light, Spec T_ray, Spec r_l, Spec r_u, Spec T_maj, int ch, Spec f_hat, Spec Ll, float p_l, float scatterPDF, Spec r_p) {
    if (walked) {
        T_ray = mul_tmaj_ratio<GREY>(T_ray, T_maj, ch);
        r_l = mul_tmaj_ratio<GREY>(r_l, T_maj, ch);
        r_u = mul_tmaj_ratio<GREY>(r_u, T_maj, ch);
    }
    if (!nonzero(T_ray)) return sp(0.f);
    r_l = r_l * (r_p * p_l);
    r_u = r_u * (r_p * scatterPDF);
    if (delta_light) return wdiv(f_hat * T_ray * Ll, avg(r_l));  // IsDeltaLight (:1248-1249)
    return wdiv(f_hat * T_ray * Ll, avg(r_l + r_u));
}

// ---- one tracking step of a walk (media_sampleTMaj.h:66-114 == :190-246), cut where the callback sits --------------
// A walk is: { while segments: [zero majorant: T *= exp, next] else inner loop { draw t; collision -> callback | leave } }.
// State between two steps: the iterator, the current segment, the position in it, T_maj, u, the private RNG.
template <class Medium>
struct WalkState {
    typename Medium::Iter iter;
    Spec sigma_maj;   // of the current segment (scaled)
    float seg_tMax, tMin;
    Spec T_maj;
    float u;
    Rng rng;
    int count;
    bool in_seg;
};
enum { WALK_MOVED = 0, WALK_COLLISION = 1, WALK_END = 2 };
// advance: everything up to (not including) the collision.  Returns WALK_COLLISION with *t_out set when a tentative collision
// was drawn inside the current segment, WALK_END when the iterator ran out, WALK_MOVED otherwise (call again).
template <class Medium, bool CAP>
VDEV int walk_advance(WalkState<Medium> &w, int ch, float scale, float *t_out, int cap = 10000) {
    if (!w.in_seg) {
        MajSeg seg;
        // (An empty cell costs a whole tracking round here -- T_maj *= FastExp(-0) == 1 and nothing else.  Crossing a RUN of empty
        // cells inside this call was tried: bit-identical, and up to 2x slower on a cloud with empty space around it -- the lanes in
        // the run spin while the rest of the wavefront waits, where the round structure keeps everyone stepping.  HISTORY.)
        if (!w.iter.next(&seg)) return WALK_END;
        seg.sigma_maj = seg.sigma_maj * scale;
        const float smaj = ch_of(seg.sigma_maj, ch);
        if (smaj == 0) {
            if (nonzero(seg.sigma_maj)) {  // (zero in every channel: the factor is FastExp(-0) == 1 exactly, and x * 1 is x)
                float dt = seg.tMax - seg.tMin;
                if (isinf_(dt)) dt = kFltMax;
                w.T_maj = w.T_maj * fast_exp(seg.sigma_maj * -dt);
            }
            return WALK_MOVED;
        }
        w.sigma_maj = seg.sigma_maj;
        w.seg_tMax = seg.tMax;
        w.tMin = seg.tMin;
        w.in_seg = true;
    }
    const float smaj = ch_of(w.sigma_maj, ch);
    if constexpr (CAP) w.count++;
    const float t = w.tMin + sample_exponential(w.u, smaj);
    w.u = w.rng.uniform();
    if (t < w.seg_tMax) {
        if (CAP && w.count > cap) {  // media_sampleTMaj.h:216-219: leaves the inner loop only
            w.in_seg = false;
            return WALK_MOVED;
        }
        *t_out = t;
        return WALK_COLLISION;
    }
    float dt = w.seg_tMax - w.tMin;
    if (isinf_(dt)) dt = kFltMax;
    w.T_maj = w.T_maj * fast_exp(w.sigma_maj * -dt);
    w.in_seg = false;
    return WALK_MOVED;
}

// =====================================================================================================================
// kernels
// =====================================================================================================================
#ifdef VSPG_WF_STATS  // diagnostic build: where the lanes of the distance walk are (scripts/wf_stats.py reads them back)
// [0] loop iterations  [1] lanes with a job  [2] advance rounds run  [3] lanes in them  [4] collision steps run
// [5] lanes in them  [6] lanes with a job in draining iterations  [7] draining iterations (the wavefront has no job left to claim)
__device__ unsigned long long g_wf_stats[16];
#define VSPG_WF_STAT(k, i, v) st_[(i)] += (unsigned long long)(v)
#define VSPG_WF_STAT_DECL unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define VSPG_WF_STAT_FLUSH(k) do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_wf_stats[(k) * 8 + i_], st_[i_]); } while (0)
#else
#define VSPG_WF_STAT(k, i, v)
#define VSPG_WF_STAT_DECL
#define VSPG_WF_STAT_FLUSH(k)
#endif
constexpr int kWfBlock = 256;
#ifndef VSPG_WF_WALK_WAVES
#define VSPG_WF_WALK_WAVES 4
#endif
#ifndef VSPG_WF_SHADOW_WAVES
#define VSPG_WF_SHADOW_WAVES 5
#endif
constexpr int kWfWalkWavesPerSimd = VSPG_WF_WALK_WAVES;  // launch bound of the walk kernels (register budget 128)
constexpr int kWfShadowWavesPerSimd = VSPG_WF_SHADOW_WAVES;  // the shadow walk carries less state (<= 96 registers)
#ifndef VSPG_WF_MERGED_WAVES
#define VSPG_WF_MERGED_WAVES 4   // (5: 96 VGPRs + 12-36 B of scratch; measured slower on every workload but NanoVDB semantics)
#endif
constexpr int kWfMergedWavesPerSimd = VSPG_WF_MERGED_WAVES;  // k_wf_walk (both kinds of job: the two kinds' running state shares registers)
constexpr int kWfRefill = 16;
#ifndef VSPG_WF_VERTEX_WAVES
#define VSPG_WF_VERTEX_WAVES 3   // launch bound of the unguided vertex kernel (waves per SIMD)
#endif
#ifndef VSPG_WF_VERTEX_WAVES_GUIDED
#define VSPG_WF_VERTEX_WAVES_GUIDED 2   // ... of the guided one (its product mixture lives in registers).  Measured at 3 (168 VGPRs, round 4, same
#endif                                  // box): cloud-guided 9.6 -> 10.9 ms per wave -- the spills cost more than the third wavefront hides
#ifndef VSPG_WF_VERTEX_WAVES_BND
#define VSPG_WF_VERTEX_WAVES_BND 2      // ... of the unguided one with the medium-boundary code compiled in (the light ray's segment chain, the
#endif                                  // interface skip): at 3 it spills ~300 B per lane; 2 measured 6.19 against 6.50 ms per cloud-scene wave
#ifndef VSPG_WF_CLAIM
#define VSPG_WF_CLAIM 128
#endif
constexpr int kWfClaim = VSPG_WF_CLAIM;  // jobs a walk wavefront claims per returning atomic  // a walk wavefront refills its idle lanes once this many are idle

template <class REC>
struct WfCountersT : PathCountersT<REC> {
    uint32_t paths;
    VDEV void path() { paths++; }
    VDEV void zero() { this->segments = this->volume_scatters = this->surface_hits = this->density_queries = this->shadow_rays = this->shadow_queries = paths = 0; }
};
using WfCounters = WfCountersT<NullRecorder>;
template <class REC>
VDEV void wf_flush_counters(const WfCountersT<REC> &pc, unsigned long long *g) {
    __shared__ unsigned int s_c[CNT_COUNT];
    if (threadIdx.x < CNT_COUNT) s_c[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t v[CNT_COUNT] = {pc.paths, pc.segments, pc.volume_scatters, pc.surface_hits, pc.density_queries, pc.shadow_rays, pc.shadow_queries};
    for (int k = 0; k < CNT_COUNT; ++k) {
        uint32_t x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&s_c[k], x);
    }
    __syncthreads();
    if (threadIdx.x < CNT_COUNT && s_c[threadIdx.x]) atomicAdd(&g[threadIdx.x], (unsigned long long)s_c[threadIdx.x]);
}

// the block's medium: GridMedium stages its 16^3 majorant grid in LDS (every DDA step reads it), the 64^3 one stays in L2
template <class Medium>
VDEV Medium wf_block_medium(const DScene &S) {
    const float *maj_ptr = nullptr;
    static_assert(!Medium::kMajLds || Medium::kRes == kMajRes, "GridMediumT's MAJLDS: only the 16^3 majorant grid has an LDS copy");
    if constexpr (Medium::kRes == kMajRes) {
        __shared__ float s_maj[kMajRes * kMajRes * kMajRes];
        const float4 *src = reinterpret_cast<const float4 *>(S.majorant);
        float4 *dst = reinterpret_cast<float4 *>(s_maj);
        for (int i = threadIdx.x; i < kMajRes * kMajRes * kMajRes / 4; i += kWfBlock) dst[i] = src[i];
        maj_ptr = s_maj;
    }
    stage_scene_lds(S);
    __syncthreads();
    return MediumMaker<Medium>::make(S, maj_ptr);
}

// ---- the begin of a segment: li_segment_a up to the traversal ----------------------------------------------------------
// Scene intersection, the segment's RNG, VSP fetch, the resampling routine's prologue; stores the path (with `extra` flags: what
// the caller still has to say about the PREVIOUS vertex) and, if the traversal has anything to do, the distance-walk job.
// SKIPS (boundary scenes, passes that record nothing): a segment that needs no distance walk and ends on a non-emissive interface
// surface -- the camera ray on its way to the cloud's bounding shape, a bounce off the ground back into it -- has no vertex and adds
// nothing to L: Li's `continue` (:399-404) is taken right here (SkipIntersection, GetMedium: li_surface_pre's skip branch) and the
// segment behind the boundary begun in its place, instead of a launch of every kernel to find that out.
template <class Medium, bool GUIDED, bool SKIPS = false, class PC>
VDEV void wf_segment_begin(const WfArgs &a, const DScene &S, const Medium &medium, unsigned slot, PathState &st, Sampler &sampler, int ch,
                           IsgSample &isg, PC &pc, uint32_t extra, bool *walk) {
    const WfPool &P = a.P;
    constexpr int G = Medium::kGrey;
    // ---- li_segment_a up to the traversal (:312-325, sample_distance / SampleT_maj_Resampling prologue) --
    Isect si;
    bool in_medium;
    for (int chain = 0;; ++chain) {
        pc.segment();
        si = scene_intersect(S, st.ro, st.rd, kInf);
        // :318 `if (ray.medium && !std::isinf(tMax))`: no distance sampling for a ray outside the medium or one that escapes the scene
        in_medium = !has_bnd<Medium::kBnd>(S) ? S.medium_type != VSPG_MEDIUM_NONE : st.in_medium;
        if constexpr (SKIPS && Medium::kBnd == 1) {
            if (!in_medium && si.hit && chain < kWfChainMax) {
                const int sfl = surf_flags(S, si.quad);
                const bool emits = !is_tri(si.quad) && !is_sphere(si.quad) && quad_at(si.quad).is_light;
                if ((sfl & SURF_INTERFACE) && !emits) {
                    st.ro = offset_ray_origin(surf_pi(S, si.quad, si.p), si.n, st.rd);
                    st.in_medium = S.medium_type != VSPG_MEDIUM_NONE && get_medium(sfl, si.n, st.rd, st.in_medium);
                    continue;
                }
            }
        }
        break;
    }
    const float tMax = si.hit ? si.t : kInf;
    extra |= FL_LIVE | (si.hit ? (uint32_t)WFL_HIT : 0u);
    P.set3(WF_VXP, slot, si.p);
    P.i(WF_VXG, slot) = si.quad;
    P.f(WF_VXT, slot) = si.t;
    if (in_medium && si.hit) {
        Rng rng;
        {
            uint64_t hash0 = hash_float(sampler.get1d());
            uint64_t hash1 = hash_float(sampler.get1d());
            rng.set_sequence(hash0, hash1);
        }
        bool guide = false;
        if (has_bnd<Medium::kBnd>(S) && st.depth == 0) {  // the camera segment may reach the medium through a boundary, an iteration or more after
            int px, py;                           // the path started: the pixel's primary VSP is read where it is used
            wf_pixel_of(slot, a.tilesX, &px, &py);
            st.vsp0 = (a.vsp_ready & VSP_READY) ? a.vsp_buf[(size_t)py * S.xres + px] : 0.5f;
        }
        const float vsp = fetch_vsp<GUIDED>(S, st, &guide);
        if (st.depth == 0) isg.vsp_used = guide ? vsp : -1.f;
        const float u = sampler.get1d();
        const float tM = tMax * len(st.rd);
        const V3 rdn = normalize(st.rd);
        const auto iter = medium.sample_ray(st.ro, rdn, tM);
        // The pre-pass (media_sampleTMaj.h:153-168) serves the VSP-guided case (majorantScale, the zero-candidate
        // compensation); otherwise all it decides is the early return for a ray whose majorants are all zero.  For
        // a grey medium the traversal itself gives that ray the same result -- every cell multiplies T_maj by
        // FastExp(-0) == 1, no callback runs, no sampler dimension is drawn -- so unguided segments (every segment past
        // the camera ray when only the primary VSP guides) skip the DDA sweep and go straight to the walk.  (Chromatic
        // media keep it: with a zero majorant in the hero channel only, the early return and the traversal differ in
        // the other channels of T_maj.)
        float totalLength = 0.f;
#ifdef VSPG_WF_EXPERIMENT_NO_PREPASS  // (timing experiment only, WRONG results: what does the sweep cost?)
        if (false) {
#else
        if (guide || !Medium::kGrey) {
#endif
            auto pre = iter;
            while (true) {
                MajSeg seg;
                if (!pre.next(&seg)) break;
                const float smaj = ch_of(seg.sigma_maj, ch);
                if (smaj == 0) continue;
                totalLength += smaj * (seg.tMax - seg.tMin);
            }
        } else if (iter.tMin < iter.tMax) {
            totalLength = 1.f;  // (only its being non-zero matters below)
        }
        if (guide) extra |= WFL_GUIDE;
        if (totalLength == 0.f) {
            extra |= WFL_NOWALK;
        } else {
            float majorantScale = 1.0f, vrc = vsp;
            if (guide) {
                float minTotalLength = -logf_(1 - vsp);
                if (minTotalLength > totalLength) {
                    majorantScale = minTotalLength / totalLength;
                    totalLength = minTotalLength;
                }
                float expNegTotalLength = fast_exp(-totalLength);
                vrc = wdiv(vsp, 1 - expNegTotalLength);
            }
            wf_store_iter(P, slot, iter, ch);
            P.set3(WF_RDN, slot, rdn);
            P.store_rng(WF_WRNG, slot, rng);
            P.f(WF_WU, slot) = u;
            P.f(WF_MSCALE, slot) = majorantScale;
            P.f(WF_VRC, slot) = vrc;
            *walk = true;
        }
    } else {
        extra |= WFL_NODIST;
    }
    wf_store_path<G>(P, slot, st, sampler, ch, isg, extra);
}

// ---- the next segment's begin as its own kernel (guided pipelines) -----------------------------------------------------------
// The guided vertex keeps its product mixture in registers and fills the register file on its own.  Inlined behind it, the
// segment's begin (BVH walk, DDA pre-pass) pushed the allocator into spilling inside the mixture loops (cloud-guided 11.0 -> 11.4
// ms per wave); as a real call it cost more still (12.4: the path goes through the stack).  So guided pipelines keep the two
// apart: k_wf_vertex parks the path, this kernel walks the same list and begins the segments.
template <class Medium, bool GUIDED = true, bool TRAIN = false>
__global__ __launch_bounds__(kWfBlock, 3) void k_wf_begin(WfArgs a, int it) {
    const DScene &S = *a.scene;
    const Medium medium = wf_block_medium<Medium>(S);
    const WfPool &P = a.P;
    constexpr int G = Medium::kGrey;
    WfIter *I = &a.iters[it];
    const unsigned n = I->n_active;
    const unsigned int *list = (it & 1) ? a.list_active2 : a.list_active;
    WfCountersT<typename std::conditional<TRAIN, PathRecorder, NullRecorder>::type> pc;
    pc.zero();
    __shared__ unsigned int s_stage[kWfStageRounds * kWfBlock], s_cnt, s_gbase;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const WfStage stB{s_stage, &s_cnt};
    int round = 0;
    for (unsigned base = blockIdx.x * kWfBlock; base < n; base += gridDim.x * kWfBlock, ++round) {
        const unsigned idx = base + threadIdx.x;
        bool walk = false;
        unsigned slot = 0;
        if (idx < n) {
            slot = list[idx];
            const uint32_t fl0 = P.u(WF_FLAGS, slot);
            if (!(fl0 & WFL_DEAD)) {
                PathState st;
                Sampler sampler;
                IsgSample isg;
                int ch;
                const uint32_t fl = wf_load_path<G>(P, slot, st, sampler, &ch, isg);
                st.gs.vsp_next = P.f(WF_GSVSP, slot);
                wf_segment_begin<Medium, GUIDED, !TRAIN>(a, S, medium, slot, st, sampler, ch, isg, pc, fl & (uint32_t)(WFL_SHADOW_WALK | WFL_DELTA), &walk);
            }
        }
        stB.push(walk, slot);
        if ((round % kWfStageRounds) == kWfStageRounds - 1) stB.flush(a.list_walk, &I->n_walk, &s_gbase);
    }
    stB.flush(a.list_walk, &I->n_walk, &s_gbase);
    wf_flush_counters(pc, a.counters);
}

// ---- camera rays + the first segment's begin -----------------------------------------------------------------------------
template <class Medium, bool GUIDED = false, bool TRAIN = false>
__global__ __launch_bounds__(kWfBlock, 3) void k_wf_start(WfArgs a) {
    const DScene &S = *a.scene;
    const Medium medium = wf_block_medium<Medium>(S);
    const unsigned n = a.n_items;
    WfIter *I = &a.iters[0];
    WfCountersT<typename std::conditional<TRAIN, PathRecorder, NullRecorder>::type> pc;
    pc.zero();
    __shared__ unsigned int s_stage[2][kWfStageRounds * kWfBlock], s_cnt[2], s_gbase[2];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const WfStage stA{s_stage[0], &s_cnt[0]}, stB{s_stage[1], &s_cnt[1]};
    int round = 0;
    for (unsigned base = blockIdx.x * kWfBlock; base < n; base += gridDim.x * kWfBlock, ++round) {
        const unsigned slot = base + threadIdx.x;
        bool alive = false, walk = false;
        if (slot < n) {
            int px, py;
            wf_pixel_of(slot, a.tilesX, &px, &py);
            if (px < S.xres && py < S.yres) {
                PathState st;
                Sampler sampler;
                IsgSample isg;
                int ch = 0;
                start_path(S, a.vsp_buf, a.vsp_ready, px, py, a.jump, sampler, st, &ch, isg);
                alive = true;
                if constexpr (TRAIN) {
                    wf_rec_load(a, slot, pc.rec);
                    pc.rec.reset();
                    wf_rec_store(a, slot, pc.rec);
                }
                wf_segment_begin<Medium, GUIDED, !TRAIN>(a, S, medium, slot, st, sampler, ch, isg, pc, 0u, &walk);
            }
        }
        stA.push(alive, slot);
        stB.push(walk, slot);
        if ((round % kWfStageRounds) == kWfStageRounds - 1) wf_flush_stages<2>({stA, stB}, {a.list_active, a.list_walk}, {&I->n_active, &I->n_walk}, s_gbase);
    }
    wf_flush_stages<2>({stA, stB}, {a.list_active, a.list_walk}, {&I->n_active, &I->n_walk}, s_gbase);
    wf_flush_counters(pc, a.counters);
}

// ---- the two walk kernels share the job-claiming loop ---------------------------------------------------------------
// claim: lanes with `want` set receive the next jobs of list[0, n) (slot in *slot_out, returns true); the wavefront keeps a
// local range [next, end) claimed kWfClaim jobs at a time with one returning atomic (64 at a time put 32 k returning
// atomics per launch on one word: a third of the kernel's time).
// Job order.  Tried (round 2): longest-processing-time-first -- the distance-walk jobs bucketed by the ray's majorant optical
// depth (the expected number of tentative collisions, known from the segment's pre-pass), thickest bucket first, so that a launch
// does not end on a few long walks.  With four buckets the lanes holding a job rose from 44.6 to 52.2 of 64 and the loop iterations
// fell by 15 % (scripts/wf_stats.py) -- and the wave got 5 % SLOWER (11.7 -> 12.3 ms): jobs in list order are jobs in slot order,
// so a wavefront's claim reads and writes neighbouring records (whole cache lines); bucketed, every 16-byte group of a job's
// record is its own line.  Two buckets (only the thickest 10-15 % pulled forward) were within +-0.5 %.  Dropped.
//
// Draining.  30 % of the distance walk's loop iterations on the 256^3 cloud are run by wavefronts that have no job left to claim,
// with 14.6 of 64 lanes still walking (NanoVDB semantics: 40 %, 10.2 lanes; the others run 57.7 lanes).  Tried (round 2): a
// draining wavefront down to 16 / 24 / 32 / 48 lanes PARKS its walks -- full tracking state into the record -- and a second
// launch resumes them packed, 64 to a wavefront; bit-identical, and 12-15 % SLOWER (11.1 -> 12.4-13.0 ms; 16.3 -> 17.5): the
// draining iterations run on a mostly empty chip and are cheap in time, the park / resume traffic and the second launch's own
// ramp and tail are not.  Dropped.
//
// Claim size.  The job cursor is ONE hot word (~88 returning atomics per microsecond on this part), so claims are large; but a
// launch ends on the wavefronts still working through their last claim, and the later path-loop iterations have a tenth of
// the first one's jobs.  Per launch: jobs / (4 * resident wavefronts), clamped to [32, 256] -- 126 for the first iteration of a
// 1080p wave (the fixed 128 this replaced), 32 for the short lists, which then spread over every wavefront.  Measured on the
// 256^3 cloud (scripts/gpu_variants_wl.sh, same box): fixed 128: 11.7 ms; fixed 256 / 512: 11.6-12.2 / 12.5; this rule: 11.1-11.7;
// shrinking claims towards the end of a list (guided self-scheduling, mode 2): 13.4 -- the atomics cost more than the tail.
#ifndef VSPG_WF_CLAIM_MODE
#define VSPG_WF_CLAIM_MODE 1   // 0: fixed kWfClaim; 1: per launch (above); 2: of the jobs LEFT
#endif
#ifndef VSPG_WF_CLAIM_DIV
#define VSPG_WF_CLAIM_DIV 4
#endif
#ifndef VSPG_WF_CLAIM_MIN
#define VSPG_WF_CLAIM_MIN 32
#endif
#ifndef VSPG_WF_CLAIM_MAX
#define VSPG_WF_CLAIM_MAX 256
#endif
struct WfClaim {
    unsigned next, end;
    bool exhausted;
    unsigned seen;   // the list position of this wavefront's latest claim (an estimate of the global cursor)
    unsigned seg, tried;  // the stretch this wavefront claims from, stretches it has found run out
};
VDEV WfClaim wf_claim_init(int segs) { return WfClaim{0u, 0u, false, 0u, blockIdx.x % (unsigned)segs, 0u}; }
VDEV unsigned wf_claim_size(unsigned n, unsigned seen) {
    if (VSPG_WF_CLAIM_MODE == 0) return (unsigned)kWfClaim;
    const unsigned waves = gridDim.x * (blockDim.x >> 6);
    const unsigned left = VSPG_WF_CLAIM_MODE == 2 ? (n > seen ? n - seen : 0u) : n;
    unsigned sz = left / (waves * (unsigned)VSPG_WF_CLAIM_DIV);
    sz = sz < (unsigned)VSPG_WF_CLAIM_MIN ? (unsigned)VSPG_WF_CLAIM_MIN : sz;
    return sz > (unsigned)VSPG_WF_CLAIM_MAX ? (unsigned)VSPG_WF_CLAIM_MAX : sz;
}
// the next chunk of list positions for a wavefront.  With ONE cursor per list the cursor is a hot word where lists are short and
// walks brief: a returning atomic per 73 jobs is 18 k of them per launch on the boundary scene, against ~88 per microsecond that one
// address serves (round 5: claims twice as large measured 3.5 % faster there, and cost tails elsewhere).  So the list may be dealt
// out to SEGS cursors (each on a 128-byte line of its own) in stripes of 16 chunks, stripe j to cursor
// j % segs: every cursor serves the whole length of the list, a chunk never straddles two stripes (the chunk size is fixed per
// launch), and a cursor's positions grow with it, so one that has passed the end has run out for good: `segs` of them found so = the
// list is exhausted.  A wavefront starts at its workgroup's cursor and moves round.  The host deals boundary scenes' job streams
// (k_wf_walk<Medium, kWfSegs>) out to 8 cursors (cloud-scene 4.47 -> 4.26 ms) and leaves the others at one: the all-filling cloud's wave took 8.5 % LONGER with 8 --
// its walks are long, the cursor is not its problem, and one front sweeping the image keeps the density fetches of the whole chip
// in one part of the volume.
template <unsigned SEGS>
VDEV bool wf_claim_refill(WfClaim &c, unsigned n, unsigned int *heads) {
    constexpr unsigned segs = SEGS;
    static_assert(VSPG_WF_CLAIM_MODE != 2, "striped cursors need a chunk size that is fixed per launch");
    const int lane = threadIdx.x & 63;
    const unsigned sz = wf_claim_size(n, c.seen), stripe = sz * 16u;
    auto pos_of = [&](unsigned base) { return ((base / stripe) * segs + c.seg) * stripe + base % stripe; };
    while (c.tried < segs) {
        // (a look before the returning atomic: cursors only grow, so one seen past the end is, and a wavefront at the end of the
        // list finds that out with plain loads instead of a round of atomics)
        unsigned pos = n;
        if (lane == 0 && (segs == 1u || pos_of(__hip_atomic_load(&heads[c.seg * 32u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < n)) pos = pos_of(atomicAdd(&heads[c.seg * 32u], sz));
        pos = __builtin_amdgcn_readfirstlane(pos);
        if (pos < n) {
            c.next = pos;
            c.end = pos + sz < n ? pos + sz : n;
            c.seen = c.end;
            return true;
        }
        c.seg = (c.seg + 1u) % segs;
        c.tried++;
    }
    c.exhausted = true;
    return false;
}
VDEV bool wf_claim(WfClaim &c, bool want, const unsigned int *list, unsigned n, unsigned int *head, unsigned *slot_out) {
    const int lane = threadIdx.x & 63;
    const unsigned long long need = __ballot(want);
    const unsigned cnt = (unsigned)__popcll(need);
    const unsigned rank = (unsigned)__popcll(need & ((1ull << lane) - 1ull));
    bool got = false;
    unsigned served = 0;
    while (served < cnt) {
        if (c.next >= c.end) {
            if (c.exhausted || !wf_claim_refill<1u>(c, n, head)) break;
        }
        const unsigned avail = c.end - c.next;
        const unsigned take = cnt - served < avail ? cnt - served : avail;
        if (want && !got && rank >= served && rank < served + take) {
            *slot_out = list[c.next + (rank - served)];
            got = true;
        }
        c.next += take;
        served += take;
    }
    return got;
}

// the same over a job STREAM without a list of its own: hands out positions (k_wf_walk maps them to its two lists)
template <unsigned SEGS>
VDEV bool wf_claim_index(WfClaim &c, bool want, unsigned n, unsigned int *head, unsigned *index_out) {
    const int lane = threadIdx.x & 63;
    const unsigned long long need = __ballot(want);
    const unsigned cnt = (unsigned)__popcll(need);
    const unsigned rank = (unsigned)__popcll(need & ((1ull << lane) - 1ull));
    bool got = false;
    unsigned served = 0;
    while (served < cnt) {
        if (c.next >= c.end) {
            if (c.exhausted || !wf_claim_refill<SEGS>(c, n, head)) break;
        }
        const unsigned avail = c.end - c.next;
        const unsigned take = cnt - served < avail ? cnt - served : avail;
        if (want && !got && rank >= served && rank < served + take) {
            *index_out = c.next + (rank - served);
            got = true;
        }
        c.next += take;
        served += take;
    }
    return got;
}

// ---- SampleT_maj_Resampling's traversal (media_sampleTMaj.h:178-247) + the reservoir callback (:691-719) --------------
template <class Medium>
__global__ __launch_bounds__(kWfBlock, kWfWalkWavesPerSimd) void k_wf_dist_walk(WfArgs a, int it) {
    const DScene &S = *a.scene;
    const Medium medium = wf_block_medium<Medium>(S);
    const WfPool &P = a.P;
    WfIter *I = &a.iters[it];
    const unsigned n = I->n_walk;
    WfCounters pc;
    pc.zero();
    WfClaim claim = wf_claim_init(1);
    bool active = false, result = false;  // result: a finished walk whose result is still in registers
    unsigned slot = 0;
    WalkState<Medium> w;
    VSPG_WF_STAT_DECL;
    w.iter = medium.empty_iter();
    w.sigma_maj = sp(0.f); w.seg_tMax = w.tMin = 0.f; w.T_maj = sp(1.f); w.u = 0.f; w.rng.state = w.rng.inc = 0; w.count = 0; w.in_seg = false;
    V3 ro = mk(0, 0, 0), rdn = mk(0, 0, 1);
    float scale = 1.f;
    int ch = 0;
    Sampler sampler;
    sampler.rng.state = sampler.rng.inc = 0;
    // reservoir state (:684-690)
    float weightSum = 0, sel_wi = 0;
    Spec trRatioEst = sp(1.f), beta_rs = sp(1.f), r_u_rs = sp(1.f), sel_num = sp(0.f), sel_den = sp(0.f);
    V3 sel_p = mk(0, 0, 0);
    while (true) {
        const unsigned n_idle = (unsigned)__popcll(__ballot(!active));
        if (n_idle == 64u || (n_idle >= (unsigned)a.walk_refill && !(claim.exhausted && claim.next >= claim.end))) {
            if (result) {  // park the finished walks' results
                wf_store_walk_result(P, slot, a.compact_results != 0, WfWalkResult{w.T_maj, trRatioEst, beta_rs, r_u_rs, sel_num, sel_den, weightSum, sel_wi, sel_p});
                P.store_rng(WF_RNG, slot, sampler.rng);
                result = false;
            }
            unsigned ns = 0;
            if (wf_claim(claim, !active, a.list_walk, n, &I->seg_head[0][0][0], &ns)) {
                slot = ns;
                w.iter = wf_load_iter(P, slot, medium, &ch);
                w.in_seg = false;
                w.T_maj = sp(1.f);
                w.u = P.f(WF_WU, slot);
                P.load_rng(WF_WRNG, slot, w.rng);
                w.count = 0;
                ro = P.v3(WF_RO, slot);
                rdn = P.v3(WF_RDN, slot);
                scale = P.f(WF_MSCALE, slot);
                P.load_rng(WF_RNG, slot, sampler.rng);
                weightSum = 0; sel_wi = 0;
                trRatioEst = beta_rs = r_u_rs = sp(1.f);
                sel_num = sel_den = sp(0.f);
                sel_p = mk(0, 0, 0);
                active = true;
            }
            if (__ballot(active) == 0ull) break;  // nothing in flight and the list has run out
        }
        // ---- one tracking step ------------------------------------------------------------------------------------
        float t = 0.f;
        int r = WALK_MOVED;
        VSPG_WF_STAT(0, 0, 1); VSPG_WF_STAT(0, 1, __popcll(__ballot(active))); VSPG_WF_STAT(0, 2, 1); VSPG_WF_STAT(0, 3, __popcll(__ballot(active)));
        if (claim.exhausted && claim.next >= claim.end) { VSPG_WF_STAT(0, 7, 1); VSPG_WF_STAT(0, 6, __popcll(__ballot(active))); }  // draining: no job left to claim
        if (active) {
            r = walk_advance<Medium, true>(w, ch, scale, &t);
        }
        // lanes that only moved (crossed into the next majorant cell, skipped an empty one) try again before the wavefront
        // enters the collision code, so that code runs with most lanes: up to walk_rounds tries while >= 8 lanes would sit idle
        for (int rr = 1; rr < a.walk_rounds; ++rr) {
            const bool again = active && r == WALK_MOVED;
            if (__popcll(__ballot(again)) < 8) break;
            VSPG_WF_STAT(0, 2, 1); VSPG_WF_STAT(0, 3, __popcll(__ballot(again)));
            if (again) r = walk_advance<Medium, true>(w, ch, scale, &t);
        }
        VSPG_WF_STAT(0, 4, __ballot(active && r == WALK_COLLISION) != 0ull); VSPG_WF_STAT(0, 5, __popcll(__ballot(active && r == WALK_COLLISION)));
        if (active && r == WALK_COLLISION) {
            w.T_maj = w.T_maj * fast_exp(w.sigma_maj * -(t - w.tMin));
            const V3 p = ro + rdn * t;
            const MediumProps mp = medium.sample_point(p);
            // the resampling callback (:691-719); it never stops the traversal
            pc.density_query();
            const Spec sigma_maj = w.sigma_maj, T_maj = w.T_maj;
            const Spec sigma_t = mp.sigma_t;
            const Spec sigma_n = clamp_zero(sigma_maj - sigma_t);
            const float wi = ch_of(wdiv(sigma_t, sigma_maj) * trRatioEst, ch);
            if (wi > 0) {
                weightSum += wi;
                if (sampler.get1d() < wdiv(wi, weightSum)) {
                    const float pdf = ch_of(T_maj, ch) * ch_of(sigma_t, ch);
                    sel_num = wdiv(beta_rs * T_maj * mp.sigma_s, pdf);
                    sel_den = wdiv(r_u_rs * T_maj * sigma_t, pdf);
                    sel_p = p;
                    sel_wi = wi;
                }
            }
            const float pdf = ch_of(T_maj, ch) * ch_of(sigma_n, ch);
            beta_rs = beta_rs * wdiv(T_maj * sigma_n, pdf);
            r_u_rs = r_u_rs * wdiv(T_maj * sigma_n, pdf);
            trRatioEst = trRatioEst * wdiv(sigma_n, sigma_maj);
            w.T_maj = sp(1.f);
            w.tMin = t;
        } else if (active && r == WALK_END) {
            active = false;
            result = true;
        }
    }
    VSPG_WF_STAT_FLUSH(0);
    wf_flush_counters(pc, a.counters);
}

// ---- the vertex of a pipeline path: li_segment_b with the NEE's shadow walk deferred ----------------------------------------
// Vertex set-up, the NEE's light sample and shadow-ray set-up (queued as a shadow-walk job, or added on the spot when no walk
// is needed), Russian roulette and the new direction (the guided vertex does all of it in li_vertex_guided_impl).  Returns what
// became of the path: finished here (film written), ended but listed for its outstanding NEE (stored, WFL_DEAD), or going on
// (`extra_out`: the flags of an outstanding shadow walk, for the caller's store).
enum { WFV_FINISHED = 0, WFV_DEAD_LISTED = 1, WFV_CONTINUES = 2 };
template <class Medium, bool GUIDED, bool TRAIN, class PC>
VDEV int wf_vertex(const WfArgs &a, const DScene &S, const Medium &medium, unsigned slot, PathState &st, Sampler &sampler, int ch, IsgSample &isg,
                   PC &pc, const Vertex &vx, uint32_t *extra_out, bool *shadow_out) {
    const WfPool &P = a.P;
    constexpr int G = Medium::kGrey;
    bool shadow = false;
    VertexCtx c;
    vertex_setup<false, true, Medium::kBnd>(S, st, vx, c);
    float survivalProb = 1.f;
    uint32_t extra = 0u;
    ShadowSetup ss;
    ss.status = 0;
    bool nee = false, cont = true;
    Spec beta_nee = st.beta;
    if constexpr (GUIDED) {
        {   // guided RR (:274-285): the pixel's contribution estimate, once the image-space buffer is ready
            int qx, qy;
            wf_pixel_of(slot, a.tilesX, &qx, &qy);
            st.guideRR = S.prm.rrguiding && S.contrib_ready;
            st.pce = st.guideRR ? S.contrib[(size_t)qy * S.xres + qx] : 0.f;
        }
        cont = li_vertex_guided_impl<Medium, true, true>(
            S, medium, st, ch, sampler, pc, vx.volume, vx.p, vx.g, c.si, c.intr, c.bsdf, nullptr, 0, [&](const auto &gd, bool use_gd) {
                nee = true;
                beta_nee = st.beta;
                ss = sample_Ld_begin(S, medium, ch, st.r_u, c.intr, &c.bsdf, sampler, pc, &gd, use_gd);
                return sp(0.f);
            });
    } else {
        survivalProb = vertex_pre(S, st, sampler, vx);
        if (S.prm.usenee && (vx.volume || c.bsdf.has_lobes)) {  // :479 / :833
            nee = true;
            ss = sample_Ld_begin(S, medium, ch, st.r_u, c.intr, &c.bsdf, sampler, pc);
        }
    }
    if (nee) {
        if (ss.status == 1) {
            const float tM = ss.tMax * len(ss.ld);
            const V3 rdn = normalize(ss.ld);
            const auto iter = medium.sample_ray(ss.lo, rdn, tM);
            if (iter.tMin >= iter.tMax) {
                ss.status = 2;  // the ray misses the medium's bounds: the walk would return T_maj = 1 at once
            } else {
                wf_store_iter(P, slot, iter, ch, WF_SIT, WF_SITP);
                P.set3(WF_SRDN, slot, rdn);
                P.set3(WF_SLO, slot, ss.lo);
                P.store_rng(WF_SWRNG, slot, ss.rng);
                P.f(WF_SWU2, slot) = ss.us;
                P.sets(WF_FHAT, slot, ss.f_hat);
                P.sets(WF_LSL, slot, ss.L);
                P.f(WF_PL, slot) = ss.p_l;
                P.f(wf_spdf_field<G>(), slot) = ss.scatterPDF;
                P.sets(WF_BNEE, slot, beta_nee);
                extra |= WFL_SHADOW_WALK | (ss.delta_light ? (uint32_t)WFL_DELTA : 0u);
                shadow = true;
            }
        }
        if (!shadow) {  // no walk to wait for: the estimate's tail (:1233-1251) and the addition (:483 / :836) here
            Spec Ld = sp(0.f);
            if (ss.status == 2)
                Ld = sample_Ld_end<G>(false, ss.delta_light, sp(1.f), sp(1.f), sp(1.f), sp(1.f), ch, ss.f_hat, ss.L, ss.p_l, ss.scatterPDF, st.r_u);
            else if (ss.status == 3)
                Ld = ss.Ld;
            st.L = st.L + beta_nee * Ld;
            if constexpr (TRAIN) pc.rec.add_scattered_direct_light(Ld);
        }
    }
    // ---- Russian roulette + new direction (:842-874 / :487-606); the guided vertex has done both ------------------
    if constexpr (!GUIDED) cont = vertex_tail<true, Medium::kBnd>(S, st, sampler, vx, c, survivalProb);
    if (!cont) {
        if (shadow) {  // its shadow walk is out: the slot stays listed for that one addition
            wf_rec_store(a, slot, pc.rec);
            wf_store_path<G>(P, slot, st, sampler, ch, isg, extra | FL_LIVE | WFL_DEAD);
            *shadow_out = true;
            return WFV_DEAD_LISTED;
        }
        wf_finish_path(a, slot, st, isg);
        pc.path();
        wf_rec_finish(a, slot, pc.rec);
        return WFV_FINISHED;
    }
    *extra_out = extra;
    *shadow_out = shadow;
    return WFV_CONTINUES;
}

// ---- the vertex kernel: [deferred NEE result] + segment end + vertex + the next segment's begin ----------------------------
// GUIDED: the vertex code is li_vertex_guided_impl -- cache init, NEE set-up with the guided PDF, Russian roulette, MIS / RIS
// direction sampling, the next segment's VSP -- with the product mixture in registers (GStoreReg; ~240 VGPRs: 2 waves per SIMD)
// TRAIN (a18, with GUIDED): the pass records path segments (the NEE's result reaches its record when it is added to L)
template <class Medium, bool GUIDED = false, bool TRAIN = false>
__global__ __launch_bounds__(kWfBlock, GUIDED ? VSPG_WF_VERTEX_WAVES_GUIDED : (Medium::kBnd == 1 ? VSPG_WF_VERTEX_WAVES_BND : VSPG_WF_VERTEX_WAVES)) void k_wf_vertex(WfArgs a, int it) {
    const DScene &S = *a.scene;
    const Medium medium = wf_block_medium<Medium>(S);
    const WfPool &P = a.P;
    constexpr int G = Medium::kGrey;
    WfIter *I = &a.iters[it], *In = &a.iters[it + 1];
    const unsigned n = I->n_active;
    const unsigned int *list_in = (it & 1) ? a.list_active2 : a.list_active;
    unsigned int *list_out = (it & 1) ? a.list_active : a.list_active2;
    WfCountersT<typename std::conditional<TRAIN, PathRecorder, NullRecorder>::type> pc;
    pc.zero();
    __shared__ unsigned int s_stage[3][kWfStageRounds * kWfBlock], s_cnt[3], s_gbase[3];
    if (threadIdx.x < 3) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const WfStage stA{s_stage[0], &s_cnt[0]}, stB{s_stage[1], &s_cnt[1]}, stC{s_stage[2], &s_cnt[2]};
    int round = 0;
    // Medium boundaries: a segment that needs no distance walk (a ray in vacuum, a ray that escapes, an all-zero majorant run)
    // has nothing to wait for -- its end, the vertex behind it and the next begin run in this same launch, unless the vertex's
    // NEE sent a shadow walk out (its result must reach L first).  A path through the reference's cloud scenes spends an
    // iteration less per boundary crossing: every vacuum segment (camera to the bounding sphere, sphere to ground or sky) used
    // to cost a launch of all three kernels and a trip of the record through HBM.  Round 4 ran such a chain in the lane that
    // began it, while the rest of its wavefront waited (lane utilisation 0.36 on the cloud scene); round 5 REGROUPS: the slot
    // goes onto a ring in the workgroup's LDS (the path is in its record anyway -- every link of a chain starts from there) and
    // the workgroup takes its next 256 paths from the ring whenever that holds as many, from the list otherwise, and drains the
    // ring at the end.  Per path the same links in the same order; which lane runs them is nobody's business.
    constexpr bool kChain = !GUIDED && Medium::kBnd == 1;
    constexpr bool kRegroup = kChain && VSPG_WF_DENSE_CLAIM != 0 && VSPG_WF_REGROUP != 0;
    // Ring size: a list round is taken only while fewer than one block's worth (B = 256) of links are SEEN waiting, and what is seen
    // lags the truth by at most the previous round's pushes (<= B): the true occupancy at a decision is < 2 B, after a list round
    // < 3 B, and a ring round never raises it; the entries being read plus a round's pushes stay under 4 B.  Sized at twice that
    // (4 B measured the same: 4.47 against 4.46 ms, although it leaves room for a fourth resident workgroup per CU).
    constexpr unsigned kRing = 8 * kWfBlock;
    enum { SRC_LIST = 0, SRC_RING = 1, SRC_RETRY = 2, SRC_DONE = 3 };
    __shared__ unsigned int s_ring[kRegroup ? kRing : 1], s_tail;
    if (threadIdx.x == 0) s_tail = 0;  // (ordered before the first push by the first round's barrier)
    unsigned ring_head = 0;            // thread 0's
    int prev_src = SRC_LIST;
    bool exhausted = false;
#if VSPG_WF_DENSE_CLAIM
    __shared__ unsigned int s_chunk[2][3];  // per round {source, base, count}
    for (;; ++round) {
        if (threadIdx.x == 0) {
            unsigned src = SRC_LIST, base = 0, cnt = kWfBlock;
            if constexpr (kRegroup) {
                const unsigned avail = *(volatile unsigned int *)&s_tail - ring_head;
                if (avail >= (unsigned)kWfBlock || (exhausted && avail > 0u)) {
                    src = SRC_RING;
                    base = ring_head;
                    cnt = avail < (unsigned)kWfBlock ? avail : (unsigned)kWfBlock;
                    ring_head += cnt;
                } else if (!exhausted) {
                    base = atomicAdd(&I->dense_head, (unsigned)kWfBlock);
                    if (base >= n) {
                        exhausted = true;
                        src = SRC_RETRY;
                    }
                } else {
                    // nothing on the ring as seen from here -- but a slower wavefront may still be pushing (last round's links):
                    // one empty round, whose barrier orders those pushes before the next look; empty twice in a row is empty
                    src = prev_src == SRC_RETRY ? SRC_DONE : SRC_RETRY;
                }
                prev_src = (int)src;
            } else {
                base = atomicAdd(&I->dense_head, (unsigned)kWfBlock);
                if (base >= n) src = SRC_DONE;
            }
            s_chunk[round & 1][0] = src; s_chunk[round & 1][1] = base; s_chunk[round & 1][2] = cnt;
        }
        __syncthreads();  // (two slots: the next claim is written while a slow wavefront may still be reading this one)
        const unsigned src = s_chunk[round & 1][0], base = s_chunk[round & 1][1], cnt = s_chunk[round & 1][2];
        if (src == SRC_DONE) break;
        bool have = false;
        unsigned slot = 0;
        int chain0 = 0;
        if (src == SRC_LIST) {
            have = base + threadIdx.x < n;
            if (have) slot = list_in[base + threadIdx.x];
        } else if (src == SRC_RING) {
            have = threadIdx.x < cnt;
            if (have) {
                const unsigned e = s_ring[(base + threadIdx.x) % kRing];
                slot = e & 0x0fffffffu;
                chain0 = (int)(e >> 28);
            }
        }
#else
    for (unsigned base = blockIdx.x * kWfBlock; base < n; base += gridDim.x * kWfBlock, ++round) {
        const bool have = base + threadIdx.x < n;
        unsigned slot = 0;
        const int chain0 = 0;
        if (have) slot = list_in[base + threadIdx.x];
#endif
        bool next = false, walk = false, shadow = false;  // next: the slot goes onto the next iteration's list
        bool requeue = false;
        int chain = chain0;
        for (; have; ++chain) {
            next = false;
            walk = false;
            bool again = false;
            PathState st;
            Sampler sampler;
            IsgSample isg;
            int ch;
            const uint32_t fl = wf_load_path<G>(P, slot, st, sampler, &ch, isg);
            if constexpr (TRAIN) wf_rec_load(a, slot, pc.rec);
            bool alive = false;
            // ---- the previous vertex's NEE, if its shadow walk was out (:483 / :836 from the estimate on) ----------------
            if (fl & WFL_SHADOW_WALK) {
                const WfShadowResult sr = wf_load_shadow_result(P, slot, a.compact_results != 0);
                const Spec Ld = sample_Ld_end<G>(true, (fl & WFL_DELTA) != 0, sr.T_ray, sr.r_l, sr.r_u,
                                                 sr.T_res, ch, P.sp3(WF_FHAT, slot), P.sp3(WF_LSL, slot), P.f(WF_PL, slot),
                                                 P.f(wf_spdf_field<G>(), slot), st.r_u);  // (the vertex code after the NEE leaves r_u alone)
                st.L = st.L + P.sp3(WF_BNEE, slot) * Ld;
                if constexpr (TRAIN) pc.rec.add_scattered_direct_light(Ld);  // :485 / :838 (add_scatter_data keeps the SCAT bit)
            }
            Vertex vx;
            int kind = EV_PASS;
            bool skipped = false;  // the segment ended on an interface (no BSDF): no vertex, the next segment begins behind it
            if (fl & WFL_DEAD) {
                // the path ended at the previous vertex; that addition was all it waited for
            } else {
                int px, py;
                wf_pixel_of(slot, a.tilesX, &px, &py);
                Isect si;
                si.hit = (fl & WFL_HIT) != 0;
                si.quad = P.i(WF_VXG, slot);
                si.t = P.f(WF_VXT, slot);
                si.p = P.v3(WF_VXP, slot);
                if (is_tri(si.quad)) {
                    const DTri &T = S.tris[tri_of(si.quad)];
                    si.n = V3{T.nx, T.ny, T.nz};
                } else if (is_sphere(si.quad)) {  // si.p = the object-space hit point (kSpherePrim)
                    si.n = sphere_interaction<false>(S.spheres[sphere_of(si.quad)], si.p).n;
                } else {
                    si.n = ld3(quad_at(si.quad).n);
                }
                vx.volume = false;
                vx.p = si.p;
                vx.g = 0;
                vx.quad = si.quad;
                vx.t = si.t;
                if (!(fl & WFL_NODIST)) {
                    // ---- sample_distance after the traversal (:721-802) ---------------------------------------------
                    const bool guide = (fl & WFL_GUIDE) != 0;
                    float weightSum = 0, sel_wi = 0, vrc = 0;
                    Spec trRatioEst = sp(1.f), beta_rs = sp(1.f), r_u_rs = sp(1.f), sel_num = sp(0.f), sel_den = sp(0.f), T_maj = sp(1.f);
                    V3 sel_p = mk(0, 0, 0);
                    if (!(fl & WFL_NOWALK)) {
                        const WfWalkResult wr = wf_load_walk_result(P, slot, a.compact_results != 0);
                        T_maj = wr.T_maj; weightSum = wr.weightSum; trRatioEst = wr.trRatioEst; beta_rs = wr.beta_rs; r_u_rs = wr.r_u_rs;
                        sel_p = wr.sel_p; sel_wi = wr.sel_wi; sel_num = wr.sel_num; sel_den = wr.sel_den;
                        vrc = P.f(WF_VRC, slot);
                        P.load_rng(WF_RNG, slot, sampler.rng);  // the traversal drew one sampler dimension per candidate (:702)
                    }
                    float sel_sTTr = sel_wi;
                    beta_rs = mul_tmaj_ratio<G>(beta_rs, T_maj, ch);
                    r_u_rs = mul_tmaj_ratio<G>(r_u_rs, T_maj, ch);
                    if (st.depth == 0 && S.tr_calc) {  // trBuffer->AddSample (:727-728, trbuffer.h:40-45); one sample per pixel per pass, passes in order
                        const size_t pix = (size_t)py * S.xres + px;
                        const int ns = S.tr_spp[pix] + 1;
                        S.tr_spp[pix] = ns;
                        const float alpha = wrcp((float)ns);
                        float *tb = S.tr_rgb + pix * 3;
                        tb[0] = (1.f - alpha) * tb[0] + alpha * trRatioEst.r;
                        tb[1] = (1.f - alpha) * tb[1] + alpha * trRatioEst.g;
                        tb[2] = (1.f - alpha) * tb[2] + alpha * trRatioEst.b;
                    }
                    const float trScalar = ch_of(trRatioEst, ch);
                    float surf_wi = trScalar;
                    if (guide && trScalar < 1 && trScalar > 0 && weightSum > 0) {
                        float volRatio = vrc * S.prm.vspmisratio + (1 - trScalar) * (1 - S.prm.vspmisratio);
                        float surfRatio = 1 - volRatio;
                        surf_wi = wdiv(surfRatio, volRatio) * weightSum;
                    }
                    weightSum += surf_wi;
                    if (weightSum != 0) {
                        bool selectSurface = false;
                        if (sampler.get1d() < wdiv(surf_wi, weightSum)) {
                            sel_wi = surf_wi;
                            sel_sTTr = trScalar;
                            sel_num = beta_rs;
                            sel_den = r_u_rs;
                            selectSurface = true;
                        }
                        const float factor = wdiv(weightSum * sel_sTTr, sel_wi);
                        bool term = false;
                        if (!selectSurface) {
                            if (st.depth == 0) {
                                isg.valid = true;
                                isg.surface_event = false;
                            }
                            if (st.depth++ >= S.prm.maxdepth) term = true;
                            else pc.volume_scatter();
                        }
                        if (!term) {
                            st.beta = st.beta * (sel_num * factor);
                            st.r_u = st.r_u * sel_den;
                            if (has_nan(st.beta) || has_nan(st.r_u) || has_inf(st.beta) || has_inf(st.r_u)) term = true;
                        }
                        if (term) kind = EV_TERMINATE;
                        else if (!selectSurface) {
                            if constexpr (TRAIN) {  // :798-802 (transmittanceWeight starts every path-loop iteration at 1, :317)
                                const Spec tw = sp(1.f) * ((sel_num * factor) / sel_den);
                                pc.rec.add_transmittance_weight(tw);
                                pc.rec.new_segment(sel_p, true);
                            }
                            kind = EV_SCATTER;
                            vx.volume = true;
                            vx.p = sel_p;
                            vx.g = medium.g;
                        }
                    }
                }
                if (kind == EV_TERMINATE || (!(fl & WFL_NODIST) && (!nonzero(st.beta) || !nonzero(st.r_u)))) {  // :343-344
                    alive = false;
                } else if (kind == EV_SCATTER) {
                    alive = true;
                } else {
                    const int pre = li_surface_pre<true, Medium::kBnd>(S, st, isg, pc, si, sp(1.f));
                    alive = pre != LI_END;
                    skipped = pre == LI_SKIP;
                }
            }
            // (the next segment's begin -- scene intersection, majorant pre-pass -- has ONE call site behind both ways into it: a
            // wavefront that holds crossings and vertices runs it once, not once per kind)
            bool begin = false;
            uint32_t extra = 0u;
            if (!alive) {
                wf_finish_path(a, slot, st, isg);
                pc.path();
                wf_rec_finish(a, slot, pc.rec);
            } else if (skipped) {  // :399-404: SkipIntersection moved the ray past the boundary; the path loop goes round again
                wf_rec_store(a, slot, pc.rec);
                if constexpr (GUIDED) {  // (gs.vsp_next: the previous vertex's estimate stays the next segment's -- no vertex in between)
                    wf_store_path<G>(P, slot, st, sampler, ch, isg, FL_LIVE);
                } else {
                    begin = true;
                }
                next = true;
            } else {
                // ---- li_segment_b: the vertex (wf_vertex), then the next segment's begin ---------------------------------------
                const int fate = wf_vertex<Medium, GUIDED, TRAIN>(a, S, medium, slot, st, sampler, ch, isg, pc, vx, &extra, &shadow);
                if (fate == WFV_DEAD_LISTED) {
                    next = true;
                } else if (fate == WFV_CONTINUES) {
                    wf_rec_store(a, slot, pc.rec);
                    if constexpr (GUIDED) {  // k_wf_begin takes it from here
                        P.f(WF_GSVSP, slot) = st.gs.vsp_next;
                        wf_store_path<G>(P, slot, st, sampler, ch, isg, extra | FL_LIVE);
                    } else {
                        begin = true;
                    }
                    next = true;
                }
            }
            if constexpr (!GUIDED) {
                if (begin) {
                    wf_segment_begin<Medium, GUIDED, !TRAIN>(a, S, medium, slot, st, sampler, ch, isg, pc, extra, &walk);
                    again = !walk && !shadow;  // (a crossing has no NEE: `shadow` is the vertex's)
                }
            }
            if (!kChain || !again || chain >= kWfChainMax) break;
            if constexpr (kRegroup) {  // the next link runs in whichever lane the ring hands it to
                requeue = true;
                next = false;
                break;
            }
        }
        if constexpr (kRegroup) {
            const unsigned long long m = __ballot(requeue);
            if (m != 0ull) {
                const int lane = threadIdx.x & 63;
                const int leader = __ffsll((long long)m) - 1;
                unsigned int rb = 0;
                if (lane == leader) rb = atomicAdd(&s_tail, (unsigned int)__popcll(m));
                rb = __shfl(rb, leader);
                if (requeue) s_ring[(rb + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))) % kRing] = slot | ((unsigned)(chain + 1) << 28);
            }
        }
        stA.push(next, slot);
        stB.push(walk, slot);
        stC.push(shadow, slot);
        if ((round % kWfStageRounds) == kWfStageRounds - 1)
            wf_flush_stages<3>({stA, stB, stC}, {list_out, a.list_walk, a.list_shadow}, {&In->n_active, &In->n_walk, &I->n_shadow}, s_gbase);
    }
    wf_flush_stages<3>({stA, stB, stC}, {list_out, a.list_walk, a.list_shadow}, {&In->n_active, &In->n_walk, &I->n_shadow}, s_gbase);
    wf_flush_counters(pc, a.counters);
}

// ---- vspsamplingmethod "nds" (NDS / NDS+) on grid media: the whole segment and the vertex in one dense kernel -----------------
// SampleT_maj_OpticalDepthSpace's callback ENDS the segment (a real collision scatters or absorbs), and it draws from the same
// RNG as the traversal: the walk cannot be cut out as a job the way the resampling routine's can.  So the segment runs in the
// lane (li_segment_a: intersection, distance sampling with its delta-tracking walk, the event), then the vertex (wf_vertex);
// what the pipeline still takes out of the lane is the NEE's ratio-tracking walk -- k_wf_shadow_walk, lanes regrouped over jobs
// -- whose result the next launch adds first thing, as in the resampling pipeline.  One launch per path-loop iteration.
template <class Medium, bool GUIDED = false, bool TRAIN = false>
__global__ __launch_bounds__(kWfBlock, 2) void k_wf_segment_vertex(WfArgs a, int it) {
    const DScene &S = *a.scene;
    const Medium medium = wf_block_medium<Medium>(S);
    const WfPool &P = a.P;
    constexpr int G = Medium::kGrey;
    const bool first = it == 0;
    WfIter *I = &a.iters[it], *In = &a.iters[it + 1];
    const unsigned n = first ? a.n_items : I->n_active;
    const unsigned int *list_in = (it & 1) ? a.list_active2 : a.list_active;
    unsigned int *list_out = (it & 1) ? a.list_active : a.list_active2;
    WfCountersT<typename std::conditional<TRAIN, PathRecorder, NullRecorder>::type> pc;
    pc.zero();
    __shared__ unsigned int s_stage[2][kWfStageRounds * kWfBlock], s_cnt[2], s_gbase[2];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const WfStage stA{s_stage[0], &s_cnt[0]}, stC{s_stage[1], &s_cnt[1]};
    int round = 0;
#if VSPG_WF_DENSE_CLAIM   // (here a chunk holds whole distance walks: their lengths differ by orders of magnitude)
    __shared__ unsigned int s_chunk[2];
    for (;; ++round) {
        if (threadIdx.x == 0) s_chunk[round & 1] = atomicAdd(&I->dense_head, (unsigned)kWfBlock);
        __syncthreads();
        const unsigned base = s_chunk[round & 1];
        if (base >= n) break;
#else
    for (unsigned base = blockIdx.x * kWfBlock; base < n; base += gridDim.x * kWfBlock, ++round) {
#endif
        const unsigned idx = base + threadIdx.x;
        bool next = false, shadow = false;
        unsigned slot = 0;
        if (idx < n) {
            slot = first ? idx : list_in[idx];
            int px, py;
            wf_pixel_of(slot, a.tilesX, &px, &py);
            PathState st;
            Sampler sampler;
            IsgSample isg;
            int ch = 0;
            bool run = false;
            if (first) {
                if (px < S.xres && py < S.yres) {
                    start_path(S, a.vsp_buf, a.vsp_ready, px, py, a.jump, sampler, st, &ch, isg);
                    if constexpr (TRAIN) {
                        wf_rec_load(a, slot, pc.rec);
                        pc.rec.reset();
                    }
                    run = true;
                }
            } else {
                const uint32_t fl = wf_load_path<G>(P, slot, st, sampler, &ch, isg);
                if constexpr (TRAIN) wf_rec_load(a, slot, pc.rec);
                if constexpr (GUIDED) st.gs.vsp_next = P.f(WF_GSVSP, slot);
                if (has_bnd<Medium::kBnd>(S) && st.depth == 0) st.vsp0 = (a.vsp_ready & VSP_READY) ? a.vsp_buf[(size_t)py * S.xres + px] : 0.5f;  // (see wf_segment_begin)
                if (Medium::kEmit && S.temperature) {  // the wavelength sample is the pixel sample's first number: drawn again rather than carried in the record
                    Sampler s0;
                    s0.start_pixel_sample(px, py, S.seed, a.jump);
                    st.lu = s0.get1d();
                }
                if (fl & WFL_SHADOW_WALK) {  // the previous vertex's NEE (:483 / :836 from the estimate on)
                    const WfShadowResult sr = wf_load_shadow_result(P, slot, a.compact_results != 0);
                    const Spec Ld = sample_Ld_end<G>(true, (fl & WFL_DELTA) != 0, sr.T_ray, sr.r_l, sr.r_u,
                                                     sr.T_res, ch, P.sp3(WF_FHAT, slot), P.sp3(WF_LSL, slot), P.f(WF_PL, slot),
                                                     P.f(wf_spdf_field<G>(), slot), st.r_u);
                    st.L = st.L + P.sp3(WF_BNEE, slot) * Ld;
                    if constexpr (TRAIN) pc.rec.add_scattered_direct_light(Ld);
                }
                if (fl & WFL_DEAD) {
                    wf_finish_path(a, slot, st, isg);
                    pc.path();
                    wf_rec_finish(a, slot, pc.rec);
                } else {
                    run = true;
                }
            }
            if (run) {
                Vertex vx;
                const int alive = li_segment_a<Medium, GUIDED, SEG_ANY>(S, medium, a.vsp_buf, a.vsp_ready, px, py, st, ch, sampler, isg, pc, vx);
                if (alive == LI_END) {
                    wf_finish_path(a, slot, st, isg);
                    pc.path();
                    wf_rec_finish(a, slot, pc.rec);
                } else if (alive == LI_SKIP) {  // a medium boundary was crossed (:399-404): the loop goes round again, no vertex
                    wf_rec_store(a, slot, pc.rec);
                    if constexpr (GUIDED) P.f(WF_GSVSP, slot) = st.gs.vsp_next;
                    wf_store_path<G>(P, slot, st, sampler, ch, isg, FL_LIVE);
                    next = true;
                } else {
                    uint32_t extra = 0u;
                    const int fate = wf_vertex<Medium, GUIDED, TRAIN>(a, S, medium, slot, st, sampler, ch, isg, pc, vx, &extra, &shadow);
                    if (fate == WFV_DEAD_LISTED) {
                        next = true;
                    } else if (fate == WFV_CONTINUES) {
                        wf_rec_store(a, slot, pc.rec);
                        if constexpr (GUIDED) P.f(WF_GSVSP, slot) = st.gs.vsp_next;
                        wf_store_path<G>(P, slot, st, sampler, ch, isg, extra | FL_LIVE);
                        next = true;
                    }
                }
            }
        }
        stA.push(next, slot);
        stC.push(shadow, slot);
        if ((round % kWfStageRounds) == kWfStageRounds - 1) wf_flush_stages<2>({stA, stC}, {list_out, a.list_shadow}, {&In->n_active, &I->n_shadow}, s_gbase);
    }
    wf_flush_stages<2>({stA, stC}, {list_out, a.list_shadow}, {&In->n_active, &I->n_shadow}, s_gbase);
    wf_flush_counters(pc, a.counters);
}

// ---- the shadow ray's ratio-tracking walk (SampleT_maj, media_sampleTMaj.h:49-117, with the callback of :1207-1232) -----
template <class Medium>
__global__ __launch_bounds__(kWfBlock, kWfShadowWavesPerSimd) void k_wf_shadow_walk(WfArgs a, int it) {
    const DScene &S = *a.scene;
    const Medium medium = wf_block_medium<Medium>(S);
    const WfPool &P = a.P;
    WfIter *I = &a.iters[it];
    const unsigned n = I->n_shadow;
    WfCounters pc;
    pc.zero();
    WfClaim claim = wf_claim_init(1);
    bool active = false, result = false;
    unsigned slot = 0;
    WalkState<Medium> w;
    w.iter = medium.empty_iter();
    w.sigma_maj = sp(0.f); w.seg_tMax = w.tMin = 0.f; w.T_maj = sp(1.f); w.u = 0.f; w.rng.state = w.rng.inc = 0; w.count = 0; w.in_seg = false;
    V3 ro = mk(0, 0, 0), rdn = mk(0, 0, 1);
    int ch = 0;
    Spec T_ray = sp(1.f), r_l = sp(1.f), r_u = sp(1.f), T_res = sp(1.f);
    while (true) {
        const unsigned n_idle = (unsigned)__popcll(__ballot(!active));
        if (n_idle == 64u || (n_idle >= (unsigned)a.walk_refill && !(claim.exhausted && claim.next >= claim.end))) {
            if (result) {
                wf_store_shadow_result(P, slot, a.compact_results != 0, WfShadowResult{T_ray, r_l, r_u, T_res});
                result = false;
            }
            unsigned ns = 0;
            if (wf_claim(claim, !active, a.list_shadow, n, &I->seg_head[1][0][0], &ns)) {
                slot = ns;
                w.iter = wf_load_iter(P, slot, medium, &ch, WF_SIT, WF_SITP);
                w.in_seg = false;
                w.T_maj = sp(1.f);
                w.u = P.f(WF_SWU2, slot);
                P.load_rng(WF_SWRNG, slot, w.rng);
                ro = P.v3(WF_SLO, slot);
                rdn = P.v3(WF_SRDN, slot);
                T_ray = r_l = r_u = sp(1.f);
                active = true;
            }
            if (__ballot(active) == 0ull) break;
        }
        float t = 0.f;
        int r = WALK_MOVED;
        if (active) {
            r = walk_advance<Medium, false>(w, ch, 1.f, &t);
        }
        // lanes that only moved (crossed into the next majorant cell, skipped an empty one) try again before the wavefront
        // enters the collision code, so that code runs with most lanes: up to walk_rounds tries while >= 8 lanes would sit idle
        for (int rr = 1; rr < a.walk_rounds; ++rr) {
            const bool again = active && r == WALK_MOVED;
            if (__popcll(__ballot(again)) < 8) break;
            if (again) r = walk_advance<Medium, false>(w, ch, 1.f, &t);
        }
        if (active && r == WALK_COLLISION) {
            w.T_maj = w.T_maj * fast_exp(w.sigma_maj * -(t - w.tMin));
            const V3 p = ro + rdn * t;
            const MediumProps mp = medium.sample_point(p);
            pc.shadow_query();
            // ratio tracking (:1207-1232)
            const Spec sigma_maj = w.sigma_maj, T_maj = w.T_maj;
            const float pdf = ch_of(T_maj, ch) * ch_of(sigma_maj, ch);
            const Spec sigma_n = medium.sigma_n(mp, sigma_maj);
            T_ray = T_ray * wdiv(T_maj * sigma_n, pdf);
            r_l = r_l * wdiv(T_maj * sigma_maj, pdf);
            r_u = r_u * wdiv(T_maj * sigma_n, pdf);
            const Spec Tr = wdiv(T_ray, avg(r_l + r_u));
            if (maxc(Tr) < 0.05f) {
                const float q = 0.75f;
                if (w.rng.uniform() < q) T_ray = sp(0.f);
                else T_ray = wdiv(T_ray, 1 - q);
            }
            if (!nonzero(T_ray)) {  // the callback stops the traversal: SampleT_maj returns 1
                T_res = sp(1.f);
                active = false;
                result = true;
            } else {
                w.T_maj = sp(1.f);
                w.tMin = t;
            }
        } else if (active && r == WALK_END) {
            T_res = w.T_maj;
            active = false;
            result = true;
        }
    }
    wf_flush_counters(pc, a.counters);
}

// ---- BOTH walks of a path-loop iteration in one persistent kernel (round 5) ---------------------------------------------------------
// The shadow walks of iteration i and the distance walks of iteration i + 1 are produced by the same vertex kernel and consumed by
// the same next one; as two kernels side by side (rounds 2-4) each drained on its own -- 30 % of a walk kernel's loop iterations ran
// in wavefronts with no job left to claim, at 15 lanes -- and each kept its own wavefronts resident.  Here they are ONE job stream
// (the shadow jobs first, then the distance jobs; one cursor): a lane runs one job of either kind, a wavefront refills its idle lanes
// from whatever is next, the majorant-cell advance (DDA step, exponential sample, the tentative-collision draw) is the same code
// for both kinds and only the collision's callback differs: the reservoir update (:691-719) or ratio tracking (:1207-1232).  A
// wavefront holds both kinds only around the point where the stream changes kind.  Per job nothing changes: same bits.
// Measured (same box, ms per 1080p wave, merged / two kernels side by side): cloud-scene 5.26 / 5.56, cloud-scene-guided 7.57 / 8.00,
// cloud-guided 9.37 / 9.53 -- but the all-filling cloud 8.75 / 8.19-8.41 and NanoVDB semantics 14.4 / 12.8: side by side the two
// kernels keep more wavefronts resident (four of the distance walk's + five of the shadow walk's compete for a SIMD's registers),
// which is what long, dense job lists want; short lists (boundary scenes: 7-9 iterations over thinning lists) and the guided
// pipelines (a third kernel, k_wf_begin, in the chain) want fewer launches and tails.  The host picks accordingly (wf_render_pass).
// it_d / it_s: the iteration whose distance / shadow jobs run, -1: none.
// SEGS: job cursors the stream is dealt out to (wf_claim_refill; a template parameter: as a kernel argument it cost every walk kernel
// scalar-register spills in its loop).
template <class Medium, int SEGS = 1>
__global__ __launch_bounds__(kWfBlock, kWfMergedWavesPerSimd) void k_wf_walk(WfArgs a, int it_d, int it_s) {
    const DScene &S = *a.scene;
    const Medium medium = wf_block_medium<Medium>(S);
    const WfPool &P = a.P;
    const unsigned n_s = it_s >= 0 ? a.iters[it_s].n_shadow : 0u, n_d = it_d >= 0 ? a.iters[it_d].n_walk : 0u;
    const unsigned n = n_s + n_d;
    unsigned int *const head = it_d >= 0 ? &a.iters[it_d].seg_head[2][0][0] : &a.iters[it_s].seg_head[3][0][0];  // (the control blocks are zeroed once per pass)
    WfCounters pc;
    pc.zero();
    WfClaim claim = wf_claim_init(SEGS);
    bool active = false, result = false, shadow = false;
    unsigned slot = 0;
    WalkState<Medium> w;
    w.iter = medium.empty_iter();
    w.sigma_maj = sp(0.f); w.seg_tMax = w.tMin = 0.f; w.T_maj = sp(1.f); w.u = 0.f; w.rng.state = w.rng.inc = 0; w.count = 0; w.in_seg = false;
    V3 ro = mk(0, 0, 0), rdn = mk(0, 0, 1);
    float scale = 1.f;
    int ch = 0;
    Sampler sampler;
    sampler.rng.state = sampler.rng.inc = 0;
    // one set of registers for the two kinds' running state: reservoir {trRatioEst, beta_rs, r_u_rs} | ratio tracking {T_ray, r_l, r_u}
    Spec xa = sp(1.f), xb = sp(1.f), xc = sp(1.f);
    float weightSum = 0, sel_wi = 0;
    Spec sel_num = sp(0.f), sel_den = sp(0.f);   // (shadow jobs: sel_num carries the residual T_maj)
    V3 sel_p = mk(0, 0, 0);
    VSPG_WF_STAT_DECL;
    while (true) {
        const unsigned n_idle = (unsigned)__popcll(__ballot(!active));
        if (n_idle == 64u || (n_idle >= (unsigned)a.walk_refill && !(claim.exhausted && claim.next >= claim.end))) {
            if (result) {  // park the finished walk's result
                if (shadow) {
                    wf_store_shadow_result(P, slot, a.compact_results != 0, WfShadowResult{xa, xb, xc, sel_num});
                } else {
                    wf_store_walk_result(P, slot, a.compact_results != 0, WfWalkResult{w.T_maj, xa, xb, xc, sel_num, sel_den, weightSum, sel_wi, sel_p});
                    P.store_rng(WF_RNG, slot, sampler.rng);
                }
                result = false;
            }
            unsigned idx = 0;
            if (wf_claim_index<(unsigned)SEGS>(claim, !active, n, head, &idx)) {
                shadow = idx < n_s;
                slot = shadow ? a.list_shadow[idx] : a.list_walk[idx - n_s];
                w.in_seg = false;
                w.T_maj = sp(1.f);
                w.count = 0;
                xa = xb = xc = sp(1.f);
                if (shadow) {
                    w.iter = wf_load_iter(P, slot, medium, &ch, WF_SIT, WF_SITP);
                    w.u = P.f(WF_SWU2, slot);
                    P.load_rng(WF_SWRNG, slot, w.rng);
                    ro = P.v3(WF_SLO, slot);
                    rdn = P.v3(WF_SRDN, slot);
                    scale = 1.f;
                } else {
                    w.iter = wf_load_iter(P, slot, medium, &ch);
                    w.u = P.f(WF_WU, slot);
                    P.load_rng(WF_WRNG, slot, w.rng);
                    ro = P.v3(WF_RO, slot);
                    rdn = P.v3(WF_RDN, slot);
                    scale = P.f(WF_MSCALE, slot);
                    P.load_rng(WF_RNG, slot, sampler.rng);
                    weightSum = 0; sel_wi = 0;
                    sel_num = sel_den = sp(0.f);
                    sel_p = mk(0, 0, 0);
                }
                active = true;
            }
            if (__ballot(active) == 0ull) break;  // nothing in flight and the stream has run out
        }
        // ---- one tracking step ------------------------------------------------------------------------------------
        float t = 0.f;
        int r = WALK_MOVED;
        const int cap = shadow ? 0x7fffffff : 10000;  // (the 10 000-collision guard belongs to the resampling routine only, media_sampleTMaj.h:216-219)
        VSPG_WF_STAT(0, 0, 1); VSPG_WF_STAT(0, 1, __popcll(__ballot(active))); VSPG_WF_STAT(0, 2, 1); VSPG_WF_STAT(0, 3, __popcll(__ballot(active)));
        if (claim.exhausted && claim.next >= claim.end) { VSPG_WF_STAT(0, 7, 1); VSPG_WF_STAT(0, 6, __popcll(__ballot(active))); }  // draining: no job left to claim
        if (active) r = walk_advance<Medium, true>(w, ch, scale, &t, cap);
        for (int rr = 1; rr < a.walk_rounds; ++rr) {
            const bool again = active && r == WALK_MOVED;
            if (__popcll(__ballot(again)) < 8) break;
            VSPG_WF_STAT(0, 2, 1); VSPG_WF_STAT(0, 3, __popcll(__ballot(again)));
            if (again) r = walk_advance<Medium, true>(w, ch, scale, &t, cap);
        }
        VSPG_WF_STAT(0, 4, __ballot(active && r == WALK_COLLISION) != 0ull); VSPG_WF_STAT(0, 5, __popcll(__ballot(active && r == WALK_COLLISION)));
        if (active && r == WALK_COLLISION) {
            w.T_maj = w.T_maj * fast_exp(w.sigma_maj * -(t - w.tMin));
            const V3 p = ro + rdn * t;
            const MediumProps mp = medium.sample_point(p);
            const Spec sigma_maj = w.sigma_maj, T_maj = w.T_maj;
            if (shadow) {  // ratio tracking (:1207-1232): xa = T_ray, xb = r_l, xc = r_u
                pc.shadow_query();
                const float pdf = ch_of(T_maj, ch) * ch_of(sigma_maj, ch);
                const Spec sigma_n = medium.sigma_n(mp, sigma_maj);
                xa = xa * wdiv(T_maj * sigma_n, pdf);
                xb = xb * wdiv(T_maj * sigma_maj, pdf);
                xc = xc * wdiv(T_maj * sigma_n, pdf);
                const Spec Tr = wdiv(xa, avg(xb + xc));
                if (maxc(Tr) < 0.05f) {
                    const float q = 0.75f;
                    if (w.rng.uniform() < q) xa = sp(0.f);
                    else xa = wdiv(xa, 1 - q);
                }
                if (!nonzero(xa)) {  // the callback stops the traversal: SampleT_maj returns 1
                    sel_num = sp(1.f);
                    active = false;
                    result = true;
                } else {
                    w.T_maj = sp(1.f);
                    w.tMin = t;
                }
            } else {  // the resampling callback (:691-719): xa = trRatioEst, xb = beta_rs, xc = r_u_rs; it never stops the traversal
                pc.density_query();
                const Spec sigma_t = mp.sigma_t;
                const Spec sigma_n = clamp_zero(sigma_maj - sigma_t);
                const float wi = ch_of(wdiv(sigma_t, sigma_maj) * xa, ch);
                if (wi > 0) {
                    weightSum += wi;
                    if (sampler.get1d() < wdiv(wi, weightSum)) {
                        const float pdf = ch_of(T_maj, ch) * ch_of(sigma_t, ch);
                        sel_num = wdiv(xb * T_maj * mp.sigma_s, pdf);
                        sel_den = wdiv(xc * T_maj * sigma_t, pdf);
                        sel_p = p;
                        sel_wi = wi;
                    }
                }
                const float pdf = ch_of(T_maj, ch) * ch_of(sigma_n, ch);
                xb = xb * wdiv(T_maj * sigma_n, pdf);
                xc = xc * wdiv(T_maj * sigma_n, pdf);
                xa = xa * wdiv(sigma_n, sigma_maj);
                w.T_maj = sp(1.f);
                w.tMin = t;
            }
        } else if (active && r == WALK_END) {
            if (shadow) sel_num = w.T_maj;  // the residual majorant transmittance (T_res)
            active = false;
            result = true;
        }
    }
    VSPG_WF_STAT_FLUSH(0);
    wf_flush_counters(pc, a.counters);
}

VSPG_NS_END  // namespace vspg
