// vspg_wf_launch.h -- the host side of one wavefront-pipeline pass: the kernel launches of vspg_wavefront.h in order.
//
// The pipeline's kernels are the bulk of the library's device code (dense kernels x {medium layout, grey, guided, training,
// medium boundaries}): they are instantiated in translation units of their own -- vspg_wf_grid.hip (GridMedium) and
// vspg_wf_nvdb.hip (NanoVDB semantics) -- which `make -j` compiles beside vspg_capi.hip.  vspg_capi.hip prepares a pass (buffers,
// grid sizes, streams) as a plain WfLaunch and calls wf_dispatch_grid / wf_dispatch_nvdb; those pick the instantiation and run
// wf_run_pass below.  Return value: 0, a hipError_t, or WF_E_NOT_DRAINED.
#pragma once
#include <hip/hip_runtime.h>

#include "vspg_wavefront.h"

VSPG_NS_BEGIN

struct WfLaunch {
    WfArgs a;
    unsigned dense, walk, swalk, mwalk;   // grid sizes: dense kernels, distance walk, shadow walk, the merged walk (k_wf_walk)
    bool serial;                   // everything on one stream (VSPG_WF_SERIAL)
    int segs;                      // job cursors of k_wf_walk's stream (1 or kWfSegs: wf_claim_refill)
    bool merged;                   // both walks of an iteration in ONE kernel (k_wf_walk: the default; VSPG_WF_MERGED=0: two kernels side by side)
    bool bnd;                      // the scene has medium boundaries: the BND instantiations, iterations until the list runs dry
    bool nds;                      // vspsamplingmethod "nds": segment + vertex in one dense kernel
    bool emit;                     // ... over a medium with a temperature grid: the NDS kernel's emissive instantiations
    int maxdepth, base_iters, max_iters;
    hipStream_t s, s2;
    hipEvent_t ev_vertex, ev_shadow;
};
constexpr int WF_E_NOT_DRAINED = -1000;

int wf_dispatch_grid(const WfLaunch &L, bool guided, bool train, bool grey);
int wf_dispatch_nvdb(const WfLaunch &L, bool guided, bool train, bool grey);

#define WFCHK(expr)                               \
    do {                                          \
        const hipError_t e_ = (expr);             \
        if (e_ != hipSuccess) return (int)e_;     \
    } while (0)

// Medium: the dense kernels' instantiation (its kBnd says whether the boundary code is compiled in); WalkMedium: the two walk
// kernels' -- the grey layout whenever the medium's coefficients are bitwise grey, never the boundary flavour (a walk job is a ray
// segment inside the medium: boundaries are the dense kernels' business).
// NDS_ONLY: only the "nds" branch is instantiated (the emissive media: their other kernels would be the plain medium's over again).
// NO_NDS: only the resampling pipeline is instantiated (the fast-arithmetic translation units, vspg_fast.hip).
template <class Medium, bool GUIDED, bool TRAIN, class WalkMedium, bool NDS_ONLY = false, bool NO_NDS = false>
int wf_run_pass(const WfLaunch &L) {
    const WfArgs &a = L.a;
    const hipStream_t s = L.s, s2 = L.s2;
    const bool bnd = L.bnd, serial = L.serial;
    // (medium boundaries) has the list of iteration `it` run dry?  One small read-back per iteration past the ones every pass needs.
    // Once the host reads an iteration's list length anyway it sizes that iteration's launches by it: the tail iterations of a
    // boundary scene hold a few thousand paths, and a full grid of persistent workgroups (each stages the majorant grid into its LDS
    // before it finds the job list empty) costs 50-60 us per walk launch whatever the list holds.  One lane per job at most: the
    // tail is bound by the length of a walk, not by throughput.
    unsigned dense = L.dense, walk = L.walk, swalk = L.swalk, mwalk = L.mwalk;
    auto list_empty = [&](int it, bool *empty) -> int {
        unsigned int na = 0;
        WFCHK(hipMemcpyAsync(&na, &a.iters[it].n_active, sizeof na, hipMemcpyDeviceToHost, s));
        WFCHK(hipStreamSynchronize(s));
        *empty = na == 0;
        const unsigned blocks = (na + (unsigned)kWfBlock - 1u) / (unsigned)kWfBlock;
        dense = blocks < L.dense ? (blocks ? blocks : 1u) : L.dense;
        walk = blocks < L.walk ? (blocks ? blocks : 1u) : L.walk;
        swalk = blocks < L.swalk ? (blocks ? blocks : 1u) : L.swalk;
        mwalk = blocks < L.mwalk ? (blocks ? blocks : 1u) : L.mwalk;
        return 0;
    };
    if (!NO_NDS && L.nds) {  // NDS / NDS+: segment + vertex in one kernel, the NEE's walk regrouped
      if constexpr (!NO_NDS) {
        for (int it = 0; it < L.max_iters; ++it) {
            if (bnd && it > L.base_iters) {
                bool empty = false;
                if (const int rc = list_empty(it, &empty)) return rc;
                if (empty) break;
            }
            hipLaunchKernelGGL((k_wf_segment_vertex<Medium, GUIDED, TRAIN>), dim3(dense), dim3(kWfBlock), 0, s, a, it);
            if (bnd || it < L.maxdepth) hipLaunchKernelGGL(k_wf_shadow_walk<WalkMedium>, dim3(swalk), dim3(kWfBlock), 0, s, a, it);
        }
      }
    } else if constexpr (!NDS_ONLY) {
        hipLaunchKernelGGL((k_wf_start<Medium, GUIDED, TRAIN>), dim3(L.dense), dim3(kWfBlock), 0, s, a);
        if (L.merged) {
            // vertex(it) hands out the shadow jobs of iteration it AND (itself or through k_wf_begin) the distance jobs of it + 1; the
            // next vertex kernel consumes both: one walk kernel between them, one stream
            int shadow_out = -1;  // the iteration whose shadow jobs wait for a walk
            for (int it = 0; it < L.max_iters; ++it) {
                if (bnd && it > L.base_iters) {
                    bool empty = false;
                    if (const int rc = list_empty(it, &empty)) return rc;
                    if (empty) break;
                }
                if (L.segs > 1) hipLaunchKernelGGL((k_wf_walk<WalkMedium, kWfSegs>), dim3(mwalk), dim3(kWfBlock), 0, s, a, it, shadow_out);
                else hipLaunchKernelGGL((k_wf_walk<WalkMedium, 1>), dim3(mwalk), dim3(kWfBlock), 0, s, a, it, shadow_out);
                hipLaunchKernelGGL((k_wf_vertex<Medium, GUIDED, TRAIN>), dim3(dense), dim3(kWfBlock), 0, s, a, it);
                shadow_out = -1;
                if (bnd || it < L.maxdepth) {
                    if constexpr (GUIDED) hipLaunchKernelGGL((k_wf_begin<Medium, GUIDED, TRAIN>), dim3(dense), dim3(kWfBlock), 0, s, a, it + 1);
                    shadow_out = it;
                }
            }
            if (shadow_out >= 0) {
                if (L.segs > 1) hipLaunchKernelGGL((k_wf_walk<WalkMedium, kWfSegs>), dim3(mwalk), dim3(kWfBlock), 0, s, a, -1, shadow_out);
                else hipLaunchKernelGGL((k_wf_walk<WalkMedium, 1>), dim3(mwalk), dim3(kWfBlock), 0, s, a, -1, shadow_out);
            }
        } else {
        for (int it = 0; it < L.max_iters; ++it) {
            if (bnd && it > L.base_iters) {
                bool empty = false;
                if (const int rc = list_empty(it, &empty)) return rc;
                if (empty) break;
            }
            hipLaunchKernelGGL(k_wf_dist_walk<WalkMedium>, dim3(walk), dim3(kWfBlock), 0, s, a, it);
            if (it > 0 && !serial) WFCHK(hipStreamWaitEvent(s, L.ev_shadow, 0));
            hipLaunchKernelGGL((k_wf_vertex<Medium, GUIDED, TRAIN>), dim3(dense), dim3(kWfBlock), 0, s, a, it);
            if (bnd || it < L.maxdepth) {
                // (guided: the next segments begin BEFORE the shadow walk starts -- launched after it, the dense begin kernel crawled in
                // the slots the persistent walk left over and the next distance walk waited for it)
                if constexpr (GUIDED) hipLaunchKernelGGL((k_wf_begin<Medium, GUIDED, TRAIN>), dim3(dense), dim3(kWfBlock), 0, s, a, it + 1);
                if (!serial) {
                    WFCHK(hipEventRecord(L.ev_vertex, s));
                    WFCHK(hipStreamWaitEvent(s2, L.ev_vertex, 0));
                }
                hipLaunchKernelGGL(k_wf_shadow_walk<WalkMedium>, dim3(swalk), dim3(kWfBlock), 0, s2, a, it);
                if (!serial) WFCHK(hipEventRecord(L.ev_shadow, s2));
            }
        }
        if (bnd && !serial) WFCHK(hipStreamWaitEvent(s, L.ev_shadow, 0));  // (the last shadow walk finds an empty list; the caller's stream still waits for it)
    }
    }
    WFCHK(hipGetLastError());
    if (bnd) {
        bool empty = true;
        if (const int rc = list_empty(L.max_iters, &empty)) return rc;
        if (!empty) return WF_E_NOT_DRAINED;
    }
    return 0;
}

// one medium layout's instantiations: {grey, chromatic} x {unguided, guided, guided + training} x {no boundaries, boundaries},
// + the NDS kernel's for media with a temperature grid
// the pipeline's kernels read a 16^3 majorant grid from their block's LDS copy with LDS instructions (GridMediumT's MAJLDS;
// -DVSPG_WF_MAJ_LDS=0: through the generic pointer, as rounds 2-4 did -- for A/B builds)
#ifndef VSPG_WF_MAJ_LDS
#define VSPG_WF_MAJ_LDS 1
#endif
template <bool NVDB> constexpr bool kWfMajLds = !NVDB && VSPG_WF_MAJ_LDS != 0;
template <bool NVDB>
int wf_dispatch(const WfLaunch &L, bool guided, bool train, bool grey) {
#define VSPG_WF_CASE(GREY, BNDV)                                                                                        \
    do {                                                                                                                \
        using M = GridMediumT<NVDB, GREY, BNDV, false, kWfMajLds<NVDB>>;                                                          \
        using WM = GridMediumT<NVDB, GREY, -1, false, kWfMajLds<NVDB>>;                                                           \
        if (guided && train) return wf_run_pass<M, true, true, WM>(L);                                                  \
        if (guided) return wf_run_pass<M, true, false, WM>(L);                                                          \
        return wf_run_pass<M, false, false, WM>(L);                                                                     \
    } while (0)
    if (L.emit && L.nds) {  // (a medium with a temperature grid is never run on the grey layout: its emission is not grey)
#define VSPG_WF_EMIT_CASE(BNDV)                                                                                        \
    do {                                                                                                                \
        using M = GridMediumT<NVDB, false, BNDV, true, kWfMajLds<NVDB>>;                                                          \
        using WM = GridMediumT<NVDB, false, -1, false, kWfMajLds<NVDB>>;                                                          \
        if (guided && train) return wf_run_pass<M, true, true, WM, true>(L);                                            \
        if (guided) return wf_run_pass<M, true, false, WM, true>(L);                                                    \
        return wf_run_pass<M, false, false, WM, true>(L);                                                               \
    } while (0)
        if (L.bnd) VSPG_WF_EMIT_CASE(1);
        VSPG_WF_EMIT_CASE(0);
#undef VSPG_WF_EMIT_CASE
    }
    if (L.bnd) {
        if (grey) VSPG_WF_CASE(true, 1);
        VSPG_WF_CASE(false, 1);
    }
    if (grey) VSPG_WF_CASE(true, 0);
    VSPG_WF_CASE(false, 0);
#undef VSPG_WF_CASE
}

// the unguided resampling pipeline only: {grey, chromatic} x {no boundaries, boundaries} (vspg_fast.hip)
template <bool NVDB>
int wf_dispatch_unguided(const WfLaunch &L, bool grey) {
    if (L.nds || L.emit) return WF_E_NOT_DRAINED - 1;  // (the caller routes those to the exact instantiations)
#define VSPG_WF_UCASE(GREY, BNDV) return wf_run_pass<GridMediumT<NVDB, GREY, BNDV, false, kWfMajLds<NVDB>>, false, false, GridMediumT<NVDB, GREY, -1, false, kWfMajLds<NVDB>>, false, true>(L)
    if (L.bnd) {
        if (grey) VSPG_WF_UCASE(true, 1);
        VSPG_WF_UCASE(false, 1);
    }
    if (grey) VSPG_WF_UCASE(true, 0);
    VSPG_WF_UCASE(false, 0);
#undef VSPG_WF_UCASE
}

VSPG_NS_END  // namespace vspg

// The fast-arithmetic translation units (vspg_fast.hip compiled with -DVSPG_ARITH=1 / 2, vspg_arith.h) behind plain C entry points:
// their types live in an inline namespace of their own, so the launch records cross as untyped pointers (same struct, same source).
extern "C" {
int vspg_arith1_wg3(const void *wg3_launch);
int vspg_arith2_wg3(const void *wg3_launch);
int vspg_arith1_wf_grid(const void *wf_launch, int grey);
int vspg_arith2_wf_grid(const void *wf_launch, int grey);
int vspg_arith1_trace(const void *trace_launch);
int vspg_arith2_trace(const void *trace_launch);
}
