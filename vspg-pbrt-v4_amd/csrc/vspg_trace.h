// vspg_trace.h -- k_trace_paths: the --debugstart replay (integrators.cpp:77-95) -- chosen (pixel, sample) pairs run through the
// per-lane path loop, radiance and segment count returned per path.  The parity tests compare these with the oracle's replay.
#pragma once
#include "vspg_path.h"

VSPG_NS_BEGIN

constexpr int kTraceBlock = 256;
static_assert(kTraceBlock == kGuideBlock, "the guiding scratch in LDS is sized for kTraceBlock threads");

// guided RR (:274-285): the pixel's contribution estimate, once the image-space buffer is ready
VDEV void load_contribution_estimate(const DScene &S, int px, int py, PathState &st) {
    st.guideRR = S.prm.rrguiding && S.contrib_ready;
    st.pce = st.guideRR ? S.contrib[(size_t)py * S.xres + px] : 0.f;
}


template <class Medium, bool GUIDED>
__global__ __launch_bounds__(kTraceBlock) void k_trace_paths(const DScene *__restrict__ Sp, const float *__restrict__ vsp_buf,
                                                        int vsp_ready, int n, const int32_t *__restrict__ pixel_xy,
                                                        const int32_t *__restrict__ sample_index, float *__restrict__ out_L,
                                                        int32_t *__restrict__ out_seg) {
    const DScene &S = *Sp;
    stage_scene_lds(S);
    __syncthreads();
    int i = blockIdx.x * kTraceBlock + threadIdx.x;
    if (i >= n) return;
    const Medium medium = MediumMaker<Medium>::make(S, S.majorant);
    int px = pixel_xy[2 * i], py = pixel_xy[2 * i + 1];
    PathCounters pc = {0, 0, 0, 0, 0};
    Sampler sampler;
    PathState st;
    IsgSample isg;
    int ch;
    float *glds = nullptr;
    if constexpr (GUIDED) glds = guide_lds();
    start_path(S, vsp_buf, vsp_ready, px, py, sample_index[i], sampler, st, &ch, isg);
    if constexpr (GUIDED) load_contribution_estimate(S, px, py, st);
    while (li_segment<Medium, GUIDED>(S, medium, vsp_buf, vsp_ready, px, py, st, ch, sampler, isg, pc, glds, kTraceBlock)) {
    }
    Spec L = finish_radiance(st.L);
    out_L[3 * i] = L.r; out_L[3 * i + 1] = L.g; out_L[3 * i + 2] = L.b;
    if (out_seg) out_seg[i] = (int32_t)pc.segments;
}


// the unguided replays behind one call (vspg_capi.hip launches them directly; the fast-arithmetic translation units export this)
struct TraceLaunch {
    const DScene *dscene;
    const float *vsp;
    int vsp_ready, n;
    const int32_t *pixel_xy, *sample_index;
    float *out_L;
    int32_t *out_seg;
    hipStream_t stream;
    int grid;   // 1: GridMedium ("uniformgrid"), 0: HomogeneousMedium
};
inline int trace_launch_unguided(const TraceLaunch &T) {
    const dim3 grid((unsigned)((T.n + kTraceBlock - 1) / kTraceBlock)), block(kTraceBlock);
    if (T.grid) hipLaunchKernelGGL((k_trace_paths<GridMedium, false>), grid, block, 0, T.stream, T.dscene, T.vsp, T.vsp_ready, T.n, T.pixel_xy, T.sample_index, T.out_L, T.out_seg);
    else hipLaunchKernelGGL((k_trace_paths<HomogeneousMedium, false>), grid, block, 0, T.stream, T.dscene, T.vsp, T.vsp_ready, T.n, T.pixel_xy, T.sample_index, T.out_L, T.out_seg);
    return (int)hipGetLastError();
}

VSPG_NS_END  // namespace vspg
