// vspg_fast.hip -- the tolerance-mode instantiations of the path kernels (vspg_arith.h): compiled twice by csrc/Makefile,
//   vspg_fast1.o  -DVSPG_ARITH=1  weight-only quotients as a * v_rcp_f32(b); trajectories stay the oracle's
//   vspg_fast2.o  -DVSPG_ARITH=2  -fno-hip-fp32-correctly-rounded-divide-sqrt: every division / square root at 2.5 ulp, native log / sin / cos
// What they serve (vspg_renderer_set_arithmetic refuses the rest): the unguided rectangle-scene instantiations of the workgroup kernel
// (k_render_wave_wg3 -- the headline workload) and the unguided resampling pipeline over GridMedium (cloud, cloud-scene).
#include <hip/hip_runtime.h>

#include "vspg_trace.h"
#include "vspg_wg3.h"
#include "vspg_wf_launch.h"

static_assert(VSPG_ARITH == 1 || VSPG_ARITH == 2, "vspg_fast.hip is the fast-arithmetic translation unit: build it with -DVSPG_ARITH=1 or 2");

#if VSPG_ARITH == 1
#define VSPG_FAST_SYM(name) vspg_arith1_##name
#else
#define VSPG_FAST_SYM(name) vspg_arith2_##name
#endif

extern "C" int VSPG_FAST_SYM(wg3)(const void *wg3_launch) {
    return vspg::wg3_launch_unguided(*static_cast<const vspg::Wg3Launch *>(wg3_launch));
}
extern "C" int VSPG_FAST_SYM(wf_grid)(const void *wf_launch, int grey) {
    return vspg::wf_dispatch_unguided<false>(*static_cast<const vspg::WfLaunch *>(wf_launch), grey != 0);
}
extern "C" int VSPG_FAST_SYM(trace)(const void *trace_launch) {
    return vspg::trace_launch_unguided(*static_cast<const vspg::TraceLaunch *>(trace_launch));
}
