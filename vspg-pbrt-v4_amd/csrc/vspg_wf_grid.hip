// vspg_wf_grid.hip -- the wavefront pipeline's kernels for GridMedium ("uniformgrid"), instantiated here so that `make -j` compiles
// them beside vspg_capi.hip (vspg_wf_launch.h).
#include "vspg_wf_launch.h"
VSPG_NS_BEGIN
int wf_dispatch_grid(const WfLaunch &L, bool guided, bool train, bool grey) { return wf_dispatch<false>(L, guided, train, grey); }
VSPG_NS_END  // namespace vspg
