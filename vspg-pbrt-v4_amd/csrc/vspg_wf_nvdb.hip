// vspg_wf_nvdb.hip -- the wavefront pipeline's kernels for NanoVDBMedium semantics (64^3 majorants), instantiated here so that
// `make -j` compiles them beside vspg_capi.hip (vspg_wf_launch.h).
#include "vspg_wf_launch.h"
VSPG_NS_BEGIN
int wf_dispatch_nvdb(const WfLaunch &L, bool guided, bool train, bool grey) { return wf_dispatch<true>(L, guided, train, grey); }
VSPG_NS_END  // namespace vspg
