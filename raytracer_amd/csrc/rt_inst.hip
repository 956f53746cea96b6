// rt_inst.hip -- explicit instantiations of the tracing kernels (rt_kernels.h).  Compiled once per
//   RT_INST_CLS  = scene class (0 general, 1 flat, 2 general with nested leaves)
//   RT_INST_ST   = true / false: the counting variant or not
//   RT_INST_PART = 0 render_kernel variants, 1 wf_level_kernel variants, 2 the batch kernels (color_at, probe, shade_rays)
// so that the library's ~50 large kernels build in parallel (Makefile).
#include "rt_kernels.h"

namespace rtk {
#define RT_DEFINE_RENDER(...) template __global__ void __VA_ARGS__(RT_RENDER_ARGS);
#define RT_DEFINE_WF(...) template __global__ void __VA_ARGS__(RT_WF_ARGS);
#if RT_INST_PART == 0
#if RT_INST_CLS == 1
RT_RENDER_LIST_FLAT(RT_DEFINE_RENDER, RT_INST_ST)
#else
RT_RENDER_LIST_COMPOUND(RT_DEFINE_RENDER, RT_INST_ST, RT_INST_CLS)
#endif
#elif RT_INST_PART == 1
#if RT_INST_CLS == 1
RT_WF_LIST_FLAT(RT_DEFINE_WF, RT_INST_ST)
#else
RT_WF_LIST_COMPOUND(RT_DEFINE_WF, RT_INST_ST, RT_INST_CLS)
#endif
#else
#define RT_MISC_X(ID, ARGS) template __global__ void ID ARGS;
RT_MISC_LIST(RT_INST_CLS)
#endif
}  // namespace rtk
