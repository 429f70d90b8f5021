// The shading program of ONE option set: k_raygen and k_shade (shade_kernel.h) with the stage's options pinned by -DTR_SPEC_* macros,
// the way the reference compiles its options into the ray-tracing pipeline as #defines (src/path_tracer_stage.cc:30-116,
// src/rt_camera_stage.cc, compiled at run time through src/shader_source.cc).  specialize.cc builds this translation unit - through
// hipRTC when a stage with a new option set is first rendered, or ahead of time into the kernel cache (trhip_pt_precompile; the
// reference's presets and the sets the tests use are compiled by __graft_entry__.build()) - and launches the kernels through the
// module API.  The arithmetic follows the flags: IEEE fp32 without contraction, or - TR_SHADE_NATIVE_MATH and
// -fno-hip-fp32-correctly-rounded-divide-sqrt - what Vulkan asks of the reference's GLSL (shade_fast.hip).
// An instance renders the same bits as the general kernel of the same arithmetic (tests/test_specialization.py).
#include "shade_kernel.h"

#ifndef TR_SPEC_COUNT
#define TR_SPEC_COUNT 0
#endif

namespace tr {

#if TR_SPEC_PROGRAM == 1
// Ray generation is a program of its own: it is compiled at IEEE fp32 whatever the arithmetic of the shading kernels (camera rays,
// like the traversal, are bit-equal to the oracle's in both modes).
extern "C" __global__ __launch_bounds__(KB) void trhip_spec_raygen(SceneView sv, PtParams P, PathBuffers pb) { raygen_paths<SpecMacros>(sv, P, pb); }
#else
extern "C" __global__ __launch_bounds__(KB, TR_SHADE_WAVES) void trhip_spec_shade(SceneView sv, PtParams P, PathBuffers pb, int bounce, const uint* queue, uint* bc,
                                                                                  uint* next_queue) {
    shade_bounce<TR_SPEC_COUNT != 0, false, SpecMacros>(sv, P, pb, bounce, queue, bc, next_queue);
}

extern "C" __global__ __launch_bounds__(KB, TR_SHADE_LAST_WAVES) void trhip_spec_shade_last(SceneView sv, PtParams P, PathBuffers pb, int bounce, const uint* queue,
                                                                                            uint* bc, uint* next_queue) {
    shade_bounce<TR_SPEC_COUNT != 0, true, SpecMacros>(sv, P, pb, bounce, queue, bc, next_queue);
}

#endif

}  // namespace tr
