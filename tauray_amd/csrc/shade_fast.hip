// k_shade for the option set of the reference's command line, compiled a second time with the arithmetic a Vulkan implementation
// is allowed to use for the reference's GLSL - what the reference itself runs on:
//   * `/` at 2.5 ulp and sqrt / inversesqrt through the hardware's v_sqrt_f32 / v_rsq_f32 (SPIR-V precision requirements of the
//     Vulkan specification: division 2.5 ULP, inversesqrt 2 ULP, sqrt inherited from 1 / inversesqrt) instead of the correctly
//     rounded sequences of the default build (-fno-hip-fp32-correctly-rounded-divide-sqrt);
//   * sin, cos, pow through v_sin_f32 / v_cos_f32 / v_exp_f32 / v_log_f32 (common.h: TR_SHADE_NATIVE_MATH).
// Multiply-adds do not contract here either (measured worth nothing; it keeps the instances of the kernel bit-identical).
// Traversal, the triangle test and every other kernel stay in path_tracer.hip at IEEE fp32: hits are bit-exact against the
// oracle, radiance is compared within the tolerance DESIGN.md section 3 states.  The IEEE k_shade of the same option set has
// 11 640 instructions, this one 5 900: the C library's sin / cos / pow and the correctly rounded division and square-root
// sequences are code the reference never executed.  sponza_teapots: 1.22 -> 0.97 ms per frame (profiles/r3/shade_arithmetic_ab.txt:
// native sin / cos / pow alone 1.04, division / sqrt alone 1.15, contraction alone 1.21).
// trhip_pt_set_shading_arithmetic(pt, 1) or TRHIP_SHADE_FAST=0 select the IEEE instances (path_tracer.hip).
#ifndef TR_SHADE_LIBM_MATH     // A/B builds: only the division / sqrt / contraction flags, sin / cos / pow from the C library
#define TR_SHADE_NATIVE_MATH 1
#endif
#include "pt_kernels.h"

namespace tr {

void launch_shade_fast(bool cli, bool count, bool last, uint blocks, hipStream_t stream, const SceneView& sv, const PtParams& P, const PathBuffers& pb, int bounce,
                       const uint* queue, uint* bc, uint* next_queue) {
    // (counting + command-line set: the general counting instance, as in the IEEE build)
#define TR_LAUNCH(...) hipLaunchKernelGGL((k_shade<__VA_ARGS__>), dim3(blocks), dim3(KB), 0, stream, sv, P, pb, bounce, queue, bc, next_queue)
    if (cli) {
        if (last) { if (count) TR_LAUNCH(true, true, SpecCli); else TR_LAUNCH(false, true, SpecCli); }
        else { if (count) TR_LAUNCH(true, false, SpecCli); else TR_LAUNCH(false, false, SpecCli); }
    } else {
        if (last) { if (count) TR_LAUNCH(true, true, SpecGeneral); else TR_LAUNCH(false, true, SpecGeneral); }
        else { if (count) TR_LAUNCH(true, false, SpecGeneral); else TR_LAUNCH(false, false, SpecGeneral); }
    }
#undef TR_LAUNCH
}

}  // namespace tr

#if TR_SHADE_TIMELINE
// the phase timeline of the shade kernels of this translation unit (shade_timeline.h): read out / reset, instrument builds only
extern "C" int trhip_debug_shade_timeline(unsigned long long* out, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(tr::g_shade_tl), sizeof(unsigned long long) * tr::STL_WORDS) != hipSuccess) return 1;
    if (reset) { static unsigned long long zero[tr::STL_WORDS]; if (hipMemcpyToSymbol(HIP_SYMBOL(tr::g_shade_tl), zero, sizeof(zero)) != hipSuccess) return 1; }
    return 0;
}
#endif
