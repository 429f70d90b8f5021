// A small frame's paths kept resident: one kernel walks every path of the launch through all its bounces.
//
// The queue schedule (path_tracer.hip) runs a bounce as two launches - trace, shade - and a launch ends with its slowest ray: a
// frame of four bounces is a chain of ten launches, each as long as the longest of ALL its rays.  For a full 1080p frame that is
// what keeps 256 CUs busy (2 M rays per launch, compaction between bounces, four lanes to fill the tails).  For the strip of a
// frame one GPU of eight renders (1920 x 136: 261 k paths, 4 080 waves - fewer than the chip holds at once) it is all latency: every
// launch lasts about as long as its longest ray and the frame takes the SUM over the bounces of the longest ray of each, 0.90 ms
// where the work is 0.5 ms.  Here a wave keeps its 64 paths from the camera ray to the last bounce:
//     per bounce   closest hit (trace_closest_wave4) -> shade_path -> the bounce's shadow ray (trace_shadow_wave4)
// with no queue, no compaction, no atomics and nothing to wait for between waves, so a wave lasts as long as ITS longest rays and
// the frame as long as its slowest wave.  Paths that end leave their lanes idle (the wave still runs as long as its longest path),
// which is the price of not compacting and the reason this schedule is only chosen while all paths of the launch are resident at
// once (PtStage::render: up to 4 waves per SIMD at the 128 registers the kernel is built for).
//
// Same arithmetic, same bits as the queue schedule: ray generation and the resolve stay launches of their own (IEEE fp32 kernels of
// path_tracer.hip), the traversal here is the queue kernels' code with its divisions pinned to correct rounding (common.h div_rn),
// shade_path is the body of k_shade, and a path's own order of additions (emission of bounce b, then the light sample of b) is kept
// because the lane that shades a path also traces its shadow ray.  Device-only, free of host headers.
#pragma once
#include "shade_kernel.h"
#include "trace_lanes.h"

namespace tr {

namespace {

#ifndef TR_FRAME_WAVES
#define TR_FRAME_WAVES 4      // waves per SIMD the resident kernel is built for (128 VGPRs): 262 144 paths on 256 CUs
#endif

template <bool COUNT, typename S>
TR_DEV void frame_resident(const SceneView& sv, const PtParams& P_, const PathBuffers& pb, uint* bc, int* s_stack, int* s_owner) {
    PtParams P = P_;
    S::pin(P);
    const QuadCtx qc = make_quad_ctx(s_stack, s_owner, pb);
    const uint n = P.n_ids;
    TraceStats st = {};
    uint closest_rays = 0, shadow_rays = 0, max_vis = 0, surf = 0;
    int overflow = 0;
    const uint wave_id = (blockIdx.x * KB + threadIdx.x) >> 6, n_waves = (gridDim.x * KB) >> 6;
    bool first = true;
    while (true) {
        uint base = 0;
        if (first) base = wave_id * 64u;      // persistent waves, as in the queue kernels: the first chunk by wave id, later ones from a cursor
        else {
            if (n <= n_waves * 64u) break;
            if ((threadIdx.x & 63) == 0) base = n_waves * 64u + atomicAdd(&bc[BC_CUR_CLOSEST], 64u);
            base = __shfl(base, 0);
        }
        first = false;
        if (base >= n) break;
        const uint qi = base + (threadIdx.x & 63);
        const uint id = qi + P.id_offset;
        u4 misc = {0, 0, 0, 1};
        if (qi < n) misc = pb.misc[id];
        bool live = qi < n && !(misc.w & 1u);
        for (int bounce = 0; bounce < P.opt.max_bounces; ++bounce) {
            if (__ballot(live) == 0) break;
            closest_lane<COUNT>(sv, P, pb, bounce, nullptr, qi, n, s_stack + threadIdx.x, qc, st, overflow, max_vis, closest_rays, live);
            ShadeOut o;
            if (live) shade_path<COUNT, false, S>(sv, P, pb, bounce, id, misc, o, surf);
            // the bounce's shadow ray, by the lane that shaded the path: contrib *= shadow_ray(...) lands in the path's sums before the
            // next bounce adds to them, as in the queue schedule (shadow(b) rides with closest(b + 1), k_shade(b + 1) follows)
            shadow_ray<COUNT>(sv, P, pb, o.want_shadow, F4(o.sh_o, o.sh_tmax), F4(o.sh_d, __uint_as_float(id)), F4(o.sh_c, o.sh_lum), [&] { return o.sh_w; },
                              s_stack + threadIdx.x, qc, st, overflow, shadow_rays);
            live = o.alive;
        }
    }
    flush_trace_counters<COUNT>(P, pb, overflow, 4000, closest_rays, shadow_rays, st, max_vis);
    if (COUNT && P.count_work) {
        for (int off = 32; off > 0; off >>= 1) surf += __shfl_xor(surf, off);
        if ((threadIdx.x & 63) == 0) add64(pb.counters, CNT_SURF, surf);
    }
}

template <bool COUNT, typename S = SpecGeneral>
__global__ __launch_bounds__(KB, TR_FRAME_WAVES) void k_frame_resident(SceneView sv, PtParams P, PathBuffers pb, uint* bc) {
    __shared__ int s_stack[TR_STACK_WORDS];
    __shared__ int s_owner[(KB / 64) * TR_OWNER_WORDS];
    frame_resident<COUNT, S>(sv, P, pb, bc, s_stack, s_owner);
}

}  // namespace

}  // namespace tr
