// One wave's share of a trace launch: the closest-hit rays / shadow rays of 64 queue slots, and the counters a launch keeps.
// What the queue kernels of path_tracer.hip (k_trace_closest, k_trace_shadow, k_trace_fused) are made of.  Device-only, free of host headers.
#pragma once
#include "trace.h"
#include "trace_quad.h"
#include "pt_state.h"

namespace tr {

namespace {

// ---------------------------------------------------------------------------------------------------
// The closest-hit rays of queue slots base .. base + 63 (path_tracer.glsl:387-403), one wave: trace, store the hit records of
// the paths.  Every lane of the wave calls this; the traversal re-deals the last rays of the chunk over quads (trace_quad.h).
template <bool COUNT>
TR_DEV void closest_lane(const SceneView& sv, const PtParams& P, const PathBuffers& pb, int bounce, const uint* queue, uint qi, uint n, int* lds_stack,
                         const QuadCtx& qc, TraceStats& st, int& overflow, uint& max_vis, uint& rays) {
    bool valid = qi < n;
    uint id = 0;
    u4 misc = {0, 0, 0, 1};
    f4 o = F4(0), d = F4(0);
    // the ray is fetched together with the path's flags, not behind them: one round trip less before the traversal starts, and a
    // queue holds live paths only (the flag matters at bounce 0, where the ids are all launch ids)
    TL(const unsigned long long tl_chunk = tl_now();)
    if (valid) { id = queue ? queue[qi] : qi + P.id_offset; misc = pb.misc[id]; o = pb.org_pdf[id]; d = pb.dir_reg[id]; valid = !(misc.w & 1u); }
    TL(tl_data_arrived(); tl_misc(qc.tl, 3, tl_now() - tl_chunk);)
    // payload.random_seed of this trace: k_raygen stored the seed of bounce 0, every closest-hit trace advances it once
    // (path_tracer.glsl:387-403; DESIGN.md on the any-hit hash)
    for (int b = 0; b < bounce; ++b) pcg(misc.x);
    HitRecord hit;
    const bool include_lights = !(P.opt.hide_lights && bounce == 0);
    const uint before = st.nodes;
#if TR_QUAD_SWITCH > 0
    trace_closest_wave4<0, COUNT>(sv, valid, F3(o), F3(d), bounce == 0 ? 0.0f : P.opt.min_ray_dist, __builtin_huge_valf(), include_lights, misc.x,
                                  lds_stack, qc, hit, st, overflow);
#else
    if (valid) trace_closest4<0, COUNT>(sv, F3(o), F3(d), bounce == 0 ? 0.0f : P.opt.min_ray_dist, __builtin_huge_valf(), include_lights,
                                        misc.x, lds_stack, hit, st, overflow);
#endif
    if (COUNT) st.cnodes += st.nodes - before;      // every lane: in the quad tail lane 0 of a quad counts for the quad's ray
    if (!valid) return;
    if (COUNT) {
        const uint vis = st.nodes - before;
        max_vis = max(max_vis, vis);
        if (vis > 100000u && vis > atomicMax(&pb.counters[CNT_MAXVIS], vis)) {   // debugging aid: remember a pathological ray
            float* dbg = reinterpret_cast<float*>(pb.counters + CNT_DBG);
            dbg[0] = o.x; dbg[1] = o.y; dbg[2] = o.z; dbg[3] = d.x; dbg[4] = d.y; dbg[5] = d.z; dbg[6] = (float)bounce; dbg[7] = (float)id;
            dbg[8] = o.w; dbg[9] = d.w;
        }
    }
    pb.hit[id] = make_int4(hit.instance_id, hit.primitive_id, __float_as_int(hit.u), __float_as_int(hit.v));
    rays++;
    TL(tl_misc(qc.tl, 5, tl_now() - tl_chunk);)
}

// LDS and global scratch of a wave's quad tail
TR_DEV QuadCtx make_quad_ctx(int* s_stack, int* s_owner, const PathBuffers& pb TL(, uint* s_tl = nullptr)) {
    QuadCtx qc;
    TL(qc.tl = s_tl ? s_tl + (threadIdx.x >> 6) * TL_WORDS : nullptr;)
    const uint wave = threadIdx.x >> 6;
    qc.wave_stack = s_stack + (threadIdx.x & ~63u);
    qc.owner_tab = s_owner + wave * TR_OWNER_WORDS;
    qc.spill = pb.qspill + ((size_t)blockIdx.x * (KB / 64) + wave) * (16u * TR_QSPILL);
    return qc;
}

// The shadow rays of slots base .. base + 63 of the bounce's shadow queue, one wave: contrib *= shadow_ray(...)
// (path_tracer.glsl:35-52, 462-463) and add_demodulated_color of the result.  Every lane of the wave calls this.
// `contrib` and `lobes` are fetched behind the traversal, by the rays that turn out visible: four registers the traversal loop (64 in all for
// the shadow kernel) does not have to carry, for one more round trip at the end of a chunk of 64 rays.
template <bool COUNT, typename CONTRIB, typename LOBES>
TR_DEV void shadow_ray(const SceneView& sv, const PtParams& P, const PathBuffers& pb, bool valid, f4 o, f4 d, CONTRIB&& contrib, LOBES&& lobes, int* lds_stack, const QuadCtx& qc,
                       TraceStats& st, int& overflow, uint& rays) {
#if TR_QUAD_SWITCH > 0 && !defined(TR_NO_SHADOW_QUADS)
    float vis = trace_shadow_wave4<COUNT>(sv, valid, F3(o), F3(d), P.opt.min_ray_dist, o.w, lds_stack, qc, st, overflow);
#else
    float vis = 1.0f;
    if (valid) vis = trace_shadow4<COUNT>(sv, F3(o), F3(d), P.opt.min_ray_dist, o.w, lds_stack, st, overflow);
#endif
    if (!valid) return;
    const uint id = __float_as_uint(d.w);
    if (vis != 0.0f) {
        const f4 c = contrib();
        // clamp_contribution_mul on the occluded radiance (path_tracer.glsl:462-463): c.w = luminance before visibility
        float m = c.w * vis;
        if (c.w > 0.0f && m > P.opt.indirect_clamping) vis *= P.opt.indirect_clamping / m;
        const f3 radiance = F3(c.x * vis, c.y * vis, c.z * vis);
        const f2 w = lobes();
        // add_demodulated_color; a zero weight adds exactly nothing, so that target is left alone
        if (w.x != 0.0f) { f4 d4 = pb.diffuse[id]; d4.x += radiance.x * w.x; d4.y += radiance.y * w.x; d4.z += radiance.z * w.x; pb.diffuse[id] = d4; }
        if (w.y != 0.0f) { f4 r4 = pb.reflection[id]; r4.x += radiance.x * w.y; r4.y += radiance.y * w.y; r4.z += radiance.z * w.y; pb.reflection[id] = r4; }
    }
    rays++;
}
template <bool COUNT>
TR_DEV void shadow_lane(const SceneView& sv, const PtParams& P, const PathBuffers& pb, uint qi, uint n, int* lds_stack, const QuadCtx& qc,
                        TraceStats& st, int& overflow, uint& rays) {
    const bool valid = qi < n;
    f4 o = F4(0), d = F4(0);
    if (valid) { o = pb.sh_org_tmax[qi]; d = pb.sh_dir_id[qi]; }
    shadow_ray<COUNT>(sv, P, pb, valid, o, d, [&] { return pb.sh_contrib[qi]; }, [&] { return pb.sh_lobes[qi]; }, lds_stack, qc, st, overflow, rays);
}

template <bool COUNT>
TR_DEV void flush_trace_counters(const PtParams& P, const PathBuffers& pb, int overflow, int overflow_tag, uint closest_rays, uint shadow_rays,
                                 TraceStats st, uint max_vis) {
    if (overflow) { pb.counters[CNT_OVERFLOW] = 1; pb.counters[CNT_DBG + 12] = (uint)overflow_tag; }
    if (!P.count_work) return;
    for (int off = 32; off > 0; off >>= 1) {
        closest_rays += __shfl_xor(closest_rays, off); shadow_rays += __shfl_xor(shadow_rays, off);
        if (COUNT) {
            st.nodes += __shfl_xor(st.nodes, off); st.tris += __shfl_xor(st.tris, off); st.alpha += __shfl_xor(st.alpha, off);
            st.ph_node += __shfl_xor(st.ph_node, off); st.ph_tri += __shfl_xor(st.ph_tri, off); st.ph_node16 += __shfl_xor(st.ph_node16, off);
            st.ph_node8 += __shfl_xor(st.ph_node8, off); st.lv_node16 += __shfl_xor(st.lv_node16, off);
            st.ph_qnode += __shfl_xor(st.ph_qnode, off); st.ph_qtri += __shfl_xor(st.ph_qtri, off); st.cnodes += __shfl_xor(st.cnodes, off);
            for (int b = 0; b < 8; ++b) st.ph_hist[b] += __shfl_xor(st.ph_hist[b], off);
            st.maxsp = max(st.maxsp, (uint)__shfl_xor(st.maxsp, off)); max_vis = max(max_vis, (uint)__shfl_xor(max_vis, off));
        }
    }
    if ((threadIdx.x & 63) == 0) {
        add64(pb.counters, CNT_CLOSEST, closest_rays);
        add64(pb.counters, CNT_SHADOWRAYS, shadow_rays);
        if (COUNT) {
            add64(pb.counters, CNT_NODES, st.nodes); add64(pb.counters, CNT_TRIS, st.tris); add64(pb.counters, CNT_ALPHA, st.alpha);
            add64(pb.counters, CNT_PH_NODE, st.ph_node); add64(pb.counters, CNT_PH_TRI, st.ph_tri); add64(pb.counters, CNT_PH_NODE16, st.ph_node16);
            add64(pb.counters, CNT_PH_NODE8, st.ph_node8); add64(pb.counters, CNT_LV_NODE16, st.lv_node16);
            add64(pb.counters, CNT_PH_QNODE, st.ph_qnode); add64(pb.counters, CNT_PH_QTRI, st.ph_qtri); add64(pb.counters, CNT_CNODES, st.cnodes);
            for (int b = 0; b < 8; ++b) add64(pb.counters, CNT_PH_HIST + 2 * b, st.ph_hist[b]);
            atomicMax(&pb.counters[CNT_MAXSP], st.maxsp); atomicMax(&pb.counters[CNT_MAXVIS], max_vis);
        }
    }
}

}  // namespace

}  // namespace tr
