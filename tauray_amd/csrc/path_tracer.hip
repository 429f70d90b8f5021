// Wavefront path tracer for gfx950: the MI355X-native replacement of Tauray's path_tracer_stage
// dispatch (src/path_tracer_stage.cc:118-147 -> shader/path_tracer.rgen + hit/miss shader table).
//
// One vkCmdTraceRaysKHR pass becomes, per sample:
//   k_raygen -> k_trace_closest(0) -> k_shade(0) -> [k_trace_fused(b): closest(b) + shadow(b - 1)] -> k_shade(b) ... -> k_resolve
// Path state lives in HBM as float4/uint4 SoA streams; live paths are compacted with wave ballots into id queues
// between bounces so later bounces launch dense waves.  Launches are sized for the worst case and read the live
// counts from device memory: there is no host round trip inside a frame.  A frame is cut into four lanes (slices of
// the path ids with their own queues) that run this loop concurrently on their own streams (PtStage::render).
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "pt_kernels.h"
#include "resolve_kernel.h"
#include "specialize.h"

namespace tr {

namespace {

// Register budgets of the traversal kernels (waves per SIMD): the loops are latency-bound, so the shadow kernel runs at
// the full 8 waves (<= 64 VGPRs); the closest-hit kernel sorts four children and spills below 80 VGPRs, 6 waves win.
#ifndef TR_CLOSEST_WAVES
#define TR_CLOSEST_WAVES 6
#endif
#ifndef TR_SHADOW_WAVES
#define TR_SHADOW_WAVES 8
#endif

// Persistent waves: each wave takes its first 64 rays by wave id (no atomic: avoids a burst of ~7000 dequeues on one word
// at kernel start) and later chunks from a device-side cursor, which starts past the statically assigned range.
// SOLO only names the instance launched while detailed timing serialises the frame, so that a profiler lists the
// kernel running alone (the roofline measurement) apart from the overlapped launches of normal frames.
// WIDE: the scene has RGBA16 textures.  The instances for scenes without (nearly all) pin SceneView::wide_textures to 0, and the
// any-hit alpha test compiles to the one-dword fetch it always was (texture.h; profiles/r4/texture_format_ab.txt).
template <bool COUNT, bool SOLO, bool WIDE = true>
__global__ __launch_bounds__(KB, TR_CLOSEST_WAVES) void k_trace_closest(SceneView sv, PtParams P, PathBuffers pb, int bounce, const uint* queue,
                                                      uint* bc) {
    if (!WIDE) sv.wide_textures = 0;
    __shared__ int s_stack_rows[TR_STACK_WORDS];
    int* const s_stack = s_stack_rows + TR_STACK_ROW0;     // row -1 exists (LaneStack, trace.h)
    __shared__ int s_owner[(KB / 64) * TR_OWNER_WORDS];
    TL(__shared__ uint s_tl[(KB / 64) * TL_WORDS]; for (uint i = threadIdx.x; i < (KB / 64) * TL_WORDS; i += KB) s_tl[i] = 0; __syncthreads();)
    const QuadCtx qc = make_quad_ctx(s_stack, s_owner, pb TL(, s_tl));
    const uint n = queue ? bc[BC_QUEUE] : P.n_ids;
    TraceStats st = {};
    uint rays = 0, max_vis = 0;
    int overflow = 0;
    const uint wave_id = (blockIdx.x * KB + threadIdx.x) >> 6, n_waves = (gridDim.x * KB) >> 6;
    bool first = true;
    while (true) {
        uint base = 0;
        if (first) base = wave_id * 64u;
        else {
            if (n <= n_waves * 64u) break;   // the static first chunks covered the queue: no cursor traffic at all
            if ((threadIdx.x & 63) == 0) base = n_waves * 64u + atomicAdd(&bc[BC_CUR_CLOSEST], 64u);
            base = __shfl(base, 0);
        }
        first = false;
        if (base >= n) break;
        closest_lane<COUNT>(sv, P, pb, bounce, queue, base + (threadIdx.x & 63), n, s_stack + threadIdx.x, qc, st, overflow, max_vis, rays);
    }
    flush_trace_counters<COUNT>(P, pb, overflow, 1000 + bounce, rays, 0u, st, max_vis);
    TL(__syncthreads(); for (uint i = threadIdx.x; i < (KB / 64) * TL_WORDS; i += KB) if (s_tl[i]) atomicAdd(&g_timeline[i % TL_WORDS], (unsigned long long)s_tl[i]);)
}

template <bool COUNT, bool WIDE = true>
__global__ __launch_bounds__(KB, TR_SHADOW_WAVES) void k_trace_shadow(SceneView sv, PtParams P, PathBuffers pb, uint* bc) {
    if (!WIDE) sv.wide_textures = 0;
    __shared__ int s_stack_rows[TR_STACK_WORDS];
    int* const s_stack = s_stack_rows + TR_STACK_ROW0;     // row -1 exists (LaneStack, trace.h)
    __shared__ int s_owner[(KB / 64) * TR_OWNER_WORDS];
    const QuadCtx qc = make_quad_ctx(s_stack, s_owner, pb);
    const uint n = bc[BC_SHADOW];
    TraceStats st = {};
    uint rays = 0;
    int overflow = 0;
    const uint wave_id = (blockIdx.x * KB + threadIdx.x) >> 6, n_waves = (gridDim.x * KB) >> 6;
    bool first = true;
    while (true) {
        uint base = 0;
        if (first) base = wave_id * 64u;
        else {
            if (n <= n_waves * 64u) break;
            if ((threadIdx.x & 63) == 0) base = n_waves * 64u + atomicAdd(&bc[BC_CUR_SHADOW], 64u);
            base = __shfl(base, 0);
        }
        first = false;
        if (base >= n) break;
        shadow_lane<COUNT>(sv, P, pb, base + (threadIdx.x & 63), n, s_stack + threadIdx.x, qc, st, overflow, rays);
    }
    flush_trace_counters<COUNT>(P, pb, overflow, 2000, 0u, rays, st, 0u);
}

// Closest-hit rays of bounce b and the shadow rays of bounce b - 1 in one launch.  The two are independent (different
// state arrays), and in the lane schedule a launch lasts as long as its slowest wave: one launch with one tail instead of
// two launches with two.  Chunk g of the launch is a closest-hit chunk while g < chunks_c (the longer rays go first), a
// shadow chunk afterwards.
#ifndef TR_TAIL_PIECES
#define TR_TAIL_PIECES 0
#endif
#ifndef TR_TAIL16_DIV
#define TR_TAIL16_DIV 8
#endif
#ifndef TR_TAIL32_DIV
#define TR_TAIL32_DIV 8
#endif
template <bool WIDE>
__global__ __launch_bounds__(KB, TR_CLOSEST_WAVES) void k_trace_fused(SceneView sv, PtParams P, PathBuffers pb, int bounce, const uint* queue,
                                                                      uint* bc, uint* bc_prev) {
    if (!WIDE) sv.wide_textures = 0;
    __shared__ int s_stack_rows[TR_STACK_WORDS];
    int* const s_stack = s_stack_rows + TR_STACK_ROW0;     // row -1 exists (LaneStack, trace.h)
    __shared__ int s_owner[(KB / 64) * TR_OWNER_WORDS];
    const QuadCtx qc = make_quad_ctx(s_stack, s_owner, pb);
    const uint nc = bc[BC_QUEUE], ns = bc_prev[BC_SHADOW];
    const uint chunks_c = (nc + 63u) >> 6, total = chunks_c + ((ns + 63u) >> 6);
    TraceStats st = {};
    uint closest_rays = 0, shadow_rays = 0, max_vis = 0;
    int overflow = 0;
    const uint wave_id = (blockIdx.x * KB + threadIdx.x) >> 6, n_waves = (gridDim.x * KB) >> 6;
#if TR_TAIL_PIECES
    // The end of the launch in smaller pieces (an experiment, profiles/r5/tail_pieces_ab.txt): the last chunks behind the static ones are
    // handed out as halves (32 rays) and then quarters (16 rays, which a wave traces one ray per quad from the start), so that the waves
    // still running when the queue drains hold less each.  Piece p -> chunk g, offset, size: the same function in every wave.
    const uint dyn = total > n_waves ? total - n_waves : 0u;
    const uint tail16 = dyn < n_waves / TR_TAIL16_DIV ? dyn : n_waves / TR_TAIL16_DIV;                          // chunks cut into four
    const uint tail32 = dyn - tail16 < n_waves / TR_TAIL32_DIV ? dyn - tail16 : n_waves / TR_TAIL32_DIV;        // chunks cut into two
    const uint full = total - tail16 - tail32, pieces = full + 2u * tail32 + 4u * tail16;
#else
    const uint pieces = total;
#endif
    bool first = true;
    while (true) {
        uint p = 0;
        if (first) p = wave_id;
        else {
            if (pieces <= n_waves) break;   // the static first chunks covered both queues
            if ((threadIdx.x & 63) == 0) p = n_waves + atomicAdd(&bc[BC_CUR_CLOSEST], 1u);
            p = __shfl(p, 0);
        }
        first = false;
        if (p >= pieces) break;
        uint g = p, off = 0u, size = 64u;
#if TR_TAIL_PIECES
        if (p >= full) {
            const uint q = p - full;
            if (q < 2u * tail32) { g = full + (q >> 1); off = (q & 1u) << 5; size = 32u; }
            else { const uint r = q - 2u * tail32; g = full + tail32 + (r >> 2); off = (r & 3u) << 4; size = 16u; }
        }
#endif
        const uint lane = threadIdx.x & 63;
        const uint slot = lane < size ? off + lane : 0xFFFFFFFFu;      // lanes beyond the piece hold no ray
        if (g < chunks_c) closest_lane<false>(sv, P, pb, bounce, queue, slot == 0xFFFFFFFFu ? slot : (g << 6) + slot, nc, s_stack + threadIdx.x, qc, st, overflow, max_vis, closest_rays);
        else shadow_lane<false>(sv, P, pb, slot == 0xFFFFFFFFu ? slot : ((g - chunks_c) << 6) + slot, ns, s_stack + threadIdx.x, qc, st, overflow, shadow_rays);
    }
    flush_trace_counters<false>(P, pb, overflow, 3000 + bounce, closest_rays, shadow_rays, st, max_vis);
}


// write_all_outputs, first-sample part (path_tracer.glsl:549-563): albedo, material, normal, position, screen motion and
// instance id of the first hit
TR_DEV void write_first_hit_gbuffer(const SceneView& sv, const PtParams& P, uint launch_id, const SurfacePoint& v, const SampledMaterial& mat, bool surface,
                                    int4 h) {
    if (P.samples_accumulated + P.previous_samples != 0) return;
    if (!(P.T.albedo || P.T.material || P.T.normal || P.T.pos || P.T.instance_id || P.T.screen_motion)) return;
    uint lx, ly, lz;
    launch_coord(P.L, launch_id, lx, ly, lz);
    int wx, wy;
    if (!get_write_pixel_pos(P.L, lx, ly, wx, wy) || (uint)wx >= P.target_w || (uint)wy >= P.target_h) return;
    const size_t pix = ((size_t)lz * P.target_h + (uint)wy) * P.target_w + (uint)wx;
    if (P.T.albedo) reinterpret_cast<f4*>(P.T.albedo)[pix] = mat.albedo;
    if (P.T.material)   // pack_gbuffer_material (gbuffer.glsl:256-260)
        reinterpret_cast<f4*>(P.T.material)[pix] = F4(mat.metallic, mat.roughness, (mat.ior_out / mat.ior_in) * 0.25f, mat.transmittance);
    if (P.T.normal) {   // octahedral_pack (math.glsl:480-485)
        f3 nn = v.mapped_normal / (fabsf(v.mapped_normal.x) + fabsf(v.mapped_normal.y) + fabsf(v.mapped_normal.z));
        f2 o = nn.z >= 0.0f ? F2(nn.x, nn.y)
                            : F2((1 - fabsf(nn.y)) * ((nn.x >= 0.0f ? 1.0f : 0.0f) * 2 - 1), (1 - fabsf(nn.x)) * ((nn.y >= 0.0f ? 1.0f : 0.0f) * 2 - 1));
        reinterpret_cast<f2*>(P.T.normal)[pix] = o;
    }
    if (P.T.pos) reinterpret_cast<f4*>(P.T.pos)[pix] = F4(v.pos, 0);
    if (P.T.instance_id) reinterpret_cast<int*>(P.T.instance_id)[pix] = surface ? h.x : -1;
    if (P.T.screen_motion) {   // write_gbuffer_screen_motion (path_tracer.glsl:557-562); lights and misses: prev_pos = pos
        const f3 prev_pos = surface ? surface_prev_pos(sv, h.x, h.y, __int_as_float(h.z), __int_as_float(h.w)) : v.pos;
        const f3 m = get_camera_projection(sv.prev_cameras[global_viewport(P, lz)], P.opt.projection, prev_pos);
        reinterpret_cast<f2*>(P.T.screen_motion)[pix] = F2(m.x, m.y);
    }
}

// The first-hit gbuffer entries of the path tracer, as their own pass right after bounce 0 (only launched when such a
// target is bound and the frame starts an accumulation): it redoes get_intersection_info for the primary hit so that
// k_shade, which runs every bounce of every frame, does not carry this code and its registers.
__global__ __launch_bounds__(KB) void k_first_hit_gbuffer(SceneView sv, PtParams P, PathBuffers pb) {
    uint i = blockIdx.x * KB + threadIdx.x;
    if (i >= P.n_ids) return;
    const uint id = i + P.id_offset;
    const u4 misc = pb.misc[id];
    if (misc.w & 1u) return;
    const int4 h = pb.hit[id];
    // after k_shade(0) the path's origin / direction are those of the next ray for surviving paths: the primary ray is
    // rebuilt from the camera instead
    uint lx, ly, lz;
    launch_coord(P.L, misc.z, lx, ly, lz);
    int px, py;
    if (!get_pixel_pos(P.L, lx, ly, px, py)) return;
    LocalSampler ls = init_local_sampler(u4{(uint)px, (uint)py, global_viewport(P, lz), P.rng_sample}, sample_counter_of(P, lz), P.rng_seed, P.opt.sampler);
    f2 cam_offset = F2(0.0f);
    if (P.opt.film != 0) {
        f4 r = u4_to_unit(pcg4d(ls.rs));
        if (P.opt.film == 1) cam_offset = F2(r.x, r.y) * 2.0f - 1.0f;
        else cam_offset = sample_blackman_harris_concentric_disk(F2(r.x, r.y)) * 2.0f;
        cam_offset = cam_offset * (2.0f * P.opt.film_radius);
    }
    f2 dof_u = F2(0.5f);
    if (P.opt.depth_of_field) { f4 r = u4_to_unit(pcg4d(ls.rs)); dof_u = F2(r.x, r.y); }
    f3 pos, view;
    get_screen_camera_ray(P.L, px, py, sv.cameras[global_viewport(P, lz)], P.opt.projection, P.opt.depth_of_field != 0, cam_offset, dof_u, pos, view);
    SampledMaterial mat;
    mat.albedo = F4(0); mat.metallic = 1; mat.roughness = 0; mat.emission = F3(0);
    mat.transmittance = 0; mat.ior_in = 1; mat.ior_out = 1; mat.f0 = 0;
    SurfacePoint v;
    v.pos = pos; v.hard_normal = F3(0); v.smooth_normal = F3(0); v.mapped_normal = F3(0); v.tri_light_pdf = 0;
    const bool surface = h.x >= 0;
    if (surface) {
        shade_surface(sv, h.x, h.y, __int_as_float(h.z), __int_as_float(h.w), view, pos, false, P.opt.tri_light_mode, P.opt.pre_transformed_vertices != 0, v, mat);
        mat.albedo.w = 1.0f;
    } else if (h.y >= 0) {   // sphere light (path_tracer.glsl:139-158)
        const PointLight pl = sv.point_lights[h.y];
        v.pos = pos + __int_as_float(h.z) * view;
        v.mapped_normal = normalize(v.pos - pl.pos);
        mat.albedo = F4(0, 0, 0, 1);
    } else {                 // miss (path_tracer.glsl:160-199)
        v.pos = pos;
        v.mapped_normal = -view;
        mat.albedo = F4(0);
    }
    write_first_hit_gbuffer(sv, P, misc.z, v, mat, surface, h);
}

// ---------------------------------------------------------------------------------------------------
// direct_stage (src/direct_stage.cc:30-127, shader/direct.rgen:57-132): the first hit with lights hidden plus
// SAMPLES_PER_PASS light samples from it.  k_direct handles light sample `sample` of every path (sample 0 also sets up the
// first-hit terms); the shadow rays go through the same queue and k_trace_shadow_direct adds the visible ones.  The
// colour accumulator of the pass lives in first_emis (rgb; a = first-hit alpha).
template <bool COUNT>
__global__ __launch_bounds__(KB, TR_SHADE_WAVES) void k_direct(SceneView sv, PtParams P, PathBuffers pb, int sample, uint* bc) {
    const uint n = P.n_ids;
    const uint n_round = (n + 63u) & ~63u;
    uint surf = 0;
    for (uint qi = blockIdx.x * KB + threadIdx.x; qi < n_round; qi += gridDim.x * KB) {
        bool active = qi < n;
        uint id = 0;
        u4 misc = {0, 0, 0, 1};
        if (active) { id = qi + P.id_offset; misc = pb.misc[id]; active = !(misc.w & 1u); }
        bool want_shadow = false;
        f3 sh_o = F3(0), sh_d = F3(0), sh_c = F3(0), sh_cw = F3(0);
        f2 sh_w = F2(0.0f);
        float sh_tmax = 0;
        if (active) {
            const f4 o4 = pb.org_pdf[id], d4 = pb.dir_reg[id];
            const int4 h = pb.hit[id];
            const f3 pos = F3(o4), view = F3(d4);
            u4 rs = pb.rng[id];
            // ---- get_intersection_info (path_tracer.glsl:91-201); sphere lights are masked out of the primary ray
            SampledMaterial mat;
            mat.albedo = F4(0); mat.metallic = 1; mat.roughness = 0; mat.emission = F3(0);
            mat.transmittance = 0; mat.ior_in = 1; mat.ior_out = 1; mat.f0 = 0;
            SurfacePoint v;
            v.pos = pos; v.hard_normal = F3(0); v.smooth_normal = F3(0); v.mapped_normal = F3(0); v.tri_light_pdf = 0;
            f3 light = F3(0);
            const bool surface = h.x >= 0;
            if (surface) {
                if (COUNT && sample == 0) surf++;
                shade_surface(sv, h.x, h.y, __int_as_float(h.z), __int_as_float(h.w), view, pos, P.nee_tri != 0, P.opt.tri_light_mode,
                              P.opt.pre_transformed_vertices != 0, v, mat);
                mat.albedo.w = 1.0f;
                if (P.nee_tri) { light = mat.emission; mat.emission = F3(0); }
            } else {
                f4 c = sv.environment_factor;
                if (sv.environment_proj >= 0) {
                    f2 uv;
                    uv.y = asinf(-view.y) / TR_PI + 0.5f;
                    uv.x = atan2f(view.z, view.x) / (2 * TR_PI) + 0.5f;
                    f4 t = sample_envmap(sv, uv);
                    c.x *= t.x; c.y *= t.y; c.z *= t.z;
                }
                for (uint i = 0; i < sv.directional_light_count; ++i) {
                    const DirectionalLight dl = sv.directional_lights[i];
                    if (dl.dir_cutoff >= 1.0f) continue;
                    float visible = stepf(dl.dir_cutoff, dot(view, -dl.dir));
                    f3 dc = visible * dl.color / (2.0f * TR_PI * (1.0f - dl.dir_cutoff));
                    if (P.nee_dir) light += dc; else mat.emission += dc;
                }
                v.pos = pos;
                v.mapped_normal = -view;
                mat.albedo = F4(0);
                if (P.nee_env) light += F3(c); else mat.emission += F3(c);
            }
            const float spp = (float)P.opt.samples_per_pass;
            f4 color = F4(0), dif = F4(0), ref = F4(0);
            bool touched = false;
            if (sample == 0) {
                const f3 c0 = (light + mat.emission) * spp;   // color += (light + emission) * SAMPLES_PER_PASS
                color = F4(c0, mat.albedo.w);
                touched = true;
                write_first_hit_gbuffer(sv, P, misc.z, v, mat, surface, h);
            }
            if (surface) {
                const m3 tbn = create_tangent_space(v.mapped_normal);
                const f3 shading_view = view_to_tangent_space(view, tbn);
                u4 coord;
                {
                    uint lx, ly, lz;
                    launch_coord(P.L, misc.z, lx, ly, lz);
                    int px = 0, py = 0;
                    if (P.opt.sampler == SAMPLER_SOBOL_OWEN) get_pixel_pos(P.L, lx, ly, px, py);
                    coord = u4{(uint)px, (uint)py, global_viewport(P, lz) + P.rng_seed, P.rng_sample + sample_counter_of(P, lz)};
                }
                const bool any_nee = (P.nee_point && sv.point_light_count > 0) || (P.nee_dir && sv.directional_light_count > 0) ||
                                     (P.nee_tri && sv.tri_light_count > 0) || (P.nee_env && sv.environment_proj >= 0);
                u4 rnd = ray_sample_uint(rs, coord, misc.y, (uint)sample, P.opt.sampler, P.max_sobol_bounces);
                if (any_nee) {   // next_event_estimation (path_tracer.glsl:302-344) without a MIS define: the light pdf alone
                    Lobes lobes = {0, 0, 0, 0};
                    f3 out_dir;
                    float out_length = 0.0f, light_pdf;
                    f3 contrib = sample_explicit_light(sv, P, rnd, v.pos, out_dir, out_length, light_pdf);
                    f3 shading_light = mulT(out_dir, tbn);
                    float nee_bsdf_pdf = material_bsdf_pdf(P.opt.bounce_mode, shading_light, shading_view, mat, lobes);
                    correct_lobes_for_normal_map(out_dir, v.hard_normal, lobes);
                    const bool cast = contrib.x > 0.0001f || contrib.y > 0.0001f || contrib.z > 0.0001f;
                    const f3 radiance = contrib / nee_mis_pdf(P, light_pdf, nee_bsdf_pdf);
                    const f3 cw = modulate_bsdf(mat, lobes);
                    const f2 w = F2(lobes.diffuse + lobes.transmission, lobes.dielectric_reflection + lobes.metallic_reflection);
                    if (cast) {
                        want_shadow = true;
                        sh_o = v.pos; sh_d = out_dir; sh_tmax = out_length; sh_c = radiance; sh_cw = cw; sh_w = w;
                    } else {
                        if (!touched) { color = pb.first_emis[id]; dif = pb.diffuse[id]; ref = pb.reflection[id]; touched = true; }
                        color.x += radiance.x * cw.x; color.y += radiance.y * cw.y; color.z += radiance.z * cw.z;
                        dif.x += radiance.x * w.x; dif.y += radiance.y * w.x; dif.z += radiance.z * w.x;
                        ref.x += radiance.x * w.y; ref.y += radiance.y * w.y; ref.z += radiance.z * w.y;
                    }
                }
                // diffuse.a = reflection.a = 1 / length(first_hit_vertex.pos - pos), set by every sample
                const float inv_len = 1.0f / length(v.pos - pos);
                if (touched) { dif.w = inv_len; ref.w = inv_len; }
                else { pb.diffuse[id].w = inv_len; pb.reflection[id].w = inv_len; }
                pb.rng[id] = rs;
            }
            if (touched) { pb.first_emis[id] = color; pb.diffuse[id] = dif; pb.reflection[id] = ref; }
        }
        uint sslot = wave_append(&bc[BC_SHADOW], want_shadow);
        if (want_shadow) {
            pb.sh_org_tmax[sslot] = F4(sh_o, sh_tmax);
            pb.sh_dir_id[sslot] = F4(sh_d, __uint_as_float(id));
            pb.sh_contrib[sslot] = F4(sh_c, 0);
            pb.sh_lobes[sslot] = sh_w;
            pb.sh_cweight[sslot] = F4(sh_cw, 0);
        }
    }
    if (COUNT && P.count_work) {
        for (int off = 32; off > 0; off >>= 1) surf += __shfl_xor(surf, off);
        if ((threadIdx.x & 63) == 0) add64(pb.counters, CNT_SURF, surf);
    }
}

template <bool COUNT>
__global__ __launch_bounds__(KB, TR_SHADOW_WAVES) void k_trace_shadow_direct(SceneView sv, PtParams P, PathBuffers pb, uint* bc) {
    __shared__ int s_stack_rows[TR_STACK_WORDS];
    int* const s_stack = s_stack_rows + TR_STACK_ROW0;     // row -1 exists (LaneStack, trace.h)
    const uint n = bc[BC_SHADOW];
    TraceStats st = {};
    uint rays = 0;
    int overflow = 0;
    for (uint qi = blockIdx.x * KB + threadIdx.x; qi < ((n + 63u) & ~63u); qi += gridDim.x * KB) {
        if (qi >= n) continue;
        const f4 o = pb.sh_org_tmax[qi], d = pb.sh_dir_id[qi], c = pb.sh_contrib[qi];
        const float vis = trace_shadow4<COUNT>(sv, F3(o), F3(d), P.opt.min_ray_dist, o.w, s_stack + threadIdx.x, st, overflow);
        rays++;
        if (vis == 0.0f) continue;
        const uint id = __float_as_uint(d.w);
        const f3 radiance = F3(c.x * vis, c.y * vis, c.z * vis);
        const f2 w = pb.sh_lobes[qi];
        const f4 cw = pb.sh_cweight[qi];
        f4 col = pb.first_emis[id];
        col.x += radiance.x * cw.x; col.y += radiance.y * cw.y; col.z += radiance.z * cw.z;
        pb.first_emis[id] = col;
        if (w.x != 0.0f) { f4 d4 = pb.diffuse[id]; d4.x += radiance.x * w.x; d4.y += radiance.y * w.x; d4.z += radiance.z * w.x; pb.diffuse[id] = d4; }
        if (w.y != 0.0f) { f4 r4 = pb.reflection[id]; r4.x += radiance.x * w.y; r4.y += radiance.y * w.y; r4.z += radiance.z * w.y; pb.reflection[id] = r4; }
    }
    flush_trace_counters<COUNT>(P, pb, overflow, 5000, 0u, rays, st, 0u);
}

__global__ void k_direct_next_sample(uint* bc) { bc[BC_SHADOW] = 0; }   // the shadow queue of the previous light sample is done

// direct.rgen:main: color /= SAMPLES_PER_PASS; diffuse /= SAMPLES_PER_PASS (reflection is not divided); write_all_outputs
__global__ __launch_bounds__(KB) void k_resolve_direct(PtParams P, PathBuffers pb) {
    uint i = blockIdx.x * KB + threadIdx.x;
    if (i >= P.n_ids) return;
    i += P.id_offset;
    uint lx, ly, lz;
    launch_coord(P.L, i, lx, ly, lz);
    int wx, wy;
    if (!get_write_pixel_pos(P.L, lx, ly, wx, wy)) return;
    if ((uint)wx >= P.target_w || (uint)wy >= P.target_h) return;
    const float spp = (float)P.opt.samples_per_pass;
    const size_t idx = ((size_t)lz * P.target_h + (uint)wy) * P.target_w + (uint)wx;
    const uint prev_samples = P.samples_accumulated + P.previous_samples;
    const float keep = prev_samples != 0 ? (float)prev_samples / (float)((uint)P.opt.samples_per_pass + prev_samples) : 0.0f;
    auto accumulate = [&](void* image, f4 value) {
        f4* target = reinterpret_cast<f4*>(image);
        if (prev_samples != 0) value = mix4(value, target[idx], keep);
        target[idx] = value;
    };
    if (P.T.color) { const f4 c = pb.first_emis[i]; accumulate(P.T.color, F4(c.x / spp, c.y / spp, c.z / spp, P.opt.transparent_background ? c.w : 1.0f)); }
    if (P.T.diffuse) { const f4 d = pb.diffuse[i]; accumulate(P.T.diffuse, F4(d.x / spp, d.y / spp, d.z / spp, d.w / spp)); }
    if (P.T.reflection) accumulate(P.T.reflection, pb.reflection[i]);
}

uint calculate_shuffled_strips_b(uint sx, uint sy) {   // src/distribution_strategy.cc:62-69
    uint n = sx * sy, b = 31;
    while ((n >> b) < 128 && b > 0) b--;
    return b;
}

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return set_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

}  // namespace

// shade_fast.hip: the ahead-of-time k_shade instances (command-line option set or general) at the accuracy Vulkan asks of the reference's GLSL
void launch_shade_fast(bool cli, bool count, bool last, uint blocks, hipStream_t stream, const SceneView& sv, const PtParams& P, const PathBuffers& pb, int bounce,
                       const uint* queue, uint* bc, uint* next_queue);

void get_ray_count(const trhip_distribution& d, uint& w, uint& h) {   // src/distribution_strategy.cc:33-61
    if (d.strategy == 0) { w = d.size_x; h = d.size_y; }
    else if (d.strategy == 1) { w = d.size_x; h = (d.size_y - d.index + d.count - 1) / d.count; }
    else { w = d.count; h = 1; }
}

constexpr int PT_LANES = 4;   // independent slices of a frame that run concurrently (see PtStage::render)
static uint trace_grid_cap() {   // most blocks a persistent trace launch gets (TRHIP_GRID_BLOCKS); sizes the quad-tail spill buffer
    static const uint cap = getenv("TRHIP_GRID_BLOCKS") ? (uint)atoi(getenv("TRHIP_GRID_BLOCKS")) : 256u * 8u;
    return cap;
}
static uint fused_grid_factor() {   // blocks of a fused launch per block of paths of its lane (two queues: 2 gives every chunk of both a wave)
    static const uint f = getenv("TRHIP_FUSED_GRID_FACTOR") ? (uint)std::max(1, atoi(getenv("TRHIP_FUSED_GRID_FACTOR"))) : 2u;
    return f;
}
// words of the quad-tail spill buffer one trace launch of `blocks` blocks needs: a slice per wave
static size_t qspill_words(size_t blocks) { return std::max<size_t>(blocks, 1) * (KB / 64) * 16u * TR_QSPILL; }
struct TimedSpan { int kind; hipEvent_t a, b; };
enum { T_CLOSEST = 0, T_SHADOW = 1, T_SHADE = 2, T_RAYGEN = 3, T_RESOLVE = 4, T_KINDS = 5 };

struct PtStage::Impl {
    PathBuffers pb{};
    size_t capacity = 0;
    float prev_frame_ms = 0.0f;        // device time of the stage's previous frame (0 = unknown): picks the enqueue order of the lanes
    size_t qspill_lane_words = 0;      // one region of PathBuffers::qspill: the largest trace launch of a lane
    size_t qspill_regions = 0;         // regions allocated: one per lane in use, two for a single lane whose shadow launches run on the side stream
    hipEvent_t ev[2]{};
    bool ev_init = false;
    std::vector<hipEvent_t> pool;      // recycled events for per-launch timing
    std::vector<TimedSpan> pending;    // recorded, not yet resolved
    float acc_ms[T_KINDS] = {0, 0, 0, 0, 0};
    uint acc_launches[T_KINDS] = {0, 0, 0, 0, 0};
    uint frames = 0;
    hipStream_t side = nullptr;        // lane 1, or (single lane) shadow rays of bounce b while closest(b+1) runs on the caller's stream
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t lane_stream[PT_LANES] = {};   // lanes 2.. (lane 0 = caller's stream, lane 1 = side)
    hipEvent_t lane_join[PT_LANES] = {};
    hipEvent_t pass_done[PT_LANES] = {};      // sample lanes: k_resolve of the lane's latest pass
    hipEvent_t get_event() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreateWithFlags(&e, hipEventDisableSystemFence); return e;
    }
};

PtStage::PtStage(DeviceScene* scene, const trhip_pt_options& o) : scene(scene), opt(o), impl(new Impl()) {
    dist = trhip_distribution{0, 0, 0, 0, 1, 1};
}

PtStage::~PtStage() {
    free_buffers();
    if (impl->pb.counters) (void)hipFree(impl->pb.counters);
    if (impl->pb.bounce) (void)hipFree(impl->pb.bounce);
    if (impl->pb.qspill) (void)hipFree(impl->pb.qspill);
    if (impl->ev_init) for (auto& e : impl->ev) (void)hipEventDestroy(e);
    for (auto& sp : impl->pending) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (auto& e : impl->pool) (void)hipEventDestroy(e);
    if (impl->side) { stream_pool_release(impl->side); (void)hipEventDestroy(impl->ev_fork); (void)hipEventDestroy(impl->ev_join); }
    for (int l = 2; l < PT_LANES; ++l) if (impl->lane_stream[l]) { stream_pool_release(impl->lane_stream[l]); (void)hipEventDestroy(impl->lane_join[l]); }
    for (auto& e : impl->pass_done) if (e) (void)hipEventDestroy(e);
    delete impl;
}

void PtStage::free_buffers() {
    PathBuffers& pb = impl->pb;
    void* ptrs[] = {pb.org_pdf, pb.dir_reg, pb.atten_alpha, pb.diffuse, pb.reflection, pb.plobes, pb.first_mat, pb.first_emis, pb.rng, pb.misc, pb.hit,
                    pb.sum_color, pb.sum_diffuse, pb.sum_reflection, pb.sh_org_tmax, pb.sh_dir_id, pb.sh_contrib, pb.sh_lobes, pb.sh_cweight, pb.queue[0], pb.queue[1]};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    uint *counters = pb.counters, *bounce = pb.bounce;
    int* qspill = pb.qspill;
    pb = PathBuffers{};
    pb.counters = counters; pb.bounce = bounce; pb.qspill = qspill;
    impl->capacity = 0;
}

int PtStage::ensure_buffers(size_t n, bool lobe_sums) {
    PathBuffers& pb = impl->pb;
    if (!pb.counters) {
        HIPCHK(hipMalloc(&pb.counters, PT_LANES * CNT_WORDS * sizeof(uint)));   // one block of counters per lane
        HIPCHK(hipMemset(pb.counters, 0, PT_LANES * CNT_WORDS * sizeof(uint)));
        HIPCHK(hipMalloc(&pb.bounce, (size_t)PT_LANES * BC_STRIDE * ((size_t)opt.max_bounces + 2u) * sizeof(uint)));
    }
    if (!impl->ev_init) { for (auto& e : impl->ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableSystemFence)); impl->ev_init = true; }
    if (n <= impl->capacity && (!lobe_sums || pb.sum_diffuse)) return 0;
    free_buffers();
    HIPCHK(hipMalloc(&pb.org_pdf, n * 16)); HIPCHK(hipMalloc(&pb.dir_reg, n * 16)); HIPCHK(hipMalloc(&pb.atten_alpha, n * 16));
    HIPCHK(hipMalloc(&pb.diffuse, n * 16)); HIPCHK(hipMalloc(&pb.reflection, n * 16)); HIPCHK(hipMalloc(&pb.plobes, n * 8));
    HIPCHK(hipMalloc(&pb.first_mat, n * 16)); HIPCHK(hipMalloc(&pb.first_emis, n * 16)); HIPCHK(hipMalloc(&pb.rng, n * 16));
    HIPCHK(hipMalloc(&pb.misc, n * 16)); HIPCHK(hipMalloc(&pb.hit, n * 16)); HIPCHK(hipMalloc(&pb.sum_color, n * 16));
    HIPCHK(hipMalloc(&pb.sh_org_tmax, n * 16)); HIPCHK(hipMalloc(&pb.sh_dir_id, n * 16)); HIPCHK(hipMalloc(&pb.sh_contrib, n * 16));
    HIPCHK(hipMalloc(&pb.sh_lobes, n * 8));
    if (lobe_sums) { HIPCHK(hipMalloc(&pb.sum_diffuse, n * 16)); HIPCHK(hipMalloc(&pb.sum_reflection, n * 16)); }
    HIPCHK(hipMalloc(&pb.queue[0], n * 4)); HIPCHK(hipMalloc(&pb.queue[1], n * 4));
    impl->capacity = n;
    return 0;
}

// The quad-tail spill buffer: `regions` slices of `blocks` blocks' worth each (grows only).  Sized by the lanes a frame really
// uses and by the grid its trace launches really get - a frame slot running one lane of 1024-block launches holds 30 MB, a lone
// frame on four lanes 118 MB (it was PT_LANES x the 2048-block cap = 236 MB for every stage).
static int ensure_qspill(PathBuffers& pb, size_t& lane_words, size_t& regions_have, size_t regions, size_t blocks) {
    const size_t words = qspill_words(blocks);
    if (words <= lane_words && regions <= regions_have) return 0;
    if (pb.qspill) HIPCHK(hipFree(pb.qspill));
    pb.qspill = nullptr;
    lane_words = std::max(lane_words, words);
    regions_have = std::max(regions_have, regions);
    HIPCHK(hipMalloc(&pb.qspill, regions_have * lane_words * sizeof(int)));
    return 0;
}

PtStage::Program PtStage::choose_program() {
    static const bool cli_instances = !(getenv("TRHIP_SHADE_CLI") && atoi(getenv("TRHIP_SHADE_CLI")) == 0);
    const bool wide = scene->wide_textures != 0;      // RGBA16 textures: the kernel instances with the two-format texel fetch
    const bool cli_set = cli_instances && is_cli_default_set(opt) && scene->shade_tris != nullptr && !wide;     // k_shade<.., SpecCli>: reads the ShadeTri records, RGBA8 texels
    static const bool shade_fast_env = !(getenv("TRHIP_SHADE_FAST") && atoi(getenv("TRHIP_SHADE_FAST")) == 0);
    const bool shade_fast = ieee_shading < 0 ? shade_fast_env : ieee_shading == 0;
    static const bool specialize_env = !(getenv("TRHIP_SPECIALIZE") && atoi(getenv("TRHIP_SPECIALIZE")) == 0);
    const SpecKernels *spec_shade = nullptr, *spec_raygen = nullptr;
    std::string key;
    if (!cli_set && !direct && (specialize < 0 ? specialize_env : specialize != 0)) {
        SpecRequest rq{opt, scene->shade_tris != nullptr && !opt.pre_transformed_vertices, !shade_fast, count_work != 0, SPEC_SHADE, wide};
        std::string why;
        spec_shade = spec_kernels(rq, &why);
        if (spec_shade) { rq.program = SPEC_RAYGEN; spec_raygen = spec_kernels(rq, &why); }
        if (!spec_shade || !spec_raygen) {
            static bool warned = false;
            if (!warned) fprintf(stderr, "[trhip] no specialised shading program for {%s}: %s - rendering with the general kernels\n", spec_key(rq).c_str(), why.c_str());
            warned = true;
            spec_shade = nullptr; spec_raygen = nullptr;
        }
    }
    return Program{cli_set, shade_fast, wide, spec_shade, spec_raygen, key};
}

// trhip_pt_get_program: the choice above with an identity the ranks of a job can compare
int PtStage::get_program(trhip_program_info* out) {
    memset(out, 0, sizeof(*out));
    const Program p = choose_program();
    out->kind = p.shade ? 2 : (p.cli_set ? 1 : 0);
    out->ieee = p.shade_fast ? 0 : 1;
    SpecRequest rq{opt, scene->shade_tris != nullptr && !opt.pre_transformed_vertices, !p.shade_fast, count_work != 0, SPEC_SHADE, p.wide};
    const std::string key = (out->kind == 2 ? "compiled: " : out->kind == 1 ? "command-line set, ahead of time: " : "general kernels: ") + spec_key(rq);
    snprintf(out->key, sizeof(out->key), "%s", key.c_str());
    unsigned long long h = spec_sources_hash();
    h = spec_fnv1a(h, &out->kind, sizeof(out->kind)); h = spec_fnv1a(h, &out->ieee, sizeof(out->ieee));
    const int dir = direct ? 1 : 0;
    h = spec_fnv1a(h, &dir, sizeof(dir));
    // the general kernels read the options as data: their identity is the build; the other two pin option fields
    if (out->kind != 0) { const std::string k = spec_key(rq); h = spec_fnv1a(h, k.c_str(), k.size()); }
    if (p.shade) { h = spec_fnv1a(h, &p.shade->code_hash, 8); h = spec_fnv1a(h, &p.raygen->code_hash, 8); }
    out->identity = h;
    return 0;
}

int PtStage::render(const trhip_pt_targets& targets, uint target_w, uint target_h, uint viewports, hipStream_t stream) {
    if (!scene->accel_built) return set_error("trhip_pt_render: call trhip_scene_build_accel first");
    if (viewports == 0 || viewports % frame_batch != 0) return set_error("trhip_pt_render: the layers of a launch are a whole number of frames (trhip_pt_set_frame_batch)");
    const uint frame_views = viewports / frame_batch;
    if ((uint64_t)shard_vp_base + (uint64_t)(frame_views - 1) * shard_vp_stride >= scene->camera_count)
        return set_error("trhip_pt_render: viewport count exceeds uploaded cameras");
    if (frame_batch > 1 && (accumulated_samples != 0 || direct)) return set_error("trhip_pt_render: a frame batch renders independent frames of the path tracer (reset the accumulation first)");
    if (opt.samples_per_pass <= 0 || opt.samples_per_pixel % opt.samples_per_pass != 0)
        return set_error("trhip_pt_render: samples_per_pixel must be a multiple of samples_per_pass");
    if (dist.size_x == 0 || dist.size_y == 0) return set_error("trhip_pt_render: distribution not set");
    PtParams P{};
    P.opt = opt;
    uint lw, lh;
    get_ray_count(dist, lw, lh);
    P.L.size_x = dist.size_x; P.L.size_y = dist.size_y; P.L.strategy = dist.strategy; P.L.index = dist.index;
    P.L.count = dist.strategy == 2 ? calculate_shuffled_strips_b(dist.size_x, dist.size_y) : dist.count;
    P.L.primary = dist.primary; P.L.launch_w = lw; P.L.launch_h = lh;
    P.viewports = viewports;
    const size_t n = (size_t)lw * lh * viewports;
    if (n == 0) return 0;
    if (n > 0xFFFFFFF0ull) return set_error("trhip_pt_render: launch too large");
    P.n_launch = (uint)n; P.id_offset = 0; P.n_ids = (uint)n;
    P.max_sobol_bounces = (uint)(opt.max_bounces > 8 ? 8 : opt.max_bounces);   // shader/sobol_lookup_table.glsl:4-14
    // src/rt_stage.cc:81; a sample shard renders every shard_sample_stride-th sample of samples_per_pixel * stride per frame
    P.sample_counter = frame_counter * (uint)opt.samples_per_pixel * shard_sample_stride;
    P.vp_base = shard_vp_base; P.vp_stride = shard_vp_stride;
    P.frame_views = frame_views; P.frame_counter_step = (uint)opt.samples_per_pixel * shard_sample_stride;
    { uint s = opt.rng_seed; P.rng_seed = s != 0 ? pcg(s) : 0; }                  // src/rt_stage.cc:82
    P.samples_accumulated = accumulated_samples;
    P.target_w = target_w; P.target_h = target_h;
    P.nee_point = opt.nee_point > 0; P.nee_dir = opt.nee_directional > 0; P.nee_env = opt.nee_envmap > 0; P.nee_tri = opt.nee_triangles > 0;
    {   // get_nee_sampling_probabilities (shader/rt.glsl:302-335): scene constants, evaluated once in fp32
        float point = (P.nee_point && scene->point_light_count > 0) ? opt.nee_point : 0.0f;
        float tri = (P.nee_tri && scene->tri_light_count > 0) ? opt.nee_triangles : 0.0f;
        float dir = (P.nee_dir && scene->directional_light_count > 0) ? opt.nee_directional : 0.0f;
        float env = (P.nee_env && scene->environment_proj >= 0) ? opt.nee_envmap : 0.0f;
        float sum = point + tri + dir + env;
        float inv_sum = sum <= 0.0f ? 0.0f : (1.0f / sum + 1e-5f);
        P.prob_point = point * inv_sum; P.prob_tri = tri * inv_sum; P.prob_dir = dir * inv_sum; P.prob_env = env * inv_sum;
    }
    P.count_work = 1;
    P.bounce_words = (uint)BC_STRIDE * ((uint)opt.max_bounces + 2u);
    P.fused_resolve = opt.samples_per_pass == 1;
    P.T = targets;
    P.tm_display = nullptr; P.tm_op = tm_info.op; P.tm_exposure = tm_info.exposure; P.tm_gamma = tm_info.gamma; P.tm_grid = tm_info.alpha_grid_background;
    const bool timing = detailed_timing != 0;
    // A frame of several one-sample passes can keep whole samples in flight instead of slices of one (see "sample lanes" below):
    // every lane then needs path state for all n paths.
    static const bool sample_lanes_enabled = !(getenv("TRHIP_SAMPLE_LANES") && atoi(getenv("TRHIP_SAMPLE_LANES")) == 0);
    const int passes_total = opt.samples_per_pixel / opt.samples_per_pass;
    const int sample_lane_count = std::min(passes_total, PT_LANES);
    const bool sample_lanes = sample_lanes_enabled && !direct && !timing && lanes == 0 && opt.samples_per_pass == 1 && passes_total >= 2 &&
                              n * (size_t)sample_lane_count * 224u <= ((size_t)8 << 30);
    if (int rc = ensure_buffers(sample_lanes ? n * (size_t)sample_lane_count : n, targets.diffuse || targets.reflection)) return rc;
    PathBuffers& pb = impl->pb;
    SceneView sv = scene->view();
    if (opt.pre_transformed_vertices) {   // PRE_TRANSFORMED_VERTICES: shade from scene_stage's world-space vertex copy
        if (int rc = ensure_world_vertices(*scene, stream)) return rc;
        sv.vertices = scene->world_vertices; sv.spans = scene->world_spans;
    }
    const bool count = count_work != 0;
    // Which shading program renders this stage.  The kernels exist in two arithmetics - IEEE fp32 (path_tracer.hip) and, the default,
    // the accuracy Vulkan asks of the reference's GLSL (shade_fast.hip; trhip_pt_set_shading_arithmetic) - and, in each, as
    //  * the ahead-of-time instances of the command-line option set (SpecCli; TRHIP_SHADE_CLI=0 switches them off),
    //  * an instance compiled for this stage's option set the first time it renders (shade_spec.hip through specialize.cc: hipRTC,
    //    or the kernel cache; trhip_pt_set_specialization / TRHIP_SPECIALIZE=0 switch that off),
    //  * the general instances, which read every option from the parameter block - what renders when neither of the above applies.
    // All three render the same bits in the same arithmetic.
    const Program prog = choose_program();
    const bool wide = prog.wide, cli_set = prog.cli_set, shade_fast = prog.shade_fast;
    const SpecKernels *spec_shade = prog.shade, *spec_raygen = prog.raygen;
    auto launch_raygen = [&](uint blocks, hipStream_t on, const PtParams& LP, const PathBuffers& lb) {
        if (spec_raygen) {
            SceneView a0 = sv; PtParams a1 = LP; PathBuffers a2 = lb;
            void* args[] = {&a0, &a1, &a2};
            (void)hipModuleLaunchKernel(spec_raygen->raygen, blocks, 1, 1, KB, 1, 1, 0, on, args, nullptr);
        } else hipLaunchKernelGGL(k_raygen<SpecGeneral>, dim3(blocks), dim3(KB), 0, on, sv, LP, lb);
    };
    auto launch_shade = [&](bool count, bool last, uint blocks, hipStream_t on, const PtParams& LP, const PathBuffers& lb, int bounce, const uint* q, uint* bc, uint* qn) {
        if (spec_shade) {
            SceneView a0 = sv; PtParams a1 = LP; PathBuffers a2 = lb;
            void* args[] = {&a0, &a1, &a2, &bounce, &q, &bc, &qn};
            (void)hipModuleLaunchKernel(last ? spec_shade->shade_last : spec_shade->shade, blocks, 1, 1, KB, 1, 1, 0, on, args, nullptr);
        } else if (shade_fast) launch_shade_fast(cli_set, count, last, blocks, on, sv, LP, lb, bounce, q, bc, qn);
        else if (last) {
            if (count) hipLaunchKernelGGL((k_shade<true, true>), dim3(blocks), dim3(KB), 0, on, sv, LP, lb, bounce, q, bc, qn);
            else if (cli_set) hipLaunchKernelGGL((k_shade<false, true, SpecCli>), dim3(blocks), dim3(KB), 0, on, sv, LP, lb, bounce, q, bc, qn);
            else hipLaunchKernelGGL((k_shade<false, true>), dim3(blocks), dim3(KB), 0, on, sv, LP, lb, bounce, q, bc, qn);
        }
        else if (count) hipLaunchKernelGGL((k_shade<true, false>), dim3(blocks), dim3(KB), 0, on, sv, LP, lb, bounce, q, bc, qn);
        else if (cli_set) hipLaunchKernelGGL((k_shade<false, false, SpecCli>), dim3(blocks), dim3(KB), 0, on, sv, LP, lb, bounce, q, bc, qn);
        else hipLaunchKernelGGL((k_shade<false, false>), dim3(blocks), dim3(KB), 0, on, sv, LP, lb, bounce, q, bc, qn);
    };
    // Concurrency inside a frame.  The trace kernels are persistent and leave the chip under-filled while their last
    // waves finish, the bounce loop is a chain of dependent launches, and trace (VALU-bound) and shade (latency-bound)
    // want different resources.  Two ways to fill the gaps, both bit-neutral:
    //  * lanes (default, four = the HIP runtime's hardware queues): the path ids are cut into slices with their own
    //    queues and counters, and every slice runs its whole bounce loop on its own stream, so one slice's shade overlaps
    //    another's traversal;
    //  * TRHIP_LANES=1: one lane; shadow(b) shares the launch of closest(b + 1) (k_trace_fused), or - TRHIP_FUSED=0 - runs
    //    on a side stream next to it.
    // Per-kernel timing (trhip_pt_set_profiling) wants kernels that own the chip: one lane, one stream.
    static const int lanes_env = getenv("TRHIP_LANES") ? atoi(getenv("TRHIP_LANES")) : PT_LANES;
    static const bool overlap_enabled = !(getenv("TRHIP_OVERLAP") && atoi(getenv("TRHIP_OVERLAP")) == 0);
    // Small frames (the shards of a multi-GPU job) are bound by the latency of one ray's dependent fetches per kernel, not by
    // throughput; extra lanes pay as long as a lane still has a few hundred blocks.  One frame at a time, sponza_teapots, lanes
    // 1 / 2 / 4 (profiles/r5/lanes_by_share.txt, every lane on a hardware pipe of its own - stream_pool.hip; round 3's
    // thresholds were measured with whatever pipes the streams had landed on): 65 k paths 0.50 / 0.52 / 0.55 ms, 130 k
    // 0.60 / 0.57 / 0.61, 261 k 0.94 / 0.79 / 0.74, 518 k 1.49 / 1.40 / 1.17, 1.04 M 2.52 / 2.30 / 2.16, 2.07 M 4.48 / 3.96 / 3.70.
    static const size_t lanes_min_paths = getenv("TRHIP_LANES_MIN_PATHS") ? (size_t)atol(getenv("TRHIP_LANES_MIN_PATHS")) : (size_t)100000;
    //  * sample lanes (a frame of two or more one-sample passes, the offline case - BASELINE config 3 is 4096 of them): the lanes
    //    take turns with whole samples instead of sharing one, each with path state of its own, so that up to four samples are
    //    in flight like the frames of a renderer with frame slots (3.4 instead of 4.0 ms per sample on sponza_class).  A pass
    //    ends in k_resolve, which blends into the targets and therefore runs in pass order: each one waits for the previous
    //    pass's, on whatever lane that ran.
    //  * a stage of a renderer with frames in flight (trhip_pt_set_frame_slots): the frames overlap each other, so one lane per frame -
    //    except with two slots, where two lanes each put the pair on all four hardware pipes (two in flight, one frame per launch:
    //    3.57 -> 3.43 ms on sponza_teapots, 3.05 -> 2.92 sponza_class, 1.80 -> 1.66 test.glb; profiles/r5/two_in_flight_lanes.txt)
    static const size_t slots2_min_paths = getenv("TRHIP_SLOTS2_MIN_PATHS") ? (size_t)atol(getenv("TRHIP_SLOTS2_MIN_PATHS")) : (size_t)50000;
    const int auto_lanes = frame_slots >= 3 ? 1
                         : frame_slots == 2 ? (n < slots2_min_paths ? 1 : std::min(lanes_env, 2))
                         : (n < lanes_min_paths ? 1 : std::max(1, std::min(n < 2 * lanes_min_paths ? std::min(lanes_env, 2) : lanes_env, PT_LANES)));
    const int n_lanes = sample_lanes ? sample_lane_count : (timing ? 1 : (lanes > 0 ? std::min(lanes, PT_LANES) : auto_lanes));
    // shadow(b) rides in the launch of closest(b + 1) unless kernels are being timed or counted one by one
    static const bool fused_enabled = !(getenv("TRHIP_FUSED") && atoi(getenv("TRHIP_FUSED")) == 0);
    const bool first_hit_targets = targets.albedo || targets.material || targets.normal || targets.pos || targets.instance_id || targets.screen_motion;
    const bool fused = fused_enabled && !timing && !count;
    const bool overlap = overlap_enabled && !timing && !fused && n_lanes == 1;
    if ((overlap || n_lanes > 1) && !impl->side) {
        if (int rc = stream_pool_acquire(&impl->side, &stream, 1)) return rc;      // on another hardware pipe than the caller's stream
        HIPCHK(hipEventCreateWithFlags(&impl->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&impl->ev_join, hipEventDisableTiming));
    }
    // Trace launches of frames that share the chip - with the other frames in flight (frame slots, one lane each) or with the
    // other three lanes of their own frame - get four blocks per CU instead of eight, which leaves room for the launches next
    // to them (four slots: sponza_teapots 4.33 -> 4.18 ms per frame, sponza_class 3.59 -> 3.44; four lanes of a lone frame:
    // 4.78 -> 4.52 ms, profiles/r2/schedule_sweep.txt).  Only a kernel that is timed alone wants the whole chip.
    // Round 5 (profiles/r5/sync_schedule_sweep.txt, lone_frame_grids.txt): with the round's faster traversal the four pixel lanes of a lone frame want
    // smaller grids still - three blocks per CU for the trace launches and for k_shade (768 / 768 instead of 1024 / 2048): a lane's launch
    // then leaves half the chip to the other lanes' launches, which is what lanes are for; one frame at a time 4.06 -> 3.72 ms on
    // sponza_teapots.  Frame slots (one lane per frame, whole frames per launch) and sample lanes keep the larger grids: pipelined 3.29 ms
    // against 3.37 with the small ones.
    const bool pixel_lanes = n_lanes >= 3 && !sample_lanes;
    const bool few_slots = (n_lanes == 1 && (frame_slots == 2 || frame_slots == 3)) || (n_lanes == 2 && frame_slots == 2);      // trhip_pt_set_frame_slots: trace launches of three blocks per CU
    const uint grid_cap = (!timing && !getenv("TRHIP_GRID_BLOCKS")) ? ((pixel_lanes || few_slots) ? 768u : 1024u) : trace_grid_cap();
    // A trace kernel that is timed alone gets exactly the blocks that are resident at its register budget (persistent waves: a
    // block that has to wait for a slot only lengthens the tail): 1536 for the closest-hit kernel, 0.544 -> 0.527 ms per launch
    // on sponza_teapots against the 2048 both used to get; the shadow kernel's budget is 8 per CU, i.e. 2048.
    static const uint n_cu = [] { int dev = 0, cu = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev); return (uint)std::max(cu, 1); }();
    const uint closest_cap = (timing && !getenv("TRHIP_GRID_BLOCKS")) ? std::min(n_cu * (uint)TR_CLOSEST_WAVES, trace_grid_cap()) : grid_cap;   // the spill buffer of the quad tail is sized by trace_grid_cap()
    const uint shadow_cap = (timing && !getenv("TRHIP_GRID_BLOCKS")) ? std::min(n_cu * (uint)TR_SHADOW_WAVES, trace_grid_cap()) : grid_cap;
    {   // spill regions: one per lane; a single lane whose shadow launches overlap the next closest-hit launch on the side stream
        // needs a second one (both kernels index their slices by block and wave)
        const size_t lane_paths = sample_lanes ? n : (n + (size_t)n_lanes - 1) / (size_t)n_lanes;
        const size_t lane_blocks = std::min<size_t>(std::max(std::max(closest_cap, shadow_cap), grid_cap), fused_grid_factor() * ((lane_paths + KB - 1) / KB) + 1);
        if (int rc = ensure_qspill(impl->pb, impl->qspill_lane_words, impl->qspill_regions, (size_t)std::max(n_lanes, overlap ? 2 : 1), lane_blocks)) return rc;
    }
    auto& ev = impl->ev;
    // per-launch event pair, recorded on the launch stream, resolved lazily in get_timings()
    auto timed = [&](int kind, hipStream_t on, auto&& launch) {
        if (!timing) { launch(); return; }
        TimedSpan sp{kind, impl->get_event(), impl->get_event()};
        (void)hipEventRecord(sp.a, on);
        launch();
        (void)hipEventRecord(sp.b, on);
        impl->pending.push_back(sp);
    };
    if (timing_pending) {      // how long the previous frame of this stage took, if it is over (one frame at a time: always): picks the enqueue order below
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) impl->prev_frame_ms = ms;
        else (void)hipGetLastError();
    }
    HIPCHK(hipEventRecord(ev[0], stream));
    if (direct) {   // direct_stage: one lane on the caller's stream (src/direct_stage.cc:104-127)
        if (!pb.sh_cweight) HIPCHK(hipMalloc(&pb.sh_cweight, impl->capacity * 16));
        PtParams LP = P;
        LP.opt.hide_lights = 1;      // the primary ray is traced with mask 0xFF ^ 0x02
        LP.opt.mis_mode = 0;         // no MIS define in direct_stage: nee_mis_pdf is the light pdf
        PathBuffers lb = pb;
        const uint blocks_all = (LP.n_ids + KB - 1) / KB;
        const uint blocks_q = blocks_all < grid_cap ? blocks_all : grid_cap;
        const int passes = opt.samples_per_pixel / opt.samples_per_pass;
        for (int pass = 0; pass < passes; ++pass) {
            LP.previous_samples = (uint)pass * (uint)opt.samples_per_pass;
            LP.sample_in_pass = 0;
            LP.rng_sample = shard_sample_base + shard_sample_stride * LP.previous_samples;
            timed(T_RAYGEN, stream, [&] { hipLaunchKernelGGL(k_raygen<SpecGeneral>, dim3(blocks_all), dim3(KB), 0, stream, sv, LP, lb); });
            timed(T_CLOSEST, stream, [&] {
                auto kc = count ? k_trace_closest<true, false> : k_trace_closest<false, false>;
                hipLaunchKernelGGL(kc, dim3(blocks_q), dim3(KB), 0, stream, sv, LP, lb, 0, (const uint*)nullptr, lb.bounce);
            });
            for (int smp = 0; smp < opt.samples_per_pass; ++smp) {
                if (smp > 0) hipLaunchKernelGGL(k_direct_next_sample, dim3(1), dim3(1), 0, stream, lb.bounce);
                timed(T_SHADE, stream, [&] {
                    if (count) hipLaunchKernelGGL(k_direct<true>, dim3(blocks_q), dim3(KB), 0, stream, sv, LP, lb, smp, lb.bounce);
                    else hipLaunchKernelGGL(k_direct<false>, dim3(blocks_q), dim3(KB), 0, stream, sv, LP, lb, smp, lb.bounce);
                });
                timed(T_SHADOW, stream, [&] {
                    auto ks = count ? k_trace_shadow_direct<true> : k_trace_shadow_direct<false>;
                    hipLaunchKernelGGL(ks, dim3(blocks_q), dim3(KB), 0, stream, sv, LP, lb, lb.bounce);
                });
            }
            timed(T_RESOLVE, stream, [&] { hipLaunchKernelGGL(k_resolve_direct, dim3(blocks_all), dim3(KB), 0, stream, LP, lb); });
        }
        HIPCHK(hipEventRecord(ev[1], stream));
        HIPCHK(hipGetLastError());
        timing_pending = true;
        impl->frames++;
        frame_counter++;
        accumulated_samples += (uint)opt.samples_per_pixel;
        return 0;
    }
    for (int l = 2; l < n_lanes; ++l) if (!impl->lane_stream[l]) {
        hipStream_t taken[PT_LANES] = {stream, impl->side};
        for (int k = 2; k < l; ++k) taken[k] = impl->lane_stream[k];
        if (int rc = stream_pool_acquire(&impl->lane_stream[l], taken, l)) return rc;       // every lane on a pipe of its own
        HIPCHK(hipEventCreateWithFlags(&impl->lane_join[l], hipEventDisableTiming));
    }
    if (n_lanes != last_lanes || stream != last_main) {     // what trhip_pt_get_lane_pipes reports (looked up when the schedule changes)
        last_lanes = n_lanes; last_main = stream;
        for (int l = 0; l < n_lanes && l < 4; ++l) {
            int c = -1;
            (void)stream_pool_class(l == 0 ? stream : (l == 1 ? impl->side : impl->lane_stream[l]), &c);
            last_lane_pipes[l] = c;
        }
    }
    if (n_lanes > 1) {   // fork
        HIPCHK(hipEventRecord(impl->ev_fork, stream));
        HIPCHK(hipStreamWaitEvent(impl->side, impl->ev_fork, 0));
        for (int l = 2; l < n_lanes; ++l) HIPCHK(hipStreamWaitEvent(impl->lane_stream[l], impl->ev_fork, 0));
    }
    // Slices: a lane may render its pixels as several slices one after the other (TRHIP_LANE_SLICES, an experiment: the frame ends with
    // the tails of smaller slices).  Slice i runs on the stream of lane i mod n_lanes and reuses that lane's counters and spill region.
    static const int slices_env = getenv("TRHIP_LANE_SLICES") ? std::max(1, std::min(atoi(getenv("TRHIP_LANE_SLICES")), 4)) : 1;
    const int n_slices = (sample_lanes || n_lanes < 2) ? n_lanes : n_lanes * slices_env;
    const uint per_lane = sample_lanes ? (uint)n : (uint)((((n + n_slices - 1) / n_slices + 63) / 64) * 64);   // whole 8x8 tiles per slice
    {   // interleave the slices' tiles when the image is whole tiles and divides evenly (always true for 1080p / 4 lanes)
        const size_t tiles = n / 64;
        const bool even = !sample_lanes && viewports == 1 && (lw & 7u) == 0 && (lh & 7u) == 0 && n_slices > 1 && tiles % (size_t)n_slices == 0 &&
                          (size_t)per_lane * (size_t)n_slices == n;
        P.L.tile_lanes = even ? (uint)n_slices : 1u;
        P.L.tiles_per_lane = even ? (uint)(tiles / (size_t)n_slices) : 0u;
    }
    if (sample_lanes) for (int l = 0; l < n_lanes; ++l) if (!impl->pass_done[l]) HIPCHK(hipEventCreateWithFlags(&impl->pass_done[l], hipEventDisableTiming));
    struct LaneCtx { hipStream_t ls; PtParams LP; PathBuffers lb; uint blocks_all, blocks_q, blocks_f; bool shadow_in_flight; };
    LaneCtx lane_ctx[PT_LANES * 4];
    int lanes_used = 0;
    for (int slice = 0; slice < n_slices; ++slice) {
        const int lane = slice % n_lanes;      // the stream, the counters and the spill region
        LaneCtx& c = lane_ctx[slice];
        c.ls = lane == 0 ? stream : (lane == 1 ? impl->side : impl->lane_stream[lane]);
        c.LP = P;
        c.LP.id_offset = sample_lanes ? 0u : (uint)slice * per_lane;
        if (c.LP.id_offset >= n) break;
        c.LP.n_ids = std::min(per_lane, (uint)n - c.LP.id_offset);
        PathBuffers& lb = c.lb;
        lb = pb;   // the lane's view: its own queues / shadow queue / counters; the per-path arrays shared (slices of one sample) or its own (sample lanes)
        lb.counters = pb.counters + lane * CNT_WORDS;
        lb.bounce = pb.bounce + (size_t)lane * P.bounce_words;
        lb.qspill = pb.qspill + (size_t)lane * impl->qspill_lane_words;
        const size_t q0 = sample_lanes ? (size_t)lane * n : (size_t)c.LP.id_offset;
        lb.queue[0] = pb.queue[0] + q0; lb.queue[1] = pb.queue[1] + q0;
        lb.sh_org_tmax = pb.sh_org_tmax + q0; lb.sh_dir_id = pb.sh_dir_id + q0;
        lb.sh_contrib = pb.sh_contrib + q0; lb.sh_lobes = pb.sh_lobes + q0;
        if (sample_lanes) {
            const size_t o = (size_t)lane * n;
            lb.org_pdf += o; lb.dir_reg += o; lb.atten_alpha += o; lb.diffuse += o; lb.reflection += o; lb.plobes += o; lb.first_mat += o; lb.first_emis += o;
            lb.rng += o; lb.misc += o; lb.hit += o;
        }
        c.blocks_all = (c.LP.n_ids + KB - 1) / KB;
        // persistent-style launch for the queue kernels: enough blocks to fill the chip, grid-stride over the queue
        c.blocks_q = c.blocks_all < grid_cap ? c.blocks_all : grid_cap;
        // The fused launch carries two queues (closest-hit rays of this bounce, shadow rays of the last one): a small frame - a rank's share
        // of a multi-GPU job - leaves the chip room for a wave per chunk of both, and then the launch lasts as long as the longer of the two
        // traversals instead of their sum (a 1/8 strip of sponza_teapots: 165 -> 130 us per fused launch, profiles/r5/strip_timeline_1_8.txt)
        c.blocks_f = std::min(grid_cap, c.blocks_all * fused_grid_factor());
        c.shadow_in_flight = false;
        lanes_used = slice + 1;
    }
    const int passes = passes_total;
    // every launch of one pass of one lane
    // `only`: -2 = the whole pass; otherwise one step of it, so that the steps of several lanes can be enqueued in turn
    // (-1 = ray generation, 0 .. max_bounces - 1 = that bounce, max_bounces = accumulation and resolve)
    auto enqueue_pass = [&](int lane, int pass, int only) -> int {
        LaneCtx& c = lane_ctx[lane];
        const hipStream_t ls = c.ls;
        PtParams& LP = c.LP;
        PathBuffers& lb = c.lb;
        const uint blocks_all = c.blocks_all, blocks_q = c.blocks_q, blocks_f = c.blocks_f;
        bool& shadow_in_flight = c.shadow_in_flight;
        {
            LP.previous_samples = (uint)pass * (uint)opt.samples_per_pass;
            for (int s = 0; s < opt.samples_per_pass; ++s) {
                LP.sample_in_pass = (uint)s;
                LP.rng_sample = shard_sample_base + shard_sample_stride * (LP.previous_samples + LP.sample_in_pass);
                if (only == -2 || only == -1) timed(T_RAYGEN, ls, [&] {
                    launch_raygen(blocks_all, ls, LP, lb);
                });
                for (int bounce = 0; bounce < opt.max_bounces; ++bounce) {
                    if (only != -2 && only != bounce) continue;
                    const uint* q = bounce == 0 ? nullptr : lb.queue[bounce & 1];
                    uint* qn = lb.queue[(bounce + 1) & 1];
                    uint* bc = lb.bounce + BC_STRIDE * bounce;
                    if (fused && bounce > 0) {
                        // closest(b) together with shadow(b - 1): one launch, one tail
                        hipLaunchKernelGGL(wide ? k_trace_fused<true> : k_trace_fused<false>, dim3(blocks_f), dim3(KB), 0, ls, sv, LP, lb, bounce, q, bc, bc - BC_STRIDE);
                    } else {
                        timed(T_CLOSEST, ls, [&] {
                            auto kc = count ? k_trace_closest<true, false> : (timing ? (wide ? k_trace_closest<false, true, true> : k_trace_closest<false, true, false>)
                                                                                     : (wide ? k_trace_closest<false, false, true> : k_trace_closest<false, false, false>));
                            hipLaunchKernelGGL(kc, dim3(std::min(blocks_all, closest_cap)), dim3(KB), 0, ls, sv, LP, lb, bounce, q, bc);
                        });
                    }
                    if (shadow_in_flight) { HIPCHK(hipStreamWaitEvent(ls, impl->ev_join, 0)); shadow_in_flight = false; }
                    timed(T_SHADE, ls, [&] {
                        // k_shade holds three waves per SIMD (768 resident blocks) and strides over the queue; 2048 blocks since the
                        // trace launches of a frame slot shrank to 1024 (round 2 sweep: profiles/r2/schedule_sweep.txt)
                        static const uint shade_cap_env = getenv("TRHIP_SHADE_BLOCKS") ? (uint)atoi(getenv("TRHIP_SHADE_BLOCKS")) : 0u;
                        const uint shade_cap = shade_cap_env ? shade_cap_env : ((pixel_lanes || (n_lanes == 2 && frame_slots == 2)) ? 768u : 2048u);
                        const uint blocks_s = timing ? blocks_q : (blocks_all < shade_cap ? blocks_all : shade_cap);   // alone on the chip it wants the full grid
                        static const bool last_variant = !(getenv("TRHIP_SHADE_LAST") && atoi(getenv("TRHIP_SHADE_LAST")) == 0);
                        const bool last = last_variant && bounce == opt.max_bounces - 1;
                        launch_shade(count, last, blocks_s, ls, LP, lb, bounce, q, bc, qn);
                    });
                    if (bounce == 0 && first_hit_targets && s == opt.samples_per_pass - 1 && LP.samples_accumulated + LP.previous_samples == 0)
                        hipLaunchKernelGGL(k_first_hit_gbuffer, dim3(blocks_all), dim3(KB), 0, ls, sv, LP, lb);
                    if (bounce < opt.max_bounces - 1 && !fused) {
                        hipStream_t ss = ls;
                        if (overlap) {   // fork: shadow(b) on the side stream, closest(b+1) follows on the caller's stream
                            HIPCHK(hipEventRecord(impl->ev_fork, ls));
                            HIPCHK(hipStreamWaitEvent(impl->side, impl->ev_fork, 0));
                            ss = impl->side;
                        }
                        timed(T_SHADOW, ss, [&] {
                            auto ks = count ? k_trace_shadow<true> : (wide ? k_trace_shadow<false, true> : k_trace_shadow<false, false>);
                            // on the side stream the launch runs next to closest(b + 1) of the same lane: its quad tails spill into
                            // the second region, not into the slices the closest-hit waves are using
                            PathBuffers sb = lb;
                            if (overlap) sb.qspill = pb.qspill + impl->qspill_lane_words;
                            hipLaunchKernelGGL(ks, dim3(std::min(blocks_all, shadow_cap)), dim3(KB), 0, ss, sv, LP, sb, bc);
                        });
                        if (overlap) { HIPCHK(hipEventRecord(impl->ev_join, impl->side)); shadow_in_flight = true; }
                    }
                }
                if (only != -2 && only != opt.max_bounces) continue;
                if (shadow_in_flight) { HIPCHK(hipStreamWaitEvent(ls, impl->ev_join, 0)); shadow_in_flight = false; }
                if (!LP.fused_resolve) hipLaunchKernelGGL(k_accumulate_sample, dim3(blocks_all), dim3(KB), 0, ls, LP, lb);
            }
            if (only != -2 && only != opt.max_bounces) return 0;
            // sample lanes: the targets have seen pass - 1 before this pass blends into them
            if (sample_lanes && pass > 0) HIPCHK(hipStreamWaitEvent(ls, impl->pass_done[(pass - 1) % n_lanes], 0));
            LP.tm_display = (pass == passes - 1 && !direct) ? reinterpret_cast<f4*>(tm_display) : nullptr;      // the frame's last pass leaves the final colour
            timed(T_RESOLVE, ls, [&] { hipLaunchKernelGGL(k_resolve, dim3(blocks_all), dim3(KB), 0, ls, LP, lb); });
            if (sample_lanes && pass + 1 < passes) HIPCHK(hipEventRecord(impl->pass_done[lane], ls));
        }
        return 0;
    };
    // Enqueue order of the lanes of a one-pass frame (same bits either way; profiles/r3/enqueue_order_ab.txt).  Lane after lane leaves
    // the lanes a dozen launches (~60 us of host time) apart, so they run out of phase - one shades while another traces - which is
    // worth 3 % on a 4.2 ms frame (sponza_teapots); step by step in turn keeps them in phase, which is worth 6 % on a 2.2 ms frame
    // (test.glb) and costs 3 % on the long one; lanes one or two steps apart gain on neither.  So: in turn when the previous frame
    // of this stage took less than 3 ms, lane after lane otherwise.  TRHIP_ENQUEUE=lanes | step | skew<k> pins the order.
    static const char* order_env = getenv("TRHIP_ENQUEUE");
    const bool interleave = order_env ? strcmp(order_env, "lanes") != 0 : (impl->prev_frame_ms > 0.0f && impl->prev_frame_ms < 3.0f);
    if (sample_lanes) {
        for (int pass = 0; pass < passes; ++pass) if (int rc = enqueue_pass(pass % lanes_used, pass, -2)) return rc;
    } else if (interleave && lanes_used > 1 && passes == 1 && opt.samples_per_pass == 1) {
        // TRHIP_ENQUEUE=skew<k>: lane l runs k steps behind lane l - 1 (0 = all lanes in step)
        static const int skew = (order_env && !strncmp(order_env, "skew", 4)) ? atoi(order_env + 4) : 0;
        const int n_steps = opt.max_bounces + 2;
        for (int first = 0; first < lanes_used; first += n_lanes) {      // slices of one stream one after the other (they share its counters)
            const int here = std::min(n_lanes, lanes_used - first);
            for (int t = 0; t < n_steps + skew * (here - 1); ++t)
                for (int lane = 0; lane < here; ++lane) {
                    const int step = t - skew * lane;
                    if (step < 0 || step >= n_steps) continue;
                    if (int rc = enqueue_pass(first + lane, 0, step - 1)) return rc;
                }
        }
    } else {
        for (int lane = 0; lane < lanes_used; ++lane)
            for (int pass = 0; pass < passes; ++pass) if (int rc = enqueue_pass(lane, pass, -2)) return rc;
    }
    if (n_lanes > 1) {   // join
        HIPCHK(hipEventRecord(impl->ev_join, impl->side));
        HIPCHK(hipStreamWaitEvent(stream, impl->ev_join, 0));
        for (int l = 2; l < n_lanes; ++l) {
            HIPCHK(hipEventRecord(impl->lane_join[l], impl->lane_stream[l]));
            HIPCHK(hipStreamWaitEvent(stream, impl->lane_join[l], 0));
        }
    }
    HIPCHK(hipEventRecord(ev[1], stream));
    HIPCHK(hipGetLastError());
    timing_pending = true;
    impl->frames++;
    // rt_stage::update: frame_counter++ ; rt_camera_stage::update: accumulated_samples += samples_per_pixel
    frame_counter += frame_batch;
    accumulated_samples += (uint)opt.samples_per_pixel;
    return 0;
}

int PtStage::get_counters(trhip_counters* out, hipStream_t stream) {
    memset(out, 0, sizeof(*out));
    if (!impl->pb.counters) return 0;
    HIPCHK(hipStreamSynchronize(stream));
    uint hl[PT_LANES][CNT_WORDS];
    HIPCHK(hipMemcpy(hl, impl->pb.counters, sizeof(hl), hipMemcpyDeviceToHost));
    uint* h = hl[0];
    auto rd = [&](int i) { uint64_t v = 0; for (int l = 0; l < PT_LANES; ++l) v += (uint64_t)hl[l][i] | ((uint64_t)hl[l][i + 1] << 32); return v; };
    out->closest_rays = rd(CNT_CLOSEST); out->shadow_rays = rd(CNT_SHADOWRAYS); out->node_visits = rd(CNT_NODES);
    out->tri_tests = rd(CNT_TRIS); out->alpha_tests = rd(CNT_ALPHA); out->surface_hits = rd(CNT_SURF);
    for (int l = 1; l < PT_LANES; ++l) { h[CNT_OVERFLOW] |= hl[l][CNT_OVERFLOW]; h[CNT_MAXSP] = std::max(h[CNT_MAXSP], hl[l][CNT_MAXSP]); h[CNT_MAXVIS] = std::max(h[CNT_MAXVIS], hl[l][CNT_MAXVIS]); }
    out->stack_overflows = h[CNT_OVERFLOW];
    if (getenv("TRHIP_DEBUG"))
        fprintf(stderr, "[trhip] closest-hit loop: %llu node phases (%llu with <= 16 active rays serving %llu visits, %llu with <= 8), %llu triangle phases; quad tail: %llu node phases, %llu triangle phases; %llu node visits\n",
                (unsigned long long)rd(CNT_PH_NODE), (unsigned long long)rd(CNT_PH_NODE16), (unsigned long long)rd(CNT_LV_NODE16), (unsigned long long)rd(CNT_PH_NODE8),
                (unsigned long long)rd(CNT_PH_TRI), (unsigned long long)rd(CNT_PH_QNODE), (unsigned long long)rd(CNT_PH_QTRI), (unsigned long long)rd(CNT_NODES));
    if (getenv("TRHIP_DEBUG")) {
        fprintf(stderr, "[trhip] per-lane node phases by live rays (1-8, 9-16, ..., 57-64):");
        for (int b = 0; b < 8; ++b) fprintf(stderr, " %llu", (unsigned long long)rd(CNT_PH_HIST + 2 * b));
        fprintf(stderr, "\n");
    }
    if (getenv("TRHIP_DEBUG")) { float* f = (float*)(h + CNT_DBG); fprintf(stderr, "[trhip] overflow %u src %u; max stack depth %u; max node visits per ray %u; worst ray o=(%g %g %g) d=(%g %g %g) bounce %g id %g pdf %g reg %g\n", h[CNT_OVERFLOW], h[CNT_DBG + 12], h[CNT_MAXSP], h[CNT_MAXVIS], f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], f[8], f[9]); }
    return 0;
}

// Wave-level phase statistics of the counting trace kernels (the denominators of "rays per vector instruction")
int PtStage::get_phase_counters(trhip_phase_counters* out, hipStream_t stream) {
    memset(out, 0, sizeof(*out));
    if (!impl->pb.counters) return 0;
    HIPCHK(hipStreamSynchronize(stream));
    uint hl[PT_LANES][CNT_WORDS];
    HIPCHK(hipMemcpy(hl, impl->pb.counters, sizeof(hl), hipMemcpyDeviceToHost));
    auto rd = [&](int i) { uint64_t v = 0; for (int l = 0; l < PT_LANES; ++l) v += (uint64_t)hl[l][i] | ((uint64_t)hl[l][i + 1] << 32); return v; };
    out->lane_node_phases = rd(CNT_PH_NODE); out->lane_tri_phases = rd(CNT_PH_TRI);
    out->quad_node_phases = rd(CNT_PH_QNODE); out->quad_tri_phases = rd(CNT_PH_QTRI);
    out->lane_node_phases_le16 = rd(CNT_PH_NODE16); out->lane_node_visits_le16 = rd(CNT_LV_NODE16);
    out->closest_node_visits = rd(CNT_CNODES);
    for (int b = 0; b < 8; ++b) out->lane_node_phase_hist[b] = rd(CNT_PH_HIST + 2 * b);
    return 0;
}

int PtStage::reset_counters() {
    if (impl->pb.counters) HIPCHK(hipMemset(impl->pb.counters, 0, PT_LANES * CNT_WORDS * sizeof(uint)));
    if (int rc = resolve_pending()) return rc;
    for (int k = 0; k < T_KINDS; ++k) { impl->acc_ms[k] = 0; impl->acc_launches[k] = 0; }
    impl->frames = 0;
    return 0;
}

int PtStage::resolve_pending() {
    for (auto& sp : impl->pending) {
        HIPCHK(hipEventSynchronize(sp.b));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, sp.a, sp.b));
        impl->acc_ms[sp.kind] += ms;
        impl->acc_launches[sp.kind]++;
        impl->pool.push_back(sp.a); impl->pool.push_back(sp.b);
    }
    impl->pending.clear();
    return 0;
}

int PtStage::get_timings(trhip_timings* out) {
    memset(out, 0, sizeof(*out));
    if (timing_pending) {
        HIPCHK(hipEventSynchronize(impl->ev[1]));
        HIPCHK(hipEventElapsedTime(&out->path_tracing_ms, impl->ev[0], impl->ev[1]));
    }
    if (int rc = resolve_pending()) return rc;
    out->trace_closest_ms = impl->acc_ms[T_CLOSEST]; out->trace_shadow_ms = impl->acc_ms[T_SHADOW]; out->shade_ms = impl->acc_ms[T_SHADE];
    out->raygen_ms = impl->acc_ms[T_RAYGEN]; out->resolve_ms = impl->acc_ms[T_RESOLVE];
    out->trace_closest_launches = impl->acc_launches[T_CLOSEST]; out->trace_shadow_launches = impl->acc_launches[T_SHADOW];
    out->shade_launches = impl->acc_launches[T_SHADE]; out->frames = impl->frames;
    return 0;
}

}  // namespace tr

#if TR_TIMELINE
// Read-out of the phase timeline (trace_timeline.h); only in the variant library built with -DTR_TIMELINE=1, not part of include/trhip.h.
extern "C" int trhip_debug_timeline(unsigned long long* out, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(tr::g_timeline), sizeof(unsigned long long) * tr::TL_WORDS) != hipSuccess) return 1;
    if (reset) { static unsigned long long zero[tr::TL_WORDS]; if (hipMemcpyToSymbol(HIP_SYMBOL(tr::g_timeline), zero, sizeof(zero)) != hipSuccess) return 1; }
    return 0;
}
#endif
