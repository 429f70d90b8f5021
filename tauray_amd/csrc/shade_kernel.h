// The shading kernels of the wavefront path tracer: k_raygen (path_tracer.rgen:88-101) and k_shade, one bounce of evaluate_ray
// (path_tracer.glsl:385-498), with the light sampling and MIS code they call.  Compiled three ways from this one source:
//   * path_tracer.hip: IEEE fp32, the general instances (every option is data) and the instances of the command-line option set;
//   * shade_fast.hip: the same instances at the accuracy Vulkan asks of the reference's GLSL;
//   * shade_spec.hip: one option set pinned by -DTR_SPEC_* macros, ahead of time or through hipRTC when a stage with that option set
//     is first rendered (specialize.cc) - what the reference does with its #defines (src/path_tracer_stage.cc:30-116).
// Device-only and free of host headers.
#pragma once
#include "pt_state.h"

namespace tr {

namespace {

// ---------------------------------------------------------------------------------------------------
// MIS (path_tracer.glsl:54-89)
TR_DEV float bsdf_mis_pdf(const SceneView& sv, const PtParams& P, float pl_pdf, float dl_pdf, float tri_pdf, float env_pdf, float bsdf_pdf) {
    if (bsdf_pdf == 0.0f) return 1.0f;
    float avg_nee_pdf =
        dl_pdf * P.prob_dir / (float)max(sv.directional_light_count, 1u) +
        tri_pdf * P.prob_tri / (float)max(sv.tri_light_count, 1u) +
        env_pdf * P.prob_env +
        pl_pdf * P.prob_point / (float)max(sv.point_light_count, 1u);
    if (P.opt.mis_mode == 2) return (avg_nee_pdf * avg_nee_pdf + bsdf_pdf * bsdf_pdf) / bsdf_pdf;
    if (P.opt.mis_mode == 1) return avg_nee_pdf + bsdf_pdf;
    return avg_nee_pdf > 0 ? __builtin_huge_valf() : bsdf_pdf;
}
TR_DEV float nee_mis_pdf(const PtParams& P, float nee_pdf, float bsdf_pdf) {
    if (nee_pdf <= 0.0f) return -nee_pdf;
    if (P.opt.mis_mode == 2) return (nee_pdf * nee_pdf + bsdf_pdf * bsdf_pdf) / nee_pdf;
    if (P.opt.mis_mode == 1) return nee_pdf + bsdf_pdf;
    return nee_pdf;
}
TR_DEV float clamp_contribution_mul(const PtParams& P, f3 contrib) {   // path_tracer.glsl:356-365
    if (P.opt.indirect_clamping > 0.0f) {
        float m = rgb_to_luminance(contrib);
        if (m > P.opt.indirect_clamping) return P.opt.indirect_clamping / m;
    }
    return 1;
}

// sample_environment_map (rt.glsl:251-285), in two halves: which entry of the alias table the sample reads, and the sample once
// that entry is there (sample_explicit_light fetches it together with whatever records the other lanes of the wave chose)
TR_DEV int environment_alias_index(const SceneView& sv, u4 rnd) {
    const uint sx = sv.env_w, sy = sv.env_h;
    const uint ipx = clampu(rnd.x / (0xFFFFFFFFu / sx), 0u, sx - 1u), ipy = clampu(rnd.y / (0xFFFFFFFFu / sy), 0u, sy - 1u);
    return (int)(ipx + ipy * sx);
}
TR_DEV f3 sample_environment_map(const SceneView& sv, u4 rnd, int i, const AliasEntry at, f3& dir, float& length, float& pdf) {
    f3 color = F3(sv.environment_factor);
    if (sv.environment_proj >= 0) {
        uint sx = sv.env_w, sy = sv.env_h;
        const uint pixel_count = sx * sy;
        pdf = at.pdf;
        if (rnd.z > at.probability) { i = (int)at.alias_id; pdf = at.alias_pdf; }
        int ppx = (int)((uint)i % sx), ppy = (int)((uint)i / sx);
        f2 off = F2((float)(uint)(rnd.x * pixel_count), (float)(uint)(rnd.y * pixel_count)) * TR_INV_UINT32_MAX;
        f2 uv = (F2((float)ppx, (float)ppy) + off) / F2((float)sx, (float)sy);
        dir = uv_to_latlong_direction(uv);
        color = color * F3(sample_envmap(sv, uv));
    } else {
        pdf = 1.0f / (4.0f * TR_PI);
        dir = sample_sphere(F2((float)rnd.x, (float)rnd.y) * TR_INV_UINT32_MAX);
    }
    length = __builtin_huge_valf();
    return color;
}
TR_DEV float sample_environment_map_pdf(const SceneView& sv, f3 dir) {   // rt.glsl:287-299
    if (sv.environment_proj >= 0) {
        uint i = (uint)latlong_direction_to_pixel_id(dir, (int)sv.env_w, (int)sv.env_h);
        uint n = sv.env_w * sv.env_h;
        if (i >= n) i = n - 1;   // the GLSL read is out of bounds for the last half texel row/column
        return sv.alias_table[i].pdf;
    }
    return 1.0f / (4.0f * TR_PI);
}

// sample_explicit_light (path_tracer.glsl:203-289).
// Round 6: select, fetch, compute.  The reference's chain of `(u.w -= probability) < 0` tests picks one kind of light per path, so the
// lanes of a wave sample different kinds, and with a fetch inside every branch a wave paid the round trips of all of them one after
// the other - point light record, triangle light record (+ its texture), alias table entry (+ envmap texels), directional light
// record: half of the 16 000 clocks the light sample took of a shade wave's 50 000 (profiles/r6/shade_phase_timeline.txt).  Now every
// lane first decides its kind and its record - a PointLight, a TriLight (64 bytes each), an AliasEntry (16) or a DirectionalLight
// (32) - then all lanes fetch theirs at once through one per-lane pointer into the same four 16-byte registers, and the branches that
// follow only compute (the environment's texels and an emissive triangle's texture remain fetches of their own).  Same tests in the
// same order, same arithmetic on the same values.
TR_DEV f3 sample_explicit_light(const SceneView& sv, const PtParams& P, u4 rnd, f3 pos, f3& out_dir, float& out_length, float& pdf) {
    f4 u = u4_to_unit(rnd);
    enum { NONE = 0, POINT, TRI, ENV, DIR };
    int kind = NONE;
    if (P.nee_point && (u.w -= P.prob_point) < 0) kind = POINT;
    else if (P.nee_tri && (u.w -= P.prob_tri) < 0) kind = TRI;
    else if (P.nee_env && (u.w -= P.prob_env) < 0) kind = ENV;
    else if (P.nee_dir && (u.w -= P.prob_dir) < 0) kind = DIR;
    // (opaque to the optimiser: it would otherwise thread every test above straight into its branch below and put a copy of the fetch
    // back into each - the structure this function was rewritten to get rid of)
    asm volatile("" : "+v"(kind));
    // the lane's record: address and size in 16-byte words
    // (the counts by value first: a conditional expression over members of the view would select between their addresses and pin the view in memory)
    const int n_point = (int)sv.point_light_count, n_tri = (int)sv.tri_light_count, n_dir = (int)sv.directional_light_count;
    const int light_count = kind == POINT ? n_point : kind == TRI ? n_tri : n_dir;
    const int light_index = clampi((int)(u.z * light_count), 0, light_count - 1);   // random_sample_point_light and its siblings
    // (four masked fetches into the same registers, not one fetch through a selected pointer: a pointer chosen among the arrays of
    // the scene view makes the compiler keep the whole view in scratch memory and every load of the kernel a flat one)
    u4 r0 = {0u, 0u, 0u, 0u}, r1 = r0, r2 = r0, r3 = r0;
    int alias_i = 0;
    if (kind == POINT) { const u4* rec = reinterpret_cast<const u4*>(sv.point_lights + light_index); r0 = rec[0]; r1 = rec[1]; r2 = rec[2]; r3 = rec[3]; }
    if (kind == TRI) { const u4* rec = reinterpret_cast<const u4*>(sv.tri_lights + light_index); r0 = rec[0]; r1 = rec[1]; r2 = rec[2]; r3 = rec[3]; }
    if (kind == DIR) { const u4* rec = reinterpret_cast<const u4*>(sv.directional_lights + light_index); r0 = rec[0]; r1 = rec[1]; }
    if (kind == ENV && sv.environment_proj >= 0) { alias_i = environment_alias_index(sv, rnd); r0 = *reinterpret_cast<const u4*>(sv.alias_table + alias_i); }
#define TR_F(x) __uint_as_float(x)
    STL(STL_L_SELECT);

    if (kind == POINT) {
        float weight = (float)max(light_count, 1);
        PointLight pl;
        pl.color = F3(TR_F(r0.x), TR_F(r0.y), TR_F(r0.z)); pl.dir = F3(TR_F(r0.w), TR_F(r1.x), TR_F(r1.y)); pl.pos = F3(TR_F(r1.z), TR_F(r1.w), TR_F(r2.x));
        pl.radius = TR_F(r2.y); pl.dir_cutoff = TR_F(r2.z); pl.dir_falloff = TR_F(r2.w);
        pl.cutoff_radius = TR_F(r3.x); pl.spot_radius = TR_F(r3.y); pl.shadow_map_index = (int)r3.z; pl.padding = (int)r3.w;
        f3 color;
        sample_point_light(pl, F2(u.x, u.y), pos, out_dir, out_length, color, pdf);
        pdf *= P.prob_point / weight;
        return color;
    }
    if (kind == TRI) {
        f3 A = F3(TR_F(r0.x), TR_F(r0.y), TR_F(r0.z)) - pos, B = F3(TR_F(r0.w), TR_F(r1.x), TR_F(r1.y)) - pos, C = F3(TR_F(r1.z), TR_F(r1.w), TR_F(r2.x)) - pos;
        f3 color = r9g9b9e5_to_rgb(r2.y);
        const int emission_tex_id = (int)r3.w;
        float tri_pdf = 0.0f;
        out_dir = sample_triangle_light(P.opt.tri_light_mode, F2(u.x, u.y), A, B, C, tri_pdf);
        out_length = ray_plane_intersection_dist(out_dir, A, B, C);
        if (isinf(tri_pdf) || tri_pdf <= 0 || out_length <= P.opt.min_ray_dist || any_nan(out_dir)) {
            pdf = 1.0f; out_dir = F3(0);
            return F3(0);
        }
        if (emission_tex_id >= 0) {
            f3 bary = get_barycentric_coords(out_dir * out_length, A, B, C);
            f2 uv = bary.x * unpack_half2x16(r3.x) + bary.y * unpack_half2x16(r3.y) + bary.z * unpack_half2x16(r3.z);
            color = color * F3(sample_texture(sv, emission_tex_id, uv));
        }
        out_length -= P.opt.min_ray_dist;
        pdf = P.prob_tri * tri_pdf / light_count;
        STL(STL_L_TRI);
        return color;
    }
    if (kind == ENV) {
        const AliasEntry at = {r0.x, r0.y, TR_F(r0.z), TR_F(r0.w)};
        f3 color = sample_environment_map(sv, rnd, alias_i, at, out_dir, out_length, pdf);
        pdf *= P.prob_env;
        STL(STL_L_ENV);
        return color;
    }
    if (kind == DIR) {
        DirectionalLight dl;
        dl.color = F3(TR_F(r0.x), TR_F(r0.y), TR_F(r0.z)); dl.shadow_map_index = (int)r0.w; dl.dir = F3(TR_F(r1.x), TR_F(r1.y), TR_F(r1.z)); dl.dir_cutoff = TR_F(r1.w);
        out_length = __builtin_huge_valf();
        out_dir = sample_cone(F2(u.x, u.y), -dl.dir, dl.dir_cutoff);   // sample_directional_light (light.glsl:119-129)
        pdf = dl.dir_cutoff >= 1.0f ? -1.0f : 1.0f / (2.0f * TR_PI * (1.0f - dl.dir_cutoff));
        f3 color = pdf > 0 ? dl.color * pdf : dl.color;
        pdf *= P.prob_dir / light_count;
        STL(STL_L_DIR);
        return color;
    }
#undef TR_F
    out_dir = F3(0); out_length = 0; pdf = 1.0f;
    return F3(0);
}

TR_DEV void correct_lobes_for_normal_map(f3 sample_dir, f3 geometric_normal, Lobes& l) {   // path_tracer.glsl:291-300
    if (dot(geometric_normal, sample_dir) < 0) { l.diffuse = 0; l.dielectric_reflection = 0; l.metallic_reflection = 0; }
    else l.transmission = 0;
}

// One bounce of evaluate_ray (path_tracer.glsl:385-498) for every live path of the queue.
#ifndef TR_SHADE_NOLOOP
#define TR_SHADE_NOLOOP 0       // 1: one pass per thread instead of the persistent loop (an experiment: profiles/r6/shade_noloop_ab.txt)
#endif
#ifndef TR_SHADE_WAVES
#define TR_SHADE_WAVES (TR_SHADE_NOLOOP ? 4 : 3)
#endif
#ifndef TR_SHADE_LAST_WAVES
#define TR_SHADE_LAST_WAVES 5   // the last bounce only collects emission: no light or BSDF sampling, no queue appends
#endif
// LAST: the instance for bounce == max_bounces - 1, where every path is terminal (path_tracer.glsl:445): compiled without the
// NEE / BSDF half of the loop body and without the block-wide appends (and their barriers).
//
// S: the option set the instance is compiled for.  The reference compiles its options into the pipeline as #defines
// (src/path_tracer_stage.cc:30-116); here the options are data, and an instance that is built for one option set overwrites
// those fields of its parameter block with constants, so that the compiler drops the other samplers, film filters, bounce modes,
// light modes, MIS rules and their registers.
//   * SpecGeneral: nothing pinned - every option is read from the parameter block (18 855 instructions at IEEE fp32);
//   * SpecCli: the option set of the reference's command line (uniform-random sampler, point film, power
//     MIS, material bounces, solid-angle triangle lights, no roulette / clamping / regularisation / depth of field / hidden lights /
//     white first-bounce albedo / transparent background / pre-transformed vertices) - what every BASELINE config renders with;
//     ahead-of-time instances in path_tracer.hip and shade_fast.hip;
//   * SpecMacros: whatever -DTR_SPEC_* say (shade_spec.hip: any other option set, compiled when a stage first needs it).
// S::pinned: the sampler is known at compile time; S::shade_tris: surface hits read the ShadeTri records (common.h); S::wide_textures:
// the scene may hold RGBA16 textures (false pins SceneView::wide_textures to 0: the one-dword texel fetch, texture.h).
struct SpecGeneral {
    static constexpr bool pinned = false, shade_tris = false, wide_textures = true;
    TR_DEV static void pin(PtParams&) {}
};
struct SpecCli {
    static constexpr bool pinned = true, shade_tris = true, wide_textures = false;
    TR_DEV static void pin(PtParams& P) {
        P.opt.sampler = 0; P.opt.film = 0; P.opt.mis_mode = 2; P.opt.bounce_mode = 2; P.opt.tri_light_mode = 1;
        P.opt.russian_roulette_delta = 0.0f; P.opt.indirect_clamping = 0.0f; P.opt.regularization_gamma = 0.0f;
        P.opt.depth_of_field = 0; P.opt.hide_lights = 0; P.opt.use_white_albedo_on_first_bounce = 0; P.opt.transparent_background = 0;
        P.opt.pre_transformed_vertices = 0;
    }
};
#ifdef TR_SPEC_SAMPLER
// One option set as macros (specialize.cc writes them from the stage's options; see spec_key there for the list).  The three
// thresholds keep their run-time values when they are in use; only "off" (0) is pinned, which is what removes code.
struct SpecMacros {
    static constexpr bool pinned = true, shade_tris = TR_SPEC_SHADE_TRIS != 0, wide_textures = TR_SPEC_WIDE_TEXTURES != 0;
    TR_DEV static void pin(PtParams& P) {
        P.opt.sampler = TR_SPEC_SAMPLER; P.opt.film = TR_SPEC_FILM; P.opt.mis_mode = TR_SPEC_MIS; P.opt.bounce_mode = TR_SPEC_BOUNCE_MODE;
        P.opt.tri_light_mode = TR_SPEC_TRI_LIGHT_MODE; P.opt.projection = TR_SPEC_PROJECTION;
        if (!TR_SPEC_ROULETTE) P.opt.russian_roulette_delta = 0.0f;
        if (!TR_SPEC_CLAMPING) P.opt.indirect_clamping = 0.0f;
        if (!TR_SPEC_REGULARIZATION) P.opt.regularization_gamma = 0.0f;
        P.opt.depth_of_field = TR_SPEC_DOF; P.opt.hide_lights = TR_SPEC_HIDE_LIGHTS; P.opt.use_white_albedo_on_first_bounce = TR_SPEC_WHITE_ALBEDO;
        P.opt.transparent_background = TR_SPEC_TRANSPARENT; P.opt.pre_transformed_vertices = TR_SPEC_PRE_TRANSFORMED;
        P.nee_point = TR_SPEC_NEE_POINT; P.nee_tri = TR_SPEC_NEE_TRI; P.nee_dir = TR_SPEC_NEE_DIR; P.nee_env = TR_SPEC_NEE_ENV;
        if (!TR_SPEC_NEE_POINT) P.prob_point = 0.0f;
        if (!TR_SPEC_NEE_TRI) P.prob_tri = 0.0f;
        if (!TR_SPEC_NEE_DIR) P.prob_dir = 0.0f;
        if (!TR_SPEC_NEE_ENV) P.prob_env = 0.0f;
    }
};
#endif
// What one bounce of one path leaves for the kernel that called it: whether the path goes on, and its shadow ray if it cast one
// (contrib *= shadow_ray(...) happens in the trace kernel, including the clamp on the occluded value).
struct ShadeOut {
    bool alive = false;          // continues to the next bounce
    bool want_shadow = false;
    f3 sh_o = {0, 0, 0}, sh_d = {0, 0, 0}, sh_c = {0, 0, 0};
    f2 sh_w = {0, 0};
    float sh_tmax = 0, sh_lum = 0;
};

// One bounce of evaluate_ray (path_tracer.glsl:385-498) for path `id`: reads the path's state and hit record, adds emission (with
// MIS) to the sample's demodulated sums, samples a light (the shadow ray goes to `o`) and the BSDF, writes the next ray back.
// P: the launch parameters with the option set already pinned (S::pin).
template <bool COUNT, bool LAST, typename S>
TR_DEV void shade_path(const SceneView& sv, const PtParams& P, const PathBuffers& pb, int bounce, uint id, u4 misc, ShadeOut& o, uint& surf) {
    bool& alive = o.alive;
    bool& want_shadow = o.want_shadow;
    f3 &sh_o = o.sh_o, &sh_d = o.sh_d, &sh_c = o.sh_c;
    f2& sh_w = o.sh_w;
    float &sh_tmax = o.sh_tmax, &sh_lum = o.sh_lum;
        const f4 o4 = pb.org_pdf[id], d4 = pb.dir_reg[id], a4 = pb.atten_alpha[id];
        const int4 h = pb.hit[id];
        f3 pos = F3(o4), view = F3(d4);
        float bsdf_pdf = o4.w, regularization = d4.w;
        f3 attenuation = F3(a4);
        // demodulated light of this sample: known to be zero before bounce 0, otherwise fetched only by the paths that add to
        // it in this kernel (emitters, envmap/light hits, NEE samples too dim for a shadow ray)
        f4 dif = F4(0), ref = F4(0);
        bool have = bounce == 0;
        f2 pl = bounce == 0 ? F2(0.0f, 1.0f) : pb.plobes[id];   // primary_lobes = (0,0,0,1) (path_tracer.glsl:383)
        u4 rs = pb.rng[id];
        STL(STL_STATE);
        // (payload.random_seed advances once per closest-hit trace: closest_lane derives the seed of its bounce from the one k_raygen
        // stored, so no kernel rewrites misc)

        // ---- get_intersection_info (path_tracer.glsl:91-201)
        SampledMaterial mat;
        mat.albedo = F4(0); mat.metallic = 1; mat.roughness = 0; mat.emission = F3(0);
        mat.transmittance = 0; mat.ior_in = 1; mat.ior_out = 1; mat.f0 = 0;
        SurfacePoint v;
        v.pos = pos; v.hard_normal = F3(0); v.smooth_normal = F3(0); v.mapped_normal = F3(0); v.tri_light_pdf = 0;
        float pl_pdf = 0, dl_pdf = 0, tri_pdf = 0, env_pdf = 0;
        f3 light = F3(0);
        bool surface = false;
        if (h.x >= 0) {
            surface = true;
            if (COUNT) surf++;
            shade_surface(sv, h.x, h.y, __int_as_float(h.z), __int_as_float(h.w), view, pos, P.nee_tri != 0, P.opt.tri_light_mode, P.opt.pre_transformed_vertices != 0, v, mat, S::shade_tris);
            mat.albedo.w = 1.0f;
            if (P.nee_tri) {
                tri_pdf = v.tri_light_pdf;
                light = mat.emission;
                mat.emission = F3(0);
            }
        } else if (h.y >= 0) {
            const PointLight pl = sv.point_lights[h.y];
            f3 c = get_spotlight_intensity(pl, view) * pl.color / (pl.radius * pl.radius * TR_PI);
            if (P.nee_point) { light = c; pl_pdf = sample_point_light_pdf(pl, pos); }
            else mat.emission = c;
            v.pos = pos + __int_as_float(h.z) * view;
            v.mapped_normal = normalize(v.pos - pl.pos);
            mat.albedo = F4(0, 0, 0, 1);
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
                if (P.nee_dir) { light += dc; dl_pdf += visible * sample_directional_light_pdf(dl); }
                else mat.emission += dc;
            }
            v.pos = pos;
            v.mapped_normal = -view;
            mat.albedo = F4(0);
            if (P.nee_env) {
                light += F3(c);
                env_pdf = sv.environment_proj >= 0 ? sample_environment_map_pdf(sv, view) : 0.0f;
            } else mat.emission += F3(c);
        }
        if (!surface) { STL(STL_NOSURFACE); }
        const bool terminal = LAST || !surface || bounce == P.opt.max_bounces - 1;

        // ---- emission with MIS (path_tracer.glsl:413-435)
        float mis_pdf = bsdf_mis_pdf(sv, P, pl_pdf, dl_pdf, tri_pdf, env_pdf, bsdf_pdf);
        float mis_weight = 1.0f;
        if (bsdf_pdf != 0) { attenuation = attenuation / bsdf_pdf; mis_weight = bsdf_pdf / mis_pdf; }
        light = attenuation * mis_weight * (mat.emission + light);
        if (bounce != 0) light *= clamp_contribution_mul(P, light);

        // add_demodulated_color(primary_lobes, light, diffuse, reflection) (path_tracer.glsl:435, material.glsl:66-73)
        if (bounce == 0 || light.x != 0.0f || light.y != 0.0f || light.z != 0.0f) {
            if (!have) { dif = pb.diffuse[id]; ref = pb.reflection[id]; have = true; }
            dif.x += light.x * pl.x; dif.y += light.y * pl.x; dif.z += light.z * pl.x;
            ref.x += light.x * pl.y; ref.y += light.y * pl.y; ref.z += light.z * pl.y;
        }
        if (bounce == 0) {   // first_hit_vertex / first_hit_material (path_tracer.glsl:437-442)
            pb.first_mat[id] = F4(F3(mat.albedo), mat.metallic);
            pb.first_emis[id] = F4(light, mat.albedo.w);
        }

        STL(STL_EMIT_MIS);
        if (P.opt.regularization_gamma != 0.0f) {   // PATH_SPACE_REGULARIZATION (path_tracer.glsl:437-444)
            if (bsdf_pdf != 0.0f) regularization *= fmax2(1 - P.opt.regularization_gamma / tpow(bsdf_pdf, 0.25f), 0.0f);
            mat.roughness = 1.0f - ((1.0f - mat.roughness) * regularization);
        }

        if (!terminal) {
            const m3 tbn = create_tangent_space(v.mapped_normal);
            const f3 shading_view = view_to_tangent_space(view, tbn);
            u4 coord = {0, 0, 0, 0};   // only the Sobol-Owen sampler hashes the launch coordinate again (uniform branch)
            if (P.opt.sampler == SAMPLER_SOBOL_OWEN) {
                uint lx, ly, lz;
                launch_coord(P.L, misc.z, lx, ly, lz);
                int px = 0, py = 0;
                get_pixel_pos(P.L, lx, ly, px, py);
                coord = u4{(uint)px, (uint)py, global_viewport(P, lz) + P.rng_seed, P.rng_sample + sample_counter_of(P, lz)};
            }
            // ---- next_event_estimation (path_tracer.glsl:302-344, 449-472)
            const bool any_nee = (P.nee_point && sv.point_light_count > 0) || (P.nee_dir && sv.directional_light_count > 0) ||
                                 (P.nee_tri && sv.tri_light_count > 0) || (P.nee_env && sv.environment_proj >= 0);
            u4 rnd = ray_sample_uint(rs, coord, misc.y, (uint)bounce * 2u, P.opt.sampler, P.max_sobol_bounces);
            Lobes lobes = {0, 0, 0, 0};
            if (any_nee) {
                f3 out_dir;
                float out_length = 0.0f, light_pdf;
                f3 contrib = sample_explicit_light(sv, P, rnd, v.pos, out_dir, out_length, light_pdf);
                STL(STL_LIGHT);
                f3 shading_light = mulT(out_dir, tbn);
                float nee_bsdf_pdf = material_bsdf_pdf(P.opt.bounce_mode, shading_light, shading_view, mat, lobes);
                correct_lobes_for_normal_map(out_dir, v.hard_normal, lobes);
                bool cast = contrib.x > 0.0001f || contrib.y > 0.0001f || contrib.z > 0.0001f;
                contrib = contrib / nee_mis_pdf(P, light_pdf, nee_bsdf_pdf);
                f3 radiance = attenuation * contrib;
                float clamp_lum = 0.0f;   // > 0: indirect clamping applies to (radiance * visibility)
                if (bounce != 0) {
                    radiance *= modulate_bsdf(mat, lobes);
                    if (P.opt.indirect_clamping > 0.0f) clamp_lum = rgb_to_luminance(radiance);
                } else {
                    // primary_lobes = lobes (path_tracer.glsl:466)
                    pl = F2(lobes.diffuse + lobes.transmission, lobes.dielectric_reflection + lobes.metallic_reflection);
                }
                if (cast) {
                    // contrib *= shadow_ray(...) happens in k_trace_shadow, including the clamp on the occluded value
                    want_shadow = true;
                    sh_o = v.pos; sh_d = out_dir; sh_tmax = out_length; sh_c = radiance; sh_lum = clamp_lum; sh_w = pl;
                } else {
                    float mul = (clamp_lum > P.opt.indirect_clamping && clamp_lum > 0.0f) ? P.opt.indirect_clamping / clamp_lum : 1.0f;
                    if (!have) { dif = pb.diffuse[id]; ref = pb.reflection[id]; have = true; }
                    const f3 r = radiance * mul;
                    dif.x += r.x * pl.x; dif.y += r.y * pl.x; dif.z += r.z * pl.x;
                    ref.x += r.x * pl.y; ref.y += r.y * pl.y; ref.z += r.z * pl.y;
                }
            }
            STL(STL_NEE_EVAL);
            if (bounce == 1) {   // diffuse.a = reflection.a = 1 / length(v.pos - pos) (path_tracer.glsl:470-471)
                const float inv_len = 1.0f / length(v.pos - pos);
                if (have) { dif.w = inv_len; ref.w = inv_len; }
                else { pb.diffuse[id].w = inv_len; pb.reflection[id].w = inv_len; }
            }
            // ---- BSDF sampling (path_tracer.glsl:475-497)
            Lobes bl = {0, 0, 0, 0};
            f4 ray_sample = u4_to_unit(ray_sample_uint(rs, coord, misc.y, (uint)bounce * 2u + 1u, P.opt.sampler, P.max_sobol_bounces));
            f3 new_dir;
            material_bsdf_sample(P.opt.bounce_mode, ray_sample, shading_view, mat, new_dir, bl, bsdf_pdf);
            view = mul(tbn, new_dir);
            correct_lobes_for_normal_map(v.hard_normal, view, bl);
            if (bounce != 0) attenuation *= modulate_bsdf(mat, bl);
            else pl = F2(bl.diffuse + bl.transmission, bl.dielectric_reflection + bl.metallic_reflection);   // primary_lobes = lobes
            pos = v.pos;
            alive = true;
            if (P.opt.russian_roulette_delta > 0) {   // USE_RUSSIAN_ROULETTE: the survivor weight is never applied
                float qi_ = fmin2(1.0f, 1.0f / P.opt.russian_roulette_delta);
                if (ray_sample.w > qi_) alive = false;
            }
            if (fmax2(attenuation.x, fmax2(attenuation.y, attenuation.z)) <= 0.0f) alive = false;
        }
        STL(STL_BSDF);
        // ---- write back
        if (have) { pb.diffuse[id] = dif; pb.reflection[id] = ref; }
        if (alive) {
            pb.org_pdf[id] = F4(pos, bsdf_pdf);
            pb.dir_reg[id] = F4(view, regularization);
            pb.atten_alpha[id] = F4(attenuation, 0);
            if (bounce == 0) pb.plobes[id] = pl;
            pb.rng[id] = rs;
        }
        STL(STL_WRITEBACK);
}

// The arguments of k_shade / trhip_spec_shade as they lie in the kernel-argument segment (by-value arguments in order, each at its
// natural alignment: the layout of this struct).  shade_bounce reads the scene view and the parameters through it (below).
struct ShadeKernArgs { SceneView sv; PtParams P; PathBuffers pb; int bounce; const uint* queue; uint* bc; uint* next_queue; };
#ifndef TR_SHADE_FRESH_ARGS
#define TR_SHADE_FRESH_ARGS 1
#endif
template <typename T>
TR_DEV void load_kernarg(T& dst, const T __attribute__((address_space(4)))* src) {      // word by word out of the constant address space
    static_assert(sizeof(T) % 4 == 0, "kernel arguments are whole words");
    uint* d = reinterpret_cast<uint*>(&dst);
    const uint __attribute__((address_space(4)))* s = (const uint __attribute__((address_space(4)))*)src;
#pragma unroll
    for (uint i = 0; i < sizeof(T) / 4; ++i) d[i] = s[i];
}

template <bool COUNT, bool LAST, typename S>
TR_DEV void shade_bounce(const SceneView& sv_, const PtParams& P_, const PathBuffers& pb, int bounce, const uint* queue, uint* bc, uint* next_queue) {
    PtParams P = P_;
    S::pin(P);
    SceneView sv = sv_;
    if (!S::wide_textures) sv.wide_textures = 0;
    // the Sobol index of the Z samplers lives in PathBuffers::misc
    const bool misc_needed = !S::pinned || P.opt.sampler == SAMPLER_SOBOL_Z2 || P.opt.sampler == SAMPLER_SOBOL_Z3;
    const uint n = queue ? bc[BC_QUEUE] : P.n_ids;
    const uint n_round = LAST ? n : ((n + (uint)KB - 1u) & ~((uint)KB - 1u));   // whole blocks take part in the appends
    uint surf = 0;
#if TR_SHADE_TIMELINE
    stl_begin();
#endif
    // A persistent grid that strides over the queue.  -DTR_SHADE_NOLOOP=1 (an experiment, profiles/r6/shade_noloop_ab.txt): one pass per thread,
    // launched with a thread per queue slot - the kernel alone is 28 % faster (126 registers, nothing hoisted, nothing spilled), but a grid of
    // one-pass blocks gives its slots away to the other lanes' persistent trace kernels as its blocks finish, and whole frames lose 1-5 %.
#if TR_SHADE_NOLOOP
    for (uint qi = blockIdx.x * KB + threadIdx.x, once = 1; once && qi < n_round; once = 0) {
#else
    for (uint qi = blockIdx.x * KB + threadIdx.x; qi < n_round; qi += gridDim.x * KB) {
#endif
#if TR_SHADE_FRESH_ARGS && !TR_SHADE_NOLOOP
        // Every pass of the loop reads the scene view and the parameters afresh, through a pointer to the kernel-argument segment the
        // optimiser cannot see through.  Round 6 (profiles/r6/shade_phase_timeline.txt, shade_noloop_ab.txt): around this loop the compiler
        // hoisted everything invariant - reciprocals of the light counts and of the environment's size, the pieces of uniform divisions -
        // into registers that stay live around the back edge, and spilled six of the kernel's 168 to scratch, each reload behind a full
        // `s_waitcnt vmcnt(0)` in the middle of the body's gathers.  With nothing to hoist the kernel needs 158 registers and no scratch:
        // k_shade 0.89 -> 0.795 ms per frame on sponza_teapots, frames -2 ... -4 % (profiles/r6/shade_fresh_args_ab.txt).  The loop itself
        // stays: a persistent grid keeps the slots it has while the other lanes' trace kernels run.  The argument layout this relies on
        // (by-value arguments in order at their natural alignment = struct ShadeKernArgs) is the AMDGPU kernel ABI; every parity test
        // renders through it.
        typedef const ShadeKernArgs __attribute__((address_space(4)))* KernArgPtr;      // the constant address space: scalar loads
        KernArgPtr ka = (KernArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        PtParams P;
        load_kernarg(P, &ka->P);
        S::pin(P);
        SceneView sv;
        load_kernarg(sv, &ka->sv);
        if (!S::wide_textures) sv.wide_textures = 0;
#endif
        STL(STL_ITER);
        STL(STL_CALIBRATION);
        bool active = qi < n;
        uint id = 0;
        u4 misc = {0, 0, 0, 1};
        if (active) {
            id = queue ? queue[qi] : qi + P.id_offset;
            // PathBuffers::misc is written by k_raygen only.  What this kernel wants from it on a queue-driven bounce - the path's
            // launch id - is the path id itself (the Sobol index is for the Sobol-Z samplers, the dead flag for bounce 0, where
            // the ids are all launch ids): 16 bytes per path and bounce not read when the sampler is known not to need them
            if (!misc_needed && queue) misc = u4{0u, 0u, id, 0u};
            else { misc = pb.misc[id]; active = !(misc.w & 1u); }
        }
        STL(STL_ID);
        ShadeOut o;
        if (active) shade_path<COUNT, LAST, S>(sv, P, pb, bounce, id, misc, o, surf);
        const bool alive = o.alive, want_shadow = o.want_shadow;
        // ---- queue compaction (wave ballots)
        if (LAST) continue;     // nothing survives the last bounce
        uint sslot, nslot;
        block_append2(&bc[BC_SHADOW], want_shadow, sslot, &bc[BC_STRIDE + BC_QUEUE], alive, nslot);
        STL(STL_APPEND);
        if (want_shadow) {
            pb.sh_org_tmax[sslot] = F4(o.sh_o, o.sh_tmax);
            pb.sh_dir_id[sslot] = F4(o.sh_d, __uint_as_float(id));
            pb.sh_contrib[sslot] = F4(o.sh_c, o.sh_lum);
            pb.sh_lobes[sslot] = o.sh_w;
        }
        if (alive) next_queue[nslot] = id;
        STL(STL_QUEUE);
    }
#if TR_SHADE_TIMELINE
    stl_end(bounce);
#endif
    if (COUNT && P.count_work) {
        for (int off = 32; off > 0; off >>= 1) surf += __shfl_xor(surf, off);
        if ((threadIdx.x & 63) == 0) add64(pb.counters, CNT_SURF, surf);
    }
}

// ---------------------------------------------------------------------------------------------------
// path_tracer.rgen:88-101 + get_world_camera_ray (path_tracer.glsl:504-533)
template <typename S>
TR_DEV void raygen_paths(const SceneView& sv, const PtParams& P_, const PathBuffers& pb) {
    PtParams P = P_;
    S::pin(P);
    if (blockIdx.x == 0) for (uint k = threadIdx.x; k < P.bounce_words; k += KB) pb.bounce[k] = 0;   // queue lengths and work cursors of this sample
    uint i = blockIdx.x * KB + threadIdx.x;
    if (i >= P.n_ids) return;
    i += P.id_offset;
    uint lx, ly, lz;
    launch_coord(P.L, i, lx, ly, lz);
    int px, py;
    bool valid = get_pixel_pos(P.L, lx, ly, px, py);
    u4 misc = {0, 0, i, valid ? 0u : 1u};
    if (P.sample_in_pass == 0 && !P.fused_resolve) {
        pb.sum_color[i] = F4(0, 0, 0, 1);
        if (pb.sum_diffuse) { pb.sum_diffuse[i] = F4(0); pb.sum_reflection[i] = F4(0); }
    }
    if (!valid) { pb.misc[i] = misc; return; }
    LocalSampler ls = init_local_sampler(u4{(uint)px, (uint)py, global_viewport(P, lz), P.rng_sample}, sample_counter_of(P, lz),
                                         P.rng_seed, P.opt.sampler);
    f2 cam_offset = F2(0.0f);
    if (P.opt.film != 0) {   // control.antialiasing == 1
        f4 r = u4_to_unit(pcg4d(ls.rs));   // generate_film_sample
        if (P.opt.film == 1) cam_offset = F2(r.x, r.y) * 2.0f - 1.0f;
        else cam_offset = sample_blackman_harris_concentric_disk(F2(r.x, r.y)) * 2.0f;
        cam_offset = cam_offset * (2.0f * P.opt.film_radius);
    }
    f2 dof_u = F2(0.5f);
    if (P.opt.depth_of_field) { f4 r = u4_to_unit(pcg4d(ls.rs)); dof_u = F2(r.x, r.y); }
    f3 origin, dir;
    get_screen_camera_ray(P.L, px, py, sv.cameras[global_viewport(P, lz)], P.opt.projection, P.opt.depth_of_field != 0, cam_offset, dof_u, origin, dir);
    misc.x = pcg4d(ls.rs).x;      // payload.random_seed = pcg4d(lsampler.rs.seed).x  (path_tracer.glsl:384)
    misc.y = ls.sobol_index;
    pb.org_pdf[i] = F4(origin, 0.0f);            // bsdf_pdf = 0
    pb.dir_reg[i] = F4(dir, 1.0f);               // regularization = 1
    pb.atten_alpha[i] = F4(1, 1, 1, 1);          // attenuation = 1
    pb.rng[i] = ls.rs;
    pb.misc[i] = misc;
}
template <typename S = SpecGeneral>
__global__ __launch_bounds__(KB) void k_raygen(SceneView sv, PtParams P, PathBuffers pb) { raygen_paths<S>(sv, P, pb); }

template <bool COUNT, bool LAST, typename S = SpecGeneral>
__global__ __launch_bounds__(KB, LAST ? TR_SHADE_LAST_WAVES : TR_SHADE_WAVES) void k_shade(SceneView sv, PtParams P, PathBuffers pb, int bounce, const uint* queue,
                                              uint* bc, uint* next_queue) {
    shade_bounce<COUNT, LAST, S>(sv, P, pb, bounce, queue, bc, next_queue);
}

}  // namespace

}  // namespace tr
