// tauray_hip.hh - C++17 host layer over the trhip C ABI, keeping Tauray's renderer/stage surface for the
// path-tracer hot path (header-only; link with -ltrhip).
//
// Class and function names, option fields and the render() sequence follow the reference:
//   distribution_strategy / distribution_params ... src/distribution_strategy.{hh,cc}
//   scene_stage .................................... src/scene_stage.{hh,cc} (uploads + acceleration structure)
//   path_tracer_stage (+ options) .................. src/path_tracer_stage.{hh,cc}, rt_camera_stage, rt_stage
//   stitch_stage, tonemap_stage .................... src/stitch_stage.{hh,cc}, src/tonemap_stage.{hh,cc}
//   rt_renderer .................................... src/rt_renderer.{hh,cc}: per-device stages, transfer, stitch, tonemap
//   headless ....................................... src/headless.{hh,cc}: readback, NaN report, file naming, EXR/RAW writers
//   load_balancer .................................. src/load_balancer.{hh,cc}
// Errors are thrown as std::runtime_error carrying trhip_last_error(), like the reference.
#ifndef TAURAY_HIP_HH
#define TAURAY_HIP_HH
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>
#ifdef TAURAY_HIP_WITH_ZLIB
#include <zlib.h>
#endif

#include "trhip.h"
#include "tauray_exr.hh"

namespace tr
{

inline void check(int rc)
{
    if(rc != 0) throw std::runtime_error(trhip_last_error());
}

struct uvec2 { uint32_t x = 0, y = 0; };

//==============================================================================
// distribution_strategy.hh
//==============================================================================
enum distribution_strategy
{
    DISTRIBUTION_DUPLICATE = 0,
    DISTRIBUTION_SCANLINE = 1,
    DISTRIBUTION_SHUFFLED_STRIPS = 2
};

struct distribution_params
{
    uvec2 size;
    distribution_strategy strategy = DISTRIBUTION_SCANLINE;
    unsigned index = 0;
    unsigned count = 1;
    bool primary = true;
};

inline uvec2 get_distribution_render_size(const distribution_params& p)
{
    switch(p.strategy)
    {
    case DISTRIBUTION_DUPLICATE: return p.size;
    case DISTRIBUTION_SCANLINE: return uvec2{p.size.x, (p.size.y - p.index + p.count - 1) / p.count};
    default: return uvec2{p.count, 1};
    }
}

inline uvec2 get_distribution_target_size(const distribution_params& p)
{
    if(p.primary) return p.size;
    if(p.strategy == DISTRIBUTION_SHUFFLED_STRIPS) return uvec2{p.size.x, (p.count + p.size.x - 1) / p.size.x};
    return get_distribution_render_size(p);
}

// src/distribution_strategy.cc:21-31 returns the frame size for shuffled strips.  The ids a device can be handed reach past the
// pixel count, though: the 2^b regions are padded to a common size (45 x 51 = 2 295 pixels are 16 regions of 144 = 2 304 ids), a
// device with (nearly) the whole frame gets `count` > pixels, and its partial image - `count` pixels in rows of size.x - is a row
// taller than the frame.  The reference's targets are Vulkan images, whose out-of-bounds stores are dropped and whose padded ids
// fall on no pixel anyway; a linear buffer needs the padded range, or the last valid pixels of such a share are written past its end
// (found by tests/test_cpp_host.py::test_cpp_random_multi_device_runs: workloads 0, 1 on two devices).
inline uvec2 get_distribution_target_max_size(const distribution_params& p)
{
    if(p.strategy == DISTRIBUTION_SHUFFLED_STRIPS)
    {
        unsigned n = p.size.x * p.size.y, b = 31;
        while((n >> b) < 128 && b > 0) b--;                  // calculate_shuffled_strips_b
        const size_t regions = size_t(1) << b, padded = ((size_t(n) + regions - 1) / regions) << b;
        return uvec2{p.size.x, (unsigned)((padded + p.size.x - 1) / p.size.x)};
    }
    return get_distribution_target_size(p);
}

inline uvec2 get_ray_count(const distribution_params& p)
{
    if(p.strategy == DISTRIBUTION_SHUFFLED_STRIPS) return uvec2{p.count, 1};
    return get_distribution_render_size(p);
}

inline unsigned calculate_shuffled_strips_b(uvec2 size)
{
    unsigned n = size.x * size.y;
    unsigned b = 31;
    while((n >> b) < 128 && b > 0) b--;
    return b;
}

inline unsigned calculate_shuffled_strips_pixels_per_device(uvec2 size, float max_ratio)
{
    unsigned b = calculate_shuffled_strips_b(size);
    size_t n_regions = size_t(1) << b;
    size_t region = (size_t(size.x) * size.y + n_regions - 1) / n_regions;
    return (unsigned)std::ceil(max_ratio * region * (1 << b));
}

inline distribution_params get_device_distribution_params(
    uvec2 full_image_size, distribution_strategy strategy, double workload_offset, double workload_size,
    unsigned device_index, unsigned device_count, bool primary
){
    distribution_params d;
    d.strategy = strategy;
    d.size = full_image_size;
    d.primary = primary;
    if(strategy == DISTRIBUTION_SHUFFLED_STRIPS)
    {
        unsigned before = calculate_shuffled_strips_pixels_per_device(full_image_size, workload_offset);
        unsigned after = calculate_shuffled_strips_pixels_per_device(full_image_size, workload_offset + workload_size);
        d.index = before;
        d.count = after - before;
    }
    else
    {
        d.index = device_index;
        d.count = device_count;
    }
    return d;
}

inline trhip_distribution to_abi(const distribution_params& p)
{
    return trhip_distribution{p.size.x, p.size.y, (int32_t)p.strategy, p.index, p.count, p.primary ? 1u : 0u};
}

//==============================================================================
// device (context holds one per HIP device; --fake-devices repeats a device, src/context.cc:415-416)
//==============================================================================
class device
{
public:
    explicit device(int hip_device): hip_device(hip_device) { check(trhip_device_create(hip_device, &h)); }
    device(const device&) = delete;
    ~device() { trhip_device_destroy(h); }

    void* alloc(size_t bytes) { void* p = nullptr; check(trhip_malloc(h, bytes, &p)); return p; }
    void free(void* p) { trhip_free(h, p); }
    void sync(void* stream = nullptr) { check(trhip_sync(h, stream)); }
    // Frame slots: a stream per frame in flight (MAX_FRAMES_IN_FLIGHT, src/context.hh:26); `dependencies` between stages on
    // different streams (src/dependency.hh) = stream_wait: `stream` continues once everything now on `on` has finished.
    void* create_stream() { void* st = nullptr; check(trhip_stream_create(h, &st)); return st; }
    void destroy_stream(void* stream) { check(trhip_stream_destroy(h, stream)); }
    void stream_wait(void* stream, void* on) { check(trhip_stream_wait(h, stream, on)); }

    trhip_device* h = nullptr;
    int hip_device;
};

//==============================================================================
// scene: the flattened scene_stage inputs (SURVEY.md Appendix A arrays)
//==============================================================================
struct gltf_animation;      // node tree + animation clips of a loaded glTF file (include/tauray_gltf.hh)

struct scene_data
{
    std::vector<uint8_t> instances, spans, vertices, indices, point_lights, directional_lights, texture_infos, texels,
        envmap, alias_table, cameras, non_opaque;
    uint32_t envmap_width = 0, envmap_height = 0;
    float environment_factor[4] = {0, 0, 0, 0};
    uint32_t gather_emissive_triangles = 0;
    uint32_t projection = 0;
    // Skinned vertex groups (mesh::get_skin + model::get_joints, src/mesh.hh:32-36, src/model.hh): the instance's vertices above are
    // the bind pose; `joint_transforms` (column-major mat4 per joint: global transform of the joint node * inverse bind matrix,
    // model::update_joints src/model.cc:107-118) is the pose scene_stage::set_scene applies - the file's rest pose when a loader
    // filled it in.  Not part of the .trsc dump.
    struct skinned_mesh { uint32_t instance = 0; std::vector<trhip_skin> skins; std::vector<float> joint_transforms; };
    std::vector<skinned_mesh> skinned;
    // What tr::scene_animator needs to play the file's animation clips (tauray_gltf.hh); null for scenes without a node tree.
    std::shared_ptr<gltf_animation> animation;
    std::vector<uint8_t> previous_cameras;       // camera_pair.previous of the current frame (empty = the cameras themselves)

    uint32_t instance_count() const { return (uint32_t)(instances.size() / 288); }
    uint32_t camera_count() const { return (uint32_t)(cameras.size() / 320); }
    uint32_t point_light_count() const { return (uint32_t)(point_lights.size() / 64); }
    uint32_t directional_light_count() const { return (uint32_t)(directional_lights.size() / 32); }
    bool has_tri_lights() const
    {
        for(uint32_t i = 0; i < instance_count(); ++i)
        {
            const float* e = reinterpret_cast<const float*>(instances.data() + 288 * i + 208 + 32);   // mat.emission_factor
            if(e[0] != 0 || e[1] != 0 || e[2] != 0) return true;
        }
        return false;
    }
};

// .trsc reader (written by tauray_amd/scene_io.py)
inline scene_data load_scene_dump(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if(!f) throw std::runtime_error("Failed to open " + path);
    char magic[4]; uint32_t version = 0;
    f.read(magic, 4); f.read(reinterpret_cast<char*>(&version), 4);
    if(std::memcmp(magic, "TRSC", 4) != 0 || version != 1) throw std::runtime_error(path + " is not a version-1 scene dump");
    scene_data s;
    std::vector<uint8_t>* sections[] = {&s.instances, &s.spans, &s.vertices, &s.indices, &s.point_lights, &s.directional_lights,
        &s.texture_infos, &s.texels, &s.envmap, &s.alias_table, &s.cameras, &s.non_opaque};
    for(auto* sec: sections)
    {
        uint64_t n = 0;
        f.read(reinterpret_cast<char*>(&n), 8);
        sec->resize(n);
        f.read(reinterpret_cast<char*>(sec->data()), (std::streamsize)n);
    }
    f.read(reinterpret_cast<char*>(&s.envmap_width), 4);
    f.read(reinterpret_cast<char*>(&s.envmap_height), 4);
    f.read(reinterpret_cast<char*>(s.environment_factor), 16);
    f.read(reinterpret_cast<char*>(&s.gather_emissive_triangles), 4);
    f.read(reinterpret_cast<char*>(&s.projection), 4);
    if(!f) throw std::runtime_error(path + " is truncated");
    return s;
}

class scene_stage
{
public:
    explicit scene_stage(device& dev): dev(&dev) {}

    void set_scene(const scene_data& s)
    {
        trhip_scene_desc d = {};
        d.instances = s.instances.data(); d.spans = s.spans.data(); d.instance_count = s.instance_count();
        d.vertices = s.vertices.data(); d.vertex_count = (uint32_t)(s.vertices.size() / 48);
        d.indices = reinterpret_cast<const uint32_t*>(s.indices.data()); d.index_count = (uint32_t)(s.indices.size() / 4);
        d.point_lights = s.point_lights.data(); d.point_light_count = s.point_light_count();
        d.directional_lights = s.directional_lights.data(); d.directional_light_count = s.directional_light_count();
        d.texture_infos = s.texture_infos.data(); d.texture_count = (uint32_t)(s.texture_infos.size() / 16);
        d.texels = s.texels.data();
        if(!s.envmap.empty())
        {
            d.envmap = reinterpret_cast<const float*>(s.envmap.data());
            d.envmap_width = s.envmap_width; d.envmap_height = s.envmap_height;
            d.alias_table = s.alias_table.data();
        }
        std::memcpy(d.environment_factor, s.environment_factor, 16);
        d.cameras = s.cameras.data(); d.camera_count = s.camera_count();
        d.non_opaque = s.non_opaque.data();
        d.gather_emissive_triangles = s.gather_emissive_triangles;
        check(trhip_scene_upload(dev->h, &d));
        // skinned meshes: the uploaded vertices are the bind pose; pose them before the build (the reference runs skinning.comp on
        // the first scene update, src/scene_stage.cc:1543-1567)
        for(const scene_data::skinned_mesh& sk: s.skinned)
        {
            set_skin(sk.instance, sk.skins.data(), (uint32_t)sk.skins.size());
            skin(sk.instance, sk.joint_transforms.data(), (uint32_t)(sk.joint_transforms.size() / 16));
        }
        check(trhip_scene_set_build_mode(dev->h, 0));     // first build of a scene: ePreferFastTrace
        check(trhip_scene_build_accel(dev->h, &accel));
    }

    // Per-frame scene changes of scene_stage::update (src/scene_stage.cc:1066-1116, 1543-1612).  `update_acceleration`
    // keeps the tree and recomputes its boxes (a BLAS/TLAS update) or, with `rebuild`, builds it again on the device.
    void update_instances(const void* instances_288, uint32_t count) { check(trhip_scene_update_instances(dev->h, instances_288, count)); }
    void set_previous_cameras(const void* camera_data_320, uint32_t count) { check(trhip_scene_set_previous_cameras(dev->h, camera_data_320, count)); }
    // mesh(mesh* animation_source) + mesh::skin_data: bind pose (nullptr = the uploaded vertices) and one skin per vertex
    void set_skin(uint32_t instance, const trhip_skin* skins, uint32_t vertex_count, const void* source_vertices_48 = nullptr)
    {
        check(trhip_scene_set_skin(dev->h, instance, source_vertices_48, skins, vertex_count));
    }
    // model::update_joints + shader/skinning.comp: column-major mat4 per joint
    void skin(uint32_t instance, const float* joint_transforms, uint32_t joint_count) { check(trhip_scene_skin(dev->h, instance, joint_transforms, joint_count)); }
    void update_cameras(const void* camera_data_320, uint32_t count) { check(trhip_scene_update_cameras(dev->h, camera_data_320, count)); }
    // One frame's worth of scene changes (scene_stage::update after update(scene, dt), src/scene.cc:226-235): the instance records,
    // joint matrices and cameras of `s` - as tr::scene_animator::update left them - go to the device and the acceleration
    // structure is updated once.
    void apply(const scene_data& s, bool rebuild = false)
    {
        update_instances(s.instances.data(), s.instance_count());
        for(const scene_data::skinned_mesh& sk: s.skinned) skin(sk.instance, sk.joint_transforms.data(), (uint32_t)(sk.joint_transforms.size() / 16));
        update_cameras(s.cameras.data(), s.camera_count());
        if(!s.previous_cameras.empty()) set_previous_cameras(s.previous_cameras.data(), (uint32_t)(s.previous_cameras.size() / 320));
        if(s.point_light_count() || s.directional_light_count())
            check(trhip_scene_update_lights(dev->h, s.point_lights.data(), s.point_light_count(), s.directional_lights.data(), s.directional_light_count()));
        update_acceleration(rebuild);
    }
    void update_acceleration(bool rebuild = false)
    {
        // geometry that is rebuilt after its first build is dynamic: ePreferFastBuild (src/acceleration_structure.cc:129-131)
        if(rebuild) check(trhip_scene_set_build_mode(dev->h, 1));
        check(rebuild ? trhip_scene_build_accel(dev->h, &accel) : trhip_scene_refit_accel(dev->h, &accel));
    }

    device* dev;
    trhip_accel_info accel = {};
};

//==============================================================================
// rt_common.hh enums + path_tracer_stage
//==============================================================================
enum class film_filter { POINT = 0, BOX, BLACKMAN_HARRIS };
enum class multiple_importance_sampling_mode { MIS_DISABLED, MIS_BALANCE_HEURISTIC, MIS_POWER_HEURISTIC };
enum class bounce_sampling_mode { HEMISPHERE, COSINE_HEMISPHERE, MATERIAL };
enum class tri_light_sampling_mode { AREA, SOLID_ANGLE, HYBRID };
enum class sampler_type { UNIFORM_RANDOM = 0, SOBOL_OWEN, SOBOL_Z_ORDER_2D, SOBOL_Z_ORDER_3D };

struct light_sampling_weights
{
    float point_lights = 1.0f;
    float directional_lights = 1.0f;
    float envmap = 1.0f;
    float emissive_triangles = 1.0f;
};

class path_tracer_stage
{
public:
    // rt_stage::options + rt_camera_stage::options + path_tracer_stage::options; defaults are the reference's
    // CLI defaults (src/options.hh), not the struct defaults, because that is what `tauray scene.glb` renders with.
    struct options
    {
        int max_ray_depth = 8;
        float min_ray_dist = 1e-4f;
        int rng_seed = 0;
        sampler_type local_sampler = sampler_type::UNIFORM_RANDOM;
        distribution_params distribution;
        size_t active_viewport_count = 1;
        int samples_per_pixel = 1;
        int samples_per_pass = 1;
        int projection = 0;                 // camera::projection_type
        bool transparent_background = false;
        bool use_white_albedo_on_first_bounce = false;
        bool hide_lights = false;
        bool pre_transformed_vertices = false;   // --pre-transform-vertices (src/options.hh)
        film_filter film = film_filter::POINT;
        multiple_importance_sampling_mode mis_mode = multiple_importance_sampling_mode::MIS_POWER_HEURISTIC;
        float film_radius = 0.5f;
        float russian_roulette_delta = 0;
        float indirect_clamping = 0;
        float regularization_gamma = 0.0f;
        bool depth_of_field = false;
        light_sampling_weights sampling_weights;
        bounce_sampling_mode bounce_mode = bounce_sampling_mode::MATERIAL;
        tri_light_sampling_mode tri_light_mode = tri_light_sampling_mode::SOLID_ANGLE;
    };

    path_tracer_stage(device& dev, scene_stage& ss, void* color_target, const options& opt, bool direct_only = false)
    : dev(&dev), ss(&ss), color(color_target), opt(opt)
    {
        trhip_pt_options o = {};
        o.max_bounces = opt.max_ray_depth; o.min_ray_dist = opt.min_ray_dist; o.rng_seed = (uint32_t)opt.rng_seed;
        o.sampler = (int)opt.local_sampler; o.samples_per_pixel = opt.samples_per_pixel; o.samples_per_pass = opt.samples_per_pass;
        o.projection = opt.projection; o.film = (int)opt.film; o.film_radius = opt.film_radius; o.mis_mode = (int)opt.mis_mode;
        o.russian_roulette_delta = opt.russian_roulette_delta; o.indirect_clamping = opt.indirect_clamping;
        o.regularization_gamma = opt.regularization_gamma; o.depth_of_field = opt.depth_of_field;
        o.nee_point = opt.sampling_weights.point_lights; o.nee_directional = opt.sampling_weights.directional_lights;
        o.nee_envmap = opt.sampling_weights.envmap; o.nee_triangles = opt.sampling_weights.emissive_triangles;
        o.bounce_mode = (int)opt.bounce_mode; o.tri_light_mode = (int)opt.tri_light_mode; o.hide_lights = opt.hide_lights;
        o.use_white_albedo_on_first_bounce = opt.use_white_albedo_on_first_bounce;
        o.transparent_background = opt.transparent_background;
        o.pre_transformed_vertices = opt.pre_transformed_vertices;
        check(direct_only ? trhip_direct_create(dev.h, &o, &pt) : trhip_pt_create(dev.h, &o, &pt));
        reset_distribution_params(opt.distribution);
    }
    path_tracer_stage(const path_tracer_stage&) = delete;
    ~path_tracer_stage() { trhip_pt_destroy(pt); }

    void reset_accumulated_samples() { check(trhip_pt_reset_accumulation(pt, 0)); }
    void reset_sample_counter() { check(trhip_pt_reset_accumulation(pt, 1)); }
    void reset_distribution_params(distribution_params distribution)
    {
        opt.distribution = distribution;
        trhip_distribution d = to_abi(distribution);
        check(trhip_pt_set_distribution(pt, &d));
    }
    // one stage per frame slot: slot k of F renders frames k, k + F, ... (rt_stage::frame_counter, src/rt_stage.cc:81-86)
    void set_frame_counter(uint32_t frame_counter) { check(trhip_pt_set_frame_counter(pt, frame_counter)); }
    // `frames` consecutive frames per run(): the target holds frames * active_viewport_count layers, frame-major (trhip_pt_set_frame_batch)
    void set_frame_batch(uint32_t frames) { check(trhip_pt_set_frame_batch(pt, frames)); frame_batch = frames; }
    // slices of a frame run concurrently inside the stage: 0 = automatic, 1 = none (several frames in flight instead)
    void set_lanes(int lanes) { check(trhip_pt_set_lanes(pt, lanes)); }
    void set_frame_slots(int slots) { check(trhip_pt_set_frame_slots(pt, slots)); }
    // the last pass of a frame also writes tonemap(colour) into `display` (a renderer with nothing between this stage and its tonemap stage)
    void set_fused_tonemap(void* display, const trhip_tonemap_info* info) { check(trhip_pt_set_fused_tonemap(pt, display, info)); }
    // which shading program renders this stage (general kernels / the command-line set's ahead-of-time instances / compiled for the option
    // set), resolved now, and its identity - what the devices of a job compare before the first frame (trhip_pt_get_program)
    trhip_program_info program() const { trhip_program_info p; check(trhip_pt_get_program(pt, &p)); return p; }
    // view / sample shard of a multi-device job: global viewport and sample addressing
    void set_shard(uint32_t viewport_base, uint32_t viewport_stride, uint32_t sample_base = 0, uint32_t sample_stride = 1)
    {
        check(trhip_pt_set_shard(pt, viewport_base, viewport_stride, sample_base, sample_stride));
    }
    // stage::run: enqueue the frame (all passes) on `stream` (default: the device's stream)
    void run(void* stream = nullptr)
    {
        uvec2 ts = get_distribution_target_size(opt.distribution);
        check(trhip_pt_render(pt, color, ts.x, ts.y, (uint32_t)opt.active_viewport_count * frame_batch, stream));
    }
    // the same frame into a gbuffer (src/gbuffer.hh: the entries path_tracer.rgen writes); null members are skipped
    using gbuffer_target = trhip_pt_targets;
    void run(const gbuffer_target& targets)
    {
        uvec2 ts = get_distribution_target_size(opt.distribution);
        check(trhip_pt_render_targets(pt, &targets, ts.x, ts.y, (uint32_t)opt.active_viewport_count, nullptr));
    }
    float get_duration_ms() { trhip_timings t; check(trhip_pt_get_timings(pt, &t)); return t.path_tracing_ms; }
    trhip_counters get_counters() { trhip_counters c; check(trhip_pt_get_counters(pt, &c)); return c; }

    device* dev;
    scene_stage* ss;
    void* color;
    options opt;
    trhip_pt* pt = nullptr;
    uint32_t frame_batch = 1;
};

// direct_stage (src/direct_stage.{hh,cc}): first hit + samples_per_pass light samples, same surface as path_tracer_stage.
// Its reference defaults differ (Blackman-Harris film of radius 1, hybrid tri-light sampling): set them in `options`.
class direct_stage: public path_tracer_stage
{
public:
    direct_stage(device& dev, scene_stage& ss, void* color_target, const options& opt)
    : path_tracer_stage(dev, ss, color_target, opt, true) {}
};

//==============================================================================
// tonemap_stage / load_balancer
//==============================================================================
class tonemap_stage
{
public:
    enum operator_type { LINEAR = 0, GAMMA_CORRECTION, FILMIC, REINHARD, REINHARD_LUMINANCE };
    struct options
    {
        operator_type tonemap_operator = FILMIC;
        float exposure = 1.0f;
        float gamma = 2.2f;
        bool alpha_grid_background = false;
    };
    tonemap_stage(device& dev, const options& opt): dev(&dev), opt(opt) {}
    trhip_tonemap_info info() const { return {(int32_t)opt.tonemap_operator, opt.exposure, opt.gamma, opt.alpha_grid_background ? 16 : 0}; }
    void run(const void* in, void* out, uvec2 size, uint32_t layers, void* stream = nullptr)
    {
        const trhip_tonemap_info i = info();
        check(trhip_tonemap(dev->h, in, out, size.x, size.y, layers, &i, stream));
    }
    device* dev;
    options opt;
};

class load_balancer
{
public:
    explicit load_balancer(size_t device_count, std::vector<double> initial = {}): workloads(std::move(initial))
    {
        workloads.resize(device_count);
        double sum = 0, add = 0;
        for(double w: workloads) sum += w;
        if(sum == 0) { add = 1.0; sum = (double)workloads.size(); }
        for(double& w: workloads) w = (std::max(w, 0.0) + add) / sum;
    }
    // times[i] = "path tracing" timer of device i
    const std::vector<double>& update(const std::vector<double>& times)
    {
        double sum_speed = 0;
        for(size_t i = 0; i < workloads.size(); ++i) sum_speed += std::max(workloads[i] / times[i], 0.0);   // a zero time gives inf: no update
        if(sum_speed > 0 && std::isfinite(sum_speed))
            for(size_t i = 0; i < workloads.size(); ++i)
                workloads[i] = workloads[i] * 0.9 + (workloads[i] / times[i]) / sum_speed * 0.1;
        return workloads;
    }
    std::vector<double> workloads;
};

//==============================================================================
// rt_renderer<Pipeline> (src/rt_renderer.hh:28-77): all devices in one process, like the reference.  Pipeline is
// path_tracer_stage or direct_stage (the reference instantiates the template for those, src/rt_renderer.cc:410-412);
// `rt_renderer` and `direct_renderer` below are the two instantiations (src/rt_renderer.hh:75-77).
//
// render() never blocks the host, like the reference's (src/rt_renderer.cc:84-133, src/stage.cc:35-76): every device has
// one stream per frame slot; a device's path tracing and the peer copy of its partial frame go onto its slot stream, the
// display device's slot stream waits for those streams (trhip_stream_wait_peer = the `dependencies` the reference hands
// from stage to stage), then stitches every partial in one launch and tonemaps.  The next frame of a slot starts on a
// non-display device only after the display device has stitched the slot's previous frame (the receive buffer is free
// again) - a stream dependency as well.  Frame slots (MAX_FRAMES_IN_FLIGHT, src/context.hh:26) work with any number of
// devices.
//==============================================================================
template<typename Pipeline>
class basic_rt_renderer
{
public:
    struct options: path_tracer_stage::options
    {
        tonemap_stage::options tonemap;
        bool accumulate = false;
        // Frame slots: frame i renders, is gathered and tonemapped on the streams of slot i % N while its predecessors are
        // still running; `display` and finish_frame() refer to the frame render() was last called for, frame_slots[k] to
        // the others.
        int max_frames_in_flight = 1;
        // Frames per launch: B > 1 makes every render() call B consecutive frames (trhip_pt_set_frame_batch): every image holds
        // B * active_viewport_count layers, frame-major, and `display` B tonemapped frames.  For frames that do not accumulate.
        // Bigger launches: less of a frame is the tail of its kernels (the pipelined figure of bench.py uses two).
        int frames_per_launch = 1;
    };

    // `devices`: HIP device index per logical device (repeat an index for --fake-devices); device 0 displays.
    basic_rt_renderer(const std::vector<int>& devices, const scene_data& scene, uvec2 size, options opt)
    : size(size), opt(opt)
    {
        if(devices.empty()) throw std::runtime_error("rt_renderer needs at least one device");
        if(devices.size() == 1) this->opt.distribution.strategy = DISTRIBUTION_DUPLICATE;   // src/tauray.cc:519-521
        const int n_slots = std::max(this->opt.max_frames_in_flight, 1);
        if(n_slots > 1 && this->opt.accumulate)
            throw std::runtime_error("rt_renderer: accumulating frames depend on each other, frames in flight must be 1");
        batch = (uint32_t)std::max(this->opt.frames_per_launch, 1);
        if(batch > 1 && this->opt.accumulate)
            throw std::runtime_error("rt_renderer: accumulating frames depend on each other, frames per launch must be 1");
        per_device.resize(devices.size());
        std::vector<double> ratios(devices.size(), 1.0 / devices.size());
        double cumulative = 0;
        const size_t layers = this->opt.active_viewport_count * batch;
        display_bytes = size_t(size.x) * size.y * 16 * layers;
        for(size_t i = 0; i < devices.size(); ++i)
        {
            per_device_data& d = per_device[i];
            d.dev = std::make_unique<device>(devices[i]);
            d.scene_update = std::make_unique<scene_stage>(*d.dev);
            d.scene_update->set_scene(scene);               // scene replicated on every device (src/gpu_buffer.hh:63-116)
            d.dist = get_device_distribution_params(size, this->opt.distribution.strategy, cumulative, ratios[i], (unsigned)i,
                                                    (unsigned)devices.size(), i == 0);
            cumulative += ratios[i];
            // non-primary targets are allocated for the largest share set_device_workloads can hand the device
            // (get_distribution_target_max_size, src/rt_renderer.cc init_resources)
            const uvec2 ms = get_distribution_target_max_size(d.dist);
            d.max_bytes = size_t(ms.x) * ms.y * 16 * layers;
            d.slots.resize((size_t)n_slots);
            for(slot_data& sl: d.slots)
            {
                sl.stream = d.dev->create_stream();
                sl.color = d.dev->alloc(d.max_bytes);
                check(trhip_memset(d.dev->h, sl.color, 0, d.max_bytes, nullptr));
                path_tracer_stage::options po = this->opt;
                po.distribution = d.dist;
                sl.ray_tracer = std::make_unique<Pipeline>(*d.dev, *d.scene_update, sl.color, po);
                if(n_slots > 1) { sl.ray_tracer->set_frame_slots(n_slots); }   // the frames in flight fill the chip between them
                if(batch > 1) sl.ray_tracer->set_frame_batch(batch);
                if(i != 0) sl.gbuffer_copy = per_device[0].dev->alloc(d.max_bytes);   // receive buffer on the display device
            }
            d.dev->sync();
        }
        frame_slots.resize((size_t)n_slots);
        for(frame_slot& fs: frame_slots) fs.display = per_device[0].dev->alloc(display_bytes);
        display = frame_slots[0].display;
        tonemap = std::make_unique<tonemap_stage>(*per_device[0].dev, this->opt.tonemap);
        // One device: nothing sits between the path tracer and the tonemap stage (no transfer, no stitch), and the stage writes the slot's
        // display image while it writes its colour target - the same bits without a second pass over the frame.  TRHIP_FUSED_TONEMAP=0: off.
        const char* fe = getenv("TRHIP_FUSED_TONEMAP");
        fused_tonemap = std::is_same<Pipeline, path_tracer_stage>::value && per_device.size() == 1 && !(fe && atoi(fe) == 0);
        fused_info.assign(frame_slots.size(), trhip_tonemap_info{-1, 0.0f, 0.0f, 0});     // what each slot's stage was last told: render() keeps it current
    }

    ~basic_rt_renderer()
    {
        finish_all();
        for(size_t i = 0; i < per_device.size(); ++i)
            for(slot_data& sl: per_device[i].slots)
            {
                sl.ray_tracer.reset();
                if(sl.gbuffer_copy) per_device[0].dev->free(sl.gbuffer_copy);
                per_device[i].dev->free(sl.color);
                per_device[i].dev->destroy_stream(sl.stream);
            }
        for(frame_slot& fs: frame_slots) per_device[0].dev->free(fs.display);
    }

    void reset_accumulation(bool reset_sample_counter = false)
    {
        for(auto& d: per_device)
            for(slot_data& sl: d.slots)
            {
                sl.ray_tracer->reset_accumulated_samples();
                if(reset_sample_counter) sl.ray_tracer->reset_sample_counter();
            }
        if(reset_sample_counter) frame_index = 0;
        accumulated_frames = 0;
    }

    // waits for the frame of the last render() call (all of it: path tracing, gather, stitch and tonemap end on the
    // display device's slot stream)
    void finish_frame() { if(current_slot >= 0) finish_slot(current_slot); }
    void finish_slot(int k) { per_device[0].dev->sync(per_device[0].slots[(size_t)k].stream); }
    void finish_all()
    {
        for(auto& d: per_device) { for(slot_data& sl: d.slots) d.dev->sync(sl.stream); d.dev->sync(); }
    }

    // The scene changed (an animation step): every device's copy follows (the scene is replicated, src/gpu_buffer.hh:63-116).
    // Frames in flight read the old scene: they are finished first.
    void update_scene(const scene_data& s, bool rebuild = false)
    {
        finish_all();
        for(auto& d: per_device) d.scene_update->apply(s, rebuild);
    }

    // rt_renderer::render (src/rt_renderer.cc:84-133): ray tracers -> transfers -> stitch -> tonemap.  Enqueues only.
    void render()
    {
        const size_t k = (frame_index / batch) % frame_slots.size();
        current_slot = (int)k;
        const uint32_t layers = (uint32_t)opt.active_viewport_count * batch;
        device& display_device = *per_device[0].dev;
        void* const display_stream = per_device[0].slots[k].stream;
        if(fused_tonemap)
        {   // the stage's copy of the tonemap parameters follows tonemap->opt: an edit between frames takes effect on the next frame,
            // as it does when the tonemap stage itself runs (several devices)
            const trhip_tonemap_info ti = tonemap->info();
            if(std::memcmp(&ti, &fused_info[k], sizeof(ti)) != 0)
            {
                per_device[0].slots[k].ray_tracer->set_fused_tonemap(frame_slots[k].display, &ti);
                fused_info[k] = ti;
            }
        }
        for(size_t i = 0; i < per_device.size(); ++i)
        {
            per_device_data& d = per_device[i];
            slot_data& sl = d.slots[k];
            if(!opt.accumulate) sl.ray_tracer->reset_accumulated_samples();
            if(frame_slots.size() > 1 || batch > 1) sl.ray_tracer->set_frame_counter(frame_index);   // one stage per slot: slot k renders frames k, k + N, ... (B at a time)
            if(i != 0)   // the slot's previous frame has been stitched on the display device: its receive buffer is free
                check(trhip_stream_wait_peer(d.dev->h, sl.stream, display_device.h, display_stream));
            sl.ray_tracer->run(sl.stream);
            if(i != 0) check(trhip_copy_peer(display_device.h, sl.gbuffer_copy, d.dev->h, sl.color, d.target_bytes(layers), sl.stream));
        }
        if(per_device.size() > 1)
        {
            std::vector<trhip_distribution> dists;
            std::vector<const void*> partials;
            std::vector<uint32_t> ws, hs;
            for(size_t i = 1; i < per_device.size(); ++i)
            {
                per_device_data& d = per_device[i];
                check(trhip_stream_wait_peer(display_device.h, display_stream, d.dev->h, d.slots[k].stream));
                const uvec2 ts = get_distribution_target_size(d.dist);
                dists.push_back(to_abi(d.dist)); partials.push_back(d.slots[k].gbuffer_copy); ws.push_back(ts.x); hs.push_back(ts.y);
            }
            check(trhip_stitch_batch(display_device.h, (uint32_t)dists.size(), dists.data(), partials.data(), ws.data(), hs.data(),
                                     per_device[0].slots[k].color, layers, stitch_blend_ratio, display_stream));
            stitch_blend_ratio = 1.0f;      // src/rt_renderer.cc:122
        }
        display = frame_slots[k].display;
        if(!fused_tonemap) tonemap->run(per_device[0].slots[k].color, display, size, layers, display_stream);
        frame_index += batch;
        accumulated_frames++;
    }

    // rt_renderer::set_device_workloads (src/rt_renderer.cc:135-183): only shuffled strips can be re-balanced
    void set_device_workloads(const std::vector<double>& ratios)
    {
        if(opt.distribution.strategy != DISTRIBUTION_SHUFFLED_STRIPS) return;
        finish_all();
        double cumulative = 0;
        for(size_t i = 0; i < per_device.size(); ++i)
        {
            double ratio = std::min(std::max(ratios[i], 0.0), 1.0 - cumulative);
            per_device[i].dist = get_device_distribution_params(size, opt.distribution.strategy, cumulative, ratio, (unsigned)i,
                                                                (unsigned)per_device.size(), i == 0);
            cumulative += ratio;
            for(slot_data& sl: per_device[i].slots)
            {
                sl.ray_tracer->reset_distribution_params(per_device[i].dist);
                if(i != 0) sl.ray_tracer->reset_accumulated_samples();
            }
        }
        // the non-primary devices start over with one sample: blend their pixels into what the display device has
        // accumulated instead of replacing it (src/rt_renderer.cc:176-181)
        if(opt.accumulate && per_device.size() > 1) stitch_blend_ratio = 1.0f / float(accumulated_frames + 1);
    }

    // "path tracing" timers of the most recent frame, one per device (waits for them)
    std::vector<double> get_path_tracing_times()
    {
        std::vector<double> t;
        const size_t k = current_slot < 0 ? 0 : (size_t)current_slot;
        for(auto& d: per_device) t.push_back(d.slots[k].ray_tracer->get_duration_ms());
        return t;
    }

    struct slot_data
    {
        void* stream = nullptr;
        std::unique_ptr<Pipeline> ray_tracer;
        void* color = nullptr;          // the device's (partial) colour target of this slot
        void* gbuffer_copy = nullptr;   // non-primary devices: where the partial lands on the display device
    };
    struct per_device_data
    {
        std::unique_ptr<device> dev;
        std::unique_ptr<scene_stage> scene_update;
        distribution_params dist;
        size_t max_bytes = 0;
        std::vector<slot_data> slots;
        size_t target_bytes(size_t layers) const { const uvec2 ts = get_distribution_target_size(dist); return size_t(ts.x) * ts.y * 16 * layers; }
    };
    std::vector<per_device_data> per_device;
    struct frame_slot { void* display = nullptr; };   // tonemapped RGBA32F on the display device
    std::vector<frame_slot> frame_slots;   // options.max_frames_in_flight of them (at least one)
    int current_slot = -1;
    uint32_t batch = 1;                    // options.frames_per_launch
    uint32_t frame_index = 0;
    uvec2 size;
    options opt;
    void* display = nullptr;          // frame_slots[current_slot].display
    size_t display_bytes = 0;
    std::unique_ptr<tonemap_stage> tonemap;
    bool fused_tonemap = false;
    std::vector<trhip_tonemap_info> fused_info;
    unsigned accumulated_frames = 0;
    float stitch_blend_ratio = 1.0f;
};
using rt_renderer = basic_rt_renderer<path_tracer_stage>;       // path_tracer_renderer
using direct_renderer = basic_rt_renderer<direct_stage>;        // direct_renderer

//==============================================================================
// headless (src/headless.{hh,cc}): readback + writers.  EXR: scanline file, channels B,G,R[,A] like the reference's,
// half or float, every codec of src/headless.hh:25-32 (include/tauray_exr.hh); PIZ is the default as in the reference
// (src/headless.hh:56).
//==============================================================================
class headless
{
public:
    enum image_file_type { EXR = 0, RAW, EMPTY };
    enum pixel_format { RGB16, RGB32, RGBA16, RGBA32 };
    enum compression_type { NONE = 0, RLE = 1, ZIPS = 2, ZIP = 3, PIZ = 4 };   // values = OpenEXR compression codes (src/headless.hh:25-32)

    struct options
    {
        uvec2 size;
        std::string output_prefix = "capture";
        image_file_type output_file_type = EXR;
        pixel_format output_format = RGB16;
        compression_type output_compression = PIZ;      // src/headless.hh:56
        bool single_frame = false;
        bool skip_nan_check = false;
        unsigned first_frame_index = 0;
        unsigned display_count = 1;
        // a view shard (one process per GPU, viewport v on rank v mod N): local layer l is display display_index_base + l *
        // display_index_stride of display_count_total (0 = display_count) - what the file names are made of
        unsigned display_index_base = 0, display_index_stride = 1, display_count_total = 0;
    };

    explicit headless(const options& opt): opt(opt) {}

    static uint16_t float_to_half(float f) { return exr::float_to_half(f); }

    std::string get_filename(unsigned display_index, unsigned frame_number) const
    {
        std::string filename = opt.output_prefix;                                       // src/headless.cc:305-309
        if((opt.display_count_total ? opt.display_count_total : opt.display_count) > 1)
            filename += std::to_string(opt.display_index_base + display_index * opt.display_index_stride) + "_";
        if(!opt.single_frame) filename += std::to_string(frame_number);
        return filename + (opt.output_file_type == EXR ? ".exr" : ".raw");
    }

    // finish_image + save_image for every display layer; returns the number of NaN pixels reported
    size_t save(device& dev, const void* display_image, unsigned frame_number)
    {
        const size_t pixels = size_t(opt.size.x) * opt.size.y;
        if(opt.output_file_type == EMPTY && opt.skip_nan_check) return 0;      // nothing to look at, nothing to write: no readback
        std::vector<float> mem(pixels * 4 * opt.display_count);
        check(trhip_download(dev.h, mem.data(), display_image, mem.size() * 4, nullptr));
        size_t nan_pixels = 0;
        for(unsigned d = 0; d < opt.display_count; ++d)
        {
            const float* img = mem.data() + pixels * 4 * d;
            if(!opt.skip_nan_check)
                for(size_t j = 0; j < pixels; ++j)
                    if(std::isnan(img[4 * j]) || std::isnan(img[4 * j + 1]) || std::isnan(img[4 * j + 2]) || std::isnan(img[4 * j + 3]))
                    {
                        std::fprintf(stderr, "NaN pixel at: %zu, %zu\n", j % opt.size.x, j / opt.size.x);
                        nan_pixels++;
                    }
            if(opt.output_file_type == EMPTY) continue;
            const std::string filename = get_filename(d, frame_number);
            if(opt.output_file_type == RAW) write_raw(filename, img, pixels);
            else write_exr(filename, img);
        }
        return nan_pixels;
    }

    // writes one display layer (RGBA32F, `img`) as save() would, without a device: for tools and tests
    void write_image(const std::string& filename, const float* img) const
    {
        if(opt.output_file_type == RAW) write_raw(filename, img, size_t(opt.size.x) * opt.size.y);
        else if(opt.output_file_type == EXR) write_exr(filename, img);
    }

    options opt;

private:
    void write_raw(const std::string& filename, const float* img, size_t pixels) const
    {
        std::ofstream f(filename, std::ios::binary);
        if(!f) throw std::runtime_error("Failed to write " + filename);
        f.write(reinterpret_cast<const char*>(img), (std::streamsize)(pixels * 16));
    }

    void write_exr(const std::string& filename, const float* img) const
    {
        const bool alpha = opt.output_format == RGBA16 || opt.output_format == RGBA32;
        const bool half = opt.output_format == RGB16 || opt.output_format == RGBA16;
        const std::vector<uint8_t> out = exr::encode(img, opt.size.x, opt.size.y, alpha, half, (int)opt.output_compression);
        std::ofstream f(filename, std::ios::binary);
        if(!f) throw std::runtime_error("Failed to write " + filename);
        f.write(reinterpret_cast<const char*>(out.data()), (std::streamsize)out.size());
    }
};

}

#endif
