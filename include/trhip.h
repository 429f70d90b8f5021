/* trhip - C ABI of the MI355X-native path-tracing core.
 *
 * This is the drop-in boundary for Tauray's path_tracer_stage hot path.  The
 * reference has no FFI: its extension point is C++ subclassing
 * (docs/DEVELOPERS.md:3-27; rt_renderer<Pipeline> in src/rt_renderer.hh:28-77).
 * Each entry point below names the reference interface it replaces; the
 * reference-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions: one opaque trhip_device per HIP device, all calls from one host
 * thread, work is enqueued on the caller's hipStream_t (passed as void*; NULL =
 * the default stream) and is asynchronous unless stated.  Every function
 * returns 0 on success, non-zero on failure with a message available from
 * trhip_last_error() (the reference throws std::runtime_error instead).
 * All POD layouts are the reference's GPU-side structs.
 */
#ifndef TRHIP_H
#define TRHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct trhip_device trhip_device;
typedef struct trhip_pt trhip_pt;

/* ---- device / memory (replaces tr::context + tr::device, src/context.hh, src/device.hh) */
int trhip_device_create(int hip_device, trhip_device** out);
void trhip_device_destroy(trhip_device* dev);
const char* trhip_last_error(void);
int trhip_malloc(trhip_device* dev, size_t bytes, void** out);           /* gpu_buffer (src/gpu_buffer.hh) */
int trhip_free(trhip_device* dev, void* ptr);
int trhip_upload(trhip_device* dev, void* dst_dev, const void* src_host, size_t bytes, void* stream);
int trhip_download(trhip_device* dev, void* dst_host, const void* src_dev, size_t bytes, void* stream); /* headless readback, src/headless.cc:292-303 */
int trhip_memset(trhip_device* dev, void* dst_dev, int value, size_t bytes, void* stream);
int trhip_sync(trhip_device* dev, void* stream);
/* Frames in flight (MAX_FRAMES_IN_FLIGHT, src/context.hh:26).  Every call takes the stream it is ordered on; a frame
 * slot is a stream plus the stages and images it owns, and `dependencies` between stages on different streams
 * (src/dependency.hh; timeline-semaphore waits in multi_device_stage::run, src/stage.cc:35-76) become
 * trhip_stream_wait: work enqueued on `stream` after the call starts only when everything enqueued on `on` before the
 * call has finished.  No host synchronisation.  NULL is the default stream. */
int trhip_stream_create(trhip_device* dev, void** stream_out);
int trhip_stream_destroy(trhip_device* dev, void* stream);
int trhip_stream_wait(trhip_device* dev, void* stream, void* on);
/* Streams are kept for the life of the process and handed out by the hardware pipe their queue sits on (csrc/stream_pool.hip: queues
 * of one pipe do not overlap launches larger than the chip, so the slots of a renderer and the lanes of a stage are spread over the
 * pipes).  trhip_stream_create returns an idle stream of the pipe with the fewest takers; trhip_stream_destroy hands it back.
 * `pipe_class_out`: the pipe of any stream of this process, the caller's own included (small integers from 0 in the order pipes
 * were seen; -1 with TRHIP_PIPE_PROBE=0): two streams of one class serialise chip-filling launches. */
int trhip_stream_pipe_class(trhip_device* dev, void* stream, int32_t* pipe_class_out);
/* ---- Process requirements (what the library needs from the process it is loaded into; nothing in the reference corresponds -
 * the Vulkan driver owns its queues, src/context.hh:26, src/stage.cc:35-76).
 *  1. GPU_MAX_HW_QUEUES >= 8 in the environment BEFORE the first HIP call of the process (the HIP runtime reads it once).  A lone
 *     frame is cut into four lanes on four streams that must sit on four different hardware pipes to overlap; with the runtime's
 *     default of four hardware queues the streams of one process share fewer pipes: sponza_teapots 1920x1080 renders in 4.55 ms
 *     instead of 3.69, a 1/8 strip in 1.0 instead of 0.75.  Frames are the same bits either way.  The
 *     Python mirror (tauray_amd/_lib.py), bench.py, the tests and the CLI (tauray_amd/host/tauray_hip_cli.cc) set it with
 *     setenv(.., overwrite = 0) before they touch HIP; an embedding application must do the same in its main() or launcher.
 *  2. Streams: create the streams you render on with trhip_stream_create (they come classified by pipe).  A stream of your own
 *     (or NULL - the default stream is classified by trhip_device_create) works too; the first stage that renders on it classifies it with a ~1 ms experiment if - and only if - the stream
 *     is idle and not capturing at that moment; otherwise its pipe stays unknown (-1) and the stage's lanes may share it.
 *     trhip_stream_pipe_class is the explicit form: it SYNCHRONISES `stream`, runs the experiment and remembers the result for
 *     that stream identity.  The library never synchronises a caller's stream anywhere else.
 *  3. trhip_device_get_info tells how many distinct pipes the process reaches (4 on MI355X with requirement 1 met; fewer = the
 *     -19 % case); with TRHIP_DEBUG=1 it and trhip_pt_render warn once on stderr when lanes share a pipe.
 *  4. TRHIP_PIPE_CLASSES=0,1,2,3 pins the classes of the library's streams in creation order and skips every experiment (for
 *     hosts that cannot afford the ~10 ms of probing at start-up or run under a tool that serialises queues);
 *     TRHIP_PIPE_PROBE=0 switches classification off altogether (every stream -1). */
typedef struct trhip_device_info {
    uint32_t struct_size;
    int32_t hip_device;
    char name[64];                    /* gcnArchName, e.g. "gfx950:sramecc+:xnack-" */
    char pci_bus_id[24];              /* "0000:05:00.0" */
    uint8_t uuid[16];
    int32_t compute_units;
    int32_t pipe_classes;             /* distinct hardware pipes the library's streams landed on (0: TRHIP_PIPE_PROBE=0) */
    int32_t pool_streams;             /* streams the library holds on this device (never destroyed) */
    int32_t hw_queues_env;            /* GPU_MAX_HW_QUEUES as this process sees it; 0 = unset (the runtime's default of 4) */
} trhip_device_info;
int trhip_device_get_info(trhip_device* dev, trhip_device_info* out);
/* The same dependency between streams of two devices of one process (the timeline semaphores the reference exports
 * between devices, src/device_transfer.cc:318-347, src/rt_renderer.cc:98-127): work enqueued on `stream` of `dev` after
 * the call starts only when everything enqueued on `on` of `on_dev` before the call has finished. */
int trhip_stream_wait_peer(trhip_device* dev, void* stream, trhip_device* on_dev, void* on);
/* device -> device copy over xGMI (replaces the pinned-host bounce of src/device_transfer.cc:140-290 when all
 * devices live in one process; with one process per GPU the same transfer is an RCCL send/recv) */
int trhip_copy_peer(trhip_device* dst_dev, void* dst, trhip_device* src_dev, const void* src, size_t bytes, void* stream);

/* ---- texture files (host only, no device involved).  What stb_image does for the reference's glTF loader through tinygltf
 * (src/gltf.cc:520-576): a PNG (any colour type and bit depth, interlaced or not) or a baseline / extended-sequential / progressive JPEG file
 * in memory becomes RGBA8, row 0 = top row of the file (include/tauray_image.hh; the C++ loader includes that header, the
 * Python mirror calls this entry point, so both flatten a scene to the same bytes).  *rgba_out is released with
 * trhip_image_free.  channels_in_file: 1 grey, 2 grey + alpha, 3 RGB, 4 RGBA. */
int trhip_image_decode(const void* data, size_t bytes, uint32_t* width, uint32_t* height, uint32_t* channels_in_file, uint8_t** rgba_out);
/* The same file as the texels a scene stores: RGBA8 (*bits = 8), or - a PNG of 16 bits per sample - RGBA16 in host byte order
 * (*bits = 16, 8 bytes per texel): the reference keeps such an image as R16G16B16A16Unorm (src/gltf.cc:548-556).  Released with
 * trhip_image_free. */
int trhip_image_decode_texels(const void* data, size_t bytes, uint32_t* width, uint32_t* height, uint32_t* channels_in_file, uint32_t* bits, uint8_t** texels_out);
void trhip_image_free(uint8_t* rgba);

/* OpenEXR files (include/tauray_exr.hh; what tinyexr is to the reference).  trhip_exr_decode = read_exr of src/texture.cc:70-163:
 * a single-part scanline or tiled file (NONE / RLE / ZIPS / ZIP / PIZ; HALF, FLOAT or UINT channels) becomes interleaved floats,
 * *channels of them per pixel (at most four: R, G, B, A by the first letter of the channel names, or file order when a channel is
 * called anything else), row 0 = top of the data window.  trhip_exr_encode = the file headless::save_image writes
 * (src/headless.cc:355-412) from RGBA32F pixels: channels [A,] B, G, R as half or float; compression = the OpenEXR code
 * (0 none, 1 RLE, 2 ZIPS, 3 ZIP, 4 PIZ - the reference's default, src/headless.hh:56).  Results are released with trhip_exr_free. */
int trhip_exr_decode(const void* data, size_t bytes, uint32_t* width, uint32_t* height, uint32_t* channels, float** pixels_out);
int trhip_exr_encode(const float* rgba, uint32_t width, uint32_t height, int alpha, int half, int compression, uint8_t** bytes_out, size_t* size_out);
void trhip_exr_free(void* p);

/* ---- scene (replaces scene_stage::update's uploads, src/scene_stage.cc:1026-1496) */
typedef struct trhip_scene_desc {
    const void* instances;            /* 288-byte `instance` records (shader/scene.glsl:43-53) */
    const void* spans;                /* u32x4 per instance: vertex_offset, vertex_count, index_offset, triangle_count */
    uint32_t instance_count;
    const void* vertices;             /* 48-byte `vertex` records, model space (src/mesh.hh:19-25) */
    uint32_t vertex_count;
    const uint32_t* indices;          /* per-instance, relative to the instance's vertex span */
    uint32_t index_count;
    const void* point_lights;         /* 64 B (shader/light.glsl:15-27); point lights first, then spotlights */
    uint32_t point_light_count;
    const void* directional_lights;   /* 32 B (shader/light.glsl:7-13) */
    uint32_t directional_light_count;
    const void* texture_infos;        /* u32x4 per texture: width, height, texel_offset (in 4-byte words of `texels`), format (0 = RGBA8, 1 = RGBA16) */
    uint32_t texture_count;
    const uint8_t* texels;            /* RGBA8 (4 bytes per texel) or RGBA16 (8 bytes, host byte order) by the texture's format, row 0 first */
    const float* envmap;              /* RGBA32F lat-long, or NULL (then environment_factor is ignored, proj = -1) */
    uint32_t envmap_width, envmap_height;
    const void* alias_table;          /* 16 B entries (src/environment_map.hh:37-43), one per envmap texel */
    float environment_factor[4];
    const void* cameras;              /* 320-byte camera_data (shader/camera.glsl:13-23), one per viewport */
    uint32_t camera_count;
    const uint8_t* non_opaque;        /* per instance: material::potentially_transparent (src/material.cc:7-11) */
    uint32_t gather_emissive_triangles; /* scene_stage::options::gather_emissive_triangles (src/tauray.cc:384) */
} trhip_scene_desc;

typedef struct trhip_accel_info {
    uint32_t triangle_count;
    uint32_t node_count;
    uint32_t tri_light_count;
    float build_ms;                   /* device time of the whole build */
    float bounds_min[3], bounds_max[3];
    uint32_t node_bytes;              /* bytes one node visit reads (112: six box planes + child refs of a 4-wide node) */
    uint32_t leaf_count;              /* leaves of the tree (= triangle_count: one triangle per leaf); node_count = leaf_count - 1 */
} trhip_accel_info;

/* Copies the scene to the device (the reference's buffers byte for byte) and derives, next to them, one record
 * per index triangle with its three vertices side by side (128 bytes of positions, normals and texture coordinates = one cache line, and
 * 48 bytes of tangents apart), which the shading kernels read instead of indices + vertices (same values, fewer cache lines; rebuilt for
 * a mesh by trhip_scene_skin).  Spans that do not start at a whole
 * triangle, or that share indices over different vertex ranges, are valid input: such a scene gets no records and the general kernels. */
int trhip_scene_upload(trhip_device* dev, const trhip_scene_desc* desc);
int trhip_scene_update_cameras(trhip_device* dev, const void* camera_data, uint32_t count); /* src/scene_stage.cc:1145-1174 */
/* Moving lights: replaces the 64-byte point / spot light records and the 32-byte directional light records of the uploaded
 * scene, same counts (what scene_stage::update rewrites when a light's transformable changed, src/scene_stage.cc:1287-1354).
 * Triangle lights follow their instances (trhip_scene_update_instances + the acceleration-structure update). */
int trhip_scene_update_lights(trhip_device* dev, const void* point_lights, uint32_t point_light_count, const void* directional_lights,
                              uint32_t directional_light_count);
/* camera_pair.previous of every viewport (shader/scene.glsl:176-185); the current cameras until set.  Feeds the motion
 * features and the screen-motion target. */
int trhip_scene_set_previous_cameras(trhip_device* dev, const void* camera_data, uint32_t count);
/* Dynamic scenes: replaces the 288-byte instance records (model, model_normal, model_prev, material) of the uploaded
 * scene, same count and meshes (what scene_stage::update rewrites per frame, src/scene_stage.cc:1066-1116).  The
 * acceleration structure is invalidated: call trhip_scene_build_accel again (full rebuild on the device). */
int trhip_scene_update_instances(trhip_device* dev, const void* instances, uint32_t count);
/* Skinned meshes.  trhip_scene_set_skin marks the mesh of `instance` as animated: `source` is the bind-pose vertex array
 * (mesh::get_animation_source(), src/mesh.hh:47,73; NULL = the vertices uploaded for that instance) and `skins` one
 * {uvec4 joints, vec4 weights} record per vertex (mesh::skin_data, src/mesh.hh:32-36).  trhip_scene_skin runs
 * shader/skinning.comp over it with `joint_count` column-major mat4 joint transforms (global transform * inverse bind
 * matrix, model::update_joints src/model.cc:107-118) and rewrites the instance's vertices in the scene, as
 * scene_stage::record_skinning does per frame (src/scene_stage.cc:1543-1567).  Instances that share the vertex span
 * move together, like instances of one mesh in the reference.  The acceleration structure is invalidated: follow with
 * trhip_scene_refit_accel (the BLAS *update* of src/scene_stage.cc:1569-1612) or trhip_scene_build_accel. */
typedef struct trhip_skin { uint32_t joints[4]; float weights[4]; } trhip_skin;
int trhip_scene_set_skin(trhip_device* dev, uint32_t instance, const void* source_vertices, const trhip_skin* skins, uint32_t vertex_count);
int trhip_scene_skin(trhip_device* dev, uint32_t instance, const float* joint_transforms, uint32_t joint_count);
/* Reads back the (possibly skinned) 48-byte vertices of one instance; for tests and tools. */
int trhip_scene_get_vertices(trhip_device* dev, uint32_t instance, void* out_host, uint32_t max_count);
/* After trhip_scene_update_instances: keeps the topology of the last build and recomputes the world triangles, every
 * child box (level by level, bottom-up) and the tri lights - an acceleration-structure *update* instead of a build
 * (src/acceleration_structure.cc:376-422).  Results are identical to a rebuild; traversal gets slower as the
 * transforms drift from the ones the tree was built for. */
int trhip_scene_refit_accel(trhip_device* dev, trhip_accel_info* out);
/* Replaces vkCmdBuildAccelerationStructuresKHR (src/acceleration_structure.cc:198,266,421) with an
 * on-device build (pre-transform -> bounds -> Morton -> radix sort -> PLOC clustering -> 4-wide collapse) and
 * runs extract_tri_lights (shader/extract_tri_lights.comp:17-54).  Synchronous. */
int trhip_scene_build_accel(trhip_device* dev, trhip_accel_info* out);
/* What the reference tells the driver per mesh (src/acceleration_structure.cc:127-133): static geometry is built with
 * ePreferFastTrace, dynamic geometry with ePreferFastBuild | eAllowUpdate.  prefer_fast_build = 0 (the default):
 * trhip_scene_build_accel follows the clustering with tree-optimisation rounds (parallel reinsertion) and chooses the 4-wide
 * nodes by cost (csrc/bvh_optimize.h): 12 % fewer node visits per ray, 16 instead of 9 ms for a million triangles; != 0: it does
 * neither - for callers that rebuild every frame.  Hits, and therefore frames, do not depend on the choice. */
int trhip_scene_set_build_mode(trhip_device* dev, int prefer_fast_build);
/* copies the 64-byte tri_light records back to the host (test hook) */
int trhip_scene_get_tri_lights(trhip_device* dev, void* out_host, uint32_t max_count);

/* ---- path_tracer_stage (src/path_tracer_stage.{hh,cc}, src/rt_camera_stage.{hh,cc}, src/rt_stage.{hh,cc}) */
typedef struct trhip_pt_options {     /* == path_tracer_stage::options flattened (src/path_tracer_stage.hh:13-30) */
    int32_t max_bounces;              /* rt_stage::options::max_ray_depth -> MAX_BOUNCES */
    float min_ray_dist;
    uint32_t rng_seed;                /* raw option; pcg() applied if non-zero (src/rt_stage.cc:82) */
    int32_t sampler;                  /* rt_stage::sampler_type: 0 uniform-random, 1 sobol-owen, 2 sobol-z 2D, 3 sobol-z 3D */
    int32_t samples_per_pixel;
    int32_t samples_per_pass;
    int32_t projection;               /* camera::projection_type: 0 perspective, 1 orthographic, 2 equirectangular */
    int32_t film;                     /* film_filter: 0 point, 1 box, 2 blackman-harris */
    float film_radius;
    int32_t mis_mode;                 /* 0 disabled, 1 balance, 2 power */
    float russian_roulette_delta;
    float indirect_clamping;
    float regularization_gamma;
    int32_t depth_of_field;
    float nee_point, nee_directional, nee_envmap, nee_triangles;  /* light_sampling_weights; 0 disables the class */
    int32_t bounce_mode;              /* bounce_sampling_mode: 0 hemisphere, 1 cosine hemisphere, 2 material */
    int32_t tri_light_mode;           /* tri_light_sampling_mode: 0 area, 1 solid angle, 2 hybrid */
    int32_t hide_lights;
    int32_t use_white_albedo_on_first_bounce;
    int32_t transparent_background;
    int32_t pre_transformed_vertices; /* PRE_TRANSFORMED_VERTICES: shade from the world-space vertex copy of shader/pre_transform.comp (built on first use) */
} trhip_pt_options;

typedef struct trhip_distribution {   /* == distribution_params (src/distribution_strategy.hh:21-28) */
    uint32_t size_x, size_y;
    int32_t strategy;                 /* 0 duplicate, 1 scanline, 2 shuffled strips */
    uint32_t index, count;
    uint32_t primary;
} trhip_distribution;

typedef struct trhip_counters {       /* per trhip_pt, cumulative since the last reset */
    uint64_t closest_rays, shadow_rays;           /* rays actually traced (-> Mray/s) */
    uint64_t node_visits, tri_tests, alpha_tests, surface_hits;   /* only counted when counting is enabled */
    uint64_t stack_overflows;
} trhip_counters;

typedef struct trhip_timings {        /* hipEvent timers with the reference's stage names (src/timer.cc) */
    float path_tracing_ms;            /* "path tracing (N viewports)" of the last trhip_pt_render */
    /* Per-kernel device time, cumulative since the last trhip_pt_reset_counters; only collected while
     * detailed timing is on (event pairs around every launch, no host synchronisation). */
    float trace_closest_ms, trace_shadow_ms, shade_ms, raygen_ms, resolve_ms;
    uint32_t trace_closest_launches, trace_shadow_launches, shade_launches, frames;
} trhip_timings;

int trhip_pt_create(trhip_device* dev, const trhip_pt_options* opt, trhip_pt** out);   /* path_tracer_stage ctor */
/* direct_stage (src/direct_stage.{hh,cc}, shader/direct.rgen): the first hit of every pixel with sphere lights hidden plus
 * samples_per_pass light samples from it; no bounces.  Same handle type and the same calls as the path tracer (set
 * distribution, render, render_targets, counters, timings).  Of the options it reads the sampler, sample counts, film,
 * projection, light-sampling weights, bounce and tri-light modes, min_ray_dist and transparent_background; MIS, clamping,
 * regularisation and roulette do not apply (the reference sets no such defines for it) and max_bounces only sizes the
 * Sobol table. */
int trhip_direct_create(trhip_device* dev, const trhip_pt_options* opt, trhip_pt** out);
void trhip_pt_destroy(trhip_pt* pt);
int trhip_pt_set_distribution(trhip_pt* pt, const trhip_distribution* dist);  /* rt_camera_stage::reset_distribution_params */
int trhip_pt_reset_accumulation(trhip_pt* pt, int reset_sample_counter);      /* reset_accumulated_samples / reset_sample_counter */
/* rt_stage::frame_counter (src/rt_stage.cc:81-86) of the next frame: with one stage per frame slot, slot k of F renders
 * frames k, k + F, ... and sets the counter before each of them (sample_counter = frame_counter * samples_per_pixel). */
int trhip_pt_set_frame_counter(trhip_pt* pt, uint32_t frame_counter);
/* Several consecutive frames in one launch.  After trhip_pt_set_frame_batch(pt, B) a render call takes B * V layers (V = the
 * viewports of one frame) and renders frames f .. f + B - 1 of the stage's frame counter into them, frame-major: layer l is
 * viewport l % V of frame f + l / V, with exactly the samples a separate render call for that frame would have drawn; the frame
 * counter advances by B.  For frames that do not accumulate (offline frames: the caller resets the accumulation between them,
 * src/tauray.cc:1101).  What it is for: a rank of a pixel-sharded multi-GPU job traces an eighth of a frame per call, launches
 * that are too small to fill the chip; four frames per launch cost 12 % less per frame. */
int trhip_pt_set_frame_batch(trhip_pt* pt, uint32_t frames);
/* How many slices of a frame the stage runs concurrently on its own streams: 0 = automatic
 * (by frame size: four lanes from 200 k paths, two from 100 k, each on a hardware pipe of its own - and by trhip_pt_set_frame_slots:
 * two lanes per stage with two slots, one with more), 1 = everything on the caller's stream. */
int trhip_pt_set_lanes(trhip_pt* pt, int lanes);
/* A hint from a renderer with frame slots (MAX_FRAMES_IN_FLIGHT stages of the same scene, src/context.hh:26): how many stages render
 * next to this one on the device.  The stage sizes its persistent launches by it - with two or three frames in flight a trace launch of
 * three blocks per CU leaves the room the neighbours need (two slots: -0 ... 5 %, three: -3 ... 5 %), with four or more the larger grids
 * stay (profiles/r5/frame_slot_grids.txt).  0 (the default) = unknown.  Frames are the same bits whatever the hint. */
int trhip_pt_set_frame_slots(trhip_pt* pt, int slots);
/* The schedule of the stage's last render: how many lanes it ran (1 ... 4) and the hardware pipe class (trhip_stream_pipe_class) of each
 * lane's stream, the caller's stream first.  Lanes on one pipe would have run one after the other. */
int trhip_pt_get_lane_pipes(trhip_pt* pt, int32_t* lanes_out, int32_t pipe_classes_out[4]);
/* View and sample sharding across devices (the reference itself only shards pixels,
 * src/distribution_strategy.cc).  Local layer l of the target shows viewport viewport_base + l * viewport_stride: that
 * viewport's camera (shader/scene.glsl:176-185) and its RNG stream (the viewport index seeds the sampler,
 * shader/sampling.glsl:32-45).  Local sample s of a frame is sample sample_base + s * sample_stride of the pixel's
 * sequence, out of samples_per_pixel * sample_stride per frame (every shard takes the same number).  With these a
 * shard's pixels are the ones a single device renders for that viewport / those samples.  Defaults 0, 1, 0, 1. */
int trhip_pt_set_shard(trhip_pt* pt, uint32_t viewport_base, uint32_t viewport_stride, uint32_t sample_base, uint32_t sample_stride);
/* One frame: update() + every pass of record_command_buffer_pass (src/path_tracer_stage.cc:118-147).
 * `color` is the device RGBA32F image2DArray [viewports][target_h][target_w] where target size is
 * get_distribution_target_size(dist) (src/distribution_strategy.cc:6-19). */
int trhip_pt_render(trhip_pt* pt, void* color_dev, uint32_t target_w, uint32_t target_h, uint32_t viewports, void* stream);
/* The same frame into any subset of the gbuffer targets path_tracer.rgen writes (write_all_outputs,
 * shader/path_tracer.glsl:535-576; gbuffer_target of src/gbuffer.hh): device images [viewports][target_h][target_w],
 * null = not requested.  color / diffuse / reflection are running means over the accumulated samples
 * (shader/gbuffer.glsl:18-28,68-78,118-128); diffuse and reflection are the demodulated light of material.glsl:66-73
 * with a = 1/length of the second path segment.  albedo, material (metallic, roughness, ior/4, transmittance:
 * gbuffer.glsl:256-260), normal (octahedral, math.glsl:480-485), pos and instance_id describe the first hit and are
 * written by the first sample only. */
typedef struct trhip_pt_targets {
    void* color;        /* RGBA32F */
    void* diffuse;      /* RGBA32F */
    void* reflection;   /* RGBA32F */
    void* albedo;       /* RGBA32F */
    void* material;     /* RGBA32F */
    void* normal;       /* RG32F   */
    void* pos;          /* RGBA32F, world space, w = 0 */
    void* instance_id;  /* R32I, -1 = no surface */
    void* screen_motion;/* RG32F: get_camera_projection(previous camera, previous position).xy (shader/camera.glsl:61-67) */
} trhip_pt_targets;
int trhip_pt_render_targets(trhip_pt* pt, const trhip_pt_targets* targets, uint32_t target_w, uint32_t target_h, uint32_t viewports, void* stream);
/* Arithmetic of the shading kernel.  The reference's GLSL runs at the accuracy Vulkan asks of an implementation (SPIR-V
 * precision requirements: `/` 2.5 ULP, inversesqrt 2 ULP, sin / cos 2^-11 absolute, pow through exp2 and log2), and so do the
 * shading kernels here by default (csrc/shade_fast.hip, csrc/shade_spec.hip: v_rcp / v_rsq / v_sqrt /
 * v_sin / v_cos / v_exp / v_log).  ieee != 0: every shading kernel of this stage computes in IEEE fp32 with the C library's
 * sin / cos / pow, expression by expression like the CPU oracle - slower (1.22 instead of 0.97 ms per 1080p frame of the
 * million-triangle bench scene), for comparisons that want the last bit.  Ray traversal and the ray-triangle test are IEEE
 * fp32 in either mode: hits do not depend on it.  The environment variable TRHIP_SHADE_FAST=0 makes ieee the default. */
int trhip_pt_set_shading_arithmetic(trhip_pt* pt, int ieee);
/* Shading program of the stage.  The reference compiles a stage's options into its pipeline as #defines when the stage is built
 * (src/path_tracer_stage.cc:30-116, shaderc at run time through src/shader_source.cc).  Here the ray generation and shading
 * kernels exist ahead of time for the command-line option set and in a general form that reads every option as data; a stage
 * with any other option set gets a program compiled for it the first time it renders (csrc/shade_spec.hip through hipRTC: the
 * sampler, film filter, MIS rule, bounce and triangle-light modes, the light classes in use and the on / off state of roulette,
 * clamping, regularisation, depth of field ... become constants; a few seconds once, then the kernel cache -
 * trhip_kernel_cache_dir - serves it).  Same bits as the general kernels of the same arithmetic, fewer instructions.
 * enable: 1 = specialise (the default unless TRHIP_SPECIALIZE=0), 0 = always the general kernels.  If no program can be built
 * (no libhiprtc) the stage renders with the general kernels and says so once on stderr. */
int trhip_pt_set_specialization(trhip_pt* pt, int enable);
/* Compiles the programs of an option set into the kernel cache ahead of time, so that the first frame does not wait for
 * the compiler: ray generation and shading, for `arch` (NULL = "gfx950"); shade_tris = the scene will have whole-triangle
 * index spans (what trhip_scene_upload derives ShadeTri records from - true for every glTF file), ieee / count_work as in
 * trhip_pt_set_shading_arithmetic / trhip_pt_set_profiling.  Needs no GPU and no device handle. */
int trhip_pt_precompile(const trhip_pt_options* opt, int shade_tris, int ieee, int count_work, const char* arch);
/* Which shading program renders this stage, resolved now (a program for the option set is loaded from the kernel cache or compiled, as the
 * first render would do): the reference compiles one pipeline per stage from its options (src/path_tracer_stage.cc:30-116) and every
 * device of a job gets the same one.  Here a stage can end up on three kinds of kernels, and at the default arithmetic two kinds are two
 * implementations inside Vulkan's accuracy, not the same bits - so the ranks of a multi-GPU job compare
 * `identity` before the first frame (tr::process_rt_renderer, tauray_amd.renderer.RtRenderer: all-gather, mismatch = error) and
 * bench.py records `kind` instead of inferring it from the options.
 *   kind: 0 = the general kernels (every option read from the parameter block), 1 = the ahead-of-time instances of the reference's
 *         command-line option set, 2 = a program compiled for this option set (hipRTC / kernel cache);
 *   ieee: 1 = IEEE fp32 shading, 0 = Vulkan-grade arithmetic;
 *   identity: FNV-1a over kind, arithmetic, the pinned option fields, the embedded device sources of this build of the library, the
 *         build's id (trhip_build_id: every source of csrc/ and the compile flags, so two builds whose ahead-of-time kernels differ - kinds
 *         0 and 1 live in path_tracer.o / shade_fast.o, not in the embedded sources - differ here too) and, for kind 2, the bytes of the
 *         code objects that were loaded;
 *   key: the pinned fields as text (what TRHIP_DEBUG prints). */
typedef struct trhip_program_info { int32_t kind, ieee; uint64_t identity; char key[240]; } trhip_program_info;
int trhip_pt_get_program(trhip_pt* pt, trhip_program_info* out);
uint64_t trhip_build_id(void);              /* 64 bits of SHA-256 over the library's sources and compile flags, fixed when it was built */
const char* trhip_kernel_cache_dir(void);   /* TRHIP_KERNEL_CACHE, else kernel_cache/ next to libtrhip.so, else ~/.cache/trhip; "" = none writable */
int trhip_pt_set_profiling(trhip_pt* pt, int count_work, int detailed_timing);
int trhip_pt_get_counters(trhip_pt* pt, trhip_counters* out);     /* synchronises the stream */
int trhip_pt_reset_counters(trhip_pt* pt);
int trhip_pt_get_timings(trhip_pt* pt, trhip_timings* out);       /* synchronises the stream */
/* Wave-level statistics of the closest-hit loop, cumulative like trhip_counters and only collected while work counting is on
 * (trhip_pt_set_profiling): how many node / triangle phases the waves executed one ray per lane and one ray per quad
 * (csrc/trace_quad.h), and the per-lane node phases by the number of live rays (1-8, 9-16, ..., 57-64).  node_visits divided by
 * the node phases is the number of rays one vector instruction of the traversal serves - what the VALU roofline of bench.py
 * multiplies the issue rate with (a hardware lane count cannot tell a quad's four lanes from four rays). */
typedef struct trhip_phase_counters {
    uint64_t lane_node_phases, lane_tri_phases, quad_node_phases, quad_tri_phases;
    uint64_t lane_node_phases_le16, lane_node_visits_le16;
    uint64_t lane_node_phase_hist[8];
    uint64_t closest_node_visits;     /* node visits of the closest-hit rays alone (trhip_counters::node_visits includes shadow rays) */
} trhip_phase_counters;
int trhip_pt_get_phase_counters(trhip_pt* pt, trhip_phase_counters* out);   /* synchronises the stream */
/* Peak vector-instruction issue rate of the device as it runs now: a loop of independent v_fma_f32 at eight waves per SIMD,
 * in 10^9 wave-level instructions per second (MI355X_MICROARCH.md: 2 cycles per wave64 instruction on a SIMD-32; the clock is
 * what the box sustains).  The peak of the VALU roofline in bench.py; about 2 ms of device time. */
int trhip_calibrate_valu(trhip_device* dev, float* ginst_per_s);
/* Peak rate at which the vector L1 caches (TCP, one per CU) take cache-line accesses, in 10^9 accesses per second over the device:
 * independent 16-byte loads out of an L1-resident footprint, every lane of a wave in a 128-byte line of its own (64 accesses per
 * wave instruction).  One access per clock and CU (tools/ubench/l1_tags.hip, profiles/r3/l1_tag_rate.json) - what the counter
 * TCP_TOTAL_CACHE_ACCESSES counts, and the peak of the L1 level of bench.py's roofline: a traversal step reads its node with seven
 * loads per lane, seven accesses to one line.  About 1 ms of device time. */
int trhip_calibrate_l1(trhip_device* dev, float* gaccesses_per_s);

/* ---- feature_stage (src/feature_stage.cc:22-104): 0 albedo, 1 world normal, 2 view normal, 3 world pos,
 *      4 view pos, 5 distance, 6 world motion, 7 view motion, 8 screen motion, 9 instance id */
int trhip_feature_render(trhip_device* dev, int feature, const trhip_distribution* dist, int projection,
                         uint32_t viewport, float min_ray_dist, const float default_value[4],
                         void* color_dev, uint32_t target_w, uint32_t target_h, void* stream);

/* ---- ray-level queries (parity hooks for traceRayEXT, shader/path_tracer.glsl:38-50,387-403).
 * rays: 8 floats each {ox, oy, oz, tmin, dx, dy, dz, tmax}; hits: {i32 instance, i32 primitive, f32 u, f32 v, f32 t}.
 * seeds == NULL selects the feature renderer's fixed alpha cutoff (shader/rt_feature.rahit:17). */
/* Two queries on one device must not run at the same time (the closest-hit query keeps one spill buffer for the deep stack
 * entries of its quad tails per device): enqueue them on one stream, or order the streams with trhip_stream_wait. */
int trhip_trace_closest(trhip_device* dev, uint32_t n, const void* rays_dev, const void* seeds_dev,
                        int include_lights, void* hits_dev, void* stream);
int trhip_trace_shadow(trhip_device* dev, uint32_t n, const void* rays_dev, void* visibility_dev, void* stream);

/* ---- stitch_stage (src/stitch_stage.cc:128-196, shader/stitch_scanline.comp, stitch_shuffled_strips.comp).
 * Scatters one non-primary device's partial image into the primary (full-size) image. */
int trhip_stitch(trhip_device* dev, const trhip_distribution* partial_dist, const void* partial_dev,
                 uint32_t partial_w, uint32_t partial_h, void* primary_dev, uint32_t viewports,
                 float blend_ratio, void* stream);
/* The partial images of all non-primary devices in one launch (the reference dispatches the stitch shader once per
 * device, src/stitch_stage.cc:150-196): `count` entries of what trhip_stitch takes.  With blend_ratio < 1 the partials
 * must not overlap, which the distribution strategies guarantee. */
int trhip_stitch_batch(trhip_device* dev, uint32_t count, const trhip_distribution* partial_dists, const void* const* partials_dev,
                       const uint32_t* partial_ws, const uint32_t* partial_hs, void* primary_dev, uint32_t viewports,
                       float blend_ratio, void* stream);

/* ---- tonemap_stage (src/tonemap_stage.cc:139-164, shader/tonemap*.comp) */
typedef struct trhip_tonemap_info {
    int32_t op;                       /* tonemap_stage::operator_type: 0 linear, 1 gamma, 2 filmic, 3 reinhard, 4 reinhard luminance */
    float exposure, gamma;
    int32_t alpha_grid_background;    /* 0 or grid size (16 when not headless) */
} trhip_tonemap_info;
int trhip_tonemap(trhip_device* dev, const void* in_dev, void* out_dev, uint32_t width, uint32_t height,
                  uint32_t layers, const trhip_tonemap_info* info, void* stream);
/* A renderer with nothing between its path tracer and its tonemap stage (one device, no stitch, no denoiser: rt_renderer with one
 * device, src/rt_renderer.cc) can have the stage's last pass write the display image as it writes the colour target: `display_dev`
 * (same width x height x layers as the colour target, RGBA32F) receives trhip_tonemap(colour) pixel by pixel, the same bits, without the
 * second pass over the frame (a 1080p frame: 47 us after the last lane has finished).  NULL turns it off.  Path tracer stages only. */
int trhip_pt_set_fused_tonemap(trhip_pt* pt, void* display_dev, const trhip_tonemap_info* info);

#ifdef __cplusplus
}
#endif
#endif
