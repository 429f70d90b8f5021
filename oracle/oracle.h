/* TEST INFRASTRUCTURE - CPU oracle for the path_tracer_stage hot path.
 *
 * A scalar restatement of the reference's GLSL (shader/path_tracer.*, rt.glsl,
 * ggx.glsl, light.glsl, math.glsl, sampling.glsl, ...) with its own SAH BVH
 * standing in for the Vulkan driver's traceRayEXT.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (libtrhip.so) never does.
 *
 * Parity status: pinned against the reference's golden EXRs for test.glb
 * (distance, world-pos, view-pos, world-normal, view-normal, albedo: tight;
 * path-tracer: statistical).  See tests/test_oracle_golden.py.
 */
#ifndef TR_ORACLE_H
#define TR_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_scene_desc {
    const void* instances;            /* 288-byte `instance` records (shader/scene.glsl:43-53) */
    const void* spans;                /* {vertex_offset, vertex_count, index_offset, triangle_count} u32x4 per instance */
    uint32_t instance_count;
    const void* vertices;             /* 48-byte `vertex` records, model space */
    uint32_t vertex_count;
    const uint32_t* indices;
    uint32_t index_count;
    const void* point_lights;         /* 64 B each */
    uint32_t point_light_count;
    const void* directional_lights;   /* 32 B each */
    uint32_t directional_light_count;
    const void* texture_infos;        /* {width, height, texel_offset, pad} u32x4 */
    uint32_t texture_count;
    const uint8_t* texels;            /* RGBA8 */
    const float* envmap;              /* RGBA32F lat-long or NULL */
    uint32_t envmap_width, envmap_height;
    const void* alias_table;          /* 16 B entries or NULL */
    float environment_factor[4];
    const void* cameras;              /* 320-byte camera_data, one per viewport */
    uint32_t camera_count;
    const uint8_t* non_opaque;        /* per instance: 1 = any-hit (potentially transparent) */
    uint32_t gather_emissive_triangles;
} oracle_scene_desc;

typedef struct oracle_pt_options {
    int32_t max_bounces;
    float min_ray_dist;
    uint32_t rng_seed;
    int32_t sampler;                  /* 0 uniform-random, 1 sobol-owen, 2 sobol-z 2D, 3 sobol-z 3D */
    int32_t samples_per_pixel;
    int32_t samples_per_pass;
    int32_t projection;               /* 0 perspective, 1 orthographic, 2 equirectangular */
    int32_t film;                     /* 0 point, 1 box, 2 blackman-harris */
    float film_radius;
    int32_t mis_mode;                 /* 0 disabled, 1 balance, 2 power */
    float russian_roulette_delta;
    float indirect_clamping;
    float regularization_gamma;
    int32_t depth_of_field;
    float nee_point, nee_directional, nee_envmap, nee_triangles;
    int32_t bounce_mode;              /* 0 hemisphere, 1 cosine hemisphere, 2 material */
    int32_t tri_light_mode;           /* 0 area, 1 solid angle, 2 hybrid */
    int32_t hide_lights;
    int32_t use_white_albedo_on_first_bounce;
    int32_t transparent_background;
    int32_t pre_transformed_vertices;
} oracle_pt_options;

typedef struct oracle_distribution {
    uint32_t size_x, size_y;
    int32_t strategy;                 /* 0 duplicate, 1 scanline, 2 shuffled strips */
    uint32_t index, count;
    uint32_t primary;
} oracle_distribution;

typedef struct oracle_hit {
    int32_t instance_id;
    int32_t primitive_id;
    float bary_u, bary_v;
    float t;
} oracle_hit;

typedef struct oracle_counters {
    uint64_t closest_rays, shadow_rays, node_visits, tri_tests, alpha_tests, surface_hits;
} oracle_counters;

typedef struct oracle_scene oracle_scene;

oracle_scene* oracle_scene_create(const oracle_scene_desc* desc);
void oracle_scene_destroy(oracle_scene* s);
uint32_t oracle_scene_tri_light_count(const oracle_scene* s);
/* copies 64-byte tri_light records (extract_tri_lights.comp) */
void oracle_scene_get_tri_lights(const oracle_scene* s, void* out);

/* One frame of path_tracer_stage: all passes (samples_per_pixel / samples_per_pass).
 * `color` is the RGBA32F target of get_distribution_target_size(dist) x viewports. */
int oracle_pt_render(oracle_scene* s, const oracle_pt_options* opt, const oracle_distribution* dist,
                     uint32_t viewport_count, uint32_t frame_counter, uint32_t samples_accumulated,
                     float* color, uint32_t target_w, uint32_t target_h, int threads);

/* The same frame with the other gbuffer targets path_tracer.rgen can write (write_all_outputs,
 * shader/path_tracer.glsl:535-576; formats of shader/gbuffer.glsl): any pointer may be null.
 * color/diffuse/reflection/albedo/material/pos: 4 floats per pixel; normal: 2 floats (octahedral);
 * instance_id: 1 int32.  Same target size as `color`. */
typedef struct oracle_pt_targets {
    float* color; float* diffuse; float* reflection; float* albedo; float* material; float* normal; float* pos;
    int32_t* instance_id;
    float* screen_motion;   /* 2 floats: get_camera_projection(previous camera, previous position).xy */
} oracle_pt_targets;
int oracle_pt_render_targets(oracle_scene* s, const oracle_pt_options* opt, const oracle_distribution* dist,
                             uint32_t viewport_count, uint32_t frame_counter, uint32_t samples_accumulated,
                             const oracle_pt_targets* targets, uint32_t target_w, uint32_t target_h, int threads);
/* direct_stage (src/direct_stage.cc:30-127, shader/direct.rgen): first hit + samples_per_pass light samples; same options struct
 * (max_bounces only sizes the Sobol table; MIS, clamping, regularisation, roulette do not apply) */
int oracle_direct_render_targets(oracle_scene* s, const oracle_pt_options* opt, const oracle_distribution* dist,
                                 uint32_t viewport_count, uint32_t frame_counter, uint32_t samples_accumulated,
                                 const oracle_pt_targets* targets, uint32_t target_w, uint32_t target_h, int threads);

/* feature_stage (src/feature_stage.cc:33-65): 0 albedo, 1 world normal, 2 view normal,
 * 3 world pos, 4 view pos, 5 distance, 6 world motion, 7 view motion, 8 screen motion, 9 instance id */
/* camera_pair.previous per viewport (defaults to the current cameras) */
/* shader/skinning.comp over one mesh: 48-byte vertices, {uvec4 joints, vec4 weights} skins, column-major mat4 joints */
void oracle_skin_vertices(const void* source, const void* skins, uint32_t vertex_count, const float* joint_transforms, uint32_t joint_count, void* destination);
/* view / sample shard of the following oracle_pt_render* calls; same meaning as trhip_pt_set_shard */
int oracle_scene_set_shard(oracle_scene* s, uint32_t viewport_base, uint32_t viewport_stride, uint32_t sample_base, uint32_t sample_stride);
int oracle_scene_set_previous_cameras(oracle_scene* s, const void* camera_data_array, uint32_t count);
int oracle_feature_render(oracle_scene* s, int feature, const oracle_distribution* dist, int projection,
                          uint32_t viewport, float min_ray_dist, const float default_value[4],
                          float* color, uint32_t target_w, uint32_t target_h, int threads);

/* Closest-hit / shadow queries on explicit rays (ray = ox oy oz tmin dx dy dz tmax). */
void oracle_trace_closest(oracle_scene* s, uint32_t n, const float* rays, const uint32_t* seeds,
                          int include_lights, oracle_hit* out, int threads);
void oracle_trace_shadow(oracle_scene* s, uint32_t n, const float* rays, float* visibility, int threads);

void oracle_tonemap(const float* in, float* out, uint32_t pixel_count, int op, float exposure, float gamma);

void oracle_get_counters(oracle_scene* s, oracle_counters* out);
void oracle_reset_counters(oracle_scene* s);

/* Known-answer hooks for the pure functions (tests/test_known_answers.py). */
uint32_t oracle_pcg(uint32_t* seed);
void oracle_pcg2d(uint32_t seed[2], uint32_t out[2]);
void oracle_pcg4d(uint32_t seed[4], uint32_t out[4]);
void oracle_init_random_sampler(const uint32_t coord[4], uint32_t out_seed[4]);
void oracle_generate_sobol_sample(uint32_t index, uint32_t bounce, uint32_t max_sobol_bounces, uint32_t out[4]);
void oracle_owen_scramble_2d(const uint32_t x[4], const uint32_t seed[4], uint32_t out[4]);
uint32_t oracle_owen_scramble_4d(uint32_t x, uint32_t seed);
uint32_t oracle_owen_scramble_8d(uint32_t x, uint32_t seed);
uint32_t oracle_get_permutation_n(int n, uint32_t permutation, uint32_t dimension);
uint32_t oracle_morton_2d(uint32_t x, uint32_t y);
uint32_t oracle_morton_3d(uint32_t x, uint32_t y, uint32_t z);
void oracle_ray_sample_uint(int sampler, int max_bounces, const uint32_t coord[4], uint32_t sample_counter,
                            uint32_t rng_seed_raw, uint32_t bounce_index, uint32_t out[4]);
uint32_t oracle_rgb_to_r9g9b9e5(const float rgb[3]);
void oracle_r9g9b9e5_to_rgb(uint32_t v, float rgb[3]);
uint32_t oracle_pack_half2x16(float x, float y);
uint32_t oracle_permute_region_id(uint32_t i, uint32_t size_x, uint32_t size_y, uint32_t b);
void oracle_camera_ray(const void* camera_data, int projection, float px, float py, float sw, float sh,
                       float dof_u, float dof_v, int dof, float origin[3], float dir[3]);
void oracle_sample_cone(float u0, float u1, const float dir[3], float cos_theta_min, float out[3]);
void oracle_sample_spherical_triangle(float u0, float u1, const float A[3], const float B[3], const float C[3],
                                      float out_dir[3], float* pdf);
void oracle_ggx_vndf_sample(const float view[3], float roughness, float u1, float u2, float out[3]);
/* material = {albedo rgba, metallic, roughness, transmittance, ior_in, ior_out, f0} (10 floats) */
void oracle_ggx_bsdf_sample(const float u[4], const float view[3], const float material[10],
                            float out_dir[3], float lobes[4], float* pdf);
float oracle_ggx_bsdf_pdf(const float out_dir[3], const float view[3], const float material[10], float lobes[4]);
float oracle_sample_blackman_harris(float u);

#ifdef __cplusplus
}
#endif
#endif
