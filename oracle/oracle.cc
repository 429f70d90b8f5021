// TEST INFRASTRUCTURE - CPU oracle for Tauray's path_tracer_stage hot path.
//
// Scalar C++ restatement of the reference's GLSL, function by function, each
// citing the reference file:line it follows (paths relative to the reference
// checkout).  The Vulkan driver's BVH build / traceRayEXT - which is not in the
// reference's source tree - is replaced by a binned-SAH BVH with Woop-style
// watertight ray/triangle tests (Woop, Benthin, Wald 2013).
//
// Documented deviations from the reference (both implementation-defined there):
//  * any-hit order: the reference advances payload.random_seed once per
//    any-hit invocation, in driver traversal order (shader/rt_common.rahit:21).
//    Here the alpha-test random number is a traversal-order-independent hash
//    of (seed, instance, primitive) and the seed advances once per closest-hit
//    trace.  Statistically equivalent; makes results independent of the BVH.
//  * equal-t hits are resolved towards the lower (instance, primitive).
//
// Compile with -ffp-contract=off (see Makefile): arithmetic is plain IEEE fp32
// so the HIP kernels can reproduce geometry bit-for-bit.
#include "oracle.h"
#include "glsl.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace gl;

namespace {

#define M_PI_F 3.14159265359f          // shader/math.glsl:4
#define M_1_SQRT3 0.57735026918962576451f
#define INV_UINT32_MAX 2.3283064365386963e-10f
static const float RAY_MAX_DIST = std::numeric_limits<float>::infinity();  // float(1e39), shader/rt.glsl:25

// ----------------------------------------------------------------------------
// Data layouts (Appendix A of SURVEY.md; all tightly packed)
// ----------------------------------------------------------------------------
#pragma pack(push, 4)
struct vertex { vec3 pos; vec3 normal; vec2 uv; vec4 tangent; };                 // shader/scene.glsl:14-20
struct material {                                                                  // shader/material.glsl:9-22
    vec4 albedo_factor, metallic_roughness_factor, emission_factor;
    float transmittance, ior, normal_factor; uint flags;
    int albedo_tex_id, metallic_roughness_tex_id, normal_tex_id, emission_tex_id;
};
struct instance {                                                                  // shader/scene.glsl:43-53
    int light_base_id, sh_grid_index; uint pad; float shadow_terminator_mul;
    mat4 model, model_normal, model_prev; material mat;
};
struct directional_light { vec3 color; int shadow_map_index; vec3 dir; float dir_cutoff; };  // shader/light.glsl:7-13
struct point_light {                                                               // shader/light.glsl:15-27
    vec3 color, dir, pos; float radius, dir_cutoff, dir_falloff, cutoff_radius, spot_radius;
    int shadow_map_index, padding;
};
struct tri_light {                                                                 // shader/light.glsl:29-37
    vec3 pos[3]; uint emission_factor, instance_id, primitive_id; uint uv[3]; int emission_tex_id;
};
struct alias_table_entry { uint alias_id, probability; float pdf, alias_pdf; };  // shader/alias_table.glsl:7-13
struct camera_data {                                                               // shader/camera.glsl:13-23
    mat4 view, view_inverse, view_proj, proj_inverse; vec4 origin, dof_params, projection_info, pan;
};
struct mesh_span { uint vertex_offset, vertex_count, index_offset, triangle_count; };
struct texture_info { uint width, height, texel_offset, format; };      // format 1: RGBA16 texels (R16G16B16A16Unorm, src/gltf.cc:548-556), two 4-byte words each
#pragma pack(pop)
static_assert(sizeof(vertex) == 48 && sizeof(material) == 80 && sizeof(instance) == 288, "layout");
static_assert(sizeof(directional_light) == 32 && sizeof(point_light) == 64 && sizeof(tri_light) == 64, "layout");
static_assert(sizeof(camera_data) == 320 && sizeof(alias_table_entry) == 16, "layout");

struct sampled_material {                                                          // shader/material.glsl:24-36
    vec4 albedo; float metallic, roughness; vec3 emission;
    float transmittance, ior_in, ior_out, f0; uint flags; float shadow_terminator_mul;
};
struct bsdf_lobes { float transmission, diffuse, dielectric_reflection, metallic_reflection; };  // material.glsl:44-50
#define MATERIAL_LOBE_REFLECTION 3
#define MATERIAL_LOBE_TRANSMISSION 2
#define MATERIAL_LOBE_DIFFUSE 1
#define MATERIAL_LOBE_ALL 0

struct vertex_data {                                                               // shader/scene.glsl:22-41
    vec3 pos, prev_pos, hard_normal, smooth_normal, mapped_normal; vec2 uv; vec3 tangent, bitangent;
    bool back_facing; int instance_id, primitive_id;
};
struct pt_vertex_data { vec3 pos, prev_pos, hard_normal, smooth_normal, mapped_normal; int instance_id; };  // path_tracer.glsl:11-22
struct intersection_pdf { float point_light_pdf, directional_light_pdf, tri_light_pdf, envmap_pdf; };
struct hit_payload { uint random_seed; int instance_id, primitive_id; vec2 barycentrics; };  // rt_common_payload.glsl:4-21

// ----------------------------------------------------------------------------
// RNG  (shader/math.glsl:75-122)
// ----------------------------------------------------------------------------
inline uint pcg(uint& seed) {
    seed = seed * 747796405u + 2891336453u;
    seed = ((seed >> ((seed >> 28) + 4)) ^ seed) * 277803737u;
    seed = (seed >> 22) ^ seed;
    return seed;
}
inline uvec2 pcg2d(uvec2& seed) {
    seed = seed * 1664525u + 1013904223u;
    seed = seed + uvec2{seed.y, seed.x} * 1664525u;
    seed = (seed >> 16) ^ seed;
    seed = seed + uvec2{seed.y, seed.x} * 1664525u;
    seed = (seed >> 16) ^ seed;
    return seed;
}
inline uvec4 pcg4d(uvec4& seed) {
    seed = seed * 1664525u + 1013904223u;
    seed = seed + uvec4{seed.y, seed.z, seed.x, seed.y} * uvec4{seed.w, seed.x, seed.y, seed.z};
    seed = (seed >> 16) ^ seed;
    seed = seed + uvec4{seed.y, seed.z, seed.x, seed.y} * uvec4{seed.w, seed.x, seed.y, seed.z};
    return seed;
}

// ----------------------------------------------------------------------------
// Sobol / Owen  (shader/math.glsl:125-201, 260-278)
// ----------------------------------------------------------------------------
static const uint sobol_lookup_table[256][4] = {
#include "sobol_table.inc"
};

uvec4 generate_sobol_sample(uint index, uint bounce, uint max_sobol_bounces) {
    uvec4 x = {0, 0, 0, 0};
    if (bounce >= max_sobol_bounces) {
        x = uvec4{index, bounce, bounce * index, 0};
        return pcg4d(x);
    }
    // top set bit excluded; index 0 => findLSB = -1 and the loop body never runs
    for (int bit = findLSB(index); bit < findMSB(index); bit++) {
        uint mask = (index >> bit) & 1u;
        if (mask != 0) {
            const uint* r = sobol_lookup_table[bounce * 32 + bit];
            x = x ^ uvec4{r[0], r[1], r[2], r[3]};
        }
    }
    return x;
}

uint get_permutation_n(int n, uint permutation, uint dimension) {
    uint res = 0;
    for (int i = n - 1; i >= 0; --i) {
        uint q = permutation % (uint)(n - i);
        permutation /= (uint)(n - i);
        if ((uint)i == dimension) res = q;
        if (dimension > (uint)i) res += uint(res >= q);
    }
    return res;
}

uvec4 owen_scramble_2d(uvec4 x, uvec4 seed) {
    x = bitfieldReverse(x);
    x = x ^ (x * 0x3D20ADEAu);
    x = x + seed;
    x = x * ((seed >> 16u) | 1u);
    x = x ^ (x * 0x05526C56u);
    x = x ^ (x * 0x53A22864u);
    x = bitfieldReverse(x);
    return x;
}
uint owen_scramble_4d(uint x, uint seed) {
    uint result = 0;
    for (uint i = 0; i < 32; i += 2) {
        uvec2 s = {seed, x & ((~3u) << i)};
        result |= get_permutation_n(4, pcg2d(s).x % 24u, (x >> i) & 3u) << i;
    }
    return result;
}
uint owen_scramble_8d(uint x, uint seed) {
    uint result = 0;
    for (uint i = 0; i < 32; i += 3) {
        uvec2 s = {seed, x & ((~7u) << i)};
        result |= get_permutation_n(8, pcg2d(s).x % 40320u, (x >> i) & 7u) << i;
    }
    return result;
}
uint morton_2d(uint xx, uint yy) {
    uint v[2] = {xx, yy};
    for (int k = 0; k < 2; ++k) {
        uint x = v[k] & 0x0000ffffu;
        x = (x ^ (x << 8u)) & 0x00ff00ffu;
        x = (x ^ (x << 4u)) & 0x0f0f0f0fu;
        x = (x ^ (x << 2u)) & 0x33333333u;
        x = (x ^ (x << 1u)) & 0x55555555u;
        v[k] = x;
    }
    return v[0] + 2u * v[1];
}
uint morton_3d(uint xx, uint yy, uint zz) {
    uint v[3] = {xx, yy, zz};
    for (int k = 0; k < 3; ++k) {
        uint x = v[k] & 0x000003ffu;
        x = (x ^ (x << 16u)) & 0xff0000ffu;
        x = (x ^ (x << 8u)) & 0x0300f00fu;
        x = (x ^ (x << 4u)) & 0x030c30c3u;
        x = (x ^ (x << 2u)) & 0x09249249u;
        v[k] = x;
    }
    return v[0] + 2u * v[1] + 4u * v[2];
}

// ----------------------------------------------------------------------------
// Samplers  (shader/sampling.glsl, random_sampler.glsl, sobol_*_sampler.glsl)
// ----------------------------------------------------------------------------
enum { SAMPLER_UNIFORM = 0, SAMPLER_SOBOL_OWEN = 1, SAMPLER_SOBOL_Z2 = 2, SAMPLER_SOBOL_Z3 = 3 };

struct local_sampler {
    uvec4 rs_seed;       // random_sampler.seed
    uvec4 owen_seed;     // sobol_owen_sampler.seed
    uint sobol_index;    // sobol_z_sampler.sobol_index
};

uvec4 init_random_sampler(uvec4 coord) {   // random_sampler.glsl:11-19
    uvec4 seed = coord;
    seed.y ^= pcg(seed.x);
    seed.z ^= pcg(seed.y);
    seed.w ^= pcg(seed.z);
    return seed;
}

local_sampler init_local_sampler(uvec4 coord, uint sample_counter, uint rng_seed, int sampler) {  // sampling.glsl:32-45
    local_sampler ls;
    coord.w += sample_counter;
    coord.z += rng_seed;
    ls.sobol_index = 0;
    ls.owen_seed = uvec4{0, 0, 0, 0};
    if (sampler == SAMPLER_SOBOL_Z3)        // sobol_z_sampler.glsl:24-39
        ls.sobol_index = owen_scramble_8d(morton_3d(coord.x, coord.y, coord.w), coord.w >> 10u);
    else if (sampler == SAMPLER_SOBOL_Z2)
        ls.sobol_index = owen_scramble_4d(morton_2d(coord.x, coord.y), coord.w);
    else if (sampler == SAMPLER_SOBOL_OWEN) {  // sobol_owen_sampler.glsl:23-28
        uvec4 c = coord;
        ls.owen_seed = pcg4d(c);
    }
    ls.rs_seed = init_random_sampler(coord);
    return ls;
}

uvec4 generate_uniform_random_uint(local_sampler& ls) { return pcg4d(ls.rs_seed); }   // random_sampler.glsl:25-28
vec4 generate_uniform_random(local_sampler& ls) { return to_float(generate_uniform_random_uint(ls)) * INV_UINT32_MAX; }

uvec4 generate_ray_sample_uint(local_sampler& ls, uint bounce_index, int sampler, uint max_sobol_bounces) {  // sampling.glsl:66-75
    if (sampler == SAMPLER_SOBOL_Z2 || sampler == SAMPLER_SOBOL_Z3)
        return generate_sobol_sample(ls.sobol_index, bounce_index, max_sobol_bounces);
    if (sampler == SAMPLER_SOBOL_OWEN) {     // sobol_owen_sampler.glsl:11-21 (sampler passed by value)
        uvec4 seed = ls.owen_seed;
        uint index = seed.w;
        seed.w = bounce_index;
        uvec4 hashed = pcg4d(seed);          // mutates the local copy
        uint shuffled_index = owen_scramble_2d(uvec4{index, index, index, index}, hashed).x;
        seed.w = index;
        return owen_scramble_2d(generate_sobol_sample(shuffled_index, bounce_index, max_sobol_bounces),
                                uvec4{seed.y, seed.z, seed.w, seed.x});
    }
    return generate_uniform_random_uint(ls);
}

// ----------------------------------------------------------------------------
// math.glsl sampling helpers
// ----------------------------------------------------------------------------
vec3 create_tangent(vec3 normal) {                     // math.glsl:12-20
    vec3 major;
    if (fabsf(normal.x) < M_1_SQRT3) major = V3(1, 0, 0);
    else if (fabsf(normal.y) < M_1_SQRT3) major = V3(0, 1, 0);
    else major = V3(0, 0, 1);
    return normalize(cross(normal, major));
}
mat3 create_tangent_space(vec3 normal) {               // math.glsl:26-31
    vec3 tangent = create_tangent(normal);
    vec3 bitangent = cross(normal, tangent);
    return M3(tangent, bitangent, normal);
}
vec3 view_to_tangent_space(vec3 view, const mat3& tbn) {   // math.glsl:472-478
    vec3 tview = (-view) * tbn;
    if (tview.z < 1e-5f) tview = V3(tview.x, tview.y, max(tview.z, 1e-5f));
    return normalize(tview);
}
vec2 sample_concentric_disk(vec2 u) {                  // math.glsl:205-218
    vec2 uo = 2.0f * u - 1.0f;
    vec2 abs_uo = {fabsf(uo.x), fabsf(uo.y)};
    if (abs_uo.x < 0.0001f && abs_uo.y < 0.0001f) return V2(0);
    vec2 rt = (abs_uo.x > abs_uo.y) ? V2(uo.x, M_PI_F / 4 * (uo.y / uo.x))
                                    : V2(uo.y, M_PI_F / 2 - M_PI_F / 4 * (uo.x / uo.y));
    return rt.x * V2(cosf(rt.y), sinf(rt.y));
}
float sample_blackman_harris(float u) {                // math.glsl:220-228
    bool flip = u > 0.5f;
    u = flip ? 1 - u : u;
    float vx = -0.33518669f * powf(u, 0.5f), vy = -0.51620529f * powf(u, 0.3333333333f);
    float vz = 1.87406934f * powf(u, 0.25f), vw = -0.66315464f * powf(u, 0.2f);
    float s = 0.29627329f * u + vx + vy + vz + vw;
    return flip ? 1 - s : s;
}
vec2 sample_blackman_harris_concentric_disk(vec2 u) {  // math.glsl:230-241
    vec2 uo = 2.0f * u - 1.0f;
    vec2 abs_uo = {fabsf(uo.x), fabsf(uo.y)};
    if (abs_uo.x < 0.0001f && abs_uo.y < 0.0001f) return V2(0);
    vec2 rt = (abs_uo.x > abs_uo.y) ? V2(u.x, M_PI_F / 4 * (uo.y / uo.x))
                                    : V2(u.y, M_PI_F / 2 - M_PI_F / 4 * (uo.x / uo.y));
    return (2.0f * sample_blackman_harris(rt.x) - 1.0f) * V2(cosf(rt.y), sinf(rt.y));
}
vec2 sample_regular_polygon(vec2 u, float angle, uint sides) {   // math.glsl:281-292
    float side = floorf(u.x * sides);
    u.x = fract(u.x * sides);
    float side_radians = (2.0f * M_PI_F) / sides;
    float a1 = side_radians * side + angle;
    float a2 = side_radians * (side + 1) + angle;
    vec2 b = V2(sinf(a1), cosf(a1));
    vec2 c = V2(sinf(a2), cosf(a2));
    u = u.x + u.y > 1 ? 1 - u : u;
    return b * u.x + c * u.y;
}
vec3 sample_cosine_hemisphere(vec2 u) {                // math.glsl:294-298
    vec2 d = sample_concentric_disk(u);
    return V3(d.x, d.y, sqrtf(max(0.0f, 1 - dot(d, d))));
}
float pdf_cosine_hemisphere(vec3 dir) { return max(dir.z, 0.0f) * (1.0f / M_PI_F); }   // math.glsl:300-303
vec3 sample_sphere(vec2 u) {                           // math.glsl:305-315
    float cos_theta = 2 * u.x - 1;
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float phi = u.y * 2 * M_PI_F;
    return V3(cosf(phi) * sin_theta, sinf(phi) * sin_theta, cos_theta);
}
vec3 sample_hemisphere(vec2 u) {                       // math.glsl:317-327
    float cos_theta = u.x;
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float phi = u.y * 2 * M_PI_F;
    return V3(cosf(phi) * sin_theta, sinf(phi) * sin_theta, cos_theta);
}
vec3 sample_cone(vec2 u, vec3 dir, float cos_theta_min) {   // math.glsl:342-358
    float cos_theta = mix(1.0f, cos_theta_min, u.x);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float phi = u.y * 2 * M_PI_F;
    vec3 o = create_tangent_space(dir) * V3(cosf(phi) * sin_theta, sinf(phi) * sin_theta, cos_theta);
    return dot(o, dir) <= cos_theta_min ? dir : o;
}
vec3 sample_triangle_area(vec2 u, vec3 A, vec3 B, vec3 C) {  // math.glsl:360-371
    float alpha = u.x, beta = u.y;
    if (alpha + beta > 1) { alpha = 1 - alpha; beta = 1 - beta; }
    float gamma = 1 - beta - alpha;
    return alpha * A + beta * B + gamma * C;
}
float determinant_accurate(vec3 nA, vec3 nB, vec3 nC) {      // math.glsl:373-382
    float div = inversesqrt(2.0f * fabsf(nB.x) + 2.0f);
    float e = nB.x > 0 ? div : -div;
    vec3 h = nB * div + V3(e, 0, 0);
    vec3 a = nA - 2.0f * h * dot(h, nA);
    vec3 c = nC - 2.0f * h * dot(h, nC);
    return fabsf(a.y * c.z - c.y * a.z);
}
vec3 sample_spherical_triangle(vec2 xi, vec3 A, vec3 B, vec3 C, float& pdf) {   // math.glsl:385-419
    vec3 nA = normalize(A), nB = normalize(B), nC = normalize(C);
    float dAB = dot(nA, nB), dBC = dot(nB, nC), dAC = dot(nA, nC);
    float div = inversesqrt(2.0f * fabsf(nB.x) + 2.0f);
    float e = nB.x > 0 ? div : -div;
    vec3 h = nB * div + V3(e, 0, 0);
    vec3 a = nA - 2.0f * h * (dAB * div + e * nA.x);
    vec3 c = nC - 2.0f * h * (dBC * div + e * nC.x);
    float G0 = fabsf(a.y * c.z - c.y * a.z);
    float G1 = dAC + dBC;
    float G2 = 1.0f + dAB;
    float solid_angle = 2.0f * atan2f(G0, G1 + G2);
    pdf = 1.0f / solid_angle;
    float chosen_split = xi.x * solid_angle * 0.5f;
    vec3 r = (G0 * cosf(chosen_split) - G1 * sinf(chosen_split)) * nA + G2 * sinf(chosen_split) * nC;
    vec3 Ch = 2.0f * dot(nA, r) * r / dot(r, r) - nA;
    float d = dot(Ch, nB);
    float z = 1 - xi.y + d * xi.y;
    float st = sqrtf((1.0f - z * z) / (1.0f - d * d));
    return (z - st * d) * nB + st * Ch;
}
float spherical_triangle_solid_angle(vec3 nA, vec3 nB, vec3 nC) {   // math.glsl:422-429
    return 2.0f * atan2f(determinant_accurate(nA, nB, nC), 1.0f + (dot(nA, nB) + (dot(nB, nC) + dot(nA, nC))));
}
float triangle_area_pdf(vec3 p, vec3 a, vec3 b, vec3 c) {           // math.glsl:441-447
    vec3 normal = cross(a - b, a - c);
    float p_dist2 = dot(p, p);
    return 2.0f * p_dist2 * sqrtf(p_dist2) / fabsf(dot(normal, p));
}
float ray_plane_intersection_dist(vec3 dir, vec3 A, vec3 B, vec3 C) {   // math.glsl:450-456
    vec3 pn = normalize(cross(A - B, A - C));
    float pw = dot(A, pn);
    return fabsf(pw / dot(pn, dir));
}
vec3 get_barycentric_coords(vec3 p, vec3 A, vec3 B, vec3 C) {        // math.glsl:458-470
    vec3 ba = B - A, ca = C - A, pa = p - A;
    float bb = dot(ba, ba), bc = dot(ba, ca), cc = dot(ca, ca), pb = dot(pa, ba), pc = dot(pa, ca);
    float denom = 1.0f / (bb * cc - bc * bc);
    vec3 bary;
    bary.y = (cc * pb - bc * pc) * denom;
    bary.z = (bb * pc - bc * pb) * denom;
    bary.x = 1.0f - bary.y - bary.z;
    return bary;
}

// color.glsl
vec3 inverse_srgb_correction(vec3 col) {               // color.glsl:7-12
    vec3 low = col * 0.07739938f;
    vec3 high = pow3(V3(fmaf(col.x, 0.94786729f, 0.05213270f), fmaf(col.y, 0.94786729f, 0.05213270f),
                        fmaf(col.z, 0.94786729f, 0.05213270f)), V3(2.4f));
    return V3(0.04045f < col.x ? high.x : low.x, 0.04045f < col.y ? high.y : low.y, 0.04045f < col.z ? high.z : low.z);
}
float rgb_to_luminance(vec3 col) { return dot(col, V3(0.2126f, 0.7152f, 0.0722f)); }
uint rgb_to_r9g9b9e5(vec3 color) {                      // color.glsl:19-28
    int ex[3];
    frexpf(color.x, &ex[0]); frexpf(color.y, &ex[1]); frexpf(color.z, &ex[2]);
    int e = clamp(max(ex[0], max(ex[1], ex[2])), -16, 15);
    float sc = exp2f((float)-e) * 512.0f;
    int r = clamp((int)floorf(color.x * sc), 0, 511);
    int g = clamp((int)floorf(color.y * sc), 0, 511);
    int b = clamp((int)floorf(color.z * sc), 0, 511);
    return (uint)r | ((uint)g << 9) | ((uint)b << 18) | ((uint)(e + 16) << 27);
}
vec3 r9g9b9e5_to_rgb(uint rgbe) {                       // color.glsl:30-34
    int r = rgbe & 0x1FF, g = (rgbe >> 9) & 0x1FF, b = (rgbe >> 18) & 0x1FF, a = (rgbe >> 27) & 0x1FF;
    return V3((float)r, (float)g, (float)b) * (1.0f / 512.0f) * exp2f((float)(a - 16));
}

// alias_table.glsl
int latlong_direction_to_pixel_id(vec3 dir, int sx, int sy) {   // alias_table.glsl:22-27
    vec2 uv = V2(atan2f(dir.z, dir.x) * 0.5f, asinf(-dir.y)) / M_PI_F + 0.5f;
    int px = (int)(uv.x * sx + 0.5f), py = (int)(uv.y * sy + 0.5f);
    return px + py * sx;
}
vec3 uv_to_latlong_direction(vec2 uv) {                 // alias_table.glsl:29-35
    uv = (uv - 0.5f) * M_PI_F;
    vec3 dir = V3(cosf(2.0f * uv.x), -sinf(uv.y), sinf(2.0f * uv.x));
    float s = sqrtf(1 - dir.y * dir.y);
    dir.x *= s; dir.z *= s;
    return dir;
}

// ----------------------------------------------------------------------------
// Scene + BVH (stand-in for the Vulkan acceleration structure)
// ----------------------------------------------------------------------------
struct wtri { vec3 v0, v1, v2; int inst, prim; uint gid; uint8_t non_opaque; };
struct bnode { vec3 bmin, bmax; int left, right; int first, count; };   // count>0 => leaf

struct counters_t { std::atomic<uint64_t> closest{0}, shadow{0}, nodes{0}, tris{0}, alpha{0}, surface{0}; };

}  // namespace

struct oracle_scene {
    // oracle_scene_set_shard: view / sample shard of a multi-device job (mirrors trhip_pt_set_shard)
    uint shard_vp_base = 0, shard_vp_stride = 1, shard_sample_base = 0, shard_sample_stride = 1;
    std::vector<instance> instances;
    std::vector<mesh_span> spans;
    std::vector<vertex> vertices;
    std::vector<uint> indices;
    std::vector<point_light> point_lights;
    std::vector<directional_light> directional_lights;
    std::vector<tri_light> tri_lights;
    std::vector<texture_info> tex_infos;
    std::vector<uint8_t> texels;
    std::vector<vec4> envmap;
    uint env_w = 0, env_h = 0;
    std::vector<alias_table_entry> alias_table;
    vec4 environment_factor;
    int environment_proj = -1;
    std::vector<camera_data> cameras;
    std::vector<camera_data> prev_cameras;   // camera_pair.previous (shader/scene.glsl:176-185); = cameras until set
    std::vector<uint8_t> non_opaque;
    // pre_transform.comp output (world-space vertex copy, one span per instance); built on first use
    std::vector<vertex> world_vertices;
    std::vector<mesh_span> world_spans;
    std::vector<wtri> tris;       // BVH order
    std::vector<bnode> nodes;
    counters_t counters;
};

namespace {

inline vec3 transform_point(const mat4& m, vec3 p) {   // (model * vec4(pos, 1)).xyz, GLSL evaluation order
    vec4 r = m * V4(p, 1.0f);
    return V3(r);
}

struct bbox { vec3 lo, hi; };
inline bbox empty_box() { float inf = std::numeric_limits<float>::infinity(); return {V3(inf), V3(-inf)}; }
inline void grow(bbox& b, vec3 p) { b.lo = min(b.lo, p); b.hi = max(b.hi, p); }
inline void grow(bbox& b, const bbox& o) { b.lo = min(b.lo, o.lo); b.hi = max(b.hi, o.hi); }
inline float half_area(const bbox& b) { vec3 d = b.hi - b.lo; return d.x * d.y + d.y * d.z + d.z * d.x; }

struct build_ctx {
    std::vector<wtri>& tris;
    std::vector<bbox> boxes;
    std::vector<vec3> cents;
    std::vector<uint> order;
    std::vector<bnode>& nodes;
};

int build_node(build_ctx& c, uint begin, uint end) {
    int id = (int)c.nodes.size();
    c.nodes.push_back({});
    bbox b = empty_box(), cb = empty_box();
    for (uint i = begin; i < end; ++i) { grow(b, c.boxes[c.order[i]]); grow(cb, c.cents[c.order[i]]); }
    c.nodes[id].bmin = b.lo; c.nodes[id].bmax = b.hi;
    uint n = end - begin;
    auto make_leaf = [&]() { c.nodes[id].first = (int)begin; c.nodes[id].count = (int)n; c.nodes[id].left = c.nodes[id].right = -1; return id; };
    if (n <= 2) return make_leaf();
    const int NB = 16;
    float best_cost = std::numeric_limits<float>::infinity();
    int best_axis = -1, best_split = -1;
    for (int axis = 0; axis < 3; ++axis) {
        float lo = idx(cb.lo, axis), hi = idx(cb.hi, axis);
        if (!(hi > lo)) continue;
        bbox bb[NB]; uint cnt[NB];
        for (int k = 0; k < NB; ++k) { bb[k] = empty_box(); cnt[k] = 0; }
        float scale = NB / (hi - lo);
        for (uint i = begin; i < end; ++i) {
            uint t = c.order[i];
            int k = min(NB - 1, (int)((idx(c.cents[t], axis) - lo) * scale));
            grow(bb[k], c.boxes[t]); cnt[k]++;
        }
        float la[NB], ra[NB]; uint lc[NB], rc[NB];
        bbox acc = empty_box(); uint ac = 0;
        for (int k = 0; k < NB; ++k) { grow(acc, bb[k]); ac += cnt[k]; la[k] = half_area(acc); lc[k] = ac; }
        acc = empty_box(); ac = 0;
        for (int k = NB - 1; k >= 0; --k) { grow(acc, bb[k]); ac += cnt[k]; ra[k] = half_area(acc); rc[k] = ac; }
        for (int k = 0; k < NB - 1; ++k) {
            if (lc[k] == 0 || rc[k + 1] == 0) continue;
            float cost = la[k] * lc[k] + ra[k + 1] * rc[k + 1];
            if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = k; }
        }
    }
    uint mid;
    if (best_axis < 0) {
        if (n <= 4) return make_leaf();
        mid = begin + n / 2;
    } else {
        float leaf_cost = half_area(b) * n;
        if (n <= 4 && leaf_cost <= best_cost + half_area(b)) return make_leaf();
        float lo = idx(cb.lo, best_axis), hi = idx(cb.hi, best_axis);
        float scale = NB / (hi - lo);
        auto it = std::partition(c.order.begin() + begin, c.order.begin() + end, [&](uint t) {
            int k = min(NB - 1, (int)((idx(c.cents[t], best_axis) - lo) * scale));
            return k <= best_split;
        });
        mid = (uint)(it - c.order.begin());
        if (mid == begin || mid == end) mid = begin + n / 2;
    }
    int l = build_node(c, begin, mid);
    int r = build_node(c, mid, end);
    c.nodes[id].left = l; c.nodes[id].right = r; c.nodes[id].count = 0; c.nodes[id].first = 0;
    return id;
}

void build_scene_accel(oracle_scene& s) {
    std::vector<wtri> tris;
    uint gid = 0;
    for (size_t i = 0; i < s.instances.size(); ++i) {
        const instance& o = s.instances[i];
        const mesh_span& sp = s.spans[i];
        for (uint p = 0; p < sp.triangle_count; ++p, ++gid) {
            wtri t;
            const uint* ix = &s.indices[sp.index_offset + 3 * p];
            t.v0 = transform_point(o.model, s.vertices[sp.vertex_offset + ix[0]].pos);
            t.v1 = transform_point(o.model, s.vertices[sp.vertex_offset + ix[1]].pos);
            t.v2 = transform_point(o.model, s.vertices[sp.vertex_offset + ix[2]].pos);
            t.inst = (int)i; t.prim = (int)p; t.gid = gid; t.non_opaque = s.non_opaque[i];
            tris.push_back(t);
        }
    }
    build_ctx c{tris, {}, {}, {}, s.nodes};
    c.boxes.resize(tris.size()); c.cents.resize(tris.size()); c.order.resize(tris.size());
    for (size_t i = 0; i < tris.size(); ++i) {
        bbox b = empty_box();
        grow(b, tris[i].v0); grow(b, tris[i].v1); grow(b, tris[i].v2);
        c.boxes[i] = b; c.cents[i] = (b.lo + b.hi) * 0.5f; c.order[i] = (uint)i;
    }
    s.nodes.clear();
    s.nodes.reserve(tris.size() * 2 + 1);
    if (!tris.empty()) build_node(c, 0, (uint)tris.size());
    s.tris.resize(tris.size());
    for (size_t i = 0; i < tris.size(); ++i) s.tris[i] = tris[c.order[i]];
}

// extract_tri_lights.comp:17-54
void extract_tri_lights(oracle_scene& s) {
    s.tri_lights.clear();
    for (size_t i = 0; i < s.instances.size(); ++i) {
        const instance& o = s.instances[i];
        if (o.light_base_id < 0) continue;
        const mesh_span& sp = s.spans[i];
        if (s.tri_lights.size() < (size_t)o.light_base_id + sp.triangle_count)
            s.tri_lights.resize((size_t)o.light_base_id + sp.triangle_count);
        for (uint p = 0; p < sp.triangle_count; ++p) {
            tri_light& l = s.tri_lights[o.light_base_id + p];
            const uint* ix = &s.indices[sp.index_offset + 3 * p];
            const vertex& v0 = s.vertices[sp.vertex_offset + ix[0]];
            const vertex& v1 = s.vertices[sp.vertex_offset + ix[1]];
            const vertex& v2 = s.vertices[sp.vertex_offset + ix[2]];
            l.emission_tex_id = o.mat.emission_tex_id;
            l.emission_factor = rgb_to_r9g9b9e5(V3(o.mat.emission_factor));
            l.instance_id = (uint)i; l.primitive_id = p;
            l.pos[0] = transform_point(o.model, v0.pos);
            l.pos[1] = transform_point(o.model, v1.pos);
            l.pos[2] = transform_point(o.model, v2.pos);
            l.uv[0] = packHalf2x16(v0.uv); l.uv[1] = packHalf2x16(v1.uv); l.uv[2] = packHalf2x16(v2.uv);
        }
    }
}

// ----------------------------------------------------------------------------
// Textures: bilinear, repeat, LOD 0, unnormalised RGBA8 (src/sampler_table.cc:8-17)
// ----------------------------------------------------------------------------
inline int wrap_repeat(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }

vec4 sample_texture(const oracle_scene& s, int tex_id, vec2 uv) {
    const texture_info& ti = s.tex_infos[tex_id];
    int w = (int)ti.width, h = (int)ti.height;
    // a texture unit returns a texel for any coordinate; non-finite ones are defined as 0 (see csrc/texture.h)
    if (!(fabsf(uv.x) < INFINITY)) uv.x = 0.0f;
    if (!(fabsf(uv.y) < INFINITY)) uv.y = 0.0f;
    float x = uv.x * (float)w - 0.5f, y = uv.y * (float)h - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    int x0 = wrap_repeat((int)fx0, w), y0 = wrap_repeat((int)fy0, h);
    int x1 = wrap_repeat((int)fx0 + 1, w), y1 = wrap_repeat((int)fy0 + 1, h);
    const uint8_t* base = s.texels.data() + (size_t)ti.texel_offset * 4;
    auto fetch = [&](int xx, int yy) {
        if (ti.format == 1) {
            const uint16_t* q = reinterpret_cast<const uint16_t*>(base) + ((size_t)yy * w + xx) * 4;
            return V4(q[0], q[1], q[2], q[3]) * (1.0f / 65535.0f);
        }
        const uint8_t* p = base + ((size_t)yy * w + xx) * 4;
        return V4(p[0], p[1], p[2], p[3]) * (1.0f / 255.0f);
    };
    vec4 c00 = fetch(x0, y0), c10 = fetch(x1, y0), c01 = fetch(x0, y1), c11 = fetch(x1, y1);
    vec4 top = c00 * (1.0f - fx) + c10 * fx;
    vec4 bot = c01 * (1.0f - fx) + c11 * fx;
    return top * (1.0f - fy) + bot * fy;
}

vec4 sample_envmap(const oracle_scene& s, vec2 uv) {
    int w = (int)s.env_w, h = (int)s.env_h;
    float x = uv.x * (float)w - 0.5f, y = uv.y * (float)h - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    int x0 = wrap_repeat((int)fx0, w), y0 = wrap_repeat((int)fy0, h);
    int x1 = wrap_repeat((int)fx0 + 1, w), y1 = wrap_repeat((int)fy0 + 1, h);
    vec4 c00 = s.envmap[(size_t)y0 * w + x0], c10 = s.envmap[(size_t)y0 * w + x1];
    vec4 c01 = s.envmap[(size_t)y1 * w + x0], c11 = s.envmap[(size_t)y1 * w + x1];
    vec4 top = c00 * (1.0f - fx) + c10 * fx;
    vec4 bot = c01 * (1.0f - fx) + c11 * fx;
    return top * (1.0f - fy) + bot * fy;
}

// ----------------------------------------------------------------------------
// Ray / triangle: watertight test (Woop et al. 2013), no culling
// ----------------------------------------------------------------------------
struct ray_pre { int kx, ky, kz; float Sx, Sy, Sz; vec3 org, dir, inv_dir; };

ray_pre make_ray(vec3 org, vec3 dir) {
    ray_pre r;
    r.org = org; r.dir = dir;
    float ax = fabsf(dir.x), ay = fabsf(dir.y), az = fabsf(dir.z);
    int kz = (ax > ay) ? (ax > az ? 0 : 2) : (ay > az ? 1 : 2);
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    if (idx(dir, kz) < 0.0f) { int t = kx; kx = ky; ky = t; }
    r.kx = kx; r.ky = ky; r.kz = kz;
    r.Sx = idx(dir, kx) / idx(dir, kz);
    r.Sy = idx(dir, ky) / idx(dir, kz);
    r.Sz = 1.0f / idx(dir, kz);
    r.inv_dir = V3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    return r;
}

inline bool tri_intersect(const ray_pre& r, const wtri& tr, float tmin, float tmax, float& t, float& bu, float& bv) {
    const vec3 A = tr.v0 - r.org, B = tr.v1 - r.org, C = tr.v2 - r.org;
    const float Akz = idx(A, r.kz), Bkz = idx(B, r.kz), Ckz = idx(C, r.kz);
    const float Ax = idx(A, r.kx) - r.Sx * Akz, Ay = idx(A, r.ky) - r.Sy * Akz;
    const float Bx = idx(B, r.kx) - r.Sx * Bkz, By = idx(B, r.ky) - r.Sy * Bkz;
    const float Cx = idx(C, r.kx) - r.Sx * Ckz, Cy = idx(C, r.ky) - r.Sy * Ckz;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        double CxBy = (double)Cx * (double)By, CyBx = (double)Cy * (double)Bx;
        U = (float)(CxBy - CyBx);
        double AxCy = (double)Ax * (double)Cy, AyCx = (double)Ay * (double)Cx;
        V = (float)(AxCy - AyCx);
        double BxAy = (double)Bx * (double)Ay, ByAx = (double)By * (double)Ax;
        W = (float)(BxAy - ByAx);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    const float det = U + V + W;
    if (det == 0.0f) return false;
    const float Az = r.Sz * Akz, Bz = r.Sz * Bkz, Cz = r.Sz * Ckz;
    const float T = U * Az + V * Bz + W * Cz;
    const float rcp = 1.0f / det;
    const float tt = T * rcp;
    if (!(tt > tmin && tt < tmax)) return false;
    t = tt; bu = V * rcp; bv = W * rcp;
    return true;
}

inline bool box_intersect(const ray_pre& r, vec3 bmin, vec3 bmax, float tmin, float tmax, float& tnear) {
    float t0 = tmin, t1 = tmax;
    for (int a = 0; a < 3; ++a) {
        float inv = idx(r.inv_dir, a);
        float tn = (idx(bmin, a) - idx(r.org, a)) * inv;
        float tf = (idx(bmax, a) - idx(r.org, a)) * inv;
        if (std::signbit(inv)) { float x = tn; tn = tf; tf = x; }   // by the direction's sign: comparing would mis-order a NaN
        // 0 * inf = NaN when the origin lies exactly on a plane of an axis the ray does not move along: the ray is inside
        // that slab (on its boundary), so the axis does not constrain the interval
        t0 = tn > t0 ? tn : t0;
        t1 = tf < t1 ? tf : t1;
    }
    // entry against min(exit, tmax) widened by a pad that covers the rounding of the slab distances and the error of the
    // triangle test's distance (a different expression): the closest hit must not depend on the shape of the tree
    if (t0 > t1 * 1.000004f) return false;
    tnear = t0;
    return true;
}

// Order-independent stand-in for `generate_single_uniform_random(payload.random_seed)` (rt_common.rahit:21).
inline float alpha_cutoff_hash(uint seed, int instance_id, int primitive_id) {
    uint k = (uint)instance_id * 0x9E3779B9u + (uint)primitive_id;
    uint h = seed ^ pcg(k);
    return (float)pcg(h) * INV_UINT32_MAX;
}

// get_interpolated_vertex_light (rt.glsl:103-117)
vec2 interpolated_uv(const oracle_scene& s, int instance_id, int primitive_id, vec2 bary) {
    const mesh_span& sp = s.spans[instance_id];
    const uint* ix = &s.indices[sp.index_offset + 3 * primitive_id];
    const vertex& v0 = s.vertices[sp.vertex_offset + ix[0]];
    const vertex& v1 = s.vertices[sp.vertex_offset + ix[1]];
    const vertex& v2 = s.vertices[sp.vertex_offset + ix[2]];
    vec3 b = V3(1.0f - bary.x - bary.y, bary.x, bary.y);
    return v0.uv * b.x + v1.uv * b.y + v2.uv * b.z;
}

// is_material_skippable (rt.glsl:136-144)
bool is_material_skippable(const oracle_scene& s, int instance_id, vec2 uv, float alpha_cutoff) {
    const material& mat = s.instances[instance_id].mat;
    vec4 albedo = mat.albedo_factor;
    if (mat.albedo_tex_id >= 0) albedo = albedo * sample_texture(s, mat.albedo_tex_id, uv);
    return albedo.w <= alpha_cutoff;
}


// ----------------------------------------------------------------------------
// traceRayEXT stand-ins
// ----------------------------------------------------------------------------
struct thread_counters { uint64_t closest = 0, shadow = 0, nodes = 0, tris = 0, alpha = 0, surface = 0; };

// sphere_intersection (rt_common.glsl:36-46) / get_point_light_hit_t (:48-51)
float sphere_intersection(vec3 sphere_pos, float sphere_radius, vec3 ray_origin, vec3 ray_direction) {
    vec3 oc = ray_origin - sphere_pos;
    float a = dot(ray_direction, ray_direction);
    float b = 2.0f * dot(oc, ray_direction);
    float c = dot(oc, oc) - sphere_radius * sphere_radius;
    float discriminant = b * b - 4.0f * a * c;
    if (discriminant < 0) return -1.0f;
    return (-b - sqrtf(discriminant)) / (2.0f * a);
}

// Closest hit: rt_common.rchit/.rmiss/.rahit + rt_common_point_light.rint/.rchit.
// `alpha_mode`: 0 = stochastic alpha (path tracer), 1 = fixed cutoff 1e-4 (rt_feature.rahit:17).
void trace_closest(const oracle_scene& s, vec3 org, vec3 dir, float tmin, float tmax, bool include_lights,
                   int alpha_mode, hit_payload& payload, float& hit_t, thread_counters& tc) {
    tc.closest++;
    payload.instance_id = -1;
    payload.primitive_id = -1;
    payload.barycentrics = V2(0);
    float best_t = tmax;
    uint best_gid = 0xFFFFFFFFu;
    bool found = false;
    // non-finite or zero-direction rays are outside traceRayEXT's contract (ggx.glsl:343-348 can emit out_dir = 0 at
    // bounce 0): defined as a miss
    const bool finite_ray = std::isfinite(org.x) && std::isfinite(org.y) && std::isfinite(org.z) &&
                            std::isfinite(dir.x) && std::isfinite(dir.y) && std::isfinite(dir.z) &&
                            (dir.x != 0.0f || dir.y != 0.0f || dir.z != 0.0f);
    if (!s.nodes.empty() && finite_ray) {
        ray_pre r = make_ray(org, dir);
        int stack[128];
        int sp = 0;
        stack[sp++] = 0;
        while (sp > 0) {
            int ni = stack[--sp];
            const bnode& n = s.nodes[ni];
            float tn;
            tc.nodes++;
            if (!box_intersect(r, n.bmin, n.bmax, tmin, best_t, tn)) continue;
            if (n.count > 0) {
                for (int k = 0; k < n.count; ++k) {
                    const wtri& tr = s.tris[n.first + k];
                    float t, bu, bv;
                    tc.tris++;
                    // closed upper bound so that equal-t candidates are seen by the tie-break
                    if (!tri_intersect(r, tr, tmin, std::numeric_limits<float>::infinity(), t, bu, bv)) continue;
                    if (!(t < best_t || (t == best_t && found && tr.gid < best_gid))) continue;
                    if (!(t < tmax)) continue;
                    if (tr.non_opaque) {
                        tc.alpha++;
                        vec2 uv = interpolated_uv(s, tr.inst, tr.prim, V2(bu, bv));
                        float cutoff = alpha_mode == 0 ? alpha_cutoff_hash(payload.random_seed, tr.inst, tr.prim) : 0.0001f;
                        if (is_material_skippable(s, tr.inst, uv, cutoff)) continue;
                    }
                    best_t = t; best_gid = tr.gid; found = true;
                    payload.instance_id = tr.inst;
                    payload.primitive_id = tr.prim;
                    payload.barycentrics = V2(bu, bv);
                }
            } else {
                // near child first
                const bnode& l = s.nodes[n.left];
                const bnode& rr = s.nodes[n.right];
                float tl, trr;
                bool hl = box_intersect(r, l.bmin, l.bmax, tmin, best_t, tl);
                bool hr = box_intersect(r, rr.bmin, rr.bmax, tmin, best_t, trr);
                if (hl && hr) {
                    if (tl <= trr) { stack[sp++] = n.right; stack[sp++] = n.left; }
                    else { stack[sp++] = n.left; stack[sp++] = n.right; }
                } else if (hl) stack[sp++] = n.left;
                else if (hr) stack[sp++] = n.right;
            }
        }
    }
    if (include_lights && finite_ray) {
        for (size_t i = 0; i < s.point_lights.size(); ++i) {
            const point_light& pl = s.point_lights[i];
            if (pl.radius == 0.0f) continue;   // degenerate AABB at the origin (src/scene_stage.cc:1366-1368)
            float hit = sphere_intersection(pl.pos, pl.radius, org, dir);
            if (hit > 0 && hit > tmin && hit < best_t) {
                best_t = hit; found = true;
                payload.instance_id = -1;
                payload.primitive_id = (int)i;
                payload.barycentrics = V2(hit, 0);
            }
        }
    }
    hit_t = found ? best_t : -1.0f;
    if (alpha_mode == 0) pcg(payload.random_seed);   // one advance per closest-hit trace (see header)
}

// shadow_ray (path_tracer.glsl:35-52) with rt_common_shadow.rahit/.rchit; lights excluded (mask 0xFD)
float trace_shadow(const oracle_scene& s, vec3 org, vec3 dir, float tmin, float tmax, thread_counters& tc) {
    tc.shadow++;
    float visibility = 1.0f;
    if (s.nodes.empty()) return visibility;
    if (!(std::isfinite(org.x) && std::isfinite(org.y) && std::isfinite(org.z) && std::isfinite(dir.x) && std::isfinite(dir.y) &&
          std::isfinite(dir.z) && (dir.x != 0.0f || dir.y != 0.0f || dir.z != 0.0f))) return visibility;
    ray_pre r = make_ray(org, dir);
    int stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        int ni = stack[--sp];
        const bnode& n = s.nodes[ni];
        float tn;
        tc.nodes++;
        if (!box_intersect(r, n.bmin, n.bmax, tmin, tmax, tn)) continue;
        if (n.count > 0) {
            for (int k = 0; k < n.count; ++k) {
                const wtri& tr = s.tris[n.first + k];
                float t, bu, bv;
                tc.tris++;
                if (!tri_intersect(r, tr, tmin, tmax, t, bu, bv)) continue;
                if (!tr.non_opaque) return 0.0f;
                tc.alpha++;
                const material& mat = s.instances[tr.inst].mat;
                vec2 uv = interpolated_uv(s, tr.inst, tr.prim, V2(bu, bv));
                float alpha = mat.albedo_factor.w;
                if (mat.albedo_tex_id >= 0) alpha *= sample_texture(s, mat.albedo_tex_id, uv).w;
                visibility *= 1.0f - alpha;
                if (visibility == 0.0f) return 0.0f;
            }
        } else {
            stack[sp++] = n.left;
            stack[sp++] = n.right;
        }
    }
    return visibility;
}

// ----------------------------------------------------------------------------
// ggx.glsl
// ----------------------------------------------------------------------------
float ggx_fresnel_schlick(float cos_d, float f0) { return f0 + (1.0f - f0) * powf(max(1.0f - cos_d, 0.0f), 5.0f); }  // :30-33
float ggx_fresnel(float cos_d, const sampled_material& mat) {     // ggx.glsl:36-49
    if (mat.ior_in > mat.ior_out) {
        float inv_eta = mat.ior_in / mat.ior_out;
        float sin_theta2 = inv_eta * inv_eta * (1.0f - cos_d * cos_d);
        if (sin_theta2 >= 1.0f) return 1.0f;
        cos_d = sqrtf(1.0f - sin_theta2);
    } else if (mat.ior_in == mat.ior_out) return 0.0f;
    return ggx_fresnel_schlick(cos_d, mat.f0);
}
float fresnel_importance(float cos_d, const sampled_material& mat) {   // ggx.glsl:54-67
    if (mat.ior_in > mat.ior_out) {
        float inv_eta = mat.ior_in / mat.ior_out;
        float sin_theta2 = inv_eta * inv_eta * (1.0f - cos_d * cos_d);
        if (sin_theta2 >= 1.0f) return 1.0f;
        cos_d = sqrtf(1.0f - sin_theta2);
    } else if (mat.ior_in == mat.ior_out) return 0.0f;
    return mat.f0 + (max(1.0f - mat.roughness, mat.f0) - mat.f0) * powf(1.0f - cos_d, 5.0f);
}
float ggx_fresnel_refl(float cos_d, const sampled_material& mat) { return ggx_fresnel_schlick(cos_d, mat.f0); }  // :76-79
float ggx_masking(float v_dot_n, float v_dot_h, float a) {         // ggx.glsl:82-87
    float a2 = a * a;
    return step(0.0f, v_dot_n * v_dot_h) * 2.0f / (1.0f + sqrtf(1.0f + a2 / (v_dot_n * v_dot_n) - a2));
}
float ggx_masking_shadowing(float v_dot_n, float v_dot_h, float l_dot_n, float l_dot_h, float a) {   // :90-97
    float a2 = a * a;
    return step(0.0f, v_dot_n * v_dot_h) * step(0.0f, l_dot_n * l_dot_h) * 4.0f /
           ((1.0f + sqrtf(1.0f + a2 / max(v_dot_n * v_dot_n, 1e-18f) - a2)) *
            (1.0f + sqrtf(1.0f + a2 / max(l_dot_n * l_dot_n, 1e-18f) - a2)));
}
float ggx_masking_shadowing_predivided(float v_dot_n, float v_dot_h, float l_dot_n, float l_dot_h, float a) {  // :101-109
    float a2 = a * a;
    float denom1 = fabsf(l_dot_n) * sqrtf(a2 + (1.0f - a2) * v_dot_n * v_dot_n);
    float denom2 = fabsf(v_dot_n) * sqrtf(a2 + (1.0f - a2) * l_dot_n * l_dot_n);
    return step(0.0f, v_dot_n * v_dot_h) * step(0.0f, l_dot_n * l_dot_h) * 0.5f / (denom1 + denom2);
}
float ggx_distribution(float h_dot_n, float a) {                   // ggx.glsl:114-119
    float a2 = a * a;
    float denom = h_dot_n * h_dot_n * (a2 - 1.0f) + 1.0f;
    return a2 / (M_PI_F * denom * denom);
}
void ggx_brdf_inner(vec3 out_dir, vec3 view_dir, vec3 h, float fresnel, float distribution, float cos_d,
                    const sampled_material& mat, bsdf_lobes& bsdf) {   // ggx.glsl:123-147
    float cos_l = out_dir.z, cos_v = view_dir.z;
    float geometry = ggx_masking_shadowing_predivided(cos_v, cos_d, cos_l, dot(out_dir, h), mat.roughness);
    float kd = (1.0f - fresnel) * (1.0f - mat.metallic) * (1.0f - mat.transmittance);
    cos_l = max(cos_l, 0.0f);
    bsdf.diffuse += kd * cos_l / M_PI_F;
    bsdf.dielectric_reflection += fresnel * geometry * distribution * cos_l * (1.0f - mat.metallic);
    bsdf.metallic_reflection += geometry * distribution * cos_l * mat.metallic;
}
vec3 ggx_vndf_sample(vec3 view, float roughness, float u1, float u2) {   // ggx.glsl:215-236
    vec3 v = normalize(V3(roughness * view.x, roughness * view.y, view.z));
    vec3 t1 = v.z < 0.9999f ? normalize(cross(v, V3(0, 0, 1))) : V3(1, 0, 0);
    vec3 t2 = cross(t1, v);
    float inv_a = 1.0f + v.z;
    float a = 1.0f / inv_a;
    float r = sqrtf(u1);
    float phi = u2 < a ? u2 * inv_a * M_PI_F : M_PI_F + (u2 - a) / (1.0f - a) * M_PI_F;
    float p1 = r * cosf(phi);
    float p2 = r * sinf(phi) * (u2 < a ? 1.0f : v.z);
    float p3 = sqrtf(max(0.0f, 1.0f - p1 * p1 - p2 * p2));
    vec3 n = p1 * t1 + p2 * t2 + p3 * v;
    return normalize(V3(roughness * n.x, roughness * n.y, max(0.0f, n.z)));
}
// ggx_bsdf_sample_core (ggx.glsl:240-375) with eval_all_lobes = true (ggx_bsdf_sample :377-388)
void ggx_bsdf_sample(vec4 uniform_random, vec3 view_dir, const sampled_material& mat, vec3& out_dir,
                     bsdf_lobes& bsdf, float& pdf) {
    const bool zero_roughness = mat.roughness < 0.001f;
    vec3 h = zero_roughness ? V3(0, 0, 1) : ggx_vndf_sample(view_dir, mat.roughness, uniform_random.x, uniform_random.y);
    float cos_d = dot(view_dir, h);
    float fresnel = ggx_fresnel(cos_d, mat);
    float cos_v = view_dir.z;
    float max_albedo = max(mat.albedo.x, max(mat.albedo.y, mat.albedo.z));
    float specular_cutoff = mix(1.0f, fresnel_importance(view_dir.z, mat), (1 - mat.metallic) * max_albedo);
    float diffuse_cutoff = 1.0f - mat.transmittance;
    float specular_probability = specular_cutoff;
    float diffuse_probability = (1.0f - specular_cutoff) * diffuse_cutoff;
    float transmissive_probability = (1.0f - specular_cutoff) * (1.0f - diffuse_cutoff);
    float u = uniform_random.z;
    pdf = 0.0f;
    out_dir = V3(0);
    if (u <= specular_cutoff) {   // Reflective
        out_dir = reflect(-view_dir, h);
        float cos_l = out_dir.z, cos_h = h.z;
        float G1 = ggx_masking(cos_v, cos_d, mat.roughness);
        float D = zero_roughness ? 4 * cos_l * cos_v : ggx_distribution(cos_h, mat.roughness);
        pdf = G1 * D / (4 * fabsf(cos_v)) * specular_probability;
        float diffuse_pdf = (zero_roughness ? 0 : pdf_cosine_hemisphere(out_dir) * diffuse_probability);
        pdf += diffuse_pdf;
        ggx_brdf_inner(out_dir, view_dir, h, fresnel, D, cos_d, mat, bsdf);
        if (zero_roughness) {
            bsdf.diffuse = 0;
            bsdf.dielectric_reflection /= pdf;
            bsdf.metallic_reflection /= pdf;
            pdf = 0;
        }
    } else {
        u = clamp((u - specular_cutoff) / (1 - specular_cutoff), 0.0f, 0.99999f);
        if (u <= diffuse_cutoff) {   // Diffuse
            u = clamp(u / diffuse_cutoff, 0.0f, 0.99999f);
            out_dir = sample_cosine_hemisphere(V2(u, uniform_random.w));
            h = normalize(view_dir + out_dir);
            float cos_h = h.z;
            cos_d = dot(view_dir, h);
            fresnel = ggx_fresnel_refl(cos_d, mat);
            float G1 = ggx_masking(cos_v, cos_d, mat.roughness);
            float D = (zero_roughness ? 0 : ggx_distribution(cos_h, mat.roughness));
            pdf = pdf_cosine_hemisphere(out_dir) * diffuse_probability;
            float specular_pdf = G1 * D / (4 * fabsf(cos_v)) * specular_probability;
            pdf += specular_pdf;
            ggx_brdf_inner(out_dir, view_dir, h, fresnel, D, cos_d, mat, bsdf);
            if (zero_roughness) { bsdf.dielectric_reflection = 0; bsdf.metallic_reflection = 0; }
        } else {   // Transmissive
            out_dir = normalize(refract(-view_dir, h, mat.ior_in / mat.ior_out));
            if (any_nan(out_dir)) { out_dir = V3(0); pdf = 0; return; }
            float cos_l = out_dir.z, cos_h = h.z;
            float cos_o = dot(out_dir, h);
            float G2 = ggx_masking_shadowing(cos_v, cos_d, cos_l, cos_o, mat.roughness);
            float G1 = ggx_masking(cos_v, cos_d, mat.roughness);
            float D = zero_roughness ? 4 * cos_l * cos_v : ggx_distribution(cos_h, mat.roughness);
            float denom = mat.ior_in / mat.ior_out * cos_d + cos_o;
            bsdf.transmission += fabsf(cos_d * cos_o) * mat.transmittance * (1.0f - mat.metallic) * (1.0f - fresnel) * G2 * D /
                                 (denom * denom * fabsf(cos_v));
            pdf = (fabsf(cos_d * cos_o) * G1 * D) / (denom * denom * fabsf(cos_v)) * transmissive_probability;
            if (zero_roughness) { bsdf.transmission /= pdf; pdf = 0; }
        }
    }
}
// ggx_bsdf_lobe_pdf with lobe = MATERIAL_LOBE_ALL (ggx.glsl:403-500, :502-510)
float ggx_bsdf_pdf(vec3 out_dir, vec3 view_dir, const sampled_material& mat, bsdf_lobes& bsdf) {
    float cos_l = out_dir.z, cos_v = view_dir.z;
    vec3 h;
    if (cos_l > 0) h = normalize(view_dir + out_dir);
    else h = (mat.ior_in > mat.ior_out ? 1.0f : -1.0f) * normalize(mat.ior_out * out_dir + mat.ior_in * view_dir);
    float cos_h = h.z;
    float cos_d = dot(view_dir, h);
    float cos_o = dot(out_dir, h);
    float fresnel = ggx_fresnel(cos_d, mat);
    float geometry = ggx_masking_shadowing_predivided(cos_v, cos_d, cos_l, cos_o, mat.roughness);
    const bool zero_roughness = mat.roughness < 0.001f;
    float distribution = zero_roughness ? 0 : ggx_distribution(cos_h, mat.roughness);
    float max_albedo = max(mat.albedo.x, max(mat.albedo.y, mat.albedo.z));
    float specular_cutoff = mix(1.0f, fresnel_importance(view_dir.z, mat), (1 - mat.metallic) * max_albedo);
    float diffuse_cutoff = 1.0f - mat.transmittance;
    float specular_probability = specular_cutoff;
    float diffuse_probability = (1.0f - specular_cutoff) * diffuse_cutoff;
    float transmissive_probability = (1.0f - specular_cutoff) * (1.0f - diffuse_cutoff);
    float G1 = ggx_masking(cos_v, cos_d, mat.roughness);
    float pdf = 0.0f;
    if (cos_l > 0) {
        float kd = (1.0f - fresnel) * (1.0f - mat.metallic) * (1.0f - mat.transmittance);
        float diffuse_pdf = pdf_cosine_hemisphere(out_dir) * diffuse_probability;
        if (!std::isnan(diffuse_pdf) && !std::isinf(diffuse_pdf) && diffuse_pdf > 0.0f) {
            bsdf.diffuse += kd * cos_l / M_PI_F;
            pdf += diffuse_pdf;
        }
        float specular_pdf = G1 * distribution / (4 * fabsf(cos_v)) * specular_probability;
        if (!std::isnan(specular_pdf) && !std::isinf(specular_pdf) && specular_pdf > 0.0f) {
            bsdf.dielectric_reflection += fresnel * geometry * distribution * cos_l * (1.0f - mat.metallic);
            bsdf.metallic_reflection += geometry * distribution * cos_l * mat.metallic;
            pdf += specular_pdf;
        }
    } else {
        float denom = mat.ior_in / mat.ior_out * cos_d + cos_o;
        geometry *= 4.0f;
        float transmit_pdf = (fabsf(cos_d * cos_o) * G1 * distribution) / (fabsf(cos_v) * denom * denom * M_PI_F) * transmissive_probability;
        if (!std::isnan(transmit_pdf) && !std::isinf(transmit_pdf) && transmit_pdf > 0.0f) {
            bsdf.transmission += -cos_l * fabsf(cos_d * cos_o) * mat.transmittance * (1.0f - mat.metallic) * (1.0f - fresnel) * geometry * distribution / (denom * denom);
            pdf += transmit_pdf;
        }
    }
    return pdf;
}
// material_bsdf_sample / material_bsdf_pdf (ggx.glsl:512-552); bounce_mode 0 hemisphere, 1 cosine, 2 material
void material_bsdf_sample(int bounce_mode, vec4 uniform_random, vec3 view_dir, const sampled_material& mat,
                          vec3& out_dir, bsdf_lobes& bsdf, float& pdf) {
    if (bounce_mode == 0) {
        if (mat.transmittance > 0.0f) { out_dir = sample_sphere(V2(uniform_random.x, uniform_random.y)); pdf = 0.25f / M_PI_F; }
        else { out_dir = sample_hemisphere(V2(uniform_random.x, uniform_random.y)); pdf = 0.5f / M_PI_F; }
        ggx_bsdf_pdf(out_dir, view_dir, mat, bsdf);
    } else if (bounce_mode == 1) {
        float split = mat.transmittance * 0.5f;
        out_dir = (uniform_random.z < split ? -1.0f : 1.0f) * sample_cosine_hemisphere(V2(uniform_random.x, uniform_random.y));
        pdf = fabsf(out_dir.z / M_PI_F) * (uniform_random.z < split ? split : 1.0f - split);
        ggx_bsdf_pdf(out_dir, view_dir, mat, bsdf);
    } else ggx_bsdf_sample(uniform_random, view_dir, mat, out_dir, bsdf, pdf);
}
float material_bsdf_pdf(int bounce_mode, vec3 out_dir, vec3 view_dir, const sampled_material& mat, bsdf_lobes& bsdf) {
    if (bounce_mode == 0) {
        ggx_bsdf_pdf(out_dir, view_dir, mat, bsdf);
        if (mat.transmittance == 0 && out_dir.z <= 0) return 0.0f;
        return mat.transmittance > 0.0f ? 0.25f / M_PI_F : 0.5f / M_PI_F;
    } else if (bounce_mode == 1) {
        ggx_bsdf_pdf(out_dir, view_dir, mat, bsdf);
        if (mat.transmittance == 0 && out_dir.z <= 0) return 0.0f;
        float split = mat.transmittance * 0.5f;
        return fabsf(out_dir.z / M_PI_F) * (out_dir.z < 0 ? split : 1.0f - split);
    }
    return ggx_bsdf_pdf(out_dir, view_dir, mat, bsdf);
}

// material.glsl:52-73
vec3 modulate_bsdf(const sampled_material& mat, const bsdf_lobes& bsdf) {
    return V3(mat.albedo) * (bsdf.metallic_reflection + bsdf.transmission + bsdf.diffuse) + bsdf.dielectric_reflection;
}
vec3 modulate_color(const sampled_material& mat, vec3 diffuse, vec3 reflected) {
    float approx_fresnel = 0.02f;
    diffuse = diffuse * V3(mat.albedo) * (1 - mat.metallic);
    reflected = reflected * mix(V3(approx_fresnel), V3(mat.albedo), mat.metallic) / mix(approx_fresnel, 1.0f, mat.metallic);
    return diffuse + reflected;
}
void add_demodulated_color(const bsdf_lobes& primary_bsdf, vec3 light_color, vec3& diffuse, vec3& reflected) {
    diffuse += light_color * (primary_bsdf.diffuse + primary_bsdf.transmission);
    reflected += light_color * (primary_bsdf.dielectric_reflection + primary_bsdf.metallic_reflection);
}

// ----------------------------------------------------------------------------
// light.glsl
// ----------------------------------------------------------------------------
float get_spotlight_intensity(const point_light& l, vec3 dir) {     // light.glsl:45-58
    if (l.dir_falloff > 0) {
        float cutoff = dot(dir, -l.dir);
        cutoff = cutoff > l.dir_cutoff ? 1.0f - powf(max(1.0f - cutoff, 0.0f) / (1.0f - l.dir_cutoff), l.dir_falloff) : 0.0f;
        return cutoff;
    }
    return 1.0f;
}
void sample_point_light(const point_light& pl, vec2 u, vec3 pos, vec3& out_dir, float& out_length, vec3& color, float& pdf) {  // :76-106
    vec3 dir = pos - pl.pos;
    float dist2 = dot(dir, dir);
    float k = 1.0f - pl.radius * pl.radius / dist2;
    float dir_cutoff = k > 0 ? sqrtf(k) : -1.0f;
    out_dir = sample_cone(u, -normalize(dir), dir_cutoff);
    float b = dot(dir, out_dir);
    out_length = -b - sqrtf(max(b * b - dist2 + pl.radius * pl.radius, 0.0f));
    color = get_spotlight_intensity(pl, normalize(-dir)) * pl.color;
    if (pl.radius == 0.0f) pdf = -dist2;
    else { color = color / (pl.radius * pl.radius * M_PI_F); pdf = 1 / (2.0f * M_PI_F * (1.0f - dir_cutoff)); }
}
float sample_point_light_pdf(const point_light& pl, vec3 pos) {     // light.glsl:108-117
    vec3 dir = pos - pl.pos;
    float dist2 = dot(dir, dir);
    float k = 1.0f - pl.radius * pl.radius / dist2;
    float dir_cutoff = k > 0 ? sqrtf(k) : -1.0f;
    if (pl.radius == 0.0f) return 0;
    return 1 / (2.0f * M_PI_F * (1.0f - dir_cutoff));
}
void sample_directional_light(const directional_light& dl, vec2 u, vec3& out_dir, vec3& color, float& pdf) {  // :119-129
    out_dir = sample_cone(u, -dl.dir, dl.dir_cutoff);
    pdf = dl.dir_cutoff >= 1.0f ? -1.0f : 1.0f / (2.0f * M_PI_F * (1.0f - dl.dir_cutoff));
    color = pdf > 0 ? dl.color * pdf : dl.color;
}
float sample_directional_light_pdf(const directional_light& dl) {   // light.glsl:131-134
    return dl.dir_cutoff >= 1.0f ? 0.0f : 1.0f / (2.0f * M_PI_F * (1.0f - dl.dir_cutoff));
}
// tri_light_mode: 0 area, 1 solid angle, 2 hybrid (light.glsl:136-179)
float sample_triangle_light_pdf(int mode, vec3 P, vec3 A, vec3 B, vec3 C) {
    if (mode == 0) return triangle_area_pdf(P, A, B, C);
    if (mode == 1) return 1.0f / spherical_triangle_solid_angle(normalize(A), normalize(B), normalize(C));
    float solid_angle = spherical_triangle_solid_angle(normalize(A), normalize(B), normalize(C));
    return solid_angle > 1e-6f ? 1.0f / solid_angle : triangle_area_pdf(P, A, B, C);
}
vec3 sample_triangle_light(int mode, vec2 u, vec3 A, vec3 B, vec3 C, float& pdf) {
    if (mode == 1) return sample_spherical_triangle(u, A, B, C, pdf);
    if (mode == 2) {
        float solid_angle = spherical_triangle_solid_angle(normalize(A), normalize(B), normalize(C));
        if (solid_angle > 1e-6f) return sample_spherical_triangle(u, A, B, C, pdf);
    }
    vec3 P = sample_triangle_area(u, A, B, C);
    pdf = triangle_area_pdf(P, A, B, C);
    return normalize(P);
}

// ----------------------------------------------------------------------------
// camera.glsl:25-59,99-124 and rt.glsl:234-247
// ----------------------------------------------------------------------------
void get_camera_ray(const camera_data& cam, int projection, bool dof, vec2 pixel_coord, vec2 screen_size, vec2 dof_u,
                    vec3& origin, vec3& dir) {
    vec2 uv = pixel_coord / screen_size;
    if (projection == 2) {
        // equirectangular_camera_data_buffer: {view, view_inverse, origin, fov} in the first 152 bytes
        const float* raw = reinterpret_cast<const float*>(&cam);
        vec4 eorigin = V4(raw[32], raw[33], raw[34], raw[35]);
        vec2 fov = V2(raw[36], raw[37]);
        uv = (uv * 2.0f - 1.0f) * fov;
        vec2 c = V2(cosf(uv.x), cosf(uv.y)), s = V2(sinf(uv.x), sinf(uv.y));
        vec3 t = V3(s.x * c.y, s.y, -c.x * c.y);
        dir = normalize(V3(cam.view_inverse * V4(t, 0)));
        origin = V3(eorigin);
        return;
    }
    uv = uv * 2.0f - 1.0f;
    if (projection == 0) {
        if (dof) {
            vec2 aperture_offset = cam.dof_params.w == 0 ? sample_concentric_disk(dof_u)
                                                         : sample_regular_polygon(dof_u, cam.dof_params.z, (uint)cam.dof_params.w);
            vec3 view_origin = V3(aperture_offset.x * cam.dof_params.y, aperture_offset.y * cam.dof_params.y, 0);
            vec3 view_dir = V3(cam.proj_inverse * V4(uv.x, uv.y, 1, 1)) * cam.dof_params.x;
            view_dir = normalize(view_dir - view_origin);
            origin = V3(cam.view_inverse * V4(view_origin, 1.0f));
            dir = V3(cam.view_inverse * V4(view_dir, 0));
            dir = normalize(dir);
        } else {
            origin = V3(cam.origin);
            vec4 t = cam.proj_inverse * V4(uv.x, uv.y, 1, 1);
            dir = normalize(V3(cam.view_inverse * V4(t.x, t.y, t.z, 0)));
        }
    } else {
        origin = V3(cam.view_inverse * (cam.proj_inverse * V4(uv.x, uv.y, 0, 1)));
        dir = normalize(V3(cam.view_inverse * V4(0, 0, -1, 0)));
    }
}

// distribution helpers (rt.glsl:170-231)
struct launch_ctx {
    oracle_distribution dist;     // `count` already holds b for shuffled strips (rt_camera_stage.cc:86-87)
    uvec3 launch_id;
    uvec2 launch_size;
    uint viewport = 0;            // the viewport layer launch_id.z shows (differs from it only in a view shard)
};
uint permute_region_id(uint i, uint size_x, uint size_y, uint b) {
    uint region_size = ((size_x * size_y) + (1u << b) - 1) >> b;
    uint region_id = i / region_size;
    uint k = b == 0 ? 0 : (bitfieldReverse(region_id) >> (32 - b));
    return k * region_size + i % region_size;
}
bool get_pixel_pos(const launch_ctx& L, ivec2& p) {
    if (L.dist.strategy == 0) { p = {(int)L.launch_id.x, (int)L.launch_id.y}; return true; }
    if (L.dist.strategy == 1) { p = {(int)L.launch_id.x, (int)(L.launch_id.y * L.dist.count + L.dist.index)}; return true; }
    uint j = permute_region_id(L.dist.index + L.launch_id.x, L.dist.size_x, L.dist.size_y, L.dist.count);
    if (j < L.dist.size_x * L.dist.size_y) { p = {(int)(j % L.dist.size_x), (int)(j / L.dist.size_x)}; return true; }
    return false;
}
bool get_write_pixel_pos(const launch_ctx& L, ivec3& p) {
    if (L.dist.strategy == 0) { p = {(int)L.launch_id.x, (int)L.launch_id.y, (int)L.launch_id.z}; return true; }
    if (L.dist.strategy == 1) {
        uint y = L.launch_id.y;
        if (L.dist.primary == 1) y = y * L.dist.count + L.dist.index;
        p = {(int)L.launch_id.x, (int)y, (int)L.launch_id.z};
        return true;
    }
    uvec3 wp = {L.launch_id.x % L.dist.size_x, L.launch_id.x / L.dist.size_x, L.launch_id.z};
    uint j = permute_region_id(L.dist.index + L.launch_id.x, L.dist.size_x, L.dist.size_y, L.dist.count);
    if (L.dist.primary == 1) wp = {j % L.dist.size_x, j / L.dist.size_x, L.launch_id.z};
    if (j < L.dist.size_x * L.dist.size_y) { p = {(int)wp.x, (int)wp.y, (int)wp.z}; return true; }
    return false;
}
uvec2 get_screen_size(const launch_ctx& L) {
    if (L.dist.strategy == 0) return L.launch_size;
    return {L.dist.size_x, L.dist.size_y};
}
void get_screen_camera_ray(const launch_ctx& L, ivec2 pixel, const camera_data& cam, int projection, bool dof,
                           vec2 pixel_offset, vec2 dof_u, vec3& origin, vec3& dir) {   // rt.glsl:234-241
    vec2 p = V2((float)pixel.x, (float)pixel.y) + (pixel_offset * 0.5f + 0.5f);
    uvec2 size = get_screen_size(L);
    p.y = (float)size.y - p.y;
    get_camera_ray(cam, projection, dof, p, V2((float)size.x, (float)size.y), dof_u, origin, dir);
}

// ----------------------------------------------------------------------------
// rt.glsl: vertex interpolation + material sampling
// ----------------------------------------------------------------------------
struct pt_ctx {
    const oracle_scene* s;
    oracle_pt_options opt;
    uint max_sobol_bounces;
    bool nee_point, nee_dir, nee_env, nee_tri;
    uint sample_counter, rng_seed;
    thread_counters* tc;
};

vertex_data get_interpolated_vertex(const pt_ctx& c, vec3 view, vec2 barycentrics, int instance_id, int primitive_id,
                                    vec3 pos, float& pdf) {   // rt.glsl:27-101
    const oracle_scene& s = *c.s;
    const instance& o = s.instances[instance_id];
    const bool pre = c.opt.pre_transformed_vertices != 0;   // PRE_TRANSFORMED_VERTICES: the stage binds scene_stage's pre-transformed copy
    const mesh_span& sp = pre ? s.world_spans[instance_id] : s.spans[instance_id];
    const std::vector<vertex>& verts = pre ? s.world_vertices : s.vertices;
    const uint* ix = &s.indices[sp.index_offset + 3 * primitive_id];
    const vertex& v0 = verts[sp.vertex_offset + ix[0]];
    const vertex& v1 = verts[sp.vertex_offset + ix[1]];
    const vertex& v2 = verts[sp.vertex_offset + ix[2]];
    vec3 b = V3(1.0f - barycentrics.x - barycentrics.y, barycentrics.x, barycentrics.y);
    vec4 avg_tangent = v0.tangent * b.x + v1.tangent * b.y + v2.tangent * b.z;
    vertex_data interp;
    vec4 model_pos = V4(v0.pos * b.x + v1.pos * b.y + v2.pos * b.z, 1);
    interp.pos = pre ? V3(model_pos) : V3(o.model * model_pos);
    {   // CALC_PREV_VERTEX_POS (rt.glsl:73-79).  With pre-transformed vertices the reference forms model_prev * inverse(model) *
        // model_pos with the driver's inverse(); here the model-space point comes from the untransformed vertices instead.
        const mesh_span& osp = s.spans[instance_id];
        const uint* oix = &s.indices[osp.index_offset + 3 * primitive_id];
        vec3 object_pos = s.vertices[osp.vertex_offset + oix[0]].pos * b.x + s.vertices[osp.vertex_offset + oix[1]].pos * b.y +
                          s.vertices[osp.vertex_offset + oix[2]].pos * b.z;
        interp.prev_pos = V3(o.model_prev * V4(object_pos, 1));
    }
    pdf = 0.0f;
    if (c.nee_tri) {
        if (o.light_base_id >= 0) {
            vec3 p0 = pre ? v0.pos : V3(o.model * V4(v0.pos, 1));
            vec3 p1 = pre ? v1.pos : V3(o.model * V4(v1.pos, 1));
            vec3 p2 = pre ? v2.pos : V3(o.model * V4(v2.pos, 1));
            pdf = sample_triangle_light_pdf(c.opt.tri_light_mode, interp.pos - pos, p0 - pos, p1 - pos, p2 - pos);
        }
    }
    mat3 mn = M3(o.model_normal);
    vec3 sn = v0.normal * b.x + v1.normal * b.y + v2.normal * b.z;
    interp.smooth_normal = normalize(pre ? sn : mn * sn);
    vec3 at = V3(avg_tangent);
    interp.tangent = normalize(pre ? at : mn * at);
    interp.bitangent = normalize(cross(interp.smooth_normal, interp.tangent) * avg_tangent.w);
    interp.uv = v0.uv * b.x + v1.uv * b.y + v2.uv * b.z;
    vec3 hn = cross(v1.pos - v0.pos, v2.pos - v0.pos);
    interp.hard_normal = normalize(pre ? hn : mn * hn);
    interp.back_facing = dot(interp.hard_normal, view) > 0;
    if (interp.back_facing) {
        interp.smooth_normal = -interp.smooth_normal;
        interp.hard_normal = -interp.hard_normal;
    }
    interp.mapped_normal = interp.smooth_normal;
    interp.instance_id = instance_id;
    interp.primitive_id = primitive_id;
    return interp;
}

sampled_material sample_material(const oracle_scene& s, int instance_id, vertex_data& v) {   // scene.glsl:87-152, rt.glsl:119-134
    const instance& o = s.instances[instance_id];
    const material& mat = o.mat;
    sampled_material res;
    res.albedo = mat.albedo_factor;
    if (mat.albedo_tex_id >= 0) {
        vec4 tex_col = sample_texture(s, mat.albedo_tex_id, v.uv);
        vec3 lin = inverse_srgb_correction(V3(tex_col));
        res.albedo = res.albedo * V4(lin, tex_col.w);
    }
    vec2 mr = V2(mat.metallic_roughness_factor.x, mat.metallic_roughness_factor.y);
    if (mat.metallic_roughness_tex_id >= 0) {
        vec4 t = sample_texture(s, mat.metallic_roughness_tex_id, v.uv);
        mr = mr * V2(t.z, t.y);
    }
    res.metallic = mr.x;
    res.roughness = mr.y * mr.y;
    if (mat.normal_tex_id >= 0) {
        mat3 tbn = M3(v.tangent, v.bitangent, v.smooth_normal);
        vec4 t = sample_texture(s, mat.normal_tex_id, v.uv);
        vec3 ts_normal = normalize(V3(t) * 2.0f - 1.0f);
        v.mapped_normal = normalize(tbn * (ts_normal * V3(mat.normal_factor, mat.normal_factor, 1.0f)));
        v.mapped_normal = any_nan(v.mapped_normal) ? v.smooth_normal : v.mapped_normal;
    }
    res.emission = V3(mat.emission_factor);
    if (mat.emission_tex_id >= 0) res.emission = res.emission * V3(sample_texture(s, mat.emission_tex_id, v.uv));
    res.transmittance = mat.transmittance;
    if (v.back_facing && res.transmittance > 0.0001f) { res.ior_in = mat.ior; res.ior_out = 1.0f; }
    else { res.ior_in = 1.0f; res.ior_out = mat.ior; }
    float f0 = (res.ior_out - res.ior_in) / (res.ior_out + res.ior_in);
    f0 *= f0;
    res.f0 = f0;
    res.flags = mat.flags;
    res.shadow_terminator_mul = o.shadow_terminator_mul;
    return res;
}

// rt.glsl:251-335
void get_nee_sampling_probabilities(const pt_ctx& c, float& point, float& triangle, float& directional, float& envmap) {
    const oracle_scene& s = *c.s;
    point = (c.nee_point && !s.point_lights.empty()) ? c.opt.nee_point : 0.0f;
    triangle = (c.nee_tri && !s.tri_lights.empty()) ? c.opt.nee_triangles : 0.0f;
    directional = (c.nee_dir && !s.directional_lights.empty()) ? c.opt.nee_directional : 0.0f;
    envmap = (c.nee_env && s.environment_proj >= 0) ? c.opt.nee_envmap : 0.0f;
    float sum = point + triangle + directional + envmap;
    float inv_sum = sum <= 0.0f ? 0.0f : (1.0f / sum + 1e-5f);
    point *= inv_sum; triangle *= inv_sum; directional *= inv_sum; envmap *= inv_sum;
}
vec3 sample_environment_map(const oracle_scene& s, uvec3 rand, vec3& shadow_ray_direction, float& shadow_ray_length, float& pdf) {
    vec3 color = V3(s.environment_factor);
    if (s.environment_proj >= 0) {
        uvec2 size = {s.env_w, s.env_h};
        const uint pixel_count = size.x * size.y;
        uvec2 ip = {clamp(rand.x / (0xFFFFFFFFu / size.x), 0u, size.x - 1u), clamp(rand.y / (0xFFFFFFFFu / size.y), 0u, size.y - 1u)};
        int i = int(ip.x + ip.y * size.x);
        alias_table_entry at = s.alias_table[i];
        pdf = at.pdf;
        if (rand.z > at.probability) { i = int(at.alias_id); pdf = at.alias_pdf; }
        ivec2 p = {int((uint)i % size.x), int((uint)i / size.x)};
        vec2 off = V2((float)(uint)(rand.x * pixel_count), (float)(uint)(rand.y * pixel_count)) * INV_UINT32_MAX;
        vec2 uv = (V2((float)p.x, (float)p.y) + off) / V2((float)size.x, (float)size.y);
        shadow_ray_direction = uv_to_latlong_direction(uv);
        color = color * V3(sample_envmap(s, uv));
    } else {
        pdf = 1.0f / (4.0f * M_PI_F);
        shadow_ray_direction = sample_sphere(V2((float)rand.x, (float)rand.y) * INV_UINT32_MAX);
    }
    shadow_ray_length = RAY_MAX_DIST;
    return color;
}
float sample_environment_map_pdf(const oracle_scene& s, vec3 dir) {
    if (s.environment_proj >= 0) {
        uint i = (uint)latlong_direction_to_pixel_id(dir, (int)s.env_w, (int)s.env_h);
        if (i >= s.alias_table.size()) i = (uint)s.alias_table.size() - 1;   // the GLSL read is out of bounds here
        return s.alias_table[i].pdf;
    }
    return 1.0f / (4.0f * M_PI_F);
}

// ----------------------------------------------------------------------------
// path_tracer.glsl
// ----------------------------------------------------------------------------
float bsdf_mis_pdf(const pt_ctx& c, const intersection_pdf& nee_pdf, float bsdf_pdf) {   // :54-76
    if (bsdf_pdf == 0.0f) return 1.0f;
    const oracle_scene& s = *c.s;
    float point_prob, triangle_prob, dir_prob, envmap_prob;
    get_nee_sampling_probabilities(c, point_prob, triangle_prob, dir_prob, envmap_prob);
    float avg_nee_pdf =
        nee_pdf.directional_light_pdf * dir_prob / (float)max((uint)s.directional_lights.size(), 1u) +
        nee_pdf.tri_light_pdf * triangle_prob / (float)max((uint)s.tri_lights.size(), 1u) +
        nee_pdf.envmap_pdf * envmap_prob +
        nee_pdf.point_light_pdf * point_prob / (float)max((uint)s.point_lights.size(), 1u);
    if (c.opt.mis_mode == 2) return (avg_nee_pdf * avg_nee_pdf + bsdf_pdf * bsdf_pdf) / bsdf_pdf;
    if (c.opt.mis_mode == 1) return avg_nee_pdf + bsdf_pdf;
    return avg_nee_pdf > 0 ? std::numeric_limits<float>::infinity() : bsdf_pdf;
}
float nee_mis_pdf(const pt_ctx& c, float nee_pdf, float bsdf_pdf) {                       // :78-89
    if (nee_pdf <= 0.0f) return -nee_pdf;
    if (c.opt.mis_mode == 2) return (nee_pdf * nee_pdf + bsdf_pdf * bsdf_pdf) / nee_pdf;
    if (c.opt.mis_mode == 1) return nee_pdf + bsdf_pdf;
    return nee_pdf;
}

bool get_intersection_info(const pt_ctx& c, const hit_payload& payload, vec3 origin, vec3 view, pt_vertex_data& v,
                           intersection_pdf& nee_pdf, sampled_material& mat, vec3& light) {   // :91-201
    const oracle_scene& s = *c.s;
    nee_pdf = {0, 0, 0, 0};
    mat = sampled_material{};
    mat.metallic = 1;
    mat.albedo = V4(0);
    // the other members stay unwritten in the GLSL for lights and misses; the material gbuffer target packs ior_out / ior_in,
    // so both implementations define them: roughness 0, transmittance 0, ior 1 / 1
    mat.ior_in = 1.0f; mat.ior_out = 1.0f;
    v = pt_vertex_data{};
    if (payload.instance_id >= 0) {
        c.tc->surface++;
        float pdf = 0.0f;
        vertex_data vd = get_interpolated_vertex(c, view, payload.barycentrics, payload.instance_id, payload.primitive_id, origin, pdf);
        mat = sample_material(s, payload.instance_id, vd);
        mat.albedo.w = 1.0f;
        if (c.nee_tri) {
            nee_pdf.tri_light_pdf = pdf == 0.0f ? 0.0f : pdf;
            light = mat.emission;
            mat.emission = V3(0);
        } else light = V3(0);
        v.pos = vd.pos; v.hard_normal = vd.hard_normal; v.smooth_normal = vd.smooth_normal;
        v.mapped_normal = vd.mapped_normal; v.instance_id = vd.instance_id; v.prev_pos = vd.prev_pos;
        return true;
    } else if (payload.primitive_id >= 0) {
        const point_light& pl = s.point_lights[payload.primitive_id];
        vec3 color = get_spotlight_intensity(pl, view) * pl.color / (pl.radius * pl.radius * M_PI_F);
        if (c.nee_point) {
            mat.emission = V3(0);
            light = color;
            nee_pdf.point_light_pdf = sample_point_light_pdf(pl, origin);
        } else { light = V3(0); mat.emission = color; }
        v.pos = origin + payload.barycentrics.x * view;
        v.prev_pos = v.pos;   // path_tracer.glsl:151
        v.mapped_normal = normalize(v.pos - pl.pos);
        v.instance_id = -1;
        mat.albedo = V4(0, 0, 0, 1);
        return false;
    } else {
        vec4 color = s.environment_factor;
        if (s.environment_proj >= 0) {
            vec2 uv;
            uv.y = asinf(-view.y) / M_PI_F + 0.5f;
            uv.x = atan2f(view.z, view.x) / (2 * M_PI_F) + 0.5f;
            vec4 t = sample_envmap(s, uv);
            color.x *= t.x; color.y *= t.y; color.z *= t.z;
        }
        mat.emission = V3(0);
        light = V3(0);
        for (size_t i = 0; i < s.directional_lights.size(); ++i) {
            const directional_light& dl = s.directional_lights[i];
            if (dl.dir_cutoff >= 1.0f) continue;
            float visible = step(dl.dir_cutoff, dot(view, -dl.dir));
            vec3 dcolor = visible * dl.color / (2.0f * M_PI_F * (1.0f - dl.dir_cutoff));
            if (c.nee_dir) {
                light += dcolor;
                nee_pdf.directional_light_pdf += visible * sample_directional_light_pdf(dl);
            } else mat.emission += dcolor;
        }
        v.instance_id = -1;
        v.pos = origin;
        v.prev_pos = v.pos;   // path_tracer.glsl:188
        v.mapped_normal = -view;
        mat.albedo = V4(0);
        if (c.nee_env) {
            light += V3(color);
            nee_pdf.envmap_pdf = s.environment_proj >= 0 ? sample_environment_map_pdf(s, view) : 0.0f;
        } else mat.emission += V3(color);
        return false;
    }
}

vec3 sample_explicit_light(const pt_ctx& c, uvec4 rand_uint, vec3 pos, vec3& out_dir, float& out_length, float& pdf) {   // :203-289
    const oracle_scene& s = *c.s;
    float point_prob, triangle_prob, dir_prob, envmap_prob;
    get_nee_sampling_probabilities(c, point_prob, triangle_prob, dir_prob, envmap_prob);
    vec4 u = to_float(rand_uint) * INV_UINT32_MAX;
    if (c.nee_point && (u.w -= point_prob) < 0) {
        const int light_count = int(s.point_lights.size());
        // random_sample_point_light (light.glsl:39-43)
        int light_index = clamp(int(u.z * light_count), 0, light_count - 1);
        float weight = (float)max(light_count, 1);
        const point_light& pl = s.point_lights[light_index];
        vec3 color;
        sample_point_light(pl, V2(u.x, u.y), pos, out_dir, out_length, color, pdf);
        pdf *= point_prob / weight;
        return color;
    }
    if (c.nee_tri && (u.w -= triangle_prob) < 0) {
        const int light_count = int(s.tri_lights.size());
        int light_index = clamp(int(u.z * light_count), 0, light_count - 1);
        const tri_light& tl = s.tri_lights[light_index];
        vec3 A = tl.pos[0] - pos, B = tl.pos[1] - pos, C = tl.pos[2] - pos;
        vec3 color = r9g9b9e5_to_rgb(tl.emission_factor);
        float tri_pdf = 0.0f;
        out_dir = sample_triangle_light(c.opt.tri_light_mode, V2(u.x, u.y), A, B, C, tri_pdf);
        out_length = ray_plane_intersection_dist(out_dir, A, B, C);
        if (std::isinf(tri_pdf) || tri_pdf <= 0 || out_length <= c.opt.min_ray_dist || any_nan(out_dir)) {
            pdf = 1.0f;
            out_dir = V3(0);
            return V3(0);
        }
        if (tl.emission_tex_id >= 0) {
            vec3 bary = get_barycentric_coords(out_dir * out_length, A, B, C);
            vec2 uv = bary.x * unpackHalf2x16(tl.uv[0]) + bary.y * unpackHalf2x16(tl.uv[1]) + bary.z * unpackHalf2x16(tl.uv[2]);
            color = color * V3(sample_texture(s, tl.emission_tex_id, uv));
        }
        out_length -= c.opt.min_ray_dist;
        pdf = triangle_prob * tri_pdf / light_count;
        return color;
    }
    if (c.nee_env && (u.w -= envmap_prob) < 0) {
        vec3 color = sample_environment_map(s, uvec3{rand_uint.x, rand_uint.y, rand_uint.z}, out_dir, out_length, pdf);
        pdf *= envmap_prob;
        return color;
    }
    if (c.nee_dir && (u.w -= dir_prob) < 0) {
        const int light_count = int(s.directional_lights.size());
        int light_index = clamp(int(u.z * light_count), 0, light_count - 1);
        const directional_light& dl = s.directional_lights[light_index];
        out_length = RAY_MAX_DIST;
        vec3 color;
        sample_directional_light(dl, V2(u.x, u.y), out_dir, color, pdf);
        pdf *= dir_prob / light_count;
        return color;
    }
    // "Should never be reached, hopefully." - out params are left undefined in GLSL; defined here.
    out_dir = V3(0); out_length = 0; pdf = 1.0f;
    return V3(0);
}

void correct_lobes_for_normal_map(vec3 sample_dir, vec3 geometric_normal, bsdf_lobes& lobes) {   // :291-300
    if (dot(geometric_normal, sample_dir) < 0) { lobes.diffuse = 0; lobes.dielectric_reflection = 0; lobes.metallic_reflection = 0; }
    else lobes.transmission = 0;
}

vec3 next_event_estimation(const pt_ctx& c, uvec4 rand_uint, const mat3& tbn, vec3 shading_view, const sampled_material& mat,
                           const pt_vertex_data& v, bsdf_lobes& lobes) {   // :302-344
    const oracle_scene& s = *c.s;
    bool any = (c.nee_point && !s.point_lights.empty()) || (c.nee_dir && !s.directional_lights.empty()) ||
               (c.nee_tri && !s.tri_lights.empty()) || (c.nee_env && s.environment_proj >= 0);
    if (!any) return V3(0);
    vec3 out_dir;
    float out_length = 0.0f, light_pdf;
    vec3 contrib = sample_explicit_light(c, rand_uint, v.pos, out_dir, out_length, light_pdf);
    vec3 shading_light = out_dir * tbn;
    lobes = bsdf_lobes{0, 0, 0, 0};
    float bsdf_pdf = material_bsdf_pdf(c.opt.bounce_mode, shading_light, shading_view, mat, lobes);
    correct_lobes_for_normal_map(out_dir, v.hard_normal, lobes);
    if (contrib.x > 0.0001f || contrib.y > 0.0001f || contrib.z > 0.0001f)
        contrib *= trace_shadow(s, v.pos, out_dir, c.opt.min_ray_dist, out_length, *c.tc);
    contrib = contrib / nee_mis_pdf(c, light_pdf, bsdf_pdf);
    return contrib;
}

float clamp_contribution_mul(const pt_ctx& c, vec3 contrib) {   // :356-365
    if (c.opt.indirect_clamping > 0.0f) {
        float m = rgb_to_luminance(contrib);
        if (m > c.opt.indirect_clamping) return c.opt.indirect_clamping / m;
    }
    return 1;
}

void evaluate_ray(const pt_ctx& c, local_sampler& lsampler, vec3 pos, vec3 view, vec4& diffuse, vec4& reflection,
                  pt_vertex_data& first_hit_vertex, sampled_material& first_hit_material) {   // :367-499
    const oracle_scene& s = *c.s;
    const uint MAX_BOUNCES = (uint)c.opt.max_bounces;
    vec3 attenuation = V3(1);
    diffuse = V4(0); reflection = V4(0);
    vec3 diffuse_rgb = V3(0), reflection_rgb = V3(0);
    float regularization = 1.0f;
    float bsdf_pdf = 0.0f;
    bsdf_lobes primary_lobes = {0, 0, 0, 1};
    hit_payload payload;
    { uvec4& sd = lsampler.rs_seed; payload.random_seed = pcg4d(sd).x; }
    for (uint bounce = 0; bounce < MAX_BOUNCES; ++bounce) {
        float hit_t;
        bool include_lights = !(c.opt.hide_lights && bounce == 0);
        trace_closest(s, pos, view, bounce == 0 ? 0.0f : c.opt.min_ray_dist, RAY_MAX_DIST, include_lights, 0, payload, hit_t, *c.tc);
        pt_vertex_data v;
        sampled_material mat;
        intersection_pdf nee_pdf;
        vec3 light;
        bool terminal = !get_intersection_info(c, payload, pos, view, v, nee_pdf, mat, light) || bounce == MAX_BOUNCES - 1;
        float mis_pdf = bsdf_mis_pdf(c, nee_pdf, bsdf_pdf);
        float mis_weight = 1.0f;
        if (bsdf_pdf != 0) {
            attenuation = attenuation / bsdf_pdf;
            mis_weight = bsdf_pdf / mis_pdf;
        }
        light = attenuation * mis_weight * (mat.emission + light);
        if (bounce != 0) light *= clamp_contribution_mul(c, light);
        add_demodulated_color(primary_lobes, light, diffuse_rgb, reflection_rgb);
        if (bounce == 0) {
            first_hit_vertex = v;
            first_hit_material = mat;
            first_hit_material.emission = light;
        }
        if (c.opt.regularization_gamma != 0.0f) {   // PATH_SPACE_REGULARIZATION
            if (bsdf_pdf != 0.0f) regularization *= max(1 - c.opt.regularization_gamma / powf(bsdf_pdf, 0.25f), 0.0f);
            mat.roughness = 1.0f - ((1.0f - mat.roughness) * regularization);
        }
        mat3 tbn = create_tangent_space(v.mapped_normal);
        vec3 shading_view = view_to_tangent_space(view, tbn);
        if (!terminal) {
            bsdf_lobes lobes = {0, 0, 0, 0};
            vec3 radiance = attenuation * next_event_estimation(
                c, generate_ray_sample_uint(lsampler, bounce * 2, c.opt.sampler, c.max_sobol_bounces), tbn, shading_view, mat, v, lobes);
            if (bounce != 0) {
                radiance *= modulate_bsdf(mat, lobes);
                radiance *= clamp_contribution_mul(c, radiance);
            } else primary_lobes = lobes;
            add_demodulated_color(primary_lobes, radiance, diffuse_rgb, reflection_rgb);
            if (bounce == 1) diffuse.w = reflection.w = 1.0f / length(v.pos - pos);
        }
        if (terminal) break;
        bsdf_lobes lobes = {0, 0, 0, 0};
        vec4 ray_sample = to_float(generate_ray_sample_uint(lsampler, bounce * 2 + 1, c.opt.sampler, c.max_sobol_bounces)) * INV_UINT32_MAX;
        material_bsdf_sample(c.opt.bounce_mode, ray_sample, shading_view, mat, view, lobes, bsdf_pdf);
        view = tbn * view;
        correct_lobes_for_normal_map(v.hard_normal, view, lobes);
        if (bounce != 0) attenuation *= modulate_bsdf(mat, lobes);
        else primary_lobes = lobes;
        pos = v.pos;
        if (c.opt.russian_roulette_delta > 0) {   // USE_RUSSIAN_ROULETTE; the visibility /= qi result is never used
            float qi = min(1.0f, 1.0f / c.opt.russian_roulette_delta);
            if (ray_sample.w > qi) break;
        }
        if (max(attenuation.x, max(attenuation.y, attenuation.z)) <= 0.0f) break;
    }
    diffuse.x = diffuse_rgb.x; diffuse.y = diffuse_rgb.y; diffuse.z = diffuse_rgb.z;
    reflection.x = reflection_rgb.x; reflection.y = reflection_rgb.y; reflection.z = reflection_rgb.z;
}

void get_world_camera_ray(const pt_ctx& c, const launch_ctx& L, ivec2 pixel, const camera_data& cam, local_sampler& lsampler,
                          vec3& origin, vec3& dir) {   // :504-533
    vec2 cam_offset = V2(0.0f);
    if (c.opt.film != 0) {   // control.antialiasing == 1
        if (c.opt.film == 1) {
            vec4 r = generate_uniform_random(lsampler);
            cam_offset = V2(r.x, r.y) * 2.0f - 1.0f;
        } else {
            vec4 r = generate_uniform_random(lsampler);
            cam_offset = sample_blackman_harris_concentric_disk(V2(r.x, r.y)) * 2.0f;
        }
        cam_offset = cam_offset * (2.0f * c.opt.film_radius);
    }
    vec2 dof_u = V2(0.5f);
    if (c.opt.depth_of_field) { vec4 r = generate_uniform_random(lsampler); dof_u = V2(r.x, r.y); }
    get_screen_camera_ray(L, pixel, cam, c.opt.projection, c.opt.depth_of_field != 0, cam_offset, dof_u, origin, dir);
}

// get_camera_projection (shader/camera.glsl:61-67 perspective / orthographic, :126-134 equirectangular)
vec3 get_camera_projection(const camera_data& cam, int projection, vec3 world_pos) {
    if (projection == 2) {
        vec3 t = V3(cam.view * V4(world_pos, 1.0f));
        float t_len = length(t);
        t = t / t_len;
        const float* raw = reinterpret_cast<const float*>(&cam);   // equirect layout: view, view_inverse, origin, fov (src/camera.cc:409-415)
        const float fov[2] = {raw[36], raw[37]};
        vec2 a = V2(atan2f(t.x, -t.z), asinf(t.y));
        return V3((a.x / fov[0]) * 0.5f + 0.5f, (a.y / fov[1]) * 0.5f + 0.5f, t_len);
    }
    vec4 projected_pos = cam.view_proj * V4(world_pos, 1.0f);
    return V3((projected_pos.x / projected_pos.w) * 0.5f + 0.5f, (projected_pos.y / projected_pos.w) * 0.5f + 0.5f, projected_pos.w);
}

// shader/pre_transform.comp:26-42 (dispatched per instance by src/scene_stage.cc:1685-1723)
void ensure_world_vertices(oracle_scene& s) {
    if (!s.world_spans.empty()) return;
    s.world_spans.resize(s.instances.size());
    size_t total = 0;
    for (size_t i = 0; i < s.instances.size(); ++i) total += s.spans[i].vertex_count;
    s.world_vertices.resize(total);
    uint offset = 0;
    for (size_t i = 0; i < s.instances.size(); ++i) {
        const instance& o = s.instances[i];
        const mesh_span& sp = s.spans[i];
        s.world_spans[i] = sp;
        s.world_spans[i].vertex_offset = offset;
        const mat3 mn = M3(o.model_normal);
        // determinant(mat3(o.model_normal)), cofactor expansion along the first column
        const float det = mn.c[0].x * (mn.c[1].y * mn.c[2].z - mn.c[2].y * mn.c[1].z)
                        - mn.c[1].x * (mn.c[0].y * mn.c[2].z - mn.c[2].y * mn.c[0].z)
                        + mn.c[2].x * (mn.c[0].y * mn.c[1].z - mn.c[1].y * mn.c[0].z);
        for (uint k = 0; k < sp.vertex_count; ++k) {
            vertex v = s.vertices[sp.vertex_offset + k];
            v.pos = V3(o.model * V4(v.pos, 1));
            v.normal = normalize(mn * v.normal);
            vec3 t = normalize(mn * V3(v.tangent));
            v.tangent = V4(t, v.tangent.w);
            if (det < 0) { v.normal = -v.normal; v.tangent = V4(-t.x, -t.y, -t.z, -v.tangent.w); }
            s.world_vertices[offset + k] = v;
        }
        offset += sp.vertex_count;
    }
}

// write_all_outputs (path_tracer.glsl:535-576) + accumulate_gbuffer_* (gbuffer.glsl:18-28,68-78,118-128)
void write_all_outputs(const pt_ctx& c, const launch_ctx& L, ivec3 wp, uint previous_samples, uint samples_accumulated, const oracle_pt_targets& T,
                       uint target_w, uint target_h, vec3 col, float alpha, vec4 sum_diffuse, vec4 sum_reflection, float diffuse_div, float reflection_div,
                       const pt_vertex_data& first_hit_vertex, const sampled_material& first_hit_material) {
    const oracle_scene& s = *c.s;
    const int spp = c.opt.samples_per_pass;
    if ((uint)wp.x >= target_w || (uint)wp.y >= target_h) return;
    const size_t pix = ((size_t)wp.z * target_h + wp.y) * target_w + wp.x;
    const uint prev_samples = samples_accumulated + previous_samples;
    if (prev_samples == 0) {   // only the first sample writes the gbuffer (path_tracer.glsl:549-563)
        if (T.albedo) { float* a = T.albedo + pix * 4; a[0] = first_hit_material.albedo.x; a[1] = first_hit_material.albedo.y; a[2] = first_hit_material.albedo.z; a[3] = first_hit_material.albedo.w; }
        if (T.material) {   // pack_gbuffer_material (gbuffer.glsl:256-260)
            float* m = T.material + pix * 4;
            float ior = first_hit_material.ior_out / first_hit_material.ior_in;
            m[0] = first_hit_material.metallic; m[1] = first_hit_material.roughness; m[2] = ior * 0.25f; m[3] = first_hit_material.transmittance;
        }
        if (T.normal) {     // octahedral_pack (math.glsl:480-485)
            vec3 n = first_hit_vertex.mapped_normal;
            n = n / (fabsf(n.x) + fabsf(n.y) + fabsf(n.z));
            vec2 o = n.z >= 0.0f ? V2(n.x, n.y)
                                 : V2((1 - fabsf(n.y)) * ((n.x >= 0.0f ? 1.0f : 0.0f) * 2 - 1), (1 - fabsf(n.x)) * ((n.y >= 0.0f ? 1.0f : 0.0f) * 2 - 1));
            T.normal[pix * 2] = o.x; T.normal[pix * 2 + 1] = o.y;
        }
        if (T.pos) { float* q = T.pos + pix * 4; q[0] = first_hit_vertex.pos.x; q[1] = first_hit_vertex.pos.y; q[2] = first_hit_vertex.pos.z; q[3] = 0; }
        if (T.instance_id) T.instance_id[pix] = first_hit_vertex.instance_id;
        if (T.screen_motion) {   // write_gbuffer_screen_motion(get_camera_projection(get_prev_camera(), prev_pos)): rg32f keeps xy
            vec3 m = get_camera_projection(s.prev_cameras[L.viewport], c.opt.projection, first_hit_vertex.prev_pos);
            T.screen_motion[pix * 2] = m.x; T.screen_motion[pix * 2 + 1] = m.y;
        }
    }
    auto accumulate = [&](float* target, vec4 value) {   // accumulate_gbuffer_{color,diffuse,reflection} (gbuffer.glsl:18-28,68-78,118-128)
        if (!target) return;
        float* px = target + pix * 4;
        if (prev_samples != 0) {
            vec4 prev = V4(px[0], px[1], px[2], px[3]);
            uint total = (uint)spp + prev_samples;
            value = mix(value, prev, (float)prev_samples / (float)total);
        }
        px[0] = value.x; px[1] = value.y; px[2] = value.z; px[3] = value.w;
    };
    accumulate(T.color, V4(col, alpha));
    accumulate(T.diffuse, sum_diffuse / diffuse_div);
    accumulate(T.reflection, sum_reflection / reflection_div);
}

// path_tracer.rgen:77-127 + write_all_outputs (path_tracer.glsl:535-576) + accumulate_gbuffer_color (gbuffer.glsl:18-28)
void pt_invocation(const pt_ctx& c, const launch_ctx& L, uint previous_samples, uint samples_accumulated, const oracle_pt_targets& T,
                   uint target_w, uint target_h) {
    const oracle_scene& s = *c.s;
    ivec2 pixel;
    ivec3 wp;
    if (!get_pixel_pos(L, pixel) || !get_write_pixel_pos(L, wp)) return;
    const camera_data& cam = s.cameras[L.viewport];
    pt_vertex_data first_hit_vertex{};
    sampled_material first_hit_material{};
    vec3 sum_color = V3(0);
    vec4 sum_diffuse = V4(0), sum_reflection = V4(0);
    const int spp = c.opt.samples_per_pass;
    for (int i = 0; i < spp; ++i) {
        local_sampler lsampler = init_local_sampler(uvec4{(uint)pixel.x, (uint)pixel.y, L.viewport, s.shard_sample_base + s.shard_sample_stride * (previous_samples + (uint)i)},
                                                    c.sample_counter, c.rng_seed, c.opt.sampler);
        vec3 origin, dir;
        get_world_camera_ray(c, L, pixel, cam, lsampler, origin, dir);
        vec4 diffuse, reflection;
        evaluate_ray(c, lsampler, origin, dir, diffuse, reflection, first_hit_vertex, first_hit_material);
        vec4 old_albedo = first_hit_material.albedo;
        if (c.opt.use_white_albedo_on_first_bounce) { first_hit_material.albedo.x = first_hit_material.albedo.y = first_hit_material.albedo.z = 1; }
        sum_color += first_hit_material.emission + modulate_color(first_hit_material, V3(diffuse), V3(reflection));
        sum_diffuse = sum_diffuse + diffuse;
        sum_reflection = sum_reflection + reflection;
        first_hit_material.albedo = old_albedo;
    }
    vec3 col = sum_color / (float)spp;
    const float alpha = c.opt.transparent_background ? first_hit_material.albedo.w : 1.0f;
    write_all_outputs(c, L, wp, previous_samples, samples_accumulated, T, target_w, target_h, col, alpha, sum_diffuse, sum_reflection, (float)spp, (float)spp,
                      first_hit_vertex, first_hit_material);
}

// shader/direct.rgen:57-132 (direct_stage): one primary ray with lights hidden, SAMPLES_PER_PASS light samples at its hit
void direct_invocation(const pt_ctx& c, const launch_ctx& L, uint previous_samples, uint samples_accumulated, const oracle_pt_targets& T,
                       uint target_w, uint target_h) {
    const oracle_scene& s = *c.s;
    ivec2 pixel;
    ivec3 wp;
    if (!get_pixel_pos(L, pixel) || !get_write_pixel_pos(L, wp)) return;
    const camera_data& cam = s.cameras[L.viewport];
    const int spp = c.opt.samples_per_pass;
    local_sampler lsampler = init_local_sampler(uvec4{(uint)pixel.x, (uint)pixel.y, L.viewport, c.s->shard_sample_base + c.s->shard_sample_stride * previous_samples}, c.sample_counter, c.rng_seed, c.opt.sampler);
    vec3 origin, dir;
    get_world_camera_ray(c, L, pixel, cam, lsampler, origin, dir);
    // evaluate_direct_ray
    pt_vertex_data first_hit_vertex{};
    sampled_material first_hit_material{};
    hit_payload payload;
    { uvec4& sd = lsampler.rs_seed; payload.random_seed = pcg4d(sd).x; }
    vec3 color = V3(0);
    vec4 diffuse = V4(0), reflection = V4(0);
    float hit_t;
    trace_closest(s, origin, dir, 0.0f, RAY_MAX_DIST, false /* mask 0xFF ^ 0x02 */, 0, payload, hit_t, *c.tc);
    intersection_pdf nee_pdf;
    vec3 light;
    const bool terminal = !get_intersection_info(c, payload, origin, dir, first_hit_vertex, nee_pdf, first_hit_material, light);
    color += (light + first_hit_material.emission) * (float)spp;
    if (!terminal) {
        mat3 tbn = create_tangent_space(first_hit_vertex.mapped_normal);
        vec3 shading_view = -dir * tbn;
        if (shading_view.z < 0.00001f) shading_view = V3(shading_view.x, shading_view.y, std::max(shading_view.z, 0.00001f));
        shading_view = normalize(shading_view);
        vec3 diffuse_rgb = V3(0), reflection_rgb = V3(0);
        for (int i = 0; i < spp; ++i) {
            bsdf_lobes lobes = {0, 0, 0, 0};
            vec3 radiance = next_event_estimation(c, generate_ray_sample_uint(lsampler, (uint)i, c.opt.sampler, c.max_sobol_bounces), tbn, shading_view,
                                                  first_hit_material, first_hit_vertex, lobes);
            color += radiance * modulate_bsdf(first_hit_material, lobes);
            add_demodulated_color(lobes, radiance, diffuse_rgb, reflection_rgb);
            diffuse.w = reflection.w = 1.0f / length(first_hit_vertex.pos - origin);
        }
        diffuse = V4(diffuse_rgb, diffuse.w); reflection = V4(reflection_rgb, reflection.w);
    }
    // main(): color /= SAMPLES_PER_PASS; diffuse /= SAMPLES_PER_PASS; (reflection is not divided)
    const float alpha = c.opt.transparent_background ? first_hit_material.albedo.w : 1.0f;
    write_all_outputs(c, L, wp, previous_samples, samples_accumulated, T, target_w, target_h, color / (float)spp, alpha, diffuse, reflection, (float)spp, 1.0f,
                      first_hit_vertex, first_hit_material);
}

uvec2 get_ray_count(const oracle_distribution& d) {   // src/distribution_strategy.cc:33-61
    if (d.strategy == 0) return {d.size_x, d.size_y};
    if (d.strategy == 1) return {d.size_x, (d.size_y - d.index + d.count - 1) / d.count};
    return {d.count, 1};
}
uint calculate_shuffled_strips_b(uint sx, uint sy) {   // src/distribution_strategy.cc:62-69
    uint n = sx * sy;
    uint b = 31;
    while ((n >> b) < 128 && b > 0) b--;
    return b;
}

void flush_counters(oracle_scene& s, const thread_counters& tc) {
    s.counters.closest += tc.closest; s.counters.shadow += tc.shadow; s.counters.nodes += tc.nodes;
    s.counters.tris += tc.tris; s.counters.alpha += tc.alpha; s.counters.surface += tc.surface;
}

}  // namespace

// ============================================================================
// C API
// ============================================================================
extern "C" {

oracle_scene* oracle_scene_create(const oracle_scene_desc* d) {
    oracle_scene* s = new oracle_scene();
    s->instances.assign((const instance*)d->instances, (const instance*)d->instances + d->instance_count);
    s->spans.assign((const mesh_span*)d->spans, (const mesh_span*)d->spans + d->instance_count);
    s->vertices.assign((const vertex*)d->vertices, (const vertex*)d->vertices + d->vertex_count);
    s->indices.assign(d->indices, d->indices + d->index_count);
    s->point_lights.assign((const point_light*)d->point_lights, (const point_light*)d->point_lights + d->point_light_count);
    s->directional_lights.assign((const directional_light*)d->directional_lights,
                                 (const directional_light*)d->directional_lights + d->directional_light_count);
    s->tex_infos.assign((const texture_info*)d->texture_infos, (const texture_info*)d->texture_infos + d->texture_count);
    size_t texel_count = 0;
    for (auto& ti : s->tex_infos) texel_count = std::max(texel_count, (size_t)ti.texel_offset + (size_t)ti.width * ti.height * (ti.format == 1 ? 2u : 1u));
    s->texels.assign(d->texels, d->texels + texel_count * 4);
    // scene_metadata (src/scene_stage.cc:1341-1352): the factor is the environment map's; without a map it is zero, proj -1
    s->environment_factor = V4(0.0f);
    if (d->envmap && d->envmap_width && d->envmap_height) {
        s->environment_factor = V4(d->environment_factor[0], d->environment_factor[1], d->environment_factor[2], d->environment_factor[3]);
        s->env_w = d->envmap_width; s->env_h = d->envmap_height;
        size_t n = (size_t)s->env_w * s->env_h;
        s->envmap.assign((const vec4*)d->envmap, (const vec4*)d->envmap + n);
        s->alias_table.assign((const alias_table_entry*)d->alias_table, (const alias_table_entry*)d->alias_table + n);
        s->environment_proj = 0;
    } else {
        // no environment map: scene_stage writes factor 0, proj -1 (src/scene_stage.cc:1340-1350)
        s->environment_proj = -1;
    }
    s->cameras.assign((const camera_data*)d->cameras, (const camera_data*)d->cameras + d->camera_count);
    s->prev_cameras = s->cameras;
    s->non_opaque.assign(d->non_opaque, d->non_opaque + d->instance_count);
    build_scene_accel(*s);
    if (d->gather_emissive_triangles) extract_tri_lights(*s);
    return s;
}
void oracle_scene_destroy(oracle_scene* s) { delete s; }
int oracle_scene_set_previous_cameras(oracle_scene* s, const void* camera_data_array, uint32_t count) {
    if (count != s->cameras.size()) return 1;
    s->prev_cameras.assign((const camera_data*)camera_data_array, (const camera_data*)camera_data_array + count);
    return 0;
}
uint32_t oracle_scene_tri_light_count(const oracle_scene* s) { return (uint32_t)s->tri_lights.size(); }
// shader/skinning.comp:44-71 over one mesh.  GLSL's inverse() is the driver's; restated as the cofactor expansion over
// 2x2 sub-determinants.  Directions use w = 0, so only the upper-left 3x3 of inverse(skin_mat) reaches them and the
// fourth column of the transpose contributes +0.
void oracle_skin_vertices(const void* source, const void* skins_in, uint32_t vertex_count, const float* joint_transforms, uint32_t joint_count, void* destination) {
    struct skin { uint joints[4]; float weights[4]; };
    const vertex* src_v = (const vertex*)source;
    const skin* skins = (const skin*)skins_in;
    const mat4* joints = (const mat4*)joint_transforms;
    vertex* dst_v = (vertex*)destination;
    for (uint32_t i = 0; i < vertex_count; ++i) {
        const skin s = skins[i];
        mat4 m;
        for (int c = 0; c < 4; ++c) {
            vec4 col = joints[s.joints[0] < joint_count ? s.joints[0] : 0].c[c] * s.weights[0];
            for (int k = 1; k < 4; ++k) col = col + joints[s.joints[k] < joint_count ? s.joints[k] : 0].c[c] * s.weights[k];
            m.c[c] = col;
        }
        auto a = [&](int r, int c) { const vec4& v = m.c[c]; return r == 0 ? v.x : r == 1 ? v.y : r == 2 ? v.z : v.w; };
        const float s0 = a(0,0) * a(1,1) - a(1,0) * a(0,1), s1 = a(0,0) * a(1,2) - a(1,0) * a(0,2), s2 = a(0,0) * a(1,3) - a(1,0) * a(0,3);
        const float s3 = a(0,1) * a(1,2) - a(1,1) * a(0,2), s4 = a(0,1) * a(1,3) - a(1,1) * a(0,3), s5 = a(0,2) * a(1,3) - a(1,2) * a(0,3);
        const float c5 = a(2,2) * a(3,3) - a(3,2) * a(2,3), c4 = a(2,1) * a(3,3) - a(3,1) * a(2,3), c3 = a(2,1) * a(3,2) - a(3,1) * a(2,2);
        const float c2 = a(2,0) * a(3,3) - a(3,0) * a(2,3), c1 = a(2,0) * a(3,2) - a(3,0) * a(2,2), c0 = a(2,0) * a(3,1) - a(3,0) * a(2,1);
        const float det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
        float inv[3][3];
        inv[0][0] = ( a(1,1) * c5 - a(1,2) * c4 + a(1,3) * c3) / det;
        inv[0][1] = (-a(0,1) * c5 + a(0,2) * c4 - a(0,3) * c3) / det;
        inv[0][2] = ( a(3,1) * s5 - a(3,2) * s4 + a(3,3) * s3) / det;
        inv[1][0] = (-a(1,0) * c5 + a(1,2) * c2 - a(1,3) * c1) / det;
        inv[1][1] = ( a(0,0) * c5 - a(0,2) * c2 + a(0,3) * c1) / det;
        inv[1][2] = (-a(3,0) * s5 + a(3,2) * s2 - a(3,3) * s1) / det;
        inv[2][0] = ( a(1,0) * c4 - a(1,1) * c2 + a(1,3) * c0) / det;
        inv[2][1] = (-a(0,0) * c4 + a(0,1) * c2 - a(0,3) * c0) / det;
        inv[2][2] = ( a(3,0) * s4 - a(3,1) * s2 + a(3,3) * s0) / det;
        const mat3 it = M3(V3(inv[0][0], inv[0][1], inv[0][2]), V3(inv[1][0], inv[1][1], inv[1][2]), V3(inv[2][0], inv[2][1], inv[2][2]));  // transpose(inverse)
        const vertex src = src_v[i];
        vertex dst = src;
        dst.pos = V3(m * V4(src.pos.x, src.pos.y, src.pos.z, 1.0f));
        dst.normal = normalize(it * src.normal + V3(0.0f, 0.0f, 0.0f));
        const vec3 t = normalize(it * V3(src.tangent) + V3(0.0f, 0.0f, 0.0f));
        dst.tangent = V4(t.x, t.y, t.z, src.tangent.w);
        dst_v[i] = dst;
    }
}

int oracle_scene_set_shard(oracle_scene* s, uint32_t viewport_base, uint32_t viewport_stride, uint32_t sample_base, uint32_t sample_stride) {
    if (viewport_stride == 0 || sample_stride == 0 || sample_base >= sample_stride) return 1;
    s->shard_vp_base = viewport_base; s->shard_vp_stride = viewport_stride; s->shard_sample_base = sample_base; s->shard_sample_stride = sample_stride;
    return 0;
}
void oracle_scene_get_tri_lights(const oracle_scene* s, void* out) { memcpy(out, s->tri_lights.data(), s->tri_lights.size() * sizeof(tri_light)); }

int oracle_pt_render(oracle_scene* s, const oracle_pt_options* opt, const oracle_distribution* dist_in, uint32_t viewport_count,
                     uint32_t frame_counter, uint32_t samples_accumulated, float* color, uint32_t target_w, uint32_t target_h,
                     int threads) {
    oracle_pt_targets T{};
    T.color = color;
    return oracle_pt_render_targets(s, opt, dist_in, viewport_count, frame_counter, samples_accumulated, &T, target_w, target_h, threads);
}

static int render_targets_impl(oracle_scene* s, const oracle_pt_options* opt, const oracle_distribution* dist_in, uint32_t viewport_count,
                               uint32_t frame_counter, uint32_t samples_accumulated, const oracle_pt_targets* targets, uint32_t target_w,
                               uint32_t target_h, int threads, bool direct);

int oracle_pt_render_targets(oracle_scene* s, const oracle_pt_options* opt, const oracle_distribution* dist_in, uint32_t viewport_count,
                             uint32_t frame_counter, uint32_t samples_accumulated, const oracle_pt_targets* targets, uint32_t target_w,
                             uint32_t target_h, int threads) {
    return render_targets_impl(s, opt, dist_in, viewport_count, frame_counter, samples_accumulated, targets, target_w, target_h, threads, false);
}

/* direct_stage (src/direct_stage.cc, shader/direct.rgen): no MIS define is set for it, so nee_mis_pdf is the light pdf */
int oracle_direct_render_targets(oracle_scene* s, const oracle_pt_options* opt, const oracle_distribution* dist_in, uint32_t viewport_count,
                                 uint32_t frame_counter, uint32_t samples_accumulated, const oracle_pt_targets* targets, uint32_t target_w,
                                 uint32_t target_h, int threads) {
    oracle_pt_options o = *opt;
    o.mis_mode = 0;
    return render_targets_impl(s, &o, dist_in, viewport_count, frame_counter, samples_accumulated, targets, target_w, target_h, threads, true);
}

static int render_targets_impl(oracle_scene* s, const oracle_pt_options* opt, const oracle_distribution* dist_in, uint32_t viewport_count,
                               uint32_t frame_counter, uint32_t samples_accumulated, const oracle_pt_targets* targets, uint32_t target_w,
                               uint32_t target_h, int threads, bool direct) {
    if (viewport_count == 0 || (uint64_t)s->shard_vp_base + (uint64_t)(viewport_count - 1) * s->shard_vp_stride >= s->cameras.size()) return 1;
    if (opt->pre_transformed_vertices) ensure_world_vertices(*s);
    pt_ctx c;
    c.s = s; c.opt = *opt;
    c.max_sobol_bounces = (uint)(opt->max_bounces > 8 ? 8 : opt->max_bounces);   // sobol_lookup_table.glsl:4-14
    c.nee_point = opt->nee_point > 0; c.nee_dir = opt->nee_directional > 0;
    c.nee_env = opt->nee_envmap > 0; c.nee_tri = opt->nee_triangles > 0;
    c.sample_counter = frame_counter * (uint)opt->samples_per_pixel * s->shard_sample_stride;   // rt_stage.cc:81, rt_camera_stage.cc:59
    uint seed = opt->rng_seed;
    c.rng_seed = seed != 0 ? pcg(seed) : 0;                                       // rt_stage.cc:82
    oracle_distribution dist = *dist_in;
    if (dist.strategy == 2) dist.count = dist_in->count;   // ray count uses the pixel count; b is derived below
    uvec2 rays = get_ray_count(dist);
    launch_ctx base;
    base.dist = dist;
    if (dist.strategy == 2) base.dist.count = calculate_shuffled_strips_b(dist.size_x, dist.size_y);
    base.launch_size = rays;
    const int passes = opt->samples_per_pixel / opt->samples_per_pass;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    const int tiles_x = (int)((rays.x + 15u) / 16u), n_tiles = tiles_x * (int)((rays.y + 15u) / 16u);
    int n_threads = 1;
#ifdef _OPENMP
    n_threads = std::max(omp_get_max_threads(), 1);
#endif
    const int tiles_per_grab = std::max(1, std::min(16, n_tiles / (n_threads * 16)));
    const int n_grabs = (n_tiles + tiles_per_grab - 1) / tiles_per_grab;
    for (int pass = 0; pass < passes; ++pass) {
        uint previous_samples = (uint)pass * (uint)opt->samples_per_pass;
        for (uint z = 0; z < viewport_count; ++z) {
#pragma omp parallel
            {
                thread_counters tc;
                pt_ctx lc = c;
                lc.tc = &tc;
                // 16 x 16 pixel tiles handed out sixteen at a time; launch ids are independent, so the order is
                // scheduling only.  Sixteen tiles per grab would leave a 256-thread host two grabs per thread at 1080p, so the grab
                // shrinks until every thread can expect at least sixteen of them.
#pragma omp for schedule(dynamic, 1)
                for (int g = 0; g < n_grabs; ++g) {
                    for (int t = g * tiles_per_grab; t < std::min((g + 1) * tiles_per_grab, n_tiles); ++t) {
                        const uint x0 = (uint)(t % tiles_x) * 16u, y0 = (uint)(t / tiles_x) * 16u;
                        for (uint y = y0; y < std::min(y0 + 16u, rays.y); ++y) {
                            for (uint x = x0; x < std::min(x0 + 16u, rays.x); ++x) {
                                launch_ctx L = base;
                                L.launch_id = {x, y, z};
                                L.viewport = s->shard_vp_base + z * s->shard_vp_stride;
                                if (direct) direct_invocation(lc, L, previous_samples, samples_accumulated, *targets, target_w, target_h);
                                else pt_invocation(lc, L, previous_samples, samples_accumulated, *targets, target_w, target_h);
                            }
                        }
                    }
                }
#pragma omp critical
                flush_counters(*s, tc);
            }
        }
    }
    return 0;
}

int oracle_feature_render(oracle_scene* s, int feature, const oracle_distribution* dist_in, int projection, uint32_t viewport,
                          float min_ray_dist, const float default_value[4], float* color, uint32_t target_w, uint32_t target_h,
                          int threads) {
    if (viewport >= s->cameras.size()) return 1;
    pt_ctx c;
    c.s = s;
    memset(&c.opt, 0, sizeof(c.opt));
    c.opt.projection = projection;
    c.nee_point = c.nee_dir = c.nee_env = c.nee_tri = false;
    oracle_distribution dist = *dist_in;
    uvec2 rays = get_ray_count(dist);
    launch_ctx base;
    base.dist = dist;
    if (dist.strategy == 2) base.dist.count = calculate_shuffled_strips_b(dist.size_x, dist.size_y);
    base.launch_size = rays;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        thread_counters tc;
        pt_ctx lc = c;
        lc.tc = &tc;
#pragma omp for schedule(dynamic, 4)
        for (int y = 0; y < (int)rays.y; ++y) {
            for (uint x = 0; x < rays.x; ++x) {
                launch_ctx L = base;
                L.launch_id = {x, (uint)y, viewport};
                ivec2 pixel; ivec3 wp;
                if (!get_pixel_pos(L, pixel) || !get_write_pixel_pos(L, wp)) continue;
                const camera_data& cam = s->cameras[viewport];
                vec3 origin, dir;
                // rt_feature.rgen:21-45: ray from cam.origin with tmin = min_ray_dist
                get_screen_camera_ray(L, pixel, cam, projection, false, V2(0), V2(0.5f), origin, dir);
                vec3 ray_origin = projection == 2 ? origin : V3(cam.origin);
                hit_payload payload;
                payload.random_seed = 0;
                float hit_t;
                trace_closest(*s, ray_origin, dir, min_ray_dist, RAY_MAX_DIST, false, 1, payload, hit_t, tc);
                vec4 data = V4(default_value[0], default_value[1], default_value[2], default_value[3]);
                if (payload.instance_id >= 0) {   // rt_feature.rchit:16-27 with FEATURE of src/feature_stage.cc:33-65
                    float pdf;
                    vertex_data v = get_interpolated_vertex(lc, dir, payload.barycentrics, payload.instance_id, payload.primitive_id, ray_origin, pdf);
                    sampled_material mat = sample_material(*s, payload.instance_id, v);
                    switch (feature) {
                        default:
                        case 0: data = mat.albedo; break;
                        case 1: data = V4(v.mapped_normal, 1); break;
                        case 2: data = V4(V3(cam.view * V4(v.mapped_normal, 0)), 1); break;
                        case 3: data = V4(v.pos, 1); break;
                        case 4: data = cam.view * V4(v.pos, 1); break;
                        case 5: data = V4(hit_t, hit_t, hit_t, 1); break;
                        case 6: data = V4(v.pos - v.prev_pos, 1); break;
                        case 7: data = V4(V3(cam.view * V4(v.pos, 1) - s->prev_cameras[viewport].view * V4(v.prev_pos, 1)), 1); break;
                        case 8: data = V4(get_camera_projection(s->prev_cameras[viewport], projection, v.prev_pos), 1); break;
                        case 9: data = V4((float)payload.instance_id, (float)payload.primitive_id, 0, 1); break;
                    }
                }
                if ((uint)wp.x >= target_w || (uint)wp.y >= target_h) continue;
                float* px = color + (((size_t)0 * target_h + wp.y) * target_w + wp.x) * 4;
                px[0] = data.x; px[1] = data.y; px[2] = data.z; px[3] = data.w;
            }
        }
#pragma omp critical
        flush_counters(*s, tc);
    }
    return 0;
}

void oracle_trace_closest(oracle_scene* s, uint32_t n, const float* rays, const uint32_t* seeds, int include_lights,
                          oracle_hit* out, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        thread_counters tc;
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < (int64_t)n; ++i) {
            const float* r = rays + i * 8;
            hit_payload p;
            p.random_seed = seeds ? seeds[i] : 0;
            float t;
            trace_closest(*s, V3(r[0], r[1], r[2]), V3(r[4], r[5], r[6]), r[3], r[7], include_lights != 0, seeds ? 0 : 1, p, t, tc);
            out[i].instance_id = p.instance_id; out[i].primitive_id = p.primitive_id;
            out[i].bary_u = p.barycentrics.x; out[i].bary_v = p.barycentrics.y; out[i].t = t;
        }
#pragma omp critical
        flush_counters(*s, tc);
    }
}

void oracle_trace_shadow(oracle_scene* s, uint32_t n, const float* rays, float* visibility, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        thread_counters tc;
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < (int64_t)n; ++i) {
            const float* r = rays + i * 8;
            visibility[i] = trace_shadow(*s, V3(r[0], r[1], r[2]), V3(r[4], r[5], r[6]), r[3], r[7], tc);
        }
#pragma omp critical
        flush_counters(*s, tc);
    }
}

// tonemap.glsl:35-55 + tonemap_{gamma,filmic,reinhard,reinhard_luminance}.comp; op as tonemap_stage.hh:17-24
void oracle_tonemap(const float* in, float* out, uint32_t pixel_count, int op, float exposure, float gamma) {
    if (op == 0) gamma = 1.0f;   // tonemap_stage.cc:159
    for (uint32_t i = 0; i < pixel_count; ++i) {
        vec4 col = V4(in[4 * i], in[4 * i + 1], in[4 * i + 2], in[4 * i + 3]);
        vec3 c;
        if (op == 0 || op == 1) c = V3(col) * exposure;
        else if (op == 2) {
            c = clamp(V3(col) * exposure, V3(0), V3(1000));
            c = max(V3(0.0f), c - 0.004f);
            c = pow3((c * (6.2f * c + 0.5f)) / (c * (6.2f * c + 1.7f) + 0.06f), V3(2.2f));
        } else if (op == 3) {
            c = clamp(V3(col) * exposure, V3(0), V3(1000));
            c = c / (V3(1.0f) + c);
        } else {
            c = clamp(V3(col) * exposure, V3(0), V3(1000));
            float lum = rgb_to_luminance(c);
            float new_lum = lum / (1.0f + lum);
            c = c / max(lum, 1e-4f) * new_lum;
        }
        if (gamma != 1.0f) c = pow3(c, V3(1.0f / gamma));
        out[4 * i] = c.x; out[4 * i + 1] = c.y; out[4 * i + 2] = c.z; out[4 * i + 3] = col.w;
    }
}

void oracle_get_counters(oracle_scene* s, oracle_counters* o) {
    o->closest_rays = s->counters.closest; o->shadow_rays = s->counters.shadow; o->node_visits = s->counters.nodes;
    o->tri_tests = s->counters.tris; o->alpha_tests = s->counters.alpha; o->surface_hits = s->counters.surface;
}
void oracle_reset_counters(oracle_scene* s) {
    s->counters.closest = 0; s->counters.shadow = 0; s->counters.nodes = 0; s->counters.tris = 0; s->counters.alpha = 0; s->counters.surface = 0;
}

// ---- known-answer hooks
uint32_t oracle_pcg(uint32_t* seed) { return pcg(*seed); }
void oracle_pcg2d(uint32_t seed[2], uint32_t out[2]) { uvec2 s = {seed[0], seed[1]}; uvec2 r = pcg2d(s); seed[0] = s.x; seed[1] = s.y; out[0] = r.x; out[1] = r.y; }
void oracle_pcg4d(uint32_t seed[4], uint32_t out[4]) {
    uvec4 s = {seed[0], seed[1], seed[2], seed[3]};
    uvec4 r = pcg4d(s);
    seed[0] = s.x; seed[1] = s.y; seed[2] = s.z; seed[3] = s.w;
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void oracle_init_random_sampler(const uint32_t coord[4], uint32_t o[4]) {
    uvec4 r = init_random_sampler(uvec4{coord[0], coord[1], coord[2], coord[3]});
    o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
}
void oracle_generate_sobol_sample(uint32_t index, uint32_t bounce, uint32_t msb, uint32_t o[4]) {
    uvec4 r = generate_sobol_sample(index, bounce, msb); o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
}
void oracle_owen_scramble_2d(const uint32_t x[4], const uint32_t seed[4], uint32_t o[4]) {
    uvec4 r = owen_scramble_2d(uvec4{x[0], x[1], x[2], x[3]}, uvec4{seed[0], seed[1], seed[2], seed[3]});
    o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
}
uint32_t oracle_owen_scramble_4d(uint32_t x, uint32_t seed) { return owen_scramble_4d(x, seed); }
uint32_t oracle_owen_scramble_8d(uint32_t x, uint32_t seed) { return owen_scramble_8d(x, seed); }
uint32_t oracle_get_permutation_n(int n, uint32_t permutation, uint32_t dimension) { return get_permutation_n(n, permutation, dimension); }
uint32_t oracle_morton_2d(uint32_t x, uint32_t y) { return morton_2d(x, y); }
uint32_t oracle_morton_3d(uint32_t x, uint32_t y, uint32_t z) { return morton_3d(x, y, z); }
void oracle_ray_sample_uint(int sampler, int max_bounces, const uint32_t coord[4], uint32_t sample_counter, uint32_t rng_seed_raw,
                            uint32_t bounce_index, uint32_t o[4]) {
    uint seed = rng_seed_raw;
    uint rs = seed != 0 ? pcg(seed) : 0;
    local_sampler ls = init_local_sampler(uvec4{coord[0], coord[1], coord[2], coord[3]}, sample_counter, rs, sampler);
    uvec4 r = generate_ray_sample_uint(ls, bounce_index, sampler, (uint)(max_bounces > 8 ? 8 : max_bounces));
    o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
}
uint32_t oracle_rgb_to_r9g9b9e5(const float rgb[3]) { return rgb_to_r9g9b9e5(V3(rgb[0], rgb[1], rgb[2])); }
void oracle_r9g9b9e5_to_rgb(uint32_t v, float rgb[3]) { vec3 c = r9g9b9e5_to_rgb(v); rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z; }
uint32_t oracle_pack_half2x16(float x, float y) { return packHalf2x16(V2(x, y)); }
uint32_t oracle_permute_region_id(uint32_t i, uint32_t sx, uint32_t sy, uint32_t b) { return permute_region_id(i, sx, sy, b); }
void oracle_camera_ray(const void* cam, int projection, float px, float py, float sw, float sh, float dof_u, float dof_v, int dof,
                       float origin[3], float dir[3]) {
    vec3 o, d;
    get_camera_ray(*(const camera_data*)cam, projection, dof != 0, V2(px, py), V2(sw, sh), V2(dof_u, dof_v), o, d);
    origin[0] = o.x; origin[1] = o.y; origin[2] = o.z; dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}
void oracle_sample_cone(float u0, float u1, const float dir[3], float cos_theta_min, float out[3]) {
    vec3 r = sample_cone(V2(u0, u1), V3(dir[0], dir[1], dir[2]), cos_theta_min); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void oracle_sample_spherical_triangle(float u0, float u1, const float A[3], const float B[3], const float C[3], float out_dir[3], float* pdf) {
    vec3 r = sample_spherical_triangle(V2(u0, u1), V3(A[0], A[1], A[2]), V3(B[0], B[1], B[2]), V3(C[0], C[1], C[2]), *pdf);
    out_dir[0] = r.x; out_dir[1] = r.y; out_dir[2] = r.z;
}
void oracle_ggx_vndf_sample(const float view[3], float roughness, float u1, float u2, float out[3]) {
    vec3 r = ggx_vndf_sample(V3(view[0], view[1], view[2]), roughness, u1, u2); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
static sampled_material material_from_floats(const float m[10]) {
    sampled_material mat{};
    mat.albedo = V4(m[0], m[1], m[2], m[3]); mat.metallic = m[4]; mat.roughness = m[5]; mat.transmittance = m[6];
    mat.ior_in = m[7]; mat.ior_out = m[8]; mat.f0 = m[9];
    return mat;
}
void oracle_ggx_bsdf_sample(const float u[4], const float view[3], const float material[10], float out_dir[3], float lobes[4], float* pdf) {
    sampled_material mat = material_from_floats(material);
    bsdf_lobes l = {0, 0, 0, 0};
    vec3 od;
    ggx_bsdf_sample(V4(u[0], u[1], u[2], u[3]), V3(view[0], view[1], view[2]), mat, od, l, *pdf);
    out_dir[0] = od.x; out_dir[1] = od.y; out_dir[2] = od.z;
    lobes[0] = l.transmission; lobes[1] = l.diffuse; lobes[2] = l.dielectric_reflection; lobes[3] = l.metallic_reflection;
}
float oracle_ggx_bsdf_pdf(const float out_dir[3], const float view[3], const float material[10], float lobes[4]) {
    sampled_material mat = material_from_floats(material);
    bsdf_lobes l = {0, 0, 0, 0};
    float pdf = ggx_bsdf_pdf(V3(out_dir[0], out_dir[1], out_dir[2]), V3(view[0], view[1], view[2]), mat, l);
    lobes[0] = l.transmission; lobes[1] = l.diffuse; lobes[2] = l.dielectric_reflection; lobes[3] = l.metallic_reflection;
    return pdf;
}
float oracle_sample_blackman_harris(float u) { return sample_blackman_harris(u); }

}  // extern "C"
