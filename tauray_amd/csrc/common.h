// Shared device/host definitions for the trhip kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <string.h>
#include <cstring>

#include "../../include/trhip.h"

#define TR_DEV __device__ __forceinline__
#define TR_HD __host__ __device__ __forceinline__

namespace tr {

typedef uint32_t uint;

// ---------------------------------------------------------------------------
// small vector layer (explicit evaluation order; see DESIGN.md "fp contract")
// ---------------------------------------------------------------------------
struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };
struct u4 { uint x, y, z, w; };
struct u2 { uint x, y; };

TR_HD f2 F2(float a, float b) { return {a, b}; }
TR_HD f2 F2(float a) { return {a, a}; }
TR_HD f3 F3(float a, float b, float c) { return {a, b, c}; }
TR_HD f3 F3(float a) { return {a, a, a}; }
TR_HD f3 F3(const f4& v) { return {v.x, v.y, v.z}; }
TR_HD f4 F4(float a, float b, float c, float d) { return {a, b, c, d}; }
TR_HD f4 F4(const f3& v, float w) { return {v.x, v.y, v.z, w}; }
TR_HD f4 F4(float a) { return {a, a, a, a}; }

TR_HD f2 operator+(f2 a, f2 b) { return {a.x + b.x, a.y + b.y}; }
TR_HD f2 operator-(f2 a, f2 b) { return {a.x - b.x, a.y - b.y}; }
TR_HD f2 operator*(f2 a, f2 b) { return {a.x * b.x, a.y * b.y}; }
TR_HD f2 operator/(f2 a, f2 b) { return {a.x / b.x, a.y / b.y}; }
TR_HD f2 operator*(f2 a, float s) { return {a.x * s, a.y * s}; }
TR_HD f2 operator*(float s, f2 a) { return {a.x * s, a.y * s}; }
TR_HD f2 operator/(f2 a, float s) { return {a.x / s, a.y / s}; }
TR_HD f2 operator+(f2 a, float s) { return {a.x + s, a.y + s}; }
TR_HD f2 operator-(f2 a, float s) { return {a.x - s, a.y - s}; }
TR_HD f2 operator-(float s, f2 a) { return {s - a.x, s - a.y}; }
TR_HD float dot(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }

TR_HD f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
TR_HD f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
TR_HD f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
TR_HD f3 operator/(f3 a, f3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
TR_HD f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
TR_HD f3 operator*(float s, f3 a) { return {a.x * s, a.y * s, a.z * s}; }
TR_HD f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
TR_HD f3 operator+(f3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
TR_HD f3 operator-(f3 a, float s) { return {a.x - s, a.y - s, a.z - s}; }
TR_HD f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
TR_HD f3& operator+=(f3& a, f3 b) { a = a + b; return a; }
TR_HD f3& operator*=(f3& a, f3 b) { a = a * b; return a; }
TR_HD f3& operator*=(f3& a, float s) { a = a * s; return a; }
TR_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
TR_HD f3 cross(f3 a, f3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
TR_HD float length(f3 a) { return sqrtf(dot(a, a)); }
TR_HD f3 normalize(f3 a) { float l = length(a); return {a.x / l, a.y / l, a.z / l}; }
TR_HD float comp(const f3& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

TR_HD f4 operator+(f4 a, f4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
TR_HD f4 operator-(f4 a, f4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
TR_HD f4 operator*(f4 a, f4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
TR_HD f4 operator*(f4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
TR_HD f4 operator*(float s, f4 a) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }

TR_HD float fmin2(float a, float b) { return b < a ? b : a; }   // GLSL min
TR_HD float fmax2(float a, float b) { return a < b ? b : a; }   // GLSL max
TR_HD float clampf(float x, float lo, float hi) { return fmin2(fmax2(x, lo), hi); }
TR_HD int clampi(int x, int lo, int hi) { int t = x < lo ? lo : x; return t > hi ? hi : t; }
TR_HD uint clampu(uint x, uint lo, uint hi) { uint t = x < lo ? lo : x; return t > hi ? hi : t; }
TR_HD float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
TR_HD f3 mix3(f3 a, f3 b, float t) { return a * (1.0f - t) + b * t; }
TR_HD f4 mix4(f4 a, f4 b, float t) { return a * (1.0f - t) + b * t; }
TR_HD float stepf(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
TR_HD f3 max3(f3 a, f3 b) { return {fmax2(a.x, b.x), fmax2(a.y, b.y), fmax2(a.z, b.z)}; }
TR_HD f3 min3(f3 a, f3 b) { return {fmin2(a.x, b.x), fmin2(a.y, b.y), fmin2(a.z, b.z)}; }
TR_HD bool any_nan(f3 a) { return isnan(a.x) || isnan(a.y) || isnan(a.z); }

// sin / cos / pow of the shading code.  GLSL leaves their accuracy to the Vulkan implementation (SPIR-V precision table: sin and
// cos 2^-11 absolute on [-pi, pi], pow inherited from exp2(y * log2(x)) at a few ulp each); the IEEE build calls the C library's
// functions like the oracle, the shading translation unit built for speed (TR_SHADE_NATIVE_MATH: shade_fast.hip) the hardware's
// v_sin_f32 / v_cos_f32 / v_exp_f32 / v_log_f32.  Traversal and the triangle test use neither.
#if defined(TR_SHADE_NATIVE_MATH) && defined(__HIP_DEVICE_COMPILE__)
TR_DEV float tsin(float x) { return __sinf(x); }
TR_DEV float tcos(float x) { return __cosf(x); }
TR_DEV float tpow(float x, float y) { return y == 0.0f ? 1.0f : __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }
// a base that is a difference which can round a few ulps below zero (1 - cos at normal incidence): exp2(y * log2(x)) is NaN there where
// the C library's powf of the IEEE build returns -1e-35 for an odd integer exponent; clamped, both modes stay finite and agree to that
TR_DEV float tpow_ge0(float x, float y) { return tpow(x < 0.0f ? 0.0f : x, y); }
#else
TR_HD float tsin(float x) { return sinf(x); }
TR_HD float tcos(float x) { return cosf(x); }
TR_HD float tpow(float x, float y) { return powf(x, y); }
TR_HD float tpow_ge0(float x, float y) { return powf(x, y); }
#endif

// column-major matrices, as glm / GLSL
struct m3 { f3 c[3]; };
struct m4 { f4 c[4]; };
TR_HD f3 mul(const m3& m, f3 v) { return m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z; }          // M * v
TR_HD f3 mulT(f3 v, const m3& m) { return {dot(v, m.c[0]), dot(v, m.c[1]), dot(v, m.c[2])}; }   // v * M
TR_HD f4 mul(const m4& m, f4 v) { return m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z + m.c[3] * v.w; }
TR_HD m3 upper3(const m4& m) { return {{F3(m.c[0]), F3(m.c[1]), F3(m.c[2])}}; }
TR_HD f3 transform_point(const m4& m, f3 p) { return F3(mul(m, F4(p, 1.0f))); }

// ---------------------------------------------------------------------------
// POD layouts shared with the host
// ---------------------------------------------------------------------------
#pragma pack(push, 4)
struct Vertex { f3 pos; f3 normal; f2 uv; f4 tangent; };
struct Material {
    f4 albedo_factor, metallic_roughness_factor, emission_factor;
    float transmittance, ior, normal_factor; uint flags;
    int albedo_tex_id, metallic_roughness_tex_id, normal_tex_id, emission_tex_id;
};
struct Instance {
    int light_base_id, sh_grid_index; uint pad; float shadow_terminator_mul;
    m4 model, model_normal, model_prev; Material mat;
};
struct DirectionalLight { f3 color; int shadow_map_index; f3 dir; float dir_cutoff; };
struct PointLight {
    f3 color, dir, pos; float radius, dir_cutoff, dir_falloff, cutoff_radius, spot_radius;
    int shadow_map_index, padding;
};
struct TriLight { f3 pos[3]; uint emission_factor, instance_id, primitive_id; uint uv[3]; int emission_tex_id; };
struct AliasEntry { uint alias_id, probability; float pdf, alias_pdf; };
struct CameraData { m4 view, view_inverse, view_proj, proj_inverse; f4 origin, dof_params, projection_info, pan; };
struct MeshSpan { uint vertex_offset, vertex_count, index_offset, triangle_count; };
// The three vertices of one indexed triangle side by side: what k_shade reads for a hit instead of three indices and three 48-byte
// vertices from up to six cache lines.  Positions, normals and texture coordinates - what every hit needs, 96 bytes - fill one 128-byte
// record, i.e. exactly one cache line; the three tangents, which only a normal-mapped material reads, sit in an array of their own
// (SceneView::shade_tangents, three per record).  Until round 4 a record was the three vertices whole (144 bytes from two or three
// lines).  Record index = index_offset / 3 + primitive (instances of one mesh share the records); built at upload and after skinning
// (csrc/bvh_build.hip build_shade_tris), absent when the spans do not allow that index.
struct ShadeTri { f3 pos[3]; f3 normal[3]; f2 uv[3]; float pad[8]; };
struct Skin { uint joints[4]; float weights[4]; };   // mesh::skin_data (src/mesh.hh:32-36), `skin` of shader/skinning.comp:10-14
// texel_offset: where the texture starts in SceneView::texels, in 4-byte words; format: TEXTURE_FORMAT_RGBA8 (one word per texel) or
// TEXTURE_FORMAT_RGBA16 (two: what the reference stores a 16-bit PNG as, src/gltf.cc:548-556)
struct TextureInfo { uint width, height, texel_offset, format; };
enum { TEXTURE_FORMAT_RGBA8 = 0, TEXTURE_FORMAT_RGBA16 = 1 };
#pragma pack(pop)
static_assert(sizeof(Vertex) == 48 && sizeof(Material) == 80 && sizeof(Instance) == 288 && sizeof(ShadeTri) == 128, "layout");
static_assert(sizeof(DirectionalLight) == 32 && sizeof(PointLight) == 64 && sizeof(TriLight) == 64, "layout");
static_assert(sizeof(CameraData) == 320 && sizeof(AliasEntry) == 16, "layout");

// ---------------------------------------------------------------------------
// Acceleration structure in HBM
// ---------------------------------------------------------------------------
// 64-byte BVH2 node: both child boxes + child references.  What the builder clusters, optimises and then collapses into
// Bvh4Node; never traversed.
//   child >= 0 : internal node index;  child < 0 : leaf, triangle index = ~child
struct alignas(64) BvhNode {
    float lo0[3], hi0[3], lo1[3], hi1[3];
    int child0, child1;
    uint pad0, pad1;
};
static_assert(sizeof(BvhNode) == 64, "node layout");

// 4-wide node, one 128-byte line: child boxes in SoA (one dwordx4 per plane), child references as in BvhNode
// (>= 0 inner node, < 0 ~triangle).  Empty slots hold an inverted box (lo = +inf, hi = -inf) and are never hit.
struct alignas(128) Bvh4Node {
    float lox[4], hix[4], loy[4], hiy[4], loz[4], hiz[4];   // per axis lo at 32*axis, hi 16 bytes later: a ray reads its near plane at
                                                            // 32*axis + 16*(direction < 0) and its far plane at that offset ^ 16
    int child[4];
    int pad[4];
};
static_assert(sizeof(Bvh4Node) == 128, "Bvh4Node layout");

// 48-byte world-space triangle record, stored in Morton (leaf) order.
//   inst_flags: bits 0..30 instance id, bit 31 = non-opaque (runs the any-hit path)
//   alpha (non-opaque triangles only): what the any-hit test needs of the material.  Bit 31 clear: the bits of the candidate's alpha
//   itself - the material has no albedo texture, alpha = albedo_factor.a for every point of the triangle, no fetch at all.  Bit 31 set:
//   index of the triangle's AlphaTri record.  (Round 5: the test used to walk instance -> span -> three indices -> three vertices ->
//   texture table -> texels, four dependent round trips inside a triangle phase that every other lane of the wave waits through.)
// -DTR_TRI_STRIDE64=1 (an experiment, profiles/r5/tri_stride_ab.txt): records padded to 64 bytes, so that none straddles a 128-byte line
// (three of eight 48-byte records do)
#ifndef TR_TRI_STRIDE64
#define TR_TRI_STRIDE64 0
#endif
struct alignas(TR_TRI_STRIDE64 ? 64 : 16) TriRecord {
    float v0[3], v1[3], v2[3];
    uint inst_flags, prim, alpha;
};
static_assert(sizeof(TriRecord) == (TR_TRI_STRIDE64 ? 64 : 48), "tri layout");
// One per triangle of a non-opaque instance, at alpha_base[instance] + primitive: texture coordinates of the three vertices,
// albedo_factor.a and the albedo texture (get_interpolated_vertex_light + the alpha tap of shader/rt_common.rahit:15-24 in one 32-byte fetch).
struct alignas(16) AlphaTri { f2 uv0, uv1, uv2; float factor; int tex; };
static_assert(sizeof(AlphaTri) == 32, "alpha record layout");

struct HitRecord { int instance_id, primitive_id; float u, v, t; };

// Everything a kernel needs to see the scene (passed by value).
struct SceneView {
    const Instance* instances;
    const MeshSpan* spans;
    const Vertex* vertices;
    const uint* indices;
    const PointLight* point_lights;
    const DirectionalLight* directional_lights;
    const TriLight* tri_lights;
    const TextureInfo* tex_infos;
    const uint8_t* texels;
    const f4* envmap;
    const AliasEntry* alias_table;
    const CameraData* cameras;
    const CameraData* prev_cameras;   // camera_pair.previous (shader/scene.glsl:176-185)
    const MeshSpan* obj_spans;        // the uploaded model-space vertices even when `vertices` is the pre-transformed copy
    const Vertex* obj_vertices;
    const ShadeTri* shade_tris;  // null = none
    const f4* shade_tangents;    // three per record
    const TriRecord* tris;
    const AlphaTri* alpha_tris;  // records of the non-opaque triangles (TriRecord::alpha)
    const Bvh4Node* nodes4;      // the 4-wide fp32 nodes the traversal reads; node 0 is the root
    f4 environment_factor;
    int environment_proj;
    uint instance_count, point_light_count, directional_light_count, tri_light_count;
    uint env_w, env_h;
    uint tri_count, node_count;
    uint wide_textures;          // some texture of the scene is TEXTURE_FORMAT_RGBA16
};

}  // namespace tr
