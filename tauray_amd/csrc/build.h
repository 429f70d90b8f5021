// Host-side device scene container shared by the API and the BVH builder.
#pragma once
#include <string>
#include <vector>

#include "common.h"
#include "shading.h"

namespace tr {

int set_error(const std::string& msg);   // records trhip_last_error(); returns 1

struct DeviceScene {
    // uploaded by trhip_scene_upload
    Instance* instances = nullptr;
    MeshSpan* spans = nullptr;
    Vertex* vertices = nullptr;
    uint* indices = nullptr;
    PointLight* point_lights = nullptr;
    DirectionalLight* directional_lights = nullptr;
    TextureInfo* tex_infos = nullptr;
    uint8_t* texels = nullptr;
    f4* envmap = nullptr;
    AliasEntry* alias_table = nullptr;
    CameraData* cameras = nullptr;
    CameraData* prev_cameras = nullptr;  // defaults to a copy of `cameras`
    uint8_t* non_opaque = nullptr;
    uint* alpha_base = nullptr;          // per instance: first AlphaTri record of its triangles (non-opaque instances; 0xFFFFFFFF otherwise)
    AlphaTri* alpha_tris = nullptr;      // filled by the build and the refit (k_pretransform / k_retransform)
    uint alpha_count = 0;
    uint* tri_prefix = nullptr;          // instance_count + 1 prefix sums of triangle counts
    ShadeTri* shade_tris = nullptr;      // index_count / 3 records, or null (see common.h ShadeTri)
    f4* shade_tangents = nullptr;        // ... and their tangents, three per record, behind the records in the same allocation
    // scene_stage's pre-transformed vertex copy (shader/pre_transform.comp): one span per instance; vertices built on first use
    std::vector<MeshSpan> host_spans, host_world_spans;
    MeshSpan* world_spans = nullptr;
    Vertex* world_vertices = nullptr;
    uint world_vertex_count = 0;
    f4 environment_factor = {0, 0, 0, 0};
    int environment_proj = -1;
    uint instance_count = 0, point_light_count = 0, directional_light_count = 0, camera_count = 0, texture_count = 0;
    uint env_w = 0, env_h = 0, tri_count = 0, vertex_count = 0, index_count = 0;
    uint gather_emissive_triangles = 0, host_tri_light_count = 0;
    uint wide_textures = 0;              // some texture is RGBA16 (texture.h takes the one-dword path otherwise)
    // built by trhip_scene_build_accel
    BvhNode* nodes = nullptr;
    Bvh4Node* nodes4 = nullptr;
    TriRecord* tris = nullptr;
    TriLight* tri_lights = nullptr;
    uint node_count = 0, tri_light_count = 0;
    int builder = 1;                     // 0 = Karras LBVH, 1 = PLOC over the Morton order (TRHIP_BUILDER=lbvh|ploc)
    uint build_rounds = 0;
    int ploc_radius = 16;                // neighbour search radius of the PLOC rounds (TRHIP_PLOC_RADIUS)
    int optimise_rounds = 8;             // reinsertion rounds after the build (bvh_optimize.h; TRHIP_BVH_OPT)
    uint leaf_count = 0;                 // leaves of the tree = records in `tris`
    uint accel_tri_count = 0;            // triangle count of the scene the structure was built for
    bool collapse_by_cost = true;        // 4-wide nodes chosen by least area sum (k_collapse_cost) instead of greedily (TRHIP_COLLAPSE=greedy)
    bool fast_build = false;             // trhip_scene_set_build_mode: no optimisation rounds
    int optimise_modulus = 1;            // a node searches every optimise_modulus-th round (TRHIP_BVH_OPT_MOD)
    int dfs_layout = 1;                  // depth-first node order (TRHIP_NODE_LAYOUT=dfs|build)
    bool accel_built = false;
    uint accel_capacity = 0xFFFFFFFFu;   // leaf count the output buffers were allocated for
    void* scratch = nullptr;             // build temporaries, kept between builds
    float bounds_lo[3] = {0, 0, 0}, bounds_hi[3] = {0, 0, 0};   // centroid bounds of the last build
    // refit support (built on the first trhip_scene_refit_accel after a build): live 4-wide nodes in breadth-first order
    uint* level_nodes = nullptr;
    float* node_bounds = nullptr;        // 6 floats per node: union of its child boxes
    std::vector<uint> level_offsets;     // level l = level_nodes[level_offsets[l] .. level_offsets[l + 1])
    bool levels_valid = false;
    size_t scratch_bytes = 0;

    SceneView view() const {
        SceneView v;
        v.instances = instances; v.spans = spans; v.vertices = vertices; v.indices = indices;
        v.point_lights = point_lights; v.directional_lights = directional_lights; v.tri_lights = tri_lights;
        v.tex_infos = tex_infos; v.texels = texels; v.envmap = envmap; v.alias_table = alias_table;
        v.cameras = cameras; v.prev_cameras = prev_cameras ? prev_cameras : cameras; v.obj_spans = spans; v.obj_vertices = vertices; v.shade_tris = shade_tris; v.shade_tangents = shade_tangents; v.tris = tris; v.alpha_tris = alpha_tris; v.nodes4 = nodes4;
        v.environment_factor = environment_factor; v.environment_proj = environment_proj;
        v.instance_count = instance_count; v.point_light_count = point_light_count;
        v.directional_light_count = directional_light_count; v.tri_light_count = tri_light_count;
        v.env_w = env_w; v.env_h = env_h; v.wide_textures = wide_textures; v.tri_count = accel_built ? tri_count : 0; v.node_count = node_count;
        return v;
    }
    void free_accel() {
        if (nodes) (void)hipFree(nodes);
        if (nodes4) (void)hipFree(nodes4);
        if (tris) (void)hipFree(tris);
        if (tri_lights) (void)hipFree(tri_lights);
        nodes = nullptr; nodes4 = nullptr; tris = nullptr; tri_lights = nullptr; node_count = 0; tri_light_count = 0; accel_built = false;
        accel_capacity = 0xFFFFFFFFu;
        if (level_nodes) (void)hipFree(level_nodes);
        if (node_bounds) (void)hipFree(node_bounds);
        level_nodes = nullptr; node_bounds = nullptr; level_offsets.clear(); levels_valid = false;
    }
    // skinned meshes (mesh(mesh* animation_source), src/mesh.hh:47): bind-pose vertices + skin records per instance
    struct SkinSlot { Vertex* source = nullptr; Skin* skins = nullptr; m4* joints = nullptr; uint joint_capacity = 0; uint vertex_count = 0; };
    std::vector<SkinSlot> skin_slots;
    void free_skins() {
        for (SkinSlot& k : skin_slots) { if (k.source) (void)hipFree(k.source); if (k.skins) (void)hipFree(k.skins); if (k.joints) (void)hipFree(k.joints); }
        skin_slots.clear();
    }
    void free_all() {
        free_accel();
        free_skins();
        void* ptrs[] = {instances, spans, vertices, indices, point_lights, directional_lights, tex_infos, texels, envmap,
                        alias_table, cameras, prev_cameras, non_opaque, tri_prefix, world_spans, world_vertices, scratch, shade_tris, alpha_base, alpha_tris};
        for (void* p : ptrs) if (p) (void)hipFree(p);
        *this = DeviceScene();
    }
};

int build_accel(DeviceScene& ds, hipStream_t stream, trhip_accel_info* info);
int ensure_world_vertices(DeviceScene& ds, hipStream_t stream);
int build_shade_tris(DeviceScene& ds, int instance, hipStream_t stream);   // instance < 0: every mesh (after an upload); else the mesh of that instance (after skinning)
int skin_instance(DeviceScene& ds, uint instance, const float* joint_transforms, uint joint_count, hipStream_t stream);   // skinning.comp
int refit_accel(DeviceScene& ds, hipStream_t stream, trhip_accel_info* info);   // same tree, new boxes (after trhip_scene_update_instances)   // pre_transform.comp per instance

}  // namespace tr
