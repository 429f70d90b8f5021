// BVH traversal + watertight ray/triangle test for gfx950 (wave64).
//
// Replaces the Vulkan driver's traceRayEXT (shader/path_tracer.glsl:38-50,387-403)
// and the hit/miss shader table (shader/rt_common*.r*).  One ray per lane; each
// lane keeps its short traversal stack in LDS (column layout stack[entry][lane]
// -> conflict free for ds_read/write_b32 regardless of per-lane depth) and
// spills to a private array only past LDS_STACK entries.
#pragma once
#include "common.h"
#include "rng.h"
#include "texture.h"

namespace tr {

#define TR_BLOCK 256
#ifndef TR_VOTE
#define TR_VOTE 8          // closest-hit traversal: lanes holding a leaf wait until this many do (0 disables the vote)
#endif
#define TR_LDS_STACK 16
#define TR_SPILL_STACK 112

struct RayPre {
    f3 org, dir, inv_dir;
    int kx, ky, kz;
    float Sx, Sy, Sz;
};

TR_DEV RayPre make_ray(f3 org, f3 dir) {
    RayPre r;
    r.org = org; r.dir = dir;
    float ax = fabsf(dir.x), ay = fabsf(dir.y), az = fabsf(dir.z);
    int kz = (ax > ay) ? (ax > az ? 0 : 2) : (ay > az ? 1 : 2);
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    if (comp(dir, kz) < 0.0f) { int t = kx; kx = ky; ky = t; }
    r.kx = kx; r.ky = ky; r.kz = kz;
    r.Sx = comp(dir, kx) / comp(dir, kz);
    r.Sy = comp(dir, ky) / comp(dir, kz);
    r.Sz = 1.0f / comp(dir, kz);
    r.inv_dir = F3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    return r;
}

TR_DEV bool ray_is_finite(f3 o, f3 d) {
    return isfinite(o.x) && isfinite(o.y) && isfinite(o.z) && isfinite(d.x) && isfinite(d.y) && isfinite(d.z) &&
           (d.x != 0.0f || d.y != 0.0f || d.z != 0.0f);
}

// Watertight test (Woop, Benthin, Wald 2013), no culling.  Plain IEEE fp32 without
// contraction: the shared-edge guarantee needs both products of each edge function
// rounded, and the CPU oracle evaluates exactly the same expression tree.
TR_DEV bool tri_intersect(const RayPre& r, f3 v0, f3 v1, f3 v2, float tmin, float tmax, float& t, float& bu, float& bv) {
#pragma clang fp contract(off)
    const f3 A = v0 - r.org, B = v1 - r.org, C = v2 - r.org;
    const float Akz = comp(A, r.kz), Bkz = comp(B, r.kz), Ckz = comp(C, r.kz);
    const float Ax = comp(A, r.kx) - r.Sx * Akz, Ay = comp(A, r.ky) - r.Sy * Akz;
    const float Bx = comp(B, r.kx) - r.Sx * Bkz, By = comp(B, r.ky) - r.Sy * Bkz;
    const float Cx = comp(C, r.kx) - r.Sx * Ckz, Cy = comp(C, r.ky) - r.Sy * Ckz;
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

// Conservative slab test; returns entry distance in tnear.
TR_DEV bool box_intersect(const RayPre& r, const float* lo, const float* hi, float tmin, float tmax, float& tnear) {
    float tx0 = (lo[0] - r.org.x) * r.inv_dir.x, tx1 = (hi[0] - r.org.x) * r.inv_dir.x;
    float ty0 = (lo[1] - r.org.y) * r.inv_dir.y, ty1 = (hi[1] - r.org.y) * r.inv_dir.y;
    float tz0 = (lo[2] - r.org.z) * r.inv_dir.z, tz1 = (hi[2] - r.org.z) * r.inv_dir.z;
    float nx = fminf(tx0, tx1), fx = fmaxf(tx0, tx1);
    float ny = fminf(ty0, ty1), fy = fmaxf(ty0, ty1);
    float nz = fminf(tz0, tz1), fz = fmaxf(tz0, tz1);
    // fminf/fmaxf drop NaNs (0 * inf when the origin lies on a slab plane of a zero-direction axis)
    float t0 = fmaxf(fmaxf(nx, ny), fmaxf(nz, tmin));
    // far planes widened by 1 + 2*gamma(3) so rounding can never cull a true hit
    float t1 = fminf(fminf(fminf(fx, fy), fz) * 1.0000003576278687f, tmax);
    tnear = t0;
    return t0 <= t1;
}

// Per-lane traversal stack: first TR_LDS_STACK entries in LDS, the rest in a private spill array.
// `sp` must stay in a VGPR and the LDS access must stay a ds_read/ds_write: the spill array is therefore a separate
// local (a struct member array drags the whole struct, sp included, into scratch) and pop() reads LDS
// unconditionally (an if/else over the two memories is if-converted into a generic pointer + flat_load).
struct LaneStack {
    int* lds;           // &stack[0][lane_in_block]; stride TR_BLOCK
    int sp;
    int overflow;       // count of dropped pushes (an int in a VGPR, not a wave-level predicate)
    TR_DEV void init(int* base) { lds = base; sp = 0; overflow = 0; }
    TR_DEV void push(int* spill, int v) {
        const bool ok = sp < TR_LDS_STACK + TR_SPILL_STACK;
        if (sp < TR_LDS_STACK) lds[sp * TR_BLOCK] = v;
        else if (ok) spill[sp - TR_LDS_STACK] = v;
        overflow += ok ? 0 : 1;
        sp += ok ? 1 : 0;
    }
    TR_DEV int pop(const int* spill) {
        sp--;
        int v = lds[(sp < TR_LDS_STACK ? sp : 0) * TR_BLOCK];
        asm volatile("" : "+v"(v));   // pin the ds_read: no select-of-pointers + flat_load
        if (sp >= TR_LDS_STACK) v = spill[sp - TR_LDS_STACK];
        return v;
    }
};

struct TraceStats { uint nodes, tris, alpha, maxsp; };

// get_interpolated_vertex_light (shader/rt.glsl:103-117): uv at a candidate hit
TR_DEV f2 candidate_uv(const SceneView& sv, int inst, int prim, float bu, float bv) {
    const MeshSpan sp = sv.spans[inst];
    const uint* ix = sv.indices + sp.index_offset + 3u * (uint)prim;
    const Vertex* vb = sv.vertices + sp.vertex_offset;
    f2 uv0 = vb[ix[0]].uv, uv1 = vb[ix[1]].uv, uv2 = vb[ix[2]].uv;
    float b0 = 1.0f - bu - bv;
    return uv0 * b0 + uv1 * bu + uv2 * bv;
}

// Candidate alpha of a non-opaque triangle (albedo_factor.a * texture alpha).
TR_DEV float candidate_alpha(const SceneView& sv, int inst, int prim, float bu, float bv) {
    const Material& mat = sv.instances[inst].mat;
    float alpha = mat.albedo_factor.w;
    int tex = mat.albedo_tex_id;
    if (tex >= 0) alpha *= sample_texture(sv, tex, candidate_uv(sv, inst, prim, bu, bv)).w;
    return alpha;
}

// Traversal-order independent stand-in for generate_single_uniform_random(payload.random_seed)
// (shader/rt_common.rahit:21); see DESIGN.md "any-hit order".
TR_DEV float alpha_cutoff_hash(uint seed, int instance_id, int primitive_id) {
    uint k = (uint)instance_id * 0x9E3779B9u + (uint)primitive_id;
    uint h = seed ^ pcg(k);
    return (float)pcg(h) * 2.3283064365386963e-10f;
}

// Closest hit over triangles (+ sphere lights).  ALPHA_MODE 0: stochastic alpha keyed by `seed`
// (shader/rt_common.rahit:15-24); 1: fixed cutoff 1e-4 (shader/rt_feature.rahit:17).
template <int ALPHA_MODE, bool COUNT>
TR_DEV void trace_closest(const SceneView& sv, f3 org, f3 dir, float tmin, float tmax, bool include_lights, uint seed,
                          int* lds_stack, HitRecord& hit, TraceStats& st, int& overflow) {
    hit.instance_id = -1; hit.primitive_id = -1; hit.u = 0; hit.v = 0; hit.t = -1.0f;
    float best_t = tmax;
    bool found = false;
    uint best_inst = 0xFFFFFFFFu, best_prim = 0xFFFFFFFFu;
    RayPre r = make_ray(org, dir);
    // A ray with a non-finite origin/direction or a zero direction is outside traceRayEXT's contract; the reference
    // produces one when a refraction sample fails at bounce 0 (ggx.glsl:343-348 sets out_dir = vec3(0)).  It is defined
    // here as a miss; without this guard a zero direction passes the slab test of every box on its positive side.
    const bool finite_ray = ray_is_finite(org, dir);
    if (sv.tri_count > 0 && finite_ray) {
        LaneStack stk;
        int spill[TR_SPILL_STACK];
        stk.init(lds_stack);
        int node = sv.node_count > 0 ? 0 : -1;   // single-triangle scene: leaf ~0 == -1
        while (true) {
#if TR_VOTE > 0
            // wave vote: run the (expensive) triangle branch only when enough lanes hold a leaf; leaf lanes wait otherwise
            const bool at_leaf = node < 0;
            const int n_leaf = __popcll(__ballot(at_leaf)), n_all = __popcll(__ballot(true));
            const bool leaf_phase = n_leaf >= TR_VOTE || n_leaf == n_all;
            if (at_leaf != leaf_phase) continue;
#endif
            if (node >= 0) {
                const BvhNode n = sv.nodes[node];
                if (COUNT) st.nodes++;
                float t0, t1;
                bool h0 = box_intersect(r, n.lo0, n.hi0, tmin, best_t, t0);
                bool h1 = box_intersect(r, n.lo1, n.hi1, tmin, best_t, t1);
                if (h0 && h1) {
                    bool first0 = t0 <= t1;
                    stk.push(spill, first0 ? n.child1 : n.child0);
                    if (COUNT) st.maxsp = max(st.maxsp, (uint)stk.sp);
                    node = first0 ? n.child0 : n.child1;
                    continue;
                } else if (h0) { node = n.child0; continue; }
                else if (h1) { node = n.child1; continue; }
            } else {
                const TriRecord tr = sv.tris[~node];
                if (COUNT) st.tris++;
                float t, bu, bv;
                f3 v0 = F3(tr.v0[0], tr.v0[1], tr.v0[2]), v1 = F3(tr.v1[0], tr.v1[1], tr.v1[2]), v2 = F3(tr.v2[0], tr.v2[1], tr.v2[2]);
                if (tri_intersect(r, v0, v1, v2, tmin, __builtin_huge_valf(), t, bu, bv)) {
                    const uint inst = tr.inst_flags & 0x7FFFFFFFu;
                    const bool closer = t < best_t ||
                        (t == best_t && found && (inst < best_inst || (inst == best_inst && tr.prim < best_prim)));
                    if (closer && t < tmax) {
                        bool accept = true;
                        if (tr.inst_flags & 0x80000000u) {
                            if (COUNT) st.alpha++;
                            float a = candidate_alpha(sv, (int)inst, (int)tr.prim, bu, bv);
                            float cutoff = ALPHA_MODE == 0 ? alpha_cutoff_hash(seed, (int)inst, (int)tr.prim) : 0.0001f;
                            accept = !(a <= cutoff);   // is_material_skippable (shader/rt.glsl:136-144)
                        }
                        if (accept) {
                            best_t = t; found = true; best_inst = inst; best_prim = tr.prim;
                            hit.instance_id = (int)inst; hit.primitive_id = (int)tr.prim; hit.u = bu; hit.v = bv;
                        }
                    }
                }
            }
            if (stk.sp == 0) break;
            node = stk.pop(spill);
        }
        overflow += stk.overflow;
    }
    if (include_lights && finite_ray) {
        // rt_common_point_light.rint:11-17 / .rchit:10-15, shader/rt_common.glsl:36-51
        for (uint i = 0; i < sv.point_light_count; ++i) {
            const PointLight& pl = sv.point_lights[i];
            float radius = pl.radius;
            if (radius == 0.0f) continue;
            f3 oc = org - pl.pos;
            float a = dot(dir, dir);
            float b = 2.0f * dot(oc, dir);
            float c = dot(oc, oc) - radius * radius;
            float disc = b * b - 4.0f * a * c;
            if (disc < 0) continue;
            float h = (-b - sqrtf(disc)) / (2.0f * a);
            if (h > 0 && h > tmin && h < best_t) {
                best_t = h; found = true;
                hit.instance_id = -1; hit.primitive_id = (int)i; hit.u = h; hit.v = 0;
            }
        }
    }
    hit.t = found ? best_t : -1.0f;
}

// shadow_ray (shader/path_tracer.glsl:35-52) + rt_common_shadow.rahit/.rchit: product of (1 - alpha)
// over non-opaque hits, 0 on the first opaque hit; lights are excluded (mask 0xFD).
template <bool COUNT>
TR_DEV float trace_shadow(const SceneView& sv, f3 org, f3 dir, float tmin, float tmax, int* lds_stack, TraceStats& st,
                          int& overflow) {
    float visibility = 1.0f;
    if (sv.tri_count == 0 || !ray_is_finite(org, dir)) return visibility;
    RayPre r = make_ray(org, dir);
    LaneStack stk;
    int spill[TR_SPILL_STACK];
    stk.init(lds_stack);
    int node = sv.node_count > 0 ? 0 : -1;
    while (true) {
#ifdef TR_VOTE_SHADOW
        const bool at_leaf = node < 0;
        const int n_leaf = __popcll(__ballot(at_leaf)), n_all = __popcll(__ballot(true));
        const bool leaf_phase = n_leaf >= TR_VOTE_SHADOW || n_leaf == n_all;
        if (at_leaf != leaf_phase) continue;
#endif
        if (node >= 0) {
            const BvhNode n = sv.nodes[node];
            if (COUNT) st.nodes++;
            float t0, t1;
            bool h0 = box_intersect(r, n.lo0, n.hi0, tmin, tmax, t0);
            bool h1 = box_intersect(r, n.lo1, n.hi1, tmin, tmax, t1);
            if (h0 && h1) { stk.push(spill, n.child1); node = n.child0; continue; }
            else if (h0) { node = n.child0; continue; }
            else if (h1) { node = n.child1; continue; }
        } else {
            const TriRecord tr = sv.tris[~node];
            if (COUNT) st.tris++;
            float t, bu, bv;
            f3 v0 = F3(tr.v0[0], tr.v0[1], tr.v0[2]), v1 = F3(tr.v1[0], tr.v1[1], tr.v1[2]), v2 = F3(tr.v2[0], tr.v2[1], tr.v2[2]);
            if (tri_intersect(r, v0, v1, v2, tmin, tmax, t, bu, bv)) {
                if (!(tr.inst_flags & 0x80000000u)) { visibility = 0.0f; break; }
                if (COUNT) st.alpha++;
                float alpha = candidate_alpha(sv, (int)(tr.inst_flags & 0x7FFFFFFFu), (int)tr.prim, bu, bv);
                visibility *= 1.0f - alpha;
                if (visibility == 0.0f) break;
            }
        }
        if (stk.sp == 0) break;
        node = stk.pop(spill);
    }
    overflow += stk.overflow;
    return visibility;
}

// =====================================================================================================================
// 8-wide compressed BVH traversal.  Stack entries are node groups (child_base, hit bits << 24 | imask); one entry
// per level at most, so twelve LDS entries cover trees over 10^9 triangles before spilling.
#define TR_LDS_STACK8 12
#define TR_SPILL_STACK8 20

struct LaneStack8 {
    uint2* lds;          // &stack[0][lane_in_block]; stride TR_BLOCK
    int sp;
    int overflow;
    TR_DEV void init(int* base) { lds = reinterpret_cast<uint2*>(base); sp = 0; overflow = 0; }
    TR_DEV void push(uint2* spill, uint2 v) {
        const bool ok = sp < TR_LDS_STACK8 + TR_SPILL_STACK8;
        if (sp < TR_LDS_STACK8) lds[sp * TR_BLOCK] = v;
        else if (ok) spill[sp - TR_LDS_STACK8] = v;
        overflow += ok ? 0 : 1;
        sp += ok ? 1 : 0;
    }
    TR_DEV uint2 pop(const uint2* spill) {
        sp--;
        uint2 v = lds[(sp < TR_LDS_STACK8 ? sp : 0) * TR_BLOCK];
        asm volatile("" : "+v"(v.x), "+v"(v.y));   // keep the LDS read a ds_read_b64
        if (sp >= TR_LDS_STACK8) v = spill[sp - TR_LDS_STACK8];
        return v;
    }
};

struct Ray8 {
    f3 org, inv_dir;     // inv_dir with zero components replaced by +-huge so slabs outside the origin reject
    uint octinv;         // 7 - octant
};

TR_DEV Ray8 make_ray8(f3 org, f3 dir) {
    Ray8 r;
    r.org = org;
    const float eps = 1e-30f;
    r.inv_dir.x = 1.0f / (fabsf(dir.x) > eps ? dir.x : copysignf(eps, dir.x));
    r.inv_dir.y = 1.0f / (fabsf(dir.y) > eps ? dir.y : copysignf(eps, dir.y));
    r.inv_dir.z = 1.0f / (fabsf(dir.z) > eps ? dir.z : copysignf(eps, dir.z));
    uint oct = (dir.x < 0.0f ? 4u : 0u) | (dir.y < 0.0f ? 2u : 0u) | (dir.z < 0.0f ? 1u : 0u);
    r.octinv = 7u - oct;
    return r;
}

// Tests the eight quantised child boxes of one node.  Returns hit bits: bits 24..31 internal children in traversal
// priority order (slot ^ octinv), bits 0..23 leaf triangles.
TR_DEV uint intersect_node8(const Ray8& r, const uint4 n0, const uint4 n1, const uint4 n2, const uint4 n3, const uint4 n4, float tmin, float tmax) {
    const float px = __uint_as_float(n0.x), py = __uint_as_float(n0.y), pz = __uint_as_float(n0.z);
    const float sx = __uint_as_float((n0.w & 0xFFu) << 23), sy = __uint_as_float(((n0.w >> 8) & 0xFFu) << 23),
                sz = __uint_as_float(((n0.w >> 16) & 0xFFu) << 23);
    uint hitmask = 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const uint meta4 = half ? n1.w : n1.z;
        const uint qlx = half ? n2.y : n2.x, qly = half ? n2.w : n2.z;
        const uint qlz = half ? n3.y : n3.x, qhx = half ? n3.w : n3.z;
        const uint qhy = half ? n4.y : n4.x, qhz = half ? n4.w : n4.z;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int sh = 8 * k;
            // plane positions decoded exactly as the builder verified them: q * scale + p.  The product is exact
            // (q <= 255, scale a power of two), so one fma gives the same bits as mul + add.
            const float x0 = __fmaf_rn((float)((qlx >> sh) & 0xFFu), sx, px), x1 = __fmaf_rn((float)((qhx >> sh) & 0xFFu), sx, px);
            const float y0 = __fmaf_rn((float)((qly >> sh) & 0xFFu), sy, py), y1 = __fmaf_rn((float)((qhy >> sh) & 0xFFu), sy, py);
            const float z0 = __fmaf_rn((float)((qlz >> sh) & 0xFFu), sz, pz), z1 = __fmaf_rn((float)((qhz >> sh) & 0xFFu), sz, pz);
            const float tx0 = (x0 - r.org.x) * r.inv_dir.x, tx1 = (x1 - r.org.x) * r.inv_dir.x;
            const float ty0 = (y0 - r.org.y) * r.inv_dir.y, ty1 = (y1 - r.org.y) * r.inv_dir.y;
            const float tz0 = (z0 - r.org.z) * r.inv_dir.z, tz1 = (z1 - r.org.z) * r.inv_dir.z;
            const float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), tmin));
            const float tf = fminf(fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1)) * 1.0000003576278687f, tmax);
            const uint m = (meta4 >> sh) & 0xFFu;
            // an empty slot has meta 0 (no bits); branch-free insertion of the child's bits at its priority position
            const uint inner = ((m & 0x18u) == 0x18u) ? r.octinv : 0u;
            const uint bits = (m >> 5) << ((m ^ inner) & 0x1Fu);
            hitmask |= (tn <= tf) ? bits : 0u;
        }
    }
    return hitmask;
}

TR_DEV void load_node8(const Bvh8Node* nodes, uint index, uint4& n0, uint4& n1, uint4& n2, uint4& n3, uint4& n4) {
    const uint4* p = reinterpret_cast<const uint4*>(nodes + index);
    n0 = p[0]; n1 = p[1]; n2 = p[2]; n3 = p[3]; n4 = p[4];
}

template <int ALPHA_MODE, bool COUNT>
TR_DEV void trace_closest8(const SceneView& sv, f3 org, f3 dir, float tmin, float tmax, bool include_lights, uint seed,
                           int* lds_stack, HitRecord& hit, TraceStats& st, int& overflow) {
    hit.instance_id = -1; hit.primitive_id = -1; hit.u = 0; hit.v = 0; hit.t = -1.0f;
    float best_t = tmax;
    bool found = false;
    uint best_inst = 0xFFFFFFFFu, best_prim = 0xFFFFFFFFu;
    const bool finite_ray = ray_is_finite(org, dir);
    if (sv.tri_count > 0 && finite_ray) {
        const RayPre r = make_ray(org, dir);
        const Ray8 r8 = make_ray8(org, dir);
        LaneStack8 stk;
        uint2 spill[TR_SPILL_STACK8];
        stk.init(lds_stack);
        uint2 G = make_uint2(0u, 0x80000000u);     // the root as a one-child node group
        while (true) {
            uint2 T;
            if (G.y > 0x00FFFFFFu) {
                const uint hits = G.y;
                const int bit = 31 - __clz((int)hits);
                G.y &= ~(1u << bit);
                if (G.y > 0x00FFFFFFu) { stk.push(spill, G); if (COUNT) st.maxsp = max(st.maxsp, (uint)stk.sp); }
                const uint slot = ((uint)bit - 24u) ^ r8.octinv;
                const uint rel = __popc(hits & ~(0xFFFFFFFFu << slot) & 0xFFu);
                uint4 n0, n1, n2, n3, n4;
                load_node8(sv.nodes8, G.x + rel, n0, n1, n2, n3, n4);
                if (COUNT) st.nodes++;
                const uint hm = intersect_node8(r8, n0, n1, n2, n3, n4, tmin, best_t);
                G = make_uint2(n1.x, (hm & 0xFF000000u) | (n0.w >> 24));
                T = make_uint2(n1.y, hm & 0x00FFFFFFu);
            } else {
                T = make_uint2(0u, 0u);
            }
            while (T.y) {
                const int b = __ffs((int)T.y) - 1;
                T.y &= T.y - 1u;
                const TriRecord tr = sv.tris[T.x + (uint)b];
                if (COUNT) st.tris++;
                float t, bu, bv;
                f3 v0 = F3(tr.v0[0], tr.v0[1], tr.v0[2]), v1 = F3(tr.v1[0], tr.v1[1], tr.v1[2]), v2 = F3(tr.v2[0], tr.v2[1], tr.v2[2]);
                if (tri_intersect(r, v0, v1, v2, tmin, __builtin_huge_valf(), t, bu, bv)) {
                    const uint inst = tr.inst_flags & 0x7FFFFFFFu;
                    const bool closer = t < best_t ||
                        (t == best_t && found && (inst < best_inst || (inst == best_inst && tr.prim < best_prim)));
                    if (closer && t < tmax) {
                        bool accept = true;
                        if (tr.inst_flags & 0x80000000u) {
                            if (COUNT) st.alpha++;
                            float a = candidate_alpha(sv, (int)inst, (int)tr.prim, bu, bv);
                            float cutoff = ALPHA_MODE == 0 ? alpha_cutoff_hash(seed, (int)inst, (int)tr.prim) : 0.0001f;
                            accept = !(a <= cutoff);
                        }
                        if (accept) {
                            best_t = t; found = true; best_inst = inst; best_prim = tr.prim;
                            hit.instance_id = (int)inst; hit.primitive_id = (int)tr.prim; hit.u = bu; hit.v = bv;
                        }
                    }
                }
            }
            if (G.y <= 0x00FFFFFFu) {
                if (stk.sp == 0) break;
                G = stk.pop(spill);
            }
        }
        overflow += stk.overflow;
    }
    if (include_lights && finite_ray) {
        for (uint i = 0; i < sv.point_light_count; ++i) {
            const PointLight& pl = sv.point_lights[i];
            float radius = pl.radius;
            if (radius == 0.0f) continue;
            f3 oc = org - pl.pos;
            float a = dot(dir, dir);
            float b = 2.0f * dot(oc, dir);
            float c = dot(oc, oc) - radius * radius;
            float disc = b * b - 4.0f * a * c;
            if (disc < 0) continue;
            float h = (-b - sqrtf(disc)) / (2.0f * a);
            if (h > 0 && h > tmin && h < best_t) {
                best_t = h; found = true;
                hit.instance_id = -1; hit.primitive_id = (int)i; hit.u = h; hit.v = 0;
            }
        }
    }
    hit.t = found ? best_t : -1.0f;
}

template <bool COUNT>
TR_DEV float trace_shadow8(const SceneView& sv, f3 org, f3 dir, float tmin, float tmax, int* lds_stack, TraceStats& st, int& overflow) {
    float visibility = 1.0f;
    if (sv.tri_count == 0 || !ray_is_finite(org, dir)) return visibility;
    const RayPre r = make_ray(org, dir);
    const Ray8 r8 = make_ray8(org, dir);
    LaneStack8 stk;
    uint2 spill[TR_SPILL_STACK8];
    stk.init(lds_stack);
    uint2 G = make_uint2(0u, 0x80000000u);
    bool done = false;
    while (!done) {
        uint2 T;
        if (G.y > 0x00FFFFFFu) {
            const uint hits = G.y;
            const int bit = 31 - __clz((int)hits);
            G.y &= ~(1u << bit);
            if (G.y > 0x00FFFFFFu) stk.push(spill, G);
            const uint slot = ((uint)bit - 24u) ^ r8.octinv;
            const uint rel = __popc(hits & ~(0xFFFFFFFFu << slot) & 0xFFu);
            uint4 n0, n1, n2, n3, n4;
            load_node8(sv.nodes8, G.x + rel, n0, n1, n2, n3, n4);
            if (COUNT) st.nodes++;
            const uint hm = intersect_node8(r8, n0, n1, n2, n3, n4, tmin, tmax);
            G = make_uint2(n1.x, (hm & 0xFF000000u) | (n0.w >> 24));
            T = make_uint2(n1.y, hm & 0x00FFFFFFu);
        } else {
            T = make_uint2(0u, 0u);
        }
        while (T.y) {
            const int b = __ffs((int)T.y) - 1;
            T.y &= T.y - 1u;
            const TriRecord tr = sv.tris[T.x + (uint)b];
            if (COUNT) st.tris++;
            float t, bu, bv;
            f3 v0 = F3(tr.v0[0], tr.v0[1], tr.v0[2]), v1 = F3(tr.v1[0], tr.v1[1], tr.v1[2]), v2 = F3(tr.v2[0], tr.v2[1], tr.v2[2]);
            if (tri_intersect(r, v0, v1, v2, tmin, tmax, t, bu, bv)) {
                if (!(tr.inst_flags & 0x80000000u)) { visibility = 0.0f; done = true; break; }
                if (COUNT) st.alpha++;
                float alpha = candidate_alpha(sv, (int)(tr.inst_flags & 0x7FFFFFFFu), (int)tr.prim, bu, bv);
                visibility *= 1.0f - alpha;
                if (visibility == 0.0f) { done = true; break; }
            }
        }
        if (!done && G.y <= 0x00FFFFFFu) {
            if (stk.sp == 0) break;
            G = stk.pop(spill);
        }
    }
    overflow += stk.overflow;
    return visibility;
}


// ---------------------------------------------------------------------------------------------------------------
// Stepwise 8-wide traversal for the persistent kernels: `step()` performs one node visit plus the triangles of that
// node and reports completion, so a wave can hand finished lanes a new ray instead of idling until its slowest
// ray is done (lane-level dynamic fetch, Aila & Laine 2009).
template <bool SHADOW, int ALPHA_MODE, bool COUNT>
struct Trav8 {
    RayPre r;
    Ray8 r8;
    uint2 G;
    LaneStack8 stk;
    float tmin, tmax, best_t;      // best_t: closest distance (closest-hit) or accumulated visibility (shadow)
    uint best_inst, best_prim, seed;
    float hu, hv;
    bool found;

    TR_DEV void begin(f3 org, f3 dir, float tmin_, float tmax_, uint seed_, int* lds_stack) {
        r = make_ray(org, dir);
        r8 = make_ray8(org, dir);
        G = make_uint2(0u, 0x80000000u);
        stk.init(lds_stack);
        tmin = tmin_; tmax = tmax_; seed = seed_;
        best_t = SHADOW ? 1.0f : tmax_;
        best_inst = 0xFFFFFFFFu; best_prim = 0xFFFFFFFFu; hu = 0; hv = 0; found = false;
    }

    // returns true when the ray is finished
    TR_DEV bool step(const SceneView& sv, uint2* spill, TraceStats& st) {
        uint2 T = make_uint2(0u, 0u);
        if (G.y > 0x00FFFFFFu) {
            const uint hits = G.y;
            const int bit = 31 - __clz((int)hits);
            G.y &= ~(1u << bit);
            if (G.y > 0x00FFFFFFu) { stk.push(spill, G); if (COUNT) st.maxsp = max(st.maxsp, (uint)stk.sp); }
            const uint slot = ((uint)bit - 24u) ^ r8.octinv;
            const uint rel = __popc(hits & ~(0xFFFFFFFFu << slot) & 0xFFu);
            uint4 n0, n1, n2, n3, n4;
            load_node8(sv.nodes8, G.x + rel, n0, n1, n2, n3, n4);
            if (COUNT) st.nodes++;
            const uint hm = intersect_node8(r8, n0, n1, n2, n3, n4, tmin, SHADOW ? tmax : best_t);
            G = make_uint2(n1.x, (hm & 0xFF000000u) | (n0.w >> 24));
            T = make_uint2(n1.y, hm & 0x00FFFFFFu);
        }
        while (T.y) {
            const int b = __ffs((int)T.y) - 1;
            T.y &= T.y - 1u;
            const TriRecord tr = sv.tris[T.x + (uint)b];
            if (COUNT) st.tris++;
            float t, bu, bv;
            f3 v0 = F3(tr.v0[0], tr.v0[1], tr.v0[2]), v1 = F3(tr.v1[0], tr.v1[1], tr.v1[2]), v2 = F3(tr.v2[0], tr.v2[1], tr.v2[2]);
            if (!tri_intersect(r, v0, v1, v2, tmin, SHADOW ? tmax : __builtin_huge_valf(), t, bu, bv)) continue;
            const uint inst = tr.inst_flags & 0x7FFFFFFFu;
            if (SHADOW) {
                if (!(tr.inst_flags & 0x80000000u)) { best_t = 0.0f; return true; }
                if (COUNT) st.alpha++;
                best_t *= 1.0f - candidate_alpha(sv, (int)inst, (int)tr.prim, bu, bv);
                if (best_t == 0.0f) return true;
            } else {
                const bool closer = t < best_t || (t == best_t && found && (inst < best_inst || (inst == best_inst && tr.prim < best_prim)));
                if (closer && t < tmax) {
                    bool accept = true;
                    if (tr.inst_flags & 0x80000000u) {
                        if (COUNT) st.alpha++;
                        float a = candidate_alpha(sv, (int)inst, (int)tr.prim, bu, bv);
                        float cutoff = ALPHA_MODE == 0 ? alpha_cutoff_hash(seed, (int)inst, (int)tr.prim) : 0.0001f;
                        accept = !(a <= cutoff);
                    }
                    if (accept) { best_t = t; found = true; best_inst = inst; best_prim = tr.prim; hu = bu; hv = bv; }
                }
            }
        }
        if (G.y <= 0x00FFFFFFu) {
            if (stk.sp == 0) return true;
            G = stk.pop(spill);
        }
        return false;
    }

};


// Stepwise BVH2 traversal (one node OR one triangle per step) for the lane-refill kernels.
template <bool SHADOW, int ALPHA_MODE, bool COUNT>
struct Trav2 {
    RayPre r;
    LaneStack stk;
    int node;
    float tmin, tmax, best_t;
    uint best_inst, best_prim, seed;
    float hu, hv;
    bool found;

    TR_DEV void begin(const SceneView& sv, f3 org, f3 dir, float tmin_, float tmax_, uint seed_, int* lds_stack) {
        r = make_ray(org, dir);
        stk.init(lds_stack);
        node = sv.node_count > 0 ? 0 : -1;
        tmin = tmin_; tmax = tmax_; seed = seed_;
        best_t = SHADOW ? 1.0f : tmax_;
        best_inst = 0xFFFFFFFFu; best_prim = 0xFFFFFFFFu; hu = 0; hv = 0; found = false;
    }

    TR_DEV bool step(const SceneView& sv, int* spill, TraceStats& st) {
        if (node >= 0) {
            const BvhNode n = sv.nodes[node];
            if (COUNT) st.nodes++;
            float t0, t1;
            const float far = SHADOW ? tmax : best_t;
            bool h0 = box_intersect(r, n.lo0, n.hi0, tmin, far, t0);
            bool h1 = box_intersect(r, n.lo1, n.hi1, tmin, far, t1);
            if (h0 && h1) {
                bool first0 = SHADOW ? true : (t0 <= t1);
                stk.push(spill, first0 ? n.child1 : n.child0);
                if (COUNT) st.maxsp = max(st.maxsp, (uint)stk.sp);
                node = first0 ? n.child0 : n.child1;
                return false;
            } else if (h0) { node = n.child0; return false; }
            else if (h1) { node = n.child1; return false; }
        } else {
            const TriRecord tr = sv.tris[~node];
            if (COUNT) st.tris++;
            float t, bu, bv;
            f3 v0 = F3(tr.v0[0], tr.v0[1], tr.v0[2]), v1 = F3(tr.v1[0], tr.v1[1], tr.v1[2]), v2 = F3(tr.v2[0], tr.v2[1], tr.v2[2]);
            if (tri_intersect(r, v0, v1, v2, tmin, SHADOW ? tmax : __builtin_huge_valf(), t, bu, bv)) {
                const uint inst = tr.inst_flags & 0x7FFFFFFFu;
                if (SHADOW) {
                    if (!(tr.inst_flags & 0x80000000u)) { best_t = 0.0f; return true; }
                    if (COUNT) st.alpha++;
                    best_t *= 1.0f - candidate_alpha(sv, (int)inst, (int)tr.prim, bu, bv);
                    if (best_t == 0.0f) return true;
                } else {
                    const bool closer = t < best_t || (t == best_t && found && (inst < best_inst || (inst == best_inst && tr.prim < best_prim)));
                    if (closer && t < tmax) {
                        bool accept = true;
                        if (tr.inst_flags & 0x80000000u) {
                            if (COUNT) st.alpha++;
                            float a = candidate_alpha(sv, (int)inst, (int)tr.prim, bu, bv);
                            float cutoff = ALPHA_MODE == 0 ? alpha_cutoff_hash(seed, (int)inst, (int)tr.prim) : 0.0001f;
                            accept = !(a <= cutoff);
                        }
                        if (accept) { best_t = t; found = true; best_inst = inst; best_prim = tr.prim; hu = bu; hv = bv; }
                    }
                }
            }
        }
        if (stk.sp == 0) return true;
        node = stk.pop(spill);
        return false;
    }
};

// sphere lights + hit record (rt_common_point_light.rint/.rchit), shared by the stepwise traversals
template <class TRAV>
TR_DEV void finish_closest_hit(const SceneView& sv, TRAV& tv, f3 org, f3 dir, bool include_lights, HitRecord& hit) {
    hit.instance_id = tv.found ? (int)tv.best_inst : -1; hit.primitive_id = tv.found ? (int)tv.best_prim : -1; hit.u = tv.hu; hit.v = tv.hv;
    if (include_lights) {
        for (uint i = 0; i < sv.point_light_count; ++i) {
            const PointLight& pl = sv.point_lights[i];
            float radius = pl.radius;
            if (radius == 0.0f) continue;
            f3 oc = org - pl.pos;
            float a = dot(dir, dir);
            float b = 2.0f * dot(oc, dir);
            float c = dot(oc, oc) - radius * radius;
            float disc = b * b - 4.0f * a * c;
            if (disc < 0) continue;
            float h = (-b - sqrtf(disc)) / (2.0f * a);
            if (h > 0 && h > tv.tmin && h < tv.best_t) {
                tv.best_t = h; tv.found = true;
                hit.instance_id = -1; hit.primitive_id = (int)i; hit.u = h; hit.v = 0;
            }
        }
    }
    hit.t = tv.found ? tv.best_t : -1.0f;
}

// LDS words per block for the per-lane stacks of either traversal
#define TR_STACK_WORDS(WIDE) ((WIDE) ? 2 * TR_LDS_STACK8 * TR_BLOCK : TR_LDS_STACK * TR_BLOCK)

template <int ALPHA_MODE, bool COUNT, bool WIDE>
TR_DEV void trace_closest_any(const SceneView& sv, f3 org, f3 dir, float tmin, float tmax, bool include_lights, uint seed,
                              int* lds_stack, HitRecord& hit, TraceStats& st, int& overflow) {
    if (WIDE) trace_closest8<ALPHA_MODE, COUNT>(sv, org, dir, tmin, tmax, include_lights, seed, lds_stack, hit, st, overflow);
    else trace_closest<ALPHA_MODE, COUNT>(sv, org, dir, tmin, tmax, include_lights, seed, lds_stack, hit, st, overflow);
}
template <bool COUNT, bool WIDE>
TR_DEV float trace_shadow_any(const SceneView& sv, f3 org, f3 dir, float tmin, float tmax, int* lds_stack, TraceStats& st, int& overflow) {
    if (WIDE) return trace_shadow8<COUNT>(sv, org, dir, tmin, tmax, lds_stack, st, overflow);
    return trace_shadow<COUNT>(sv, org, dir, tmin, tmax, lds_stack, st, overflow);
}

}  // namespace tr
