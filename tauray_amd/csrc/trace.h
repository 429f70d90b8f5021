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
#include "trace_timeline.h"

namespace tr {

#ifndef TR_BLOCK
#define TR_BLOCK 256
#endif
#ifndef TR_VOTE
#define TR_VOTE 32         // closest-hit traversal: lanes holding a leaf wait until this many do - or half of the live lanes, which
                           // with 32 is always the smaller number (0 disables the vote).  8 before the quad tail took over the
                           // thin end of a wave, 16 after it; re-measured on the static-build tree: 12 / 16 / 24 / 32 = 3.89 / 3.85 /
                           // 3.81 / 3.79 ms per frame, and 3 / 5 / 6 eighths of the live lanes instead of half: 3.84 / 3.84 / 3.88
#endif
#ifndef TR_VOTE_SHADOW_WAVE
#define TR_VOTE_SHADOW_WAVE 16   // the same vote in the per-lane loop of trace_shadow_wave4 (0: none); 8 / 16 measured: -1 ... -2 % shadow time
#endif
// The slab test compares a box's entry distance with min(exit distance, current closest hit / tmax) * TR_SLAB_PAD.  The pad
// covers the rounding of both slab distances and, on the tmax side, the error of the distance the triangle test computes
// (a different expression): without it a box can be culled against a hit that one of its own triangles would have beaten
// by a few ulps, and which of two nearly coincident surfaces wins then depends on the shape of the tree.
#ifndef TR_SLAB_PAD
#define TR_SLAB_PAD 1.000004f
#endif
#define TR_LDS_STACK 16
#ifndef TR_SPILL_STACK
#define TR_SPILL_STACK 112
#endif

struct RayPre {
    f3 org, dir, inv_dir;
    int kx, ky, kz;
    float Sx, Sy, Sz;
    uint nox, noy, noz;   // byte offsets of the near x / y / z planes inside a Bvh4Node (far plane: offset ^ 16)
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
    r.nox = (__float_as_uint(r.inv_dir.x) >> 31) << 4;
    r.noy = ((__float_as_uint(r.inv_dir.y) >> 31) << 4) | 32u;
    r.noz = ((__float_as_uint(r.inv_dir.z) >> 31) << 4) | 64u;
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

// Per-lane traversal stack: first TR_LDS_STACK entries in LDS, the rest in a private spill array.
// `sp` must stay in a VGPR and the LDS access must stay a ds_read/ds_write: the spill array is therefore a separate
// local (a struct member array drags the whole struct, sp included, into scratch) and pop() reads LDS
// unconditionally (an if/else over the two memories is if-converted into a generic pointer + flat_load).
struct LaneStack {
    int* lds;           // &stack[0][lane_in_block]; stride TR_BLOCK
    int sp;
    int overflow;       // count of dropped pushes (an int in a VGPR, not a wave-level predicate)
    TR_DEV void init(int* base) { lds = base; sp = 0; overflow = 0; }
    // The common case costs a compare, an address and the ds_write.  Past the end of the spill array pushes land on its last
    // slot and are counted: the traversal still terminates (sp stays balanced) and the frame is reported as invalid.
    TR_DEV void push(int* spill, int v) {
        if (sp < TR_LDS_STACK) lds[sp * TR_BLOCK] = v;
        else {
            const int k = sp - TR_LDS_STACK;
            if (k >= TR_SPILL_STACK) overflow++;
            spill[k < TR_SPILL_STACK ? k : TR_SPILL_STACK - 1] = v;
        }
        sp++;
    }
    TR_DEV int pop(const int* spill) {
        sp--;
        int v = lds[(sp < TR_LDS_STACK ? sp : 0) * TR_BLOCK];
        asm volatile("" : "+v"(v));   // pin the ds_read: no select-of-pointers + flat_load
        if (sp >= TR_LDS_STACK) v = spill[sp - TR_LDS_STACK < TR_SPILL_STACK ? sp - TR_LDS_STACK : TR_SPILL_STACK - 1];
        return v;
    }
};

struct TraceStats {
    uint nodes, tris, alpha, maxsp;
    // divergence statistics of the closest-hit loop (counting builds only), kept by the first active lane of a wave: node / triangle
    // phases executed, node phases that ran with at most 16 / 8 active rays, and the ray visits those phases served
    uint ph_node, ph_tri, ph_node16, ph_node8, lv_node16;
    uint ph_qnode, ph_qtri;   // phases of the quad-cooperative tail (trace_quad.h)
    uint ph_hist[8];          // per-lane node phases by live rays: 1-8, 9-16, ..., 57-64
    uint cnodes;              // node visits of closest-hit rays alone (`nodes` also counts the shadow rays of a fused launch)
};

// Candidate alpha of a non-opaque triangle: albedo_factor.a * texture alpha at the uv of the candidate hit
// (get_interpolated_vertex_light, shader/rt.glsl:103-117; shader/rt_common.rahit:15-24).  `word` = TriRecord::alpha: the alpha itself for
// an untextured material, else the triangle's AlphaTri record - one 32-byte fetch where the instance, its span, three indices and three
// vertices used to be fetched one behind the other (same values, same expressions: common.h, bvh_build.hip alpha_word).
TR_DEV float candidate_alpha(const SceneView& sv, uint word, float bu, float bv) {
    if (!(word & 0x80000000u)) return __uint_as_float(word);
    const AlphaTri a = sv.alpha_tris[word & 0x7FFFFFFFu];
    float alpha = a.factor;
    if (a.tex >= 0) {
        const float b0 = 1.0f - bu - bv;
        alpha *= sample_texture_alpha(sv, a.tex, a.uv0 * b0 + a.uv1 * bu + a.uv2 * bv);
    }
    return alpha;
}

// Traversal-order independent stand-in for generate_single_uniform_random(payload.random_seed)
// (shader/rt_common.rahit:21); see DESIGN.md "any-hit order".
TR_DEV float alpha_cutoff_hash(uint seed, int instance_id, int primitive_id) {
    uint k = (uint)instance_id * 0x9E3779B9u + (uint)primitive_id;
    uint h = seed ^ pcg(k);
    return (float)pcg(h) * 2.3283064365386963e-10f;
}

// =====================================================================================================================
// 4-wide fp32 BVH: half the dependent node fetches of the binary tree for about the same box-test arithmetic.
struct Hit4 { float t[4]; int c[4]; };

// The ray's direction signs pick the near and the far plane of every axis at load time (per-lane byte offsets into the
// 128-byte node), so the slab test needs no min / max to order them: per child 6 sub, 6 mul, max + max3, min3 + pad + min,
// one compare.  NaNs (0 * inf: origin on a plane of an axis the ray does not move along) are dropped by min / max, i.e.
// that axis does not constrain the interval.  Empty slots hold an inverted infinite box: their near distance is +inf (or
// their far distance -inf) for every ray, so they never pass and need no test of their own.
TR_DEV void box4_intersect(const RayPre& r, const Bvh4Node* nodes, int node, float tmin, float tmax, Hit4& h TL(, TlPhase* tlp = nullptr)) {
    f4 nxv, fxv, nyv, fyv, nzv, fzv;
    int c0, c1, c2, c3;
    {
        const char* base = reinterpret_cast<const char*>(nodes);
        const uint t = (uint)node << 7;
        uint ax = t | r.nox, ay = t | r.noy, az = t | r.noz;
        // opaque to the optimiser: once it splits the known +32 / +64 out of ay / az as immediate offsets, it addresses the far
        // planes with 64-bit adds instead of the base + 32-bit offset form
        asm volatile("" : "+v"(ay), "+v"(az));
        nxv = *reinterpret_cast<const f4*>(base + (size_t)ax); fxv = *reinterpret_cast<const f4*>(base + (size_t)(ax ^ 16u));
        nyv = *reinterpret_cast<const f4*>(base + (size_t)ay); fyv = *reinterpret_cast<const f4*>(base + (size_t)(ay ^ 16u));
        nzv = *reinterpret_cast<const f4*>(base + (size_t)az); fzv = *reinterpret_cast<const f4*>(base + (size_t)(az ^ 16u));
        const int4 ch = *reinterpret_cast<const int4*>(base + (size_t)t + 96);
        c0 = ch.x; c1 = ch.y; c2 = ch.z; c3 = ch.w;
    }
    TL(if (tlp) tlp->loads_issued();)
    const float nx[4] = {nxv.x, nxv.y, nxv.z, nxv.w}, ny[4] = {nyv.x, nyv.y, nyv.z, nyv.w}, nz[4] = {nzv.x, nzv.y, nzv.z, nzv.w};
    const float fx[4] = {fxv.x, fxv.y, fxv.z, fxv.w}, fy[4] = {fyv.x, fyv.y, fyv.z, fyv.w}, fz[4] = {fzv.x, fzv.y, fzv.z, fzv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float tx0 = (nx[k] - r.org.x) * r.inv_dir.x, tx1 = (fx[k] - r.org.x) * r.inv_dir.x;
        const float ty0 = (ny[k] - r.org.y) * r.inv_dir.y, ty1 = (fy[k] - r.org.y) * r.inv_dir.y;
        const float tz0 = (nz[k] - r.org.z) * r.inv_dir.z, tz1 = (fz[k] - r.org.z) * r.inv_dir.z;
        const float t0 = fmaxf(fmaxf(tx0, ty0), fmaxf(tz0, tmin));
        const float t1 = fminf(fminf(fminf(tx1, ty1), tz1), tmax) * TR_SLAB_PAD;
        h.t[k] = t0 <= t1 ? t0 : __builtin_huge_valf();
    }
    // keeps the load of the child ids next to the plane loads: left alone, the compiler sinks it into the "some child is hit"
    // branch, one more dependent round trip per node
    asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
    h.c[0] = c0; h.c[1] = c1; h.c[2] = c2; h.c[3] = c3;
}


#define TR_CE4(a, b) { const bool sw = h.t[b] < h.t[a]; const float ta = h.t[a], tb = h.t[b]; const int ca = h.c[a], cb = h.c[b]; \
                       h.t[a] = sw ? tb : ta; h.t[b] = sw ? ta : tb; h.c[a] = sw ? cb : ca; h.c[b] = sw ? ca : cb; }

// Closest hit over triangles (+ sphere lights), one ray per lane.  ALPHA_MODE 0: stochastic alpha keyed by `seed`
// (shader/rt_common.rahit:15-24); 1: fixed cutoff 1e-4 (shader/rt_feature.rahit:17).
template <int ALPHA_MODE, bool COUNT>
TR_DEV void trace_closest4(const SceneView& sv, f3 org, f3 dir, float tmin, float tmax, bool include_lights, uint seed,
                           int* lds_stack, HitRecord& hit, TraceStats& st, int& overflow) {
    hit.instance_id = -1; hit.primitive_id = -1; hit.u = 0; hit.v = 0; hit.t = -1.0f;
    float best_t = tmax;
    bool found = false;
    uint best_inst = 0xFFFFFFFFu, best_prim = 0xFFFFFFFFu;
    RayPre r = make_ray(org, dir);
    const bool finite_ray = ray_is_finite(org, dir);
    if (sv.tri_count > 0 && finite_ray) {
        LaneStack stk;
        int spill[TR_SPILL_STACK];
        stk.init(lds_stack);
        int node = sv.node_count > 0 ? 0 : -1;   // single-triangle scene: leaf ~0 == -1
        while (true) {
#if TR_VOTE > 0
            const bool at_leaf = node < 0;
            const int n_leaf = __popcll(__ballot(at_leaf)), n_all = __popcll(__ballot(true));
            const bool leaf_phase = n_leaf >= TR_VOTE || n_leaf == n_all;
            if (at_leaf != leaf_phase) continue;
#endif
            if (COUNT) {
                const unsigned long long m = __ballot(true), mn = __ballot(node >= 0);
                if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) {
                    if (mn) { st.ph_node++; if (__popcll(m) <= 16) { st.ph_node16++; st.lv_node16 += (uint)__popcll(mn); } if (__popcll(m) <= 8) st.ph_node8++; }
                    else st.ph_tri++;
                }
            }
            if (node >= 0) {
                Hit4 h;
                box4_intersect(r, sv.nodes4, node, tmin, best_t, h);
                if (COUNT) st.nodes++;
                TR_CE4(0, 1) TR_CE4(2, 3) TR_CE4(0, 2) TR_CE4(1, 3) TR_CE4(1, 2)
                if (h.t[0] < __builtin_huge_valf()) {
                    if (h.t[3] < __builtin_huge_valf()) stk.push(spill, h.c[3]);
                    if (h.t[2] < __builtin_huge_valf()) stk.push(spill, h.c[2]);
                    if (h.t[1] < __builtin_huge_valf()) stk.push(spill, h.c[1]);
                    if (COUNT) st.maxsp = max(st.maxsp, (uint)stk.sp);
                    node = h.c[0];
                    continue;
                }
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
                            float a = candidate_alpha(sv, tr.alpha, bu, bv);
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
            if (stk.sp == 0) break;
            node = stk.pop(spill);
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
            float hh = (-b - sqrtf(disc)) / (2.0f * a);
            if (hh > 0 && hh > tmin && hh < best_t) {
                best_t = hh; found = true;
                hit.instance_id = -1; hit.primitive_id = (int)i; hit.u = hh; hit.v = 0;
            }
        }
    }
    hit.t = found ? best_t : -1.0f;
}

// shadow_ray (shader/path_tracer.glsl:35-52) + rt_common_shadow.rahit/.rchit: product of (1 - alpha)
// over non-opaque hits, 0 on the first opaque hit; lights are excluded (mask 0xFD).
template <bool COUNT>
TR_DEV float trace_shadow4(const SceneView& sv, f3 org, f3 dir, float tmin, float tmax, int* lds_stack, TraceStats& st, int& overflow) {
    float visibility = 1.0f;
    if (sv.tri_count == 0 || !ray_is_finite(org, dir)) return visibility;
    RayPre r = make_ray(org, dir);
    LaneStack stk;
    int spill[TR_SPILL_STACK];
    stk.init(lds_stack);
    int node = sv.node_count > 0 ? 0 : -1;   // single-triangle scene: leaf ~0 == -1
    while (true) {
        if (node >= 0) {
            int next = 0x7FFFFFFF;
            {
                Hit4 h;
                box4_intersect(r, sv.nodes4, node, tmin, tmax, h);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (h.t[k] < __builtin_huge_valf()) {
                        if (next == 0x7FFFFFFF) next = h.c[k];
                        else stk.push(spill, h.c[k]);
                    }
                }
            }
            if (COUNT) st.nodes++;
            if (next != 0x7FFFFFFF) { node = next; continue; }
        } else {
            const TriRecord tr = sv.tris[~node];
            if (COUNT) st.tris++;
            float t, bu, bv;
            f3 v0 = F3(tr.v0[0], tr.v0[1], tr.v0[2]), v1 = F3(tr.v1[0], tr.v1[1], tr.v1[2]), v2 = F3(tr.v2[0], tr.v2[1], tr.v2[2]);
            if (tri_intersect(r, v0, v1, v2, tmin, tmax, t, bu, bv)) {
                if (!(tr.inst_flags & 0x80000000u)) { visibility = 0.0f; break; }
                if (COUNT) st.alpha++;
                float alpha = candidate_alpha(sv, tr.alpha, bu, bv);
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

// LDS words per block for the per-lane stacks
#define TR_STACK_WORDS (TR_LDS_STACK * TR_BLOCK)

}  // namespace tr
