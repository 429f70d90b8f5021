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

// A ray as the traversal keeps it: in the frame the watertight triangle test works in - components in the order (kx, ky, kz), kz the axis
// the direction is largest along (Woop et al. 2013).  Round 5: the slab test does not care in which order it takes the axes (its
// near / far planes are picked by per-lane byte offsets anyway), and the triangle test picks the vertex components it wants with explicit
// selects on 4 kx / 4 ky / 4 kz (tri_intersect) - a component picked out of a vector by a run-time index is lowered to exec-mask branches:
// nine such picks per triangle test were 150 of its 335 instructions
// (profiles/r5/trace_phase_timeline.txt: a triangle phase spent 2 400 clocks there, twice a node phase's slab tests and sort).
struct RayPre {
    f3 op;                // origin in the order (kx, ky, kz)
    f3 ip;                // 1 / direction, same order
    float Sx, Sy;         // shear constants of the triangle test; the third one, 1 / direction[kz], is ip.z
    uint nkx, nky, nkz;   // byte offsets of the near planes of the axes kx / ky / kz inside a Bvh4Node: 32 k + 16 (direction[k] < 0); far plane: offset ^ 16
};
TR_DEV uint tri_component_offset(uint nk) { return (nk >> 3) & 12u; }     // 4 k: byte offset of component k of a vertex

TR_DEV RayPre make_ray(f3 org, f3 dir) {
    RayPre r;
    float ax = fabsf(dir.x), ay = fabsf(dir.y), az = fabsf(dir.z);
    int kz = (ax > ay) ? (ax > az ? 0 : 2) : (ay > az ? 1 : 2);
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    if (comp(dir, kz) < 0.0f) { int t = kx; kx = ky; ky = t; }
    const float dkx = comp(dir, kx), dky = comp(dir, ky), dkz = comp(dir, kz);
    r.Sx = dkx / dkz;
    r.Sy = dky / dkz;
    r.op = F3(comp(org, kx), comp(org, ky), comp(org, kz));
    r.ip = F3(1.0f / dkx, 1.0f / dky, 1.0f / dkz);
    r.nkx = ((uint)kx << 5) | ((__float_as_uint(r.ip.x) >> 31) << 4);
    r.nky = ((uint)ky << 5) | ((__float_as_uint(r.ip.y) >> 31) << 4);
    r.nkz = ((uint)kz << 5) | ((__float_as_uint(r.ip.z) >> 31) << 4);
    return r;
}

TR_DEV bool ray_is_finite(f3 o, f3 d) {
    return isfinite(o.x) && isfinite(o.y) && isfinite(o.z) && isfinite(d.x) && isfinite(d.y) && isfinite(d.z) &&
           (d.x != 0.0f || d.y != 0.0f || d.z != 0.0f);
}

// Experiment -DTR_PAIR_LEAVES=1 (bvh_build.hip pair_leaves_postpass): a leaf reference ~(index | 1 << 30) names two records, index and index + 1.
#ifndef TR_PAIR_LEAVES
#define TR_PAIR_LEAVES 0
#endif
TR_DEV uint leaf_index(uint complemented_ref) { return TR_PAIR_LEAVES ? (complemented_ref & 0x3FFFFFFFu) : complemented_ref; }
TR_DEV bool leaf_is_pair(uint complemented_ref) { return TR_PAIR_LEAVES && (complemented_ref & 0x40000000u) != 0; }
// in front of the `if (tri_intersect(... leaf_index(ref) + member ...))` of a leaf: once, or once per member of a pair
#if TR_PAIR_LEAVES
#define TR_LEAF_MEMBERS(ref, more) for (uint member = 0; member <= (leaf_is_pair(ref) ? 1u : 0u) && (more); ++member)
#else
#define TR_LEAF_MEMBERS(ref, more) constexpr uint member = 0;
#endif

// What a triangle test hands back: the candidate and the words of the record behind the vertices.
struct TriHit { float t, bu, bv; uint inst_flags, prim, alpha; };

// Watertight test (Woop, Benthin, Wald 2013), no culling, of triangle record `index`.  Plain IEEE fp32 without
// contraction: the shared-edge guarantee needs both products of each edge function
// rounded, and the CPU oracle evaluates exactly the same expression tree.  The nine vertex components are taken in the ray's order.
// Wave priority around the fetch of a phase (an experiment, -DTR_SETPRIO=1: low while a phase issues its loads, high from there to the
// next phase's loads; =2: the other way round; profiles/r5/setprio_ab.txt).  Off: no instruction.
#ifndef TR_SETPRIO
#define TR_SETPRIO 0
#endif
#if TR_SETPRIO == 1
#define TR_PRIO_BEGIN() __builtin_amdgcn_s_setprio(0)
#define TR_PRIO_ISSUED() __builtin_amdgcn_s_setprio(2)
#elif TR_SETPRIO == 2
#define TR_PRIO_BEGIN() __builtin_amdgcn_s_setprio(2)
#define TR_PRIO_ISSUED() __builtin_amdgcn_s_setprio(0)
#else
#define TR_PRIO_BEGIN()
#define TR_PRIO_ISSUED()
#endif

TR_DEV bool tri_intersect(const RayPre& r, const TriRecord* tris, uint index, float tmin, float tmax, TriHit& o TL(, TlPhase* tlp = nullptr)) {
#pragma clang fp contract(off)
    TR_PRIO_BEGIN();
#ifndef TR_TRI_FETCH_DWORDS
    // The record is three 16-byte loads - what a lane's L1 pays for is accesses, not bytes - and every component is picked out of the three
    // registers of its vertex with two selects (18 selects per test).  Round 5 measured both ways of not branching over the axes
    // (profiles/r5/triangle_fetch_ab.txt): nine 4-byte loads at per-lane offsets 4 kx / 4 ky / 4 kz cost no vector instruction at all and ten
    // L1 accesses instead of three; the selects win by 4 % of the closest-hit kernel - the vector L1 is the busiest unit of these kernels
    // (0.65 of its access rate), the VALUs are not (0.43).
    const f4* p = reinterpret_cast<const f4*>(reinterpret_cast<const char*>(tris) + (size_t)index * (uint)sizeof(TriRecord));
    const f4 q0 = p[0], q1 = p[1], q2 = p[2];       // x0 y0 z0 x1 | y1 z1 x2 y2 | z2 inst prim alpha
    TR_PRIO_ISSUED();
    o.inst_flags = __float_as_uint(q2.y); o.prim = __float_as_uint(q2.z); o.alpha = __float_as_uint(q2.w);
    const uint kx4 = tri_component_offset(r.nkx), ky4 = tri_component_offset(r.nky), kz4 = tri_component_offset(r.nkz);
    auto pick = [](float x, float y, float z, uint k4) { const float xy = k4 == 4u ? y : x; return k4 == 8u ? z : xy; };
    const float v0x = pick(q0.x, q0.y, q0.z, kx4), v1x = pick(q0.w, q1.x, q1.y, kx4), v2x = pick(q1.z, q1.w, q2.x, kx4);
    const float v0y = pick(q0.x, q0.y, q0.z, ky4), v1y = pick(q0.w, q1.x, q1.y, ky4), v2y = pick(q1.z, q1.w, q2.x, ky4);
    const float v0z = pick(q0.x, q0.y, q0.z, kz4), v1z = pick(q0.w, q1.x, q1.y, kz4), v2z = pick(q1.z, q1.w, q2.x, kz4);
#else
    // variant for the A/B: nine 4-byte loads at per-lane offsets, no select
    const char* base = reinterpret_cast<const char*>(tris);
    const uint rec = index * (uint)sizeof(TriRecord);
    const uint ox = rec + tri_component_offset(r.nkx), oy = rec + tri_component_offset(r.nky), oz = rec + tri_component_offset(r.nkz);
    const float v0x = *reinterpret_cast<const float*>(base + (size_t)ox), v1x = *reinterpret_cast<const float*>(base + (size_t)ox + 12), v2x = *reinterpret_cast<const float*>(base + (size_t)ox + 24);
    const float v0y = *reinterpret_cast<const float*>(base + (size_t)oy), v1y = *reinterpret_cast<const float*>(base + (size_t)oy + 12), v2y = *reinterpret_cast<const float*>(base + (size_t)oy + 24);
    const float v0z = *reinterpret_cast<const float*>(base + (size_t)oz), v1z = *reinterpret_cast<const float*>(base + (size_t)oz + 12), v2z = *reinterpret_cast<const float*>(base + (size_t)oz + 24);
    const uint* tail = reinterpret_cast<const uint*>(base + (size_t)rec + 36);
    o.inst_flags = tail[0]; o.prim = tail[1]; o.alpha = tail[2];
#endif
    TL(if (tlp) tlp->loads_issued();)
    const float Akz = v0z - r.op.z, Bkz = v1z - r.op.z, Ckz = v2z - r.op.z;
    const float Ax = (v0x - r.op.x) - r.Sx * Akz, Ay = (v0y - r.op.y) - r.Sy * Akz;
    const float Bx = (v1x - r.op.x) - r.Sx * Bkz, By = (v1y - r.op.y) - r.Sy * Bkz;
    const float Cx = (v2x - r.op.x) - r.Sx * Ckz, Cy = (v2y - r.op.y) - r.Sy * Ckz;
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
    const float Az = r.ip.z * Akz, Bz = r.ip.z * Bkz, Cz = r.ip.z * Ckz;
    const float T = U * Az + V * Bz + W * Cz;
    const float rcp = 1.0f / det;
    const float tt = T * rcp;
    if (!(tt > tmin && tt < tmax)) return false;
    o.t = tt; o.bu = V * rcp; o.bv = W * rcp;
    return true;
}

// Per-lane traversal stack: the top entry in a register, the next TR_LDS_STACK in LDS, the rest in a private spill array.
// `sp` must stay in a VGPR and the LDS access must stay a ds_read/ds_write: the spill array is therefore a separate
// local (a struct member array drags the whole struct, sp included, into scratch) and pop() reads LDS
// unconditionally (an if/else over the two memories is if-converted into a generic pointer + flat_load).
//
// Round 5 (profiles/r5/trace_phase_timeline.txt): a wave gets an instruction issued every 6-7 clocks at six waves per SIMD whatever
// its kind, and the three conditional pushes of a node phase + the pop were ~120 instructions and an LDS round trip - more clocks than
// the four slab tests and the sort.  Now: entry sp - 1 lives in `tos`, rows 0 .. sp - 2 in memory; a node phase stores the old top and
// up to two of the sorted children with three unconditional ds_writes (rows above the new top hold junk, which nothing reads), and a pop
// hands out the register and refills it with a ds_read nobody waits for until the next pop or push.  Row -1 exists (the kernels
// allocate one row more and pass the address of row 0): the store of the "old top" of an empty stack lands there.
struct LaneStack {
    int* lds;           // &stack[0][lane_in_block]; stride TR_BLOCK; row -1 is writable
    int sp;             // entries, the one in `tos` included
    int tos;
    bool overflow;      // a store was dropped (a lane mask in scalar registers: the per-lane counter this used to be held a vector register for nothing)
    TR_DEV void init(int* base) { lds = base; sp = 0; tos = 0; overflow = false; }
    // row r of the memory part (r >= -1).  Past the end of the spill array stores land on its last slot and are counted: the traversal
    // still terminates (sp stays balanced) and the frame is reported as invalid.
    TR_DEV void store_row(int* spill, int r, int v) {
        if (r < TR_LDS_STACK) lds[r * TR_BLOCK] = v;
        else {
            const int k = r - TR_LDS_STACK;
            if (k >= TR_SPILL_STACK) overflow = true;
            spill[k < TR_SPILL_STACK ? k : TR_SPILL_STACK - 1] = v;
        }
    }
    TR_DEV void push(int* spill, int v) {
        store_row(spill, sp - 1, tos);
        tos = v;
        sp++;
    }
    // The hit children of a node phase other than the one the ray descends into, nearest last: `m` of (c3, c2, c1) are valid - the last m.
    TR_DEV void push_sorted(int* spill, int m, int c1, int c2, int c3) {
        if (sp + 2 <= TR_LDS_STACK) {     // rows sp - 1, sp, sp + 1 are LDS rows
            int* row = lds + (sp - 1) * TR_BLOCK;
            row[0] = tos;
            row[TR_BLOCK] = m == 3 ? c3 : c2;
            row[2 * TR_BLOCK] = c2;
            tos = m > 0 ? c1 : tos;
            sp += m;
        } else {
            if (m > 2) push(spill, c3);
            if (m > 1) push(spill, c2);
            if (m > 0) push(spill, c1);
        }
    }
    // ... in slot order, for the any-hit loop (which descends into the first hit child and keeps the others in the order of their slots):
    // `n` valid entries among (a, b, c), packed to the front
    TR_DEV void push_packed(int* spill, int n, int a, int b, int c) {
        if (sp + 2 <= TR_LDS_STACK) {
            // bottom to top: old top, a, b | top = c (n = 3);  old top, a | top = b (n = 2);  old top | top = a (n = 1)
            int* row = lds + (sp - 1) * TR_BLOCK;
            row[0] = tos;
            row[TR_BLOCK] = a;
            row[2 * TR_BLOCK] = b;
            tos = n == 3 ? c : (n == 2 ? b : (n == 1 ? a : tos));
            sp += n;
        } else {
            if (n > 0) push(spill, a);
            if (n > 1) push(spill, b);
            if (n > 2) push(spill, c);
        }
    }
    TR_DEV int pop(const int* spill) {
        const int out = tos;
        sp--;
        const int r = sp - 1;       // the row that becomes the top: -1 (junk, never handed out) when the stack is empty now
        int v = lds[(r < TR_LDS_STACK ? r : 0) * TR_BLOCK];
        asm volatile("" : "+v"(v));   // pin the ds_read: no select-of-pointers + flat_load
        if (r >= TR_LDS_STACK) v = spill[r - TR_LDS_STACK < TR_SPILL_STACK ? r - TR_LDS_STACK : TR_SPILL_STACK - 1];
        tos = v;
        return out;
    }
    // every entry in memory (rows 0 .. sp - 1): what the re-deal to quads reads (trace_quad.h)
    TR_DEV void flush(int* spill) { store_row(spill, sp - 1, tos); }
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
struct Node4Data { f4 nxv, fxv, nyv, fyv, nzv, fzv; int4 ch; };      // near / far planes of the axes kx, ky, kz and the child references
TR_DEV void box4_load(const RayPre& r, const Bvh4Node* nodes, int node, Node4Data& d) {
    const char* base = reinterpret_cast<const char*>(nodes);
    const uint t = (uint)node << 7;
    const uint ax = t | r.nkx, ay = t | r.nky, az = t | r.nkz;
    d.nxv = *reinterpret_cast<const f4*>(base + (size_t)ax); d.fxv = *reinterpret_cast<const f4*>(base + (size_t)(ax ^ 16u));
    d.nyv = *reinterpret_cast<const f4*>(base + (size_t)ay); d.fyv = *reinterpret_cast<const f4*>(base + (size_t)(ay ^ 16u));
    d.nzv = *reinterpret_cast<const f4*>(base + (size_t)az); d.fzv = *reinterpret_cast<const f4*>(base + (size_t)(az ^ 16u));
    d.ch = *reinterpret_cast<const int4*>(base + (size_t)t + 96);
}
TR_DEV void box4_test(const RayPre& r, const Node4Data& d, float tmin, float tmax, Hit4& h) {
    int c0 = d.ch.x, c1 = d.ch.y, c2 = d.ch.z, c3 = d.ch.w;
    const float nx[4] = {d.nxv.x, d.nxv.y, d.nxv.z, d.nxv.w}, ny[4] = {d.nyv.x, d.nyv.y, d.nyv.z, d.nyv.w}, nz[4] = {d.nzv.x, d.nzv.y, d.nzv.z, d.nzv.w};
    const float fx[4] = {d.fxv.x, d.fxv.y, d.fxv.z, d.fxv.w}, fy[4] = {d.fyv.x, d.fyv.y, d.fyv.z, d.fyv.w}, fz[4] = {d.fzv.x, d.fzv.y, d.fzv.z, d.fzv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // the three slabs in the ray's axis order: max / min over them do not depend on the order
        const float tx0 = (nx[k] - r.op.x) * r.ip.x, tx1 = (fx[k] - r.op.x) * r.ip.x;
        const float ty0 = (ny[k] - r.op.y) * r.ip.y, ty1 = (fy[k] - r.op.y) * r.ip.y;
        const float tz0 = (nz[k] - r.op.z) * r.ip.z, tz1 = (fz[k] - r.op.z) * r.ip.z;
        const float t0 = fmaxf(fmaxf(tx0, ty0), fmaxf(tz0, tmin));
        const float t1 = fminf(fminf(fminf(tx1, ty1), tz1), tmax) * TR_SLAB_PAD;
        h.t[k] = t0 <= t1 ? t0 : __builtin_huge_valf();
    }
    // keeps the load of the child ids next to the plane loads: left alone, the compiler sinks it into the "some child is hit"
    // branch, one more dependent round trip per node
    asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
    h.c[0] = c0; h.c[1] = c1; h.c[2] = c2; h.c[3] = c3;
}
TR_DEV void box4_intersect(const RayPre& r, const Bvh4Node* nodes, int node, float tmin, float tmax, Hit4& h TL(, TlPhase* tlp = nullptr)) {
    Node4Data d;
    TR_PRIO_BEGIN();
    box4_load(r, nodes, node, d);
    TR_PRIO_ISSUED();
    TL(if (tlp) tlp->loads_issued();)
    box4_test(r, d, tmin, tmax, h);
}

// What the any-hit loop does with the four slab tests of a node: the first hit child in slot order is where the ray goes next, the other
// hit children go onto the stack in slot order (unordered descent: an occluder anywhere ends the ray).  Returns false without a hit.
TR_DEV bool shadow_descend(const Hit4& h, LaneStack& stk, int* spill, int& node) {
    const bool h0 = h.t[0] < __builtin_huge_valf(), h1 = h.t[1] < __builtin_huge_valf(), h2 = h.t[2] < __builtin_huge_valf(), h3 = h.t[3] < __builtin_huge_valf();
    const int n = (int)h0 + (int)h1 + (int)h2 + (int)h3;
    if (n == 0) return false;
    // registers, not elements of an array: a select between array elements is turned into a load at a selected address, which puts the array into scratch
    int c0 = h.c[0], c1 = h.c[1], c2 = h.c[2], c3 = h.c[3];
    asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
    const int first = h0 ? c0 : (h1 ? c1 : (h2 ? c2 : c3));
    // the hit children behind the first, packed to the front: a = the second hit, b = the third, c = the fourth
    const int a = (h0 && h1) ? c1 : ((h2 && (h0 != h1)) ? c2 : c3);
    const int b = (h0 && h1 && h2) ? c2 : c3;
    stk.push_packed(spill, n - 1, a, b, c3);
    node = first;
    return true;
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
                    // sorted: the hit children come first, so the number of further hits says which of c[1..3] go onto the stack
                    const int m = (int)(h.t[1] < __builtin_huge_valf()) + (int)(h.t[2] < __builtin_huge_valf()) + (int)(h.t[3] < __builtin_huge_valf());
                    stk.push_sorted(spill, m, h.c[1], h.c[2], h.c[3]);
                    if (COUNT) st.maxsp = max(st.maxsp, (uint)stk.sp);
                    node = h.c[0];
                    continue;
                }
            } else {
                TriHit tr;
                if (COUNT) st.tris++;
                TR_LEAF_MEMBERS((uint)~node, true)
                if (tri_intersect(r, sv.tris, leaf_index((uint)~node) + member, tmin, __builtin_huge_valf(), tr)) {
                    const float t = tr.t, bu = tr.bu, bv = tr.bv;
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
        overflow += stk.overflow ? 1 : 0;
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
            Hit4 h;
            box4_intersect(r, sv.nodes4, node, tmin, tmax, h);
            if (COUNT) st.nodes++;
            if (shadow_descend(h, stk, spill, node)) continue;
        } else {
            TriHit tr;
            if (COUNT) st.tris++;
#if TR_PAIR_LEAVES
            bool occluded = false;
            for (uint member = 0; member <= (leaf_is_pair((uint)~node) ? 1u : 0u) && !occluded; ++member)
            if (tri_intersect(r, sv.tris, leaf_index((uint)~node) + member, tmin, tmax, tr)) {
                if (!(tr.inst_flags & 0x80000000u)) { visibility = 0.0f; occluded = true; }
                else {
                    if (COUNT) st.alpha++;
                    float alpha = candidate_alpha(sv, tr.alpha, tr.bu, tr.bv);
                    visibility *= 1.0f - alpha;
                    if (visibility == 0.0f) occluded = true;
                }
            }
            if (occluded) break;
#else
            if (tri_intersect(r, sv.tris, (uint)~node, tmin, tmax, tr)) {
                if (!(tr.inst_flags & 0x80000000u)) { visibility = 0.0f; break; }
                if (COUNT) st.alpha++;
                float alpha = candidate_alpha(sv, tr.alpha, tr.bu, tr.bv);
                visibility *= 1.0f - alpha;
                if (visibility == 0.0f) break;
            }
#endif
        }
        if (stk.sp == 0) break;
        node = stk.pop(spill);
    }
    overflow += stk.overflow ? 1 : 0;
    return visibility;
}

// LDS words per block for the per-lane stacks: TR_LDS_STACK rows and, in front of them, the row a store to "row -1" lands in (LaneStack).
// Kernels declare `__shared__ int s_stack_rows[TR_STACK_WORDS]` and hand out `s_stack_rows + TR_STACK_ROW0`.
#define TR_STACK_ROW0 TR_BLOCK
#define TR_STACK_WORDS ((TR_LDS_STACK + 1) * TR_BLOCK)

}  // namespace tr
