// Experiment (round 5; -DTR_RAYS2=1, a variant library only; profiles/r5/two_rays_per_lane_ab.txt): two rays per lane.
//
// The closest-hit waves wait for their node fetches about two fifths of their time at six waves per SIMD (80 registers); the review's
// lever: let a lane carry two rays and fetch the nodes of both before it waits for either - four waves of 128 registers hold eight
// fetches in flight per lane and SIMD where six waves of 80 hold six.  A chunk is 128 queue slots (slot A = lane, slot B = lane + 64);
// a node phase loads the nodes of both slots (idle slots load the root: no branch between the two groups of loads, so the scheduler
// keeps them together), then runs the slab tests, the sort and the stack work of each; a triangle phase does the same with the records.
// No quad tail here: the comparison is against the one-ray kernel without its tail (-DTR_QUAD_SWITCH=0).  Same arithmetic, same
// (t, instance, primitive) order: same hits.
#pragma once
#include "trace_lanes.h"

namespace tr {

namespace {

struct RaySlot {
    RayPre r;
    float tmin, best_t, best_u, best_v;
    uint best_inst, best_prim, seed;
    LaneStack stk;
    int node;
    bool live, finite;
};

template <bool COUNT>
TR_DEV void slot_consider(const SceneView& sv, RaySlot& s, const TriHit& tr, TraceStats& st) {
    const uint inst = tr.inst_flags & 0x7FFFFFFFu;
    const bool closer = tr.t < s.best_t || (tr.t == s.best_t && s.best_inst != 0xFFFFFFFFu && (inst < s.best_inst || (inst == s.best_inst && tr.prim < s.best_prim)));
    if (closer) {
        bool accept = true;
        if (tr.inst_flags & 0x80000000u) {
            if (COUNT) st.alpha++;
            const float a = candidate_alpha(sv, tr.alpha, tr.bu, tr.bv);
            accept = !(a <= alpha_cutoff_hash(s.seed, (int)inst, (int)tr.prim));
        }
        if (accept) { s.best_t = tr.t; s.best_inst = inst; s.best_prim = tr.prim; s.best_u = tr.bu; s.best_v = tr.bv; }
    }
}

TR_DEV void slot_after_node(RaySlot& s, int* spill, Hit4& h) {
    TR_CE4(0, 1) TR_CE4(2, 3) TR_CE4(0, 2) TR_CE4(1, 3) TR_CE4(1, 2)
    if (h.t[0] < __builtin_huge_valf()) {
        const int m = (int)(h.t[1] < __builtin_huge_valf()) + (int)(h.t[2] < __builtin_huge_valf()) + (int)(h.t[3] < __builtin_huge_valf());
        s.stk.push_sorted(spill, m, h.c[1], h.c[2], h.c[3]);
        s.node = h.c[0];
    } else if (s.stk.sp == 0) s.live = false;
    else s.node = s.stk.pop(spill);
}

// The closest-hit rays of queue slots base .. base + 127, one wave.
template <bool COUNT>
TR_DEV void closest_lane2(const SceneView& sv, const PtParams& P, const PathBuffers& pb, int bounce, const uint* queue, uint base, uint n, int* lds_a, int* lds_b,
                          TraceStats& st, int& overflow, uint& rays) {
    const uint lane = threadIdx.x & 63u;
    RaySlot S[2];
    uint ids[2];
    bool valid[2];
    f3 org[2], dir[2];
    int spill_a[TR_SPILL_STACK], spill_b[TR_SPILL_STACK];
    const float tmin = bounce == 0 ? 0.0f : P.opt.min_ray_dist;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint qi = base + lane + 64u * (uint)k;
        valid[k] = qi < n;
#ifdef TR_RAYS2_DEBUG_ONE_SLOT      // bisecting: only slot TR_RAYS2_DEBUG_ONE_SLOT carries rays (chunks of 64)
        valid[k] = k == TR_RAYS2_DEBUG_ONE_SLOT && base + lane < n;
#endif
        ids[k] = 0;
        u4 misc = {0, 0, 0, 1};
        f4 o = F4(0), d = F4(0, 0, 1, 0);
#ifdef TR_RAYS2_DEBUG_ONE_SLOT
        const uint qj = base + lane;
#else
        const uint qj = qi;
#endif
        if (valid[k]) { ids[k] = queue ? queue[qj] : qj + P.id_offset; misc = pb.misc[ids[k]]; o = pb.org_pdf[ids[k]]; d = pb.dir_reg[ids[k]]; valid[k] = !(misc.w & 1u); }
        for (int b = 0; b < bounce; ++b) pcg(misc.x);
        org[k] = F3(o); dir[k] = F3(d);
        RaySlot& s = S[k];
        s.r = make_ray(org[k], dir[k]);
        s.tmin = tmin; s.best_t = __builtin_huge_valf(); s.best_u = 0; s.best_v = 0; s.best_inst = 0xFFFFFFFFu; s.best_prim = 0xFFFFFFFFu; s.seed = misc.x;
        s.stk.init(k == 0 ? lds_a : lds_b);
        s.node = 0;
        s.finite = valid[k] && ray_is_finite(org[k], dir[k]);
        s.live = s.finite && sv.tri_count > 0 && sv.node_count > 0;
    }
    while (true) {
        const bool n0 = S[0].live && S[0].node >= 0, n1 = S[1].live && S[1].node >= 0;
        const bool l0 = S[0].live && S[0].node < 0, l1 = S[1].live && S[1].node < 0;
        const int n_node = __popcll(__ballot(n0)) + __popcll(__ballot(n1)), n_leaf = __popcll(__ballot(l0)) + __popcll(__ballot(l1));
        if (n_node + n_leaf == 0) break;
        const bool tri_phase = n_node == 0 || n_leaf >= 2 * TR_VOTE || n_leaf >= n_node;
        if (!tri_phase) {
            Hit4 h0, h1;
            // both fetches are issued before either is waited for; a slot without a node to visit fetches the root
            Node4Data d0, d1;
            box4_load(S[0].r, sv.nodes4, n0 ? S[0].node : 0, d0);
            box4_load(S[1].r, sv.nodes4, n1 ? S[1].node : 0, d1);
            box4_test(S[0].r, d0, S[0].tmin, S[0].best_t, h0);
            box4_test(S[1].r, d1, S[1].tmin, S[1].best_t, h1);
            if (COUNT) st.nodes += (uint)n0 + (uint)n1;
            if (n0) slot_after_node(S[0], spill_a, h0);
            if (n1) slot_after_node(S[1], spill_b, h1);
        } else {
            TriHit t0, t1;
            const bool hit0 = tri_intersect(S[0].r, sv.tris, l0 ? (uint)~S[0].node : 0u, S[0].tmin, __builtin_huge_valf(), t0);
            const bool hit1 = tri_intersect(S[1].r, sv.tris, l1 ? (uint)~S[1].node : 0u, S[1].tmin, __builtin_huge_valf(), t1);
            if (COUNT) st.tris += (uint)l0 + (uint)l1;
            if (l0) { if (hit0) slot_consider<COUNT>(sv, S[0], t0, st); if (S[0].stk.sp == 0) S[0].live = false; else S[0].node = S[0].stk.pop(spill_a); }
            if (l1) { if (hit1) slot_consider<COUNT>(sv, S[1], t1, st); if (S[1].stk.sp == 0) S[1].live = false; else S[1].node = S[1].stk.pop(spill_b); }
        }
    }
    const bool include_lights = !(P.opt.hide_lights && bounce == 0);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        RaySlot& s = S[k];
        overflow += s.stk.overflow ? 1 : 0;
        if (!valid[k]) continue;
        HitRecord hit;
        hit.instance_id = -1; hit.primitive_id = -1; hit.u = 0; hit.v = 0;
        bool found = s.best_inst != 0xFFFFFFFFu;
        float best_t = s.best_t;
        if (found) { hit.instance_id = (int)s.best_inst; hit.primitive_id = (int)s.best_prim; hit.u = s.best_u; hit.v = s.best_v; }
        if (include_lights && s.finite) {
            for (uint i = 0; i < sv.point_light_count; ++i) {
                const PointLight& pl = sv.point_lights[i];
                float radius = pl.radius;
                if (radius == 0.0f) continue;
                f3 oc = org[k] - pl.pos;
                float a = dot(dir[k], dir[k]);
                float b = 2.0f * dot(oc, dir[k]);
                float c = dot(oc, oc) - radius * radius;
                float disc = b * b - 4.0f * a * c;
                if (disc < 0) continue;
                float hh = (-b - sqrtf(disc)) / (2.0f * a);
                if (hh > 0 && hh > tmin && hh < best_t) { best_t = hh; found = true; hit.instance_id = -1; hit.primitive_id = (int)i; hit.u = hh; hit.v = 0; }
            }
        }
        pb.hit[ids[k]] = make_int4(hit.instance_id, hit.primitive_id, __float_as_int(hit.u), __float_as_int(hit.v));
        rays++;
    }
}

}  // namespace

}  // namespace tr
