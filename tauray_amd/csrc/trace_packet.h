// trace_packet.h - closest hit for a wave of rays that travel together: the primary rays of an 8 x 8 pixel tile (launch_coord deals
// tiles to waves).  The wave walks the tree once, on one stack: a node is visited when any of its rays enters it, every lane tests its
// own ray against the node's four boxes, and the node comes through the scalar cache - one line for the wave instead of a line per
// lane, so a step waits for one fetch that is almost always near instead of for the farthest of thirty.  The
// price is the union: a ray is walked through nodes only its neighbours needed - small for rays a pixel apart.
//
// Hits are the per-lane loop's hits: every triangle a ray could reach lies in a leaf the wave visits, all live lanes test every visited
// triangle with the same test, and a lane keeps the candidate the per-lane loop would keep (nearest t, then lowest instance, then
// lowest primitive; non-opaque candidates pass the same alpha test).  Box tests use the arithmetic of box4_intersect, so a ray enters
// exactly the boxes it enters there; only the order of visits differs, which the tie rule makes irrelevant.
#pragma once
#include "trace.h"

namespace tr {

#ifndef TR_PACKET_STACK
#define TR_PACKET_STACK 96          // entries of the wave's stack (LDS); a 4-wide tree of a million triangles needs about 30
#endif

// `wave_stack`: TR_PACKET_STACK ints of LDS owned by this wave.  Every lane of the wave calls this.
template <int ALPHA_MODE, bool COUNT>
TR_DEV void trace_closest_packet(const SceneView& sv, bool valid, f3 org, f3 dir, float tmin, float tmax, bool include_lights, uint seed,
                                 int* wave_stack, HitRecord& hit, TraceStats& st, int& overflow) {
    hit.instance_id = -1; hit.primitive_id = -1; hit.u = 0; hit.v = 0; hit.t = -1.0f;
    float best_t = tmax, best_u = 0.0f, best_v = 0.0f;
    uint best_inst = 0xFFFFFFFFu, best_prim = 0xFFFFFFFFu;
    const RayPre r = make_ray(org, dir);
    const bool finite_ray = valid && ray_is_finite(org, dir);
    const bool live = finite_ray && sv.tri_count > 0;
    const bool gx = r.nox & 16u, gy = r.noy & 16u, gz = r.noz & 16u;      // near plane = hi
    if (__ballot(live) != 0ull) {
        int sp = 0;
        int node = sv.node_count > 0 ? 0 : -1;      // wave-uniform
        while (true) {
            node = __builtin_amdgcn_readfirstlane(node);
            if (node >= 0) {
                const Bvh4Node& nd = sv.nodes4[node];      // uniform address: scalar loads
                if (COUNT && live) st.nodes++;
                float t0[4];
                unsigned long long any[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // box4_intersect's arithmetic: (near plane - o) * (1 / d) per axis, near / far picked by the direction's sign
                    const float xl = (nd.lox[k] - r.org.x) * r.inv_dir.x, xh = (nd.hix[k] - r.org.x) * r.inv_dir.x;
                    const float yl = (nd.loy[k] - r.org.y) * r.inv_dir.y, yh = (nd.hiy[k] - r.org.y) * r.inv_dir.y;
                    const float zl = (nd.loz[k] - r.org.z) * r.inv_dir.z, zh = (nd.hiz[k] - r.org.z) * r.inv_dir.z;
                    const float tx0 = gx ? xh : xl, tx1 = gx ? xl : xh, ty0 = gy ? yh : yl, ty1 = gy ? yl : yh, tz0 = gz ? zh : zl, tz1 = gz ? zl : zh;
                    const float a = fmaxf(fmaxf(tx0, ty0), fmaxf(tz0, tmin));
                    const float b = fminf(fminf(fminf(tx1, ty1), tz1), best_t) * TR_SLAB_PAD;
                    const bool h = live && a <= b;
                    t0[k] = h ? a : __builtin_huge_valf();
                    any[k] = __ballot(h);
                }
                // order of the children some ray enters: by the entry distances of the first lane that enters anything (its misses
                // last); any order finds the same hits
                const unsigned long long entered = any[0] | any[1] | any[2] | any[3];
                if (entered != 0ull) {
                    const int lead = __ffsll((long long)entered) - 1;
                    float key[4];
                    int ref[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        key[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t0[k]), lead));
                        ref[k] = nd.child[k];
                        if (any[k] == 0ull) { key[k] = __builtin_huge_valf(); ref[k] = 0x7FFFFFFF; }      // nobody enters: not visited
                        else if (!(key[k] < __builtin_huge_valf())) key[k] = 3.0e38f;                       // entered by others only: after the lead's own
                    }
#define TR_PCE(a, b) { const bool sw = key[b] < key[a]; const float ka = key[a], kb = key[b]; const int ra = ref[a], rb = ref[b]; \
                       key[a] = sw ? kb : ka; key[b] = sw ? ka : kb; ref[a] = sw ? rb : ra; ref[b] = sw ? ra : rb; }
                    TR_PCE(0, 1) TR_PCE(2, 3) TR_PCE(0, 2) TR_PCE(1, 3) TR_PCE(1, 2)
#undef TR_PCE
                    // far ones onto the stack (uniform values, every lane writes the same word), nearest next
#pragma unroll
                    for (int k = 3; k >= 1; --k)
                        if (ref[k] != 0x7FFFFFFF) {
                            if (sp < TR_PACKET_STACK) wave_stack[sp] = ref[k]; else overflow++;
                            sp += sp < TR_PACKET_STACK ? 1 : 0;
                        }
                    node = ref[0];
                    continue;
                }
            } else {
                const TriRecord& tr = sv.tris[~node];      // uniform address
                if (COUNT && live) st.tris++;
                float t, bu, bv;
                const f3 v0 = F3(tr.v0[0], tr.v0[1], tr.v0[2]), v1 = F3(tr.v1[0], tr.v1[1], tr.v1[2]), v2 = F3(tr.v2[0], tr.v2[1], tr.v2[2]);
                if (live && tri_intersect(r, v0, v1, v2, tmin, __builtin_huge_valf(), t, bu, bv)) {
                    const uint inst = tr.inst_flags & 0x7FFFFFFFu, prim = tr.prim;
                    const bool closer = t < best_t || (t == best_t && best_inst != 0xFFFFFFFFu && (inst < best_inst || (inst == best_inst && prim < best_prim)));
                    if (closer && t < tmax) {
                        bool accept = true;
                        if (tr.inst_flags & 0x80000000u) {
                            if (COUNT) st.alpha++;
                            const float a = candidate_alpha(sv, (int)inst, (int)prim, bu, bv);
                            const float cutoff = ALPHA_MODE == 0 ? alpha_cutoff_hash(seed, (int)inst, (int)prim) : 0.0001f;
                            accept = !(a <= cutoff);
                        }
                        if (accept) { best_t = t; best_inst = inst; best_prim = prim; best_u = bu; best_v = bv; }
                    }
                }
            }
            if (sp == 0) break;
            --sp;
            node = wave_stack[sp];
        }
    }
    bool found = best_inst != 0xFFFFFFFFu;
    if (found) { hit.instance_id = (int)best_inst; hit.primitive_id = (int)best_prim; hit.u = best_u; hit.v = best_v; }
    // sphere lights, as trace_closest4 (shader/rt_common.rint: nearest root in front of the origin)
    if (include_lights && finite_ray) {
        for (uint i = 0; i < sv.point_light_count; ++i) {
            const PointLight& pl = sv.point_lights[i];
            const float radius = pl.radius;
            if (radius == 0.0f) continue;
            const f3 oc = org - pl.pos;
            const float a = dot(dir, dir);
            const float b = 2.0f * dot(oc, dir);
            const float c = dot(oc, oc) - radius * radius;
            const float disc = b * b - 4.0f * a * c;
            if (disc < 0) continue;
            const float hh = (-b - sqrtf(disc)) / (2.0f * a);
            if (hh > 0 && hh > tmin && hh < best_t) {
                best_t = hh; found = true;
                hit.instance_id = -1; hit.primitive_id = (int)i; hit.u = hh; hit.v = 0;
            }
        }
    }
    hit.t = found ? best_t : -1.0f;
}

}  // namespace tr
