// Wave-level traversal with a quad-cooperative tail (gfx950, wave64).
//
// A wave starts 64 rays together, one per lane (trace.h), and runs as long as its longest ray: on the bench scenes more than
// half of the node phases of the closest-hit loop execute with at most 16 live rays, 3 of 64 lanes busy on average
// (tools/phase_probe.py).  When at most TR_QUAD_SWITCH rays are left the wave re-deals them: every surviving ray gets a quad
// of four lanes, lane q of the quad tests child q of the 4-wide node (one box instead of four, no sorting network: the order
// of the four entry distances comes from three quad-permute DPP reads), hit leaves are tested by the lanes that found them
// (siblings in parallel) and the far children are pushed by their lanes in one LDS write.  A node phase of the tail costs
// about 60 vector instructions instead of 150.  The arithmetic per box and per triangle is the per-lane code's, candidates
// resolve by the same (t, instance, primitive) order, so hits are bit-identical.
//
// The quad keeps using the traversal stack of the lane the ray came from (its LDS column); entries past TR_LDS_STACK live in a
// per-wave slice of a global buffer (QuadCtx::spill) instead of the owner's private scratch.
#pragma once
#include "trace.h"

namespace tr {

#ifndef TR_QUAD_SWITCH
#define TR_QUAD_SWITCH 16      // live rays at which a wave switches to one ray per quad (0 = never)
#endif
#ifndef TR_QUAD_VOTE
#define TR_QUAD_VOTE 4         // lanes holding a triangle at which the quads of a wave run a triangle phase (2 / 4 / 8 / 16 measured)
#endif
#define TR_QSPILL TR_SPILL_STACK   // stack entries per quad beyond the LDS part: the per-lane loop's depth (deepest stack seen on the bench scenes: 26)
static_assert(TR_QUAD_SWITCH <= 16, "a wave has sixteen quads");

#define TR_OWNER_WORDS 16     // LDS words per wave behind QuadCtx::owner_tab

struct QuadCtx {
    int* wave_stack;   // LDS: stack column of lane 0 of this wave; entry e of lane l at [e * TR_BLOCK + l]
    int* owner_tab;    // LDS: 16 words of this wave
    int* spill;        // global: 16 * TR_QSPILL words of this wave
    TL(uint* tl;)      // LDS: TL_WORDS words of this wave (trace_timeline.h)
};

// quad permutes: lane q reads lane (q + k) & 3 of its quad
template <int CTRL> TR_DEV int quad_perm(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
TR_DEV int qrot1(int v) { return quad_perm<0x39>(v); }
TR_DEV int qrot2(int v) { return quad_perm<0x4E>(v); }
TR_DEV int qrot3(int v) { return quad_perm<0x93>(v); }
TR_DEV float qrot1f(float v) { return __int_as_float(qrot1(__float_as_int(v))); }
TR_DEV float qrot2f(float v) { return __int_as_float(qrot2(__float_as_int(v))); }
TR_DEV int bperm(int byte_addr, int v) { return __builtin_amdgcn_ds_bpermute(byte_addr, v); }
TR_DEV float bpermf(int byte_addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v))); }

// This lane's place in its quad, read off the hardware each time it is wanted: kept in a register across the quad loop it was the value the
// allocator spilled, and every node phase of the tail began with a scratch load and a wait for it (three instructions here instead).
TR_DEV int quad_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l & 3;
}

TR_DEV void wave_sync_lds() {   // LDS writes of this wave are visible to its other lanes afterwards (no other wave is involved)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// What the four lanes of a quad share about their ray (same values in all four) plus the lane's own candidate.
#ifndef TR_QUAD_FETCH
#define TR_QUAD_FETCH 0
#endif
struct QuadRay {
    RayPre r;        // the ray as its owner kept it (trace.h)
    float tmin;
#if TR_QUAD_FETCH
    uint off_a, off_b;     // byte offsets inside a node of the two 16-byte chunks this lane fetches for its quad (quad_child_box)
#endif
};

// 4 x 4 transpose inside a quad: lane i comes in with row i in (r0 .. r3) and leaves with column i.  Two butterfly stages of
// quad-permute DPP reads (lane ^ 1, lane ^ 2), four selects + two reads per register pair: 16 vector instructions.
TR_DEV void quad_transpose(int q, float& r0, float& r1, float& r2, float& r3) {
    const bool odd = (q & 1) != 0, hi = (q & 2) != 0;
    auto swap1 = [&](float& a, float& b) {      // with lane ^ 1
        const float t = __int_as_float(quad_perm<0xB1>(__float_as_int(odd ? a : b)));
        a = odd ? t : a; b = odd ? b : t;
    };
    auto swap2 = [&](float& a, float& b) {      // with lane ^ 2
        const float t = __int_as_float(quad_perm<0x4E>(__float_as_int(hi ? a : b)));
        a = hi ? t : a; b = hi ? b : t;
    };
    swap1(r0, r1); swap1(r2, r3);
    swap2(r0, r2); swap2(r1, r3);
}

// Box of child q of `node` against the quad's ray; returns the child reference, `hit` and the entry distance.
// A lane reads the seven words of its child out of seven different 16-byte chunks of the node: 28 L1 accesses per quad and node.
// -DTR_QUAD_FETCH=1 (round 5, measured, not adopted: profiles/r5/quad_fetch_ab.txt) lets the quad fetch the node together instead - lane q
// loads two whole chunks, eight accesses per quad, and two 4 x 4 transposes hand every lane the column of its child: 32 vector
// instructions for 20 accesses.  Closest-hit +2 %, frames +1 %: the tail's phases serve few rays, its L1 accesses are few either way,
// and the 32 instructions sit on the critical path of the last rays of a wave.  (The same trade in the triangle test - selects for
// accesses - won, because those phases run with full waves.)
// -DTR_TAIL_PREFETCH=1 (an experiment of round 6, profiles/r6/tail_prefetch_ab.txt): a quad lane whose child is an inner node that was hit
// touches that child's cache line right away (one dword, `pf`), so that the line is on its way while the quad sorts its hits and updates its
// stack; the value is consumed here, behind the next phase's own loads (older loads return first: no extra wait).
#ifndef TR_TAIL_PREFETCH
#define TR_TAIL_PREFETCH 0
#endif
TR_DEV int quad_child_box(const QuadRay& r, const Bvh4Node* nodes, int node, int q, float tmax, bool& hit, float& t0 TL(, TlPhase* tlp = nullptr), int pf = 0) {
    float nx, fx, ny, fy, nz, fz;
    int c;
    {
        const char* base = reinterpret_cast<const char*>(nodes);
#if TR_QUAD_FETCH
        const uint t = (uint)node << 7;
        const f4 a = *reinterpret_cast<const f4*>(base + (size_t)(t | r.off_a));
        const f4 b = *reinterpret_cast<const f4*>(base + (size_t)(t | r.off_b));
        TL(if (tlp) tlp->loads_issued();)
        float a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w, b0 = b.x, b1 = b.y, b2 = b.z, b3 = b.w;
        quad_transpose(q, a0, a1, a2, a3);      // rows: near x, far x, near y, far y
        quad_transpose(q, b0, b1, b2, b3);      // rows: near z, far z, references, references
        nx = a0; fx = a1; ny = a2; fy = a3; nz = b0; fz = b1;
        c = __float_as_int(b2);
#else
        const uint t = ((uint)node << 7) | ((uint)q << 2);
        const uint ax = t | r.r.nkx, ay = t | r.r.nky, az = t | r.r.nkz;
        nx = *reinterpret_cast<const float*>(base + (size_t)ax); fx = *reinterpret_cast<const float*>(base + (size_t)(ax ^ 16u));
        ny = *reinterpret_cast<const float*>(base + (size_t)ay); fy = *reinterpret_cast<const float*>(base + (size_t)(ay ^ 16u));
        nz = *reinterpret_cast<const float*>(base + (size_t)az); fz = *reinterpret_cast<const float*>(base + (size_t)(az ^ 16u));
        c = *reinterpret_cast<const int*>(base + (size_t)t + 96);
#if TR_TAIL_PREFETCH
        asm volatile("" : : "v"(pf));
#endif
        TL(if (tlp) tlp->loads_issued();)
#endif
    }
    // the arithmetic of box4_intersect for one child
    const float tx0 = (nx - r.r.op.x) * r.r.ip.x, tx1 = (fx - r.r.op.x) * r.r.ip.x;
    const float ty0 = (ny - r.r.op.y) * r.r.ip.y, ty1 = (fy - r.r.op.y) * r.r.ip.y;
    const float tz0 = (nz - r.r.op.z) * r.r.ip.z, tz1 = (fz - r.r.op.z) * r.r.ip.z;
    t0 = fmaxf(fmaxf(tx0, ty0), fmaxf(tz0, r.tmin));
    const float t1 = fminf(fminf(fminf(tx1, ty1), tz1), tmax) * TR_SLAB_PAD;
    hit = t0 <= t1;
#if !TR_QUAD_FETCH
    asm volatile("" : "+v"(c));
#endif
    return c;
}

// The quad's stack: entries below TR_LDS_STACK in the owner lane's LDS column, the rest in the wave's global slice.
struct QuadStack {
    int* lds;        // owner's column
    int* glob;       // this quad's TR_QSPILL words
    int sp;
    int overflow;
    TR_DEV void store(int pos, int v) {
        if (pos < TR_LDS_STACK) lds[pos * TR_BLOCK] = v;
        else {
            const int k = pos - TR_LDS_STACK;
            if (k >= TR_QSPILL) overflow++;
            glob[k < TR_QSPILL ? k : TR_QSPILL - 1] = v;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // read back by the other lanes of the quad
        }
    }
    TR_DEV int load(int pos) {
        int v = lds[(pos < TR_LDS_STACK ? pos : 0) * TR_BLOCK];
        asm volatile("" : "+v"(v));
        if (pos >= TR_LDS_STACK) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const int k = pos - TR_LDS_STACK;
            v = glob[k < TR_QSPILL ? k : TR_QSPILL - 1];
        }
        return v;
    }
};

// Re-deals the live rays of a wave over its quads: ray k (in lane order) goes to quad k.  Every lane of the wave calls this.
// What both trace loops share of a ray travels here: the slab test's and the triangle test's constants, the current node and
// the stack (the quad keeps the owner's LDS column; entries the owner had spilled to private scratch move to the quad's slice
// of the wave's global buffer).
struct QuadDeal {
    int q, qd;            // lane in the quad, quad in the wave
    int my_rank;          // for a live lane: which quad took its ray
    int src, back;        // ds_bpermute addresses: the lane the quad's ray came from / lane 0 of the quad that took this lane's ray
    bool has_ray;         // this quad got a ray
};
TR_DEV QuadDeal quad_deal(unsigned long long act, bool live, const RayPre& r, float tmin, int node, const LaneStack& stk, const int* spill, const QuadCtx& qc,
                          int& overflow, QuadRay& qr, QuadStack& qs, int& qnode) {
    QuadDeal d;
    const int lane = threadIdx.x & 63, n_act = __popcll(act);
    d.q = lane & 3; d.qd = lane >> 2;
    d.my_rank = __popcll(act & ((1ull << lane) - 1ull));
    if (live) {
        qc.owner_tab[d.my_rank] = lane;
        for (int e = TR_LDS_STACK; e < stk.sp; ++e) {
            const int k = e - TR_LDS_STACK;
            if (k < TR_QSPILL) qc.spill[d.my_rank * TR_QSPILL + k] = spill[k < TR_SPILL_STACK ? k : TR_SPILL_STACK - 1];
            else overflow++;
        }
    }
    wave_sync_lds();
    d.has_ray = d.qd < n_act;
    const int owner = d.has_ray ? qc.owner_tab[d.qd] : lane;
    d.src = owner << 2;
    d.back = d.my_rank << 4;
    const int src = d.src;
    qr.r.op = F3(bpermf(src, r.op.x), bpermf(src, r.op.y), bpermf(src, r.op.z));
    qr.r.ip = F3(bpermf(src, r.ip.x), bpermf(src, r.ip.y), bpermf(src, r.ip.z));
    qr.r.Sx = bpermf(src, r.Sx); qr.r.Sy = bpermf(src, r.Sy);
    const uint packed = (uint)bperm(src, (int)(r.nkx | (r.nky << 8) | (r.nkz << 16)));
    qr.r.nkx = packed & 0xFFu; qr.r.nky = (packed >> 8) & 0xFFu; qr.r.nkz = (packed >> 16) & 0xFFu;
    qr.tmin = bpermf(src, tmin);
#if TR_QUAD_FETCH
    {   // which two chunks of a node this lane fetches for its quad: lane 0 near x + near z, 1 far x + far z, 2 near y + references, 3 far y + references
        const uint flip = (uint)(d.q & 1) << 4;
        qr.off_a = ((d.q & 2) ? qr.r.nky : qr.r.nkx) ^ flip;
        qr.off_b = (d.q & 2) ? 96u : (qr.r.nkz ^ flip);
    }
#endif
    qnode = bperm(src, node);
    qs.lds = qc.wave_stack + owner;
    qs.glob = qc.spill + d.qd * TR_QSPILL;
    qs.sp = bperm(src, stk.sp);
    qs.overflow = 0;
    return d;
}

// What a quad does with the children its lanes found: `inner` = this lane's child is an inner node that was hit, `key` orders
// the hits (smaller first; unique per lane).  The first becomes the quad's node, the others go onto the stack with the second
// on top; without a hit the quad pops its next node or is done.  Quad-uniform control flow; returns the stack depth reached.
TR_DEV void quad_descend(bool inner, uint key, int c, QuadStack& qs, int& qnode, bool& qlive) {
    const uint k1 = (uint)qrot1((int)key), k2 = (uint)qrot2((int)key), k3 = (uint)qrot3((int)key);
    const int rank = (int)(k1 < key) + (int)(k2 < key) + (int)(k3 < key);
    int n_inner = inner ? 1 : 0;
    n_inner += qrot1(n_inner); n_inner += qrot2(n_inner);
    int nx = (inner && rank == 0) ? c : 0;
    nx |= qrot1(nx); nx |= qrot2(nx);
    if (n_inner > 0) {
        if (inner && rank >= 1) qs.store(qs.sp + n_inner - 1 - rank, c);
        qs.sp += n_inner - 1;
        qnode = nx;
    } else if (qs.sp == 0) qlive = false;
    else { qs.sp--; qnode = qs.load(qs.sp); }
}

// A leaf the ray brought along from the per-lane phase (its current node or an entry of its stack: the per-lane loop pushes
// leaves, the quads do not): lane 0 of the quad takes it as its pending triangle, the quad goes on with the next entry.
TR_DEV void quad_inherited_leaf(int q, int& pend, QuadStack& qs, int& qnode, bool& qlive) {
    if (q == 0) pend = ~qnode;
    if (qs.sp == 0) qlive = false;
    else { qs.sp--; qnode = qs.load(qs.sp); }
}

// Closest hit for the rays of one wave.  Every lane of the wave calls this (`valid` = the lane has a ray); parameters and
// result as trace_closest4.
template <int ALPHA_MODE, bool COUNT>
TR_DEV void trace_closest_wave4(const SceneView& sv, bool valid, f3 org, f3 dir, float tmin, float tmax, bool include_lights, uint seed,
                                int* lds_stack, const QuadCtx& qc, HitRecord& hit, TraceStats& st, int& overflow) {
    hit.instance_id = -1; hit.primitive_id = -1; hit.u = 0; hit.v = 0; hit.t = -1.0f;
    float best_t = tmax, best_u = 0.0f, best_v = 0.0f;
    uint best_inst = 0xFFFFFFFFu, best_prim = 0xFFFFFFFFu;   // none found yet
    RayPre r = make_ray(org, dir);
    const bool finite_ray = valid && ray_is_finite(org, dir);
    bool live = finite_ray && sv.tri_count > 0;
    LaneStack stk;
    int spill[TR_SPILL_STACK];
    stk.init(lds_stack);
    int node = sv.node_count > 0 ? 0 : -1;
    TL(const unsigned long long tl_enter = tl_now(), tl_wall0 = tl_wall();)

    // candidate of a triangle test against the lane's best so far (shader/rt_common.rahit:15-24 for non-opaque geometry)
    auto consider = [&](const TriHit& tr, float t, float bu, float bv) {
        const uint inst = tr.inst_flags & 0x7FFFFFFFu;
        const bool closer = t < best_t || (t == best_t && best_inst != 0xFFFFFFFFu && (inst < best_inst || (inst == best_inst && tr.prim < best_prim)));
        if (closer) {
            bool accept = true;
            if (tr.inst_flags & 0x80000000u) {
                if (COUNT) st.alpha++;
                const float a = candidate_alpha(sv, tr.alpha, bu, bv);
                const float cutoff = ALPHA_MODE == 0 ? alpha_cutoff_hash(seed, (int)inst, (int)tr.prim) : 0.0001f;
                accept = !(a <= cutoff);
            }
            if (accept) { best_t = t; best_inst = inst; best_prim = tr.prim; best_u = bu; best_v = bv; }
        }
    };

    // ---- one ray per lane while more than TR_QUAD_SWITCH rays are live
    while (true) {
        const unsigned long long act = __ballot(live);
        if (__popcll(act) <= TR_QUAD_SWITCH) break;
        if (live) {
            const bool at_leaf = node < 0;
#if TR_VOTE > 0
            const int n_leaf = __popcll(__ballot(at_leaf)), n_all = __popcll(__ballot(true));
            // a triangle phase when TR_VOTE lanes hold a leaf - or half of the live ones, once the wave has thinned out
            const bool leaf_phase = n_leaf >= (TR_VOTE < ((n_all + 1) >> 1) ? TR_VOTE : ((n_all + 1) >> 1)) || n_leaf == n_all;
#else
            const bool leaf_phase = at_leaf;
#endif
            if (COUNT) {
                const unsigned long long m = __ballot(true);
                if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) {
                    if (!leaf_phase) { st.ph_node++; st.ph_hist[(__popcll(m) - 1) >> 3]++; if (__popcll(m) <= 16) { st.ph_node16++; st.lv_node16 += (uint)__popcll(__ballot(!at_leaf)); } if (__popcll(m) <= 8) st.ph_node8++; }
                    else st.ph_tri++;
                }
            }
            if (at_leaf == leaf_phase) {
                bool descend = false;
                TL(TlPhase tlp; const int tl_units = __popcll(__ballot(true)); tlp.begin();)
                if (!at_leaf) {
                    Hit4 h;
                    box4_intersect(r, sv.nodes4, node, tmin, best_t, h TL(, &tlp));
                    if (COUNT) st.nodes++;
                    TR_CE4(0, 1) TR_CE4(2, 3) TR_CE4(0, 2) TR_CE4(1, 3) TR_CE4(1, 2)
                    TL(asm volatile("" : "+v"(h.t[0]), "+v"(h.t[1]), "+v"(h.t[2]), "+v"(h.t[3]), "+v"(h.c[0]), "+v"(h.c[1]), "+v"(h.c[2]), "+v"(h.c[3])); tlp.mark_a();)
                    if (h.t[0] < __builtin_huge_valf()) {
                        // sorted: the hit children come first, so the number of further hits says which of c[1..3] go onto the stack
                        const int m = (int)(h.t[1] < __builtin_huge_valf()) + (int)(h.t[2] < __builtin_huge_valf()) + (int)(h.t[3] < __builtin_huge_valf());
                        stk.push_sorted(spill, m, h.c[1], h.c[2], h.c[3]);
                        if (COUNT) st.maxsp = max(st.maxsp, (uint)stk.sp);
                        node = h.c[0];
                        descend = true;
                    }
                } else {
                    TriHit tr;
                    if (COUNT) st.tris++;
#if TR_TIMELINE
                    const bool tl_hit = tri_intersect(r, sv.tris, (uint)~node, tmin, __builtin_huge_valf(), tr, &tlp);
                    tlp.mark_a();
                    if (tl_hit) { tlp.alpha = (tr.inst_flags & 0x80000000u) != 0 && (tr.t < best_t || tr.t == best_t); consider(tr, tr.t, tr.bu, tr.bv); }
                    tlp.mark_b();
#else
                    TR_LEAF_MEMBERS((uint)~node, true)
                    if (tri_intersect(r, sv.tris, leaf_index((uint)~node) + member, tmin, __builtin_huge_valf(), tr)) consider(tr, tr.t, tr.bu, tr.bv);
#endif
                }
                if (!descend) {
                    if (stk.sp == 0) live = false;
                    else node = stk.pop(spill);
                }
                TL(tlp.end(qc.tl, leaf_phase ? TL_LT : TL_LN, tl_units, (tl_units - 1) >> 3, leaf_phase ? TL_LT_WAIT_HIST : TL_LN_WAIT_HIST);)
            }
        }
    }

    // ---- the tail: one ray per quad
    const unsigned long long act = __ballot(live);
    const int n_act = __popcll(act);
#ifdef TR_QUAD_DEBUG
    const float dbg_u = live ? (float)stk.sp : -1.0f, dbg_v = live ? (float)((node < 0 ? 1000 : 0) + n_act) : -1.0f;
#endif
    if (n_act > 0) {
        QuadRay qr; QuadStack qs; int qnode;
        TL(const unsigned long long tl_deal = tl_now();)
        if (live) stk.flush(spill);     // the quads read the stack from memory: the top entry leaves its register
        const QuadDeal deal = quad_deal(act, live, r, tmin, node, stk, spill, qc, overflow, qr, qs, qnode);
        const int src = deal.src;
        const uint qseed = (uint)bperm(src, (int)seed);
        // every lane of the quad starts from the owner's best candidate; the quad shares the culling bound
        float lt = bpermf(src, best_t), lu = bpermf(src, best_u), lv = bpermf(src, best_v);
        uint linst = (uint)bperm(src, (int)best_inst), lprim = (uint)bperm(src, (int)best_prim);
        float qbest = lt;
        bool qlive = deal.has_ray;
        int pend = -1;      // triangle this lane has to test
#if TR_TAIL_PREFETCH
        int qpf = 0;
#endif
        TL(tl_misc(qc.tl, 2, tl_now() - tl_deal);)
        while (true) {
            int w = pend >= 0 ? 1 : 0;
            w |= qrot1(w); w |= qrot2(w);
            const bool qwait = w != 0;                      // a lane of this quad holds a triangle
            if (__ballot(qlive || qwait) == 0) break;
            const bool can_node = qlive && !qwait;
            const bool tri_phase = __popcll(__ballot(pend >= 0)) >= TR_QUAD_VOTE || __ballot(can_node) == 0;
            if (COUNT && (threadIdx.x & 63) == 0) { if (tri_phase) st.ph_qtri++; else st.ph_qnode++; }
            if (tri_phase) {
                if (pend >= 0) {
                    TL(TlPhase tlp; const int tl_units = __popcll(__ballot(true)); tlp.begin();)
                    TriHit tr;
                    if (COUNT) st.tris++;
                    TR_LEAF_MEMBERS((uint)pend, true)
                    if (tri_intersect(qr.r, sv.tris, leaf_index((uint)pend) + member, qr.tmin, __builtin_huge_valf(), tr TL(, &tlp))) {
                        const float t = tr.t, bu = tr.bu, bv = tr.bv;
                        const uint inst = tr.inst_flags & 0x7FFFFFFFu;
                        bool accept = t < lt || (t == lt && linst != 0xFFFFFFFFu && (inst < linst || (inst == linst && tr.prim < lprim)));
                        if (accept && (tr.inst_flags & 0x80000000u)) {
                            if (COUNT) st.alpha++;
                            const float a = candidate_alpha(sv, tr.alpha, bu, bv);
                            const float cutoff = ALPHA_MODE == 0 ? alpha_cutoff_hash(qseed, (int)inst, (int)tr.prim) : 0.0001f;
                            accept = !(a <= cutoff);
                        }
                        // selects, not a branch: hipcc 7.2 lowered the branchy form (five assignments under `if (accept)`) to code that
                        // kept the old lt / lv for accepted non-opaque candidates (the pair went through one 64-bit move that the
                        // alpha branch overwrote) - found as Suzanne hits with t = inf, tools/ab_dump.py
                        lt = accept ? t : lt; lu = accept ? bu : lu; lv = accept ? bv : lv;
                        linst = accept ? inst : linst; lprim = accept ? tr.prim : lprim;
                    }
                    TL(tlp.end(qc.tl, TL_QT, tl_units, (tl_units - 1) >> 3, -1);)
                }
                pend = -1;
                float m = lt;
                m = fminf(m, qrot1f(m)); m = fminf(m, qrot2f(m));
                qbest = fminf(qbest, m);
            } else if (can_node && qnode < 0) quad_inherited_leaf(quad_lane(), pend, qs, qnode, qlive);
            else if (can_node) {
                bool hitb; float t0;
                TL(TlPhase tlp; const int tl_units = __popcll(__ballot(true)) >> 2; tlp.begin();)
                const int q = quad_lane();
#if TR_TAIL_PREFETCH
                const int c = quad_child_box(qr, sv.nodes4, qnode, q, qbest, hitb, t0 TL(, &tlp), qpf);
#else
                const int c = quad_child_box(qr, sv.nodes4, qnode, q, qbest, hitb, t0 TL(, &tlp));
#endif
                if (COUNT && q == 0) st.nodes++;
                const bool inner = hitb && c >= 0;
#if TR_TAIL_PREFETCH
                if (inner) qpf = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(sv.nodes4) + ((size_t)(uint)c << 7) + 96);
#endif
                if (hitb && c < 0) pend = ~c;
                // order of the inner children that were hit: entry distance, ties by slot (two low mantissa bits carry the slot)
                quad_descend(inner, inner ? ((__float_as_uint(t0) & ~3u) | (uint)q) : 0xFFFFFFFFu, c, qs, qnode, qlive);
                if (COUNT) st.maxsp = max(st.maxsp, (uint)qs.sp);
                TL(tlp.end(qc.tl, TL_QN, tl_units, (tl_units - 1) >> 1, TL_QN_WAIT_HIST);)
            }
        }
        // the quad's result: smallest (t, instance, primitive) of its four lanes
        {
            float ot = qrot1f(lt), ou = qrot1f(lu), ov = qrot1f(lv); uint oi = (uint)qrot1((int)linst), op = (uint)qrot1((int)lprim);
            bool take = ot < lt || (ot == lt && (oi < linst || (oi == linst && op < lprim)));
            lt = take ? ot : lt; lu = take ? ou : lu; lv = take ? ov : lv; linst = take ? oi : linst; lprim = take ? op : lprim;
            ot = qrot2f(lt); ou = qrot2f(lu); ov = qrot2f(lv); oi = (uint)qrot2((int)linst); op = (uint)qrot2((int)lprim);
            take = ot < lt || (ot == lt && (oi < linst || (oi == linst && op < lprim)));
            lt = take ? ot : lt; lu = take ? ou : lu; lv = take ? ov : lv; linst = take ? oi : linst; lprim = take ? op : lprim;
        }
        int qo = qs.overflow;
        qo += qrot1(qo); qo += qrot2(qo);
        // back to the lanes the rays came from: the owner of rank k reads lane 4 k
        const int back = deal.back;
        const float rt = bpermf(back, lt), ru = bpermf(back, lu), rv = bpermf(back, lv);
        const uint ri = (uint)bperm(back, (int)linst), rp = (uint)bperm(back, (int)lprim);
        const int ro = bperm(back, qo);
        if (live) { best_t = rt; best_u = ru; best_v = rv; best_inst = ri; best_prim = rp; overflow += ro; }
    }
    overflow += stk.overflow ? 1 : 0;
    TL(tl_misc(qc.tl, 0, 1); tl_misc(qc.tl, 1, tl_now() - tl_enter); tl_misc(qc.tl, 4, tl_wall() - tl_wall0);)

    bool found = best_inst != 0xFFFFFFFFu;
    if (found) { hit.instance_id = (int)best_inst; hit.primitive_id = (int)best_prim; hit.u = best_u; hit.v = best_v; }
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
            float hh = (-b - sqrtf(disc)) / (2.0f * a);
            if (hh > 0 && hh > tmin && hh < best_t) {
                best_t = hh; found = true;
                hit.instance_id = -1; hit.primitive_id = (int)i; hit.u = hh; hit.v = 0;
            }
        }
    }
    hit.t = found ? best_t : -1.0f;
#ifdef TR_QUAD_DEBUG
    hit.u = dbg_u; hit.v = dbg_v;      // what the ray looked like when the wave switched to quads
#endif
}

// Any-hit visibility for the shadow rays of one wave (trace_shadow4 with the quad-cooperative tail).  Every lane of the wave
// calls this; returns the product of (1 - alpha) over the non-opaque hits, 0 after an opaque one.
template <bool COUNT>
TR_DEV float trace_shadow_wave4(const SceneView& sv, bool valid, f3 org, f3 dir, float tmin, float tmax, int* lds_stack, const QuadCtx& qc,
                                TraceStats& st, int& overflow) {
    float visibility = 1.0f;
    bool live = valid && sv.tri_count > 0 && ray_is_finite(org, dir);
    RayPre r = make_ray(org, dir);
    LaneStack stk;
    int spill[TR_SPILL_STACK];
    stk.init(lds_stack);
    int node = sv.node_count > 0 ? 0 : -1;
    while (true) {
        if (__popcll(__ballot(live)) <= TR_QUAD_SWITCH) break;
#if TR_VOTE_SHADOW_WAVE > 0
        bool go = live;
        if (live) {     // wave vote as in the closest-hit loop: lanes holding a leaf wait until enough of them do
            const bool at_leaf = node < 0;
            const int n_leaf = __popcll(__ballot(at_leaf)), n_all = __popcll(__ballot(true));
            const bool leaf_phase = n_leaf >= TR_VOTE_SHADOW_WAVE || n_leaf == n_all;
            go = at_leaf == leaf_phase;
        }
        if (go) {
#else
        if (live) {
#endif
            bool descend = false;
            if (node >= 0) {
                Hit4 h;
                box4_intersect(r, sv.nodes4, node, tmin, tmax, h);
                if (COUNT) st.nodes++;
                descend = shadow_descend(h, stk, spill, node);
            } else {
                TriHit tr;
                if (COUNT) st.tris++;
                TR_LEAF_MEMBERS((uint)~node, live)
                if (tri_intersect(r, sv.tris, leaf_index((uint)~node) + member, tmin, tmax, tr)) {
                    if (!(tr.inst_flags & 0x80000000u)) { visibility = 0.0f; live = false; }
                    else {
                        if (COUNT) st.alpha++;
                        const float alpha = candidate_alpha(sv, tr.alpha, tr.bu, tr.bv);
                        visibility *= 1.0f - alpha;
                        if (visibility == 0.0f) live = false;
                    }
                }
            }
            if (live && !descend) {
                if (stk.sp == 0) live = false;
                else node = stk.pop(spill);
            }
        }
    }
    const unsigned long long act = __ballot(live);
    const int n_act = __popcll(act);
    if (n_act > 0) {
        QuadRay qr; QuadStack qs; int qnode;
        if (live) stk.flush(spill);
        const QuadDeal deal = quad_deal(act, live, r, tmin, node, stk, spill, qc, overflow, qr, qs, qnode);
        const int q = deal.q, src = deal.src;
        const float qtmax = bpermf(src, tmax);
        const float owner_vis = bpermf(src, visibility);
        float lvis = q == 0 ? owner_vis : 1.0f;     // the owner's product so far rides in lane 0 of the quad
        bool qlive = deal.has_ray;
        int pend = -1;
        while (true) {
            int w = pend >= 0 ? 1 : 0;
            w |= qrot1(w); w |= qrot2(w);
            const bool qwait = w != 0;
            if (__ballot(qlive || qwait) == 0) break;
            const bool can_node = qlive && !qwait;
            const bool tri_phase = __popcll(__ballot(pend >= 0)) >= TR_QUAD_VOTE || __ballot(can_node) == 0;
            if (tri_phase) {
                if (pend >= 0) {
                    TriHit tr;
                    if (COUNT) st.tris++;
                    TR_LEAF_MEMBERS((uint)pend, true)
                    if (tri_intersect(qr.r, sv.tris, leaf_index((uint)pend) + member, qr.tmin, qtmax, tr)) {
                        if (!(tr.inst_flags & 0x80000000u)) lvis = 0.0f;
                        else {
                            if (COUNT) st.alpha++;
                            lvis *= 1.0f - candidate_alpha(sv, tr.alpha, tr.bu, tr.bv);
                        }
                    }
                }
                pend = -1;
                int z = lvis == 0.0f ? 1 : 0;
                z |= qrot1(z); z |= qrot2(z);
                if (z) qlive = false;      // occluded: nothing left to find
            } else if (can_node && qnode < 0) quad_inherited_leaf(quad_lane(), pend, qs, qnode, qlive);
            else if (can_node) {
                bool hitb; float t0;
                const int q = quad_lane();
                const int c = quad_child_box(qr, sv.nodes4, qnode, q, qtmax, hitb, t0);
                if (COUNT && q == 0) st.nodes++;
                const bool inner = hitb && c >= 0;
                if (hitb && c < 0) pend = ~c;
                quad_descend(inner, inner ? (uint)q : 0xFFFFFFFFu, c, qs, qnode, qlive);      // slot order, as the per-lane loop descends
            }
        }
        float v = lvis;
        v *= qrot1f(v); v *= qrot2f(v);
        int qo = qs.overflow;
        qo += qrot1(qo); qo += qrot2(qo);
        const int back = deal.back;
        const float rv = bpermf(back, v);
        const int ro = bperm(back, qo);
        if (live) { visibility = rv; overflow += ro; }
    }
    overflow += stk.overflow ? 1 : 0;
    return visibility;
}

}  // namespace tr
