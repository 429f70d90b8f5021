// Triangle pre-splitting for static ("prefer fast trace") builds: large or diagonal triangles enter the Morton sort as several
// references, each with the tight box of the part of the triangle inside one cell of the Morton grid, so that the clustering is
// not forced to wrap one huge box around them (Karras & Aila, "Fast Parallel Construction of High-Quality Bounding Volume
// Hierarchies", 2013, section 4; the driver behind the reference's ePreferFastTrace builds, src/acceleration_structure.cc:129-133,
// is free to do the same).  Included by bvh_build.hip only.
//
// A reference is a leaf of its own: a copy of the 48-byte triangle record (traversal needs no indirection) with the clipped box
// as leaf box.  The triangle test is the triangle's own, so a closest-hit query finds the same (t, instance, primitive) through
// whichever reference it reaches first, and reaching a second one changes nothing (equal candidates do not replace each
// other).  An any-hit query multiplies (1 - alpha) per candidate, so only opaque triangles are split: the first opaque candidate
// ends such a query whichever reference holds it.
//
// Why no hit is lost: every point P of triangle T lies in the box of at least one of T's references.  By induction over the
// splits - P lies in the parent reference's box B; the split plane x_k = s puts P into one closed half-space; the child box of
// that side is B intersected with the bounding box of (T clipped to the half-space), whose corners are T's vertices on that side
// (exact) and the points where T's edges cross the plane (computed, then padded by 8 ulps of the larger coordinate of the edge).
#pragma once

namespace tr {
namespace {

#define PRESPLIT_MAX_PER_TRI 32      // splits one triangle may receive
#define PRESPLIT_CANDIDATES 32       // scale factors tried per search pass

struct SplitGrid { float lo[3], ext[3], inv[3]; };     // the Morton grid: scene bounds, extent, 2^21 / extent

TR_DEV uint split_quant(const SplitGrid& g, int k, float x) {
    const float f = (x - g.lo[k]) * g.inv[k];
    return (uint)fminf(fmaxf(f, 0.0f), 2097151.0f);
}

// The most important spatial-median plane of the Morton grid that cuts the box strictly inside (the coarsest level over the
// three axes; ties go to the longer axis).  Returns its level bit h (20 = the root plane, 0 = the finest), or -1 if no grid plane
// cuts the box.
TR_DEV int split_plane(const SplitGrid& g, const float* lo, const float* hi, int& axis, float& pos) {
    int best_h = -1;
    float best_ext = -1.0f;
    for (int k = 0; k < 3; ++k) {
        if (!(hi[k] > lo[k])) continue;
        const uint qa = split_quant(g, k, lo[k]), qb = split_quant(g, k, hi[k]);
        if (qa == qb) continue;
        const int h = 31 - __clz((int)(qa ^ qb));
        const uint cell = (qb >> h) << h;                          // the multiple of 2^h in (qa, qb]
        const float p = g.lo[k] + (float)cell * (g.ext[k] * (1.0f / 2097152.0f));
        if (!(p > lo[k] && p < hi[k])) continue;
        const float e = hi[k] - lo[k];
        if (h > best_h || (h == best_h && e > best_ext)) { best_h = h; best_ext = e; axis = k; pos = p; }
    }
    return best_h;
}

TR_DEV void tri_bounds(const TriRecord& t, float* lo, float* hi) {
    for (int k = 0; k < 3; ++k) { lo[k] = fminf(fminf(t.v0[k], t.v1[k]), t.v2[k]); hi[k] = fmaxf(fmaxf(t.v0[k], t.v1[k]), t.v2[k]); }
}

// priority of a triangle: (2^i * (A_aabb - A_ideal))^(1/3), i = level of the most important plane cutting its box (0 = root);
// A_ideal = |cross(e1, e2)|_1, the box area infinitely many splits would approach
__global__ __launch_bounds__(BT) void k_split_priority(uint n, const TriRecord* tris, SplitGrid g, float* prio, uint* max_bits) {
    const uint i = blockIdx.x * BT + threadIdx.x;
    float p = 0.0f;
    if (i < n) {
        const TriRecord t = tris[i];
        if (!(t.inst_flags & 0x80000000u)) {
            float lo[3], hi[3];
            tri_bounds(t, lo, hi);
            int axis = 0; float pos = 0.0f;
            const int h = split_plane(g, lo, hi, axis, pos);
            if (h >= 0) {
                const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
                const float a_aabb = 2.0f * (dx * dy + dy * dz + dz * dx);
                const f3 e1 = F3(t.v1[0] - t.v0[0], t.v1[1] - t.v0[1], t.v1[2] - t.v0[2]), e2 = F3(t.v2[0] - t.v0[0], t.v2[1] - t.v0[1], t.v2[2] - t.v0[2]);
                const f3 c = cross(e1, e2);
                const float a_ideal = fabsf(c.x) + fabsf(c.y) + fabsf(c.z);
                const float excess = a_aabb - a_ideal;
                if (excess > 0.0f && isfinite(excess)) p = cbrtf(ldexpf(excess, h - 20));
            }
        }
        prio[i] = p;
    }
    float m = p;
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(max_bits, __float_as_uint(m));
}

TR_DEV uint split_count(float prio, float scale) {
    const float s = prio * scale;
    return s >= (float)PRESPLIT_MAX_PER_TRI ? (uint)PRESPLIT_MAX_PER_TRI : (uint)s;
}

// total number of splits for each of PRESPLIT_CANDIDATES scale factors (integer sums: the result does not depend on thread order)
struct SplitScales { float s[PRESPLIT_CANDIDATES]; };
__global__ __launch_bounds__(BT) void k_split_totals(uint n, const float* prio, SplitScales sc, unsigned long long* totals) {
    __shared__ uint s_sum[PRESPLIT_CANDIDATES];
    if (threadIdx.x < PRESPLIT_CANDIDATES) s_sum[threadIdx.x] = 0;
    __syncthreads();
    const uint i = blockIdx.x * BT + threadIdx.x;
    const float p = i < n ? prio[i] : 0.0f;
    for (int c = 0; c < PRESPLIT_CANDIDATES; ++c) {
        uint v = split_count(p, sc.s[c]);
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_sum[c], v);
    }
    __syncthreads();
    if (threadIdx.x < PRESPLIT_CANDIDATES && s_sum[threadIdx.x]) atomicAdd(&totals[threadIdx.x], (unsigned long long)s_sum[threadIdx.x]);
}

__global__ __launch_bounds__(BT) void k_split_counts(uint n, const float* prio, float scale, uint* count /* n + BT entries for the scan */) {
    const uint i = blockIdx.x * BT + threadIdx.x;
    count[i] = i < n ? 1u + split_count(prio[i], scale) : 0u;
}

// Bounding boxes of the two halves of the triangle on either side of x_axis = pos, each intersected with the parent box.
TR_DEV void split_boxes(const TriRecord& t, int axis, float pos, const float* plo, const float* phi, float* llo, float* lhi, float* rlo, float* rhi) {
    const float inf = __builtin_huge_valf();
    for (int k = 0; k < 3; ++k) { llo[k] = rlo[k] = inf; lhi[k] = rhi[k] = -inf; }
    const float* v[3] = {t.v0, t.v1, t.v2};
    for (int e = 0; e < 3; ++e) {
        const float* a = v[e];
        const float* b = v[e == 2 ? 0 : e + 1];
        const float da = a[axis], db = b[axis];
        if (da <= pos) for (int k = 0; k < 3; ++k) { llo[k] = fminf(llo[k], a[k]); lhi[k] = fmaxf(lhi[k], a[k]); }
        if (da >= pos) for (int k = 0; k < 3; ++k) { rlo[k] = fminf(rlo[k], a[k]); rhi[k] = fmaxf(rhi[k], a[k]); }
        if ((da < pos && db > pos) || (da > pos && db < pos)) {
            const float w = (pos - da) / (db - da);
            for (int k = 0; k < 3; ++k) {
                float x = a[k] + w * (b[k] - a[k]);
                const float pad = 4.8e-7f * fmaxf(fabsf(a[k]), fabsf(b[k]));     // 8 ulps of the larger end: covers the rounding of w and of the interpolation
                float xl = x - pad, xh = x + pad;
                if (k == axis) { xl = pos; xh = pos; }
                llo[k] = fminf(llo[k], xl); lhi[k] = fmaxf(lhi[k], xh);
                rlo[k] = fminf(rlo[k], xl); rhi[k] = fmaxf(rhi[k], xh);
            }
        }
    }
    for (int k = 0; k < 3; ++k) {
        llo[k] = fmaxf(llo[k], plo[k]); lhi[k] = fminf(lhi[k], phi[k]);
        rlo[k] = fmaxf(rlo[k], plo[k]); rhi[k] = fminf(rhi[k], phi[k]);
    }
    lhi[axis] = fminf(lhi[axis], pos);
    rlo[axis] = fmaxf(rlo[axis], pos);
}

// Emits the 1 + s references of every triangle: a depth-first recursion over (box, splits left) pairs.  A box is cut by the most
// important grid plane through it (by the middle of its longest axis when no grid plane cuts it), the remaining splits go to
// the halves in proportion to their longest extents.
__global__ __launch_bounds__(BT) void k_split_emit(uint n, const TriRecord* tris, const uint* offset, const uint* count, SplitGrid g, TriRecord* refs, float* ref_box) {
    const uint i = blockIdx.x * BT + threadIdx.x;
    if (i >= n) return;
    const TriRecord t = tris[i];
    uint out = offset[i];
    const uint end = out + count[i];
    float slo[PRESPLIT_MAX_PER_TRI + 1][3], shi[PRESPLIT_MAX_PER_TRI + 1][3];
    int ssplits[PRESPLIT_MAX_PER_TRI + 1];
    int sp = 0;
    tri_bounds(t, slo[0], shi[0]);
    ssplits[0] = (int)(count[i] - 1u);
    sp = 1;
    while (sp > 0) {
        --sp;
        float lo[3], hi[3];
        for (int k = 0; k < 3; ++k) { lo[k] = slo[sp][k]; hi[k] = shi[sp][k]; }
        const int s = ssplits[sp];
        int axis = 0; float pos = 0.0f;
        bool can = false;
        if (s > 0) {
            can = split_plane(g, lo, hi, axis, pos) >= 0;
            if (!can) {      // below the grid's resolution: the middle of the longest axis
                float e = 0.0f;
                for (int k = 0; k < 3; ++k) if (hi[k] - lo[k] > e) { e = hi[k] - lo[k]; axis = k; }
                pos = lo[axis] + 0.5f * e;
                can = e > 0.0f && pos > lo[axis] && pos < hi[axis];
            }
        }
        float llo[3], lhi[3], rlo[3], rhi[3];
        float wl = 0.0f, wr = 0.0f;
        if (can) {
            split_boxes(t, axis, pos, lo, hi, llo, lhi, rlo, rhi);
            // The clipped boxes are conservative, so near a corner of the box the triangle may not reach one side of the plane at
            // all: then the box stays whole.
            for (int k = 0; k < 3; ++k) { wl = fmaxf(wl, lhi[k] - llo[k]); wr = fmaxf(wr, rhi[k] - rlo[k]); can = can && lhi[k] >= llo[k] && rhi[k] >= rlo[k]; }
        }
        if (!can) {
            // a leaf of the recursion, or a box that cannot be cut: the splits it still holds become copies, so that every slot
            // the scan reserved is written
            for (int c = 0; c <= s && out < end; ++c, ++out) {
                refs[out] = t;
                for (int k = 0; k < 3; ++k) { ref_box[6 * (size_t)out + k] = lo[k]; ref_box[6 * (size_t)out + 3 + k] = hi[k]; }
            }
            continue;
        }
        const int rest = s - 1;
        int sl = (wl + wr) > 0.0f ? (int)floorf((float)rest * (wl / (wl + wr)) + 0.5f) : rest / 2;
        sl = sl < 0 ? 0 : (sl > rest ? rest : sl);
        for (int k = 0; k < 3; ++k) { slo[sp][k] = rlo[k]; shi[sp][k] = rhi[k]; }
        ssplits[sp] = rest - sl; sp++;
        for (int k = 0; k < 3; ++k) { slo[sp][k] = llo[k]; shi[sp][k] = lhi[k]; }
        ssplits[sp] = sl; sp++;
    }
}

// Morton keys of the references: centre of the reference's box in the grid the split planes come from
__global__ __launch_bounds__(BT) void k_morton_refs(uint n, const float* ref_box, SplitGrid g, unsigned long long* keys, uint* vals) {
    const uint i = blockIdx.x * BT + threadIdx.x;
    if (i >= n) return;
    uint q[3];
    for (int k = 0; k < 3; ++k) q[k] = split_quant(g, k, (ref_box[6 * (size_t)i + k] + ref_box[6 * (size_t)i + 3 + k]) * 0.5f);
    keys[i] = (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
    vals[i] = i;
}

__global__ __launch_bounds__(BT) void k_gather_refs(uint n, const TriRecord* unsorted, const float* unsorted_box, const uint* sorted_vals, TriRecord* sorted, float* leaf_box) {
    const uint i = blockIdx.x * BT + threadIdx.x;
    if (i >= n) return;
    const uint s = sorted_vals[i];
    sorted[i] = unsorted[s];
    for (int k = 0; k < 6; ++k) leaf_box[6 * (size_t)i + k] = unsorted_box[6 * (size_t)s + k];
}

}  // namespace
}  // namespace tr
