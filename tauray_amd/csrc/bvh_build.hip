// On-device acceleration-structure build: the software replacement for the driver's BLAS/TLAS build
// (reference call sites src/acceleration_structure.cc:198,266,421; recorded by src/scene_stage.cc:1620-1662).
//   world triangles (model * vec4(pos, 1), GLSL order) + centroid bounds -> 63-bit Morton keys -> radix sort (rocPRIM)
//   -> PLOC clustering over the Morton order (or Karras 2012 LBVH + atomic refit, TRHIP_BUILDER=lbvh)
//   -> depth-first relabelling -> in-place collapse to 4-wide fp32 nodes (128 B) + 48-byte triangle records.
// trhip_scene_refit_accel keeps the tree and recomputes its boxes level by level.  Also here: extract_tri_lights
// (shader/extract_tri_lights.comp:17-54) and the pre-transformed vertex copy (shader/pre_transform.comp:26-42).
#include <algorithm>
#include <vector>
#include "build.h"

#include <rocprim/rocprim.hpp>

namespace tr {

namespace {

constexpr int BT = 256;

// order-preserving float <-> uint map for atomicMin/Max on bounds
TR_DEV uint float_flip(float f) { uint u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
TR_HD float float_unflip(uint u) {
    uint v = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float f;
#if defined(__HIP_DEVICE_COMPILE__)
    f = __uint_as_float(v);
#else
    memcpy(&f, &v, 4);
#endif
    return f;
}

// instance/primitive of a global triangle id via binary search over the per-instance prefix sums
TR_DEV void locate_triangle(const uint* tri_prefix, uint instance_count, uint gid, uint& inst, uint& prim) {
    uint lo = 0, hi = instance_count;   // prefix has instance_count + 1 entries
    while (hi - lo > 1) {
        uint mid = (lo + hi) >> 1;
        if (tri_prefix[mid] <= gid) lo = mid; else hi = mid;
    }
    inst = lo;
    prim = gid - tri_prefix[lo];
}

// TriRecord::alpha of a triangle of a non-opaque instance, and its AlphaTri record (common.h).  The material can change between a build
// and a refit (trhip_scene_update_instances replaces the whole instance record), so both write it.
TR_DEV uint alpha_word(const Material& mat, uint record, AlphaTri* alpha_tris, f2 uv0, f2 uv1, f2 uv2) {
    AlphaTri a;
    a.uv0 = uv0; a.uv1 = uv1; a.uv2 = uv2; a.factor = mat.albedo_factor.w; a.tex = mat.albedo_tex_id;
    alpha_tris[record] = a;
    const uint bits = __float_as_uint(a.factor);
    return (a.tex < 0 && !(bits & 0x80000000u)) ? bits : (0x80000000u | record);
}

// world-space triangle = (model * vec4(pos, 1)).xyz in the GLSL evaluation order; also accumulates the
// centroid bounds used to quantise Morton codes.
__global__ __launch_bounds__(BT) void k_pretransform(SceneView sv, const uint* tri_prefix, const uint8_t* non_opaque, const uint* alpha_base, AlphaTri* alpha_tris,
                                                     TriRecord* tris_unsorted, uint* cbounds /*6 flipped uints of the centroid bounds; [16..21] the same for the triangles' bounds*/) {
    float cmin[3] = {__builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf()};
    float cmax[3] = {-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf()};
    float smin[3] = {__builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf()};
    float smax[3] = {-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf()};
    // a bounded grid walks the triangles: the twelve bounds end in twelve atomics per block on two cache lines, and 16 000 waves of
    // them (one set per wave of a grid of one thread per triangle) took a millisecond of a 4 ms rebuild - the atomics of a line are
    // served one after the other ("Atomics")
    for (uint gid = blockIdx.x * BT + threadIdx.x; gid < sv.tri_count; gid += gridDim.x * BT) {
        uint inst, prim;
        locate_triangle(tri_prefix, sv.instance_count, gid, inst, prim);
        const MeshSpan sp = sv.spans[inst];
        const uint* ix = sv.indices + sp.index_offset + 3u * prim;
        const Vertex* vb = sv.vertices + sp.vertex_offset;
        const m4 model = sv.instances[inst].model;
        f3 p0 = transform_point(model, vb[ix[0]].pos);
        f3 p1 = transform_point(model, vb[ix[1]].pos);
        f3 p2 = transform_point(model, vb[ix[2]].pos);
        TriRecord t;
        t.v0[0] = p0.x; t.v0[1] = p0.y; t.v0[2] = p0.z;
        t.v1[0] = p1.x; t.v1[1] = p1.y; t.v1[2] = p1.z;
        t.v2[0] = p2.x; t.v2[1] = p2.y; t.v2[2] = p2.z;
        t.inst_flags = inst | (non_opaque[inst] ? 0x80000000u : 0u);
        t.prim = prim;
        t.alpha = non_opaque[inst] ? alpha_word(sv.instances[inst].mat, alpha_base[inst] + prim, alpha_tris, vb[ix[0]].uv, vb[ix[1]].uv, vb[ix[2]].uv) : 0u;
        tris_unsorted[gid] = t;
        f3 lo = min3(min3(p0, p1), p2), hi = max3(max3(p0, p1), p2);
        f3 c = (lo + hi) * 0.5f;
        const float cc[3] = {c.x, c.y, c.z}, ll[3] = {lo.x, lo.y, lo.z}, hh[3] = {hi.x, hi.y, hi.z};
        for (int k = 0; k < 3; ++k) { cmin[k] = fminf(cmin[k], cc[k]); cmax[k] = fmaxf(cmax[k], cc[k]); smin[k] = fminf(smin[k], ll[k]); smax[k] = fmaxf(smax[k], hh[k]); }
    }
    // wave reduce, block reduce through LDS, then one set of atomics per block
    __shared__ float s_red[BT / 64][12];
    for (int k = 0; k < 3; ++k) {
        float mn = cmin[k], mx = cmax[k], sn = smin[k], sx = smax[k];
        for (int off = 32; off > 0; off >>= 1) {
            mn = fminf(mn, __shfl_xor(mn, off));
            mx = fmaxf(mx, __shfl_xor(mx, off));
            sn = fminf(sn, __shfl_xor(sn, off));
            sx = fmaxf(sx, __shfl_xor(sx, off));
        }
        if ((threadIdx.x & 63) == 0) { float* w = s_red[threadIdx.x >> 6]; w[k] = mn; w[3 + k] = mx; w[6 + k] = sn; w[9 + k] = sx; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = (int)threadIdx.x;
        float mn = s_red[0][k], mx = s_red[0][3 + k], sn = s_red[0][6 + k], sx = s_red[0][9 + k];
        for (int w = 1; w < BT / 64; ++w) { mn = fminf(mn, s_red[w][k]); mx = fmaxf(mx, s_red[w][3 + k]); sn = fminf(sn, s_red[w][6 + k]); sx = fmaxf(sx, s_red[w][9 + k]); }
        if (mn <= mx) {
            atomicMin(&cbounds[k], float_flip(mn));
            atomicMax(&cbounds[3 + k], float_flip(mx));
            atomicMin(&cbounds[16 + k], float_flip(sn));     // bounds of the triangles themselves: the grid of the pre-split
            atomicMax(&cbounds[19 + k], float_flip(sx));
        }
    }
}

TR_DEV unsigned long long expand21(uint v) {   // spread 21 bits to every third bit
    unsigned long long x = v & 0x1FFFFFull;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(BT) void k_morton(uint n, const TriRecord* tris, const uint* cbounds, unsigned long long* keys, uint* vals) {
    uint i = blockIdx.x * BT + threadIdx.x;
    if (i >= n) return;
    float lo[3], inv[3];
    for (int k = 0; k < 3; ++k) {
        lo[k] = float_unflip(cbounds[k]);
        float ext = float_unflip(cbounds[3 + k]) - lo[k];
        inv[k] = ext > 0 ? 2097151.0f / ext : 0.0f;
    }
    const TriRecord t = tris[i];
    float c[3];
    for (int k = 0; k < 3; ++k) {
        float mn = fminf(fminf(t.v0[k], t.v1[k]), t.v2[k]), mx = fmaxf(fmaxf(t.v0[k], t.v1[k]), t.v2[k]);
        c[k] = (mn + mx) * 0.5f;
    }
    uint q[3];
    for (int k = 0; k < 3; ++k) {
        float f = (c[k] - lo[k]) * inv[k];
        q[k] = (uint)fminf(fmaxf(f, 0.0f), 2097151.0f);
    }
    keys[i] = (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
    vals[i] = i;
}

// Karras 2012: common-prefix length between sorted keys i and j, ties broken by index
TR_DEV int delta(const unsigned long long* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    unsigned long long a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((uint)i ^ (uint)j);
    return __clzll((long long)(a ^ b));
}

// one thread per internal node i in [0, n-1): children + parent links.
// refs: >= 0 internal node, < 0 leaf ~leaf_index (leaf_index = position in sorted order)
__global__ __launch_bounds__(BT) void k_hierarchy(int n, const unsigned long long* keys, int2* children, uint* subtree_size, int* parent_internal, int* parent_leaf) {
    int i = blockIdx.x * BT + threadIdx.x;
    if (i >= n - 1) return;
    int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta(keys, n, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int gamma = i + s * d + min(d, 0);
    int left = (min(i, j) == gamma) ? ~gamma : gamma;
    int right = (max(i, j) == gamma + 1) ? ~(gamma + 1) : (gamma + 1);
    children[i] = make_int2(left, right);
    subtree_size[i] = (uint)(max(i, j) - min(i, j) + 1);   // leaves covered by this node
    if (left >= 0) parent_internal[left] = i; else parent_leaf[~left] = i;
    if (right >= 0) parent_internal[right] = i; else parent_leaf[~right] = i;
    if (i == 0) parent_internal[0] = -1;
}

// gather triangles into Morton order and emit leaf boxes
__global__ __launch_bounds__(BT) void k_gather_leaves(uint n, const TriRecord* unsorted, const uint* sorted_vals, TriRecord* sorted, float* leaf_box /*6 per leaf*/) {
    uint i = blockIdx.x * BT + threadIdx.x;
    if (i >= n) return;
    const TriRecord t = unsorted[sorted_vals[i]];
    sorted[i] = t;
    for (int k = 0; k < 3; ++k) {
        leaf_box[6 * i + k] = fminf(fminf(t.v0[k], t.v1[k]), t.v2[k]);
        leaf_box[6 * i + 3 + k] = fmaxf(fmaxf(t.v0[k], t.v1[k]), t.v2[k]);
    }
}

// bottom-up refit: the second thread to reach a node owns it.  Agent-scope release/acquire around the
// arrival counter makes the first child's box (written by another CU, possibly another XCD) visible.
__global__ __launch_bounds__(BT) void k_refit(int n, const int2* children, const int* parent_internal, const int* parent_leaf,
                                              const float* leaf_box, float* node_box /*6 per internal*/, uint* arrive, BvhNode* nodes) {
    int leaf = blockIdx.x * BT + threadIdx.x;
    if (leaf >= n) return;
    int node = parent_leaf[leaf];
    while (node >= 0) {
        __threadfence();   // release: our child's box is published before we announce arrival
        uint prev = __hip_atomic_fetch_add(&arrive[node], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == 0) return;            // sibling subtree not finished: the other thread continues
        __threadfence();   // acquire
        const int2 ch = children[node];
        BvhNode out;
        float lo[3], hi[3];
        {
            const float* b0 = ch.x >= 0 ? node_box + 6 * (size_t)ch.x : leaf_box + 6 * (size_t)(~ch.x);
            const float* b1 = ch.y >= 0 ? node_box + 6 * (size_t)ch.y : leaf_box + 6 * (size_t)(~ch.y);
            for (int k = 0; k < 3; ++k) {
                float l0 = __hip_atomic_load(&b0[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                float h0 = __hip_atomic_load(&b0[3 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                float l1 = __hip_atomic_load(&b1[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                float h1 = __hip_atomic_load(&b1[3 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                out.lo0[k] = l0; out.hi0[k] = h0; out.lo1[k] = l1; out.hi1[k] = h1;
                lo[k] = fminf(l0, l1); hi[k] = fmaxf(h0, h1);
            }
        }
        out.child0 = ch.x; out.child1 = ch.y; out.pad0 = 0; out.pad1 = 0;
        nodes[node] = out;
        for (int k = 0; k < 3; ++k) {
            __hip_atomic_store(&node_box[6 * (size_t)node + k], lo[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&node_box[6 * (size_t)node + 3 + k], hi[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        node = parent_internal[node];
    }
}


// ---------------------------------------------------------------------------------------------------------------
// PLOC (parallel locally-ordered clustering, Meister & Bittner 2018) on the Morton-sorted leaves: every cluster
// looks for the neighbour within +-PLOC_RADIUS positions whose merged box has the smallest area; mutual nearest
// neighbours merge; the cluster array is compacted and the round repeats until one cluster is left.  Gives
// near-SAH trees while staying a radix-sort-based on-device build.  Internal node ids are handed out from the top
// so that the last merge (the root) is node 0.
#define PLOC_RADIUS 16

TR_DEV float merged_area(const float* a, const float* b) {
    float dx = fmaxf(a[3], b[3]) - fminf(a[0], b[0]);
    float dy = fmaxf(a[4], b[4]) - fminf(a[1], b[1]);
    float dz = fmaxf(a[5], b[5]) - fminf(a[2], b[2]);
    return dx * dy + dy * dz + dz * dx;
}

// The live cluster count of a round sits in device memory (c_ptr): the host only learns it every few rounds and sizes
// the grids by its last known value, so a round costs launches but no host round trip.
__global__ __launch_bounds__(BT) void k_ploc_nn(const uint* c_ptr, uint radius, const float* cbox, uint* nn) {
    const uint c = *c_ptr;
    uint i = blockIdx.x * BT + threadIdx.x;
    if (i >= c) return;
    float mine[6];
    for (int k = 0; k < 6; ++k) mine[k] = cbox[6 * (size_t)i + k];
    uint lo = i > radius ? i - radius : 0, hi = min(c - 1, i + radius);
    float best = __builtin_huge_valf(); uint bj = i;
    for (uint j = lo; j <= hi; ++j) {
        if (j == i) continue;
        float a = merged_area(mine, cbox + 6 * (size_t)j);
        if (a < best) { best = a; bj = j; }
    }
    nn[i] = bj;
}

__global__ __launch_bounds__(BT) void k_ploc_merge(const uint* c_ptr, uint n_leaves, const int* cref, const float* cbox, const uint* nn, uint* valid, int* out_ref,
                                                   float* out_box, uint* alloc, int2* children, float* node_box, uint* subtree_size, int* parent_internal) {
    const uint c = *c_ptr;
    uint i = blockIdx.x * BT + threadIdx.x;
    if (i >= c) { if (i < gridDim.x * BT) valid[i] = 0; return; }   // keeps the scan input defined past the live range
    uint j = nn[i];
    const bool mutual = j != i && nn[j] == i;
    if (mutual && i > j) { valid[i] = 0; return; }
    int ref = cref[i];
    float box[6];
    for (int k = 0; k < 6; ++k) box[k] = cbox[6 * (size_t)i + k];
    if (mutual) {
        const int other = cref[j];
        const float* ob = cbox + 6 * (size_t)j;
        for (int k = 0; k < 3; ++k) { box[k] = fminf(box[k], ob[k]); box[3 + k] = fmaxf(box[3 + k], ob[3 + k]); }
        const uint id = (n_leaves - 2u) - atomicAdd(alloc, 1u);
        children[id] = make_int2(ref, other);
        if (ref >= 0) parent_internal[ref] = (int)id;
        if (other >= 0) parent_internal[other] = (int)id;
        for (int k = 0; k < 6; ++k) node_box[6 * (size_t)id + k] = box[k];
        subtree_size[id] = (ref < 0 ? 1u : subtree_size[ref]) + (other < 0 ? 1u : subtree_size[other]);
        ref = (int)id;
    }
    valid[i] = 1;
    out_ref[i] = ref;
    for (int k = 0; k < 6; ++k) out_box[6 * (size_t)i + k] = box[k];
}

__global__ __launch_bounds__(BT) void k_ploc_compact(const uint* c_ptr, uint* c_next, const uint* valid, const uint* pos, const int* in_ref, const float* in_box,
                                                     int* cref, float* cbox) {
    const uint c = *c_ptr;
    uint i = blockIdx.x * BT + threadIdx.x;
    if (i >= c) return;
    if (i == c - 1) *c_next = pos[i] + valid[i];   // cluster count of the next round
    if (!valid[i]) return;
    uint p = pos[i];
    cref[p] = in_ref[i];
    for (int k = 0; k < 6; ++k) cbox[6 * (size_t)p + k] = in_box[6 * (size_t)i + k];
}

// The last rounds in one workgroup: once PLOC_TAIL clusters or fewer are left, a round of three grid launches and a scan is all
// launch latency (a 1 M triangle build spent about thirty of its seventy-odd rounds there, 20-30 us each).  The clusters move into LDS, one per
// thread, and the same three steps - nearest neighbour, merge of the mutual pairs, compaction - repeat between barriers until one
// cluster is left.  Same neighbour choice (ascending scan, first minimum wins) and same merge rule as the grid kernels, so the
// same tree; node ids come from the same allocator.
#define PLOC_TAIL 1024
__global__ __launch_bounds__(PLOC_TAIL) void k_ploc_tail(const uint* c_ptr, uint n_leaves, uint radius, const int* cref, const float* cbox, uint* alloc, int2* children,
                                                         float* node_box, uint* subtree_size, int* parent_internal, uint* rounds_out) {
    __shared__ float s_box[6][PLOC_TAIL];
    __shared__ int s_ref[PLOC_TAIL];
    __shared__ uint s_nn[PLOC_TAIL];
    __shared__ uint s_wave[PLOC_TAIL / 64];
    const uint i = threadIdx.x, lane = i & 63u, w = i >> 6;
    uint c = *c_ptr;
    if (i < c) {
        s_ref[i] = cref[i];
        for (int k = 0; k < 6; ++k) s_box[k][i] = cbox[6 * (size_t)i + k];
    }
    __syncthreads();
    uint rounds = 0;
    while (c > 1) {
        uint bj = i;
        float mine[6];
        if (i < c) {
            for (int k = 0; k < 6; ++k) mine[k] = s_box[k][i];
            const uint lo = i > radius ? i - radius : 0, hi = min(c - 1, i + radius);
            float best = __builtin_huge_valf();
            for (uint j = lo; j <= hi; ++j) {
                if (j == i) continue;
                const float other[6] = {s_box[0][j], s_box[1][j], s_box[2][j], s_box[3][j], s_box[4][j], s_box[5][j]};
                const float a = merged_area(mine, other);
                if (a < best) { best = a; bj = j; }
            }
        }
        s_nn[i] = bj;
        __syncthreads();
        bool keep = false;
        int ref = 0;
        if (i < c) {
            const uint j = bj;
            const bool mutual = j != i && s_nn[j] == i;
            if (!(mutual && i > j)) {
                keep = true;
                ref = s_ref[i];
                if (mutual) {
                    const int other = s_ref[j];
                    for (int k = 0; k < 3; ++k) { mine[k] = fminf(mine[k], s_box[k][j]); mine[3 + k] = fmaxf(mine[3 + k], s_box[3 + k][j]); }
                    const uint id = (n_leaves - 2u) - atomicAdd(alloc, 1u);
                    children[id] = make_int2(ref, other);
                    if (ref >= 0) parent_internal[ref] = (int)id;
                    if (other >= 0) parent_internal[other] = (int)id;
                    for (int k = 0; k < 6; ++k) node_box[6 * (size_t)id + k] = mine[k];
                    subtree_size[id] = (ref < 0 ? 1u : subtree_size[ref]) + (other < 0 ? 1u : subtree_size[other]);
                    ref = (int)id;
                }
            }
        }
        const unsigned long long kept = __ballot(keep);
        if (lane == 0) s_wave[w] = (uint)__popcll(kept);
        __syncthreads();   // every read of this round's clusters is done, and the subtree sizes written above are visible to the block
        uint before = 0, total = 0;
        for (uint k = 0; k < PLOC_TAIL / 64; ++k) { const uint v = s_wave[k]; total += v; before += k < w ? v : 0u; }
        if (keep) {
            const uint p = before + (uint)__popcll(kept & ((1ull << lane) - 1ull));
            s_ref[p] = ref;
            for (int k = 0; k < 6; ++k) s_box[k][p] = mine[k];
        }
        c = total;
        ++rounds;
        __syncthreads();
    }
    if (i == 0) *rounds_out = rounds;
}

__global__ __launch_bounds__(BT) void k_ploc_init(uint n, int* cref) {
    uint i = blockIdx.x * BT + threadIdx.x;
    if (i < n) cref[i] = ~(int)i;
}

__global__ __launch_bounds__(BT) void k_identity(uint n, int* v) {
    uint i = blockIdx.x * BT + threadIdx.x;
    if (i < n) v[i] = (int)i;
}

// Depth-first relabelling: a node's slot is its pre-order index among the internal nodes, so a left child sits right
// after its parent and every subtree is one contiguous run of 64-byte nodes (keeps the lower levels of one ray's walk
// within a few cache lines / one TLB page instead of scattered by merge order).
__global__ __launch_bounds__(BT) void k_dfs_order(uint n_internal, const int2* children, const uint* subtree_size, const int* parent_internal, int* new_id) {
    uint i = blockIdx.x * BT + threadIdx.x;
    if (i >= n_internal) return;
    uint pos = 0;
    int cur = (int)i;
    for (int p = parent_internal[cur]; p >= 0; cur = p, p = parent_internal[cur]) {
        const int2 ch = children[p];
        pos += 1u;
        if (ch.y == cur && ch.x >= 0) pos += subtree_size[ch.x] - 1u;   // skip the left subtree's internal nodes
    }
    new_id[i] = (int)pos;
}

// 64-byte traversal nodes from (children, boxes)
__global__ __launch_bounds__(BT) void k_emit_nodes(uint n_internal, const int2* children, const float* node_box, const float* leaf_box, const int* new_id,
                                                   BvhNode* nodes) {
    uint i = blockIdx.x * BT + threadIdx.x;
    if (i >= n_internal) return;
    const int2 ch = children[i];
    const float* b0 = ch.x >= 0 ? node_box + 6 * (size_t)ch.x : leaf_box + 6 * (size_t)(~ch.x);
    const float* b1 = ch.y >= 0 ? node_box + 6 * (size_t)ch.y : leaf_box + 6 * (size_t)(~ch.y);
    BvhNode out;
    for (int k = 0; k < 3; ++k) { out.lo0[k] = b0[k]; out.hi0[k] = b0[3 + k]; out.lo1[k] = b1[k]; out.hi1[k] = b1[3 + k]; }
    out.child0 = ch.x >= 0 ? new_id[ch.x] : ch.x;
    out.child1 = ch.y >= 0 ? new_id[ch.y] : ch.y;
    out.pad0 = 0; out.pad1 = 0;
    nodes[new_id[i]] = out;
}

}  // namespace
}  // namespace tr
#include "bvh_optimize.h"
namespace tr {
namespace {

// BVH2 -> BVH4, in place: every binary node keeps its (depth-first) index and adopts up to four descendants: the ones
// k_collapse_cost chose (`dec`: least sum of 4-wide node areas), or - dec = nullptr, TRHIP_COLLAPSE=greedy - found by
// repeatedly opening the adopted inner node with the largest box.  Nodes that were adopted away are never referenced
// again; they stay as dead 128-byte lines, which costs memory but neither bandwidth nor cache (a node is one line).
__global__ __launch_bounds__(BT) void k_collapse4(uint n_internal, const int2* children, const float* node_box, const float* leaf_box, const int* new_id,
                                                  const uint8_t* dec, Bvh4Node* nodes4) {
    uint i = blockIdx.x * BT + threadIdx.x;
    if (i >= n_internal) return;
    constexpr int W = 4;
    int cand[W];
    cand[0] = children[i].x; cand[1] = children[i].y;
    int ncand = 2;
    if (dec) ncand = collapse_children(children, dec, (int)i, cand);
    while (!dec && ncand < W) {
        int best = -1; float best_area = -1.0f;
        for (int c = 0; c < ncand; ++c) {
            if (cand[c] < 0) continue;
            const float* b = node_box + 6 * (size_t)cand[c];
            float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
            float area = dx * dy + dy * dz + dz * dx;
            if (area > best_area) { best_area = area; best = c; }
        }
        if (best < 0) break;
        const int2 ch = children[cand[best]];
        cand[best] = ch.x; cand[ncand++] = ch.y;
    }
    Bvh4Node out;
    for (int c = 0; c < 4; ++c) {
        if (c < ncand) {
            const float* b = cand[c] >= 0 ? node_box + 6 * (size_t)cand[c] : leaf_box + 6 * (size_t)(~cand[c]);
            out.lox[c] = b[0]; out.loy[c] = b[1]; out.loz[c] = b[2]; out.hix[c] = b[3]; out.hiy[c] = b[4]; out.hiz[c] = b[5];
            out.child[c] = cand[c] >= 0 ? new_id[cand[c]] : cand[c];
        } else {
            out.lox[c] = out.loy[c] = out.loz[c] = __builtin_huge_valf();
            out.hix[c] = out.hiy[c] = out.hiz[c] = -__builtin_huge_valf();
            out.child[c] = 0x7FFFFFFF;
        }
        out.pad[c] = 0;
    }
    nodes4[(size_t)new_id[i]] = out;
}

// shader/extract_tri_lights.comp:17-54 (all emissive instances in one launch)
__global__ __launch_bounds__(BT) void k_extract_tri_lights(SceneView sv, const uint* tri_prefix, TriLight* out) {
    uint gid = blockIdx.x * BT + threadIdx.x;
    if (gid >= sv.tri_count) return;
    uint inst, prim;
    locate_triangle(tri_prefix, sv.instance_count, gid, inst, prim);
    const Instance& o = sv.instances[inst];
    if (o.light_base_id < 0) return;
    const MeshSpan sp = sv.spans[inst];
    const uint* ix = sv.indices + sp.index_offset + 3u * prim;
    const Vertex* vb = sv.vertices + sp.vertex_offset;
    const Vertex v0 = vb[ix[0]], v1 = vb[ix[1]], v2 = vb[ix[2]];
    TriLight l;
    l.emission_tex_id = o.mat.emission_tex_id;
    l.emission_factor = rgb_to_r9g9b9e5(F3(o.mat.emission_factor));
    l.instance_id = inst;
    l.primitive_id = prim;
    l.pos[0] = transform_point(o.model, v0.pos);
    l.pos[1] = transform_point(o.model, v1.pos);
    l.pos[2] = transform_point(o.model, v2.pos);
    l.uv[0] = pack_half2x16(v0.uv); l.uv[1] = pack_half2x16(v1.uv); l.uv[2] = pack_half2x16(v2.uv);
    out[(uint)o.light_base_id + prim] = l;
}

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return set_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

}  // namespace

// shader/pre_transform.comp:26-42
__global__ __launch_bounds__(BT) void k_pre_transform_vertices(uint vertex_count, const Instance* instance, const Vertex* in, Vertex* out) {
    uint i = blockIdx.x * BT + threadIdx.x;
    if (i >= vertex_count) return;
    const m4 model = instance->model;
    const m3 mn = upper3(instance->model_normal);
    // determinant(mat3(o.model_normal)), cofactor expansion along the first column
    const float det = mn.c[0].x * (mn.c[1].y * mn.c[2].z - mn.c[2].y * mn.c[1].z)
                    - mn.c[1].x * (mn.c[0].y * mn.c[2].z - mn.c[2].y * mn.c[0].z)
                    + mn.c[2].x * (mn.c[0].y * mn.c[1].z - mn.c[1].y * mn.c[0].z);
    Vertex v = in[i];
    v.pos = transform_point(model, v.pos);
    v.normal = normalize(mul(mn, v.normal));
    f3 t = normalize(mul(mn, F3(v.tangent)));
    v.tangent = F4(t, v.tangent.w);
    if (det < 0) { v.normal = -v.normal; v.tangent = F4(-t.x, -t.y, -t.z, -v.tangent.w); }
    out[i] = v;
}

int ensure_world_vertices(DeviceScene& ds, hipStream_t stream) {
    if (ds.world_vertices || ds.world_vertex_count == 0) return 0;
    HIPCHK(hipMalloc(&ds.world_vertices, (size_t)ds.world_vertex_count * sizeof(Vertex)));
    for (uint i = 0; i < ds.instance_count; ++i) {   // one dispatch per instance, as src/scene_stage.cc:1685-1723 records them
        const MeshSpan& sp = ds.host_spans[i];
        if (sp.vertex_count == 0) continue;
        hipLaunchKernelGGL(k_pre_transform_vertices, dim3((sp.vertex_count + BT - 1) / BT), dim3(BT), 0, stream, sp.vertex_count, ds.instances + i,
                           ds.vertices + sp.vertex_offset, ds.world_vertices + ds.host_world_spans[i].vertex_offset);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// shader/skinning.comp:44-71, dispatched per skinned mesh by scene_stage::record_skinning (src/scene_stage.cc:1543-1567).
// skin_mat = sum w_k * joint[k]; positions go through skin_mat, normals and tangents through transpose(inverse(skin_mat)).
// GLSL leaves the arithmetic of inverse() to the driver; here it is the cofactor expansion over 2x2 sub-determinants
// (only the upper-left 3x3 of the inverse reaches a direction), identical in oracle/oracle.cc so results match bit for bit.
// The previous-position buffer the shader also fills is read by the raster path only (shader/forward.vert:28).
__global__ __launch_bounds__(BT) void k_skinning(uint vertex_count, const Vertex* source, const Skin* skins, const m4* joints, uint joint_count, Vertex* destination) {
    uint i = blockIdx.x * BT + threadIdx.x;
    if (i >= vertex_count) return;
    const Skin s = skins[i];
    float a[4][4];   // a[r][c]
    for (int c = 0; c < 4; ++c) {
        f4 col = F4(0);
        for (int k = 0; k < 4; ++k) {
            const uint j = s.joints[k] < joint_count ? s.joints[k] : 0u;   // out-of-range joint ids are undefined behaviour in the shader
            const f4 t = joints[j].c[c] * s.weights[k];
            col = k == 0 ? t : col + t;
        }
        a[0][c] = col.x; a[1][c] = col.y; a[2][c] = col.z; a[3][c] = col.w;
    }
    const float s0 = a[0][0] * a[1][1] - a[1][0] * a[0][1], s1 = a[0][0] * a[1][2] - a[1][0] * a[0][2], s2 = a[0][0] * a[1][3] - a[1][0] * a[0][3];
    const float s3 = a[0][1] * a[1][2] - a[1][1] * a[0][2], s4 = a[0][1] * a[1][3] - a[1][1] * a[0][3], s5 = a[0][2] * a[1][3] - a[1][2] * a[0][3];
    const float c5 = a[2][2] * a[3][3] - a[3][2] * a[2][3], c4 = a[2][1] * a[3][3] - a[3][1] * a[2][3], c3 = a[2][1] * a[3][2] - a[3][1] * a[2][2];
    const float c2 = a[2][0] * a[3][3] - a[3][0] * a[2][3], c1 = a[2][0] * a[3][2] - a[3][0] * a[2][2], c0 = a[2][0] * a[3][1] - a[3][0] * a[2][1];
    const float det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    float inv[3][3];   // inv[r][c], upper-left block of inverse(skin_mat)
    inv[0][0] = ( a[1][1] * c5 - a[1][2] * c4 + a[1][3] * c3) / det;
    inv[0][1] = (-a[0][1] * c5 + a[0][2] * c4 - a[0][3] * c3) / det;
    inv[0][2] = ( a[3][1] * s5 - a[3][2] * s4 + a[3][3] * s3) / det;
    inv[1][0] = (-a[1][0] * c5 + a[1][2] * c2 - a[1][3] * c1) / det;
    inv[1][1] = ( a[0][0] * c5 - a[0][2] * c2 + a[0][3] * c1) / det;
    inv[1][2] = (-a[3][0] * s5 + a[3][2] * s2 - a[3][3] * s1) / det;
    inv[2][0] = ( a[1][0] * c4 - a[1][1] * c2 + a[1][3] * c0) / det;
    inv[2][1] = (-a[0][0] * c4 + a[0][1] * c2 - a[0][3] * c0) / det;
    inv[2][2] = ( a[3][0] * s4 - a[3][1] * s2 + a[3][3] * s0) / det;
    const Vertex src = source[i];
    Vertex dst = src;
    dst.pos = F3(a[0][0] * src.pos.x + a[0][1] * src.pos.y + a[0][2] * src.pos.z + a[0][3],
                 a[1][0] * src.pos.x + a[1][1] * src.pos.y + a[1][2] * src.pos.z + a[1][3],
                 a[2][0] * src.pos.x + a[2][1] * src.pos.y + a[2][2] * src.pos.z + a[2][3]);
    // transpose(inverse(M)) * (d, 0): component r = sum_c inv[c][r] * d_c
    const f3 n = src.normal, t = F3(src.tangent);
    // (the w = 0 column of the 4x4 product contributes +0: keeps the sign of a zero component as GLSL has it)
    dst.normal = normalize(F3(inv[0][0] * n.x + inv[1][0] * n.y + inv[2][0] * n.z + 0.0f, inv[0][1] * n.x + inv[1][1] * n.y + inv[2][1] * n.z + 0.0f,
                              inv[0][2] * n.x + inv[1][2] * n.y + inv[2][2] * n.z + 0.0f));
    const f3 tt = normalize(F3(inv[0][0] * t.x + inv[1][0] * t.y + inv[2][0] * t.z + 0.0f, inv[0][1] * t.x + inv[1][1] * t.y + inv[2][1] * t.z + 0.0f,
                               inv[0][2] * t.x + inv[1][2] * t.y + inv[2][2] * t.z + 0.0f));
    dst.tangent = F4(tt, src.tangent.w);
    destination[i] = dst;
}

__global__ __launch_bounds__(BT) void k_build_shade_tris(uint n_spans, const MeshSpan* spans, const Vertex* vertices, const uint* indices, ShadeTri* out, f4* tangents) {
    const MeshSpan sp = spans[blockIdx.y];
    const Vertex* vb = vertices + sp.vertex_offset;
    const uint* ix = indices + sp.index_offset;
    ShadeTri* o = out + sp.index_offset / 3u;
    f4* ot = tangents + (size_t)sp.index_offset;      // three per record: index_offset / 3 * 3
    for (uint t = blockIdx.x * BT + threadIdx.x; t < sp.triangle_count; t += gridDim.x * BT) {
        ShadeTri r;
        for (int k = 0; k < 3; ++k) {
            const Vertex v = vb[ix[3 * t + k]];
            r.pos[k] = v.pos; r.normal[k] = v.normal; r.uv[k] = v.uv;
            ot[3 * t + k] = v.tangent;
        }
        for (float& x : r.pad) x = 0.0f;
        o[t] = r;
    }
}

// The per-triangle vertex records k_shade reads (common.h ShadeTri).  A record is addressed by index_offset / 3 + primitive, so every
// span must start at a whole triangle and two spans over the same indices must use the same vertices; a scene that does not
// (none the loaders produce) simply has no records and is shaded by the general kernels.
int build_shade_tris(DeviceScene& ds, int instance, hipStream_t stream) {
    if (instance < 0) {
        if (ds.shade_tris) { (void)hipFree(ds.shade_tris); ds.shade_tris = nullptr; ds.shade_tangents = nullptr; }
        if (ds.index_count < 3 || ds.instance_count == 0 || getenv("TRHIP_NO_SHADE_TRIS")) return 0;
        std::vector<MeshSpan> unique;
        {
            std::vector<MeshSpan> sorted(ds.host_spans);
            std::sort(sorted.begin(), sorted.end(), [](const MeshSpan& a, const MeshSpan& b) { return a.index_offset != b.index_offset ? a.index_offset < b.index_offset : a.triangle_count > b.triangle_count; });
            uint64_t end = 0;      // first index not covered by the records so far
            for (const MeshSpan& sp : sorted) {
                if (sp.triangle_count == 0) continue;
                if (sp.index_offset % 3u != 0) return 0;
                if (!unique.empty() && sp.index_offset < end) {      // overlaps the previous mesh: fine if it is (part of) the same mesh
                    const MeshSpan& u = unique.back();
                    if (sp.vertex_offset != u.vertex_offset || (sp.index_offset - u.index_offset) % 3u != 0 || (uint64_t)sp.index_offset + 3ull * sp.triangle_count > end) return 0;
                    continue;
                }
                unique.push_back(sp);
                end = (uint64_t)sp.index_offset + 3ull * sp.triangle_count;
            }
        }
        if (unique.empty()) return 0;
        {   // records, then the tangents (three f4 per record) in the same allocation
            const size_t n_rec = ds.index_count / 3u;
            HIPCHK(hipMalloc(&ds.shade_tris, n_rec * (sizeof(ShadeTri) + 3 * sizeof(f4))));
            ds.shade_tangents = reinterpret_cast<f4*>(ds.shade_tris + n_rec);
        }
        MeshSpan* dev_spans = nullptr;
        HIPCHK(hipMalloc(&dev_spans, unique.size() * sizeof(MeshSpan)));
        HIPCHK(hipMemcpyAsync(dev_spans, unique.data(), unique.size() * sizeof(MeshSpan), hipMemcpyHostToDevice, stream));
        uint most = 0;
        for (const MeshSpan& sp : unique) most = std::max(most, sp.triangle_count);
        for (size_t first = 0; first < unique.size(); first += 65535u) {
            const uint count = (uint)std::min<size_t>(65535u, unique.size() - first);
            hipLaunchKernelGGL(k_build_shade_tris, dim3(std::min((most + BT - 1) / BT, 1024u), count), dim3(BT), 0, stream, count, dev_spans + first, ds.vertices, ds.indices, ds.shade_tris, ds.shade_tangents);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        (void)hipFree(dev_spans);
        return 0;
    }
    if (!ds.shade_tris) return 0;
    // one mesh again (its vertices were skinned): the span of the instance, read from the device copy of the spans
    const MeshSpan& sp = ds.host_spans[(size_t)instance];
    if (sp.triangle_count == 0) return 0;
    hipLaunchKernelGGL(k_build_shade_tris, dim3(std::min((sp.triangle_count + BT - 1) / BT, 1024u), 1), dim3(BT), 0, stream, 1u, ds.spans + instance, ds.vertices, ds.indices, ds.shade_tris, ds.shade_tangents);
    HIPCHK(hipGetLastError());
    return 0;
}

int skin_instance(DeviceScene& ds, uint instance, const float* joint_transforms, uint joint_count, hipStream_t stream) {
    if (instance >= ds.skin_slots.size() || !ds.skin_slots[instance].source) return set_error("trhip_scene_skin: instance has no skin (trhip_scene_set_skin)");
    DeviceScene::SkinSlot& k = ds.skin_slots[instance];
    if (joint_count == 0 || !joint_transforms) return set_error("trhip_scene_skin: no joint transforms");
    if (joint_count > k.joint_capacity) {
        if (k.joints) (void)hipFree(k.joints);
        k.joints = nullptr; k.joint_capacity = 0;
        HIPCHK(hipMalloc(&k.joints, (size_t)joint_count * sizeof(m4)));
        k.joint_capacity = joint_count;
    }
    HIPCHK(hipMemcpyAsync(k.joints, joint_transforms, (size_t)joint_count * sizeof(m4), hipMemcpyHostToDevice, stream));
    const MeshSpan& sp = ds.host_spans[instance];
    hipLaunchKernelGGL(k_skinning, dim3((k.vertex_count + BT - 1) / BT), dim3(BT), 0, stream, k.vertex_count, k.source, k.skins, k.joints, joint_count,
                       ds.vertices + sp.vertex_offset);
    HIPCHK(hipGetLastError());
    if (int rc = build_shade_tris(ds, (int)instance, stream)) return rc;
    HIPCHK(hipStreamSynchronize(stream));   // the host array may be reused by the caller
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Refit: the instance transforms changed, the tree keeps its topology (what a BLAS/TLAS *update* does in the reference,
// src/acceleration_structure.cc:376-422).  World triangles are recomputed in place, then the child boxes of the live
// nodes are rebuilt level by level from the deepest level up.
__global__ __launch_bounds__(BT) void k_retransform(SceneView sv, uint n, TriRecord* tris, const uint* alpha_base, AlphaTri* alpha_tris) {
    uint i = blockIdx.x * BT + threadIdx.x;
    if (i >= n) return;
    TriRecord t = tris[i];
    const uint inst = t.inst_flags & 0x7FFFFFFFu;
    const MeshSpan sp = sv.obj_spans[inst];
    const uint* ix = sv.indices + sp.index_offset + 3u * t.prim;
    const Vertex* vb = sv.obj_vertices + sp.vertex_offset;
    const m4 model = sv.instances[inst].model;
    const f3 p0 = transform_point(model, vb[ix[0]].pos), p1 = transform_point(model, vb[ix[1]].pos), p2 = transform_point(model, vb[ix[2]].pos);
    t.v0[0] = p0.x; t.v0[1] = p0.y; t.v0[2] = p0.z;
    t.v1[0] = p1.x; t.v1[1] = p1.y; t.v1[2] = p1.z;
    t.v2[0] = p2.x; t.v2[1] = p2.y; t.v2[2] = p2.z;
    if (t.inst_flags & 0x80000000u) t.alpha = alpha_word(sv.instances[inst].mat, alpha_base[inst] + t.prim, alpha_tris, vb[ix[0]].uv, vb[ix[1]].uv, vb[ix[2]].uv);
    tris[i] = t;
}

// breadth-first expansion of one level of live nodes (run once per build, on the first refit)
__global__ __launch_bounds__(BT) void k_expand_level(uint count, const uint* level, const Bvh4Node* nodes4, uint* next, uint* next_count) {
    uint i = blockIdx.x * BT + threadIdx.x;
    if (i >= count) return;
    const Bvh4Node& nd = nodes4[level[i]];
    for (int c = 0; c < 4; ++c) {
        const int ch = nd.child[c];
        if (ch >= 0 && ch != 0x7FFFFFFF) next[atomicAdd(next_count, 1u)] = (uint)ch;
    }
}

__global__ __launch_bounds__(BT) void k_refit_level(uint count, const uint* level, Bvh4Node* nodes4, const TriRecord* tris, float* node_bounds) {
    uint i = blockIdx.x * BT + threadIdx.x;
    if (i >= count) return;
    const uint id = level[i];
    Bvh4Node nd = nodes4[id];
    float lo[3] = {__builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf()};
    float hi[3] = {-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf()};
    for (int c = 0; c < 4; ++c) {
        const int ch = nd.child[c];
        if (ch == 0x7FFFFFFF) continue;
        float blo[3], bhi[3];
        if (ch < 0) {
            const TriRecord& t = tris[~ch];
            for (int k = 0; k < 3; ++k) { blo[k] = fminf(fminf(t.v0[k], t.v1[k]), t.v2[k]); bhi[k] = fmaxf(fmaxf(t.v0[k], t.v1[k]), t.v2[k]); }
        } else {
            for (int k = 0; k < 3; ++k) { blo[k] = node_bounds[6 * (size_t)ch + k]; bhi[k] = node_bounds[6 * (size_t)ch + 3 + k]; }
        }
        nd.lox[c] = blo[0]; nd.loy[c] = blo[1]; nd.loz[c] = blo[2]; nd.hix[c] = bhi[0]; nd.hiy[c] = bhi[1]; nd.hiz[c] = bhi[2];
        for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], blo[k]); hi[k] = fmaxf(hi[k], bhi[k]); }
    }
    nodes4[id] = nd;
    for (int k = 0; k < 3; ++k) { node_bounds[6 * (size_t)id + k] = lo[k]; node_bounds[6 * (size_t)id + 3 + k] = hi[k]; }
}

int refit_accel(DeviceScene& ds, hipStream_t stream, trhip_accel_info* info) {
    const uint n = ds.leaf_count;      // records in ds.tris: triangles, or the references of a pre-split build (their leaf boxes grow to the whole triangle here)
    if (ds.accel_tri_count != ds.tri_count || ds.accel_capacity == 0xFFFFFFFFu || (n > 0 && !ds.tris) || (n > 1 && !ds.nodes4))
        return set_error("trhip_scene_refit_accel: no acceleration structure to refit; call trhip_scene_build_accel first");
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, stream));
    SceneView sv = ds.view();
    const uint n1 = ds.node_count;
    if (n1 > 0 && !ds.levels_valid) {   // once per build: live nodes by level
        if (!ds.level_nodes) HIPCHK(hipMalloc(&ds.level_nodes, (size_t)n1 * 4));
        if (!ds.node_bounds) HIPCHK(hipMalloc(&ds.node_bounds, (size_t)n1 * 24));
        uint* counter = nullptr;
        HIPCHK(hipMalloc(&counter, 4));
        const uint root = 0;
        HIPCHK(hipMemcpyAsync(ds.level_nodes, &root, 4, hipMemcpyHostToDevice, stream));
        ds.level_offsets.assign(1, 0u);
        uint begin = 0, count = 1;
        while (count > 0) {
            ds.level_offsets.push_back(begin + count);
            if (begin + count >= n1) break;   // all node slots used: nothing can follow
            HIPCHK(hipMemsetAsync(counter, 0, 4, stream));
            hipLaunchKernelGGL(k_expand_level, dim3((count + BT - 1) / BT), dim3(BT), 0, stream, count, ds.level_nodes + begin, ds.nodes4,
                               ds.level_nodes + begin + count, counter);
            uint next = 0;
            HIPCHK(hipMemcpyAsync(&next, counter, 4, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            begin += count; count = next;
            if (ds.level_offsets.size() > 4096) { (void)hipFree(counter); return set_error("trhip_scene_refit_accel: hierarchy too deep"); }
        }
        (void)hipFree(counter);
        ds.levels_valid = true;
        if (getenv("TRHIP_DEBUG"))
            fprintf(stderr, "[trhip] 4-wide tree: %u live nodes of %u slots in %zu levels, %.3f children per node\n", ds.level_offsets.back(), n1,
                    ds.level_offsets.size() - 1, (double)(n + ds.level_offsets.back() - 1) / (double)ds.level_offsets.back());
    }
    if (n > 0) hipLaunchKernelGGL(k_retransform, dim3((n + BT - 1) / BT), dim3(BT), 0, stream, sv, n, ds.tris, ds.alpha_base, ds.alpha_tris);
    for (size_t l = ds.level_offsets.size(); l-- > 1;) {
        const uint lo = ds.level_offsets[l - 1], cnt = ds.level_offsets[l] - lo;
        if (cnt) hipLaunchKernelGGL(k_refit_level, dim3((cnt + BT - 1) / BT), dim3(BT), 0, stream, cnt, ds.level_nodes + lo, ds.nodes4, ds.tris, ds.node_bounds);
    }
    HIPCHK(hipGetLastError());
    ds.accel_built = true;
    if (ds.gather_emissive_triangles && ds.host_tri_light_count > 0 && ds.tri_lights) {
        HIPCHK(hipMemsetAsync(ds.tri_lights, 0, (size_t)ds.host_tri_light_count * sizeof(TriLight), stream));
        SceneView sv2 = ds.view();
        hipLaunchKernelGGL(k_extract_tri_lights, dim3((ds.tri_count + BT - 1) / BT), dim3(BT), 0, stream, sv2, ds.tri_prefix, ds.tri_lights);
    }
    HIPCHK(hipEventRecord(e1, stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (info) {
        memset(info, 0, sizeof(*info));
        info->triangle_count = ds.tri_count; info->leaf_count = n; info->node_count = ds.node_count; info->node_bytes = 112u; info->tri_light_count = ds.tri_light_count;
        info->build_ms = ms;
        for (int k = 0; k < 3; ++k) { info->bounds_min[k] = ds.bounds_lo[k]; info->bounds_max[k] = ds.bounds_hi[k]; }
    }
    return 0;
}

// Experiment (TRHIP_TREELET=<nodes per treelet>, off by default; profiles/r5/treelet_order_ab.txt): the live 4-wide nodes laid out treelet by
// treelet - a treelet is grown from its root by always opening the hit-likeliest (largest) child box until it holds the given number of
// nodes, its nodes are stored side by side, the subtrees hanging off it follow depth-first - and the triangle records in the order the new
// node array refers to them, so the leaves of a node are neighbours in memory.  Dead lines (binary nodes the collapse adopted away) drop
// out.  A host pass over a downloaded tree: build time is not the point of the experiment.  Hits do not depend on the layout.
static int reorder_into_treelets(DeviceScene& ds, hipStream_t stream, uint n_nodes, uint n_tris, uint treelet_nodes, bool reorder_tris) {
    if (n_nodes == 0 || n_tris == 0) return 0;
    HIPCHK(hipStreamSynchronize(stream));
    std::vector<Bvh4Node> nodes(n_nodes), out_nodes;
    std::vector<TriRecord> tris(n_tris), out_tris;
    HIPCHK(hipMemcpy(nodes.data(), ds.nodes4, (size_t)n_nodes * sizeof(Bvh4Node), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tris.data(), ds.tris, (size_t)n_tris * sizeof(TriRecord), hipMemcpyDeviceToHost));
    std::vector<int> new_id(n_nodes, -1);
    std::vector<uint> order;            // old node ids in their new order
    order.reserve(n_nodes);
    std::vector<uint> roots = {0u};     // stack of treelet roots
    auto area = [&](const Bvh4Node& nd, int c) {
        const float dx = nd.hix[c] - nd.lox[c], dy = nd.hiy[c] - nd.loy[c], dz = nd.hiz[c] - nd.loz[c];
        return dx * dy + dy * dz + dz * dx;
    };
    std::vector<std::pair<float, uint>> heap;
    while (!roots.empty()) {
        const uint root = roots.back(); roots.pop_back();
        heap.clear();
        heap.push_back({__builtin_huge_valf(), root});
        uint taken = 0;
        std::vector<uint> rest;
        while (!heap.empty()) {
            std::pop_heap(heap.begin(), heap.end());
            const uint nd = heap.back().second; heap.pop_back();
            if (taken >= treelet_nodes) { rest.push_back(nd); continue; }
            new_id[nd] = (int)order.size(); order.push_back(nd); taken++;
            for (int c = 0; c < 4; ++c) {
                const int ch = nodes[nd].child[c];
                if (ch >= 0 && ch != 0x7FFFFFFF) { heap.push_back({area(nodes[nd], c), (uint)ch}); std::push_heap(heap.begin(), heap.end()); }
            }
        }
        // the subtrees below the treelet, the largest box last so that it is laid out next
        for (size_t k = rest.size(); k-- > 0;) roots.push_back(rest[k]);
    }
    out_nodes.resize(n_nodes);
    std::vector<int> new_tri(n_tris, -1);
    uint next_tri = 0;
    for (size_t k = 0; k < order.size(); ++k) {
        Bvh4Node nd = nodes[order[k]];
        for (int c = 0; c < 4; ++c) {
            const int ch = nd.child[c];
            if (ch == 0x7FFFFFFF) continue;
            if (ch >= 0) nd.child[c] = new_id[ch];
            else if (reorder_tris) { const uint t = (uint)~ch; if (new_tri[t] < 0) new_tri[t] = (int)next_tri++; nd.child[c] = ~new_tri[t]; }
        }
        out_nodes[k] = nd;
    }
    for (size_t k = order.size(); k < n_nodes; ++k) { out_nodes[k] = nodes[0]; }   // never referenced
    HIPCHK(hipMemcpy(ds.nodes4, out_nodes.data(), (size_t)n_nodes * sizeof(Bvh4Node), hipMemcpyHostToDevice));
    if (reorder_tris) {
        out_tris.resize(n_tris);
        for (uint t = 0; t < n_tris; ++t) { if (new_tri[t] < 0) new_tri[t] = (int)next_tri++; out_tris[new_tri[t]] = tris[t]; }
        HIPCHK(hipMemcpy(ds.tris, out_tris.data(), (size_t)n_tris * sizeof(TriRecord), hipMemcpyHostToDevice));
    }
    if (getenv("TRHIP_DEBUG")) fprintf(stderr, "[trhip] treelet layout: %zu live nodes of %u in treelets of %u, triangles %s\n", order.size(), n_nodes, treelet_nodes, reorder_tris ? "in node order" : "as built");
    return 0;
}

// Experiment (TRHIP_PAIR_LEAVES=1 with a library built -DTR_PAIR_LEAVES=1; profiles/r5/pair_leaves_ab.txt): leaves of two triangles.
// A host pass over the finished 4-wide tree, bottom-up: the leaf slots of a node are paired (the two whose union box is smallest first), a
// pair takes one slot - reference ~(first | 1 << 30), its triangles adjacent in the record array - and the slots that frees are filled by
// adopting the children of inner children that fit, which removes those nodes and a level of the walk above them.  Nodes and records are
// then renumbered in depth-first order.  Refit does not understand pair references: the experiment is for static scenes.
static int pair_leaves_postpass(DeviceScene& ds, hipStream_t stream, uint n_nodes, uint n_tris) {
    if (n_nodes == 0 || n_tris < 2) return 0;
    HIPCHK(hipStreamSynchronize(stream));
    std::vector<Bvh4Node> nodes(n_nodes);
    std::vector<TriRecord> tris(n_tris);
    HIPCHK(hipMemcpy(nodes.data(), ds.nodes4, (size_t)n_nodes * sizeof(Bvh4Node), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tris.data(), ds.tris, (size_t)n_tris * sizeof(TriRecord), hipMemcpyDeviceToHost));
    struct Slot { float lo[3], hi[3]; int kind; int a, b; };      // kind 0 inner node a, 1 triangle a, 2 pair (a, b)
    std::vector<std::vector<Slot>> slots(n_nodes);
    std::vector<char> live(n_nodes, 0), dead(n_nodes, 0);
    {   // live nodes: reachable from the root
        std::vector<uint> st = {0u};
        while (!st.empty()) {
            const uint nd = st.back(); st.pop_back();
            live[nd] = 1;
            for (int c = 0; c < 4; ++c) {
                const int ch = nodes[nd].child[c];
                if (ch == 0x7FFFFFFF) continue;
                Slot sl;
                sl.lo[0] = nodes[nd].lox[c]; sl.lo[1] = nodes[nd].loy[c]; sl.lo[2] = nodes[nd].loz[c];
                sl.hi[0] = nodes[nd].hix[c]; sl.hi[1] = nodes[nd].hiy[c]; sl.hi[2] = nodes[nd].hiz[c];
                sl.kind = ch >= 0 ? 0 : 1; sl.a = ch >= 0 ? ch : ~ch; sl.b = -1;
                slots[nd].push_back(sl);
                if (ch >= 0) st.push_back((uint)ch);
            }
        }
    }
    auto area_of = [](const float* lo, const float* hi) { const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; return dx * dy + dy * dz + dz * dx; };
    size_t pairs = 0, adopted = 0;
    for (uint k = n_nodes; k-- > 0;) {      // children sit behind their parents (depth-first order): bottom-up
        if (!live[k]) continue;
        std::vector<Slot>& S = slots[k];
        while (true) {      // pair leaf slots, cheapest union first
            int bi = -1, bj = -1; float best = __builtin_huge_valf();
            for (size_t i = 0; i < S.size(); ++i) for (size_t j = i + 1; j < S.size(); ++j) {
                if (S[i].kind != 1 || S[j].kind != 1) continue;
                float lo[3], hi[3];
                for (int a = 0; a < 3; ++a) { lo[a] = std::min(S[i].lo[a], S[j].lo[a]); hi[a] = std::max(S[i].hi[a], S[j].hi[a]); }
                const float ar = area_of(lo, hi);
                if (ar < best) { best = ar; bi = (int)i; bj = (int)j; }
            }
            if (bi < 0) break;
            for (int a = 0; a < 3; ++a) { S[bi].lo[a] = std::min(S[bi].lo[a], S[bj].lo[a]); S[bi].hi[a] = std::max(S[bi].hi[a], S[bj].hi[a]); }
            S[bi].kind = 2; S[bi].b = S[bj].a;
            S.erase(S.begin() + bj);
            pairs++;
        }
        while (true) {      // adopt the children of an inner child that fit into the free slots, largest box first
            int pick = -1; float best = -1.0f;
            for (size_t i = 0; i < S.size(); ++i) {
                if (S[i].kind != 0) continue;
                const std::vector<Slot>& C = slots[(size_t)S[i].a];
                if (S.size() - 1 + C.size() > 4) continue;
                const float ar = area_of(S[i].lo, S[i].hi);
                if (ar > best) { best = ar; pick = (int)i; }
            }
            if (pick < 0) break;
            const int child = S[pick].a;
            const std::vector<Slot> C = slots[(size_t)child];
            S.erase(S.begin() + pick);
            S.insert(S.end(), C.begin(), C.end());
            dead[(size_t)child] = 1;
            adopted++;
        }
    }
    // renumber depth-first
    std::vector<int> new_id(n_nodes, -1);
    std::vector<uint> order;
    {
        std::vector<uint> st = {0u};
        while (!st.empty()) {
            const uint nd = st.back(); st.pop_back();
            new_id[nd] = (int)order.size(); order.push_back(nd);
            for (size_t i = slots[nd].size(); i-- > 0;) if (slots[nd][i].kind == 0) st.push_back((uint)slots[nd][i].a);
        }
    }
    std::vector<Bvh4Node> out_nodes(n_nodes, nodes[0]);
    std::vector<TriRecord> out_tris(n_tris);
    uint next_tri = 0;
    for (size_t k = 0; k < order.size(); ++k) {
        Bvh4Node nd;
        const std::vector<Slot>& S = slots[order[k]];
        for (int c = 0; c < 4; ++c) {
            nd.pad[c] = 0;
            if ((size_t)c >= S.size()) {
                nd.lox[c] = nd.loy[c] = nd.loz[c] = __builtin_huge_valf(); nd.hix[c] = nd.hiy[c] = nd.hiz[c] = -__builtin_huge_valf(); nd.child[c] = 0x7FFFFFFF;
                continue;
            }
            const Slot& sl = S[(size_t)c];
            nd.lox[c] = sl.lo[0]; nd.loy[c] = sl.lo[1]; nd.loz[c] = sl.lo[2]; nd.hix[c] = sl.hi[0]; nd.hiy[c] = sl.hi[1]; nd.hiz[c] = sl.hi[2];
            if (sl.kind == 0) nd.child[c] = new_id[(size_t)sl.a];
            else {
                out_tris[next_tri] = tris[(size_t)sl.a];
                if (sl.kind == 2) { out_tris[next_tri + 1] = tris[(size_t)sl.b]; nd.child[c] = ~(int)(next_tri | 0x40000000u); next_tri += 2; }
                else { nd.child[c] = ~(int)next_tri; next_tri += 1; }
            }
        }
        out_nodes[k] = nd;
    }
    if (next_tri != n_tris) return set_error("pair leaves: " + std::to_string(next_tri) + " of " + std::to_string(n_tris) + " triangles referenced");
    HIPCHK(hipMemcpy(ds.nodes4, out_nodes.data(), (size_t)n_nodes * sizeof(Bvh4Node), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ds.tris, out_tris.data(), (size_t)n_tris * sizeof(TriRecord), hipMemcpyHostToDevice));
    fprintf(stderr, "[trhip] pair leaves: %zu pairs, %zu nodes adopted away, %zu live nodes of %u\n", pairs, adopted, order.size(), n_nodes);
    return 0;
}

int build_accel(DeviceScene& ds, hipStream_t stream, trhip_accel_info* info) {
    const uint n_scene = ds.tri_count;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, stream));
    ds.accel_built = false;
    ds.levels_valid = false;   // a new tree: the refit level lists are rebuilt on demand
    SceneView sv = ds.view();
    sv.tri_count = n_scene;
    // Temporaries come out of one scratch arena that survives the call, and the outputs keep their allocation while the
    // triangle count does not change: a rebuild (dynamic scenes) performs no allocation at all.
    size_t plan_bytes = 0;
    auto plan = [&](size_t bytes) { size_t o = plan_bytes; plan_bytes += (bytes + 255) & ~(size_t)255; return o; };
    // n_tri = triangles = leaves; n_cap = what everything is sized for
    const uint n_tri = n_scene;
    const uint n_cap = n_tri;
    uint n = n_tri;
    const size_t n1 = n_cap > 1 ? n_cap - 1 : 0;     // inner nodes the buffers hold
    size_t sort_bytes = 0, scan_bytes = 0;
    if (n_cap > 0) {
        HIPCHK(rocprim::radix_sort_pairs(nullptr, sort_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (uint*)nullptr, (uint*)nullptr, n_cap, 0, 64, stream));
        HIPCHK(rocprim::exclusive_scan(nullptr, scan_bytes, (uint*)nullptr, (uint*)nullptr, 0u, n_cap + BT, rocprim::plus<uint>(), stream));
    }
    const size_t o_cbounds = plan(32 * sizeof(uint)), o_unsorted = plan((size_t)n_cap * sizeof(TriRecord)), o_keys = plan((size_t)n_cap * 8),
                 o_keys_sorted = plan((size_t)n_cap * 8), o_vals = plan((size_t)n_cap * 4), o_vals_sorted = plan((size_t)n_cap * 4),
                 o_leaf_box = plan((size_t)n_cap * 24), o_sort = plan(sort_bytes + 16), o_children = plan(n1 * sizeof(int2)),
                 o_sizes = plan(n1 * 4), o_parent = plan(n1 * 4), o_parent_leaf = plan((size_t)n_cap * 4), o_node_box = plan(n1 * 24),
                 o_arrive = plan(n1 * 4), o_cref0 = plan((size_t)n_cap * 4), o_cref1 = plan((size_t)n_cap * 4), o_cbox0 = plan((size_t)n_cap * 24),
                 o_cbox1 = plan((size_t)n_cap * 24), o_nn = plan((size_t)n_cap * 4), o_valid = plan(((size_t)n_cap + BT) * 4), o_pos = plan(((size_t)n_cap + BT) * 4),
                 o_scan = plan(scan_bytes + 16), o_new_id = plan(n1 * 4), o_nodes2 = plan(n1 * sizeof(BvhNode));
    const bool optimise = ds.optimise_rounds > 0 && !ds.fast_build && n_tri > 2;
    const size_t n_all_cap = (size_t)n_cap + n1;
    const bool dp_collapse = ds.collapse_by_cost && !ds.fast_build && n_tri > 2;   // a fast build keeps the greedy choice (the cost pass would double its time)
    const size_t o_uparent = plan(optimise || dp_collapse ? n_all_cap * 4 : 0), o_moves = plan(optimise ? n_all_cap * sizeof(OptMove) : 0), o_lock = plan(optimise ? n_all_cap * 8 : 0),
                 o_optstat = plan(64), o_ccost = plan(dp_collapse ? n1 * 12 : 0), o_cdec = plan(dp_collapse ? n1 : 0);
    if (plan_bytes > ds.scratch_bytes) {
        if (ds.scratch) (void)hipFree(ds.scratch);
        ds.scratch = nullptr; ds.scratch_bytes = 0;
        HIPCHK(hipMalloc(&ds.scratch, plan_bytes));
        ds.scratch_bytes = plan_bytes;
    }
    char* base = static_cast<char*>(ds.scratch);
    uint* cbounds = reinterpret_cast<uint*>(base + o_cbounds);   // 6 flipped centroid bounds, [6] PLOC node allocator, [8..9] cluster counts
    {
        const uint init[32] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(cbounds, init, sizeof(init), hipMemcpyHostToDevice, stream));
    }
    if (ds.accel_capacity != n_cap) {   // outputs
        ds.free_accel();
        if (n_cap > 0) HIPCHK(hipMalloc(&ds.tris, (size_t)n_cap * sizeof(TriRecord)));
        if (n1 > 0) {
            // traversal addresses a node's planes with 32-bit byte offsets (node << 7 | plane)
            if ((uint64_t)n1 * sizeof(Bvh4Node) > 0xFFFFFFFFull) return set_error("trhip_scene_build_accel: more than 2^25 nodes");
            HIPCHK(hipMalloc(&ds.nodes4, n1 * sizeof(Bvh4Node)));
        }
        ds.accel_capacity = n_cap;
    }
    if (n_tri > 0) {
        TriRecord* unsorted = reinterpret_cast<TriRecord*>(base + o_unsorted);
        unsigned long long *keys = reinterpret_cast<unsigned long long*>(base + o_keys), *keys_sorted = reinterpret_cast<unsigned long long*>(base + o_keys_sorted);
        uint *vals = reinterpret_cast<uint*>(base + o_vals), *vals_sorted = reinterpret_cast<uint*>(base + o_vals_sorted);
        float *leaf_box = reinterpret_cast<float*>(base + o_leaf_box), *node_box = reinterpret_cast<float*>(base + o_node_box);
        int2* children = reinterpret_cast<int2*>(base + o_children);
        uint* ranges = reinterpret_cast<uint*>(base + o_sizes);   // subtree sizes (leaves per internal node)
        int *parent_internal = reinterpret_cast<int*>(base + o_parent), *parent_leaf = reinterpret_cast<int*>(base + o_parent_leaf);
        uint* arrive = reinterpret_cast<uint*>(base + o_arrive);
        BvhNode* nodes2 = reinterpret_cast<BvhNode*>(base + o_nodes2);   // binary nodes: only the collapse input
        const uint tblocks = (n_tri + BT - 1) / BT;
        hipLaunchKernelGGL(k_pretransform, dim3(tblocks < 1024u ? tblocks : 1024u), dim3(BT), 0, stream, sv, ds.tri_prefix, ds.non_opaque, ds.alpha_base, ds.alpha_tris, unsorted, cbounds);
        const uint blocks = (n + BT - 1) / BT;
        hipLaunchKernelGGL(k_morton, dim3(blocks), dim3(BT), 0, stream, n, unsorted, cbounds, keys, vals);
        HIPCHK(rocprim::radix_sort_pairs(base + o_sort, sort_bytes, keys, keys_sorted, vals, vals_sorted, n, 0, 64, stream));
        hipLaunchKernelGGL(k_gather_leaves, dim3(blocks), dim3(BT), 0, stream, n, unsorted, vals_sorted, ds.tris, leaf_box);
        const size_t n_all = (size_t)n + (n > 1 ? n - 1 : 0);
        if (n > 1) {
            const uint iblocks = (n - 1 + BT - 1) / BT;
            if (ds.builder == 0) {
                // Karras 2012 LBVH + atomic bottom-up refit
                HIPCHK(hipMemsetAsync(arrive, 0, (size_t)(n - 1) * 4, stream));
                hipLaunchKernelGGL(k_hierarchy, dim3(iblocks), dim3(BT), 0, stream, (int)n, keys_sorted, children, ranges, parent_internal, parent_leaf);
                hipLaunchKernelGGL(k_refit, dim3(blocks), dim3(BT), 0, stream, (int)n, children, parent_internal, parent_leaf, leaf_box,
                                   node_box, arrive, nodes2);
            } else {
                // PLOC over the Morton order
                int* cref[2] = {reinterpret_cast<int*>(base + o_cref0), reinterpret_cast<int*>(base + o_cref1)};
                float* cbox[2] = {reinterpret_cast<float*>(base + o_cbox0), reinterpret_cast<float*>(base + o_cbox1)};
                uint *nn = reinterpret_cast<uint*>(base + o_nn), *valid = reinterpret_cast<uint*>(base + o_valid), *pos = reinterpret_cast<uint*>(base + o_pos);
                uint *alloc = cbounds + 6, *c_dev = cbounds + 8;   // count of round r in c_dev[r & 1]; k_ploc_compact writes the other one
                HIPCHK(hipMemsetAsync(parent_internal, 0xFF, (size_t)(n - 1) * 4, stream));   // root keeps -1
                hipLaunchKernelGGL(k_ploc_init, dim3(blocks), dim3(BT), 0, stream, n, cref[0]);
                HIPCHK(hipMemcpyAsync(cbox[0], leaf_box, (size_t)n * 24, hipMemcpyDeviceToDevice, stream));
                HIPCHK(hipMemcpyAsync(c_dev, &n, 4, hipMemcpyHostToDevice, stream));
                uint c = n;   // last cluster count the host has seen (upper bound of the live count)
                int rounds = 0;
                const bool tail = !getenv("TRHIP_PLOC_NO_TAIL");
                while (c > 1) {
                    if (tail && c <= PLOC_TAIL) {   // the rest in one workgroup; its round count stays on the device (cbounds[10])
                        hipLaunchKernelGGL(k_ploc_tail, dim3(1), dim3(PLOC_TAIL), 0, stream, c_dev + (rounds & 1), n, (uint)ds.ploc_radius, cref[0], cbox[0], alloc,
                                           children, node_box, ranges, parent_internal, cbounds + 10);
                        break;
                    }
                    // rounds between host checks: few while the grids are large (an over-sized grid costs), many once they are small
                    const int batch = c > (1u << 16) ? 2 : (c > 4096 ? 4 : 8);
                    const uint cb = (c + BT - 1) / BT;
                    for (int r = 0; r < batch; ++r) {
                        const uint* c_cur = c_dev + ((rounds + r) & 1);
                        hipLaunchKernelGGL(k_ploc_nn, dim3(cb), dim3(BT), 0, stream, c_cur, (uint)ds.ploc_radius, cbox[0], nn);
                        hipLaunchKernelGGL(k_ploc_merge, dim3(cb), dim3(BT), 0, stream, c_cur, n, cref[0], cbox[0], nn, valid, cref[1], cbox[1], alloc, children,
                                           node_box, ranges, parent_internal);
                        HIPCHK(rocprim::exclusive_scan(base + o_scan, scan_bytes, valid, pos, 0u, cb * BT, rocprim::plus<uint>(), stream));
                        hipLaunchKernelGGL(k_ploc_compact, dim3(cb), dim3(BT), 0, stream, c_cur, c_dev + ((rounds + r + 1) & 1), valid, pos, cref[1], cbox[1], cref[0], cbox[0]);
                    }
                    rounds += batch;
                    uint c_new = 0;
                    HIPCHK(hipMemcpyAsync(&c_new, c_dev + (rounds & 1), 4, hipMemcpyDeviceToHost, stream));
                    HIPCHK(hipStreamSynchronize(stream));
                    if (c_new >= c || rounds > 4096) return set_error("PLOC: clustering did not converge");
                    c = c_new;
                }
                ds.build_rounds = (uint)rounds;
            }
            if (optimise) {
                // static geometry, "prefer fast trace": reinsertion rounds on the binary tree (bvh_optimize.h)
                OptTree t{(uint)(n - 1), n, children, node_box, leaf_box, reinterpret_cast<int*>(base + o_uparent)};
                OptMove* moves = reinterpret_cast<OptMove*>(base + o_moves);
                unsigned long long* lock = reinterpret_cast<unsigned long long*>(base + o_lock);
                double* cost = reinterpret_cast<double*>(base + o_optstat);
                uint* applied = reinterpret_cast<uint*>(base + o_optstat + 16);
                const uint ablocks = (uint)((n_all + BT - 1) / BT);
                const bool debug = getenv("TRHIP_DEBUG") != nullptr;
                auto report = [&](int round) -> int {
                    if (!debug) return 0;
                    double h[3] = {0, 0, 0};
                    HIPCHK(hipMemsetAsync(cost, 0, 8, stream));
                    hipLaunchKernelGGL(k_opt_cost, dim3(iblocks), dim3(BT), 0, stream, t, cost);
                    HIPCHK(hipMemcpyAsync(h, cost, 24, hipMemcpyDeviceToHost, stream));
                    HIPCHK(hipStreamSynchronize(stream));
                    uint moved[2]; memcpy(moved, &h[2], 8);
                    fprintf(stderr, "[trhip] tree optimisation round %d: inner area sum %.6g, %u of %u moves applied\n", round, h[0], round ? moved[0] : 0u, round ? moved[1] : 0u);
                    if (round) {   // every box against its children, every leaf count, every parent link
                        uint bad = 0;
                        HIPCHK(hipMemsetAsync(applied, 0, 4, stream));
                        hipLaunchKernelGGL(k_opt_check, dim3(iblocks), dim3(BT), 0, stream, t, ranges, applied);
                        HIPCHK(hipMemcpyAsync(&bad, applied, 4, hipMemcpyDeviceToHost, stream));
                        HIPCHK(hipStreamSynchronize(stream));
                        if (bad) return set_error("tree optimisation: " + std::to_string(bad) + " inconsistent nodes after round " + std::to_string(round));
                    }
                    return 0;
                };
                hipLaunchKernelGGL(k_opt_parents, dim3(iblocks), dim3(BT), 0, stream, t);
                if (int rc = report(0)) return rc;
                for (int round = 0; round < ds.optimise_rounds; ++round) {
                    HIPCHK(hipMemsetAsync(lock, 0, n_all * 8, stream));
                    HIPCHK(hipMemsetAsync(applied, 0, 8, stream));
                    HIPCHK(hipMemsetAsync(arrive, 0, (size_t)(n - 1) * 4, stream));
                    hipLaunchKernelGGL(k_opt_search, dim3(ablocks), dim3(BT), 0, stream, t, moves, (uint)round % (uint)ds.optimise_modulus, (uint)ds.optimise_modulus);
                    hipLaunchKernelGGL(k_opt_lock, dim3(ablocks), dim3(BT), 0, stream, t, moves, lock, debug ? applied + 1 : nullptr);
                    hipLaunchKernelGGL(k_opt_verify, dim3(ablocks), dim3(BT), 0, stream, t, moves, lock);
                    hipLaunchKernelGGL(k_opt_apply, dim3(ablocks), dim3(BT), 0, stream, t, moves, applied);
                    hipLaunchKernelGGL(k_opt_refit, dim3(blocks), dim3(BT), 0, stream, t, arrive, ranges);
                    if (int rc = report(round + 1)) return rc;
                }
                parent_internal = t.parent;     // the first n - 1 entries are the inner nodes' parents
            }
            {
                int* new_id = reinterpret_cast<int*>(base + o_new_id);
                if (ds.dfs_layout) hipLaunchKernelGGL(k_dfs_order, dim3(iblocks), dim3(BT), 0, stream, n - 1, children, ranges, parent_internal, new_id);
                else hipLaunchKernelGGL(k_identity, dim3(iblocks), dim3(BT), 0, stream, n - 1, new_id);
                const uint8_t* dec = nullptr;
                if (dp_collapse) {
                    OptTree t{(uint)(n - 1), n, children, node_box, leaf_box, reinterpret_cast<int*>(base + o_uparent)};
                    if (!optimise) hipLaunchKernelGGL(k_opt_parents, dim3(iblocks), dim3(BT), 0, stream, t);
                    HIPCHK(hipMemsetAsync(arrive, 0, (size_t)(n - 1) * 4, stream));
                    hipLaunchKernelGGL(k_collapse_cost, dim3(blocks), dim3(BT), 0, stream, t, arrive, reinterpret_cast<float*>(base + o_ccost), reinterpret_cast<uint8_t*>(base + o_cdec));
                    dec = reinterpret_cast<const uint8_t*>(base + o_cdec);
                }
                hipLaunchKernelGGL(k_collapse4, dim3(iblocks), dim3(BT), 0, stream, n - 1, children, node_box, leaf_box, new_id, dec, ds.nodes4);
            }
        }
        HIPCHK(hipGetLastError());
        if (getenv("TRHIP_PAIR_LEAVES") && atoi(getenv("TRHIP_PAIR_LEAVES")) != 0 && n > 2) {
#if TR_PAIR_LEAVES
            if (int rc = pair_leaves_postpass(ds, stream, n - 1, n)) return rc;
#else
            return set_error("TRHIP_PAIR_LEAVES needs a library built with -DTR_PAIR_LEAVES=1 (the traversal has to know the pair references)");
#endif
        }
        if (const char* e = getenv("TRHIP_TREELET")) {
            if (n > 2 && atoi(e) > 0) if (int rc = reorder_into_treelets(ds, stream, n - 1, n, (uint)atoi(e), !getenv("TRHIP_TREELET_KEEP_TRIS"))) return rc;
        }
    }
    ds.leaf_count = n;
    ds.node_count = n > 1 ? n - 1 : 0;
    ds.accel_tri_count = n_tri;
    ds.accel_built = true;
    // tri lights
    ds.tri_light_count = 0;
    if (ds.gather_emissive_triangles && ds.host_tri_light_count > 0) {
        if (!ds.tri_lights) HIPCHK(hipMalloc(&ds.tri_lights, (size_t)ds.host_tri_light_count * sizeof(TriLight)));
        HIPCHK(hipMemsetAsync(ds.tri_lights, 0, (size_t)ds.host_tri_light_count * sizeof(TriLight), stream));
        ds.tri_light_count = ds.host_tri_light_count;
        SceneView sv2 = ds.view();
        hipLaunchKernelGGL(k_extract_tri_lights, dim3((n_tri + BT - 1) / BT), dim3(BT), 0, stream, sv2, ds.tri_prefix, ds.tri_lights);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(e1, stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    uint hb[6];
    HIPCHK(hipMemcpy(hb, cbounds, sizeof(hb), hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; ++k) { ds.bounds_lo[k] = float_unflip(hb[k]); ds.bounds_hi[k] = float_unflip(hb[3 + k]); }
    if (info) {
        info->triangle_count = n_tri;
        info->leaf_count = n;
        info->node_count = ds.node_count;
        info->node_bytes = 112u;
        info->tri_light_count = ds.tri_light_count;
        info->build_ms = ms;
        for (int k = 0; k < 3; ++k) { info->bounds_min[k] = float_unflip(hb[k]); info->bounds_max[k] = float_unflip(hb[3 + k]); }
    }
    return 0;
}

}  // namespace tr
