// Tree optimisation by parallel reinsertion (Meister & Bittner, "Parallel Reinsertion for Bounding Volume Hierarchy
// Optimization", 2018) on the binary tree the PLOC / LBVH rounds leave behind, before it is relabelled and collapsed: what the
// reference asks its driver for with ePreferFastTrace on static meshes (src/acceleration_structure.cc:129-133).
//
// One round: every node x looks for the place in the tree where hanging it - together with its parent p, which is spliced out
// and re-used as the new inner node - lowers the sum of the inner nodes' surface areas most (branch-and-bound walk: up from p,
// and at every pivot on the way down into the subtree on the other side); every candidate move stamps its gain on the nodes it
// needs untouched (64-bit atomicMax of gain | node id: the result does not depend on thread order), the moves that own all their
// stamps are applied, and the boxes are refitted bottom-up.  Included by bvh_build.hip only.
#pragma once

namespace tr {
namespace {

// Nodes carry one id here: inner node i -> i, leaf l -> n_inner + l.  `children` keeps the builder's references (>= 0 inner
// node, < 0 leaf ~l); `parent[id]` = inner node or -1 for the root (node 0).
struct OptTree {
    uint n_inner, n_leaf;
    int2* children;
    float* node_box;          // 6 per inner node
    const float* leaf_box;    // 6 per leaf
    int* parent;              // n_inner + n_leaf
};
TR_DEV int opt_id(int ref, uint n_inner) { return ref >= 0 ? ref : (int)n_inner + ~ref; }
TR_DEV int opt_ref(int id, uint n_inner) { return id < (int)n_inner ? id : ~(id - (int)n_inner); }
TR_DEV const float* opt_box(const OptTree& t, int id) { return id < (int)t.n_inner ? t.node_box + 6 * (size_t)id : t.leaf_box + 6 * (size_t)(id - (int)t.n_inner); }
TR_DEV float opt_area(const float* b) { const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2]; return dx * dy + dy * dz + dz * dx; }
TR_DEV int opt_sibling(const OptTree& t, int parent, int child) {
    const int2 ch = t.children[parent];
    const int a = opt_id(ch.x, t.n_inner);
    return a == child ? opt_id(ch.y, t.n_inner) : a;
}

__global__ __launch_bounds__(BT) void k_opt_parents(OptTree t) {
    const uint i = blockIdx.x * BT + threadIdx.x;
    if (i >= t.n_inner) return;
    const int2 ch = t.children[i];
    t.parent[opt_id(ch.x, t.n_inner)] = (int)i;
    t.parent[opt_id(ch.y, t.n_inner)] = (int)i;
    if (i == 0) t.parent[0] = -1;
}

#define OPT_STACK 96
#define OPT_MAX_STEPS 8192     // node visits one search may spend

struct OptMove { int target, pivot; float gain; };   // hang x next to `target`; `pivot` = lowest common ancestor of the old and the new place

// The best new place of every node.  `phase` / `modulus`: only nodes with id % modulus == phase search in this round.
__global__ __launch_bounds__(BT) void k_opt_search(OptTree t, OptMove* moves, uint phase, uint modulus) {
    const uint x = blockIdx.x * BT + threadIdx.x;
    const uint n_all = t.n_inner + t.n_leaf;
    if (x >= n_all) return;
    OptMove mv = {-1, -1, 0.0f};
    const int p = t.parent[x];
    if (p > 0 && x % modulus == phase) {     // the root and its children stay where they are
        float bx[6];
        { const float* b = opt_box(t, (int)x); for (int k = 0; k < 6; ++k) bx[k] = b[k]; }
        const float ax = opt_area(bx);
        float d_path = opt_area(t.node_box + 6 * (size_t)p);     // what the path below the pivot saves once x and p are gone
        float bw[6] = {__builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf()};
        int stack_node[OPT_STACK]; float stack_ind[OPT_STACK];
        int steps = 0;
        int prev = (int)x;
        for (int pivot = p; pivot >= 0 && steps < OPT_MAX_STEPS; prev = pivot, pivot = t.parent[pivot]) {
            const int sib = opt_sibling(t, pivot, prev);
            int sp = 0;
            stack_node[sp] = sib; stack_ind[sp] = 0.0f; sp++;
            while (sp > 0 && steps < OPT_MAX_STEPS) {
                --sp; ++steps;
                const int y = stack_node[sp]; const float ind = stack_ind[sp];
                const float* by = opt_box(t, y);
                float u[6];
                for (int k = 0; k < 3; ++k) { u[k] = fminf(by[k], bx[k]); u[3 + k] = fmaxf(by[3 + k], bx[3 + k]); }
                const float direct = opt_area(u);
                const float gain = d_path - ind - direct;
                if (gain > mv.gain) { mv.gain = gain; mv.target = y; mv.pivot = pivot; }
                if (y < (int)t.n_inner) {
                    const float ind2 = ind + direct - opt_area(by);       // what y's own growth costs every place below it
                    if (d_path - ind2 - ax > mv.gain && sp + 2 <= OPT_STACK) {
                        const int2 ch = t.children[y];
                        stack_node[sp] = opt_id(ch.x, t.n_inner); stack_ind[sp] = ind2; sp++;
                        stack_node[sp] = opt_id(ch.y, t.n_inner); stack_ind[sp] = ind2; sp++;
                    }
                }
            }
            const float* bs = opt_box(t, sib);
            for (int k = 0; k < 3; ++k) { bw[k] = fminf(bw[k], bs[k]); bw[3 + k] = fmaxf(bw[3 + k], bs[3 + k]); }
            if (pivot != p) d_path += opt_area(t.node_box + 6 * (size_t)pivot) - opt_area(bw);   // the pivot shrinks to what is left below it
        }
        // a move must pay for itself beyond rounding noise
        if (!(mv.gain > 1e-6f * ax)) mv.target = -1;
    }
    moves[x] = mv;
}

TR_DEV unsigned long long opt_key(float gain, uint x) { return ((unsigned long long)__float_as_uint(gain) << 32) | (unsigned long long)x; }

// The nodes a move needs untouched by any other move of the round: x, its sibling and p (rewired), and the target with its
// ancestors below the pivot.  The last is what keeps the tree a tree: two moves whose targets lie in each other's moved
// subtrees would close a cycle, and then one's x or p is an ancestor of the other's target below its pivot - a stamp both
// want.  The old ancestors of p are not protected: a concurrent move there only makes the gain an estimate (the boxes are
// refitted afterwards anyway).  f(id) is called for each node of the set.
template <typename F>
TR_DEV void opt_for_lock_set(const OptTree& t, uint x, const OptMove& mv, F f) {
    const int p = t.parent[x];
    f((int)x); f(opt_sibling(t, p, (int)x)); f(p);
    for (int a = mv.target; a >= 0 && a != mv.pivot; a = t.parent[a]) f(a);
}

__global__ __launch_bounds__(BT) void k_opt_lock(OptTree t, const OptMove* moves, unsigned long long* lock, uint* candidates) {
    const uint x = blockIdx.x * BT + threadIdx.x;
    if (x >= t.n_inner + t.n_leaf) return;
    const OptMove mv = moves[x];
    if (mv.target < 0) return;
    if (candidates) atomicAdd(candidates, 1u);
    const unsigned long long key = opt_key(mv.gain, x);
    opt_for_lock_set(t, x, mv, [&](int id) { atomicMax(&lock[id], key); });
}

// Drops the moves that do not own every stamp of their lock set (the tree is only read here).
__global__ __launch_bounds__(BT) void k_opt_verify(OptTree t, OptMove* moves, const unsigned long long* lock) {
    const uint x = blockIdx.x * BT + threadIdx.x;
    if (x >= t.n_inner + t.n_leaf) return;
    const OptMove mv = moves[x];
    if (mv.target < 0) return;
    const unsigned long long key = opt_key(mv.gain, x);
    bool mine = true;
    opt_for_lock_set(t, x, mv, [&](int id) { mine = mine && lock[id] == key; });
    if (!mine) moves[x].target = -1;
}

// Applies the surviving moves.  A move reads and writes only nodes it owns, plus two child slots it finds by ids it owns (the
// one holding p in p's parent, the one holding the target in the target's parent): two moves never touch the same word.
__global__ __launch_bounds__(BT) void k_opt_apply(OptTree t, const OptMove* moves, uint* applied) {
    const uint x = blockIdx.x * BT + threadIdx.x;
    if (x >= t.n_inner + t.n_leaf) return;
    const OptMove mv = moves[x];
    if (mv.target < 0) return;
    const int p = t.parent[x], y = mv.target;
    const int s = opt_sibling(t, p, (int)x);
    const int g = t.parent[p], q = t.parent[y];
    int* ch = reinterpret_cast<int*>(t.children);
    const int ref_p = opt_ref(p, t.n_inner), ref_s = opt_ref(s, t.n_inner), ref_y = opt_ref(y, t.n_inner), ref_x = opt_ref((int)x, t.n_inner);
    // s takes p's place
    if (ch[2 * g] == ref_p) ch[2 * g] = ref_s; else ch[2 * g + 1] = ref_s;
    t.parent[s] = g;
    // p takes y's place (q may be g or s by now: read after the write above)
    if (ch[2 * q] == ref_y) ch[2 * q] = ref_p; else ch[2 * q + 1] = ref_p;
    t.parent[p] = q;
    ch[2 * p] = ref_y; ch[2 * p + 1] = ref_x;
    t.parent[y] = p;
    atomicAdd(applied, 1u);
}

// What the bottom-up passes hand from thread to thread (boxes, leaf counts, cost tables) is written and read with agent-scope
// atomic stores and loads, which go past the caches that are not coherent between CUs and XCDs; the arrival counter is a
// relaxed agent-scope atomic.  Between the two a thread only has to wait until its own stores have completed - no cache
// write-back or invalidation, which is what made the release / acquire form of this loop (k_refit) cost 7 ms per million
// triangles instead of 1.  TRHIP_DEBUG checks every box of the finished tree against its children (k_opt_check).
TR_DEV void opt_publish() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }

// Bottom-up boxes and leaf counts of the whole tree (the second thread to reach a node owns it, as k_refit).
__global__ __launch_bounds__(BT) void k_opt_refit(OptTree t, uint* arrive, uint* subtree_size) {
    const uint leaf = blockIdx.x * BT + threadIdx.x;
    if (leaf >= t.n_leaf) return;
    int node = t.parent[t.n_inner + leaf];
    while (node >= 0) {
        opt_publish();
        const uint prev = __hip_atomic_fetch_add(&arrive[node], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == 0) return;
        const int2 ch = t.children[node];
        const float* b0 = ch.x >= 0 ? t.node_box + 6 * (size_t)ch.x : t.leaf_box + 6 * (size_t)(~ch.x);
        const float* b1 = ch.y >= 0 ? t.node_box + 6 * (size_t)ch.y : t.leaf_box + 6 * (size_t)(~ch.y);
        for (int k = 0; k < 3; ++k) {
            const float l0 = __hip_atomic_load(&b0[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), h0 = __hip_atomic_load(&b0[3 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float l1 = __hip_atomic_load(&b1[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), h1 = __hip_atomic_load(&b1[3 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&t.node_box[6 * (size_t)node + k], fminf(l0, l1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&t.node_box[6 * (size_t)node + 3 + k], fmaxf(h0, h1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const uint n0 = ch.x >= 0 ? __hip_atomic_load(&subtree_size[ch.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
        const uint n1 = ch.y >= 0 ? __hip_atomic_load(&subtree_size[ch.y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
        __hip_atomic_store(&subtree_size[node], n0 + n1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        node = t.parent[node];
    }
}


// Debug: inner nodes whose box is not exactly the union of their children's, or whose leaf count is not the children's sum.
__global__ __launch_bounds__(BT) void k_opt_check(OptTree t, const uint* subtree_size, uint* bad) {
    const uint i = blockIdx.x * BT + threadIdx.x;
    if (i >= t.n_inner) return;
    const int2 ch = t.children[i];
    const float* b0 = ch.x >= 0 ? t.node_box + 6 * (size_t)ch.x : t.leaf_box + 6 * (size_t)(~ch.x);
    const float* b1 = ch.y >= 0 ? t.node_box + 6 * (size_t)ch.y : t.leaf_box + 6 * (size_t)(~ch.y);
    bool ok = t.parent[opt_id(ch.x, t.n_inner)] == (int)i && t.parent[opt_id(ch.y, t.n_inner)] == (int)i;
    for (int k = 0; k < 3; ++k) ok = ok && t.node_box[6 * (size_t)i + k] == fminf(b0[k], b1[k]) && t.node_box[6 * (size_t)i + 3 + k] == fmaxf(b0[3 + k], b1[3 + k]);
    ok = ok && subtree_size[i] == (ch.x >= 0 ? subtree_size[ch.x] : 1u) + (ch.y >= 0 ? subtree_size[ch.y] : 1u);
    if (i == 0) ok = ok && subtree_size[0] == t.n_leaf && t.parent[0] == -1;
    if (!ok) atomicAdd(bad, 1u);
}

// sum of the inner nodes' areas (the part of the SAH cost a reinsertion can change), for TRHIP_DEBUG
__global__ __launch_bounds__(BT) void k_opt_cost(OptTree t, double* sum) {
    const uint i = blockIdx.x * BT + threadIdx.x;
    double a = i < t.n_inner ? (double)opt_area(t.node_box + 6 * (size_t)i) : 0.0;
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum, a);
}


// ---- which binary nodes become 4-wide nodes (after Ylitie, Karras & Laine 2017, section 3.1, for four-wide nodes over
// single-triangle leaves).  cost(n, i), i = 1..3: the least sum of 4-wide node areas with which the subtree of n can hang in i
// slots of a node above it: in one slot n is a 4-wide node itself (its area + the best way to deal its four slots to its two
// children), in more slots it may instead be opened and its slots dealt to the children.  Leaves cost nothing in any number of
// slots.  Bottom-up like k_opt_refit; `dec[n]` = bits 0-1: slots of the left child when n is a node (1..3), bit 2: in two
// slots n is opened, bits 3-4: in three slots n is opened with that many slots for the left child (0 = n stays as in two).
__global__ __launch_bounds__(BT) void k_collapse_cost(OptTree t, uint* arrive, float* cost /*3 per inner node*/, uint8_t* dec) {
    const uint leaf = blockIdx.x * BT + threadIdx.x;
    if (leaf >= t.n_leaf) return;
    int node = t.parent[t.n_inner + leaf];
    while (node >= 0) {
        opt_publish();
        const uint prev = __hip_atomic_fetch_add(&arrive[node], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == 0) return;
        const int2 ch = t.children[node];
        float cl[3] = {0, 0, 0}, cr[3] = {0, 0, 0};
        if (ch.x >= 0) for (int k = 0; k < 3; ++k) cl[k] = __hip_atomic_load(&cost[3 * (size_t)ch.x + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ch.y >= 0) for (int k = 0; k < 3; ++k) cr[k] = __hip_atomic_load(&cost[3 * (size_t)ch.y + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float d4 = cl[0] + cr[2]; uint m4 = 1;
        if (cl[1] + cr[1] < d4) { d4 = cl[1] + cr[1]; m4 = 2; }
        if (cl[2] + cr[0] < d4) { d4 = cl[2] + cr[0]; m4 = 3; }
        const float c1 = opt_area(t.node_box + 6 * (size_t)node) + d4;
        const float d2 = cl[0] + cr[0];
        const uint open2 = d2 < c1 ? 1u : 0u;
        const float c2 = open2 ? d2 : c1;
        float d3 = cl[0] + cr[1]; uint m3 = 1;
        if (cl[1] + cr[0] < d3) { d3 = cl[1] + cr[0]; m3 = 2; }
        const uint open3 = d3 < c2 ? m3 : 0u;
        const float c3 = open3 ? d3 : c2;
        __hip_atomic_store(&cost[3 * (size_t)node], c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&cost[3 * (size_t)node + 1], c2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&cost[3 * (size_t)node + 2], c3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dec[node] = (uint8_t)(m4 | (open2 << 2) | (open3 << 3));
        node = t.parent[node];
    }
}

// The references that fill the four slots of binary node n as a 4-wide node, in order; returns how many.
TR_DEV int collapse_children(const int2* children, const uint8_t* dec, int n, int out[4]) {
    int stack_node[8], stack_slots[8];
    int sp = 0, count = 0;
    const int2 ch = children[n];
    const int m4 = dec[n] & 3;
    stack_node[sp] = ch.y; stack_slots[sp] = 4 - m4; sp++;
    stack_node[sp] = ch.x; stack_slots[sp] = m4; sp++;
    while (sp > 0) {
        --sp;
        const int node = stack_node[sp];
        int slots = stack_slots[sp];
        if (node < 0) { out[count++] = node; continue; }
        const uint d = dec[node];
        int left = 0;                                   // slots of the left child once the node is opened; 0 = it stays a node
        if (slots == 3) { left = (int)((d >> 3) & 3u); if (left == 0) slots = 2; }
        if (slots == 2 && left == 0) left = (int)((d >> 2) & 1u);
        if (left == 0) { out[count++] = node; continue; }
        const int2 c = children[node];
        stack_node[sp] = c.y; stack_slots[sp] = slots - left; sp++;
        stack_node[sp] = c.x; stack_slots[sp] = left; sp++;
    }
    return count;
}

}  // namespace
}  // namespace tr
